import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "optin_fusions: needs a library built with -DSGLM_OPTIN_FUSIONS=1 (variant build); "
                                       "skipped on the default library, whose entry points for them return UNSUPPORTED")


def pytest_collection_modifyitems(config, items):
    if any("optin_fusions" in item.keywords for item in items):
        try:
            from sglang_npu_amd import ops
            have = ops.has_optin_fusions()
        except Exception:  # noqa: BLE001  (no library here: the gpu marker below decides)
            have = False
        if not have:
            skip_f = pytest.mark.skip(reason="opt-in fusion kernels are not in this build (-DSGLM_OPTIN_FUSIONS=1)")
            for item in items:
                if "optin_fusions" in item.keywords:
                    item.add_marker(skip_f)
    # `-m gpu` tests must never silently pass without a GPU.
    if torch.cuda.is_available():
        return
    markexpr = config.getoption("-m") or ""
    if "gpu" in markexpr and "not gpu" not in markexpr:
        pytest.exit("-m gpu requested but no GPU is visible", returncode=3)
    skip = pytest.mark.skip(reason="needs a GPU")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def _h16(arr: np.ndarray, dtype: str) -> torch.Tensor:
    t = torch.from_numpy(arr.view(np.int16).copy())
    return t.view(torch.bfloat16 if dtype == "bf16" else torch.float16)


def load_golden(name: str) -> dict:
    """Load tests/golden/<name>.npz; 16-bit float arrays (stored as uint16) become torch tensors."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    dtype = z["dtype"].item().decode() if "dtype" in z.files else None
    for k in z.files:
        a = z[k]
        if a.dtype == np.uint16 and dtype is not None:
            out[k] = _h16(a, dtype)
        elif a.dtype.kind in "SU":
            out[k] = a.item().decode() if a.dtype.kind == "S" else str(a)
        elif a.ndim == 0:
            out[k] = a.item()
        else:
            out[k] = torch.from_numpy(a.copy())
    return out


def golden_names(prefix: str):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


def tol_for(dtype, ref_f32: torch.Tensor):
    """north_star: 1e-3 on 16-bit outputs.  An output element of magnitude |x| carries half an
    ulp of rounding on each side (2^-8 |x| bf16, 2^-11 |x| fp16), so the bound is
    atol = 1e-3 + ulp(dtype) * max|ref|."""
    ulp = 2.0 ** -8 if dtype in ("bf16", torch.bfloat16) else 2.0 ** -11
    return 1e-3 + ulp * float(ref_f32.abs().max())


def tol_pair(dtype, ref: torch.Tensor):
    """Bound for comparing TWO 16-bit results (HIP vs the oracle's 16-bit output): each carries half an
    ulp of output rounding, so they may sit one full ulp apart: atol = 1e-3 + 2 * ulp/2 * max|ref|."""
    ulp = 2.0 ** -7 if dtype in ("bf16", torch.bfloat16) else 2.0 ** -10
    return 1e-3 + ulp * float(ref.float().abs().max())
