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


def _ulp(dtype) -> float:
    """Half-ulp rounding bound of a 16-bit output relative to its magnitude: |round(x) - x| <= 2^-8 |x| (bf16, 8 significand
    bits) or 2^-11 |x| (fp16, 11)."""
    return 2.0 ** -8 if dtype in ("bf16", torch.bfloat16) else 2.0 ** -11


def elem_bound(dtype, ref: torch.Tensor, pair: bool = False) -> torch.Tensor:
    """PER-ELEMENT parity bound: north_star's 1e-3 on 16-bit outputs plus the output's own rounding,
        |hip_i - ref_i| <= 1e-3 + ulp(dtype) * |ref_i|,
    ulp = 2^-8 (bf16) / 2^-11 (fp16) against an fp32 truth; `pair=True` compares TWO 16-bit results (the HIP kernel against
    the oracle's / the reference's 16-bit output): each carries its own half ulp, so the ulp term is doubled.  A small output
    next to a large one gets no slack from its neighbour (the old bound scaled everything by max|ref|)."""
    return 1e-3 + (2.0 if pair else 1.0) * _ulp(dtype) * ref.float().abs()


def p_rounding_term(dtype, attn_of_abs_v: torch.Tensor, pair: bool = False) -> torch.Tensor:
    """What rounding the softmax numerators to the 16-bit MFMA input type can move ONE output element by: every p_j carries
    a relative error of at most half an ulp (2^-9 bf16, 2^-12 fp16), so |sum_j p_j v_jd - sum_j round(p_j) v_jd| / sum_j p_j
    <= halfulp * sum_j p_j |v_jd| / sum_j p_j -- the same attention evaluated on |V| (`attn_of_abs_v`, from the oracle), per
    element.  The reference's own extend kernel rounds its probabilities the same way (extend.cpp: the s_delta tile is
    converted to the 16-bit type for the PV GEMM) and is itself 0.1-0.3 % of elements beyond 1e-3 + ulp |ref_i| of the fp32
    truth on its golden fixtures (parity bound); `pair=True`: both sides round independently."""
    half_ulp = 2.0 ** -9 if dtype in ("bf16", torch.bfloat16) else 2.0 ** -12
    return (2.0 if pair else 1.0) * half_ulp * attn_of_abs_v.detach().float().abs().cpu()


def count_beyond(got: torch.Tensor, ref: torch.Tensor, dtype, pair: bool = False) -> int:
    """How many elements lie beyond the strict per-element bound (no P-rounding term)."""
    got_f, ref_f = got.detach().float().cpu(), ref.detach().float().cpu()
    return int(((got_f - ref_f).abs() > elem_bound(dtype, ref_f, pair)).sum())


def assert_elem_close(got: torch.Tensor, ref: torch.Tensor, dtype, pair: bool = False, what: str = "", extra=None):
    """extra: an additional PER-ELEMENT allowance with a derivation of its own (p_rounding_term), never a scalar of the tensor."""
    got_f, ref_f = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got_f.shape == ref_f.shape, f"{what}: shape {tuple(got_f.shape)} vs {tuple(ref_f.shape)}"
    err = (got_f - ref_f).abs()
    bound = elem_bound(dtype, ref_f, pair)
    if extra is not None:
        bound = bound + extra.reshape(bound.shape)
    over = err - bound
    bad = over > 0
    if bool(bad.any()) or not bool(torch.isfinite(got_f).all()):
        i = int(torch.argmax(torch.where(torch.isfinite(over), over, torch.full_like(over, float("inf")))))
        raise AssertionError(
            f"{what}: {int(bad.sum())} of {bad.numel()} elements beyond 1e-3 + {'2 ' if pair else ''}ulp |ref_i|"
            f"{' + the P-rounding term' if extra is not None else ''}; worst at flat "
            f"index {i}: got {got_f.reshape(-1)[i].item():.6g}, ref {ref_f.reshape(-1)[i].item():.6g}, |err| {err.reshape(-1)[i].item():.3e}, "
            f"bound {bound.reshape(-1)[i].item():.3e}; max |err| {err.max().item():.3e}")


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
