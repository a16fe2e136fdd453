"""CPU: the C-ABI library loads, exports every symbol include/*.h declares, and rejects bad
arguments with the reference's error class (no GPU compute here)."""
import ctypes
import glob
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    syms = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        syms += re.findall(r"\b(sgl_mi355_[a-z0-9_]+)\s*\(", src)
    return sorted(set(syms))


def test_library_exports_every_declared_symbol():
    from sglang_npu_amd import _lib
    lib = _lib.lib()
    syms = _declared_symbols()
    assert len(syms) >= 6
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/ but not exported"


def test_abi_version():
    from sglang_npu_amd import _lib
    assert _lib.lib().sgl_mi355_abi_version() == _lib.ABI_VERSION


def test_invalid_arguments_raise_runtime_error():
    from sglang_npu_amd import _lib
    lib = _lib.lib()
    rc = lib.sgl_mi355_create_kv_indices(None, ctypes.c_int64(0), None, 0, None, 0, None, None, 0, None,
                                         ctypes.c_int64(70000), None)
    assert rc == 1
    with pytest.raises(RuntimeError, match="batch_size"):
        _lib.check(rc)
    # heads not divisible by kv heads
    z = ctypes.c_int64
    rc = lib.sgl_mi355_decode_attention_fwd(
        None, None, None, None, None, None, None, None, None, z(1), z(2), z(7), z(2), z(128), z(128),
        z(0), z(0), z(0), z(0), z(0), z(0), z(0), z(0), ctypes.c_float(1.0), ctypes.c_float(0.0), 0, None)
    assert rc == 1 and "multiple" in _lib.last_error()


def test_round3_entry_points_check_their_arguments_before_any_launch():
    """sgl_mi355_decode_attention_quant / sgl_mi355_fp8_scaled_mm_silu_mul_wshuffled: precondition failures are
    SGL_MI355_ERR_INVALID_ARGUMENT (1), shapes outside the fused form SGL_MI355_ERR_UNSUPPORTED (2) -- both decided on
    the host, nothing is launched (no GPU here)."""
    from sglang_npu_amd import _lib
    lib = _lib.lib()
    z, f, vp = ctypes.c_int64, ctypes.c_float, ctypes.c_void_p
    fake = vp(0x10000)  # aligned, never dereferenced: the shape is refused first
    rc = lib.sgl_mi355_decode_attention_quant(
        None, None, None, None, None, None, None, None, 0, None, None, z(64), z(4096), z(32), z(8), z(128), z(4096), z(128),
        z(1024), z(128), z(1024), z(128), z(4096), z(128), f(0.1), f(0.0), 0, None)
    if lib.sgl_mi355_has_optin_fusions():
        assert rc == 1 and "null" in _lib.last_error()
        rc = lib.sgl_mi355_decode_attention_quant(
            fake, fake, fake, fake, fake, fake, fake, fake, 0, fake, fake, z(8), z(4096), z(32), z(8), z(128), z(4096), z(128),
            z(1024), z(128), z(1024), z(128), z(4096), z(128), f(0.1), f(0.0), 0, None)
        assert rc == 2 and "pairs-of-items" in _lib.last_error()  # 8 requests x 8 kv heads = 64 items: not more than 256
    else:  # the default build: the opt-in fusions' entry points decline everything, without touching their arguments
        assert rc == 2 and "opt-in fusion" in _lib.last_error()
    rc = lib.sgl_mi355_fp8_scaled_mm_silu_mul_wshuffled(fake, fake, fake, fake, None, fake, z(1024), z(28672 + 16), z(4096), z(4096),
                                                        0, None)
    assert rc == 1 and "N % 32" in _lib.last_error()
    rc = lib.sgl_mi355_fp8_scaled_mm_silu_mul_wshuffled(fake, fake, fake, fake, None, fake, z(64), z(28672), z(4096), z(4096), 0, None)
    assert rc == 2 and "prefill sizes" in _lib.last_error()
    rc = lib.sgl_mi355_fp8_scaled_mm_silu_mul_wshuffled(fake, fake, fake, fake, None, fake, z(256), z(4096), z(4096), z(4096), 0, None)
    assert rc == 2  # 2 x 16 tiles
    rc = lib.sgl_mi355_fp8_scaled_mm_silu_mul_wshuffled(fake, fake, fake, fake, None, fake, z(0), z(4096), z(4096), z(4096), 0, None)
    assert rc == 0  # no rows: nothing to do


def test_ops_refuse_cpu_tensors():
    import torch
    from sglang_npu_amd import ops
    x = torch.zeros(2, 4, 8, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.set_kv_buffer(x, x, torch.zeros(2, dtype=torch.int64), x, x)


def test_product_does_not_import_oracle():
    """The shipped package must never route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "sglang_npu_amd")
    for path in glob.glob(os.path.join(pkg, "**", "*.py"), recursive=True) + \
            glob.glob(os.path.join(pkg, "csrc", "*")):
        src = open(path, errors="replace").read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), path
        assert "sgl_oracle" not in src, path
