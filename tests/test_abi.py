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
