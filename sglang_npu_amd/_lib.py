"""ctypes binding to the C-ABI kernel library ``lib/libsgl_mi355.so`` (include/sgl_mi355.h).

There is no CPU or PyTorch fallback: if the HIP library is missing or a call fails the
caller gets an exception.
"""
from __future__ import annotations

import ctypes
import os
import threading

_PKG = os.path.dirname(os.path.abspath(__file__))
# SGL_MI355_LIB: load another build of the same library (same-box A/B of kernel variants, tools/ab_variants.py)
LIB_PATH = os.environ.get("SGL_MI355_LIB") or os.path.join(_PKG, "lib", "libsgl_mi355.so")
ABI_VERSION = 14

_lib = None
_lock = threading.Lock()


class Mi355LibraryError(RuntimeError):
    """The HIP kernel library could not be loaded."""


def lib() -> ctypes.CDLL:
    """Load (once) and return the kernel library.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise Mi355LibraryError(
                        f"{LIB_PATH} not found: build it with `python -m sglang_npu_amd.build_ext` "
                        "(hipcc, gfx950). There is no fallback path.")
                try:
                    l = ctypes.CDLL(LIB_PATH)
                except OSError as e:  # pragma: no cover - depends on the machine
                    raise Mi355LibraryError(f"cannot load {LIB_PATH}: {e}") from e
                l.sgl_mi355_abi_version.restype = ctypes.c_int
                l.sgl_mi355_last_error.restype = ctypes.c_size_t
                l.sgl_mi355_last_error.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
                v = l.sgl_mi355_abi_version()
                if v != ABI_VERSION:
                    raise Mi355LibraryError(f"ABI mismatch: library {v}, binding {ABI_VERSION}; rebuild")
                _lib = l
    return _lib


def last_error() -> str:
    buf = ctypes.create_string_buffer(512)
    lib().sgl_mi355_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(rc: int) -> None:
    """Map a C-ABI status to the exception the reference raises for the same condition:
    TORCH_CHECK -> RuntimeError, TORCH_CHECK_NOT_IMPLEMENTED -> NotImplementedError."""
    if rc == 0:
        return
    msg = last_error()
    if rc == 2:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)
