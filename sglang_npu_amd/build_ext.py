"""Build the MI355X kernel library (hipcc, gfx950 only) in-tree.

    python -m sglang_npu_amd.build_ext [--force] [--jobs N]

Every ``csrc/*.hip`` is compiled to an object with ``hipcc --offload-arch=gfx950`` and the
objects are linked into ``sglang_npu_amd/lib/libsgl_mi355.so`` -- the C-ABI shared library
declared in ``include/sgl_mi355.h``.  No GPU is needed to build (hipcc cross-compiles).
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIB_DIR, "libsgl_mi355.so")
OBJ_DIR = os.path.join(PKG, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
            "-ffp-contract=fast-honor-pragmas", "-I", os.path.join(ROOT, "include")]


def _deps():
    return glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))


def _stale(out, srcs):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in srcs)


def _file_flags(src):
    """Per-file compiler flags: a `// hipcc-flags: ...` comment in the first lines of the source."""
    flags = []
    with open(src) as f:
        for _, line in zip(range(8), f):
            if line.startswith("// hipcc-flags:"):
                flags += line[len("// hipcc-flags:"):].split()
    return flags


def _compile(src, obj, extra):
    cmd = [HIPCC] + CXXFLAGS + _file_flags(src) + extra + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return src, r.returncode, r.stdout + r.stderr


def build(force: bool = False, jobs: int = 4, verbose: bool = True, extra=None, variant: str = "") -> str:
    """variant: build a second copy of the library with `extra` compiler flags (e.g. -DSGLM_KV_DMA_POLICY=1) into
    lib/variants/libsgl_mi355_<variant>.so for same-box A/B runs; the default build is untouched."""
    lib_path = os.path.join(LIB_DIR, "variants", f"libsgl_mi355_{variant}.so") if variant else LIB
    obj_dir = os.path.join(OBJ_DIR, "variant_" + variant) if variant else OBJ_DIR
    os.makedirs(os.path.dirname(lib_path), exist_ok=True)
    os.makedirs(obj_dir, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = _deps()
    todo, objs = [], []
    for s in srcs:
        o = os.path.join(obj_dir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + deps):
            todo.append((s, o))
    if todo:
        with cf.ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
            for src, rc, out in ex.map(lambda so: _compile(so[0], so[1], extra or []), todo):
                if verbose and out.strip():
                    print(out, file=sys.stderr)
                if rc != 0:
                    raise RuntimeError(f"hipcc failed on {src}:\n{out}")
                if verbose:
                    print(f"[build_ext] compiled {os.path.basename(src)}")
    if force or todo or _stale(lib_path, objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib_path] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
        if verbose:
            print(f"[build_ext] linked {lib_path}")
    return lib_path


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=4)
    ap.add_argument("--variant", default="", help="name of an A/B variant build (lib/variants/)")
    ap.add_argument("--flag", action="append", default=[], help="extra compiler flag of the variant, e.g. -DSGLM_X=1")
    a = ap.parse_args()
    print(build(force=a.force, jobs=a.jobs, extra=a.flag, variant=a.variant))
