#!/usr/bin/env python3
"""Same-box A/B of kernel-library builds (box-to-box variation is +-3 %, so variants are only ever compared inside one
gpurun call): runs `cmd` once per library per round, alternating, each in its own process with SGL_MI355_LIB set.

    python -m sglang_npu_amd.build_ext --variant nt --flag=-DSGLM_KV_DMA_NT=1
    python tools/ab_variants.py --libs default,nt --rounds 3 -- python tools/bench_decode.py --quick
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", required=True, help="comma-separated variant names; 'default' = lib/libsgl_mi355.so")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    for r in range(a.rounds):
        for name in a.libs.split(","):
            env = dict(os.environ)
            if name != "default":
                env["SGL_MI355_LIB"] = os.path.join(ROOT, "sglang_npu_amd", "lib", "variants", f"libsgl_mi355_{name}.so")
            out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True)
            for line in (out.stdout + out.stderr).splitlines():
                if line.strip() and "amdgpu.ids" not in line:
                    print(f"[round {r}] [{name}] {line}", flush=True)


if __name__ == "__main__":
    main()
