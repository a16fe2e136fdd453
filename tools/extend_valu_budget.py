#!/usr/bin/env python3
"""Per-tile vector-instruction budget of the extend (prefill) attention kernel, read from the ISA.

    python tools/extend_valu_budget.py [path/to/attention_extend.hip]   (default: the tree's; needs hipcc, no GPU)

Compiles ONE instantiation -- extend_mfma_kernel<bf16, D = 128, int32 indices, GH = 4, no mask, 16-bit pool, no key split>,
the kernel of a long single-request prefill -- to gfx950 assembly and walks the extend-stage tile loop along the path an
UNMASKED tile takes (every tile of a query block except the one or two on the diagonal): the QK^T block, the row-maximum
chain, the cross-half exchange, the exp / sum / pack block with the PV MFMAs, the DMA issue of the next tile.  Instructions
are binned by what they do for the softmax.  The O rescale (32 v_pk_mul_f32) sits behind a wave-uniform branch that is
taken only while some row's running maximum still moves (the first tiles of a block) and is listed apart."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "sglang_npu_amd", "csrc", "attention_extend.hip")
KERNEL = "_ZN4sglm12_GLOBAL__N_118extend_mfma_kernelILi0ELi128EiLi4ELb0ELb0ELi1ELi2ELb0EEEvNS0_10ExtendArgsE"


def compile_one(tmp):
    text = open(SRC).read()
    cut = text.index("// what the caller of the _parts entry point knows")
    one = os.path.join(tmp, "ext_one.hip")
    with open(one, "w") as f:
        f.write(text[:cut])
        f.write("template __global__ void extend_mfma_kernel<0, 128, int, 4, false, false, 1, 2, false>(ExtendArgs);\n"
                "}  // namespace\n}  // namespace sglm\n")
    out = os.path.join(tmp, "ext_one.s")
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "-std=c++17", "--offload-arch=gfx950",
                    "--cuda-device-only", "-S", "-ffp-contract=fast-honor-pragmas", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.dirname(SRC), "-I", os.path.join(ROOT, "sglang_npu_amd", "csrc"), one, "-o", out],
                   check=True, capture_output=True, text=True)
    return open(out).read()


def blocks_of(asm):
    m = re.search(r"^" + re.escape(KERNEL) + r":[^\n]*\n(.*?)^\.Lfunc_end\d+:", asm, flags=re.S | re.M)
    tail = asm[m.end():m.end() + 8000]
    meta = {k: re.search(r"^;\s*" + k + r":\s*(\d+)", tail, flags=re.M).group(1) for k in ("NumVgprs", "ScratchSize", "Occupancy")}
    out, cur = [], None
    for line in m.group(1).split("\n"):
        l = line.strip()
        if re.match(r"^\.LBB\d+_\d+:", l):
            cur = [l.split(":")[0], []]
            out.append(cur)
        elif cur is not None and l and not l.startswith((";", ".")):
            cur[1].append(l.split()[0])
    return out, meta


BINS = [("score scale (s * sm_scale * log2 e)", lambda op, blk: op == "v_pk_mul_f32" and blk == "qk"),
        ("row maximum (v_max3 chain, running max)", lambda op, blk: op.startswith(("v_max3", "v_max_f32"))),
        ("cross-half exchange of the row maximum", lambda op, blk: op.startswith(("v_permlane", "ds_bpermute")) or
         (blk == "mid" and op.startswith(("v_and_b32", "v_xor_b32", "v_lshlrev_b32", "v_cmp_lt_i32", "v_mov_b32")))),
        ("exp2 argument (s - m)", lambda op, blk: op.startswith(("v_sub_f32", "v_fma_f32"))),
        ("exp2 (transcendental, 8-cycle issue)", lambda op, blk: op.startswith("v_exp_f32")),
        ("row sum (32 adds + l = l * alpha + sum)", lambda op, blk: op.startswith(("v_add_f32", "v_fmac_f32"))),
        ("P -> 16 bits (v_cvt_pk)", lambda op, blk: op.startswith("v_cvt_pk")),
        ("LDS address arithmetic", lambda op, blk: op.startswith(("v_add_u32", "v_add3_u32", "v_lshl_add", "v_or_b32"))),
        ("guards (m == -inf, alpha != 1)", lambda op, blk: op.startswith(("v_cmp", "v_cndmask"))),
        ("other VALU", lambda op, blk: op.startswith("v_") and not op.startswith("v_mfma"))]


def main():
    with tempfile.TemporaryDirectory() as tmp:
        blocks, meta = blocks_of(compile_one(tmp))
    cnt = lambda ops, p: sum(1 for o in ops if o.startswith(p))  # noqa: E731
    # the extend stage is the LAST tile loop of the kernel: its QK^T blocks are the last ones with 16 MFMAs + 16 ds_read_b128
    qk = [i for i, (_, ops) in enumerate(blocks) if cnt(ops, "v_mfma") == 16 and cnt(ops, "ds_read_b128") == 16]
    pv = [i for i, (_, ops) in enumerate(blocks) if cnt(ops, "v_mfma") == 16 and cnt(ops, "ds_read_b64_tr") == 32]
    start = qk[-1]
    end = min(i for i in pv if i > start)
    path, rescale = [], 0
    for i in range(start, end + 1):
        name, ops = blocks[i]
        masked = cnt(ops, "v_cndmask") > 8 or cnt(ops, "v_cmp") > 8   # the per-element mask of a diagonal tile: not this path
        masked = masked or cnt(ops, "v_div_scale") > 0 or cnt(ops, "v_rndne") > 0 or cnt(ops, "v_fmaak") > 0  # (tanh of a logit cap)
        masked = masked or (len(ops) <= 4 and i not in (start, end))   # (the jump pads between the tanh blocks of the r4 source)
        if masked:
            continue
        if cnt(ops, "v_pk_mul_f32") >= 32 and i != start:              # the O rescale behind its wave-uniform branch
            rescale += cnt(ops, "v_pk_mul_f32")
            continue
        kind = "qk" if i == start else ("pv" if i == end else "mid")
        if i == start and cnt(ops, "v_pk_mul_f32") >= 32:               # (an if-converted rescale inside the QK block)
            rescale += 32
        path.append((name, kind, ops))
    # the psum adds / DMA issue may sit in the block(s) right after the PV block
    for i in range(end + 1, min(end + 4, len(blocks))):
        name, ops = blocks[i]
        if cnt(ops, "v_add_f32") >= 16 or cnt(ops, "global_load_lds") >= 4:
            path.append((name, "tail", ops))
    bins = collections.OrderedDict((b[0], 0) for b in BINS)
    other = collections.Counter()
    tot = collections.Counter()
    for name, kind, ops in path:
        for op in ops:
            if op.startswith("v_mfma"):
                tot["MFMA"] += 1
            elif op.startswith("v_"):
                tot["VALU"] += 1
                for label, test in BINS:
                    if test(op, kind):
                        bins[label] += 1
                        if label == "other VALU":
                            other[op] += 1
                        break
            elif op.startswith("ds_"):
                tot["LDS"] += 1
            elif op.startswith(("global_", "buffer_")):
                tot["VMEM (LDS-DMA)"] += 1
            elif op.startswith(("s_waitcnt", "s_nop", "s_barrier")):
                tot[op] += 1
            elif op.startswith("s_"):
                tot["SALU"] += 1
    print(f"# {os.path.relpath(SRC, ROOT)}: extend_mfma_kernel<bf16, 128, int, GH=4, plain>  NumVgprs {meta['NumVgprs']}  "
          f"scratch {meta['ScratchSize']} B  occupancy {meta['Occupancy']} waves/SIMD")
    print(f"# blocks on the unmasked-tile path: {', '.join(n for n, _, _ in path)}")
    mf = tot["MFMA"]
    print(f"per tile and wave (64 keys x 32 query rows x 1 head): {mf} MFMA (32x32x16), {tot['VALU']} other vector instructions "
          f"= {tot['VALU'] / mf:.2f} per MFMA")
    for label, n in bins.items():
        if n:
            print(f"    {n:4d}  {label}")
    if other:
        print("          (other: " + ", ".join(f"{k} {v}" for k, v in other.most_common()) + ")")
    print(f"    + {rescale} v_pk_mul_f32 of the O rescale on tiles where a row maximum moved (wave-uniform branch)")
    print("  beside them: " + ", ".join(f"{k} {v}" for k, v in tot.items() if k not in ("MFMA", "VALU")))
    issue = (tot["VALU"] - bins["exp2 (transcendental, 8-cycle issue)"]) * 4 + bins["exp2 (transcendental, 8-cycle issue)"] * 8 + mf * 8
    print(f"  vector issue cycles per tile and wave at the guide's prices (4 per VALU, 8 per v_exp_f32, 8 per MFMA): {issue}; "
          f"MFMA pipe {mf * 32}; two waves per SIMD: issue {2 * issue} vs pipe {2 * mf * 32} -> issue-bound ceiling "
          f"{min(1.0, mf * 32 / issue):.2f} of the MFMA peak")


if __name__ == "__main__":
    main()
