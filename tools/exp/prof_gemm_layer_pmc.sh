# SQ counter passes of the four FP8 layer GEMMs at M = 1024 and M = 4096 (separate passes, kernel-trace only): VERDICT r4 item 6
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for M in 1024 4096; do
export M
rm -rf gpurun_out/pmc_gl
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_gl/sq -- python3 tools/prof_gemm_layer.py > gpurun_out/pmc_gl_sq.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_gl/sq2 -- python3 tools/prof_gemm_layer.py > gpurun_out/pmc_gl_sq2.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, json, collections, os
M = os.environ["M"]
names = ["qkv 4096->6144", "o 4096->4096", "gate_up 4096->28672", "down 14336->4096"]
out = [collections.defaultdict(list) for _ in names]
kern = [None] * 4
dur = [[] for _ in names]
for d in ("sq", "sq2"):
    fs = glob.glob(f"gpurun_out/pmc_gl/{d}/*/*counter_collection.csv")
    acc = collections.defaultdict(float)
    meta = {}
    for r in csv.DictReader(open(fs[0])):
        if "fp8_gemm" in r["Kernel_Name"] and "finalize" not in r["Kernel_Name"]:
            acc[(int(r["Dispatch_Id"]), r["Counter_Name"])] += float(r["Counter_Value"])
            meta[int(r["Dispatch_Id"])] = r["Kernel_Name"]
    ids = sorted(meta)
    for n, did in enumerate(ids[:16]):
        kern[n // 4] = meta[did].replace("void sglm::(anonymous namespace)::", "")[:70]
        for (d_, c), v in acc.items():
            if d_ == did:
                out[n // 4][c].append(v)
for n, name in enumerate(names):
    o = {c: sum(x) / len(x) for c, x in out[n].items()}
    simd = o["GRBM_GUI_ACTIVE"] / 8 * 1024
    print(json.dumps({"M": int(M), "gemm": name, "kernel": kern[n],
                      "mfma_busy_of_simd_cycles": round(o["SQ_VALU_MFMA_BUSY_CYCLES"] / simd, 4),
                      "us_at_2.4GHz_from_GRBM": round(o["GRBM_GUI_ACTIVE"] / 8 / 2400, 1),
                      "wait_inst_any_of_wave_cycles": round(o["SQ_WAIT_INST_ANY"] / o["SQ_WAVE_CYCLES"], 3),
                      "wait_any_of_wave_cycles": round(o["SQ_WAIT_ANY"] / o["SQ_WAVE_CYCLES"], 3),
                      "waves_per_simd": round(o["SQ_WAVE_CYCLES"] * 4 / simd, 2),
                      "valu_per_mfma": round(o["SQ_INSTS_VALU"] / o["SQ_INSTS_MFMA"], 2),
                      "lds_per_mfma": round(o["SQ_INSTS_LDS"] / o["SQ_INSTS_MFMA"], 2),
                      **{c: round(v, 1) for c, v in o.items()}}))
PY
done
rm -rf gpurun_out/pmc_gl
