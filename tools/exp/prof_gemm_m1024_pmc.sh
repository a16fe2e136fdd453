# SQ counter passes + kernel stats of the one-tile-per-CU prefill GEMMs (separate passes, kernel-trace only); summary to stdout
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_m1024
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_m1024/sq -- python3 tools/prof_gemm_m1024.py > gpurun_out/pmc_m1024_sq.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_m1024/sq2 -- python3 tools/prof_gemm_m1024.py > gpurun_out/pmc_m1024_sq2.log 2>&1 &&
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_MEM_VIOLATIONS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d gpurun_out/pmc_m1024/sq3 -- python3 tools/prof_gemm_m1024.py > gpurun_out/pmc_m1024_sq3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc_m1024/stats -- python3 tools/prof_gemm_m1024.py > gpurun_out/pmc_m1024_stats.log 2>&1
python3 - <<'PY'
import csv, glob, json, collections
out = collections.defaultdict(dict)
for d in ("sq", "sq2", "sq3"):
    fs = glob.glob(f"gpurun_out/pmc_m1024/{d}/*/*counter_collection.csv")
    if not fs:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "fp8_gemm_tiled" in r["Kernel_Name"]:
            acc[(r["Dispatch_Id"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    # the first 8 tiled launches are down_proj, the next 8 o_proj
    ids = sorted({int(k[0]) for k in acc})
    for n, did in enumerate(ids):
        which = "down_14336" if n < 8 else "o_4096"
        for (d_, c), v in acc.items():
            if int(d_) == did:
                out[which].setdefault(c, []).append(sum(v))
for k, v in out.items():
    print(json.dumps({"gemm": k, **{c: round(sum(x) / len(x), 1) for c, x in v.items()}}))
fs = glob.glob("gpurun_out/pmc_m1024/stats/*/*kernel_stats.csv")
if fs:
    for r in csv.DictReader(open(fs[0])):
        if "fp8_gemm" in r["Name"]:
            print(r["Name"][:90], r["Calls"], r["AverageNs"])
PY
rm -rf gpurun_out/pmc_m1024
