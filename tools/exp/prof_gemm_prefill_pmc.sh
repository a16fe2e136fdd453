# PMC passes of the prefill FP8 GEMM (separate passes, kernel-trace only): effective clock and MFMA busy fraction
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_prefill
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_prefill/clk -- python3 tools/prof_gemm_prefill.py > gpurun_out/pmc_prefill_clk.log 2>&1 &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_prefill/mfma -- python3 tools/prof_gemm_prefill.py > gpurun_out/pmc_prefill_mfma.log 2>&1 &&
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F8 --kernel-trace --output-format csv -d gpurun_out/pmc_prefill/wait -- python3 tools/prof_gemm_prefill.py > gpurun_out/pmc_prefill_wait.log 2>&1
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for d in ("clk", "mfma", "wait"):
    fs = glob.glob(f"gpurun_out/pmc_prefill/{d}/*/*counter_collection.csv")
    if not fs: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "tiled" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r: acc["_dur_ns_" + d].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, v in acc.items(): out[k] = [sum(v) / len(v), len(v)]
print(json.dumps(out, indent=1))
PY
