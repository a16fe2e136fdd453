# SQ counter pass + kernel stats of the AWQ decode GEMM (separate passes, kernel-trace only); summary to stdout
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-v2}
rm -rf gpurun_out/pmc_awq_$TAG
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_awq_$TAG/sq -- python3 tools/prof_awq_decode.py > gpurun_out/pmc_awq_${TAG}_sq.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_awq_$TAG/sq2 -- python3 tools/prof_awq_decode.py > gpurun_out/pmc_awq_${TAG}_sq2.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc_awq_$TAG/stats -- python3 tools/prof_awq_decode.py > gpurun_out/pmc_awq_${TAG}_stats.log 2>&1
python3 - $TAG <<'PY'
import csv, glob, json, collections, sys
tag = sys.argv[1]
out = collections.defaultdict(dict)
for d in ("sq", "sq2"):
    fs = glob.glob(f"gpurun_out/pmc_awq_{tag}/{d}/*/*counter_collection.csv")
    if not fs:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "awq_wstream" in r["Kernel_Name"]:
            key = (r["Kernel_Name"].split("(")[0][-48:], r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
            acc[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (key, c), v in acc.items():
        out[key][c] = round(sum(v) / len(v), 1)
        out[key]["launches"] = len(v)
for k, v in out.items():
    print(json.dumps({"kernel": k[0], "grid": k[1], "wg": k[2], **v}))
fs = glob.glob(f"gpurun_out/pmc_awq_{tag}/stats/*/*kernel_stats.csv")
if fs:
    for r in csv.DictReader(open(fs[0])):
        if "awq" in r["Name"]:
            print(r["Name"][:90], r["Calls"], r["AverageNs"])
PY
