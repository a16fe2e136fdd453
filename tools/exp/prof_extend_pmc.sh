# SQ counter passes of the extend attention kernel (separate passes, kernel-trace only); CASE as tools/prof_extend.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${TAG:-x}
rm -rf gpurun_out/pmc_ext_$TAG
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_ext_$TAG/sq -- python3 tools/prof_extend.py > gpurun_out/pmc_ext_${TAG}_sq.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_ext_$TAG/sq2 -- python3 tools/prof_extend.py > gpurun_out/pmc_ext_${TAG}_sq2.log 2>&1 &&
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d gpurun_out/pmc_ext_$TAG/sq3 -- python3 tools/prof_extend.py > gpurun_out/pmc_ext_${TAG}_sq3.log 2>&1
python3 - <<PY
import csv, glob, json, collections
tot = collections.defaultdict(list)
for d in ("sq", "sq2", "sq3"):
    fs = glob.glob(f"gpurun_out/pmc_ext_$TAG/{d}/*/*counter_collection.csv")
    if not fs:
        continue
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        if "extend_" in r["Kernel_Name"]:
            acc[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            name = r["Kernel_Name"][:120]
    for (d_, c), v in acc.items():
        tot[c].append(v)
print(json.dumps({"tag": "$TAG", "case": "${CASE:-1,4096,0}", "kernel": name,
                  **{c: round(sum(x) / len(x), 1) for c, x in tot.items()}}))
PY
rm -rf gpurun_out/pmc_ext_$TAG
