# PMC passes of the decode GEMMs on fragment-major weights (separate passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_gemm
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_gemm/fetch -- python3 tools/prof_gemm_decode.py > gpurun_out/pmc_gemm_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_gemm/write -- python3 tools/prof_gemm_decode.py > gpurun_out/pmc_gemm_write.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc_gemm/stats -- python3 tools/prof_gemm_decode.py > gpurun_out/pmc_gemm_stats.log 2>&1
python3 - <<'PY'
import csv, glob, json, collections
def per_kernel(counter, d):
    f = glob.glob(f"gpurun_out/pmc_gemm/{d}/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and "fp8_gemm" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][-60:], r["Grid_Size"] if "Grid_Size" in r else "")].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
fs, ws = per_kernel("FETCH_SIZE", "fetch"), per_kernel("WRITE_SIZE", "write")
out = []
for k in fs:
    out.append({"kernel": k[0], "grid": k[1], "launches": fs[k][1], "FETCH_SIZE_KB_raw": round(fs[k][0], 1),
                "WRITE_SIZE_KB_raw": round(ws.get(k, (0, 0))[0], 1),
                "hbm_MB_per_launch": round((fs[k][0] * 2 + ws.get(k, (0, 0))[0]) * 1024 / 1e6, 2)})
print(json.dumps(out, indent=1))
PY
