# PMC passes of the decode kernel (separate passes, kernel-trace only -- no other trace domain with --pmc)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_r2
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r2/fetch -- python3 tools/prof_decode.py > gpurun_out/pmc_r2_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r2/write -- python3 tools/prof_decode.py > gpurun_out/pmc_r2_write.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc_r2/stats -- python3 tools/prof_decode.py > gpurun_out/pmc_r2_stats.log 2>&1
python3 - <<'PY'
import csv, glob, json
def avg(counter, d):
    f = glob.glob(f"gpurun_out/pmc_r2/{d}/*/*counter_collection.csv")[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and "decode_mfma" in r["Kernel_Name"]]
    return sum(vals) / len(vals), len(vals)
fs, n1 = avg("FETCH_SIZE", "fetch")
ws, n2 = avg("WRITE_SIZE", "write")
st = glob.glob("gpurun_out/pmc_r2/stats/*/*kernel_stats.csv")[0]
row = [r for r in csv.DictReader(open(st)) if "decode_mfma" in r["Name"]][0]
alg = 64 * 2048 * 8 * 2 * 128 * 2 + 4 * 64 * 2048 + 2 * 64 * 32 * 2 * 128
hbm = fs * 1024 * 2 + ws * 1024
print(json.dumps({"FETCH_SIZE_KB_per_launch_raw": round(fs, 2), "WRITE_SIZE_KB_per_launch_raw": round(ws, 2), "launches": [n1, n2],
                  "hbm_bytes_per_launch": int(hbm), "algorithmic_bytes_per_launch": alg,
                  "traffic_over_algorithmic": round(hbm / alg, 4), "kernel": row["Name"][:120],
                  "avg_us_isolated_profiled": round(float(row["AverageNs"]) / 1e3, 2), "calls": int(row["Calls"])}))
PY
