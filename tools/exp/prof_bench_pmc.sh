# HBM traffic of the decode attention kernel INSIDE the bench step (separate --pmc passes, kernel-trace only), plus the
# kernel stats and the layer breakdown of the same command.  Output: gpurun_out/r04_decode_pmc_instep.json etc.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs"
rm -rf gpurun_out/pmc_r4
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r4/fetch -- python3 $ARGS > gpurun_out/pmc_r4_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r4/write -- python3 $ARGS > gpurun_out/pmc_r4_write.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc_r4/stats -- python3 bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/pmc_r4_stats.log 2>&1
python3 - <<'PY'
import csv, glob, json
def per_launch(counter, d):
    f = glob.glob(f"gpurun_out/pmc_r4/{d}/*/*counter_collection.csv")[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if r["Counter_Name"] == counter and "decode_mfma_pair" in r["Kernel_Name"]]
    return vals
fs, ws = per_launch("FETCH_SIZE", "fetch"), per_launch("WRITE_SIZE", "write")
# bench.py: ctx 2048, warm-up passes and steps advance the sequence length by one each; the exact length of every
# launch is not in the trace, so the ratio is taken against the MEAN algorithmic bytes over the lengths the run visits
# (2048 .. 2048 + ~20: +-0.5 % around the mean)
n = min(len(fs), len(ws))
mean_fetch, mean_write = sum(fs) / len(fs), sum(ws) / len(ws)
hbm = mean_fetch * 1024 * 2 + mean_write * 1024
def alg(ctx): return 64 * ctx * 8 * 2 * 128 * 2 + 4 * 64 * ctx + 2 * 64 * 32 * 2 * 128
lo, hi = alg(2048), alg(2048 + 24)
out = {"kernel": "decode_mfma_pair_kernel inside bench.py's decode step (graph replay + eager instrumented passes)",
       "launches": [len(fs), len(ws)], "FETCH_SIZE_KB_per_launch_raw": round(mean_fetch, 2),
       "WRITE_SIZE_KB_per_launch_raw": round(mean_write, 2), "hbm_bytes_per_launch": int(hbm),
       "algorithmic_bytes_per_launch_ctx2048": lo, "algorithmic_bytes_per_launch_ctx2072": hi,
       "traffic_over_algorithmic": round(hbm / ((lo + hi) / 2), 4),
       "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `python3 bench.py --steps 8 --warmup 2 "
                 "--no-cpu-baseline --no-other-configs`; FETCH_SIZE x 2 (gfx950: 128-B units reported as 64-B) + WRITE_SIZE, KB"}
json.dump(out, open("gpurun_out/r04_decode_pmc_instep.json", "w"), indent=1)
print(json.dumps(out))
PY
cp gpurun_out/pmc_r4/stats/*/*kernel_stats.csv gpurun_out/r04_bench_tp1_kernel_stats.csv 2>/dev/null
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pmc_r4/trace -- python3 bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/pmc_r4_trace.log 2>&1
python3 tools/layer_breakdown.py gpurun_out/pmc_r4/trace/*/*kernel_trace.csv > gpurun_out/r04_layer_breakdown_decode.txt 2>&1
cat gpurun_out/r04_layer_breakdown_decode.txt
python3 tools/layer_breakdown.py gpurun_out/pmc_r4/trace/*/*kernel_trace.csv "extend_mfma_kernel<0, 128, int, 2, false, false, 2" fp8_gemm_tiled3 > gpurun_out/r04_layer_breakdown_prefill.txt 2>&1
cat gpurun_out/r04_layer_breakdown_prefill.txt
# the per-launch counter rows of the decode attention kernel (small), then drop the raw traces (tens of MB)
python3 - <<'PY'
import csv, glob
for counter, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    f = glob.glob(f"gpurun_out/pmc_r4/{d}/*/*counter_collection.csv")[0]
    with open(f"gpurun_out/r04_decode_pmc_instep_{counter}.csv", "w", newline="") as out:
        w = csv.writer(out)
        w.writerow(["Dispatch_Id", "Kernel", "Grid_Size", "Counter_Name", "Counter_Value"])
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "decode_mfma_pair" in r["Kernel_Name"]:
                w.writerow([r["Dispatch_Id"], r["Kernel_Name"][:58], r["Grid_Size"], counter, r["Counter_Value"]])
PY
# (round 5, ADVICE r4: this breakdown used to run AFTER the traces were deleted and wrote only an error; the round-5 profile
#  set is tools/exp/prof_bench_r05.sh, which takes every breakdown before it drops the traces)
python3 tools/layer_breakdown.py gpurun_out/pmc_r4/trace/*/*kernel_trace.csv "extend_mfma_kernel<0, 128, int, 2, false, false, 2" "fp8_gemm_wstream_kernel<0, 8" > gpurun_out/r04_layer_breakdown_prefill_128.txt 2>&1
cat gpurun_out/r04_layer_breakdown_prefill_128.txt
rm -rf gpurun_out/pmc_r4
