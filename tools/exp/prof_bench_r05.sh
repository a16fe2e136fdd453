# Round 5 profile set of the bench command (separate --pmc passes, kernel-trace only): in-step HBM traffic of the decode attention
# kernel, kernel stats, layer breakdowns of the decode step in both call orders.  Output: gpurun_out/r05_*.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs"
rm -rf gpurun_out/pmc_r5
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r5/fetch -- python3 $ARGS > gpurun_out/pmc_r5_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r5/write -- python3 $ARGS > gpurun_out/pmc_r5_write.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc_r5/stats -- python3 bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/pmc_r5_stats.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, json
def per_launch(counter, d):
    f = glob.glob(f"gpurun_out/pmc_r5/{d}/*/*counter_collection.csv")[0]
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if r["Counter_Name"] == counter and "decode_mfma_pair" in r["Kernel_Name"]]
fs, ws = per_launch("FETCH_SIZE", "fetch"), per_launch("WRITE_SIZE", "write")
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
json.dump(out, open("gpurun_out/r05_decode_pmc_instep.json", "w"), indent=1)
print(json.dumps(out))
for counter, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    f = glob.glob(f"gpurun_out/pmc_r5/{d}/*/*counter_collection.csv")[0]
    with open(f"gpurun_out/r05_decode_pmc_instep_{counter}.csv", "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(["Dispatch_Id", "Kernel", "Grid_Size", "Counter_Name", "Counter_Value"])
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "decode_mfma_pair" in r["Kernel_Name"]:
                w.writerow([r["Dispatch_Id"], r["Kernel_Name"][:58], r["Grid_Size"], counter, r["Counter_Value"]])
PY
cp gpurun_out/pmc_r5/stats/*/*kernel_stats.csv gpurun_out/r05_bench_tp1_kernel_stats.csv
# layer breakdowns BEFORE the traces are dropped: one trace per call order
for order in reference fused; do
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pmc_r5/trace_$order -- python3 bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-other-configs --call-order $order > gpurun_out/pmc_r5_trace_$order.log 2>&1
  python3 tools/layer_breakdown.py gpurun_out/pmc_r5/trace_$order/*/*kernel_trace.csv decode_mfma_pair > gpurun_out/r05_layer_breakdown_decode_$order.txt 2>&1
  # ... and the most common OTHER layer length of the same trace (the other call order runs beside it for min(steps, 10) steps)
  for n in 8 9 10 11 12 13 14; do LAYER_COUNT=$n python3 tools/layer_breakdown.py gpurun_out/pmc_r5/trace_$order/*/*kernel_trace.csv decode_mfma_pair >> gpurun_out/r05_layer_breakdown_decode_${order}_by_count.txt 2>/dev/null; done
  echo "== $order"; cat gpurun_out/r05_layer_breakdown_decode_$order.txt
done
python3 tools/layer_breakdown.py gpurun_out/pmc_r5/trace_reference/*/*kernel_trace.csv "extend_mfma_kernel<0, 128, int, 2, false, false, 2" fp8_gemm_tiled3 > gpurun_out/r05_layer_breakdown_prefill.txt 2>&1
cat gpurun_out/r05_layer_breakdown_prefill.txt
rm -rf gpurun_out/pmc_r5
