# kernel trace of the default bench command -> median decoder-layer breakdowns: decode step, 1024-token prefill, 128-token prefill
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_lb
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lb -- python3 bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/r04_bench_profiled.json 2> gpurun_out/r04_bench_profiled.err || exit 1
T=$(ls gpurun_out/prof_lb/*/*kernel_trace.csv | head -1)
{ echo "## decode step (bs=64, ctx 2048), one layer"; python tools/layer_breakdown.py $T decode_mfma;
  echo; echo "## 1024-token prefill (bs=1), one layer"; python tools/layer_breakdown.py $T "extend_mfma_kernel<0, 128, int, 2, false, false, 2" fp8_gemm_tiled3;
  echo; echo "## 128-token prefill (bs=1; eager and graph-replayed passes mixed: read the busy column), one layer"; python tools/layer_breakdown.py $T "extend_mfma_kernel<0, 128, int, 2, false, false, 2" "fp8_gemm_wstream_kernel<0, 8"; } > gpurun_out/r04_layer_breakdowns.txt 2>&1
cp $(ls gpurun_out/prof_lb/*/*kernel_stats.csv | head -1) gpurun_out/r04_bench_tp1_kernel_stats.csv
cat gpurun_out/r04_layer_breakdowns.txt
rm -rf gpurun_out/prof_lb
