# the round's closing set: GPU suite, the default bench line, the profile set of the bench command, the 70B TP = 8 rank breakdowns
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
step r05_final_pytest 900 python -m pytest tests -m gpu -q -p no:cacheprovider
tail -3 gpurun_out/r05_final_pytest.log
step r05_bench_tp1 600 python bench.py --steps 20 --warmup 5
tail -c 300 gpurun_out/r05_bench_tp1.log
step r05_prof_bench 900 bash tools/exp/prof_bench_r05.sh
grep -A18 "^== " gpurun_out/r05_prof_bench.log | head -60
for order in reference fused; do
  rm -rf gpurun_out/prof70
  step r05_bench70_$order 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof70 -- python3 bench.py --model llama3-70b --emulate-tp 8 --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs --call-order $order
  for n in 9 10 11 12 13 14 15 16; do LAYER_COUNT=$n python tools/layer_breakdown.py gpurun_out/prof70/*/*kernel_trace.csv decode_mfma 2>/dev/null; done > gpurun_out/r05_layer_breakdown_70b_tp8_rank_$order.txt
  head -40 gpurun_out/r05_layer_breakdown_70b_tp8_rank_$order.txt
done
rm -rf gpurun_out/prof70
