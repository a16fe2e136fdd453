cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
step r05_t2_pytest 600 python -m pytest tests/test_backend_gpu.py tests/test_decode_gpu.py tests/test_custom_allreduce_gpu.py -q -p no:cacheprovider
tail -5 gpurun_out/r05_t2_pytest.log
SGL_MI355_LIB=sglang_npu_amd/lib/variants/libsgl_mi355_ar_nowait.so step r05_t2_race_nowait 300 python -m pytest tests/test_custom_allreduce_gpu.py -k staging -q -p no:cacheprovider
tail -5 gpurun_out/r05_t2_race_nowait.log
for order in reference fused; do
  rm -rf gpurun_out/prof70
  step r05_bench70_$order 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof70 -- python3 bench.py --model llama3-70b --emulate-tp 8 --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs --call-order $order
  python tools/layer_breakdown.py gpurun_out/prof70/*/*kernel_trace.csv > gpurun_out/r05_layer_breakdown_70b_tp8_rank_$order.txt; cat gpurun_out/r05_layer_breakdown_70b_tp8_rank_$order.txt
  tail -3 gpurun_out/r05_bench70_$order.err
done
rm -rf gpurun_out/prof70
