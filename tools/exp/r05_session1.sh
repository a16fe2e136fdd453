source tools/gpu_steps.sh
step r05_t1_pytest 900 python -m pytest tests -m gpu -q -p no:cacheprovider
tail -40 gpurun_out/r05_t1_pytest.log
SGL_MI355_LIB=sglang_npu_amd/lib/variants/libsgl_mi355_ar_nowait.so step r05_t1_race_nowait 300 python -m pytest tests/test_custom_allreduce_gpu.py -k staging -q -p no:cacheprovider
tail -15 gpurun_out/r05_t1_race_nowait.log
SGL_MI355_SHARE_GPU=1 step r05_rehearsal_ws2 600 python bench.py --gpus 2 --steps 8 --warmup 2
tail -c 600 gpurun_out/r05_rehearsal_ws2.log; tail -20 gpurun_out/r05_rehearsal_ws2.err
step r05_bench_tp1_a 600 python bench.py --steps 20 --warmup 5
tail -c 1500 gpurun_out/r05_bench_tp1_a.log; tail -5 gpurun_out/r05_bench_tp1_a.err
