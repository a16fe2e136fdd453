cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
step r05_t9_pytest 900 python -m pytest tests/test_custom_allreduce_gpu.py tests/test_tp_model_gpu.py tests/test_fp8_gpu.py tests/test_fp8_wshuffled_gpu.py tests/test_bench_self_launch_gpu.py -q -p no:cacheprovider
tail -8 gpurun_out/r05_t9_pytest.log
SGL_MI355_SHARE_GPU=1 step r05_rehearsal_ws2 600 python bench.py --gpus 2 --steps 8 --warmup 2
tail -c 400 gpurun_out/r05_rehearsal_ws2.log; tail -4 gpurun_out/r05_rehearsal_ws2.err
SGL_MI355_SHARE_GPU=1 step r05_rehearsal_ws4 600 python bench.py --gpus 4 --steps 8 --warmup 2
tail -c 400 gpurun_out/r05_rehearsal_ws4.log; tail -4 gpurun_out/r05_rehearsal_ws4.err
