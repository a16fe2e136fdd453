cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
step r05_nk_pytest 900 python -m pytest tests/test_decode_newkv_gpu.py tests/test_backend_gpu.py tests/test_model_parity_gpu.py tests/test_decode_gpu.py -q -p no:cacheprovider -x
tail -12 gpurun_out/r05_nk_pytest.log
SGL_MI355_DECODE_KV_WRITE_FUSION=1 step r05_nk_bench_on 400 python bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline
tail -c 1500 gpurun_out/r05_nk_bench_on.log
step r05_nk_bench_off 400 python bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline
tail -c 1500 gpurun_out/r05_nk_bench_off.log
SGL_MI355_DECODE_KV_WRITE_FUSION=1 step r05_nk_bench_on2 400 python bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline
tail -c 600 gpurun_out/r05_nk_bench_on2.log
