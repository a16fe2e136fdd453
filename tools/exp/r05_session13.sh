cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
step r05_t13_pytest 900 python -m pytest tests/test_randomized_gpu.py tests/test_extend_gpu.py tests/test_extend_parts_gpu.py tests/test_high_address_gpu.py tests/test_fp8kv_gpu.py tests/test_backend_gpu.py -q -p no:cacheprovider
tail -15 gpurun_out/r05_t13_pytest.log | cut -c1-400
