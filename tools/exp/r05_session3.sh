cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
step r05_t3_pytest 600 python -m pytest tests/test_extend_gpu.py tests/test_extend_parts_gpu.py tests/test_high_address_gpu.py tests/test_backend_gpu.py tests/test_fp8kv_gpu.py tests/test_fp8kv_e5m2_gpu.py -q -p no:cacheprovider
tail -5 gpurun_out/r05_t3_pytest.log
step r05_t3_extend_new 300 python tools/bench_extend_cases.py
SGL_MI355_LIB=sglang_npu_amd/lib/variants/libsgl_mi355_extend_r4.so step r05_t3_extend_r4 300 python tools/bench_extend_cases.py
step r05_t3_extend_new2 300 python tools/bench_extend_cases.py
echo NEW; cat gpurun_out/r05_t3_extend_new.log; echo R4; cat gpurun_out/r05_t3_extend_r4.log; echo NEW2; cat gpurun_out/r05_t3_extend_new2.log
