cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
step r05_s3_pytest 600 python -m pytest tests/test_decode_split_merge_gpu.py tests/test_decode_gpu.py tests/test_fp8kv_gpu.py -q -p no:cacheprovider
tail -4 gpurun_out/r05_s3_pytest.log
FORMS=merged,merged_fp8 step r05_s3_probe 600 python tools/exp/decode_small_probe.py 8 1
cat gpurun_out/r05_s3_probe.log
