cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
step r05_def_pytest 900 python -m pytest tests/test_fp8_companion_gpu.py tests/test_deferred_gpu.py tests/test_tp_model_gpu.py tests/test_backend_gpu.py tests/test_decode_split_merge_gpu.py -q -p no:cacheprovider -x
tail -30 gpurun_out/r05_def_pytest.log
step r05_def_70ref 400 python bench.py --model llama3-70b --emulate-tp 8 --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs --call-order reference
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r05_def_70ref.log") if x.startswith("{")][-1]
d=json.loads(l)
print("70b ref", d["value"], d["ms_per_step"], d.get("fused_ms_per_step"))
PY
