cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
step r05_t8_pytest 600 python -m pytest tests/test_fp8_companion_gpu.py tests/test_fp8_wshuffled_gpu.py tests/test_backend_gpu.py tests/test_model_parity_gpu.py tests/test_elementwise_gpu.py -q -p no:cacheprovider
tail -12 gpurun_out/r05_t8_pytest.log
step r05_t8_bench 600 python bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r05_t8_bench.log') if l.startswith('{')][-1])
for k in ['value','ms_per_step','value_dropin','value_fused','fused_ms_per_step','dropin_ms_per_step','ttft_ms_p50','ttft_ms_p50_128','ttft_ms_p50_graph']: print(k,d.get(k))
print(d['roofline']['frac'])
"
SGL_MI355_NO_FP8_COMPANION=1 step r05_t8_bench_nocomp 600 python bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r05_t8_bench_nocomp.log') if l.startswith('{')][-1])
for k in ['value','ms_per_step','value_dropin','value_fused','fused_ms_per_step','dropin_ms_per_step','ttft_ms_p50','ttft_ms_p50_128','ttft_ms_p50_graph']: print(k,d.get(k))
"
