cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
step r05_t10_pytest 600 python -m pytest tests/test_elementwise_gpu.py tests/test_model_parity_gpu.py tests/test_backend_gpu.py tests/test_tp_model_gpu.py -q -p no:cacheprovider
tail -4 gpurun_out/r05_t10_pytest.log
SGL_MI355_SHARE_GPU=1 SGL_MI355_BENCH_EXTRA_70B=1 step r05_rehearsal_ws2_70b 900 python bench.py --gpus 2 --steps 4 --warmup 1
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r05_rehearsal_ws2_70b.log') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['launcher'])
print(d.get('config5_llama3_70b'))
"
tail -5 gpurun_out/r05_rehearsal_ws2_70b.err
step r05_t10_bench 600 python bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r05_t10_bench.log') if l.startswith('{')][-1])
for k in ['value','ms_per_step','value_fused','ttft_ms_p50','ttft_ms_p50_128','ttft_ms_p50_graph']: print(k,d.get(k))
"
