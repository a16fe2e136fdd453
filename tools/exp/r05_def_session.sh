cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
step r05_def_pytest 900 python -m pytest tests/test_deferred_gpu.py -q -p no:cacheprovider -x
tail -30 gpurun_out/r05_def_pytest.log
step r05_def_full 900 python -m pytest tests -m gpu -q -p no:cacheprovider
tail -4 gpurun_out/r05_def_full.log
step r05_def_on 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
python - <<'PY'
import json
for n in ("on",):
    l=[x for x in open(f"gpurun_out/r05_def_{n}.log") if x.startswith("{")][-1]
    d=json.loads(l)
    print(n, d["value"], d["ms_per_step"], d["value_fused"], d["roofline"]["frac"], "ttft", d["ttft_ms_p50"], d["ttft_ms_p50_graph"], d["ttft_ms_p50_128"])
    for oc in d["other_configs"]: print(oc["config"][:40], oc["ms_per_step"], oc.get("fused_ms_per_step"), oc["ttft_ms_p50"])
PY
