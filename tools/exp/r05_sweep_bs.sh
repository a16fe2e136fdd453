cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
for b in 1 8 32 48 128; do
  step r05_sw_on_$b 300 python bench.py --batch $b --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline
  SGL_MI355_NO_DEFERRED_EPILOGUE=1 step r05_sw_off_$b 300 python bench.py --batch $b --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline
done
python - <<'PY'
import json
for b in (1,8,32,48,128):
    row=[]
    for n in ("on","off"):
        try:
            l=[x for x in open(f"gpurun_out/r05_sw_{n}_{b}.log") if x.startswith("{")][-1]
            d=json.loads(l); row.append((d["ms_per_step"], d["fused_ms_per_step"]))
        except Exception as e: row.append(("err", str(e)[:40]))
    print("bs", b, "reference on/off:", row[0][0], row[1][0], " fused:", row[0][1], row[1][1])
PY
