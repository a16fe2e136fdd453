# Batch-size / context / KV-dtype sweep of the headline model (one GPU): one JSON summary line per point.
# usage: bash tools/sweep_bench.sh > gpurun_out/r03_sweep.txt
for spec in "1 2048 auto" "8 2048 auto" "16 2048 auto" "32 2048 auto" "64 512 auto" "64 2048 auto" "64 8192 auto" "128 2048 auto" "256 2048 auto" "64 2048 fp8_e4m3" "128 2048 fp8_e4m3"; do
  set -- $spec
  timeout -k 10 300 python bench.py --batch $1 --ctx $2 --kv-dtype $3 --steps 16 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(dict(batch=$1, ctx=$2, kv='$3', ms_per_step=d['ms_per_step'], tokens_per_s=d['value'], attn_frac_hbm=d['roofline']['frac'], attn_us=d['roofline'].get('avg_launch_us'))))"
done
