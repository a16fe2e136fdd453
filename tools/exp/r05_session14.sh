cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
A="bench.py --model llama3-70b --emulate-tp 8 --steps 16 --warmup 3 --no-cpu-baseline --no-other-configs --call-order fused"
step r05_t14_base 300 python $A
SGL_MI355_GATE_UP_PARTIALS_MAX_N=8192 step r05_t14_gup 300 python $A
SGL_MI355_DECODE_WAVES=8 step r05_t14_w8 300 python $A
SGL_MI355_NO_SPLIT_MERGE_FUSION=1 step r05_t14_nomerge 300 python $A
step r05_t14_base2 300 python $A
for f in base gup w8 nomerge base2; do python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r05_t14_$f.log') if l.startswith('{')][-1])
print('$f', d['ms_per_step'], d['fused_ms_per_step'], d['dropin_ms_per_step'], d['roofline']['avg_launch_us'])
"; done
