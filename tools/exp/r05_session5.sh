cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
export CASES="1,32,8,512,0;1,32,8,1024,0;1,32,8,2048,0;1,32,8,4096,0;4,32,8,512,2048;4,32,8,2048,0;8,8,1,1024,0"
for i in 1 2; do
step r05_t5_base_$i 300 python tools/bench_extend_cases.py
SGL_MI355_LIB=sglang_npu_amd/lib/variants/libsgl_mi355_extend_chains.so step r05_t5_chains_$i 300 python tools/bench_extend_cases.py
done
for f in base_1 chains_1 base_2 chains_2; do echo $f; python3 -c "
import json
for l in open('gpurun_out/r05_t5_$f.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['B'],d['Hq'],d['Hkv'],d['L'],d['prefix'],d['kernel_us'],d['parts_us'])
"; done
