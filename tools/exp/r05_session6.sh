cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
# item 6: counters of the layer GEMMs at M = 1024 / 4096
step r05_gemm_pmc 600 bash tools/exp/prof_gemm_layer_pmc.sh
cat gpurun_out/r05_gemm_pmc.log | cut -c1-420
# item 7, experiment 1 and 2: existing tile forms / raster groups forced on the straggler shapes (M = 4096 qkv, o; M = 1024 gate_up)
export MS=1024,4096
step r05_t6_auto 200 python tools/bench_gemm_prefill_wshuf.py
SGL_MI355_T3_CB=3 step r05_t6_cb3 200 python tools/bench_gemm_prefill_wshuf.py
SGL_MI355_TILED_V3=1 step r05_t6_v3_256 200 python tools/bench_gemm_prefill_wshuf.py
SGL_MI355_T3_GN=2 step r05_t6_gn2 200 python tools/bench_gemm_prefill_wshuf.py
SGL_MI355_T3_GN=8 step r05_t6_gn8 200 python tools/bench_gemm_prefill_wshuf.py
SGL_MI355_T3_KS=1 step r05_t6_ks1 200 python tools/bench_gemm_prefill_wshuf.py
step r05_t6_auto2 200 python tools/bench_gemm_prefill_wshuf.py
for f in auto cb3 v3_256 gn2 gn8 ks1 auto2; do echo $f; python3 -c "
import json
for l in open('gpurun_out/r05_t6_$f.log'):
    if l.startswith('{'):
        d=json.loads(l); print('  ',d['M'],d['K'],d['N'],d['us'],d['TFLOPs'])
"; done
