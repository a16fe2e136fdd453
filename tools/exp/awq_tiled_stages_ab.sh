# same-box A/B of the tiled AWQ GEMM variants: SGL_MI355_AWQ_TILED="stages,rows"
for v in auto 3,128 2,128 3,64 2,64; do echo "variant $v"; if [ $v = auto ]; then unset SGL_MI355_AWQ_TILED; else export SGL_MI355_AWQ_TILED=$v; fi; python tools/bench_awq_tiled.py 2>&1 | grep -v amdgpu | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    print(d['K'],d['N'],d['M'],d['fused_us'],d['fused_TFLOPs'], 'lib', d['lib_gemm_on_fp16_copy_us'])
"; done
