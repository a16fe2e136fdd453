cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SGL_MI355_LIB=sglang_npu_amd/lib/variants/libsgl_mi355_fusions.so timeout -k 10 300 python -m pytest tests/test_attn_quant_fusion_gpu.py tests/test_decode_fused_qkv_gpu.py -x -q > gpurun_out/r04_t7_fusions_variant.log 2>&1; echo "variant tests rc=$?"; tail -2 gpurun_out/r04_t7_fusions_variant.log
timeout -k 10 300 python -m pytest tests/test_attn_quant_fusion_gpu.py tests/test_decode_fused_qkv_gpu.py tests/test_abi.py -q 2>&1 | tail -2
rm -rf gpurun_out/prof70
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof70 -- python3 bench.py --model llama3-70b --emulate-tp 8 --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/r04_bench70_prof.json 2> gpurun_out/r04_bench70_prof.err || exit 1
python tools/layer_breakdown.py gpurun_out/prof70/*/*kernel_trace.csv > gpurun_out/r04_layer_breakdown_70b_tp8_rank.txt; cat gpurun_out/r04_layer_breakdown_70b_tp8_rank.txt
rm -rf gpurun_out/prof70
