cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
SGL_MI355_LIB=$PWD/sglang_npu_amd/lib/variants/libsgl_mi355_fusions.so step r05_nk2_variant 900 python -m pytest tests/test_decode_newkv_gpu.py tests/test_decode_fused_qkv_gpu.py tests/test_attn_quant_fusion_gpu.py tests/test_decode_gpu.py tests/test_backend_gpu.py tests/test_model_parity_gpu.py -q -p no:cacheprovider
tail -5 gpurun_out/r05_nk2_variant.log
