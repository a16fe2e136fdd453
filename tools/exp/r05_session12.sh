cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
V=$GRAFT_REPO_ROOT/sglang_npu_amd/lib/variants
SGL_MI355_LIB=$V/libsgl_mi355_ext_inter.so step r05_t12_pytest 600 python -m pytest tests/test_extend_gpu.py tests/test_extend_parts_gpu.py tests/test_high_address_gpu.py tests/test_fp8kv_gpu.py tests/test_fp8kv_e5m2_gpu.py tests/test_backend_gpu.py -q -p no:cacheprovider
tail -4 gpurun_out/r05_t12_pytest.log
export CASES="1,32,8,256,0;1,32,8,512,0;1,32,8,1024,0;1,32,8,2048,0;1,32,8,4096,0;1,32,8,8192,0;4,32,8,512,2048;4,32,8,2048,0;8,8,1,1024,0;16,32,8,128,1024"
for i in 1 2; do
step r05_t12_base_$i 300 python tools/bench_extend_cases.py
SGL_MI355_LIB=$V/libsgl_mi355_ext_inter.so step r05_t12_inter_$i 300 python tools/bench_extend_cases.py
done
for f in base_1 inter_1 base_2 inter_2; do echo $f; python3 -c "
import json
for l in open('gpurun_out/r05_t12_$f.log'):
    if l.startswith('{'):
        d=json.loads(l); print(' ',d['B'],d['Hq'],d['Hkv'],d['L'],d['prefix'],d['kernel_us'],d['parts_us'])
"; done
SGL_MI355_LIB=$V/libsgl_mi355_ext_timing.so CASE=1,4096,0 step r05_t12_phase_4096 300 python tools/exp/extend_phase_times.py
cat gpurun_out/r05_t12_phase_4096.log
