cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
step r05_t4_pytest 300 python -m pytest tests/test_high_address_gpu.py -q -p no:cacheprovider
tail -3 gpurun_out/r05_t4_pytest.log
(for c in 1,1024,0 1,4096,0 4,512,2048; do TAG=$(echo $c | tr , _) CASE=$c timeout -k 10 300 bash tools/exp/prof_extend_pmc.sh || exit 1; done) > gpurun_out/r05_extend_pmc_new.txt 2> gpurun_out/r05_extend_pmc_new.err
cat gpurun_out/r05_extend_pmc_new.txt
(for c in 1,1024,0 1,4096,0; do TAG=$(echo $c | tr , _) CASE=$c SGL_MI355_LIB=$GRAFT_REPO_ROOT/sglang_npu_amd/lib/variants/libsgl_mi355_extend_r4.so timeout -k 10 300 bash tools/exp/prof_extend_pmc.sh || exit 1; done) > gpurun_out/r05_extend_pmc_r4.txt 2> gpurun_out/r05_extend_pmc_r4.err
cat gpurun_out/r05_extend_pmc_r4.txt
