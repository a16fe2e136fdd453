cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
export SGL_MI355_LIB=$PWD/sglang_npu_amd/lib/variants/libsgl_mi355_dec_timing.so
step r05_dpt_a 300 python tools/exp/decode_phase_times.py
cat gpurun_out/r05_dpt_a.log; tail -3 gpurun_out/r05_dpt_a.err
CASE=64,8,1,256,1 step r05_dpt_b 300 python tools/exp/decode_phase_times.py
cat gpurun_out/r05_dpt_b.log
CASE=64,8,1,4096,4 step r05_dpt_c 300 python tools/exp/decode_phase_times.py
cat gpurun_out/r05_dpt_c.log
FP8_OUT=1 step r05_dpt_d 300 python tools/exp/decode_phase_times.py
cat gpurun_out/r05_dpt_d.log
