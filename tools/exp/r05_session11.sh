cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
export SGL_MI355_LIB=$GRAFT_REPO_ROOT/sglang_npu_amd/lib/variants/libsgl_mi355_ext_timing.so
CASE=1,4096,0 step r05_t11_phase_4096 300 python tools/exp/extend_phase_times.py
cat gpurun_out/r05_t11_phase_4096.log; tail -3 gpurun_out/r05_t11_phase_4096.err
CASE=4,2048,0 step r05_t11_phase_4x2048 300 python tools/exp/extend_phase_times.py
cat gpurun_out/r05_t11_phase_4x2048.log
