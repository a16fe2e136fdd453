cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
step r05_small_w4 600 python tools/exp/decode_small_probe.py 8 1
cat gpurun_out/r05_small_w4.log; tail -3 gpurun_out/r05_small_w4.err
SGL_MI355_DECODE_WAVES=2 step r05_small_w2 600 python tools/exp/decode_small_probe.py 8 1
cat gpurun_out/r05_small_w2.log; tail -3 gpurun_out/r05_small_w2.err
