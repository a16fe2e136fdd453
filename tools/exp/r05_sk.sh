cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
step r05_sk_auto 300 python tools/exp/splitk_prefill_probe.py
SGL_MI355_TILED_V3=1 step r05_sk_v1 300 python tools/exp/splitk_prefill_probe.py
SGL_MI355_TILED_V3=2 step r05_sk_v2 300 python tools/exp/splitk_prefill_probe.py
cat gpurun_out/r05_sk_auto.log gpurun_out/r05_sk_v1.log gpurun_out/r05_sk_v2.log
