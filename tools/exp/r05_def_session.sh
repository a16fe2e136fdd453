cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
SGL_MI355_SHARE_GPU=1 step r05_rehearsal_ws2 600 python bench.py --gpus 2 --steps 8 --warmup 2
tail -c 600 gpurun_out/r05_rehearsal_ws2.log; tail -4 gpurun_out/r05_rehearsal_ws2.err
SGL_MI355_SHARE_GPU=1 step r05_rehearsal_ws4 600 python bench.py --gpus 4 --steps 8 --warmup 2
tail -c 600 gpurun_out/r05_rehearsal_ws4.log; tail -4 gpurun_out/r05_rehearsal_ws4.err
