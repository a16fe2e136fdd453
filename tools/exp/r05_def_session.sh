cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
step r05_def_full 900 python -m pytest tests -m gpu -q -p no:cacheprovider
tail -4 gpurun_out/r05_def_full.log
