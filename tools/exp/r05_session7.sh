cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
step r05_t7_pytest 900 python -m pytest tests -m gpu -q -p no:cacheprovider
tail -4 gpurun_out/r05_t7_pytest.log
step r05_prof_bench 900 bash tools/exp/prof_bench_r05.sh
tail -60 gpurun_out/r05_prof_bench.log
