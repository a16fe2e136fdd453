cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/gpu_steps.sh
export SEED=${SEED:-77}
N=200 step r05_fz_extend 400 python tools/exp/fuzz_extend.py
N=200 step r05_fz_parts 400 python tools/exp/fuzz_extend_parts.py
N=250 step r05_fz_decode 400 python tools/exp/fuzz_decode.py
step r05_fz_gemm 400 python tools/exp/fuzz_gemm.py
step r05_fz_misc 400 python tools/exp/fuzz_misc.py
step r05_fz_g16 400 python tools/exp/fuzz_gemm16_tiled.py
step r05_fz_ar 400 python tools/exp/fuzz_allreduce.py
for n in extend parts decode gemm misc g16 ar; do tail -1 gpurun_out/r05_fz_$n.log; done
