# same-box A/B of the unsplit FP8 weight streamer's (phase length, consumer waves): SGL_MI355_WSTREAM_FORCE="PH,nc"
for f in none 4,4 4,5 4,6 8,4; do
  if [ $f = none ]; then unset SGL_MI355_WSTREAM_FORCE; else export SGL_MI355_WSTREAM_FORCE=$f; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); g=d['roofline_gemm']
print('$f', d['ms_per_step'], [(s['name'], s['decode']['us']) for s in g['shapes']])"
done
