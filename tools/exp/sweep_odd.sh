# odd sizes, for crashes / cliffs rather than numbers
run() { timeout -k 10 300 python bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs 2>gpurun_out/odd_err.txt | python3 -c "
import json, sys
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '->', d['ms_per_step'], 'ms', d['value'], 'tok/s')
except Exception as e:
    print('$*', '-> FAILED', e); print(open('gpurun_out/odd_err.txt').read()[-600:])"; }
run --batch 33 --ctx 2000
run --batch 7 --ctx 4097
run --batch 3 --ctx 33
run --model llama2-7b --quant awq --batch 17 --ctx 3000
run --model llama3-70b --emulate-tp 8 --batch 65
run --model llama3-70b --emulate-tp 8 --batch 200
run --emulate-tp 4 --batch 96
run --emulate-tp 2 --batch 129 --kv-dtype fp8_e4m3
run --quant none --batch 200
run --model qwen2-0.5b --quant none --batch 130 --ctx 1000
run --batch 64 --ctx 2048 --no-graph
