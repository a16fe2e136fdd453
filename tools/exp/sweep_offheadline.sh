run() { timeout -k 10 300 python bench.py "$@" --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import json, sys
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '->', d['ms_per_step'], 'ms', d['value'], 'tok/s')
except Exception as e:
    print('$*', '-> FAILED', e)"; }
run --batch 48
run --batch 65
run --batch 100
run --model llama2-7b --quant awq --batch 128
run --model llama2-7b --quant awq --batch 32
run --quant none --batch 128
run --model llama3-70b --emulate-tp 8 --batch 128
run --model llama3-70b --emulate-tp 8 --batch 16
run --model qwen2-0.5b --quant none --batch 64
run --batch 64 --ctx 128
run --batch 64 --ctx 4096 --kv-dtype fp8_e5m2
