for o in 0 6 8 7; do timeout -k 10 150 python bench.py --streams 10240 --chunks-per-step 1 --steps 300 --warmup 20 --no-cpu-baseline --no-host-fed --no-side-config --opt lstm=$o --details gpurun_out/bench_ns_$o.json 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lstm', $o, d['value'], d['ms_per_step'])"; python - <<PY
import json
d=json.load(open('gpurun_out/bench_ns_$o.json'))
print({k:v['ms_per_launch'] for k,v in d['kernels'].items()})
PY
done
