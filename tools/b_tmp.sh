python -m pytest tests -m gpu -x -q -k "split16" 2>&1 | tail -2
for a in "--precision split16 --streams 4096 --chunks-per-step 16" "--precision split16"; do
python bench.py --no-cpu-baseline $a 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'], {k:v['ms_per_launch'] for k,v in d['kernels'].items()})"
done
