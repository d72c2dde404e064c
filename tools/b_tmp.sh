python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for a in "" "--opt fe_overlap=1" "--streams 4096 --chunks-per-step 16 --opt fe_overlap=1" "--precision split16 --opt fe_overlap=1" "--precision split16 --streams 4096 --chunks-per-step 16 --opt fe_overlap=1" "--model v4" "--model v4 --opt fe_overlap=0"; do
python bench.py --no-cpu-baseline $a 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'], {k:v['ms_per_launch'] for k,v in d['kernels'].items()})"
done
