#!/bin/bash
# A/B of one engine option on one box, alternating, same bench arguments:   gpurun -- 'bash tools/ab_opt.sh fe_xcd=0 [bench.py arguments]'
# "A" runs with --opt $1, "B" with the defaults.
OPT=$1; shift
ARGS="${@:---no-side-config --no-host-fed --no-cpu-baseline}"
for i in 1 2 3; do
  for which in A B; do
    if [ $which = A ]; then X="--opt $OPT"; else X=""; fi
    echo -n "$which ($X): "
    timeout -k 10 300 python bench.py $ARGS $X 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernels_ms'])" || exit 1
  done
done
