#!/bin/bash
# A/B on one box: the library built from an earlier commit (vadc_amd/csrc/build/prev/libvadc_amd_prev.so: git archive <commit> | tar -x -C /tmp/prev && make -C
# /tmp/prev/vadc_amd/csrc) against the tree's, alternating, same bench arguments.  gpurun -- 'bash tools/ab_prev.sh [bench.py arguments]'
P=$GRAFT_REPO_ROOT/vadc_amd/csrc/build/prev/libvadc_amd_prev.so
[ -f "$P" ] || { echo "ab_prev.sh: $P is missing (build the earlier commit's library there first): nothing to compare with" >&2; exit 1; }
ARGS="${@:---no-side-config --no-host-fed --no-cpu-baseline}"
for i in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then export VADC_AMD_LIB=$P; else unset VADC_AMD_LIB; fi
    echo -n "$which: "
    timeout -k 10 300 python bench.py $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernels_ms'])" || exit 1
  done
done
