P=$GRAFT_REPO_ROOT/vadc_amd/csrc/build/prev/libvadc_amd_prev.so
for i in 1 2; do
  echo "--- prev"; VADC_AMD_LIB=$P timeout -k 10 100 python tools/l1_rate.py 24576 10 0 || exit 1
  VADC_AMD_LIB=$P timeout -k 10 200 python bench.py --no-side-config --no-host-fed --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernels_ms'])" || exit 1
  echo "--- new"; timeout -k 10 100 python tools/l1_rate.py 24576 10 0 || exit 1
  timeout -k 10 200 python bench.py --no-side-config --no-host-fed --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernels_ms'])" || exit 1
done
