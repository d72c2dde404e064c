cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for o in 1 2; do
  rm -rf gpurun_out/ab$o
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab$o -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --opt fe_nps=$o > gpurun_out/ab$o.log 2>&1
  tail -1 gpurun_out/ab$o.log | cut -c1-200
  f=$(find gpurun_out/ab$o -name "*kernel_stats.csv" | head -1)
  head -8 "$f" | cut -c1-200
done
