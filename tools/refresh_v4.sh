cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
NB="--no-cpu-baseline --no-host-fed --no-side-config"
B="python3 bench.py --steps 3 --warmup 1 $NB"
W="--model v4 --precision fp32 --streams 4096 --chunks-per-step 16"
rm -rf $O/prof_fetch_v4_fp32 $O/prof_write_v4_fp32 $O/prof_kt_v4_fp32
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch_v4_fp32 -- $B $W > $O/prof_fetch_v4_fp32.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write_v4_fp32 -- $B $W > $O/prof_write_v4_fp32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt_v4_fp32 -- python3 bench.py $NB $W > $O/prof_kt_v4_fp32.log 2>&1
python bench.py --model v4 --no-cpu-baseline --no-side-config --details $O/bench_v4_256x96_details.json 2>/dev/null | tail -1 > $O/bench_v4_256x96.json
python bench.py --model v4 --streams 4096 --chunks-per-step 16 --no-cpu-baseline --details $O/bench_v4_4096x16_details.json 2>/dev/null | tail -1 > $O/bench_v4_4096x16.json
echo done
