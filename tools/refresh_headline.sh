#!/bin/bash
# The headline line and the rocprofv3 summaries it is checked against, from ONE box (boxes of the pool differ by 5 - 7 %):
#   gpurun --timeout 900 -- 'bash tools/refresh_headline.sh'
# then: python tools/rocprof_reduce.py --kernel-trace gpurun_out/prof_kt --fetch gpurun_out/prof_fetch --write gpurun_out/prof_write --out profiles/rNN --tag bench_256x96
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
rm -rf $O/prof_kt $O/prof_fetch $O/prof_write
python bench.py --details $O/bench_default_details.json 2>/dev/null | tail -1 > $O/bench_default.json || exit 1
python bench.py --steps 20 --warmup 5 --details $O/bench_driver_form_details.json 2>/dev/null | tail -1 > $O/bench_driver_form.json || exit 1
NB="--no-cpu-baseline --no-host-fed --no-side-config"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -- python3 bench.py $NB > $O/prof_kt.log 2>&1 || exit 1
B="python3 bench.py --steps 3 --warmup 1 $NB"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- $B > $O/prof_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- $B > $O/prof_write.log 2>&1 || exit 1
python -c "import json; d=json.loads(open('$O/bench_default.json').read()); print('default', d['value'], d['ms_per_step'], d['kernels_ms'])"
python -c "import json; d=json.loads(open('$O/bench_driver_form.json').read()); print('driver form', d['value'], d['ms_per_step'])"
grep -h "k_frontend_sym\|k_layer1_regs<12\|k_enc_fused" $O/prof_kt/*/*_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,150-
