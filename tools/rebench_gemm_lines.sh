#!/bin/bash
# the bench lines of the GEMM-front-end configurations, side by side.   gpurun -- 'bash tools/rebench_gemm_lines.sh'
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
O=gpurun_out
python bench.py --details $O/bench_default_details.json 2>/dev/null | tail -1 > $O/bench_default.json
python bench.py --steps 20 --warmup 5 --details $O/bench_driver_form_details.json 2>/dev/null | tail -1 > $O/bench_driver_form.json
python bench.py --precision fast_stft --streams 4096 --chunks-per-step 16 --no-cpu-baseline --details $O/bench_fast_stft_4096x16_details.json 2>/dev/null | tail -1 > $O/bench_fast_stft_4096x16.json
python bench.py --model v4 --no-cpu-baseline --no-side-config --details $O/bench_v4_256x96_details.json 2>/dev/null | tail -1 > $O/bench_v4_256x96.json
python bench.py --model v4 --streams 4096 --chunks-per-step 16 --no-cpu-baseline --details $O/bench_v4_4096x16_details.json 2>/dev/null | tail -1 > $O/bench_v4_4096x16.json
python -c "
import json
for f in ('bench_default','bench_driver_form','bench_v4_4096x16','bench_v4_256x96','bench_fast_stft_4096x16'):
    d=json.loads(open('gpurun_out/'+f+'.json').read())
    print(f, d['value'], d['config'].get('frontend_kernel'), d['roofline'].get('traffic_over_algorithmic'), {k:(v['value'],v.get('frontend_kernel')) for k,v in d.get('configs',{}).items()})
"
