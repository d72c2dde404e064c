#!/bin/bash
# The bench lines kept under profiles/rNN (one JSON line each).  Run through gpurun from the repo root; outputs land in gpurun_out/.
O=gpurun_out
python bench.py --details $O/bench_default_details.json 2>/dev/null | tail -1 > $O/bench_default.json
python bench.py --steps 20 --warmup 5 --details $O/bench_driver_form_details.json 2>/dev/null | tail -1 > $O/bench_driver_form.json
python bench.py --no-graph --no-cpu-baseline --no-side-config --details $O/bench_256x96_eager_details.json 2>/dev/null | tail -1 > $O/bench_256x96_eager.json
python bench.py --streams 4096 --chunks-per-step 16 --no-cpu-baseline --details $O/bench_4096x16_details.json 2>/dev/null | tail -1 > $O/bench_4096x16.json
python bench.py --streams 4096 --chunks-per-step 16 --no-cpu-baseline --no-graph --details $O/bench_4096x16_eager_details.json 2>/dev/null | tail -1 > $O/bench_4096x16_eager.json
python bench.py --precision split16 --streams 4096 --chunks-per-step 16 --no-cpu-baseline --details $O/bench_split16_4096x16_details.json 2>/dev/null | tail -1 > $O/bench_split16_4096x16.json
python bench.py --precision fast_stft --streams 4096 --chunks-per-step 16 --no-cpu-baseline --details $O/bench_fast_stft_4096x16_details.json 2>/dev/null | tail -1 > $O/bench_fast_stft_4096x16.json
python bench.py --model v4 --no-cpu-baseline --no-side-config --details $O/bench_v4_256x96_details.json 2>/dev/null | tail -1 > $O/bench_v4_256x96.json
python bench.py --model v4 --streams 4096 --chunks-per-step 16 --no-cpu-baseline --details $O/bench_v4_4096x16_details.json 2>/dev/null | tail -1 > $O/bench_v4_4096x16.json
python tests/reports/parity_report.py > $O/parity_report.log 2>&1
tail -1 $O/parity_report.log
python bench.py --streams 10240 --chunks-per-step 1 --no-cpu-baseline --no-side-config --details $O/bench_10240x1_details.json 2>/dev/null | tail -1 > $O/bench_10240x1.json
python tools/sweep_streams.py > $O/sweep_streams.log 2>&1
