#!/bin/bash
# what rows of Y on 128-byte boundaries (pitch 32 floats instead of 25) would buy the front end.  Time: tools/fe_bench FE_POWER (2,000 launches per
# variant, four rounds in turn); bytes written: a rocprofv3 --pmc WRITE_SIZE pass per variant.   gpurun --timeout 600 -- 'bash tools/y_pitch_probe.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
FE_POWER=1 tools/fe_bench 24576 sym - - mask32 2>&1 | grep -E "round [0-9]" | tee $O/y_pitch_time.log
for v in OPT3 YP32; do
   rm -rf $O/ypitch_$v
   FE_POWER=1 FE_ONLY=$v rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/ypitch_$v -- tools/fe_bench 24576 sym - - mask32 > $O/ypitch_$v.log 2>&1
done
python3 - <<'PY'
import csv, glob
for v in ("OPT3", "YP32"):
    tot, n = 0.0, 0
    for f in glob.glob(f"gpurun_out/ypitch_{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_frontend_sym" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE":
                tot += float(r["Counter_Value"]); n += 1
    if n:
        print(f"{v}: WRITE_SIZE {tot / n * 1024 / 1e6:.1f} MB per launch over {n} launches (Y = {24576 * 129 * 25 * 4 / 1e6:.1f} MB, padded rows {24576 * 129 * 32 * 4 / 1e6:.1f} MB)")
PY
