#!/bin/bash
# PMC passes over the Silero v5 encoder (tools/v5_rate.py): where a wave's cycles go.  gpurun --timeout 600 -- 'bash tools/v5_pmc.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
rm -rf $O/v5pmcA $O/v5pmcB $O/v5pmcC $O/v5pmcD $O/v5kt
B="python3 tools/v5_rate.py --steps 3 --warmup 1 $V5_SHAPE"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/v5kt -- python3 tools/v5_rate.py --steps 40 --warmup 5 $V5_SHAPE > $O/v5kt.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $O/v5pmcA -- $B > $O/v5pmcA.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/v5pmcB -- $B > $O/v5pmcB.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM --output-format csv -d $O/v5pmcC -- $B > $O/v5pmcC.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/v5pmcD -- $B > $O/v5pmcD.log 2>&1
rm -rf $O/v5pmcE
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/v5pmcE -- $B > $O/v5pmcE.log 2>&1
# (a pass with TCC_HIT_sum / TCC_MISS_sum / TCP_TCC_READ_REQ_sum beside FETCH_SIZE never came back on this pool: 7 GPU-minutes until the silence guard; not repeated)
python3 - <<'PY'
import csv, glob, collections, json, shutil
summary = {"workload": "Silero v5 shapes, tools/v5_rate.py (default 256 streams x 288 windows per call)", "counters_per_launch": {}, "note": "SQ_* cycle counters are in quad-cycles; FETCH_SIZE / WRITE_SIZE in KB from separate passes: hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE counts 64 B per 128-B request, MI355X_MICROARCH.md section HBM)"}
for d in ("v5pmcA", "v5pmcB", "v5pmcC", "v5pmcD", "v5pmcE"):
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_v5_" not in k: continue
            acc[k[:40]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k[:40], r["Counter_Name"])] += 1
        for k in acc:
            print(d, k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
            summary["counters_per_launch"].setdefault(k, {}).update({c: round(v / n[(k, c)]) for c, v in acc[k].items()})
for f in glob.glob("gpurun_out/v5kt/**/*kernel_stats.csv", recursive=True):
    print(open(f).read())
    shutil.copy(f, "gpurun_out/v5_kernel_stats.csv")
json.dump(summary, open("gpurun_out/v5_pmc.json", "w"), indent=1)
PY
