#!/bin/bash
# PMC passes over k_layer1_regs alone (tools/enc_rate.py): where a wave's cycles go.  gpurun --timeout 600 -- 'bash tools/l1r_pmc.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
rm -rf $O/l1rpmcA $O/l1rpmcB $O/l1rpmcC
B="python3 tools/l1_rate.py 24576 2 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $O/l1rpmcA -- $B > $O/l1rpmcA.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/l1rpmcB -- $B > $O/l1rpmcB.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC --output-format csv -d $O/l1rpmcC -- $B > $O/l1rpmcC.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("l1rpmcA", "l1rpmcB", "l1rpmcC"):
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_layer1_regs" not in k: continue
            acc[k[:40]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k[:40], r["Counter_Name"])] += 1
        for k in acc:
            print(d, k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
