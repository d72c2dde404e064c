#!/bin/bash
# Re-create everything under profiles/rNN from one GPU box.  Run through gpurun from the repo root:
#   gpurun --timeout 1100 -- 'bash tools/refresh_profiles.sh'      (outputs land in gpurun_out/, then: python tools/rocprof_reduce.py ...)
# rocprofv3 rules on this pool: program directly after `--`, PMC passes separate from tracing, one counter group per pass.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
rm -rf $O/pmcA $O/pmcB $O/pmcC $O/pmcD
NB="--no-cpu-baseline --no-host-fed --no-side-config"
B="python3 bench.py --steps 3 --warmup 1 $NB"
if [ -z "$SKIP_HEADLINE" ]; then       # (SKIP_HEADLINE=1: tools/refresh_headline.sh has taken the headline's trace and traffic passes on the box its line came from)
rm -rf $O/prof_kt $O/prof_fetch $O/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -- python3 bench.py $NB > $O/prof_kt.log 2>&1
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- $B > $O/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- $B > $O/prof_write.log 2>&1
echo "traffic passes done"
fi
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 --output-format csv -d $O/pmcA -- $B > $O/pmcA.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmcB -- $B > $O/pmcB.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $O/pmcC -- $B > $O/pmcC.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/pmcD -- $B > $O/pmcD.log 2>&1
echo "compute passes done"
# HBM traffic of the 4096-stream workloads (BASELINE configs 3 and 4)
for w in "v4 fp32" "v31 split16"; do
   set -- $w
   W="--model $1 --precision $2 --streams 4096 --chunks-per-step 16"
   rm -rf $O/prof_fetch_$1_$2 $O/prof_write_$1_$2 $O/prof_kt_$1_$2
   rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch_$1_$2 -- $B $W > $O/prof_fetch_$1_$2.log 2>&1
   rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write_$1_$2 -- $B $W > $O/prof_write_$1_$2.log 2>&1
   rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt_$1_$2 -- python3 bench.py $NB $W > $O/prof_kt_$1_$2.log 2>&1
   echo "$w done"
done
# the driver's form (20 timed steps after 5 warm-up steps): kernel trace for tools/trace_timeline.py
rm -rf $O/prof_kt_driver
rocprofv3 --kernel-trace --output-format csv -d $O/prof_kt_driver -- python3 bench.py --steps 20 --warmup 5 $NB > $O/prof_kt_driver.log 2>&1
python3 tools/trace_timeline.py $O/prof_kt_driver 20 > $O/driver_form_timeline.txt 2>&1
echo "driver-form trace done"
