#!/bin/bash
# PMC-compute passes (instruction mix, busy / wait counters, LDS) for the two 4096-stream workloads -- BASELINE config 4 (Silero v4) and config 3 (v3.1 SPLIT16) --
# so that every BASELINE configuration has trace + traffic + compute under profiles/rNN.  Run through gpurun from the repo root:
#   gpurun --timeout 1100 -- 'bash tools/refresh_pmc_compute_4096.sh'
# then reduce with tools/rocprof_reduce.py --pmc gpurun_out/pmcA_<tag> ... (one counter group per pass; no tracing beside --pmc).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
NB="--no-cpu-baseline --no-host-fed --no-side-config"
B="python3 bench.py --steps 3 --warmup 1 $NB"
for w in "v4 fp32" "v31 split16"; do
   set -- $w
   T=$1_$2
   W="--model $1 --precision $2 --streams 4096 --chunks-per-step 16"
   rm -rf $O/pmcA_$T $O/pmcB_$T $O/pmcC_$T $O/pmcD_$T
   rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 --output-format csv -d $O/pmcA_$T -- $B $W > $O/pmcA_$T.log 2>&1 &&
   rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmcB_$T -- $B $W > $O/pmcB_$T.log 2>&1 &&
   rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $O/pmcC_$T -- $B $W > $O/pmcC_$T.log 2>&1 &&
   rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/pmcD_$T -- $B $W > $O/pmcD_$T.log 2>&1 || exit 1
   echo "$w compute passes done"
done
