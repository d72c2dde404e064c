#!/bin/bash
# timing-only ablations of k_layer1_regs (results are wrong): what the stage costs without its input DMA, its transformer block, the normalization offset, the MFMAs, the operand splits.  gpurun -- 'bash tools/l1r_ablate.sh'
cd "$(dirname "$0")/.." || exit 1
for v in "" "-DVADC_L1R_ABL_NODMA" "-DVADC_L1R_ABL_NOBLOCK" "-DVADC_L1R_ABL_NOBLOCK -DVADC_L1R_ABL_NODMA" "-DVADC_L1R_ABL_NONORM" "-DVADC_ENC_ABL_NOMFMA" "-DVADC_ENC_ABL_NOSPLIT" ${L1R_EXTRA}; do
   export VADC_AMD_LIB=$(bash tools/abl_build.sh kernels_layer1_regs.hip $v | tail -1)
   echo "== $v"; python tools/l1_rate.py 24576 5 0 2>&1 | grep "layer1=0"
done
