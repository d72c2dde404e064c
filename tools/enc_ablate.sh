#!/bin/bash
# timing-only ablations of k_enc_fused (results are wrong): no MFMA, no split.  gpurun -- 'bash tools/enc_ablate.sh'
cd "$(dirname "$0")/.." || exit 1
for v in "" "-DVADC_ENC_ABL_NOMFMA" "-DVADC_ENC_ABL_NOSPLIT" "-DVADC_ENC_ABL_NOMFMA -DVADC_ENC_ABL_NOSPLIT" "-fno-slp-vectorize"; do
   export VADC_AMD_LIB=$(bash tools/abl_build.sh kernels_encoder_fused.hip $v | tail -1)
   echo "== $v"; python tools/enc_rate.py 24576 10 2>&1 | grep encoder=0; python tools/enc_rate.py 6144 10 2>&1 | grep encoder=0
done
