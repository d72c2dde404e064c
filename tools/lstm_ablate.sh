#!/bin/bash
# timing-only ablations of k_lstm_layer (results are wrong).  gpurun -- 'bash tools/lstm_ablate.sh'
cd "$(dirname "$0")/.." || exit 1
for v in "" "-DVADC_LSTM_ABL_NOXLOAD" "-DVADC_LSTM_ABL_NOMFMA" "-DVADC_LSTM_ABL_NOGATES" "-DVADC_LSTM_ABL_NOXLOAD -DVADC_LSTM_ABL_NOMFMA -DVADC_LSTM_ABL_NOGATES"; do
   export VADC_AMD_LIB=$(bash tools/abl_build.sh kernels_lstm.hip $v | tail -1)
   echo "== $v"; python tools/lstm_rate.py 256 96 5 2>&1 | grep lstm=7
done
