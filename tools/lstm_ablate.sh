#!/bin/bash
# timing-only ablations of k_lstm_layer (results are wrong).  gpurun -- 'bash tools/lstm_ablate.sh'
cd "$(dirname "$0")/.." || exit 1
build() { (cd vadc_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c kernels_lstm.hip -o build/kernels_lstm.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libvadc_amd.so build/engine.o build/kernels_frontend.o build/kernels_frontend_gemm.o build/kernels_encoder_mfma.o build/kernels_encoder_fused.o build/kernels_lstm.o build/kernels_v5.o); }
for v in "" "-DVADC_LSTM_ABL_NOXLOAD" "-DVADC_LSTM_ABL_NOMFMA" "-DVADC_LSTM_ABL_NOGATES" "-DVADC_LSTM_ABL_NOXLOAD -DVADC_LSTM_ABL_NOMFMA -DVADC_LSTM_ABL_NOGATES"; do
   build $v
   echo "== $v"; python tools/lstm_rate.py 256 96 5 2>&1 | grep lstm=7
done
build
