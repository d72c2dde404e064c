#!/bin/bash
# timing-only ablations of k_enc_fused (results are wrong): no MFMA, no split.  gpurun -- 'bash tools/enc_ablate.sh'
cd "$(dirname "$0")/.." || exit 1
build() { (cd vadc_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c kernels_encoder_fused.hip -o build/kernels_encoder_fused.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libvadc_amd.so build/engine.o build/kernels_frontend.o build/kernels_frontend_gemm.o build/kernels_encoder_mfma.o build/kernels_encoder_fused.o build/kernels_lstm.o build/kernels_v5.o); }
for v in "" "-DVADC_ENC_ABL_NOMFMA" "-DVADC_ENC_ABL_NOSPLIT" "-DVADC_ENC_ABL_NOMFMA -DVADC_ENC_ABL_NOSPLIT" "-fno-slp-vectorize"; do
   build $v
   echo "== $v"; python tools/enc_rate.py 24576 10 2>&1 | grep encoder=0; python tools/enc_rate.py 6144 10 2>&1 | grep encoder=0
done
