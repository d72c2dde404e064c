#!/bin/bash
# timing-only ablations of k_layer1_regs (results are wrong): what the stage costs without its input DMA, its transformer block, the normalization offset, the MFMAs, the operand splits.  gpurun -- 'bash tools/l1r_ablate.sh'
cd "$(dirname "$0")/.." || exit 1
build() { (cd vadc_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c kernels_layer1_regs.hip -o build/kernels_layer1_regs.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libvadc_amd.so build/engine.o build/kernels_frontend.o build/kernels_frontend_gemm.o build/kernels_encoder_mfma.o build/kernels_encoder_fused.o build/kernels_layer1_regs.o build/kernels_lstm.o build/kernels_v5.o); }
for v in "" "-DVADC_L1R_ABL_NODMA" "-DVADC_L1R_ABL_NOBLOCK" "-DVADC_L1R_ABL_NOBLOCK -DVADC_L1R_ABL_NODMA" "-DVADC_L1R_ABL_NONORM" "-DVADC_ENC_ABL_NOMFMA" "-DVADC_ENC_ABL_NOSPLIT" ${L1R_EXTRA}; do
   build $v
   echo "== $v"; python tools/l1_rate.py 24576 5 0 2>&1 | grep "layer1=0"
done
