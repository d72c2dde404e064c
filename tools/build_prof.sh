#!/bin/bash
# build libvadc_amd.so with extra defines for kernels_encoder_mfma.hip (e.g. -DVADC_PHASE_PROF); usage: tools/build_prof.sh [defines...]
set -e
cd "$(dirname "$0")/../vadc_amd/csrc"
make -j4 > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c kernels_encoder_mfma.hip -o build/kernels_encoder_mfma.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c kernels_frontend_gemm.hip -o build/kernels_frontend_gemm.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libvadc_amd.so build/engine.o build/kernels_frontend.o build/kernels_frontend_gemm.o build/kernels_encoder_mfma.o build/kernels_lstm.o build/kernels_v5.o
touch kernels_encoder_mfma.hip kernels_frontend_gemm.hip   # so that a plain make rebuilds the product object
