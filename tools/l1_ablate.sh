#!/bin/bash
# timing-only ablations of k_layer1's channel loop (results are wrong).  gpurun -- 'bash tools/l1_ablate.sh'
cd "$(dirname "$0")/.." || exit 1
build() { (cd vadc_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c kernels_encoder_mfma.hip -o build/kernels_encoder_mfma.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libvadc_amd.so build/engine.o build/kernels_frontend.o build/kernels_frontend_gemm.o build/kernels_encoder_mfma.o build/kernels_encoder_fused.o build/kernels_lstm.o build/kernels_v5.o); }
for v in "" "-DVADC_L1_ABL_NOMFMA" "-DVADC_L1_ABL_NODW" "-DVADC_L1_ABL_NOMFMA -DVADC_L1_ABL_NODW" "-DVADC_L1_ABL_NOLOAD" "-DVADC_L1_ABL_NOTF" "-DVADC_L1_ABL_NOTF -DVADC_L1_ABL_NOLOAD" "-DVADC_L1_ABL_NOTF -DVADC_L1_ABL_NOMFMA -DVADC_L1_ABL_NODW" "-DVADC_L1_ABL_NOTF -DVADC_L1_ABL_NOMFMA -DVADC_L1_ABL_NODW -DVADC_L1_ABL_NOLOAD"; do
   build $v
   echo "== $v"; python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-host-fed --no-side-config 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['kernels_ms'])"
done
build
