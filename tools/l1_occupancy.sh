#!/bin/bash
# k_layer1 compiled for 5 / 6 waves per SIMD (= workgroups per CU) and input-ring depths.  gpurun -- 'bash tools/l1_occupancy.sh'
cd "$(dirname "$0")/.." || exit 1
build() { (cd vadc_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c kernels_encoder_mfma.hip -o build/kernels_encoder_mfma.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libvadc_amd.so build/engine.o build/kernels_frontend.o build/kernels_frontend_gemm.o build/kernels_encoder_mfma.o build/kernels_encoder_fused.o build/kernels_lstm.o build/kernels_v5.o); }
for v in "-DVADC_L1_WAVES=5" "-DVADC_L1_WAVES=6 -DVADC_L1_XR=16" "-DVADC_L1_WAVES=6 -DVADC_L1_XR=8" "-DVADC_L1_WAVES=6 -DVADC_L1_XR=24"; do
   build $v
   echo "== $v"; python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-host-fed --no-side-config 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['value'], d['kernels_ms'])"
done
build
