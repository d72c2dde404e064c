#!/bin/bash
# k_layer1 compiled for 5 / 6 waves per SIMD (= workgroups per CU) and input-ring depths.  gpurun -- 'bash tools/l1_occupancy.sh'
cd "$(dirname "$0")/.." || exit 1
for v in "-DVADC_L1_WAVES=5" "-DVADC_L1_WAVES=6 -DVADC_L1_XR=16" "-DVADC_L1_WAVES=6 -DVADC_L1_XR=8" "-DVADC_L1_WAVES=6 -DVADC_L1_XR=24"; do
   export VADC_AMD_LIB=$(bash tools/abl_build.sh kernels_encoder_mfma.hip $v | tail -1)
   echo "== $v"; python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-host-fed --no-side-config 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['value'], d['kernels_ms'])"
done
