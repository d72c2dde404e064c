#!/bin/bash
# timing-only ablations of k_layer1's channel loop (results are wrong).  gpurun -- 'bash tools/l1_ablate.sh'
cd "$(dirname "$0")/.." || exit 1
for v in "" "-DVADC_L1_ABL_NOMFMA" "-DVADC_L1_ABL_NODW" "-DVADC_L1_ABL_NOMFMA -DVADC_L1_ABL_NODW" "-DVADC_L1_ABL_NOLOAD" "-DVADC_L1_ABL_NOTF" "-DVADC_L1_ABL_NOTF -DVADC_L1_ABL_NOLOAD" "-DVADC_L1_ABL_NOTF -DVADC_L1_ABL_NOMFMA -DVADC_L1_ABL_NODW" "-DVADC_L1_ABL_NOTF -DVADC_L1_ABL_NOMFMA -DVADC_L1_ABL_NODW -DVADC_L1_ABL_NOLOAD"; do
   export VADC_AMD_LIB=$(bash tools/abl_build.sh kernels_encoder_mfma.hip $v | tail -1)
   echo "== $v"; python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-host-fed --no-side-config 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['kernels_ms'])"
done
