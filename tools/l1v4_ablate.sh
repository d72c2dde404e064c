#!/bin/bash
# timing-only ablations of k_layer1_regs_v4 (results are wrong): the upper bound of what pre-split projection operands / a folded normalization offset / a magnitude
# array from the front end could buy the Silero v4 first stage, inside the step at 4096 x 16.   gpurun -- 'bash tools/l1v4_ablate.sh'
cd "$(dirname "$0")/.." || exit 1
set -e
export VADC_AMD_ABL_SKIP_L1_SELFCHECK=1      # the ablated kernels fail the create-time self-check by construction: keep them anyway
for v in "" "-DVADC_L1V4_ABL_XPRESPLIT" "-DVADC_L1V4_ABL_NOOFF" "-DVADC_L1V4_ABL_XPRESPLIT -DVADC_L1V4_ABL_NOOFF" "-DVADC_L1V4_ABL_NOEXP" "-DVADC_L1V4_ABL_XPRESPLIT -DVADC_L1V4_ABL_NOOFF -DVADC_L1V4_ABL_NOEXP" "-DVADC_ENC_ABL_NOSPLIT" "-DVADC_ENC_ABL_NOMFMA"; do
   export VADC_AMD_LIB=$(bash tools/abl_build.sh kernels_layer1_regs_v4.hip $v | tail -1)
   echo "== ${v:-product}"
   python bench.py --model v4 --streams 4096 --chunks-per-step 16 --steps 60 --warmup 10 --no-cpu-baseline --no-host-fed --no-side-config 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value', d['value'], 'ms_per_step', d['ms_per_step'], 'kernels_ms', d['kernels_ms'])"
done
