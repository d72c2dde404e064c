#!/bin/bash
# cycle stamps inside k_layer1_regs (one wave per workgroup): share of a chunk's time spent waiting for its DMA, in the normalization offset, the conv
# block, the transformer block and the store.  gpurun -- 'bash tools/l1r_phases.sh'   (a variant of the library with -DVADC_L1R_PHASE_PROF beside the product: tools/abl_build.sh)
cd "$(dirname "$0")/.." || exit 1
export VADC_AMD_LIB=$(bash tools/abl_build.sh kernels_layer1_regs.hip -DVADC_L1R_PHASE_PROF | tail -1)
python - <<'PY'
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from vadc_amd.engine import Engine
from vadc_amd import _lib
blob = open("tests/golden/reference_fixtures/silero_v31_16k.testtensor", "rb").read()
n = 24576
e = Engine(blob, max_streams=256, max_chunks_per_call=96, device=0)
x = (np.random.default_rng(1).standard_normal((n, 129, 25)) * 2.0).astype(np.float32)
L = C.CDLL(os.environ["VADC_AMD_LIB"])
out = (C.c_ulonglong * 8)()
e.stage_from_stage(x, "normalized", "layer1")
L.vadc_amd_debug_l1r_phases(out, 1)
for _ in range(3): e.stage_from_stage(x, "normalized", "layer1")
L.vadc_amd_debug_l1r_phases(out, 1)
v = np.array(list(out)[:6], dtype=np.float64)
names = ["loop top (store drain)", "wait for the DMA", "normalization offset", "conv block (+ DMA issue)", "transformer block", "store"]
for nm, c in zip(names, v): print(f"{nm:28s} {c / v.sum() * 100:5.1f} %   {c / (3 * 256 * 12):9.0f} cycles per chunk")
e.close()
PY
