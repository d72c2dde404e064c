"""Cycles per phase of k_frontend_gemm (a -DVADC_PHASE_PROF build of kernels_frontend_gemm.hip: tools/abl_build.sh), per workgroup iteration:
    export VADC_AMD_LIB=$(bash tools/abl_build.sh kernels_frontend_gemm.hip -DVADC_PHASE_PROF | tail -1); python tools/gemm_phase_report.py [v31|v4] [streams] [chunks]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vadc_amd import synth, _lib
from vadc_amd.engine import Engine
model = sys.argv[1] if len(sys.argv) > 1 else "v31"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
C = int(sys.argv[3]) if len(sys.argv) > 3 else 64
path = "tests/golden/silero_v4_16k.testtensor" if model == "v4" else "tests/golden/reference_fixtures/silero_v31_16k.testtensor"
blob = open(path, "rb").read()
e = Engine(blob, max_streams=S, max_chunks_per_call=C, device=0, precision=0 if model == "v4" else 2)
pcm = np.ascontiguousarray(np.tile(synth.make_streams(16, C, seed0=1), (S // 16, 1)))
for i in range(5): e.run(pcm)
_lib.load().vadc_gemm_phase_report()
e.close()
