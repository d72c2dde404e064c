import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vadc_amd import synth, _lib
from vadc_amd.engine import Engine
blob = open("tests/golden/reference_fixtures/silero_v31_16k.testtensor", "rb").read()
S, C = 256, 64
e = Engine(blob, max_streams=S, max_chunks_per_call=C, device=0, precision=2)
pcm = np.ascontiguousarray(np.tile(synth.make_streams(16, C, seed0=1), (S // 16, 1)))
for i in range(5): e.run(pcm)
_lib.load().vadc_gemm_phase_report()
e.close()
