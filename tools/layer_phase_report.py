import ctypes, subprocess, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vadc_amd import synth, _lib
from vadc_amd.engine import Engine
v4 = len(sys.argv) > 1 and sys.argv[1] == "v4"          # python tools/layer_phase_report.py [v4]
blob = open("tests/golden/silero_v4_16k.testtensor" if v4 else "tests/golden/reference_fixtures/silero_v31_16k.testtensor", "rb").read()
S, C = 256, 64
e = Engine(blob, max_streams=S, max_chunks_per_call=C, device=0)
pcm = synth.make_streams(16, C, seed0=1)
pcm = np.ascontiguousarray(np.tile(pcm, (S // 16, 1)))
for i in range(5): e.run(pcm)
L = _lib.load()
L.vadc_phase_report()
e.close()
