"""Soak: 272 streams (17 tiles: a ragged last tile) x 96 chunks x 120 synchronous calls with the state carried on the device; three streams (first, middle, last) are
recomputed by the oracle over all 11,520 chunks.  python tests/reports/soak_report.py   (GPU box; ~2 minutes, most of it the oracle)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from vadc_amd import synth
from vadc_amd.engine import Engine
from oracle import oracle as O
blob = open("tests/golden/reference_fixtures/silero_v31_16k.testtensor", "rb").read()
S, Cn, calls = 272, 96, 120
pcm = synth.make_streams(S, Cn * 4, seed0=99)            # 4 distinct windows per stream, cycled
e = Engine(blob, max_streams=S, max_chunks_per_call=Cn, device=0)
outs = []
t = time.time()
for k in range(calls):
    w = k % 4
    outs.append(e.run(pcm[:, w * Cn * 1536:(w + 1) * Cn * 1536])[[0, 137, 271]])
print("gpu calls done in %.1f s" % (time.time() - t))
e.close()
got = np.concatenate(outs, axis=1)                       # [3, calls*Cn, 2]
orc = O.Oracle(blob)
worst = 0.0
for j, s in enumerate((0, 137, 271)):
    seq = np.concatenate([pcm[s, (k % 4) * Cn * 1536:((k % 4) + 1) * Cn * 1536] for k in range(calls)])
    want = orc.forward_stream(seq)
    d = float(np.abs(got[j] - want).max())
    worst = max(worst, d)
    print("stream", s, "chunks", want.shape[0], "max |dp| =", d)
print("worst", worst, "OK" if worst <= 1e-4 else "FAIL")
