import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np
from vadc_amd import synth
from vadc_amd.engine import Engine
from oracle import oracle as O
g = np.load("tests/golden/python_reference_v4_long_windows.npz")
blob4 = open("tests/golden/silero_v4_16k.testtensor", "rb").read()
orc = O.OracleV4(blob4)
base = synth.make_streams(16, 400, seed0=52000)
for w in (832, 896, 960, 1024):
    n = 640 if w < 1024 else 560
    e = Engine(blob4, max_streams=2, max_chunks_per_call=80, device=0); e.set_window(w)
    pcm = np.ascontiguousarray(base[[2, 12], : n * w])
    got = np.concatenate([e.run(pcm[:, i * w:(i + 80) * w]) for i in range(0, n, 80)], axis=1)[:, :, 1]
    oc = orc.forward_streams(pcm, window=w)
    for j, s in enumerate((2, 12)):
        ref = g[f"probs64_w{w}_s{s}"]
        print(w, s, "engine vs float64 reference: max %.3e  oracle vs reference: max %.3e  engine vs oracle: max %.3e" % (np.abs(got[j] - ref).max(), np.abs(oc[j] - ref).max(), np.abs(got[j] - oc[j]).max()), flush=True)
    e.close()
