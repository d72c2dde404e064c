"""k_layer1_regs_v4 (option "layer1" = 0) against k_layer_mfma's K = 1 form (option "layer1" = 1) on Silero v4's first stage -- a bring-up aid, no oracle
involved.  Usage: l1_regs_debug_v4.py [chunks]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vadc_amd import synth
from vadc_amd.engine import Engine
blob = open("tests/golden/silero_v4_16k.testtensor", "rb").read()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
x = (synth.make_streams(1, n, seed0=3).astype(np.float32) / 32768.0).reshape(-1)
e = Engine(blob, max_streams=4, max_chunks_per_call=max(64, n), device=0)
e.set_option("layer1", 1); ref = e.stage_from_samples(x, "layer1")
e.set_option("layer1", 0); got = e.stage_from_samples(x, "layer1")
d = np.abs(got - ref)
print("layer1", got.shape, "max |regs - mfma| =", float(d.max()), "ref max", float(np.abs(ref).max()))
if d.max() > 1e-4:
    np.set_printoptions(precision=3, suppress=True, linewidth=220)
    print("   per-chunk max err", d.reshape(d.shape[0], -1).max(axis=1))
    print("   chunk0 per-step max err", d[0].max(axis=0))
    print("   chunk0 per-channel max err", d[0].max(axis=1))
    print("ref chunk0\n", ref[0]); print("got chunk0\n", got[0])
for st in ("layer2", "layer4"):
    e.set_option("layer1", 1); r = e.stage_from_samples(x, st)
    e.set_option("layer1", 0); g = e.stage_from_samples(x, st)
    print(st, "max |regs - mfma| =", float(np.abs(g - r).max()), "ref max", float(np.abs(r).max()))
e.close()
