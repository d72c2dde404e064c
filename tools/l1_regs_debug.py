"""k_layer1_regs (option "layer1" = 0) against the K = 1 fp32-MFMA form of k_layer_mfma (option "layer1" = 1): the layer-1 output of whole chunks and the
three op-level taps -- a bring-up aid, no oracle involved.  Usage: l1_regs_debug.py [chunks] [dump]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vadc_amd import synth
from vadc_amd.engine import Engine
blob = open("tests/golden/reference_fixtures/silero_v31_16k.testtensor", "rb").read()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
x = (synth.make_streams(1, n, seed0=3).astype(np.float32) / 32768.0).reshape(-1)
e = Engine(blob, max_streams=4, max_chunks_per_call=max(64, n), device=0)
e.set_option("layer1", 1); ref = e.stage_from_samples(x, "layer1")
e.set_option("layer1", 0); got = e.stage_from_samples(x, "layer1")
d = np.abs(got - ref)
print("layer1", got.shape, "max |regs - mfma| =", float(d.max()), "ref max", float(np.abs(ref).max()))
if d.max() > 1e-4 or len(sys.argv) > 2:
    np.set_printoptions(precision=3, suppress=True, linewidth=220)
    print("   per-chunk max err", d.reshape(d.shape[0], -1).max(axis=1))
    print("   chunk0 per-step max err", d[0].max(axis=0))
    print("   chunk0 per-channel max err", d[0].max(axis=1))
    print("ref chunk0\n", ref[0]); print("got chunk0\n", got[0])
rng = np.random.default_rng(5)
y = np.maximum(rng.standard_normal((n, 16, 25)).astype(np.float32), 0)
for what in ("layer_norm", "attention", "transformer_block"):
    e.set_option("layer1", 1); r = e.layer1_block(y, what)
    e.set_option("layer1", 0); g = e.layer1_block(y, what)
    dd = np.abs(g - r)
    print(what, "max |regs - mfma| =", float(dd.max()), "ref max", float(np.abs(r).max()))
    if dd.max() > 1e-4:
        np.set_printoptions(precision=3, suppress=True, linewidth=220)
        print("   chunk0 per-step max err", dd[0].max(axis=0))
        print("   chunk0 per-channel max err", dd[0].max(axis=1))
e.close()
