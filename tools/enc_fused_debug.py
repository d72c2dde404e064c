"""Stage-by-stage comparison of k_enc_fused with the per-layer kernels (option "encoder" = 5) -- a bring-up aid, no oracle involved."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vadc_amd import synth
from vadc_amd.engine import Engine
blob = open("tests/golden/reference_fixtures/silero_v31_16k.testtensor", "rb").read()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
x = (synth.make_streams(1, n, seed0=3).astype(np.float32) / 32768.0).reshape(-1)
e = Engine(blob, max_streams=4, max_chunks_per_call=64, device=0)
for stage in ("layer2", "layer3", "layer4"):
    e.set_option("encoder", 5)
    ref = e.stage_from_samples(x, stage)
    e.set_option("encoder", 0)
    got = e.stage_from_samples(x, stage)
    d = np.abs(got - ref)
    print(stage, got.shape, "max |fused - per-layer| =", float(d.max()), "ref max", float(np.abs(ref).max()))
    if d.max() > 1e-4:
        i = np.unravel_index(np.argmax(d), d.shape)
        print("   worst at", i, got[i], ref[i])
        print("   per-chunk max err", d.reshape(d.shape[0], -1).max(axis=1))
        print("   chunk0 per-step max err", d[0].max(axis=0))
        print("   chunk0 per-channel max err", d[0].max(axis=1))
for a, b in (("layer1", "layer2"), ("layer2", "layer3"), ("layer3", "layer4"), ("layer2", "layer4")):
    e.set_option("encoder", 5)
    src = e.stage_from_samples(x, a)
    ref = e.stage_from_stage(src, a, b)
    e.set_option("encoder", 0)
    got = e.stage_from_stage(src, a, b)
    print(a, "->", b, "max |fused - per-layer| =", float(np.abs(got - ref).max()))
e.close()
if len(sys.argv) > 2:
    e = Engine(blob, max_streams=4, max_chunks_per_call=64, device=0)
    e.set_option("encoder", 5); ref = e.stage_from_samples(x, "layer2")
    e.set_option("encoder", 0); got = e.stage_from_samples(x, "layer2")
    np.set_printoptions(precision=2, suppress=True, linewidth=200)
    print("ref chunk0\n", ref[0]); print("got chunk0\n", got[0])
    e.close()
