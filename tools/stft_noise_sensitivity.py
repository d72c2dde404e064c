"""How far do the probabilities move when the STFT is computed EXACTLY (float64, rounded once to fp32) instead of with the reference's fp32 reduction tree?
The engine's own stage taps do both: (a) samples -> magnitude tap (the tree, bit-exact to the reference) -> layers -> recurrence; (b) float64 magnitudes -> the same
layers -> the same recurrence.  If (b) is already further from (a) than the 1e-4 bar, no front end that is merely ACCURATE (a GEMM, however precise) can be a parity mode
against the C backend: only one that reproduces the reference's rounding can.        python tools/stft_noise_sensitivity.py [streams=16] [chunks=400]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vadc_amd.engine import Engine
from vadc_amd import synth, testtensor as tt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = os.path.join(ROOT, "tests", "golden", "reference_fixtures", "silero_v31_16k.testtensor")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
Cn = int(sys.argv[2]) if len(sys.argv) > 2 else 400
basis = [a for _, a in tt.load(W) if a.size == 258 * 256][0].reshape(258, 256).astype(np.float64)
pcm = synth.make_streams(S, Cn, seed0=9000)
x = pcm.astype(np.float32) / np.float32(32768)
e = Engine(open(W, "rb").read(), max_streams=S, max_chunks_per_call=Cn, device=0)
res = {}
for form in ("tree", "float64", "float64_of_fp32_products"):
    enc = np.empty((S, Cn, 64, 7), np.float32)
    for s in range(S):
        ch = x[s].reshape(Cn, 1536)
        if form == "tree":
            mag = e.stage_from_samples(ch, "magnitude")
        else:
            pad = np.concatenate([ch[:, 128:0:-1], ch, ch[:, -2:-130:-1]], axis=1).astype(np.float64)         # reflect pad 128 either side (stft.c)
            fr = np.lib.stride_tricks.sliding_window_view(pad, 256, axis=1)[:, ::64][:, :25]                  # [Cn, 25, 256]
            conv = np.einsum("ctk,bk->cbt", fr, basis)
            mag = np.sqrt(conv[:, :129] ** 2 + conv[:, 129:] ** 2).astype(np.float32)
        enc[s] = e.stage_from_stage(mag, "magnitude", "layer4")
    e.reset_streams()
    res[form] = e.lstm_decoder(enc)[:, :, 1]
    if form == "float64":
        break
e.close()
d = np.abs(res["float64"] - res["tree"])
print(f"{S} streams x {Cn} chunks: |p(float64 STFT) - p(reference's fp32 tree)|  max {d.max():.3e}  p99.9 {np.quantile(d, 0.999):.3e}  mean {d.mean():.3e}   (bar: 1e-4)")
