#!/usr/bin/env python3
"""How much do the per-chunk probabilities move when the STFT is evaluated in a different fp32 summation order?

The reference sums its 256 products per output in a fixed fp32 tree (stft.c:115-184).  k_frontend reproduces that tree
bit for bit; the MFMA front end (k_frontend_gemm) cannot (v_mfma_f32_16x16x4_f32 accumulates internally, and the
real-input folding x[n] +- x[256-n] halves the taps).  log1p(2^20 |X|) amplifies differences in near-silent bins, so
this script measures the end effect on the OUTPUT with the CPU oracle: the oracle's exact path versus the oracle fed
with magnitudes from (a) a float64 STFT rounded once to fp32 (the "true" value both orders approximate) and (b) an fp32
folded evaluation with a different order.  Test infrastructure only (imports oracle/).
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from vadc_amd import synth
from vadc_amd.testtensor import load

W = os.path.join(ROOT, "tests", "golden", "reference_fixtures", "silero_v31_16k.testtensor")
ts = [a for _, a in load(W)]
basis = ts[0].reshape(258, 256)
SPEC = [(True, 2), (True, 2), (False, 1), (True, 1)]
lw, lb, dw, db = ts[95], ts[96], ts[97], ts[98]


def layers_for(t_in=25):
    out, idx = [], 1
    for has_proj, stride in SPEC:
        n = 24 if has_proj else 22
        L = O.make_layer(ts[idx:idx + n], has_proj, stride, t_in)
        out.append(L); idx += n; t_in = L.t_out
    return out

LAYERS = layers_for()


def from_magnitude(mag, h, c):
    """mag [129,25] fp32 -> prob, state carried (h, c [2,64])."""
    x = O.adaptive_norm(mag)                       # so_adaptive_norm takes magnitudes (log1p inside, misc.c:40-96)
    for L in LAYERS:
        x = O.transformer_layer(x, L)
    seq, h, c = O.lstm_seq(np.ascontiguousarray(x.T), lw, lb, h, c)
    p = O.decoder(np.ascontiguousarray(seq.T), dw, db)
    return p, h, c


def pad(x):
    return np.concatenate([x[128:0:-1], x, x[-2:-130:-1]])


def mag_f64(x):
    xp = pad(x.astype(np.float64))
    fr = np.stack([xp[64 * f:64 * f + 256] for f in range(25)], 1)       # [256,25]
    conv = basis.astype(np.float64) @ fr
    return np.sqrt(conv[:129] ** 2 + conv[129:] ** 2).astype(np.float32)


def mag_folded_f32(x):
    xp = pad(x.astype(np.float32))
    fr = np.stack([xp[64 * f:64 * f + 257] if 64 * f + 257 <= xp.size else np.concatenate([xp[64 * f:], [0]]) for f in range(25)], 1).astype(np.float32)
    xs = fr[0:128].copy(); xd = fr[0:128].copy()
    xs[1:] = fr[1:128] + fr[255:128:-1]; xd[1:] = fr[1:128] - fr[255:128:-1]
    xs[0] = fr[128]; xd[0] = 0
    bre = basis[:129, :128].copy(); bre[:, 0] = basis[:129, 128]
    bim = basis[129:, :128].copy(); bim[:, 0] = 0
    re = np.zeros((129, 25), np.float32); im = np.zeros((129, 25), np.float32)
    for s in range(32):                            # K-step order of the kernel: taps {s, 32+s, 64+s, 96+s}
        idx = [s, 32 + s, 64 + s, 96 + s]
        re = (re + (bre[:, idx].astype(np.float64) @ xs[idx].astype(np.float64))).astype(np.float32)
        im = (im + (bim[:, idx].astype(np.float64) @ xd[idx].astype(np.float64))).astype(np.float32)
    return np.sqrt(re * re + im * im).astype(np.float32)


def main():
    S, n = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 150
    pcm = synth.make_streams(S, n, seed0=77)
    orc = O.Oracle(open(W, "rb").read())
    worst = {"f64": 0.0, "folded": 0.0}
    for s in range(S):
        x = (pcm[s].astype(np.float32) / np.float32(32768.0)).reshape(n, 1536)
        h0, c0 = orc.new_state()
        st = {k: (np.zeros((2, 64), np.float32), np.zeros((2, 64), np.float32)) for k in worst}
        for i in range(n):
            pe, taps = orc.forward_chunk(x[i], h0, c0, taps=True)
            if s == 0 and i == 0:                  # the composition below must reproduce the oracle exactly
                _, m = O.stft_magnitude(x[i], basis)
                pz, _, _ = from_magnitude(m, *orc.new_state())
                assert np.array_equal(pz, pe), (pz, pe)
            for k, fn in (("f64", mag_f64), ("folded", mag_folded_f32)):
                p, hh, cc = from_magnitude(fn(x[i]), *st[k])
                st[k] = (hh, cc)
                worst[k] = max(worst[k], float(abs(p[1] - pe[1])))
        print(f"stream {s}: max |dp| so far {worst}", flush=True)
    print("RESULT", worst)

if __name__ == "__main__":
    main()
