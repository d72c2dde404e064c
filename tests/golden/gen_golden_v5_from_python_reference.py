#!/usr/bin/env python3
"""Generate tests/golden/silero_v5_seeded.testtensor and tests/golden/python_reference_v5.npz by IMPORTING the reference's PyTorch restatement of
Silero v5 (/root/reference/silero_vad.py::Silero_Vad_5) in the build container.

The reference ships NO v5 weights (its C test test.c:2027-2196 loads an untracked container; vadc runs v5 through onnxruntime with an external
model).  So the weights here are SEEDED random tensors of the v5 shapes (torch's default initialisers under torch.manual_seed(5), the decoder conv
scaled up so that the probabilities use the whole range; the STFT basis is the reference's forward_basis_buffer from the v3.1 container), and these
vectors pin SHAPES and ARITHMETIC -- chunking with the 64-sample context (vadc.c:105-162), right reflect pad, hop 128, the four k = 3 convs,
LSTM(128) with carried state, decoder -- not a trained model.  Evaluated in float64.

    python tests/golden/gen_golden_v5_from_python_reference.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import silero_vad  # noqa: E402  (reference, build container only)

from vadc_amd import synth, testtensor  # noqa: E402


def build_model():
    torch.manual_seed(5)
    m = silero_vad.Silero_Vad_5()
    basis = testtensor.load(os.path.join(HERE, "reference_fixtures", "silero_v31_16k.testtensor"))[0][1]
    with torch.no_grad():
        m.stft.forward_basis_buffer.copy_(torch.from_numpy(basis.copy()))
        m.decoder.decoder[2].weight.mul_(40.0)               # spread the probabilities over (0, 1)
        # the magnitudes of 16-bit audio / 32768 are small: give the first conv a gain so that the encoder is not all bias
        m.encoder[0].reparam_conv.weight.mul_(30.0)
    m.eval()
    return m


def container(m):
    sd = m.state_dict()
    f = lambda k: sd[k].detach().numpy().astype(np.float32)
    ts = [("forward_basis_buffer", f("stft.forward_basis_buffer"))]
    for l in range(4):
        ts += [(f"reparam_conv_{l}_weights", f(f"encoder.{l}.reparam_conv.weight")), (f"reparam_conv_{l}_biases", f(f"encoder.{l}.reparam_conv.bias"))]
    w = np.concatenate([f("decoder.rnn.weight_ih_l0"), f("decoder.rnn.weight_hh_l0")], axis=1)[None]       # utils.py:93-97
    b = (f("decoder.rnn.bias_ih_l0") + f("decoder.rnn.bias_hh_l0"))[None]                                    # utils.py:99-101
    ts += [("lstm_weights", w), ("lstm_biases", b), ("decoder_weights", f("decoder.decoder.2.weight")), ("decoder_biases", f("decoder.decoder.2.bias"))]
    return ts


@torch.no_grad()
def run_stream(m, pcm_i16):
    """vadc.c:105-162 (process_chunks_v5): every 512-sample window is preceded by the stream's previous 64 samples (zeros at the start)"""
    x = torch.from_numpy(pcm_i16.astype(np.float32) / np.float32(32768)).double()
    n = x.numel() // 512
    h = torch.zeros(1, 1, 128, dtype=torch.float64); c = torch.zeros(1, 1, 128, dtype=torch.float64)
    ctx = torch.zeros(64, dtype=torch.float64)
    probs = []
    for i in range(n):
        w = x[i * 512:(i + 1) * 512]
        out, h, c = m(torch.cat([ctx, w]).reshape(1, 576), h, c)
        ctx = w[-64:]
        probs.append(float(out.reshape(-1)[0]))
    return np.asarray(probs), h.reshape(128).numpy().copy(), c.reshape(128).numpy().copy()


def main():
    m = build_model()
    wpath = os.path.join(HERE, "silero_v5_seeded.testtensor")
    testtensor.dump(wpath, container(m))
    print("wrote", wpath, os.path.getsize(wpath), "bytes")
    m64 = m.double()
    pcm = {f"speech{k}": synth.speech_like(144 * 512, seed=100 + k) for k in range(3)}
    for kind in ("zeros", "noise", "square"):
        pcm[kind] = synth.control_stream(kind, 48 * 512, seed=7)
    out = {}
    for name, x in pcm.items():
        p, h, c = run_stream(m64, x)
        out[f"pcm_{name}"] = x; out[f"probs64_{name}"] = p; out[f"h64_{name}"] = h; out[f"c64_{name}"] = c
        print(f"{name}: {p.size} chunks, p[min,max]=({p.min():.4f},{p.max():.4f})")
    # stage taps of one chunk (chunk 20 of speech0, with its true context)
    x = pcm["speech0"].astype(np.float64) / 32768.0
    with torch.no_grad():
        inp = torch.from_numpy(x[20 * 512 - 64:21 * 512]).reshape(1, 576)
        spect = m64.stft(inp)
        e = m64.encoder
        c0 = e[0](spect); c1 = e[1](c0); c2 = e[2](c1); c3 = e[3](c2)
        for k, v in (("magnitude", spect), ("c0", c0), ("c1", c1), ("c2", c2), ("c3", c3)):
            out[f"tap20_{k}"] = v.squeeze(0).numpy()
    path = os.path.join(HERE, "python_reference_v5.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
