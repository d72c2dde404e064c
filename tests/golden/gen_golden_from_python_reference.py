#!/usr/bin/env python3
"""Generate tests/golden/python_reference_v31.npz by IMPORTING the reference's PyTorch restatement
(/root/reference/silero_vad.py::Silero_V3) in the build container.  The reference's python cannot
travel to the GPU box, so only the resulting vectors (inputs + expected outputs) are committed.

    python tests/golden/gen_golden_from_python_reference.py

Weights: tests/golden/reference_fixtures/silero_v31_16k.testtensor, mapped onto the torch module through
the reference's own key maps (/root/reference/utils.py:114-222, inverted).  LSTM: W[:, :64] -> weight_ih,
W[:, 64:] -> weight_hh, fused bias -> bias_ih (bias_hh = 0) (utils.py:93-101).  adaptive_normalization.filter_
is the 7-tap constant of misc.c:5-13.

The module is evaluated in float64 ("truth": SURVEY.md Appendix F -- an independent fp32 evaluation differs
from the C backend by up to 1.5e-4 purely through STFT rounding, fp64 stays within 9.2e-5) and the stage
outputs of a few chunks are stored alongside.
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
import utils as ref_utils  # noqa: E402

from vadc_amd import synth, testtensor  # noqa: E402

FILTER = [0.03663284704089164733887, 0.11128076165914535522461, 0.21674531698226928710938,
          0.27068215608596801757812, 0.21674531698226928710938, 0.11128076165914535522461,
          0.03663284704089164733887]


def build_model(weights_path):
    named = dict(testtensor.load(weights_path))
    sd = {}

    def put(prefix, keymap):
        for k, torch_key in keymap.items():
            sd[torch_key.replace("_model1.", "")] = torch.from_numpy(named[f"{prefix}.{k}"].copy())

    put("transformer_l1", ref_utils.transformer_l1_key_map())
    put("transformer_l2", ref_utils.transformer_l2_key_map(4))
    put("transformer_l3", ref_utils.transformer_l3_key_map(9))
    put("transformer_l4", ref_utils.transformer_l2_key_map(14))
    sd["feature_extractor.forward_basis_buffer"] = torch.from_numpy(named["forward_basis_buffer"].copy())
    sd["adaptive_normalization.filter_"] = torch.tensor(FILTER, dtype=torch.float32).reshape(1, 1, 7)
    W, B = named["weights"], named["biases"]
    for l in range(2):
        sd[f"lstm.weight_ih_l{l}"] = torch.from_numpy(W[l][:, :64].copy())
        sd[f"lstm.weight_hh_l{l}"] = torch.from_numpy(W[l][:, 64:].copy())
        sd[f"lstm.bias_ih_l{l}"] = torch.from_numpy(B[l].copy())
        sd[f"lstm.bias_hh_l{l}"] = torch.zeros(256)
    sd["decoder.1.weight"] = torch.from_numpy(named["decoder_weights"].copy())
    sd["decoder.1.bias"] = torch.from_numpy(named["decoder_biases"].copy())
    m = silero_vad.Silero_V3(16000)
    # BatchNorm bookkeeping buffers are not part of the weights file
    own = m.state_dict()
    for k in own:
        if k.endswith("num_batches_tracked"):
            sd[k] = own[k]
    missing, unexpected = m.load_state_dict(sd, strict=True), None
    m.eval()
    return m


@torch.no_grad()
def run_stream(m, pcm_i16, dtype):
    x = torch.from_numpy(pcm_i16.astype(np.float32) / np.float32(32768)).to(dtype).reshape(-1, 1536)
    h = torch.zeros(2, 1, 64, dtype=dtype)
    c = torch.zeros(2, 1, 64, dtype=dtype)
    probs = []
    for i in range(x.shape[0]):
        out, h, c = m(x[i:i + 1], h, c)
        probs.append(out.reshape(2).numpy().copy())
    return np.stack(probs), h.reshape(2, 64).numpy().copy(), c.reshape(2, 64).numpy().copy()


@torch.no_grad()
def stage_taps(m, chunk_f32, h, c, dtype):
    x = torch.from_numpy(chunk_f32).to(dtype).reshape(1, 1536)
    spect = m.feature_extractor(x)
    norm = m.adaptive_normalization(spect)
    cb1 = m.first_layer(norm)
    enc = m.encoder
    l1 = enc[3](enc[2](enc[1](enc[0](cb1))))
    l2 = enc[8](enc[7](enc[6](enc[5](enc[4](l1)))))
    l3 = enc[13](enc[12](enc[11](enc[10](enc[9](l2)))))
    l4 = enc[18](enc[17](enc[16](enc[15](enc[14](l3)))))
    lstm_out, (hn, cn) = m.lstm(l4.permute(0, 2, 1), (h, c))
    out = m.decoder(lstm_out.permute(0, 2, 1))
    f = lambda t: t.squeeze(0).numpy().astype(np.float64)
    return dict(magnitude=f(spect), normalized=f(norm), l1=f(l1), l2=f(l2), l3=f(l3), l4=f(l4),
                lstm_out=f(lstm_out), probs=out.reshape(2).numpy().astype(np.float64),
                hn=hn.reshape(2, 64).numpy().astype(np.float64), cn=cn.reshape(2, 64).numpy().astype(np.float64))


def main():
    weights = os.path.join(HERE, "reference_fixtures", "silero_v31_16k.testtensor")
    m32 = build_model(weights)
    m64 = build_model(weights).double()

    n_speech, n_chunks = 3, 48
    pcm = {f"speech{k}": synth.speech_like(n_chunks * 1536, seed=100 + k) for k in range(n_speech)}
    for kind in ("zeros", "noise", "square"):
        pcm[kind] = synth.control_stream(kind, 16 * 1536, seed=7)

    out = {}
    for name, x in pcm.items():
        p64, h64, c64 = run_stream(m64, x, torch.float64)
        p32, _, _ = run_stream(m32, x, torch.float32)
        out[f"pcm_{name}"] = x
        out[f"probs64_{name}"] = p64
        out[f"probs32_{name}"] = p32
        out[f"h64_{name}"] = h64
        out[f"c64_{name}"] = c64
        print(f"{name}: p[min,max]=({p64[:,1].min():.4f},{p64[:,1].max():.4f})  |fp32-fp64|max={np.abs(p64 - p32).max():.2e}")

    # stage taps (fp64) for chunk 0 (zero state) and a mid-stream chunk of speech0
    x = pcm["speech0"].astype(np.float32) / np.float32(32768)
    z = torch.zeros(2, 1, 64, dtype=torch.float64)
    for ci in (0, 20):
        # state before chunk ci
        h, c = z.clone(), z.clone()
        with torch.no_grad():
            for i in range(ci):
                _, h, c = m64(torch.from_numpy(x[i * 1536:(i + 1) * 1536]).double().reshape(1, 1536), h, c)
        taps = stage_taps(m64, x[ci * 1536:(ci + 1) * 1536], h, c, torch.float64)
        out[f"tap{ci}_h_in"] = h.reshape(2, 64).numpy()
        out[f"tap{ci}_c_in"] = c.reshape(2, 64).numpy()
        for k, v in taps.items():
            out[f"tap{ci}_{k}"] = v

    path = os.path.join(HERE, "python_reference_v31.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
