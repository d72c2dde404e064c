#!/usr/bin/env python3
"""Generate tests/golden/python_reference_v4.npz by IMPORTING the reference's PyTorch restatement of Silero v4
(/root/reference/silero_vad.py::Silero_V4, 16 kHz) in the build container.  v4 has no C implementation in the
reference (silero.h:59) -- it runs only through onnxruntime there -- so this class is the in-tree statement of
the v4 arithmetic, and these vectors are PyTorch-vs-build (not ORT-vs-build) goldens.

    python -m vadc_amd.onnx_weights /root/reference/silero_vad_v4.onnx tests/golden/silero_v4_16k.testtensor
    python tests/golden/gen_golden_v4_from_python_reference.py

Weights: tests/golden/silero_v4_16k.testtensor (extracted from the reference's silero_vad_v4.onnx by
vadc_amd/onnx_weights.py: 16 kHz branch, ONNX LSTM gate order i,o,f,c re-ordered to i,f,g,o, conv+BN already folded
by the exporter -> loaded into Conv1d with an identity BatchNorm).  Evaluated in float64 ("truth") and float32.
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


def build_model(weights_path, sr=16000):
    ts = [a for _, a in testtensor.load(weights_path)]
    sd = {}
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a).copy())
    sd["feature_extractor.forward_basis_buffer"] = T(ts[0])
    sd["adaptive_normalization.filter_"] = T(ts[35]).reshape(1, 1, 7)

    def block(prefix, i0, proj=True):
        names = ["dw_conv.0.weight", "dw_conv.0.bias", "pw_conv.0.weight", "pw_conv.0.bias"] + (["proj.weight", "proj.bias"] if proj else [])
        for k, n in enumerate(names):
            sd[f"{prefix}.{n}"] = T(ts[i0 + k])

    def conv_bn(ci, bi, i0, ch):
        sd[f"encoder.{ci}.weight"] = T(ts[i0]); sd[f"encoder.{ci}.bias"] = T(ts[i0 + 1])
        sd[f"encoder.{bi}.weight"] = torch.ones(ch); sd[f"encoder.{bi}.bias"] = torch.zeros(ch)
        sd[f"encoder.{bi}.running_mean"] = torch.zeros(ch)
        sd[f"encoder.{bi}.running_var"] = torch.ones(ch) - 1e-5      # identity: (x - 0) / sqrt(1 - eps + eps)

    block("first_layer.0", 1)
    conv_bn(0, 1, 7, 16)
    block("encoder.3.0", 9)
    conv_bn(4, 5, 15, 32)
    block("encoder.7.0", 17, proj=False)
    conv_bn(8, 9, 21, 32)
    block("encoder.11.0", 23)
    conv_bn(12, 13, 29, 64)
    W, B = ts[31], ts[32]
    for l in range(2):
        sd[f"decoder.rnn.weight_ih_l{l}"] = T(W[l][:, :64]); sd[f"decoder.rnn.weight_hh_l{l}"] = T(W[l][:, 64:])
        sd[f"decoder.rnn.bias_ih_l{l}"] = T(B[l]); sd[f"decoder.rnn.bias_hh_l{l}"] = torch.zeros(256)
    sd["decoder.decoder.1.weight"] = T(ts[33]); sd["decoder.decoder.1.bias"] = T(ts[34])
    m = silero_vad.Silero_V4(sr)
    own = m.state_dict()
    for k in own:
        if k.endswith("num_batches_tracked"):
            sd[k] = own[k]
    m.load_state_dict(sd, strict=True)
    m.eval()
    return m


@torch.no_grad()
def run_stream(m, pcm_i16, dtype, window=1536):
    x = torch.from_numpy(pcm_i16.astype(np.float32) / np.float32(32768)).to(dtype).reshape(-1, window)
    h = torch.zeros(2, 1, 64, dtype=dtype); c = torch.zeros(2, 1, 64, dtype=dtype)
    probs = []
    for i in range(x.shape[0]):
        out, h, c = m(x[i:i + 1], h, c)
        probs.append(float(out.reshape(-1)[0]))
    return np.asarray(probs), h.reshape(2, 64).numpy().copy(), c.reshape(2, 64).numpy().copy()


@torch.no_grad()
def stage_taps(m, chunk_f32, h, c, dtype):
    x = torch.from_numpy(chunk_f32).to(dtype).reshape(1, 1536)
    spect = m.feature_extractor(x)
    norm = m.adaptive_normalization(spect)
    cb1 = m.first_layer(torch.cat([spect, norm], 1))
    e = m.encoder
    l1 = e[2](e[1](e[0](cb1)))
    l2 = e[6](e[5](e[4](e[3](l1))))
    l3 = e[10](e[9](e[8](e[7](l2))))
    l4 = e[14](e[13](e[12](e[11](l3))))
    lstm_out, (hn, cn) = m.decoder.rnn(l4.permute(0, 2, 1), (h, c))
    dec = m.decoder.decoder(lstm_out.permute(0, 2, 1))
    f = lambda t: t.squeeze(0).numpy().astype(np.float64)
    return dict(magnitude=f(spect), normalized=f(norm), l1=f(l1), l2=f(l2), l3=f(l3), l4=f(l4), lstm_out=f(lstm_out),
                prob=np.float64(dec.mean()), hn=hn.reshape(2, 64).numpy().astype(np.float64), cn=cn.reshape(2, 64).numpy().astype(np.float64))


def main():
    weights = os.path.join(HERE, "silero_v4_16k.testtensor")
    m32 = build_model(weights)
    m64 = build_model(weights).double()
    pcm = {f"speech{k}": synth.speech_like(48 * 1536, seed=100 + k) for k in range(3)}
    for kind in ("zeros", "noise", "square"):
        pcm[kind] = synth.control_stream(kind, 16 * 1536, seed=7)
    out = {}
    for name, x in pcm.items():
        p64, h64, c64 = run_stream(m64, x, torch.float64)
        p32, _, _ = run_stream(m32, x, torch.float32)
        out[f"pcm_{name}"] = x; out[f"probs64_{name}"] = p64; out[f"probs32_{name}"] = p32
        out[f"h64_{name}"] = h64; out[f"c64_{name}"] = c64
        print(f"{name}: p[min,max]=({p64.min():.4f},{p64.max():.4f})  |fp32-fp64|max={np.abs(p64 - p32).max():.2e}")
    x = pcm["speech0"].astype(np.float32) / np.float32(32768)
    z = torch.zeros(2, 1, 64, dtype=torch.float64)
    for ci in (0, 20):
        h, c = z.clone(), z.clone()
        with torch.no_grad():
            for i in range(ci):
                _, h, c = m64(torch.from_numpy(x[i * 1536:(i + 1) * 1536]).double().reshape(1, 1536), h, c)
        out[f"tap{ci}_h_in"] = h.reshape(2, 64).numpy(); out[f"tap{ci}_c_in"] = c.reshape(2, 64).numpy()
        for k, v in stage_taps(m64, x[ci * 1536:(ci + 1) * 1536], h, c, torch.float64).items():
            out[f"tap{ci}_{k}"] = v
    path = os.path.join(HERE, "python_reference_v4.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    # 512- and 1024-sample windows (the v4 graph takes 512 ... 1536 samples: onnx_helpers.c:164-170, --sequence_count vadc.c:743-752): the same
    # streams cut into shorter chunks, float64 probabilities and final state
    outw = {}
    for window in (512, 1024):
        for name in ("speech0", "speech1", "noise", "square"):
            x = pcm[name][: (pcm[name].size // window) * window]
            p64, h64, c64 = run_stream(m64, x, torch.float64, window)
            outw[f"probs64_w{window}_{name}"] = p64; outw[f"h64_w{window}_{name}"] = h64; outw[f"c64_w{window}_{name}"] = c64
            print(f"window {window} {name}: {p64.size} chunks, p[min,max]=({p64.min():.4f},{p64.max():.4f})")
    pathw = os.path.join(HERE, "python_reference_v4_windows.npz")
    np.savez_compressed(pathw, **outw)
    print("wrote", pathw, os.path.getsize(pathw), "bytes  (pcm: the streams of python_reference_v4.npz)")
    # windows between the thirds (round 5): 768 and 1280 samples = 12 and 20 STFT frames, whose strided stages meet ODD lengths (12 -> 6 -> 3 -> 2, 20 -> 10 -> 5 -> 3)
    outm = {}
    for window in (768, 1280):
        for name in ("speech0", "speech1", "noise", "square"):
            x = pcm[name][: (pcm[name].size // window) * window]
            p64, h64, c64 = run_stream(m64, x, torch.float64, window)
            outm[f"probs64_w{window}_{name}"] = p64; outm[f"h64_w{window}_{name}"] = h64; outm[f"c64_w{window}_{name}"] = c64
            print(f"window {window} {name}: {p64.size} chunks, p[min,max]=({p64.min():.4f},{p64.max():.4f})")
    pathm = os.path.join(HERE, "python_reference_v4_windows_768_1280.npz")
    np.savez_compressed(pathm, **outm)
    print("wrote", pathm, os.path.getsize(pathm), "bytes  (pcm: the streams of python_reference_v4.npz)")
    # the 8 kHz branch of the v4 graph (silero_vad.py::Silero_V4(8000): third strided conv with stride 1; weights `model_8k.*` of the reference's
    # silero_vad_v4.onnx -> tests/golden/silero_v4_8k.testtensor): windows of 768 / 512 / 256 samples = 12 / 8 / 4 STFT frames.  The streams
    # are the same sample sequences, now read as 8 kHz audio.
    m64_8k = build_model(os.path.join(HERE, "silero_v4_8k.testtensor"), 8000).double()
    out8 = {}
    for window in (768, 512, 256):
        for name in ("speech0", "speech1", "noise", "square"):
            x = pcm[name][: (pcm[name].size // window) * window]
            p64, h64, c64 = run_stream(m64_8k, x, torch.float64, window)
            out8[f"probs64_w{window}_{name}"] = p64; out8[f"h64_w{window}_{name}"] = h64; out8[f"c64_w{window}_{name}"] = c64
            print(f"8 kHz window {window} {name}: {p64.size} chunks, p[min,max]=({p64.min():.4f},{p64.max():.4f})")
    path8 = os.path.join(HERE, "python_reference_v4_8k.npz")
    np.savez_compressed(path8, **out8)
    print("wrote", path8, os.path.getsize(path8), "bytes  (pcm: the streams of python_reference_v4.npz)")
    # every multiple of 64 samples (round 6): windows that are no multiple of 256 -- 9, 11, 13, 15, 17, 19, 21, 22 and 23 STFT frames at 16 kHz, 5, 7 and 11 at 8 kHz -- whose
    # right reflect pad starts inside the next 64-sample block and whose strided stages meet every parity of lengths (9 -> 5 -> 3 -> 2, 13 -> 7 -> 4 -> 2, 23 -> 12 -> 6 -> 3, ...)
    out64 = {}
    for window in (576, 704, 832, 960, 1088, 1216, 1344, 1408, 1472, 520, 1000, 1336, 1528):      # (the last four: no multiple of 64 -- trailing samples that fill no frame)
        for name in ("speech0", "speech1", "square"):
            x = pcm[name][: (pcm[name].size // window) * window]
            p64, h64, c64 = run_stream(m64, x, torch.float64, window)
            out64[f"probs64_w{window}_{name}"] = p64; out64[f"h64_w{window}_{name}"] = h64; out64[f"c64_w{window}_{name}"] = c64
            print(f"window {window} {name}: {p64.size} chunks, p[min,max]=({p64.min():.4f},{p64.max():.4f})")
    for window in (320, 448, 704):
        for name in ("speech0", "speech1", "square"):
            x = pcm[name][: (pcm[name].size // window) * window]
            p64, h64, c64 = run_stream(m64_8k, x, torch.float64, window)
            out64[f"probs64_8k_w{window}_{name}"] = p64; out64[f"h64_8k_w{window}_{name}"] = h64; out64[f"c64_8k_w{window}_{name}"] = c64
            print(f"8 kHz window {window} {name}: {p64.size} chunks, p[min,max]=({p64.min():.4f},{p64.max():.4f})")
    path64 = os.path.join(HERE, "python_reference_v4_windows_64.npz")
    np.savez_compressed(path64, **out64)
    print("wrote", path64, os.path.getsize(path64), "bytes  (pcm: the streams of python_reference_v4.npz)")


if __name__ == "__main__":
    main()
