"""vadc_amd.onnx_weights.silero_v3_tensors: silero_vad_v3.onnx -> the 99-tensor container of the reference's C backend (tensor.h:114-191).
The reference ships both the ONNX file and the container made from the same PyTorch weights (testdata/silero_v31_16k.testtensor, here under
tests/golden/reference_fixtures/): everything the exporter left alone must come out bit-identical, the BatchNorm it folded into the strided convs must
fold the fixture's conv + BatchNorm to the same numbers, and the oracle (silero_v3.c:72-215 restated) must give the same probabilities from both.
The ONNX file is the reference's (not copied here): the test runs where /root/reference is present and is skipped elsewhere."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from vadc_amd import onnx_weights, synth, testtensor as tt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ONNX = "/root/reference/silero_vad_v3.onnx"
FIXTURE = os.path.join(ROOT, "tests", "golden", "reference_fixtures", "silero_v31_16k.testtensor")

pytestmark = pytest.mark.skipif(not os.path.exists(ONNX), reason="the reference's silero_vad_v3.onnx is not on this machine")


def test_v3_onnx_gives_the_c_backends_container():
    got = onnx_weights.silero_v3_tensors(ONNX)
    want = tt.load(FIXTURE)
    assert [n for n, _ in got] == [n for n, _ in want]
    fx = dict(want)
    for (name, a), (_, b) in zip(got, want):
        assert a.shape == b.shape and a.dtype == np.float32, name
        if ".conv_" in name or ".batch_norm_" in name:
            continue
        assert np.array_equal(a, b), name                       # dw / pw / proj convs, Linear weights (transposed back), LayerNorms, LSTM (gates re-ordered), decoder
    ex = dict(got)
    for layer in range(1, 5):
        p = f"transformer_l{layer}."
        scale = fx[p + "batch_norm_weights"] / np.sqrt(fx[p + "batch_norm_running_var"] + np.float32(onnx_weights.V3_BN_EPS))
        w = fx[p + "conv_weights"] * scale[:, None, None]
        b = (fx[p + "conv_biases"] - fx[p + "batch_norm_running_mean"]) * scale + fx[p + "batch_norm_biases"]
        assert np.abs(w - ex[p + "conv_weights"]).max() <= 1e-6 and np.abs(b - ex[p + "conv_biases"]).max() <= 1e-6, layer
        # ... and the BatchNorm that goes out with the folded conv is the identity in batch_norm's own arithmetic (misc.c:98-141)
        assert np.all(np.sqrt(ex[p + "batch_norm_running_var"] + np.float32(onnx_weights.V3_BN_EPS)) == 1.0)


def test_the_oracle_gives_the_same_probabilities_from_both_containers():
    pcm = synth.make_streams(2, 12, seed0=31)
    a = O.Oracle(tt.dumps(onnx_weights.silero_v3_tensors(ONNX)))
    b = O.Oracle(open(FIXTURE, "rb").read())
    for s in range(2):
        pa, pb = a.forward_stream(pcm[s]), b.forward_stream(pcm[s])
        assert np.abs(pa - pb).max() <= 2e-6, s
