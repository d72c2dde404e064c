"""The CPU oracle against committed golden vectors:

* tests/golden/c_reference_v31.npz  -- outputs of the reference C backend compiled in place
  (gen_golden_from_c_reference.py).  The oracle reproduces the reference's reduction order, so the bar is
  BIT-EXACT probabilities, state and STFT magnitudes.
* tests/golden/python_reference_v31.npz -- the reference's PyTorch restatement evaluated in float64
  (gen_golden_from_python_reference.py).  Bar: |dp| <= 1e-4 (north-star tolerance; SURVEY.md Appendix F puts the
  C backend's own fp32 noise against fp64 at <= 9.2e-5) and stage outputs within a relative 2e-3 of peak.
"""
import os

import numpy as np
import pytest

from oracle import oracle as O
from vadc_amd import testtensor as tt

from conftest import GOLDEN

STREAMS = ["speech0", "speech1", "speech2", "zeros", "noise", "square"]


@pytest.fixture(scope="module")
def orc(weights_blob):
    return O.Oracle(weights_blob)


@pytest.fixture(scope="module")
def gold_c():
    return np.load(os.path.join(GOLDEN, "c_reference_v31.npz"))


@pytest.fixture(scope="module")
def gold_py():
    return np.load(os.path.join(GOLDEN, "python_reference_v31.npz"))


@pytest.mark.parametrize("name", STREAMS)
def test_probs_bit_exact_vs_c_reference(orc, gold_c, gold_py, name):
    pcm = gold_py[f"pcm_{name}"]
    h, c = orc.new_state()
    probs = orc.forward_stream(pcm, h, c)
    assert np.array_equal(probs.view(np.uint32), gold_c[f"probs_{name}"].view(np.uint32))
    assert np.array_equal(h.view(np.uint32), gold_c[f"h_{name}"].view(np.uint32))
    assert np.array_equal(c.view(np.uint32), gold_c[f"c_{name}"].view(np.uint32))


def test_stft_magnitude_bit_exact_vs_c_reference(orc, gold_c, gold_py, weights_blob):
    basis = tt.loads(weights_blob)[0][1]
    x = gold_py["pcm_speech0"].astype(np.float32) / np.float32(32768)
    want = gold_c["stft_mag_speech0_chunks_0_20"]
    for k, ci in enumerate((0, 20)):
        _, mag = O.stft_magnitude(x[ci * 1536:(ci + 1) * 1536], basis)
        assert np.array_equal(mag.view(np.uint32), want[k].view(np.uint32))


@pytest.mark.parametrize("name", STREAMS)
def test_probs_vs_python_reference_fp64(orc, gold_py, name):
    probs = orc.forward_stream(gold_py[f"pcm_{name}"])
    d = np.abs(probs.astype(np.float64) - gold_py[f"probs64_{name}"])
    assert d.max() <= 1e-4, d.max()


@pytest.mark.parametrize("ci", [0, 20])
def test_stage_taps_vs_python_reference_fp64(orc, gold_py, ci):
    x = gold_py["pcm_speech0"].astype(np.float32) / np.float32(32768)
    h = gold_py[f"tap{ci}_h_in"].astype(np.float32).copy()
    c = gold_py[f"tap{ci}_c_in"].astype(np.float32).copy()
    out, taps = orc.forward_chunk(x[ci * 1536:(ci + 1) * 1536], h, c, taps=True)
    for k in ("magnitude", "normalized", "l1", "l2", "l3", "l4", "lstm_out"):
        want = gold_py[f"tap{ci}_{k}"]
        scale = max(1.0, float(np.abs(want).max()))
        err = float(np.abs(taps[k].astype(np.float64) - want).max()) / scale
        assert err < 2e-3, (k, err)
    assert np.abs(out - gold_py[f"tap{ci}_probs"]).max() <= 1e-4
    assert np.abs(h - gold_py[f"tap{ci}_hn"]).max() < 1e-3
    assert np.abs(c - gold_py[f"tap{ci}_cn"]).max() < 1e-3


def test_state_carry_equals_one_long_call(orc, gold_py):
    """Feeding a stream chunk by chunk == feeding it in one call (state is the only coupling)."""
    pcm = gold_py["pcm_speech1"][: 12 * 1536]
    whole = orc.forward_stream(pcm)
    h, c = orc.new_state()
    parts = np.concatenate([orc.forward_stream(pcm[i * 1536:(i + 1) * 1536], h, c) for i in range(12)])
    assert np.array_equal(whole.view(np.uint32), parts.view(np.uint32))


def test_malformed_weights_rejected(weights_blob):
    with pytest.raises(ValueError):
        O.Oracle(weights_blob[:-4])
    with pytest.raises(ValueError):
        O.Oracle(b"\x02\x00\x00\x00" + weights_blob[4:])
