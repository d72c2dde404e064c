"""Live check of the restatement against the reference's C sources compiled in place
(oracle/_ref/libvadc_ref.so).  Only runs where that build exists (the build container, or a GPU box that
received the prebuilt file); skipped otherwise.  Bar: bit-exact."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from vadc_amd import synth, testtensor as tt

pytestmark = pytest.mark.skipif(not os.path.exists(O.REF_LIB_PATH), reason="oracle/_ref not built here")


@pytest.fixture(scope="module")
def pair(weights_path, weights_blob):
    return O.Oracle(weights_blob), O.Reference(weights_path)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_stream_bit_exact(pair, seed):
    orc, ref = pair
    pcm = synth.speech_like(40 * 1536, seed=seed)
    x = pcm.astype(np.float32) / np.float32(32768)
    ref.reset()
    want = ref.run(x, batch=8)
    h, c = orc.new_state()
    got = orc.forward_stream(pcm, h, c)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    hr, cr = ref.state()
    assert np.array_equal(h.view(np.uint32), hr.view(np.uint32)) and np.array_equal(c.view(np.uint32), cr.view(np.uint32))


def test_stages_bit_exact(pair, weights_blob):
    orc, ref = pair
    basis = tt.loads(weights_blob)[0][1]
    x = synth.speech_like(4 * 1536, seed=5).astype(np.float32) / np.float32(32768)
    mags = ref.stft(x)
    norm = ref.adaptive_norm(mags)
    enc = ref.encoder(norm)
    for i in range(4):
        h, c = orc.new_state()
        _, taps = orc.forward_chunk(x[i * 1536:(i + 1) * 1536], h, c, taps=True)
        assert np.array_equal(taps["magnitude"].view(np.uint32), mags[i].view(np.uint32))
        assert np.array_equal(taps["normalized"].view(np.uint32), norm[i].view(np.uint32))
        assert np.array_equal(taps["l4"].view(np.uint32), enc[i].view(np.uint32))
