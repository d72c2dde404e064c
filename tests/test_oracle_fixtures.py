"""The CPU oracle against the reference's own 18 in-tree known-answer fixtures
(/root/reference/testdata/*.testtensor, copied as data into tests/golden/reference_fixtures/).
Each test mirrors one test of /root/reference/test.c (line given) with the same tolerance
(atol 1e-4 everywhere, 1e-10 for the decoder: test.c:198)."""
import numpy as np
import pytest

from oracle import oracle as O
from vadc_amd import testtensor as tt

ATOL = 1e-4


def load(fixture_path, name):
    return [a for _, a in tt.load(fixture_path(name))]


def maxerr(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


def test_dw_conv_129(fixture_path):                      # test.c:545
    x, w, b, ref = load(fixture_path, "dw_conv_129")
    assert maxerr(O.dw_conv_k5(x, w, b), ref) < ATOL


def test_pw_conv_129_16(fixture_path):                   # test.c:581
    x, w, b, ref = load(fixture_path, "pw_conv_129_16")
    assert maxerr(O.conv_k1(x, w, b), ref) < ATOL


def test_first_layer_conv_block(fixture_path):           # test.c:820
    dw_w, dw_b, pw_w, pw_b, pj_w, pj_b, x, ref = load(fixture_path, "first_layer_conv_block")
    assert maxerr(O.conv_block(x, dw_w, dw_b, pw_w, pw_b, pj_w, pj_b), ref) < ATOL


def test_decoder(fixture_path):                          # test.c:170 (atol 1e-10)
    x, w, b, ref = load(fixture_path, "decoder_test")
    assert maxerr(O.decoder(x[0], w, b), ref.reshape(-1)) < 1e-10


def test_softmax(fixture_path):                          # test.c:900
    x, ref = load(fixture_path, "softmax_test")
    assert maxerr(O.softmax_rows(x), ref) < ATOL


def test_layer_norm(fixture_path):                       # test.c:931
    x, w, b, ref = load(fixture_path, "layernorm_test")
    assert maxerr(O.layer_norm(x, w, b), ref) < ATOL


def test_batch_norm(fixture_path):                       # test.c:966
    x, mean, var, w, b, ref = load(fixture_path, "batchnorm_test")
    for i in range(x.shape[0]):
        assert maxerr(O.batch_norm(x[i], mean, var, w, b), ref[i]) < ATOL


def test_dual_head_attention(fixture_path):              # test.c:1105
    x, w, b, pw, pb, ref = load(fixture_path, "dual_head_attention_test")
    assert maxerr(O.attention(x, w, b, pw, pb), ref) < ATOL


def test_transformer_block_16_16_48(fixture_path):       # test.c:1143 (norm1,norm2 BEFORE linear1,linear2)
    qkv_w, qkv_b, out_w, out_b, n1w, n1b, n2w, n2b, l1w, l1b, l2w, l2b, x, ref = \
        load(fixture_path, "transformer_block_test_16_16_48")
    d = 16
    dummy = [np.zeros((d, 1, 5), np.float32), np.zeros(d, np.float32), np.zeros((d, d, 1), np.float32), np.zeros(d, np.float32)]
    tail = [np.zeros((d, d, 1), np.float32), np.zeros(d, np.float32)] + [np.ones(d, np.float32)] * 4
    layer = O.make_layer(dummy + [qkv_w, qkv_b, out_w, out_b, n1w, n1b, l1w, l1b, l2w, l2b, n2w, n2b] + tail,
                         has_proj=False, stride=1, t_in=25)
    assert maxerr(O.transformer_block(x, layer), ref) < ATOL


def _run_layers(tensors, spec, x):
    """spec: list of (has_proj, stride); tensors consumed positionally like fill_transformer_weights (tensor.h:114)."""
    idx = 0
    t_in = x.shape[-1]
    for has_proj, stride in spec:
        n = 24 if has_proj else 22
        layer = O.make_layer(tensors[idx:idx + n], has_proj, stride, t_in)
        x = O.transformer_layer(x, layer)
        idx += n
        t_in = x.shape[-1]
    return x, idx


L1, L2, L3, L4 = (True, 2), (True, 2), (False, 1), (True, 1)


@pytest.mark.parametrize("name,spec", [
    ("transformer_first_layer", [L1]),                   # test.c:1196
    ("transformer_layers_1_2", [L1, L2]),                # test.c:1239
    ("transformer_layers_3", [L3]),                      # test.c:1918
    ("transformer_layers_1_2_3", [L1, L2, L3]),          # test.c:1320
    ("transformer_layers_1_2_3_4", [L1, L2, L3, L4]),    # test.c:1392
])
def test_transformer_layers(fixture_path, name, spec):
    ts = load(fixture_path, name)
    x, ref = ts[-2], ts[-1]
    out, used = _run_layers(ts[:-2], spec, x[0])
    assert used == len(ts) - 2
    assert maxerr(out, ref[0]) < ATOL


def test_adaptive_normalization_encoder(fixture_path):   # test.c:1434
    ts = load(fixture_path, "adaptive_normalization_encoder")
    x, ref = ts[-2], ts[-1]
    xn = O.adaptive_norm(x[0])
    out, _ = _run_layers(ts[:-2], [L1, L2, L3, L4], xn)
    assert maxerr(out, ref[0]) < ATOL


def test_adaptive_audio_normalization(fixture_path):     # test.c:1071 (batched, per-item mean)
    x, ref = load(fixture_path, "adaptive_audio_normalization_test")
    for i in range(x.shape[0]):
        assert maxerr(O.adaptive_norm(x[i]), ref[i]) < ATOL


def test_lstm(fixture_path):                             # test.c:243: out = [7 outputs ; h[2] ; c[2]]
    x, h0, c0, w, b, ref = load(fixture_path, "lstm_nito_reference_randn")
    out, hn, cn = O.lstm_seq(x, w, b, h0, c0)
    got = np.concatenate([out, hn, cn], axis=0)
    assert got.shape == ref.shape == (11, 64)
    assert maxerr(got, ref) < ATOL


def test_transpose2d():                                  # test.c:862 -- layout helper used throughout
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    assert np.array_equal(a.T, np.array([[0, 3], [1, 4], [2, 5]], np.float32))
