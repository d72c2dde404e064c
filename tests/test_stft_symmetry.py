"""The derivation behind k_frontend_sym (vadc_amd/csrc/kernels_frontend.hip), checked on the CPU against the oracle's bit-exact STFT:
the reference's basis has the bin-mirror and quarter-mirror DFT symmetries BIT FOR BIT, so the reference's reduction tree (stft.c:115-184)
evaluated for bins 0..32 yields, with 7 additions per derived row on sign-flipped lane sums, the magnitudes of all 129 bins -- the same
bits as evaluating the tree for every bin.  float32 numpy here mirrors the kernel's arithmetic step by step."""
import numpy as np
import pytest

from oracle import oracle as O
from vadc_amd import synth, testtensor as tt

f32 = np.float32


@pytest.fixture(scope="module")
def basis(weights_blob):
    return tt.loads(weights_blob)[0][1].reshape(258, 256).astype(f32)


def test_reference_basis_has_the_dft_symmetries_bitwise(basis):
    """what vadc_amd_create verifies (engine.hip basis_has_dft_symmetries) before it lets k_frontend_sym run"""
    re, im = basis[:129], basis[129:]
    n = np.arange(256)
    sg = np.where(n % 2 == 0, 1.0, -1.0).astype(f32)
    for b in range(65):
        assert np.array_equal(re[128 - b], sg * re[b]) and np.array_equal(im[128 - b], -sg * im[b])
        exp_re = np.select([n % 4 == 0, n % 4 == 1, n % 4 == 2, n % 4 == 3], [re[b], -im[b], -re[b], im[b]])
        exp_im = np.select([n % 4 == 0, n % 4 == 1, n % 4 == 2, n % 4 == 3], [-im[b], -re[b], im[b], re[b]])
        assert np.array_equal(re[64 - b], exp_re) and np.array_equal(im[64 - b], exp_im)


def _lane_sums(xf, k):
    """xf [25, 256] frames, k [256] taps -> v[frame][l]: products, j tree and group sums of stft.c:141-167 per tree lane l"""
    p = (xf * k[None, :]).astype(f32).reshape(25, 4, 8, 8)        # [frame][i][j][l]
    a01 = p[:, :, 0] + p[:, :, 1]; a23 = p[:, :, 2] + p[:, :, 3]; a45 = p[:, :, 4] + p[:, :, 5]; a67 = p[:, :, 6] + p[:, :, 7]
    g = (a01 + a23) + (a45 + a67)
    return (g[:, 0] + g[:, 1]) + (g[:, 2] + g[:, 3])


def _frames(chunk):
    pad = np.concatenate([chunk[128:0:-1], chunk, chunk[-2:-130:-1]]).astype(f32)
    return np.stack([pad[64 * n:64 * n + 256] for n in range(25)])


@pytest.mark.parametrize("kind,seed", [("speech", 1), ("speech", 2), ("noise", 3), ("zeros", 4), ("square", 5)])
def test_129_bins_from_33_base_bins_are_the_oracle_bits(weights_blob, basis, kind, seed):
    orc = O.Oracle(weights_blob)
    pcm = synth.speech_like(2 * 1536, seed) if kind == "speech" else synth.control_stream(kind, 2 * 1536, seed)
    x = pcm.astype(f32) / f32(32768)
    mag = lambda r, i: np.sqrt((r * r).astype(f32) + (i * i).astype(f32)).astype(f32)
    for c in range(2):
        ch = x[c * 1536:(c + 1) * 1536]
        h, cc = orc.new_state()
        _, taps = orc.forward_chunk(ch, h, cc, taps=True)
        want = taps["magnitude"]                                  # [129, 25], the reference's bits (tests/test_oracle_vs_ref.py)
        xf = _frames(ch)
        got = np.full((129, 25), np.nan, f32)
        for b in range(33):
            vr, vi = _lane_sums(xf, basis[b]), _lane_sums(xf, basis[129 + b])
            E = [[None] * 4 for _ in range(8)]
            for lp in range(4):
                rx, ry, ix, iy = vr[:, 2 * lp], vr[:, 2 * lp + 1], vi[:, 2 * lp], vi[:, 2 * lp + 1]
                E[0][lp] = rx + ry; E[1][lp] = ix + iy; E[2][lp] = rx - ry; E[3][lp] = ix - iy      # rows b, 128 - b
                E[4][lp] = rx - iy; E[5][lp] = ix + ry; E[6][lp] = rx + iy; E[7][lp] = ix - ry      # rows 64 - b, 64 + b (sign alternates with lp)
            y = [(e[0] + e[1]) + (e[2] + e[3]) if k < 4 else (e[0] - e[1]) + (e[2] - e[3]) for k, e in enumerate(E)]
            got[b] = mag(y[0], y[1]); got[128 - b] = mag(y[2], y[3]); got[64 - b] = mag(y[4], y[5]); got[64 + b] = mag(y[6], y[7])
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (kind, c)
