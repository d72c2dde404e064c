"""Silero v5 shapes on the HIP path (through the C-ABI) against the v5 CPU oracle and the goldens generated from the reference's
PyTorch class silero_vad.py::Silero_Vad_5 with seeded weights (tests/golden/gen_golden_v5_from_python_reference.py).
Needs an MI355X: `-m gpu`.  Bar: per-chunk probability within 1e-4 (north star)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, run_cli
from oracle import oracle as O
from vadc_amd import synth, testtensor as tt
from vadc_amd.engine import Engine, VadcAmdError, MODEL_V5
from vadc_amd.staging import to_device, to_host

pytestmark = pytest.mark.gpu

PROB_TOL = 1e-4
V5_WEIGHTS = os.path.join(GOLDEN, "silero_v5_seeded.testtensor")
STREAMS = ["speech0", "speech1", "speech2", "zeros", "noise", "square"]


@pytest.fixture(scope="module")
def blob():
    return open(V5_WEIGHTS, "rb").read()


@pytest.fixture(scope="module")
def eng(blob):
    e = Engine(blob, max_streams=64, max_chunks_per_call=192, device=0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def orc(blob):
    return O.OracleV5(blob)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "python_reference_v5.npz"))


def streams512(S, n, seed0):
    """S streams x n 512-sample windows of the synthetic speech-like signal"""
    c = -(-n * 512 // 1536)
    return np.ascontiguousarray(synth.make_streams(S, c, seed0=seed0)[:, :n * 512])


def test_model_kind_comes_from_the_container(eng):
    c = eng.caps()
    assert c["model_kind"] == MODEL_V5 and c["is_silero_v5"] == 1 and c["lstm_steps_per_chunk"] == 1
    assert (c["input_size_min"], c["input_size_max"], c["context_size"], c["lstm_hidden_size"]) == (512, 512, 64, 128)
    assert eng.window == 512
    with pytest.raises(VadcAmdError):
        eng.stage_from_samples(np.zeros(512, np.float32), "magnitude")


@pytest.mark.parametrize("name", STREAMS)
def test_probabilities_and_state_match_python_reference(eng, gold, name):
    pcm = gold[f"pcm_{name}"]
    eng.reset_streams()
    p = eng.run(pcm.reshape(1, -1))[0]
    assert np.array_equal(p[:, 0], p[:, 1])
    assert float(np.abs(p[:, 1] - gold[f"probs64_{name}"]).max()) < PROB_TOL
    h, c = eng.get_state(0)
    assert float(np.abs(h.reshape(-1) - gold[f"h64_{name}"]).max()) < PROB_TOL
    assert float(np.abs(c.reshape(-1) - gold[f"c64_{name}"]).max()) < 5 * PROB_TOL


def test_f32_and_s16_inputs_agree(eng, gold):
    pcm = gold["pcm_speech2"].reshape(1, -1)
    eng.reset_streams(); a = eng.run(pcm)
    eng.reset_streams(); b = eng.run(pcm.astype(np.float32) / np.float32(32768))
    assert np.array_equal(a, b)


@pytest.mark.parametrize("S,n", [(1, 1), (3, 7), (16, 33), (24, 40), (64, 96)])
def test_probabilities_match_oracle_many_streams(eng, orc, S, n):
    pcm = streams512(S, n, seed0=5151 + S)
    eng.reset_streams()
    got = eng.run(pcm)[:, :, 1]
    ref = orc.forward_streams(pcm)
    assert float(np.abs(got - ref).max()) < PROB_TOL


def test_context_and_state_carry_across_calls(eng, orc):
    S, n = 5, 60
    pcm = streams512(S, n, seed0=99)
    eng.reset_streams()
    whole = eng.run(pcm)
    eng.reset_streams()
    parts = [eng.run(pcm[:, a * 512:b * 512]) for a, b in ((0, 1), (1, 2), (2, 31), (31, 60))]
    assert np.array_equal(np.concatenate(parts, axis=1), whole)      # bit-identical: the context reaches the next call
    # without the context the second call differs (the first 64 padded samples are the previous window's tail)
    eng.reset_streams()
    cold = eng.run(pcm[:, 512:1024])
    assert not np.array_equal(cold, whole[:, 1:2])


def test_forked_calls_and_fp32_recurrence(blob, orc):
    """calls of >= 2048 windows fork onto the engine's streams (encoder of call k+1 beside the recurrence of call k, deferred joins): context, state
    and the hand-off buffers carry over exactly as on one stream; the split-fp16 recurrence (default) and the fp32-MFMA one (option lstm=3) both
    hold the 1e-4 bar"""
    S, n = 64, 120
    pcm = streams512(S, n, seed0=2024)
    ref = orc.forward_streams(pcm)
    out = {}
    for lstm in (0, 3):
        e = Engine(blob, max_streams=S, max_chunks_per_call=n, device=0)
        try:
            e.set_option("lstm", lstm)
            whole = e.run(pcm)
            assert e.get_option("lstm_kernel") == (3 if lstm == 3 else 6)
            e.reset_streams()
            e.set_option("defer_join", 1)
            torch = pytest.importorskip("torch")
            d_in = [to_device(np.ascontiguousarray(pcm[:, a * 512:b * 512])) for a, b in ((0, 40), (40, 41), (41, 80), (80, 120))]
            d_out = [torch.empty(S, x.shape[1] // 512, 2, device="cuda") for x in d_in]
            st = torch.cuda.Stream()
            for x, y in zip(d_in, d_out):                        # forked, small (on the caller's stream), forked, forked
                e.run_device(x.data_ptr(), np.int16, S, y.shape[1], y.data_ptr(), st.cuda_stream)
            e.join(st.cuda_stream)
            st.synchronize()
            parts = np.concatenate([to_host(y) for y in d_out], axis=1)
            assert np.array_equal(parts, whole)
            assert float(np.abs(whole[:, :, 1] - ref).max()) < PROB_TOL
            out[lstm] = whole
        finally:
            e.close()
    assert float(np.abs(out[0] - out[3]).max()) < PROB_TOL


def test_encoder_forms_agree_and_fall_back(blob, orc):
    """round 6: the encoder's GEMMs (folded STFT, four k = 3 convs, the LSTM's input projection) run as split-fp16 MFMAs at fp32 accuracy (k_v5_encoder_h3:
    "frontend_kernel" 2) when every weight x 256 fits fp16's range and the basis has the real-DFT fold symmetries; option "encoder" = 3, a weight outside that
    range or a basis without the symmetries select the fp32-MFMA form (k_v5_encoder: 1).  Every form holds the 1e-4 bar against the oracle on ragged tiles of 16."""
    S, n = 21, 37                                               # 777 windows: 48 whole tiles of 16 and a ragged one, tiles across stream boundaries
    pcm = streams512(S, n, seed0=4711)
    ref = orc.forward_streams(pcm)
    out = {}
    for enc in (0, 3):
        e = Engine(blob, max_streams=S, max_chunks_per_call=n, device=0)
        try:
            e.set_option("encoder", enc)
            out[enc] = e.run(pcm)
            assert e.get_option("frontend_kernel") == (1 if enc == 3 else 2)
            assert float(np.abs(out[enc][:, :, 1] - ref).max()) < PROB_TOL
            e.reset_streams()
            assert np.array_equal(e.run(pcm.astype(np.float32) / np.float32(32768)), out[enc])       # f32 samples of 16-bit audio: the same integers
        finally:
            e.close()
    assert float(np.abs(out[0] - out[3]).max()) < PROB_TOL
    ts = tt.loads(blob)
    big = [(k, v.copy()) for k, v in ts]
    big[3][1][5, 7, 1] = 300.0                                  # x 256 leaves fp16's range
    odd = [(k, v.copy()) for k, v in ts]
    odd[0][1][40, 0, 100] *= 1.0000001                          # re row of bin 40 no longer even about tap 128
    for other in (big, odd):
        b2 = tt.dumps(other)
        e = Engine(b2, max_streams=S, max_chunks_per_call=n, device=0)
        try:
            got = e.run(pcm)
            assert e.get_option("frontend_kernel") == 1
            assert float(np.abs(got[:, :, 1] - O.OracleV5(b2).forward_streams(pcm)).max()) < PROB_TOL
        finally:
            e.close()


def test_encoder_h3_at_the_bench_shape(blob, orc):
    """256 streams x 288 windows (what tools/v5_rate.py and bench.py's configs["v5_256x288"] time) through the split-fp16 encoder, forked: sampled streams against the oracle"""
    S, n = 256, 288
    base = streams512(16, n, seed0=11)
    pcm = np.ascontiguousarray(np.tile(base, (S // 16, 1)))
    e = Engine(blob, max_streams=S, max_chunks_per_call=n, device=0)
    try:
        got = e.run(pcm)
        assert e.get_option("frontend_kernel") == 2
    finally:
        e.close()
    ref = orc.forward_streams(base[:6])
    assert float(np.abs(got[:6, :, 1] - ref).max()) < PROB_TOL
    assert np.array_equal(got[:16], got[16:32]) and np.array_equal(got[:16], got[-16:])


def test_reset_of_selected_streams_clears_context_and_state(eng):
    S, n = 4, 12
    pcm = streams512(S, n, seed0=7)
    eng.reset_streams()
    first = eng.run(pcm)
    eng.reset_streams(np.array([1, 3], np.int32))
    again = eng.run(pcm)
    assert np.array_equal(again[1], first[1]) and np.array_equal(again[3], first[3])
    assert not np.array_equal(again[0], first[0])


def test_device_buffers_and_caller_stream(eng, orc):
    torch = pytest.importorskip("torch")
    S, n = 20, 24
    pcm = streams512(S, n, seed0=31)
    d_in = to_device(pcm)
    d_out = torch.empty(S, n, 2, device="cuda")
    st = torch.cuda.Stream()
    eng.reset_streams(); eng.synchronize()
    with torch.cuda.stream(st):
        eng.run_device(d_in.data_ptr(), np.int16, S, n, d_out.data_ptr(), st.cuda_stream)
    st.synchronize()
    ref = orc.forward_streams(pcm)
    assert float(np.abs(to_host(d_out)[:, :, 1] - ref).max()) < PROB_TOL


def test_cli_with_v5_weights(gold):
    """the POSIX CLI (host/vadc_hip.c) takes the model kind and the 512-sample window from the container's caps (vadc.c:743-752 clamps the
    default 1536 to the backend's range; onnx_helpers.c:158-160 for v5)"""
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "host", "vadc_hip")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "host")])
    pcm = gold["pcm_speech0"]
    r = run_cli([exe, "--model", V5_WEIGHTS, "--raw_probabilities"], pcm.tobytes())
    assert r.returncode == 0, r.stderr.decode()
    assert "Running with sequence count 512" in r.stderr.decode()
    got = np.array([float(x) for x in r.stdout.decode().splitlines()], np.float32)
    assert got.size == gold["probs64_speech0"].size
    assert float(np.abs(got - gold["probs64_speech0"]).max()) <= PROB_TOL + 5e-7        # %f quantises to 5e-7


def test_save_and_restore_a_stream_onto_another_slot(eng):
    """h, c AND the 64-sample context are the state of a v5 stream: saved after 20 windows and restored onto another slot, the stream continues
    bit-identically; without the context it does not (vadc_amd_get_context / vadc_amd_set_context)"""
    n = 40
    pcm = streams512(3, n, seed0=321)
    eng.reset_streams()
    whole = eng.run(pcm)
    eng.reset_streams()
    eng.run(pcm[:, :20 * 512])
    h, c = eng.get_state(1)
    ctx = eng.get_context(1)
    assert np.array_equal(ctx, pcm[1, 20 * 512 - 64:20 * 512].astype(np.float32) / np.float32(32768))
    eng.reset_streams()
    other = np.ascontiguousarray(pcm[[2, 1, 0]])               # stream 1's continuation now runs in slot 1 of a freshly reset engine ...
    eng.set_state(1, h, c)
    eng.set_context(1, ctx)
    cont = eng.run(other[:, 20 * 512:])
    assert np.array_equal(cont[1], whole[1, 20:])
    eng.reset_streams()
    eng.set_state(1, h, c)                                     # ... and differs when the context is left behind
    assert not np.array_equal(eng.run(other[:, 20 * 512:])[1, :1], whole[1, 20:21])
