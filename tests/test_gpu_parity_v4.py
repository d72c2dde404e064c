"""Silero v4 / 16 kHz on the HIP path (through the C-ABI) against the v4 CPU oracle and the goldens generated from the
reference's PyTorch class silero_vad.py::Silero_V4 (tests/golden/gen_golden_v4_from_python_reference.py).
Needs an MI355X: `-m gpu`.  Bar: per-chunk probability within 1e-4 (north star); stage taps within 2e-4 of float64 torch
(relative to the stage's magnitude)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, run_cli
from oracle import oracle as O
from vadc_amd import synth
from vadc_amd.engine import Engine, VadcAmdError, MODEL_V4
from vadc_amd.staging import to_device, to_host

pytestmark = pytest.mark.gpu

PROB_TOL = 1e-4
TAP_TOL = 2e-4
V4_WEIGHTS = os.path.join(GOLDEN, "silero_v4_16k.testtensor")
STREAMS = ["speech0", "speech1", "speech2", "zeros", "noise", "square"]


@pytest.fixture(scope="module")
def blob():
    return open(V4_WEIGHTS, "rb").read()


@pytest.fixture(scope="module")
def eng(blob):
    e = Engine(blob, max_streams=64, max_chunks_per_call=64, device=0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def orc(blob):
    return O.OracleV4(blob)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "python_reference_v4.npz"))


def f32(pcm):
    return pcm.astype(np.float32) / np.float32(32768)


def test_model_kind_comes_from_the_container(eng):
    c = eng.caps()
    assert c["model_kind"] == MODEL_V4 and c["lstm_steps_per_chunk"] == 3
    assert (c["input_size_min"], c["input_size_max"], c["output_stride"], c["silero_probability_out_index"]) == (512, 1536, 2, 1)
    with pytest.raises(VadcAmdError):
        eng.set_option("encoder", 3)                 # split-fp16 / fp32 GEMM forms exist for the v3.1 transformer layers only


@pytest.mark.parametrize("name", STREAMS)
def test_probabilities_match_python_reference(eng, gold, name):
    pcm = gold[f"pcm_{name}"]
    eng.reset_streams()
    p = eng.run(pcm.reshape(1, -1))[0]
    assert np.array_equal(p[:, 0], p[:, 1])          # v4 has one output; both slots carry it
    assert float(np.abs(p[:, 1] - gold[f"probs64_{name}"]).max()) < PROB_TOL


def test_probabilities_match_oracle_many_streams(eng, orc):
    S, n = 24, 40                                    # ragged vs the 16-stream LSTM tile
    pcm = synth.make_streams(S, n, seed0=4242)
    eng.reset_streams()
    got = eng.run(pcm)[:, :, 1]
    ref = orc.forward_streams(pcm)
    assert float(np.abs(got - ref).max()) < PROB_TOL


def test_chunk_group_pipeline_matches_single_launch(eng, orc):
    """64 streams x 64 chunks = 4096 chunks per call: the engine forks onto its two internal streams and pipelines four
    chunk groups (front end/encoder of group g+1 under the LSTM of group g); results must not depend on the grouping"""
    pcm = synth.make_streams(64, 64, seed0=777)
    out = {}
    for groups in (1, 4, 0):
        eng.set_option("groups", groups)
        eng.reset_streams()
        out[groups] = eng.run(pcm)[:, :, 1]
    eng.set_option("groups", 0)
    assert np.array_equal(out[1], out[4]) and np.array_equal(out[1], out[0])
    ref = orc.forward_streams(pcm[:6])
    assert float(np.abs(out[1][:6] - ref).max()) < PROB_TOL


@pytest.mark.parametrize("S,C,calls", [(1, 3, 3), (18, 7, 2), (129, 16, 2), (700, 3, 2), (2049, 1, 2)])
def test_shapes_sweep_against_oracle(blob, orc, S, C, calls):
    """ragged stream counts and several calls with carried state (ragged vs the 4-chunk front-end groups and 21-chunk last stage)"""
    e = Engine(blob, max_streams=S, max_chunks_per_call=C, device=0)
    base = synth.make_streams(min(S, 6), C * calls, seed0=2000 + S)
    pcm = np.ascontiguousarray(np.tile(base, ((S + base.shape[0] - 1) // base.shape[0], 1))[:S])
    got = np.concatenate([e.run(pcm[:, k * C * 1536:(k + 1) * C * 1536]) for k in range(calls)], axis=1)[:, :, 1]
    e.close()
    for s in sorted({0, S - 1, S // 2, min(S - 1, 16)}):
        want = orc.forward_stream(pcm[s])
        assert float(np.abs(got[s] - want).max()) < PROB_TOL, (s, float(np.abs(got[s] - want).max()))
    for s in range(base.shape[0], S, max(1, S // 7)):
        assert np.array_equal(got[s], got[s % base.shape[0]])


def test_config4_full_size_4096_streams(blob, orc):
    """BASELINE config 4 at its stated size (Silero v4, 4096 streams x 16 chunks per call: the >= 2048-stream scheduling branch, no LSTM CU
    partition): determinism, range, stream independence (the same audio in different slots gives the same bits), state carry over two calls =
    one call of twice the length, hipGraph replay = eager, and every one of the 64 distinct streams against the oracle"""
    import torch
    S, Cn = 4096, 16
    base = synth.make_streams(64, 2 * Cn, seed0=7400)
    pcm = np.ascontiguousarray(np.tile(base, (S // 64, 1)))
    e = Engine(blob, max_streams=S, max_chunks_per_call=2 * Cn, device=0)
    try:
        a = np.concatenate([e.run(pcm[:, : Cn * 1536]), e.run(pcm[:, Cn * 1536:])], axis=1)
        e.reset_streams()
        b = e.run(pcm)                                       # one call of 32 chunks
        assert np.array_equal(a, b)
        assert np.isfinite(a).all() and (a >= 0).all() and (a <= 1).all()
        assert np.array_equal(a[:64], a[64 * 17: 64 * 18]) and np.array_equal(a[:64], a[S - 64:])
        want = orc.forward_streams(base)
        assert float(np.abs(a[:64, :, 1] - want).max()) < PROB_TOL
        # graph replay at this size, two caller streams / buffers as bench.py drives it
        d_in = [to_device(np.ascontiguousarray(pcm[:, i * Cn * 1536:(i + 1) * Cn * 1536])) for i in range(2)]
        sts = [torch.cuda.Stream(), torch.cuda.Stream()]
        e.set_option("graph", 1)
        for rep in range(2):
            e.reset_streams()
            outs = [torch.empty((S, Cn, 2), dtype=torch.float32, device="cuda:0") for _ in range(2)]
            for i in range(2):
                with torch.cuda.stream(sts[i]):
                    e.run_device(d_in[i].data_ptr(), np.int16, S, Cn, outs[i].data_ptr(), sts[i].cuda_stream)
            torch.cuda.synchronize()
            g = np.concatenate([to_host(o) for o in outs], axis=1)
            assert np.array_equal(g, a), rep
        e.set_option("graph", 0)
    finally:
        e.close()


@pytest.mark.parametrize("ci", [0, 20])
def test_stage_taps(eng, gold, ci):
    x = f32(gold["pcm_speech0"])[ci * 1536:(ci + 1) * 1536]
    for stage, key in (("magnitude", "magnitude"), ("normalized", "normalized"), ("layer1", "l1"), ("layer2", "l2"),
                       ("layer3", "l3"), ("layer4", "l4")):
        got = eng.stage_from_samples(x, stage)[0]
        ref = gold[f"tap{ci}_{key}"]
        scale = max(1.0, float(np.abs(ref).max()))
        assert got.shape == ref.shape
        # log1p(2^20 m) amplifies the fp32 rounding noise of near-silent bins (|dm| ~ 1e-7 -> 0.1 in the log domain when
        # m ~ 1e-6); the stages downstream are insensitive to it (their taps and the probabilities hold the tight bar)
        tol = 2e-2 if stage == "normalized" else TAP_TOL * scale
        err = float(np.abs(got - ref).max())
        assert err < tol, (stage, err)


def test_frontend_variants(eng, orc, gold):
    """option frontend=1: the STFT tree kernel with the v4 geometry (pad 96, 24 frames) is bit-identical to the oracle's tree;
    frontend=0 (default): the folded GEMM on the matrix cores agrees with it to fp32 rounding of a 256-tap sum"""
    x = f32(gold["pcm_speech1"])[:7 * 1536]          # 7 chunks: ragged vs the 4-chunk workgroup iteration
    eng.set_option("frontend", 1)
    tree = eng.stage_from_samples(x, "magnitude")
    tree_n = eng.stage_from_samples(x, "normalized")
    eng.set_option("frontend", 0)
    gemm = eng.stage_from_samples(x, "magnitude")
    gemm_n = eng.stage_from_samples(x, "normalized")
    for i in range(7):
        h, c = orc.new_state()
        _, taps = orc.forward_chunk(x[i * 1536:(i + 1) * 1536], h, c, taps=True)
        assert np.array_equal(tree[i].view(np.uint32), taps["magnitude"].view(np.uint32))
    assert float(np.abs(gemm - tree).max()) < 2e-5 * max(1.0, float(np.abs(tree).max()))
    assert float(np.abs(gemm_n - tree_n).max()) < 5e-3     # log1p(2^20 m) amplifies rounding noise in near-silent bins


@pytest.mark.parametrize("n", [1, 7, 9, 23, 100])
def test_first_stage_register_kernel_agrees_with_the_k1_form(eng, orc, gold, n):
    """k_layer1_regs_v4 (the default at the 1536-sample window: the chunk by LDS-DMA into the wave's own buffer, two overlapping 16-column tiles,
    split-fp16 MFMAs over K = 516) against the K = 1 fp32-MFMA form of k_layer_mfma (option layer1=1): different kernels, the same math to fp32
    rounding; both the oracle's.  n: one chunk .. several chunks per wave of the persistent grid"""
    x = np.tile(f32(gold["pcm_speech1"])[: 25 * 1536], 4)[: n * 1536]
    try:
        eng.set_option("layer1", 1); a = eng.stage_from_samples(x, "layer1")
        eng.set_option("layer1", 0); b = eng.stage_from_samples(x, "layer1")
    finally:
        eng.set_option("layer1", 0)
    assert not np.array_equal(a.view(np.uint32), b.view(np.uint32))       # two different kernels did run
    assert float(np.abs(a - b).max()) < 5e-5 * max(1.0, float(np.abs(a).max())), float(np.abs(a - b).max())
    want = []
    for i in range(n):
        h, c = orc.new_state()
        _, taps = orc.forward_chunk(x[i * 1536:(i + 1) * 1536], h, c, taps=True)
        want.append(taps["l1"])
    assert float(np.abs(b - np.stack(want)).max()) < TAP_TOL * max(1.0, float(np.abs(b).max()))


def test_first_stage_over_many_rounds_of_its_input_ring(blob, orc):
    """k_layer1_regs_v4 with its ring of three k-block slabs per wave, twelve waves per workgroup, wave-major slots (round 5): 10,007 distinct chunks = 3.3
    rounds of the grid, every phase of the ring.  Every chunk against the K = 1 fp32-MFMA form, 48 spread over the rounds against the oracle."""
    from vadc_amd import synth
    n = 10007
    rng = np.random.default_rng(11)
    base = synth.make_streams(8, 16, seed0=57).astype(np.float32).reshape(-1, 1536) / np.float32(32768)
    x = base[rng.integers(0, base.shape[0], n)] * rng.uniform(0.05, 1.0, (n, 1)).astype(np.float32)
    e = Engine(blob, max_streams=256, max_chunks_per_call=40, device=0)
    try:
        e.set_option("layer1", 1); a = e.stage_from_samples(x, "layer1")
        e.set_option("layer1", 0); b = e.stage_from_samples(x, "layer1")
        assert e.get_option("layer1_kernel") == 0
    finally:
        e.close()
    assert not np.array_equal(a.view(np.uint32), b.view(np.uint32))
    scale = max(1.0, float(np.abs(a).max()))
    d = np.abs(a - b).reshape(n, -1).max(axis=1)
    assert float(d.max()) < 2e-4 * scale, (float(d.max()), int(d.argmax()))      # two fp32-grade forms of one stage: a ring slip would be a gross error
    for i in list(np.linspace(0, n - 1, 48).astype(int)) + [int(d.argmax())]:      # ... and the chunk the two disagree on most is the oracle's too
        h, c = orc.new_state()
        _, taps = orc.forward_chunk(x[i], h, c, taps=True)
        assert float(np.abs(b[i] - taps["l1"]).max()) < TAP_TOL * scale, int(i)


@pytest.mark.parametrize("variant", [0, 1])
def test_probabilities_both_frontends(eng, gold, variant):
    eng.set_option("frontend", variant)
    try:
        for name in ("speech0", "noise"):
            eng.reset_streams()
            p = eng.run(gold[f"pcm_{name}"].reshape(1, -1))[0]
            assert float(np.abs(p[:, 1] - gold[f"probs64_{name}"]).max()) < PROB_TOL
    finally:
        eng.set_option("frontend", 0)


def test_state_is_carried_and_split_invariant(eng, gold):
    pcm = gold["pcm_speech2"].reshape(1, -1)
    eng.reset_streams()
    whole = eng.run(pcm)
    eng.reset_streams()
    parts = np.concatenate([eng.run(pcm[:, :17 * 1536]), eng.run(pcm[:, 17 * 1536:])], axis=1)
    assert float(np.abs(whole - parts).max()) < 1e-6
    h, c = eng.get_state(0)
    assert float(np.abs(h - gold["h64_speech2"]).max()) < 2e-4 and float(np.abs(c - gold["c64_speech2"]).max()) < 2e-4


def test_lstm_decoder_tap(eng, orc, gold):
    """LSTM + v4 decoder (mean_t sigmoid(w . relu(h_t) + b)) fed with the golden encoder output"""
    enc = gold["tap20_l4"].astype(np.float32).reshape(1, 1, 64, 3)
    eng.reset_streams()
    eng.set_state(0, gold["tap20_h_in"].astype(np.float32), gold["tap20_c_in"].astype(np.float32))
    p = eng.lstm_decoder(enc)[0, 0]
    assert abs(float(p[1]) - float(gold["tap20_prob"])) < PROB_TOL


def test_device_pointer_and_graph_paths(eng, gold):
    import torch
    pcm = np.ascontiguousarray(np.tile(gold["pcm_speech0"][:32 * 1536], (8, 1)))
    eng.reset_streams()
    ref = eng.run(pcm)
    d_in = to_device(pcm)
    d_out = torch.empty((8, 32, 2), dtype=torch.float32, device="cuda:0")
    for graph in (0, 1):
        eng.set_option("graph", graph)
        eng.reset_streams()
        st = torch.cuda.Stream(device="cuda:0")
        eng.run_device(d_in.data_ptr(), np.int16, 8, 32, d_out.data_ptr(), hip_stream=st.cuda_stream)
        st.synchronize()
        assert float(np.abs(to_host(d_out) - ref).max()) < 1e-6
    eng.set_option("graph", 0)


def test_cli_with_v4_weights(gold):
    """the POSIX CLI (host/vadc_hip.c) takes the model kind from the weights container: `--model silero_v4_16k.testtensor`"""
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "host", "vadc_hip")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "host")])
    pcm = gold["pcm_speech0"]
    r = run_cli([exe, "--model", V4_WEIGHTS, "--raw_probabilities"], pcm.tobytes())
    assert r.returncode == 0, r.stderr.decode()
    got = np.array([float(x) for x in r.stdout.decode().splitlines()], np.float32)
    assert got.size == gold["probs64_speech0"].size
    assert float(np.abs(got - gold["probs64_speech0"]).max()) <= PROB_TOL + 5e-7        # %f quantises to 5e-7
    r = run_cli([exe, "--model", V4_WEIGHTS], pcm.tobytes())
    sec, _ = O.segments(gold["probs64_speech0"].astype(np.float32))
    assert r.stdout.decode().splitlines() == ["%.2f,%.2f" % (a, b) for a, b in sec]


# ---------------------------------------------------------------------------------------------- window sizes (vadc --sequence_count)
@pytest.mark.parametrize("window", [512, 768, 1024, 1280])
def test_window_sizes_vs_python_reference_and_oracle(blob, orc, gold, window):
    """Silero v4 with 512- ... 1280-sample windows (option "window": the v4 graph takes 512 ... 1536 samples, onnx_helpers.c:164-170; 768 and 1280 -- 12 and 20
    frames, odd lengths in the strided stages: 12 -> 6 -> 3 -> 2, 20 -> 10 -> 5 -> 3 -- since round 5): caps report the range, probabilities against the float64
    PyTorch goldens and the oracle, many ragged streams, state carried over calls, forked calls"""
    gw = np.load(os.path.join(GOLDEN, "python_reference_v4_windows.npz" if window in (512, 1024) else "python_reference_v4_windows_768_1280.npz"))
    e = Engine(blob, max_streams=40, max_chunks_per_call=150, device=0)
    try:
        c = e.caps()
        assert (c["input_size_min"], c["input_size_max"], c["window_samples"]) == (512, 1536, 1536)
        e.set_window(window)
        c = e.caps()
        assert c["input_size_step"] == 64
        assert c["window_samples"] == window and c["lstm_steps_per_chunk"] == {512: 1, 768: 2, 1024: 2, 1280: 3}[window] and e.get_option("window") == window
        for name in ("speech0", "speech1", "noise", "square"):
            pcm = gold[f"pcm_{name}"]
            pcm = pcm[: (pcm.size // window) * window]
            e.reset_streams()
            p = e.run(pcm.reshape(1, -1))[0]
            assert np.array_equal(p[:, 0], p[:, 1])
            assert float(np.abs(p[:, 1] - gw[f"probs64_w{window}_{name}"]).max()) < PROB_TOL, name
        # ragged stream count, two calls with carried state, enough chunks to fork (37 x 60 = 2220 chunks per call)
        S, n = 37, 120
        pcm = synth.make_streams(S, (n * window + 1535) // 1536, seed0=6100 + window)[:, : n * window]
        e.reset_streams()
        got = np.concatenate([e.run(pcm[:, : 60 * window]), e.run(pcm[:, 60 * window:])], axis=1)[:, :, 1]
        want = orc.forward_streams(pcm, window=window)
        assert float(np.abs(got - want).max()) < PROB_TOL
        # stage taps have the window's shapes
        x = f32(gold["pcm_speech0"])[: 5 * window]
        assert e.stage_from_samples(x, "magnitude").shape == (5, 129, window // 64)
        assert e.stage_from_samples(x, "layer4").shape == (5, 64, {512: 1, 768: 2, 1024: 2, 1280: 3}[window])
        with pytest.raises(VadcAmdError):
            e.set_option("window", 1000)                              # not a multiple of 64
        with pytest.raises(VadcAmdError):
            e.set_option("window", 1792)
        e.set_window(1536)
        e.reset_streams()
        p = e.run(gold["pcm_speech0"].reshape(1, -1))[0]
        assert float(np.abs(p[:, 1] - gold["probs64_speech0"]).max()) < PROB_TOL
    finally:
        e.close()


def test_window_option_is_v4_only():
    from conftest import WEIGHTS
    e = Engine(open(WEIGHTS, "rb").read(), max_streams=1, max_chunks_per_call=1, device=0)
    try:
        c = e.caps()
        assert (c["input_size_min"], c["input_size_max"]) == (1536, 1536)      # the C backend's contract (silero.h:41-42)
        with pytest.raises(VadcAmdError):
            e.set_option("window", 512)
        e.set_option("window", 1536)
    finally:
        e.close()


def test_cli_sequence_count_selects_the_v4_window(gold):
    """`--sequence_count` (vadc.c:743-752, 1117): clamped to the backend's range; with a Silero v4 container every multiple of 64 in 512 .. 1536 is run (other
    values rounded down to one), one %f line per FULL chunk of that size"""
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "host", "vadc_hip")
    gw = dict(np.load(os.path.join(GOLDEN, "python_reference_v4_windows.npz")))
    gw.update(np.load(os.path.join(GOLDEN, "python_reference_v4_windows_768_1280.npz")))
    pcm = gold["pcm_speech0"]
    gw.update(np.load(os.path.join(GOLDEN, "python_reference_v4_windows_64.npz")))
    for arg, window in (("512", 512), ("1024", 1024), ("1100", 1088), ("100", 512), ("9999", 1536), ("768", 768), ("1000", 960), ("1280", 1280), ("1500", 1472), ("576", 576)):
        r = run_cli([exe, "--model", V4_WEIGHTS, "--raw_probabilities", "--sequence_count", arg], pcm.tobytes())
        assert r.returncode == 0, r.stderr.decode()
        assert f"Running with sequence count {window}" in r.stderr.decode()
        got = np.array([float(x) for x in r.stdout.decode().splitlines()], np.float32)
        want = gw[f"probs64_w{window}_speech0"] if window != 1536 else gold["probs64_speech0"]
        assert got.shape == want.shape and float(np.abs(got - want).max()) < PROB_TOL + 5e-7



def test_long_streams_around_a_geometry_boundary_vs_the_float64_reference():
    """640 (560) consecutive chunks of two synthetic streams at 832 / 896 / 960 / 1024 samples per chunk -- three windows that run the 1024-sample geometry with masked frames
    and that geometry itself -- against the reference's PyTorch class in float64 (tests/golden/python_reference_v4_long_windows.npz, committed generator): within 1e-4
    everywhere (measured: <= 6.4e-5).  The fp32 ORACLE is not the yardstick here: on stream 2 at 960 samples it sits 1.5e-4 from the float64 statement itself (a quiet
    passage around chunk 601, tests/test_oracle_v4.py), where the engine is 9e-6 from it."""
    g = np.load(os.path.join(GOLDEN, "python_reference_v4_long_windows.npz"))
    blob = open(V4_WEIGHTS, "rb").read()
    base = synth.make_streams(16, 400, seed0=52000)
    for w in (832, 896, 960, 1024):
        n = 640 if w < 1024 else 560
        e = Engine(blob, max_streams=2, max_chunks_per_call=80, device=0)
        try:
            e.set_window(w)
            pcm = np.ascontiguousarray(base[[2, 12], : n * w])
            got = np.concatenate([e.run(pcm[:, i * w:(i + 80) * w]) for i in range(0, n, 80)], axis=1)[:, :, 1]
        finally:
            e.close()
        for j, s_ in enumerate((2, 12)):
            d = float(np.abs(got[j] - g[f"probs64_w{w}_s{s_}"]).max())
            assert d <= 1e-4, (w, s_, d)


@pytest.mark.parametrize("window", [576, 832, 960, 1088, 1408, 1472])      # (one to three frames short of each built geometry; all nine in-between windows: tests/test_oracle_v4.py on the CPU)
def test_windows_of_every_multiple_of_64(blob, orc, gold, window):
    """round 6: the reference's onnxruntime path admits every count in 512 ... 1536 (onnx_helpers.c:164-170); the engine serves every multiple of 64 samples.  A
    window that is no multiple of 256 runs the next larger built geometry -- the chunk re-laid-out so that its first frames are the window's own frames (samples,
    then the 96 the right reflect pad reads), the surplus steps masked stage by stage -- with the LSTM steps of its own frame count: probabilities against the float64
    PyTorch goldens (Silero_V4 at that length) and the oracle on ragged streams, f32 = s16 bit for bit, state carried over calls, forked calls, graph replay"""
    import torch
    g = np.load(os.path.join(GOLDEN, "python_reference_v4_windows_64.npz"))
    f = window // 64
    t1 = (f + 1) // 2; t2 = (t1 + 1) // 2; t3 = (t2 + 1) // 2
    e = Engine(blob, max_streams=40, max_chunks_per_call=150, device=0)
    try:
        e.set_window(window)
        c = e.caps()
        assert c["window_samples"] == window and c["lstm_steps_per_chunk"] == t3 and c["input_size_step"] == 64
        for name in ("speech0", "speech1", "square"):
            pcm = gold[f"pcm_{name}"]
            pcm = pcm[: (pcm.size // window) * window]
            e.reset_streams()
            p = e.run(pcm.reshape(1, -1))[0]
            assert float(np.abs(p[:, 1] - g[f"probs64_w{window}_{name}"]).max()) < PROB_TOL, name
            e.reset_streams()
            assert float(np.abs(e.run(f32(pcm).reshape(1, -1))[0] - p).max()) < 5e-5      # f32 samples: k_frontend_gemm instead of k_frontend_gemm2, the same re-laid-out chunk
        S, n = 37, 120
        pcm = synth.make_streams(S, (n * window + 1535) // 1536, seed0=6300 + window)[:, : n * window]
        e.reset_streams()
        got = np.concatenate([e.run(pcm[:, : 60 * window]), e.run(pcm[:, 60 * window:])], axis=1)[:, :, 1]
        want = orc.forward_streams(pcm, window=window)
        assert float(np.abs(got - want).max()) < PROB_TOL
        # deferred joins + graph replay from device buffers: the same bits as the synchronous calls
        d_in = [to_device(np.ascontiguousarray(pcm[:, k * 60 * window:(k + 1) * 60 * window])) for k in range(2)]
        d_out = [torch.empty(S, 60, 2, device="cuda") for _ in range(2)]
        st = torch.cuda.Stream()
        e.set_option("defer_join", 1); e.set_option("graph", 1)
        for rep_ in range(2):
            e.reset_streams()
            for k in range(2):
                e.run_device(d_in[k].data_ptr(), np.int16, S, 60, d_out[k].data_ptr(), st.cuda_stream)
            e.join(st.cuda_stream); st.synchronize()
            assert np.array_equal(np.concatenate([to_host(o) for o in d_out], axis=1)[:, :, 1], got), rep_
        e.set_option("defer_join", 0); e.set_option("graph", 0)
        with pytest.raises(VadcAmdError):
            e.stage_from_samples(f32(gold["pcm_speech0"])[: 2 * window], "magnitude")       # stage taps exist at the built windows
    finally:
        e.close()


@pytest.mark.parametrize("window", [320, 448, 704])
def test_8khz_windows_of_every_multiple_of_64(gold, window):
    g = np.load(os.path.join(GOLDEN, "python_reference_v4_windows_64.npz"))
    blob8 = open(os.path.join(GOLDEN, "silero_v4_8k.testtensor"), "rb").read()
    orc8 = O.OracleV4(blob8)
    e = Engine(blob8, max_streams=40, max_chunks_per_call=300, device=0)
    try:
        e.set_window(window)
        for name in ("speech0", "speech1", "square"):
            pcm = gold[f"pcm_{name}"]
            pcm = pcm[: (pcm.size // window) * window]
            e.reset_streams()
            p = e.run(pcm.reshape(1, -1))[0]
            assert float(np.abs(p[:, 1] - g[f"probs64_8k_w{window}_{name}"]).max()) < PROB_TOL, name
        S, n = 21, 150
        pcm = synth.make_streams(S, (n * window + 1535) // 1536, seed0=6400 + window)[:, : n * window]
        e.reset_streams()
        got = e.run(pcm)[:, :, 1]
        assert float(np.abs(got - orc8.forward_streams(pcm, window=window)).max()) < PROB_TOL
    finally:
        e.close()


# ---------------------------------------------------------------------------------------------- the 8 kHz branch of the v4 graph
V4_8K_WEIGHTS = os.path.join(GOLDEN, "silero_v4_8k.testtensor")


@pytest.mark.parametrize("window", [768, 512, 256])
def test_8khz_branch_vs_python_reference_and_oracle(gold, window):
    """the 37-tensor container (the graph's `model_8k.*` weights; third strided conv with stride 1, silero_vad.py:178-181): caps say 8000 Hz and
    256 ... 768-sample windows; probabilities against the float64 PyTorch goldens (Silero_V4(8000)) and the oracle; ragged streams, forked calls"""
    g8 = np.load(os.path.join(GOLDEN, "python_reference_v4_8k.npz"))
    blob8 = open(V4_8K_WEIGHTS, "rb").read()
    orc8 = O.OracleV4(blob8)
    e = Engine(blob8, max_streams=40, max_chunks_per_call=300, device=0)
    try:
        c = e.caps()
        assert (c["model_kind"], c["sample_rate"], c["input_size_min"], c["input_size_max"], c["window_samples"]) == (MODEL_V4, 8000, 256, 768, 768)
        if window != 768:
            e.set_window(window)
        assert e.caps()["window_samples"] == window and e.caps()["lstm_steps_per_chunk"] == window // 256
        for name in ("speech0", "speech1", "noise", "square"):
            pcm = gold[f"pcm_{name}"]
            pcm = pcm[: (pcm.size // window) * window]
            e.reset_streams()
            p = e.run(pcm.reshape(1, -1))[0]
            assert float(np.abs(p[:, 1] - g8[f"probs64_w{window}_{name}"]).max()) < PROB_TOL, name
        S, n = 37, 120
        pcm = synth.make_streams(S, (n * window + 1535) // 1536, seed0=6900 + window)[:, : n * window]
        e.reset_streams()
        got = np.concatenate([e.run(pcm[:, : 60 * window]), e.run(pcm[:, 60 * window:])], axis=1)[:, :, 1]
        want = orc8.forward_streams(pcm, window=window)
        assert float(np.abs(got - want).max()) < PROB_TOL
        x = f32(gold["pcm_speech0"])[: 5 * window]
        assert e.stage_from_samples(x, "layer3").shape == (5, 32, window // 256) and e.stage_from_samples(x, "layer4").shape == (5, 64, window // 256)
        with pytest.raises(VadcAmdError):
            e.set_option("window", 1536)
    finally:
        e.close()


def test_cli_with_the_8khz_container(gold):
    import subprocess
    from conftest import ROOT
    g8 = np.load(os.path.join(GOLDEN, "python_reference_v4_8k.npz"))
    r = subprocess.run([os.path.join(ROOT, "host", "vadc_hip"), "--model", V4_8K_WEIGHTS, "--raw_probabilities"], input=gold["pcm_speech0"].tobytes(),
                       capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert "Running with sequence count 768" in r.stderr.decode()            # the default 1536 clamped to the branch's maximum (vadc.c:743-752)
    got = np.array([float(x) for x in r.stdout.decode().splitlines()], np.float32)
    assert float(np.abs(got - g8["probs64_w768_speech0"]).max()) < PROB_TOL + 5e-7


def test_first_stage_k1_form_on_every_remainder_of_chunks_per_workgroup(blob, orc):
    """k_layer_mfma's K = 1 form of the v4 first stage (option "layer1" = 1 -- the fallback of k_layer1_regs_v4: a weight outside fp16's range, a failed
    self-check) runs 8 waves / 5 chunks per workgroup (the middle chunk in two pieces that overlap by four steps, so every lane finds its depthwise-conv
    neighbours in its own wave): the oracle's probabilities on every remainder of chunks per workgroup"""
    for S, Cn in ((3, 13), (64, 40), (1, 1), (2, 5), (7, 6)):
        pcm = synth.make_streams(S, Cn, seed0=77 + S)
        e = Engine(blob, max_streams=S, max_chunks_per_call=Cn, device=0)
        try:
            e.set_option("layer1", 1)
            got = e.run(pcm)
            assert e.get_option("layer1_kernel") == 1
        finally:
            e.close()
        assert float(np.abs(got[:, :, 1] - orc.forward_streams(pcm)).max()) < PROB_TOL


def test_8khz_container_without_dft_symmetries_is_refused():
    """the 8 kHz branch runs on the GEMM front end only (the tree kernel is built for the 1536-sample / 24-frame geometry of the 16 kHz branch); a 37-tensor
    container whose basis lacks the real-DFT symmetries that front end needs used to be accepted and would have read 1536-sample chunks from
    768-sample buffers: vadc_amd_create refuses it"""
    from vadc_amd import testtensor as tt
    ts = tt.load(V4_8K_WEIGHTS)
    basis = ts[0][1].copy()
    basis.reshape(-1).view(np.uint32)[5 * 256 + 37] += 1          # one ulp in one tap breaks the mirror symmetry
    ts[0] = (ts[0][0], basis)
    with pytest.raises(VadcAmdError) as ei:
        Engine(tt.dumps(ts), max_streams=2, max_chunks_per_call=4, device=0)
    assert ei.value.code == -2 and "8 kHz" in str(ei.value)


def test_a_rejected_option_leaves_captured_graphs_alone(blob):
    """vadc_amd_set_option validates before it drops the captured graphs: after a rejected call the next replay is a replay (same bits, and the
    engine still reports graph mode)"""
    pcm = synth.make_streams(4, 6, seed0=77)
    e = Engine(blob, max_streams=4, max_chunks_per_call=6, device=0)
    try:
        e.set_option("graph", 1)
        a = e.run(pcm)
        with pytest.raises(VadcAmdError):
            e.set_option("window", 777)
        with pytest.raises(VadcAmdError):
            e.set_option("no_such_option", 1)
        e.reset_streams()
        b = e.run(pcm)
        assert e.get_option("graph") == 1
    finally:
        e.close()
    assert np.array_equal(a, b)
