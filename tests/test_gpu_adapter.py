"""The literal backend trio (include/vadc_backend_hip.h: backend_init / backend_create_tensors / backend_run, silero.h:48-81) EXECUTED on the GPU:
tests/c/adapter_run.c drives it like vadc.c does (config setup :686-795, process_chunks :56-103, process_chunks_v5 :105-162) over layout mirrors of
the reference types, and its probabilities are compared with the goldens generated from the reference (C backend for v3.1, the reference's PyTorch
classes for v4 windows and the v5 shapes)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(ROOT, "tests", "golden")
PROB_TOL = 1e-4


@pytest.fixture(scope="module")
def runner(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("adapter") / "adapter_run")
    lib = os.path.join(ROOT, "vadc_amd")
    subprocess.check_call(["gcc", "-std=gnu11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "c"),
                           os.path.join(ROOT, "tests", "c", "adapter_run.c"), "-L", lib, "-lvadc_amd", f"-Wl,-rpath,{lib}", "-o", exe])
    return exe


def run_adapter(runner, tmp_path, weights, batch, seq, pcm):
    x = (pcm.astype(np.float32) / np.float32(32768))             # vadc.c:883,898
    fin, fout = str(tmp_path / "in.f32"), str(tmp_path / "out.f32")
    x.tofile(fin)
    r = subprocess.run([runner, weights, str(batch), str(seq), fin, fout], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return np.fromfile(fout, np.float32), r.stderr


@pytest.mark.parametrize("batch", [1, 7, 96])
def test_trio_v31_against_the_c_reference(runner, tmp_path, batch):
    """Silero v3.1 through backend_init -> backend_create_tensors -> backend_run at --batch 1 / 7 / 96: every full batch matches the reference C
    backend's probabilities (c_reference_v31.npz) to 1e-4; a zero-padded last batch (48 chunks at batch 7 / 96) is run like the reference runs it --
    its padded windows pass through the LSTM after the real ones, so the real windows still match"""
    py = np.load(os.path.join(GOLDEN, "python_reference_v31.npz"))
    ref = np.load(os.path.join(GOLDEN, "c_reference_v31.npz"))
    w = os.path.join(GOLDEN, "reference_fixtures", "silero_v31_16k.testtensor")
    for name in ("speech0", "noise"):
        got, log = run_adapter(runner, tmp_path, w, batch, 1536, py["pcm_" + name])
        want = ref["probs_" + name][:, 1]
        assert got.shape == want.shape and f"batch {batch}" in log
        assert float(np.abs(got - want).max()) <= PROB_TOL, name


def test_trio_v5_row_compaction(runner, tmp_path):
    """a Silero v5 container: the caller builds rows of 64 + 512 samples and carries the context itself (vadc.c:105-162); backend_run takes the windows
    and keeps the context on the device -- the probabilities are the PyTorch class's (python_reference_v5.npz), at batch 1 and at batch 8"""
    g = np.load(os.path.join(GOLDEN, "python_reference_v5.npz"))
    w = os.path.join(GOLDEN, "silero_v5_seeded.testtensor")
    for batch in (1, 8):
        got, log = run_adapter(runner, tmp_path, w, batch, 512, g["pcm_speech1"])
        want = g["probs64_speech1"]
        assert got.shape == want.shape and "v5 1" in log
        assert float(np.abs(got - want).max()) <= PROB_TOL, batch


@pytest.mark.parametrize("seq", [512, 768, 1024, 1280, 1536, 960, 1472])      # (960, 1472: windows between the built ones -- every multiple of 64 is served)
def test_trio_v4_forwards_the_sequence_count(runner, tmp_path, seq):
    """a Silero v4 container accepts --sequence_count 512 ... 1536 (caps.input_size_min / max as ort_init reports them, onnx_helpers.c:164-170); the
    caller sizes its buffers for THAT window (vadc.c:743-781), so the adapter must make the engine run it (it used to keep 1536 and read past the
    caller's buffer).  Probabilities against the reference's PyTorch class per window size."""
    py = np.load(os.path.join(GOLDEN, "python_reference_v4.npz"))
    w = os.path.join(GOLDEN, "silero_v4_16k.testtensor")
    pcm = py["pcm_speech0"]
    gfile = "python_reference_v4_windows.npz" if seq in (512, 1024) else ("python_reference_v4_windows_768_1280.npz" if seq in (768, 1280) else "python_reference_v4_windows_64.npz")
    want = py["probs64_speech0"] if seq == 1536 else np.load(os.path.join(GOLDEN, gfile))[f"probs64_w{seq}_speech0"]
    got, log = run_adapter(runner, tmp_path, w, 1, seq, pcm)
    assert got.shape == want.shape and f"of {seq} samples" in log
    assert float(np.abs(got - want).max()) <= PROB_TOL


def test_trio_v4_refuses_a_window_it_cannot_run(runner, tmp_path):
    """1000 lies inside [input_size_min, input_size_max] but is no multiple of 64 samples (a count in between is not built): the adapter aborts with a message (like
    an onnxruntime error, onnx_helpers.h:5-14) instead of running the wrong window"""
    py = np.load(os.path.join(GOLDEN, "python_reference_v4.npz"))
    w = os.path.join(GOLDEN, "silero_v4_16k.testtensor")
    x = (py["pcm_noise"].astype(np.float32) / np.float32(32768))
    fin, fout = str(tmp_path / "in.f32"), str(tmp_path / "out.f32")
    x.tofile(fin)
    r = subprocess.run([runner, w, "1", "1000", fin, fout], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "sequence_count 1000" in r.stderr
