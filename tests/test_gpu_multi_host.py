"""host/vadc_hip_multi.c -- the multi-GPU host in C (one engine per device, one thread each, ncclGather of the probabilities to device 0 per step):
its gathered probabilities against the CPU oracle.  One device always (RCCL with one rank: the whole code path but the wire); two devices where the
box has them."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, WEIGHTS
from oracle import oracle as O           # checker
from vadc_amd import synth

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "host", "vadc_hip_multi")
PROB_TOL = 1e-4


def _exe():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "vadc_hip_multi"])
    return EXE


def _run_and_check(tmp_path, gpus, S, C, K):
    pcm = synth.make_streams(gpus * S, K * C, seed0=2200 + gpus)
    fin, fout = str(tmp_path / "in.s16"), str(tmp_path / "out.f32")
    np.ascontiguousarray(pcm).tofile(fin)
    r = subprocess.run([_exe(), "--model", WEIGHTS, "--gpus", str(gpus), "--streams-per-gpu", str(S), "--chunks", str(C), "--steps", str(K), "--pcm", fin, "--dump", fout],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == gpus and line["steps"] == K and line["scaling"] == "weak" and line["value"] > 0
    got = np.fromfile(fout, np.float32).reshape(K, gpus * S, C, 2)              # rank r's block = global streams [r S, (r + 1) S): contiguous blocks
    want = O.Oracle(open(WEIGHTS, "rb").read()).forward_streams(pcm)             # [gpus * S, K * C]
    got_p = np.concatenate([got[k, :, :, 1] for k in range(K)], axis=1)
    d = np.abs(got_p - want)
    assert float(d.max()) <= PROB_TOL, (float(d.max()), np.unravel_index(d.argmax(), d.shape))


def test_one_device_gather_matches_the_oracle(tmp_path):
    _run_and_check(tmp_path, 1, 19, 5, 4)                                        # a ragged stream tile, state carried over 4 steps, more steps than step buffers


def test_two_devices_gather_matches_the_oracle(tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    _run_and_check(tmp_path, 2, 19, 5, 4)


def test_bench_line_of_the_c_host():
    r = subprocess.run([_exe(), "--model", WEIGHTS, "--gpus", "1", "--streams-per-gpu", "64", "--chunks", "32", "--steps", "6", "--warmup", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "config"):
        assert k in line
    assert line["warmup"] == 2 and abs(line["value"] - 64 * 32 * 6 * 0.096 / (line["ms_per_step"] * 6e-3)) / line["value"] < 1e-3
