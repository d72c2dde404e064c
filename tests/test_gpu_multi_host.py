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
    assert line["total_streams"] == gpus * S and abs(line["value_per_gpu"] - line["value"] / gpus) <= 0.06 and f"WHOLE JOB over {gpus} GPU" in line["metric"]
    assert line["rccl"]["world_size"] == gpus and line["rccl"]["nccl_version"] > 0 and line["rccl"]["bytes_per_chunk"] == 4 and line["rccl"]["bytes_per_rank_and_step"] == 4 * S * C
    got = np.fromfile(fout, np.float32).reshape(K, gpus * S, C)                 # the speech probability alone (4 B per chunk on the wire); rank r's block = global streams [r S, (r + 1) S)
    want = O.Oracle(open(WEIGHTS, "rb").read()).forward_streams(pcm)             # [gpus * S, K * C]
    got_p = np.concatenate([got[k] for k in range(K)], axis=1)
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
    for k in ("metric", "value", "value_per_gpu", "total_streams", "rccl", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "config"):
        assert k in line
    assert line["warmup"] == 2 and abs(line["value"] - 64 * 32 * 6 * 0.096 / (line["ms_per_step"] * 6e-3)) / line["value"] < 1e-3


def test_devices_list_and_its_errors(tmp_path):
    """--devices names the HIP devices in rank order; a device that is not visible, a device named twice and more GPUs than are visible are refused with a message that
    says how many devices there are (no hang, no crash, exit code 1)"""
    import torch
    n = torch.cuda.device_count()
    fout = str(tmp_path / "out.f32")
    pcm = synth.make_streams(8, 2 * 3, seed0=2300)
    fin = str(tmp_path / "in.s16")
    np.ascontiguousarray(pcm).tofile(fin)
    base = [_exe(), "--model", WEIGHTS, "--streams-per-gpu", "8", "--chunks", "3", "--steps", "2"]
    r = subprocess.run(base + ["--devices", "0", "--pcm", fin, "--dump", fout], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.fromfile(fout, np.float32).reshape(2, 8, 3)
    want = O.Oracle(open(WEIGHTS, "rb").read()).forward_streams(pcm)
    assert float(np.abs(np.concatenate([got[k] for k in range(2)], axis=1) - want).max()) <= PROB_TOL
    for extra, text in ((["--devices", str(n)], "visible device"), (["--devices", "0,0"], "twice"), (["--gpus", str(n + 1)], "visible"), (["--devices", "0", "--gpus", "2"], "lists 1")):
        r = subprocess.run(base + extra, capture_output=True, text=True, timeout=120)
        assert r.returncode == 1 and text in r.stderr, (extra, r.returncode, r.stderr[-500:])
