"""`python bench.py --gpus N` started WITHOUT a launcher must bring up N ranks itself (the driver's SCALE runs call it that way) and report the
number of ranks the process group saw.  `--dry-run` runs exactly that skeleton -- spawn before any torch / HIP import, 127.0.0.1 rendezvous,
contiguous stream blocks, the per-step gather through vadc_amd.shard.ProbabilityGather (the helper the GPU path uses), barrier,
max-over-ranks timing, rank 0 prints ONE JSON line -- over gloo on the CPU with a stand-in for the engine."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(*args, env=None):
    r = subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def _ranks_that_fit(n):
    """a one-GPU box of this pool lets six processes onto its card at once: when this (pytest) process holds a GPU context of its own already -- the files ran in another
    order than the default one, in which this file comes before the first that touches the GPU --, six ranks would be the seventh process and the box ends the run"""
    import torch
    return n - 1 if n >= 6 and torch.cuda.is_initialized() else n


def _assert_n_rank_schema(d, n, S, Cn, c5_S, c5_C, backend):
    """what the first 8-GPU run must carry to be decisive (BASELINE config 5): `value` = the whole job under a label that says so, the per-GPU figure
    beside it, the totals, what the PROCESS GROUP reports about the one collective (4 B per chunk), and config 5's shape timed under the same ranks"""
    assert d["n_gpus"] == n and d["total_streams"] == S * n and d["scaling"] == "weak"
    assert abs(d["value_per_gpu"] - d["value"] / n) <= 0.06
    assert d["config"]["streams_per_gpu"] == S and d["config"]["chunks_per_step"] == Cn
    if n == 1:
        assert "per GPU" in d["metric"] and "WHOLE JOB" not in d["metric"] and d["rccl"]["world_size"] == 1 and "configs" not in d or not any(k.startswith("1x") for k in d.get("configs", {}))
        return
    assert f"WHOLE JOB over {n} GPUs" in d["metric"] and "value_per_gpu" in d["metric"]
    r = d["rccl"]
    assert r["backend"] == backend and r["world_size"] == n and r["bytes_per_chunk"] == 4 and r["bytes_per_rank_and_step"] == 4 * S * Cn
    assert "efficiency_vs_1gpu" in d
    c = d["configs"][f"{n}x{c5_S}x{c5_C}"]
    for k in ("value", "per_gpu", "ms_per_step", "steps", "n_gpus", "total_streams", "streams_per_gpu", "chunks_per_step", "rccl", "efficiency_vs_1gpu", "one_gpu_figure"):
        assert k in c, k
    assert c["n_gpus"] == n and c["total_streams"] == n * c5_S and c["streams_per_gpu"] == c5_S and c["chunks_per_step"] == c5_C
    assert abs(c["per_gpu"] - c["value"] / n) <= 0.06 and c["rccl"]["world_size"] == n and c["rccl"]["bytes_per_rank_and_step"] == 4 * c5_S * c5_C
    want = n * c5_S * c5_C * 0.096 / (c["ms_per_step"] * 1e-3)
    assert abs(c["value"] - want) / want < 1e-2


@pytest.mark.parametrize("n", [1, 2, 3, 8])
def test_bench_spawns_its_own_ranks(n, tmp_path):
    """world 8 = the node the driver's scaling run uses (BASELINE config 5): spawn, rendezvous, contiguous stream blocks, the per-step gather and max-over-ranks
    timing with EIGHT ranks, over gloo; every rank keeps to its own share of the CPUs this job may use; the line has the N-rank schema, and with a 1-GPU
    figure at hand (the cache an N = 1 run leaves, or --one-gpu-json) the efficiencies are filled in"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    one = tmp_path / "one.json"
    one.write_text(json.dumps({"5x3": 1000.0, "7x2": 500.0}))
    d = _run("--gpus", str(n), "--dry-run", "--steps", "4", "--warmup", "0", "--streams", "5", "--chunks-per-step", "3", "--config5-shape", "7x2", "--config5-steps", "3",
             "--one-gpu-json", str(one), env=env)
    assert d["n_gpus"] == n and d["dry_run"] is True and d["gather_verified"] is True
    assert d["total_streams"] == 5 * n and d["steps"] == 4 and d["scaling"] == "weak"
    have = len(os.sched_getaffinity(0))
    assert d["rank_cpus"] == (have // n if n > 1 and have >= n else have)
    _assert_n_rank_schema(d, n, 5, 3, 7, 2, "gloo")
    if n > 1:
        assert abs(d["efficiency_vs_1gpu"] - d["value"] / n / 1000.0) < 1e-3
        c = d["configs"][f"{n}x7x2"]
        assert c["one_gpu_figure"] == 500.0 and abs(c["efficiency_vs_1gpu"] - c["value"] / n / 500.0) < 1e-3


def test_default_config5_shape_is_baselines():
    """BASELINE config 5 = 32,768 streams sharded 4096 per GPU: what `--gpus 8` times beside the headline unless told otherwise"""
    src = open(BENCH).read()
    assert '"--config5-shape", default="4096x16"' in src
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "32768 total streams sharded 4096/GPU" in base["configs"][4]


def test_bench_runs_as_one_rank_under_a_launcher():
    """with RANK / WORLD_SIZE in the environment (torch.distributed.run) the process is one rank and does not spawn: two such processes rendezvous"""
    from test_multi_rank_gloo import _free_port
    port = str(_free_port())
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--steps", "2", "--streams", "4", "--chunks-per-step", "2"],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    d = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["gather_verified"] is True
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]           # only rank 0 prints


def test_cpu_baseline_worker_reports_a_rate():
    r = subprocess.run([sys.executable, BENCH, "--cpu-worker", "--cpu-seconds", "0.3", "--cpu-batch", "96"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["kind"] in ("reference", "port") and d["chunks_per_s"] > 100


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 6])       # (six ranks: as many processes as a one-GPU box lets onto its card at once)
def test_multi_rank_gpu_code_path_on_one_gpu(n):
    """`--one-gpu-rehearsal`: N ranks share GPU 0 and gather over gloo through the host -- a one-GPU box cannot form an RCCL group, but everything
    else of the multi-rank GPU path runs: per-rank engines, deferred joins, the side-stream gather behind vadc_amd_join, per-buffer gather events,
    barrier and max-over-ranks timing -- for the headline shape AND for the config 5 entry (small blocks here), with the N-rank schema of the line"""
    n = _ranks_that_fit(n)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    d = _run("--gpus", str(n), "--one-gpu-rehearsal", "--steps", "7", "--warmup", "2", "--streams", "32", "--chunks-per-step", "8", "--no-cpu-baseline", "--no-host-fed",
             "--config5-shape", "48x4", "--config5-steps", "6", env=env)
    assert d["steps"] == 7 and d["value"] > 0
    _assert_n_rank_schema(d, n, 32, 8, 48, 4, "gloo")
    c = d["configs"][f"{n}x48x4"]
    assert c["precision"] == "split16" and c["hipgraph"] is True and c["value"] > 0 and "rehearsal" in c["rccl"]["note"]


@pytest.mark.gpu
@pytest.mark.parametrize("world,total", [(1, 21), (2, 37), (3, 100), (6, 83)])      # (six ranks: as many processes as a one-GPU box lets onto its card at once)
def test_multi_rank_answers_are_the_oracles(world, total, tmp_path):
    """the N > 1 rank path PROVES its answers on one GPU: `--one-gpu-rehearsal --verify-dump` runs the rank code (per-rank engines on a RAGGED
    contiguous partition of the streams, deferred joins, side-stream gathers behind vadc_amd_join, five steps back to back from reset state) and
    writes rank 0's gathered [total_streams, chunks] of every step; streams of EVERY rank -- first, middle and last of its block -- are recomputed
    with the CPU oracle.  A wrong stream -> rank mapping, a gather that reads a buffer before its call has finished, or state that does not carry
    from step to step fails here."""
    import numpy as np
    from oracle import oracle as O
    from vadc_amd import shard, synth
    world = _ranks_that_fit(world)
    dump = str(tmp_path / "gathered.npz")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    Cn = 6
    # (world 1: no gather keeps a step's probabilities and steps i, i + buffers share an output buffer -- the dump takes a copy per step, behind its join)
    d = _run("--gpus", str(world), *(["--one-gpu-rehearsal"] if world > 1 else []), "--steps", "3", "--warmup", "1", "--streams", str(-(-total // world)), "--total-streams", str(total),
             "--chunks-per-step", str(Cn), "--no-cpu-baseline", "--no-host-fed", "--verify-dump", dump, env=env)
    assert d["n_gpus"] == world
    g = np.load(dump)
    probs, NB, K = g["probs"], int(g["buffers"]), int(g["steps"])
    assert probs.shape == (K, total, Cn) and int(g["world"]) == world             # the gather ships the speech probability alone (4 B per chunk)
    blob = open(os.path.join(ROOT, "tests", "golden", "reference_fixtures", "silero_v31_16k.testtensor"), "rb").read()
    orc = O.Oracle(blob)
    for r in range(world):
        lo, hi = shard.stream_block(r, world, total)
        for s in sorted({lo, (lo + hi) // 2, hi - 1}):
            pcm = synth.make_streams(1, NB * Cn, seed0=5000 + s)[0]                     # bench.py --verify-dump: seed = 5000 + global stream id
            seq = np.concatenate([pcm[(k % NB) * Cn * 1536:((k % NB) + 1) * Cn * 1536] for k in range(K)])
            want = orc.forward_stream(seq).reshape(K, Cn, 2)[:, :, 1]
            assert float(np.abs(probs[:, s] - want).max()) <= 1e-4, (r, s)


_RCCL_ONE_RANK = r"""
import os, sys, json
sys.path.insert(0, %(root)r)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
import bench
from vadc_amd import shard
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%(port)d", rank=0, world_size=1, device_id=torch.device("cuda", 0))
S, Cn = 48, 5
probs = torch.arange(S * Cn * 2, dtype=torch.float32, device="cuda:0").view(S, Cn, 2)
side = torch.cuda.Stream()
buf = [torch.empty((S, Cn), dtype=torch.float32, device="cuda:0")]
with torch.cuda.stream(side):                                            # the call ProbabilityGather.gather makes for world > 1, on a side stream as bench.py issues it
    send = probs[:, :, 1].contiguous()
    dist.gather(send, buf, dst=0)
side.synchronize()
ok_gather = bool(torch.equal(buf[0], probs[:, :, 1]))
t = torch.tensor([1.25], dtype=torch.float64, device="cuda:0")          # the max-over-ranks of the timed region
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
g = shard.ProbabilityGather(S, Cn, "cuda:0")
facts = bench.collective_facts(torch, dist, 2, False, g)                 # (world = 2: the branch that asks the process group)
print(json.dumps({"gather": ok_gather, "max": float(t.item()), "facts": facts, "backend": dist.get_backend()}))
sys.stdout.flush()
dist.destroy_process_group()
os._exit(0)
"""


@pytest.mark.gpu
def test_rccl_primitives_with_one_rank():
    """what the N > 1 path asks of RCCL, on the real library with the one rank a one-GPU box can form: dist.gather into a list on the destination rank from a side
    stream (ProcessGroupNCCL's gather), all_reduce(MAX) of a float64 device scalar, barrier, and what collective_facts reads off the process group (backend "nccl",
    world size, ncclGetVersion) -- so that the first 8-GPU run does not start by finding one of them unsupported.  The wire (xGMI, several ranks) stays unexercised."""
    from test_multi_rank_gloo import _free_port
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK % {"root": ROOT, "port": _free_port()}], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["gather"] is True and d["max"] == 1.25 and d["backend"] == "nccl"
    assert d["facts"]["backend"] == "nccl" and d["facts"]["world_size"] == 1 and d["facts"]["bytes_per_chunk"] == 4 and "nccl_version" in d["facts"]


@pytest.mark.gpu
def test_rccl_gather_between_two_gpus():
    """the real collective: two ranks on two GPUs, backend nccl (= RCCL), the path's ProbabilityGather.  Needs two visible GPUs -- a one-GPU box skips."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: RCCL needs two")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    d = _run("--gpus", "2", "--steps", "5", "--warmup", "2", "--streams", "64", "--chunks-per-step", "8", "--no-cpu-baseline", "--no-host-fed",
             "--config5-shape", "128x4", "--config5-steps", "6", env=env)
    assert d["n_gpus"] == 2 and d["value"] > 0
    _assert_n_rank_schema(d, 2, 64, 8, 128, 4, "nccl")
    assert "nccl_version" in d["rccl"]
