#!/usr/bin/env python3
"""bench.py -- throughput of the Silero v3.1 hot path on MI355X (metric of BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--streams S] [--chunks-per-step C]

A "step" = one pass of the hot path (front end -> 4 encoder layers -> LSTM+decoder) over one batch of
synthetic 16 kHz s16le audio: S independent streams x C consecutive 1536-sample chunks per stream, LSTM state
carried on the device from step to step.  Defaults: S = 256 (BASELINE config 2), C = 96 = the window vadc hands its backend per
stream and call (`chunks_count = 96` vadc.c:799, `--batch` default 96 vadc.c:1116).  Inputs are resident in HBM before the timed region.
value = streams x chunks x 0.096 s / wall_s  (audio-seconds per second == concurrent real-time streams),
whole job over all ranks.

Multi-GPU (`--gpus N`): one process per GPU.  Under a launcher (`python -m torch.distributed.run ... bench.py --gpus N`: RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment) this process IS one rank; started plainly (`python bench.py --gpus N`) it spawns the N ranks itself --
before it imports torch or touches HIP -- waits for them and relays rank 0's JSON line.  Streams are sharded across ranks in contiguous
blocks (vadc_amd/shard.py) with no data-path collective; the per-step speech probabilities are gathered to rank 0 with ONE RCCL gather
(north star), inside the timed region.  `--dry-run` runs the same rank skeleton (spawn, rendezvous, sharding, gather, barrier, max-over-ranks
timing) over gloo on the CPU with a stand-in for the engine: what tests/test_bench_spawn.py exercises where there is no GPU.

The JSON line is kept short (the driver reads the tail of stdout); its verbose form -- per-kernel executed / algorithmic FLOP, pipes, notes -- goes to
`--details` (default gpurun_out/bench_details.json).  Besides the contract fields the line carries
  roofline     -- dominant kernel (by CU-time): EXECUTED FLOP per launch / HIP-event duration against the peak of the pipe that executes them
                  (events recorded inside the timed region, on the kernel's own stream, on every 8th step); the algorithmic (dense-basis,
                  SURVEY.md 8(d)) figure rides along
  kernels_ms   -- HIP-event average launch duration of every kernel of the step (k_lstm_l1, the recurrence's second layer, is launched BESIDE the first layer of
                  the same call and follows its progress -- engine option "lstm_trail" --, so its event pair includes its wait for that layer to start)
  cpu_baseline -- the reference C backend (oracle/_ref, kind "reference") or the CPU oracle (kind "port") on this box's host cores: all
                  cores (one process per core, value = aggregate), one core at batch 96 and at batch 1 (BASELINE config 1)
  host_fed     -- the same step through vadc_amd_run_s16 (pageable host buffers in and out: PCIe-inclusive); never `value`
  stage_fracs  -- {kernel: [fraction of its binding pipe's peak, pipe]} for every kernel of the step (STFT against the non-FMA vector
                  peak, the GEMM stages against the matrix peak)
  configs      -- timed in the same run after the headline's timed region, each with value, ms_per_step and the dominant kernel's roofline fraction:
                  "4096x16"  the largest single-GPU configuration (BASELINE config 3: 4096 streams x 16 chunks, SPLIT16 precision, graph replay);
                  "10240x1"  the north star's literal shape (>= 10 k concurrent streams at real time: 10,240 streams x ONE chunk per call), with the
                             latency a chunk sees (`latency_ms`: one isolated call, issue -> probabilities complete; the budget is the 96 ms until the
                             stream's next chunk) and the rate through the asynchronous host-buffer entry point (`host_fed`);
                  "v4_4096x16"  BASELINE config 4: Silero v4 16k at 4096 streams x 16 chunks (GEMM STFT on the fp16 matrix pipe, hipGraph replay);
                  "256x96_fp32_mfma"  the headline workload with every GEMM as literal fp32 MFMA (options encoder = 3, lstm = 3, layer1 = 1): what the
                             split-fp16 x 3 arithmetic of `dtype` buys
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from vadc_amd.staging import pinned, to_device, to_host, to_host_tensor      # noqa: E402  (numpy <-> device through page-locked buffers; imports numpy only: torch stays unimported until a rank needs it)

CHUNK_SECONDS = 1536 / 16000.0
PEAKS = {"valu_nofma": 78.65,     # fp32 vector ALU with every product and every sum rounded separately (the reference's tree): half of the FMA peak
         "fp32": 157.3,           # MI355X_MICROARCH.md: fp32 vector == fp32 matrix peak (FMA)
         "fp16": 2500.0}          # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense
# MAC per chunk by kernel: total, and the part that runs as split-fp16 MFMA (3 fp16 MFMAs per fp32 one)
MAC_V31 = {"k_frontend": (1_651_200, 0), "k_layer1": (181_053, 0), "k_layer2": (112_208, 87_040), "k_layer3": (61_600, 57_344),
           "k_layer4": (236_768, 229_376), "k_lstm": (458_752 + 896, 458_752),
           "k_enc234": (112_208 + 61_600 + 236_768, 0)}      # layers 2-4 in one launch (k_enc_fused): executed work counted in kernel_cost
MAC_V4 = {"k_frontend": (1_585_152, 0), "k_layer1": (232_176, 0), "k_layer2": (19_392, 0), "k_layer3": (10_176, 0), "k_layer4": (25_056, 0),
          "k_lstm": (196_608 + 192, 196_608),
          "k_enc234": (19_392 + 10_176 + 25_056, 0)}       # stages 2-4 in one launch (k_enc_fused_v4): executed work counted in kernel_cost
PATH_FLOP_PER_CHUNK = {"v31": 2 * 2_702_477, "v4": 2 * 2_068_752}      # whole path
FRONTEND_KERNELS = {0: "k_frontend_sym", 1: "k_frontend_fl", 2: "k_frontend_gemm2", 3: "k_frontend (v4 tree)"}


def metric_label(model, world):
    """N = 1: BASELINE.json's metric.  N > 1: `value` is the whole job (the driver's contract) -- the label says so, and value_per_gpu carries the metric's per-GPU figure"""
    m = "Silero v3.1 16k" if model == "v31" else "Silero v4 16k (BASELINE config 4; not the headline metric)"
    if world == 1:
        return f"audio-seconds/sec (= real-time streams) per GPU, {m}"
    return f"audio-seconds/sec (= real-time streams), WHOLE JOB over {world} GPUs (value_per_gpu = value / {world}), {m}"


def frontend_name(eng, fe_kernel):
    """the front-end kernel that ran, by name: the engine reports 2 for either GEMM form; this bench feeds s16, which is k_frontend_gemm2's (f32 input: k_frontend_gemm)"""
    return FRONTEND_KERNELS.get(fe_kernel)


def dtype_label(mode, fe_kernel):
    """what the arithmetic runs in -- by the front-end kernel that ran, not by the precision mode alone (Silero v4's STFT is the GEMM form in every mode)"""
    gemms = "split-f16x3 GEMMs (f32 accumulate)"
    if fe_kernel == 2:
        return f"split-f16x3 GEMM STFT (f32 accumulate) + {gemms}"
    return f"f32 STFT + {gemms}" + (", f32-MFMA fallbacks refused" if mode == 1 else "")


def workload_label(model, mode, fe_kernel):
    if model == "v4":
        stft = ("STFT as a folded real-input GEMM on the fp16 matrix pipe (split-fp16 x 3, fp32 accumulation: the v4 parity target is a framework convolution in any fp32 order)"
                if fe_kernel == 2 else "STFT by the fp32 tree kernel")
        return f"BASELINE config 4: {stft} + split-fp16 x 3 GEMMs with fp32 accumulation (22-bit operands)"
    return {0: "parity mode: exact fp32 STFT tree + split-fp16 x 3 GEMMs with fp32 accumulation (22-bit operands; BASELINE config 2 says fp32: the literal fp32-MFMA engine is configs[256x96_fp32_mfma])",
            1: "SPLIT16 precision mode (BASELINE config 3)", 2: "FAST_STFT throughput mode (GEMM STFT: outside the 1e-4 bar)"}[mode]


def kernel_cost(model, name, fe_kernel, layer_major=False, layer1_regs=False):
    """-> (algorithmic FLOP per chunk, {pipe: executed FLOP per chunk}).  Executed = what the kernel issues: the symmetric front end
    evaluates 33 of the 129 bins' trees, split-fp16 GEMMs issue three fp16 MFMAs per fp32 product, the folded GEMM front end half the taps."""
    if name in ("k_lstm", "k_lstm_l1") and layer_major:      # k_lstm_layer: "k_lstm" is layer 0 alone, "k_lstm_l1" layer 1 + decoder
        mac, mac16 = (MAC_V4 if model == "v4" else MAC_V31)["k_lstm"]
        mac, mac16 = (mac16 // 2, mac16 // 2) if name == "k_lstm" else (mac - mac16 // 2, mac16 // 2)
    else:
        mac, mac16 = (MAC_V4 if model == "v4" else MAC_V31)[name]
    alg = 2 * mac
    if name == "k_frontend":
        frames = 24 if model == "v4" else 25
        if fe_kernel == 0:      # per position: 33 base bins x (2 x (256 mul + 248 add) + 56 lane-tree adds) + 129 x (re^2, im^2, +)
            return alg, {"valu_nofma": frames * (33 * (2 * 504 + 56) + 129 * 3)}
        if fe_kernel == 2:      # folded real-input DFT: 256 rows x K = 128, three split-fp16 MFMAs per k-block
            return alg, {"fp16": 3 * 2 * 256 * 128 * frames}
        return alg, {"valu_nofma": frames * (129 * 2 * 511 + 129 * 3)}
    if name == "k_layer1" and model == "v4" and layer1_regs:
        # k_layer1_regs_v4: 96 v_mfma_f32_16x16x32_f16 for K = 516 (16 k blocks x 3 split terms x 2 tiles) + 4 for bin 128 + 4 for the strided conv
        return alg, {"fp16": (96 + 4 + 4) * 16384}
    if name == "k_layer1" and layer1_regs:
        # k_layer1_regs issues v_mfma_f32_16x16x32_f16 only, per chunk (two 16-column tiles for its 25 steps, idle columns and zero k slots included):
        # 48 for the 258 -> 16 conv block (8 k blocks x 3 split terms x 2 tiles), 4 for the Nyquist channel, 60 for the D = 16 transformer block and the
        # strided conv (two instructions per K = 16 product); depthwise conv, operand splits, softmax and LayerNorm are vector work
        return alg, {"fp16": (48 + 4 + 60) * 16384}
    if name == "k_enc234" and model == "v4":
        # k_enc_fused_v4 issues v_mfma_f32_16x16x32_f16 only (16,384 FLOP each, idle columns included), per chunk: stage 2 six for the conv block + six for the strided
        # conv (one 16-column tile per chunk), stages 3 and 4 on one tile per two chunks: (6 + 6) / 2 and (12 + 12 + 24) / 2
        return alg, {"fp16": (12 + 6 + 24) * 16384}
    if name == "k_enc234":
        # k_enc_fused issues v_mfma_f32_16x16x32_f16 only (16,384 FLOP each, zero-padded k and idle columns included): per chunk 60 in layer 2 (one
        # 16-column tile per chunk), 30 in layer 3 and 105 in layer 4 (one tile per two chunks); depthwise conv, softmax and LayerNorm are vector work
        return alg, {"fp16": (60 + 30 + 105) * 16384}
    exe = {}
    if mac16:
        exe["fp16"] = 3 * 2 * mac16
    if mac - mac16:
        exe["fp32"] = 2 * (mac - mac16)
    return alg, exe


# ------------------------------------------------------------------------------------------------- CPU baseline
def _cpu_runner(model, weights_path):
    from oracle import oracle as O                      # the checker, timed here as the reported baseline -- never the product path
    from vadc_amd import synth
    base = 2048                                         # chunks of one synthetic speech stream (3.3 min of audio), repeated
    pcm = synth.speech_like(base * 1536, seed=9)
    try:
        if model == "v4":                               # the reference has no C implementation of v4: only the restatement exists
            raise FileNotFoundError
        ref = O.Reference(weights_path)
        x = pcm.astype(np.float32) / np.float32(32768)
        return "reference", base, (lambda n, batch: ref.run(x[: n * 1536], batch=batch))
    except (FileNotFoundError, OSError, ValueError):
        blob = open(weights_path, "rb").read()
        orc = O.OracleV4(blob) if model == "v4" else O.Oracle(blob)
        return "port", base, (lambda n, batch: orc.forward_stream(pcm[: n * 1536]))


def cpu_worker(model, weights_path, seconds, batch):
    """one core: chunks per second over ~`seconds` of work; prints one JSON line (run as a child process, one per core)"""
    kind, base, run = _cpu_runner(model, weights_path)
    run(8, batch)
    n_unit = 256 if batch > 1 else 128
    t0 = time.perf_counter(); run(n_unit, batch); dt = time.perf_counter() - t0
    reps = max(1, int(round(seconds / max(dt, 1e-6))))
    t0 = time.perf_counter()
    done = 0
    for r in range(reps):
        run(n_unit, batch); done += n_unit
    dt = time.perf_counter() - t0
    print(json.dumps({"kind": kind, "chunks": done, "seconds": dt, "chunks_per_s": done / dt}), flush=True)


def usable_cores():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup's CPU quota (a GPU box hands a 1-GPU job 16 of its 256 hardware
    threads through cpu.max; more busy processes than that only take turns) and by 64."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_baseline(model, weights_path, seconds=10.0):
    """Reported baseline, not the target.  Children are plain `python bench.py --cpu-worker` processes started BEFORE this process touches the GPU."""
    def spawn(batch, secs):
        return subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", "--model", model, "--cpu-seconds", str(secs), "--cpu-batch", str(batch)],
                                stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    def collect(procs):
        out = []
        for p in procs:
            o, _ = p.communicate(timeout=600)
            out.append(json.loads(o.strip().splitlines()[-1]))
        return out
    cores = usable_cores()
    one1 = collect([spawn(1, 3.0)])[0]                  # BASELINE config 1: batch = 1, one core, the machine otherwise idle
    one96 = collect([spawn(96, 3.0)])[0]
    allc = collect([spawn(96, seconds) for _ in range(cores)])
    agg = sum(r["chunks_per_s"] for r in allc)
    return {"value": round(agg * CHUNK_SECONDS, 1), "unit": "audio-seconds/sec", "cores": cores, "kind": allc[0]["kind"],
            "per_core": round(agg * CHUNK_SECONDS / cores, 2),
            "single_core_batch96": round(one96["chunks_per_s"] * CHUNK_SECONDS, 2),
            "single_core_batch1": round(one1["chunks_per_s"] * CHUNK_SECONDS, 2),
            "sample": f"one process per usable core ({cores}: affinity mask capped by the cgroup CPU quota), each {allc[0]['chunks']} consecutive chunks of one synthetic speech stream at batch 96 "
                      f"({sum(r['seconds'] for r in allc):.0f} core-seconds); single-core points: {one96['chunks']} chunks at batch 96, {one1['chunks']} at batch 1 (BASELINE config 1)"}


def pin_rank_to_its_cpus(local_rank, local_world):
    """N ranks of one node share the CPUs this job may use (a GPU box hands a job a slice of its hardware threads): each rank keeps to its own
    share of the affinity mask, so that eight issuing threads do not migrate over -- and evict each other from -- the same few cores.
    Returns the CPUs the rank ended up with (for the line / the tests); leaves the mask alone when there are fewer CPUs than ranks."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
        per = len(cpus) // max(local_world, 1)
        if local_world > 1 and per >= 1:
            mine = cpus[local_rank * per:(local_rank + 1) * per]
            os.sched_setaffinity(0, mine)
            return mine
        return cpus
    except (AttributeError, OSError):
        return []


# ------------------------------------------------------------------------------------------------- rank spawning
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (this parent has not imported torch and never touches HIP),
    relay rank 0's stdout, exit with the worst return code."""
    env0 = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(n), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # supervise every rank: one that dies at start-up (bad device, out of memory) would leave the others in the rendezvous or in the gather until the
    # collective's timeout -- stop them and report its return code instead
    import threading
    out_box = []
    reader = threading.Thread(target=lambda: out_box.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + 3600
    failed = None
    while any(p.poll() is None for p in procs):
        bad = [p for p in procs if p.poll() not in (None, 0)]
        if bad or time.time() > deadline:
            failed = bad[0].returncode if bad else 124
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    reader.join(timeout=10)
    sys.stdout.write(out_box[0] if out_box else "")
    sys.stdout.flush()
    return abs(failed) if failed else max(abs(p.returncode or 0) for p in procs)


# ------------------------------------------------------------------------------------------------- dry run (CPU, gloo)
def dry_run(args, world, rank):
    """The rank skeleton without a GPU: gloo rendezvous, stream sharding, the per-step gather through the SAME helper the GPU path uses,
    barrier + max-over-ranks timing, rank 0 prints the line with the N-rank schema of the real run (job total under a truthful metric label, value_per_gpu,
    total_streams, what the process group reports, and the BASELINE config 5 entry configs["<N>xSxC"] timed under the same ranks).  The engine is replaced
    by values that encode (global stream, chunk, column, step)."""
    import torch
    import torch.distributed as dist
    from vadc_amd import shard
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)

    def region(S, Cn, steps):
        total = S * world
        lo, hi = shard.stream_block(rank, world, total)
        g = shard.ProbabilityGather(total, Cn, "cpu")
        s = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1)
        c = torch.arange(Cn, dtype=torch.float32).view(1, -1, 1)
        k = torch.arange(2, dtype=torch.float32).view(1, 1, 2)
        ok = True
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            g.gather(s * 1000 + c * 2 + k + i * 0.25)              # the stand-in's [S, Cn, 2] "probabilities": the gather ships element 1 alone
            if rank == 0 and i == steps - 1:
                sa = torch.arange(total, dtype=torch.float32).view(-1, 1)
                ok = bool(torch.equal(g.result(), sa * 1000 + c[:, :, 0] * 2 + 1 + i * 0.25))
        if world > 1:
            dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        return ok, total, dt, round(total * Cn * steps * CHUNK_SECONDS / dt, 1), collective_facts(torch, dist, world, False, g)

    S, Cn = args.streams, args.chunks_per_step
    ok, total, dt, value, facts = region(S, Cn, args.steps)
    out = {"metric": "dry run (no GPU, stand-in engine): " + metric_label(args.model, world), "value": value, "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 4), "higher_is_better": True, "scaling": "weak",
           "dry_run": True, "gather_verified": ok, "total_streams": total, "value_per_gpu": round(value / world, 1), "rccl": facts, "rank_cpus": args.rank_cpus_n,
           "config": {"workload": f"stand-in engine, {S} streams/rank x {Cn} chunks/step", "streams_per_gpu": S, "chunks_per_step": Cn,
                      "parallelism": f"streams sharded over {world} rank(s), gloo gather"}}
    if world > 1:
        one = one_gpu_figures(args.one_gpu_json)
        out["efficiency_vs_1gpu"] = vs_one_gpu(value, world, one, f"{S}x{Cn}")
        c5_S, c5_C = (int(v) for v in args.config5_shape.lower().split("x"))
        ok5, total5, dt5, value5, facts5 = region(c5_S, c5_C, args.config5_steps)
        ok = ok and ok5
        out["gather_verified"] = ok
        out["configs"] = {f"{world}x{c5_S}x{c5_C}": {"value": value5, "per_gpu": round(value5 / world, 1), "ms_per_step": round(dt5 / args.config5_steps * 1e3, 4),
                                                     "steps": args.config5_steps, "n_gpus": world, "total_streams": total5, "streams_per_gpu": c5_S, "chunks_per_step": c5_C,
                                                     "rccl": facts5, "efficiency_vs_1gpu": vs_one_gpu(value5, world, one, f"{c5_S}x{c5_C}"),
                                                     "one_gpu_figure": one.get(f"{c5_S}x{c5_C}")}}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


# ------------------------------------------------------------------------------------------------- one rank's step discipline
_CALLER_STREAMS = []


def caller_streams(torch, n):
    """the streams this process issues its steps from: made once and shared by the headline and every side configuration.  HIP deals a process's plain streams onto four
    hardware queues in turn, and an engine driven from a NEW stream per configuration ran 7 - 19 % slower whenever that stream's queue was an unlucky one
    (tools/engine_order_probe.py): a side configuration must not be measured on its position in this file"""
    while len(_CALLER_STREAMS) < n:
        _CALLER_STREAMS.append(torch.cuda.Stream())
    return _CALLER_STREAMS[:n]


class StepLoop:
    """How one rank issues its steps -- shared by the headline's timed region and by every configuration timed beside it, at any number of ranks.
    NB input / output buffers used in turn.  Deferred joins (default): every step is issued from streams[0], which a call does not block (engine option
    defer_join); with several ranks a side stream joins the call (vadc_amd_join: a device-side wait) and carries the ONE collective, the gather of the
    step's speech probabilities (4 B per chunk) to rank 0, so that step k's gather runs beside step k + 1's kernels.  strict: NB caller streams in turn,
    each strictly ordered."""

    def __init__(self, torch, eng, S, Cn, d_in, d_probs, gather, world=1, rehearsal=False, defer_join=True):
        self.torch, self.eng, self.S, self.Cn, self.d_in, self.d_probs = torch, eng, S, Cn, d_in, d_probs
        self.NB = len(d_in)
        self.gather, self.world, self.rehearsal, self.defer_join = gather, world, rehearsal, defer_join
        self.streams = caller_streams(torch, max(self.NB, 2 if world > 1 else 1))
        self.gathered = [None] * self.NB          # per step buffer: the event behind the gather that last read it (several ranks only)
        if defer_join:
            eng.set_option("defer_join", 1)

    def step(self, i, gather=None):
        torch, eng, S, Cn = self.torch, self.eng, self.S, self.Cn
        gather = gather if gather is not None else self.gather
        b = i % self.NB
        if self.defer_join:
            if self.world > 1 and self.gathered[b] is not None:
                self.streams[0].wait_event(self.gathered[b])          # the gather that read d_probs[b] NB steps ago comes before this call rewrites it
            eng.run_device(self.d_in[b].data_ptr(), np.int16, S, Cn, self.d_probs[b].data_ptr(), self.streams[0].cuda_stream)
            if self.world > 1:
                side = self.streams[1 + b % (len(self.streams) - 1)]
                eng.join(side.cuda_stream)
                with torch.cuda.stream(side):
                    gather.gather(to_host_tensor(self.d_probs[b]) if self.rehearsal else self.d_probs[b])
                    self.gathered[b] = torch.cuda.Event()
                    self.gathered[b].record(side)
            return
        st = self.streams[b]
        with torch.cuda.stream(st):
            eng.run_device(self.d_in[b].data_ptr(), np.int16, S, Cn, self.d_probs[b].data_ptr(), st.cuda_stream)
            if self.world > 1:
                gather.gather(to_host_tensor(self.d_probs[b]) if self.rehearsal else self.d_probs[b])       # the only collective: final probability gather (RCCL)


def collective_facts(torch, dist, world, rehearsal, gather):
    """what the process group itself reports about the job's one collective (not what the command line asked for)"""
    if world == 1 or not dist.is_initialized():
        return {"backend": None, "world_size": 1, "collective": None}
    out = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "collective": "gather -> rank 0, one per step, on a side stream behind vadc_amd_join",
           "bytes_per_chunk": gather.bytes_per_chunk, "bytes_per_rank_and_step": gather.bytes_per_chunk * gather.mx * gather.send.shape[1]}
    if out["backend"] == "nccl":
        try:
            out["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())      # ncclGetVersion of the RCCL this process loaded
        except Exception as ex:                                                               # noqa: BLE001 -- a fact we report, never a reason to fail the run
            out["nccl_version"] = f"unavailable ({type(ex).__name__})"
    if rehearsal:
        out["note"] = "one-GPU rehearsal: gloo through the host, all ranks on GPU 0 -- exercises the rank code, not RCCL / xGMI"
    return out


# ------------------------------------------------------------------------------------------------- configurations timed beside the headline
_SIDE_INPUT = {}


def _side_input(S, Cn, NB, rank):
    """the synthetic input of a side configuration, [S, NB * Cn * 1536] s16: 16 distinct signals tiled over the streams.  The last one made is kept (a caller that times
    several engines at one shape -- tests/test_gpu_partition_rules.py -- spent most of its time making it again)"""
    from vadc_amd import synth
    key = (S, Cn, NB, rank)
    if key not in _SIDE_INPUT:
        _SIDE_INPUT.clear()
        base = synth.make_streams(16, NB * Cn, seed0=777 + 100 * rank)
        _SIDE_INPUT[key] = np.ascontiguousarray(np.tile(base, (S // 16 + 1, 1))[:S])
    return _SIDE_INPUT[key]


def side_config(torch, blob, dev, local_rank, model, S, Cn, precision, steps=20, warmup=5, opts=None, latency=False, host_fed=False,
                world=1, rank=0, rehearsal=False):
    """Another configuration timed in the same run, after the headline's timed region: same step discipline (StepLoop: deferred joins, one issuing stream,
    three input buffers, graph replay), device time from torch's synchronize on both sides.  Reported beside `value`, never instead of it.
    world > 1: EVERY rank runs the configuration on its own S streams under the same process group -- the gather inside the timed region, barriers on both
    sides, max over ranks -- and the entry carries the job's total, the per-GPU figure and what the process group reports.
    latency: also the time of ONE isolated call from its issue to its probabilities being complete (median of 20); host_fed: also the rate through
    vadc_amd_run_s16_async (page-locked host buffers in and out)."""
    import torch.distributed as dist
    from vadc_amd import shard, synth
    from vadc_amd.engine import Engine
    eng = Engine(blob, max_streams=S, max_chunks_per_call=Cn, device=local_rank, precision=precision)
    for k_, v_ in (opts or {}).items():
        eng.set_option(k_, v_)
    NB = 3
    pcm = _side_input(S, Cn, NB, rank)
    d_in = [to_device(np.ascontiguousarray(pcm[:, i * Cn * 1536:(i + 1) * Cn * 1536]), dev) for i in range(NB)]
    d_out = [torch.empty((S, Cn, 2), dtype=torch.float32, device=dev) for _ in range(NB)]
    gather = shard.ProbabilityGather(S * world, Cn, "cpu" if rehearsal else dev)
    loop = StepLoop(torch, eng, S, Cn, d_in, d_out, gather, world, rehearsal, True)
    eng.set_option("groups", 1)
    step = loop.step
    st = loop.streams[0]
    for i in range(2 * NB):
        step(i)
    torch.cuda.synchronize()
    eng.set_option("graph", 1)
    for i in range(2 * NB + warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    eng.reset_kernel_times()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.set_profiling(i % 8 == min(3, steps - 1))
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    eng.set_profiling(False)
    kt = {k: ms / n for k, (n, ms) in eng.kernel_times().items() if n}
    fe_kernel = eng.get_option("frontend_kernel")
    dom = max(kt, key=kt.get)
    _, exe = kernel_cost(model, dom, fe_kernel, "k_lstm_l1" in kt, eng.get_option("layer1_kernel") == 0)
    pipe = max(exe, key=lambda p_: exe[p_] / PEAKS[p_])
    out = {"value": round(S * world * Cn * steps * CHUNK_SECONDS / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps,
           "precision": {0: "fp32", 1: "split16", 2: "fast_stft"}[precision], "hipgraph": True,
           "roofline_kernel": dom, "roofline_frac": round(exe[pipe] * S * Cn / (kt[dom] * 1e-3) / 1e12 / PEAKS[pipe], 4),
           "frontend_kernel": frontend_name(eng, fe_kernel),
           "kernels_ms": {k: round(v, 4) for k, v in kt.items()},
           "resolved": {"lstm": eng.get_option("lstm_kernel"), "lstm_cus": eng.get_option("lstm_cus"), "shared": eng.get_option("lstm_shared")}}      # what the engine's rules (or the forced options) came to
    if world > 1:
        out.update({"n_gpus": world, "total_streams": S * world, "streams_per_gpu": S, "chunks_per_step": Cn, "per_gpu": round(out["value"] / world, 1),
                    "rccl": collective_facts(torch, dist, world, rehearsal, gather)})
    if opts:
        out["options"] = dict(opts)
    if latency:
        lat = []
        for i in range(20):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step(i)
            eng.join(st.cuda_stream)
            st.synchronize()
            lat.append(time.perf_counter() - t1)
        out["latency_ms"] = round(float(np.median(lat)) * 1e3, 3)
        out["latency_budget_ms"] = 96.0
    if host_fed:
        host = [pinned(pcm[:, i * Cn * 1536:(i + 1) * Cn * 1536]).numpy() for i in range(NB)]      # page-locked by torch's host allocator (see the headline's host_fed leg)
        outs = [pinned(np.empty((S, Cn, 2), np.float32)).numpy() for _ in range(NB)]
        for i in range(18):
            eng.run_async(host[i % NB], outs[i % NB])
        eng.wait_async()
        n_host = 48
        t1 = time.perf_counter()
        for i in range(n_host):
            eng.run_async(host[i % NB], outs[i % NB])
        eng.wait_async()
        dta = time.perf_counter() - t1
        out["host_fed"] = {"value": round(S * Cn * n_host * CHUNK_SECONDS / dta, 1), "ms_per_step": round(dta / n_host * 1e3, 3), "pcie_gb_per_s": round(S * Cn * 3072 * n_host / dta / 1e9, 1)}
    torch.cuda.synchronize()
    eng.close()
    return out


def side_config_v5(torch, blob, dev, local_rank, S, Cn, steps=60, warmup=10):
    """Silero v5 SHAPES (process_chunks_v5 vadc.c:105-162, silero_vad.py:290-434) on seeded weights -- the reference ships none, so this is no
    BASELINE config: S streams x Cn 512-sample windows (32 ms each) per call, s16 resident in HBM, deferred joins (the encoder of call k + 1 beside the recurrence
    of call k).  value = audio-seconds per second; the encoder's executed split-fp16 MFMA FLOP against the fp16 matrix peak."""
    from vadc_amd import synth
    from vadc_amd.engine import Engine
    W = 512
    eng = Engine(blob, max_streams=S, max_chunks_per_call=Cn, device=local_rank)
    NB = 3
    base = synth.make_streams(16, -(-NB * Cn * W // 1536), seed0=11)[:, :NB * Cn * W]
    pcm = np.ascontiguousarray(np.tile(base, (-(-S // 16), 1))[:S])
    d_in = [to_device(np.ascontiguousarray(pcm[:, i * Cn * W:(i + 1) * Cn * W]), dev) for i in range(NB)]
    d_out = [torch.empty((S, Cn, 2), dtype=torch.float32, device=dev) for _ in range(NB)]
    st = caller_streams(torch, 1)[0]
    eng.set_option("defer_join", 1)

    def step(i):
        eng.run_device(d_in[i % NB].data_ptr(), np.int16, S, Cn, d_out[i % NB].data_ptr(), st.cuda_stream)
    for i in range(2 * NB + warmup):
        step(i)
    torch.cuda.synchronize()
    eng.reset_kernel_times()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.set_profiling(i % 8 == min(3, steps - 1))
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.set_profiling(False)
    kt = {k: ms / n for k, (n, ms) in eng.kernel_times().items() if n}
    # k_v5_encoder_h3 per tile of 16 windows, v_mfma_f32_16x16x32_f16 (16,384 FLOP each, three per fp32 product): folded STFT 768, conv 0 1,248, conv 1 288,
    # conv 2 48, conv 3 48, W_ih 384 = 2,784 -> 174 per window
    exe = 174 * 16384
    out = {"value": round(S * Cn * steps * (W / 16000.0) / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps, "window": W,
           "encoder_kernel": {1: "k_v5_encoder (fp32 MFMA)", 2: "k_v5_encoder_h3 (split-fp16 MFMA)"}.get(eng.get_option("frontend_kernel")),
           "kernels_ms": {k: round(v, 4) for k, v in kt.items()},
           "note": "Silero v5 shapes on SEEDED weights (the reference ships none): not a BASELINE config"}
    if "k_frontend" in kt and eng.get_option("frontend_kernel") == 2:
        out["roofline_kernel"] = "k_frontend"
        out["roofline_frac"] = round(exe * S * Cn / (kt["k_frontend"] * 1e-3) / 1e12 / PEAKS["fp16"], 4)
    eng.close()
    return out


def one_gpu_figures(path):
    """the 1-GPU figures an N-rank run compares itself with: written by the N = 1 run of this bench (same box, same tree), or handed over with --one-gpu-json"""
    try:
        d = json.load(open(path))
        return d if isinstance(d, dict) else {}
    except (OSError, ValueError):
        return {}


def vs_one_gpu(entry_value, world, one, key):
    """efficiency = (job total / N) / the 1-GPU figure of the same shape -- informative only; the driver computes its own from the per-N lines"""
    ref = one.get(key)
    return round(entry_value / world / ref, 4) if ref else None


# ------------------------------------------------------------------------------------------------- one rank
def run_rank(args, world, rank, local_rank):
    weights_path = os.path.join(ROOT, "tests", "golden", "reference_fixtures", "silero_v31_16k.testtensor")
    if args.model == "v4":
        weights_path = os.path.join(ROOT, "tests", "golden", "silero_v4_16k.testtensor")
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.model, weights_path)    # before this process initialises the GPU: its children are plain CPU processes

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path (--dry-run rehearses the rank skeleton over gloo)")
    rehearsal = args.one_gpu_rehearsal        # N ranks on ONE GPU with gloo (a single-GPU box cannot form an RCCL group): exercises this rank code, not RCCL
    if rehearsal:
        local_rank = 0
    # one process per GPU, bound by LOCAL_RANK before this process makes any HIP call (device_count() does not initialise the runtime on this image)
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} is bound to device {local_rank} (LOCAL_RANK), but {torch.cuda.device_count()} device(s) are visible")
    torch.cuda.set_device(local_rank)
    assert torch.cuda.current_device() == local_rank
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from vadc_amd import shard, synth
    from vadc_amd.engine import Engine

    dev = f"cuda:{local_rank}"
    blob = open(weights_path, "rb").read()
    S, Cn = args.streams, args.chunks_per_step
    total_streams = args.total_streams if args.total_streams > 0 else S * world
    if args.total_streams > 0:                       # a ragged total (--verify-dump runs): this rank's block of the contiguous partition
        lo_, hi_ = shard.stream_block(rank, world, total_streams)
        S = hi_ - lo_
    eng = Engine(blob, max_streams=S, max_chunks_per_call=Cn, device=local_rank, precision={"fp32": 0, "split16": 1, "fast_stft": 2}[args.precision])
    assert eng.caps()["device"] == local_rank, (eng.caps()["device"], local_rank)      # the engine's kernels run on this rank's own GPU
    mode = eng.caps()["precision"]
    eng.set_option("groups", args.groups)
    for kv in args.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))

    # synthetic input: 16 distinct speech-like streams per rank tiled over S, NB step buffers used in turn
    NB = args.caller_streams
    if args.verify_dump:                              # every GLOBAL stream its own signal (seed = 5000 + global id): what tests/test_bench_spawn.py re-creates
        lo_, _ = shard.stream_block(rank, world, total_streams)
        pcm = np.concatenate([synth.make_streams(1, NB * Cn, seed0=5000 + lo_ + i) for i in range(S)])
    else:
        base = synth.make_streams(min(S, 16), NB * Cn, seed0=1234 + 100 * rank)
        pcm = np.ascontiguousarray(np.tile(base, ((S + base.shape[0] - 1) // base.shape[0], 1))[:S])
    d_in = [to_device(np.ascontiguousarray(pcm[:, i * Cn * 1536:(i + 1) * Cn * 1536]), dev) for i in range(NB)]
    d_probs = [torch.empty((S, Cn, 2), dtype=torch.float32, device=dev) for _ in range(NB)]
    gather = shard.ProbabilityGather(total_streams, Cn, "cpu" if rehearsal else dev)      # weak scaling: S streams per GPU, contiguous blocks; 4 B per chunk
    assert gather.hi - gather.lo == S
    # The engine's internal in-order streams overlap the stages of consecutive steps: front end + encoder of step k+2 beside LSTM layer 0 of step
    # k+1 beside layer 1 of step k (layer-major LSTM, <= 512 streams), or front end + encoder beside the whole LSTM.  Default: all steps issued
    # from ONE stream, which a call does not block (defer_join); --strict-join: NB caller streams in turn, each strictly ordered.
    loop = StepLoop(torch, eng, S, Cn, d_in, d_probs, gather, world, rehearsal, args.defer_join)
    step, streams, gathered = loop.step, loop.streams, loop.gathered

    for i in range(2 * NB):                # setup, not warm-up: the first calls create the engine's internal streams / CU masks and touch every buffer once
        step(i)
    torch.cuda.synchronize()
    eng.set_profiling(False)
    if args.graph:                        # setup as well: capture and instantiate every (input buffer, hand-off buffer) pairing -- host work, the GPU mostly idles
        eng.set_option("graph", 1)
        for i in range(2 * NB):
            step(i)
        torch.cuda.synchronize()
    for i in range(args.warmup):          # the W warm-up steps run what the timed steps run (graph replay), right before them
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    eng.reset_kernel_times()
    separate_pass = args.no_kernel_timing
    # Per-kernel HIP events (two hipEventRecord per launch, on the launch's stream) cost ~2.5 % of the step when every launch
    # of the timed region carries them (each boundary between two timed kernels costs ~12 us); they are recorded on every 8th step of the
    # timed region instead (still "live", still on the kernel's own stream), which keeps `value` within ~0.6 % of an event-free run.
    prof_every = 8
    n_prof = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        if not separate_pass:
            on = (i % prof_every) == min(3, args.steps - 1)   # 20 steps: steps 3, 11, 19 (not the pipeline-filling first one); a run of < 4 steps times its last one
            eng.set_profiling(on)
            n_prof += int(on)
        step(i)
    issued = time.perf_counter() - t0         # host time to ISSUE the K steps (the device is still running them)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    eng.set_profiling(False)
    if separate_pass:                     # kernel durations from an eager pass outside the timed region
        eng.set_option("graph", 0)
        eng.set_profiling(True)
        n_prof = 4
        for i in range(n_prof):
            step(i)
        torch.cuda.synchronize()
        eng.set_profiling(False)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if args.verify_dump:
        # The N-rank path proves its answers: from reset state, K steps issued back to back exactly as in the timed region (deferred joins, side-stream
        # gathers behind vadc_amd_join, no host synchronisation in between), each step gathered into a buffer set of its own; rank 0 writes the K gathered
        # [total_streams, chunks] speech probabilities.  tests/test_bench_spawn.py recomputes sampled streams of EVERY rank with the CPU oracle.
        K = 2 * NB - 1
        eng.synchronize(); torch.cuda.synchronize()
        eng.set_option("graph", 1 if args.graph else 0)
        eng.reset_streams()
        for b_ in range(NB):
            gathered[b_] = None
        gv = [shard.ProbabilityGather(total_streams, Cn, "cpu" if rehearsal else dev) for _ in range(K)]
        if world > 1:
            dist.barrier()
        single = []                            # one GPU: no gather keeps a step's probabilities, and steps i, i + NB share d_probs[i % NB] -- a copy per step, behind its join
        for i in range(K):
            step(i, gv[i])
            if world == 1:
                if args.defer_join:
                    eng.join(streams[0].cuda_stream)
                with torch.cuda.stream(streams[0] if args.defer_join else streams[i % NB]):
                    single.append(d_probs[i % NB][:, :, 1].clone())
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if rank == 0:
            res = np.stack([to_host(g.result() if world > 1 else single[i]) for i, g in enumerate(gv)])
            np.savez(args.verify_dump, probs=res, total_streams=total_streams, world=world, chunks=Cn, buffers=NB, steps=K)

    out = None
    if rank == 0:
        chunks_per_step = total_streams * Cn
        value = chunks_per_step * args.steps * CHUNK_SECONDS / elapsed
        kt = eng.kernel_times()
        fe_kernel = eng.get_option("frontend_kernel")
        # The LSTM chain runs concurrently on its own small CU partition; weigh every kernel's
        # time by the share of the chip it occupies so that "dominant" means dominant in CU-time, not in wall time
        # of a kernel that leaves 240 CUs to the others.
        n_cus = torch.cuda.get_device_properties(local_rank).multi_processor_count
        cu_share = {k: 1.0 for k in kt}
        lstm_cus = eng.get_option("lstm_cus")                     # CUs the engine reserved for the LSTM chain (0: whole chip)
        layer_major = kt.get("k_lstm_l1", (0, 0.0))[0] > 0        # two layer launches, each on half of the LSTM's CU partition
        cu_share["k_lstm"] = (lstm_cus / n_cus / (2 if layer_major else 1)) if lstm_cus > 0 else 1.0
        cu_share["k_lstm_l1"] = cu_share["k_lstm"]
        per_kernel = {}
        for k, (n_l, ms) in kt.items():
            if not n_l:
                continue
            alg, exe = kernel_cost(args.model, k, fe_kernel, layer_major, eng.get_option("layer1_kernel") == 0)
            per_launch = S * Cn * n_prof / n_l                    # chunks one launch processes (a step may be split into chunk groups)
            sec = ms / n_l / 1e3
            # binding pipe = the one whose executed FLOP take longest at its peak (the pipes can overlap: this is the LOWER bound on the kernel's time)
            pipe = max(exe, key=lambda p: exe[p] / PEAKS[p])
            per_kernel[k] = {"ms_per_launch": round(ms / n_l, 4), "cu_share": round(cu_share[k], 4), "chunks_per_launch": int(per_launch),
                             "algorithmic_flop_per_chunk": alg, "algorithmic_tflops": round(alg * per_launch / sec / 1e12, 3),
                             "executed_flop_per_chunk": exe, "pipe": pipe,
                             "executed_tflops": round(exe[pipe] * per_launch / sec / 1e12, 3),
                             "frac_of_pipe_peak": round(exe[pipe] * per_launch / sec / 1e12 / PEAKS[pipe], 4)}
        dom = max(per_kernel, key=lambda k: kt[k][1] * cu_share[k])
        d = per_kernel[dom]
        traffic = None
        traffic_source = None
        try:   # HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, KB -> B): only a pass collected
            # on exactly this workload AND this front-end kernel counts (tools/rocprof_reduce.py writes both into the file), else null
            default_workload = args.model == "v31" and mode == 0 and S == 256 and Cn == 96
            name = "latest_pmc_traffic.json" if default_workload else f"latest_pmc_traffic_{args.model}_{args.precision}_{S}x{Cn}.json"
            prof = json.load(open(os.path.join(ROOT, "profiles", name)))
            if (prof.get("streams") == S and prof.get("chunks_per_step") == Cn and prof.get("model", "v31") == args.model
                    and prof.get("precision", "fp32") == args.precision and prof.get("frontend_kernel") == frontend_name(eng, fe_kernel)
                    and dom in prof.get("kernels", {})):
                traffic = prof["kernels"][dom]["hbm_bytes_per_launch"]
                traffic_source = f"profiles/{name} (rocprofv3 --pmc pass of this workload, committed; not measured in this run)"
        except (OSError, ValueError):
            pass
        out = {
            "metric": metric_label(args.model, world),
            "value": round(value, 1), "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": dtype_label(mode, fe_kernel), "data": "synthetic",
            "config": {"workload": f"Silero {'v3.1' if args.model == 'v31' else 'v4'} 16k, batch={S} streams/GPU x {Cn} chunks/step, "
                                   f"{workload_label(args.model, mode, fe_kernel)}, s16le input resident in HBM",
                       "streams_per_gpu": S, "chunks_per_step": Cn, "hipgraph": bool(args.graph), "frontend_kernel": frontend_name(eng, fe_kernel),
                       "parallelism": f"streams sharded over {world} GPU(s), RCCL gather of probabilities"},
            "roofline": {"bound": "mfma" if d["pipe"] in ("fp16", "fp32") else "valu", "kernel": dom, "achieved": d["executed_tflops"], "peak": PEAKS[d["pipe"]],
                         "unit": "TFLOP/s", "frac": d["frac_of_pipe_peak"], "traffic": traffic, "traffic_source": traffic_source,
                         "avg_launch_ms": d["ms_per_launch"], "chunks_per_launch": d["chunks_per_launch"], "pipe": d["pipe"],
                         # the same rate against the fp32 FMA peak (what the vector ALU could do if the reference's tree allowed contraction), and the
                         # measured HBM bytes of the launch over its algorithmic bytes (the front end: 3,072 B of s16 samples per chunk; others: the whole path's 3.1 KB)
                         "frac_of_fma_peak": round(d["executed_tflops"] / PEAKS["fp32"], 4),
                         "algorithmic_bytes_per_launch": d["chunks_per_launch"] * (3072 if dom == "k_frontend" else 3136),
                         "traffic_over_algorithmic": (round(traffic / (d["chunks_per_launch"] * (3072 if dom == "k_frontend" else 3136)), 2) if traffic else None),
                         "executed_flop_per_chunk": d["executed_flop_per_chunk"][d["pipe"]],
                         "algorithmic_flop_per_chunk": d["algorithmic_flop_per_chunk"], "algorithmic_tflops": d["algorithmic_tflops"],
                         # the whole path in the dense formulation of SURVEY.md 8(d), as a rate only (no fraction: the kernels execute fewer FLOP than it counts)
                         "path_flop_per_chunk": PATH_FLOP_PER_CHUNK[args.model],
                         "path_algorithmic_tflops": round(PATH_FLOP_PER_CHUNK[args.model] * S * Cn * args.steps / elapsed / 1e12, 3),
                         "note": "dominant kernel by CU-time.  achieved = EXECUTED FLOP of that kernel's binding pipe per launch / HIP-event launch duration; peak = that pipe: "
                                 "valu_nofma 78.65 (fp32 vector ALU, products and sums rounded separately as the reference's STFT tree demands = half the FMA peak), "
                                 "fp32 157.3 (vector == matrix), fp16 2500 (split-fp16 GEMMs: 3 MFMAs per fp32 product).  k_frontend_sym evaluates the tree for 33 of "
                                 "the 129 bins (the rest follow from the basis' DFT symmetries, bit-exactly), so it EXECUTES 27 % of the dense-basis FLOP that "
                                 "algorithmic_* counts (SURVEY.md 8(d))"},
            "kernels": per_kernel,
            "lstm_trail": bool(eng.get_option("lstm_trail_used")) and layer_major,      # layer 1 of the recurrence beside layer 0 of the same call (what the last step did:
                                                                                          # under a tool that serialises kernels the engine does not)
            "stage_fracs": {k: [v["frac_of_pipe_peak"], v["pipe"]] for k, v in per_kernel.items()},
            "chunks_per_sec": round(chunks_per_step * args.steps / elapsed, 1),
            "host_issue_ms_per_step": round(issued / args.steps * 1e3, 4),   # what graph replay saves is host time: compare with --no-graph
            # `value` is the WHOLE JOB over all ranks (the driver's contract); the metric's "per GPU" figure is this one
            "value_per_gpu": round(value / world, 1), "total_streams": total_streams,
            "rccl": collective_facts(torch, dist, world, rehearsal, gather),
        }
        one = one_gpu_figures(args.one_gpu_json) if world > 1 else {}
        if world > 1:
            out["efficiency_vs_1gpu"] = vs_one_gpu(value, world, one, f"{S}x{Cn}")      # null without a 1-GPU figure of this shape (--one-gpu-json / the N = 1 run's cache)
        if world == 1 and not args.no_host_fed:
            # PCIe-inclusive rates of the host-buffer entry points (what a drop-in backend_run caller pays); reported, never `value`.
            #   value       -- vadc_amd_run_s16_async: page-locked host buffers, H2D of call k+1 beside the kernels of call k beside the D2H of call k-1
            #   synchronous -- vadc_amd_run_s16: pageable buffers, copy -> run -> copy
            # (the asynchronous leg's buffers come page-locked from torch's host allocator: the engine takes such ranges as they are -- hipHostRegister of heap pages is one more
            # userptr mapping this pool's runtime was seen to fault on, vadc_amd/staging.py; the synchronous leg keeps an ordinary pageable array)
            host = [pinned(pcm[:, i * Cn * 1536:(i + 1) * Cn * 1536]).numpy() for i in range(NB)]
            outs = [pinned(np.empty((S, Cn, 2), np.float32)).numpy() for _ in range(NB)]
            pageable = np.ascontiguousarray(pcm[:, : Cn * 1536])
            eng.set_option("groups", 1)
            # a SUSTAINED rate: 18 warm-up calls (the runtime's one-off stalls -- first use of each staging slot, page-locking, a 6.7 ms stall inside the
            # eleventh hipMemcpyAsync of a process -- are behind), then 48 calls: the pipeline's fill and drain (one copy in, one step + copy out: 2.3 ms)
            # is 3 % of the timed region.  Timed over 12 calls after 6 it read 1.18-1.21 M (38 GB/s) for the same steady state.
            for i in range(18):
                eng.run_async(host[i % NB], outs[i % NB])
            eng.wait_async()
            n_host = 48
            t1 = time.perf_counter()
            for i in range(n_host):
                eng.run_async(host[i % NB], outs[i % NB])
            eng.wait_async()
            dta = time.perf_counter() - t1
            eng.set_option("defer_join", 0)                       # what a plain synchronous caller gets: the call pipelines its own chunk groups
            eng.set_option("groups", 0)
            eng.run(pageable)
            n_sync = 4
            t1 = time.perf_counter()
            for _ in range(n_sync):
                eng.run(pageable)
            dts = time.perf_counter() - t1
            out["host_fed"] = {"value": round(S * Cn * n_host * CHUNK_SECONDS / dta, 1), "unit": "audio-seconds/sec", "ms_per_step": round(dta / n_host * 1e3, 3),
                               "pcie_gb_per_s": round(S * Cn * 3072 * n_host / dta / 1e9, 1),
                               "synchronous": round(S * Cn * n_sync * CHUNK_SECONDS / dts, 1),
                               "note": "vadc_amd_run_s16_async: page-locked host s16 in, probabilities out, three calls in flight (H2D 3 KB + D2H 8 B per chunk inside the "
                                       "timed region); `synchronous` = vadc_amd_run_s16 on pageable buffers (copy -> run -> copy)"}
        if world == 1 and not args.no_side_config and not args.verify_dump and args.model == "v31" and not (S == 4096 and Cn == 16):
            eng.close()
            eng = None
            blob_v4 = open(os.path.join(ROOT, "tests", "golden", "silero_v4_16k.testtensor"), "rb").read()
            out["configs"] = {"4096x16": side_config(torch, blob, dev, local_rank, args.model, 4096, 16, 1, steps=60, warmup=10),      # (60 steps: what the N-rank entry configs["<N>x4096x16"] runs)
                              "10240x1": side_config(torch, blob, dev, local_rank, args.model, 10240, 1, 0, steps=200, warmup=20, latency=True, host_fed=True),
                              "256x96_fp32_mfma": side_config(torch, blob, dev, local_rank, args.model, 256, 96, 0, opts={"encoder": 3, "lstm": 3, "layer1": 1}),
                              # BASELINE config 4: Silero v4 at 4096 streams.  Last: an engine created in front of the 10,240 x 1 configuration moved that one's internal
                              # streams onto other hardware queues and cost it 40 % (1.80 against 2.94 M alone and in this order: DESIGN.md section 6)
                              "v4_4096x16": side_config(torch, blob_v4, dev, local_rank, "v4", 4096, 16, 0, steps=100, warmup=10)}
            # Silero v5 shapes, after every BASELINE configuration: 256 streams x 288 windows (= 96 v3.1 chunks of audio per stream and call) and 4096 x 48
            blob_v5 = open(os.path.join(ROOT, "tests", "golden", "silero_v5_seeded.testtensor"), "rb").read()
            out["configs"]["v5_256x288"] = side_config_v5(torch, blob_v5, dev, local_rank, 256, 288)
            out["configs"]["v5_4096x48"] = side_config_v5(torch, blob_v5, dev, local_rank, 4096, 48)
        if cpu is not None:
            out["cpu_baseline"] = cpu

    # BASELINE config 5's shape under the SAME N ranks and the same collective: 4096 streams per GPU x 16 chunks per step, SPLIT16 (config 3's arithmetic),
    # graph replay, the gather inside the timed region, barriers on both sides, max over ranks.  Every rank takes part; rank 0 reports it as
    # configs["<N>x4096x16"].  `value` stays on the default shape so that the N = 1 line agrees with the single-GPU bench.
    c5_S, c5_C = (int(v) for v in args.config5_shape.lower().split("x"))
    if world > 1 and not args.no_side_config and not args.verify_dump and args.model == "v31":
        if eng is not None:
            eng.close()
            eng = None
        c5 = side_config(torch, blob, dev, local_rank, args.model, c5_S, c5_C, 1, steps=args.config5_steps, warmup=10, world=world, rank=rank, rehearsal=rehearsal)
        if rank == 0:
            c5["efficiency_vs_1gpu"] = vs_one_gpu(c5["value"], world, one, f"{c5_S}x{c5_C}")
            c5["one_gpu_figure"] = one.get(f"{c5_S}x{c5_C}")
            out.setdefault("configs", {})[f"{world}x{c5_S}x{c5_C}"] = c5

    if rank == 0:
        if world == 1 and args.model == "v31" and not args.verify_dump:
            # what an N-rank run of this bench on the same box compares itself with (efficiency_vs_1gpu): this run's figures by shape
            try:
                os.makedirs(os.path.dirname(os.path.abspath(args.one_gpu_json)), exist_ok=True)
                figs = {f"{S}x{Cn}": out["value"]}
                figs.update({k: v["value"] for k, v in out.get("configs", {}).items() if k in ("4096x16",)})
                with open(args.one_gpu_json, "w") as f:
                    json.dump(figs, f)
            except OSError:
                pass
        # The line the driver parses stays SHORT (it reads the tail of stdout): the per-kernel accounting and the long notes go to a details file
        # (the same object, verbose), the line keeps every contract field plus the per-kernel launch times.
        try:
            os.makedirs(os.path.dirname(os.path.abspath(args.details)), exist_ok=True)
            with open(args.details, "w") as f:
                json.dump(out, f)
        except OSError:
            pass
        line = dict(out)
        line["roofline"] = {k: v for k, v in out["roofline"].items() if k != "note"}
        line["kernels_ms"] = {k: v["ms_per_launch"] for k, v in out["kernels"].items()}
        del line["kernels"]
        if "host_fed" in line:
            line["host_fed"] = {k: v for k, v in out["host_fed"].items() if k != "note"}
        if "cpu_baseline" in line:
            c = dict(out["cpu_baseline"])
            c["sample"] = c["sample"][:120]
            line["cpu_baseline"] = c
        line["details"] = os.path.relpath(os.path.abspath(args.details), ROOT)
        if "configs" in line:
            line["configs"] = {k: {kk: vv for kk, vv in v.items() if kk not in ("kernels_ms", "note")} for k, v in out["configs"].items()}
        print(json.dumps(line, separators=(",", ":")), flush=True)
    if eng is not None:
        eng.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500,
                    help="timed steps (default 500: the fill and drain of the two-deep step pipeline -- one LSTM launch that nothing overlaps -- "
                         "is a few %% of a 20-step run and 0.1 %% of this one)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--streams", type=int, default=256, help="streams PER GPU (BASELINE config 2: 256)")
    ap.add_argument("--chunks-per-step", type=int, default=96, help="chunks per stream and step (default 96 = vadc's window, vadc.c:799)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-fed", action="store_true")
    ap.add_argument("--total-streams", type=int, default=0, help="(with --verify-dump) total streams over all ranks, partitioned in contiguous blocks -- may be ragged")
    ap.add_argument("--verify-dump", default="", metavar="FILE.npz",
                    help="after the timed region: reset the state, run 2 x buffers - 1 steps with one distinct signal per GLOBAL stream and write rank 0's gathered "
                         "probabilities of every step (what tests/test_bench_spawn.py checks against the CPU oracle)")
    ap.add_argument("--no-side-config", action="store_true", help="skip the 4096 x 16 (BASELINE config 3) measurement that rides along with the default line")
    ap.add_argument("--model", choices=["v31", "v4"], default="v31",
                    help="v31 = Silero v3.1 (BASELINE headline, default); v4 = Silero v4 16k (BASELINE config 4, not the headline)")
    ap.add_argument("--precision", choices=["fp32", "split16", "fast_stft"], default="fp32",
                    help="fp32 = the parity mode (default, BASELINE config 2); split16 = BASELINE config 3 (exact STFT + split-fp16 GEMMs, within 1e-4); "
                         "fast_stft = throughput mode, STFT as a GEMM, NOT within 1e-4 (see include/vadc_amd.h)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT",
                    help="engine tuning switch (vadc_amd_set_option), e.g. --opt frontend=1; experiments only")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="experiment: no per-kernel HIP events inside the timed region (kernel table then comes from a separate pass)")
    ap.add_argument("--no-graph", dest="graph", action="store_false",
                    help="eager launches only.  Default: hipGraph replay of the step's kernel sequences (north star: hipGraph-captured steady-state steps); "
                         "every 8th step of the timed region is still issued eagerly so that its kernels carry HIP events")
    ap.set_defaults(graph=True)
    ap.add_argument("--groups", type=int, default=1,
                    help="chunk groups per step inside the engine (1: whole step per launch; steps overlap each other "
                         "through the two caller streams; 0 = engine default for single-stream callers)")
    ap.add_argument("--strict-join", dest="defer_join", action="store_false",
                    help="every call makes its own stream wait for its completion (strict stream semantics) and the steps rotate over --caller-streams streams.  "
                         "Default: engine option defer_join = 1 -- all steps are issued from ONE stream, which a call does not block; the consumer of a step's "
                         "probabilities (the RCCL gather, the final synchronize) joins it with vadc_amd_join")
    ap.set_defaults(defer_join=True)
    ap.add_argument("--caller-streams", type=int, default=3, help="step buffers used in turn (= caller streams with --strict-join)")
    ap.add_argument("--details", default=os.path.join(ROOT, "gpurun_out", "bench_details.json"),
                    help="where the verbose form of the line goes (per-kernel executed / algorithmic FLOP, pipes, notes)")
    ap.add_argument("--config5-shape", default="4096x16", metavar="SxC",
                    help="with --gpus N > 1: streams per GPU x chunks per step of the BASELINE config 5 measurement that rides along as configs[\"<N>xSxC\"] (default 4096x16; "
                         "tests use small blocks)")
    ap.add_argument("--config5-steps", type=int, default=60)
    ap.add_argument("--one-gpu-json", default=os.path.join(ROOT, "gpurun_out", "bench_one_gpu.json"),
                    help="the 1-GPU figures by shape ({\"256x96\": v, \"4096x16\": v}): written by an N = 1 run, read by an N > 1 run for efficiency_vs_1gpu")
    ap.add_argument("--one-gpu-rehearsal", action="store_true",
                    help="N ranks share GPU 0 and gather over gloo through the host: rehearses the multi-rank code path on a one-GPU box (its numbers mean nothing)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: rehearse spawn / rendezvous / sharding / gather / timing over gloo on the CPU")
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-batch", type=int, default=96, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_worker:
        wp = os.path.join(ROOT, "tests", "golden", "silero_v4_16k.testtensor" if args.model == "v4" else os.path.join("reference_fixtures", "silero_v31_16k.testtensor"))
        cpu_worker(args.model, wp, args.cpu_seconds, args.cpu_batch)
        return 0
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.caller_streams < 1 or (args.gpus > 1 and args.defer_join and args.caller_streams < 2):
        raise SystemExit("--caller-streams must be >= 1, and >= 2 on several GPUs with deferred joins (one issuing stream + at least one side stream for the gather)")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus)                  # nothing above imported torch or touched HIP
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    args.rank_cpus_n = len(pin_rank_to_its_cpus(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))))
    if args.dry_run:
        return dry_run(args, world, rank)
    rc = run_rank(args, world, rank, local_rank)
    # A process that has used the GPU leaves WITHOUT the HIP runtime's exit handlers: beside another process's GPU context (the other ranks of this job) they hung one
    # short-lived process in about two hundred on ROCm 7.2 -- after main() had returned, every engine destroyed (tools/cli_teardown_probe.py).  Everything is flushed first.
    # (Not under a profiler -- rocprofv3 writes its files from exit handlers -- and not when the process is alone on its GPU.)
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
    if world > 1 and not profiled:
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(int(rc or 0))
    return rc


if __name__ == "__main__":
    sys.exit(main())
