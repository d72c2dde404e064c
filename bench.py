#!/usr/bin/env python3
"""bench.py -- throughput of the Silero v3.1 hot path on MI355X (metric of BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--streams S] [--chunks-per-step C]

A "step" = one pass of the hot path (front end -> 4 encoder layers -> LSTM+decoder) over one batch of
synthetic 16 kHz s16le audio: S independent streams x C consecutive 1536-sample chunks per stream, LSTM state
carried on the device from step to step.  Defaults: S = 256 (BASELINE config 2), C = 96 = the window vadc hands its backend per
stream and call (`chunks_count = 96` vadc.c:799, `--batch` default 96 vadc.c:1116).  Inputs are resident in HBM before the timed region.
value = streams x chunks x 0.096 s / wall_s  (audio-seconds per second == concurrent real-time streams),
whole job over all ranks.  For N > 1 the driver launches one rank per GPU (torch.distributed, RCCL); streams
are sharded across ranks with no data-path collective; the per-step speech probabilities are gathered to
rank 0 with one RCCL gather (north star), inside the timed region.

The JSON line also carries
  roofline     -- dominant kernel: algorithmic FLOP per launch / HIP-event duration vs the fp32 peak (events recorded inside the
                  timed region, on the kernel's own stream, on every 4th step)
  cpu_baseline -- the CPU oracle (kind "port") or oracle/_ref (kind "reference") timed on this box's host cores
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHUNK_SECONDS = 1536 / 16000.0
# algorithmic work per chunk, FLOP = 2 x MAC
FLOP_PER_CHUNK = {
    "k_frontend": 2 * 1_651_200,
    "k_layer1": 2 * 181_053, "k_layer2": 2 * 112_208, "k_layer3": 2 * 61_600, "k_layer4": 2 * 236_768,
    "k_lstm": 2 * (458_752 + 896),                 # both layers incl. the input projection + decoder
}
# Silero v4 (BASELINE config 4, `--model v4`): parity-test configuration, not the headline
FLOP_PER_CHUNK_V4 = {
    "k_frontend": 2 * 1_585_152,
    "k_layer1": 2 * 232_176, "k_layer2": 2 * 19_392, "k_layer3": 2 * 10_176, "k_layer4": 2 * 25_056,
    "k_lstm": 2 * (196_608 + 192),
}
PATH_FLOP_PER_CHUNK = {"v31": 2 * 2_702_477, "v4": 2 * 2_068_752}      # whole path
PEAK_FP32_TFLOPS = 157.3          # MI355X_MICROARCH.md: vector == matrix fp32 peak
PEAK_FP16_TFLOPS = 2500.0         # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense


def cpu_baseline(blob, weights_path, seconds_budget=12.0, model="v31"):
    """Reported baseline, not the target: the oracle on ONE host core over a bounded sample."""
    from oracle import oracle as O
    from vadc_amd import synth
    base = 2048                                  # chunks of one synthetic speech stream (3.3 min of audio), repeated
    pcm = synth.speech_like(base * 1536, seed=9)
    kind, runner = "port", None
    try:
        if model == "v4":                       # the reference has no C implementation of v4: only the restatement exists
            raise FileNotFoundError
        ref = O.Reference(weights_path)
        x = pcm.astype(np.float32) / np.float32(32768)
        kind, runner = "reference", (lambda n: ref.run(x[: n * 1536], batch=96))
    except (FileNotFoundError, OSError, ValueError):
        orc = O.OracleV4(blob) if model == "v4" else O.Oracle(blob)
        runner = lambda n: orc.forward_stream(pcm[: n * 1536])
    runner(8)
    t0 = time.perf_counter(); runner(256); dt = time.perf_counter() - t0
    reps = int(min(64, max(1, round(seconds_budget * 256 / max(dt, 1e-6) / base))))      # ~seconds_budget of CPU work
    t0 = time.perf_counter()
    for _ in range(reps):
        runner(base)
    dt = time.perf_counter() - t0
    return {"value": round(reps * base * CHUNK_SECONDS / dt, 2), "unit": "audio-seconds/sec", "cores": 1, "kind": kind,
            "sample": f"{reps} x {base} consecutive chunks of one synthetic speech stream ({dt:.1f} s of CPU work), single thread"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500,
                    help="timed steps (default 500 = about 1 s: the fill and drain of the two-deep step pipeline -- one LSTM launch that nothing overlaps -- "
                         "is 2.7 %% of a 20-step run and 0.1 %% of this one)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--streams", type=int, default=256, help="streams PER GPU (BASELINE config 2: 256)")
    ap.add_argument("--chunks-per-step", type=int, default=96, help="chunks per stream and step (default 96 = vadc's window, vadc.c:799)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--model", choices=["v31", "v4"], default="v31",
                    help="v31 = Silero v3.1 (BASELINE headline, default); v4 = Silero v4 16k (BASELINE config 4, not the headline)")
    ap.add_argument("--precision", choices=["fp32", "split16"], default="fp32",
                    help="fp32 = the parity mode (default, BASELINE config 2); split16 = BASELINE config 3: STFT as a split-fp16 GEMM on "
                         "the matrix pipe instead of the reference's reduction tree (|dp| up to ~1e-4, see include/vadc_amd.h)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT",
                    help="engine tuning switch (vadc_amd_set_option), e.g. --opt frontend=1; experiments only")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="experiment: no per-kernel HIP events inside the timed region (kernel table then comes from a separate pass)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a hipGraph (kernel timing then comes from a separate pass)")
    ap.add_argument("--groups", type=int, default=1,
                    help="chunk groups per step inside the engine (1: whole step per launch; steps overlap each other "
                         "through the two caller streams; 0 = engine default for single-stream callers)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from vadc_amd import synth
    from vadc_amd.engine import Engine

    weights_path = os.path.join(ROOT, "tests", "golden", "reference_fixtures", "silero_v31_16k.testtensor")
    if args.model == "v4":
        weights_path = os.path.join(ROOT, "tests", "golden", "silero_v4_16k.testtensor")
        FLOP_PER_CHUNK.clear(); FLOP_PER_CHUNK.update(FLOP_PER_CHUNK_V4)
    blob = open(weights_path, "rb").read()
    S, Cn = args.streams, args.chunks_per_step
    eng = Engine(blob, max_streams=S, max_chunks_per_call=Cn, device=local_rank, precision=1 if args.precision == "split16" else 0)
    split16 = args.precision == "split16" and eng.caps()["precision"] == 1
    eng.set_option("groups", args.groups)
    for kv in args.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))

    # synthetic input: 16 distinct speech-like streams per rank tiled over S, two alternating step buffers
    base = synth.make_streams(min(S, 16), 2 * Cn, seed0=1234 + 100 * rank)
    pcm = np.ascontiguousarray(np.tile(base, ((S + base.shape[0] - 1) // base.shape[0], 1))[:S])
    d_in = [torch.from_numpy(np.ascontiguousarray(pcm[:, i * Cn * 1536:(i + 1) * Cn * 1536])).to(f"cuda:{local_rank}") for i in range(2)]
    d_probs = [torch.empty((S, Cn, 2), dtype=torch.float32, device=f"cuda:{local_rank}") for _ in range(2)]
    gather_list = [torch.empty_like(d_probs[0]) for _ in range(world)] if (world > 1 and rank == 0) else None
    from vadc_amd import shard
    lo, hi = shard.stream_block(rank, world, S * world)     # weak scaling: S streams per GPU, contiguous blocks
    assert hi - lo == S
    # Two caller streams used alternately: each step is strictly ordered on its own stream, and the engine's
    # internal in-order streams overlap step k+1's front end + encoder with step k's LSTM.
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def step(i):
        st = streams[i & 1]
        with torch.cuda.stream(st):
            eng.run_device(d_in[i & 1].data_ptr(), np.int16, S, Cn, d_probs[i & 1].data_ptr(), st.cuda_stream)
            if world > 1:
                dist.gather(d_probs[i & 1], gather_list, dst=0)   # the only collective: final probability gather

    for i in range(2):                     # setup, not warm-up: the first calls create the engine's internal streams / CU masks and touch every buffer once
        step(i)
    torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    eng.reset_kernel_times()
    separate_pass = args.graph or args.no_kernel_timing
    eng.set_profiling(not separate_pass)  # per-kernel HIP events need eager launches
    if args.graph:
        eng.set_option("graph", 1)
        step(0); step(1)                  # capture both input buffers outside the timed region
        torch.cuda.synchronize()
    # Per-kernel HIP events (two hipEventRecord per launch, on the launch's stream) cost ~2.5 % of the step when every launch
    # of the timed region carries them; they are recorded on every 4th step of the timed region instead (still "live", still
    # on the kernel's own stream), which keeps `value` within ~0.6 % of an event-free run.
    prof_every = 4
    n_prof = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        if not separate_pass:
            on = (i % prof_every) == 0
            eng.set_profiling(on)
            n_prof += int(on)
        step(i)
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

    t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        chunks_per_step = S * Cn * world
        value = chunks_per_step * args.steps * CHUNK_SECONDS / elapsed
        kt = eng.kernel_times()
        # The LSTM chain runs concurrently on its own small CU partition; weigh every kernel's
        # time by the share of the chip it occupies so that "dominant" means dominant in CU-time, not in wall time
        # of a kernel that leaves 240 CUs to the others.
        n_cus = torch.cuda.get_device_properties(local_rank).multi_processor_count
        cu_share = {k: 1.0 for k in kt}
        lstm_cus = eng.get_option("lstm_cus")                     # CUs the engine reserved for the LSTM chain (0: whole chip)
        cu_share["k_lstm"] = (lstm_cus / n_cus) if lstm_cus > 0 else 1.0
        dom = max(kt, key=lambda k: kt[k][1] * cu_share[k])
        launches, total_ms = kt[dom]
        avg_s = total_ms / max(launches, 1) / 1e3
        # chunks one launch of the dominant kernel processes (a step may be split into chunk groups)
        chunks_per_launch = S * Cn * n_prof / max(launches, 1)
        achieved = FLOP_PER_CHUNK[dom] * chunks_per_launch / avg_s / 1e12
        traffic = None
        try:   # HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, KB -> B)
            # a workload only gets the bytes of passes collected on exactly that workload (tools/rocprof_reduce.py names the files)
            default_workload = args.model == "v31" and not split16 and S == 256 and Cn == 96
            name = "latest_pmc_traffic.json" if default_workload else f"latest_pmc_traffic_{args.model}_{args.precision}_{S}x{Cn}.json"
            prof = json.load(open(os.path.join(ROOT, "profiles", name)))
            if (prof.get("streams") == S and prof.get("chunks_per_step") == Cn and prof.get("model", "v31") == args.model
                    and prof.get("precision", "fp32") == args.precision and dom in prof.get("kernels", {})):
                traffic = prof["kernels"][dom]["hbm_bytes_per_launch"]
        except (OSError, ValueError):
            pass
        per_kernel = {}
        for k, (n_l, ms) in kt.items():
            if n_l:
                per_kernel[k] = {"ms_per_launch": round(ms / n_l, 4), "cu_share": round(cu_share[k], 4),
                                 "tflops": round(FLOP_PER_CHUNK[k] * (S * Cn * n_prof / n_l) / (ms / n_l / 1e3) / 1e12, 3)}
        out = {
            "metric": "audio-seconds/sec (= real-time streams) per GPU, Silero v3.1 16k" if args.model == "v31" else
                      "audio-seconds/sec (= real-time streams) per GPU, Silero v4 16k (BASELINE config 4; not the headline metric)",
            "value": round(value, 1), "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if not split16 else "split-f16 (2 x fp16, fp32 accumulate) front end + f32", "data": "synthetic",
            "config": {"workload": f"Silero {'v3.1' if args.model == 'v31' else 'v4'} 16k, batch={S} streams/GPU x {Cn} chunks/step, "
                                   f"{'fp32' if not split16 else 'SPLIT16 precision mode (BASELINE config 3; not the parity mode)'}, s16le input resident in HBM",
                       "streams_per_gpu": S, "chunks_per_step": Cn, "hipgraph": bool(args.graph), "parallelism": f"streams sharded over {world} GPU(s), RCCL gather of probabilities"},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_TFLOPS, 4), "traffic": traffic,
                         "avg_launch_ms": round(avg_s * 1e3, 4), "chunks_per_launch": int(chunks_per_launch),
                         "algorithmic_flop_per_chunk": FLOP_PER_CHUNK[dom],
                         # blended figure over the whole path: all kernels' algorithmic FLOP per chunk
                         # x chunks/s of the job on one GPU, against the same fp32 peak
                         "path_flop_per_chunk": PATH_FLOP_PER_CHUNK[args.model],
                         "path_achieved": round(PATH_FLOP_PER_CHUNK[args.model] * S * Cn * args.steps / elapsed / 1e12, 3),
                         "path_frac": round(PATH_FLOP_PER_CHUNK[args.model] * S * Cn * args.steps / elapsed / 1e12 / PEAK_FP32_TFLOPS, 4),
                         "note": "dominant kernel by CU-time; fp32 peak (vector == matrix); the bit-exact STFT is unfused "
                                 "mul+add (2 VALU instructions per MAC) => its ceiling is frac 0.5"},
            "kernels": per_kernel,
            "chunks_per_sec": round(chunks_per_step * args.steps / elapsed, 1),
        }
        gemm_fe = (args.model == "v4" and eng.get_option("frontend") == 0) or split16
        if gemm_fe and dom == "k_frontend":
            # the GEMM front end executes the FOLDED real-input DFT as split-fp16 products on the fp16 matrix pipe: 3 MFMAs per
            # k-block, 256 rows (8 re + 8 im tiles) x K = 128 per position -- price it on what it executes against that pipe's peak
            frames = 24 if args.model == "v4" else 25
            executed = 3 * 2 * 256 * 128 * frames
            ach = executed * chunks_per_launch / avg_s / 1e12
            out["roofline"].update({"achieved": round(ach, 3), "peak": PEAK_FP16_TFLOPS, "frac": round(ach / PEAK_FP16_TFLOPS, 4),
                                    "executed_mfma_flop_per_chunk": executed,
                                    "algorithmic_tflops_dense_basis": round(achieved, 3)})
        if args.model == "v4" or split16:
            out["roofline"]["note"] = ("dominant kernel by CU-time; fp32 peak (vector == matrix).  FLOP are counted for the DENSE basis "
                                       "(SURVEY.md 8(d): 2 x 258 x 256 x 24 per chunk for the front end); k_frontend_gemm folds the "
                                       "real-input DFT (x[n] +- x[256-n]) and EXECUTES half of them, as split-fp16 products on the fp16 matrix pipe "
                                       "(3 x v_mfma_f32_16x16x32_f16 per k-block).  When k_frontend is the dominant kernel its achieved / peak / frac are EXECUTED "
                                       "split-fp16 MFMA FLOP against the fp16 dense peak (the kernel is not bound by the matrix pipe: HBM writes for the v4 geometry, "
                                       "the per-tile barrier for v3.1's, DESIGN.md 4.5); path_frac stays algorithmic FLOP (dense basis) against the fp32 roof")
            out["roofline"]["executed_flop_per_chunk_frontend"] = 2 * (129 * 128 + 128 * 128) * (24 if args.model == "v4" else 25)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(blob, weights_path, model=args.model)
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
