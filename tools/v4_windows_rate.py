#!/usr/bin/env python3
"""Silero v4 at every window the engine serves (option "window" = --sequence_count of the reference's onnxruntime path, onnx_helpers.c:164-170, vadc.c:743-752):
S streams x C windows per call, s16 resident in HBM, deferred joins, graph replay, K timed calls.  The default window (1536) runs k_frontend_gemm2 + k_layer1_regs_v4 +
k_enc_fused_v4; the others k_frontend_gemm2 + one k_layer_mfma launch per stage.  One JSON line per window (for profiles/rNN/v4_windows.jsonl)."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=4096)
    ap.add_argument("--chunks", type=int, default=16)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rate", type=int, default=16000, choices=[16000, 8000])
    ap.add_argument("--windows", type=int, nargs="*", default=None, help="windows to time (default: the built ones of the branch)")
    a = ap.parse_args()
    wins = tuple(a.windows) if a.windows else ((1536, 1280, 1024, 768, 512) if a.rate == 16000 else (768, 512, 256))
    if len(wins) > 1:
        # one process per window: the FOURTH engine a process creates runs 15 - 40 % slower, every kernel of it (its streams land on hardware queues that earlier engines'
        # streams still hold: DESIGN.md 7a) -- a 768-sample window measured fourth read 3.5 M instead of 4.6 M
        import subprocess
        for W in wins:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--streams", str(a.streams), "--chunks", str(a.chunks), "--steps", str(a.steps), "--warmup", str(a.warmup),
                            "--rate", str(a.rate), "--windows", str(W)], check=False)
        return
    import torch
    from vadc_amd.engine import Engine
    from vadc_amd.staging import to_device, to_host
    from vadc_amd import synth
    name = "silero_v4_16k.testtensor" if a.rate == 16000 else "silero_v4_8k.testtensor"
    blob = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
    default_window = 1536 if a.rate == 16000 else 768
    windows = tuple(a.windows) if a.windows else ((1536, 1280, 1024, 768, 512) if a.rate == 16000 else (768, 512, 256))
    for W in windows:
        eng = Engine(blob, max_streams=a.streams, max_chunks_per_call=a.chunks, device=0)
        if W != default_window:
            eng.set_window(W)
        eng.set_option("defer_join", 1)
        n = a.chunks * W
        base = synth.make_streams(16, -(-n // 1536), seed0=11)[:, :n]
        pcm = np.ascontiguousarray(np.tile(base, (-(-a.streams // 16), 1))[:a.streams])
        d_in = to_device(pcm)
        d_out = torch.empty(a.streams, a.chunks, 2, device="cuda")
        st = torch.cuda.Stream()
        for _ in range(3):
            eng.run_device(d_in.data_ptr(), np.int16, a.streams, a.chunks, d_out.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        eng.set_option("graph", 1)
        for _ in range(a.warmup):
            eng.run_device(d_in.data_ptr(), np.int16, a.streams, a.chunks, d_out.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        eng.reset_kernel_times()
        t0 = time.perf_counter()
        for i in range(a.steps):
            eng.set_profiling(i % 8 == 3)
            eng.run_device(d_in.data_ptr(), np.int16, a.streams, a.chunks, d_out.data_ptr(), st.cuda_stream)
        eng.join(st.cuda_stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        eng.set_profiling(False)
        kt = {k: round(ms / c, 4) for k, (c, ms) in eng.kernel_times().items() if c}
        print(json.dumps({"model": f"Silero v4 {a.rate // 1000} kHz", "window": W, "lstm_steps": eng.caps()["lstm_steps_per_chunk"], "streams": a.streams, "chunks_per_step": a.chunks,
                          "value": round(a.streams * a.chunks * a.steps * (W / a.rate) / dt, 1), "unit": "audio-seconds/sec", "ms_per_step": round(dt / a.steps * 1e3, 4),
                          "layer1_kernel": {0: "k_layer1_regs_v4", 1: "k_layer_mfma (per stage)"}.get(eng.get_option("layer1_kernel")), "kernels_ms": kt}), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
