#!/usr/bin/env python3
"""Throughput of the hot path versus the number of concurrent streams (report at B in {1, 256, 4096},
sweep 1..4096).  Runs bench.py once per point on this GPU and writes gpurun_out/sweep_streams.json.
   python tools/sweep_streams.py [--model v31|v4]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
model = sys.argv[sys.argv.index("--model") + 1] if "--model" in sys.argv else "v31"
POINTS = [(1, 96), (4, 96), (16, 96), (64, 96), (128, 96), (192, 96), (256, 96), (288, 96), (320, 96), (384, 96), (512, 96),      # (streams, chunks per step)
          (640, 32), (768, 32), (896, 32), (1024, 32), (1280, 32), (1664, 32), (2048, 32), (3072, 32), (4096, 16),           # the scheduling rules' whole range
          (256, 8), (256, 16), (1024, 1), (4096, 1), (4096, 4), (16384, 1)]   # serving with minimum latency: few 96-ms chunks per call
rows = []
for S, C in POINTS:
    for graph in ((True, False) if S <= 16 else (True,)):           # graph replay is bench.py's default; small calls also as eager launches
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--streams", str(S), "--chunks-per-step", str(C), "--no-cpu-baseline", "--no-host-fed", "--no-side-config",
               "--model", model, "--steps", "150", "--warmup", "10"] + ([] if graph else ["--no-graph"])
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print("FAILED", S, C, out.stderr[-400:], file=sys.stderr); continue
        d = json.loads(line[-1])
        rows.append({"streams": S, "chunks_per_step": C, "hipgraph": graph, "audio_seconds_per_sec": d["value"], "ms_per_step": d["ms_per_step"],
                     "path_algorithmic_tflops": d["roofline"].get("path_algorithmic_tflops"), "kernels_ms": d.get("kernels_ms")})
        print(rows[-1], flush=True)
json.dump({"model": model, "unit": "audio-seconds/sec (= real-time streams), one MI355X, fp32", "points": rows},
          open(os.path.join(ROOT, "gpurun_out", f"sweep_streams_{model}.json"), "w"), indent=1)
