#!/usr/bin/env python3
"""Which LSTM kernel for which call shape: bench.py per (streams, chunks per step) x option "lstm" (0 = the engine's rule, 6, 7, 8).
   python tools/lstm_variant_sweep.py [variants, e.g. 0,6,7,8]   -> gpurun_out/lstm_variant_sweep.json"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,6,7,8").split(",")]
POINTS = [(1, 96), (16, 96), (64, 96), (256, 96), (512, 96), (1024, 32), (2048, 32), (4096, 16), (4096, 1), (10240, 1), (16384, 1)]
rows = []
for S, C in POINTS:
    for v in variants:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--streams", str(S), "--chunks-per-step", str(C), "--no-cpu-baseline", "--no-host-fed", "--no-side-config",
               "--steps", "120", "--warmup", "10", "--opt", f"lstm={v}"]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print("FAILED", S, C, v, out.stderr[-300:], file=sys.stderr); continue
        d = json.loads(line[-1])
        rows.append({"streams": S, "chunks_per_step": C, "lstm": v, "audio_seconds_per_sec": d["value"], "ms_per_step": d["ms_per_step"]})
        print(rows[-1], flush=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "lstm_variant_sweep.json"), "w"), indent=1)
