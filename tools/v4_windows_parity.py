"""Silero v4 at every window the engine serves against the oracle on 16 long synthetic streams: max / p99.9 / mean |dp| per window.   python tools/v4_windows_parity.py [windows ...]"""
import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle as O
from vadc_amd import synth
from vadc_amd.engine import Engine
EIGHT = len(sys.argv) > 1 and sys.argv[1] == "8k"                     # python tools/v4_windows_parity.py 8k [windows ...]: the graph's 8 kHz branch (256 ... 768)
blob4 = open("tests/golden/silero_v4_8k.testtensor" if EIGHT else "tests/golden/silero_v4_16k.testtensor", "rb").read()
orc = O.OracleV4(blob4)
base = synth.make_streams(16, 400, seed0=52000)
WINDOWS = [int(a) for a in sys.argv[(2 if EIGHT else 1):]] or (list(range(256, 769, 64)) if EIGHT else list(range(512, 1537, 64)))
for w, opts in [(w_, {}) for w_ in WINDOWS]:
    e = Engine(blob4, max_streams=16, max_chunks_per_call=80, device=0)
    e.set_window(w)
    for k, v in opts.items(): e.set_option(k, v)
    n = 640
    pcm = np.ascontiguousarray(base[:, : n * w]) if n * w <= base.shape[1] else np.ascontiguousarray(base[:, : (base.shape[1] // w) * w])
    n = pcm.shape[1] // w
    n -= n % 80
    pcm = np.ascontiguousarray(pcm[:, : n * w])
    x = (pcm.astype(np.float32) / np.float32(32768)) if os.environ.get("F32") else pcm      # F32=1: the f32 entry point (k_frontend_gemm's first form)
    got = np.concatenate([e.run(x[:, i * w:(i + 80) * w]) for i in range(0, n, 80)], axis=1)[:, :, 1]
    want = orc.forward_streams(pcm, window=w)
    d = np.abs(got.astype(np.float64) - want)
    i = np.unravel_index(d.argmax(), d.shape)
    print(json.dumps({"window": w, "opts": opts, "chunks": n, "max": float(d.max()), "p999": float(np.quantile(d, 0.999)), "mean": float(d.mean()), "at": [int(i[0]), int(i[1])], "p_at": float(want[i])}), flush=True)
    e.close()
