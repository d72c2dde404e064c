#!/usr/bin/env python3
"""Run-to-run determinism of the shipped engine under its own concurrency: for each call shape the same calls (carried state, deferred joins, graph replay and eager
launches in turn) are issued again and again from reset state and every run's probabilities are compared with the first run's, bit for bit.  The kernels of
consecutive calls overlap on the device (front end + encoder of call k+1 beside the recurrence of call k, on shared or partitioned CUs depending on the shape), so a
result that depends on what runs beside it shows here.  One JSON line per shape.   python tools/soak_determinism.py [scale]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vadc_amd import synth
from vadc_amd.engine import Engine
from vadc_amd.staging import to_device, to_host
BLOBS = {"v31": "tests/golden/reference_fixtures/silero_v31_16k.testtensor", "v4": "tests/golden/silero_v4_16k.testtensor", "v5": "tests/golden/silero_v5_seeded.testtensor"}
bits = lambda a: np.ascontiguousarray(a).view(np.uint32)
scale = 1.0
def soak(S, Cn, calls, reps, opts=None, model="v31"):
    blob = open(BLOBS[model], "rb").read()
    e = Engine(blob, max_streams=S, max_chunks_per_call=Cn, device=0)
    if opts and "window" in opts:                                   # (Silero v4 at another window: before the inputs are cut)
        e.set_window(opts["window"])
    W = e.window                                                   # samples per chunk (1536; Silero v5: 512)
    base = synth.make_streams(min(S, 48), -(-calls * Cn * W // 1536), seed0=4000 + S)
    pcm = np.ascontiguousarray(base[np.arange(S) % base.shape[0]])
    d_in = [to_device(np.ascontiguousarray(pcm[:, k * Cn * W:(k + 1) * Cn * W])) for k in range(calls)]
    for k_, v_ in (opts or {}).items():
        if k_ != "window": e.set_option(k_, v_)
    e.set_option("defer_join", 1)
    st = torch.cuda.Stream()
    first, bad, worst = None, 0, 0.0
    reps = max(2, int(reps * scale))
    for rep in range(reps):
        e.set_option("graph", rep & 1); e.reset_streams()
        d_out = [torch.empty((S, Cn, 2), dtype=torch.float32, device="cuda:0") for _ in range(calls)]
        for k in range(calls):
            e.run_device(d_in[k].data_ptr(), np.int16, S, Cn, d_out[k].data_ptr(), st.cuda_stream)
        e.join(st.cuda_stream); st.synchronize()
        r = np.concatenate([to_host(o) for o in d_out], axis=1)
        if first is None: first = r
        elif not np.array_equal(bits(first), bits(r)):
            bad += 1; worst = max(worst, float(np.abs(first - r).max()))
    rec = ({"model": model, "streams": S, "chunks_per_call": Cn, "calls": calls, "runs": reps, "options": opts or {}, "lstm_kernel": e.get_option("lstm_kernel"), "lstm_cus": e.get_option("lstm_cus"),
                      "lstm_trail_used": e.get_option("lstm_trail_used"), "runs_differing_from_the_first": bad, "max_abs_dp": worst})
    print(json.dumps(rec), flush=True)
    e.close()
    return rec
SHAPES = [(10240, 1, 8, 600), (16384, 1, 4, 200), (4096, 1, 8, 300), (256, 96, 3, 120), (256, 8, 8, 300), (100, 24, 4, 300), (640, 8, 6, 200), (1024, 4, 8, 200), (4096, 16, 2, 60), (256, 96, 3, 60, {"lstm_trail": 0}), (10240, 1, 8, 200, {"lstm": 6}),
          (256, 96, 3, 60, None, "v4"), (4096, 16, 2, 40, None, "v4"), (10240, 1, 6, 150, None, "v4"), (768, 32, 3, 60, None, "v4"),
          (256, 96, 3, 60, None, "v5"), (4096, 16, 2, 40, None, "v5"), (64, 192, 2, 60, None, "v5"),
          (256, 288, 2, 60, None, "v5"), (4096, 48, 2, 30, None, "v5"),                                                    # round 6: the bench shapes of the split-fp16 v5 kernels
          (256, 96, 3, 60, {"window": 960}, "v4"), (4096, 16, 2, 40, {"window": 1472}, "v4"), (768, 32, 3, 60, {"window": 576}, "v4")]      # round 6: windows between the built geometries
if __name__ == "__main__":
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    for a in SHAPES:
        soak(*a)
