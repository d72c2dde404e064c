#!/usr/bin/env python3
"""How long the device takes to reach its sustained clock after an idle gap: per-step completion times of the 256 x 96 step (graph replay, deferred join) for the
first steps behind idle gaps of different lengths.  Run on the GPU box: python tools/clock_ramp.py > gpurun_out/clock_ramp.log"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vadc_amd import synth
from vadc_amd.engine import Engine

S, Cn, NB = 256, 96, 3
blob = open(os.path.join(ROOT, "tests", "golden", "reference_fixtures", "silero_v31_16k.testtensor"), "rb").read()
eng = Engine(blob, max_streams=S, max_chunks_per_call=Cn, device=0)
eng.set_option("defer_join", 1)
base = synth.make_streams(16, NB * Cn, seed0=1234)
pcm = np.ascontiguousarray(np.tile(base, (S // 16, 1)))
d_in = [torch.from_numpy(np.ascontiguousarray(pcm[:, i * Cn * 1536:(i + 1) * Cn * 1536])).to("cuda:0") for i in range(NB)]
d_probs = [torch.empty((S, Cn, 2), dtype=torch.float32, device="cuda:0") for _ in range(NB)]
st = torch.cuda.Stream()
side = torch.cuda.Stream()
def step(i):
    eng.run_device(d_in[i % NB].data_ptr(), np.int16, S, Cn, d_probs[i % NB].data_ptr(), st.cuda_stream)
for i in range(2 * NB): step(i)
torch.cuda.synchronize()
eng.set_option("graph", 1)
for i in range(2 * NB): step(i)
torch.cuda.synchronize()
N = 120
for gap_ms in (0, 1, 5, 20, 100, 1000, 3000):
    # busy first: 600 steps, then the gap, then N steps with an event behind each
    for i in range(600): step(i)
    torch.cuda.synchronize()
    time.sleep(gap_ms / 1e3)
    evs = []
    e0 = torch.cuda.Event(enable_timing=True); e0.record(side)
    for i in range(N):
        step(i)
        eng.join(side.cuda_stream)
        e = torch.cuda.Event(enable_timing=True); e.record(side); evs.append(e)
    torch.cuda.synchronize()
    t = [e0.elapsed_time(e) for e in evs]
    d = np.diff([0.0] + t)
    blocks = [round(float(np.mean(d[a:a + 10])), 4) for a in range(0, N, 10)]
    print(json.dumps({"gap_ms": gap_ms, "first20_ms": round(t[19], 3), "ms_per_step_by_block_of_10": blocks}), flush=True)
eng.close()
