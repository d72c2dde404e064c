"""When each step of a short timed run finishes: python tools/step_timeline.py [steps=20] [warmup=5] [graph=1]
The bench's loop (one issuing stream, deferred joins, hipGraph replay) with an event behind every step's join; prints the completion time of every
step relative to the start of the timed region and the gaps between them -- where a 20-step run loses against the steady state."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vadc_amd import synth
from vadc_amd.engine import Engine
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5
graph = int(sys.argv[3]) if len(sys.argv) > 3 else 1
blob = open("tests/golden/reference_fixtures/silero_v31_16k.testtensor", "rb").read()
S, Cn, NB = 256, 96, 3
eng = Engine(blob, max_streams=S, max_chunks_per_call=Cn, device=0)
eng.set_option("groups", 1)
for kv in sys.argv[4:]:
    k, v = kv.split("="); eng.set_option(k, int(v))
base = synth.make_streams(16, NB * Cn, seed0=1234)
pcm = np.ascontiguousarray(np.tile(base, (S // 16, 1)))
d_in = [torch.from_numpy(np.ascontiguousarray(pcm[:, i * Cn * 1536:(i + 1) * Cn * 1536])).cuda() for i in range(NB)]
d_probs = [torch.empty((S, Cn, 2), dtype=torch.float32, device="cuda") for _ in range(NB)]
main, side = torch.cuda.Stream(), torch.cuda.Stream()
eng.set_option("defer_join", 1)
def step(i):
    b = i % NB
    eng.run_device(d_in[b].data_ptr(), np.int16, S, Cn, d_probs[b].data_ptr(), main.cuda_stream)
for i in range(2 * NB): step(i)
torch.cuda.synchronize()
for i in range(W): step(i)
torch.cuda.synchronize()
if graph:
    eng.set_option("graph", 1)
    for i in range(2 * NB): step(i)
    torch.cuda.synchronize()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
evs[0].record(main)
t0 = time.perf_counter()
for i in range(K):
    step(i)
    eng.join(side.cuda_stream)
    evs[i + 1].record(side)
issued = time.perf_counter() - t0
torch.cuda.synchronize()
el = time.perf_counter() - t0
ts = [evs[0].elapsed_time(e) for e in evs[1:]]
print(f"{K} steps: {el * 1e3:.3f} ms wall ({el / K * 1e3:.4f} per step), issued in {issued * 1e3:.3f} ms")
print("done at (ms):", " ".join(f"{t:.2f}" for t in ts))
print("gaps   (ms):", " ".join(f"{b - a:.3f}" for a, b in zip([0.0] + ts[:-1], ts)))
eng.close()
