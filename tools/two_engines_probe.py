"""Probe: do two engines running side by side on one GPU (kernels of different calls free to overlap) beat one engine with the same total stream count?
An upper bound on what a second front-end + encoder stream inside the engine could win."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from vadc_amd.engine import Engine
from vadc_amd import synth
blob = open("tests/golden/reference_fixtures/silero_v31_16k.testtensor", "rb").read()
def make(S, C):
    e = Engine(blob, max_streams=S, max_chunks_per_call=C, device=0)
    e.set_option("defer_join", 1)
    e.set_option("groups", 1)
    base = synth.make_streams(16, C, seed0=3)
    pcm = np.ascontiguousarray(np.tile(base, (S // 16, 1)))
    return e, torch.from_numpy(pcm).cuda(), torch.empty(S, C, 2, device="cuda")
def run(engs, steps):
    sts = [torch.cuda.Stream() for _ in engs]
    for _ in range(10):
        for (e, i, o), st in zip(engs, sts): e.run_device(i.data_ptr(), np.int16, i.shape[0], o.shape[1], o.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for (e, i, o), st in zip(engs, sts): e.run_device(i.data_ptr(), np.int16, i.shape[0], o.shape[1], o.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tot = sum(i.shape[0] * o.shape[1] for e, i, o in engs) * steps * 0.096
    return tot / dt
C = 96
one = [make(512, C)]
print("one engine 512 streams:", round(run(one, 200)))
one[0][0].close()
two = [make(256, C), make(256, C)]
print("two engines 256 streams each:", round(run(two, 200)))
for e, _, _ in two: e.close()
one = [make(256, C)]
print("one engine 256 streams:", round(run(one, 300)))
