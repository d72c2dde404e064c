#!/usr/bin/env python3
"""Single-stream latency of the drop-in path: one `backend_run`-shaped call (1 stream x n chunks, host buffers in, host
probabilities out, synchronous) as the reference's process_chunks issues it (vadc.c:56-103; window = 96 chunks).
Prints ms per call and the real-time factor; eager launches vs hipGraph replay on device-resident buffers."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vadc_amd import synth
from vadc_amd.engine import Engine

W = os.path.join(ROOT, "tests", "golden", "reference_fixtures", "silero_v31_16k.testtensor")
blob = open(W, "rb").read()
for n in (1, 8, 96):
    eng = Engine(blob, max_streams=1, max_chunks_per_call=n, device=0)
    pcm = synth.speech_like(n * 1536, seed=3).reshape(1, -1)
    for _ in range(5): eng.run(pcm)
    t0 = time.perf_counter(); reps = 50
    for _ in range(reps): eng.run(pcm)
    host_ms = (time.perf_counter() - t0) / reps * 1e3
    d_in = torch.from_numpy(pcm).to("cuda:0"); d_out = torch.empty((1, n, 2), dtype=torch.float32, device="cuda:0")
    st = torch.cuda.Stream()
    res = {}
    for graph in (0, 1):
        eng.set_option("graph", graph)
        for _ in range(5): eng.run_device(d_in.data_ptr(), np.int16, 1, n, d_out.data_ptr(), st.cuda_stream)
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.run_device(d_in.data_ptr(), np.int16, 1, n, d_out.data_ptr(), st.cuda_stream)
            st.synchronize()
        res[graph] = (time.perf_counter() - t0) / reps * 1e3
    print(f"1 stream x {n:3d} chunks: host-buffer call {host_ms:7.3f} ms ({n * 0.096 / host_ms * 1e3:8.1f}x real time) | "
          f"device buffers eager {res[0]:6.3f} ms, hipGraph {res[1]:6.3f} ms")
    eng.close()
