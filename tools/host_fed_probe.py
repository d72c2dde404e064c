"""PCIe-inclusive rate of vadc_amd_run_s16_async by number of H2D copy streams (option "h2d_streams") and kind of host memory (numpy buffers the engine
page-locks with hipHostRegister / torch pinned tensors), one fresh engine per point.  python tools/host_fed_probe.py
The FIRST engine of a process is the representative one (later engines of the same process have been seen 1.7x faster: stream -> hardware queue luck)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vadc_amd import synth
from vadc_amd.engine import Engine
blob = open("tests/golden/reference_fixtures/silero_v31_16k.testtensor", "rb").read()
S, Cn, NB = 256, 96, 3
base = synth.make_streams(16, Cn, seed0=5)
pcm = np.ascontiguousarray(np.tile(base, (S // 16, 1)))
for kind in ("registered numpy", "torch pinned"):
    if kind == "torch pinned":
        hbuf = [torch.from_numpy(pcm).pin_memory() for _ in range(NB)]
        host = [h.numpy() for h in hbuf]
        obuf = [torch.empty((S, Cn, 2), dtype=torch.float32).pin_memory() for _ in range(NB)]
        outs = [o.numpy() for o in obuf]
    else:
        host = [pcm.copy() for _ in range(NB)]
        outs = [np.empty((S, Cn, 2), np.float32) for _ in range(NB)]
    for parts in (1, 2, 3, 4):
        e = Engine(blob, max_streams=S, max_chunks_per_call=Cn, device=0)
        e.set_option("h2d_streams", parts); e.set_option("graph", 1)
        for i in range(18): e.run_async(host[i % NB], outs[i % NB])
        e.wait_async()
        n = 48
        t = time.perf_counter()
        for i in range(n): e.run_async(host[i % NB], outs[i % NB])
        e.wait_async()
        dt = time.perf_counter() - t
        print(f"{kind:18s} h2d_streams={parts}: {S * Cn * n * 0.096 / dt / 1e6:.3f} M audio-s/s, {dt / n * 1e3:.3f} ms per call, {S * Cn * 3072 * n / dt / 1e9:.1f} GB/s")
        e.close()
