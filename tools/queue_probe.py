"""An engine's rate can depend on which engines the PROCESS created before it: 10,240 x 1 (no CU partition, plain prioritised streams) runs at 2.95 M audio-s/s as the first engine
and after a default 256 x 96 engine (partition 2 x 16 CUs), and at 1.77 M after a 256 x 96 engine with option lstm = 6 (one masked partition of 16 CUs) -- every time, alternating.
Every pair of the engine's streams still overlaps (probed with k_probe_overlap); what changes is WHEN the recurrence of call k starts: beside call k + 1's front end (good) or
0.2 ms later beside its persistent first-layer / encoder kernels, which then take twice as long (k_layer1 0.063 -> 0.12 ms, k_enc234 0.05 -> 0.12).  Stream -> hardware-queue
placement is the runtime's; bench.py orders its side configurations so that none is measured behind such a predecessor.   python tools/queue_probe.py"""
import os, sys
ROOT="/root/repo"; sys.path.insert(0, ROOT)
import torch, bench
b31=open(os.path.join(ROOT,"tests","golden","reference_fixtures","silero_v31_16k.testtensor"),"rb").read()
dev=torch.device("cuda",0)
def r(S,Cn,opts=None):
    x=bench.side_config(torch, b31, dev, 0, "v31", S, Cn, 0, steps=100, warmup=10, opts=opts)
    print(S,Cn,opts, round(x["value"]/1e6,3), x["ms_per_step"], x["kernels_ms"], flush=True)
FM = {"full_mask_streams": int(os.environ.get("FM", "0"))}      # the other form: 0 = plain streams (round 4), 1 = all three masked
r(10240,1); r(10240,1, FM)
for i in range(4):
    r(256,96, {"lstm":6} if i%2 else None)
    r(10240,1); r(10240,1, FM)
for S, Cn in ((4096, 16), (16384, 1), (3072, 32), (4096, 4)):
    r(S, Cn); r(S, Cn, FM); r(S, Cn); r(S, Cn, FM)
