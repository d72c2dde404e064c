"""(the harness that first showed DESIGN.md 4.1 (d): with the engine of commit b91f1ec, whose k_frontend_ri -- option "fe_opt" = 11 -- had the swapped pair as the SECOND source of its
packed additions, 37 - 50 of 6,000 runs differ; at HEAD the same option is the fixed form and no run does)
engine A runs stage taps alone (front end; front end + first layers) again and again while engine B, from another thread, keeps the device busy with whole steps
(10,240 x 1, layer-major LSTM): does a stage of A ever change its bits?"""
import sys, os, threading
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vadc_amd import synth
from vadc_amd.engine import Engine
blob = open("tests/golden/reference_fixtures/silero_v31_16k.testtensor", "rb").read()
bits = lambda a: np.ascontiguousarray(a).view(np.uint32)
S = 10240
base = synth.make_streams(48, 8, seed0=10240)
pcm = np.ascontiguousarray(base[np.arange(S) % 48])
d_in = [torch.from_numpy(np.ascontiguousarray(pcm[:, k * 1536:(k + 1) * 1536])).cuda() for k in range(8)]
B = Engine(blob, max_streams=S, max_chunks_per_call=1, device=0)
B.set_option("defer_join", 1)
lstmB = int(sys.argv[1]) if len(sys.argv) > 1 else 7
B.set_option("lstm", lstmB)
stB = torch.cuda.Stream()
stop = False
def busy():
    d_out = torch.empty((S, 1, 2), dtype=torch.float32, device="cuda:0")
    i = 0
    while not stop:
        for k in range(8):
            B.run_device(d_in[k].data_ptr(), np.int16, S, 1, d_out.data_ptr(), stB.cuda_stream)
        B.join(stB.cuda_stream); stB.synchronize(); i += 1
x = (pcm[:2048, :1536].astype(np.float32) / 32768.0).reshape(-1)
A = Engine(blob, max_streams=2048, max_chunks_per_call=1, device=0)
for fe in (int(os.environ.get("FE", "11")),):
    A.set_option("fe_opt", fe)
    for stage in (os.environ.get("STAGE", "normalized"),):
        try:
            first = A.stage_from_samples(x, stage)
        except Exception as ex:
            print("stage", stage, ex); continue
        th = threading.Thread(target=busy); stop = False; th.start()
        bad = 0; info = []
        for rep in range(int(os.environ.get("REPS", "1500"))):
            r = A.stage_from_samples(x, stage)
            d = bits(first) != bits(r)
            if d.any():
                bad += 1; info.append((rep, sorted(set(np.nonzero(d.reshape(d.shape[0], -1))[0].tolist()))[:6]))
                for c in sorted(set(np.nonzero(d.reshape(d.shape[0], -1))[0].tolist()))[:(2 if bad <= 6 else 0)]:
                    dd = d[c]; diff = (r[c].astype(np.float64) - first[c])
                    nz = np.nonzero(dd)
                    big = np.nonzero(np.abs(diff - np.median(diff)) > 1e-3)
                    print("  BIG rep", rep, "chunk", c, "cells", len(big[0]), "bins", sorted(set(big[0].tolist())), "frames", sorted(set(big[1].tolist())), flush=True)
                    med = float(np.median(diff))
                    for bb_, ff_ in list(zip(big[0].tolist(), big[1].tolist()))[:3]:
                        got_v = r[c][bb_, ff_] - med
                        col = first[c][:, ff_]
                        near = np.nonzero(np.abs(col - got_v) < 2e-6)[0].tolist()
                        # the same lane's value in neighbouring chunks / frames?
                        print("     bin", bb_, "frame", ff_, "want", float(first[c][bb_, ff_]), "got", float(got_v), "bins of this frame with that value:", near,
                              "raw Y got (approx, + mean):", float(got_v), flush=True)
                    continue
                    print("  rep", rep, "chunk", c, "differing elements", int(dd.sum()), "bins", sorted(set(nz[0].tolist()))[:12], "frames", sorted(set(nz[1].tolist())), "diff min/max", diff.min(), diff.max(), "median diff", float(np.median(diff)), flush=True)
        stop = True; th.join()
        print("B lstm", lstmB, "A fe_opt", fe, "stage", stage, "runs that differ:", bad, "of", os.environ.get("REPS", "1500"), info[:4], flush=True)
A.close(); B.close()
