#!/usr/bin/env python3
"""Throughput of the Silero v5-shapes path (seeded weights): S streams x C 512-sample windows per call, inputs resident in HBM, K timed calls.
Not a BASELINE config (the reference ships no v5 weights); printed as one JSON line for profiles/."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--chunks", type=int, default=288)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT", help="engine option (vadc_amd_set_option), e.g. --opt lstm_cus=128")
    a = ap.parse_args()
    import torch
    from vadc_amd.engine import Engine
    from vadc_amd.staging import to_device, to_host
    from vadc_amd import synth
    blob = open(os.path.join(ROOT, "tests", "golden", "silero_v5_seeded.testtensor"), "rb").read()
    eng = Engine(blob, max_streams=a.streams, max_chunks_per_call=a.chunks, device=0)
    eng.set_option("defer_join", 1)                 # calls overlap: encoder of call k+1 beside the recurrence of call k
    for kv in a.opt:
        eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    base = synth.make_streams(16, -(-a.chunks * 512 // 1536), seed0=11)[:, :a.chunks * 512]
    pcm = np.ascontiguousarray(np.tile(base, (-(-a.streams // 16), 1))[:a.streams])
    d_in = to_device(pcm)
    d_out = torch.empty(a.streams, a.chunks, 2, device="cuda")
    st = torch.cuda.Stream()                        # not the null stream: that one synchronises with the engine's CU-masked (blocking) streams
    for _ in range(a.warmup):
        eng.run_device(d_in.data_ptr(), np.int16, a.streams, a.chunks, d_out.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng.run_device(d_in.data_ptr(), np.int16, a.streams, a.chunks, d_out.data_ptr(), st.cuda_stream)
    eng.join(st.cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    audio_s = a.streams * a.chunks * 512 / 16000.0 * a.steps
    print(json.dumps({"metric": "audio-seconds/sec per GPU, Silero v5 shapes (seeded weights; not a BASELINE config)", "value": round(audio_s / dt, 1),
                      "streams": a.streams, "chunks_per_step": a.chunks, "window": 512, "steps": a.steps, "ms_per_step": round(dt / a.steps * 1e3, 4),
                      "chunks_per_s": round(a.streams * a.chunks * a.steps / dt, 1), "options": a.opt, "lstm_cus": eng.get_option("lstm_cus")}))
    eng.close()


if __name__ == "__main__":
    main()
