"""Engines created one after the other in ONE process, each closed before the next: the rate of every one of them at the same shape (Silero v4, 4096 x 16 by default).
   python tools/engine_order_probe.py [engines=8] [model=v4|v31] [streams] [chunks]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    model = sys.argv[2] if len(sys.argv) > 2 else "v4"
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    Cn = int(sys.argv[4]) if len(sys.argv) > 4 else 16
    import torch
    from vadc_amd.engine import Engine
    from vadc_amd import synth
    from vadc_amd.staging import to_device
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    name = "silero_v4_16k.testtensor" if model == "v4" else os.path.join("reference_fixtures", "silero_v31_16k.testtensor")
    blob = open(os.path.join(root, "tests", "golden", name), "rb").read()
    base = synth.make_streams(16, Cn, seed0=11)
    d_in = to_device(np.ascontiguousarray(np.tile(base, (-(-S // 16), 1))[:S]))
    d_out = torch.empty(S, Cn, 2, device="cuda")
    st = torch.cuda.Stream()
    rows = []
    for i in range(n):
        eng = Engine(blob, max_streams=S, max_chunks_per_call=Cn, device=0)
        if os.environ.get("NEW_STREAM"):
            st = torch.cuda.Stream()                              # NEW_STREAM=1: a caller's stream of its own per engine (what the measurement tools did)
        eng.set_option("defer_join", 1)
        for _ in range(3):
            eng.run_device(d_in.data_ptr(), np.int16, S, Cn, d_out.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        eng.set_option("graph", 1)
        for _ in range(10):
            eng.run_device(d_in.data_ptr(), np.int16, S, Cn, d_out.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(60):
            eng.run_device(d_in.data_ptr(), np.int16, S, Cn, d_out.data_ptr(), st.cuda_stream)
        eng.join(st.cuda_stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rows.append(round(S * Cn * 60 * 0.096 / dt / 1e6, 2))
        eng.close()
    print(json.dumps({"model": model, "streams": S, "chunks": Cn, "M_audio_s_per_s_by_engine": rows}))


if __name__ == "__main__":
    main()
