"""Bit comparison of two builds of the library on what k_frontend_gemm2 feeds (s16 input; the stage taps take f32 and run the first form): Silero v4's probabilities
for 61 streams x 13 chunks at the default window and the two shorter ones.
    VADC_AMD_LIB=<other build> python tools/fe_gemm_bits.py dump /tmp/a.npz ;  python tools/fe_gemm_bits.py dump /tmp/b.npz ;  python tools/fe_gemm_bits.py cmp /tmp/a.npz /tmp/b.npz"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "dump":
    from vadc_amd import synth
    from vadc_amd.engine import Engine
    blob = open(os.path.join(ROOT, "tests", "golden", "silero_v4_16k.testtensor"), "rb").read()
    out = {}
    for window in (1536, 1024, 512):
        e = Engine(blob, max_streams=64, max_chunks_per_call=16, device=0)
        if window != 1536: e.set_window(window)
        pcm = synth.make_streams(16, 96, seed0=5).reshape(-1)[: 61 * 13 * window].reshape(61, -1)      # 61 streams x 13 chunks: ragged tiles and groups
        out[f"probs_{window}"] = e.run(pcm)                     # s16 input: k_frontend_gemm2
        e.close()
    np.savez(sys.argv[2], **out)
    print("wrote", sys.argv[2], {k: v.shape for k, v in out.items()})
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        same = np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32))
        print(k, "bit-identical" if same else f"DIFFERENT: max |d| {np.abs(a[k] - b[k]).max():.3e}, {int((a[k].view(np.uint32) != b[k].view(np.uint32)).sum())} words")
