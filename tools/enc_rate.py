"""k_enc_fused alone on the chip: python tools/enc_rate.py [chunks=24576] [reps=20]
Feeds a layer-1 output through vadc_amd_debug_stage_from_stage (layer1 -> layer4) with per-kernel HIP events on; prints ms per launch.
(The stage API adds host copies around the launch; the events bracket the kernel only.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vadc_amd.engine import Engine
blob = open("tests/golden/reference_fixtures/silero_v31_16k.testtensor", "rb").read()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
e = Engine(blob, max_streams=256, max_chunks_per_call=(n + 255) // 256, device=0)
rng = np.random.default_rng(1)
x = np.abs(rng.standard_normal((n, 16, 13)).astype(np.float32))
for enc in (0, 5):
    e.set_option("encoder", enc)
    e.stage_from_stage(x, "layer1", "layer4")
    e.reset_kernel_times()
    e.set_profiling(True)
    for _ in range(reps):
        e.stage_from_stage(x, "layer1", "layer4")
    e.set_profiling(False)
    kt = e.kernel_times()
    tot = sum(ms / max(c, 1) for k, (c, ms) in kt.items() if c)
    print(f"encoder={enc}: " + "  ".join(f"{k} {ms / c:.4f} ms" for k, (c, ms) in kt.items() if c) + f"   sum {tot:.4f} ms per {n} chunks")
e.close()
