"""The first encoder stage alone on the chip: python tools/l1_rate.py [chunks=24576] [reps=10] [variants=0,1]
Feeds normalized log-magnitudes through vadc_amd_debug_stage_from_stage (normalized -> layer1) with per-kernel HIP events on; prints ms per launch
for option "layer1" = 0 (k_layer1_regs) and 1 (k_layer_mfma's K = 1 form).  The input arrives by a host copy, so
it sits in whatever cache a 317 MB DMA write leaves it in; inside the step (bench.py --details) the front end has just written it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vadc_amd.engine import Engine
blob = open("tests/golden/reference_fixtures/silero_v31_16k.testtensor", "rb").read()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
variants = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,1").split(",")]
e = Engine(blob, max_streams=256, max_chunks_per_call=(n + 255) // 256, device=0)
rng = np.random.default_rng(1)
x = (rng.standard_normal((n, 129, 25)) * 2.0).astype(np.float32)
for v in variants:
    e.set_option("layer1", v)
    e.stage_from_stage(x, "normalized", "layer1")
    e.reset_kernel_times()
    e.set_profiling(True)
    for _ in range(reps):
        e.stage_from_stage(x, "normalized", "layer1")
    e.set_profiling(False)
    kt = e.kernel_times()
    print(f"layer1={v}: " + "  ".join(f"{k} {ms / c:.4f} ms" for k, (c, ms) in kt.items() if c) + f"   per {n} chunks")
e.close()
