"""The LSTM kernels alone: python tools/lstm_rate.py [streams=256] [chunks=96] [reps=10]
vadc_amd_debug_lstm_decoder with per-kernel HIP events; prints ms per launch and microseconds per recurrence slot (7 per chunk)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vadc_amd.engine import Engine
blob = open("tests/golden/reference_fixtures/silero_v31_16k.testtensor", "rb").read()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
C = int(sys.argv[2]) if len(sys.argv) > 2 else 96
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
e = Engine(blob, max_streams=S, max_chunks_per_call=C, device=0)
rng = np.random.default_rng(1)
x = np.abs(rng.standard_normal((S, C, 64, 7)).astype(np.float32))
for lk, trail in ((7, 0), (6, 0)):
    e.set_option("lstm", lk); e.set_option("lstm_trail", trail)
    e.lstm_decoder(x)
    e.reset_kernel_times(); e.set_profiling(True)
    for _ in range(reps):
        e.lstm_decoder(x)
    e.set_profiling(False)
    kt = e.kernel_times()
    print(f"lstm={lk} trail={trail}: " + "  ".join(f"{k} {ms / c:.4f} ms ({ms / c / (7 * C) * 1e3:.3f} us/slot)" for k, (c, ms) in kt.items() if c))
e.close()
