"""SPLIT16 precision mode against the FP32 (parity) mode of the same engine: where the folded split-fp16 GEMM front end moves
the log-magnitudes and by how much the probabilities follow.  Product code only (both sides are the HIP engine)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vadc_amd import synth                      # noqa: E402
from vadc_amd.engine import Engine              # noqa: E402


def main():
    S, Cn = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 16)
    seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 564
    blob = open(os.path.join(ROOT, "tests", "golden", "reference_fixtures", "silero_v31_16k.testtensor"), "rb").read()
    pcm = synth.make_streams(S, Cn, seed0=seed0)
    e32 = Engine(blob, max_streams=S, max_chunks_per_call=Cn, device=0)
    e16 = Engine(blob, max_streams=S, max_chunks_per_call=Cn, device=0, precision=2)
    p32 = e32.run(pcm)[:, :, 1].astype(np.float64)
    p16 = e16.run(pcm)[:, :, 1].astype(np.float64)
    d = np.abs(p32 - p16)
    s, c = np.unravel_index(np.argmax(d), d.shape)
    out = {"streams": S, "chunks": Cn, "max_dp": float(d.max()), "argmax": [int(s), int(c)], "p32_at_max": float(p32[s, c]),
           "p99_dp": float(np.quantile(d, 0.99)), "p999_dp": float(np.quantile(d, 0.999)), "mean_dp": float(d.mean()),
           "per_stream_max_top5": sorted([float(x) for x in d.max(axis=1)], reverse=True)[:5],
           "per_chunk_index_max": [float(x) for x in d.max(axis=0)]}
    # log-magnitudes of the worst stream, chunk by chunk: mean-removed values [129][25] per chunk
    x = (pcm[s].astype(np.float32) / np.float32(32768))
    a = e32.stage_from_samples(x, "normalized").astype(np.float64)
    b = e16.stage_from_samples(x, "normalized").astype(np.float64)
    dy = np.abs(a - b)
    out["dY_max"] = float(dy.max())
    out["dY_per_chunk_max"] = [float(v) for v in dy.reshape(Cn, -1).max(axis=1)]
    out["dY_per_frame_max"] = [float(v) for v in dy.max(axis=(0, 1))]
    out["dY_bins_above_0.01"] = int((dy.max(axis=(0, 2)) > 0.01).sum())
    out["dY_mean"] = float(dy.mean())
    big = dy > 0.05
    out["count_dY_above_0.05"] = int(big.sum())
    out["mean_logmag_where_big"] = float(a[big].mean()) if big.any() else None
    out["mean_logmag_all"] = float(a.mean())
    print(json.dumps(out))
    e32.close(); e16.close()


if __name__ == "__main__":
    main()
