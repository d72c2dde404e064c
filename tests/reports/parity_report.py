#!/usr/bin/env python3
"""Parity statistics of the HIP engine against the CPU oracle on long synthetic speech streams (GPU box).
Writes profiles/<round>/parity_report.json.   python tests/reports/parity_report.py [out.json]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O            # noqa: E402  (checker)
from vadc_amd import synth                # noqa: E402
from vadc_amd.engine import Engine        # noqa: E402


def compare(eng, orc, pcm, step, one_output=False):
    S, n = pcm.shape[0], pcm.shape[1] // 1536
    eng.reset_streams()
    got = np.concatenate([eng.run(pcm[:, i * 1536:(i + step) * 1536]) for i in range(0, n, step)], axis=1)[:, :, 1]
    want = orc.forward_streams(pcm)
    d = np.abs(got.astype(np.float64) - want).ravel()
    seg_equal = all(np.array_equal(O.segments(got[s])[1], O.segments(want[s])[1]) for s in range(S))
    return {"streams": S, "chunks_per_stream": n, "max_abs_dp": float(d.max()), "p999_abs_dp": float(np.quantile(d, 0.999)),
            "mean_abs_dp": float(d.mean()), "prob_range": [float(want.min()), float(want.max())],
            "chunks_within_1e-3_of_threshold_0.5": int((np.abs(want - 0.5) < 1e-3).sum()),
            "segment_chunk_indices_identical": bool(seg_equal), "tolerance": 1e-4}


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_report.json")
    blob = open(os.path.join(ROOT, "tests", "golden", "reference_fixtures", "silero_v31_16k.testtensor"), "rb").read()
    orc = O.Oracle(blob)
    eng = Engine(blob, max_streams=64, max_chunks_per_call=100, device=0)
    rep = compare(eng, orc, synth.make_streams(8, 1000, seed0=9000), 100)               # 8 long streams (96 s each)
    # a wide sweep: 64 streams x 400 chunks, different seeds, plus the control signals of SURVEY.md section 8(d)
    wide = synth.make_streams(61, 400, seed0=31000)
    ctrl = np.stack([synth.control_stream(k, 400 * 1536, seed=5) for k in ("zeros", "noise", "square")])
    rep["wide_sweep"] = compare(eng, orc, np.concatenate([wide, ctrl]), 100)
    eng.close()
    # BASELINE config 3 (precision SPLIT16: exact STFT + split-fp16 GEMMs) and the throughput mode (FAST_STFT: GEMM STFT) over the same 33,600 chunks
    for name, mode in (("split16", 1), ("fast_stft", 2)):
        e = Engine(blob, max_streams=64, max_chunks_per_call=100, device=0, precision=mode)
        rep[name] = compare(e, orc, np.concatenate([wide, ctrl]), 100)
        rep[name]["rounded_long_streams"] = compare(e, orc, synth.make_streams(8, 1000, seed0=9000), 100)
        if mode == 2:
            rep[name]["tolerance"] = 1e-3
        e.close()
    # Silero v4 against its own restatement (PyTorch-pinned)
    v4 = os.path.join(ROOT, "tests", "golden", "silero_v4_16k.testtensor")
    if os.path.exists(v4):
        blob4 = open(v4, "rb").read()
        e4 = Engine(blob4, max_streams=16, max_chunks_per_call=100, device=0)
        rep["silero_v4"] = compare(e4, O.OracleV4(blob4), synth.make_streams(16, 400, seed0=52000), 100)
        e4.close()
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(rep, open(out_path, "w"), indent=1)
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
