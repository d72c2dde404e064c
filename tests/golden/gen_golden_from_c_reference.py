#!/usr/bin/env python3
"""Generate tests/golden/c_reference_v31.npz from the reference's C backend compiled in place
(oracle/_ref/libvadc_ref.so, recipe oracle/build_ref.sh).  Build container only.

    make -C oracle ref && python tests/golden/gen_golden_from_c_reference.py

Inputs are the PCM streams already stored in python_reference_v31.npz (so both goldens describe the same
audio).  Stored: per-chunk [prob0, prob1] as produced by silero_run_one_batch_with_context at batch=1,
final LSTM state, the STFT magnitude stage of two chunks (bit patterns), and the same stream re-run at
batch=96 (the vadc default) to pin "batch only changes encoder grouping" (SURVEY.md Appendix D).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402


def main():
    weights = os.path.join(HERE, "reference_fixtures", "silero_v31_16k.testtensor")
    src = np.load(os.path.join(HERE, "python_reference_v31.npz"))
    ref = O.Reference(weights)
    out = {}
    for key in src.files:
        if not key.startswith("pcm_"):
            continue
        name = key[4:]
        pcm = src[key]
        x = pcm.astype(np.float32) / np.float32(32768)
        ref.reset()
        p1 = ref.run(x, batch=1)
        h, c = ref.state()
        ref.reset()
        p96 = ref.run(x, batch=min(96, x.size // 1536))
        assert np.array_equal(p1.view(np.uint32), p96.view(np.uint32)), "batch must not change results"
        out[f"probs_{name}"] = p1
        out[f"h_{name}"] = h
        out[f"c_{name}"] = c
        print(name, "prob range", p1[:, 1].min(), p1[:, 1].max())
    x = src["pcm_speech0"].astype(np.float32) / np.float32(32768)
    out["stft_mag_speech0_chunks_0_20"] = ref.stft(np.concatenate([x[0:1536], x[20 * 1536:21 * 1536]]))
    path = os.path.join(HERE, "c_reference_v31.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
