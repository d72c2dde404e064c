"""Golden vectors of vadc's segmenter FROM THE REFERENCE'S OWN CODE: oracle/_ref/ref_segmenter is feed_probability / emit_speech_segment /
combine_or_emit_speech_segment as they stand in /root/reference/vadc.c:165-299, compiled in place by oracle/build_ref.sh (the loop that feeds them and the
end-of-stream flush restate vadc.c:964-987, 1005-1027; print_speech_stats, stderr only, is empty: oracle/ref_segmenter_harness.c).  Each case = a probability
sequence + the CLI's options; the expectation is the text vadc prints.  Build container only (the reference does not travel); the vectors do:
    python tests/golden/gen_golden_segments_from_c_reference.py   ->   tests/golden/c_reference_segments.npz"""
import json, os, struct, subprocess, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_segmenter")


def chunks_of(ms, input_count):                      # vadc.c:756-768, in float32 like the reference
    chunk_ms = np.float32(input_count) / np.float32(16000) * np.float32(1000.0)
    n = int(np.float32(ms) / chunk_ms + np.float32(0.5))
    return max(n, 1)


def run_reference(probs, o):
    thr = np.float32(o["threshold"])
    neg = np.float32(thr - np.float32(o["neg_threshold_relative"]))                      # vadc.c:1243
    head = struct.pack("<iffiifii", len(probs), thr, neg, chunks_of(o["min_silence"], o["sequence_count"]), chunks_of(o["min_speech"], o["sequence_count"]),
                       np.float32(o["speech_pad"]), 1 if o["centiseconds"] else 0, o["sequence_count"])
    r = subprocess.run([EXE], input=head + np.asarray(probs, np.float32).tobytes(), capture_output=True, check=True)
    return r.stdout.decode()


def sequences():
    rng = np.random.default_rng(20260504)
    out = {}
    g = np.load(os.path.join(HERE, "c_reference_v31.npz"))
    for k in g.files:                                   # what the reference's C backend gives on this repo's test streams
        if k.startswith("probs_"):
            out["model_" + k[6:]] = g[k][:, 1]             # [chunks, 2]: the speech probability is index 1 (vadc.c:704-708)
    out["empty"] = np.zeros(0, np.float32)
    out["one_low"] = np.array([0.1], np.float32)
    out["one_high"] = np.array([0.9], np.float32)
    out["all_high_40"] = np.full(40, 0.95, np.float32)                                   # speech to the end of the stream: the final flush alone
    out["all_low_40"] = np.full(40, 0.02, np.float32)
    out["ends_in_speech"] = np.concatenate([np.full(10, 0.02), np.full(30, 0.9)]).astype(np.float32)
    out["ends_in_short_speech"] = np.concatenate([np.full(30, 0.02), np.full(2, 0.9)]).astype(np.float32)
    out["alternating"] = np.tile(np.array([0.9, 0.1], np.float32), 60)
    out["on_the_thresholds"] = np.tile(np.array([0.5, 0.35, 0.349999, 0.5, 0.499999, 0.35], np.float32), 30)
    out["bursts"] = np.concatenate([np.concatenate([np.full(int(a), 0.9), np.full(int(b), 0.05)]) for a, b in rng.integers(1, 12, (40, 2))]).astype(np.float32)
    for i in range(6):                                  # random walks through both thresholds, 400 chunks
        w = np.cumsum(rng.normal(0, 0.12, 400)) + 0.4
        out[f"walk{i}"] = (1 / (1 + np.exp(-(w - w.mean()) * 2.5))).astype(np.float32)
    for i in range(4):                                  # telegraph noise with dwell times around the min_silence / min_speech counts
        d = rng.integers(1, 9, 120)
        out[f"telegraph{i}"] = np.concatenate([np.full(n, 0.85 if j % 2 else 0.08) for j, n in enumerate(d)]).astype(np.float32) + rng.normal(0, 0.02, int(d.sum())).astype(np.float32)
    return {k: np.clip(np.asarray(v, np.float32), 0, 1) for k, v in out.items()}


OPTIONS = [
    dict(threshold=0.5, neg_threshold_relative=0.15, min_silence=200.0, min_speech=250.0, speech_pad=30.0, centiseconds=False, sequence_count=1536),      # vadc.c:1110-1124
    dict(threshold=0.5, neg_threshold_relative=0.15, min_silence=200.0, min_speech=250.0, speech_pad=30.0, centiseconds=True, sequence_count=1536),
    dict(threshold=0.3, neg_threshold_relative=0.05, min_silence=100.0, min_speech=100.0, speech_pad=0.0, centiseconds=False, sequence_count=1536),
    dict(threshold=0.8, neg_threshold_relative=0.3, min_silence=500.0, min_speech=1000.0, speech_pad=100.0, centiseconds=False, sequence_count=1536),
    dict(threshold=0.5, neg_threshold_relative=0.15, min_silence=200.0, min_speech=250.0, speech_pad=500.0, centiseconds=True, sequence_count=512),       # pads that merge neighbours
    dict(threshold=0.5, neg_threshold_relative=0.15, min_silence=10.0, min_speech=10.0, speech_pad=30.0, centiseconds=False, sequence_count=768),         # both counts clamp to 1
    dict(threshold=0.45, neg_threshold_relative=0.2, min_silence=300.0, min_speech=400.0, speech_pad=45.0, centiseconds=False, sequence_count=1280),
]


def main():
    if not os.path.exists(EXE):
        sys.exit("oracle/_ref/ref_segmenter is not built (oracle/build_ref.sh needs /root/reference)")
    seqs = sequences()
    cases, arrays = [], {}
    for name, p in seqs.items():
        arrays["probs_" + name] = p
        for oi, o in enumerate(OPTIONS):
            cases.append({"sequence": name, "options": o, "stdout": run_reference(p, o)})
    arrays["cases"] = np.frombuffer(json.dumps(cases).encode(), np.uint8)
    path = os.path.join(HERE, "c_reference_segments.npz")
    np.savez_compressed(path, **arrays)
    n_seg = sum(c["stdout"].count("\n") for c in cases)
    print(f"wrote {path}: {len(seqs)} sequences x {len(OPTIONS)} option sets = {len(cases)} cases, {n_seg} segments, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
