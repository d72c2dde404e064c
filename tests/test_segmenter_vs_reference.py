"""The segmenter ("bit-exact segment timestamp indices") against the REFERENCE'S OWN CODE, not against a restatement.

tests/golden/c_reference_segments.npz holds what feed_probability / combine_or_emit_speech_segment / emit_speech_segment of /root/reference/vadc.c:165-299
-- compiled from where they lie by oracle/build_ref.sh into oracle/_ref/ref_segmenter -- print for 26 probability sequences x 7 option sets (1,165 segments:
the C backend's probabilities on this repo's streams, streams that end in speech, values sitting ON both thresholds, dwell times around the duration
counts, pads that merge neighbours, both output formats, four window sizes).  Pinned here, on the CPU:
  * the PRODUCT's segmenter -- host/vadc_hip.c, through `--probabilities_in` (its own rounding, feeding, merging, flush and printing; no engine) -- text for text;
  * the oracle's restatement (oracle/silero_oracle.c so_segments), which the GPU tests compare the CLI's segments of real audio with;
  * the goldens themselves against the live build of the reference's functions where oracle/_ref/ref_segmenter exists.
What stays a restatement on every side: the loop that feeds the probabilities and the end-of-stream flush (vadc.c:964-987, 1005-1027 sit inside run_inference
between Win32 I/O) and the ms -> chunks rounding (:756-768); print_speech_stats (stderr, no segment arithmetic) is empty in the harness."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "c_reference_segments.npz"))
CASES = json.loads(bytes(G["cases"]).decode())
IDS = [f"{c['sequence']}-{i % 7}" for i, c in enumerate(CASES)]


@pytest.fixture(scope="module")
def cli():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "host"), "vadc_hip"], capture_output=True, text=True)
    exe = os.path.join(ROOT, "host", "vadc_hip")
    if r.returncode != 0 or not os.path.exists(exe):
        pytest.skip("host/vadc_hip does not build here: " + r.stderr[-300:])
    return exe


def test_the_goldens_cover_what_they_claim():
    assert len(CASES) == 182 and sum(c["stdout"].count("\n") for c in CASES) == 1165
    assert {c["options"]["sequence_count"] for c in CASES} == {512, 768, 1280, 1536}
    assert any(c["stdout"] == "" for c in CASES) and any(c["options"]["centiseconds"] for c in CASES)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_product_cli_prints_what_the_reference_code_prints(cli, case, tmp_path):
    o = case["options"]
    f = tmp_path / "p.f32"
    G["probs_" + case["sequence"]].astype(np.float32).tofile(f)
    args = [cli, "--probabilities_in", str(f), "--threshold", repr(o["threshold"]), "--neg_threshold_relative", repr(o["neg_threshold_relative"]),
            "--min_silence", repr(o["min_silence"]), "--min_speech", repr(o["min_speech"]), "--sequence_count", str(o["sequence_count"])]
    if o["speech_pad"] > 0:
        args += ["--speech_pad", repr(o["speech_pad"])]
    else:
        pytest.skip("the CLI, like vadc.c:1215-1218, ignores a value <= 0: --speech_pad 0 cannot be asked for")
    if o["centiseconds"]:
        args.append("--output_centi_seconds")
    r = subprocess.run(args, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert r.stdout == case["stdout"]


@pytest.mark.parametrize("case", [c for c in CASES if c["options"]["sequence_count"] == 1536], ids=[i for i, c in zip(IDS, CASES) if c["options"]["sequence_count"] == 1536])
def test_oracle_restatement_gives_the_reference_codes_segments(case):
    o = case["options"]
    sec, _ = O.segments(G["probs_" + case["sequence"]], threshold=o["threshold"], neg_threshold_relative=o["neg_threshold_relative"],
                        min_silence_ms=o["min_silence"], min_speech_ms=o["min_speech"], speech_pad_ms=o["speech_pad"])
    if o["centiseconds"]:
        txt = "".join(f"{int(float(a) * 100.0 + 0.5)},{int(float(b) * 100.0 + 0.5)}\n" for a, b in sec)
    else:
        txt = "".join("%.2f,%.2f\n" % (float(a), float(b)) for a, b in sec)
    assert txt == case["stdout"]


def test_goldens_are_what_the_live_reference_build_prints():
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_segmenter")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_segmenter not built (needs /root/reference at build time)")

    def chunks_of(ms, n):
        return max(int(np.float32(ms) / (np.float32(n) / np.float32(16000) * np.float32(1000.0)) + np.float32(0.5)), 1)
    for c in CASES:
        o, p = c["options"], G["probs_" + c["sequence"]].astype(np.float32)
        thr = np.float32(o["threshold"])
        head = struct.pack("<iffiifii", p.size, thr, np.float32(thr - np.float32(o["neg_threshold_relative"])), chunks_of(o["min_silence"], o["sequence_count"]),
                           chunks_of(o["min_speech"], o["sequence_count"]), np.float32(o["speech_pad"]), 1 if o["centiseconds"] else 0, o["sequence_count"])
        r = subprocess.run([exe], input=head + p.tobytes(), capture_output=True, check=True)
        assert r.stdout.decode() == c["stdout"], (c["sequence"], o)
