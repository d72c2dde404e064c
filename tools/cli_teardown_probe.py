"""A short-lived host CLI beside a parent process that holds a GPU context of its own (what the GPU test suite is): N runs in a row, each with the teardown marks on
(VADC_AMD_TRACE_TEARDOWN); the first one that does not come back within 30 s is reported with the marks it had written, and the loop stops.
   python tools/cli_teardown_probe.py [N=300] [extra CLI arguments ...]"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vadc_amd.engine import Engine      # noqa: E402
from vadc_amd import synth              # noqa: E402

WEIGHTS = os.path.join(ROOT, "tests", "golden", "reference_fixtures", "silero_v31_16k.testtensor")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    extra = sys.argv[2:]
    blob = open(WEIGHTS, "rb").read()
    parent = Engine(blob, max_streams=64, max_chunks_per_call=40, device=0)      # the parent's own queues, idle most of the time
    pcm_parent = synth.make_streams(64, 40, seed0=1)
    parent.run(pcm_parent)
    pcm = synth.make_streams(1, 60, seed0=7)[0].tobytes()
    exe = os.path.join(ROOT, "host", "vadc_hip")
    env = dict(os.environ, VADC_AMD_TRACE_TEARDOWN="1")
    t0 = time.time()
    worst = 0.0
    for i in range(n):
        t = time.time()
        try:
            r = subprocess.run([exe, "--model", WEIGHTS, *extra], input=pcm, capture_output=True, timeout=30, env=env)
        except subprocess.TimeoutExpired as ex:
            print(f"run {i}: did not come back; stdout {len(ex.stdout or b'')} bytes; stderr: {(ex.stderr or b'').decode(errors='replace')[-1200:]}", flush=True)
            return 1
        if r.returncode != 0:
            print(f"run {i}: rc {r.returncode}: {r.stderr.decode(errors='replace')[-1200:]}", flush=True)
            return 1
        worst = max(worst, time.time() - t)
        if i % 8 == 3:
            parent.run(pcm_parent)                                                 # the parent works now and then, as a test suite does
        if i % 25 == 24:
            print(f"{i + 1} runs, {time.time() - t0:.0f} s, slowest {worst:.2f} s", flush=True)
    parent.close()
    print(f"{n} runs came back, slowest {worst:.2f} s")
    return 0


if __name__ == "__main__":
    sys.exit(main())
