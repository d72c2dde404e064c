"""The engine's scheduling rules (which LSTM kernel: resolve_lstm, engine.hip; how many CUs the recurrence gets for itself: lstm_partition_cus) were fitted
to sweeps on one box.  This test holds them against the box it runs on: at the shapes either side of each rule's thresholds, the rule's own choice is timed
against the forced alternatives, and must be within 5 % of the best of them -- so a box with another clock or CU layout reopens a cliff of profiles/EXPERIMENTS.md item 7 loudly, not silently.  (The arithmetic is that of silero_v3.c:72-215 whichever way the rules fall: the kernels are bit-identical,
tests/test_gpu_parity.py::test_lstm_variants_agree.)"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

STEPS = 60
ALTERNATIVES = [{"lstm": 6}, {"lstm": 7}, {"lstm": 7, "lstm_cus": 32}, {"lstm": 7, "lstm_cus": 64},
                {"cu_partition": 0}]       # (five: {"lstm": 6, "lstm_cus": 32 / 64} and {"lstm": 7, "cu_partition": 0} never came first in rounds 4-5; most of a rate() is the engine's creation)


@pytest.mark.parametrize("model,S,Cn", [("v31", 256, 96), ("v31", 320, 96), ("v31", 832, 32), ("v31", 2048, 32),
                                        ("v4", 256, 96), ("v4", 320, 96), ("v4", 640, 32), ("v4", 2048, 32)])      # (Silero v4: the rule of round 5, tools/v4_partition_sweep.py)
def test_the_rules_choice_is_within_5_percent_of_the_best_forced_alternative(model, S, Cn):
    import torch
    import bench
    blob = open(os.path.join(ROOT, "tests", "golden", "reference_fixtures", "silero_v31_16k.testtensor") if model == "v31" else
                os.path.join(ROOT, "tests", "golden", "silero_v4_16k.testtensor"), "rb").read()
    dev = torch.device("cuda", 0)

    def rate(opts):
        r = bench.side_config(torch, blob, dev, 0, model, S, Cn, 0, steps=STEPS, warmup=10, opts=opts or None)
        return r["value"], (r["resolved"]["lstm"], r["resolved"]["lstm_cus"], r["resolved"]["shared"])
    rule, chosen = rate({})
    alts, same = {}, []
    for o in ALTERNATIVES:
        v, cfg = rate(o)
        # a forced alternative that comes to the configuration the rule chose IS the rule's choice, measured once more: engines created one after the other do not
        # all get the same hardware queues, and the same configuration was seen 6 % apart inside one run of this test
        if cfg == chosen and "cu_partition" not in o:
            same.append(v)
        else:
            alts[tuple(sorted(o.items()))] = v
    rule = max([rule] + same)
    best_key = max(alts, key=alts.get)
    if rule < 0.95 * alts[best_key]:
        # before failing, the two contenders once more, the rule's choice last
        again_alt = rate(dict(best_key))[0]
        rule = max(rule, rate({})[0])
        alts[best_key] = min(alts[best_key], again_alt)
    report = ", ".join(f"{dict(k)}: {v / 1e6:.3f} M" for k, v in sorted(alts.items(), key=lambda kv: -kv[1]))
    print(f"\n{model} {S} x {Cn}: rule {rule / 1e6:.3f} M (lstm {chosen[0]} on {chosen[1]} CUs; measured {1 + len(same)} x); {report}")
    assert rule >= 0.95 * max(alts.values()), f"{model} {S} x {Cn}: the rule's choice runs at {rule / 1e6:.3f} M audio-s/s; forced alternatives: {report}"
