"""The scheduling rules' choice against forced alternatives (LSTM kernel 6 / 7, partition 32 / 64 CUs or none) over a list of shapes, both models: where is the rule more
than a few per cent behind?  (tests/test_gpu_partition_rules.py asserts this at eight shapes; this prints it for any.)   python tools/rule_vs_alternatives.py v31:512x96 v4:896x32 ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
BLOB = {"v31": os.path.join(ROOT, "tests", "golden", "reference_fixtures", "silero_v31_16k.testtensor"), "v4": os.path.join(ROOT, "tests", "golden", "silero_v4_16k.testtensor")}
ALTS = [{"lstm": 6}, {"lstm": 7}, {"lstm": 6, "lstm_cus": 32}, {"lstm": 7, "lstm_cus": 32}, {"lstm": 7, "lstm_cus": 64}, {"lstm": 6, "lstm_cus": 64}, {"cu_partition": 0}, {"lstm": 7, "cu_partition": 0}]
dev = torch.device("cuda", 0)
for arg in sys.argv[1:]:
    model, shape = arg.split(":"); S, Cn = (int(x) for x in shape.split("x"))
    blob = open(BLOB[model], "rb").read()
    rate = lambda o: bench.side_config(torch, blob, dev, 0, model, S, Cn, 0, steps=100, warmup=10, opts=o or None)["value"] / 1e6
    rule = rate({})
    alts = sorted(((rate(o), o) for o in ALTS), key=lambda t: -t[0])
    flag = "   <-- rule %.1f %% behind" % (100 * (alts[0][0] / rule - 1)) if alts[0][0] > 1.03 * rule else ""
    print(f"{model} {S} x {Cn}: rule {rule:.3f} M; best {alts[0][0]:.3f} {alts[0][1]}; 2nd {alts[1][0]:.3f} {alts[1][1]}{flag}", flush=True)
