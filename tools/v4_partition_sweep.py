"""Silero v4: throughput by LSTM kernel and partition size at the stream counts where the recurrence, not the front end + encoder, sets the step (round 5: the front end
+ encoder stream got 25 % faster and the rule of round 3 -- 32 CUs from 20 tiles up -- left the chain as the bound).  python tools/v4_partition_sweep.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
blob = open(os.path.join(ROOT, "tests", "golden", "silero_v4_16k.testtensor"), "rb").read()
dev = torch.device("cuda", 0)
shapes = [(320, 96), (512, 96), (640, 32), (1024, 32), (1280, 32), (2048, 32)] if len(sys.argv) < 2 else [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for S, Cn in shapes:
    row = {}
    full = os.environ.get("V4_SWEEP_FULL") == "1"
    variants = [(f"lstm{k}_cus{c}", {"lstm": k, "lstm_cus": c}) for k in (6, 7) for c in (32, 48, 64, 96, 128)] + [("lstm6_nopart", {"lstm": 6, "cu_partition": 0})] if full else \
               [(f"lstm7_cus{c}", {"lstm": 7, "lstm_cus": c}) for c in (32, 48, 64)] + [("lstm6_cus32", {"lstm": 6, "lstm_cus": 32})]
    for name, opts in [("rule", {})] + variants:
        try:
            r = bench.side_config(torch, blob, dev, 0, "v4", S, Cn, 0, steps=100, warmup=10, opts=opts or None)
            row[name] = round(r["value"] / 1e6, 3)
        except Exception as ex:
            row[name] = str(ex)[:40]
    best = max((v, k) for k, v in row.items() if isinstance(v, float))
    print(f"{S} x {Cn}: rule {row['rule']}  best {best}  {json.dumps(row)}", flush=True)
