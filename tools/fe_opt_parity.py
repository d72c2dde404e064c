"""option "fe_opt" (k_frontend_sym's OPT mask; 11 = k_frontend_ri) against the oracle: the wide parity sweep of tests/reports/parity_report.py for each value, the magnitude tap
compared as uint32 between values, and the bench line of each.   python tools/fe_opt_parity.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from vadc_amd import synth
from vadc_amd.engine import Engine
blob = open("tests/golden/reference_fixtures/silero_v31_16k.testtensor", "rb").read()
orc = O.Oracle(blob)
wide = synth.make_streams(61, 400, seed0=31000)
ctrl = np.stack([synth.control_stream(k, 400 * 1536, seed=5) for k in ("zeros", "noise", "square")])
pcm = np.concatenate([wide, ctrl])
want = orc.forward_streams(pcm)
eng = Engine(blob, max_streams=64, max_chunks_per_call=100, device=0)
print("zero_im0 =", eng.get_option("zero_im0"))
rep, mags = {}, {}
x = (pcm[:8, :20 * 1536].astype(np.float32) / 32768.0).reshape(-1)
for opt in (0, 3, 11):                                   # (7 = log1p without its Newton step existed at commit 6cb6391 only)
    eng.set_option("fe_opt", opt); eng.reset_streams()
    got = np.concatenate([eng.run(pcm[:, i * 1536:(i + 100) * 1536]) for i in range(0, 400, 100)], axis=1)[:, :, 1]
    d = np.abs(got.astype(np.float64) - want).ravel()
    rep[opt] = {"max_abs_dp": float(d.max()), "p999": float(np.quantile(d, 0.999)), "mean": float(d.mean())}
    mags[opt] = eng.stage_from_samples(x, "magnitude").view(np.uint32)
    rep[opt]["probabilities_identical_to_fe_opt_0"] = bool(np.array_equal(got.view(np.uint32), first.view(np.uint32))) if opt else True
    if not opt: first = got
    print(opt, rep[opt], flush=True)
print("magnitude tap identical:", all(np.array_equal(mags[0], mags[k]) for k in (3, 11)))
eng.close()
json.dump(rep, open("gpurun_out/fe_opt_parity.json", "w"), indent=1)
