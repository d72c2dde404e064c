#!/usr/bin/env python3
"""Which correction terms of the split-fp16 GEMMs (W . X ~= Wl . Xh + Wh . Xl + Wh . Xh) does the 1e-4 parity bar need in encoder layers 2-4?
Rebuilds kernels_encoder_fused.hip with -DVADC_ENC_WLO=0 / -DVADC_ENC_XLO=0 (GPU box) and compares the engine with the CPU oracle on the
wide sweep of parity_report.py (64 streams x 400 chunks).   python tests/reports/enc_terms_report.py"""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def build(flags):
    c = os.path.join(ROOT, "vadc_amd", "csrc")
    objs = ["engine", "kernels_frontend", "kernels_frontend_gemm", "kernels_encoder_mfma", "kernels_encoder_fused", "kernels_lstm", "kernels_v5"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", *flags, "-c",
                           "kernels_encoder_fused.hip", "-o", "build/kernels_encoder_fused.o"], cwd=c)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-o", "../libvadc_amd.so"] + [f"build/{o}.o" for o in objs], cwd=c)

def measure():
    code = r'''
import json, sys, numpy as np
sys.path.insert(0, %r)
from oracle import oracle as O
from vadc_amd import synth
from vadc_amd.engine import Engine
blob = open("tests/golden/reference_fixtures/silero_v31_16k.testtensor", "rb").read()
wide = synth.make_streams(61, 400, seed0=31000)
ctrl = np.stack([synth.control_stream(k, 400 * 1536, seed=5) for k in ("zeros", "noise", "square")])
pcm = np.concatenate([wide, ctrl])
want = np.load("gpurun_out/enc_terms_want.npy") if __import__("os").path.exists("gpurun_out/enc_terms_want.npy") else None
if want is None:
    want = O.Oracle(blob).forward_streams(pcm); np.save("gpurun_out/enc_terms_want.npy", want)
e = Engine(blob, max_streams=64, max_chunks_per_call=100, device=0)
got = np.concatenate([e.run(pcm[:, i * 1536:(i + 100) * 1536]) for i in range(0, 400, 100)], axis=1)[:, :, 1]
d = np.abs(got.astype(np.float64) - want).ravel()
print(json.dumps({"max_abs_dp": float(d.max()), "p999_abs_dp": float(np.quantile(d, 0.999)), "mean_abs_dp": float(d.mean())}))
''' % ROOT
    out = subprocess.check_output([sys.executable, "-c", code], cwd=ROOT, text=True)
    return json.loads(out.strip().split("\n")[-1])

rep = {}
for name, flags in (("all three terms", []), ("without Wl.Xh", ["-DVADC_ENC_WLO=0"]), ("without Wh.Xl", ["-DVADC_ENC_XLO=0"]),
                    ("Wh.Xh only", ["-DVADC_ENC_WLO=0", "-DVADC_ENC_XLO=0"])):
    build(flags)
    rep[name] = measure()
    print(name, rep[name], flush=True)
build([])
json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "enc_terms_report.json"), "w"), indent=1)
