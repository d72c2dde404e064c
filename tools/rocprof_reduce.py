#!/usr/bin/env python3
"""Reduce rocprofv3 output directories into the small summaries kept under profiles/.

  python tools/rocprof_reduce.py --kernel-trace gpurun_out/prof_kt --fetch gpurun_out/prof_fetch --write gpurun_out/prof_write \
         --out profiles/r01 --tag bench_256x96 --streams 256 --chunks-per-step 96

* kernel trace:  <dir>/**/*_kernel_stats.csv  -> <out>/<tag>_kernel_stats.csv (copied, vadc kernels + totals only)
* PMC passes  :  FETCH_SIZE and WRITE_SIZE were collected in SEPARATE runs (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE);
                 per-kernel averages -> <out>/<tag>_pmc_traffic.json and profiles/latest_pmc_traffic.json, with the gfx950
                 correction of MI355X_MICROARCH.md (FETCH_SIZE counts 64 B per 128-B request): hbm = (2*FETCH + WRITE) KB.
"""
import argparse, csv, glob, json, os, re, shutil, sys
from collections import defaultdict

SHORT = [("k_frontend", "k_frontend"), ("k_lstm", "k_lstm")]
FE_NAMES = ["k_frontend_ri", "k_frontend_sym", "k_frontend_fl", "k_frontend_gemm2", "k_frontend_gemm", "k_frontend"]     # most specific first
FE_SEEN = set()


def short_name(full):
    if "k_enc_fused" in full:
        return "k_enc234"
    if "k_layer1_regs" in full:
        return "k_layer1"
    m = re.search(r"k_lstm_layer<\d+, \d+, (\d)[,>]", full) or re.search(r"k_lstm_layerILi\d+ELi\d+ELi(\d)E", full)     # rocprofv3 leaves some names mangled
    if m:
        # <TS, DEC, L, TAPDEC, TRAIL, REDO>: the REDO form (round 6: behind every pair, leaves at once unless a tile's layer 1 gave up) is a launch of its own
        redo = re.search(r"k_lstm_layer<\d+, \d+, \d, (?:true|false), (?:true|false), true>", full) or re.search(r"k_lstm_layerILi\d+ELi\d+ELi\dELb[01]ELb[01]ELb1E", full)
        return "k_lstm_redo" if redo else ("k_lstm" if m.group(1) == "0" else "k_lstm_l1")
    m = re.search(r"k_layer_mfma<(\d+), (\d+), (\d+)", full) or re.search(r"k_layer<(\d+), (\d+), (\d+)", full)
    if m:
        return {("129", "16"): "k_layer1", ("258", "16"): "k_layer1", ("16", "32"): "k_layer2", ("32", "32"): "k_layer3", ("32", "64"): "k_layer4"}[(m.group(1), m.group(2))]
    for fe in FE_NAMES:
        if fe + "<" in full or fe + "(" in full:
            FE_SEEN.add("k_frontend (v4 tree)" if fe == "k_frontend" else fe)
            break
    for key, name in SHORT:
        if key in full:
            return name
    return None


def find(d, pat):
    hits = glob.glob(os.path.join(d, "**", pat), recursive=True)
    if not hits:
        sys.exit(f"no {pat} under {d}")
    return max(hits, key=os.path.getmtime)        # gpurun merges runs into the same directory: take the newest


def pmc_avg(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    with open(find(d, "*_counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            k = short_name(row["Kernel_Name"])
            if k:
                acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel-trace"); ap.add_argument("--fetch"); ap.add_argument("--write")
    ap.add_argument("--pmc", action="append", default=[], help="directory of an extra --pmc pass; all its counters are averaged per kernel")
    ap.add_argument("--out", required=True); ap.add_argument("--tag", required=True)
    ap.add_argument("--streams", type=int, default=256); ap.add_argument("--chunks-per-step", type=int, default=96)
    ap.add_argument("--model", default="v31"); ap.add_argument("--precision", default="fp32")
    ap.add_argument("--command", default="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    if a.kernel_trace:
        src = find(a.kernel_trace, "*_kernel_stats.csv")
        with open(src) as f, open(os.path.join(a.out, a.tag + "_kernel_stats.csv"), "w") as g:
            for i, line in enumerate(f):
                if i == 0 or "vadc::" in line or "_ZN4vadc" in line:      # rocprofv3 leaves long template names mangled (k_lstm_layer)
                    g.write(line)
        print("wrote", os.path.join(a.out, a.tag + "_kernel_stats.csv"))
    if a.fetch and a.write:
        fe, wr = pmc_avg(a.fetch, "FETCH_SIZE"), pmc_avg(a.write, "WRITE_SIZE")
        out = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- {a.command}",
               "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE counts 64 B per 128-B request; MI355X_MICROARCH.md section HBM)",
               "model": a.model, "precision": a.precision, "frontend_kernel": sorted(FE_SEEN)[0] if len(FE_SEEN) == 1 else None,
               "streams": a.streams, "chunks_per_step": a.chunks_per_step, "kernels": {}}
        for k in sorted(set(fe) | set(wr)):
            out["kernels"][k] = {"FETCH_SIZE_KB": round(fe.get(k, 0.0), 1), "WRITE_SIZE_KB": round(wr.get(k, 0.0), 1),
                                 "hbm_bytes_per_launch": int((2 * fe.get(k, 0.0) + wr.get(k, 0.0)) * 1024)}
        p = os.path.join(a.out, a.tag + "_pmc_traffic.json")
        json.dump(out, open(p, "w"), indent=1)
        default_workload = a.model == "v31" and a.precision == "fp32" and a.streams == 256 and a.chunks_per_step == 96
        latest = "latest_pmc_traffic.json" if default_workload else f"latest_pmc_traffic_{a.model}_{a.precision}_{a.streams}x{a.chunks_per_step}.json"
        shutil.copy(p, os.path.join(os.path.dirname(os.path.abspath(a.out)), latest))   # what bench.py reads for `roofline.traffic`
        print("wrote", p)
    if a.pmc:
        merged = defaultdict(dict)
        for d in a.pmc:
            for k, cs in pmc_all(d).items():
                merged[k].update(cs)
        out = {"source": f"rocprofv3 --pmc <counters> (one pass per directory, no tracing) -- {a.command}",
               "note": "per-kernel averages per launch, summed over all XCDs/SEs as rocprofv3 reports them",
               "streams": a.streams, "chunks_per_step": a.chunks_per_step,
               "kernels": {k: {c: round(v, 1) for c, v in sorted(cs.items())} for k, cs in sorted(merged.items())}}
        for k, cs in out["kernels"].items():
            if cs.get("SQ_BUSY_CYCLES") and cs.get("SQ_ACTIVE_INST_VALU") is not None and cs.get("SQ_WAVE_CYCLES"):
                cs["derived_valu_active_per_wave_cycle"] = round(cs["SQ_ACTIVE_INST_VALU"] / cs["SQ_WAVE_CYCLES"], 4)
            if cs.get("SQ_INSTS_VALU") and cs.get("SQ_INSTS_VALU_MFMA_F32") is not None:
                cs["derived_mfma_share_of_valu_insts"] = round(cs["SQ_INSTS_VALU_MFMA_F32"] / cs["SQ_INSTS_VALU"], 4)
        p = os.path.join(a.out, a.tag + "_pmc_compute.json")
        json.dump(out, open(p, "w"), indent=1)
        print("wrote", p)


def pmc_all(d):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    with open(find(d, "*_counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            k = short_name(row["Kernel_Name"])
            if k:
                a = acc[k][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    return {k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in acc.items()}


if __name__ == "__main__":
    main()
