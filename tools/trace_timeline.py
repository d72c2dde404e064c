"""Timeline of the timed region from a rocprofv3 kernel trace of `bench.py --steps K`: python tools/trace_timeline.py <dir> [K]
Prints, for the last K steps: the front end's start-to-start intervals, the gap between a step's last encoder kernel and the next front end, and the
drain (last encoder kernel end -> last LSTM kernel end)."""
import csv, glob, os, sys
d = sys.argv[1]; K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
f = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f)) if "vadc" in r["Kernel_Name"]]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
def kind(n):
    if "k_frontend" in n: return "fe"
    if "k_enc_fused" in n: return "enc"
    if "k_lstm_layer" in n:      # layer = the third template argument (mangled: k_lstm_layerILi7ELi0ELi<L>ELb0E...; demangled: k_lstm_layer<7, 0, <L>, false>)
        import re
        m = re.search(r"k_lstm_layerILi\d+ELi\d+ELi(\d)E", n) or re.search(r"k_lstm_layer<\d+, \d+, (\d)", n)
        return "l0" if m and m.group(1) == "0" else "l1"
    if "k_layer_mfma" in n or "k_layer1_regs" in n: return "l1k"
    return "other"
fe = [e for e in ev if kind(e[2]) == "fe"][-K:]
t0 = fe[0][0]
sel = [e for e in ev if e[0] >= t0]
enc = [e for e in sel if kind(e[2]) == "enc"]
lst = [e for e in sel if kind(e[2]) in ("l0", "l1")]
print("steps", len(fe), "region %.3f ms (first front end start -> last kernel end)" % ((max(e[1] for e in sel) - t0) / 1e6))
iv = [(fe[i + 1][0] - fe[i][0]) / 1e3 for i in range(len(fe) - 1)]
print("front-end start-to-start us:", " ".join("%.0f" % v for v in iv))
gaps = [(fe[i + 1][0] - enc[i][1]) / 1e3 for i in range(min(len(enc), len(fe) - 1))]
print("encoder end -> next front end start us:", " ".join("%.0f" % g for g in gaps))
print("drain: last encoder end -> last LSTM end %.3f ms" % ((max(e[1] for e in lst) - enc[-1][1]) / 1e6))
for k in ("fe", "l1k", "enc", "l0", "l1"):
    ds = [(e[1] - e[0]) / 1e3 for e in sel if kind(e[2]) == k]
    if ds: print(k, "n=%d avg %.0f us  first %.0f  last %.0f" % (len(ds), sum(ds) / len(ds), ds[0], ds[-1]))
fd = [(e[1] - e[0]) / 1e3 for e in sel if kind(e[2]) == "fe"]
print("front-end durations us:", " ".join("%.0f" % v for v in fd))
l1d = [(e[1] - e[0]) / 1e3 for e in sel if kind(e[2]) == "l1k"]
print("layer-1 durations us:", " ".join("%.0f" % v for v in l1d))
