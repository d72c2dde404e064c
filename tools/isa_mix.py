"""Instruction mix of every kernel in a hipcc -S listing: python tools/isa_mix.py file.s [name-filter]"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur, cnt = None, None
def report():
    if cur is None or flt not in cur: return
    mfma = sum(v for k, v in cnt.items() if "mfma" in k)
    valu = sum(v for k, v in cnt.items() if k.startswith("v_") and "mfma" not in k)
    ds = sum(v for k, v in cnt.items() if k.startswith("ds_"))
    vm = sum(v for k, v in cnt.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_")))
    sal = sum(v for k, v in cnt.items() if k.startswith("s_"))
    print(f"{cur}: mfma {mfma}  valu {valu}  lds {ds}  vmem {vm}  salu {sal} (s_nop {cnt['s_nop']}, s_waitcnt {cnt['s_waitcnt']})")
    print("   top valu:", ", ".join(f"{k} {v}" for v, k in sorted(((v, k) for k, v in cnt.items() if k.startswith("v_") and "mfma" not in k), reverse=True)[:18]))
for ln in lines:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        report(); cur, cnt = m.group(1), collections.Counter(); continue
    if ln.startswith(".Lfunc_end"):
        report(); cur = None; continue
    if cur is not None:
        m = re.match(r"\s+([a-z][a-z_0-9]+)", ln)
        if m: cnt[m.group(1)] += 1
