"""Register / LDS / spill table of every kernel in one source file: python tools/kernel_regs.py vadc_amd/csrc/kernels_lstm.hip [name filter]
(compiles the device side to an ISA listing in /tmp and reads the kernel metadata)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = "/tmp/" + os.path.basename(src) + ".s"
extra = ["-ffp-contract=off", "-fno-slp-vectorize"] if "kernels_frontend.hip" in src else []
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-Wno-unused-command-line-argument",
                       "-I", os.path.join(ROOT, "include"), "-o", out, src] + extra)
t = open(out).read()
for blk in t.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if flt not in name:
        continue
    g = lambda k: re.search(r"\." + k + r":\s+(\d+)", blk).group(1)
    print(f"{name[:70]:70s} vgpr {g('vgpr_count'):>3s} agpr {blk.split()[0]:>3s} spill {g('vgpr_spill_count'):>3s} sgpr {g('sgpr_count'):>3s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size')}")
print("listing:", out)
