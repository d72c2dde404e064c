#!/usr/bin/env python3
"""The kernels that wait for their LDS-DMA pieces with COUNTED `s_waitcnt vmcnt(N)` (k_layer1_regs, k_layer1_regs_v4, k_lstm_layer) issue those pieces from
inline asm the compiler cannot see: the counts hold only if the compiled kernel issues exactly the vector-memory operations the source assumes, spills nothing
(scratch traffic is vector-memory traffic; an in-flight load's destination register must not move) and if nothing but those asm statements writes M0 between
an `s_mov_b32 m0` and the DMA that reads it.  (hipcc rejects "m0" in an asm clobber list -- a reserved register -- so the listing is where this is checked.)
Run by vadc_amd/csrc/Makefile on every build of these files and by tests/test_abi.py:
    python tools/check_counted_waits.py <listing.s> [<listing.s> ...]        # exit code 1 and a message on the first violation
"""
import re
import sys

# kernel name fragment (Itanium-mangled, after _ZN4vadc<len>) -> expectations on its listing
RULES = {
    "k_layer1_regsILi12ELi0": {"dma": 28, "load_dword": 8, "min_store_dword": 4},         # per iteration: 4 partial sums, 4 groups of 4 pieces, 4 stores; sums and three groups once more in the prologue
    "k_layer1_regs_v4ILi12": {"dma": 22, "load_dword": 8, "min_store_dword": 4},          # per iteration 3 + 3 + 3 + 4 pieces; 9 in the prologue
    # ring prologue 4 + one per slot.  hipcc keeps the state write-back's addresses in scratch across the slot loop (two stores in the prologue, two loads in the
    # epilogue: older than every piece the loop waits for, so the counts stand); INSIDE the loop there must be none
    "k_lstm_layerILi7ELi0ELi0ELb0ELb0": {"dma": 5, "scratch_outside_loop_ok": True},
    "k_lstm_layerILi7ELi0ELi1ELb0ELb0": {"dma": 5, "scratch_outside_loop_ok": True},
    # the TRAIL forms (layer 1 beside layer 0 of the same call): the published tile count travels through the scalar cache, no vector-memory operation is added
    "k_lstm_layerILi7ELi0ELi0ELb0ELb1": {"dma": 5, "scratch_outside_loop_ok": True},
    "k_lstm_layerILi7ELi0ELi1ELb0ELb1": {"dma": 5, "scratch_outside_loop_ok": True},
}


def inner_loop(body):
    """the text of the kernel's SLOT loop: the innermost loop that holds the MFMAs and the barrier (the TRAIL kernels wrap it in a loop over blocks and have
    polling loops around it) -- from its first block to the first label behind the last block that names it as its header"""
    lines = body.split("\n")
    best = ""
    for i, l in enumerate(lines):
        if not re.search(r"Loop Header: Depth=\d", l):
            continue
        m = next((mm for mm in (re.match(r"\.L(BB\d+_\d+):", lines[j]) for j in range(i, max(i - 4, -1), -1)) if mm), None)      # (the header's comments may sit on the lines behind its label)
        if not m:
            continue
        tag = "Header=" + m.group(1) + " "
        members = [j for j, x in enumerate(lines) if tag in x + " "]
        last = max(members) if members else i
        end = next((j for j in range(last + 1, len(lines)) if re.match(r"\.LBB\d+_\d+:", lines[j]) and tag not in lines[j] + " " and "Parent Loop " + m.group(1) not in lines[j]), len(lines))
        text = "\n".join(lines[min([i - 3] + members):end])         # (a rotated loop's first block may sit in front of its header label)
        if "v_mfma" in text and "s_barrier" in text and (not best or len(text) < len(best)):
            best = text
    return best


def kernels(txt):
    for m in re.finditer(r"^(_ZN4vadc\d+\S+):", txt, re.M):
        name = m.group(1)
        end = txt.find(".Lfunc_end", m.end())
        yield name, txt[m.end():end if end > 0 else len(txt)]


def check(path):
    txt = open(path).read()
    errors, seen = [], set()
    meta = {}
    for blk in txt.split("  - .agpr_count:")[1:]:
        nm = re.search(r"\.name:\s+(\S+)", blk).group(1)
        meta[nm] = (int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1)), int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1)))
    for name, body in kernels(txt):
        for frag, rule in RULES.items():
            if frag not in name:
                continue
            seen.add(frag)
            spill, scratch = meta.get(name, (None, None))
            if rule.get("scratch_outside_loop_ok"):
                if re.search(r"\sscratch_", inner_loop(body)) or not inner_loop(body):
                    errors.append(f"{name}: scratch instructions inside the slot loop (or no loop found)")
                # a compiler-made full drain in the slot loop (it waited there for the carried state's loads until round 4) also drains the pieces the counted waits leave in flight
                if re.search(r"s_waitcnt\s+vmcnt\(0\)", inner_loop(body)):
                    errors.append(f"{name}: s_waitcnt vmcnt(0) inside the slot loop")
            else:
                if spill != 0 or scratch != 0:
                    errors.append(f"{name}: {spill} spilled registers, {scratch} bytes of scratch (must be 0 / 0)")
                if re.search(r"\sscratch_", body):
                    errors.append(f"{name}: scratch instructions in the listing")
            dma = len(re.findall(r"\sglobal_load_lds_dwordx4\s", body))
            if dma != rule["dma"]:
                errors.append(f"{name}: {dma} global_load_lds_dwordx4, the waits count {rule['dma']}")
            if "load_dword" in rule:
                n = len(re.findall(r"\sglobal_load_dword\s", body))
                if n != rule["load_dword"]:
                    errors.append(f"{name}: {n} global_load_dword, the waits count {rule['load_dword']}")
            if "min_store_dword" in rule:
                n = len(re.findall(r"\sglobal_store_dword\s", body))
                if n < rule["min_store_dword"]:
                    errors.append(f"{name}: {n} global_store_dword, the waits count at least {rule['min_store_dword']}")
            # every write of M0 is one of ours (s_mov_b32 m0 directly in front of its DMA); the compiler's own uses of M0 would sit between them unseen
            m0_writes = len(re.findall(r"\ss_mov_b32\s+m0,", body))
            other_m0 = len(re.findall(r"\s(?!s_mov_b32)\S+\s+m0,", body))
            if m0_writes != dma + rule.get("extra_m0", 0) or other_m0:
                errors.append(f"{name}: {m0_writes} s_mov_b32 m0 for {dma} + {rule.get('extra_m0', 0)} DMA pieces, {other_m0} other writes of m0")
            if "dword_dma" in rule:
                n = len(re.findall(r"\sglobal_load_lds_dword\s", body))
                if n != rule["dword_dma"]:
                    errors.append(f"{name}: {n} global_load_lds_dword, expected {rule['dword_dma']}")
    return errors, seen


def main(paths):
    bad = []
    seen = set()
    for p in paths:
        e, s = check(p)
        bad += [f"{p}: {x}" for x in e]
        seen |= s
    for b in bad:
        print("check_counted_waits:", b, file=sys.stderr)
    if not bad:
        print(f"check_counted_waits: {len(seen)} kernel(s) in {len(paths)} listing(s): vector-memory operation counts match their waits, no spills, M0 written by the DMA statements only")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
