#!/usr/bin/env python3
"""Generator for the inner tile of k_frontend_mx2 (vadc_amd/csrc/frontend_mx_tile.inc).

One tile = 64 positions x 16 filters x 256 taps of the reference's STFT tree (stft.c:115-184):
    tap t = 64 i + 8 j + l ;  g_i[l] = ((p0+p1)+(p2+p3)) + ((p4+p5)+(p6+p7)) over j
    v[l] = (g_0 + g_1) + (g_2 + g_3) ;  y = ((v0+v1)+(v2+v3)) + ((v4+v5)+(v6+v7))
The 256 products come from 256 v_mfma_f32_16x16x1_4b_f32 with a zero accumulator (individually rounded products,
tools/mfma_k1.hip); the 255 separately rounded adds are 255 "add rows" of 16 v_add_f32 each.  hipcc cannot schedule
this (it hoists the pure MFMAs and spills), so this script IS the scheduler and the register allocator:
  * MFMA k is issued in slot k, followed by at most ROWS_PER_SLOT add rows whose operands are complete;
  * a product is consumed no earlier than the slot after its MFMA, which also satisfies the MFMA->VALU hazard
    (the hardware does not interlock it) -- where a slot is short the generator pads with s_nop;
  * 16-register buffers are allocated by liveness; an add row writes in place over its first operand.
It prints the resource summary and writes the .inc file (a string literal for one asm volatile statement).
"""
import sys

N_L, N_I, N_J = 8, 4, 8
HAZARD = 14          # instructions between an MFMA and the first VALU read of its result (8-pass SGEMM MFMA: 11 wait states; not interlocked)

class Val:
    def __init__(self, name, a=None, b=None, prod=None):
        self.name, self.a, self.b, self.prod = name, a, b, prod
        self.ready = None     # slot from which it may be read
        self.buf = None
        self.user = None
        self.cls = 0          # register-bank class: 0 = left operand of its consumer, 1 = right operand

def build():
    """The tree.  v_add_f32 is commutative bit for bit, so which operand of an add row sits in which register-bank class is
    free: `first` says which class the FIRST-computed (long-lived: it waits for its sibling subtree) operand gets at each
    tree level; the per-level choice below minimises the number of buffers (brute force over the 2^8 assignments)."""
    prods, rows = [], []
    def add(name, a, b, first):
        v = Val(name, a, b); a.user = v; b.user = v; a.cls, b.cls = first, 1 - first; rows.append(v); return v
    vs = []
    for l in range(N_L):
        gs = []
        for i in range(N_I):
            p = []
            for j in range(N_J):
                v = Val(f"p{l}{i}{j}", prod=len(prods)); prods.append(v); p.append(v)
            a = add(f"a{l}{i}", p[0], p[1], 0); b = add(f"b{l}{i}", p[2], p[3], 0); h0 = add(f"h0_{l}{i}", a, b, 1)
            c = add(f"c{l}{i}", p[4], p[5], 0); d = add(f"d{l}{i}", p[6], p[7], 0); h1 = add(f"h1_{l}{i}", c, d, 1)
            gs.append(add(f"g{l}{i}", h0, h1, 0))
        g01 = add(f"g01_{l}", gs[0], gs[1], 0); g23 = add(f"g23_{l}", gs[2], gs[3], 0)
        vs.append(add(f"v{l}", g01, g23, 1))
    s01 = add("s01", vs[0], vs[1], 0); s23 = add("s23", vs[2], vs[3], 0); s0123 = add("s0123", s01, s23, 0)
    s45 = add("s45", vs[4], vs[5], 0); s67 = add("s67", vs[6], vs[7], 0); s4567 = add("s4567", s45, s67, 0)
    y = add("y", s0123, s4567, 1)
    return prods, rows, y

def schedule(rows_per_slot=1):
    prods, rows, y = build()
    for k, p in enumerate(prods): p.ready = k + 1
    pending = list(rows)            # already in dependency (tree) order
    slots = [[] for _ in range(len(prods))]
    drain = []
    for k in range(len(prods)):
        n = 0
        while n < rows_per_slot:
            pick = None
            for r in pending:
                if r.a.ready is not None and r.b.ready is not None and r.a.ready <= k and r.b.ready <= k:
                    pick = r; break
            if pick is None: break
            pending.remove(pick); pick.ready = k; slots[k].append(pick); n += 1
    for r in pending:
        r.ready = len(prods); drain.append(r)
    return prods, slots, drain, y

def allocate(prods, slots, drain):
    """Two pools of 16-register buffers.  VGPR banks are register-index mod 4 and a v_add_f32 whose two sources sit in the
    same bank issues at half rate (tools/valu_rate2.hip), so class-1 buffers are skewed by 2 registers (MFMA tuples must
    stay even-aligned) and every add row reads one operand of each class; it writes in place over the operand whose class
    its own consumer expects."""
    free, nbuf, live, maxlive = ([], []), [0, 0], [0, 0], [0, 0]
    def get(c):
        live[c] += 1; maxlive[c] = max(maxlive[c], live[c])
        if free[c]: return free[c].pop(0)
        nbuf[c] += 1; return nbuf[c] - 1
    def put(c, b):
        live[c] -= 1; free[c].append(b)
    if not SKEW:
        for v in prods: v.cls = 0
        for s_ in slots:
            for r in s_: r.cls = r.a.cls = r.b.cls = 0
        for r in drain: r.cls = 0
    def do_row(r):
        assert (not SKEW) or r.a.cls != r.b.cls
        keep, drop = (r.a, r.b) if r.cls == r.a.cls else (r.b, r.a)
        r.buf = keep.buf; put(drop.cls, drop.buf)
    for k, p in enumerate(prods):
        p.buf = get(p.cls)
        for r in slots[k]: do_row(r)
    for r in drain: do_row(r)
    return nbuf, maxlive

# ---- emission -------------------------------------------------------------------------------------------------
VB      = 48                # first VGPR the block owns; the compiler keeps v0..v47 for the code around it
GJ      = 4                 # taps per operand fetch (one ds_read_b128 / global_load_dwordx4)
DA, DB  = 2, 3              # prefetch distance in fetch groups: x from LDS, basis from L2
LDS_BLOCK_PITCH = 68        # floats per 64-sample block in the x tile (bank spread + 16-byte alignment)
SKEW, ABLATE, PKADD = True, set(), False
SBASE   = 40                # s[40:41]: running base of the B tile (advanced by 4 KB every 16 fetch groups)

def emit(path):
    prods, slots, drain, y = schedule(1)
    nbuf, _ = allocate(prods, slots, drain)
    base = (VB, VB + 16 * nbuf[0] + 2)       # class-1 buffers are skewed by two registers (bank = index mod 4)
    va = base[1] + 16 * nbuf[1] + 2          # A operands: (DA + 1) x GJ
    if not SKEW: va = VB + 16 * nbuf[0]
    vbb = va + GJ * (DA + 1)                 # B operands: (DB + 1) x GJ
    vend = vbb + GJ * (DB + 1)
    assert vend <= 256, vend
    out, n_instr = [], 0
    mfma_at = {}
    def ins(t):
        nonlocal n_instr
        out.append(t); n_instr += 1
    def reg0(v): return base[v.cls] + 16 * v.buf
    def buf(v, e): return f"v{reg0(v) + e}"
    n_groups = len(prods) // GJ
    cur_blk = 0
    def load_a(q):
        g, jq = divmod(q, N_J // GJ); l, i = divmod(g, N_I)
        a0 = va + GJ * (q % (DA + 1))
        if "noloads" in ABLATE and q >= DA + 1: return
        for h in range(GJ // 4):
            ins(f"ds_read_b128 v[{a0 + 4 * h}:{a0 + 4 * h + 3}], %[xaddr] offset:{(LDS_BLOCK_PITCH * i + 8 * l + GJ * jq + 4 * h) * 4}")
    def load_b(q):
        nonlocal cur_blk
        if "noloads" in ABLATE and q >= DB + 1: return
        if q // (64 // GJ) != cur_blk:
            ins(f"s_add_u32 s{SBASE}, s{SBASE}, 0x1000"); ins(f"s_addc_u32 s{SBASE + 1}, s{SBASE + 1}, 0"); ins("s_nop 4")
            cur_blk = q // (64 // GJ)
        b0 = vbb + GJ * (q % (DB + 1))
        for h in range(GJ // 4):
            ins(f"global_load_dwordx4 v[{b0 + 4 * h}:{b0 + 4 * h + 3}], %[boff], s[{SBASE}:{SBASE + 1}] offset:{(q % (64 // GJ)) * 64 * GJ + 256 * h}")
    def row(r):
        # hazard: a product may only be read HAZARD instructions after its MFMA
        need = 0
        for o in (r.a, r.b):
            if o.prod is not None: need = max(need, mfma_at[o.prod] + HAZARD - n_instr)
        while need > 0:
            k = min(need, 16); ins(f"s_nop {k - 1}"); need -= k
        if "noadds" in ABLATE and r is not y: return
        if PKADD:
            for e in range(0, 16, 2):
                d, a, b = reg0(r) + e, reg0(r.a) + e, reg0(r.b) + e
                ins(f"v_pk_add_f32 v[{d}:{d + 1}], v[{a}:{a + 1}], v[{b}:{b + 1}]")
            return
        for e in range(16):
            ins(f"v_add_f32 {buf(r, e)}, {buf(r.a, e)}, {buf(r.b, e)}")
    ins(f"s_mov_b64 s[{SBASE}:{SBASE + 1}], %[bbase]")
    for q in range(DB): load_b(q)
    for q in range(DA): load_a(q)
    for k, p in enumerate(prods):
        q, j = divmod(k, GJ)
        if j == 0:
            if q + DB < n_groups: load_b(q + DB)
            if q + DA < n_groups: load_a(q + DA)
            w = GJ // 4
            if "noloads" in ABLATE: ins("s_waitcnt vmcnt(0) lgkmcnt(0)")
            else: ins(f"s_waitcnt vmcnt({w * min(DB, n_groups - 1 - q)}) lgkmcnt({w * min(DA, n_groups - 1 - q)})")
        d0 = reg0(p)
        mfma_at[k] = n_instr
        if "k4" in ABLATE:       # experiment: a 4-register-result MFMA of the same 8 passes (garbage values)
            ins(f"v_mfma_f32_16x16x4_f32 v[{d0}:{d0 + 3}], v{va + GJ * (q % (DA + 1)) + j}, v{vbb + GJ * (q % (DB + 1)) + j}, 0")
        elif "nomfma" not in ABLATE or k < 16:
            ins(f"v_mfma_f32_16x16x1_4b_f32 v[{d0}:{d0 + 15}], v{va + GJ * (q % (DA + 1)) + j}, v{vbb + GJ * (q % (DB + 1)) + j}, 0")
        for r in slots[k]: row(r)
    for r in drain: row(r)
    y0 = reg0(y)
    clob = [f"v{r}" for r in range(VB, vend) if not (y0 <= r < y0 + 16)] + [f"s{SBASE}", f"s{SBASE + 1}"]
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_mx_asm.py -- do not edit.  One STFT tile: 64 positions x 16 filters x 256 taps,\n"
                "// products on v_mfma_f32_16x16x1_4b_f32 (C = 0), the reference's 255-add tree (stft.c:115-184) on the vector ALU.\n")
        f.write(f"// registers: v{VB}..v{vend - 1} ({nbuf[0]}+{nbuf[1]} tree buffers x 16, {DA + 1} x {GJ} A, {DB + 1} x {GJ} B), s[{SBASE}:{SBASE + 1}]; "
                f"{n_instr} instructions\n")
        f.write(f"#define VADC_MX_TILE_Y_CONSTRAINT \"=&{{v[{y0}:{y0 + 15}]}}\"\n")
        f.write("#define VADC_MX_TILE_CLOBBERS " + ", ".join(f'"{c}"' for c in clob) + ', "memory"\n')
        f.write("#define VADC_MX_TILE_ASM \\\n")
        for t in out:
            f.write(f'   "{t}\\n\\t" \\\n')
        f.write('   ""\n')
    return nbuf, vend, n_instr

if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("path", nargs="?")
    ap.add_argument("--vb", type=int, default=VB); ap.add_argument("--gj", type=int, default=GJ)
    ap.add_argument("--da", type=int, default=DA); ap.add_argument("--db", type=int, default=DB)
    ap.add_argument("--pkadd", action="store_true", help="add rows as 8 v_pk_add_f32 instead of 16 v_add_f32")
    ap.add_argument("--no-skew", action="store_true", help="experiment: one buffer pool, no bank skew")
    ap.add_argument("--ablate", default="", help="experiments (tools/mx_tile_bench.hip): comma list of nomfma,noadds,noloads,k4")
    args = ap.parse_args()
    VB, GJ, DA, DB, SKEW, ABLATE = args.vb, args.gj, args.da, args.db, not args.no_skew, set(filter(None, args.ablate.split(",")))
    path = args.path
    PKADD = args.pkadd
    prods, slots, drain, y = schedule(1)
    nbuf, maxlive = allocate(prods, slots, drain)
    print(f"buffers {nbuf} (max live {maxlive}), drain rows {len(drain)}, empty slots {sum(1 for s in slots if not s)}")
    if path:
        print("emitted: buffers %s, last vgpr %d, %d instructions" % emit(path))
