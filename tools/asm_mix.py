#!/usr/bin/env python3
"""Instruction mix of the kernels in a gfx950 assembly listing (hipcc -S --cuda-device-only), per kernel whose name matches a
filter: whole body and the largest loop (the tile loop).  usage: python tools/asm_mix.py file.s <name filter>"""
import collections, re, sys

src, flt = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
kern = None
bodies = {}
for ln in lines:
    m = re.match(r"^(_Z\S+):\s*; @", ln)
    if m:
        kern = m.group(1) if flt in m.group(1) else None
        if kern:
            bodies[kern] = []
        continue
    if kern:
        if ln.startswith(".Lfunc_end"):
            kern = None
            continue
        bodies[kern].append(ln)


def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_accvgpr"): return "acc_mov"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem:" + ("atomic" if "atomic" in op else "scratch" if op.startswith("scratch") else "ld/st")
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_cvt_pk_bf16"): return "v_cvt_pk_bf16"
    if op.startswith("v_"): return "valu"
    return "other"


for k, body in bodies.items():
    ins = []
    labels = {}
    for ln in body:
        t = ln.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not t or t.startswith((";", ".")):
            continue
        ins.append(t.split(";")[0].strip())
    # phases (-DLSE_PHASE_MARKS): instruction mix between consecutive "; LSE_PHASE name" comments, in listing order
    ph, cur = collections.OrderedDict(), None
    for ln in body:
        t = ln.strip()
        m = re.match(r"^; LSE_PHASE (\S+)", t)
        if m:
            cur = m.group(1)
            ph.setdefault(cur, [])
            continue
        if cur is None or not t or t.startswith((";", ".")):
            continue
        ph[cur].append(t.split(";")[0].strip())
    if ph:
        print("==", k[:110])
        keys = ["mfma", "valu", "v_cvt_pk_bf16", "s_nop", "lds", "s_waitcnt", "salu"]
        print("   phase      n   " + " ".join(f"{x:>8s}" for x in keys) + "   v_mov cndmsk  rdlane")
        for name, li in ph.items():
            c = collections.Counter(cls(t.split()[0]) for t in li)
            mv = sum(1 for t in li if t.startswith("v_mov_b32"))
            cm = sum(1 for t in li if t.startswith("v_cndmask"))
            rl = sum(1 for t in li if t.startswith(("v_readlane", "v_writelane")))
            print(f"   {name:8s} {len(li):5d} " + " ".join(f"{c.get(x, 0):8d}" for x in keys) + f"  {mv:5d} {cm:6d} {rl:6d}")
    # loops: backward branches
    loops = []
    for i, t in enumerate(ins):
        m = re.match(r"^s_c?branch\S*\s+(\.LBB\d+_\d+)", t)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            loops.append((i - labels[m.group(1)], labels[m.group(1)], i))
    loops.sort(reverse=True)
    print("==", k[:110])
    for name, lo, hi in [("whole", 0, len(ins) - 1)] + ([("largest loop", loops[0][1], loops[0][2])] if loops else []):
        c = collections.Counter(cls(t.split()[0]) for t in ins[lo:hi + 1])
        valu = collections.Counter(t.split()[0] for t in ins[lo:hi + 1] if cls(t.split()[0]) == "valu")
        lds = collections.Counter(t.split()[0] for t in ins[lo:hi + 1] if cls(t.split()[0]) == "lds")
        n = hi - lo + 1
        print(f"  {name}: {n} instructions;", ", ".join(f"{a} {b}" for a, b in c.most_common()))
        print("     valu top:", ", ".join(f"{a} {b}" for a, b in valu.most_common(14)))
        print("     lds:", ", ".join(f"{a} {b}" for a, b in lds.most_common(10)))
