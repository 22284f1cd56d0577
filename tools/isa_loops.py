#!/usr/bin/env python3
"""Summarise the innermost loops of a kernel's ISA listing (tools/kernel_meta.sh leaves /tmp/last_kernel.s):
per loop (backward branch to a label) the instruction mix of its body.  python tools/isa_loops.py [file.s] [min_len]"""
import re, sys
path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/last_kernel.s"
minlen = int(sys.argv[2]) if len(sys.argv) > 2 else 40
L = open(path).read().split("\n")
labels = {}
for i, l in enumerate(L):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = i
loops = []
for i, l in enumerate(L):
    m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i, m.group(1)))
# innermost: no other loop strictly inside
inner = [lp for lp in loops if not any(o[0] > lp[0] and o[1] < lp[1] for o in loops)]
def cls(op):
    if op.startswith("v_fmac_f64_dpp"): return "fmac_dpp"
    if op.startswith(("v_permlane",)): return "permlane"
    if op.endswith("_dpp") or "_dpp" in op: return "dpp_other"
    if re.match(r"v_(fma|fmac|add|mul|max|min|rcp|sqrt|div|cmp\w*|cvt\w*|trig|rsq|ldexp|frexp\w*)_\w*f64", op) or re.match(r"v_cmp\w*_f64", op): return "valu_f64"
    if op.startswith("v_cndmask"): return "cndmask"
    if op.startswith("v_"): return "valu_other"
    if op.startswith("global_load"): return "gload"
    if op.startswith("global_store"): return "gstore"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    return "other"
print("%-12s %6s %5s | %8s %8s %8s %8s %8s %6s %6s %6s %5s %5s %5s %5s" % ("loop", "line", "instr", "fmac_dpp", "valu_f64", "cndmask", "valu_oth", "dpp_oth", "perml", "gload", "gstore", "lds", "wait", "nop", "salu"))
for a, b, name in sorted(inner):
    c = {}
    n = 0
    for l in L[a:b + 1]:
        m = re.match(r"\s+([a-z_0-9]+)", l)
        if not m or l.strip().startswith((";", ".")): continue
        k = cls(m.group(1)); c[k] = c.get(k, 0) + 1; n += 1
    if n < minlen: continue
    valu = sum(c.get(k, 0) for k in ("fmac_dpp", "valu_f64", "cndmask", "valu_other", "dpp_other", "permlane"))
    print("%-12s %6d %5d | %8d %8d %8d %8d %8d %6d %6d %6d %5d %5d %5d %5d  VALU %d" % (name, a + 1, n, c.get("fmac_dpp", 0), c.get("valu_f64", 0), c.get("cndmask", 0), c.get("valu_other", 0), c.get("dpp_other", 0), c.get("permlane", 0), c.get("gload", 0), c.get("gstore", 0), c.get("lds", 0), c.get("waitcnt", 0), c.get("nop", 0), c.get("salu", 0), valu))

if len(sys.argv) > 3 and sys.argv[3] == "waits":
    print("\nvmcnt waits inside each loop: (instruction index in the body: vmcnt) and the memory operations around them")
    for a, b, name in sorted(inner):
        body = [l for l in L[a:b + 1] if re.match(r"\s+[a-z_0-9]+", l) and not l.strip().startswith((";", "."))]
        if len(body) < minlen: continue
        ev = []
        for i, l in enumerate(body):
            op = l.split()[0]
            if op.startswith("global_load"): ev.append("L")
            elif op.startswith("global_store"): ev.append("S")
            elif op == "s_waitcnt" and "vmcnt" in l: ev.append("[%d:w%s]" % (i, re.search(r"vmcnt\((\d+)\)", l).group(1)))
        print("%-10s n=%d  %s" % (name, len(body), "".join(ev)))
