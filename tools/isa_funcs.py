#!/usr/bin/env python3
"""Per function of an ISA listing (tools/kernel_meta_wide.sh leaves /tmp/last_wide.s): registers, scratch, and for every loop
of 300+ instructions the scratch operations, v_readlane reloads and vmcnt(0) waits inside it.  python tools/isa_funcs.py [file.s]"""
import re, sys
path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/last_wide.s"
L = open(path).read().split("\n")
funcs = []
cur = None
for i, l in enumerate(L):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        cur = [m.group(1), i, None, {}]
        funcs.append(cur)
    m = re.match(r"^; (NumVgprs|NumAgprs|ScratchSize|TotalNumSgprs|codeLenInByte)[:=]?\s*=?\s*(\d+)", l)
    if m and cur is not None:
        cur[3][m.group(1)] = int(m.group(2))
        cur[2] = i
isinstr = lambda x: re.match(r"\s+[vsdgb][a-z]*_", x) is not None
for name, a, b, info in funcs:
    if b is None: continue
    short = re.sub(r"^_ZN?\d*", "", name)[:70]
    body = L[a:b]
    n = sum(1 for x in body if isinstr(x))
    print("%-72s instr %6d  vgpr %3d agpr %3d sgpr %3d scratch %4d B | scratch ops %4d readlane %4d" % (
        short, n, info.get("NumVgprs", 0), info.get("NumAgprs", 0), info.get("TotalNumSgprs", 0), info.get("ScratchSize", 0),
        sum(1 for x in body if "scratch_" in x), sum(1 for x in body if "v_readlane" in x)))
    labels = {}
    for i in range(a, b):
        m = re.match(r"^(\.LBB\d+_\d+):", L[i])
        if m: labels[m.group(1)] = i
    loops = []
    for i in range(a, b):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", L[i])
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            k = sum(1 for x in L[labels[m.group(1)]:i] if isinstr(x))
            if k >= 300: loops.append((labels[m.group(1)], i, m.group(1), k))
    for la, lb, ln, k in sorted(loops):
        seg = L[la:lb]
        print("    loop %-12s %6d instr: scratch %3d  readlane %4d  writelane %3d  vmcnt(0) %3d  accvgpr %4d" % (
            ln, k, sum(1 for x in seg if "scratch_" in x), sum(1 for x in seg if "v_readlane" in x),
            sum(1 for x in seg if "v_writelane" in x), sum(1 for x in seg if "vmcnt(0)" in x), sum(1 for x in seg if "v_accvgpr" in x)))
