"""Condense the rocprofv3 outputs of tools/profile_secondary.sh into profiles/rNN_secondary_kernels.json: per
secondary bench line (BASELINE configs[2..4]) the kernel-stats rows of the solver kernels, the duration of every
timed launch from the kernel trace, and the PMC counters of those launches (FETCH_SIZE x2 correction as for the
headline, MI355X_MICROARCH.md)."""
import sys, os, csv, glob, json
tag, out = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "gpurun_out", "profiles_" + tag)
os.makedirs(prof, exist_ok=True)
KERNELS = ("wide_kernel", "solve_kernel")


def find(d, pat):
    f = glob.glob(os.path.join(out, d, "**", pat), recursive=True)
    return f[0] if f else None


def sig_of(r, trace):
    """a launch configuration: lines that run the same kernel at several problem sizes (the state-dimension sweep)
    differ in grid, block or LDS size"""
    if trace:
        return (r["Kernel_Name"].split("(")[0], int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]), int(r["LDS_Block_Size"]))
    return (r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]), int(r["Workgroup_Size"]), int(r["LDS_Block_Size"]))


def counters(d):
    f = find(d, "*counter_collection.csv")
    per = {}
    if not f:
        return per
    acc = {}
    for r in csv.DictReader(open(f)):
        if not any(k in r["Kernel_Name"] for k in KERNELS):
            continue
        key = (sig_of(r, False), int(r["Dispatch_Id"]), r["Counter_Name"])
        acc[key] = acc.get(key, 0.0) + float(r["Counter_Value"])
    for (sig, disp, name), v in acc.items():
        per.setdefault(sig, {}).setdefault(disp, {})[name] = v
    return per


summary = {"source": "tools/profile_secondary.sh %s: rocprofv3 --kernel-trace --stats / --pmc passes of `python3 bench.py --config <name> ...`" % tag,
           "correction": "HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (FETCH_SIZE counts half of coalesced 8 B/lane streams on gfx950)"}
for name in ("quadruped", "state_dim", "rocket"):
    entry = {"kernel_stats": [], "launches": [], "bench_lines": []}
    ks = find(name + "/kt", "*kernel_stats.csv")
    if ks:
        for r in csv.DictReader(open(ks)):
            if any(k in r["Name"] for k in KERNELS):
                entry["kernel_stats"].append({k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MaxNs")})
    kt = find(name + "/kt", "*kernel_trace.csv")
    durs = {}
    if kt:
        rows = sorted((r for r in csv.DictReader(open(kt)) if any(k in r["Kernel_Name"] for k in KERNELS)), key=lambda r: int(r["Start_Timestamp"]))
        last_end = {}
        for r in rows:
            sig = sig_of(r, True)
            t0, t1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            # sweep points that share a kernel AND a launch configuration (n = 32 and n = 48 of the state-dimension sweep: the
            # trace does not report dynamic LDS) are told apart by the pause between them (the next point's problems are
            # generated on the host): a new group of dispatches after more than 0.3 s
            if sig not in durs or t0 - last_end[sig] > 300e6:
                durs.setdefault(sig, []).append([])
            durs[sig][-1].append(t1 - t0)
            last_end[sig] = t1
    fe, wr, sq = counters(name + "/fetch"), counters(name + "/write"), counters(name + "/sq")
    for sig, groups in durs.items():
      for point, ds in enumerate(groups):
        # the timed (fused) launch of each line is the longest dispatch of its kernel, launch configuration and sweep point
        i = max(range(len(ds)), key=lambda j: ds[j])
        rec = {"kernel": sig[0], "grid_size": sig[1], "workgroup_size": sig[2], "lds_block_size": sig[3], "point": point, "dispatches": len(ds),
               "timed_launch_ms": ds[i] / 1e6}
        for src in (fe, wr, sq):
            disp = src.get(sig, {})
            if disp:
                ids = sorted(disp)   # the PMC passes replay the same dispatch sequence: the same share of it is this sweep point's
                per = max(1, len(ids) // len(groups))
                mine = ids[point * per:(point + 1) * per] if point < len(groups) - 1 else ids[point * per:]
                best = max((disp[d] for d in mine), key=lambda c: sum(c.values()))   # the same (largest) launch in the PMC pass
                rec.update(best)
        if "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
            rec["hbm_bytes"] = (2.0 * rec["FETCH_SIZE"] + rec["WRITE_SIZE"]) * 1024.0
            rec["hbm_TBps"] = rec["hbm_bytes"] / (rec["timed_launch_ms"] * 1e-3) / 1e12
        if "SQ_WAVE_CYCLES" in rec and "SQ_ACTIVE_INST_VALU" in rec:
            rec["valu_busy_per_wave"] = rec["SQ_ACTIVE_INST_VALU"] / rec["SQ_WAVE_CYCLES"]
        entry["launches"].append(rec)
    log = os.path.join(out, name + ".kt.log")
    if os.path.exists(log):
        for ln in open(log):
            if ln.startswith("{") and '"metric"' in ln:
                d = json.loads(ln)
                entry["bench_lines"].append({k: d[k] for k in ("metric", "value", "ms_per_step", "steps", "config", "roofline")})
    summary[name] = entry
json.dump(summary, open(os.path.join(prof, tag + "_secondary_kernels.json"), "w"), indent=1)
print(json.dumps({k: [(l["kernel"][-40:], round(l["timed_launch_ms"], 2), round(l.get("hbm_TBps", 0), 3), round(l.get("valu_busy_per_wave", 0), 3)) for l in v["launches"]]
                  for k, v in summary.items() if isinstance(v, dict)}))
