"""Condense the rocprofv3 outputs of tools/profile_round.sh into the small files kept under profiles/."""
import sys, os, csv, glob, json, re
tag, out = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "gpurun_out", "profiles_" + tag)
os.makedirs(prof, exist_ok=True)
KERNEL = "solve_kernel<12, 4"
CONFIGS = {"s20": dict(steps=20, warmup=5), "s100": dict(steps=100, warmup=5)}
BATCH = 8192


def find(d, pat):
    f = glob.glob(os.path.join(out, d, "**", pat), recursive=True)
    return f[0] if f else None


def counters(d):
    f = find(d, "*counter_collection.csv")
    acc = {}
    if not f:
        return acc
    for r in csv.DictReader(open(f)):
        if KERNEL not in r["Kernel_Name"]:
            continue
        key = (r["Dispatch_Id"], r["Counter_Name"])
        acc[key] = acc.get(key, 0.0) + float(r["Counter_Value"])
    per = {}
    for (disp, name), v in acc.items():
        per.setdefault(name, []).append((int(disp), v))
    return {k: sorted(v)[-1][1] for k, v in per.items()}     # the timed (fused) launch is the last dispatch of the kernel


entries, summary = [], {"source": "tools/profile_round.sh %s: rocprofv3 --kernel-trace --stats / --pmc passes of `python3 bench.py <args> --no-cpu-baseline --no-secondary --repeats 1` (the timed K-step region of the headline line: the last dispatch of the kernel)" % tag}
for name, cfg in CONFIGS.items():
    ks = find(name + "/kt", "*kernel_stats.csv")
    if ks:
        rows = list(csv.reader(open(ks)))
        with open(os.path.join(prof, "%s_%s_kernel_stats.csv" % (tag, name)), "w", newline="") as f:
            csv.writer(f).writerows(rows[:8])
    kt = find(name + "/kt", "*kernel_trace.csv")
    launches = []
    if kt:
        rows = list(csv.DictReader(open(kt)))
        sel = [r for r in rows if KERNEL in r["Kernel_Name"]]
        if sel:
            with open(os.path.join(prof, "%s_%s_kernel_trace_solve.csv" % (tag, name)), "w", newline="") as f:
                w = csv.DictWriter(f, fieldnames=list(sel[0].keys()))
                w.writeheader()
                w.writerows(sel)
            launches = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in sel]
    fe, wr, sq = counters(name + "/fetch"), counters(name + "/write"), counters(name + "/sq")
    bench_line = None
    log = os.path.join(out, name + ".kt.log")
    if os.path.exists(log):
        for ln in open(log):
            if ln.startswith("{") and '"metric"' in ln:
                bench_line = json.loads(ln)
    summary[name] = {"bench_args": "--steps %(steps)d --warmup %(warmup)d" % cfg, "solve_kernel_launch_ns": launches,
                     "timed_launch_counters": {**fe, **wr, **sq},
                     "bench_line_under_kernel_trace": bench_line}
    if "FETCH_SIZE" in fe and "WRITE_SIZE" in wr and launches:
        # FETCH_SIZE (KB) reports half of the bytes of coalesced 8 B/lane and 16 B/lane streams on gfx950
        # (MI355X_MICROARCH.md, tools/probes/fetch_calib.hip); WRITE_SIZE is exact
        hbm = (2.0 * fe["FETCH_SIZE"] + wr["WRITE_SIZE"]) * 1024.0
        solves = BATCH * cfg["steps"]
        entries.append({"kernel": "altro::solve_kernel<12,4>", "batch": BATCH, "steps": cfg["steps"], "warmup": cfg["warmup"],
                        "steps_per_launch": cfg["steps"], "FETCH_SIZE_KB": fe["FETCH_SIZE"], "WRITE_SIZE_KB": wr["WRITE_SIZE"],
                        "hbm_bytes_per_launch": hbm, "launch_ms": launches[-1] / 1e6, "hbm_bytes_per_solve": hbm / solves,
                        "hbm_TBps": hbm / (launches[-1] * 1e-9) / 1e12, "algorithmic_bytes_per_solve": 28576})
json.dump({"source": summary["source"], "correction": "FETCH_SIZE x2 (calibrated with tools/probes/fetch_calib.hip), WRITE_SIZE exact",
           "entries": entries}, open(os.path.join(prof, tag + "_traffic.json"), "w"), indent=1)
json.dump(summary, open(os.path.join(prof, tag + "_pmc_summary.json"), "w"), indent=1)
print(json.dumps({"entries": entries})[:1500])
