"""Condense the rocprofv3 outputs of tools/profile_round.sh into the small files kept under profiles/."""
import sys, os, csv, glob, json
tag, out = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "gpurun_out", "profiles_" + tag)
os.makedirs(prof, exist_ok=True)
KERNEL = "solve_kernel<12, 4"


def find(d, pat):
    f = glob.glob(os.path.join(out, d, "**", pat), recursive=True)
    return f[0] if f else None


# kernel stats (top rows) and the trace rows of the solve kernel
ks = find("kt", "*kernel_stats.csv")
if ks:
    rows = list(csv.reader(open(ks)))
    with open(os.path.join(prof, tag + "_kernel_stats.csv"), "w", newline="") as f:
        csv.writer(f).writerows(rows[:8])
kt = find("kt", "*kernel_trace.csv")
launches = []
if kt:
    rows = list(csv.DictReader(open(kt)))
    sel = [r for r in rows if KERNEL in r["Kernel_Name"]]
    with open(os.path.join(prof, tag + "_kernel_trace_solve.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(sel[0].keys()))
        w.writeheader()
        w.writerows(sel)
    launches = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in sel]
summary = {"source": "tools/profile_round.sh %s: rocprofv3 --kernel-trace --stats / --pmc passes of `python3 bench.py --no-cpu-baseline`" % tag,
           "solve_kernel_launch_ns": launches}


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
        per.setdefault(name, []).append(v)
    return {k: max(v) for k, v in per.items()}     # the fused 100-step launch is the largest dispatch


fe, wr, sq = counters("fetch"), counters("write"), counters("sq")
summary["fused_launch_counters"] = {**fe, **wr, **sq}
if "FETCH_SIZE" in fe and "WRITE_SIZE" in wr:
    # FETCH_SIZE in KB reports half of the bytes of coalesced 8 B/lane and 16 B/lane streams on gfx950
    # (MI355X_MICROARCH.md, tools/probes/fetch_calib.hip); WRITE_SIZE is exact
    hbm = (2.0 * fe["FETCH_SIZE"] + wr["WRITE_SIZE"]) * 1024.0
    ms = max(launches) / 1e6 if launches else None
    traffic = {"source": summary["source"], "kernel": "altro::solve_kernel<12,4>", "batch": 8192, "steps": 100, "steps_per_launch": 100,
               "FETCH_SIZE_KB": fe["FETCH_SIZE"], "WRITE_SIZE_KB": wr["WRITE_SIZE"],
               "correction": "FETCH_SIZE x2 (calibrated with tools/probes/fetch_calib.hip), WRITE_SIZE exact",
               "hbm_bytes_per_launch": hbm, "launch_ms": ms, "hbm_bytes_per_solve": hbm / (8192 * 100), "algorithmic_bytes_per_solve": 28576}
    json.dump(traffic, open(os.path.join(prof, tag + "_traffic.json"), "w"), indent=1)
json.dump(summary, open(os.path.join(prof, tag + "_pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary)[:600])
