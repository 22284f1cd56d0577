#!/bin/bash
# Instruction-cache counters of the headline kernel's timed launch (the last solve_kernel dispatch of bench.py --steps 20)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_icache
rm -rf $OUT && mkdir -p $OUT
export ALTRO_HIP_LIB=${1:-$GRAFT_REPO_ROOT/altro-mpc-icra2021_amd/csrc/libaltro_hip.so}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $OUT/a -o pmc --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --repeats 1 > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $OUT/b -o pmc --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --repeats 1 > $OUT/b.log 2>&1
python3 - <<'P'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_icache"
for sub in ("a", "b"):
    for f in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        ks = [r for r in rows if "solve_kernel" in r["Kernel_Name"]]
        last = max(int(r["Dispatch_Id"]) for r in ks)
        acc = collections.OrderedDict()
        for r in ks:
            if int(r["Dispatch_Id"]) == last:
                acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        print(sub, "dispatch", last, dict(acc))
P
