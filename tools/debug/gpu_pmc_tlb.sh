#!/bin/bash
# TLB and memory-latency counters of the headline kernel's timed launch: tools/debug/gpu_pmc_tlb.sh lib.so [batch]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_tlb
rm -rf $OUT && mkdir -p $OUT
export ALTRO_HIP_LIB=${1:-$GRAFT_REPO_ROOT/altro-mpc-icra2021_amd/csrc/libaltro_hip.so}
B=${2:-8192}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for grp in "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $OUT/g$i -o pmc --output-format csv -- python3 bench.py --steps 20 --warmup 5 --batch $B --no-cpu-baseline --no-secondary --repeats 1 > $OUT/g$i.log 2>&1 || echo "group $i failed"
done
python3 - <<'P'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_tlb"
for f in sorted(glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    ks = [r for r in rows if "solve_kernel" in r["Kernel_Name"]]
    if not ks: continue
    last = max(int(r["Dispatch_Id"]) for r in ks)
    acc = collections.OrderedDict()
    for r in ks:
        if int(r["Dispatch_Id"]) == last:
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    print(dict(acc))
P
