#!/bin/bash
# rocprofv3 passes behind profiles/rNN_*: kernel trace + stats, then one PMC pass per counter group
# (separate runs: --pmc is never combined with other trace domains).  Usage: tools/profile_round.sh r01
set -e
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python3 bench.py --no-cpu-baseline > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o pmc --output-format csv -- python3 bench.py --no-cpu-baseline > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o pmc --output-format csv -- python3 bench.py --no-cpu-baseline > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/sq -o pmc --output-format csv -- python3 bench.py --no-cpu-baseline > $OUT/sq.log 2>&1
python3 tools/profile_summarise.py $TAG $OUT
