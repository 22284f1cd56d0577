#!/bin/bash
# rocprofv3 passes behind profiles/rNN_*: for each profiled bench.py command line a kernel trace + stats pass, then
# one PMC pass per counter group (separate runs: --pmc is never combined with other trace domains).
#   tools/profile_round.sh r02            (on the GPU box; outputs under gpurun_out/prof_r02, summaries under
#                                          gpurun_out/profiles_r02 -- copy those into profiles/)
# Profiled command lines: the driver's (`--steps 20 --warmup 5`) and bench.py's default (100 steps).
set -e
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for CFG in "s20:--steps 20 --warmup 5" "s100:--steps 100 --warmup 5"; do
  NAME=${CFG%%:*}
  ARGS=${CFG#*:}
  rocprofv3 --kernel-trace --stats -d $OUT/$NAME/kt -o kt --output-format csv -- python3 bench.py $ARGS --no-cpu-baseline --no-secondary --repeats 1 > $OUT/$NAME.kt.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $OUT/$NAME/fetch -o pmc --output-format csv -- python3 bench.py $ARGS --no-cpu-baseline --no-secondary --repeats 1 > $OUT/$NAME.fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $OUT/$NAME/write -o pmc --output-format csv -- python3 bench.py $ARGS --no-cpu-baseline --no-secondary --repeats 1 > $OUT/$NAME.write.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $OUT/$NAME/sq -o pmc --output-format csv -- python3 bench.py $ARGS --no-cpu-baseline --no-secondary --repeats 1 > $OUT/$NAME.sq.log 2>&1
  echo "profiled $NAME"
done
python3 tools/profile_summarise.py $TAG $OUT
