#!/bin/bash
# rocprofv3 passes behind profiles/rNN_secondary_*: kernel trace + stats, then one PMC pass per counter group (separate
# runs: --pmc is never combined with other trace domains), for the secondary bench lines of BASELINE configs[2..4]:
#   tools/profile_secondary.sh r02      (on the GPU box; raw output under gpurun_out/prof_sec_r02, summaries under
#                                        gpurun_out/profiles_r02 -- copy those into profiles/)
set -e
TAG=${1:-r02}
OUT=gpurun_out/prof_sec_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for CFG in "quadruped:--config quadruped --steps 10 --warmup 3" "state_dim:--config state_dim --steps 10 --warmup 3" "rocket:--config rocket --steps 10 --warmup 3"; do  # the steps of the default run's secondary lines
  NAME=${CFG%%:*}
  ARGS=${CFG#*:}
  rocprofv3 --kernel-trace --stats -d $OUT/$NAME/kt -o kt --output-format csv -- python3 bench.py $ARGS > $OUT/$NAME.kt.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $OUT/$NAME/fetch -o pmc --output-format csv -- python3 bench.py $ARGS > $OUT/$NAME.fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $OUT/$NAME/write -o pmc --output-format csv -- python3 bench.py $ARGS > $OUT/$NAME.write.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/$NAME/sq -o pmc --output-format csv -- python3 bench.py $ARGS > $OUT/$NAME.sq.log 2>&1
  echo "profiled $NAME"
done
python3 tools/profile_secondary_summarise.py $TAG $OUT
