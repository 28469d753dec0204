#!/bin/bash
# rocprofv3 kernel trace + PMC (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU in separate passes) for the EAM,
# ADP and GRAP configs, one frame and 64 frames. Usage inside gpurun: bash scripts/profile_configs.sh <tag>
set -u
TAG=${1:-r02}
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
for CFG in eam adp grap; do
  for NF in 1 64; do
    OUT=$ROOT/gpurun_out/prof_${TAG}_${CFG}_${NF}
    mkdir -p $OUT
    python3 $ROOT/scripts/run_config.py $CFG $NF 20 > $OUT/plain.json 2> $OUT/plain.err
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/scripts/run_config.py $CFG $NF 10 > $OUT/trace.log 2>&1
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/scripts/run_config.py $CFG $NF 5 > $OUT/pmc_fetch.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/scripts/run_config.py $CFG $NF 5 > $OUT/pmc_write.log 2>&1
    rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/scripts/run_config.py $CFG $NF 5 > $OUT/pmc_sq.log 2>&1
    (cd $ROOT && python3 scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1)
    echo "$CFG $NF: $(cat $OUT/plain.json | head -c 400)"
  done
done
