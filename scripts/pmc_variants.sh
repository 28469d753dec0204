#!/bin/bash
# VALU instruction accounting of the angular kernels: normal run against a run with the triple
# bodies switched off (TA_DEBUG_NO_TRIPLES=1, wrong results) for 1 and 16 frames.
# Usage (inside gpurun, repo root): bash scripts/pmc_variants.sh <tag>
set -u
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmcvar_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for NF in 1 16; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/full_$NF -- python3 $ROOT/scripts/run_config.py sf $NF 5 > $OUT/full_$NF.log 2>&1
  TA_DEBUG_NO_TRIPLES=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/notrip_$NF -- python3 $ROOT/scripts/run_config.py sf $NF 5 > $OUT/notrip_$NF.log 2>&1
done
cd $ROOT
python3 - <<'PY' $OUT
import csv, glob, sys, collections
out = sys.argv[1]
for tag in ("full_1", "notrip_1", "full_16", "notrip_16"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{tag}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            if "forward_v2" in k or "backward_v2" in k:
                acc[k[:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in acc.items():
        print(tag, k, {c: round(sum(v) / len(v) / 1e6, 3) for c, v in d.items()})
PY
