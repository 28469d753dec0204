#!/bin/bash
# Forward-kernel instruction accounting by phase (TA_DEBUG_SKIP bits: 1 triple bodies, 2 candidate
# scan, 4 G2 sums, 8 job sweep; wrong results by construction). Usage inside gpurun: bash scripts/pmc_phases.sh
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_phases; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for SK in 0 1 8 10 14; do
  TA_DEBUG_SKIP=$SK rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/skip$SK -- python3 $ROOT/scripts/run_config.py sf 1 5 > $OUT/skip$SK.log 2>&1
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
for sk in (0, 1, 8, 10, 14):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/pmc_phases/skip{sk}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            name = "forward" if "forward_v2" in k else "backward" if "backward_v2" in k else None
            if name:
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
                acc[name]["us"].append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-3 * 1e6)
    for k, d in sorted(acc.items()):
        print("skip", sk, k, {c: round(sum(v) / len(v) / 1e6, 3) for c, v in sorted(d.items())})
PY
