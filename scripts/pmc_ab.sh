#!/bin/bash
# SQ_INSTS_VALU / SALU / LDS of the angular kernels: working library against the A/B base library.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_ab; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/new -- python3 $ROOT/scripts/run_config.py sf 1 5 > $OUT/new.log 2>&1
TA_LIB_AB=$ROOT/tensoralloy_amd/libtensoralloy_amd_base.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/base -- python3 $ROOT/scripts/run_config.py sf 1 5 > $OUT/base.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
for tag in ("new", "base"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/pmc_ab/{tag}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            name = "forward" if "forward_v2" in k else "backward" if "backward_v2" in k else None
            if name:
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in sorted(acc.items()):
        print(tag, k, {c: round(sum(v) / len(v) / 1e6, 3) for c, v in sorted(d.items())})
PY
