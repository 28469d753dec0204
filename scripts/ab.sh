#!/bin/bash
# same-session A/B of the working library against tensoralloy_amd/libtensoralloy_amd_base.so
#   bash scripts/ab.sh <config> "<frames list>" [rounds]
CFG=${1:-sf}; FR=${2:-"1 16"}; ROUNDS=${3:-2}
for k in $(seq $ROUNDS); do
  for nf in $FR; do
    steps=$((nf > 4 ? 10 : 50))
    echo -n "new  "; python scripts/run_config.py $CFG $nf $steps | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); f = d['frames']
print(f, round(d['us_per_frame'], 1), {k: round(v * 1e3 / f, 1) for k, v in d['kernel_ms'].items()})"
    echo -n "base "; TA_LIB_AB=tensoralloy_amd/libtensoralloy_amd_base.so python scripts/run_config.py $CFG $nf $steps | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); f = d['frames']
print(f, round(d['us_per_frame'], 1), {k: round(v * 1e3 / f, 1) for k, v in d['kernel_ms'].items()})"
  done
done
