#!/bin/bash
# LDS experiment switches of the angular kernels (wrong results by construction):
#   TA_DEBUG_SKIP bit 4 (16): partner atomics to conflict-free addresses; bit 5 (32): the late partner
#   reads ({1/r H}, {G species}) conflict-free; bit 6 (64): every partner read conflict-free, no exact test
OUT=gpurun_out/lds_probe.txt; : > $OUT
for nf in 1 16; do
  steps=$((nf > 4 ? 10 : 50))
  for sk in 0 16 32 48 64 80 0; do
    echo -n "frames $nf skip $sk " >> $OUT
    TA_DEBUG_SKIP=$sk python scripts/run_config.py sf $nf $steps | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); f = d['frames']
print(round(d['us_per_frame'], 1), {k: round(v * 1e3 / f, 1) for k, v in d['kernel_ms'].items()})" >> $OUT
  done
done
cat $OUT
