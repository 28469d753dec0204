#!/bin/bash
# A/B sweep of the phase-stagger switch of the second-generation angular kernels (experiment).
# Usage (inside gpurun): bash scripts/stagger_sweep.sh > gpurun_out/stagger.txt
run() {
  python bench.py --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('%-14s %-14s %.1f us  fwd %.1f  bwd %.1f  %.2f M' % (os.environ.get('TA_STAGGER_FWD','-'), os.environ.get('TA_STAGGER_BWD','-'), d['ms_per_step']*1e3, k['g4_forward']*1e3, k['backward']*1e3, d['value']/1e6))"
}
run
for shift in 3 5 8; do
  for n in 1 2 4 6; do
    TA_STAGGER_FWD=$n,$shift run
    TA_STAGGER_BWD=$n,$shift run
  done
done
run
