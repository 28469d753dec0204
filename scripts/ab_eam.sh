#!/bin/bash
# A/B of the EAM / ADP kernels in one GPU session: lanes per atom (TA_EAM_W).
# Usage: bash scripts/ab_eam.sh [out file]
out=${1:-gpurun_out/ab_eam.txt}
: > "$out"
for rep in 1 2; do
  for kind in eam adp; do
    for F in 1 64; do
      for W in 16 32 64; do
        echo "# $kind W=$W frames=$F" >> "$out"
        TA_EAM_W=$W python scripts/run_config.py $kind $F 30 >> "$out" 2>&1 || exit 1
      done
    done
  done
done
