#!/usr/bin/env python3
"""Phase timeline of the angular kernels from a -DTA_PHASE_STAMPS run (scripts/build_stamps_lib.sh):
per kernel, the mean / median / max over the workgroups of the time between consecutive stamps, and
of each stamp relative to the kernel's first stamp (s_memrealtime ticks of 10 ns)."""
import sys
import numpy as np

names = {0: ["entry", "staged", "masks", "jobs built", "list written", "sweep done", "assembled"],
         1: ["entry", "staged", "job words", "sweep done", "barrier", "epilogue done"]}
rows = {0: [], 1: []}
for line in open(sys.argv[1]):
    p = line.split()
    k, v = int(p[0]), np.array([int(x) for x in p[2:]], dtype=np.int64)
    if v[0] and v[1]:
        rows[k].append(v)
for k in (0, 1):
    if not rows[k]:
        continue
    a = np.array(rows[k])
    n = len(names[k])
    a = a[:, :n]
    ok = (a > 0).all(axis=1)
    a = a[ok]
    t0 = a[:, 0].min()
    print(f"kernel {'forward' if k == 0 else 'backward'}: {len(a)} workgroups; first entry -> last end "
          f"{(a[:, n - 1].max() - t0) / 100:.1f} us; entry spread {(a[:, 0].max() - t0) / 100:.1f} us")
    for q in range(1, n):
        d = (a[:, q] - a[:, q - 1]) / 100.0
        rel = (a[:, q] - t0) / 100.0
        print(f"  {names[k][q - 1]:>14s} -> {names[k][q]:<14s} mean {d.mean():6.2f}  median {np.median(d):6.2f}  "
              f"max {d.max():6.2f} us | reached at mean {rel.mean():6.2f}  max {rel.max():6.2f} us")
