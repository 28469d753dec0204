#!/usr/bin/env python3
"""The MD step of the engine (new coordinates in, compute, results out) in a plain loop, for a
rocprofv3 kernel trace:  rocprofv3 --kernel-trace -d <dir> -- python3 scripts/md_loop.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import ni_frame, ni_model
from tensoralloy_amd import Engine, _lib

want = _lib.TA_WANT_ENERGY | _lib.TA_WANT_FORCES | _lib.TA_WANT_VIRIAL | _lib.TA_WANT_ATOMIC
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
atoms = ni_frame(611)
with Engine(ni_model()) as eng:
    eng.set_skin(float(os.environ.get("TA_MD_SKIN", "0.5")))   # 0: a new list every step
    eng.set_frames([atoms])
    pos = np.ascontiguousarray(atoms.positions)
    for k in range(steps + 5):
        if k == 5:
            eng.synchronize()
            t0 = time.perf_counter()
        if os.environ.get("TA_MD_THREE_CALLS"):
            eng.update_positions(pos)
            eng.compute(want)
            eng.fetch(want)
        else:
            eng.step(pos, want, view=not os.environ.get("TA_MD_COPY"))
    print("ms per step", (time.perf_counter() - t0) / steps * 1e3)
