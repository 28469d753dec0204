#!/usr/bin/env python3
"""Single-frame step time against the frame size (workgroups of the angular kernels against the
resident-workgroup capacity of the chip): python scripts/size_sweep.py [reps...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import ni_frame, ni_model
from tensoralloy_amd import Engine, _lib

want = _lib.TA_WANT_ENERGY | _lib.TA_WANT_FORCES | _lib.TA_WANT_VIRIAL | _lib.TA_WANT_ATOMIC
reps = [int(x) for x in sys.argv[1:]] or [7, 8, 9, 10, 11, 12, 14]
with Engine(ni_model()) as eng:
    for rep in reps:
        info = eng.set_frames([ni_frame(611, rep=rep)])
        ms, slots = eng.time_compute(want, 5, 50)
        n = int(info.n_atoms)
        print(json.dumps({"rep": rep, "atoms": n, "pairs": int(info.n_pairs), "us": round(ms / 50 * 1e3, 1),
                          "ns_per_atom": round(ms / 50 * 1e6 / n, 2),
                          "slots_us": {k: round(v * 1e3, 1) for k, v in slots.items() if v > 0}}))
