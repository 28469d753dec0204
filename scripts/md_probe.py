#!/usr/bin/env python3
"""Where one MD step of the drop-in goes: engine-level pieces and the calculator call (4000-atom Ni)."""
import json, os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import ni_frame, ni_model
from tensoralloy_amd import Engine, TensorAlloyCalculator, _lib

want = _lib.TA_WANT_ENERGY | _lib.TA_WANT_FORCES | _lib.TA_WANT_VIRIAL | _lib.TA_WANT_ATOMIC
nn = ni_model()
atoms = ni_frame(611)
out = {}

def timeit(fn, n=100):
    fn(); fn()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t) / n * 1e3

with Engine(nn) as eng:
    for skin in (0.0, 0.3, 0.5):
        eng.set_skin(skin)
        info = eng.set_frames([atoms])
        pos = np.ascontiguousarray(atoms.positions)
        tag = f"skin{skin}"
        out[tag] = {"pairs": int(info.n_pairs), "nnl_max": int(info.nnl_max),
                    "compute_ms": eng.time_compute(want, 3, 30, per_kernel=False)[0] / 30}
        if skin > 0:
            out[tag]["update_positions_ms"] = timeit(lambda: (eng.update_positions(pos), eng.synchronize()))
            out[tag]["compute_sync_ms"] = timeit(lambda: (eng.compute(want), eng.synchronize()))
            out[tag]["fetch_ms"] = timeit(lambda: eng.fetch(want))
            out[tag]["step_ms"] = timeit(lambda: (eng.update_positions(pos), eng.compute(want), eng.fetch(want)))
        else:
            out[tag]["set_frames_ms"] = timeit(lambda: eng.set_frames([atoms]), 30)
stem = os.path.join(tempfile.mkdtemp(), "Ni")
nn.export(stem)
rng = np.random.RandomState(0)
for skin in (0.0, 0.5):
    calc = TensorAlloyCalculator(stem + ".json", skin=skin)
    a = atoms.copy()
    props = ["energy", "forces", "stress"]
    calc.calculate(a, props)
    ts = []
    for _ in range(60):
        a.positions = a.positions + rng.normal(0, 0.002, a.positions.shape)
        t = time.perf_counter()
        calc.calculate(a, props)
        f = calc.get_forces(a)
        ts.append(time.perf_counter() - t)
    out[f"calculator_skin{skin}"] = {"ms_per_call_median": float(np.median(ts) * 1e3),
                                     "atom_steps_per_s": len(a) / float(np.median(ts)),
                                     "lists": calc._engine.list_stats()}
print(json.dumps(out))
