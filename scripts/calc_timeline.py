#!/usr/bin/env python3
"""Wall-clock split of the calculator's MD step (4000-atom Ni, skin 0.5): host time inside
update_positions / compute / fetch and what is left for the Python around them."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import ni_frame, ni_model
from tensoralloy_amd import TensorAlloyCalculator
from tensoralloy_amd.engine import Engine

acc = {"update_positions": 0.0, "compute": 0.0, "fetch": 0.0}
for name in list(acc):
    orig = getattr(Engine, name)
    def wrap(self, *a, _o=orig, _n=name, **k):
        t = time.perf_counter()
        r = _o(self, *a, **k)
        acc[_n] += time.perf_counter() - t
        return r
    setattr(Engine, name, wrap)
nn = ni_model()
stem = os.path.join(tempfile.mkdtemp(), "Ni")
nn.export(stem)
calc = TensorAlloyCalculator(stem + ".json")
a = ni_frame(611)
rng = np.random.RandomState(0)
props = ["energy", "forces", "stress"]
calc.calculate(a, props)
n = 300
for k in range(20 + n):
    if k == 20:
        for key in acc:
            acc[key] = 0.0
        total = 0.0
    a.positions = a.positions + rng.normal(0, 0.002, a.positions.shape)
    t = time.perf_counter()
    calc.calculate(a, props)
    calc.get_forces(a)
    if k >= 20:
        total += time.perf_counter() - t
out = {k: v / n * 1e6 for k, v in acc.items()}
out["total_us"] = total / n * 1e6
out["python_around_us"] = out["total_us"] - sum(acc.values()) / n * 1e6
print(json.dumps(out))
