#!/usr/bin/env python3
"""cProfile of the calculator's MD step (4000-atom Ni, skin 0.5)."""
import cProfile, os, pstats, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import ni_frame, ni_model
from tensoralloy_amd import TensorAlloyCalculator
nn = ni_model()
stem = os.path.join(tempfile.mkdtemp(), "Ni")
nn.export(stem)
calc = TensorAlloyCalculator(stem + ".json")
a = ni_frame(611)
props = ["energy", "forces", "stress"]
rng = np.random.RandomState(0)
calc.calculate(a, props)
def loop(n):
    for _ in range(n):
        a.positions = a.positions + rng.normal(0, 0.002, a.positions.shape)
        calc.calculate(a, props)
        calc.get_forces(a)
loop(20)
pr = cProfile.Profile(); pr.enable(); loop(200); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
