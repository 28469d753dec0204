#!/usr/bin/env python3
"""Soak run on the GPU: many evaluations of perturbed structures per model family; every result
must be finite, translation-invariant (sum of forces = 0) and, on a small cell, equal to the CPU
oracle. Prints a summary line per family; exits non-zero on any violation."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from bench import ni_frame, ni_model
    from tensoralloy_amd import Atoms, Engine
    from tests.helpers import (fcc, make_eam, make_grap_nn, make_nn, oracle_eam_eval, oracle_eval,
                               oracle_grap_eval)
    from tests.test_gpu_sf import _alloy
    rng = np.random.RandomState(12345)
    bad = 0
    families = [
        ("sf 4000-atom Ni", ni_model(), ni_frame(611), None, 300),
        ("sf Ni-Mo small", make_nn(["Mo", "Ni"], 6.0, True, [16, 16]), _alloy(["Ni", "Ni", "Mo"], rep=(2, 2, 2)),
         oracle_eval, 200),
        ("grap Ni-Mo small", make_grap_nn(["Mo", "Ni"], 6.0, [16, 16], moment_tensors=[0, 1, 2, 3]),
         _alloy(["Ni", "Ni", "Mo"], rep=(2, 2, 2)), oracle_grap_eval, 200),
        ("grap 4000-atom Ni", make_grap_nn(["Ni"], 6.0, [64, 64], moment_tensors=[0, 1]), ni_frame(7), None, 200),
        ("eam/adp Ni-Mo small", make_eam(["Mo", "Ni"], 6.0, adp=True), _alloy(["Ni", "Mo"], rep=(2, 2, 2), a=3.7),
         oracle_eam_eval, 200),
        ("nn-eam Ni-Mo small", make_eam(["Mo", "Ni"], 6.0, potential=None), _alloy(["Ni", "Mo"], rep=(2, 2, 2), a=3.7),
         oracle_eam_eval, 200),
        ("nn-eam 4000-atom Ni", make_eam(["Ni"], 6.5, potential=None), ni_frame(9), None, 200),
        ("grap nn-filter Ni small", make_grap_nn(["Ni"], 6.0, [16, 16], "nn", moment_tensors=[0, 1, 2, 3]),
         fcc(rep=(2, 2, 2)), oracle_grap_eval, 200),
    ]
    for name, nn, atoms, oracle, n_calls in families:
        worst_f, worst_e, checks = 0.0, 0.0, 0
        with Engine(nn) as eng:
            pos0 = atoms.positions.copy()
            for k in range(n_calls):
                atoms.positions = pos0 + rng.normal(0.0, 0.03, pos0.shape) + rng.uniform(-30, 30, 3)
                r = eng.evaluate([atoms])[0]
                ok = np.isfinite(r["energy"]) and np.isfinite(r["forces"]).all() and np.isfinite(r["virial"]).all()
                net = np.abs(r["forces"].sum(axis=0)).max()
                if not ok or net > 1e-8:
                    print(f"  {name}: call {k}: finite={ok} net force {net:.3e}")
                    bad += 1
                if oracle is not None and k % 50 == 0:
                    o = oracle(nn, atoms)
                    worst_e = max(worst_e, abs(o["energy"] - r["energy"]))
                    worst_f = max(worst_f, float(np.abs(o["forces"] - r["forces"]).max()))
                    checks += 1
        print(f"{name}: {n_calls} evaluations, oracle checks {checks}, max |dE| {worst_e:.2e} eV, "
              f"max |dF| {worst_f:.2e} eV/A")
        if worst_e > 1e-6 or worst_f > 1e-5:
            bad += 1
    print("violations:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
