#!/usr/bin/env python3
"""MD-path soak on the GPU: a trajectory of random moves (small ones that keep the Verlet list, now and
then a jump that rebuilds it, a cell change) through `Engine.step(view=True)`, every step compared with an
engine that builds an exact list from scratch. Prints a summary; exits non-zero on any violation."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(steps=150):
    from tensoralloy_amd import Atoms, Engine, _lib
    from tests.helpers import fcc, make_eam, make_nn
    want = _lib.TA_WANT_ENERGY | _lib.TA_WANT_FORCES | _lib.TA_WANT_VIRIAL | _lib.TA_WANT_ATOMIC
    rng = np.random.RandomState(7)
    bad = 0
    models = [("sf", make_nn(["Mo", "Ni"], 6.0, True, [16, 16])), ("eam", make_eam(["Mo", "Ni"], 6.0)),
              ("adp", make_eam(["Mo", "Ni"], 6.0, adp=True)), ("nn-eam", make_eam(["Mo", "Ni"], 5.8, potential=None))]
    for name, nn in models:
        base = fcc(rep=(4, 4, 4), seed=3, jitter=0.05)
        syms = ["Mo" if k % 4 == 0 else "Ni" for k in range(len(base))]
        atoms = Atoms(symbols=syms, positions=base.positions, cell=base.get_cell(complete=True), pbc=True)
        worst = [0.0, 0.0, 0.0]
        with Engine(nn) as eng, Engine(nn) as exact:
            eng.set_skin(0.5)
            eng.set_frames([atoms])
            pos = atoms.positions.copy()
            cell = np.asarray(atoms.get_cell(complete=True)).copy()
            for step in range(steps):
                pos = pos + rng.normal(0, 0.02, pos.shape)
                cells = None
                if step % 37 == 36:
                    pos[rng.randint(len(pos))] += rng.normal(0, 0.3, 3)        # a jump: rebuild
                if step % 53 == 52:
                    cell = cell * (1.0 + 0.002 * rng.normal())                  # new cell: rebuild
                    cells = cell[None]
                got = eng.step(pos, want, cells=cells, view=True)
                ref = exact.evaluate([Atoms(symbols=syms, positions=pos, cell=cell, pbc=True)])[0]
                dE = abs(float(got["energy"][0]) - ref["energy"])
                dF = float(np.abs(got["forces"] - ref["forces"]).max())
                dW = float(np.abs(got["virial"][0] - ref["virial"]).max())
                worst = [max(worst[0], dE), max(worst[1], dF), max(worst[2], dW)]
                if not (dE < 1e-8 and dF < 1e-9 and dW < 1e-7):
                    print(name, "step", step, "dE", dE, "dF", dF, "dW", dW)
                    bad += 1
            builds, reuses = eng.list_stats()
        print(f"{name}: {steps} steps, lists built / reused {builds} / {reuses}, max |dE| {worst[0]:.2e} "
              f"|dF| {worst[1]:.2e} |dW| {worst[2]:.2e}")
    print("violations:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 150))
