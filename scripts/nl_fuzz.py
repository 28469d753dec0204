#!/usr/bin/env python3
"""Neighbour-list fuzz on the GPU: random triclinic cells (thick, thin, mixed periodicity, atoms outside
the box), single frames and uneven batches, through the one-pass builder; the (i, j, S) set of every
frame must equal the oracle's, the order inside a centre must be the key order, and energies must not
depend on the builder. Prints a summary; exits non-zero on any violation."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(n_cases=200):
    from oracle import neighbors as onl
    from tensoralloy_amd import Atoms, Engine
    from tests.helpers import make_nn
    from tests.test_gpu_fuzz import _random_frame
    rng = np.random.RandomState(2026)
    bad = 0
    els = ["Mo", "Ni"]
    nn = make_nn(els, 5.0, False, [8])
    sp_of = {"Mo": 0, "Ni": 1}

    def trip(i, j, S):
        a = np.concatenate([np.asarray(i)[:, None], np.asarray(j)[:, None], np.asarray(S).reshape(-1, 3)], axis=1)
        return a[np.lexsort(a.T[::-1])]
    with Engine(nn) as eng:
        for case in range(n_cases):
            nf = 1 if case % 3 else int(rng.randint(2, 6))
            frames = []
            for _ in range(nf):
                a = _random_frame(rng, els)
                if rng.rand() < 0.3:          # thin along one axis
                    cell = np.asarray(a.get_cell(complete=True)).copy()
                    ax = rng.randint(3)
                    cell[ax] *= rng.uniform(0.25, 0.6)
                    a = Atoms(symbols=a.get_chemical_symbols(), positions=a.positions, cell=cell, pbc=a.pbc)
                frames.append(a)
            try:
                eng.set_frames(frames)
            except Exception as exc:  # noqa: BLE001
                print("case", case, "set_frames raised", exc)
                bad += 1
                continue
            i, j, S = eng.pairs()
            S = np.asarray(S).reshape(-1, 3)
            ref, off = [], 0
            sp = []
            for a in frames:
                oi, oj, oS = onl.neighbor_list(a.positions, np.asarray(a.get_cell(complete=True)), a.pbc, 5.0)
                ref.append(trip(oi + off, oj + off, oS))
                off += len(a)
                sp += [sp_of[s] for s in a.get_chemical_symbols()]
            ref = np.concatenate(ref) if ref else np.zeros((0, 5), int)
            ref = ref[np.lexsort(ref.T[::-1])]
            got = trip(i, j, S)
            if got.shape != ref.shape or not np.array_equal(got, ref):
                print("case", case, "pair set differs:", got.shape, ref.shape)
                bad += 1
                continue
            sp = np.asarray(sp)
            if bool(eng.info.nl_on_device):
                order = np.lexsort((S[:, 2], S[:, 1], S[:, 0], j, sp[j], i))
                if not np.array_equal(order, np.arange(len(i))):
                    print("case", case, "not in key order")
                    bad += 1
    print(f"{n_cases} cases, violations: {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 200))
