#!/usr/bin/env python3
"""
Per-call latency of the drop-in `TensorAlloyCalculator` (the reference's user-facing entry,
calculator.py:335-370) on the 4000-atom Ni frame: what one MD step costs end to end, with new
positions every call. `--skin 0` builds a new neighbour list every call as the reference does; the
default keeps the list under a 0.5 A Verlet skin and extracts the exact list of the step on the
device. Prints one JSON line.
"""
import argparse
import cProfile
import json
import os
import pstats
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=50)
    ap.add_argument("--rep", type=int, default=10)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--skin", type=float, default=0.5, help="Verlet skin of the calculator (0 = new list every call)")
    args = ap.parse_args()
    from bench import ni_frame, ni_model
    from tensoralloy_amd import TensorAlloyCalculator

    nn = ni_model()
    stem = os.path.join(tempfile.mkdtemp(), "Ni")
    nn.export(stem)
    calc = TensorAlloyCalculator(stem + ".json", skin=args.skin)
    atoms = ni_frame(611, rep=args.rep)
    rng = np.random.RandomState(0)
    props = ["energy", "forces", "stress"]
    calc.calculate(atoms, props)
    times = []
    prof = cProfile.Profile() if args.profile else None
    for _ in range(args.calls):
        atoms.positions = atoms.positions + rng.normal(0, 0.002, atoms.positions.shape)
        t0 = time.perf_counter()
        if prof:
            prof.enable()
        calc.calculate(atoms, props)
        f = calc.get_forces(atoms)
        if prof:
            prof.disable()
        times.append(time.perf_counter() - t0)
    t = np.array(times)
    info = calc._engine.info
    out = {"calls": args.calls, "atoms": len(atoms), "ms_per_call_median": float(np.median(t) * 1e3),
           "ms_per_call_min": float(t.min() * 1e3), "atom_steps_per_s": float(len(atoms) / np.median(t)),
           "set_frames_c_abi_ms": info.set_frames_ms, "neighbor_list_ms": info.nl_ms,
           "neighbor_list_on_device": bool(info.nl_on_device), "forces_norm": float(np.abs(f).sum()),
           "skin_A": args.skin, "lists_built_reused": list(calc._engine.list_stats())}
    print(json.dumps(out))
    if prof:
        pstats.Stats(prof).sort_stats("cumulative").print_stats(25)


if __name__ == "__main__":
    main()
