#!/usr/bin/env python3
"""GRAP (the reference's default production descriptor, defaults.toml:131-155) on the 4000-atom Ni
frame: pexp, 16 filters, moments 0..3, new mode, rc = 6.0, MLP 2 x 64. Prints one JSON line."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--frames", type=int, default=1)
    ap.add_argument("--moments", type=int, default=3)
    args = ap.parse_args()
    from bench import ni_frame
    from tensoralloy_amd import AtomicNN, Engine, UniversalTransformer, _lib
    from tensoralloy_amd.grap import GenericRadialAtomicPotential
    rl = [1.0 + 0.2 * k for k in range(16)]
    pl = [5.0 - 0.25 * k for k in range(16)]
    gd = GenericRadialAtomicPotential(["Ni"], "pexp", {"rl": rl, "pl": pl},
                                      moment_tensors=list(range(args.moments + 1)), legacy_mode=False)
    nn = AtomicNN(["Ni"], gd, hidden_sizes=[64, 64], activation="softplus", minmax_scale=False,
                  export_properties=("energy", "forces", "stress"))
    nn.attach_transformer(UniversalTransformer(["Ni"], rcut=6.0))
    nn.initialize(seed=611)
    want = _lib.TA_WANT_ENERGY | _lib.TA_WANT_FORCES | _lib.TA_WANT_VIRIAL | _lib.TA_WANT_ATOMIC
    frames = [ni_frame(611 + k) for k in range(args.frames)]
    with Engine(nn) as eng:
        info = eng.set_frames(frames)
        total_ms, slots = eng.time_compute(want, 3, args.steps)
        n = int(info.n_atoms)
        print(json.dumps({"atoms": n, "pairs": int(info.n_pairs), "D": nn.ndim(),
                          "ms_per_eval": total_ms / args.steps,
                          "atom_steps_per_s": n / (total_ms / args.steps) * 1e3,
                          "kernel_ms": {k: v for k, v in slots.items() if v > 0}}))


if __name__ == "__main__":
    main()
