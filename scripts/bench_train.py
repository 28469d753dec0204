#!/usr/bin/env python3
"""One data-parallel training step on resident frames: `--loss energy` = MLP forward
on cached descriptors + weight gradient on the GPU; `--loss full` = energy + forces + stress loss
with the analytic second-order pass (`ta_loss_gradient`; `--fd` = the central-difference path it
replaced); `--loss constants` = the constants of a Zjw04 EAM model (`ta_constant_gradient`). Then
gradient all-reduce (when launched under torch.distributed), Adam on the host, upload. One JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--rep", type=int, default=10)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--loss", choices=("energy", "full", "constants"), default="energy")
    ap.add_argument("--fd", action="store_true", help="full loss through the central difference of dE/dtheta")
    args = ap.parse_args()
    from bench import ni_frame, ni_model
    from tensoralloy_amd import Engine
    from tensoralloy_amd.train import EnergyTrainer, Trainer, flatten_weights
    frames = [ni_frame(611 + k, rep=args.rep) for k in range(args.frames)]
    n_atoms = int(sum(len(a) for a in frames))
    if args.loss == "energy":
        nn = ni_model()
        labels = np.array([-4.45 * len(a) for a in frames]) + np.random.RandomState(0).randn(len(frames))
        tr = EnergyTrainer(nn, frames, labels, device=0, learning_rate=1e-3)
    else:
        if args.loss == "constants":
            from tests.helpers import make_eam
            teacher, nn = make_eam(["Ni"], 6.5), make_eam(["Ni"], 6.5)
            c = nn.constants()
            c[1] *= 1.03
            c[6] *= 0.98
            nn.set_constants(c)
        else:
            teacher, nn = ni_model(), ni_model()
            rng = np.random.RandomState(1)
            for el in nn.elements:      # the student starts next to the teacher
                nn.weights[el] = [(w + 0.01 * rng.randn(*np.shape(w)), b) for w, b in nn.weights[el]]
        with Engine(teacher) as eng:
            ref = eng.evaluate(frames)
        tr = Trainer(nn, frames, [r["energy"] for r in ref], [r["forces"] for r in ref], [r["stress"] for r in ref],
                     device=0, learning_rate=1e-4, analytic=False if args.fd else None)
    for _ in range(3):
        tr.step()
    tr.engine.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.step()
    tr.engine.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    first, last = tr.history[0], tr.history[-1]
    print(json.dumps({"loss": args.loss + ("/central-difference" if args.fd else ""), "frames": len(frames),
                      "atoms": n_atoms, "parameters": int(len(tr.theta)), "ms_per_training_step": dt * 1e3,
                      "atom_steps_per_s": n_atoms / dt, "loss_first": first, "loss_last": last}))
    tr.close()


if __name__ == "__main__":
    main()
