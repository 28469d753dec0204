#!/usr/bin/env python3
"""One data-parallel training step of the energy loss on resident frames (first
part): MLP forward on cached descriptors + weight gradient on the GPU, gradient all-reduce (when
launched under torch.distributed), Adam on the host, weight upload. Prints one JSON line."""
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
    args = ap.parse_args()
    from bench import ni_frame, ni_model
    from tensoralloy_amd.train import EnergyTrainer
    nn = ni_model()
    frames = [ni_frame(611 + k, rep=args.rep) for k in range(args.frames)]
    labels = np.array([-4.45 * len(a) for a in frames]) + np.random.RandomState(0).randn(len(frames))
    tr = EnergyTrainer(nn, frames, labels, device=0, learning_rate=1e-3)
    for _ in range(5):
        tr.step()
    tr.engine.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, mae = tr.step()
    tr.engine.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    n_atoms = int(sum(len(a) for a in frames))
    print(json.dumps({"frames": len(frames), "atoms": n_atoms, "parameters": int(len(tr.theta)),
                      "ms_per_training_step": dt * 1e3, "atom_steps_per_s": n_atoms / dt,
                      "loss_first": tr.history[0], "loss_last": tr.history[-1]}))
    tr.close()


if __name__ == "__main__":
    main()
