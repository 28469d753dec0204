#!/usr/bin/env python3
"""One BASELINE config on the resident batch, N evaluations (for rocprofv3: scripts/profile_configs.sh).
  python scripts/run_config.py <sf|nimo|eam|adp|grap|nn_eam> [frames] [steps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import ni_frame
from tensoralloy_amd import Engine, _lib
from tests.helpers import make_eam, make_grap_nn

kind = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
want = _lib.TA_WANT_ENERGY | _lib.TA_WANT_FORCES | _lib.TA_WANT_VIRIAL | _lib.TA_WANT_ATOMIC
if kind == "sf":
    from bench import ni_model
    nn = ni_model()
elif kind == "nimo":  # BASELINE config 3: Ni4Mo cell, 3920 atoms, G2 + G4, 2 x 128 MLP
    from tensoralloy_amd import AtomicNN, SymmetryFunction, UniversalTransformer
    from tests.helpers import nimo_supercell
    nn = AtomicNN(["Ni", "Mo"], SymmetryFunction(["Ni", "Mo"]), hidden_sizes=[128, 128], activation="softplus",
                  minmax_scale=False, export_properties=("energy", "forces", "stress"))
    nn.attach_transformer(UniversalTransformer(["Ni", "Mo"], rcut=6.5, angular=True))
    nn.initialize(seed=611)
elif kind == "eam":
    nn = make_eam(["Ni"], 6.5)
elif kind == "adp":
    nn = make_eam(["Ni"], 6.5, adp=True)
elif kind == "nn_eam":
    nn = make_eam(["Ni"], 6.5, potential=None)
else:
    rl = [1.0 + 0.2 * k for k in range(16)]
    pl = [5.0 - 0.25 * k for k in range(16)]
    nn = make_grap_nn(["Ni"], 6.0, [64, 64], "pexp", {"rl": rl, "pl": pl}, moment_tensors=[0, 1, 2, 3])
with Engine(nn) as eng:
    if kind == "nimo":
        frame_list = [nimo_supercell("Ni4Mo_mp-11507", rep=(7, 7, 8), jitter=0.05, seed=611 + k) for k in range(frames)]
    else:
        frame_list = [ni_frame(611 + k) for k in range(frames)]
    info = eng.set_frames(frame_list)
    total_ms, slots = eng.time_compute(want, 3, steps)
    P, N = int(info.n_pairs), int(info.n_atoms)
    out = {"config": kind, "frames": frames, "atoms": N, "pairs": P, "ms_per_eval": total_ms / steps,
           "us_per_frame": total_ms / steps / frames * 1e3, "atom_steps_per_s": N / (total_ms / steps) * 1e3,
           # pair records read twice (density pass, force pass) + per-atom rho, F'
           "algorithmic_bytes_per_eval": 2 * 32 * P + N * 8 * 6,
           "kernel_ms": {k: v for k, v in slots.items() if v > 0}}
    out["algorithmic_GBps"] = out["algorithmic_bytes_per_eval"] / (out["ms_per_eval"] * 1e-3) / 1e9
print(json.dumps(out))
