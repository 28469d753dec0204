#!/usr/bin/env python3
"""MLP kernel time (HIP-event slot) on the 4000-atom Ni frame for a few network shapes and
activations; TA_MLP_TILE_KERNEL=1 selects the 16-row tile kernel everywhere."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import ni_frame  # noqa: E402
from tensoralloy_amd import Engine, _lib  # noqa: E402
from tests.helpers import make_nn  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1
atoms = [ni_frame(611 + f, rep=10) for f in range(frames)]
want = _lib.TA_WANT_ENERGY | _lib.TA_WANT_FORCES | _lib.TA_WANT_VIRIAL
for hidden, act in (([64, 64], "softplus"), ([64, 64], "relu"), ([64, 32], "softplus"), ([16, 16], "softplus"),
                    ([64], "softplus")):
    nn = make_nn(["Ni"], 6.5, False, hidden, activation=act)
    with Engine(nn) as eng:
        eng.set_frames(atoms)
        ms, slots = eng.time_compute(want, 5, 30)
        print(json.dumps({"hidden": hidden, "act": act, "frames": frames, "mlp_us": slots["mlp"] * 1e3,
                          "step_us": ms * 1e3, "kernel": os.environ.get("TA_MLP_TILE_KERNEL", "wave")}))
