"""Five-element (Al-Co-Cu-Fe-Ni) 4000-atom frame, default G2+G4 grid, D = 80: second-generation kernels
with 5 partner species against the first-generation ones (TA_FORCE_V1=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import ni_frame
from tensoralloy_amd import Atoms, AtomicNN, Engine, SymmetryFunction, UniversalTransformer, _lib
els = ["Al", "Co", "Cu", "Fe", "Ni"]
base = ni_frame(611)
atoms = Atoms(symbols=[els[k % 5] for k in range(len(base))], positions=base.positions, cell=np.asarray(base.get_cell()), pbc=True)
nn = AtomicNN(els, SymmetryFunction(els), hidden_sizes=[64, 64], minmax_scale=False, export_properties=("energy", "forces", "stress"))
nn.attach_transformer(UniversalTransformer(els, rcut=6.5, angular=True))
nn.initialize(seed=1)
want = _lib.TA_WANT_ENERGY | _lib.TA_WANT_FORCES | _lib.TA_WANT_VIRIAL | _lib.TA_WANT_ATOMIC
with Engine(nn) as eng:
    eng.set_frames([atoms])
    ms, slots = eng.time_compute(want, 3, 20)
    print(os.environ.get("TA_FORCE_V1", "v2"), "D", nn.ndim(), "ms", ms / 20, "M atom-steps/s", len(atoms) / (ms / 20) * 1e-3, {k: round(v, 4) for k, v in slots.items() if v > 0})
