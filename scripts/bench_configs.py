#!/usr/bin/env python3
"""
Parity + throughput of every BASELINE.json config on one MI355X (not the driver's bench line:
that is bench.py). Prints one JSON object; run through gpurun and keep the output under profiles/.

  C1  snap_Ni_id11.extxyz, G2-only rc=6.0, 1 hidden layer, energy only
  C2  4000-atom Ni, G2+G4 rc=6.5, 2x64 MLP, E+F+virial                (= bench.py)
  C3  Ni-Mo binary alloy (3920 atoms, 8:2), cross-element G2/G4, 2x128 MLP, E+F+virial
  C4  EAM (zjw04) and ADP (zjw04 + mishinh) for 4000-atom Ni, E+F+virial
  C5  batch of independent 4000-atom frames on one GPU (the per-GPU share of the 512-frame job)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bench import ni_frame, ni_model, oracle_sfmodel, host_cores  # noqa: E402
from tensoralloy_amd import (Atoms, AtomicNN, Engine, SymmetryFunction, UniversalTransformer,  # noqa: E402
                             _lib)
from tensoralloy_amd.eam import AdpNN, EamAlloyNN  # noqa: E402
from tensoralloy_amd.io import read_extxyz  # noqa: E402

WANT = _lib.TA_WANT_ENERGY | _lib.TA_WANT_FORCES | _lib.TA_WANT_VIRIAL | _lib.TA_WANT_ATOMIC


def timeit(eng, want, steps=30, warmup=3):
    total_ms, slots = eng.time_compute(want, warmup, steps)
    return total_ms / steps, {k: v for k, v in slots.items() if v > 0}


def sf_parity(nn, atoms, res):
    from oracle import csf
    m = oracle_sfmodel(nn)
    if nn._minmax_scale:
        m.minmax = nn.minmax
    prep = csf.prepare(m, atoms.get_chemical_symbols(), atoms.positions,
                       np.asarray(atoms.get_cell(complete=True)), atoms.pbc)
    t0 = time.perf_counter()
    ref = csf.run(m, prep, True, host_cores(), csf.make_cmodel(m))
    dt = time.perf_counter() - t0
    out = {"dE_eV": abs(ref["energy"] - res["energy"]), "cpu_oracle_s": dt}
    if "forces" in res:
        out["dF_max"] = float(np.abs(ref["forces"] - res["forces"]).max())
        out["dW_max"] = float(np.abs(ref["virial"] - res["virial"]).max())
    return out


def nimo_frame(seed=611):
    """BASELINE config 3: the reference's Ni4Mo_mp-11507 conventional cell (I4/m, 8 Ni + 2 Mo),
    7 x 7 x 8 = 3920 atoms, N(0, 0.05 A) jitter."""
    from tests.helpers import nimo_supercell
    return nimo_supercell("Ni4Mo_mp-11507", rep=(7, 7, 8), jitter=0.05, seed=seed)


def bench_nn_eam(steps=50):
    from tests.helpers import make_eam, oracle_eam_eval
    out = {}
    atoms = ni_frame(611)
    for tag, adp, rc in (("C4_nn_eam", False, 6.0), ("C4_nn_eam", False, 6.5), ("C4_nn_adp", True, 6.5)):
        nn = make_eam(["Ni"], rc, adp=adp, potential=None)
        with Engine(nn) as eng:
            r = eng.evaluate([atoms])[0]
            ms, slots = timeit(eng, WANT, steps=steps)
            o = oracle_eam_eval(nn, atoms)
            out[f"{tag}_rc{rc}"] = {
                "atoms": len(atoms), "pairs": int(eng.info.n_pairs), "ms_per_eval": ms,
                "atom_steps_per_s": len(atoms) / ms * 1e3, "kernel_ms": slots,
                "hidden_sizes": nn.hidden_sizes["Ni"]["rho"],
                "parity": {"dE_eV": abs(o["energy"] - r["energy"]),
                           "dF_max": float(np.abs(o["forces"] - r["forces"]).max()),
                           "dW_max": float(np.abs(o["virial"] - r["virial"]).max())}}
    return out


def bench_grap_nn(steps=50):
    """GRAP with the `nn` filter network of defaults.toml `[nn.atomic.grap.nn]` (softplus, hidden
    32-32-32 with ResNet skips, 16 filters), moments 0..3, rc 6.0, MLP 2 x 64."""
    from tests.helpers import make_grap_nn, oracle_grap_eval
    atoms = ni_frame(611)
    nn = make_grap_nn(["Ni"], 6.0, [64, 64], "nn", moment_tensors=[0, 1, 2, 3])
    with Engine(nn) as eng:
        r = eng.evaluate([atoms])[0]
        ms, slots = timeit(eng, WANT, steps=steps)
        o = oracle_grap_eval(nn, atoms)
        return {"N1_grap_nn_Ni": {
            "atoms": len(atoms), "pairs": int(eng.info.n_pairs), "D": nn.ndim(), "ms_per_eval": ms,
            "atom_steps_per_s": len(atoms) / ms * 1e3, "kernel_ms": slots,
            "parity": {"dE_eV": abs(o["energy"] - r["energy"]),
                       "dF_max": float(np.abs(o["forces"] - r["forces"]).max()),
                       "dW_max": float(np.abs(o["virial"] - r["virial"]).max())}}}


def main():
    _lib.build()
    if "--nn-eam" in sys.argv:
        print(json.dumps(bench_nn_eam(), indent=1))
        return
    if "--grap-nn" in sys.argv:
        print(json.dumps(bench_grap_nn(), indent=1))
        return
    out = {}

    # ---- C1
    atoms = read_extxyz(os.path.join(ROOT, "tests", "golden", "snap_Ni_id11.extxyz"))[0]
    clf = UniversalTransformer(["Ni"], rcut=6.0, angular=False)
    nn = AtomicNN(["Ni"], SymmetryFunction(["Ni"]), hidden_sizes=[64], activation="softplus",
                  minmax_scale=False, export_properties=("energy",))
    nn.attach_transformer(clf)
    nn.initialize(seed=611)
    with Engine(nn) as eng:
        want = _lib.TA_WANT_ENERGY | _lib.TA_WANT_ATOMIC
        r = eng.evaluate([atoms], want=want)[0]
        ms, slots = timeit(eng, want, steps=100)
        out["C1"] = {"atoms": len(atoms), "pairs": int(eng.info.n_pairs), "ms_per_eval": ms,
                     "atom_steps_per_s": len(atoms) / ms * 1e3, "parity": sf_parity(nn, atoms, r),
                     "note": "6-atom cell: launch-latency bound by construction"}

    # ---- C2 / C5
    nn = ni_model()
    with Engine(nn) as eng:
        atoms = ni_frame(611)
        r = eng.evaluate([atoms])[0]
        ms, slots = timeit(eng, WANT)
        out["C2"] = {"atoms": len(atoms), "pairs": int(eng.info.n_pairs),
                     "triples": int(eng.info.n_triples), "ms_per_eval": ms,
                     "atom_steps_per_s": len(atoms) / ms * 1e3, "kernel_ms": slots,
                     "parity": sf_parity(nn, atoms, r)}
        frames = [ni_frame(611 + k) for k in range(64)]
        eng.set_frames(frames)
        ms, slots = timeit(eng, WANT, steps=5, warmup=1)
        out["C5_per_gpu_share"] = {"frames": 64, "atoms": 64 * 4000, "ms_per_batch": ms,
                                   "atom_steps_per_s": 64 * 4000 / ms * 1e3, "kernel_ms": slots}

    # ---- C3
    atoms = nimo_frame()
    clf = UniversalTransformer(["Ni", "Mo"], rcut=6.5, angular=True)
    nn = AtomicNN(["Ni", "Mo"], SymmetryFunction(["Ni", "Mo"]), hidden_sizes=[128, 128],
                  activation="softplus", minmax_scale=False,
                  export_properties=("energy", "forces", "stress"))
    nn.attach_transformer(clf)
    nn.initialize(seed=611)
    with Engine(nn) as eng:
        r = eng.evaluate([atoms])[0]
        ms, slots = timeit(eng, WANT)
        out["C3"] = {"atoms": len(atoms), "pairs": int(eng.info.n_pairs),
                     "triples": int(eng.info.n_triples), "D": nn.ndim(), "ms_per_eval": ms,
                     "atom_steps_per_s": len(atoms) / ms * 1e3, "kernel_ms": slots,
                     "parity": sf_parity(nn, atoms, r)}

    # ---- C4
    from oracle.eam import EamModel, evaluate as eam_eval
    atoms = ni_frame(611)
    for tag, cls, pots in (("C4_eam", EamAlloyNN, "zjw04"),
                           ("C4_adp", AdpNN, {"Ni": {"rho": "zjw04", "embed": "zjw04"},
                                              "NiNi": {"phi": "zjw04", "dipole": "mishinh",
                                                       "quadrupole": "mishinh"}})):
        for rc in (6.0, 6.5):
            nn = cls(["Ni"], custom_potentials=pots)
            nn.attach_transformer(UniversalTransformer(["Ni"], rcut=rc))
            with Engine(nn) as eng:
                r = eng.evaluate([atoms])[0]
                ms, slots = timeit(eng, WANT, steps=100)
                adp = {"NiNi": nn.pair_parameters("NiNi")} if cls is AdpNN else None
                o = eam_eval(EamModel(["Ni"], rc, adp=adp), atoms.get_chemical_symbols(),
                             atoms.positions, np.asarray(atoms.get_cell()), atoms.pbc)
                out[f"{tag}_rc{rc}"] = {
                    "atoms": len(atoms), "pairs": int(eng.info.n_pairs), "ms_per_eval": ms,
                    "atom_steps_per_s": len(atoms) / ms * 1e3, "kernel_ms": slots,
                    "parity": {"dE_eV": abs(o["energy"] - r["energy"]),
                               "dF_max": float(np.abs(o["forces"] - r["forces"]).max()),
                               "dW_max": float(np.abs(o["virial"] - r["virial"]).max())}}
    # nn-EAM: the reference's default potentials (rho, phi, embed = 1 -> 64 -> 32 -> 1 networks,
    # alloy.py:110-112, Defaults.hidden_sizes)
    out.update(bench_nn_eam())
    out.update(bench_grap_nn())
    # ---- N1: the reference's default production descriptor (io/input/defaults.toml:131-155):
    # GRAP, pexp, 16 filters, moments 0..3, new mode, cosine cutoff, rc = 6.0; MLP 2 x 64
    from oracle import grap as ograp
    from tensoralloy_amd.grap import GenericRadialAtomicPotential
    rl = [1.0 + 0.2 * k for k in range(16)]
    pl = [5.0 - 0.25 * k for k in range(16)]
    for tag, els, atoms in (("N1_grap_Ni", ["Ni"], ni_frame(611)), ("N1_grap_NiMo", ["Mo", "Ni"], None)):
        if atoms is None:
            base = ni_frame(611)
            atoms = Atoms(symbols=["Mo" if k % 5 == 0 else "Ni" for k in range(len(base))],
                          positions=base.positions, cell=np.asarray(base.get_cell()), pbc=True)
        gd = GenericRadialAtomicPotential(els, "pexp", {"rl": rl, "pl": pl}, moment_tensors=[0, 1, 2, 3],
                                          legacy_mode=False)
        nn = AtomicNN(els, gd, hidden_sizes=[64, 64], activation="softplus", minmax_scale=False,
                      export_properties=("energy", "forces", "stress"))
        nn.attach_transformer(UniversalTransformer(els, rcut=6.0))
        nn.initialize(seed=611)
        with Engine(nn) as eng:
            r = eng.evaluate([atoms])[0]
            ms, slots = timeit(eng, WANT, steps=100)
            d = gd.as_dict()
            om = ograp.GrapModel(els, 6.0, algorithm="pexp", parameters=d["parameters"],
                                 moment_tensors=d["moment_tensors"], legacy_mode=False,
                                 weights=nn.weights, activation="softplus")
            t0 = time.perf_counter()
            o = ograp.evaluate(om, atoms.get_chemical_symbols(), atoms.positions,
                               np.asarray(atoms.get_cell()), atoms.pbc)
            out[tag] = {"atoms": len(atoms), "pairs": int(eng.info.n_pairs), "D": nn.ndim(),
                        "ms_per_eval": ms, "atom_steps_per_s": len(atoms) / ms * 1e3, "kernel_ms": slots,
                        "parity": {"dE_eV": abs(o["energy"] - r["energy"]),
                                   "dF_max": float(np.abs(o["forces"] - r["forces"]).max()),
                                   "dW_max": float(np.abs(o["virial"] - r["virial"]).max()),
                                   "cpu_oracle_s": time.perf_counter() - t0}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
