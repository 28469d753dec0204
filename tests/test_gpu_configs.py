"""GPU: every single-GPU configuration of BASELINE.json at its FULL size, against the CPU oracle
(tolerances of north_star: 1e-6 eV total energy, 1e-5 eV/A per force component).

  C1  snap_Ni_id11.extxyz, G2-only rc = 6.0, one hidden layer, energy only, through the calculator
  C2  4000-atom Ni, G2 + G4 rc = 6.5, MLP 2 x 64, energy + forces + virial
  C3  Ni4Mo (mp-11507) 7 x 7 x 8 = 3920 atoms, cross-element G2 / G4 (D = 20), MLP 2 x 128
  C4  EAM (zjw04) rc = 6.0 / 6.5 and ADP (zjw04 + mishinh) for the 4000-atom Ni frame
(C5, the 512-frame batch over 8 GPUs, needs 8 GPUs: bench.py `config5`; its 2-rank rehearsal is
tests/test_gpu_parallel.py.)
"""
import os

import numpy as np
import pytest

from tests.helpers import make_eam, make_nn, nimo_supercell, oracle_eam_eval, oracle_eval

pytestmark = pytest.mark.gpu

E_TOL, F_TOL, W_TOL = 1e-6, 1e-5, 1e-5
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _c_oracle(nn, atoms):
    from bench import host_cores
    from oracle import csf
    from tests.helpers import oracle_model
    m = oracle_model(nn)
    return csf.run(m, csf.prepare(m, atoms.get_chemical_symbols(), atoms.positions,
                                  np.asarray(atoms.get_cell(complete=True)), atoms.pbc),
                   True, host_cores())


def _check(res, ref, n):
    assert abs(res["energy"] - ref["energy"]) < E_TOL
    assert np.abs(res["atomic"] - ref["atomic"]).max() < E_TOL
    assert np.abs(res["forces"] - ref["forces"]).max() < F_TOL
    assert np.abs(res["virial"] - ref["virial"]).max() < W_TOL
    assert res["forces"].shape == (n, 3)


def test_c1_snap_ni_radial_energy_only_through_the_calculator(lib, tmp_path):
    from tensoralloy_amd import TensorAlloyCalculator
    from tensoralloy_amd.io import read_extxyz
    atoms = read_extxyz(os.path.join(GOLDEN, "snap_Ni_id11.extxyz"))[0]
    assert len(atoms) == 6 and all(atoms.pbc)
    # Defaults of the reference (utils.py:400-406): eta {0.05, 4, 20, 80}, omega {0}, cosine cutoff
    nn = make_nn(["Ni"], 6.0, False, [64], activation="softplus")
    assert nn.ndim() == 4
    stem = str(tmp_path / "Ni_c1")
    nn.export(stem)
    calc = TensorAlloyCalculator(stem + ".json")
    e = calc.get_potential_energy(atoms)
    ref = oracle_eval(nn, atoms, want_forces=False)
    assert abs(e - ref["energy"]) < E_TOL
    assert set(calc.results) == {"energy"}           # energy only: nothing else was asked for
    per_atom = calc.get_atomic(atoms)
    assert np.abs(per_atom - ref["atomic"]).max() < E_TOL
    # the same through the exact-list engine path (skin 0) and the debug descriptors
    calc.skin = 0.0
    calc.calculate(atoms, ["energy"], debug_mode=True)
    assert abs(calc.results["energy"] - ref["energy"]) < E_TOL
    assert np.abs(calc.results["descriptors"] - ref["descriptors"]).max() < 1e-10


def test_c2_ni_4000_full_size(lib):
    from bench import ni_frame, ni_model
    from tensoralloy_amd import Engine
    nn, atoms = ni_model(), ni_frame(611)
    with Engine(nn) as eng:
        res = eng.evaluate([atoms])[0]
        assert int(eng.info.n_atoms) == 4000 and bool(eng.info.nl_on_device)
    _check(res, _c_oracle(nn, atoms), 4000)


def test_c3_ni4mo_3920_full_size(lib):
    from tensoralloy_amd import Engine
    nn = make_nn(["Ni", "Mo"], 6.5, True, [128, 128])
    assert nn.ndim() == 20
    with Engine(nn) as eng:
        # the perfect lattice pins the structure itself: SURVEY 8(d) computed 86 neighbours per
        # atom at rc = 6.5, P = 337 120, T = 14 327 600 for this cell
        info = eng.set_frames([nimo_supercell(jitter=0.0)])
        assert (int(info.n_atoms), int(info.n_pairs), int(info.n_triples)) == (3920, 337120, 14327600)
        atoms = nimo_supercell()
        res = eng.evaluate([atoms])[0]
    assert sorted(set(atoms.get_chemical_symbols())) == ["Mo", "Ni"]
    _check(res, _c_oracle(nn, atoms), 3920)


@pytest.mark.parametrize("rc,adp", [(6.0, False), (6.5, False), (6.5, True)])
def test_c4_eam_adp_4000_full_size(lib, rc, adp):
    from bench import ni_frame
    from tensoralloy_amd import Engine
    nn, atoms = make_eam(["Ni"], rc, adp=adp), ni_frame(611)
    with Engine(nn) as eng:
        res = eng.evaluate([atoms])[0]
    _check(res, oracle_eam_eval(nn, atoms), 4000)
