"""The drop-in boundary on a real GPU: TensorAlloyCalculator behaves like the
reference's ASE calculator (tensoralloy/calculator.py:31-383)."""
import numpy as np
import pytest

from tests.helpers import fcc, pd3o2, make_nn, make_eam, oracle_eval, oracle_eam_eval

pytestmark = pytest.mark.gpu


def test_calculator_pd3o2_roundtrip(lib, tmp_path):
    from tensoralloy_amd import TensorAlloyCalculator
    nn = make_nn(["Pd", "O"], 6.5, True, [32, 32], minmax=True, static_energy={"Pd": -1.5, "O": -0.5})
    path = nn.export(str(tmp_path / "pd3o2.pb"))
    calc = TensorAlloyCalculator(path)
    assert calc.elements == ["O", "Pd"]
    assert set(calc.predict_properties) >= {"energy", "forces", "stress", "energy/atom"}
    atoms = pd3o2()
    o = oracle_eval(nn, atoms)
    e = calc.get_potential_energy(atoms)
    f = calc.get_forces(atoms)          # caller's (ASE) order
    s = calc.get_stress(atoms)
    assert abs(e - o["energy"]) < 1e-6
    assert np.abs(f - o["forces"]).max() < 1e-5
    assert np.abs(s - o["stress_voigt"]).max() < 1e-8
    assert np.abs(calc.get_atomic(atoms) - o["atomic"]).max() < 1e-6
    assert abs(calc.get_total_pressure(atoms) - o["total_pressure"]) < 1e-5
    assert calc.get_stress(atoms, voigt=False).shape == (3, 3)
    # results are GSL-ordered (element-sorted: O, O, Pd, Pd, Pd), virtual row stripped
    calc.calculate(atoms, properties=["forces", "energy/atom"])
    assert np.abs(calc.results["forces"] - o["forces"][[3, 4, 0, 1, 2]]).max() < 1e-5
    assert np.abs(calc.results["energy/atom"] - o["atomic"][[3, 4, 0, 1, 2]]).max() < 1e-6
    assert set(calc.results) == {"forces", "energy/atom"}   # calculator.py:358-369
    n0 = calc.ncalls
    calc.set_prerequisite_properties(["energy", "forces", "stress"])
    calc.calculate(atoms, properties=["energy"])
    assert set(calc.results) == {"energy", "forces", "stress"}
    assert calc.ncalls == n0 + 1
    calc.reset_call_counter()
    assert calc.ncalls == 0
    with pytest.raises(KeyError):
        calc.calculate(atoms, properties=["hessian"])
    assert calc.get_magnetic_moment(atoms) is None


def test_calculator_permuted_atoms_and_absent_element(lib, tmp_path):
    from tensoralloy_amd import TensorAlloyCalculator, Atoms
    nn = make_nn(["Pd", "O"], 6.5, True, [16, 16])
    calc = TensorAlloyCalculator(nn.export(str(tmp_path / "m.json")))
    a = pd3o2()
    perm = [0, 1, 3, 4, 2]  # Pd2O2Pd of the reference tests (test_utils.py:57-66)
    b = Atoms(symbols=[a.get_chemical_symbols()[k] for k in perm], positions=a.positions[perm],
              cell=np.asarray(a.get_cell()), pbc=a.pbc)
    fa, fb = calc.get_forces(a), calc.get_forces(b)
    assert np.abs(fa[perm] - fb).max() < 1e-9
    assert abs(calc.get_potential_energy(a) - calc.get_potential_energy(b)) < 1e-9
    # a structure without oxygen still has one (masked) O row in GSL order
    pd = Atoms(symbols=["Pd"] * 3, positions=a.positions[:3], cell=np.asarray(a.get_cell()), pbc=a.pbc)
    calc.calculate(pd, properties=["forces"])
    assert calc.results["forces"].shape == (4, 3)
    assert np.all(calc.results["forces"][0] == 0.0)
    assert calc.get_forces(pd).shape == (3, 3)


def test_calculator_eam(lib, tmp_path):
    from tensoralloy_amd import TensorAlloyCalculator
    nn = make_eam(["Ni"], 6.5)
    calc = TensorAlloyCalculator(nn.export(str(tmp_path / "Ni.zhou04.pb")))
    atoms = fcc(rep=(2, 2, 2), a=3.52, jitter=0.0)
    e = calc.get_potential_energy(atoms)
    # E/atom of Zjw04 Ni for bulk('Ni', cubic=True)*[2,2,2], rc = 6.5 (SURVEY §8c pin 7)
    assert abs(e / len(atoms) + 4.44999667) < 1e-7
    atoms = fcc(rep=(2, 2, 2), a=3.52, jitter=0.05)
    o = oracle_eam_eval(nn, atoms)
    assert np.abs(calc.get_forces(atoms) - o["forces"]).max() < 1e-5
    assert np.abs(calc.get_stress(atoms) - o["stress_voigt"]).max() < 1e-8


def test_unsupported_model_file(lib, tmp_path):
    from tensoralloy_amd import TensorAlloyCalculator
    p = tmp_path / "frozen.pb"
    p.write_bytes(b"\x0a\x00")
    with pytest.raises(ValueError):
        TensorAlloyCalculator(str(p))


def test_calculator_medium_precision_model(lib, tmp_path):
    """A 'medium' (float32) model: float32 results as from the reference's float32 graph
    (calculator.py:154-159), values from the fp64 path with the float32 eps."""
    from tensoralloy_amd import TensorAlloyCalculator
    nn = make_nn(["Ni"], 6.0, True, [16, 16], precision="medium")
    calc = TensorAlloyCalculator(nn.export(str(tmp_path / "ni_medium")))
    atoms = fcc(rep=(2, 2, 2))
    calc.calculate(atoms, properties=["energy", "forces", "stress"])
    assert calc.results["forces"].dtype == np.float32 and calc.results["stress"].dtype == np.float32
    assert isinstance(calc.results["energy"], np.float32)
    o = oracle_eval(nn, atoms)
    assert abs(float(calc.results["energy"]) - o["energy"]) < 1e-5 * max(1.0, abs(o["energy"]))
    assert np.abs(calc.get_forces(atoms) - o["forces"]).max() < 1e-5


def test_elastic_constants_of_zjw04_ni_match_the_reference_values(lib, tmp_path):
    """Known answers from the reference's own tests: EamAlloyNN(['Ni'], 'zjw04'), rc = 6.0,
    `bulk('Ni', cubic=True)` (a = 3.52): C11 = 246.61, C12 = 147.15, C44 = 124.72 GPa +- 0.01
    (tests/test_calculator.py:94-111), 247 / 147 / 125 +- 1 (nn/constraint/tests/test_elastic.py:23-57).
    Second derivatives of the GPU path's energy: pins its virial."""
    from tensoralloy_amd import Atoms, TensorAlloyCalculator, UniversalTransformer
    from tensoralloy_amd.eam import EamAlloyNN
    nn = EamAlloyNN(["Ni"], "zjw04", export_properties=["energy", "forces", "stress", "elastic"])
    nn.attach_transformer(UniversalTransformer(["Ni"], rcut=6.0))
    calc = TensorAlloyCalculator(nn.export(str(tmp_path / "Ni.zhou04.elastic.pb")))
    assert "elastic" in calc.implemented_properties
    a = 3.52
    frac = np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0]])
    cubic = Atoms(symbols=["Ni"] * 4, positions=frac * a, cell=np.eye(3) * a, pbc=True)
    C = calc.get_elastic_constant_tensor(cubic)
    assert C.shape == (6, 6) and np.abs(C - C.T).max() < 1e-12
    for k in range(3):
        assert abs(C[k, k] - 246.61) < 0.01
        assert abs(C[3 + k, 3 + k] - 124.72) < 0.01
    assert abs(C[0, 1] - 147.15) < 0.01 and abs(C[0, 2] - 147.15) < 0.01 and abs(C[1, 2] - 147.15) < 0.01
    assert np.abs(C[:3, 3:]).max() < 1e-3 and abs(C[3, 4]) < 1e-3
    # bulk modulus of a cubic crystal, K = (C11 + 2 C12) / 3 = K_VRH = 180.31 (test_calculator.py:108)
    assert abs((C[0, 0] + 2 * C[0, 1]) / 3 - 180.31) < 0.01


def test_hessian_matches_the_reference_fixture(lib, tmp_path):
    """test_files/crystals/Ni_fc2.npy: Hessian of EamAlloyNN(['Ni'], 'zjw04') at rc = 6.5 written by
    the reference in 'medium' precision (nn/constraint/tests/test_fc2.py:29-54). Here: central
    differences of the GPU forces, through `get_hessian` (calculator.py:228-241)."""
    import os
    from tensoralloy_amd import Atoms, TensorAlloyCalculator, UniversalTransformer
    from tensoralloy_amd.eam import EamAlloyNN
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "Ni_fc2.npz"))
    fc2 = z["fc2"].astype(np.float64)  # [32, 32, 3, 3]
    cell = z["cell"]
    atoms = Atoms(symbols=["Ni"] * 32, positions=z["frac"] @ cell, cell=cell, pbc=True)
    nn = EamAlloyNN(["Ni"], "zjw04", export_properties=["energy", "forces", "hessian"])
    nn.attach_transformer(UniversalTransformer(["Ni"], rcut=6.5))
    nn.precision = "medium"
    calc = TensorAlloyCalculator(nn.export(str(tmp_path / "Ni_fc2")))
    H = calc.get_hessian(atoms)                      # [96, 96]
    assert H.shape == (96, 96)
    H4 = H.reshape(32, 3, 32, 3).transpose(0, 2, 1, 3)
    assert np.abs(H4 - fc2).max() < 5e-5            # fp32 noise of the fixture
    assert np.abs(H - H.T).max() < 1e-9
    assert abs(calc.get_potential_energy(atoms) / 32 + 4.44999667) < 1e-6


def test_reference_frozen_graph_runs_unchanged(lib):
    """`TensorAlloyCalculator("Ni.zhou04.pb")`: the reference's own frozen TF graph (test fixture
    test_files/models/Ni.zhou04.pb, here gzip'ed) is loaded without TensorFlow and evaluated by the
    HIP kernels. Known answer: E / atom = -4.44999667 eV for bulk('Ni', cubic=True) * [2, 2, 2] at
    a = 3.52 A (the cell of the reference's Ni_fc2.npy fixture, SURVEY 8(c) pin 7)."""
    import os
    from tensoralloy_amd import Atoms, TensorAlloyCalculator
    from tests.conftest import GOLDEN
    calc = TensorAlloyCalculator(os.path.join(GOLDEN, "Ni.zhou04.pb.gz"))
    assert calc.elements == ["Ni"] and calc.transformer.rcut == 6.5
    assert {"energy", "forces", "stress", "hessian", "atomic", "total_stress"} <= set(calc.predict_properties)
    a = 3.52
    base = np.array([[0, 0, 0], [.5, .5, 0], [.5, 0, .5], [0, .5, .5]]) * a
    pts = np.array([base + np.array([x, y, z]) * a for x in range(2) for y in range(2) for z in range(2)]).reshape(-1, 3)
    atoms = Atoms(symbols=["Ni"] * 32, positions=pts, cell=np.eye(3) * 2 * a, pbc=True)
    e = calc.get_potential_energy(atoms)
    assert abs(e / 32 - (-4.44999667)) < 1e-8
    assert np.abs(calc.get_forces(atoms)).max() < 1e-10          # perfect lattice
    full = calc.get_property("total_stress", atoms)
    voigt = calc.get_stress(atoms)
    assert full.shape == (3, 3) and abs(full[0, 0] - voigt[0]) < 1e-14 and abs(full[1, 2] - voigt[3]) < 1e-14
    per_atom = calc.get_property("atomic", atoms)
    assert per_atom.shape == (32,) and abs(per_atom.sum() - e) < 1e-9
    mo = TensorAlloyCalculator(os.path.join(GOLDEN, "Mo.zhou04.pb.gz"))
    assert mo.elements == ["Mo"] and "hessian" not in mo.predict_properties
