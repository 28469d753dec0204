"""GPU parity of the GRAP descriptor path (csrc/ta_grap.hip) against oracle/grap.py."""
import numpy as np
import pytest

from tests.helpers import fcc, make_grap_nn, oracle_grap_eval
from tests.test_gpu_sf import _alloy, E_TOL, F_TOL, G_TOL, W_TOL

pytestmark = pytest.mark.gpu


def _compare(nn, atoms_list):
    from tensoralloy_amd import Engine
    with Engine(nn) as eng:
        res = eng.evaluate(atoms_list, descriptors=True)
    for atoms, r in zip(atoms_list, res):
        o = oracle_grap_eval(nn, atoms)
        scale = max(1.0, np.abs(o["descriptors"]).max())
        assert np.abs(r["descriptors"] - o["descriptors"]).max() < G_TOL * scale
        assert abs(r["energy"] - o["energy"]) < E_TOL
        assert np.abs(r["atomic"] - o["atomic"]).max() < E_TOL
        assert np.abs(r["forces"] - o["forces"]).max() < F_TOL
        assert np.abs(r["virial"] - o["virial"]).max() < W_TOL
        assert np.abs(r["stress"] - o["stress_voigt"]).max() < 1e-8
    return res


def test_default_production_descriptor(lib):
    """defaults.toml:131-155: pexp, 16 filters, moments 0..3, new mode, cosine cutoff, rc 6.0."""
    rl = [1.0 + 0.2 * k for k in range(16)]
    pl = [5.0 - 0.25 * k for k in range(16)]
    nn = make_grap_nn(["Ni"], 6.0, [64, 64], "pexp", {"rl": rl, "pl": pl}, moment_tensors=[0, 1, 2, 3])
    _compare(nn, [fcc(rep=(3, 3, 3)), fcc(rep=(2, 2, 2), a=3.3, seed=5)])


def test_binary_be_w_pexp_polynomial(lib):
    """The reference's own GRAP test case (test_grap.py:49-105): Be/W, pexp, moments 0 1 2,
    polynomial cutoff, rc 5; legacy and new mode must agree to 1e-6."""
    atoms = _alloy(["Be", "W"], rep=(2, 2, 2), a=3.6)
    new = make_grap_nn(["Be", "W"], 5.0, [16, 16], cutoff="polynomial")
    old = make_grap_nn(["Be", "W"], 5.0, [16, 16], cutoff="polynomial", legacy_mode=True)
    g_new = _compare(new, [atoms])[0]["descriptors"]
    g_old = _compare(old, [atoms])[0]["descriptors"]
    assert np.abs(g_new - g_old).max() < 1e-6


@pytest.mark.parametrize("algorithm,parameters,method", [
    ("sf", {"eta": [0.1, 0.5, 1.0, 4.0], "omega": [0.0, 1.5]}, "cross"),
    ("morse", {"D": [1.0, 1.0, 1.0], "gamma": [1.0, 1.0, 1.0], "r0": [3.3, 3.4, 3.5]}, "pair"),
    ("density", {"A": [1.0], "beta": [1.0, 2.0, 3.0, 4.0], "re": [4.0]}, "cross"),
    ("pexp", {"rl": [1.5, 2.0, 2.5], "pl": [1.0, 2.0, 3.0]}, "pair"),
])
def test_every_filter_family(lib, algorithm, parameters, method):
    nn = make_grap_nn(["Mo", "Ni"], 6.0, [16], algorithm, parameters, moment_tensors=[0, 1, 2, 3],
                      symmetric=True, param_space_method=method, minmax=True)
    _compare(nn, [_alloy(["Ni", "Ni", "Mo"], rep=(2, 2, 2))])


def test_legacy_moment_subsets_and_many_filters(lib):
    rl = [1.0 + 0.1 * k for k in range(20)]
    pl = [4.0 - 0.1 * k for k in range(20)]
    nn = make_grap_nn(["Ni"], 6.0, [16], "pexp", {"rl": rl, "pl": pl}, moment_tensors=[2, 0],
                      legacy_mode=True)
    assert nn.ndim() == 2 * 20
    _compare(nn, [fcc(rep=(2, 2, 2), seed=8)])
    nn = make_grap_nn(["Ni"], 6.0, [16], "pexp", {"rl": rl, "pl": pl}, moment_tensors=[1])
    assert nn.ndim() == 2 * 20  # new mode emits every moment up to the largest (grap.py:606)
    _compare(nn, [fcc(rep=(2, 2, 2), seed=8)])


def test_molecule_medium_precision_and_calculator(lib, tmp_path):
    from tensoralloy_amd import Atoms, TensorAlloyCalculator
    nn = make_grap_nn(["C", "H"], 5.0, [16, 16], moment_tensors=[0, 1, 2, 3], precision="medium")
    rng = np.random.RandomState(4)
    pos = rng.rand(9, 3) * 4.0
    atoms = Atoms(symbols=["C", "H", "H", "H", "C", "H", "H", "H", "H"], positions=pos,
                  cell=np.eye(3) * 20.0, pbc=False)
    _compare(nn, [atoms])
    calc = TensorAlloyCalculator(nn.export(str(tmp_path / "grap")))
    o = oracle_grap_eval(nn, atoms)
    assert np.abs(calc.get_forces(atoms) - o["forces"]).max() < 1e-5
    assert calc.results["forces"].dtype == np.float32


def test_native_npz_model_runs_in_the_calculator(lib, tmp_path):
    """A model in the reference's `export_to_lammps_native` format is loaded as is."""
    from tensoralloy_amd import TensorAlloyCalculator
    nn = make_grap_nn(["Mo", "Ni"], 6.0, [32, 32], moment_tensors=[0, 1, 2, 3])
    path = nn.export_to_lammps_native(str(tmp_path / "MoNi.npz"))
    calc = TensorAlloyCalculator(path)
    atoms = _alloy(["Ni", "Ni", "Mo"], rep=(2, 2, 2))
    o = oracle_grap_eval(nn, atoms)
    assert abs(calc.get_potential_energy(atoms) - o["energy"]) < E_TOL
    assert np.abs(calc.get_forces(atoms) - o["forces"]).max() < F_TOL
    assert np.abs(calc.get_stress(atoms) - o["stress_voigt"]).max() < 1e-8


def test_nn_filter_network(lib, tmp_path):
    """The `nn` algorithm (grap.py:220-270, :632-643; defaults.toml `[nn.atomic.grap.nn]`: softplus,
    hidden 32-32-32 with ResNet skips, 16 filters): one shared filter network instead of analytic
    filters. Default shape, a second shape / activation, the model file and the native `.npz`."""
    from tensoralloy_amd import TensorAlloyCalculator
    nn = make_grap_nn(["Ni"], 6.0, [64, 64], "nn", moment_tensors=[0, 1, 2, 3])
    assert nn.descriptor.algorithm.hidden_sizes == [32, 32, 32] and len(nn.descriptor.algorithm) == 16
    assert nn.ndim() == 4 * 16
    _compare(nn, [fcc(rep=(3, 3, 3)), fcc(rep=(1, 1, 1))])
    par = {"hidden_sizes": [24, 40], "num_filters": 10, "activation": "tanh", "use_resnet_dt": False}
    nn2 = make_grap_nn(["Mo", "Ni"], 5.5, [16], "nn", par, moment_tensors=[0, 1, 2], symmetric=True)
    atoms = _alloy(["Ni", "Ni", "Mo"], rep=(2, 2, 2))
    _compare(nn2, [atoms])
    o = oracle_grap_eval(nn2, atoms)
    for path in (nn2.export(str(tmp_path / "fnn")), nn2.export_to_lammps_native(str(tmp_path / "fnn_native.npz"))):
        calc = TensorAlloyCalculator(path)
        assert abs(calc.get_potential_energy(atoms) - o["energy"]) < E_TOL
        assert np.abs(calc.get_forces(atoms) - o["forces"]).max() < F_TOL
    with pytest.raises(ValueError, match="legacy_mode=False"):
        make_grap_nn(["Ni"], 6.0, [16], "nn", legacy_mode=True)


@pytest.mark.parametrize("modifier", [1, 2])
def test_nn_filter_network_input_modifiers(lib, tmp_path, modifier):
    """`h_abck_modifier` (grap.py:620-631): the filter network sees r / rcov (1) or exp(-r / rcov) (2)
    with the covalent radius of the CENTRE's element; round trip through the native `.npz`
    (`fnn::h_abck_modifier`, atomic.py:419)."""
    from tensoralloy_amd import TensorAlloyCalculator
    par = {"hidden_sizes": [32, 32], "num_filters": 8, "h_abck_modifier": modifier}
    nn = make_grap_nn(["Mo", "Ni"], 5.5, [16, 16], "nn", par, moment_tensors=[0, 1, 2, 3])
    assert nn.descriptor.algorithm.h_abck_modifier == modifier
    atoms = _alloy(["Ni", "Ni", "Mo"], rep=(2, 2, 2))
    _compare(nn, [atoms, fcc(rep=(2, 2, 2))])
    plain = make_grap_nn(["Mo", "Ni"], 5.5, [16, 16], "nn", dict(par, h_abck_modifier=0), moment_tensors=[0, 1, 2, 3])
    assert abs(oracle_grap_eval(nn, atoms)["energy"] - oracle_grap_eval(plain, atoms)["energy"]) > 1e-6
    o = oracle_grap_eval(nn, atoms)
    calc = TensorAlloyCalculator(nn.export_to_lammps_native(str(tmp_path / "mod.npz")))
    assert calc._nn.descriptor.algorithm.h_abck_modifier == modifier
    assert abs(calc.get_potential_energy(atoms) - o["energy"]) < E_TOL
    assert np.abs(calc.get_forces(atoms) - o["forces"]).max() < F_TOL


@pytest.mark.parametrize("mm", [4, 5])
def test_moment_tensors_of_rank_4_and_5(lib, tmp_path, mm):
    """max_moment 4 / 5 (grap.py:538-600: the reference switches to full 3^m tensors with unit
    weights there; here 35 / 56 packed components with multinomial weights, four column tiles of
    the MFMA kernels): analytic filters for one and two elements, the `nn` filter network, the
    native `.npz` round trip."""
    from tensoralloy_amd import TensorAlloyCalculator
    rl = [1.0 + 0.2 * k for k in range(16)]
    pl = [5.0 - 0.25 * k for k in range(16)]
    nn = make_grap_nn(["Ni"], 6.0, [32, 32], "pexp", {"rl": rl, "pl": pl}, moment_tensors=list(range(mm + 1)))
    assert nn.ndim() == (mm + 1) * len(rl)
    _compare(nn, [fcc(rep=(2, 2, 2)), fcc(rep=(2, 2, 2), a=3.3, seed=5)])
    nn2 = make_grap_nn(["Mo", "Ni"], 5.5, [16], "morse",
                       {"D": [0.5, 1.0, 1.5], "gamma": [1.0, 1.2, 1.4], "r0": [2.0, 2.5, 3.0]},
                       moment_tensors=[mm], symmetric=True)
    atoms = _alloy(["Ni", "Ni", "Mo"], rep=(2, 2, 2))
    _compare(nn2, [atoms])
    nn3 = make_grap_nn(["Ni"], 5.0, [16], "nn", {"hidden_sizes": [16, 16], "num_filters": 6},
                       moment_tensors=list(range(mm + 1)))
    _compare(nn3, [fcc(rep=(2, 2, 2), seed=3)])
    o = oracle_grap_eval(nn2, atoms)
    calc = TensorAlloyCalculator(nn2.export_to_lammps_native(str(tmp_path / "m.npz")))
    assert calc._nn.descriptor.max_moment == mm
    assert abs(calc.get_potential_energy(atoms) - o["energy"]) < E_TOL
    assert np.abs(calc.get_forces(atoms) - o["forces"]).max() < F_TOL


@pytest.mark.parametrize("kind", ["pexp_default", "sf_binary_minmax", "morse_legacy", "density_rank5", "nn_filter",
                                  "nn_filter_modifier"])
def test_analytic_hessian_vectors_of_grap_models(lib, kind):
    """Round 3: `ta_hessian_vectors` for GRAP + MLP models (ta_grap.hip::grap_hvp_kernel: the forward /
    backward expressions in dual arithmetic; replaces tf.hessians, nn/basic.py:411-421, and the cell
    derivative of the virial, nn/constraint/elastic.py:24-44) against central differences of the GPU's
    own analytic forces and virials along random directions of positions and cell, plus symmetry and
    the acoustic sum rule of the Hessian; the four analytic filter families and the filter network (`nn`)."""
    from tensoralloy_amd import Atoms, Engine
    if kind == "pexp_default":
        nn, atoms = make_grap_nn(["Ni"], 6.0, [32, 32], moment_tensors=[0, 1, 2, 3]), fcc(rep=(2, 2, 2), seed=3, jitter=0.1)
    elif kind == "sf_binary_minmax":
        nn = make_grap_nn(["Mo", "Ni"], 6.0, [16], "sf", {"eta": [0.1, 0.5, 1.0, 4.0], "omega": [0.0, 1.5]},
                          moment_tensors=[0, 1, 2], symmetric=True, param_space_method="cross", minmax=True,
                          cutoff="polynomial")
        atoms = _alloy(["Ni", "Ni", "Mo"], rep=(2, 2, 2))
    elif kind == "morse_legacy":
        nn = make_grap_nn(["Ni"], 6.0, [16], "morse", {"D": [1.0, 1.0, 1.0], "gamma": [1.0, 1.0, 1.0], "r0": [3.3, 3.4, 3.5]},
                          moment_tensors=[2, 0], legacy_mode=True)
        atoms = fcc(rep=(2, 2, 2), seed=8, jitter=0.1)
    elif kind == "density_rank5":
        nn = make_grap_nn(["Ni"], 5.0, [16], "density", {"A": [1.0], "beta": [1.0, 2.0, 3.0, 4.0], "re": [4.0]},
                          moment_tensors=[0, 1, 2, 3, 4, 5], param_space_method="cross")
        atoms = fcc(rep=(2, 2, 2), seed=5, jitter=0.1)
    elif kind == "nn_filter":     # the filter network: second r-derivative by a per-pair sweep (grap_net_d2)
        nn, atoms = make_grap_nn(["Ni"], 6.0, [16], "nn", moment_tensors=[0, 1, 2]), fcc(rep=(2, 2, 2), seed=3, jitter=0.1)
    else:                         # ... with its input modifier exp(-r / rcov) and two elements
        nn = make_grap_nn(["Mo", "Ni"], 6.0, [16], "nn", {"hidden_sizes": [32, 32], "num_filters": 8, "h_abck_modifier": 2},
                          moment_tensors=[0, 1, 2, 3])
        atoms = _alloy(["Ni", "Ni", "Mo"], rep=(2, 2, 2))
    n = len(atoms)
    h = np.asarray(atoms.get_cell(complete=True), dtype=float)
    rng = np.random.RandomState(2)
    dR = rng.normal(size=(2, n, 3))
    dh = rng.normal(size=(2, 1, 3, 3)) * 0.3
    dR[1] = 0.0
    want, eps = 1 | 2 | 4, 1e-4
    with Engine(nn) as eng:
        eng.set_frames([atoms])
        dF, dW = eng.hessian_vectors(dR=dR, dh=dh, want_virial=True)
        H = -eng.hessian_vectors()
        for d in range(2):
            fd_F, fd_W = 0.0, 0.0
            for sgn in (1.0, -1.0):
                a = Atoms(symbols=atoms.get_chemical_symbols(), positions=atoms.positions + sgn * eps * dR[d],
                          cell=h + sgn * eps * dh[d, 0], pbc=True)
                r = eng.evaluate([a], want=want)[0]
                fd_F = fd_F + sgn * r["forces"] / (2 * eps)
                fd_W = fd_W + sgn * r["virial"] / (2 * eps)
            assert np.abs(dF[d] - fd_F).max() < 2e-6 * max(1.0, np.abs(fd_F).max()), (d, np.abs(dF[d] - fd_F).max())
            assert np.abs(dW[d, 0] - fd_W).max() < 2e-6 * max(1.0, np.abs(fd_W).max()), (d, np.abs(dW[d, 0] - fd_W).max())
    Hm = H.reshape(3 * n, 3 * n)
    assert np.abs(Hm - Hm.T).max() < 1e-9 * max(1.0, np.abs(Hm).max())
    assert np.abs(Hm.sum(axis=1)).max() < 1e-8
