"""
GPU parity: the HIP path (through the C ABI) against the CPU oracle on the
same seeded inputs. Tolerances are those BASELINE.json's north_star states:
1e-6 eV on the total energy, 1e-5 eV/A per force component; descriptors and
virial are held to tighter bounds.
"""
import numpy as np
import pytest

from tests.helpers import fcc, pd3o2, make_nn, oracle_eval

pytestmark = pytest.mark.gpu

E_TOL = 1e-6     # eV, north_star
F_TOL = 1e-5     # eV/A per component, north_star
G_TOL = 1e-10    # descriptors
W_TOL = 1e-6     # eV, virial components


def _compare(nn, atoms_list):
    from tensoralloy_amd import Engine
    with Engine(nn) as eng:
        res = eng.evaluate(atoms_list, descriptors=True)
    for atoms, r in zip(atoms_list, res):
        o = oracle_eval(nn, atoms)
        assert np.abs(r["descriptors"] - o["descriptors"]).max() < G_TOL
        assert abs(r["energy"] - o["energy"]) < E_TOL
        assert np.abs(r["atomic"] - o["atomic"]).max() < E_TOL
        assert np.abs(r["forces"] - o["forces"]).max() < F_TOL
        assert np.abs(r["virial"] - o["virial"]).max() < W_TOL
        assert np.abs(r["stress"] - o["stress_voigt"]).max() < 1e-8
        assert abs(r["total_pressure"] - o["total_pressure"]) < 1e-5
    return res


def test_radial_only_single_element(lib):
    nn = make_nn(["Ni"], 6.0, False, [64])
    _compare(nn, [fcc(rep=(2, 2, 2))])


def test_angular_single_element(lib):
    nn = make_nn(["Ni"], 6.5, True, [64, 64])
    _compare(nn, [fcc(rep=(3, 3, 3))])


def test_angular_binary_pd3o2(lib):
    nn = make_nn(["Pd", "O"], 6.5, True, [32, 32], minmax=True, resnet=True)
    _compare(nn, [pd3o2()])


def test_batch_of_frames(lib):
    nn = make_nn(["Ni"], 6.5, True, [64, 64])
    frames = [fcc(rep=(2, 2, 2), seed=611 + k) for k in range(3)] + [fcc(rep=(2, 2, 3), seed=7)]
    _compare(nn, frames)


def _alloy(symbols_cycle, rep=(2, 2, 2), a=3.6, seed=3):
    atoms = fcc(rep=rep, a=a, seed=seed)
    from tensoralloy_amd import Atoms
    syms = [symbols_cycle[k % len(symbols_cycle)] for k in range(len(atoms))]
    rng = np.random.RandomState(seed)
    rng.shuffle(syms)
    return Atoms(symbols=syms, positions=atoms.positions, cell=np.asarray(atoms.get_cell()),
                 pbc=True)


def test_binary_alloy_cross_terms(lib):
    nn = make_nn(["Ni", "Mo"], 6.5, True, [128, 128])
    _compare(nn, [_alloy(["Ni", "Ni", "Ni", "Mo"], rep=(2, 2, 3))])


def test_ternary_alloy(lib):
    nn = make_nn(["Al", "Cu", "Ni"], 5.0, True, [16, 16], activation="tanh")
    _compare(nn, [_alloy(["Al", "Cu", "Ni"], rep=(2, 2, 2))])


def test_four_and_five_elements(lib):
    # default channel grid: second-generation kernels with 4 / 5 partner species
    nn = make_nn(["Al", "Cu", "Mo", "Ni"], 4.5, True, [16], activation="relu")
    _compare(nn, [_alloy(["Al", "Cu", "Ni", "Mo"], rep=(2, 2, 2))])
    nn = make_nn(["Al", "Co", "Cu", "Fe", "Ni"], 5.0, True, [16, 16], minmax=True)
    _compare(nn, [_alloy(["Al", "Co", "Cu", "Fe", "Ni"], rep=(2, 2, 3)),
                  _alloy(["Ni", "Fe", "Ni", "Co", "Al", "Cu", "Ni"], rep=(2, 2, 2), seed=9)])
    # another grid (one zeta): first-generation kernels
    nn = make_nn(["Al", "Cu", "Mo", "Ni"], 4.5, True, [16], sf_kwargs=dict(zeta=[2.0]))
    _compare(nn, [_alloy(["Al", "Cu", "Ni", "Mo"], rep=(2, 2, 2))])
    # six elements: first-generation kernels
    nn = make_nn(["Al", "Co", "Cu", "Fe", "Mo", "Ni"], 4.5, True, [8])
    _compare(nn, [_alloy(["Al", "Co", "Cu", "Fe", "Mo", "Ni"], rep=(2, 2, 2))])


def test_first_generation_kernels(lib, monkeypatch):
    monkeypatch.setenv("TA_FORCE_V1", "1")
    nn = make_nn(["Ni", "Mo"], 6.5, True, [32, 32])
    _compare(nn, [_alloy(["Ni", "Mo"], rep=(2, 2, 2))])


def test_molecule_non_periodic(lib):
    from tensoralloy_amd import Atoms
    rng = np.random.RandomState(28)
    pos = rng.rand(28, 3) * 7.0
    atoms = Atoms(symbols=["B"] * 28, positions=pos, cell=np.zeros((3, 3)), pbc=False)
    nn = make_nn(["B"], 6.0, True, [32, 32], sf_kwargs=dict(omega=[0.0, 1.5]))
    _compare(nn, [atoms])


def test_tiny_cell_self_images(lib):
    from tensoralloy_amd import Atoms
    lat = np.array([[2.479787, 0, 0], [-1.239893, 2.147558, 0], [0, 0, 24.294656]])
    rng = np.random.RandomState(11)
    frac = rng.rand(6, 3)
    atoms = Atoms(symbols=["Ni"] * 6, positions=frac @ lat, cell=lat, pbc=True)
    nn = make_nn(["Ni"], 6.0, True, [64])
    _compare(nn, [atoms])


def test_acut_differs_polynomial_cutoff_multi_beta(lib):
    nn = make_nn(["Ni"], 6.0, True, [24, 24], acut=5.0, cutoff="polynomial",
                 sf_kwargs=dict(beta=[0.005, 0.1, 1.0], gamma=[1.0, -1.0, 0.5], zeta=[1.0, 2.0, 16.0]),
                 activation="squareplus", resnet=True, minmax=True)
    _compare(nn, [fcc(rep=(2, 2, 2))])


def test_non_integer_zeta(lib):
    nn = make_nn(["Ni"], 5.5, True, [16], sf_kwargs=dict(zeta=[1.5, 4.0]), activation="elu")
    _compare(nn, [fcc(rep=(2, 2, 2))])


def test_empty_and_single_atom_frames(lib):
    from tensoralloy_amd import Atoms, Engine
    nn = make_nn(["Ni"], 6.5, True, [16])
    lone = Atoms(symbols=["Ni"], positions=[[0.0, 0.0, 0.0]], cell=np.eye(3) * 30.0, pbc=True)
    with Engine(nn) as eng:
        res = eng.evaluate([lone, fcc(rep=(2, 2, 2))])
        o = oracle_eval(nn, lone)
        assert abs(res[0]["energy"] - o["energy"]) < E_TOL
        assert np.abs(res[0]["forces"]).max() == 0.0
        res = eng.evaluate([])
        assert res == []


def test_many_neighbours_multi_pass(lib):
    """rc = 9 A: ~260 neighbours per atom (> 255): second-generation kernels, 2 lanes passes."""
    nn = make_nn(["Ni"], 9.0, True, [16], sf_kwargs=dict(eta=[0.5, 4.0]))
    _compare(nn, [fcc(rep=(2, 2, 2), a=3.4)])


def test_mid_neighbour_count(lib):
    """rc = 8 A: ~180 neighbours per atom: more than one candidate block of 64 per lane."""
    nn = make_nn(["Ni"], 8.0, True, [16], sf_kwargs=dict(eta=[0.5, 4.0]))
    _compare(nn, [fcc(rep=(2, 2, 2), a=3.5)])


def test_alloy_minmax_and_resnet_variants(lib):
    nn = make_nn(["Ni", "Mo"], 6.5, True, [32, 32], minmax=True)
    _compare(nn, [_alloy(["Ni", "Mo"], rep=(2, 2, 2)), fcc(rep=(2, 2, 2))])
    nn = make_nn(["Al", "Cu", "Ni"], 5.0, True, [16, 16], activation="tanh", resnet=True)
    _compare(nn, [_alloy(["Al", "Cu", "Ni"], rep=(2, 2, 2))])


def test_energy_only(lib, monkeypatch):
    from tensoralloy_amd import Engine, _lib
    nn = make_nn(["Ni"], 6.5, True, [64, 64])
    atoms = fcc(rep=(2, 2, 2))
    with Engine(nn) as eng:
        r = eng.evaluate([atoms], want=_lib.TA_WANT_ENERGY | _lib.TA_WANT_ATOMIC)[0]
    o = oracle_eval(nn, atoms, want_forces=False)
    assert abs(r["energy"] - o["energy"]) < E_TOL
    assert np.abs(r["atomic"] - o["atomic"]).max() < E_TOL
    assert "forces" not in r


def test_medium_precision_eps(lib):
    """'medium' models: sqrt(D.D + 1e-8) (precision.py:114) everywhere a distance is formed,
    including r_jk and the law-of-cosines angle."""
    nn = make_nn(["Mo", "Ni"], 6.0, True, [16, 16], precision="medium")
    atoms = _alloy(["Ni", "Ni", "Mo"], rep=(2, 2, 2))
    res = _compare(nn, [atoms])
    nn_hi = make_nn(["Mo", "Ni"], 6.0, True, [16, 16])
    from tensoralloy_amd import Engine
    with Engine(nn_hi) as eng:
        hi = eng.evaluate([atoms], descriptors=True)[0]
    # eps really changed the numbers (1e-8 under the roots is visible at 1e-10)
    assert np.abs(hi["descriptors"] - res[0]["descriptors"]).max() > 1e-9


@pytest.mark.parametrize("minmax", [False, True])
def test_non_symmetric_single_element(lib, minmax):
    """symmetric=False lists (j, k) and (k, j): twice the symmetric angular features."""
    nn = make_nn(["Ni"], 6.0, True, [16, 16], symmetric=False, minmax=minmax)
    _compare(nn, [fcc(rep=(2, 2, 2), seed=21)])
    from tests.helpers import AtomicNN  # noqa: F401
    with pytest.raises(ValueError, match="only for one element"):
        make_nn(["Mo", "Ni"], 6.0, True, [8], symmetric=False).to_desc()


def test_gpu_descriptors_against_the_reference_fixtures(lib):
    """The two descriptor fixtures the reference's own tests hold, applied to the GPU path directly:
    `amp_Pd3O2.npz` (AMP-generated G2+G4 of the periodic Pd3O2 slab, test_sf.py:666-691) and the
    vectors of the reference's NumPy helpers on B28.xyz (test_sf.py:159-311)."""
    import os
    from tensoralloy_amd import Atoms, Engine
    golden = os.path.join(os.path.dirname(__file__), "golden")
    g = np.load(os.path.join(golden, "amp_Pd3O2.npz"))["g"]
    nn = make_nn(["Pd", "O"], 6.5, True, [8])
    with Engine(nn) as eng:
        G = eng.evaluate([pd3o2()], descriptors=True)[0]["descriptors"]
    assert np.abs(G[3:5] - g[3:5, 0:20]).max() < 1e-12   # O rows
    assert np.abs(G[0:3] - g[0:3, 20:40]).max() < 1e-12  # Pd rows
    z = np.load(os.path.join(golden, "B28_sf.npz"))
    coords, rc = z["coords"], float(z["rc"])
    b28 = Atoms(symbols=["B"] * len(coords), positions=coords, cell=np.eye(3) * 60.0, pbc=False)
    kw = dict(eta=z["etas"].tolist(), omega=[0.0], beta=z["betas"].tolist(), gamma=z["gammas"].tolist(),
              zeta=z["zetas"].tolist())
    nn = make_nn(["B"], rc, True, [8], sf_kwargs=kw)
    with Engine(nn) as eng:
        G = eng.evaluate([b28], descriptors=True)[0]["descriptors"]
    assert np.abs(G[:, :4] - z["g2_v1"]).max() < 1e-11
    assert np.abs(G[:, 4:] - z["g4"]).max() < 1e-11
    nn = make_nn(["B"], rc, False, [8], sf_kwargs=dict(eta=z["etas"].tolist(), omega=z["omegas"].tolist()))
    with Engine(nn) as eng:
        G2 = eng.evaluate([b28], descriptors=True)[0]["descriptors"]
    assert np.abs(G2 - z["g2_v2"]).max() < 1e-11


def test_safe_pow_edge_non_integer_zeta(lib):
    """a22, `safe_pow` (extension/grad_ops.py:16-74) at the edge it exists for: a collinear triple,
    where 1 + gamma cos(theta) -> 0 and, for zeta < 1, the factor zeta base^(zeta-1) of the gradient
    blows up. The eps under every root (universal.py:470-472) keeps the base a few 1e-14 above 0
    for real geometries, so both variants (plain pow = the reference's default; the masked one of
    TENSORALLOY_USE_CUSTOM_POW) stay finite and agree; the huge, ill-conditioned derivative factor
    (~1e7) is reproduced to the accuracy the 1-ulp differences in cos(theta) allow."""
    from tensoralloy_amd import Atoms, Engine
    pos = np.array([[0.0, 0.0, 0.0], [1.3, 0.0, 0.0], [2.9, 0.0, 0.0], [0.4, 1.7, 0.3]])
    atoms = Atoms(symbols=["Ni"] * 4, positions=pos + 10.0, cell=np.eye(3) * 30.0, pbc=False)
    got = {}
    for custom in (False, True):
        for zeta in ([0.5, 1.5], [1.0, 2.5]):
            nn = make_nn(["Ni"], 4.0, True, [8], sf_kwargs=dict(eta=[0.5], zeta=zeta, gamma=[1.0, -1.0]))
            nn.use_custom_pow = custom
            with Engine(nn) as eng:
                r = eng.evaluate([atoms])[0]
            o = oracle_eval(nn, atoms)
            assert np.isfinite(r["energy"]) and np.all(np.isfinite(r["forces"]))
            assert np.all(np.isfinite(o["forces"]))
            assert abs(r["energy"] - o["energy"]) < E_TOL
            scale = max(1.0, np.abs(o["forces"]).max())
            tol = 0.05 * scale if min(zeta) < 1.0 else F_TOL
            assert np.abs(r["forces"] - o["forces"]).max() < tol
            got[(custom, tuple(zeta))] = r
    for zeta in ((0.5, 1.5), (1.0, 2.5)):  # nothing is exactly singular: the mask changes nothing
        a, b = got[(False, zeta)], got[(True, zeta)]
        assert a["energy"] == b["energy"] and np.array_equal(a["forces"], b["forces"])


def test_failed_set_frames_leaves_no_batch_behind(lib):
    """A ta_set_frames that fails (here: a species index outside the model) must not leave the
    previous batch's flags standing: the next ta_compute is refused, not run on stale buffers."""
    from tensoralloy_amd import Engine, _lib as L
    import ctypes as C
    nn = make_nn(["Ni"], 6.5, True, [16])
    atoms = fcc(rep=(2, 2, 2))
    with Engine(nn) as eng:
        eng.evaluate([atoms])
        bad = L.FrameArrays(np.full(len(atoms), 3, dtype=np.int32), atoms.positions,
                            np.asarray(atoms.get_cell(complete=True)), [True] * 3)
        arr = (L.Frame * 1)(bad.as_struct())
        rc = eng._lib.ta_set_frames(eng._handle, 1, arr, None)
        assert rc == L.TA_ERR_INVALID
        assert eng._lib.ta_compute(eng._handle, L.TA_WANT_ENERGY) == L.TA_ERR_INVALID
        null = C.POINTER(C.c_double)()
        assert eng._lib.ta_get_results(eng._handle, null, null, null, null, null) == L.TA_ERR_INVALID
        r = eng.evaluate([atoms])[0]  # and the handle is still usable
        assert abs(r["energy"] - oracle_eval(nn, atoms)["energy"]) < E_TOL


@pytest.mark.parametrize("hidden,activation,minmax", [
    ([64, 64], "softplus", False), ([64, 32], "tanh", True), ([48], "softplus", False),
    ([16, 32, 64], "squareplus", False), ([20, 50], "elu", True)])
def test_one_wavefront_mlp_kernel(lib, monkeypatch, hidden, activation, minmax):
    """Batches of 1024+ tiles run the MLP in `mlp_wave_kernel` (transposed GEMMs, activations in
    registers, weights staged in LDS); forced here for small inputs. Smaller launches of one-element
    models run `mlp_quad_kernel` (the same transposed GEMMs over four wavefronts, one barrier per layer). Same results as the oracle
    and, to round-off, as the 16-row tile kernel: single element, alloy (one grid row per
    element), ragged last tile, widths that pad to 16 / 32 / 48 / 64, one to three hidden layers."""
    from tensoralloy_amd import Engine
    frames = [fcc(rep=(2, 2, 2), jitter=0.05), fcc(rep=(3, 2, 2), a=3.4, seed=2, jitter=0.08)]   # 32 + 48 atoms
    alloy = [_alloy(["Ni", "Ni", "Mo"], rep=(2, 2, 3))]
    for nn, fr in ((make_nn(["Ni"], 6.0, True, hidden, activation=activation, minmax=minmax), frames),
                   (make_nn(["Mo", "Ni"], 5.5, True, hidden, activation=activation, minmax=minmax), alloy)):
        monkeypatch.setenv("TA_MLP_TILE_KERNEL", "1")
        with Engine(nn) as eng:
            tile = eng.evaluate(fr)
        monkeypatch.delenv("TA_MLP_TILE_KERNEL")
        monkeypatch.setenv("TA_MLP_WAVE_KERNEL", "1")
        wave = _compare(nn, fr)
        monkeypatch.delenv("TA_MLP_WAVE_KERNEL")
        # forced: single-element models of these shapes take the four-wavefront kernel
        # (`mlp_quad_kernel`; by itself from 257 tiles on), alloys the generic tile kernel
        monkeypatch.setenv("TA_MLP_QUAD_KERNEL", "1")
        quad = _compare(nn, fr)
        monkeypatch.delenv("TA_MLP_QUAD_KERNEL")
        for a, b, c in zip(tile, wave, quad):
            for other in (b, c):
                assert abs(a["energy"] - other["energy"]) < 1e-10
                assert np.abs(a["forces"] - other["forces"]).max() < 1e-10
                assert np.abs(a["atomic"] - other["atomic"]).max() < 1e-11


def test_angular_kernels_without_job_lists(lib, monkeypatch):
    """`TA_NO_JOBS=1`: the second-generation kernels without the forward -> backward job list (lanes
    re-dealt by popcount, descriptors assembled from the lanes' partial sums in LDS). The path batches
    fall back to when the list buffers are switched off; same parity gate."""
    monkeypatch.setenv("TA_NO_JOBS", "1")
    _compare(make_nn(["Ni"], 6.5, True, [16, 16]), [fcc(rep=(3, 3, 3), jitter=0.05), fcc(rep=(2, 2, 2), a=3.4, seed=4)])
    _compare(make_nn(["Mo", "Ni"], 6.0, True, [16]), [_alloy(["Ni", "Ni", "Mo"], rep=(2, 2, 2))])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["ni_default", "binary_minmax_resnet_poly", "radial_only", "multi_chunk"])
def test_analytic_hessian_vectors_of_the_descriptor_models(lib, kind):
    """Round 3: `ta_hessian_vectors` for the symmetry-function + MLP models (ta_hvp.hip: the first-generation
    backward expression in dual arithmetic with w = (dE/dG, H_mlp G-dot); replaces tf.hessians,
    nn/basic.py:411-421, and the cell derivative of the virial, nn/constraint/elastic.py:24-44) against
    central differences of the GPU's own analytic forces and virials along random directions of positions
    and cell, plus symmetry and the acoustic sum rule of the Hessian from the unit directions."""
    from tensoralloy_amd import Atoms, Engine
    from tests.helpers import fcc, make_nn
    if kind == "ni_default":
        nn, atoms = make_nn(["Ni"], 6.0, True, [32, 32]), fcc(rep=(2, 2, 2), seed=3, jitter=0.1)
    elif kind == "binary_minmax_resnet_poly":
        nn = make_nn(["Mo", "Ni"], 5.5, True, [16, 16], minmax=True, resnet=True, cutoff="polynomial", activation="tanh")
        base = fcc(rep=(2, 2, 2), seed=5, jitter=0.1)
        cell = np.asarray(base.get_cell(complete=True)).copy()
        cell[1, 0] = 0.2 * cell[0, 0]
        atoms = Atoms(symbols=["Mo" if k % 3 == 0 else "Ni" for k in range(len(base))], positions=base.positions,
                      cell=cell, pbc=True)
    elif kind == "radial_only":
        nn, atoms = make_nn(["Ni"], 6.0, False, [16]), fcc(rep=(2, 2, 2), seed=4, jitter=0.1)
    else:   # several betas: more than one angular launch
        nn = make_nn(["Ni"], 5.0, True, [16], sf_kwargs=dict(beta=[0.005, 0.02, 0.1], gamma=[1.0, -1.0], zeta=[1.0, 2.0]))
        atoms = fcc(rep=(2, 2, 2), seed=6, jitter=0.1)
    n = len(atoms)
    h = np.asarray(atoms.get_cell(complete=True), dtype=float)
    rng = np.random.RandomState(2)
    dR = rng.normal(size=(2, n, 3))
    dh = rng.normal(size=(2, 1, 3, 3)) * 0.3
    dR[1] = 0.0        # second direction: the cell alone (the elastic-constant case)
    want, eps = 1 | 2 | 4, 1e-4
    with Engine(nn) as eng:
        eng.set_frames([atoms])
        dF, dW = eng.hessian_vectors(dR=dR, dh=dh, want_virial=True)
        H = -eng.hessian_vectors()                  # [3 n, n, 3]
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
    assert np.abs(Hm.sum(axis=1)).max() < 1e-8          # acoustic sum rule: a rigid shift costs nothing
