"""Seeded random structures through every model family: triclinic cells, mixed periodicity, atoms
outside the box, 1-3 elements, random cutoffs and descriptor parameters, batches of uneven frames.
GPU (through the C ABI) against the CPU oracle at the tolerances north_star states."""
import numpy as np
import pytest

from tests.helpers import (make_nn, make_eam, make_grap_nn, oracle_eval, oracle_eam_eval,
                           oracle_grap_eval)
from tests.test_gpu_sf import E_TOL, F_TOL, W_TOL
from tensoralloy_amd import Atoms

pytestmark = pytest.mark.gpu


def _random_frame(rng, elements, n_min=8, n_max=48, min_dist=1.6):
    L = rng.uniform(5.5, 11.0, 3)
    cell = np.diag(L)
    cell[1, 0], cell[2, 0], cell[2, 1] = rng.uniform(-0.3, 0.3, 3) * L[0]
    pbc = rng.rand(3) < 0.7
    # random sequential addition jams near 38 % packing: stay well below it
    room = int(0.2 * abs(np.linalg.det(cell)) / (4.0 / 3.0 * np.pi * (0.5 * min_dist) ** 3))
    n = rng.randint(n_min, max(n_min, min(n_max, room)) + 1)
    pts = []
    for _ in range(100000):                  # random points with a minimum image distance
        if len(pts) == n:
            break
        f = rng.rand(3)
        p = f @ cell
        ok = True
        for q in pts:
            d = p - q
            fr = d @ np.linalg.inv(cell)
            fr -= np.where(pbc, np.round(fr), 0.0)
            if np.linalg.norm(fr @ cell) < min_dist:
                ok = False
                break
        if ok:
            pts.append(p)
    n = len(pts)
    pos = np.array(pts) + (rng.randint(-1, 2, (n, 3)) * pbc) @ cell  # some atoms outside the box
    syms = [elements[k] for k in rng.randint(0, len(elements), n)]
    return Atoms(symbols=syms, positions=pos, cell=cell, pbc=pbc)


def _check(res, refs):
    for r, o in zip(res, refs):
        assert abs(r["energy"] - o["energy"]) < E_TOL
        assert np.abs(r["forces"] - o["forces"]).max() < F_TOL
        assert np.abs(r["virial"] - o["virial"]).max() < W_TOL


@pytest.mark.parametrize("seed", range(6))
def test_symmetry_function_models(lib, seed):
    from tensoralloy_amd import Engine
    rng = np.random.RandomState(1000 + seed)
    elements = [["Ni"], ["Mo", "Ni"], ["Al", "Cu", "Ni"]][seed % 3]
    rc = rng.uniform(3.5, 6.0)
    angular = seed != 4
    kw = dict(eta=rng.uniform(0.05, 20.0, rng.randint(1, 5)).tolist(),
              omega=rng.uniform(0.0, 2.0, rng.randint(1, 3)).tolist(),
              beta=rng.uniform(0.005, 0.5, rng.randint(1, 3)).tolist(),
              gamma=[1.0, -1.0][:rng.randint(1, 3)], zeta=[[1.0, 4.0], [2.0], [1.0, 2.0, 8.0]][seed % 3])
    nn = make_nn(elements, rc, angular, [int(rng.randint(4, 40))] * rng.randint(1, 4),
                 activation=["softplus", "tanh", "squareplus"][seed % 3], acut=rc if seed % 2 else rc * 0.8,
                 minmax=bool(seed % 2), resnet=bool(seed % 3 == 0), cutoff=["cosine", "polynomial"][seed % 2],
                 sf_kwargs=kw, seed=seed)
    frames = [_random_frame(rng, elements) for _ in range(3)]
    with Engine(nn) as eng:
        res = eng.evaluate(frames)
    _check(res, [oracle_eval(nn, a) for a in frames])


@pytest.mark.parametrize("seed", range(4))
def test_eam_adp_models(lib, seed):
    from tensoralloy_amd import Engine
    rng = np.random.RandomState(2000 + seed)
    elements = [["Ni"], ["Mo", "Ni"], ["Al", "Cu"], ["Mo", "Ni"]][seed]
    nn = make_eam(elements, rng.uniform(4.5, 6.5), adp=seed in (0, 1),
                  potential=["zjw04", "zjw04", "zjw04xc", "zjw04xcp"][seed])
    frames = [_random_frame(rng, elements, min_dist=2.0) for _ in range(3)]
    with Engine(nn) as eng:
        res = eng.evaluate(frames)
    _check(res, [oracle_eam_eval(nn, a) for a in frames])


@pytest.mark.parametrize("seed", range(4))
def test_grap_models(lib, seed):
    from tensoralloy_amd import Engine
    rng = np.random.RandomState(3000 + seed)
    elements = [["Ni"], ["Mo", "Ni"], ["Be", "W"], ["Al", "Cu", "Ni"]][seed]
    algo, params, method = [
        ("pexp", {"rl": rng.uniform(1.0, 3.0, 7).tolist(), "pl": rng.uniform(1.0, 5.0, 7).tolist()}, "pair"),
        ("sf", {"eta": [0.1, 1.0, 4.0], "omega": [0.0, 1.0]}, "cross"),
        ("morse", {"D": [1.0, 0.5], "gamma": [0.5, 0.9], "r0": [1.5, 2.5]}, "cross"),
        ("density", {"A": [1.0, 2.0], "beta": [1.0, 3.0], "re": [3.0, 4.0]}, "pair")][seed]
    nn = make_grap_nn(elements, rng.uniform(4.0, 6.0), [int(rng.randint(8, 33))] * 2, algo, params,
                      moment_tensors=list(range(seed % 4 + 1)) if seed else [0, 1, 2, 3],
                      legacy_mode=seed == 2, symmetric=seed == 1, cutoff=["cosine", "polynomial"][seed % 2],
                      param_space_method=method, minmax=seed == 3, seed=seed)
    frames = [_random_frame(rng, elements) for _ in range(3)]
    with Engine(nn) as eng:
        res = eng.evaluate(frames)
    _check(res, [oracle_grap_eval(nn, a) for a in frames])


@pytest.mark.parametrize("seed", range(5))
def test_nn_functions_and_tables(lib, seed, tmp_path):
    """nn-EAM / nn-ADP with random shapes and activations, mixed with analytic and tabulated
    functions, and the GRAP filter network: random triclinic frames, uneven batches."""
    from tensoralloy_amd import Engine
    from tests.helpers import golden_setfl
    rng = np.random.RandomState(4000 + seed)
    if seed == 4:
        elements = ["Mo", "Ni"]
        par = {"hidden_sizes": [int(rng.randint(8, 40)) for _ in range(rng.randint(1, 4))],
               "num_filters": int(rng.randint(3, 20)), "activation": "softplus", "use_resnet_dt": True}
        nn = make_grap_nn(elements, rng.uniform(4.0, 5.5), [16], "nn", par, moment_tensors=[0, 1, 2, 3])
        frames = [_random_frame(rng, elements) for _ in range(3)]
        with Engine(nn) as eng:
            res = eng.evaluate(frames)
        _check(res, [oracle_grap_eval(nn, a) for a in frames])
        return
    elements = [["Ni"], ["Mo", "Ni"], ["Al", "Cu"], ["Al", "Cu", "Ni"]][seed]
    hs = [int(rng.randint(4, 70)) for _ in range(rng.randint(1, 4))]
    pots = None
    if seed == 2:   # tabulated + nn + analytic in one model
        path = golden_setfl("Zhou_AlCu.alloy.eam", tmp_path)
        pots = {"Al": {"rho": "spline@" + path, "embed": "nn"}, "Cu": {"rho": "nn", "embed": "zjw04"},
                "AlAl": {"phi": "nn"}, "AlCu": {"phi": "spline@" + path}, "CuCu": {"phi": "zjw04"}}
    nn = make_eam(elements, rng.uniform(4.5, 5.9), adp=seed == 1, potential=pots, hidden_sizes=hs,
                  activation=["softplus", "tanh", "softplus", "squareplus"][seed], seed=seed)
    frames = [_random_frame(rng, elements, min_dist=2.0) for _ in range(3)]
    with Engine(nn) as eng:
        res = eng.evaluate(frames)
    _check(res, [oracle_eam_eval(nn, a) for a in frames])
