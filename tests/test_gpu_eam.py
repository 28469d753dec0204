"""GPU parity of the EAM / ADP kernels against the CPU oracle (oracle/eam.py)."""
import numpy as np
import pytest

from tests.helpers import fcc, make_eam, oracle_eam_eval
from tests.test_gpu_sf import _alloy, E_TOL, F_TOL, W_TOL

pytestmark = pytest.mark.gpu


def _compare(nn, atoms_list):
    from tensoralloy_amd import Engine
    with Engine(nn) as eng:
        res = eng.evaluate(atoms_list)
    for atoms, r in zip(atoms_list, res):
        o = oracle_eam_eval(nn, atoms)
        assert abs(r["energy"] - o["energy"]) < E_TOL
        assert np.abs(r["atomic"] - o["atomic"]).max() < E_TOL
        assert np.abs(r["forces"] - o["forces"]).max() < F_TOL
        assert np.abs(r["virial"] - o["virial"]).max() < W_TOL
        assert np.abs(r["stress"] - o["stress_voigt"]).max() < 1e-8


def test_eam_ni(lib):
    _compare(make_eam(["Ni"], 6.5), [fcc(rep=(3, 3, 3)), fcc(rep=(2, 2, 2), a=3.3, seed=5)])


def test_eam_ni_rc6(lib):
    _compare(make_eam(["Ni"], 6.0), [fcc(rep=(2, 2, 2), a=3.52, jitter=0.1)])


def test_eam_binary_ni_mo(lib):
    _compare(make_eam(["Ni", "Mo"], 6.5), [_alloy(["Ni", "Ni", "Ni", "Mo"], rep=(2, 2, 3))])


def test_eam_ternary(lib):
    _compare(make_eam(["Al", "Cu", "Ni"], 6.0), [_alloy(["Al", "Cu", "Ni"], rep=(2, 2, 2), a=3.8)])


def test_adp_ni(lib):
    _compare(make_eam(["Ni"], 6.5, adp=True), [fcc(rep=(3, 3, 3), jitter=0.08)])


def test_adp_binary(lib):
    _compare(make_eam(["Mo", "Ni"], 6.0, adp=True), [_alloy(["Ni", "Ni", "Mo"], rep=(2, 2, 2))])


def test_zjw04xc_blended_embedding(lib):
    # densities on both sides of 0.85 rho_e and 1.15 rho_e: compressed, equilibrium, expanded
    frames = [fcc(rep=(2, 2, 2), a=a, seed=3) for a in (3.3, 3.52, 3.8)]
    _compare(make_eam(["Ni"], 6.0, potential="zjw04xc"), frames)
    _compare(make_eam(["Al", "Cu"], 6.5, potential="zjw04uxc"), [_alloy(["Al", "Cu"], rep=(2, 2, 2), a=3.9)])


def test_zjw04xcp_cross_term(lib):
    _compare(make_eam(["Mo", "Ni"], 6.5, potential="zjw04xcp"), [_alloy(["Ni", "Ni", "Ni", "Mo"], rep=(2, 2, 3))])
    with pytest.raises(ValueError, match="no phi constants"):
        make_eam(["Cu", "Ni"], 6.5, potential="zjw04xcp").to_desc()
