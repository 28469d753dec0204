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
