"""GPU parity of the weight-gradient path (csrc/ta_train.hip) against oracle/train.py, and a short
fit on a teacher model's energies."""
import numpy as np
import pytest

from tests.helpers import fcc, make_nn, make_grap_nn, oracle_model, oracle_grap_model, oracle_eval, oracle_grap_eval
from tests.test_gpu_sf import _alloy

pytestmark = pytest.mark.gpu


def _oracle_gradient(nn, frames, coeff, grap=False):
    from oracle.train import weight_gradients, flatten
    m = oracle_grap_model(nn) if grap else oracle_model(nn)
    total = None
    for atoms, c in zip(frames, coeff):
        o = (oracle_grap_eval if grap else oracle_eval)(nn, atoms)
        g = flatten(m, weight_gradients(m, atoms.get_chemical_symbols(), o["descriptors"] /
                                        (nn.descriptor_scale() if hasattr(nn, "descriptor_scale") else 1.0),
                                        np.full(len(atoms), c)))
        total = g if total is None else total + g
    return total


@pytest.mark.parametrize("kind", ["sf_binary_minmax_resnet", "sf_single", "grap"])
def test_weight_gradient_matches_oracle(lib, kind):
    from tensoralloy_amd import Engine
    if kind == "sf_binary_minmax_resnet":
        nn = make_nn(["Mo", "Ni"], 6.0, True, [16, 16], minmax=True, resnet=True)
        frames = [_alloy(["Ni", "Ni", "Mo"], rep=(2, 2, 2)), _alloy(["Ni", "Mo"], rep=(2, 2, 3), seed=8)]
    elif kind == "sf_single":
        nn = make_nn(["Ni"], 6.5, True, [64, 64])
        frames = [fcc(rep=(3, 3, 3)), fcc(rep=(2, 2, 2), seed=4), fcc(rep=(2, 3, 2), seed=5)]
    else:
        nn = make_grap_nn(["Mo", "Ni"], 6.0, [24, 24], moment_tensors=[0, 1, 2])
        frames = [_alloy(["Ni", "Ni", "Mo"], rep=(2, 2, 2))]
    coeff = np.random.RandomState(2).randn(len(frames))
    with Engine(nn) as eng:
        eng.set_frames(frames)
        g = eng.energy_gradient(coeff)
        assert len(g) == eng.param_count()
    ref = _oracle_gradient(nn, frames, coeff, grap=(kind == "grap"))
    assert np.abs(g - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())


def test_update_weights_and_reuse_of_descriptors(lib):
    from tensoralloy_amd import Engine
    from tensoralloy_amd.train import flatten_weights
    nn = make_nn(["Ni"], 6.0, True, [16, 16])
    other = make_nn(["Ni"], 6.0, True, [16, 16], seed=99)
    frames = [fcc(rep=(2, 2, 2)), fcc(rep=(2, 2, 3), seed=3)]
    with Engine(nn) as eng:
        eng.set_frames(frames)
        e0 = eng.energies(reuse_descriptors=False)
        eng.update_weights(flatten_weights(other))
        e1 = eng.energies(reuse_descriptors=True)        # MLP only
        full = eng.evaluate(frames)                       # everything again, forces included
    with Engine(other) as eng2:
        ref = eng2.evaluate(frames)
    assert np.abs(e0 - e1).max() > 1e-3
    for k in range(2):
        assert abs(e1[k] - ref[k]["energy"]) < 1e-10
        assert np.abs(full[k]["forces"] - ref[k]["forces"]).max() < 1e-10


def test_short_fit_recovers_a_teacher(lib):
    from tensoralloy_amd import Engine
    from tensoralloy_amd.train import EnergyTrainer
    teacher = make_nn(["Ni"], 6.0, True, [16, 16], seed=5)
    student = make_nn(["Ni"], 6.0, True, [16, 16], seed=77)
    frames = [fcc(rep=(2, 2, 2), a=3.3 + 0.05 * k, seed=k, jitter=0.08) for k in range(12)]
    with Engine(teacher) as eng:
        labels = [r["energy"] for r in eng.evaluate(frames)]
    tr = EnergyTrainer(student, frames, labels, device=0, learning_rate=0.01)
    hist = tr.fit(300)
    tr.close()
    assert hist[-1] < 0.1 * hist[0]
    with Engine(student) as eng:   # the fitted weights were written back into `student`
        pred = np.array([r["energy"] for r in eng.evaluate(frames)])
    per_atom = np.abs(pred - np.array(labels)) / 32
    assert per_atom.max() < 5 * hist[-1] + 1e-6
