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


def _oracle_total_loss(nn, frames, e_ref, f_ref, s_ref):
    from tensoralloy_amd.train import energy_loss, forces_loss, stress_loss
    outs = [oracle_eval(nn, a) for a in frames]
    n = np.array([len(a) for a in frames], dtype=float)
    le = energy_loss(np.array([o["energy"] for o in outs]), e_ref, n)[0]
    lf = forces_loss([o["forces"] for o in outs], f_ref)[0]
    ls = stress_loss(np.array([o["stress_voigt"] for o in outs]), s_ref)[0]
    return le + lf + ls


def test_force_and_stress_loss_gradient(lib):
    """Gradient of the full loss (energy + forces + stress RMSE) with respect to the weights: the
    GPU path (analytic energy gradient + directional central difference for the second-derivative
    terms) against central differences of the ORACLE's loss in single weights."""
    from tensoralloy_amd.train import Trainer, flatten_weights, unflatten_weights
    nn = make_nn(["Mo", "Ni"], 5.0, True, [8, 8], seed=3)
    teacher = make_nn(["Mo", "Ni"], 5.0, True, [8, 8], seed=11)
    frames = [_alloy(["Ni", "Ni", "Mo"], rep=(2, 2, 2), seed=1), _alloy(["Ni", "Mo"], rep=(2, 2, 2), seed=2)]
    refs = [oracle_eval(teacher, a) for a in frames]
    e_ref = np.array([o["energy"] for o in refs])
    f_ref = [o["forces"] for o in refs]
    s_ref = np.array([o["stress_voigt"] for o in refs])
    tr = Trainer(nn, frames, e_ref, f_ref, s_ref, device=0)
    total, terms, grad = tr.loss_and_gradient()
    tr.close()
    assert set(terms) == {"energy", "forces", "stress"}
    assert abs(total - _oracle_total_loss(nn, frames, e_ref, f_ref, s_ref)) < 1e-9
    theta = flatten_weights(nn)
    rng = np.random.RandomState(0)
    d = 1e-5
    for k in rng.choice(len(theta), 6, replace=False):
        if grad[k] == 0.0 and tr.mask[k] == 0.0:
            continue
        th = theta.copy(); th[k] += d
        nn.weights = unflatten_weights(nn, th)
        lp = _oracle_total_loss(nn, frames, e_ref, f_ref, s_ref)
        th[k] -= 2 * d
        nn.weights = unflatten_weights(nn, th)
        lm = _oracle_total_loss(nn, frames, e_ref, f_ref, s_ref)
        nn.weights = unflatten_weights(nn, theta)
        fd = (lp - lm) / (2 * d)
        assert abs(fd - grad[k]) < 2e-6 * max(1.0, abs(fd)), (k, fd, grad[k])


def test_fit_with_forces_lowers_force_error(lib):
    from tensoralloy_amd import Engine
    from tensoralloy_amd.train import Trainer
    teacher = make_nn(["Ni"], 5.0, True, [12, 12], seed=5)
    student = make_nn(["Ni"], 5.0, True, [12, 12], seed=77)
    frames = [fcc(rep=(2, 2, 2), a=3.4 + 0.05 * k, seed=k, jitter=0.1) for k in range(6)]
    with Engine(teacher) as eng:
        ref = eng.evaluate(frames)
    tr = Trainer(student, frames, [r["energy"] for r in ref], [r["forces"] for r in ref],
                 [r["stress"] for r in ref], device=0, learning_rate=0.01)
    hist = tr.fit(150)
    tr.close()
    assert hist[-1]["forces"] < 0.5 * hist[0]["forces"]
    assert hist[-1]["total"] < 0.5 * hist[0]["total"]


@pytest.mark.parametrize("kind", ["eam_ni", "eam_binary_mixed", "adp"])
def test_nn_eam_weight_gradient(lib, kind):
    """Gradient of sum_f c_f E_f with respect to the weights of the nn functions of an EAM / ADP
    model (rho, phi, embed, dipole, quadrupole networks): against central differences of the ORACLE's
    energies in single weights of every network, and `ta_update_weights` on the live handle."""
    from tensoralloy_amd import Engine
    from tensoralloy_amd.train import flatten_weights, trainable_mask, unflatten_weights
    from tests.helpers import make_eam, oracle_eam_eval
    if kind == "eam_ni":
        nn = make_eam(["Ni"], 6.0, potential=None, hidden_sizes=[16, 8])
        frames = [fcc(rep=(2, 2, 2)), fcc(rep=(2, 2, 3), a=3.4, seed=3)]
    elif kind == "eam_binary_mixed":
        pots = {"Ni": {"rho": "nn", "embed": "zjw04"}, "Mo": {"rho": "zjw04", "embed": "nn"},
                "NiNi": {"phi": "zjw04"}, "MoNi": {"phi": "nn"}, "MoMo": {"phi": "nn"}}
        nn = make_eam(["Mo", "Ni"], 6.0, potential=pots, hidden_sizes=[12])
        frames = [_alloy(["Ni", "Ni", "Mo"], rep=(2, 2, 2))]
    else:
        nn = make_eam(["Mo", "Ni"], 5.5, adp=True, potential=None, hidden_sizes=[8, 8])
        frames = [_alloy(["Ni", "Mo"], rep=(2, 2, 2))]
    coeff = np.random.RandomState(2).randn(len(frames))
    theta = flatten_weights(nn)
    mask = trainable_mask(nn)
    with Engine(nn) as eng:
        eng.set_frames(frames)
        assert eng.param_count() == len(theta)
        g = eng.energy_gradient(coeff) * mask
        # a second model's weights on the live handle
        eng.update_weights(theta * 0.5)
        e_half = eng.energies(reuse_descriptors=True)
    half = unflatten_weights(nn, theta * 0.5)
    saved = nn.weights
    nn.weights = half
    ref_half = np.array([oracle_eam_eval(nn, a)["energy"] for a in frames])
    nn.weights = saved
    assert np.abs(e_half - ref_half).max() < 1e-8 * max(1.0, np.abs(ref_half).max())

    def oracle_loss(vec):
        nn.weights = unflatten_weights(nn, vec)
        try:
            return sum(c * oracle_eam_eval(nn, a)["energy"] for a, c in zip(frames, coeff))
        finally:
            nn.weights = saved

    rng = np.random.RandomState(5)
    live = np.flatnonzero(mask)
    picks = list(rng.choice(live, size=10, replace=False)) + [live[0], live[-1]]
    scale = max(1.0, np.abs(g).max())
    for k in picks:
        d = 1e-5
        tp, tm = theta.copy(), theta.copy()
        tp[k] += d
        tm[k] -= d
        num = (oracle_loss(tp) - oracle_loss(tm)) / (2 * d)
        assert abs(g[k] - num) < 2e-6 * scale, (k, g[k], num)


def test_nn_eam_fit_recovers_teacher_energies(lib):
    from tensoralloy_amd import Engine
    from tensoralloy_amd.train import EnergyTrainer
    from tests.helpers import make_eam
    teacher = make_eam(["Ni"], 5.0, potential=None, hidden_sizes=[8], seed=1)
    frames = [fcc(rep=(2, 2, 2), a=a, seed=k) for k, a in enumerate((3.4, 3.5, 3.6, 3.7))]
    with Engine(teacher) as eng:
        labels = [r["energy"] for r in eng.evaluate(frames)]
    student = make_eam(["Ni"], 5.0, potential=None, hidden_sizes=[8], seed=2)
    tr = EnergyTrainer(student, frames, labels, device=0, learning_rate=0.001)
    hist = tr.fit(120)
    tr.close()
    # the RMSE loss has gradients of constant size near its minimum: Adam hovers around it
    assert min(hist) < 0.15 * hist[0] and np.mean(hist[-20:]) < 0.4 * hist[0]


@pytest.mark.parametrize("kind", ["sf_binary_minmax_resnet", "sf_single_tanh", "grap"])
def test_analytic_loss_gradient_matches_oracle(lib, kind):
    """`ta_loss_gradient`: d/dtheta (sum_f c_f E_f + D_delta E) in one analytic pass, in two parts.
    (1) The directional derivative dG of the descriptors (Jacobian from one-hot backward launches,
    pair sweep: linear maps) against a sixth-order central difference of the ORACLE's descriptors
    (accuracy of that difference: ~1e-7; the cutoffs sit between neighbour shells, because the
    cosine cutoff's second derivative jumps at rc and a difference quotient across it is poor). (2) The second-order sweep through the MLP against the
    oracle's restatement of the same recurrences (checked by finite differences in
    tests/test_train_cpu.py) on the SAME dG: 1e-9, where the finite-difference path it replaces had 2e-6."""
    from tensoralloy_amd import Atoms, Engine
    from oracle.train import tangent_weight_gradients, flatten
    grap = kind == "grap"
    if kind == "sf_binary_minmax_resnet":
        nn = make_nn(["Mo", "Ni"], 4.75, True, [16, 16], minmax=True, resnet=True)
        frames = [_alloy(["Ni", "Ni", "Mo"], rep=(2, 2, 2)), _alloy(["Ni", "Mo"], rep=(2, 2, 2), seed=8)]
    elif kind == "sf_single_tanh":
        nn = make_nn(["Ni"], 5.25, True, [24, 24], activation="tanh")
        frames = [fcc(rep=(2, 2, 2)), fcc(rep=(2, 2, 3), seed=4)]
    else:
        nn = make_grap_nn(["Mo", "Ni"], 4.75, [16, 16], moment_tensors=[0, 1, 2])
        frames = [_alloy(["Ni", "Ni", "Mo"], rep=(2, 2, 2))]
    rng = np.random.RandomState(5)
    coeff = rng.randn(len(frames))
    dR = [rng.randn(len(a), 3) * 0.3 for a in frames]
    dh = [rng.randn(3, 3) * 0.05 for _ in frames]
    with Engine(nn) as eng:
        eng.set_frames(frames)
        g, dG_gpu = eng.loss_gradient(coeff, np.concatenate(dR), np.array(dh), return_tangent=True)
        g_again = eng.loss_gradient(coeff, np.concatenate(dR), np.array(dh))   # Jacobian reused
        g_energy = eng.loss_gradient(coeff)                                     # = ta_energy_gradient
        assert np.array_equal(g_energy, eng.energy_gradient(coeff))
    assert np.abs(g - g_again).max() < 1e-12 * max(1.0, np.abs(g).max())
    m = oracle_grap_model(nn) if grap else oracle_model(nn)
    evaluate = oracle_grap_eval if grap else oracle_eval
    ref, a0 = None, 0
    for a, c, d_r, d_h in zip(frames, coeff, dR, dh):
        h0 = np.asarray(a.get_cell(complete=True), dtype=np.float64)

        def G_at(t):
            moved = Atoms(numbers=np.asarray(a.numbers).copy(), positions=a.positions + t * d_r,
                          cell=h0 + t * d_h, pbc=np.asarray(a.pbc).copy())
            return evaluate(nn, moved)["descriptors"]
        e = 1e-3
        dG_fd = (45.0 * (G_at(e) - G_at(-e)) - 9.0 * (G_at(2 * e) - G_at(-2 * e)) +
                 (G_at(3 * e) - G_at(-3 * e))) / (60.0 * e)
        mine = dG_gpu[a0:a0 + len(a)]
        assert np.abs(mine - dG_fd).max() < 1e-6 * max(1.0, np.abs(dG_fd).max())
        part = flatten(m, tangent_weight_gradients(m, a.get_chemical_symbols(), G_at(0.0), mine,
                                                   np.full(len(a), c)))
        ref = part if ref is None else ref + part
        a0 += len(a)
    assert np.abs(g - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("kind", ["zjw04", "alloy", "zjw04xc", "zjw04xcp", "sutton90", "be/1", "grimes", "adp", "adp_alloy", "mixed_embed_nn", "mixed_rho_phi_nn", "mixed_adp_nn"])
def test_empirical_constant_gradient(lib, kind):
    """The reference trains the constants of its empirical potentials (potentials.py:129-163).
    `ta_constant_gradient` = d/dconstants of  sum_f c_f E_f + sum u.F + sum Y:W  (the model-dependent
    part of the energy + forces + stress loss) in dual arithmetic; checked slot by slot against the
    7-point central difference of the same functional built from the ORACLE's energies, forces
    and virials at displaced constants."""
    import copy
    from tensoralloy_amd import Engine
    from tests.helpers import hcp, make_eam, oracle_eam_model
    from oracle.train import eam_loss_functional, central_difference_6
    rng = np.random.RandomState(11)
    if kind == "zjw04":
        nn, frames = make_eam(["Ni"], 6.0), [fcc(rep=(2, 2, 2), jitter=0.1), fcc(rep=(2, 2, 2), a=3.4, seed=3, jitter=0.05)]
    elif kind == "alloy":
        nn, frames = make_eam(["Mo", "Ni"], 6.0), [_alloy(["Ni", "Ni", "Ni", "Mo"], rep=(2, 2, 2))]
    elif kind == "zjw04xc":
        nn, frames = make_eam(["Ni"], 6.0, potential="zjw04xc"), [fcc(rep=(2, 2, 2), jitter=0.1)]
    elif kind == "zjw04xcp":
        nn, frames = make_eam(["Mo", "Ni"], 6.0, potential="zjw04xcp"), [_alloy(["Ni", "Ni", "Mo", "Mo"], rep=(2, 2, 2))]
    elif kind == "sutton90":
        nn, frames = make_eam(["Ag"], 7.0, potential="sutton90"), [fcc("Ag", a=4.09, rep=(2, 2, 2), jitter=0.08)]
    elif kind == "be/1":
        nn, frames = make_eam(["Be"], 5.0, potential="Be/1"), [hcp(rep=(3, 3, 3), jitter=0.05, seed=4)]
    elif kind == "adp":       # Zjw04 rho / phi / F + MishinH dipole and quadrupole functions (mishin.py:62-66)
        nn, frames = make_eam(["Ni"], 6.0, adp=True), [fcc(rep=(2, 2, 2), jitter=0.1), fcc(rep=(2, 2, 2), a=3.4, seed=3, jitter=0.05)]
    elif kind == "adp_alloy":
        # (a frame whose densities stay clear of the embedding thresholds: with Ni2Mo one Mo atom sits at
        # rho / rho_e = 1.15 + 6e-5 and any usable stencil straddles the branch)
        nn, frames = make_eam(["Mo", "Ni"], 6.0, adp=True), [_alloy(["Ni", "Ni", "Ni", "Mo"], rep=(2, 2, 2))]
    elif kind == "grimes":
        nn, frames = make_eam(["Pu"], 6.0, potential="grimes"), [fcc("Pu", a=4.6, rep=(2, 2, 2), jitter=0.08)]
    # round 3: models that mix networks with analytic functions (the reference trains every variable of such a
    # model, potentials.py:129-200); the networks enter the constants' gradient as plain functions
    elif kind == "mixed_embed_nn":
        pots = {"Ni": {"rho": "zjw04", "embed": "nn"}, "NiNi": {"phi": "zjw04"}}
        nn, frames = make_eam(["Ni"], 6.0, potential=pots, hidden_sizes=[8]), [fcc(rep=(2, 2, 2), jitter=0.1)]
    elif kind == "mixed_rho_phi_nn":
        pots = {"Ni": {"rho": "nn", "embed": "zjw04"}, "Mo": {"rho": "zjw04", "embed": "zjw04"},
                "NiNi": {"phi": "zjw04"}, "MoNi": {"phi": "nn"}, "MoMo": {"phi": "zjw04"}}
        nn, frames = make_eam(["Mo", "Ni"], 6.0, potential=pots, hidden_sizes=[8]), [_alloy(["Ni", "Ni", "Mo"], rep=(2, 2, 2))]
    else:
        pots = {"Ni": {"rho": "zjw04", "embed": "zjw04"}, "NiNi": {"phi": "zjw04", "dipole": "nn", "quadrupole": "mishinh"}}
        nn, frames = make_eam(["Ni"], 6.0, adp=True, potential=pots, hidden_sizes=[8]), [fcc(rep=(2, 2, 2), jitter=0.1)]
    F = len(frames)
    c = rng.normal(0, 1, F)
    u = [rng.normal(0, 0.3, (len(a), 3)) for a in frames]
    Y = []
    for _ in frames:
        y = rng.normal(0, 0.05, (3, 3))
        Y.append(0.5 * (y + y.T))
    dR = np.concatenate([a.positions @ Y[k] - u[k] for k, a in enumerate(frames)])
    dh = np.array([np.asarray(a.get_cell(complete=True)) @ Y[k] for k, a in enumerate(frames)])
    names = nn.constant_names()
    with Engine(nn) as eng:
        eng.set_frames(frames)
        flat = eng.constants()
        assert np.abs(flat - nn.constants()).max() == 0.0
        grad = eng.constant_gradient(c, dR, dh)
        grad_e = eng.constant_gradient(c, None, None)          # energy term alone
        # new constants reach the kernels: the energy moves by grad_e . delta to first order
        e0 = np.dot(c, eng.energies(reuse_descriptors=False))
        live = np.array([n is not None for n in names])
        delta = np.where(live, 1e-6 * rng.normal(0, 1, len(flat)) * np.maximum(np.abs(flat), 1e-2), 0.0)
        eng.update_constants(flat + delta)
        e1 = np.dot(c, eng.energies(reuse_descriptors=False))
        # (tolerance against the size of the individual terms: their sum may cancel, the second order does not)
        assert abs((e1 - e0) - np.dot(grad_e, delta)) < 1e-3 * np.sum(np.abs(grad_e * delta)) + 1e-12
    oframes = [(a.get_chemical_symbols(), a.positions, np.asarray(a.get_cell(complete=True)), a.pbc) for a in frames]

    def functional(slot, value, energy_only=False):
        trial = copy.deepcopy(nn)
        v = flat.copy()
        v[slot] = value
        trial.set_constants(v)
        zero_u = [np.zeros_like(x) for x in u]
        zero_Y = [np.zeros((3, 3)) for _ in Y]
        return eam_loss_functional(oracle_eam_model(trial), oframes, c, zero_u if energy_only else u,
                                   zero_Y if energy_only else Y)
    scale = max(np.abs(grad).max(), 1e-30)
    L0 = functional(0, flat[0])
    checked = 0
    for slot, name in enumerate(names):
        if name is None:
            assert grad[slot] == 0.0 and grad_e[slot] == 0.0
            continue
        # small step: the piecewise Zjw04 embedding is only piecewise smooth in the constants (an atom
        # whose rho crosses 0.85 / 1.15 rho_e inside the stencil spoils the difference, not the kernel)
        h = 1e-4 * max(abs(flat[slot]), 0.05)
        fd = central_difference_6(lambda x: functional(slot, x), flat[slot], h)
        tol = 2e-8 * max(abs(fd), 1e-3 * scale) + 1e-9 + 1e-13 * abs(L0) / h    # last: round-off of the stencil
        assert abs(grad[slot] - fd) < tol, (name, grad[slot], fd)
        if slot % 5 == 0:
            fd_e = central_difference_6(lambda x: functional(slot, x, True), flat[slot], h)
            assert abs(grad_e[slot] - fd_e) < 2e-8 * max(abs(fd_e), 1e-3 * scale) + 1e-9, (name, grad_e[slot], fd_e)
        checked += 1
    assert checked >= 2


def test_fit_of_empirical_constants_recovers_a_teacher(lib):
    """Trainer on an all-analytic EAM model: the constants are the parameters (as the reference's
    potentials are trained, potentials.py:129-163); `fixed` entries and unused slots do not move."""
    from tensoralloy_amd import Engine
    from tensoralloy_amd.train import Trainer
    from tests.helpers import make_eam
    teacher = make_eam(["Ni"], 6.0)
    frames = [fcc(rep=(2, 2, 2), a=3.4 + 0.04 * k, seed=k, jitter=0.08) for k in range(6)]
    with Engine(teacher) as eng:
        ref = eng.evaluate(frames)
    student = make_eam(["Ni"], 6.0)
    start = student.constants()
    names = student.constant_names()
    off = start.copy()
    for k, n in enumerate(names):
        if n is not None and n[1] in ("f_eq", "A", "B", "F0", "F1", "F2"):
            off[k] *= 1.04
    student.set_constants(off)
    tr = Trainer(student, frames, [r["energy"] for r in ref], [r["forces"] for r in ref],
                 [r["stress"] for r in ref], learning_rate=2e-3, fixed={"Ni": ["r_eq", "rho_e", "rho_s"]})
    assert tr.constants_mode and tr.analytic
    first = tr.step()[0]
    hist = tr.fit(150)
    tr.close()
    assert hist[-1]["total"] < 0.2 * first
    end = student.constants()
    for k, n in enumerate(names):
        if n is None or n[1] in ("r_eq", "rho_e", "rho_s"):
            assert end[k] == off[k]
    assert np.abs(end - off).max() > 0


def test_per_pair_entries_on_a_skin_filtered_batch(lib):
    """with a Verlet skin the kernels run on the exact list extracted from the
    resident skin list, whose arrays are defined only up to their own pair count. The per-pair
    gradient entry refuses such a batch instead of walking the undefined tail, `pairs()` hands out
    the resident (skin) list with `info.n_pairs` entries, and with the skin back at 0 the gradient
    is the one an unfiltered engine gives."""
    from tensoralloy_amd import Engine
    from tests.helpers import fcc
    nn = make_nn(["Ni"], 4.5, True, [8, 8], seed=5)
    frames = [fcc(rep=(3, 3, 3), seed=3)]
    n = len(frames[0])
    rng = np.random.RandomState(1)
    dR, c = rng.normal(size=(n, 3)), np.array([0.3])
    with Engine(nn) as eng:
        eng.set_frames(frames)
        eng.compute(3)
        ref = eng.loss_gradient(c, dR)
        n_exact = int(eng.info.n_pairs)
        eng.set_skin(0.5)
        info = eng.set_frames(frames)
        eng.compute(3)
        assert int(info.n_pairs) > n_exact
        with pytest.raises(ValueError, match="skin-filtered"):
            eng.loss_gradient(c, dR)
        i, j, s = eng.pairs()
        assert len(i) == int(info.n_pairs) and i.min() >= 0 and i.max() < n and j.min() >= 0 and j.max() < n
        d = frames[0].positions[j] - frames[0].positions[i] + s @ np.asarray(frames[0].get_cell(complete=True))
        r = np.linalg.norm(d, axis=1)
        assert r.max() < 4.5 + 0.5 and (r < 4.5).sum() == n_exact
        eng.set_skin(0.0)
        eng.set_frames(frames)
        eng.compute(3)
        again = eng.loss_gradient(c, dR)
    assert np.abs(again - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())


def test_trainer_shortcut_notices_a_foreign_batch(lib):
    """`Trainer.engine` is public; an evaluation of other frames between two steps
    (a validation pass) must not let the next step reuse the 'resident' batch."""
    from tensoralloy_amd.train import Trainer
    from tests.helpers import fcc
    nn = make_nn(["Ni"], 4.5, True, [8], seed=5)
    teacher = make_nn(["Ni"], 4.5, True, [8], seed=6)
    frames = [fcc(rep=(2, 2, 2), seed=k) for k in range(2)]
    refs = [oracle_eval(teacher, a) for a in frames]
    tr = Trainer(nn, frames, np.array([o["energy"] for o in refs]), [o["forces"] for o in refs], None, device=0)
    l0, _, g0 = tr.loss_and_gradient()
    tr.engine.evaluate([fcc(rep=(2, 2, 2), seed=77), fcc(rep=(2, 2, 2), seed=78)])  # same sizes, other geometry
    l1, _, g1 = tr.loss_and_gradient()
    tr.close()
    assert abs(l1 - l0) < 1e-12 and np.abs(g1 - g0).max() < 1e-10


def test_pressure_rrmse_and_l2_terms_in_the_trainer(lib):
    """The rest of the reference's total loss through the GPU trainer: total-pressure RMSE
    (nn/losses.py:459-505), relative RMSE of the forces (:53-68), L2 regulariser (:507-551), dynamic
    weights (:171-201). The gradient of the sum against central differences of the ORACLE's loss."""
    from tensoralloy_amd.train import (GPA, Trainer, energy_loss, flatten_weights, l2_regularization_loss,
                                       loss_weight_at, pressure_loss, relative_forces_loss, unflatten_weights)
    nn = make_nn(["Mo", "Ni"], 5.0, True, [8, 8], seed=3)
    teacher = make_nn(["Mo", "Ni"], 5.0, True, [8, 8], seed=11)
    frames = [_alloy(["Ni", "Ni", "Mo"], rep=(2, 2, 2), seed=1), _alloy(["Ni", "Mo"], rep=(2, 2, 2), seed=2)]
    refs = [oracle_eval(teacher, a) for a in frames]
    e_ref = np.array([o["energy"] for o in refs])
    f_ref = [o["forces"] for o in refs]
    p_ref = np.array([-(o["stress_voigt"][:3].sum()) / 3.0 / GPA for o in refs])  # nn/basic.py:394-408
    n = np.array([len(a) for a in frames], dtype=float)
    kw = dict(forces_weight=(0.5, 5.0), pressure_weight=0.01, max_train_steps=100)

    def oracle_loss(model):
        outs = [oracle_eval(model, a) for a in frames]
        le = energy_loss(np.array([o["energy"] for o in outs]), e_ref, n)[0]
        lf = relative_forces_loss([o["forces"] for o in outs], f_ref, loss_weight_at((0.5, 5.0), 0, 100))[0]
        pp = np.array([-(o["stress_voigt"][:3].sum()) / 3.0 / GPA for o in outs])
        lp = pressure_loss(pp, p_ref, "rmse", 0.01)[0]
        l2 = l2_regularization_loss(model, flatten_weights(model), l2_weight=0.2, weight=0.05, step=0)[0]
        return le + lf + lp + l2

    tr = Trainer(nn, frames, e_ref, f_ref, None, device=0, pressures=p_ref, forces_method="rrmse", l2_weight=0.2,
                 l2_loss_weight=0.05, **kw)
    total, terms, grad = tr.loss_and_gradient()
    tr.close()
    assert set(terms) == {"energy", "forces", "pressure", "l2"}
    assert abs(total - oracle_loss(nn)) < 1e-9
    theta = flatten_weights(nn)
    d = 1e-5
    for k in np.random.RandomState(1).choice(len(theta), 6, replace=False):
        if tr.mask[k] == 0.0:
            continue
        th = theta.copy(); th[k] += d
        nn.weights = unflatten_weights(nn, th)
        lp = oracle_loss(nn)
        th[k] -= 2 * d
        nn.weights = unflatten_weights(nn, th)
        lm = oracle_loss(nn)
        nn.weights = unflatten_weights(nn, theta)
        fd = (lp - lm) / (2 * d)
        assert abs(fd - grad[k]) < 2e-6 * max(1.0, abs(fd)), (k, fd, grad[k])


@pytest.mark.parametrize("kind", ["eam_ni", "eam_binary_mixed", "eam_setfl_embed", "adp", "adp_mixed"])
def test_nn_eam_analytic_force_stress_loss_gradient(lib, kind):
    """Round 3: d/dtheta [sum_f c_f E_f + D_(dR, dh) E] for the nn functions of a plain EAM model in one
    second-order pass per network (`ta_loss_gradient`, ta_eam.hip::eam_loss_gradient), against (a) the
    central difference of `ta_energy_gradient` on displaced frames that round 2 used and (b) central
    differences of the ORACLE's energies, in the direction and in single weights."""
    from tensoralloy_amd import Atoms, Engine
    from tensoralloy_amd.train import flatten_weights, trainable_mask, unflatten_weights
    from tests.helpers import make_eam, oracle_eam_eval
    if kind == "eam_ni":
        nn = make_eam(["Ni"], 6.0, potential=None, hidden_sizes=[16, 8])
        frames = [fcc(rep=(2, 2, 2), jitter=0.08), fcc(rep=(2, 2, 3), a=3.4, seed=3, jitter=0.05)]
    elif kind == "eam_binary_mixed":
        pots = {"Ni": {"rho": "nn", "embed": "zjw04"}, "Mo": {"rho": "zjw04", "embed": "nn"},
                "NiNi": {"phi": "zjw04"}, "MoNi": {"phi": "nn"}, "MoMo": {"phi": "nn"}}
        nn = make_eam(["Mo", "Ni"], 6.0, potential=pots, hidden_sizes=[12])
        frames = [_alloy(["Ni", "Ni", "Mo"], rep=(2, 2, 2))]
    elif kind == "eam_setfl_embed":
        pots = {"Ni": {"rho": "nn", "embed": "zjw04"}, "NiNi": {"phi": "nn"}}
        nn = make_eam(["Ni"], 5.5, potential=pots, hidden_sizes=[8, 8])
        frames = [fcc(rep=(2, 2, 2), jitter=0.1, seed=7)]
    elif kind == "adp":   # every function a network, dipole and quadrupole included
        nn = make_eam(["Mo", "Ni"], 5.5, adp=True, potential=None, hidden_sizes=[8, 8])
        frames = [_alloy(["Ni", "Mo"], rep=(2, 2, 2))]
    else:                 # analytic MishinH dipole, quadrupole network
        pots = {"Ni": {"rho": "nn", "embed": "nn"}, "NiNi": {"phi": "zjw04", "dipole": "mishinh", "quadrupole": "nn"}}
        nn = make_eam(["Ni"], 5.5, adp=True, potential=pots, hidden_sizes=[8])
        frames = [fcc(rep=(2, 2, 2), jitter=0.1, seed=9)]
    rng = np.random.RandomState(11)
    coeff = rng.randn(len(frames))
    dR = [rng.randn(len(a), 3) * 0.3 for a in frames]
    dh = [rng.randn(3, 3) * 0.2 for _ in frames]
    theta = flatten_weights(nn)
    mask = trainable_mask(nn)

    def displaced(eps):
        return [Atoms(numbers=np.asarray(a.numbers).copy(), positions=a.positions + eps * dR[k],
                      cell=np.asarray(a.get_cell(complete=True)) + eps * dh[k], pbc=np.asarray(a.pbc).copy())
                for k, a in enumerate(frames)]

    with Engine(nn) as eng:
        eng.set_frames(frames)
        g = eng.loss_gradient(coeff, np.concatenate(dR), np.array(dh)) * mask
        g_dir_only = eng.loss_gradient(None, np.concatenate(dR), np.array(dh)) * mask
        g_e = eng.energy_gradient(coeff) * mask
        e = 1e-4
        eng.set_frames(displaced(e))
        gp = eng.energy_gradient(np.ones(len(frames)))
        eng.set_frames(displaced(-e))
        gm = eng.energy_gradient(np.ones(len(frames)))
    fd = (gp - gm) / (2 * e) * mask
    scale = max(1.0, np.abs(g).max())
    assert np.abs(g_dir_only - fd).max() < 2e-6 * scale
    assert np.abs(g - (g_e + fd)).max() < 2e-6 * scale
    assert np.abs(g_dir_only).max() > 1e-3          # the direction really couples to the weights

    saved = nn.weights

    def oracle_J(vec):
        nn.weights = unflatten_weights(nn, vec)
        try:
            out = 0.0
            for k, a in enumerate(frames):
                out += coeff[k] * oracle_eam_eval(nn, a)["energy"]
            d = 1e-4
            ep = sum(oracle_eam_eval(nn, a)["energy"] for a in displaced(d))
            em = sum(oracle_eam_eval(nn, a)["energy"] for a in displaced(-d))
            return out + (ep - em) / (2 * d)
        finally:
            nn.weights = saved

    live = np.flatnonzero(mask)
    picks = list(np.random.RandomState(5).choice(live, size=6, replace=False)) + [live[0], live[-1]]
    for k in picks:
        d = 1e-4
        tp, tm = theta.copy(), theta.copy()
        tp[k] += d
        tm[k] -= d
        num = (oracle_J(tp) - oracle_J(tm)) / (2 * d)
        assert abs(g[k] - num) < 2e-5 * scale, (k, g[k], num)


def test_trainer_uses_the_analytic_pass_for_nn_eam(lib):
    from tensoralloy_amd import Engine
    from tensoralloy_amd.train import Trainer
    from tests.helpers import make_eam
    teacher = make_eam(["Ni"], 5.0, potential=None, hidden_sizes=[8], seed=1)
    frames = [fcc(rep=(2, 2, 2), a=a, seed=k, jitter=0.05) for k, a in enumerate((3.45, 3.55, 3.65))]
    with Engine(teacher) as eng:
        res = eng.evaluate(frames)
    student = make_eam(["Ni"], 5.0, potential=None, hidden_sizes=[8], seed=2)
    kw = dict(energies=[r["energy"] for r in res], forces=[r["forces"] for r in res],
              stresses=[r["stress"] for r in res], device=0, learning_rate=0.002)
    tr = Trainer(student, frames, **kw)
    assert tr.analytic
    la, _, ga = tr.loss_and_gradient()
    tr.close()
    student2 = make_eam(["Ni"], 5.0, potential=None, hidden_sizes=[8], seed=2)
    tf = Trainer(student2, frames, analytic=False, fd_step=1e-4, **kw)
    lf, _, gf = tf.loss_and_gradient()
    tf.close()
    assert abs(la - lf) < 1e-12 * max(1.0, abs(lf))
    assert np.abs(np.asarray(ga) - np.asarray(gf)).max() < 2e-6 * max(1.0, np.abs(gf).max())


def test_trainer_trains_weights_and_constants_of_a_mixed_model(lib):
    """A model that mixes networks with analytic functions: the reference trains every variable
    (potentials/potentials.py:129-200). theta = [weights | constants]; the two halves of the gradient
    against central differences of the trainer's own loss, then a short fit that moves both."""
    from tensoralloy_amd import Engine
    from tensoralloy_amd.train import Trainer
    from tests.helpers import make_eam
    pots = {"Ni": {"rho": "zjw04", "embed": "nn"}, "NiNi": {"phi": "zjw04"}}
    teacher = make_eam(["Ni"], 5.8, potential=pots, hidden_sizes=[8], seed=1)
    frames = [fcc(rep=(2, 2, 2), a=a, seed=k, jitter=0.05) for k, a in enumerate((3.45, 3.55, 3.65))]
    with Engine(teacher) as eng:
        res = eng.evaluate(frames)
    student = make_eam(["Ni"], 5.8, potential=pots, hidden_sizes=[8], seed=2)
    c0 = student.constants()
    student.set_constants(c0 * (1.0 + 0.02 * np.random.RandomState(3).randn(len(c0))))
    tr = Trainer(student, frames, energies=[r["energy"] for r in res], forces=[r["forces"] for r in res],
                 stresses=[r["stress"] for r in res], device=0, learning_rate=0.002, fixed={"Ni": ["r_eq"]})
    assert tr.mixed and tr.analytic
    nw = tr._n_weights
    theta0 = tr.theta.copy()
    l0, _, g = tr.loss_and_gradient()
    assert len(g) == len(theta0) and np.abs(g[:nw]).max() > 0 and np.abs(g[nw:]).max() > 0

    def loss_at(vec):
        tr.engine.update_weights(vec[:nw])
        tr.engine.update_constants(vec[nw:])
        return tr.loss_and_gradient()[0]
    live = np.flatnonzero(tr.mask * (np.abs(g) > 1e-3 * np.abs(g).max()))
    rng = np.random.RandomState(5)
    picks = list(rng.choice(live[live < nw], 3, replace=False)) + list(rng.choice(live[live >= nw], 3, replace=False))
    for k in picks:
        d = 1e-5 * max(abs(theta0[k]), 0.05)
        tp, tm = theta0.copy(), theta0.copy()
        tp[k] += d
        tm[k] -= d
        num = (loss_at(tp) - loss_at(tm)) / (2 * d)
        assert abs(g[k] - num) < 2e-4 * max(abs(num), np.abs(g).max() * 1e-3), (k, g[k], num)
    loss_at(theta0)
    hist = tr.fit(60)
    assert hist[-1]["total"] < 0.7 * hist[0]["total"]
    assert np.abs(tr.theta[nw:] - theta0[nw:]).max() > 0 and np.abs(tr.theta[:nw] - theta0[:nw]).max() > 0
    names = student.constant_names()
    k_fixed = [i for i, n in enumerate(names) if n == ("Ni", "r_eq")]
    assert all(tr.theta[nw + i] == theta0[nw + i] for i in k_fixed)
    tr.close()
