"""Host logic of the training support (no GPU): loss derivative, Adam, weight layout, the oracle's
weight gradient against finite differences, and the gradient all-reduce on two gloo ranks."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from tests.helpers import fcc, make_nn, oracle_model
from tests.test_gpu_sf import _alloy


def test_energy_loss_matches_its_derivative():
    from tensoralloy_amd.train import energy_loss
    rng = np.random.RandomState(0)
    pred, lab, n = rng.randn(7) * 3, rng.randn(7) * 3, rng.randint(2, 40, 7)
    for method in ("rmse", "logcosh"):
        for per_atom in (True, False):
            loss, mae, dl = energy_loss(pred, lab, n, method, per_atom, weight=0.7)
            for k in (0, 3, 6):
                d = 1e-6
                p = pred.copy(); p[k] += d
                lp = energy_loss(p, lab, n, method, per_atom, weight=0.7)[0]
                p = pred.copy(); p[k] -= d
                lm = energy_loss(p, lab, n, method, per_atom, weight=0.7)[0]
                assert abs((lp - lm) / (2 * d) - dl[k]) < 1e-8
    # losses.py:88-90: sqrt(mean(diff^2) + eps)
    x, y, nn_ = np.array([1.0, 2.0]), np.array([1.5, 1.0]), np.array([2, 4])
    loss, mae, _ = energy_loss(y, x, nn_)
    diff = y / nn_ - x / nn_
    assert abs(loss - np.sqrt(np.mean(diff ** 2) + np.finfo(float).eps)) < 1e-15
    assert abs(mae - np.mean(np.abs(diff))) < 1e-15


def test_adam_first_steps():
    from tensoralloy_amd.train import Adam
    opt = Adam(2, learning_rate=0.1)
    th = opt.step(np.array([1.0, -1.0]), np.array([0.5, -2.0]))
    assert np.allclose(th, [0.9, -0.9], atol=1e-6)   # first Adam step = lr * sign(g)
    opt = Adam(1, learning_rate=0.1, decay_rate=0.5, decay_steps=10)
    opt.t = 10
    assert abs(opt.learning_rate() - 0.05) < 1e-15


def test_weight_layout_roundtrip():
    from tensoralloy_amd.train import flatten_weights, unflatten_weights, trainable_mask
    nn = make_nn(["Mo", "Ni"], 6.0, True, [8, 8])
    nn.weights["Mo"][0] = (nn.weights["Mo"][0][0], None)           # a layer without bias
    flat = flatten_weights(nn)
    back = unflatten_weights(nn, flat)
    for el in nn.elements:
        for (w, b), (w2, b2) in zip(nn.weights[el], back[el]):
            assert np.array_equal(np.asarray(w), w2)
            assert (b is None and b2 is None) or np.array_equal(np.ravel(b), b2)
    D = nn.ndim()
    assert len(flat) == 2 * (D * 8 + 8 + 64 + 8 + 8 + 1)
    assert trainable_mask(nn)[D * 8:D * 8 + 8].sum() == 0 and trainable_mask(nn).sum() == len(flat) - 8


def test_oracle_weight_gradient_finite_differences():
    from oracle.sf import evaluate
    from oracle.train import weight_gradients
    nn = make_nn(["Mo", "Ni"], 6.0, True, [6, 6], minmax=True, resnet=True)
    atoms = _alloy(["Ni", "Ni", "Mo"], rep=(2, 2, 2))
    m = oracle_model(nn)
    sym, cell = atoms.get_chemical_symbols(), np.asarray(atoms.get_cell())
    out = evaluate(m, sym, atoms.positions, cell, atoms.pbc, want_forces=False)
    c = np.random.RandomState(1).randn(len(atoms))
    g = weight_gradients(m, sym, out["descriptors"], c)
    d = 1e-6
    for el, l, idx in (("Ni", 0, (2, 3)), ("Mo", 1, (0, 5)), ("Ni", 2, (4, 0))):
        W = m.weights[el][l][0]
        keep = W[idx]
        W[idx] = keep + d
        ep = float(c @ evaluate(m, sym, atoms.positions, cell, atoms.pbc, want_forces=False)["atomic"])
        W[idx] = keep - d
        em = float(c @ evaluate(m, sym, atoms.positions, cell, atoms.pbc, want_forces=False)["atomic"])
        W[idx] = keep
        assert abs((ep - em) / (2 * d) - g[el][l][0][idx]) < 1e-7
    b = m.weights["Ni"][1][1]
    keep = b[2]
    b[2] = keep + d
    ep = float(c @ evaluate(m, sym, atoms.positions, cell, atoms.pbc, want_forces=False)["atomic"])
    b[2] = keep - d
    em = float(c @ evaluate(m, sym, atoms.positions, cell, atoms.pbc, want_forces=False)["atomic"])
    b[2] = keep
    assert abs((ep - em) / (2 * d) - g["Ni"][1][1][2]) < 1e-7


def test_gradient_allreduce_two_gloo_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        import numpy as np
        sys.path.insert(0, %r)
        import torch.distributed as dist
        from tensoralloy_amd.train import Adam, allreduce_mean
        dist.init_process_group("gloo")
        rank = dist.get_rank()
        g = np.arange(5, dtype=float) * (rank + 1)          # rank 0: k, rank 1: 2k -> mean 1.5 k
        m = allreduce_mean(g)
        assert np.allclose(m, 1.5 * np.arange(5)), m
        opt = Adam(5, learning_rate=0.1)
        th = opt.step(np.ones(5), m)
        out = np.zeros(5) if rank else th
        import torch
        t = torch.from_numpy(th.copy()); dist.broadcast(t, 0)
        assert np.allclose(t.numpy(), th)                   # identical weights on both ranks
        dist.destroy_process_group()
        print("ok", rank)
    """ % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29547", str(script)],
                       capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


def test_forces_and_stress_loss_derivatives():
    from tensoralloy_amd.train import forces_loss, stress_loss
    rng = np.random.RandomState(5)
    pred = [rng.randn(4, 3), rng.randn(7, 3)]
    lab = [rng.randn(4, 3), rng.randn(7, 3)]
    for method in ("rmse", "logcosh"):
        loss, mae, d = forces_loss(pred, lab, method, weight=2.0)
        h = 1e-6
        p = [x.copy() for x in pred]; p[1][3, 2] += h
        lp = forces_loss(p, lab, method, weight=2.0)[0]
        p[1][3, 2] -= 2 * h
        lm = forces_loss(p, lab, method, weight=2.0)[0]
        assert abs((lp - lm) / (2 * h) - d[1][3, 2]) < 1e-8
        s_pred, s_lab = rng.randn(3, 6) * 0.01, rng.randn(3, 6) * 0.01
        loss, mae, ds = stress_loss(s_pred, s_lab, method, weight=0.5)
        q = s_pred.copy(); q[2, 4] += h
        lp = stress_loss(q, s_lab, method, weight=0.5)[0]
        q[2, 4] -= 2 * h
        lm = stress_loss(q, s_lab, method, weight=0.5)[0]
        assert abs((lp - lm) / (2 * h) - ds[2, 4]) < 1e-8
    # nn/losses.py:285-332: one RMSE over every component of every real atom
    allc = np.concatenate([l - p for p, l in zip(pred, lab)])
    assert abs(forces_loss(pred, lab)[0] - np.sqrt(np.mean(allc ** 2) + np.finfo(float).eps)) < 1e-15


def test_oracle_second_order_gradient_against_finite_differences():
    """oracle.train.tangent_weight_gradients (the restatement the GPU's analytic force / stress loss
    gradient is compared with): d/dtheta [sum c y + (dMLP/dG) . dG] against central differences in
    single weights, for every activation family, with and without skip connections and min-max."""
    import copy
    from oracle.sf import apply_mlp
    from oracle import train as ot
    from tensoralloy_amd.train import flatten_weights, unflatten_weights
    from tests.helpers import make_nn, oracle_model
    rng = np.random.RandomState(0)
    symbols = ["Ni", "Mo", "Ni", "Ni", "Mo"]
    for act, res, mm in (("softplus", False, False), ("tanh", True, True), ("squareplus", True, False),
                         ("elu", False, True), ("sigmoid", False, False), ("softsign", True, False),
                         ("leaky_relu", False, False)):
        nn = make_nn(["Ni", "Mo"], 5.0, True, [12, 12], activation=act, resnet=res, minmax=mm)
        m = oracle_model(nn)
        D = nn.ndim()
        G, dG, c = rng.rand(5, D) * 2, rng.randn(5, D), rng.randn(5)

        def J(model):
            y, w = apply_mlp(model, symbols, G)
            return (c * y).sum() + (w * dG).sum()
        g = ot.flatten(m, ot.tangent_weight_gradients(m, symbols, G, dG, c))
        th = flatten_weights(nn)
        for k in rng.choice(len(th), 10, replace=False):
            vals = []
            for sgn in (1.0, -1.0):
                t2 = th.copy()
                t2[k] += sgn * 1e-5
                nn2 = copy.deepcopy(nn)
                nn2.weights = unflatten_weights(nn2, t2)
                vals.append(J(oracle_model(nn2)))
            fd = (vals[0] - vals[1]) / 2e-5
            assert abs(fd - g[k]) < 1e-8 * max(1.0, abs(fd)), (act, k)


def test_pressure_relative_force_l2_and_dynamic_weights():
    """The remaining terms of the reference's total loss (nn/losses.py): total-pressure RMSE / log-cosh
    (:459-505), relative RMSE of the forces (:53-68), the L2 regulariser with its decayed weight
    (:507-551; which variables it covers: convolutional.py:207-290) and the dynamic loss weights
    (:171-201): values by the formulas, derivatives by central differences."""
    from tensoralloy_amd.train import (flatten_weights, l2_regularization_loss, loss_weight_at, pressure_loss,
                                       relative_forces_loss)
    from tests.helpers import make_nn
    rng = np.random.RandomState(9)
    h = 1e-6
    p_pred, p_lab = rng.randn(5), rng.randn(5)
    for method in ("rmse", "logcosh"):
        loss, mae, d = pressure_loss(p_pred, p_lab, method, weight=3.0)
        q = p_pred.copy(); q[2] += h
        lp = pressure_loss(q, p_lab, method, weight=3.0)[0]
        q[2] -= 2 * h
        lm = pressure_loss(q, p_lab, method, weight=3.0)[0]
        assert abs((lp - lm) / (2 * h) - d[2]) < 1e-8
        assert abs(mae - np.mean(np.abs(p_pred - p_lab))) < 1e-15
    assert abs(pressure_loss(p_pred, p_lab)[0] - np.sqrt(np.mean((p_pred - p_lab) ** 2) + np.finfo(float).eps)) < 1e-15
    with pytest.raises(ValueError):
        pressure_loss(p_pred, p_lab, "rrmse")      # the reference asserts method != rrmse
    pred = [rng.randn(4, 3), rng.randn(6, 3)]
    lab = [rng.randn(4, 3), rng.randn(6, 3)]
    loss, _, d = relative_forces_loss(pred, lab, weight=0.7)
    P, L = np.concatenate(pred), np.concatenate(lab)
    assert abs(loss - 0.7 * np.mean(np.linalg.norm(L - P, axis=1) / np.linalg.norm(L, axis=1))) < 1e-15
    q = [x.copy() for x in pred]; q[1][2, 0] += h
    lp = relative_forces_loss(q, lab, weight=0.7)[0]
    q[1][2, 0] -= 2 * h
    lm = relative_forces_loss(q, lab, weight=0.7)[0]
    assert abs((lp - lm) / (2 * h) - d[1][2, 0]) < 1e-8
    # L2: kernels and biases of the hidden layers, the kernel (not the bias) of the output layer
    nn = make_nn(["Mo", "Ni"], 5.0, False, [6, 5])
    theta = flatten_weights(nn)
    expect = 0.0
    for el in nn.elements:
        layers = nn.weights[el]
        for l, (w, b) in enumerate(layers):
            expect += 0.5 * np.sum(np.asarray(w) ** 2)
            if b is not None and l < len(layers) - 1:
                expect += 0.5 * np.sum(np.asarray(b) ** 2)
    lam = 0.01 * 0.99 ** (250 / 1000)
    loss, g = l2_regularization_loss(nn, theta, l2_weight=0.3, weight=0.01, step=250)
    assert abs(loss - lam * 0.3 * expect) < 1e-15 * max(1.0, expect)
    k = int(np.argmax(np.abs(g)))
    t = theta.copy(); t[k] += h
    lp = l2_regularization_loss(nn, t, l2_weight=0.3, weight=0.01, step=250)[0]
    t[k] -= 2 * h
    lm = l2_regularization_loss(nn, t, l2_weight=0.3, weight=0.01, step=250)[0]
    assert abs((lp - lm) / (2 * h) - g[k]) < 1e-9
    assert abs(l2_regularization_loss(nn, theta, 0.3, 0.01, step=250, decayed=False)[0] - 0.01 * 0.3 * expect) < 1e-15 * expect
    # dynamic weights
    assert loss_weight_at(2.5) == 2.5
    assert abs(loss_weight_at((1.0, 100.0), 0, 1000) - 1.0) < 1e-12
    assert abs(loss_weight_at((1.0, 100.0), 500, 1000) - 10.0) < 1e-12          # linear in log10
    assert abs(loss_weight_at((1.0, 100.0), 500, 1000, logscale=False) - 50.5) < 1e-12
    with pytest.raises(ValueError):
        loss_weight_at((1.0, 2.0), 3)
