"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): NumPy gradient of sum_atoms c_atom * y_atom with
respect to the MLP weights, i.e. what `tf.gradients(loss, trainable_variables)` produces for the
energy term of the reference's loss (nn/losses.py:204-285 through nn/convolutional.py:257-290 and
nn/atomic/atomic.py:157-195). Parity unpinned (the reference stores no gradient fixture); checked by
finite differences in tests/test_train_cpu.py.
"""
import numpy as np

from .sf import activation


def weight_gradients(model, symbols, G, atom_coeff):
    """{element: [(dW, db), ...]} for E_c = sum_atoms atom_coeff * y_atom. `model` as for
    oracle.sf.apply_mlp (elements, weights, activation, use_resnet_dt, minmax)."""
    out = {}
    atom_coeff = np.asarray(atom_coeff, dtype=np.float64)
    for el in model.elements:
        idx = np.array([k for k, s in enumerate(symbols) if s == el], dtype=np.int64)
        layers = model.weights[el]
        grads = [(np.zeros_like(np.asarray(W, dtype=float)), np.zeros(np.shape(W)[1])) for W, _ in layers]
        if len(idx) == 0:
            out[el] = grads
            continue
        x = G[idx]
        if model.minmax is not None and el in model.minmax:
            xlo, xhi = model.minmax[el]
            den = xhi - xlo
            ok = den != 0.0
            x = np.where(ok, (xhi - x) / np.where(ok, den, 1.0), 0.0)
        xs, das = [x], []
        h = x
        for l, (W, b) in enumerate(layers[:-1]):
            z = h @ W + (b if b is not None else 0.0)
            a, da = activation(model.activation, z)
            res = (l > 0 and model.use_resnet_dt and W.shape[0] == W.shape[1])
            h = a + h if res else a
            das.append((da, res))
            xs.append(h)
        Wo, bo = layers[-1]
        delta = atom_coeff[idx][:, None] * np.ones((len(idx), 1))      # dE_c / dy
        grads[-1] = (xs[-1].T @ delta, delta.sum(axis=0))
        delta = delta @ np.asarray(Wo).T
        for l in range(len(layers) - 2, -1, -1):
            W, _ = layers[l]
            da, res = das[l]
            dz = delta * da
            grads[l] = (xs[l].T @ dz, dz.sum(axis=0))
            back = dz @ np.asarray(W).T
            delta = back + delta if res else back
        out[el] = grads
    return out


def flatten(model, grads):
    parts = []
    for el in model.elements:
        for dW, db in grads[el]:
            parts.append(np.ravel(dW))
            parts.append(np.ravel(db))
    return np.concatenate(parts)


def activation2(name, x):
    """Second derivative of the reference activations (nn/utils.py:39-74)."""
    name = name.lower()
    if name == "softplus":
        s = 1.0 / (1.0 + np.exp(-x))
        return s * (1.0 - s)
    if name in ("relu", "leaky_relu"):
        return np.zeros_like(x)
    if name == "tanh":
        t = np.tanh(x)
        return -2.0 * t * (1.0 - t * t)
    if name == "sigmoid":
        s = 1.0 / (1.0 + np.exp(-x))
        return s * (1.0 - s) * (1.0 - 2.0 * s)
    if name == "softsign":
        d = 1.0 + np.abs(x)
        return -2.0 * np.sign(x) / d ** 3
    if name == "elu":
        return np.where(x > 0, 0.0, np.exp(np.minimum(x, 0.0)))
    if name == "squareplus":
        s = np.sqrt(x * x + 4.0)
        return 2.0 / s ** 3
    raise ValueError(name)


def tangent_weight_gradients(model, symbols, G, dG, atom_coeff=None):
    """{element: [(dW, db), ...]} of  sum_atoms atom_coeff y_atom  +  J,  J = sum_atoms (dMLP/dG)(G_atom) . dG_atom:
    what `tf.gradients` yields for the forces / stress terms of the reference's loss
    (nn/losses.py:285-437; J = D_delta E is linear in forces and virial) when the descriptors' change
    `dG` along the direction delta is given. Forward carries value and tangent, the reverse sweep the
    adjoints of both (see tensoralloy_amd/csrc/ta_train.hip::mlp_grad2_kernel for the recurrences)."""
    out = {}
    N = len(symbols)
    atom_coeff = np.zeros(N) if atom_coeff is None else np.asarray(atom_coeff, dtype=np.float64)
    for el in model.elements:
        idx = np.array([k for k, s in enumerate(symbols) if s == el], dtype=np.int64)
        layers = model.weights[el]
        grads = [(np.zeros_like(np.asarray(W, dtype=float)), np.zeros(np.shape(W)[1])) for W, _ in layers]
        if len(idx) == 0:
            out[el] = grads
            continue
        x, xt = G[idx], dG[idx]
        if model.minmax is not None and el in model.minmax:
            xlo, xhi = model.minmax[el]
            den = xhi - xlo
            ok = den != 0.0
            safe = np.where(ok, den, 1.0)
            x = np.where(ok, (xhi - x) / safe, 0.0)
            xt = np.where(ok, -xt / safe, 0.0)
        xs, ts, das, dds, ress = [], [], [], [], []
        for l, (W, b) in enumerate(layers):
            W = np.asarray(W, dtype=float)
            xs.append(x)
            ts.append(xt)
            z = x @ W + (b if b is not None else 0.0)
            zt = xt @ W
            if l < len(layers) - 1:
                a, da = activation(model.activation, z)
                d2 = activation2(model.activation, z)
                res = (l > 0 and model.use_resnet_dt and W.shape[0] == W.shape[1])
                x_new, xt_new = a, da * zt
                if res:
                    x_new, xt_new = x_new + x, xt_new + xt
                das.append(da)
                dds.append(d2 * zt)
                ress.append(res)
                x, xt = x_new, xt_new
            else:
                das.append(np.ones_like(z))
                dds.append(np.zeros_like(z))
                ress.append(False)
        kappa = atom_coeff[idx][:, None] * np.ones((len(idx), 1))
        nu = np.ones((len(idx), 1))
        for l in range(len(layers) - 1, -1, -1):
            W = np.asarray(layers[l][0], dtype=float)
            lam = kappa * das[l] + nu * dds[l]
            mu = nu * das[l]
            grads[l] = (xs[l].T @ lam + ts[l].T @ mu, lam.sum(axis=0))
            k_in, n_in = lam @ W.T, mu @ W.T
            if ress[l]:
                k_in, n_in = k_in + kappa, n_in + nu
            kappa, nu = k_in, n_in
        out[el] = grads
    return out


def eam_loss_functional(eam_model, frames, frame_coeff, u, Y):
    """L = sum_f c_f E_f + sum_f (sum_i u_i . F_i + Y_f : W_f) of an oracle `EamModel` — the part of an
    energy + forces + stress loss (reference nn/losses.py:204-437) that depends on the model, with
    c = dL/dE, u = dL/dF, Y = dL/dW held fixed. `frames`: (symbols, positions, cell, pbc)."""
    from .eam import evaluate
    total = 0.0
    for f, (symbols, positions, cell, pbc) in enumerate(frames):
        r = evaluate(eam_model, symbols, positions, cell, pbc)
        total += frame_coeff[f] * r["energy"] + float(np.sum(u[f] * r["forces"])) + float(np.sum(Y[f] * r["virial"]))
    return total


def central_difference_6(fn, x, h):
    """d fn / dx at x with the 7-point stencil (error O(h^6))."""
    c = (-1.0 / 60.0, 3.0 / 20.0, -3.0 / 4.0, 0.0, 3.0 / 4.0, -3.0 / 20.0, 1.0 / 60.0)
    return sum(ck * fn(x + (k - 3) * h) for k, ck in enumerate(c) if ck != 0.0) / h
