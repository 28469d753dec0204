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
