"""
Training support for the per-element MLP (first part): batch loss of the total
energies, its gradient with respect to the weights on the GPU (`ta_energy_gradient`), Adam, and the
data-parallel gradient all-reduce.

Mirrors, for the energy term, reference nn/losses.py:204-285 (`get_energy_loss`: per-atom energies,
RMSE with the dtype's eps under the root, or log-cosh), nn/opt.py:89-166 (Adam with optional
exponential learning-rate decay) and train/distribute_utils.py:56-81 (mean of the replicas'
gradients; here `torch.distributed` all-reduce: RCCL between GPUs, gloo in the CPU tests).
Force and stress terms of the loss need second derivatives of the descriptors and are not built.

Descriptors do not depend on the weights: each rank keeps its shard of frames resident, computes the
descriptors once, and every step re-runs only the MLP (forward for the loss, backward for dL/dtheta).
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np

from .parallel import shard_range, world_from_env


def flatten_weights(nn) -> np.ndarray:
    """Flat parameter vector in the C ABI's layout: per element (sorted), per layer W[in][out]
    row-major then b[out] (zeros where the layer has no bias)."""
    out = []
    for el in nn.elements:
        for w, b in nn.weights[el]:
            w = np.asarray(w, dtype=np.float64)
            out.append(w.ravel())
            out.append(np.zeros(w.shape[1]) if b is None else np.asarray(b, dtype=np.float64).ravel())
    return np.concatenate(out)


def unflatten_weights(nn, flat: np.ndarray) -> Dict[str, List]:
    """Inverse of `flatten_weights`; layers without a bias keep `None`."""
    flat = np.asarray(flat, dtype=np.float64)
    out, k = {}, 0
    for el in nn.elements:
        layers = []
        for w, b in nn.weights[el]:
            shape = np.shape(w)
            n = shape[0] * shape[1]
            w2 = flat[k:k + n].reshape(shape).copy()
            k += n
            b2 = None if b is None else flat[k:k + shape[1]].copy()
            k += shape[1]
            layers.append((w2, b2))
        out[el] = layers
    return out


def trainable_mask(nn) -> np.ndarray:
    """1 for real parameters, 0 for the bias slots of layers that have no bias."""
    out = []
    for el in nn.elements:
        for w, b in nn.weights[el]:
            shape = np.shape(w)
            out.append(np.ones(shape[0] * shape[1]))
            out.append(np.zeros(shape[1]) if b is None else np.ones(shape[1]))
    return np.concatenate(out)


def energy_loss(predictions, labels, n_atoms, method="rmse", per_atom_loss=True, weight=1.0):
    """(loss, mae, dloss/dE_f) of nn/losses.py:204-285 for `rmse` and `logcosh`."""
    y = np.asarray(predictions, dtype=np.float64)
    x = np.asarray(labels, dtype=np.float64)
    n = np.asarray(n_atoms, dtype=np.float64) if per_atom_loss else np.ones_like(y)
    xs, ys = x / n, y / n
    diff = ys - xs
    mae = float(np.mean(np.abs(diff)))
    B = max(len(y), 1)
    if method == "rmse":
        mse = float(np.mean(diff * diff)) + np.finfo(np.float64).eps  # losses.py:88-90
        loss = np.sqrt(mse)
        dl = diff / (B * loss * n)
    elif method == "logcosh":
        d = xs - ys                                                  # losses.py:108 (labels - predictions)
        loss = float(np.mean(d + np.logaddexp(0.0, -2.0 * d) - np.log(2.0)))
        dl = -np.tanh(d) / (B * n)
    else:
        raise ValueError(f"loss method '{method}' is not implemented for training")
    return float(weight * loss), mae, weight * dl


class Adam:
    """tf.train.AdamOptimizer as configured by nn/opt.py:89-166: bias-corrected step, optional
    exponential decay `lr * rate ** (step / steps)` (staircase optional)."""

    def __init__(self, n, learning_rate=0.01, beta1=0.9, beta2=0.999, epsilon=1e-8,
                 decay_rate=None, decay_steps=None, staircase=False):
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta1, beta2, epsilon
        self.decay_rate, self.decay_steps, self.staircase = decay_rate, decay_steps, staircase
        self.m = np.zeros(n)
        self.v = np.zeros(n)
        self.t = 0

    def learning_rate(self):
        if not self.decay_rate or not self.decay_steps:
            return self.lr
        p = self.t / self.decay_steps
        if self.staircase:
            p = np.floor(p)
        return self.lr * self.decay_rate ** p

    def step(self, theta, grad):
        lr = self.learning_rate()
        self.t += 1
        self.m = self.b1 * self.m + (1.0 - self.b1) * grad
        self.v = self.b2 * self.v + (1.0 - self.b2) * grad * grad
        lr_t = lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        return theta - lr_t * self.m / (np.sqrt(self.v) + self.eps)


def allreduce_mean(grad: np.ndarray, device=None) -> np.ndarray:
    """Mean of `grad` over the default process group (identity without one): the replicas'
    gradients are averaged as tf.distribute's mirrored strategy does."""
    try:
        import torch
        import torch.distributed as dist
    except Exception:
        return grad
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return grad
    t = torch.from_numpy(np.ascontiguousarray(grad))
    if device is not None and dist.get_backend() == "nccl":
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return (t / dist.get_world_size()).cpu().numpy()


class EnergyTrainer:
    """Fits the MLP weights of `nn` (SF or GRAP descriptors) to total energies.

    `frames` / `energies` are the whole data set; every rank keeps its `shard_range` block
    resident on its GPU. One `step()` = loss + gradient on the shard, mean over ranks, Adam.
    """

    def __init__(self, nn, frames: Sequence, energies: Sequence[float], device=None, method="rmse",
                 per_atom_loss=True, loss_weight=1.0, learning_rate=0.01, **adam_kwargs):
        from .engine import Engine
        rank, local_rank, world = world_from_env()
        lo, hi = shard_range(len(frames), rank, world)
        self.nn = nn
        self.rank, self.world = rank, world
        self.device = local_rank if device is None else device
        self.engine = Engine(nn, device=self.device)
        self.frames = list(frames[lo:hi])
        self.labels = np.asarray(energies, dtype=np.float64)[lo:hi]
        self.n_atoms = np.array([len(a) for a in self.frames], dtype=np.float64)
        self.method, self.per_atom_loss, self.loss_weight = method, per_atom_loss, loss_weight
        self.engine.set_frames(self.frames)
        self.engine.energies(reuse_descriptors=False)  # descriptors, once
        self.theta = flatten_weights(nn)
        self.mask = trainable_mask(nn)
        self.opt = Adam(len(self.theta), learning_rate=learning_rate, **adam_kwargs)
        self.history: List[float] = []

    def loss_and_gradient(self):
        pred = self.engine.energies(reuse_descriptors=True)
        loss, mae, dl = energy_loss(pred, self.labels, self.n_atoms, self.method, self.per_atom_loss,
                                    self.loss_weight)
        grad = self.engine.energy_gradient(dl) * self.mask
        return loss, mae, grad

    def step(self):
        loss, mae, grad = self.loss_and_gradient()
        torch_dev = None
        try:
            import torch
            torch_dev = torch.device("cuda", self.device) if torch.cuda.is_available() else None
        except Exception:
            pass
        grad = allreduce_mean(grad, torch_dev)
        self.theta = self.opt.step(self.theta, grad)
        self.engine.update_weights(self.theta)
        self.history.append(loss)
        return loss, mae

    def fit(self, steps: int):
        for _ in range(steps):
            self.step()
        self.nn.weights = unflatten_weights(self.nn, self.theta)
        return self.history

    def close(self):
        self.engine.close()
