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


def _networks(nn):
    """The model's networks in the order of the C ABI's parameter vector, as (container, key):
    per element for an `AtomicNN`; per nn-function slot (rho[element], embed[element], phi[pair],
    dipole[pair], quadrupole[pair]) for an `EamAlloyNN` / `AdpNN`."""
    if hasattr(nn, "nn_functions"):
        return [(nn.weights[sec], fn) for sec, fn in (s for s in nn.nn_functions() if s is not None)]
    return [(nn.weights, el) for el in nn.elements]


def flatten_weights(nn) -> np.ndarray:
    """Flat parameter vector in the C ABI's layout: per network (see `_networks`), per layer
    W[in][out] row-major then b[out] (zeros where the layer has no bias)."""
    out = []
    for box, key in _networks(nn):
        for w, b in box[key]:
            w = np.asarray(w, dtype=np.float64)
            out.append(w.ravel())
            out.append(np.zeros(w.shape[1]) if b is None else np.asarray(b, dtype=np.float64).ravel())
    return np.concatenate(out)


def unflatten_weights(nn, flat: np.ndarray) -> Dict[str, List]:
    """Inverse of `flatten_weights`; layers without a bias keep `None`."""
    flat = np.asarray(flat, dtype=np.float64)
    out, k = {}, 0
    eam = hasattr(nn, "nn_functions")
    names = [s for s in nn.nn_functions() if s is not None] if eam else [(None, el) for el in nn.elements]
    for (box, key), (sec, _) in zip(_networks(nn), names):
        layers = []
        for w, b in box[key]:
            shape = np.shape(w)
            n = shape[0] * shape[1]
            w2 = flat[k:k + n].reshape(shape).copy()
            k += n
            b2 = None if b is None else flat[k:k + shape[1]].copy()
            k += shape[1]
            layers.append((w2, b2))
        if eam:
            out.setdefault(sec, {})[key] = layers
        else:
            out[key] = layers
    return out


def trainable_mask(nn) -> np.ndarray:
    """1 for real parameters, 0 for the bias slots of layers that have no bias and, when the model
    was built with `fixed_atomic_static_energy=True`, for every element's output bias: the
    reference creates that bias with `trainable=False` (atomic.py:249 -> `convolution1x1(...,
    fixed_output_bias=True)`, convolutional.py:277-290), so the atomic static energies stay put."""
    frozen_output_bias = bool(getattr(nn, "_fixed_atomic_static_energy", False))
    out = []
    for box, key in _networks(nn):
        layers = box[key]
        for l, (w, b) in enumerate(layers):
            shape = np.shape(w)
            out.append(np.ones(shape[0] * shape[1]))
            frozen = b is None or (frozen_output_bias and l == len(layers) - 1)
            out.append(np.zeros(shape[1]) if frozen else np.ones(shape[1]))
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


def _lib_want_all():
    from . import _lib
    return _lib.TA_WANT_ENERGY | _lib.TA_WANT_FORCES | _lib.TA_WANT_VIRIAL | _lib.TA_WANT_ATOMIC


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
    t = torch.from_numpy(np.array(grad, dtype=np.float64, copy=True))  # the all-reduce is in place: not on the caller's array
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


# ---- forces and stress terms -----------------------------------------------------------------------

def forces_loss(predictions, labels, method="rmse", weight=1.0):
    """(loss, mae, [dloss/dF per frame]) of nn/losses.py:285-332 over all real atoms' components:
    sqrt(mean(dF^2) + eps) or mean log-cosh. `predictions` / `labels`: lists of [N_f, 3] arrays."""
    diff = np.concatenate([np.asarray(l, dtype=np.float64) - np.asarray(p, dtype=np.float64)
                           for p, l in zip(predictions, labels)])          # labels - predictions
    n = diff.size
    mae = float(np.mean(np.abs(diff)))
    if method == "rmse":
        loss = np.sqrt(float(np.mean(diff * diff)) + np.finfo(np.float64).eps)
        d_all = -diff / (n * loss)
    elif method == "logcosh":
        loss = float(np.mean(diff + np.logaddexp(0.0, -2.0 * diff) - np.log(2.0)))
        d_all = -np.tanh(diff) / n
    else:
        raise ValueError(f"loss method '{method}' is not implemented for training")
    out, k = [], 0
    for p in predictions:
        m = len(p)
        out.append(weight * d_all[k:k + m])
        k += m
    return float(weight * loss), mae, out


def stress_loss(predictions, labels, method="rmse", weight=1.0):
    """(loss, mae, dloss/dstress [B, 6]) of nn/losses.py:384-437 on Voigt stresses in eV/A^3."""
    y = np.asarray(predictions, dtype=np.float64).reshape(-1, 6)
    x = np.asarray(labels, dtype=np.float64).reshape(-1, 6)
    diff = y - x
    mae = float(np.mean(np.abs(diff)))
    if method == "rmse":
        loss = np.sqrt(float(np.mean(diff * diff)) + np.finfo(np.float64).eps)
        d = diff / (diff.size * loss)
    elif method == "logcosh":
        e = x - y
        loss = float(np.mean(e + np.logaddexp(0.0, -2.0 * e) - np.log(2.0)))
        d = -np.tanh(e) / diff.size
    else:
        raise ValueError(f"loss method '{method}' is not implemented for training")
    return float(weight * loss), mae, weight * d


GPA = 1.0 / 160.21766208  # ase.units.GPa in eV/A^3 (the reference's pressure unit, nn/basic.py:403-405)


def pressure_loss(predictions, labels, method="rmse", weight=1.0):
    """(loss, mae, dloss/dP [B]) of nn/losses.py:459-505 (`get_pressure_loss`) on total pressures
    in GPa: sqrt(mean(dP^2) + eps) or mean log-cosh of (labels - predictions); `rrmse` is refused as
    the reference asserts."""
    y = np.asarray(predictions, dtype=np.float64).ravel()
    x = np.asarray(labels, dtype=np.float64).ravel()
    diff = y - x
    mae = float(np.mean(np.abs(diff)))
    if method == "rmse":
        loss = np.sqrt(float(np.mean(diff * diff)) + np.finfo(np.float64).eps)
        d = diff / (diff.size * loss)
    elif method == "logcosh":
        e = x - y
        loss = float(np.mean(e + np.logaddexp(0.0, -2.0 * e) - np.log(2.0)))
        d = -np.tanh(e) / diff.size
    else:
        raise ValueError(f"loss method '{method}' is not available for the pressure loss")
    return float(weight * loss), mae, weight * d


def relative_forces_loss(predictions, labels, weight=1.0):
    """(loss, None, [dloss/dF per frame]) of `_get_relative_rmse_loss` (nn/losses.py:53-68), the
    `rrmse` method of the force loss: mean over the atoms of |F_label - F_pred| / |F_label|."""
    P = np.concatenate([np.asarray(p, dtype=np.float64) for p in predictions])
    L = np.concatenate([np.asarray(l, dtype=np.float64) for l in labels])
    diff = L - P
    upper = np.linalg.norm(diff, axis=1)
    lower = np.linalg.norm(L, axis=1)
    n = len(P)
    loss = float(np.mean(upper / lower))
    with np.errstate(divide="ignore", invalid="ignore"):
        g = np.where(upper[:, None] > 0.0, -diff / (upper * lower * n)[:, None], 0.0)
    out, k = [], 0
    for p in predictions:
        out.append(weight * g[k:k + len(p)])
        k += len(p)
    return float(weight * loss), None, out


def l2_regularization_loss(nn, theta, l2_weight, weight=0.01, step=0, decayed=True, decay_rate=0.99,
                           decay_steps=1000):
    """(loss, dloss/dtheta) of nn/losses.py:507-551 + the regularisers `convolution1x1` attaches
    (convolutional.py:207-290): `l2_regularizer(l2_weight)` = l2_weight sum(w^2) / 2 on the kernel AND
    the bias of every hidden layer and on the kernel (not the bias) of the output layer; the sum is
    scaled by `weight`, exponentially decayed as weight * decay_rate ** (step / decay_steps) when
    `decayed` (tf.train.exponential_decay, no staircase). `theta` in the C ABI's flat layout."""
    theta = np.asarray(theta, dtype=np.float64)
    sel = np.zeros_like(theta)
    k = 0
    for box, key in _networks(nn):
        layers = box[key]
        for l, (w, b) in enumerate(layers):
            shape = np.shape(w)
            n = shape[0] * shape[1]
            sel[k:k + n] = 1.0
            k += n
            if b is not None and l < len(layers) - 1:
                sel[k:k + shape[1]] = 1.0
            k += shape[1]
    lam = float(weight) * (decay_rate ** (step / decay_steps) if decayed else 1.0)
    l2 = 0.5 * l2_weight * float(np.sum(sel * theta * theta))
    return lam * l2, lam * l2_weight * sel * theta


def loss_weight_at(weight, step=0, max_train_steps=None, logscale=True):
    """`get_static_or_dyn_weight_tensor` (nn/losses.py:171-201): a float is the weight; a pair
    (w0, w1) moves from w0 to w1 over `max_train_steps` steps, linearly in log10 when `logscale`
    (the reference's default, `_LossOptions.logscaled_dynamic_weight`), else linearly."""
    if isinstance(weight, (int, float)):
        return float(weight)
    w0, w1 = float(weight[0]), float(weight[1])
    if max_train_steps is None:
        raise ValueError("a dynamic loss weight needs max_train_steps")
    if logscale:
        l0, l1 = np.log10(w0), np.log10(w1)
        return float(10.0 ** (l0 + (l1 - l0) / max_train_steps * step))
    return w0 + (w1 - w0) / max_train_steps * step


class Trainer:
    """Energy + forces + stress (+ total pressure, + L2) loss (nn/basic.py `get_total_loss`: the sum
    of the weighted terms). Weights may be pairs (w0, w1): dynamic weights of the reference.

    The weight gradient of the force and stress terms needs second derivatives of the energy
    (the reference: `tf.gradients` through nn/losses.py:285-437). With u = dL/dF per atom and the
    symmetric Y built from dL/dstress,
        sum u.F + sum Y.W = D_delta E,   delta R = R.Y - u,  delta h = h.Y,
    the directional derivative of the energy; for the per-atom MLP models the descriptors do not
    depend on the weights, so d/dtheta of it is ONE analytic second-order pass through the MLP
    (`ta_loss_gradient`: descriptor Jacobian once per resident batch, then a pair sweep + the MLP
    pass per step, energy term included). An EAM model whose functions are all analytic trains their
    CONSTANTS instead (potentials/potentials.py:129-163): `ta_constant_gradient` differentiates the
    same functional in dual arithmetic, one seeded constant per grid row. The nn functions of a plain EAM
    / ADP model (round 3) enter D_delta E through their values and input derivatives at known points: one
    second-order pass per network (`ta_loss_gradient` again), ADP's dipole and quadrupole networks
    included. `analytic=False` forces the central difference of g = dE/dtheta on two displaced copies of
    every frame everywhere (step `fd_step` Angstrom, error O(step^2)).
    """

    def __init__(self, nn, frames, energies, forces=None, stresses=None, device=None,
                 energy_weight=1.0, forces_weight=1.0, stress_weight=1.0, method="rmse",
                 per_atom_loss=True, learning_rate=0.01, fd_step=1e-3, analytic=None, fixed=None,
                 pressures=None, pressure_weight=1.0, forces_method=None, l2_weight=0.0, l2_loss_weight=0.01,
                 l2_decayed=True, l2_decay_rate=0.99, l2_decay_steps=1000, max_train_steps=None,
                 logscaled_dynamic_weight=True, train_constants=None, **adam_kwargs):
        from .engine import Engine
        rank, local_rank, world = world_from_env()
        lo, hi = shard_range(len(frames), rank, world)
        self.nn = nn
        self.rank, self.world = rank, world
        self.device = local_rank if device is None else device
        self.engine = Engine(nn, device=self.device)
        # an EAM model without nn functions trains the constants of its analytic functions
        # (reference potentials/potentials.py:129-163), `fixed` = {section: [names]} stay put
        self.constants_mode = hasattr(nn, "nn_functions") and not any(s is not None for s in nn.nn_functions())
        # the analytic second-order pass exists for the per-atom MLP models (and, in dual arithmetic,
        # for the constants)
        # ... and, since round 3, for the nn functions of EAM / ADP models (`ta_loss_gradient`, one
        # second-order pass per network); the library refuses the sutton90 / Be/1 / grimes families, for
        # which loss_and_gradient falls back to the central difference
        self._eam_nets = hasattr(nn, "nn_functions") and not self.constants_mode
        self.analytic = True if analytic is None else bool(analytic)
        if self.constants_mode and not self.analytic:
            raise ValueError("the constants have no finite-difference path")
        self._resident = False
        self._generation = -1
        self.frames = list(frames[lo:hi])
        self.e_ref = np.asarray(energies, dtype=np.float64)[lo:hi]
        self.f_ref = None if forces is None else [np.asarray(f, dtype=np.float64) for f in forces[lo:hi]]
        self.s_ref = None if stresses is None else np.asarray(stresses, dtype=np.float64).reshape(-1, 6)[lo:hi]
        self.n_atoms = np.array([len(a) for a in self.frames], dtype=np.float64)
        self.p_ref = None if pressures is None else np.asarray(pressures, dtype=np.float64).ravel()[lo:hi]
        self.weights = (energy_weight, forces_weight, stress_weight)
        self.pressure_weight = pressure_weight
        self.forces_method = forces_method or method   # "rrmse": relative RMSE (forces only)
        self.l2 = dict(l2_weight=l2_weight, weight=l2_loss_weight, decayed=l2_decayed, decay_rate=l2_decay_rate,
                       decay_steps=l2_decay_steps)
        self.max_train_steps, self.logscale = max_train_steps, logscaled_dynamic_weight
        self.method, self.per_atom_loss, self.fd_step = method, per_atom_loss, fd_step
        # a model that mixes networks with analytic functions trains both, as the reference does (every
        # variable of potentials/potentials.py:129-200 and of the nn functions): theta = [weights | constants]
        specs = nn.nn_functions() if hasattr(nn, "nn_functions") else []
        self.mixed = (not self.constants_mode and any(s is not None for s in specs) and
                      any(s is None for s in specs)) if train_constants is None else bool(train_constants)
        self._n_weights = None
        if self.constants_mode:
            self.theta = nn.constants()
            self.mask = nn.constant_mask(fixed)
        else:
            self.theta = flatten_weights(nn)
            self.mask = trainable_mask(nn)
            if self.mixed:
                self._n_weights = len(self.theta)
                self.theta = np.concatenate([self.theta, nn.constants()])
                self.mask = np.concatenate([self.mask, nn.constant_mask(fixed)])
        self.opt = Adam(len(self.theta), learning_rate=learning_rate, **adam_kwargs)
        self.history: List[dict] = []

    def loss_and_gradient(self):
        from .atoms import Atoms
        eng = self.engine
        # the shortcut holds only while the engine still has THIS trainer's frames resident: any
        # set_frames / update_positions on the (public) engine in between bumps its generation
        if self.analytic and self._resident and eng.batch_generation == self._generation:
            # same frames as the last step: the batch, its neighbour list, descriptors and their
            # Jacobian are resident; only the MLP changed
            eng.compute(_lib_want_all())
            res = eng._per_frame(eng.fetch(_lib_want_all()))
        else:
            res = eng.evaluate(self.frames)
            self._resident = self.analytic
            self._generation = eng.batch_generation
        pred_e = np.array([r["energy"] for r in res])
        step = self.opt.t
        we, wf, ws = (loss_weight_at(w, step, self.max_train_steps, self.logscale) for w in self.weights)
        wp = loss_weight_at(self.pressure_weight, step, self.max_train_steps, self.logscale)
        terms = {}
        loss_e, mae_e, c = energy_loss(pred_e, self.e_ref, self.n_atoms, self.method, self.per_atom_loss, we)
        terms["energy"] = loss_e
        if not self.analytic:
            grad = eng.energy_gradient(c)      # the resident batch is the undisplaced one
        u = [np.zeros((len(a), 3)) for a in self.frames]
        Y = [np.zeros((3, 3)) for _ in self.frames]
        second = False
        if self.f_ref is not None and wf != 0.0:
            if self.forces_method == "rrmse":
                lf, _, du = relative_forces_loss([r["forces"] for r in res], self.f_ref, wf)
            else:
                lf, _, du = forces_loss([r["forces"] for r in res], self.f_ref, self.forces_method, wf)
            terms["forces"] = lf
            u = du
            second = True
        ds = np.zeros((len(self.frames), 6))
        if self.s_ref is not None and ws != 0.0:
            ls, _, d = stress_loss(np.array([r["stress"] for r in res]), self.s_ref, self.method, ws)
            terms["stress"] = ls
            ds += d
            second = True
        if self.p_ref is not None and wp != 0.0:
            # total pressure P = -tr(stress) / 3 / GPa (nn/basic.py:394-408): dL/dstress_aa = -dL/dP / (3 GPa)
            lp, _, dp = pressure_loss(np.array([r["total_pressure"] for r in res]), self.p_ref, self.method, wp)
            terms["pressure"] = lp
            ds[:, :3] += (-dp / (3.0 * GPA))[:, None]
            second = True
        if ds.any():
            for k, a in enumerate(self.frames):
                V = abs(np.linalg.det(np.asarray(a.get_cell(complete=True))))
                xx, yy, zz, yz, xz, xy = ds[k] / V
                Y[k] = np.array([[xx, xy / 2, xz / 2], [xy / 2, yy, yz / 2], [xz / 2, yz / 2, zz]])
        if self.analytic:
            dR = np.concatenate([a.positions @ Y[k] - u[k] for k, a in enumerate(self.frames)]) \
                if self.frames else np.zeros((0, 3))
            dh = np.array([np.asarray(a.get_cell(complete=True), dtype=np.float64) @ Y[k]
                           for k, a in enumerate(self.frames)])
            gradient = eng.constant_gradient if self.constants_mode else eng.loss_gradient
            try:
                grad = gradient(c, dR if second else None, dh if second else None)
                if self.mixed:   # ... and the constants of the analytic functions beside the networks
                    grad = np.concatenate([grad, eng.constant_gradient(c, dR if second else None,
                                                                       dh if second else None)])
            except RuntimeError as err:
                if not (self._eam_nets and "ta_loss_gradient" in str(err)):
                    raise
                self.analytic = self._resident = False   # a family without the analytic pass
                return self.loss_and_gradient()
        elif second:
            disp, coeff = [], []
            for k, a in enumerate(self.frames):
                h = np.asarray(a.get_cell(complete=True), dtype=np.float64)
                dR = a.positions @ Y[k] - u[k]
                dh = h @ Y[k]
                scale = max(np.abs(dR).max(initial=0.0), np.abs(dh).max(), 1e-300)
                e = self.fd_step / scale
                for sgn in (1.0, -1.0):
                    disp.append(Atoms(numbers=np.asarray(a.numbers).copy(), positions=a.positions + sgn * e * dR,
                                      cell=h + sgn * e * dh, pbc=np.asarray(a.pbc).copy()))
                    coeff.append(sgn / (2.0 * e))
            eng.set_frames(disp)
            grad = grad + eng.energy_gradient(np.array(coeff))
            self._resident = False
        if self.mixed and len(grad) == self._n_weights:   # (central-difference path: the constants' part is analytic)
            Rc = np.concatenate([a.positions @ Y[k] - u[k] for k, a in enumerate(self.frames)])
            hc = np.array([np.asarray(a.get_cell(complete=True), dtype=np.float64) @ Y[k]
                           for k, a in enumerate(self.frames)])
            eng.set_frames(self.frames)
            grad = np.concatenate([grad, eng.constant_gradient(c, Rc if second else None, hc if second else None)])
        if self.l2["l2_weight"] > 0.0 and self.l2["weight"] != 0.0 and not self.constants_mode:
            nw = self._n_weights if self.mixed else len(self.theta)
            l2, g2 = l2_regularization_loss(self.nn, self.theta[:nw], step=step, **self.l2)
            terms["l2"] = l2
            # every replica adds the same regulariser: the mean over ranks (step) leaves it as it is
            grad = grad + np.concatenate([g2, np.zeros(len(self.theta) - nw)])
        total = float(sum(terms.values()))
        return total, terms, grad * self.mask

    def step(self):
        total, terms, grad = self.loss_and_gradient()
        torch_dev = None
        try:
            import torch
            torch_dev = torch.device("cuda", self.device) if torch.cuda.is_available() else None
        except Exception:
            pass
        grad = allreduce_mean(grad, torch_dev)
        self.theta = self.opt.step(self.theta, grad)
        if self.constants_mode:
            self.engine.update_constants(self.theta)
        elif self.mixed:
            self.engine.update_weights(self.theta[:self._n_weights])
            self.engine.update_constants(self.theta[self._n_weights:])
        else:
            self.engine.update_weights(self.theta)
        self.history.append(dict(terms, total=total))
        return total, terms

    def fit(self, steps: int):
        for _ in range(steps):
            self.step()
        if self.constants_mode:
            self.nn.set_constants(self.theta)
        elif self.mixed:
            self.nn.weights = unflatten_weights(self.nn, self.theta[:self._n_weights])
            self.nn.set_constants(self.theta[self._n_weights:])
        else:
            self.nn.weights = unflatten_weights(self.nn, self.theta)
        return self.history

    def close(self):
        self.engine.close()
