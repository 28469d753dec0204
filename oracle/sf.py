"""
ORACLE (test infrastructure only) — NumPy float64 restatement of the
symmetry-function + per-atom MLP hot path of the reference, with ANALYTIC
forces and virial (the reference obtains them from `tf.gradients`).

Each function cites the reference lines it follows. Written for clarity on
packed pair / triple arrays; `oracle/dense.py` restates the same maths on the
reference's dense padded layout as an independent cross-check.

  term / parameter ordering .... tensoralloy/utils.py:237-290,
                                 tensoralloy/nn/atomic/sf.py:47-51
  pair geometry (eps in norm) .. tensoralloy/transformer/universal.py:448-474
  cutoffs ...................... tensoralloy/nn/cutoff.py:20-85
  G2 ........................... tensoralloy/nn/atomic/sf.py:79-119
  G4 ........................... tensoralloy/nn/atomic/sf.py:121-182
  feature concat ............... tensoralloy/nn/atomic/sf.py:184-215
  min-max scaling .............. tensoralloy/nn/atomic/atomic.py:157-195
  MLP (1x1 conv) ............... tensoralloy/nn/convolutional.py:257-290
  activations .................. tensoralloy/nn/utils.py:39-74
  energy sum ................... tensoralloy/nn/atomic/atomic.py:270-302
  forces ....................... tensoralloy/nn/basic.py:277-290
  virial / stress / pressure ... tensoralloy/nn/basic.py:293-408
"""
import itertools

import numpy as np

from .neighbors import neighbor_list, _complete_cell

EPS64 = 1e-14  # reference precision.py:113 (`Precision.high` eps)
GPA = 1.0 / 160.21766208  # ase.units.GPa, eV/A^3


# --------------------------------------------------------------------------- #
# ordering helpers
# --------------------------------------------------------------------------- #

def radial_term_index(elements, center, other):
    """Index of term `center+other` in [AA, AB (B != A sorted)] (utils.py:265-273)."""
    order = [center] + [e for e in elements if e != center]
    return order.index(other)


def angular_term_index(elements, bj, bk):
    """Index of sorted {bj, bk} among pairs (j <= k) (utils.py:274-282)."""
    n = len(elements)
    a, b = sorted([elements.index(bj), elements.index(bk)])
    pairs = [(j, k) for j in range(n) for k in range(j, n)]
    return pairs.index((a, b))


def radial_params(eta, omega):
    """ParameterGrid({'eta','omega'}): eta outer, omega fastest (sf.py:47-48)."""
    return list(itertools.product(np.ravel(eta), np.ravel(omega)))


def angular_params(beta, gamma, zeta):
    """ParameterGrid({'beta','gamma','zeta'}): zeta fastest (sf.py:49-51)."""
    return list(itertools.product(np.ravel(beta), np.ravel(gamma), np.ravel(zeta)))


# --------------------------------------------------------------------------- #
# scalar functions and derivatives
# --------------------------------------------------------------------------- #

def cutoff(r, rc, kind="cosine"):
    """fc(r) and dfc/dr. cosine: cutoff.py:43-48; polynomial(gamma=5): :78-85."""
    x = np.minimum(r / rc, 1.0)
    inside = (r < rc)
    if kind == "cosine":
        f = 0.5 * (np.cos(np.pi * x) + 1.0)
        df = np.where(inside, -0.5 * np.pi / rc * np.sin(np.pi * x), 0.0)
    elif kind == "polynomial":
        g = 5.0
        f = 1.0 + g * x ** (g + 1.0) - (g + 1.0) * x ** g
        df = np.where(inside, g * (g + 1.0) * (x ** g - x ** (g - 1.0)) / rc, 0.0)
    else:
        raise ValueError(f"unknown cutoff function {kind}")
    return f, df


def activation(name, x):
    """Value and derivative of the reference activations (nn/utils.py:39-74)."""
    name = name.lower()
    if name == "softplus":
        return np.logaddexp(0.0, x), 1.0 / (1.0 + np.exp(-x))
    if name == "relu":
        return np.maximum(x, 0.0), (x > 0).astype(x.dtype)
    if name == "leaky_relu":  # tf.nn.leaky_relu default alpha = 0.2
        return np.where(x > 0, x, 0.2 * x), np.where(x > 0, 1.0, 0.2)
    if name == "tanh":
        t = np.tanh(x)
        return t, 1.0 - t * t
    if name == "sigmoid":
        s = 1.0 / (1.0 + np.exp(-x))
        return s, s * (1.0 - s)
    if name == "softsign":
        d = 1.0 + np.abs(x)
        return x / d, 1.0 / (d * d)
    if name == "elu":
        return np.where(x > 0, x, np.expm1(np.minimum(x, 0.0))), \
            np.where(x > 0, 1.0, np.exp(np.minimum(x, 0.0)))
    if name == "squareplus":  # 0.5 (x + sqrt(x^2 + 4)), nn/utils.py:39-47
        s = np.sqrt(x * x + 4.0)
        return 0.5 * (x + s), 0.5 * (1.0 + x / s)
    raise ValueError(f"The activation function '{name}' cannot be recognized!")


# --------------------------------------------------------------------------- #
# model container
# --------------------------------------------------------------------------- #

class SFModel:
    """
    Plain description of `AtomicNN(SymmetryFunction)` + `UniversalTransformer`.

    weights[el] = list of (W [in,out], b [out] or None); last entry is the
    linear output layer (bias present iff use_atomic_static_energy,
    atomic.py:236-259).
    """

    def __init__(self, elements, rcut, acut=None, angular=False,
                 eta=(0.05, 4.0, 20.0, 80.0), omega=(0.0,), beta=(0.005,),
                 gamma=(1.0, -1.0), zeta=(1.0, 4.0), cutoff_function="cosine",
                 hidden_sizes=None, activation="softplus", weights=None,
                 use_resnet_dt=False, minmax=None, symmetric=True, safe_pow=False):
        self.elements = sorted(set(elements))
        # which `safe_pow` of extension/grad_ops.py:16-74: False = plain tf.pow (the reference's
        # default: the gradient factor zeta base^(zeta-1) is Inf at base = 0 for zeta < 1),
        # True = the TENSORALLOY_USE_CUSTOM_POW variant (Inf values and Inf/NaN gradients -> 0)
        self.safe_pow = bool(safe_pow)
        # symmetric=False lists every neighbour pair of a centre in both orders
        # (universal.py:183-203). The reference's SymmetryFunction holds n(n+1)/2 angular
        # terms (sf.py:131-132), so this only exists for one element.
        self.symmetric = bool(symmetric)
        if not self.symmetric and angular and len(self.elements) > 1:
            raise IndexError("non-symmetric angular terms need n^2 slots, sf.py:131-132 has n(n+1)/2")
        self.rcut = float(rcut)
        self.angular = bool(angular)
        self.acut = float(acut) if acut is not None else self.rcut
        self.eta, self.omega = np.ravel(eta).astype(float), np.ravel(omega).astype(float)
        self.beta, self.gamma, self.zeta = (np.ravel(beta).astype(float),
                                            np.ravel(gamma).astype(float),
                                            np.ravel(zeta).astype(float))
        self.cutoff_function = cutoff_function
        self.hidden_sizes = hidden_sizes
        self.activation = activation
        self.weights = weights
        self.use_resnet_dt = bool(use_resnet_dt)
        self.minmax = minmax  # {el: (xlo [D], xhi [D])} or None

    @property
    def n_radial(self):
        return len(self.elements) * len(self.eta) * len(self.omega)

    @property
    def n_angular(self):
        if not self.angular:
            return 0
        n = len(self.elements)
        return n * (n + 1) // 2 * len(self.beta) * len(self.gamma) * len(self.zeta)

    @property
    def ndim(self):
        return self.n_radial + self.n_angular


def random_weights(model_elements, ndim, hidden_sizes, seed=611, output_bias=True,
                   bias_scale=0.0):
    """
    He-normal (sigma = sqrt(2/fan_in), truncated at 2 sigma) kernels and zero
    (or small random, if `bias_scale`) biases from RandomState(seed)
    (reference nn/init_ops.py:20-30, utils.py:390).
    """
    rng = np.random.RandomState(seed)
    weights = {}
    for el in sorted(set(model_elements)):
        sizes = [ndim] + list(hidden_sizes[el]) + [1]
        layers = []
        for l in range(len(sizes) - 1):
            fan_in, fan_out = sizes[l], sizes[l + 1]
            sigma = np.sqrt(2.0 / fan_in)
            w = rng.normal(0.0, sigma, size=(fan_in, fan_out))
            bad = np.abs(w) > 2 * sigma
            while bad.any():
                w[bad] = rng.normal(0.0, sigma, size=int(bad.sum()))
                bad = np.abs(w) > 2 * sigma
            is_out = l == len(sizes) - 2
            if is_out and not output_bias:
                b = None
            else:
                b = bias_scale * rng.normal(size=fan_out) if bias_scale else np.zeros(fan_out)
            layers.append((w, b))
        weights[el] = layers
    return weights


# --------------------------------------------------------------------------- #
# descriptors
# --------------------------------------------------------------------------- #

def pair_geometry(positions, cell, i, j, S, eps=EPS64):
    """D = Rj - Ri + S.h ; r = sqrt(D.D + eps) (universal.py:463-474)."""
    D = positions[j] - positions[i] + S.astype(np.float64) @ cell
    r = np.sqrt(np.sum(D * D, axis=1) + eps)
    return D, r


def apply_mlp(model, symbols, G):
    """Per-element MLP on the descriptors G [N, D] (`model` needs elements, weights, activation,
    use_resnet_dt, minmax): atomic energies [N] and dE/dG [N, D].
    Follows atomic.py:157-195 (min-max), convolutional.py:257-290 (layers, skip), atomic.py:236-259."""
    elements = model.elements
    N, Dn = G.shape
    atomic = np.zeros(N)
    dEdG = np.zeros((N, Dn))
    for el in elements:
        idx = np.array([k for k, s in enumerate(symbols) if s == el], dtype=np.int64)
        if len(idx) == 0:
            continue
        x = G[idx]
        scale = None
        if model.minmax is not None and el in model.minmax:
            xlo, xhi = model.minmax[el]
            den = xhi - xlo
            ok = den != 0.0
            safe = np.where(ok, den, 1.0)
            x = np.where(ok, (xhi - x) / safe, 0.0)  # div_no_nan, atomic.py:195
            scale = np.where(ok, -1.0 / safe, 0.0)
        layers = model.weights[el]
        hs, dacts = [x], []
        hcur = x
        for l, (W, b) in enumerate(layers[:-1]):
            zl = hcur @ W + (b if b is not None else 0.0)
            a, da = activation(model.activation, zl)
            res = (l > 0 and model.use_resnet_dt and W.shape[0] == W.shape[1])
            hnext = a + hcur if res else a  # convolutional.py:272-273
            dacts.append((da, res))
            hcur = hnext
            hs.append(hcur)
        Wo, bo = layers[-1]
        y = hcur @ Wo + (bo if bo is not None else 0.0)
        atomic[idx] = y[:, 0]
        # backward to the inputs
        delta = np.repeat(Wo.T, len(idx), axis=0)  # dy/dh_L
        for l in range(len(layers) - 2, -1, -1):
            W, _ = layers[l]
            da, res = dacts[l]
            back = (delta * da) @ W.T
            delta = back + delta if res else back
        if scale is not None:
            delta = delta * scale
        dEdG[idx] = delta
    return atomic, dEdG


def _triples_of(n):
    a, b = np.triu_indices(n, k=1)
    return a, b


def evaluate(model: SFModel, symbols, positions, cell, pbc, want_forces=True,
             eps=EPS64):
    """
    Full oracle evaluation of one structure. Returns a dict with

      descriptors [N, D] (raw G, before min-max), energy, atomic [N],
      forces [N,3], virial [3,3], stress_voigt [6], total_pressure (GPa),
      dEdG [N, D], npairs, ntriples.

    All in the caller's atom order (the GSL/VAP reordering of the reference
    is a pure permutation handled at the calculator boundary).
    """
    elements = model.elements
    symbols = list(symbols)
    R = np.asarray(positions, dtype=np.float64).reshape(-1, 3)
    N = len(R)
    pbc = np.asarray(pbc, dtype=bool).reshape(3)
    h = _complete_cell(cell, pbc)
    volume = abs(np.linalg.det(h))
    kind = model.cutoff_function

    rad = radial_params(model.eta, model.omega)
    ang = angular_params(model.beta, model.gamma, model.zeta) if model.angular else []
    nr, na = len(rad), len(ang)
    nel = len(elements)
    Dn = model.ndim
    G = np.zeros((N, Dn))

    # ---- radial list -----------------------------------------------------
    pi, pj, pS = neighbor_list(R, h, pbc, model.rcut)
    Dij, rij = pair_geometry(R, h, pi, pj, pS, eps)
    fc, dfc = cutoff(rij, model.rcut, kind)
    rc2 = model.rcut ** 2
    pterm = np.array([radial_term_index(elements, symbols[a], symbols[b])
                      for a, b in zip(pi, pj)], dtype=np.int64)
    for c, (eta, omega) in enumerate(rad):
        v = np.exp(-eta * (rij - omega) ** 2 / rc2) * fc
        np.add.at(G, (pi, pterm * nr + c), v)

    # ---- angular list ----------------------------------------------------
    tri = None
    if model.angular:
        if round(model.acut - model.rcut, 2) == 0.0:  # universal.py:823-829
            ai, aj, aS, aD, ar = pi, pj, pS, Dij, rij
        else:
            ai, aj, aS = neighbor_list(R, h, pbc, model.acut)
            aD, ar = pair_geometry(R, h, ai, aj, aS, eps)
        starts = np.searchsorted(ai, np.arange(N + 1))
        ta, tb = [], []
        for c in range(N):
            n = starts[c + 1] - starts[c]
            if n < 2:
                continue
            a, b = _triples_of(n)
            if not model.symmetric:  # (j, k) and (k, j), j != k
                a, b = np.concatenate([a, b]), np.concatenate([b, a])
            ta.append(a + starts[c])
            tb.append(b + starts[c])
        if ta:
            ta = np.concatenate(ta)
            tb = np.concatenate(tb)
        else:
            ta = tb = np.zeros(0, dtype=np.int64)
        ti = ai[ta]
        Da, Db = aD[ta], aD[tb]
        ra, rb = ar[ta], ar[tb]
        Djk = Db - Da  # = Rk - Rj + (S_ik - S_ij).h  (universal.py:213, :628-645)
        rd = np.sqrt(np.sum(Djk * Djk, axis=1) + eps)
        ac2 = model.acut ** 2
        fa, dfa = cutoff(ra, model.acut, kind)
        fb, dfb = cutoff(rb, model.acut, kind)
        fd, dfd = cutoff(rd, model.acut, kind)
        lower = 2.0 * ra * rb
        cos = np.where(lower != 0.0, (ra * ra + rb * rb - rd * rd) / np.where(lower != 0, lower, 1.0), 0.0)
        z = (ra * ra + rb * rb + rd * rd) / ac2
        tterm = np.array([angular_term_index(elements, symbols[aj[x]], symbols[aj[y]])
                          for x, y in zip(ta, tb)], dtype=np.int64)
        fprod = fa * fb * fd
        for c, (beta, gamma, zeta) in enumerate(ang):
            with np.errstate(divide="ignore", invalid="ignore"):
                pw = (1.0 + gamma * cos) ** zeta
            if model.safe_pow:
                pw = np.where(np.isinf(pw), 0.0, pw)       # grad_ops.py:25-26
            v = 2.0 ** (1.0 - zeta) * pw * np.exp(-beta * z) * fprod
            np.add.at(G, (ti, model.n_radial + tterm * na + c), v)
        tri = dict(ta=ta, tb=tb, ti=ti, Da=Da, Db=Db, Djk=Djk, ra=ra, rb=rb, rd=rd,
                   fa=fa, fb=fb, fd=fd, dfa=dfa, dfb=dfb, dfd=dfd, cos=cos, z=z,
                   tterm=tterm, ai=ai, aj=aj, aS=aS, aD=aD)

    out = {"descriptors": G.copy(), "npairs": len(pi),
           "ntriples": 0 if tri is None else len(tri["ta"]), "volume": volume}
    if model.weights is None:
        return out

    # ---- min-max + MLP forward/backward -----------------------------------
    atomic, dEdG = apply_mlp(model, symbols, G)

    energy = float(np.sum(atomic))
    out.update(energy=energy, atomic=atomic, dEdG=dEdG)
    if not want_forces:
        return out

    # ---- analytic forces and virial ----------------------------
    F = np.zeros((N, 3))
    W = np.zeros((3, 3))
    # G2
    s = np.zeros(len(pi))
    for c, (eta, omega) in enumerate(rad):
        e = np.exp(-eta * (rij - omega) ** 2 / rc2)
        dg = e * (dfc - 2.0 * eta * (rij - omega) * fc / rc2)
        s += dEdG[pi, pterm * nr + c] * dg
    f = (s / rij)[:, None] * Dij  # dE/dD_ij
    np.add.at(F, pi, f)
    np.add.at(F, pj, -f)
    W += f.T @ Dij
    # G4
    if tri is not None and len(tri["ta"]):
        t = tri
        a, b, d, c_ = t["ra"], t["rb"], t["rd"], t["cos"]
        dva = np.zeros(len(a))
        dvb = np.zeros(len(a))
        dvd = np.zeros(len(a))
        ac2 = model.acut ** 2
        dca = 1.0 / b - c_ / a
        dcb = 1.0 / a - c_ / b
        dcd = -d / (a * b)
        for ch, (beta, gamma, zeta) in enumerate(ang):
            w = dEdG[t["ti"], model.n_radial + t["tterm"] * na + ch]
            base = 1.0 + gamma * c_
            with np.errstate(divide="ignore", invalid="ignore"):
                P = base ** zeta
                dP = zeta * gamma * base ** (zeta - 1.0)   # tf.pow's gradient, Inf at base = 0, zeta < 1
            if model.safe_pow:                              # grad_ops.py:25-26, :46-49
                P = np.where(np.isinf(P), 0.0, P)
                dP = np.where(np.isfinite(dP), dP, 0.0)
            e = 2.0 ** (1.0 - zeta) * np.exp(-beta * t["z"])
            fa_, fb_, fd_ = t["fa"], t["fb"], t["fd"]
            dva += w * e * (dP * dca * fa_ - 2.0 * beta * a / ac2 * P * fa_ + P * t["dfa"]) * fb_ * fd_
            dvb += w * e * (dP * dcb * fb_ - 2.0 * beta * b / ac2 * P * fb_ + P * t["dfb"]) * fa_ * fd_
            dvd += w * e * (dP * dcd * fd_ - 2.0 * beta * d / ac2 * P * fd_ + P * t["dfd"]) * fa_ * fb_
        t_a = (dva / a)[:, None] * t["Da"]
        t_b = (dvb / b)[:, None] * t["Db"]
        t_d = (dvd / d)[:, None] * t["Djk"]
        ji = t["aj"][t["ta"]]
        ki = t["aj"][t["tb"]]
        np.add.at(F, t["ti"], t_a + t_b)
        np.add.at(F, ji, -t_a + t_d)
        np.add.at(F, ki, -t_b - t_d)
        W += t_a.T @ t["Da"] + t_b.T @ t["Db"] + t_d.T @ t["Djk"]

    stress = W / volume
    voigt = np.array([stress[0, 0], stress[1, 1], stress[2, 2],
                      stress[1, 2], stress[0, 2], stress[0, 1]])
    out.update(forces=F, virial=W, stress_voigt=voigt,
               total_pressure=float(np.trace(stress) / (-3.0 * GPA)))
    return out
