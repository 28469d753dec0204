"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's
GenericRadialAtomicPotential descriptor (GRAP) + per-element MLP + analytic forces / virial.

Follows reference tensoralloy/nn/atomic/grap.py:
  radial filters ........ SymmetryFunctionAlgorithm :124-146, MorseAlgorithm :149-169,
                          DensityExpAlgorithm :172-192, PowerExpAlgorithm :195-219
                          (+ nn/eam/potentials/generic.py:15-30, :87-99, :120-176)
  parameter space ....... Algorithm.__init__ :40-79 ('cross' = sklearn ParameterGrid over sorted
                          keys, last key fastest; 'pair' = row i of every list)
  legacy mode ........... apply_legacy_pairwise_descriptor_functions :378-468
  new mode .............. apply_model :592-683, _get_moment_coeff_tensor :494-529,
                          _get_multiplicity_tensor :470-492
  feature order ......... neighbour-species block in k-body-term order [AA, AB (B != A sorted)],
                          then filter, then moment.

Pinning: the reference holds no numeric GRAP fixture. Its own tests pin the two modes against each
other (nn/atomic/tests/test_grap.py:49-105 Be/W pexp moments 0,1,2 to 1e-6; :108-149 Fe morse
moments 0,1 to 1e-8) and the packed multiplicity/moment tensors against the full ones (:152-200);
tests/test_oracle_golden.py repeats those three checks on this file. Absolute values: parity
unpinned; forces are additionally checked by finite differences.
"""
import itertools

import numpy as np

from .neighbors import neighbor_list, _complete_cell
from .sf import EPS64, GPA, activation, apply_mlp, cutoff, pair_geometry, radial_term_index  # noqa: F401

REQUIRED_KEYS = {"sf": ["eta", "omega"], "morse": ["D", "gamma", "r0"],
                 "density": ["A", "beta", "re"], "pexp": ["rl", "pl"]}


def parameter_grid(algorithm, parameters, method):
    """List of dicts, one per filter (grap.py:40-79)."""
    keys = REQUIRED_KEYS[algorithm]
    params = {k: [float(x) for x in parameters[k]] for k in keys}
    if method == "cross":
        names = sorted(params)  # ParameterGrid: sorted keys, last one fastest
        return [dict(zip(names, combo)) for combo in itertools.product(*[params[n] for n in names])]
    sizes = {len(v) for v in params.values()}
    if len(sizes) > 1:
        raise ValueError("Hyperparameters must have the same length for gen:pair")
    return [{k: params[k][i] for k in keys} for i in range(sizes.pop())]


def radial_filter(algorithm, row, r, rc):
    """v(r) and dv/dr of one filter (without the cutoff)."""
    if algorithm == "sf":
        e = np.exp(-row["eta"] * (r - row["omega"]) ** 2 / rc ** 2)
        return e, e * (-2.0 * row["eta"] * (r - row["omega"]) / rc ** 2)
    if algorithm == "morse":
        gd = row["gamma"] * (r - row["r0"])
        e1, e2 = np.exp(-gd), np.exp(-2.0 * gd)
        return row["D"] * (e2 - 2.0 * e1), row["D"] * row["gamma"] * (-2.0 * e2 + 2.0 * e1)
    if algorithm == "density":
        e = row["A"] * np.exp(-row["beta"] * (r / row["re"] - 1.0))
        return e, e * (-row["beta"] / row["re"])
    if algorithm == "pexp":
        x = r / row["rl"]
        xp = x ** row["pl"]
        e = np.exp(-xp)
        return e, e * (-row["pl"] * xp / r)
    raise ValueError(f"GRAP: algorithm '{algorithm}' is not implemented")


# covalent radii (Cordero et al. 2008 = ase.data.covalent_radii) of the elements the tests use
COVALENT_RADII = {"H": 0.31, "Be": 0.96, "C": 0.76, "N": 0.71, "O": 0.66, "Al": 1.21, "Fe": 1.32,
                  "Ni": 1.24, "Cu": 1.32, "Mo": 1.54, "Pd": 1.39, "W": 1.62}


def nn_filters(r, net, rcov=None):
    """The `nn` algorithm (grap.py:220-270, :620-643): ONE shared 1x1 CNN `Filters` maps r to all
    K = num_filters filter values, `convolution1x1(r, hidden_sizes, num_out=K, output_bias=False,
    use_resnet_dt=...)` (convolutional.py:257-290). Returns v [P, K] and dv/dr [P, K]
    (forward-mode). `net` = dict(layers=[(W, b), ..., (W_out, None)], activation, use_resnet_dt).
    `h_abck_modifier` (grap.py:620-631): 0 the input is r; 1 r / rcov; 2 exp(-r / rcov), `rcov` [P] =
    covalent radius of the pair's CENTRE element."""
    from .sf import activation
    h = np.asarray(r, dtype=np.float64)[:, None]
    dh = np.ones_like(h)
    mod = int(net.get("h_abck_modifier", 0) or 0)
    if mod == 1:
        h, dh = h / rcov[:, None], dh / rcov[:, None]
    elif mod == 2:
        h = np.exp(-h / rcov[:, None])
        dh = -h / rcov[:, None]
    elif mod != 0:
        raise ValueError(f"Unknown H(r) modifier: {mod}")
    layers = net["layers"]
    for l, (W, b) in enumerate(layers):
        W = np.asarray(W, dtype=np.float64)
        z = h @ W + (0.0 if b is None else np.asarray(b, dtype=np.float64))
        dz = dh @ W
        if l == len(layers) - 1:
            return z, dz
        a, da = activation(net.get("activation", "softplus"), z)
        da = da * dz
        if net.get("use_resnet_dt", True) and l > 0 and W.shape[0] == W.shape[1]:
            a, da = a + h, da + dh
        h, dh = a, da


# packed moment components: exponents (nx, ny, nz) of the unit vector, in the reference's order
# (grap.py:501-511): 1 | x y z | xx xy xz yy yz zz | xxx xxy xxz xyy xyz xzz yyy yyz yzz zzz, and
# continued the same way (nx descending, then ny) for the ranks 4 and 5
COMPONENTS = [(nx, ny, deg - nx - ny) for deg in range(6) for nx in range(deg, -1, -1)
              for ny in range(deg - nx, -1, -1)]
N_COMPONENTS = {0: 1, 1: 4, 2: 10, 3: 20, 4: 35, 5: 56}


def multiplicity_tensor(max_moment, symmetric=False):
    """T[d, m]. Moments up to 3: grap.py:470-492 (with the traceless corrections when `symmetric`).
    max_moment > 3: the reference sums the FULL 3^m tensors with unit weights (get_moment_tensor /
    get_T_dm, grap.py:538-600), which on packed components is the multinomial coefficient of every
    component at every rank; `symmetric` plays no role there."""
    from math import factorial
    nd = N_COMPONENTS[max_moment]
    T = np.zeros((nd, max_moment + 1))
    for d, (nx, ny, nz) in enumerate(COMPONENTS[:nd]):
        m = nx + ny + nz
        T[d, m] = factorial(m) / (factorial(nx) * factorial(ny) * factorial(nz))
    if symmetric and max_moment <= 3:
        if max_moment >= 2:
            T[0, 2] = -1.0 / 3.0
        if max_moment >= 3:
            T[1:4, 3] = -3.0 / 5.0
    return T


def full_moment_sums(P_of_components, u, H, max_moment):
    """The reference's literal max_moment > 3 formulation for ONE centre and block, as a check of
    the packed one: M = [1, u, u (x) u, ...] flattened (1 + 3 + 9 + ... + 3^m components,
    grap.py:538-575), P = H^T M, Q[k, m] = sum over the 3^m components of rank m of P^2 (T_dm of
    ones, :577-600). `u` [n, 3], `H` [n, K] -> Q [K, max_moment + 1]."""
    blocks = [np.ones((len(u), 1))]
    for _ in range(max_moment):
        prev = blocks[-1]
        blocks.append((prev[:, :, None] * u[:, None, :]).reshape(len(u), -1))
    Q = np.zeros((H.shape[1], max_moment + 1))
    for m, M in enumerate(blocks):
        P = H.T @ M            # [K, 3^m]
        Q[:, m] = (P ** 2).sum(axis=1)
    return Q


def moment_coefficients(u, max_moment):
    """M[p, d] = ux^nx uy^ny uz^nz for the packed components."""
    nd = N_COMPONENTS[max_moment]
    return np.stack([u[:, 0] ** nx * u[:, 1] ** ny * u[:, 2] ** nz for nx, ny, nz in COMPONENTS[:nd]], axis=1)


class GrapModel:
    def __init__(self, elements, rcut, algorithm="sf", parameters=None, param_space_method="pair",
                 moment_tensors=0, cutoff_function="cosine", symmetric=False, legacy_mode=True,
                 weights=None, activation="softplus", use_resnet_dt=False, minmax=None,
                 filter_net=None):
        self.elements = sorted(set(elements))
        self.rcut = float(rcut)
        self.algorithm = algorithm
        self.filter_net = filter_net  # algorithm "nn": see nn_filters
        if algorithm == "nn":
            self.grid = [None] * int(np.shape(filter_net["layers"][-1][0])[1])
        else:
            self.grid = parameter_grid(algorithm, parameters, param_space_method)
        if isinstance(moment_tensors, int):
            moment_tensors = [moment_tensors]
        self.moment_tensors = list(set(moment_tensors))  # grap.py:295
        self.cutoff_function = cutoff_function
        self.symmetric = bool(symmetric)
        self.legacy_mode = bool(legacy_mode)
        self.weights = weights
        self.activation = activation
        self.use_resnet_dt = bool(use_resnet_dt)
        self.minmax = minmax

    @property
    def max_moment(self):
        return max(self.moment_tensors)

    @property
    def features_per_filter(self):
        if self.legacy_mode:
            return len([m for m in self.moment_tensors if m in (0, 1, 2)])  # grap.py:423-460
        return self.max_moment + 1                                          # grap.py:606

    @property
    def ndim(self):
        return self.features_per_filter * len(self.grid) * len(self.elements)


def _pairs(model, symbols, positions, cell, pbc, eps):
    R = np.asarray(positions, dtype=np.float64).reshape(-1, 3)
    pbc = np.asarray(pbc, dtype=bool).reshape(3)
    h = _complete_cell(cell, pbc)
    pi, pj, pS = neighbor_list(R, h, pbc, model.rcut)
    D, r = pair_geometry(R, h, pi, pj, pS, eps)
    block = np.array([radial_term_index(model.elements, symbols[a], symbols[b]) for a, b in zip(pi, pj)],
                     dtype=np.int64)
    return R, h, pi, pj, D, r, block


def descriptors_legacy(model, symbols, positions, cell, pbc, eps=EPS64):
    """grap.py:378-468, literally: per (term, filter, moment) sums of v fc x_a x_b / r^m, squared."""
    N = len(symbols)
    R, h, pi, pj, D, r, block = _pairs(model, symbols, positions, cell, pbc, eps)
    fc, _ = cutoff(r, model.rcut, model.cutoff_function)
    K, nf = len(model.grid), model.features_per_filter
    G = np.zeros((N, model.ndim))
    pairs9 = [(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2), (1, 0), (2, 0), (2, 1)]
    for k, row in enumerate(model.grid):
        v, _ = radial_filter(model.algorithm, row, r, model.rcut)
        col = 0
        for m in model.moment_tensors:
            base = (block * K + k) * nf + col
            if m == 0:
                np.add.at(G, (pi, base), v * fc)
            elif m == 1:
                for a in range(3):
                    s = np.zeros((N, len(model.elements)))
                    np.add.at(s, (pi, block), v * fc * D[:, a] / r)
                    for b in range(len(model.elements)):
                        G[:, (b * K + k) * nf + col] += s[:, b] ** 2
            elif m == 2:
                for a, c in pairs9:
                    s = np.zeros((N, len(model.elements)))
                    np.add.at(s, (pi, block), v * fc * D[:, a] * D[:, c] / (r * r))
                    for b in range(len(model.elements)):
                        G[:, (b * K + k) * nf + col] += s[:, b] ** 2
            else:
                continue
            col += 1
    return G


def _moments(model, symbols, positions, cell, pbc, eps):
    """P[i, block, k, d] = sum_j v_k(r) fc(r) M_d(u) and everything the backward pass needs."""
    N = len(symbols)
    R, h, pi, pj, D, r, block = _pairs(model, symbols, positions, cell, pbc, eps)
    fc, dfc = cutoff(r, model.rcut, model.cutoff_function)
    K, nel = len(model.grid), len(model.elements)
    mm = model.max_moment
    u = D / r[:, None]
    M = moment_coefficients(u, mm)
    H = np.zeros((len(r), K))
    dH = np.zeros((len(r), K))
    if model.algorithm == "nn":
        rcov = np.array([COVALENT_RADII[symbols[i]] for i in pi]) \
            if model.filter_net.get("h_abck_modifier") else None
        v, dv = nn_filters(r, model.filter_net, rcov)
        H = v * fc[:, None]
        dH = dv * fc[:, None] + v * dfc[:, None]
    for k, row in enumerate(model.grid if model.algorithm != "nn" else []):
        v, dv = radial_filter(model.algorithm, row, r, model.rcut)
        H[:, k] = v * fc
        dH[:, k] = dv * fc + v * dfc
    P = np.zeros((N, nel, K, M.shape[1]))
    np.add.at(P, (pi, block), H[:, :, None] * M[:, None, :])
    return dict(R=R, h=h, pi=pi, pj=pj, D=D, r=r, block=block, u=u, M=M, H=H, dH=dH, P=P)


def descriptors_new(model, symbols, positions, cell, pbc, eps=EPS64, geometry=None):
    """grap.py:592-683: Q = T . P^2, G = [sign(P0) sqrt(Q0 + 1e-16), Q1, ..., Qm]."""
    g = geometry or _moments(model, symbols, positions, cell, pbc, eps)
    T = multiplicity_tensor(model.max_moment, model.symmetric)
    Q = np.einsum("nbkd,dm->nbkm", g["P"] ** 2, T)
    G = Q.copy()
    G[..., 0] = np.sign(g["P"][..., 0]) * np.sqrt(Q[..., 0] + 1e-16)
    return G.reshape(len(symbols), -1)


def descriptors(model, symbols, positions, cell, pbc, eps=EPS64):
    fn = descriptors_legacy if model.legacy_mode else descriptors_new
    return fn(model, symbols, positions, cell, pbc, eps)


def evaluate(model: GrapModel, symbols, positions, cell, pbc, eps=EPS64):
    """descriptors, energy, atomic, forces, virial, stress_voigt, total_pressure, dEdG."""
    symbols = list(symbols)
    N = len(symbols)
    g = _moments(model, symbols, positions, cell, pbc, eps)
    G = descriptors(model, symbols, positions, cell, pbc, eps)
    volume = abs(np.linalg.det(g["h"]))
    out = {"descriptors": G.copy(), "volume": volume, "npairs": len(g["pi"])}
    if model.weights is None:
        return out
    atomic, dEdG = apply_mlp(model, symbols, G)
    K, nel, mm = len(model.grid), len(model.elements), model.max_moment
    nf = model.features_per_filter
    dEdG4 = dEdG.reshape(N, nel, K, nf)
    P = g["P"]
    # dE/dP[i, b, k, d] = 2 P sum_m c_m T[d, m]
    if model.legacy_mode:
        T = multiplicity_tensor(mm, False)
        c = np.zeros((N, nel, K, mm + 1))
        lin = np.zeros((N, nel, K))  # the m = 0 feature is P0 itself
        col = 0
        for m in model.moment_tensors:
            if m == 0:
                lin = dEdG4[..., col]
            elif m in (1, 2):
                c[..., m] = dEdG4[..., col]
            else:
                continue
            col += 1
        dEdP = 2.0 * P * np.einsum("nbkm,dm->nbkd", c, T)
        dEdP[..., 0] += lin
    else:
        T = multiplicity_tensor(mm, model.symmetric)
        c = dEdG4.copy()
        q0 = P[..., 0] ** 2
        c[..., 0] = dEdG4[..., 0] * np.sign(P[..., 0]) / (2.0 * np.sqrt(q0 + 1e-16))
        dEdP = 2.0 * P * np.einsum("nbkm,dm->nbkd", c, T)
    # per directed pair: dE/dD = sum_kd A[k,d] (dH_k M_d u + H_k dM_d/dD)
    pi, pj, D, r, u, M = g["pi"], g["pj"], g["D"], g["r"], g["u"], g["M"]
    A = dEdP[pi, g["block"]]                       # [P, K, nd]
    a_d = np.einsum("pkd,pk->pd", A, g["H"])
    b_d = np.einsum("pkd,pk->pd", A, g["dH"])
    nd = M.shape[1]
    deg = np.array([sum(c_) for c_ in COMPONENTS[:nd]], dtype=np.float64)
    dMdu = np.zeros((len(r), nd, 3))
    for d, (nx, ny, nz) in enumerate(COMPONENTS[:nd]):
        ex = [nx, ny, nz]
        for a in range(3):
            if ex[a] == 0:
                continue
            e2 = list(ex)
            e2[a] -= 1
            dMdu[:, d, a] = ex[a] * u[:, 0] ** e2[0] * u[:, 1] ** e2[1] * u[:, 2] ** e2[2]
    # dM/dD_c = (dM/du_c - deg M u_c) / r   (u = D / r, M homogeneous of degree deg)
    dMdD = (dMdu - (deg[None, :] * M)[:, :, None] * u[:, None, :]) / r[:, None, None]
    gpair = (b_d * M).sum(axis=1)[:, None] * u + np.einsum("pd,pdc->pc", a_d, dMdD)
    F = np.zeros((N, 3))
    np.add.at(F, pi, gpair)
    np.add.at(F, pj, -gpair)
    W = gpair.T @ D
    stress = W / volume
    voigt = np.array([stress[0, 0], stress[1, 1], stress[2, 2], stress[1, 2], stress[0, 2], stress[0, 1]])
    out.update(energy=float(atomic.sum()), atomic=atomic, dEdG=dEdG, forces=F, virial=W,
               stress_voigt=voigt, total_pressure=float(np.trace(stress) / (-3.0 * GPA)))
    return out
