"""
ORACLE (test infrastructure only) — NumPy float64 restatement of the
reference's EAM / ADP path with the analytic Zhou-Johnson-Wadley (2004)
functions and Mishin's angular-dependent terms, with analytic forces and
virial.

  zhou_exp, density_exp .......... nn/eam/potentials/generic.py:87-117
  Zjw04.rho / phi / embed ........ nn/eam/potentials/zjw04.py:187-389
  mishin_cutoff / mishin_polar ... nn/eam/potentials/generic.py:52-84
  MishinH.dipole / quadrupole .... nn/eam/potentials/mishin.py:269-315
  rho_i = sum_j rho_{s_j}(r_ij) .. nn/eam/alloy.py:128-196
  0.5 sum_j phi(r_ij) ............ nn/eam/eam.py:300-362
  F(rho_i), stitching ............ nn/eam/eam.py:401-493
  dipole / quadrupole energies ... nn/eam/adp.py:315-498 (squared PER k-body term)
  y = phi + embed (+ dip + quad) . nn/eam/eam.py:568, nn/eam/adp.py:584
  "nn" functions (the default) ... nn/eam/eam.py:174-190 (`convolution1x1` on the scalar r or
                                   rho, nn/convolutional.py:154-300: no output bias, no cutoff;
                                   padded slots are removed by the masks, eam.py:344, alloy.py:181)

The parameter values below are the published Zhou-Johnson-Wadley constants
(Phys. Rev. B 69, 144113) as listed at zjw04.py:19-152, and Mishin's Ni-Ni
ADP constants as listed at mishin.py:62-66. They are data, not code.

Pinned by the reference's fixtures test_files/lammps/Zhou_AlCu.alloy.eam,
MoNi_Zhou04.eam.alloy, zjw04_Ni.alloy.eam (function tables) and
test_files/crystals/Ni_fc2.npy (whole-structure forces through the Hessian).
ADP energies are PARITY UNPINNED (the reference's ADP tests need the missing
spline extension).
"""
import numpy as np

from .neighbors import neighbor_list, _complete_cell

EPS64 = 1e-14
GPA = 1.0 / 160.21766208

ZJW04_KEYS = ["r_eq", "f_eq", "rho_e", "rho_s", "alpha", "beta", "A", "B", "kappa", "lamda",
              "Fn0", "Fn1", "Fn2", "Fn3", "F0", "F1", "F2", "F3", "eta", "Fe"]

ZJW04 = {
    "Al": dict(r_eq=2.863924, f_eq=1.403115, rho_e=20.418205, rho_s=23.195740, alpha=6.613165,
               beta=3.527021, A=0.314873, B=0.365551, kappa=0.379846, lamda=0.759692,
               Fn0=-2.807602, Fn1=-0.301435, Fn2=1.258562, Fn3=-1.247604, F0=-2.83, F1=0.0,
               F2=0.622245, F3=-2.488244, eta=0.785902, Fe=-2.824528),
    "Cu": dict(r_eq=2.556162, f_eq=1.554485, rho_e=21.175871, rho_s=21.175395, alpha=8.127620,
               beta=4.334731, A=0.396620, B=0.548085, kappa=0.308782, lamda=0.756515,
               Fn0=-2.170269, Fn1=-0.263788, Fn2=1.088878, Fn3=-0.817603, F0=-2.19, F1=0.0,
               F2=0.561830, F3=-2.100595, eta=0.310490, Fe=-2.186568),
    "Ni": dict(r_eq=2.488746, f_eq=2.007018, rho_e=27.562015, rho_s=27.930410, alpha=8.383453,
               beta=4.471175, A=0.429046, B=0.633531, kappa=0.443599, lamda=0.820658,
               Fn0=-2.693513, Fn1=-0.076445, Fn2=0.241442, Fn3=-2.375626, F0=-2.70, F1=0.0,
               F2=0.265390, F3=-0.152856, eta=0.469000, Fe=-2.699486),
    "Mo": dict(r_eq=2.7281, f_eq=2.72371, rho_e=29.354065, rho_s=29.354065, alpha=8.393531,
               beta=4.47655, A=0.708787, B=1.120373, kappa=0.13764, lamda=0.27528,
               Fn0=-3.692913, Fn1=-0.178812, Fn2=0.38045, Fn3=-3.13365, F0=-3.71, F1=0.0,
               F2=0.875874, F3=0.776222, eta=0.790879, Fe=-3.712093),
}

MISHIN_NINI = dict(d1=4.4657e-3, d2=-1.3702e0, d3=-0.9611e-1, q1=6.4502e0, q2=0.2608e-1,
                   q3=-6.0208e0, h=3.323, rc=5.168)


# ---- scalar functions with derivatives ------------------------------------------

def zhou_exp(r, a, b, c, re):
    """f(r) = a exp(-b (r/re - 1)) / (1 + (r/re - c)^20) and df/dr."""
    x = r / re
    t = x - c
    t20 = t ** 20
    up = a * np.exp(-b * (x - 1.0))
    f = up / (1.0 + t20)
    df = f * (-b - 20.0 * t ** 19 / (1.0 + t20)) / re
    return f, df


def zjw04_rho(r, p):
    return zhou_exp(r, p["f_eq"], p["beta"], p["lamda"], p["r_eq"])


def zjw04_phi_aa(r, p):
    fa, dfa = zhou_exp(r, p["A"], p["alpha"], p["kappa"], p["r_eq"])
    fb, dfb = zhou_exp(r, p["B"], p["beta"], p["lamda"], p["r_eq"])
    return fa - fb, dfa - dfb


def zjw04_phi(r, pa, pb, same):
    """phi_AA, or phi_AB = 0.5 (rho_A/rho_B phi_BB + rho_B/rho_A phi_AA) (zjw04.py:229-243)."""
    if same:
        return zjw04_phi_aa(r, pa)
    pha, dpha = zjw04_phi_aa(r, pa)
    phb, dphb = zjw04_phi_aa(r, pb)
    ra, dra = zjw04_rho(r, pa)
    rb, drb = zjw04_rho(r, pb)
    q1, q2 = ra / rb, rb / ra
    dq1 = (dra * rb - ra * drb) / (rb * rb)
    dq2 = (drb * ra - rb * dra) / (ra * ra)
    return 0.5 * (q1 * phb + q2 * pha), 0.5 * (dq1 * phb + q1 * dphb + dq2 * pha + q2 * dpha)


def zjw04_embed(rho, p):
    """Piecewise embedding energy and dF/drho (zjw04.py:319-386)."""
    rho = np.asarray(rho, dtype=np.float64)
    rho_n, rho_0 = 0.85 * p["rho_e"], 1.15 * p["rho_e"]
    F = np.zeros_like(rho)
    dF = np.zeros_like(rho)
    m1 = rho < rho_n
    m2 = (rho >= rho_n) & (rho < rho_0)
    m3 = rho >= rho_0
    x = rho[m1] / rho_n - 1.0
    F[m1] = p["Fn0"] + p["Fn1"] * x + p["Fn2"] * x ** 2 + p["Fn3"] * x ** 3
    dF[m1] = (p["Fn1"] + 2 * p["Fn2"] * x + 3 * p["Fn3"] * x ** 2) / rho_n
    x = rho[m2] / p["rho_e"] - 1.0
    F[m2] = p["F0"] + p["F1"] * x + p["F2"] * x ** 2 + p["F3"] * x ** 3
    dF[m2] = (p["F1"] + 2 * p["F2"] * x + 3 * p["F3"] * x ** 2) / p["rho_e"]
    x = rho[m3] / p["rho_s"]
    lnx = np.log(x)
    F[m3] = p["Fe"] * (1.0 - p["eta"] * lnx) * x ** p["eta"]
    dF[m3] = -p["Fe"] * p["eta"] ** 2 * lnx * x ** (p["eta"] - 1.0) / p["rho_s"]
    return F, dF


def zjw04xc_embed(rho, p):
    """Zjw04xc / Zjw04uxc embedding: the three Zjw04 branches blended with
    c1 = sigmoid(2 (rho_n - rho)), c3 = sigmoid(2 (rho - rho_0)), c2 = 1 - c1 - c3, and
    x = rho / rho_s + 1e-8 in the third branch (reference zjw04.py:482-543)."""
    rho = np.asarray(rho, dtype=np.float64)
    rho_n, rho_0 = 0.85 * p["rho_e"], 1.15 * p["rho_e"]
    x1 = rho / rho_n - 1.0
    y1 = p["Fn0"] + p["Fn1"] * x1 + p["Fn2"] * x1 ** 2 + p["Fn3"] * x1 ** 3
    d1 = (p["Fn1"] + 2 * p["Fn2"] * x1 + 3 * p["Fn3"] * x1 ** 2) / rho_n
    x2 = rho / p["rho_e"] - 1.0
    y2 = p["F0"] + p["F1"] * x2 + p["F2"] * x2 ** 2 + p["F3"] * x2 ** 3
    d2 = (p["F1"] + 2 * p["F2"] * x2 + 3 * p["F3"] * x2 ** 2) / p["rho_e"]
    x3 = rho / p["rho_s"] + 1e-8
    lnx = np.log(x3)
    y3 = p["Fe"] * (1.0 - p["eta"] * lnx) * x3 ** p["eta"]
    d3 = -p["Fe"] * p["eta"] ** 2 * lnx * x3 ** (p["eta"] - 1.0) / p["rho_s"]
    sig = lambda z: 1.0 / (1.0 + np.exp(-z))
    c1, c3 = sig(2.0 * (rho_n - rho)), sig(2.0 * (rho - rho_0))
    c2 = 1.0 - (c1 + c3)
    dc1, dc3 = -2.0 * c1 * (1.0 - c1), 2.0 * c3 * (1.0 - c3)
    F = c1 * y1 + c2 * y2 + c3 * y3
    dF = c1 * d1 + c2 * d2 + c3 * d3 + dc1 * y1 - (dc1 + dc3) * y2 + dc3 * y3
    return F, dF


def zjw04xcp_phi_ab(r, q):
    """Cross-element phi of Zjw04xcp: the AA form with the pair's own constants
    (reference zjw04.py:676-693)."""
    fa, dfa = zhou_exp(r, q["A"], q["alpha"], q["kappa"], q["r_eq"])
    fb, dfb = zhou_exp(r, q["B"], q["beta"], q["lamda"], q["r_eq"])
    return fa - fb, dfa - dfb


def mishin_polar(r, p1, p2, p3, rc, h):
    """(p1 exp(-p2 r) + p3) psi((r - rc)/h), psi(x) = x^4/(1+x^4) for x < 0 else 0."""
    z = (r - rc) / h
    zz = np.maximum(-z, 0.0)           # relu(-x), generic.py:61
    z4 = zz ** 4
    psi = z4 / (1.0 + z4)
    dpsi = np.where(z < 0, -4.0 * zz ** 3 / (1.0 + z4) ** 2 / h, 0.0)  # d psi / d r
    left = p1 * np.exp(-p2 * r) + p3
    dleft = -p1 * p2 * np.exp(-p2 * r)
    return left * psi, dleft * psi + left * dpsi


def nn_function(x, layers, act="softplus", resnet=False):
    """Scalar function f(x) given by a 1x1 CNN (eam.py:174-190; convolutional.py:257-290) and its
    derivative f'(x), by forward-mode differentiation. `layers` = [(W [in, out], b or None), ...],
    the last one being the linear output layer."""
    from .sf import activation
    x = np.asarray(x, dtype=np.float64)
    h = x[:, None]
    dh = np.ones_like(h)
    n = len(layers)
    for l, (W, b) in enumerate(layers):
        W = np.asarray(W, dtype=np.float64)
        z = h @ W + (0.0 if b is None else np.asarray(b, dtype=np.float64))
        dz = dh @ W
        if l < n - 1:
            a, da = activation(act, z)
            da = da * dz
            if resnet and l > 0 and W.shape[0] == W.shape[1]:
                a = a + h
                da = da + dh
            h, dh = a, da
        else:
            h, dh = z, dz
    return h[:, 0], dh[:, 0]


def sutton90_rho(r, p):
    """AgSutton90 (potentials/sutton90.py:47-100): rho = (a / r)^6, phi = (b / r)^12, F = -sqrt(rho)."""
    f = (p["a"] / r) ** 6
    return f, -6.0 * f / r


def sutton90_phi(r, p):
    f = (p["b"] / r) ** 12
    return f, -12.0 * f / r


def sutton90_embed(rho, p):
    s = np.sqrt(rho)
    return -s, -0.5 / s


def _morse(r, d, g, r0):
    e1 = np.exp(-g * (r - r0))
    return d * (e1 * e1 - 2.0 * e1), 2.0 * d * g * (e1 - e1 * e1)   # generic.py:15-30, agrawal.py:20-32


def agrawal_rho(r, p):
    """AgrawalBe "Be/1" (potentials/agrawal.py:57-79)."""
    e = p["A"] * np.exp(-p["B"] * (r - p["re"]))
    ec = p["A"] * np.exp(-p["B"] * (p["rc"] - p["re"]))
    x = r / p["rc"]
    f = e - ec + p["rc"] / p["m"] * (1.0 - x ** p["m"]) * (-p["B"] * ec)
    return f, -p["B"] * e + x ** (p["m"] - 1.0) * p["B"] * ec


def agrawal_phi(r, p):
    """agrawal.py:124-152."""
    m0, dm0 = _morse(r, p["D"], p["alpha"], p["re"])
    mc, dmc = _morse(p["rc"], p["D"], p["alpha"], p["re"])
    x = r / p["rc"]
    return m0 - mc + p["rc"] / p["m"] * (1.0 - x ** p["m"]) * dmc, dm0 - x ** (p["m"] - 1.0) * dmc


def agrawal_embed(rho, p):
    """agrawal.py:81-122."""
    L = np.log(np.maximum(rho, 1e-12))
    F = p["F0"] * (1.0 - p["beta"] * L) * rho ** p["beta"] + p["F1"] * rho ** p["gamma"]
    dF = -p["F0"] * p["beta"] ** 2 * L * rho ** (p["beta"] - 1.0) + p["F1"] * p["gamma"] * rho ** (p["gamma"] - 1.0)
    return F, dF


def grimes_rho(r, p):
    """RWGrimes (potentials/grimmes.py:67-86): n / r^8 (1/2 + 1/2 erf(20 (r - 3/2)))."""
    from scipy.special import erf
    t = 20.0 * (r - 1.5)
    sw = 0.5 + 0.5 * erf(t)
    dsw = 20.0 / np.sqrt(np.pi) * np.exp(-t * t)
    return p["n"] / r ** 8 * sw, p["n"] / r ** 8 * (dsw - 8.0 * sw / r)


def grimes_phi(r, p):
    """grimmes.py:40-65: Morse + Buckingham (generic.py:15-49)."""
    m0, dm0 = _morse(r, p["D"], p["gamma"], p["r0"])
    eb = p["A"] * np.exp(-r / p["rho"])
    return m0 + eb - p["C"] / r ** 6, dm0 - eb / p["rho"] + 6.0 * p["C"] / r ** 7


def grimes_embed(rho, p):
    """grimmes.py:88-100: -G sqrt(rho)."""
    s = np.sqrt(rho)
    return -p["G"] * s, -0.5 * p["G"] / s


OTHER = {"grimes": (grimes_rho, grimes_phi, grimes_embed),
         "sutton90": (sutton90_rho, sutton90_phi, sutton90_embed),
         "be/1": (agrawal_rho, agrawal_phi, agrawal_embed)}


def spline_function(x, table):
    """Tabulated function (knots, values) as the natural cubic spline the reference builds with
    `CubicInterpolator(x, y, natural_boundary=True)` (potentials/tests/test_mishin.py:60-70), here
    by SciPy; beyond the last knot the last cubic continues. Returns f(x), f'(x)."""
    from scipy.interpolate import CubicSpline
    cs = CubicSpline(np.asarray(table[0], dtype=np.float64), np.asarray(table[1], dtype=np.float64),
                     bc_type="natural", extrapolate=True)
    x = np.asarray(x, dtype=np.float64)
    return cs(x), cs(x, 1)


# ---- model ----------------------------------------------------------------------

class EamModel:
    """elements (sorted), rcut, per-element Zjw04 parameters, optional ADP pair parameters
    keyed by the sorted pair 'AB' (dict with d1..q3, h, rc)."""

    def __init__(self, elements, rcut, params=None, adp=None, blended_embed=False, phi_pairs=None,
                 nets=None, activation="softplus", tables=None, other=None):
        self.elements = sorted(set(elements))
        self.rcut = float(rcut)
        # "nn" functions: {'rho': {el: layers}, 'embed': {el: layers}, 'phi': {'AB': layers},
        # 'dipole': {'AB': layers}, 'quadrupole': {'AB': layers}}; a function without an entry is
        # the analytic one
        self.nets = nets or {}
        self.activation = activation
        # tabulated functions, same nesting: {'rho': {el: (x, y)}, 'phi': {'AB': (x, y)}, ...}
        self.tables = tables or {}
        # elements whose analytic functions are sutton90 / Be/1: {el: (kind, constants dict)}
        self.other = other or {}
        self.params = params if params is not None else {
            e: dict(ZJW04[e]) for e in self.elements if e in ZJW04}
        self.adp = adp  # {'NiNi': {...}} or None
        self.blended_embed = bool(blended_embed)  # Zjw04xc / uxc / xcp embedding
        self.phi_pairs = phi_pairs or {}          # Zjw04xcp: {'MoNi': {r_eq, A, B, ...}}


def evaluate(model: EamModel, symbols, positions, cell, pbc, eps=EPS64):
    els = model.elements
    R = np.asarray(positions, dtype=np.float64).reshape(-1, 3)
    N = len(R)
    pbc = np.asarray(pbc, dtype=bool).reshape(3)
    h = _complete_cell(cell, pbc)
    volume = abs(np.linalg.det(h))
    spec = np.array([els.index(s) for s in symbols])
    pi, pj, pS = neighbor_list(R, h, pbc, model.rcut)
    D = R[pj] - R[pi] + pS.astype(np.float64) @ h
    r = np.sqrt(np.sum(D * D, axis=1) + eps)
    si, sj = spec[pi], spec[pj]

    rho_pair = np.zeros(len(pi))
    drho_pair = np.zeros(len(pi))
    phi_pair = np.zeros(len(pi))
    dphi_pair = np.zeros(len(pi))
    for b, eb in enumerate(els):
        m = sj == b
        if eb in model.tables.get("rho", {}):
            rho_pair[m], drho_pair[m] = spline_function(r[m], model.tables["rho"][eb])
        elif eb in model.nets.get("rho", {}):
            rho_pair[m], drho_pair[m] = nn_function(r[m], model.nets["rho"][eb], model.activation)
        elif eb in model.other:
            rho_pair[m], drho_pair[m] = OTHER[model.other[eb][0]][0](r[m], model.other[eb][1])
        else:
            rho_pair[m], drho_pair[m] = zjw04_rho(r[m], model.params[eb])  # neighbour's element, alloy.py:176
        for a, ea in enumerate(els):
            mm = m & (si == a)
            key = "".join(sorted([ea, eb]))
            if key in model.tables.get("phi", {}):
                phi_pair[mm], dphi_pair[mm] = spline_function(r[mm], model.tables["phi"][key])
            elif key in model.nets.get("phi", {}):
                phi_pair[mm], dphi_pair[mm] = nn_function(r[mm], model.nets["phi"][key], model.activation)
            elif a == b and ea in model.other:
                phi_pair[mm], dphi_pair[mm] = OTHER[model.other[ea][0]][1](r[mm], model.other[ea][1])
            elif a != b and key in model.phi_pairs:
                phi_pair[mm], dphi_pair[mm] = zjw04xcp_phi_ab(r[mm], model.phi_pairs[key])
            else:
                phi_pair[mm], dphi_pair[mm] = zjw04_phi(r[mm], model.params[ea], model.params[eb], a == b)
    rho = np.zeros(N)
    np.add.at(rho, pi, rho_pair)
    phisum = np.zeros(N)
    np.add.at(phisum, pi, phi_pair)
    F = np.zeros(N)
    dF = np.zeros(N)
    for a, ea in enumerate(els):
        m = spec == a
        if ea in model.tables.get("embed", {}):
            F[m], dF[m] = spline_function(rho[m], model.tables["embed"][ea])
            continue
        if ea in model.nets.get("embed", {}):
            F[m], dF[m] = nn_function(rho[m], model.nets["embed"][ea], model.activation)
            continue
        if ea in model.other:
            F[m], dF[m] = OTHER[model.other[ea][0]][2](rho[m], model.other[ea][1])
            continue
        embed = zjw04xc_embed if model.blended_embed else zjw04_embed
        F[m], dF[m] = embed(rho[m], model.params[ea])
    atomic = F + 0.5 * phisum
    # dE/dD of the directed pair (i -> j): centre i's terms only
    s = dF[pi] * drho_pair + 0.5 * dphi_pair
    g = (s / r)[:, None] * D

    if model.adp is not None:
        nel = len(els)
        mu = np.zeros((N, nel, 3))
        lam = np.zeros((N, nel, 3, 3))
        u = np.zeros(len(pi))
        du = np.zeros(len(pi))
        w = np.zeros(len(pi))
        dw = np.zeros(len(pi))
        for a, ea in enumerate(els):
            for b, eb in enumerate(els):
                key = "".join(sorted([ea, eb]))
                m = (si == a) & (sj == b)
                p = model.adp.get(key)
                if key in model.tables.get("dipole", {}):
                    u[m], du[m] = spline_function(r[m], model.tables["dipole"][key])
                elif key in model.nets.get("dipole", {}):
                    u[m], du[m] = nn_function(r[m], model.nets["dipole"][key], model.activation)
                elif p is not None:
                    u[m], du[m] = mishin_polar(r[m], p["d1"], p["d2"], p["d3"], p["rc"], p["h"])
                if key in model.tables.get("quadrupole", {}):
                    w[m], dw[m] = spline_function(r[m], model.tables["quadrupole"][key])
                elif key in model.nets.get("quadrupole", {}):
                    w[m], dw[m] = nn_function(r[m], model.nets["quadrupole"][key], model.activation)
                elif p is not None:
                    w[m], dw[m] = mishin_polar(r[m], p["q1"], p["q2"], p["q3"], p["rc"], p["h"])
        np.add.at(mu, (pi, sj), u[:, None] * D)
        np.add.at(lam, (pi, sj), w[:, None, None] * D[:, :, None] * D[:, None, :])
        nu = np.trace(lam, axis1=2, axis2=3)
        e_dip = 0.5 * np.sum(mu * mu, axis=(1, 2))
        e_quad = np.sum(0.5 * np.sum(lam * lam, axis=(2, 3)) - nu * nu / 6.0, axis=1)
        atomic = atomic + e_dip + e_quad
        mu_p = mu[pi, sj]                                   # [P, 3]
        Lam_p = lam[pi, sj] - (nu[pi, sj] / 3.0)[:, None, None] * np.eye(3)
        muD = np.sum(mu_p * D, axis=1)
        LD = np.einsum("pab,pb->pa", Lam_p, D)
        DLD = np.sum(D * LD, axis=1)
        g = g + (muD * du / r)[:, None] * D + u[:, None] * mu_p \
            + (DLD * dw / r)[:, None] * D + 2.0 * w[:, None] * LD

    forces = np.zeros((N, 3))
    np.add.at(forces, pi, g)
    np.add.at(forces, pj, -g)
    W = g.T @ D
    stress = W / volume
    voigt = np.array([stress[0, 0], stress[1, 1], stress[2, 2], stress[1, 2], stress[0, 2], stress[0, 1]])
    return dict(energy=float(np.sum(atomic)), atomic=atomic, forces=forces, virial=W,
                stress_voigt=voigt, total_pressure=float(np.trace(stress) / (-3.0 * GPA)),
                rho=rho, npairs=len(pi), volume=volume)


# ---- setfl reader (fixture parser; mirrors the layout read by io/lammps.py:107-221) --------

def read_setfl(path, adp=False):
    with open(path) as fp:
        lines = fp.read().split("\n")
    head = lines[3].split()
    nel = int(head[0])
    els = head[1:1 + nel]
    nrho, drho, nr, dr, rcut = lines[4].split()[:5]
    nrho, nr = int(nrho), int(nr)
    drho, dr, rcut = float(drho), float(dr), float(rcut)
    tokens = " ".join(lines[5:]).split()
    pos = 0
    out = dict(elements=els, nrho=nrho, drho=drho, nr=nr, dr=dr, rcut=rcut, embed={}, rho={},
               rphi={}, u={}, w={})
    for el in els:
        pos += 4  # Z, mass, lattice constant, lattice type
        out["embed"][el] = np.array(tokens[pos:pos + nrho], dtype=float)
        pos += nrho
        out["rho"][el] = np.array(tokens[pos:pos + nr], dtype=float)
        pos += nr
    pairs = [(i, j) for i in range(nel) for j in range(i, nel)]  # io/lammps.py:150-160
    for i, j in pairs:
        out["rphi"][els[i] + els[j]] = np.array(tokens[pos:pos + nr], dtype=float)
        pos += nr
    if adp:
        for key in ("u", "w"):
            for i, j in pairs:
                out[key][els[i] + els[j]] = np.array(tokens[pos:pos + nr], dtype=float)
                pos += nr
    return out
