"""
EAM / ADP model descriptions mirroring the reference's model classes for the
analytic-potential hot path:

  EamAlloyNN  <- reference tensoralloy/nn/eam/alloy.py:24-127 (+ EamNN, eam.py:78-130)
  AdpNN       <- reference tensoralloy/nn/eam/adp.py (dipole / quadrupole terms)

Supported potentials: `sutton90` (AgSutton90), `Be/1` (AgrawalBe) and `grimes` (RWGrimes) for
single-element rho / embed / phi, and the Zhou-Johnson-Wadley family for rho / embed / phi --
`zjw04` (nn/eam/potentials/zjw04.py:155-412), `zjw04xc` / `zjw04uxc` (sigmoid-
blended embedding, :415-568; the two differ only in which constants are
trainable) and `zjw04xcp` (own constants for cross-element phi, :571-696) -- and
`mishinh` for the ADP dipole / quadrupole (nn/eam/potentials/mishin.py:269-315).
One family per model; the other empirical parameterisations raise `ValueError`.

"nn" functions -- the reference's default (alloy.py:110-112, adp.py:120-124): rho(r), phi(r),
F(rho), u(r), w(r) each given by a 1x1 CNN on the scalar argument (eam.py:174-190,
`convolution1x1` with `Defaults.hidden_sizes`, no output bias) -- run on the GPU as 16-row fp64
MFMA tiles over the pair list / the atoms (csrc/ta_eam.hip). Any mix of "nn" and analytic
functions inside one model is allowed, as in the reference. Weights live in
`nn.weights[section][function] = [(W [in, out], b), ..., (W_out, None)]` with section = element
symbol or sorted pair key, and are written to / read from `<stem>.npz` under
`<section>/<function>/weights_<j>`, `.../biases_<j>`.

Flat parameter block handed to the C ABI (`ta_model_desc.eam_params`):
  per element (sorted): 20 doubles in `ZJW04_KEYS` order (sutton90: a, b; Be/1: `AGRAWAL_KEYS`) +
  embed kind (0 / 1) + potential kind (0 Zjw04 family, 1 sutton90, 2 Be/1);
  per unordered element pair (a <= b, row-major upper triangle): phi kind
  (0 = Zjw04, 1 = Zjw04xcp constants) + [r_eq, A, B, alpha, beta, kappa, lamda];
  for ADP, per pair: 8 doubles [d1, d2, d3, q1, q2, q3, h, rc] (all zero = none).
Tabulated functions: a potential named `spline@<path>` (the prefix the reference's input reader
accepts, train/training.py:258-262) takes that function from the LAMMPS setfl (`EamAlloyNN`) or
adp (`AdpNN`) file at <path> and evaluates it as a natural cubic spline through every tabulated
point, as the reference's `CubicInterpolator(x, y, natural_boundary=True)` does
(nn/eam/potentials/tests/test_mishin.py:36-160). `EamAlloyNN.from_setfl(path)` /
`AdpNN.from_setfl(path)` build a model whose every function comes from one file.

Function networks travel in the MLP fields of `ta_model_desc` as `n_eam_nets` slots in the order
rho[element], embed[element], phi[pair], then for ADP dipole[pair], quadrupole[pair]; a slot with
0 layers is an analytic function.
"""
from __future__ import annotations

import json
import os
from datetime import datetime
from typing import Dict, List, Sequence

import numpy as np

from . import _lib
from .utils import Defaults, get_elements_from_kbody_term, get_kbody_terms

ZJW04_KEYS = ["r_eq", "f_eq", "rho_e", "rho_s", "alpha", "beta", "A", "B", "kappa", "lamda",
              "Fn0", "Fn1", "Fn2", "Fn3", "F0", "F1", "F2", "F3", "eta", "Fe"]

# Zhou, Johnson, Wadley, Phys. Rev. B 69, 144113 (2004): published constants
# (same values as the table at reference nn/eam/potentials/zjw04.py:19-152).
ZJW04_DEFAULTS = {
    "Al": [2.863924, 1.403115, 20.418205, 23.19574, 6.613165, 3.527021, 0.314873, 0.365551, 0.379846, 0.759692, -2.807602, -0.301435, 1.258562, -1.247604, -2.83, 0.0, 0.622245, -2.488244, 0.785902, -2.824528],
    "Cu": [2.556162, 1.554485, 21.175871, 21.175395, 8.12762, 4.334731, 0.39662, 0.548085, 0.308782, 0.756515, -2.170269, -0.263788, 1.088878, -0.817603, -2.19, 0.0, 0.56183, -2.100595, 0.31049, -2.186568],
    "Ni": [2.488746, 2.007018, 27.562015, 27.93041, 8.383453, 4.471175, 0.429046, 0.633531, 0.443599, 0.820658, -2.693513, -0.076445, 0.241442, -2.375626, -2.7, 0.0, 0.26539, -0.152856, 0.469, -2.699486],
    "Ag": [2.891814, 1.106232, 14.6041, 14.604144, 9.13201, 4.870405, 0.277758, 0.419611, 0.33971, 0.750758, -1.729364, -0.255882, 0.91205, -0.561432, -1.75, 0.0, 0.744561, -1.15065, 0.783924, -1.748423],
    "Mo": [2.7281, 2.72371, 29.354065, 29.354065, 8.393531, 4.47655, 0.708787, 1.120373, 0.13764, 0.27528, -3.692913, -0.178812, 0.38045, -3.13365, -3.71, 0.0, 0.875874, 0.776222, 0.790879, -3.712093],
    "Co": [2.505979, 1.975299, 27.206789, 27.206789, 8.679625, 4.629134, 0.421378, 0.640107, 0.5, 1.0, -2.541799, -0.219415, 0.733381, -1.589003, -2.56, 0.0, 0.705845, -0.68714, 0.694608, -2.559307],
    "Mg": [3.196291, 0.544323, 7.1326, 7.1326, 10.228708, 5.455311, 0.137518, 0.22593, 0.5, 1.0, -0.896473, -0.044291, 0.162232, -0.68995, -0.9, 0.0, 0.122838, -0.22601, 0.431425, -0.899702],
    "Fe": [2.481987, 1.885957, 20.041463, 20.041463, 9.81827, 5.236411, 0.392811, 0.646243, 0.170306, 0.340613, -2.534992, -0.059605, 0.193065, -2.282322, -2.54, 0.0, 0.200269, -0.14877, 0.39175, -2.539945],
    "Pd": [2.750897, 1.595417, 21.335246, 21.940073, 8.697397, 4.638612, 0.406763, 0.59888, 0.397263, 0.754799, -2.321006, -0.473983, 1.615343, -0.231681, -2.36, 0.0, 1.481742, -1.675615, 1.13, -2.352753],
    "W": [2.74084, 3.48734, 37.234847, 37.234847, 8.900114, 4.746728, 0.882435, 1.394592, 0.139209, 0.278417, -4.946281, -0.148818, 0.365057, -4.432406, -4.96, 0.0, 0.661935, 0.348147, -0.582714, -4.961306],
    "Ta": [2.860082, 3.086341, 33.787168, 33.787168, 8.489528, 4.527748, 0.611679, 1.032101, 0.176977, 0.353954, -5.103845, -0.405524, 1.112997, -3.585325, -5.14, 0.0, 1.640098, 0.221375, 0.848843, -5.141526],
    "Zr": [3.199978, 2.230909, 30.879991, 30.879991, 8.55919, 4.564902, 0.424667, 0.640054, 0.5, 1.0, -4.485793, -0.293129, 0.990148, -3.202516, -4.51, 0.0, 0.928602, -0.98187, 0.597133, -4.509025],
}

ZJW04_FAMILY = ("zjw04", "zjw04xc", "zjw04uxc", "zjw04xcp")
# the other eam/alloy members of the reference's `available_potentials` (potentials/__init__.py:20-30):
# AgSutton90 (sutton90.py:37-44) and AgrawalBe "Be/1" (agrawal.py:49-55); constants as published there
OTHER_POTENTIALS = ("sutton90", "be/1", "grimes")
GRIMES_KEYS = ["G", "n", "A", "rho", "C", "D", "gamma", "r0"]       # 'Pu': G, n; 'PuPu': the rest
GRIMES_DEFAULTS = {"Pu": dict(G=2.168, n=3980.058, A=18600.0, rho=0.2637, C=0.0, D=0.70185,
                              gamma=1.98008, r0=2.34591)}             # grimmes.py:33-37
SUTTON90_DEFAULTS = {"Ag": {"a": 2.928323832, "b": 2.485883762}}   # 'Ag': a, 'AgAg': b
AGRAWAL_KEYS = ["A", "B", "D", "alpha", "re", "F0", "F1", "beta", "gamma", "m", "rc"]
AGRAWAL_DEFAULTS = {"Be": dict(A=1.597, B=9.49713, D=0.41246, alpha=0.36324, re=2.29, F0=-2.0393,
                               F1=12.6178, beta=0.18752, gamma=-2.28827, m=10.0, rc=5.0)}
EL_KIND = {"zjw": 0, "sutton90": 1, "be/1": 2, "grimes": 3}
_OTHER_TABLES = {"sutton90": (SUTTON90_DEFAULTS, ["a", "b"]), "be/1": (AGRAWAL_DEFAULTS, AGRAWAL_KEYS),
                 "grimes": (GRIMES_DEFAULTS, GRIMES_KEYS)}
PHI_KEYS = ["r_eq", "A", "B", "alpha", "beta", "kappa", "lamda"]
# Zjw04xcp refits (reference nn/eam/potentials/zjw04.py:608-633; the later assignment wins)
_XCP_KEYS = ["A", "B", "F0", "F1", "F2", "F3", "Fe", "Fn0", "Fn1", "Fn2", "Fn3", "alpha", "beta",
             "eta", "f_eq", "kappa", "lamda", "r_eq", "rho_e", "rho_s"]
ZJW04XCP_ELEMENTS = {
    "Ni": dict(zip(_XCP_KEYS, [0.333956, 0.576165, -3.291077, 0.395187, 0.533360, -2.154562, -3.206066,
                               -3.353943, 0.041024, -2.098675, -7.605803, 8.401944, 3.288919, 1.182809,
                               1.543016, 0.419188, 0.857673, 2.488746, 25.423122, 26.498945])),
    "Mo": dict(zip(_XCP_KEYS, [1.070439, 1.762964, -6.613181, 2.160862, 0.587255, -4.271510, -6.847272,
                               -6.931113, 1.532229, 0.354207, -2.301498, 7.639637, 5.295918, 0.642979,
                               3.321370, 0.142495, 0.211357, 2.728100, 32.766506, 21.342554])),
}
ZJW04XCP_PAIRS = {  # zjw04.py:646-649
    "MoNi": dict(A=0.949134, B=1.360144, alpha=9.168006, beta=3.449561, kappa=0.478692,
                 lamda=0.424937, r_eq=2.235219),
}

ADP_KEYS = ["d1", "d2", "d3", "q1", "q2", "q3", "h", "rc"]
# Mishin's ADP constants (reference nn/eam/potentials/mishin.py:62-76)
_NINI = [4.4657e-3, -1.3702e0, -0.9611e-1, 6.4502e0, 0.2608e-1, -6.0208e0, 3.323, 5.168]
MISHINH_DEFAULTS = {
    "NiNi": _NINI,
    "FeFe": [1.9135e-1, -1.0796e0, -0.8928e-1, -5.8954e-2, -1.3872e0, 2.4790e0, 6.202, 5.055],
    "MoMo": _NINI, "MoNi": _NINI, "BeBe": _NINI,
}


class EamAlloyNN:
    """`eam/alloy` model with analytic potentials: E_i = F(rho_i) + 1/2 sum_j phi(r_ij)."""

    scope = "EAM"
    tag = "alloy"
    _kind = _lib.TA_MODEL_EAM_ALLOY

    def __init__(self, elements: Sequence[str], custom_potentials=None, hidden_sizes=None,
                 fixed_functions=None, minimize_properties=("energy", "forces"),
                 export_properties=("energy", "forces", "stress"), parameters=None,
                 activation=None):
        self._elements = sorted(list(elements))
        self._activation = activation or Defaults.activation
        self._fixed_functions = list(fixed_functions or [])
        self._minimize_properties = list(minimize_properties)
        self._export_properties = list(export_properties)
        kbody = get_kbody_terms(self._elements, angular=False)[1]
        unique = []
        for el in self._elements:
            for term in kbody[el]:
                a, b = get_elements_from_kbody_term(term)
                ab = "".join(sorted([a, b]))
                if ab not in unique:
                    unique.append(ab)
        self._unique_kbody_terms = unique
        self._hidden_sizes = self._get_hidden_sizes(hidden_sizes)
        self.weights = {}
        self._custom_potentials = custom_potentials
        self._potentials = self._setup_potentials(custom_potentials)
        # overrides of the default constants: {"Ni": {"r_eq": ...}, "NiNi": {"d1": ...}}
        self._parameters = {k: dict(v) for k, v in (parameters or {}).items()}
        self._transformer = None
        self.precision = "high"

    # ------------------------------------------------------------------
    def _extra_functions(self):
        return ()

    def _pair_functions(self):
        return ("phi",) + tuple(self._extra_functions())

    def _get_hidden_sizes(self, hidden_sizes):
        """Nested dict {section: {function: [sizes]}} from an int, a list or a nested dict
        (alloy.py:39-90, adp.py:51-105)."""
        default = list(Defaults.hidden_sizes)
        res = {el: {"rho": list(default), "embed": list(default)} for el in self._elements}
        res.update({t: {fn: list(default) for fn in self._pair_functions()}
                    for t in self._unique_kbody_terms})
        if hidden_sizes is None:
            return res
        for sec in res:
            if isinstance(hidden_sizes, dict):
                for fn, v in hidden_sizes.get(sec, {}).items():
                    if fn in res[sec]:
                        res[sec][fn] = [int(x) for x in np.atleast_1d(v)]
            else:
                for fn in res[sec]:
                    res[sec][fn] = [int(x) for x in np.atleast_1d(hidden_sizes)]
        return res

    def _setup_potentials(self, custom):
        pots = {}
        custom = {} if custom is None else custom
        if isinstance(custom, str):
            for el in self._elements:
                pots[el] = {"rho": custom, "embed": custom}
            for term in self._unique_kbody_terms:
                pots[term] = {"phi": custom}
                for fn in self._extra_functions():
                    pots[term][fn] = custom
        else:
            for el in self._elements:
                pots[el] = {"rho": custom.get(el, {}).get("rho", "nn"),
                            "embed": custom.get(el, {}).get("embed", "nn")}
            for term in self._unique_kbody_terms:
                sec = custom.get(term, {})
                pots[term] = {"phi": sec.get("phi", "nn")}
                for fn in self._extra_functions():
                    pots[term][fn] = sec.get(fn, "nn")
        family = set()
        for key, sec in pots.items():
            for fn, name in sec.items():
                eam_ok = ZJW04_FAMILY + OTHER_POTENTIALS
                ok = {"rho": eam_ok, "embed": eam_ok, "phi": eam_ok,
                      "dipole": ("mishinh",), "quadrupole": ("mishinh",)}[fn]
                if str(name).lower() == "nn" or str(name).startswith("spline@"):
                    continue
                if str(name).lower() not in ok:
                    raise ValueError(f"potential '{name}' for {key}/{fn} is not implemented by "
                                     f"tensoralloy_amd (available: {ok})")
                if fn in ("rho", "embed", "phi") and str(name).lower() in ZJW04_FAMILY:
                    family.add(str(name).lower())
        if len(family) > 1:
            # every reference potential object carries its own constants per element; mixing
            # them inside one model needs per-function parameter sets, which the kernels lack
            raise ValueError(f"one Zjw04 variant per model, got {sorted(family)}")
        self._family = family.pop() if family else "zjw04"
        # the analytic rho, embed and same-element phi of one element share one constant block in
        # the kernels: they must come from one potential (Zjw04 family, sutton90 or Be/1); a
        # cross-element analytic phi exists only in the Zjw04 family (mixing rule / xcp constants)
        self._el_kind = {}
        for el in self._elements:
            names = {str(pots[el][fn]).lower() for fn in ("rho", "embed")} | {str(pots[el + el]["phi"]).lower()}
            kinds = {("zjw" if n in ZJW04_FAMILY else n) for n in names if n != "nn" and not n.startswith("spline@")}
            if len(kinds) > 1:
                raise ValueError(f"the analytic functions of {el} mix potentials {sorted(kinds)}; use one")
            self._el_kind[el] = kinds.pop() if kinds else "zjw"
        for term in self._unique_kbody_terms:
            a, b = get_elements_from_kbody_term(term)
            name = str(pots[term]["phi"]).lower()
            if a != b and name in OTHER_POTENTIALS:
                raise ValueError(f"{name} has no cross-element pair potential ({term})")
            if a != b and name in ZJW04_FAMILY and name != "zjw04xcp" and \
                    (self._el_kind[a] != "zjw" or self._el_kind[b] != "zjw"):
                raise ValueError(f"the Zjw04 mixing rule for {term} needs Zjw04 constants of both elements")
        return pots

    # -- reference-compatible surface ----------------------------------------------
    @property
    def elements(self):
        return self._elements

    @property
    def unique_kbody_terms(self):
        return self._unique_kbody_terms

    @property
    def potentials(self):
        return self._potentials

    @property
    def predict_properties(self):
        return self._export_properties

    @property
    def variational_energy(self):
        return "energy"

    @property
    def is_finite_temperature(self):
        return False

    @property
    def transformer(self):
        return self._transformer

    def attach_transformer(self, clf):
        if clf.angular:
            raise ValueError("EAM models need a radial-only transformer (angular=False)")
        self._transformer = clf

    @property
    def hidden_sizes(self):
        return self._hidden_sizes

    def is_nn(self, section: str, fn: str) -> bool:
        return str(self._potentials[section][fn]).lower() == "nn"

    def is_spline(self, section: str, fn: str) -> bool:
        return str(self._potentials[section][fn]).startswith("spline@")

    def _all_slots(self):
        slots = [(el, "rho") for el in self._elements] + [(el, "embed") for el in self._elements]
        n = len(self._elements)
        pairs = [self._elements[i] + self._elements[j] for i in range(n) for j in range(i, n)]
        for fn in self._pair_functions():
            slots += [(t, fn) for t in pairs]
        return slots

    def spline_table(self, section: str, fn: str):
        """The `io.Spline` of a `spline@<path>` function (files are read once per model)."""
        from . import io
        path = str(self._potentials[section][fn])[len("spline@"):]
        cache = self.__dict__.setdefault("_setfl_cache", {})
        if path not in cache:
            cache[path] = io.read_adp_setfl(path) if self._extra_functions() else io.read_eam_alloy_setfl(path)
        fl = cache[path]
        if fn in ("rho", "embed"):
            group = getattr(fl, fn)
            if section not in group:
                raise ValueError(f"{path} has no {fn} table for {section}")
            return group[section]
        a, b = get_elements_from_kbody_term(section)
        try:
            return fl.pair(fn, a, b)
        except KeyError as exc:
            raise ValueError(f"{path}: {exc.args[0]}") from None

    @classmethod
    def from_setfl(cls, path: str, elements=None, **kwargs):
        """A model whose every function is tabulated in one setfl / adp file."""
        from . import io
        path = os.path.abspath(str(path))
        fl = io.read_adp_setfl(path) if cls.tag == "adp" else io.read_eam_alloy_setfl(path)
        elements = sorted(elements or fl.elements)
        name = "spline@" + path
        pots = {el: {"rho": name, "embed": name} for el in elements}
        for i, a in enumerate(elements):
            for b in elements[i:]:
                pots[a + b] = {"phi": name}
                if cls.tag == "adp":
                    pots[a + b].update(dipole=name, quadrupole=name)
        nn = cls(elements, custom_potentials=pots, **kwargs)
        nn.__dict__.setdefault("_setfl_cache", {})[path] = fl
        return nn

    def nn_functions(self):
        """(section, function) of every "nn" function in ABI slot order; analytic ones as None."""
        return [(sec, fn) if self.is_nn(sec, fn) else None for sec, fn in self._all_slots()]

    def initialize(self, seed=Defaults.seed, bias_scale=0.0):
        """He-normal kernels truncated at 2 sigma, zero (or small random) hidden biases, no output
        bias (nn/init_ops.py:20-30; `convolution1x1(..., output_bias=False)`, eam.py:184-190)."""
        rng = np.random.RandomState(seed)
        self.weights = {}
        for slot in self.nn_functions():
            if slot is None:
                continue
            sec, fn = slot
            sizes = [1] + list(self._hidden_sizes[sec][fn]) + [1]
            layers = []
            for l in range(len(sizes) - 1):
                fan_in, fan_out = sizes[l], sizes[l + 1]
                sigma = np.sqrt(2.0 / fan_in)
                w = rng.normal(0.0, sigma, size=(fan_in, fan_out))
                bad = np.abs(w) > 2 * sigma
                while bad.any():
                    w[bad] = rng.normal(0.0, sigma, size=int(bad.sum()))
                    bad = np.abs(w) > 2 * sigma
                if l == len(sizes) - 2:
                    b = None
                else:
                    b = bias_scale * rng.normal(size=fan_out) if bias_scale else np.zeros(fan_out)
                layers.append((w, b))
            self.weights.setdefault(sec, {})[fn] = layers

    def as_dict(self):
        return {"class": self.__class__.__name__, "elements": self._elements,
                "activation": self._activation,
                "custom_potentials": self._potentials, "hidden_sizes": self._hidden_sizes,
                "fixed_functions": self._fixed_functions,
                "minimize_properties": self._minimize_properties,
                "export_properties": self._export_properties,
                "parameters": self._parameters}

    # -- parameters ---------------------------------------------------------------------
    @property
    def family(self) -> str:
        """'zjw04', 'zjw04xc', 'zjw04uxc' or 'zjw04xcp'."""
        return self._family

    def other_parameters(self, el: str) -> Dict[str, float]:
        """Constants of an element whose analytic functions are sutton90 or Be/1."""
        kind = self._el_kind[el]
        table, keys = _OTHER_TABLES[kind]
        p = dict(table.get(el, {}))
        p.update({k: v for k, v in self._parameters.get(el, {}).items() if k in keys})
        p.update({k: v for k, v in self._parameters.get(el + el, {}).items() if k in keys})
        missing = [k for k in keys if k not in p]
        if missing:
            raise ValueError(f"{kind} has no constants for {el} (missing {missing}); pass parameters=")
        return p

    def element_parameters(self, el: str) -> Dict[str, float]:
        # defaults per variant: zjw04.py:19-152; Zjw04xc adds Be := Mo (:436-438);
        # Zjw04xcp refits Ni and Mo (:608-633)
        table = dict(ZJW04_DEFAULTS)
        if self._family != "zjw04":
            table["Be"] = table["Mo"]
        if el not in table:
            if self._element_is_all_nn(el):  # no analytic function reads these constants
                p = dict.fromkeys(ZJW04_KEYS, 0.0)
                p.update(r_eq=1.0, rho_e=1.0, rho_s=1.0)
                return p
            raise ValueError(f"{self._family} has no parameters for element {el}")
        p = dict(zip(ZJW04_KEYS, table[el]))
        if self._family == "zjw04xcp" and el in ZJW04XCP_ELEMENTS:
            p.update(ZJW04XCP_ELEMENTS[el])
        p.update(self._parameters.get(el, {}))
        return p

    def _element_is_all_nn(self, el: str) -> bool:
        """No analytic function reads this element's Zjw04 constants."""
        def free(sec, fn):
            return self.is_nn(sec, fn) or self.is_spline(sec, fn)
        if not (free(el, "rho") and free(el, "embed")):
            return False
        for t in self._unique_kbody_terms:
            if el in get_elements_from_kbody_term(t) and not free(t, "phi"):
                return False
        return True

    def phi_parameters(self, a: str, b: str):
        """Constants of a cross-element phi of Zjw04xcp, else None (Zjw04 mixing rule)."""
        if a == b or self._family != "zjw04xcp":
            return None
        key = "".join(sorted([a, b]))
        if self.is_nn(key, "phi") or self.is_spline(key, "phi"):
            return None
        p = dict(ZJW04XCP_PAIRS.get(key, {}))
        p.update({k: v for k, v in self._parameters.get(key, {}).items() if k in PHI_KEYS})
        missing = [k for k in PHI_KEYS if k not in p]
        if missing:
            raise ValueError(f"zjw04xcp has no phi constants for {key} (missing {missing}); "
                             f"pass parameters={{'{key}': {{...}}}}")
        return p

    def pair_parameters(self, term: str):
        return None

    def flat_parameters(self) -> np.ndarray:
        out = []
        embed_kind = 0.0 if self._family == "zjw04" else 1.0
        for el in self._elements:
            kind = self._el_kind[el]
            if kind == "zjw":
                p = self.element_parameters(el)
                out.extend(p[k] for k in ZJW04_KEYS)
            else:
                q = self.other_parameters(el)
                vals = [q[k] for k in _OTHER_TABLES[kind][1]]
                out.extend(vals + [0.0] * (20 - len(vals)))
            out.append(embed_kind)
            out.append(float(EL_KIND[kind]))
        n = len(self._elements)
        for i in range(n):
            for j in range(i, n):
                p = self.phi_parameters(self._elements[i], self._elements[j])
                if p is None:
                    out.extend([0.0] * 8)
                else:
                    out.append(1.0)
                    out.extend(p[k] for k in PHI_KEYS)
        return np.array(out, dtype=np.float64)

    # -- the constants as trainable parameters (reference potentials/potentials.py:129-163) ------
    def constant_names(self):
        """(section, key) of every slot of the C ABI's constants vector (`ta_get_constants`:
        20 per element, then 7 per sorted pair type); None where the slot is unused."""
        names = []
        for el in self._elements:
            kind = self._el_kind[el]
            keys = ZJW04_KEYS if kind == "zjw" else _OTHER_TABLES[kind][1]
            names.extend([(el, k) for k in keys] + [None] * (20 - len(keys)))
        n = len(self._elements)
        for i in range(n):
            for j in range(i, n):
                a, b = self._elements[i], self._elements[j]
                if self.phi_parameters(a, b) is None:
                    names.extend([None] * 7)
                else:
                    names.extend([("".join(sorted([a, b])), k) for k in PHI_KEYS])
        return names

    def constants(self) -> np.ndarray:
        out = []
        for name in EamAlloyNN.constant_names(self):
            if name is None:
                out.append(0.0)
                continue
            sec, key = name
            if sec in self._elements:
                kind = self._el_kind[sec]
                p = self.element_parameters(sec) if kind == "zjw" else self.other_parameters(sec)
            else:
                a, b = get_elements_from_kbody_term(sec)
                p = self.phi_parameters(a, b)
            out.append(float(p[key]))
        return np.array(out, dtype=np.float64)

    def set_constants(self, flat):
        """Take trained values back into the model (they are written by `export`)."""
        flat = np.asarray(flat, dtype=np.float64).ravel()
        names = self.constant_names()
        if len(flat) != len(names):
            raise ValueError(f"expected {len(names)} constants")
        for name, v in zip(names, flat):
            if name is None:
                continue
            sec, key = name
            self._parameters.setdefault(sec, {})[key] = float(v)
            if sec in self._elements and key in self._parameters.get(sec + sec, {}):
                self._parameters[sec + sec][key] = float(v)   # `other_parameters` reads 'AA' last

    def constant_mask(self, fixed=None) -> np.ndarray:
        """1 for trainable slots. `fixed`: {section: [names]} as the reference's potentials take it
        (`_is_trainable`, potentials.py:119-127); unused slots are never trained."""
        fixed = fixed or {}
        return np.array([0.0 if (n is None or n[1] in fixed.get(n[0], ())) else 1.0
                         for n in self.constant_names()])

    def to_desc(self):
        clf = self._transformer
        if clf is None:
            raise ValueError("A descriptor transformer must be attached.")
        keep = []
        params = np.ascontiguousarray(self.flat_parameters())
        keep.append(params)
        desc = _lib.ModelDesc()
        desc.kind = self._kind
        desc.n_elements = len(self._elements)
        desc.rcut = float(clf.rcut)
        desc.acut = float(clf.rcut)
        desc.angular = 0
        desc.n_eam_params = len(params)
        desc.eam_params = _lib.as_dp(params)
        desc.eps = 1e-8 if self.precision == "medium" else 1e-14  # precision.py:113-114
        all_slots = self._all_slots()
        if any(self.is_spline(sec, fn) for sec, fn in all_slots):
            from .io import natural_spline_coefficients
            tn, tdx, coef = [], [], []
            for sec, fn in all_slots:
                if not self.is_spline(sec, fn):
                    tn.append(0)
                    tdx.append(0.0)
                    continue
                sp = self.spline_table(sec, fn)
                tn.append(len(sp.x))
                tdx.append(float(sp.x[1] - sp.x[0]))
                coef.append(natural_spline_coefficients(sp.x, sp.y).ravel())
            tn = np.ascontiguousarray(tn, dtype=np.int32)
            tdx = np.ascontiguousarray(tdx, dtype=np.float64)
            coef = np.ascontiguousarray(np.concatenate(coef))
            keep += [tn, tdx, coef]
            desc.n_eam_nets = len(all_slots)
            desc.eam_table_n = _lib.as_ip(tn)
            desc.eam_table_dx = _lib.as_dp(tdx)
            desc.eam_table_coef = _lib.as_dp(coef)
        slots = self.nn_functions()
        if any(s is not None for s in slots):
            n_layers, sizes, flat = [], [], []
            for slot in slots:
                if slot is None:
                    n_layers.append(0)
                    continue
                sec, fn = slot
                try:
                    layers = self.weights[sec][fn]
                except KeyError:
                    raise ValueError(f"no weights for the nn function {sec}/{fn}: call initialize() "
                                     f"or set .weights['{sec}']['{fn}']") from None
                n_layers.append(len(layers))
                sz = [1]
                for w, b in layers:
                    w = np.asarray(w, dtype=np.float64)
                    if w.ndim != 2 or w.shape[0] != sz[-1]:
                        raise ValueError(f"{sec}/{fn}: weight shape {w.shape} does not chain from {sz[-1]}")
                    sz.append(w.shape[1])
                    flat.append(w.ravel())
                    flat.append(np.zeros(w.shape[1]) if b is None
                                else np.asarray(b, dtype=np.float64).ravel())
                if sz[-1] != 1:
                    raise ValueError(f"{sec}/{fn}: the output layer must have one unit")
                sizes.extend(sz)
            il = np.ascontiguousarray(n_layers, dtype=np.int32)
            isz = np.ascontiguousarray(sizes, dtype=np.int32)
            fw = np.ascontiguousarray(np.concatenate(flat))
            keep += [il, isz, fw]
            desc.n_eam_nets = len(slots)
            desc.n_layers = _lib.as_ip(il)
            desc.layer_sizes = _lib.as_ip(isz)
            desc.weights = _lib.as_dp(fw)
            desc.activation = _lib.TA_ACT[self._activation.lower()]
        return desc, keep

    def export(self, output_graph_path: str, **_ignored):
        from .model import API_VERSION, _model_stem
        if self._transformer is None:
            raise ValueError("A transformer must be attached before exporting to a pb file.")
        stem = _model_stem(output_graph_path)
        props = {"energy": "Output/Energy/energy:0", "energy/atom": "Output/Energy/atomic:0"}
        want = set(self._export_properties)
        if want & {"forces", "stress", "total_pressure"}:
            props["forces"] = "Output/Forces/forces:0"
        if "stress" in want:
            props["stress"] = "Output/Stress/Voigt/stress:0"
            props["virial"] = "Output/Stress/Full/virial:0"
            props["total_pressure"] = "Output/Stress/pressure/GPa:0"
        # second derivatives: central differences of the analytic forces / virial (calculator.py)
        if "hessian" in want:
            props["hessian"] = "Output/Hessian/hessian:0"
        if "elastic" in want:
            props["elastic"] = "Output/Elastic/Cijkl/elastic:0"
        meta = {
            "format": "tensoralloy_amd/1",
            "Transformer/params": self._transformer.as_dict(),
            "Metadata/timestamp": str(datetime.today()),
            "Metadata/precision": self.precision,
            "Metadata/variational_energy": "energy",
            "Metadata/is_finite_temperature": 0,
            "Metadata/api": API_VERSION,
            "Metadata/ops": props,
            "nn": self.as_dict(),
            "weights": None,
        }
        data = {}
        for slot in self.nn_functions():
            if slot is None:
                continue
            sec, fn = slot
            for j, (w, b) in enumerate(self.weights[sec][fn]):
                data[f"{sec}/{fn}/weights_{j}"] = np.asarray(w, dtype=np.float64)
                if b is not None:
                    data[f"{sec}/{fn}/biases_{j}"] = np.asarray(b, dtype=np.float64)
        if data:
            meta["weights"] = os.path.basename(stem) + ".npz"
            np.savez(stem + ".npz", **data)
        with open(stem + ".json", "w") as fp:
            json.dump(meta, fp, indent=1)
        return stem + ".json"


    # -- LAMMPS tables ------------------------------------------------------------------
    def export_to_setfl(self, setfl: str, nr: int, dr: float, nrho: int, drho: float,
                        lattice_constants=None, lattice_types=None, device: int = 0, **_ignored):
        """
        Write a LAMMPS `eam/alloy` (setfl) -- or, for `AdpNN`, `adp` -- potential file, as the
        reference's `EamAlloyNN.export_to_setfl` (nn/eam/alloy.py:198-381) / `AdpNN.export_to_setfl`
        do: rho(r) and r*phi(r) on r = k dr (k < nr), F(rho) on rho = k drho (k < nrho), then
        u(r) and w(r) for ADP. The tables are evaluated on the GPU by the functions the energy
        kernels use (`ta_eam_tabulate`); plotting arguments of the reference are ignored.
        """
        from .atoms import atomic_masses, atomic_numbers
        from .engine import Engine
        if self._transformer is None:
            raise ValueError("A transformer must be attached before exporting")
        r = np.arange(nr, dtype=np.float64) * float(dr)
        rho = np.arange(nrho, dtype=np.float64) * float(drho)
        with Engine(self, device=device) as eng:
            t = eng.eam_tabulate(r, rho)
        lattice_constants = lattice_constants or {}
        lattice_types = lattice_types or {}
        els = self._elements
        n = len(els)

        def block(values):
            vals = ["%24.16e" % v for v in values]
            return "\n".join(" ".join(vals[k:k + 5]) for k in range(0, len(vals), 5)) + "\n"

        with open(setfl, "w") as fp:
            fp.write(f"Date: {datetime.today()} tensoralloy_amd\n")
            fp.write("LAMMPS setfl format\n")
            fp.write(f"Conversion by tensoralloy_amd {self.__class__.__name__} ({self._family})\n")
            fp.write(f"{n} " + " ".join(els) + "\n")
            fp.write(f"{nrho} {float(drho)!r} {nr} {float(dr)!r} {float(self._transformer.rcut)!r}\n")
            for k, el in enumerate(els):
                z = atomic_numbers[el]
                fp.write(f"{z} {atomic_masses[z]!r} {float(lattice_constants.get(el, 0.0))!r} "
                         f"{lattice_types.get(el, 'fcc')}\n")
                fp.write(block(t["embed"][k]))
                fp.write(block(t["rho"][k]))
            index = {name: k for k, name in enumerate(t["pairs"])}
            order = [(i, j) for i in range(n) for j in range(i + 1)]  # setfl: (1,1) (2,1) (2,2) ...
            for i, j in order:
                fp.write(block(t["phi"][index[els[j] + els[i]]] * r))
            if "u" in t:
                for key in ("u", "w"):
                    for i, j in order:
                        fp.write(block(t[key][index[els[j] + els[i]]]))
        return setfl


class AdpNN(EamAlloyNN):
    """`eam/adp`: EAM plus Mishin's dipole and quadrupole terms (nn/eam/adp.py)."""

    tag = "adp"
    _kind = _lib.TA_MODEL_EAM_ADP

    def _extra_functions(self):
        return ("dipole", "quadrupole")

    def pair_parameters(self, term: str):
        a, b = get_elements_from_kbody_term(term)
        key = "".join(sorted([a, b]))
        base = MISHINH_DEFAULTS.get(key)
        over = self._parameters.get(key, {})
        if base is None and not over:
            return None
        p = dict(zip(ADP_KEYS, base if base is not None else [0.0] * 6 + [1.0, 0.0]))
        p.update(over)
        return p

    def constant_names(self):
        """The EAM slots, then 8 per sorted pair type: the MishinH dipole / quadrupole constants."""
        names = list(EamAlloyNN.constant_names(self))
        n = len(self._elements)
        for i in range(n):
            for j in range(i, n):
                key = "".join(sorted([self._elements[i], self._elements[j]]))
                if self.pair_parameters(key) is None:
                    names.extend([None] * 8)
                else:
                    names.extend([(key, k) for k in ADP_KEYS])
        return names

    def constants(self) -> np.ndarray:
        n_eam = len(EamAlloyNN.constant_names(self))
        out = list(EamAlloyNN.constants(self))
        for name in self.constant_names()[n_eam:]:
            out.append(0.0 if name is None else float(self.pair_parameters(name[0])[name[1]]))
        return np.array(out, dtype=np.float64)

    def flat_parameters(self) -> np.ndarray:
        out = list(EamAlloyNN.flat_parameters(self))
        n = len(self._elements)
        for i in range(n):
            for j in range(i, n):
                term = self._elements[i] + self._elements[j]
                p = self.pair_parameters(term)
                free = all(self.is_nn(term, fn) or self.is_spline(term, fn) for fn in ("dipole", "quadrupole"))
                if p is None and free:
                    p = dict(zip(ADP_KEYS, [0.0] * 6 + [1.0, 0.0]))  # not read by nn functions
                if p is None:
                    raise ValueError(f"mishinh has no dipole/quadrupole parameters for {term}")
                out.extend(p[k] for k in ADP_KEYS)
        return np.array(out, dtype=np.float64)


def nn_from_dict(cls_name: str, cfg: dict, npz=None):
    """Rebuild a model from `as_dict()`; `npz` = the weight archive of its nn functions."""
    cfg = dict(cfg)
    cls = {"EamAlloyNN": EamAlloyNN, "AdpNN": AdpNN}[cls_name]
    nn = cls(**cfg)
    for slot in nn.nn_functions():
        if slot is None:
            continue
        sec, fn = slot
        if npz is None:
            raise ValueError(f"the model has nn functions ({sec}/{fn}) but no weight file")
        layers, j = [], 0
        while f"{sec}/{fn}/weights_{j}" in npz:
            w = np.array(npz[f"{sec}/{fn}/weights_{j}"], dtype=np.float64)
            b = np.array(npz[f"{sec}/{fn}/biases_{j}"], dtype=np.float64).ravel() \
                if f"{sec}/{fn}/biases_{j}" in npz else None
            layers.append((w, b))
            j += 1
        if not layers:
            raise ValueError(f"the weight file holds nothing for {sec}/{fn}")
        nn.weights.setdefault(sec, {})[fn] = layers
    return nn
