"""
Host-side helpers shared by the transformer mirror and the model loader.

These follow the *behaviour* of the reference helpers (same names, argument
meaning, return order) so code written against `tensoralloy.utils` keeps
working:

* `get_elements_from_kbody_term`  -> reference tensoralloy/utils.py:210-234
* `get_kbody_terms`               -> reference tensoralloy/utils.py:237-290
* `szudzik_pairing`               -> reference tensoralloy/utils.py:88-161
* `ModeKeys`, `Defaults`          -> reference tensoralloy/utils.py:322-341, :393-420
* `parameter_grid`                -> sklearn `ParameterGrid` ordering used by
                                     reference tensoralloy/nn/atomic/sf.py:47-51
                                     (keys sorted, LAST key varies fastest)
"""
from __future__ import annotations

import enum
import itertools
import re
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np

__all__ = [
    "get_elements_from_kbody_term", "get_kbody_terms", "szudzik_pairing",
    "ModeKeys", "Defaults", "parameter_grid", "GPa", "RANDOM_STATE",
]

# `ase.units.GPa` (ASE >= 3.21, CODATA 2014): 1 GPa in eV/Angstrom^3.
# Used by reference tensoralloy/nn/basic.py:394-408 and calculator.py:279-295.
GPa = 1.0 / 160.21766208

RANDOM_STATE = 611

_TERM_RE = re.compile(r"[A-Z][a-z]*")


def get_elements_from_kbody_term(kbody_term: str) -> List[str]:
    """'NiMoMo' -> ['Ni', 'Mo', 'Mo']: split before every upper-case letter."""
    return _TERM_RE.findall(kbody_term)


def get_kbody_terms(elements: Sequence[str], angular=False, symmetric=True
                    ) -> Tuple[List[str], Dict[str, List[str]], List[str]]:
    """
    Ordered k-body terms (k = 2, and 3 when `angular`).

    For centre A the order is [AA, AB (B != A, sorted) | A+sorted(B_j B_k) for
    j <= k] (or every ordered (j, k) when `symmetric` is False).
    """
    elements = sorted(set(elements))
    terms: Dict[str, List[str]] = {}
    for a in elements:
        terms[a] = [a + a] + [a + b for b in elements if b != a]
    if angular:
        n = len(elements)
        for a in elements:
            if symmetric:
                pairs = [(j, k) for j in range(n) for k in range(j, n)]
            else:
                pairs = [(j, k) for j in range(n) for k in range(n)]
            for j, k in pairs:
                if symmetric:
                    suffix = "".join(sorted([elements[j], elements[k]]))
                else:
                    suffix = elements[j] + elements[k]
                terms[a].append(a + suffix)
    all_terms = list(itertools.chain(*[terms[a] for a in elements]))
    return all_terms, terms, elements


def _fold(v):
    # non-negative integers: 0, 2, 4, ... ; negative: 1, 3, 5, ...
    v = np.asarray(v, dtype=np.int64)
    return np.where(v >= 0, 2 * v, -2 * v - 1)


def _szudzik2(x, y):
    xx, yy = _fold(x), _fold(y)
    return np.where(xx >= yy, xx * xx + xx + yy, yy * yy + xx)


def szudzik_pairing(x, *args):
    """
    Szudzik pairing of signed integers, folded left to right:
    `pair(pair(x, a0), a1) ...`. Accepts scalars or equal-length 1D arrays; a
    2D array is paired column by column.
    """
    x = np.asarray(x)
    if x.ndim == 2 and not args:
        cols = [x[:, c] for c in range(x.shape[1])]
        x, args = cols[0], cols[1:]
    z = np.asarray(x, dtype=np.int64)
    for y in args:
        z = _szudzik2(z, y)
    if z.ndim == 0:
        return int(z)
    return z


class ModeKeys(enum.Enum):
    """Running modes of the reference (tensoralloy/utils.py:322-341)."""
    TRAIN = "train"
    EVAL = "eval"
    PREDICT = "infer"
    NATIVE = "native"


class Defaults:
    """Default hyper-parameters of the reference (tensoralloy/utils.py:393-420)."""
    rc = 6.0
    k_max = 2
    eta = np.array([0.05, 4.0, 20.0, 80.0])
    omega = np.array([0.0])
    beta = np.array([0.005])
    gamma = np.array([1.0, -1.0])
    zeta = np.array([1.0, 4.0])
    cutoff_function = "cosine"
    seed = RANDOM_STATE
    activation = "softplus"
    hidden_sizes = [64, 32]


def parameter_grid(**axes: Iterable[float]) -> List[Dict[str, float]]:
    """
    All combinations of the given axes in sklearn `ParameterGrid` order: keys
    sorted alphabetically, last key varying fastest.
    """
    keys = sorted(axes)
    values = [list(np.asarray(axes[k], dtype=float).ravel()) for k in keys]
    return [dict(zip(keys, combo)) for combo in itertools.product(*values)]
