"""
ORACLE (test infrastructure only) — the reference's DENSE-layout evaluation of
G2 / G4, restated in NumPy on the arrays that `UniversalTransformer.
get_descriptors` (= `build_graph`, reference transformer/universal.py:708-726)
returns: the symmetry functions are applied to the padded tensors
[n_terms, n_el, nnl_max(, ij2k_max)] with masks and summed over the slot axes,
exactly as nn/atomic/sf.py:79-182 does. Independent of the packed formulation
in oracle/sf.py; used to check the feed-dict wire format end to end.
"""
import numpy as np

from .sf import radial_params, angular_params


def _fc(r, rc, kind):
    x = np.minimum(r / rc, 1.0)
    if kind == "cosine":
        return 0.5 * (np.cos(np.pi * x) + 1.0)
    return 1.0 + 5.0 * x ** 6 - 6.0 * x ** 5


def descriptors_from_dense(universal, elements, rcut, acut, eta, omega, beta, gamma, zeta,
                           kind="cosine"):
    """{element: [n_el, D]} from the dict returned by `get_descriptors`."""
    n = len(elements)
    out = {}
    rad = radial_params(eta, omega)
    ang = angular_params(beta, gamma, zeta)
    for el in elements:
        dists, masks = universal["radial"][el]
        blocks = []
        for t in range(n):
            r = dists[0, t, :, :, 0]
            m = masks[t, :, :, 0]
            fc = _fc(r, rcut, kind)
            for e, w in rad:
                v = np.exp(-e * (r - w) ** 2 / rcut ** 2) * fc * m  # sf.py:101-108
                blocks.append(v.sum(axis=-1))
        if universal["angular"] is not None:
            dists, masks = universal["angular"][el]
            for t in range(n * (n + 1) // 2):
                rij, rik, rjk = dists[0, t], dists[4, t], dists[8, t]
                m = masks[t]
                lower = 2.0 * rij * rik
                theta = np.where(lower != 0, (rij ** 2 + rik ** 2 - rjk ** 2) /
                                 np.where(lower != 0, lower, 1.0), 0.0)  # divide_no_nan, sf.py:145-148
                z = (rij ** 2 + rik ** 2 + rjk ** 2) / acut ** 2
                fc = _fc(rij, acut, kind) * _fc(rik, acut, kind) * _fc(rjk, acut, kind)
                for b, g, zt in ang:
                    v = 2.0 ** (1.0 - zt) * (1.0 + g * theta) ** zt * np.exp(-b * z) * fc * m
                    blocks.append(v.sum(axis=(-1, -2)))
        out[el] = np.stack(blocks, axis=-1)
    return out
