"""
ORACLE (test infrastructure only) — neighbor list with the semantics of
`ase.neighborlist.neighbor_list('ijS', atoms, rc)`, the third-party routine
the reference calls at tensoralloy/transformer/universal.py:58 and
tensoralloy/neighbor.py:84 (ASE >= 3.21, requirements.txt:3; ASE is not
installed here, so its published behaviour is restated):

  * FULL list: every directed pair (i, j, S) with |R_j - R_i + S.h| < rc
    (strict), both directions present;
  * periodic images are distinct neighbours, including self-images (i == j,
    S != 0) when the cell is smaller than rc;
  * S is relative to the positions as given (atoms may sit outside the cell);
  * non-periodic axes never shift.

Pinned by the neighbour statistics the reference cached in
tensoralloy/data/datasets/snap-Ni.db (SURVEY §8c pin 9) and by
tests/test_neighbor.py:20-36 of the reference (qm7m: nij / nnl / nijk).
"""
import itertools

import numpy as np
from scipy.spatial import cKDTree


def _complete_cell(cell, pbc):
    """Replace zero lattice vectors of non-periodic axes by unit vectors
    orthogonal to the others (what `Atoms.get_cell(complete=True)` does)."""
    cell = np.array(cell, dtype=np.float64).reshape(3, 3)
    missing = [a for a in range(3) if not np.any(cell[a])]
    if len(missing) == 3:
        return np.eye(3)
    if len(missing) == 2:
        present = [a for a in range(3) if a not in missing][0]
        v = cell[present] / np.linalg.norm(cell[present])
        # any two vectors orthogonal to v
        trial = np.eye(3)[np.argmin(np.abs(v))]
        u = np.cross(v, trial)
        u /= np.linalg.norm(u)
        w = np.cross(v, u)
        cell[missing[0]], cell[missing[1]] = u, w
    elif len(missing) == 1:
        a = missing[0]
        others = [b for b in range(3) if b != a]
        n = np.cross(cell[others[0]], cell[others[1]])
        cell[a] = n / np.linalg.norm(n)
    return cell


def neighbor_list(positions, cell, pbc, rc):
    """
    Returns (i, j, S) int arrays sorted by (i, j, Sx, Sy, Sz).
    """
    R = np.asarray(positions, dtype=np.float64).reshape(-1, 3)
    pbc = np.asarray(pbc, dtype=bool).reshape(3)
    n = len(R)
    h = _complete_cell(cell, pbc)
    hinv = np.linalg.inv(h)
    frac = R @ hinv
    wrap = np.zeros((n, 3), dtype=np.int64)
    wrap[:, pbc] = np.floor(frac[:, pbc]).astype(np.int64)
    Rw = R - wrap @ h

    vol = abs(np.linalg.det(h))
    nmax = []
    for a in range(3):
        if not pbc[a]:
            nmax.append(0)
            continue
        b, c = [x for x in range(3) if x != a]
        height = vol / np.linalg.norm(np.cross(h[b], h[c]))
        nmax.append(int(np.floor(rc / height)) + 1)

    tree = cKDTree(Rw)
    out_i, out_j, out_s = [], [], []
    for s in itertools.product(*[range(-m, m + 1) for m in nmax]):
        s = np.array(s, dtype=np.int64)
        other = cKDTree(Rw + s @ h)
        m = tree.sparse_distance_matrix(other, rc * (1.0 + 1e-9),
                                        output_type="ndarray")
        if len(m) == 0:
            continue
        i = m["i"].astype(np.int64)
        j = m["j"].astype(np.int64)
        if not s.any():
            keep = i != j
            i, j = i[keep], j[keep]
        # shift relative to the positions as given
        S = s[None, :] - wrap[j] + wrap[i]
        D = R[j] - R[i] + S @ h
        d = np.sqrt(np.sum(D * D, axis=1))
        keep = d < rc
        out_i.append(i[keep])
        out_j.append(j[keep])
        out_s.append(S[keep])
    if not out_i:
        z = np.zeros(0, dtype=np.int64)
        return z, z.copy(), np.zeros((0, 3), dtype=np.int64)
    i = np.concatenate(out_i)
    j = np.concatenate(out_j)
    S = np.concatenate(out_s)
    order = np.lexsort((S[:, 2], S[:, 1], S[:, 0], j, i))
    return i[order], j[order], S[order]


def neighbor_sizes(i, species, n_species):
    """nij, nnl (max per (centre, neighbour species) count), nijk (symmetric)."""
    n = len(species)
    nij = len(i)
    counts = np.bincount(i, minlength=n)
    nijk = int(np.sum(counts * (counts - 1) // 2))
    return nij, counts, nijk
