"""
Virtual-atom map: permutation between an `Atoms`' own order ("local") and the
element-sorted, 1-based "global symbol list" (GSL) order with a virtual atom at
row 0 and room for `max_occurs[element]` atoms per element.

Interface of reference tensoralloy/transformer/vap.py:18-197 (attribute and
method names, index conventions, error messages); the maps themselves are
computed here from one stable argsort by element instead of the reference's
per-atom counters, and `map_array` scatters / gathers with index arrays.
"""
from __future__ import annotations

from collections import Counter
from typing import List

import numpy as np


class VirtualAtomMap:
    REAL_ATOM_START = 1

    def __init__(self, max_occurs: Counter, symbols: List[str]):
        self._symbols = list(symbols)
        self._max_occurs = max_occurs
        elements = sorted(max_occurs.keys())
        capacity = np.array([max_occurs[e] for e in elements], dtype=np.int64)
        self._max_vap_natoms = int(capacity.sum()) + 1
        n = len(self._symbols)

        # rank of every atom among the atoms of its own element, in local order
        code_of = {e: k for k, e in enumerate(elements)}
        codes = np.fromiter((code_of[s] for s in self._symbols), dtype=np.int64, count=n)
        grouped = np.argsort(codes, kind="stable")
        present = np.bincount(codes, minlength=len(elements))
        group_begin = np.cumsum(present) - present
        rank = np.empty(n, dtype=np.int64)
        rank[grouped] = np.arange(n) - group_begin[codes[grouped]]
        # GSL row = 1 + (rows reserved for the elements before it) + rank
        block_begin = np.cumsum(capacity) - capacity
        self.local_to_gsl = self.REAL_ATOM_START + block_begin[codes] + rank

        self._mask = np.zeros(self._max_vap_natoms, dtype=bool)
        self._mask[self.local_to_gsl] = True
        # local order == GSL order (one element, or atoms already sorted by element, no padding):
        # the maps are a shift by the virtual row and callers may skip the gather / scatter
        n = len(self.local_to_gsl)
        self.is_identity = bool(self._max_vap_natoms == n + self.REAL_ATOM_START and
                                np.array_equal(self.local_to_gsl, np.arange(n) + self.REAL_ATOM_START))
        self._vap_symbols = ["X"] + [e for e, c in zip(elements, capacity) for _ in range(int(c))]
        # dictionary views with the reference's keys: 1-based local index -> GSL row, and back
        self.local_to_gsl_map = {0: 0}
        self.local_to_gsl_map.update(
            {i + self.REAL_ATOM_START: int(g) for i, g in enumerate(self.local_to_gsl)})
        self.gsl_to_local_map = {0: -1}
        self.gsl_to_local_map.update({int(g): i for i, g in enumerate(self.local_to_gsl)})

    @property
    def vap_symbols(self):
        return self._vap_symbols

    @property
    def symbols(self):
        return self._symbols

    @property
    def max_vap_natoms(self):
        return self._max_vap_natoms

    @property
    def max_occurs(self) -> Counter:
        return self._max_occurs

    @property
    def atom_masks(self) -> np.ndarray:
        return self._mask

    def map_array(self, array: np.ndarray, reverse=False):
        """local -> GSL (rows of absent atoms and the virtual atom are zero), or
        GSL -> local when `reverse`. Rank 2 `[atoms, c]` or rank 3 `[batch, atoms, c]`."""
        array = np.asarray(array)
        rank = array.ndim
        if rank not in (2, 3):
            raise ValueError("The rank should be 2 or 3")
        batched = array if rank == 3 else array[np.newaxis]
        if reverse:
            out = batched[:, self.local_to_gsl]
        else:
            want = (batched.shape[0], len(self._symbols), batched.shape[2])
            if batched.shape != want:
                raise ValueError(f"The shape should be {want}")
            out = np.zeros((want[0], self._max_vap_natoms, want[2]), dtype=array.dtype)
            out[:, self.local_to_gsl] = batched
        return out if rank == 3 else out[0]

    def map_positions(self, positions: np.ndarray, reverse=False):
        return self.map_array(positions, reverse=reverse)

    def map_forces(self, forces: np.ndarray, reverse=False):
        return self.map_array(forces, reverse=reverse)

    def reverse_map_hessian(self, hessian: np.ndarray, phonopy_format=False):
        """`[n_vap, 3, n_vap, 3]` in GSL order -> local order, as `[3n, 3n]` or, with
        `phonopy_format`, `[n, n, 3, 3]`."""
        hessian = np.asarray(hessian)
        if hessian.ndim != 4 or hessian.shape[1] != 3 or hessian.shape[3] != 3:
            raise ValueError("The input array should be a 4D matrix of shape [Np, 3, Np, 3]")
        rows = self.local_to_gsl
        local = hessian[np.ix_(rows, np.arange(3), rows, np.arange(3))]
        if phonopy_format:
            return np.ascontiguousarray(local.transpose(0, 2, 1, 3))
        return local.reshape(3 * len(rows), 3 * len(rows))
