"""
Virtual-atom map: permutation between an `Atoms`' own order ("local") and the
element-sorted, 1-based "global symbol list" order with a virtual atom at row 0.

Mirrors reference tensoralloy/transformer/vap.py:18-197 (same attribute and
method names, same index conventions) with vectorised NumPy instead of loops.
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
        self._max_vap_natoms = int(sum(max_occurs.values()) + 1)
        istart = VirtualAtomMap.REAL_ATOM_START
        elements = sorted(max_occurs.keys())
        offsets = np.concatenate(([0], np.cumsum([max_occurs[e] for e in elements])[:-1]))
        delta = Counter()
        index_map = {}
        mask = np.zeros(self._max_vap_natoms, dtype=bool)
        for i, symbol in enumerate(self._symbols):
            idx_new = int(offsets[elements.index(symbol)]) + delta[symbol] + istart
            index_map[i + istart] = idx_new
            delta[symbol] += 1
            mask[idx_new] = True
        reverse_map = {v: k - 1 for k, v in index_map.items()}
        index_map[0] = 0
        reverse_map[0] = -1
        self._mask = mask
        self.local_to_gsl_map = index_map
        self.gsl_to_local_map = reverse_map
        self._vap_symbols = ["X"]
        for element in elements:
            self._vap_symbols.extend([element] * self._max_occurs[element])
        # array forms for vectorised use
        n = len(self._symbols)
        self.local_to_gsl = np.array([index_map[i + istart] for i in range(n)], dtype=np.int64)
        self._gather_fwd = np.array(
            [reverse_map.get(i, -1) + istart for i in range(self._max_vap_natoms)],
            dtype=np.int64)

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
        """local -> GSL (a zero row is inserted for the virtual atom), or
        GSL -> local when `reverse`."""
        array = np.asarray(array)
        rank = np.ndim(array)
        if rank == 2:
            array = array[np.newaxis, ...]
        elif rank <= 1 or rank > 3:
            raise ValueError("The rank should be 2 or 3")
        if not reverse:
            if array.shape[1] == len(self._symbols):
                array = np.insert(array, 0, np.asarray(0, dtype=array.dtype), axis=1)
            else:
                shape = (array.shape[0], len(self._symbols), array.shape[2])
                raise ValueError(f"The shape should be {shape}")
            output = array[:, self._gather_fwd]
        else:
            output = array[:, self.local_to_gsl]
        if rank == 2:
            output = np.squeeze(output, axis=0)
        return output

    def map_positions(self, positions: np.ndarray, reverse=False):
        return self.map_array(positions, reverse=reverse)

    def map_forces(self, forces: np.ndarray, reverse=False):
        return self.map_array(forces, reverse=reverse)

    def reverse_map_hessian(self, hessian: np.ndarray, phonopy_format=False):
        rank = np.ndim(hessian)
        if rank != 4 or hessian.shape[1] != 3 or hessian.shape[3] != 3:
            raise ValueError("The input array should be a 4D matrix of shape [Np, 3, Np, 3]")
        idx = self.local_to_gsl
        h = np.asarray(hessian)[idx][:, :, idx, :]  # [n, 3, n, 3]
        n = len(idx)
        if phonopy_format:
            return np.transpose(h, (0, 2, 1, 3)).copy()
        return h.reshape(n * 3, n * 3)
