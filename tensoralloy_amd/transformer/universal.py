"""
`UniversalTransformer`: host-side mirror of reference
tensoralloy/transformer/universal.py:236-908 (PREDICT path) and
tensoralloy/transformer/base.py:135-226.

The reference class has two halves: (1) host code that turns an `Atoms` into
index maps (neighbour list -> `v2g_map`, `ilist`, `jlist`, `n1`, ... — the
feed dict), and (2) TensorFlow graph code that scatters distances into dense
padded tensors. Here (1) is reproduced (vectorised, same keys / dtypes / slot
semantics) so existing callers of `get_np_feed_dict` keep working, while (2) is
replaced by the HIP library, which consumes the packed neighbour list directly
and never builds the dense tensors. `get_descriptors(features)` is kept for
parity/debugging and returns the same dense arrays as NumPy.

Batch variants (`BatchUniversalTransformer`, TFRecord encode/decode,
universal.py:921-1388) belong to the training data path and are out of scope.
"""
from __future__ import annotations

from collections import Counter
from typing import Dict, List

import numpy as np

from .. import _lib
from ..atoms import chemical_symbols
from ..utils import get_elements_from_kbody_term, get_kbody_terms
from .metadata import AngularMetadata, RadialMetadata
from .vap import VirtualAtomMap


class UniversalTransformer:
    """The universal transformer for all models."""

    def __init__(self, elements: List[str], rcut, acut=None, angular=False, periodic=True,
                 symmetric=True, use_computed_dists=True):
        for element in elements:
            if element not in chemical_symbols:
                raise ValueError(f"{element} is not a valid chemical symbol!")
        if angular and acut is None:
            acut = rcut
        all_kbody_terms, kbody_terms_for_element, elements = get_kbody_terms(
            elements, angular=angular, symmetric=symmetric)
        max_nr_terms = max_na_terms = 0
        for element in elements:
            terms = kbody_terms_for_element[element]
            max_na_terms = max(max_na_terms, len(
                [x for x in terms if len(get_elements_from_kbody_term(x)) == 3]))
            max_nr_terms = max(max_nr_terms, len(
                [x for x in terms if len(get_elements_from_kbody_term(x)) == 2]))
        self._all_kbody_terms = all_kbody_terms
        self._kbody_terms_for_element = kbody_terms_for_element
        self._max_na_terms = max_na_terms
        self._max_nr_terms = max_nr_terms
        self._rcut = rcut
        self._acut = acut
        self._elements = elements
        self._n_elements = len(elements)
        self._periodic = periodic
        self._angular = angular
        self._symmetric = symmetric
        self._use_computed_dists = use_computed_dists
        self._vap_transformers: Dict[str, VirtualAtomMap] = {}
        self._vap_by_numbers: Dict[bytes, VirtualAtomMap] = {}
        self._z_lookup = None

    # ---- reference-compatible properties ----------------------------------
    def as_dict(self) -> Dict:
        return {"class": self.__class__.__name__, "elements": self._elements,
                "rcut": self._rcut, "acut": self._acut, "angular": self._angular,
                "periodic": self._periodic, "symmetric": self._symmetric,
                "use_computed_dists": self._use_computed_dists}

    @property
    def descriptor(self):
        return "universal"

    @property
    def rc(self):
        return self._rcut

    @property
    def rcut(self) -> float:
        return self._rcut

    @property
    def acut(self) -> float:
        return self._acut

    @property
    def elements(self) -> List[str]:
        return self._elements

    @property
    def n_elements(self) -> int:
        return self._n_elements

    @property
    def periodic(self):
        return self._periodic

    @property
    def angular(self):
        return self._angular

    @property
    def symmetric(self):
        return self._symmetric

    @property
    def use_computed_dists(self):
        return self._use_computed_dists

    @use_computed_dists.setter
    def use_computed_dists(self, flag: bool):
        self._use_computed_dists = flag

    @property
    def all_kbody_terms(self):
        return self._all_kbody_terms

    @property
    def kbody_terms_for_element(self) -> Dict[str, List[str]]:
        return self._kbody_terms_for_element

    @property
    def max_occurs(self):
        return None  # only defined for batch transformers in the reference

    # ---- VAP --------------------------------------------------------------
    def get_vap_transformer(self, atoms) -> VirtualAtomMap:
        """One `VirtualAtomMap` per reduced formula (base.py:199-226)."""
        # same atomic numbers in the same order <=> same reduced formula: skip rebuilding
        # the formula string (1 ms for 4000 atoms) on every MD step
        numbers = getattr(atoms, "numbers", None)
        key = numbers.tobytes() if isinstance(numbers, np.ndarray) else None
        if key is not None and key in self._vap_by_numbers:
            return self._vap_by_numbers[key]
        formula = atoms.get_chemical_formula(mode="reduce")
        if formula not in self._vap_transformers:
            symbols = atoms.get_chemical_symbols()
            max_occurs = Counter()
            counter = Counter(symbols)
            for element in self._elements:
                max_occurs[element] = max(1, counter[element])
            self._vap_transformers[formula] = VirtualAtomMap(max_occurs, symbols)
        if key is not None:
            if len(self._vap_by_numbers) > 64:
                self._vap_by_numbers.clear()
            self._vap_by_numbers[key] = self._vap_transformers[formula]
        return self._vap_transformers[formula]

    # ---- species / neighbour helpers ----------------------------------------
    def species_indices(self, atoms) -> np.ndarray:
        """Index of every atom's element in the sorted element list."""
        numbers = getattr(atoms, "numbers", None)
        if isinstance(numbers, np.ndarray):
            if self._z_lookup is None:
                from ..atoms import chemical_symbols
                table = np.full(len(chemical_symbols), -1, dtype=np.int32)
                for k, e in enumerate(self._elements):
                    table[chemical_symbols.index(e)] = k
                self._z_lookup = table
            idx = self._z_lookup[numbers]
            if len(idx) and idx.min() < 0:
                from ..atoms import chemical_symbols
                bad = chemical_symbols[int(numbers[np.argmin(idx)])]
                raise ValueError(f"element {bad} is not supported by this model")
            return idx
        lookup = {e: k for k, e in enumerate(self._elements)}
        try:
            return np.array([lookup[s] for s in atoms.get_chemical_symbols()], dtype=np.int32)
        except KeyError as exc:
            raise ValueError(f"element {exc.args[0]} is not supported by this model") from None

    def _neighbors(self, atoms, rc):
        species = self.species_indices(atoms)
        cell = np.asarray(atoms.get_cell(complete=True), dtype=np.float64)
        pbc = np.asarray(atoms.pbc, dtype=bool) if self._periodic else np.zeros(3, bool)
        i, j, S, _ = _lib.neighbor_list(species, atoms.positions, cell, pbc,
                                        self._n_elements, rc)
        return species, cell, i, j, S

    @staticmethod
    def _radial_term(si, sj):
        # index in [AA, AB (B != A sorted)]  (utils.py:265-273)
        return np.where(sj == si, 0, np.where(sj < si, sj + 1, sj))

    def _angular_term(self, sj, sk):
        n = self._n_elements
        a, b = np.minimum(sj, sk), np.maximum(sj, sk)
        return a * n - (a * (a - 1)) // 2 + (b - a)

    # ---- metadata (feed-dict index maps) ------------------------------------
    def _radial_metadata(self, atoms, vap, rc, dtype):
        species, cell, i, j, S = self._neighbors(atoms, rc)
        P = len(i)
        t = self._radial_term(species[i], species[j]).astype(np.int32)
        l2g = vap.local_to_gsl
        ig = l2g[i].astype(np.int32)
        jg = l2g[j].astype(np.int32)
        # slot = running count of (centre, term) in list order (universal.py:90-99)
        key = i.astype(np.int64) * (self._n_elements + 1) + t
        change = np.ones(P, dtype=bool)
        if P:
            change[1:] = key[1:] != key[:-1]
        # list is sorted by (centre, neighbour species): groups are contiguous
        first = np.maximum.accumulate(np.where(change, np.arange(P), 0)) if P else np.zeros(0, int)
        inc = (np.arange(P) - first).astype(np.int32)
        v2g = np.zeros((P, 5), dtype=np.int32)
        v2g[:, 0], v2g[:, 1], v2g[:, 2], v2g[:, 4] = t, ig, inc, 1
        D = atoms.positions[j] - atoms.positions[i] + S.astype(np.float64) @ cell
        r = np.sqrt(np.sum(D * D, axis=1))
        rij = np.concatenate((r[:, None], D), axis=1).T.astype(dtype)
        meta = RadialMetadata(v2g_map=v2g, ilist=ig, jlist=jg, n1=S.astype(dtype), rij=rij)
        return meta, dict(i=i, j=j, S=S, inc=inc, species=species)

    def _angular_metadata(self, atoms, vap, radial, aux, dtype):
        if not self._symmetric:
            raise ValueError("symmetric=False angular terms are not implemented by tensoralloy_amd")
        i, j, S, inc, species = aux["i"], aux["j"], aux["S"], aux["inc"], aux["species"]
        N = len(species)
        starts = np.searchsorted(i, np.arange(N + 1))
        ta, tb = [], []
        for c in range(N):
            n = starts[c + 1] - starts[c]
            if n < 2:
                continue
            a, b = np.triu_indices(n, k=1)
            ta.append(a + starts[c])
            tb.append(b + starts[c])
        ta = np.concatenate(ta) if ta else np.zeros(0, dtype=np.int64)
        tb = np.concatenate(tb) if tb else np.zeros(0, dtype=np.int64)
        T = len(ta)
        sj, sk = species[j[ta]], species[j[tb]]
        term = self._angular_term(sj, sk).astype(np.int32)
        # key pair: j when symbol_j < symbol_k else k (universal.py:196-202)
        keyp = np.where(sj < sk, ta, tb)
        # running count per (centre, term, key pair) in enumeration order
        gid = keyp.astype(np.int64) * self._max_na_terms + term
        order = np.argsort(gid, kind="stable")
        sg = gid[order]
        first = np.ones(T, dtype=bool)
        if T:
            first[1:] = sg[1:] != sg[:-1]
        start = np.maximum.accumulate(np.where(first, np.arange(T), 0)) if T else np.zeros(0, int)
        cnt = np.empty(T, dtype=np.int32)
        cnt[order] = (np.arange(T) - start).astype(np.int32)
        v2g = np.zeros((T, 5), dtype=np.int32)
        v2g[:, 0] = term
        v2g[:, 1] = radial.ilist[ta]
        v2g[:, 2] = inc[keyp]
        v2g[:, 3] = cnt
        v2g[:, 4] = 1
        n1 = S[ta].astype(dtype)
        n2 = S[tb].astype(dtype)
        rijk = np.zeros((12, T), dtype=dtype)
        rijk[0:4] = radial.rij[:, ta]
        rijk[4:8] = radial.rij[:, tb]
        djk = rijk[5:8] - rijk[1:4]
        rijk[8] = np.sqrt(np.sum(djk * djk, axis=0))
        rijk[9:12] = djk
        return AngularMetadata(v2g_map=v2g, ilist=radial.ilist[ta], jlist=radial.jlist[ta],
                               klist=radial.jlist[tb], n1=n1, n2=n2, n3=n2 - n1, rijk=rijk)

    def get_metadata(self, atoms, vap: VirtualAtomMap):
        """(RadialMetadata, AngularMetadata | None), universal.py:787-849."""
        dtype = np.float64
        radial, aux = self._radial_metadata(atoms, vap, self._rcut, dtype)
        angular = None
        if self._angular:
            if np.round(self._acut - self._rcut, 2) == 0.0:
                ref, ref_aux = radial, aux
            else:
                ref, ref_aux = self._radial_metadata(atoms, vap, self._acut, dtype)
            angular = self._angular_metadata(atoms, vap, ref, ref_aux, dtype)
        return radial, angular

    def get_np_feed_dict(self, atoms):
        """The feed dict of universal.py:851-893 (same keys, dtypes, shapes)."""
        np_dtype = np.float64
        vap = self.get_vap_transformer(atoms)
        radial, angular = self.get_metadata(atoms, vap)
        positions = vap.map_positions(atoms.positions)
        cell = np.asarray(atoms.get_cell(complete=True), dtype=np_dtype)
        feed = dict()
        if self._use_computed_dists:
            feed["positions"] = positions.astype(np_dtype)
            feed["cell"] = cell
            feed["volume"] = np_dtype(atoms.get_volume())
        feed["n_atoms_vap"] = np.int32(vap.max_vap_natoms)
        nnl = radial.v2g_map[:, 2].max() + 1 if len(radial.v2g_map) else 0
        feed["nnl_max"] = np.int32(nnl)
        feed["atom_masks"] = vap.atom_masks.astype(np_dtype)
        feed["etemperature"] = np_dtype(atoms.info.get("etemperature", 0.0))
        feed["row_splits"] = np.int32([1] + [vap.max_occurs[e] for e in self._elements])
        feed.update(radial.as_dict(use_computed_dists=self._use_computed_dists))
        if self._angular:
            ij2k = angular.v2g_map[:, 3].max() + 1 if len(angular.v2g_map) else 0
            feed["ij2k_max"] = np.int32(ij2k)
            feed.update(angular.as_dict(use_computed_dists=self._use_computed_dists))
        return feed

    def get_feed_dict(self, atoms):
        """The reference keys TF placeholders; without TF the keys are the names."""
        return self.get_np_feed_dict(atoms)

    def get_placeholder_features(self) -> Dict:
        """The reference returns the dict of `tf.placeholder`s it feeds (transformer/base.py:191,
        created by `_initialize_placeholders`, universal.py:728-785). Without TF a placeholder is its
        specification: the same keys in the same order, each a `Placeholder(name, dtype, shape)` with
        `None` for a dimension that varies with the structure and the tensor name the frozen graph
        uses (`Placeholders/<key>:0`). `get_np_feed_dict` fills exactly these keys."""
        from collections import namedtuple
        Placeholder = namedtuple("Placeholder", ["name", "dtype", "shape"])
        f, i = np.dtype(np.float64), np.dtype(np.int32)
        spec = []
        if self._use_computed_dists:
            spec += [("positions", f, (None, 3)), ("cell", f, (3, 3)), ("volume", f, ())]
        spec += [("n_atoms_vap", i, ()), ("nnl_max", i, ()), ("atom_masks", f, (None,)),
                 ("etemperature", f, ()), ("row_splits", i, (self.n_elements + 1,)),
                 ("g2.v2g_map", i, (None, 5))]
        if self._use_computed_dists:
            spec += [("g2.ilist", i, (None,)), ("g2.jlist", i, (None,)), ("g2.n1", f, (None, 3))]
        else:
            spec += [("g2.rij", f, (4, None))]
        if self._angular:
            spec += [("ij2k_max", i, ()), ("g4.v2g_map", i, (None, 5))]
            if self._use_computed_dists:
                spec += [("g4.ilist", i, (None,)), ("g4.jlist", i, (None,)), ("g4.klist", i, (None,)),
                         ("g4.n1", f, (None, 3)), ("g4.n2", f, (None, 3)), ("g4.n3", f, (None, 3))]
            else:
                spec += [("g4.rijk", f, (12, None))]
        return {k: Placeholder(f"Placeholders/{k}:0", dt, shape) for k, dt, shape in spec}

    def get_constant_features(self, atoms):
        return self.get_np_feed_dict(atoms)

    # ---- dense universal descriptors (parity / debugging only) --------------
    def get_descriptors(self, features: dict):
        """
        NumPy counterpart of `build_graph` (universal.py:708-726): scatters the
        pair / triple geometry into the dense padded arrays of the reference,
        {"radial": {el: (dists [4,nr,n_el,nnl,1], masks [nr,n_el,nnl,1])},
         "angular": {el: (dists [12,na,n_el,nnl,ij2k], masks)} | None,
         "atom_masks": {el: mask}}.
        """
        R, h = features["positions"], features["cell"]
        eps = 1e-14

        def geom(il, jl, n):
            D = R[jl] - R[il] + (n @ h if self._periodic else 0.0)
            return np.sqrt(np.sum(D * D, axis=1) + eps), D

        splits = np.asarray(features["row_splits"])
        bounds = np.cumsum(splits)

        def split(dense, masks):
            out = {}
            for k, el in enumerate(self._elements):
                lo, hi = bounds[k], bounds[k + 1]
                out[el] = (dense[:, :, lo:hi], masks[:, lo:hi])
            return out

        nvap, nnl = int(features["n_atoms_vap"]), int(features["nnl_max"])
        m = features["g2.v2g_map"]
        r, D = geom(features["g2.ilist"], features["g2.jlist"], features["g2.n1"])
        dense = np.zeros((4, self._max_nr_terms, nvap, nnl, 1))
        masks = np.zeros((self._max_nr_terms, nvap, nnl, 1))
        idx = (m[:, 0], m[:, 1], m[:, 2], m[:, 3])
        for c, v in enumerate((r, D[:, 0], D[:, 1], D[:, 2])):
            np.add.at(dense[c], idx, v)
        np.add.at(masks, idx, m[:, 4].astype(float))
        out = {"radial": split(dense, masks), "angular": None}
        if self._angular:
            ij2k = int(features["ij2k_max"])
            m = features["g4.v2g_map"]
            rij, Dij = geom(features["g4.ilist"], features["g4.jlist"], features["g4.n1"])
            rik, Dik = geom(features["g4.ilist"], features["g4.klist"], features["g4.n2"])
            rjk, Djk = geom(features["g4.jlist"], features["g4.klist"], features["g4.n3"])
            dense = np.zeros((12, self._max_na_terms, nvap, nnl, ij2k))
            masks = np.zeros((self._max_na_terms, nvap, nnl, ij2k))
            idx = (m[:, 0], m[:, 1], m[:, 2], m[:, 3])
            vals = (rij, Dij[:, 0], Dij[:, 1], Dij[:, 2], rik, Dik[:, 0], Dik[:, 1], Dik[:, 2],
                    rjk, Djk[:, 0], Djk[:, 1], Djk[:, 2])
            for c, v in enumerate(vals):
                np.add.at(dense[c], idx, v)
            np.add.at(masks, idx, m[:, 4].astype(float))
            out["angular"] = split(dense, masks)
        am = np.asarray(features["atom_masks"])
        out["atom_masks"] = {el: am[bounds[k]:bounds[k + 1]]
                             for k, el in enumerate(self._elements)}
        return out

    build_graph = get_descriptors
