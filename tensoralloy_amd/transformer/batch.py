"""
`BatchUniversalTransformer`: the padded per-structure records of the reference's training pipeline
(reference tensoralloy/transformer/universal.py:921-1388, base.py:365-437), as NumPy arrays.

In the reference this class turns an `Atoms` object into a `tf.train.Example` (`encode`), decodes it
inside the input pipeline (`decode_protobuf`) and scatters a mini-batch into dense tensors
(`get_descriptors`). Without TensorFlow the record is a dict of arrays with the SAME keys, dtypes and
padded shapes the protobuf holds (`encode`), `decode` splits the merged index arrays the way
`_decode_example` does, and `get_descriptors` returns the dense padded arrays with a leading batch
axis. The GPU path does not need any of this -- frames are packed, not padded (`Engine.set_frames`)
-- so the class exists for callers that produce or consume the reference's record layout;
`frames(batch)` hands such records back as `Atoms` for the engine.
"""
from __future__ import annotations

from collections import Counter
from typing import Dict, List

import numpy as np

from .metadata import AngularMetadata, RadialMetadata
from .universal import UniversalTransformer
from .vap import VirtualAtomMap


class BatchUniversalTransformer(UniversalTransformer):
    """The universal transformer for mini-batch training (universal.py:921-964)."""

    def __init__(self, max_occurs: Counter, rcut, acut=None, angular=False, periodic=True,
                 symmetric=True, nij_max=None, nijk_max=None, nnl_max=None, ij2k_max=None,
                 batch_size=None, use_forces=True, use_stress=False):
        elements = sorted(max_occurs.keys())
        UniversalTransformer.__init__(self, elements=elements, rcut=rcut, acut=acut, angular=angular,
                                      periodic=periodic, symmetric=symmetric, use_computed_dists=True)
        self._use_forces = bool(use_forces)
        self._use_stress = bool(use_stress)
        self._nij_max = nij_max
        self._nijk_max = nijk_max
        self._nnl_max = nnl_max
        self._ij2k_max = ij2k_max
        self._batch_size = batch_size
        self._max_occurs = Counter(max_occurs)
        self._max_n_atoms = sum(self._max_occurs.values())
        self._batch_vaps: Dict[str, VirtualAtomMap] = {}

    # -- reference-compatible surface -------------------------------------------------------
    def as_dict(self):
        return {"class": self.__class__.__name__, "max_occurs": dict(self._max_occurs),
                "rcut": self._rcut, "acut": self._acut, "angular": self._angular,
                "nij_max": self._nij_max, "nijk_max": self._nijk_max, "nnl_max": self._nnl_max,
                "ij2k_max": self._ij2k_max, "batch_size": self._batch_size,
                "use_forces": self._use_forces, "use_stress": self._use_stress}

    @property
    def use_forces(self):
        return self._use_forces

    @property
    def use_stress(self):
        return self._use_stress

    @property
    def batch_size(self):
        return self._batch_size

    @property
    def nij_max(self):
        return self._nij_max

    @property
    def nijk_max(self):
        return self._nijk_max

    @property
    def nnl_max(self):
        return self._nnl_max

    @property
    def ij2k_max(self):
        return self._ij2k_max

    @property
    def max_occurs(self):
        return self._max_occurs

    @property
    def max_n_atoms(self):
        return self._max_n_atoms

    def as_descriptor_transformer(self) -> UniversalTransformer:
        """The `UniversalTransformer` with the same settings (universal.py:1040-1048)."""
        return UniversalTransformer(elements=sorted(self._max_occurs.keys()), rcut=self._rcut,
                                    acut=self._acut, angular=self._angular, periodic=self._periodic,
                                    symmetric=self._symmetric)

    def get_vap_transformer(self, atoms) -> VirtualAtomMap:
        """One map per reduced formula, sized by the DATA SET's `max_occurs`
        (universal.py:1196-1218), not by the structure's own counts."""
        formula = atoms.get_chemical_formula(mode="reduce")
        if formula not in self._batch_vaps:
            symbols = atoms.get_chemical_symbols()
            counts = Counter(symbols)
            for el, n in counts.items():
                if n > self._max_occurs.get(el, 0):
                    raise ValueError(f"{n} {el} atoms exceed max_occurs[{el}] = {self._max_occurs.get(el, 0)}")
            self._batch_vaps[formula] = VirtualAtomMap(self._max_occurs, symbols)
        return self._batch_vaps[formula]

    def get_metadata(self, atoms, vap: VirtualAtomMap):
        """Index maps padded with zero rows (mask 0) to `nij_max` / `nijk_max`, the TRAIN mode of
        `get_radial_metadata` / `get_angular_metadata` (universal.py:1050-1085, :46-233)."""
        radial, angular = UniversalTransformer.get_metadata(self, atoms, vap)

        def pad(a, n, what):
            if n is None:
                return a
            if len(a) > n:
                raise ValueError(f"the structure has {len(a)} {what}, more than the declared maximum {n}")
            out = np.zeros((n,) + a.shape[1:], dtype=a.dtype)
            out[:len(a)] = a
            return out

        n = self._nij_max
        radial = RadialMetadata(v2g_map=pad(radial.v2g_map, n, "pairs (nij)"), ilist=pad(radial.ilist, n, "pairs"),
                                jlist=pad(radial.jlist, n, "pairs"), n1=pad(radial.n1, n, "pairs"), rij=None)
        if angular is not None:
            n = self._nijk_max
            angular = AngularMetadata(
                v2g_map=pad(angular.v2g_map, n, "triples (nijk)"), ilist=pad(angular.ilist, n, "triples"),
                jlist=pad(angular.jlist, n, "triples"), klist=pad(angular.klist, n, "triples"),
                n1=pad(angular.n1, n, "triples"), n2=pad(angular.n2, n, "triples"),
                n3=pad(angular.n3, n, "triples"), rijk=None)
        return radial, angular

    # -- records ----------------------------------------------------------------------------------
    def encode(self, atoms) -> Dict[str, np.ndarray]:
        """The payload of the reference's `tf.train.Example` (base.py:383-437, :365-378;
        universal.py:1178-1231) as arrays: `positions` [max_n_atoms + 1, 3] in GSL order,
        `cell`, `volume`, `n_atoms_vap` (the reference stores `len(atoms)` under this key,
        base.py:411), `energy`, `free_energy`, `atom_masks`, `eentropy`, `etemperature`,
        `forces` / `stress` when used, `g2.indices` [nij_max, 7] = (v2g_map, ilist, jlist),
        `g2.shifts` [nij_max, 3], `g4.indices` [nijk_max, 8] = (v2g_map, ilist, jlist, klist),
        `g4.shifts` [nijk_max, 9] = (n1, n2, n3). Labels come from `atoms.info` / an attached
        calculator; a structure without labels gets zeros, as the reference's dummy calculator."""
        vap = self.get_vap_transformer(atoms)
        f8 = np.float64
        info = getattr(atoms, "info", {}) or {}

        def label(name, default):
            if name in info:
                return np.asarray(info[name], dtype=f8)
            calc = getattr(atoms, "calc", None)
            res = getattr(calc, "results", None) or {}
            if name in res:
                return np.asarray(res[name], dtype=f8)
            return np.asarray(default, dtype=f8)

        energy = np.atleast_1d(label("energy", 0.0))
        etemp = np.atleast_1d(f8(info.get("etemperature", 0.0)))
        eentropy = np.atleast_1d(f8(info.get("eentropy", 0.0)))
        out = {"positions": vap.map_positions(atoms.positions).astype(f8),
               "cell": np.asarray(atoms.get_cell(complete=True), dtype=f8),
               "n_atoms_vap": np.int64(len(atoms)),
               "volume": np.atleast_1d(f8(atoms.get_volume())),
               "energy": energy, "free_energy": energy - etemp * eentropy,
               "atom_masks": vap.atom_masks.astype(f8), "eentropy": eentropy, "etemperature": etemp}
        if self._use_forces:
            out["forces"] = vap.map_forces(label("forces", np.zeros((len(atoms), 3)))).astype(f8)
        if self._use_stress:
            out["stress"] = label("stress", np.zeros(6)).reshape(6)
        radial, angular = self.get_metadata(atoms, vap)
        out["g2.indices"] = np.concatenate((radial.v2g_map, radial.ilist[:, None], radial.jlist[:, None]),
                                           axis=1).astype(np.int32)
        out["g2.shifts"] = radial.n1.astype(f8)
        if angular is not None:
            out["g4.indices"] = np.concatenate((angular.v2g_map, angular.ilist[:, None], angular.jlist[:, None],
                                                angular.klist[:, None]), axis=1).astype(np.int32)
            out["g4.shifts"] = np.concatenate((angular.n1, angular.n2, angular.n3), axis=1).astype(f8)
        return out

    def decode(self, example: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
        """Split the merged arrays the way `_decode_example` does (universal.py:1233-1293):
        `g2.v2g_map` [nij_max, 5], `g2.ilist`, `g2.jlist`, `g2.n1`, and for angular models
        `g4.v2g_map`, `g4.ilist`, `g4.jlist`, `g4.klist`, `g4.n1`, `g4.n2`, `g4.n3`."""
        out = {k: v for k, v in example.items() if not k.startswith(("g2.", "g4."))}
        g2 = np.asarray(example["g2.indices"])
        out["g2.v2g_map"], out["g2.ilist"], out["g2.jlist"] = g2[:, :5], g2[:, 5], g2[:, 6]
        out["g2.n1"] = np.asarray(example["g2.shifts"])
        if "g4.indices" in example:
            g4 = np.asarray(example["g4.indices"])
            out["g4.v2g_map"] = g4[:, :5]
            out["g4.ilist"], out["g4.jlist"], out["g4.klist"] = g4[:, 5], g4[:, 6], g4[:, 7]
            sh = np.asarray(example["g4.shifts"])
            out["g4.n1"], out["g4.n2"], out["g4.n3"] = sh[:, 0:3], sh[:, 3:6], sh[:, 6:9]
        return out

    def batch(self, examples: List[Dict[str, np.ndarray]]) -> Dict[str, np.ndarray]:
        """Stack decoded records along a new leading axis (what `tf.data` batching does)."""
        dec = [self.decode(e) for e in examples]
        return {k: np.stack([np.asarray(d[k]) for d in dec]) for k in dec[0]}

    def get_descriptors(self, batch_features: Dict[str, np.ndarray]):
        """Dense padded universal descriptors of a mini-batch (universal.py:1343-1388):
        {"radial": {el: (dists [4, batch, nr, n_el, nnl_max, 1], masks [batch, nr, n_el, nnl_max, 1])},
         "angular": ... | None, "atom_masks": {el: [batch, n_el]}} -- the single-structure arrays of
        `UniversalTransformer.get_descriptors` with `nnl_max` / `ij2k_max` of the data set, stacked."""
        if self._nnl_max is None or (self._angular and self._ij2k_max is None):
            raise ValueError("nnl_max (and ij2k_max for angular models) must be set")
        B = len(batch_features["positions"])
        per = []
        for b in range(B):
            f = {k: np.asarray(v[b]) for k, v in batch_features.items()}
            f["n_atoms_vap"] = np.int32(self._max_n_atoms + 1)
            f["nnl_max"] = np.int32(self._nnl_max)
            f["row_splits"] = np.int32([1] + [self._max_occurs[e] for e in self._elements])
            if self._angular:
                f["ij2k_max"] = np.int32(self._ij2k_max)
            per.append(UniversalTransformer.get_descriptors(self, f))
        out = {"radial": {}, "angular": None if not self._angular else {}, "atom_masks": {}}
        for el in self._elements:
            out["radial"][el] = (np.stack([p["radial"][el][0] for p in per], axis=1),
                                 np.stack([p["radial"][el][1] for p in per], axis=0))
            if self._angular:
                out["angular"][el] = (np.stack([p["angular"][el][0] for p in per], axis=1),
                                      np.stack([p["angular"][el][1] for p in per], axis=0))
            out["atom_masks"][el] = np.stack([p["atom_masks"][el] for p in per], axis=0)
        return out

    def frames(self, batch_features: Dict[str, np.ndarray], symbols_of: List[List[str]]):
        """`Atoms` objects (caller's atom order) of the records of a batch, for `Engine.set_frames`;
        `symbols_of[b]` = the chemical symbols of structure b in its original order."""
        from ..atoms import Atoms
        out = []
        for b, symbols in enumerate(symbols_of):
            vap = VirtualAtomMap(self._max_occurs, symbols)
            pos = vap.map_positions(np.asarray(batch_features["positions"][b]), reverse=True)
            out.append(Atoms(symbols=symbols, positions=pos, cell=np.asarray(batch_features["cell"][b]),
                             pbc=self._periodic))
        return out
