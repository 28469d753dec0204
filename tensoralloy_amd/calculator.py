"""
`TensorAlloyCalculator`: drop-in for reference tensoralloy/calculator.py:31-383.

Same constructor, properties and methods; `calculate` runs the HIP library
instead of a TensorFlow session. `self.results` holds the same keys and shapes
as the reference: GSL-ordered (element-sorted), virtual-row-stripped arrays for
`forces` and `energy/atom`; `get_forces` / `get_atomic` map them back to the
caller's atom order exactly as the reference does (calculator.py:217-249).

Differences (documented in INTEGRATION.md):
  * `graph_model_path` names a `<name>.json` + `<name>.npz` pair written by
    `AtomicNN.export` / `EamAlloyNN.export`, the reference's native `.npz`, or a frozen
    TensorFlow GraphDef `.pb` of the reference itself: its constants are read without
    TensorFlow (`tensoralloy_amd/graphdef.py`; Zjw04-family EAM graphs and symmetry-function
    `AtomicNN` graphs; anything else raises `ValueError`).
  * `session`, `graph`, `get_op` are TensorFlow objects in the reference; here
    they raise `AttributeError`.
  * `hessian` and `elastic` are central differences of the analytic forces / virial (the
    reference differentiates its graph twice, basic.py:411-421, constraint/elastic.py:24-92);
    `eentropy` and `enthalpy` are not implemented; `free_energy` only where it is the energy op.
"""
from __future__ import annotations

from typing import List

import numpy as np

from . import _lib
from .atoms import BaseCalculator, all_changes
from .engine import Engine
from .model import EXPORTABLE_PROPERTIES, load_model
from .utils import GPa, ModeKeys


class TensorAlloyCalculator(BaseCalculator):
    """ASE-Calculator for tensoralloy_amd model files."""

    implemented_properties = list(EXPORTABLE_PROPERTIES)
    default_parameters = {}
    nolabel = True

    def __init__(self, graph_model_path: str, atoms=None, serial_mode=False, device: int = 0,
                 skin: float = 0.5):
        """
        graph_model_path : the exported model to load.
        atoms            : the target `Atoms` object.
        serial_mode      : accepted for compatibility (the reference limits TF
                           to one CPU thread, calculator.py:70-75); ignored.
        device           : HIP device index (new; default 0).
        skin             : Verlet skin in Angstrom (new; default 0.5). Successive calls for the
                           same system keep the neighbour list while no atom has moved further
                           than skin / 2; 0 builds an exact list on every call as the reference
                           does. Results do not depend on it beyond summation order.
        """
        super().__init__(restart=None, ignore_bad_restart_file=False, label=None, atoms=atoms)
        self._graph_model_path = graph_model_path
        self._mode = ModeKeys.PREDICT
        nn, clf, meta = load_model(graph_model_path)
        self._nn = nn
        self._transformer = clf
        self._meta = meta
        self._get_ops()
        self._engine = Engine(nn, device=device)
        self._skin = float(skin)
        self._engine.set_skin(self._skin)
        self._vap_cache = (None, None)
        self.implemented_properties = self._predict_properties
        self._ncalls = 0
        self._prerequisite_properties = []

    @property
    def skin(self) -> float:
        return self._skin

    @skin.setter
    def skin(self, value: float):
        self._skin = float(value)
        self._engine.set_skin(self._skin)

    def _vap_for(self, atoms):
        """`get_vap_transformer(atoms)`, remembered for the system of the previous call."""
        key = np.asarray(atoms.numbers).tobytes()
        if self._vap_cache[0] != key:
            self._vap_cache = (key, self.transformer.get_vap_transformer(atoms))
        return self._vap_cache[1]

    # -- TF-specific members of the reference ------------------------------------
    @property
    def session(self):
        raise AttributeError("tensoralloy_amd has no TensorFlow session")

    @property
    def graph(self):
        raise AttributeError("tensoralloy_amd has no TensorFlow graph")

    def get_op(self, name):
        raise AttributeError("tensoralloy_amd has no TensorFlow graph")

    # -- metadata ---------------------------------------------------------------------
    def _get_ops(self):
        ops = {prop: name for prop, name in self._meta["Metadata/ops"].items()
               if name.endswith(":0")}
        # graphs exported by the reference's older API also list `atomic` (= the per-atom energies,
        # today's `energy/atom`), `total_stress` (the full 3 x 3 stress), and `free_energy` /
        # `enthalpy` ops; a zero-temperature model's free energy IS its energy op. What this build
        # cannot produce (eentropy, enthalpy with a Pulay stress) is not offered.
        producible = {"energy", "energy/atom", "atomic", "forces", "stress", "virial", "total_stress",
                      "total_pressure", "hessian", "elastic"}
        if ops.get("free_energy") and ops.get("free_energy") == ops.get("energy"):
            producible.add("free_energy")
        ops = {k: v for k, v in ops.items() if k in producible}
        if not ops:
            raise Exception("Validated Ops cannot be found")  # calculator.py:161
        self._ops = ops
        self._predict_properties = list(ops.keys())
        self._fp_precision = self._meta.get("Metadata/precision", "high")
        if self._fp_precision not in ("high", "medium"):
            raise ValueError(f"unknown precision {self._fp_precision!r}")
        # 'medium' models are float32 graphs in the reference (calculator.py:154-159); here they
        # are evaluated in float64 with the float32 eps and the results are cast to float32
        self._fp_dtype = np.float64 if self._fp_precision == "high" else np.float32
        self._is_finite_temperature = bool(int(self._meta.get("Metadata/is_finite_temperature", 0)))
        self._variational_energy = self._meta.get("Metadata/variational_energy", "energy")
        self._api_version = self._meta.get("Metadata/api", "1.1")

    @property
    def elements(self) -> List[str]:
        return self._transformer.elements

    @property
    def transformer(self):
        return self._transformer

    @property
    def predict_properties(self):
        return self._predict_properties

    def get_model_timestamp(self):
        return self._meta.get("Metadata/timestamp")

    @property
    def api_version(self):
        return self._api_version

    @property
    def variational_energy(self):
        return self._variational_energy

    # -- property getters ----------------------------------------------------------------
    def get_potential_energy(self, atoms=None, force_consistent=False):
        return self.get_property("energy", atoms)

    def get_magnetic_moment(self, atoms=None):
        return None

    def get_magnetic_moments(self, atoms=None):
        return None

    def get_electron_entropy(self, atoms=None):
        return self.get_property("eentropy", atoms=atoms)

    def get_free_energy(self, atoms=None):
        return self.get_property("free_energy", atoms=atoms)

    def get_atomic(self, atoms=None, prop="energy"):
        """Per-atom values in the caller's atom order (calculator.py:217-226)."""
        atoms = atoms if atoms is not None else self.atoms
        values = self.get_property(f"{prop}/atom", atoms=atoms)
        values = np.insert(values, 0, 0, 0)
        clf = self.transformer.get_vap_transformer(atoms)
        return clf.map_array(values.reshape((-1, 1)), reverse=True).flatten()

    def get_hessian(self, atoms=None):
        """d2E/dR2 as [3N, 3N] in the caller's atom order (calculator.py:228-241)."""
        atoms = atoms if atoms is not None else self.atoms
        hessian = self.get_property("hessian", atoms)
        clf = self.transformer.get_vap_transformer(atoms)
        return clf.reverse_map_hessian(hessian)

    def get_forces(self, atoms=None):
        atoms = atoms if atoms is not None else self.atoms
        gsl = self.get_property("forces", atoms)
        # the engine delivers the caller's order; `results` holds the GSL-ordered copy the reference
        # exposes. `get_property` hands out copies, so the shortcut is keyed on the evaluation
        # counter: while `results["forces"]` is still the array the last `calculate` stored, mapping
        # it back (calculator.py:247-249) would only undo the permutation.
        cached = getattr(self, "_forces_local", None)
        if cached is not None and cached[0] == self._ncalls and \
                self.results.get("forces") is cached[1] and cached[2].shape == gsl.shape:
            return cached[2].copy()
        forces = np.insert(gsl, 0, 0, 0)
        return self._vap_for(atoms).map_forces(forces, reverse=True)

    def get_stress(self, atoms=None, voigt=True):
        """Stress in eV/Angstrom**3; Voigt order (xx, yy, zz, yz, xz, xy)."""
        if atoms is None:
            atoms = self.atoms
        stress = self.get_property("stress", atoms)
        if not voigt:
            xx, yy, zz, yz, xz, xy = stress
            stress = np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])
        return stress

    def get_total_pressure(self, atoms=None):
        """Total pressure in GPa (calculator.py:279-295)."""
        stress = self.get_stress(atoms)
        return np.mean(stress[:3]) * (-1.0) / GPa

    def get_elastic_constant_tensor(self, atoms=None):
        """6 x 6 elastic constants in GPa, upper triangle mirrored (calculator.py:297-322)."""
        atoms = atoms if atoms is not None else self.atoms
        assert np.all(atoms.pbc)
        elastic = np.array(self.get_property("elastic", atoms, allow_calculation=True))
        for i in range(6):
            for j in range(i + 1, 6):
                elastic[j, i] = elastic[i, j]
        return elastic

    # -- second derivatives ------------------------------------------------------------
    _FD_STEP = 1e-4  # Angstrom; central differences of analytic first derivatives

    def _displaced(self, atoms, positions=None, cell=None):
        from .atoms import Atoms
        return Atoms(numbers=np.asarray(atoms.numbers).copy(),
                     positions=atoms.positions if positions is None else positions,
                     cell=np.asarray(atoms.get_cell(complete=True)) if cell is None else cell,
                     pbc=np.asarray(atoms.pbc).copy())

    def _hessian(self, atoms, vap):
        """[n_vap, 3, n_vap, 3], GSL order, virtual atom = zero row/column: the layout of
        `tf.hessians(energy, positions)` (basic.py:411-421). H = -dF/dR."""
        n = len(atoms)
        d = self._FD_STEP
        want = _lib.TA_WANT_ENERGY | _lib.TA_WANT_FORCES | _lib.TA_WANT_VIRIAL
        H = np.zeros((n, 3, n, 3))
        analytic = self._analytic_hessian(atoms)
        if analytic is not None:
            H = analytic
        jobs = [] if analytic is not None else \
            [(i, a, sgn) for i in range(n) for a in range(3) for sgn in (1.0, -1.0)]
        chunk = max(2, min(len(jobs), (1 << 18) // max(n, 1) // 2 * 2))
        for k0 in range(0, len(jobs), chunk):
            part = jobs[k0:k0 + chunk]
            frames = []
            for i, a, sgn in part:
                pos = atoms.positions.copy()
                pos[i, a] += sgn * d
                frames.append(self._displaced(atoms, positions=pos))
            res = self._engine.evaluate(frames, want=want)
            for (i, a, sgn), r in zip(part, res):
                H[i, a] -= sgn * r["forces"] / (2.0 * d)
        H = 0.5 * (H + H.transpose(2, 3, 0, 1))
        nv = vap.max_vap_natoms
        out = np.zeros((nv, 3, nv, 3))
        idx = np.asarray(vap.local_to_gsl)
        out[np.ix_(idx, range(3), idx, range(3))] = H
        return out

    def _analytic_hessian(self, atoms):
        """H[i, a, j, b] = d^2 E / dR_ia dR_jb from `ta_hessian_vectors` (dual-number tangents through the
        analytic force kernels: exact, no step), or None where the model has no analytic path."""
        n = len(atoms)
        try:
            self._engine.set_frames([atoms])
            H = np.zeros((n, 3, n, 3))
            dF = self._engine.hessian_vectors()          # [3 n, n, 3] = d F / d R_(k, a)
        except ValueError:
            return None
        for k in range(n):
            for a in range(3):
                H[k, a] = -dF[3 * k + a]
        return H

    def _elastic(self, atoms):
        """C[vi, vj] = ((dW_ij/dh)^T h)_kl / V / GPa with the positions held fixed, as the reference's
        op differentiates the virial with respect to the cell placeholder
        (nn/constraint/elastic.py:24-44); Voigt rows/columns xx yy zz yz xz xy."""
        d = self._FD_STEP
        want = _lib.TA_WANT_ENERGY | _lib.TA_WANT_FORCES | _lib.TA_WANT_VIRIAL
        h = np.asarray(atoms.get_cell(complete=True), dtype=np.float64)
        dW = np.zeros((3, 3, 3, 3))  # [i, j, a, b] = dW_ij / dh_ab
        try:   # analytic: the nine unit cell directions, positions fixed (`ta_hessian_vectors`)
            self._engine.set_frames([atoms])
            dh = np.zeros((9, 1, 3, 3))
            for a in range(3):
                for b in range(3):
                    dh[3 * a + b, 0, a, b] = 1.0
            _, dWa = self._engine.hessian_vectors(dh=dh, want_virial=True)
            for a in range(3):
                for b in range(3):
                    dW[:, :, a, b] = dWa[3 * a + b, 0]
            frames = []
        except ValueError:
            frames, keys = [], []
            for a in range(3):
                for b in range(3):
                    for sgn in (1.0, -1.0):
                        cell = h.copy()
                        cell[a, b] += sgn * d
                        frames.append(self._displaced(atoms, cell=cell))
                        keys.append((a, b, sgn))
        if frames:
            res = self._engine.evaluate(frames, want=want)
            for (a, b, sgn), r in zip(keys, res):
                dW[:, :, a, b] += sgn * r["virial"] / (2.0 * d)
        volume = abs(np.linalg.det(h))
        pairs = [(0, 0), (1, 1), (2, 2), (1, 2), (0, 2), (0, 1)]
        C = np.zeros((6, 6))
        for vi, (i, j) in enumerate(pairs):
            cij = dW[i, j].T @ h / volume / GPa
            for vj, (k, l) in enumerate(pairs):
                C[vi, vj] = cij[k, l]
        return C

    def set_prerequisite_properties(self, properties: List[str]):
        for prop in properties:
            if prop in self.implemented_properties:
                self._prerequisite_properties.append(prop)

    # -- the hot path ------------------------------------------------------------------------
    def calculate(self, atoms=None, properties=("energy", "forces"), system_changes=all_changes,
                  debug_mode=False, extra_ops=None):
        """
        Evaluate `properties` (plus the prerequisite ones) for `atoms`.
        Replaces `Session.run(ops, feed_dict)` (calculator.py:355-370).
        """
        BaseCalculator.calculate(self, atoms, properties, system_changes)
        atoms = atoms if atoms is not None else self.atoms
        properties = set(properties).union(self._prerequisite_properties)
        for target in properties:
            if target not in self._ops:
                raise KeyError(target)  # self._ops[target], calculator.py:360
        want = _lib.TA_WANT_ENERGY | _lib.TA_WANT_ATOMIC
        if properties & {"forces", "stress", "virial", "total_pressure", "total_stress"}:
            want |= _lib.TA_WANT_FORCES | _lib.TA_WANT_VIRIAL
        vap = self._vap_for(atoms)
        second = {}
        if "hessian" in properties:
            second["hessian"] = self._hessian(atoms, vap)
        if "elastic" in properties:
            second["elastic"] = self._elastic(atoms)
        res = self._engine.evaluate_md(atoms, want=want, descriptors=debug_mode)
        results = dict(second)
        local_forces = None
        for target in properties:
            if target == "energy":
                results[target] = res["energy"]
            elif target == "free_energy":
                results[target] = res["energy"]
            elif target == "total_stress":
                if "stress" not in res:
                    raise ValueError("'total_stress' needs a cell with three lattice vectors: volume not defined")
                v = res["stress"]
                results[target] = np.array([[v[0], v[5], v[4]], [v[5], v[1], v[3]], [v[4], v[3], v[2]]])
            elif target in ("energy/atom", "atomic"):
                # GSL order, virtual row stripped (atomic.py:289-299)
                results[target] = res["atomic"] if vap.is_identity else \
                    vap.map_array(res["atomic"].reshape(-1, 1))[1:, 0]
            elif target == "forces":
                # (identity map: GSL order is the caller's order, no gather through a padded copy)
                results[target] = res["forces"] if vap.is_identity else vap.map_forces(res["forces"])[1:]
                local_forces = res["forces"]
            elif target in ("stress", "virial", "total_pressure"):
                if target not in res:
                    raise ValueError(f"'{target}' needs a cell with three lattice vectors: "
                                     "volume not defined")
                results[target] = res[target]
        if debug_mode:
            results["descriptors"] = res["descriptors"]
        if self._fp_dtype is not np.float64:
            results = {k: (np.asarray(v, dtype=self._fp_dtype) if isinstance(v, np.ndarray)
                           else self._fp_dtype(v)) for k, v in results.items()}
            if local_forces is not None:
                local_forces = np.asarray(local_forces, dtype=self._fp_dtype)
        self.results = results
        self._ncalls += 1
        self._forces_local = (self._ncalls, results["forces"], local_forces) \
            if local_forces is not None else None

    def reset_call_counter(self):
        self._ncalls = 0

    @property
    def ncalls(self):
        return self._ncalls
