"""
`Engine`: thin Python owner of one `ta_handle` (one GPU, one HIP stream).

This is the only place where Python meets the C ABI for evaluation. It plays
the role of `tf.Session` in the reference calculator (calculator.py:79, :368):
load once, then run many structures. Frames of a batch are independent units
and are evaluated by a single set of kernel launches.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Sequence

import numpy as np

from . import _lib
from .utils import GPa


class Engine:
    def __init__(self, nn, device: int = 0):
        self._lib = _lib.load()
        self._nn = nn
        self._clf = nn.transformer
        if self._clf is None:
            raise ValueError("A descriptor transformer must be attached.")
        desc, keep = nn.to_desc()
        self._handle = C.c_void_p()
        rc = self._lib.ta_create(C.byref(desc), int(device), C.byref(self._handle))
        del keep
        if rc != _lib.TA_OK:
            _lib.check(self._lib, None, rc)
        self.device = int(device)
        self.info = None
        self._frames = None
        self._volumes = None
        self._md_sig = None   # (n, numbers, pbc) of the single resident frame of `evaluate_md`
        self._md_cell = None
        self.batch_generation = 0  # bumped whenever the resident batch or its coordinates change

    # -- lifetime ----------------------------------------------------------------
    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.ta_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        _lib.check(self._lib, self._handle, rc)

    # -- batch ----------------------------------------------------------------------
    def set_frames(self, atoms_list: Sequence) -> _lib.BatchInfo:
        """Neighbour lists + upload; afterwards the batch is resident in HBM."""
        frames = []
        periodic = self._clf.periodic
        for atoms in atoms_list:
            pbc = np.asarray(atoms.pbc, dtype=bool) if periodic else np.zeros(3, dtype=bool)
            frames.append(_lib.FrameArrays(self._clf.species_indices(atoms), atoms.positions,
                                           np.asarray(atoms.get_cell(complete=True)), pbc))
        arr = (_lib.Frame * max(len(frames), 1))(*[f.as_struct() for f in frames])
        info = _lib.BatchInfo()
        self._check(self._lib.ta_set_frames(self._handle, len(frames), arr, C.byref(info)))
        self.info = info
        self._frames = frames
        self.batch_generation += 1  # what is resident changed (train.Trainer's shortcut checks this)
        self._md_sig = None
        self._volumes = np.array([abs(np.linalg.det(f.cell)) for f in frames])
        self._natoms = np.array([len(f.species) for f in frames], dtype=np.int64)
        return info

    def set_nn_tables(self, on: bool):
        """nn pair functions of an EAM / ADP model through the library's Hermite tables (default for
        inference) or evaluated exactly for every pair (`ta_set_nn_tables`). Set it before
        `set_frames`."""
        self._check(self._lib.ta_set_nn_tables(self._handle, 1 if on else 0))

    def set_skin(self, skin: float):
        """Verlet skin in Angstrom for the lists built from now on (0 = exact list)."""
        self._check(self._lib.ta_set_skin(self._handle, float(skin)))

    def update_positions(self, positions, cells=None) -> bool:
        """New coordinates for the resident frames (all atoms of the batch, in order; `cells`
        [n_frames, 3, 3] or None = unchanged). Returns True when the neighbour list was rebuilt."""
        pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 3)
        if len(pos) != int(self.info.n_atoms):
            raise ValueError("positions for every atom of the resident batch are needed")
        null = C.POINTER(C.c_double)()
        cptr = null
        if cells is not None:
            cells = np.ascontiguousarray(cells, dtype=np.float64).reshape(-1, 3, 3)
            if len(cells) != int(self.info.n_frames):
                raise ValueError("one cell per resident frame")
            cptr = _lib.as_dp(cells)
        rebuilt = C.c_int32(0)
        self._check(self._lib.ta_update_positions(self._handle, _lib.as_dp(pos), cptr, C.byref(rebuilt)))
        self.batch_generation += 1
        if cells is not None:  # only once the call has succeeded
            self._volumes = np.abs(np.linalg.det(cells))
        if rebuilt.value:
            # a rebuilt list has new pair / triple counts: scripts derive bytes per evaluation from them
            n_pairs, n_triples, nnl = C.c_int64(0), C.c_int64(0), C.c_int32(0)
            self._check(self._lib.ta_list_sizes(self._handle, C.byref(n_pairs), C.byref(n_triples), C.byref(nnl)))
            self.info.n_pairs, self.info.n_triples, self.info.nnl_max = n_pairs.value, n_triples.value, nnl.value
        return bool(rebuilt.value)

    def _view_array(self, ptr, shape):
        """numpy array over library memory at `ptr`; the staging buffer does not move between steps, so the
        wrapper is made once per (address, shape)."""
        addr = C.cast(ptr, C.c_void_p).value
        if not addr:
            return None
        key = (addr, shape)
        arr = self._view_cache.get(key)
        if arr is None:
            if len(self._view_cache) > 16:
                self._view_cache.clear()
            n = int(np.prod(shape))
            arr = np.frombuffer((C.c_double * n).from_address(addr), dtype=np.float64).reshape(shape)
            self._view_cache[key] = arr
        return arr

    def step(self, positions, want: int, cells=None, view=False) -> dict:
        """One MD step of the resident batch in one library call (`ta_step`): new coordinates in,
        evaluation, results out. Same result dict as `fetch`; the output arrays are reused from call to
        call (copy what must outlive the next step). `view=True` (`ta_step_view`): the arrays ARE the
        library's page-locked staging memory the device wrote, valid until the next call on this engine
        (no copy out of it: 128 KB per step for 4000 atoms)."""
        pos = positions if (type(positions) is np.ndarray and positions.dtype == np.float64 and
                            positions.flags.c_contiguous) else np.ascontiguousarray(positions, dtype=np.float64)
        N, F = int(self.info.n_atoms), int(self.info.n_frames)
        if pos.size != 3 * N:
            raise ValueError("positions for every atom of the resident batch are needed")
        st = self.__dict__.get("_step_state")
        if st is None:   # ctypes objects of the call, made once
            null = C.POINTER(C.c_double)()
            ptrs = tuple(C.POINTER(C.c_double)() for _ in range(4))
            rebuilt = C.c_int32(0)
            st = self._step_state = (null, ptrs, tuple(C.byref(p) for p in ptrs), rebuilt, C.byref(rebuilt))
            self._view_cache = {}
        null, ptrs, refs, rebuilt, rebuilt_ref = st
        cptr = null
        if cells is not None:
            cells = np.ascontiguousarray(cells, dtype=np.float64).reshape(-1, 3, 3)
            if len(cells) != F:
                raise ValueError("one cell per resident frame")
            cptr = _lib.as_dp(cells)
        want = int(want)
        want_f = bool(want & (_lib.TA_WANT_FORCES | _lib.TA_WANT_VIRIAL))
        if view:
            rc = self._lib.ta_step_view(self._handle, pos.ctypes.data_as(_lib._dp), cptr, want, refs[0], refs[1],
                                        refs[2], refs[3], rebuilt_ref)
            if rc:
                self._check(rc)
            energy = self._view_array(ptrs[0], (F,))
            if energy is None:
                energy = np.empty(0)
            forces, virial = self._view_array(ptrs[1], (N, 3)), self._view_array(ptrs[2], (F, 3, 3))
            atomic = self._view_array(ptrs[3], (N,))
        else:
            buf = getattr(self, "_step_buf", None)
            if buf is None or buf[0] != (N, F):
                buf = ((N, F), np.empty(F), np.empty((N, 3)), np.empty((F, 3, 3)), np.empty(N))
                self._step_buf = buf
            _, energy, forces, virial, atomic = buf
            self._check(self._lib.ta_step(
                self._handle, _lib.as_dp(pos), cptr, want, _lib.as_dp(energy),
                _lib.as_dp(forces) if want_f else null, _lib.as_dp(virial) if want_f else null,
                _lib.as_dp(atomic) if want & _lib.TA_WANT_ATOMIC else null, rebuilt_ref))
        self.batch_generation += 1
        if cells is not None:
            self._volumes = np.abs(np.linalg.det(cells))
        if rebuilt.value:
            n_pairs, n_triples, nnl = C.c_int64(0), C.c_int64(0), C.c_int32(0)
            self._check(self._lib.ta_list_sizes(self._handle, C.byref(n_pairs), C.byref(n_triples), C.byref(nnl)))
            self.info.n_pairs, self.info.n_triples, self.info.nnl_max = n_pairs.value, n_triples.value, nnl.value
        out = {"energy": energy}
        if want_f and forces is not None:
            out["forces"], out["virial"] = forces, virial
        if want & _lib.TA_WANT_ATOMIC and atomic is not None:
            out["atomic"] = atomic
        return out

    def list_stats(self):
        """(lists built, lists reused) by this engine."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.ta_list_stats(self._handle, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def compute(self, want: int):
        self._check(self._lib.ta_compute(self._handle, int(want)))

    def set_stream(self, stream_ptr):
        """Run on a caller-owned hipStream_t (int handle); None = the engine's own stream.
        0 is the handle of the legacy default stream (what `torch.cuda.current_stream().cuda_stream`
        returns when no other stream was made current): it is passed on as `hipStreamLegacy`,
        because a NULL argument means "the handle's own stream" to the C ABI."""
        HIP_STREAM_LEGACY = 1  # hip_runtime_api.h: #define hipStreamLegacy ((hipStream_t)1)
        if stream_ptr is None:
            ptr = 0
        else:
            ptr = int(stream_ptr) or HIP_STREAM_LEGACY
        self._check(self._lib.ta_set_stream(self._handle, C.c_void_p(ptr)))

    def synchronize(self):
        self._check(self._lib.ta_synchronize(self._handle))

    def fetch(self, want: int, descriptors=False) -> dict:
        """Copy back what `want` asked for; arrays cover the whole batch."""
        info = self.info
        N, F = int(info.n_atoms), int(info.n_frames)
        # (the library fills every array it is handed, or fails)
        out = {"energy": np.empty(F)}
        forces = virial = atomic = desc = None
        if want & (_lib.TA_WANT_FORCES | _lib.TA_WANT_VIRIAL):
            forces = np.empty((N, 3))
            virial = np.empty((F, 3, 3))
        if want & _lib.TA_WANT_ATOMIC:
            atomic = np.empty(N)
        if descriptors:
            desc = np.empty((N, int(info.descriptor_dim)))
        null = C.POINTER(C.c_double)()
        self._check(self._lib.ta_get_results(
            self._handle, _lib.as_dp(out["energy"]),
            _lib.as_dp(forces) if forces is not None else null,
            _lib.as_dp(virial) if virial is not None else null,
            _lib.as_dp(atomic) if atomic is not None else null,
            _lib.as_dp(desc) if desc is not None else null))
        if forces is not None:
            out["forces"], out["virial"] = forces, virial
        if atomic is not None:
            out["atomic"] = atomic
        if desc is not None:
            scale = getattr(self._nn, "descriptor_scale", None)
            out["descriptors"] = desc * scale() if scale is not None else desc
        return out

    def evaluate(self, atoms_list: Sequence, want: int = None, descriptors=False) -> List[dict]:
        """One dict per frame: energy, atomic, forces, virial, stress (Voigt,
        eV/A^3), total_pressure (GPa) in the caller's atom order."""
        if want is None:
            want = (_lib.TA_WANT_ENERGY | _lib.TA_WANT_FORCES | _lib.TA_WANT_VIRIAL |
                    _lib.TA_WANT_ATOMIC)
        self.set_frames(atoms_list)
        self.compute(want)
        return self._per_frame(self.fetch(want, descriptors=descriptors))

    def _per_frame(self, res: dict) -> List[dict]:
        out, a = [], 0
        for f, n in enumerate(self._natoms):
            d = {"energy": float(res["energy"][f])}
            if "atomic" in res:
                d["atomic"] = res["atomic"][a:a + n]
            if "forces" in res:
                d["forces"] = res["forces"][a:a + n]
                w = res["virial"][f]
                d["virial"] = w
                # a frame without three lattice vectors has no volume (ASE's `get_volume`, which
                # feeds the reference's `volume` placeholder at universal.py:865, raises for it):
                # the virial is still defined, stress and pressure are not and are left out
                if self._volumes[f] > 0.0:
                    s = w / self._volumes[f]                       # basic.py:317
                    d["stress"] = np.array([s[0, 0], s[1, 1], s[2, 2], s[1, 2], s[0, 2], s[0, 1]])
                    d["total_pressure"] = float(np.trace(s) / (-3.0 * GPa))  # basic.py:403-405
            if "descriptors" in res:
                d["descriptors"] = res["descriptors"][a:a + n]
            out.append(d)
            a += n
        return out

    def evaluate_md(self, atoms, want: int = None, descriptors=False) -> dict:
        """One structure, as `evaluate([atoms])[0]`, but when it is the system of the previous call
        (same atoms, species, periodicity) only its coordinates are sent and the neighbour list is
        kept as long as the Verlet skin allows (`set_skin`): the MD / relaxation loop of the
        reference's calculator (calculator.py:335-370) without its per-call feed-dict rebuild."""
        if want is None:
            want = (_lib.TA_WANT_ENERGY | _lib.TA_WANT_FORCES | _lib.TA_WANT_VIRIAL |
                    _lib.TA_WANT_ATOMIC)
        periodic = self._clf.periodic
        pbc = tuple(bool(x) for x in atoms.pbc) if periodic else (False, False, False)
        sig = (len(atoms), np.asarray(atoms.numbers).tobytes(), pbc)
        cell = np.ascontiguousarray(atoms.get_cell(complete=True), dtype=np.float64).reshape(3, 3)
        if sig == self._md_sig and self.info is not None and not descriptors:
            same_cell = np.array_equal(cell, self._md_cell)
            res = self.step(atoms.positions, want, None if same_cell else cell[None], view=True)
            self._md_cell = cell
            # (the staging memory is rewritten by the next call: hand out copies, the only ones made)
            return self._per_frame({k: v.copy() for k, v in res.items()})[0]
        if sig == self._md_sig and self.info is not None:
            same_cell = np.array_equal(cell, self._md_cell)
            self.update_positions(atoms.positions, None if same_cell else cell[None])
            self._md_cell = cell
        else:
            self.set_frames([atoms])
            self._md_sig, self._md_cell = sig, cell
        self.compute(want)
        return self._per_frame(self.fetch(want, descriptors=descriptors))[0]

    # -- measurement -------------------------------------------------------------------
    def time_compute(self, want: int, warmup: int, steps: int, per_kernel=True):
        total = C.c_double(0.0)
        slots = np.zeros(_lib.TA_N_KERNEL_SLOTS)
        self._check(self._lib.ta_time_compute(
            self._handle, int(want), int(warmup), int(steps), C.byref(total),
            _lib.as_dp(slots) if per_kernel else C.POINTER(C.c_double)()))
        return total.value, dict(zip(_lib.KERNEL_SLOTS, slots.tolist()))

    def count_contributing_triples(self) -> int:
        """Triples of the resident batch with all three sides below acut (non-zero G4 terms)."""
        n = C.c_int64(0)
        self._check(self._lib.ta_count_contributing_triples(self._handle, C.byref(n)))
        return int(n.value)

    def measure_hbm_copy(self, nbytes: int = 1 << 30, reps: int = 10) -> float:
        """Achievable device-to-device copy rate in GB/s (read + written bytes)."""
        g = C.c_double(0.0)
        self._check(self._lib.ta_measure_hbm_copy(self._handle, int(nbytes), int(reps), C.byref(g)))
        return g.value

    def batch_energy_device_ptr(self) -> int:
        p = C.c_void_p()
        self._check(self._lib.ta_batch_energy_device_ptr(self._handle, C.byref(p)))
        return p.value

    def copy_batch_energy(self, dst_device_ptr: int):
        """Enqueue a D2D copy of the batch energy (one double) on the engine's stream."""
        self._check(self._lib.ta_copy_batch_energy(self._handle, C.c_void_p(int(dst_device_ptr))))

    # -- training support ---------------------------------------------------
    def param_count(self) -> int:
        n = C.c_int64(0)
        self._check(self._lib.ta_param_count(self._handle, C.byref(n)))
        return int(n.value)

    def update_weights(self, flat):
        """Replace the MLP weights of the live handle (layout of `train.flatten_weights`)."""
        flat = np.ascontiguousarray(flat, dtype=np.float64).ravel()
        self._check(self._lib.ta_update_weights(self._handle, _lib.as_dp(flat), len(flat)))

    def energy_gradient(self, frame_coeff) -> np.ndarray:
        """sum_f frame_coeff[f] dE_f/dtheta for the resident batch, flat parameter layout."""
        coeff = np.ascontiguousarray(frame_coeff, dtype=np.float64).ravel()
        if len(coeff) != int(self.info.n_frames):
            raise ValueError("one coefficient per resident frame")
        grad = np.zeros(self.param_count())
        self._check(self._lib.ta_energy_gradient(self._handle, _lib.as_dp(coeff), _lib.as_dp(grad), len(grad)))
        return grad

    def loss_gradient(self, frame_coeff=None, dR=None, dh=None, return_tangent=False):
        """d/dtheta (sum_f frame_coeff[f] E_f + D_delta E) for the resident batch, delta = (dR
        [n_atoms, 3], dh [n_frames, 3, 3]): the gradient of an energy + forces + stress loss in one
        analytic pass (`ta_loss_gradient`); flat parameter layout."""
        null = C.POINTER(C.c_double)()
        N, F = int(self.info.n_atoms), int(self.info.n_frames)

        def arr(a, shape):
            if a is None:
                return None, null
            a = np.ascontiguousarray(a, dtype=np.float64).reshape(shape)
            return a, _lib.as_dp(a)
        c, cp = arr(frame_coeff, (F,))
        r, rp = arr(dR, (N, 3))
        hh, hp = arr(dh, (F, 9))
        grad = np.zeros(self.param_count())
        tangent = np.zeros((N, int(self.info.descriptor_dim))) if return_tangent else None
        self._check(self._lib.ta_loss_gradient(self._handle, cp, rp, hp, _lib.as_dp(grad), len(grad),
                                               _lib.as_dp(tangent) if return_tangent else null))
        if return_tangent:   # directional derivative of the raw descriptors [N, D]
            scale = getattr(self._nn, "descriptor_scale", None)
            return grad, (tangent * scale() if scale is not None else tangent)
        return grad

    # -- constants of the analytic EAM functions as parameters (potentials.py:129-163) ---------
    def constant_count(self) -> int:
        n = C.c_int64(0)
        self._check(self._lib.ta_constant_count(self._handle, C.byref(n)))
        return int(n.value)

    def constants(self) -> np.ndarray:
        """Flat vector: 20 per element (the model's `eam_el` rows), then 7 per sorted pair type
        (Zjw04xcp cross terms)."""
        out = np.zeros(self.constant_count())
        self._check(self._lib.ta_get_constants(self._handle, _lib.as_dp(out), len(out)))
        return out

    def update_constants(self, flat):
        flat = np.ascontiguousarray(flat, dtype=np.float64).ravel()
        self._check(self._lib.ta_update_constants(self._handle, _lib.as_dp(flat), len(flat)))

    def constant_gradient(self, frame_coeff=None, dR=None, dh=None) -> np.ndarray:
        """d/dconstants (sum_f frame_coeff[f] E_f + D_delta E) for the resident batch
        (`ta_constant_gradient`), arguments as `loss_gradient`."""
        null = C.POINTER(C.c_double)()
        N, F = int(self.info.n_atoms), int(self.info.n_frames)

        def arr(a, shape):
            if a is None:
                return None, null
            a = np.ascontiguousarray(a, dtype=np.float64).reshape(shape)
            return a, _lib.as_dp(a)
        c, cp = arr(frame_coeff, (F,))
        r, rp = arr(dR, (N, 3))
        hh, hp = arr(dh, (F, 9))
        grad = np.zeros(self.constant_count())
        self._check(self._lib.ta_constant_gradient(self._handle, cp, rp, hp, _lib.as_dp(grad), len(grad)))
        return grad

    def hessian_vectors(self, dR=None, dh=None, want_virial=False):
        """Analytic directional derivatives of the forces (and virials) of the resident batch
        (`ta_hessian_vectors`): dR [n_dir, N, 3] and / or dh [n_dir, F, 3, 3]; both None = the 3 N unit
        displacements. Returns dF [n_dir, N, 3] (= -H v) and, with `want_virial`, dW [n_dir, F, 3, 3].
        Raises ValueError for models without the analytic path."""
        N, F = int(self.info.n_atoms), int(self.info.n_frames)
        null = C.POINTER(C.c_double)()
        rp = hp = null
        if dR is None and dh is None:
            n_dir = 3 * N
        else:
            n_dir = len(dR) if dR is not None else len(dh)
        if dR is not None:
            dR = np.ascontiguousarray(dR, dtype=np.float64).reshape(n_dir, N, 3)
            rp = _lib.as_dp(dR)
        if dh is not None:
            dh = np.ascontiguousarray(dh, dtype=np.float64).reshape(n_dir, F, 9)
            hp = _lib.as_dp(dh)
        dF = np.zeros((n_dir, N, 3))
        dW = np.zeros((n_dir, F, 3, 3)) if want_virial else None
        # at most 65535 directions per library call; the unit displacements are chunked by `first`
        chunk = 32768 if (dR is None and dh is None) else n_dir
        for first in range(0, n_dir, max(chunk, 1)):
            m = min(chunk, n_dir - first)
            self._check(self._lib.ta_hessian_vectors(
                self._handle, m, first, rp, hp, _lib.as_dp(dF[first:]),
                _lib.as_dp(dW[first:]) if want_virial else null))
        return (dF, dW) if want_virial else dF

    def energies(self, reuse_descriptors=True) -> np.ndarray:
        """Frame energies of the resident batch; with `reuse_descriptors` only the MLP is re-run."""
        want = _lib.TA_WANT_ENERGY | (_lib.TA_WANT_REUSE_DESCRIPTORS if reuse_descriptors else 0)
        self.compute(want)
        return self.fetch(_lib.TA_WANT_ENERGY)["energy"]

    def set_batch_energy_target(self, dst_device_ptr):
        """Later `compute` calls write the batch energy straight to this device address
        (None = the library's own buffer): no copy before a collective."""
        self._check(self._lib.ta_set_batch_energy_target(self._handle, C.c_void_p(int(dst_device_ptr or 0))))

    def eam_tabulate(self, r, rho) -> dict:
        """rho(r), phi(r), F(rho) (and u, w for ADP) of an EAM model on the given abscissae,
        evaluated by the device functions of the energy kernels. Rows: sorted elements; pairs
        a <= b in upper-triangle order."""
        r = np.ascontiguousarray(r, dtype=np.float64).ravel()
        rho = np.ascontiguousarray(rho, dtype=np.float64).ravel()
        nel = len(self._nn.elements)
        npair = nel * (nel + 1) // 2
        adp = getattr(self._nn, "tag", "") == "adp"
        out = {"rho": np.zeros((nel, len(r))), "phi": np.zeros((npair, len(r))),
               "embed": np.zeros((nel, len(rho)))}
        null = C.POINTER(C.c_double)()
        if adp:
            out["u"], out["w"] = np.zeros((npair, len(r))), np.zeros((npair, len(r)))
        self._check(self._lib.ta_eam_tabulate(
            self._handle, len(r), _lib.as_dp(r), len(rho), _lib.as_dp(rho), _lib.as_dp(out["rho"]),
            _lib.as_dp(out["phi"]), _lib.as_dp(out["embed"]),
            _lib.as_dp(out["u"]) if adp else null, _lib.as_dp(out["w"]) if adp else null))
        out["pairs"] = [self._nn.elements[a] + self._nn.elements[b]
                        for a in range(nel) for b in range(a, nel)]
        return out

    def pairs(self):
        P = int(self.info.n_pairs)
        i = np.zeros(max(P, 1), dtype=np.int32)
        j = np.zeros(max(P, 1), dtype=np.int32)
        s = np.zeros((max(P, 1), 3), dtype=np.int32)
        self._check(self._lib.ta_get_pairs(self._handle, _lib.as_ip(i), _lib.as_ip(j), _lib.as_ip(s)))
        return i[:P], j[:P], s[:P]
