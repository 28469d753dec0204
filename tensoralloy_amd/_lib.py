"""
ctypes binding of `libtensoralloy_amd.so` (C ABI: include/tensoralloy_amd.h).

The library is the product: if it is missing or cannot be loaded this module
raises — there is no Python/NumPy fallback for the hot path.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
REPO_DIR = PKG_DIR.parent
CSRC_DIR = PKG_DIR / "csrc"
INCLUDE_DIR = REPO_DIR / "include"
LIB_PATH = PKG_DIR / "libtensoralloy_amd.so"

SOURCES = ["ta_api.hip", "ta_kernels.hip", "ta_kernels_v2.hip", "ta_mlp.hip", "ta_eam.hip",
           "ta_nlist.hip", "ta_grap.hip", "ta_train.hip", "ta_hvp.hip", "ta_neighbor.cpp"]
OBJ_DIR = CSRC_DIR / "build"

TA_OK = 0
TA_ERR_INVALID, TA_ERR_UNSUPPORTED, TA_ERR_HIP, TA_ERR_NOMEM = -1, -2, -3, -4
TA_WANT_ENERGY, TA_WANT_FORCES, TA_WANT_VIRIAL, TA_WANT_ATOMIC, TA_WANT_DESCRIPTORS = 1, 2, 4, 8, 16
TA_WANT_REUSE_DESCRIPTORS = 32
TA_MODEL_SF_MLP, TA_MODEL_EAM_ALLOY, TA_MODEL_EAM_ADP, TA_MODEL_GRAP_MLP = 1, 2, 3, 4
TA_CUTOFF = {"cosine": 0, "polynomial": 1}
TA_ACT = {"relu": 0, "softplus": 1, "tanh": 2, "squareplus": 3, "leaky_relu": 4,
          "sigmoid": 5, "softsign": 6, "elu": 7}
TA_N_KERNEL_SLOTS = 10
TA_ABI_VERSION = 4  # include/tensoralloy_amd.h: TA_ABI_VERSION
KERNEL_SLOTS = ["pair_geometry", "g4_forward", "descriptor_reduce", "mlp", "backward",
                "force_gather", "frame_reduce", "eam", "neighbor_update", "grap_forward"]

# every symbol include/tensoralloy_amd.h declares
EXPORTED_SYMBOLS = [
    "ta_device_count", "ta_create", "ta_destroy", "ta_last_error", "ta_set_frames",
    "ta_compute", "ta_get_results", "ta_eval", "ta_set_stream", "ta_synchronize", "ta_time_compute",
    "ta_batch_energy_device_ptr", "ta_copy_batch_energy", "ta_get_pairs", "ta_neighbor_list", "ta_free",
    "ta_eam_tabulate", "ta_set_batch_energy_target", "ta_param_count", "ta_update_weights",
    "ta_energy_gradient", "ta_measure_hbm_copy", "ta_set_skin", "ta_update_positions", "ta_list_stats",
    "ta_count_contributing_triples", "ta_loss_gradient", "ta_constant_count", "ta_get_constants", "ta_update_constants",
    "ta_constant_gradient", "ta_list_sizes", "ta_abi_version", "ta_model_desc_size", "ta_set_nn_tables", "ta_step", "ta_hessian_vectors", "ta_view_results", "ta_step_view",
]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class ModelDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("n_elements", C.c_int32),
        ("rcut", C.c_double), ("acut", C.c_double),
        ("angular", C.c_int32), ("cutoff_function", C.c_int32),
        ("n_eta", C.c_int32), ("n_omega", C.c_int32), ("n_beta", C.c_int32),
        ("n_gamma", C.c_int32), ("n_zeta", C.c_int32),
        ("eta", _dp), ("omega", _dp), ("beta", _dp), ("gamma", _dp), ("zeta", _dp),
        ("activation", C.c_int32), ("use_resnet_dt", C.c_int32), ("minmax_scale", C.c_int32),
        ("n_layers", _ip), ("layer_sizes", _ip), ("weights", _dp),
        ("xlo", _dp), ("xhi", _dp),
        ("n_eam_params", C.c_int32), ("eam_params", _dp),
        ("eps", C.c_double),
        ("n_grap_params", C.c_int32), ("grap_params", _dp),
        ("n_eam_nets", C.c_int32),
        ("eam_table_n", _ip), ("eam_table_dx", _dp), ("eam_table_coef", _dp),
        ("safe_pow", C.c_int32),
    ]


class Frame(C.Structure):
    _fields_ = [("n_atoms", C.c_int32), ("species", _ip), ("positions", _dp),
                ("cell", _dp), ("pbc", _ip)]


class BatchInfo(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("n_atoms", C.c_int64), ("n_pairs", C.c_int64),
                ("n_triples", C.c_int64), ("nnl_max", C.c_int32), ("descriptor_dim", C.c_int32),
                ("nl_on_device", C.c_int32), ("reserved_", C.c_int32),
                ("nl_ms", C.c_double), ("set_frames_ms", C.c_double)]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _compile_flags():
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread", f"-I{INCLUDE_DIR}",
            f"-I{CSRC_DIR}"] + os.environ.get("TA_EXTRA_HIPCC_FLAGS", "").split()


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every HIP source for gfx950 into the in-tree shared library: one object per
    translation unit, compiled in parallel (only the stale ones), then one link."""
    srcs = [CSRC_DIR / s for s in SOURCES]
    headers = list(CSRC_DIR.glob("*.h")) + [INCLUDE_DIR / "tensoralloy_amd.h"]
    deps = srcs + headers
    stamp = OBJ_DIR / "flags.txt"
    flags_same = stamp.exists() and stamp.read_text() == " ".join(_compile_flags())
    if not force and flags_same and LIB_PATH.exists():
        newest = max(p.stat().st_mtime for p in deps)
        if LIB_PATH.stat().st_mtime >= newest:
            return LIB_PATH
    # several ranks may get here at once (torchrun): one builds, the others wait for the lock and
    # then find the library up to date; the output appears atomically
    import fcntl
    from concurrent.futures import ThreadPoolExecutor
    lock_path = str(LIB_PATH) + ".lock"
    with open(lock_path, "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            OBJ_DIR.mkdir(exist_ok=True)
            flags = _compile_flags()
            flags_same = stamp.exists() and stamp.read_text() == " ".join(flags)
            if not force and flags_same and LIB_PATH.exists() and \
                    LIB_PATH.stat().st_mtime >= max(p.stat().st_mtime for p in deps):
                return LIB_PATH
            if not flags_same:
                force = True
            hdr_time = max(p.stat().st_mtime for p in headers)

            def compile_one(src):
                obj = OBJ_DIR / (src.stem + ".o")
                if not force and obj.exists() and obj.stat().st_mtime >= max(src.stat().st_mtime, hdr_time):
                    return obj
                cmd = [hipcc_path()] + flags + ["-c", str(src), "-o", str(obj)]
                if verbose:
                    print(" ".join(cmd))
                subprocess.run(cmd, check=True)
                return obj

            try:
                jobs = max(1, min(len(srcs), len(os.sched_getaffinity(0))))
            except AttributeError:
                jobs = 4
            with ThreadPoolExecutor(max_workers=jobs) as pool:
                objs = list(pool.map(compile_one, srcs))
            stamp.write_text(" ".join(flags))
            tmp = str(LIB_PATH) + f".tmp{os.getpid()}"
            cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + \
                  [str(o) for o in objs] + ["-o", tmp]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
            os.replace(tmp, LIB_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


_lib = None


def load():
    """Load the shared library (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # TA_LIB_AB=<path of another build of the same ABI>: kernel A/B runs inside one GPU session
    # (box-to-box differences of ~5 % otherwise hide a 2 % kernel change). Never silent, and only a
    # library that reports this binding's ABI version (struct layouts included) is accepted.
    path = os.environ.get("TA_LIB_AB") or str(LIB_PATH)
    if path != str(LIB_PATH):
        import sys
        sys.stderr.write(f"tensoralloy_amd: TA_LIB_AB is set, loading {path} instead of {LIB_PATH}\n")
    lib = C.CDLL(path)
    try:
        lib.ta_abi_version.restype = C.c_int
        version, desc_size = lib.ta_abi_version(), lib.ta_model_desc_size()
    except AttributeError:
        version, desc_size = -1, -1
    if version != TA_ABI_VERSION or desc_size != C.sizeof(ModelDesc):
        raise ImportError(f"{path}: ABI version {version} / ta_model_desc of {desc_size} bytes, this binding "
                          f"is version {TA_ABI_VERSION} / {C.sizeof(ModelDesc)} bytes: rebuild the library")
    H = C.c_void_p
    lib.ta_device_count.restype = C.c_int
    lib.ta_create.argtypes = [C.POINTER(ModelDesc), C.c_int, C.POINTER(H)]
    lib.ta_destroy.argtypes = [H]
    lib.ta_last_error.argtypes = [H]
    lib.ta_last_error.restype = C.c_char_p
    lib.ta_set_frames.argtypes = [H, C.c_int32, C.POINTER(Frame), C.POINTER(BatchInfo)]
    lib.ta_compute.argtypes = [H, C.c_uint32]
    lib.ta_get_results.argtypes = [H, _dp, _dp, _dp, _dp, _dp]
    lib.ta_eval.argtypes = [H, C.c_int32, C.POINTER(Frame), C.c_uint32, _dp, _dp, _dp, _dp]
    lib.ta_synchronize.argtypes = [H]
    lib.ta_set_stream.argtypes = [H, C.c_void_p]
    lib.ta_time_compute.argtypes = [H, C.c_uint32, C.c_int32, C.c_int32, _dp, _dp]
    lib.ta_measure_hbm_copy.argtypes = [H, C.c_int64, C.c_int32, _dp]
    lib.ta_batch_energy_device_ptr.argtypes = [H, C.POINTER(C.c_void_p)]
    lib.ta_copy_batch_energy.argtypes = [H, C.c_void_p]
    lib.ta_set_batch_energy_target.argtypes = [H, C.c_void_p]
    lib.ta_param_count.argtypes = [H, C.POINTER(C.c_int64)]
    lib.ta_update_weights.argtypes = [H, _dp, C.c_int64]
    lib.ta_energy_gradient.argtypes = [H, _dp, _dp, C.c_int64]
    lib.ta_loss_gradient.argtypes = [H, _dp, _dp, _dp, _dp, C.c_int64, _dp]
    lib.ta_constant_count.argtypes = [H, C.POINTER(C.c_int64)]
    lib.ta_get_constants.argtypes = [H, _dp, C.c_int64]
    lib.ta_update_constants.argtypes = [H, _dp, C.c_int64]
    lib.ta_constant_gradient.argtypes = [H, _dp, _dp, _dp, _dp, C.c_int64]
    lib.ta_get_pairs.argtypes = [H, _ip, _ip, _ip]
    lib.ta_neighbor_list.argtypes = [C.POINTER(Frame), C.c_int32, C.c_double,
                                     C.POINTER(C.c_int64), C.POINTER(_ip), C.POINTER(_ip),
                                     C.POINTER(_ip), C.POINTER(_ip)]
    lib.ta_count_contributing_triples.argtypes = [H, C.POINTER(C.c_int64)]
    lib.ta_set_skin.argtypes = [H, C.c_double]
    lib.ta_update_positions.argtypes = [H, _dp, _dp, _ip]
    lib.ta_list_stats.argtypes = [H, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.ta_list_sizes.argtypes = [H, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    lib.ta_set_nn_tables.argtypes = [H, C.c_int]
    lib.ta_hessian_vectors.argtypes = [H, C.c_int32, C.c_int32, _dp, _dp, _dp, _dp]
    lib.ta_step.argtypes = [H, _dp, _dp, C.c_uint32, _dp, _dp, _dp, _dp, _ip]
    _dpp = C.POINTER(_dp)
    lib.ta_view_results.argtypes = [H, C.c_uint32, _dpp, _dpp, _dpp, _dpp]
    lib.ta_step_view.argtypes = [H, _dp, _dp, C.c_uint32, _dpp, _dpp, _dpp, _dpp, _ip]
    lib.ta_free.argtypes = [C.c_void_p]
    lib.ta_eam_tabulate.argtypes = [H, C.c_int32, _dp, C.c_int32, _dp, _dp, _dp, _dp, _dp, _dp]
    lib.ta_free.restype = None
    _lib = lib
    return lib


def check(lib, handle, rc):
    """Map C error codes to the exceptions the reference raises for this path."""
    if rc == TA_OK:
        return
    msg = lib.ta_last_error(handle)
    msg = msg.decode("utf-8", "replace") if msg else f"error {rc}"
    if rc in (TA_ERR_INVALID, TA_ERR_UNSUPPORTED):
        raise ValueError(msg)
    if rc == TA_ERR_NOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def as_dp(a: np.ndarray):
    return a.ctypes.data_as(_dp)


def as_ip(a: np.ndarray):
    return a.ctypes.data_as(_ip)


class FrameArrays:
    """Keeps the NumPy arrays behind a `Frame` alive."""

    def __init__(self, species, positions, cell, pbc):
        self.species = np.ascontiguousarray(species, dtype=np.int32)
        self.positions = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 3)
        self.cell = np.ascontiguousarray(cell, dtype=np.float64).reshape(3, 3)
        self.pbc = np.ascontiguousarray(np.asarray(pbc).astype(bool), dtype=np.int32).reshape(3)
        if len(self.species) != len(self.positions):
            raise ValueError("species and positions disagree on the number of atoms")

    def as_struct(self) -> Frame:
        return Frame(len(self.species), as_ip(self.species), as_dp(self.positions),
                     as_dp(self.cell), as_ip(self.pbc))


def neighbor_list(species, positions, cell, pbc, n_elements, rc):
    """Host-only neighbour list of the library (no GPU needed)."""
    lib = load()
    fa = FrameArrays(species, positions, cell, pbc)
    fr = fa.as_struct()
    n = C.c_int64(0)
    pi, pj, ps, pr = _ip(), _ip(), _ip(), _ip()
    rc_ = lib.ta_neighbor_list(C.byref(fr), n_elements, float(rc), C.byref(n), C.byref(pi),
                               C.byref(pj), C.byref(ps), C.byref(pr))
    check(lib, None, rc_)
    P = n.value
    try:
        i = np.ctypeslib.as_array(pi, shape=(max(P, 1),))[:P].copy()
        j = np.ctypeslib.as_array(pj, shape=(max(P, 1),))[:P].copy()
        s = np.ctypeslib.as_array(ps, shape=(max(3 * P, 1),))[:3 * P].copy().reshape(-1, 3)
        r = np.ctypeslib.as_array(pr, shape=(max(P, 1),))[:P].copy()
    finally:
        for p in (pi, pj, ps, pr):
            lib.ta_free(p)
    return i, j, s, r
