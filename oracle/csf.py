"""
ORACLE (test infrastructure only) — ctypes wrapper of oracle/c/sf_oracle.c.
Only tests, `__graft_entry__.smoke()` and bench.py's cpu_baseline leg use it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from .neighbors import neighbor_list, _complete_cell
from .sf import radial_params, angular_params

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "c", "libsf_oracle.so")

ACT = {"relu": 0, "softplus": 1, "tanh": 2, "squareplus": 3, "leaky_relu": 4, "sigmoid": 5,
       "softsign": 6, "elu": 7}

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class CModel(C.Structure):
    _fields_ = [("n_elements", C.c_int), ("rcut", C.c_double), ("acut", C.c_double),
                ("angular", C.c_int), ("cutoff", C.c_int), ("n_rad", C.c_int), ("n_ang", C.c_int),
                ("eta", _dp), ("omega", _dp), ("beta", _dp), ("gamma", _dp), ("zeta", _dp),
                ("activation", C.c_int), ("use_resnet_dt", C.c_int), ("minmax", C.c_int),
                ("n_layers", _ip), ("layer_sizes", _ip), ("weights", _dp),
                ("xlo", _dp), ("xhi", _dp)]


def build(force=False):
    src = os.path.join(HERE, "c", "sf_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", os.path.join(HERE, "c"), "-B"], check=True,
                       stdout=subprocess.DEVNULL)
    return LIB


_lib = None


def load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        _lib.sf_oracle_eval.argtypes = [C.POINTER(CModel), C.c_int, _ip, _dp, _dp, C.c_long, _ip,
                                        _ip, _ip, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp]
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def make_cmodel(model):
    """oracle.sf.SFModel -> (CModel, keepalive)."""
    keep = []

    def dp(a):
        a = _d(a)
        keep.append(a)
        return a.ctypes.data_as(_dp)

    def ip(a):
        a = np.ascontiguousarray(a, dtype=np.int32)
        keep.append(a)
        return a.ctypes.data_as(_ip)

    rad = np.array(radial_params(model.eta, model.omega)).reshape(-1, 2)
    ang = np.array(angular_params(model.beta, model.gamma, model.zeta)).reshape(-1, 3) \
        if model.angular else np.zeros((0, 3))
    m = CModel()
    m.n_elements = len(model.elements)
    m.rcut, m.acut = model.rcut, model.acut
    m.angular = int(model.angular)
    m.cutoff = 0 if model.cutoff_function == "cosine" else 1
    m.n_rad, m.n_ang = len(rad), len(ang)
    m.eta, m.omega = dp(rad[:, 0]), dp(rad[:, 1])
    m.beta, m.gamma, m.zeta = dp(ang[:, 0]), dp(ang[:, 1]), dp(ang[:, 2])
    m.activation = ACT[model.activation.lower()]
    m.use_resnet_dt = int(model.use_resnet_dt)
    D = model.ndim
    if model.weights is not None:
        n_layers, sizes, flat = [], [], []
        for el in model.elements:
            layers = model.weights[el]
            n_layers.append(len(layers))
            s = [D]
            for w, b in layers:
                s.append(w.shape[1])
                flat.append(_d(w).ravel())
                flat.append(np.zeros(w.shape[1]) if b is None else _d(b).ravel())
            sizes.extend(s)
        m.n_layers, m.layer_sizes = ip(n_layers), ip(sizes)
        m.weights = dp(np.concatenate(flat))
    m.minmax = int(model.minmax is not None)
    if model.minmax is not None:
        m.xlo = dp(np.concatenate([np.ravel(model.minmax[el][0]) for el in model.elements]))
        m.xhi = dp(np.concatenate([np.ravel(model.minmax[el][1]) for el in model.elements]))
    return m, keep


def prepare(model, symbols, positions, cell, pbc):
    """Neighbour list + arrays (outside any timed region)."""
    R = _d(positions).reshape(-1, 3)
    h = _d(_complete_cell(cell, pbc))
    rmax = max(model.rcut, model.acut) if model.angular else model.rcut
    i, j, S = neighbor_list(R, h, pbc, rmax)
    species = np.array([model.elements.index(s) for s in symbols], dtype=np.int32)
    return dict(R=R, h=h, i=np.ascontiguousarray(i, dtype=np.int32),
                j=np.ascontiguousarray(j, dtype=np.int32),
                S=np.ascontiguousarray(S, dtype=np.int32), species=species)


def run(model, prep, want_forces=True, nthreads=0, cmodel=None):
    lib = load()
    if cmodel is None:
        cmodel = make_cmodel(model)
    m, keep = cmodel
    N = len(prep["species"])
    D = model.ndim
    energy = C.c_double(0.0)
    atomic, forces, virial = np.zeros(N), np.zeros((N, 3)), np.zeros((3, 3))
    desc = np.zeros((N, D))
    rc = lib.sf_oracle_eval(C.byref(m), N, prep["species"].ctypes.data_as(_ip),
                            prep["R"].ctypes.data_as(_dp), prep["h"].ctypes.data_as(_dp),
                            len(prep["i"]), prep["i"].ctypes.data_as(_ip),
                            prep["j"].ctypes.data_as(_ip), prep["S"].ctypes.data_as(_ip),
                            int(want_forces), int(nthreads), C.byref(energy),
                            atomic.ctypes.data_as(_dp), forces.ctypes.data_as(_dp),
                            virial.ctypes.data_as(_dp), desc.ctypes.data_as(_dp))
    if rc != 0:
        raise MemoryError("sf_oracle_eval failed")
    return dict(energy=energy.value, atomic=atomic, forces=forces, virial=virial,
                descriptors=desc)


def evaluate(model, symbols, positions, cell, pbc, want_forces=True, nthreads=0):
    return run(model, prepare(model, symbols, positions, cell, pbc), want_forces, nthreads)
