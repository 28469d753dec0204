"""
Host-side model descriptions that mirror the reference's model classes for
the hot path (names, constructor arguments and `as_dict()` contents match), and
the model-file format that replaces the frozen TensorFlow `.pb`.

  SymmetryFunction  <- reference tensoralloy/nn/atomic/sf.py:26-77
  AtomicNN          <- reference tensoralloy/nn/atomic/atomic.py:60-132 (+ BasicNN,
                       tensoralloy/nn/basic.py:99-160)
  AtomicNN.export   <- reference tensoralloy/nn/basic.py:1017-1153: the JSON
                       constants `Transformer/params` and `Metadata/*` become
                       `<name>.json`, the frozen variables become `<name>.npz`
                       with the key naming of `export_to_lammps_native`
                       (atomic.py:452-478): `weights_{element}_{layer}`,
                       `biases_{element}_{layer}`, plus `xlo_{element}`,
                       `xhi_{element}` for the min-max variables.

These classes hold parameters only; all arithmetic happens in the HIP library.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from datetime import datetime
from typing import Dict, List, Optional, Sequence, Union

import numpy as np

from . import _lib
from .utils import Defaults, get_kbody_terms, parameter_grid

API_VERSION = "1.1"  # reference nn/basic.py:43

#: properties the reference can export (nn/basic.py:74-92)
EXPORTABLE_PROPERTIES = ["energy", "eentropy", "free_energy", "atomic", "forces", "stress",
                         "total_pressure", "hessian", "elastic"]
#: properties this build computes
SUPPORTED_PROPERTIES = ["energy", "atomic", "forces", "stress", "total_pressure"]


def _safe_select(a, b):
    if a is None:
        return b
    if hasattr(a, "__len__") and len(a) == 0:
        return b
    return a


class SymmetryFunction:
    """Behler G2 / G4 symmetry-function descriptor parameters."""

    def __init__(self, elements: Sequence[str], eta=Defaults.eta, omega=Defaults.omega,
                 beta=Defaults.beta, gamma=Defaults.gamma, zeta=Defaults.zeta,
                 cutoff_function="cosine"):
        self._elements = sorted(list(elements))
        self._eta = np.asarray(eta, dtype=np.float64).ravel()
        self._omega = np.asarray(omega, dtype=np.float64).ravel()
        self._beta = np.asarray(beta, dtype=np.float64).ravel()
        self._gamma = np.asarray(gamma, dtype=np.float64).ravel()
        self._zeta = np.asarray(zeta, dtype=np.float64).ravel()
        if cutoff_function not in _lib.TA_CUTOFF:
            # the reference silently uses the polynomial cutoff for any other
            # string (sf.py:70-77); refuse instead of guessing
            raise ValueError(f"Unknown cutoff function: {cutoff_function}")
        self._cutoff_function = cutoff_function
        self._radial_parameters = parameter_grid(eta=self._eta, omega=self._omega)
        self._angular_parameters = parameter_grid(beta=self._beta, gamma=self._gamma,
                                                  zeta=self._zeta)

    @property
    def name(self):
        return "SF"

    @property
    def elements(self):
        return self._elements

    @property
    def radial_parameters(self):
        return self._radial_parameters

    @property
    def angular_parameters(self):
        return self._angular_parameters

    @property
    def cutoff_function(self):
        return self._cutoff_function

    def as_dict(self):
        return {"class": self.__class__.__name__, "elements": self._elements,
                "eta": self._eta.tolist(), "omega": self._omega.tolist(),
                "gamma": self._gamma.tolist(), "zeta": self._zeta.tolist(),
                "beta": self._beta.tolist(), "cutoff_function": self._cutoff_function}

    def ndim(self, angular: bool) -> int:
        n = len(self._elements)
        d = n * len(self._radial_parameters)
        if angular:
            d += n * (n + 1) // 2 * len(self._angular_parameters)
        return d


class AtomicNN:
    """
    Per-element MLP on symmetry-function descriptors: parameter container with
    the constructor of the reference `AtomicNN`.
    """

    scope = "Atomic"

    def __init__(self, elements: Sequence[str], descriptor: Union[SymmetryFunction, dict],
                 hidden_sizes=None, activation=None, kernel_initializer="he_normal",
                 minmax_scale=True, use_resnet_dt=False, atomic_static_energy=None,
                 use_atomic_static_energy=True, fixed_atomic_static_energy=False,
                 minimize_properties=("energy", "forces"),
                 export_properties=("energy", "forces")):
        self._elements = sorted(list(elements))
        self._hidden_sizes = self._get_hidden_sizes(
            _safe_select(hidden_sizes, Defaults.hidden_sizes))
        self._activation = _safe_select(activation, Defaults.activation)
        if self._activation.lower() not in _lib.TA_ACT:
            raise ValueError(
                f"The activation function '{self._activation}' cannot be recognized!")
        for prop in export_properties:
            if prop not in EXPORTABLE_PROPERTIES:
                raise ValueError(f"'{prop}' is not an exportable property.")
        self._kernel_initializer = kernel_initializer
        self._minmax_scale = bool(minmax_scale)
        self._use_resnet_dt = bool(use_resnet_dt)
        self._atomic_static_energy = dict(atomic_static_energy or {})
        self._use_atomic_static_energy = bool(use_atomic_static_energy)
        self._fixed_atomic_static_energy = bool(fixed_atomic_static_energy)
        self._minimize_properties = list(minimize_properties)
        self._export_properties = list(export_properties)
        if isinstance(descriptor, dict):
            d = dict(descriptor)
            cls = d.pop("class", None) or d.get("@class", "SymmetryFunction")
            d.pop("@module", None)
            d.pop("@class", None)
            if cls == "SymmetryFunction":
                descriptor = SymmetryFunction(**d)
            elif cls == "GenericRadialAtomicPotential":
                from .grap import GenericRadialAtomicPotential
                descriptor = GenericRadialAtomicPotential(**d)
            else:
                raise ValueError(f"Unsupported descriptor: {cls}")
        self._descriptor = descriptor
        self._transformer = None
        self.precision = "high"  # 'medium' = float32 reference model: eps 1e-8, float32 results
        #: {element: [(W [in, out], b [out] | None), ...]} last entry = output layer
        self.weights: Dict[str, List] = {}
        #: {element: (xlo [D], xhi [D])}
        self.minmax: Dict[str, tuple] = {}

    def _get_hidden_sizes(self, hidden_sizes) -> Dict[str, List[int]]:
        if isinstance(hidden_sizes, dict):
            out = {}
            for el in self._elements:
                v = hidden_sizes.get(el, Defaults.hidden_sizes)
                out[el] = [int(x) for x in np.atleast_1d(v)]
            return out
        sizes = [int(x) for x in np.atleast_1d(hidden_sizes)]
        return {el: list(sizes) for el in self._elements}

    # -- reference-compatible surface ------------------------------------
    @property
    def elements(self):
        return self._elements

    @property
    def hidden_sizes(self):
        return self._hidden_sizes

    @property
    def descriptor(self):
        return self._descriptor

    @property
    def predict_properties(self):
        return self._export_properties

    @property
    def variational_energy(self):
        return "energy"

    @property
    def is_finite_temperature(self):
        return False

    @property
    def transformer(self):
        return self._transformer

    def attach_transformer(self, clf):
        self._transformer = clf

    def as_dict(self):
        return {"class": self.__class__.__name__, "elements": self._elements,
                "hidden_sizes": self._hidden_sizes, "activation": self._activation,
                "kernel_initializer": self._kernel_initializer,
                "minmax_scale": self._minmax_scale, "use_resnet_dt": self._use_resnet_dt,
                "use_atomic_static_energy": self._use_atomic_static_energy,
                "fixed_atomic_static_energy": self._fixed_atomic_static_energy,
                "atomic_static_energy": self._atomic_static_energy,
                "minimize_properties": self._minimize_properties,
                "export_properties": self._export_properties,
                "descriptor": self._descriptor.as_dict()}

    # -- weights -----------------------------------------------------------
    def ndim(self) -> int:
        if self._transformer is None:
            raise ValueError("A descriptor transformer must be attached.")
        return self._descriptor.ndim(self._transformer.angular)

    def ndim_angular(self) -> int:
        return self.ndim() - self._descriptor.ndim(False)

    def initialize(self, seed=Defaults.seed, bias_scale=0.0):
        """
        Random-init variables the way the reference initialises them: He-normal
        kernels (sigma = sqrt(2 / fan_in), truncated at 2 sigma,
        nn/init_ops.py:20-30), zero biases, output bias = atomic static energy
        (atomic.py:236-259), xlo = 1000 / xhi = 0 (atomic.py:176-177).
        """
        rng = np.random.RandomState(seed)
        D = self.ndim()
        for el in self._elements:
            sizes = [D] + list(self._hidden_sizes[el]) + [1]
            layers = []
            for l in range(len(sizes) - 1):
                fan_in, fan_out = sizes[l], sizes[l + 1]
                sigma = np.sqrt(2.0 / fan_in)
                w = rng.normal(0.0, sigma, size=(fan_in, fan_out))
                bad = np.abs(w) > 2 * sigma
                while bad.any():
                    w[bad] = rng.normal(0.0, sigma, size=int(bad.sum()))
                    bad = np.abs(w) > 2 * sigma
                last = l == len(sizes) - 2
                if last:
                    b = (np.full(1, float(self._atomic_static_energy.get(el, 0.0)))
                         if self._use_atomic_static_energy else None)
                else:
                    b = bias_scale * rng.normal(size=fan_out) if bias_scale else np.zeros(fan_out)
                layers.append((w, b))
            self.weights[el] = layers
            if self._minmax_scale:
                self.minmax[el] = (np.full(D, 1000.0), np.zeros(D))

    # -- model file ----------------------------------------------------------
    def export(self, output_graph_path: str, **_ignored):
        """
        Write `<stem>.json` + `<stem>.npz`. `output_graph_path` may end in
        `.json`, `.npz`, `.pb` (the reference's extension) or nothing.
        """
        if self._transformer is None:
            raise ValueError("A transformer must be attached before exporting to a pb file.")
        if not self.weights:
            raise ValueError("The model has no weights: call initialize() or set .weights")
        stem = _model_stem(output_graph_path)
        props = {"energy": "Output/Energy/energy:0", "energy/atom": "Output/Energy/atomic:0"}
        want = set(self._export_properties)
        if want & {"forces", "stress", "total_pressure"}:
            props["forces"] = "Output/Forces/forces:0"
        if "stress" in want:
            props["stress"] = "Output/Stress/Voigt/stress:0"
            props["virial"] = "Output/Stress/Full/virial:0"
            props["total_pressure"] = "Output/Stress/pressure/GPa:0"
        # second derivatives: central differences of the analytic forces / virial (calculator.py)
        if "hessian" in want:
            props["hessian"] = "Output/Hessian/hessian:0"
        if "elastic" in want:
            props["elastic"] = "Output/Elastic/Cijkl/elastic:0"
        meta = {
            "format": "tensoralloy_amd/1",
            "Transformer/params": self._transformer.as_dict(),
            "Metadata/timestamp": str(datetime.today()),
            "Metadata/precision": self.precision,
            "Metadata/variational_energy": self.variational_energy,
            "Metadata/is_finite_temperature": 0,
            "Metadata/api": API_VERSION,
            "Metadata/ops": props,
            "nn": self.as_dict(),
            "weights": os.path.basename(stem) + ".npz",
        }
        data = {}
        for i, el in enumerate(self._elements):
            for j, (w, b) in enumerate(self.weights[el]):
                data[f"weights_{i}_{j}"] = np.asarray(w, dtype=np.float64)
                if b is not None:
                    data[f"biases_{i}_{j}"] = np.asarray(b, dtype=np.float64)
            if self._minmax_scale:
                xlo, xhi = self.minmax[el]
                data[f"xlo_{i}"] = np.asarray(xlo, dtype=np.float64)
                data[f"xhi_{i}"] = np.asarray(xhi, dtype=np.float64)
        for j, (w, b) in enumerate(getattr(self._descriptor, "filter_weights", None) or []):
            data[f"fnn::weights_0_{j}"] = np.asarray(w, dtype=np.float64)   # GRAP `nn` filter network
            if b is not None:
                data[f"fnn::biases_0_{j}"] = np.asarray(b, dtype=np.float64)
        np.savez(stem + ".npz", **data)
        with open(stem + ".json", "w") as fp:
            json.dump(meta, fp, indent=1)
        return stem + ".json"

    def export_to_lammps_native(self, model_path: str, dtype=np.float64):
        """
        Write the `.npz` the reference exports for its LAMMPS pair style
        (`AtomicNN.export_to_lammps_native`, atomic.py:304-480; the only TF-free weight interchange
        format the reference defines): same keys, dtypes and conventions. GRAP descriptors only.
        `load_model` / `TensorAlloyCalculator` read such a file directly.
        """
        if getattr(self._descriptor, "name", "") != "GRAP":
            raise ValueError("The descriptor GenericRadialAtomicPotential is required")
        if self._transformer is None:
            raise ValueError("A transformer must be attached before exporting to a pb file.")
        if self._minmax_scale:
            raise ValueError("the native format has no slot for min-max scaling (atomic.py:360-478)")
        from .atoms import atomic_masses, atomic_numbers
        sizes = list(self._hidden_sizes[self._elements[0]])
        for el in self._elements[1:]:
            if list(self._hidden_sizes[el]) != sizes:
                raise ValueError("Layer sizes of all elements must be the same")
        layer_sizes = np.array(sizes + [1], dtype=np.int32)
        actfn_map = {"relu": 0, "softplus": 1, "tanh": 2, "squareplus": 3}
        if self._activation.lower() not in actfn_map:
            raise ValueError(f"activation '{self._activation}' has no code in the native format")
        chars = []
        for el in self._elements:
            chars.extend([ord(el[0]), 0] if len(el) == 1 else [ord(c) for c in el])
        clf, gd = self._transformer, self._descriptor
        data = {"rmax": dtype(clf.rcut), "nelt": np.int32(len(self._elements)),
                "masses": np.array([atomic_masses[atomic_numbers[el]] for el in self._elements], dtype=dtype),
                "numbers": np.array(chars, dtype=np.int32), "tdnp": np.int32(0),
                "precision": np.int32(64 if dtype == np.float64 else 32), "use_fnn": np.int32(0)}
        algo = gd.algorithm.as_dict(convert_to_pairs=True)
        if gd.algorithm.name == "nn":  # atomic.py:408-438
            if not gd.filter_weights:
                raise ValueError("GRAP/nn: the descriptor has no filter weights")
            if algo["activation"].lower() not in actfn_map:
                raise ValueError(f"activation '{algo['activation']}' has no code in the native format")
            data["use_fnn"] = np.int32(1)
            data["fnn::nlayers"] = np.int32(len(algo["hidden_sizes"]) + 1)
            data["fnn::layer_sizes"] = np.array(list(algo["hidden_sizes"]) + [algo["num_filters"]], dtype=np.int32)
            data["fnn::num_filters"] = np.int32(algo["num_filters"])
            data["fnn::actfn"] = np.int32(actfn_map[algo["activation"].lower()])
            data["fnn::use_resnet_dt"] = np.int32(algo["use_resnet_dt"])
            data["fnn::apply_output_bias"] = np.int32(0)
            data["fnn::h_abck_modifier"] = np.int32(algo["h_abck_modifier"])
            for j, (w, b) in enumerate(gd.filter_weights):
                data[f"fnn::weights_0_{j}"] = np.squeeze(np.asarray(w, dtype=dtype))
                if j < len(gd.filter_weights) - 1:
                    data[f"fnn::biases_0_{j}"] = (np.zeros(np.shape(w)[1], dtype=dtype) if b is None
                                                  else np.asarray(b, dtype=dtype))
        else:
            method = {"pexp": 0, "morse": 1, "density": 2, "sf": 3}[gd.algorithm.name]
            data["descriptor::method"] = np.int32(method)
            for key, values in algo["parameters"].items():
                data[f"descriptor::{key}"] = np.array(values, dtype=dtype)
        data["nlayers"] = np.int32(len(layer_sizes))
        data["max_moment"] = np.int32(gd.max_moment)
        data["actfn"] = np.int32(actfn_map[self._activation.lower()])
        data["fctype"] = np.int32({"cosine": 0, "polynomial": 1}[gd.cutoff_function])
        data["layer_sizes"] = layer_sizes
        data["use_resnet_dt"] = np.int32(self._use_resnet_dt)
        data["apply_output_bias"] = np.int32(self._use_atomic_static_energy)
        data["is_T_symmetric"] = np.int32(gd.is_T_symmetric)
        for i, el in enumerate(self._elements):
            layers = self.weights[el]
            for j, (w, b) in enumerate(layers[:-1]):
                data[f"weights_{i}_{j}"] = np.asarray(w, dtype=dtype)
                data[f"biases_{i}_{j}"] = (np.zeros(np.shape(w)[1], dtype=dtype) if b is None
                                           else np.asarray(b, dtype=dtype))
            wo, bo = layers[-1]
            data[f"weights_{i}_{len(layers) - 1}"] = np.asarray(wo, dtype=dtype).ravel()
            if self._use_atomic_static_energy:
                data[f"biases_{i}_{len(layers) - 1}"] = (np.zeros(1, dtype=dtype) if bo is None
                                                         else np.asarray(bo, dtype=dtype).ravel())
        np.savez(model_path, **data)
        return model_path if str(model_path).endswith(".npz") else str(model_path) + ".npz"

    # -- C ABI -----------------------------------------------------------------
    def to_desc(self):
        """Flatten into a `ta_model_desc`; returns (desc, keepalive list)."""
        clf = self._transformer
        if clf is None:
            raise ValueError("A descriptor transformer must be attached.")
        is_grap = getattr(self._descriptor, "name", "SF") == "GRAP"
        if is_grap and clf.angular:
            raise ValueError("GRAP is a radial descriptor: the transformer must have angular=False")
        nonsym = bool(clf.angular and not clf.symmetric)
        if nonsym and len(self._elements) > 1:
            # the reference sizes its output list for the symmetric term count (sf.py:131-132),
            # so its SymmetryFunction also fails (IndexError) for n >= 2 elements
            raise ValueError("symmetric=False angular symmetry functions exist only for one element "
                             "(reference nn/atomic/sf.py:131-132 holds n(n+1)/2 angular terms)")
        sf = self._descriptor
        D = self.ndim()
        keep = []

        def dptr(a):
            a = np.ascontiguousarray(a, dtype=np.float64)
            keep.append(a)
            return _lib.as_dp(a)

        def iptr(a):
            a = np.ascontiguousarray(a, dtype=np.int32)
            keep.append(a)
            return _lib.as_ip(a)

        # symmetric=False with one element: every {j, k} is listed as (j, k) and (k, j)
        # (universal.py:183-203), so the angular features are exactly twice the symmetric
        # ones. The library always sums unordered triples; the factor is folded into the
        # first layer (or into xlo / xhi when min-max scaling comes first).
        n_rad = D - self.ndim_angular() if nonsym else D
        n_layers, sizes, flat = [], [], []
        for el in self._elements:
            layers = self.weights[el]
            n_layers.append(len(layers))
            s = [D]
            for k, (w, b) in enumerate(layers):
                w = np.asarray(w, dtype=np.float64)
                if w.ndim != 2 or w.shape[0] != s[-1]:
                    raise ValueError(f"weight shape {w.shape} does not chain from {s[-1]}")
                if nonsym and k == 0 and not self._minmax_scale:
                    w = w.copy()
                    w[n_rad:] *= 2.0
                s.append(w.shape[1])
                flat.append(w.ravel())
                flat.append(np.zeros(w.shape[1]) if b is None
                            else np.asarray(b, dtype=np.float64).ravel())
            sizes.extend(s)
        desc = _lib.ModelDesc()
        desc.n_elements = len(self._elements)
        desc.rcut = float(clf.rcut)
        desc.acut = float(clf.acut if clf.acut is not None else clf.rcut)
        desc.cutoff_function = _lib.TA_CUTOFF[sf.cutoff_function]
        if is_grap:
            desc.kind = _lib.TA_MODEL_GRAP_MLP
            desc.angular = 0
            gp = sf.flat_parameters()
            desc.n_grap_params = len(gp)
            desc.grap_params = dptr(gp)
        else:
            desc.kind = _lib.TA_MODEL_SF_MLP
            desc.angular = int(bool(clf.angular))
            desc.n_eta, desc.n_omega = len(sf._eta), len(sf._omega)
            desc.n_beta, desc.n_gamma, desc.n_zeta = len(sf._beta), len(sf._gamma), len(sf._zeta)
            desc.eta, desc.omega = dptr(sf._eta), dptr(sf._omega)
            desc.beta, desc.gamma, desc.zeta = dptr(sf._beta), dptr(sf._gamma), dptr(sf._zeta)
        desc.activation = _lib.TA_ACT[self._activation.lower()]
        desc.use_resnet_dt = int(self._use_resnet_dt)
        desc.minmax_scale = int(self._minmax_scale)
        desc.n_layers = iptr(n_layers)
        desc.layer_sizes = iptr(sizes)
        desc.weights = dptr(np.concatenate(flat))
        if self._minmax_scale:
            scale = np.ones(D)
            if nonsym:
                scale[n_rad:] = 0.5
            desc.xlo = dptr(np.concatenate([np.ravel(self.minmax[el][0]) * scale for el in self._elements]))
            desc.xhi = dptr(np.concatenate([np.ravel(self.minmax[el][1]) * scale for el in self._elements]))
        desc.n_eam_params = 0
        desc.eps = 1e-8 if self.precision == "medium" else 1e-14  # precision.py:113-114
        desc.safe_pow = int(self.use_custom_pow)
        return desc, keep

    @property
    def use_custom_pow(self) -> bool:
        """Which `safe_pow` the model runs with. The reference picks it at import time from the
        environment variable TENSORALLOY_USE_CUSTOM_POW (extension/grad_ops.py:16); the same variable
        is read here when the engine is created, and `nn.use_custom_pow = True / False` overrides."""
        import os
        override = getattr(self, "_use_custom_pow", None)
        return bool(os.environ.get("TENSORALLOY_USE_CUSTOM_POW", False)) if override is None else override

    @use_custom_pow.setter
    def use_custom_pow(self, value):
        self._use_custom_pow = None if value is None else bool(value)

    def descriptor_scale(self):
        """Factors that turn the library's (symmetric) descriptors into this model's."""
        clf = self._transformer
        scale = np.ones(self.ndim())
        if clf is not None and clf.angular and not clf.symmetric:
            scale[self.ndim() - self.ndim_angular():] = 2.0
        return scale


def _model_stem(path: str) -> str:
    path = str(path)
    for ext in (".json", ".npz", ".pb.gz", ".pb"):
        if path.endswith(ext):
            return path[: -len(ext)]
    return path


_ACT_OPS = {"Softplus": "softplus", "Relu": "relu", "Tanh": "tanh", "LeakyRelu": "leaky_relu",
             "Sigmoid": "sigmoid", "Softsign": "softsign", "Elu": "elu"}


def load_graph_def(path: str):
    """
    Load a model from the reference's frozen TensorFlow GraphDef (`BasicNN.export`,
    basic.py:1075-1092; what `TensorAlloyCalculator.__init__` reads, calculator.py:128-170) without
    TensorFlow: `tensoralloy_amd.graphdef` walks the protobuf and hands back the `Const` nodes.
    Returns (nn, transformer, metadata) like `load_model`.

    * Empirical EAM models (the Zjw04 family): the constants are the frozen shared variables
      `EAM/Shared/<El>/<name>` (potentials.py:129-200). Checked against the reference's own fixtures
      `test_files/models/{Ni,Mo}.zhou04.pb`.
    * `AtomicNN` with symmetry functions: weights `Atomic/<El>/Conv1d{j}/{kernel,bias}`,
      `Atomic/<El>/Output/{kernel,bias}` (convolutional.py:154-300), min-max bounds
      `Atomic/<El>/MinMax/{xlo,xhi}`, the (eta, omega) / (beta, gamma, zeta) grids from the scalar
      constants `.../G2/<tau>/{eta,omega}` and `.../G4/<tau>/{beta,gamma,zeta}` (sf.py:96-101,
      :156-162), activation and cutoff from the op types. The reference ships no such file, so this
      branch is exercised only by its structure; anything it cannot identify raises ValueError.
    """
    from .graphdef import read_graph_model, read_node_ops
    from .transformer import UniversalTransformer
    meta_s, consts = read_graph_model(path)
    params = json.loads(meta_s["Transformer/params"])
    cls = params.pop("class", "UniversalTransformer")
    params.pop("predict_properties", None)
    if cls != "UniversalTransformer":
        raise ValueError(f"Unsupported transformer: {cls}")  # calculator.py:142
    clf = UniversalTransformer(**params)
    ops = json.loads(meta_s.get("Metadata/ops", "{}"))
    meta = {"format": "graphdef", "Transformer/params": dict(params, **{"class": cls}),
            "Metadata/ops": ops,
            "Metadata/precision": meta_s.get("Metadata/precision", "high"),
            "Metadata/timestamp": meta_s.get("Metadata/timestamp"),
            "Metadata/tf_version": meta_s.get("Metadata/tf_version"),
            "Metadata/api": meta_s.get("Metadata/api", "1.0"),
            "Metadata/variational_energy": meta_s.get("Metadata/variational_energy", "energy"),
            "Metadata/is_finite_temperature": int(meta_s.get("Metadata/is_finite_temperature", 0) or 0)}
    if meta["Metadata/is_finite_temperature"]:
        raise ValueError(f"{path}: finite-temperature models are not implemented by tensoralloy_amd")
    export = [k for k, v in ops.items() if str(v).endswith(":0") and k in EXPORTABLE_PROPERTIES]
    names = list(consts)
    if any(n.startswith("EAM/") or n.startswith("ADP/") for n in names):
        from .eam import EamAlloyNN
        if any("/Dipole/" in n or "/Quadrupole/" in n or n.startswith("ADP/") for n in names):
            raise ValueError(f"{path}: ADP graphs are not read from GraphDef files; export the model "
                             f"as <name>.json + <name>.npz")
        if not any("/Zjw04" in n for n in names):
            raise ValueError(f"{path}: only empirical Zjw04-family EAM graphs are read from GraphDef "
                             f"files (no '/Zjw04*/' scope found)")
        family = "zjw04"
        for tag, key in (("/Zjw04xcp/", "zjw04xcp"), ("/Zjw04uxc/", "zjw04uxc"), ("/Zjw04xc/", "zjw04xc")):
            if any(tag in n for n in names):
                family = key
        parameters = {}
        for n, v in consts.items():
            if n.startswith("EAM/Shared/"):
                _, _, sec, key = n.split("/", 3)
                parameters.setdefault(sec, {})[key] = float(np.asarray(v))
        missing = [el for el in clf.elements if el not in parameters]
        if missing:
            raise ValueError(f"{path}: no EAM/Shared constants for {missing}")
        nn = EamAlloyNN(clf.elements, custom_potentials=family, parameters=parameters,
                        export_properties=export or ("energy", "forces", "stress"))
        nn.attach_transformer(clf)
        nn.precision = meta["Metadata/precision"]
        return nn, clf, meta
    if not any(n.startswith("Atomic/") for n in names):
        raise ValueError(f"{path}: neither an EAM nor an AtomicNN graph")
    # AtomicNN graphs: the reference ships no such frozen file, so what follows reads the STRUCTURE the
    # reference's exporter writes (scopes, constant names, which ops appear) but has no fixture that
    # pins it end to end. It refuses whatever it cannot identify unambiguously and says what it assumed.
    import warnings
    warnings.warn(f"{path}: AtomicNN GraphDef import is structural (cutoff, activation, ResNet and bias flags "
                  f"are inferred from the graph's ops); check a known energy, or export the model as "
                  f"<name>.json + <name>.npz", RuntimeWarning, stacklevel=2)
    node_ops = read_node_ops(path)
    if any("/Filters/" in n or "/Moment" in n for n in names):
        raise ValueError(f"{path}: GRAP graphs are not read from GraphDef files; use the native .npz "
                         f"(export_to_lammps_native)")

    def grid(kind, key):
        vals = {}
        for n, v in consts.items():
            parts = n.split("/")
            if len(parts) >= 3 and parts[-1] == key and parts[-3] == kind and parts[-2].isdigit():
                vals.setdefault(int(parts[-2]), float(np.asarray(v)))
        return [vals[k] for k in sorted(vals)]

    def unique(seq):
        out = []
        for x in seq:
            if x not in out:
                out.append(x)
        return out
    eta, omega = unique(grid("G2", "eta")), unique(grid("G2", "omega"))
    if not eta or not omega:
        raise ValueError(f"{path}: no G2 parameter constants found")
    kw = dict(eta=eta, omega=omega)
    if clf.angular:
        kw.update(beta=unique(grid("G4", "beta")), gamma=unique(grid("G4", "gamma")),
                  zeta=unique(grid("G4", "zeta")))
        if not (kw["beta"] and kw["gamma"] and kw["zeta"]):
            raise ValueError(f"{path}: angular transformer but no G4 parameter constants")
    cutoff = "cosine" if any(op == "Cos" for op in node_ops.values()) else "polynomial"
    acts = unique(_ACT_OPS[op] for n, op in node_ops.items() if n.startswith("Atomic/") and op in _ACT_OPS)
    if len(acts) != 1:
        raise ValueError(f"{path}: cannot identify the activation function (found {acts})")
    weights, minmax = {}, {}
    for el in clf.elements:
        # hidden layers are scoped `Conv1d1`, `Conv1d2`, ... by the reference (`Conv{rank}d{j + 1}`,
        # convolutional.py:262); a count from 0 is accepted as well
        layers, j = [], (0 if f"Atomic/{el}/Conv1d0/kernel" in consts else 1)
        while f"Atomic/{el}/Conv1d{j}/kernel" in consts:
            w = np.asarray(consts[f"Atomic/{el}/Conv1d{j}/kernel"], dtype=np.float64)
            b = consts.get(f"Atomic/{el}/Conv1d{j}/bias")
            layers.append((w.reshape(w.shape[-2], w.shape[-1]), None if b is None else np.asarray(b, np.float64).ravel()))
            j += 1
        wo = consts.get(f"Atomic/{el}/Output/kernel")
        if wo is None or not layers:
            raise ValueError(f"{path}: weights of element {el} not found")
        wo = np.asarray(wo, dtype=np.float64)
        bo = consts.get(f"Atomic/{el}/Output/bias")
        layers.append((wo.reshape(wo.shape[-2], wo.shape[-1]), None if bo is None else np.asarray(bo, np.float64).ravel()))
        weights[el] = layers
        if f"Atomic/{el}/MinMax/xlo" in consts:
            minmax[el] = (np.asarray(consts[f"Atomic/{el}/MinMax/xlo"], np.float64).ravel(),
                          np.asarray(consts[f"Atomic/{el}/MinMax/xhi"], np.float64).ravel())
    hidden = {el: [w.shape[1] for w, _ in weights[el][:-1]] for el in clf.elements}
    resnet = any(op in ("Add", "AddV2") and "/Conv1d" in n and n.startswith("Atomic/") and "BiasAdd" not in n
                 for n, op in node_ops.items())
    nn = AtomicNN(clf.elements, SymmetryFunction(clf.elements, cutoff_function=cutoff, **kw),
                  hidden_sizes=hidden, activation=acts[0], minmax_scale=bool(minmax), use_resnet_dt=resnet,
                  use_atomic_static_energy=all(l[-1][1] is not None for l in weights.values()),
                  export_properties=export or ("energy", "forces", "stress"))
    nn.attach_transformer(clf)
    nn.precision = meta["Metadata/precision"]
    for el in clf.elements:
        if weights[el][0][0].shape[0] != nn.ndim():
            raise ValueError(f"{path}: the first layer of {el} takes {weights[el][0][0].shape[0]} inputs, "
                             f"the recovered descriptor has {nn.ndim()}")
        nn.weights[el] = weights[el]
        if minmax:
            nn.minmax[el] = minmax[el]
    return nn, clf, meta


def load_lammps_native(path: str):
    """
    Read a model in the reference's native `.npz` format (`export_to_lammps_native`,
    atomic.py:304-480). Returns (nn, transformer, metadata dict) like `load_model`.
    The format carries the moments 0..max_moment of the non-legacy GRAP (atomic.py:441, grap.py:606).
    """
    from .grap import GenericRadialAtomicPotential
    from .transformer import UniversalTransformer
    npz = np.load(path)
    if int(npz["tdnp"]) != 0:
        raise ValueError(f"{path}: temperature-dependent models are not implemented by tensoralloy_amd")
    chars = np.asarray(npz["numbers"], dtype=int).reshape(-1, 2)
    elements = ["".join(chr(c) for c in row if c) for row in chars]
    if int(npz["use_fnn"]) != 0:   # the `nn` filter network: hyper-parameters and weights from fnn::*
        method = "nn"
        parameters = {"h_abck_modifier": int(npz["fnn::h_abck_modifier"]) if "fnn::h_abck_modifier" in npz.files else 0,
                      "ckpt": str(path)}
    else:
        method = {0: "pexp", 1: "morse", 2: "density", 3: "sf"}[int(npz["descriptor::method"])]
        keys = {"pexp": ["rl", "pl"], "morse": ["D", "gamma", "r0"], "density": ["A", "beta", "re"],
                "sf": ["eta", "omega"]}[method]
        parameters = {k: np.atleast_1d(npz[f"descriptor::{k}"]).astype(float).tolist() for k in keys}
    max_moment = int(npz["max_moment"])
    gd = GenericRadialAtomicPotential(
        elements, method, parameters, param_space_method="pair",
        moment_tensors=list(range(max_moment + 1)),
        cutoff_function={0: "cosine", 1: "polynomial"}[int(npz["fctype"])],
        symmetric=bool(int(npz["is_T_symmetric"])), legacy_mode=False)
    layer_sizes = [int(x) for x in np.atleast_1d(npz["layer_sizes"])]
    activation = {0: "relu", 1: "softplus", 2: "tanh", 3: "squareplus"}[int(npz["actfn"])]
    bias_out = bool(int(npz["apply_output_bias"]))
    nn = AtomicNN(elements, gd, hidden_sizes=layer_sizes[:-1], activation=activation, minmax_scale=False,
                  use_resnet_dt=bool(int(npz["use_resnet_dt"])), use_atomic_static_energy=bias_out,
                  export_properties=("energy", "forces", "stress"))
    clf = UniversalTransformer(elements, rcut=float(npz["rmax"]), angular=False)
    nn.attach_transformer(clf)
    order = sorted(range(len(elements)), key=lambda i: elements[i])  # the file's own element order
    L = len(layer_sizes)
    for i in order:
        layers = []
        for j in range(L - 1):
            layers.append((np.array(npz[f"weights_{i}_{j}"], dtype=np.float64),
                           np.array(npz[f"biases_{i}_{j}"], dtype=np.float64).ravel()))
        wo = np.array(npz[f"weights_{i}_{L - 1}"], dtype=np.float64).reshape(-1, 1)
        bo = np.array(npz[f"biases_{i}_{L - 1}"], dtype=np.float64).ravel() if bias_out else None
        layers.append((wo, bo))
        nn.weights[elements[i]] = layers
    nn.precision = "high" if int(npz["precision"]) == 64 else "medium"
    meta = {"format": "tensoralloy/native-npz", "Transformer/params": clf.as_dict(),
            "Metadata/precision": nn.precision, "Metadata/api": API_VERSION,
            "Metadata/variational_energy": "energy", "Metadata/is_finite_temperature": 0,
            "Metadata/ops": {"energy": "Output/Energy/energy:0", "energy/atom": "Output/Energy/atomic:0",
                             "forces": "Output/Forces/forces:0", "stress": "Output/Stress/Voigt/stress:0",
                             "virial": "Output/Stress/Full/virial:0",
                             "total_pressure": "Output/Stress/pressure/GPa:0"},
            "nn": nn.as_dict()}
    return nn, clf, meta


def load_model(graph_model_path: str):
    """
    Read a model written by `AtomicNN.export` (or `EamAlloyNN.export`), or a native `.npz` written by
    the reference's `export_to_lammps_native`. Returns (nn, transformer, metadata dict).
    """
    from .transformer import UniversalTransformer

    path = str(graph_model_path)
    if path.endswith(".npz") and os.path.exists(path):
        with np.load(path) as probe:
            native = "descriptor::method" in probe.files or "use_fnn" in probe.files
        if native:
            return load_lammps_native(path)
    if path.endswith(".pb") or path.endswith(".pb.gz"):
        stem = _model_stem(path)
        if not os.path.exists(stem + ".json"):
            return load_graph_def(path)
    stem = _model_stem(path)
    with open(stem + ".json") as fp:
        meta = json.load(fp)
    if meta.get("format") != "tensoralloy_amd/1":
        raise ValueError(f"{stem}.json: unknown model format {meta.get('format')!r}")
    params = dict(meta["Transformer/params"])
    cls = params.pop("class")
    params.pop("predict_properties", None)
    if cls != "UniversalTransformer":
        raise ValueError(f"Unsupported transformer: {cls}")  # calculator.py:142
    clf = UniversalTransformer(**params)
    cfg = dict(meta["nn"])
    nn_cls = cfg.pop("class")
    if nn_cls == "AtomicNN":
        nn = AtomicNN(**cfg)
        nn.attach_transformer(clf)
        npz = np.load(os.path.join(os.path.dirname(stem) or ".", meta["weights"]))
        for i, el in enumerate(nn.elements):
            layers = []
            j = 0
            while f"weights_{i}_{j}" in npz:
                w = np.array(npz[f"weights_{i}_{j}"], dtype=np.float64)
                if w.ndim == 1:
                    w = w.reshape(-1, 1)
                b = np.array(npz[f"biases_{i}_{j}"], dtype=np.float64).ravel() \
                    if f"biases_{i}_{j}" in npz else None
                layers.append((w, b))
                j += 1
            if not layers:
                raise ValueError(f"{stem}.npz holds no weights for element {el}")
            nn.weights[el] = layers
            if nn._minmax_scale:
                nn.minmax[el] = (np.array(npz[f"xlo_{i}"]), np.array(npz[f"xhi_{i}"]))
        if getattr(getattr(nn.descriptor, "algorithm", None), "name", "") == "nn":
            layers, j = [], 0
            while f"fnn::weights_0_{j}" in npz:
                b = np.array(npz[f"fnn::biases_0_{j}"], dtype=np.float64).ravel() \
                    if f"fnn::biases_0_{j}" in npz else None
                layers.append((np.array(npz[f"fnn::weights_0_{j}"], dtype=np.float64), b))
                j += 1
            if not layers:
                raise ValueError(f"{stem}.npz holds no weights of the GRAP filter network")
            nn.descriptor.filter_weights = layers
    elif nn_cls in ("EamAlloyNN", "AdpNN"):
        from .eam import nn_from_dict
        npz = None
        if meta.get("weights"):
            npz = np.load(os.path.join(os.path.dirname(stem) or ".", meta["weights"]))
        nn = nn_from_dict(nn_cls, cfg, npz)
        nn.attach_transformer(clf)
    else:
        raise ValueError(f"Unsupported model class: {nn_cls}")
    precision = meta.get("Metadata/precision", "high")
    if precision not in ("high", "medium"):
        raise ValueError(f"{stem}.json: unknown precision {precision!r}")
    nn.precision = precision
    return nn, clf, meta
