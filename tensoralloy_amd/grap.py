"""
`GenericRadialAtomicPotential` (GRAP): the descriptor the reference ships as its default
`pair_style = "atomic/grap"` (io/input/defaults.toml:5, :131-155). Mirror of reference
tensoralloy/nn/atomic/grap.py:272-377: same constructor, properties and `as_dict()`; the
arithmetic lives in csrc/ta_grap.hip.

Radial filters ("algorithms", grap.py:124-219): `sf` (eta, omega), `morse` (D, gamma, r0),
`density` (A, beta, re), `pexp` (rl, pl), and `nn` (`NNAlgorithm`, grap.py:220-270: one shared
filter network r -> K filter values, `convolution1x1(r, hidden_sizes, num_out=num_filters,
output_bias=False, use_resnet_dt=True)`, grap.py:632-643; weights in
`descriptor.filter_weights = [(W, b), ..., (W_out, None)]`; `h_abck_modifier` 0, 1, 2; only the
non-legacy formulation, as in the reference). Moment tensors up to rank 5: above rank 3 the
reference sums the full 3^m tensors with unit weights (get_moment_tensor / get_T_dm, grap.py:538-600),
here the packed components carry their multinomial coefficients instead: the same numbers.
"""
from __future__ import annotations

import itertools
from typing import Dict, List, Sequence, Union

import numpy as np

GRAP_ALGORITHMS = {"sf": 0, "morse": 1, "density": 2, "pexp": 3, "nn": 4}
_ACT_IDS = {"relu": 0, "softplus": 1, "tanh": 2, "squareplus": 3, "leaky_relu": 4, "sigmoid": 5,
            "softsign": 6, "elu": 7}
REQUIRED_KEYS = {"sf": ["eta", "omega"], "morse": ["D", "gamma", "r0"],
                 "density": ["A", "beta", "re"], "pexp": ["rl", "pl"]}


class Algorithm:
    """One family of radial filters and its hyper-parameter space (grap.py:32-121)."""

    def __init__(self, name: str, parameters: Dict[str, Sequence[float]], param_space_method="cross"):
        if name not in REQUIRED_KEYS:
            raise ValueError(f"GRAP: algorithm '{name}' is not implemented")
        if param_space_method not in ("cross", "pair"):
            raise ValueError("param_space_method must be 'cross' or 'pair'")
        self.name = name
        self.required_keys = REQUIRED_KEYS[name]
        for key in self.required_keys:
            if key not in parameters or len(parameters[key]) < 1:
                raise ValueError(f"GRAP/{name}: parameter '{key}' is missing or empty")
        self._params = {key: [float(x) for x in parameters[key]] for key in self.required_keys}
        self._param_space_method = param_space_method
        if param_space_method == "cross":
            # sklearn ParameterGrid: keys sorted, the last key varies fastest
            names = sorted(self._params)
            self._grid = [dict(zip(names, combo))
                          for combo in itertools.product(*[self._params[n] for n in names])]
        else:
            if len({len(v) for v in self._params.values()}) > 1:
                raise ValueError("Hyperparameters must have the same length for gen:pair")
            size = len(self._params[self.required_keys[0]])
            self._grid = [{key: self._params[key][i] for key in self._params} for i in range(size)]

    def __len__(self):
        return len(self._grid)

    def __getitem__(self, item):
        return self._grid[item]

    def as_dict(self, convert_to_pairs=False):
        if not convert_to_pairs:
            return {"algorithm": self.name, "parameters": {k: list(v) for k, v in self._params.items()},
                    "param_space_method": self._param_space_method}
        parameters = {key: [float(row[key]) for row in self._grid] for key in self._params}
        return {"algorithm": self.name, "parameters": parameters, "param_space_method": "pair"}

    def constants(self) -> np.ndarray:
        """[K, 3] filter constants in the order the C ABI documents."""
        out = np.zeros((len(self._grid), 3))
        for k, row in enumerate(self._grid):
            for c, key in enumerate(self.required_keys):
                out[k, c] = row[key]
        return out


class NNAlgorithm:
    """The neural-network filter model (grap.py:220-270): hyper-parameters only; the weights live
    in `GenericRadialAtomicPotential.filter_weights`."""

    required_keys = ["activation_fn", "hidden_sizes", "num_filters", "use_reset_dt", "ckpt",
                     "trainable", "h_abck_modifier"]
    name = "nn"

    def __init__(self, parameters: dict):
        parameters = parameters or {}
        self.use_resnet_dt = bool(parameters.get("use_resnet_dt", True))
        self.hidden_sizes = [int(x) for x in parameters.get("hidden_sizes", [32, 32, 32])]
        self.activation = parameters.get("activation", "softplus")
        self.num_filters = int(parameters.get("num_filters", 16))
        self.ckpt = parameters.get("ckpt", None)
        self.trainable = parameters.get("trainable", True)
        self.h_abck_modifier = int(parameters.get("h_abck_modifier", 0) or 0)
        if self.h_abck_modifier not in (0, 1, 2):
            raise ValueError(f"Unknown H(r) modifier: {self.h_abck_modifier}")  # grap.py:630-631
        if self.activation.lower() not in _ACT_IDS:
            raise ValueError(f"The activation function '{self.activation}' cannot be recognized!")
        if not self.hidden_sizes:
            raise ValueError("GRAP/nn: at least one hidden layer")

    def __len__(self):
        return self.num_filters

    def __getitem__(self, item):
        return self.__dict__[item]

    def as_dict(self, convert_to_pairs=False):
        return {"use_resnet_dt": self.use_resnet_dt, "hidden_sizes": list(self.hidden_sizes),
                "activation": self.activation, "num_filters": self.num_filters,
                "trainable": self.trainable, "ckpt": self.ckpt,
                "h_abck_modifier": self.h_abck_modifier}


class GenericRadialAtomicPotential:
    """The generic atomic potential with polarized radial interactions."""

    def __init__(self, elements: List[str], algorithm="sf", parameters=None,
                 param_space_method="pair", moment_tensors: Union[int, List[int]] = 0,
                 cutoff_function="cosine", symmetric=False, legacy_mode=True, h_abck_modifier=None):
        self._elements = sorted(list(elements))
        if isinstance(moment_tensors, int):
            moment_tensors = [moment_tensors]
        moment_tensors = list(set(int(m) for m in moment_tensors))  # grap.py:295
        if any(m < 0 for m in moment_tensors):
            raise ValueError("moment tensors must be >= 0")
        if max(moment_tensors) > 5:
            raise ValueError("The maximum angular moment should be <= 5")  # grap.py:580-581
        if cutoff_function not in ("cosine", "polynomial"):
            raise ValueError(f"Unknown cutoff function: {cutoff_function}")
        self.filter_weights = None
        if algorithm == "nn":
            if legacy_mode:
                # the legacy formulation calls `self._algo.compute`, which NNAlgorithm lacks
                raise ValueError("GRAP: the 'nn' algorithm exists only with legacy_mode=False "
                                 "(grap.py:620-643 vs :424-428)")
            self._algo = NNAlgorithm(parameters)
            if self._algo.ckpt:
                self.load_filter_checkpoint(self._algo.ckpt)
        else:
            self._algo = Algorithm(algorithm, parameters or {}, param_space_method)
        self._moment_tensors = moment_tensors
        self._cutoff_function = cutoff_function
        self._parameters = parameters
        self._param_space_method = param_space_method
        self._legacy_mode = bool(legacy_mode)
        self._symmetric = bool(symmetric)
        if len(self._algo) > 32:
            raise ValueError("GRAP: at most 32 radial filters")

    @property
    def name(self):
        return "GRAP"

    @property
    def elements(self):
        return self._elements

    @property
    def algorithm(self):
        return self._algo

    @property
    def max_moment(self):
        return max(self._moment_tensors)

    @property
    def moment_tensors(self):
        return list(self._moment_tensors)

    @property
    def is_T_symmetric(self):
        return self._symmetric

    @property
    def legacy_mode(self):
        return self._legacy_mode

    @property
    def cutoff_function(self):
        return self._cutoff_function

    def as_dict(self):
        return {"@class": self.__class__.__name__, "@module": "tensoralloy.nn.atomic.grap",
                "elements": self._elements, "algorithm": self._algo.name,
                "parameters": self._parameters, "param_space_method": self._param_space_method,
                "moment_tensors": self._moment_tensors, "cutoff_function": self._cutoff_function,
                "symmetric": self._symmetric, "legacy_mode": self._legacy_mode}

    @property
    def features_per_filter(self) -> int:
        if self._legacy_mode:
            return len([m for m in self._moment_tensors if m in (0, 1, 2)])  # grap.py:423-460
        return self.max_moment + 1                                           # grap.py:606

    def ndim(self, angular: bool = False) -> int:
        return self.features_per_filter * len(self._algo) * len(self._elements)

    # -- the `nn` filter network ------------------------------------------------------------
    def initialize_filters(self, seed=611, bias_scale=0.0):
        """He-normal kernels truncated at 2 sigma, zero (or small random) hidden biases, no output
        bias (nn/init_ops.py:20-30; grap.py:632-643)."""
        if self._algo.name != "nn":
            raise ValueError("only the 'nn' algorithm has filter weights")
        rng = np.random.RandomState(seed)
        sizes = [1] + list(self._algo.hidden_sizes) + [self._algo.num_filters]
        layers = []
        for l in range(len(sizes) - 1):
            sigma = np.sqrt(2.0 / sizes[l])
            w = rng.normal(0.0, sigma, size=(sizes[l], sizes[l + 1]))
            bad = np.abs(w) > 2 * sigma
            while bad.any():
                w[bad] = rng.normal(0.0, sigma, size=int(bad.sum()))
                bad = np.abs(w) > 2 * sigma
            if l == len(sizes) - 2:
                b = None
            else:
                b = bias_scale * rng.normal(size=sizes[l + 1]) if bias_scale else np.zeros(sizes[l + 1])
            layers.append((w, b))
        self.filter_weights = layers

    def load_filter_checkpoint(self, ckpt: str):
        """Filter weights from a native `.npz` (`fnn::*` keys, atomic.py:409-438;
        convolutional.py:214-233 reads the same file as a checkpoint)."""
        npz = np.load(ckpt)
        sizes = [int(x) for x in np.atleast_1d(npz["fnn::layer_sizes"])]
        act = {0: "relu", 1: "softplus", 2: "tanh", 3: "squareplus"}[int(npz["fnn::actfn"])]
        self._algo.activation = act
        self._algo.hidden_sizes = sizes[:-1]
        self._algo.num_filters = sizes[-1]
        self._algo.use_resnet_dt = bool(int(npz["fnn::use_resnet_dt"]))
        full = [1] + sizes
        layers = []
        for j in range(len(sizes)):
            w = np.array(npz[f"fnn::weights_0_{j}"], dtype=np.float64).reshape(full[j], full[j + 1])
            b = np.array(npz[f"fnn::biases_0_{j}"], dtype=np.float64).ravel() \
                if f"fnn::biases_0_{j}" in npz.files else None
            layers.append((w, b))
        self.filter_weights = layers

    def flat_parameters(self) -> np.ndarray:
        if self._legacy_mode and self.max_moment > 2:
            raise ValueError("GRAP legacy mode implements moments 0, 1, 2 only (grap.py:423-460)")
        mask = sum(1 << m for m in self._moment_tensors)
        head = [GRAP_ALGORITHMS[self._algo.name], len(self._algo), self.max_moment,
                int(self._legacy_mode), int(self._symmetric), mask]
        if self._algo.name == "nn":
            # [dense layers incl. output, activation id, use_resnet_dt, h_abck_modifier,
            #  sizes 1, h1, ..., K, then per layer W [in][out] row-major and b [out] (zeros: none)]
            if not self.filter_weights:
                raise ValueError("GRAP/nn: no filter weights: call initialize_filters() or set "
                                 ".filter_weights")
            sizes, flat = [1], []
            for w, b in self.filter_weights:
                w = np.asarray(w, dtype=np.float64)
                if w.ndim != 2 or w.shape[0] != sizes[-1]:
                    raise ValueError(f"GRAP/nn: weight shape {w.shape} does not chain from {sizes[-1]}")
                sizes.append(w.shape[1])
                flat += [w.ravel(), np.zeros(w.shape[1]) if b is None else np.asarray(b, dtype=np.float64).ravel()]
            if sizes[-1] != len(self._algo):
                raise ValueError("GRAP/nn: the output layer must have num_filters units")
            extra = [len(self.filter_weights), _ACT_IDS[self._algo.activation.lower()],
                     int(self._algo.use_resnet_dt), self._algo.h_abck_modifier] + sizes
            if self._algo.h_abck_modifier:
                # the network's input is r / rcov or exp(-r / rcov) with the covalent radius of the
                # CENTRE's element (grap.py:623-629): one radius per element closes the block
                from .atoms import atomic_numbers, covalent_radii
                flat.append(np.array([covalent_radii[atomic_numbers[el]] for el in self._elements]))
            return np.concatenate([np.array(head + extra, dtype=np.float64)] + flat)
        return np.concatenate([np.array(head, dtype=np.float64), self._algo.constants().ravel()])
