"""
`GenericRadialAtomicPotential` (GRAP): the descriptor the reference ships as its default
`pair_style = "atomic/grap"` (io/input/defaults.toml:5, :131-155). Mirror of reference
tensoralloy/nn/atomic/grap.py:272-377: same constructor, properties and `as_dict()`; the
arithmetic lives in csrc/ta_grap.hip.

Radial filters ("algorithms", grap.py:124-219): `sf` (eta, omega), `morse` (D, gamma, r0),
`density` (A, beta, re), `pexp` (rl, pl). The `nn` algorithm (a trained filter network) is out of
scope and raises `ValueError`, and so do moment tensors above rank 3 (the reference switches to
full 3^m tensors there, grap.py:531-590).
"""
from __future__ import annotations

import itertools
from typing import Dict, List, Sequence, Union

import numpy as np

GRAP_ALGORITHMS = {"sf": 0, "morse": 1, "density": 2, "pexp": 3}
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


class GenericRadialAtomicPotential:
    """The generic atomic potential with polarized radial interactions."""

    def __init__(self, elements: List[str], algorithm="sf", parameters=None,
                 param_space_method="pair", moment_tensors: Union[int, List[int]] = 0,
                 cutoff_function="cosine", symmetric=False, legacy_mode=True, h_abck_modifier=None):
        self._elements = sorted(list(elements))
        if isinstance(moment_tensors, int):
            moment_tensors = [moment_tensors]
        moment_tensors = list(set(int(m) for m in moment_tensors))  # grap.py:295
        if algorithm == "nn":
            raise ValueError("GRAP: the 'nn' filter network is not implemented by tensoralloy_amd")
        if any(m < 0 for m in moment_tensors):
            raise ValueError("moment tensors must be >= 0")
        if max(moment_tensors) > 3:
            raise ValueError("GRAP: moment tensors above rank 3 are not implemented by tensoralloy_amd")
        if cutoff_function not in ("cosine", "polynomial"):
            raise ValueError(f"Unknown cutoff function: {cutoff_function}")
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

    def flat_parameters(self) -> np.ndarray:
        if self._legacy_mode and self.max_moment > 2:
            raise ValueError("GRAP legacy mode implements moments 0, 1, 2 only (grap.py:423-460)")
        mask = sum(1 << m for m in self._moment_tensors)
        head = [GRAP_ALGORITHMS[self._algo.name], len(self._algo), self.max_moment,
                int(self._legacy_mode), int(self._symmetric), mask]
        return np.concatenate([np.array(head, dtype=np.float64), self._algo.constants().ravel()])
