"""
Containers for the per-structure index maps (the reference's feed-dict wire
format). Mirrors reference tensoralloy/transformer/metadata.py:19-107.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np


@dataclass(frozen=True)
class RadialMetadata:
    v2g_map: np.ndarray
    ilist: np.ndarray
    jlist: np.ndarray
    n1: np.ndarray
    rij: Optional[np.ndarray]

    def as_dict(self, use_computed_dists=True):
        if use_computed_dists:
            return {"g2.v2g_map": self.v2g_map, "g2.ilist": self.ilist,
                    "g2.jlist": self.jlist, "g2.n1": self.n1}
        return {"g2.v2g_map": self.v2g_map, "g2.rij": self.rij}


@dataclass(frozen=True)
class AngularMetadata:
    v2g_map: np.ndarray
    ilist: np.ndarray
    jlist: np.ndarray
    klist: np.ndarray
    n1: np.ndarray
    n2: np.ndarray
    n3: np.ndarray
    rijk: Optional[np.ndarray]

    def as_dict(self, use_computed_dists=True):
        if use_computed_dists:
            return {"g4.v2g_map": self.v2g_map, "g4.ilist": self.ilist,
                    "g4.jlist": self.jlist, "g4.klist": self.klist,
                    "g4.n1": self.n1, "g4.n2": self.n2, "g4.n3": self.n3}
        return {"g4.v2g_map": self.v2g_map, "g4.rijk": self.rijk}
