"""
Minimal extended-XYZ reader (the input format of BASELINE configs 1 and of the
reference's `test_files/*.extxyz`). Only what the hot path needs: species,
positions, `Lattice`, `pbc`, and per-frame / per-atom labels kept in `info`.
The reference reads these files through ASE (tensoralloy/io/read.py:43-235),
which is not a dependency here.
"""
from __future__ import annotations

import re
from typing import List

import numpy as np

from .atoms import Atoms

_KV = re.compile(r'(\w+)=("([^"]*)"|(\S+))')


def read_extxyz(path: str) -> List[Atoms]:
    with open(path) as fp:
        lines = fp.read().split("\n")
    frames, k = [], 0
    while k < len(lines) and lines[k].strip():
        n = int(lines[k])
        header = {m.group(1): (m.group(3) if m.group(3) is not None else m.group(4))
                  for m in _KV.finditer(lines[k + 1])}
        props = header.get("Properties", "species:S:1:pos:R:3").split(":")
        cols, c = [], 0
        for name, kind, width in zip(props[0::3], props[1::3], props[2::3]):
            cols.append((name, kind, c, c + int(width)))
            c += int(width)
        rows = [ln.split() for ln in lines[k + 2:k + 2 + n]]
        data = {}
        for name, kind, lo, hi in cols:
            if kind == "S":
                data[name] = [r[lo] for r in rows]
            else:
                data[name] = np.array([[float(x) for x in r[lo:hi]] for r in rows])
        cell = np.zeros((3, 3))
        if "Lattice" in header:
            cell = np.array([float(x) for x in header["Lattice"].split()]).reshape(3, 3)
        pbc = [True] * 3 if "Lattice" in header else [False] * 3
        if "pbc" in header:
            pbc = [t.upper().startswith("T") for t in header["pbc"].split()]
        info = {}
        for key, val in header.items():
            if key in ("Lattice", "Properties", "pbc"):
                continue
            try:
                parts = [float(x) for x in val.split()]
                info[key] = parts[0] if len(parts) == 1 else np.array(parts)
            except ValueError:
                info[key] = val
        for name in data:
            if name not in ("species", "pos"):
                info[name] = data[name]
        frames.append(Atoms(symbols=data["species"], positions=data["pos"], cell=cell, pbc=pbc,
                            info=info))
        k += 2 + n
    return frames
