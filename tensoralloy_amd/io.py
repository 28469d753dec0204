"""
File formats of the hot path's inputs: extended XYZ and LAMMPS setfl / ADP potential tables.

Minimal extended-XYZ reader (the input format of BASELINE configs 1 and of the
reference's `test_files/*.extxyz`). Only what the hot path needs: species,
positions, `Lattice`, `pbc`, and per-frame / per-atom labels kept in `info`.
The reference reads these files through ASE (tensoralloy/io/read.py:43-235),
which is not a dependency here.
"""
from __future__ import annotations

import gzip
import re
from dataclasses import dataclass, field
from typing import Dict, List

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


# --------------------------------------------------------------------------- #
# LAMMPS setfl (eam/alloy) and ADP tables  <- reference tensoralloy/io/lammps.py:62-235
# --------------------------------------------------------------------------- #

@dataclass
class Spline:
    """A tabulated function: knots `x`, values `y`, to be interpolated by a cubic spline with
    natural boundaries (io/lammps.py:62-73; `CubicInterpolator(x, y, natural_boundary=True)` is
    how the reference evaluates one, nn/eam/potentials/tests/test_mishin.py:60-70)."""
    bc_start: float
    bc_end: float
    x: np.ndarray
    y: np.ndarray
    natural_boundary: bool = True


@dataclass
class SetFL:
    """Contents of an `eam/alloy` or `adp` setfl file (io/lammps.py:75-93). `phi[key].y` is phi(r)
    itself: the file stores r * phi(r), the reader divides by r except at r = 0 where the raw
    value stays (io/lammps.py:196-199). Pair keys are the two symbols in the file's element order;
    `pair(a, b)` looks a term up in either order."""
    elements: List[str]
    rho: Dict[str, Spline]
    phi: Dict[str, Spline]
    embed: Dict[str, Spline]
    dipole: Dict[str, Spline]
    quadrupole: Dict[str, Spline]
    nr: int
    dr: float
    nrho: int
    drho: float
    rcut: float
    atomic_masses: List[float] = field(default_factory=list)
    lattice_constants: List[float] = field(default_factory=list)
    lattice_types: List[str] = field(default_factory=list)

    def pair(self, group: str, a: str, b: str) -> Spline:
        table = getattr(self, group)
        for key in (a + b, b + a):
            if key in table:
                return table[key]
        raise KeyError(f"no {group} table for {a}-{b}")


def _read_setfl(filename: str, is_adp: bool) -> SetFL:
    opener = gzip.open if str(filename).endswith(".gz") else open
    with opener(filename, "rb") as fp:
        # some published tables have "\r\r\n" line ends (Be_Agrawal.eam.alloy): drop every \r
        lines = fp.read().decode("utf-8", "replace").replace("\r", "").split("\n")
    head = lines[3].split()
    n_el = int(head[0])
    elements = head[1:1 + n_el]
    if len(elements) != n_el:
        raise ValueError(f"{filename}: line 4 names {len(elements)} elements, expected {n_el}")
    v = lines[4].split()
    nrho, drho, nr, dr, rcut = int(v[0]), float(v[1]), int(v[2]), float(v[3]), float(v[4])
    # values may come one or several per line (the reference's reader handles one per line only)
    tok = " ".join(lines[5:]).split()
    pos = 0
    r = np.linspace(0.0, nr * dr, nr, endpoint=False)      # io/lammps.py:100
    rho_x = np.linspace(0.0, nrho * drho, nrho, endpoint=False)

    def take(n):
        nonlocal pos
        if pos + n > len(tok):
            raise ValueError(f"{filename}: file ends inside a table")
        out = np.array(tok[pos:pos + n], dtype=np.float64)
        pos += n
        return out

    rho, embed, masses, consts, types = {}, {}, [], [], []
    for el in elements:
        masses.append(float(tok[pos + 1]))
        consts.append(float(tok[pos + 2]))
        types.append(tok[pos + 3])
        pos += 4
        embed[el] = Spline(0.0, 0.0, rho_x, take(nrho))
        rho[el] = Spline(0.0, 0.0, r, take(nr))
    # setfl order of the pair tables: (1,1), (2,1), (2,2), (3,1), ... (LAMMPS pair_eam_alloy).
    # The reference's reader walks (1,1), (1,2), (2,2), ... (io/lammps.py:153-160), which is the
    # same for one and two elements -- all its fixtures -- and mislabels tables beyond that.
    order = [(i, j) for i in range(n_el) for j in range(i + 1)]
    phi, dipole, quadrupole = {}, {}, {}
    for i, j in order:
        y = take(nr)
        y[1:] = y[1:] / r[1:]                               # io/lammps.py:196-199
        phi[elements[j] + elements[i]] = Spline(0.0, 0.0, r, y)
    if is_adp:
        for group in (dipole, quadrupole):
            for i, j in order:
                group[elements[j] + elements[i]] = Spline(0.0, 0.0, r, take(nr))
    return SetFL(elements=elements, rho=rho, phi=phi, embed=embed, dipole=dipole,
                 quadrupole=quadrupole, nr=nr, dr=dr, nrho=nrho, drho=drho, rcut=rcut,
                 atomic_masses=masses, lattice_constants=consts, lattice_types=types)


def read_eam_alloy_setfl(filename: str) -> SetFL:
    """Read a LAMMPS eam/alloy setfl file (io/lammps.py:224-228)."""
    return _read_setfl(filename, is_adp=False)


def read_adp_setfl(filename: str) -> SetFL:
    """Read a LAMMPS adp setfl file (io/lammps.py:231-235)."""
    return _read_setfl(filename, is_adp=True)


def natural_spline_coefficients(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """Piecewise cubics [n - 1][4] of the natural cubic spline through (x_k, y_k) on a uniform
    grid: on [x_k, x_k+1], f = c0 + c1 t + c2 t^2 + c3 t^3 with t = x - x_k. Second derivatives
    from the tridiagonal system (Thomas algorithm), zero at both ends."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    n = len(x)
    if n < 3:
        raise ValueError("a spline table needs at least 3 points")
    h = (x[-1] - x[0]) / (n - 1)
    # h/6 M[k-1] + 2h/3 M[k] + h/6 M[k+1] = (y[k+1] - 2 y[k] + y[k-1]) / h,  M[0] = M[n-1] = 0
    rhs = (y[2:] - 2.0 * y[1:-1] + y[:-2]) * (6.0 / (h * h))
    m = n - 2
    cp = np.empty(m)
    dp = np.empty(m)
    cp[0] = 1.0 / 4.0
    dp[0] = rhs[0] / 4.0
    for k in range(1, m):
        den = 4.0 - cp[k - 1]
        cp[k] = 1.0 / den
        dp[k] = (rhs[k] - dp[k - 1]) / den
    M = np.zeros(n)
    M[m] = dp[m - 1]
    for k in range(m - 2, -1, -1):
        M[k + 1] = dp[k] - cp[k] * M[k + 2]
    c = np.empty((n - 1, 4))
    c[:, 0] = y[:-1]
    c[:, 1] = (y[1:] - y[:-1]) / h - h * (2.0 * M[:-1] + M[1:]) / 6.0
    c[:, 2] = 0.5 * M[:-1]
    c[:, 3] = (M[1:] - M[:-1]) / (6.0 * h)
    return c
