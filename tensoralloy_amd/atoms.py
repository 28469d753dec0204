"""
A minimal stand-in for `ase.Atoms` and `ase.calculators.calculator.Calculator`.

The reference's boundary is an ASE calculator (tensoralloy/calculator.py:31).
ASE (>= 3.21, requirements.txt:3) is a third-party dependency that is not
installed in the build image; when it IS importable the real classes are used,
otherwise these shims provide exactly the members the hot path touches:

  Atoms:      positions, numbers, cell, pbc, info, get_chemical_symbols(),
              get_cell(complete=True), get_volume(), get_chemical_formula(mode=),
              get_positions(), copy(), __len__
  Calculator: results / atoms bookkeeping, check_state(), get_property(),
              get_potential_energy(), get_forces(), get_stress()
"""
from __future__ import annotations

import copy as _copy
from collections import Counter
from typing import Sequence

import numpy as np

try:  # pragma: no cover - exercised only where ASE exists
    from ase import Atoms as _AseAtoms
    from ase.calculators.calculator import (Calculator as _AseCalculator, all_changes as
                                            _ase_all_changes, PropertyNotImplementedError as
                                            _AsePNIE)
    HAVE_ASE = True
except Exception:  # ModuleNotFoundError in this image
    _AseAtoms = None
    HAVE_ASE = False

# periodic table (symbols only; index = atomic number)
chemical_symbols = [
    "X", "H", "He", "Li", "Be", "B", "C", "N", "O", "F", "Ne", "Na", "Mg", "Al", "Si", "P",
    "S", "Cl", "Ar", "K", "Ca", "Sc", "Ti", "V", "Cr", "Mn", "Fe", "Co", "Ni", "Cu", "Zn",
    "Ga", "Ge", "As", "Se", "Br", "Kr", "Rb", "Sr", "Y", "Zr", "Nb", "Mo", "Tc", "Ru", "Rh",
    "Pd", "Ag", "Cd", "In", "Sn", "Sb", "Te", "I", "Xe", "Cs", "Ba", "La", "Ce", "Pr", "Nd",
    "Pm", "Sm", "Eu", "Gd", "Tb", "Dy", "Ho", "Er", "Tm", "Yb", "Lu", "Hf", "Ta", "W", "Re",
    "Os", "Ir", "Pt", "Au", "Hg", "Tl", "Pb", "Bi", "Po", "At", "Rn", "Fr", "Ra", "Ac", "Th",
    "Pa", "U", "Np", "Pu", "Am", "Cm", "Bk", "Cf", "Es", "Fm", "Md", "No", "Lr"]
atomic_numbers = {s: z for z, s in enumerate(chemical_symbols)}
# covalent radii in Angstrom (Cordero et al., Dalton Trans. 2008, 2832: the table behind
# `ase.data.covalent_radii`, which the reference's GRAP `h_abck_modifier` reads, grap.py:623-629);
# index = atomic number, 0.2 for the dummy element as in ASE, nan beyond Cm
covalent_radii = [
    0.2, 0.31, 0.28, 1.28, 0.96, 0.84, 0.76, 0.71, 0.66, 0.57, 0.58, 1.66, 1.41, 1.21, 1.11, 1.07,
    1.05, 1.02, 1.06, 2.03, 1.76, 1.70, 1.60, 1.53, 1.39, 1.39, 1.32, 1.26, 1.24, 1.32, 1.22,
    1.22, 1.20, 1.19, 1.20, 1.20, 1.16, 2.20, 1.95, 1.90, 1.75, 1.64, 1.54, 1.47, 1.46, 1.42,
    1.39, 1.45, 1.44, 1.42, 1.39, 1.39, 1.38, 1.39, 1.40, 2.44, 2.15, 2.07, 2.04, 2.03, 2.01,
    1.99, 1.98, 1.98, 1.96, 1.94, 1.92, 1.92, 1.89, 1.90, 1.87, 1.87, 1.75, 1.70, 1.62, 1.51,
    1.44, 1.41, 1.36, 1.36, 1.32, 1.45, 1.46, 1.48, 1.40, 1.50, 1.50, 2.60, 2.21, 2.15, 2.06,
    2.00, 1.96, 1.90, 1.87, 1.80, 1.69] + [float("nan")] * 7
# standard atomic weights (IUPAC abridged values; mass number of the longest-lived isotope for the
# radioactive elements), index = atomic number. Only written into exported model files.
atomic_masses = [
    1.0, 1.008, 4.002602, 6.94, 9.0121831, 10.81, 12.011, 14.007, 15.999, 18.998403163, 20.1797,
    22.98976928, 24.305, 26.9815385, 28.085, 30.973761998, 32.06, 35.45, 39.948, 39.0983, 40.078,
    44.955908, 47.867, 50.9415, 51.9961, 54.938044, 55.845, 58.933194, 58.6934, 63.546, 65.38,
    69.723, 72.630, 74.921595, 78.971, 79.904, 83.798, 85.4678, 87.62, 88.90584, 91.224, 92.90637,
    95.95, 97.90721, 101.07, 102.90550, 106.42, 107.8682, 112.414, 114.818, 118.710, 121.760,
    127.60, 126.90447, 131.293, 132.90545196, 137.327, 138.90547, 140.116, 140.90766, 144.242,
    144.91276, 150.36, 151.964, 157.25, 158.92535, 162.500, 164.93033, 167.259, 168.93422, 173.054,
    174.9668, 178.49, 180.94788, 183.84, 186.207, 190.23, 192.217, 195.084, 196.966569, 200.592,
    204.38, 207.2, 208.98040, 208.98243, 209.98715, 222.01758, 223.01974, 226.02541, 227.02775,
    232.0377, 231.03588, 238.02891, 237.04817, 244.06421, 243.06138, 247.07035, 247.07031,
    251.07959, 252.0830, 257.09511, 258.09843, 259.1010, 262.110]

all_changes = ["positions", "numbers", "cell", "pbc", "initial_charges", "initial_magmoms"]


class PropertyNotImplementedError(NotImplementedError):
    """Raised when a calculator is asked for a property it cannot compute."""


def complete_cell(cell) -> np.ndarray:
    """`Atoms.get_cell(complete=True)`: fill zero lattice vectors with unit
    vectors orthogonal to the others."""
    cell = np.array(cell, dtype=np.float64).reshape(3, 3)
    if cell.any(axis=1).all():  # three lattice vectors: nothing to fill in
        return cell
    missing = [a for a in range(3) if not np.any(cell[a])]
    if len(missing) == 3:
        return np.eye(3)
    if len(missing) == 2:
        present = [a for a in range(3) if a not in missing][0]
        v = cell[present] / np.linalg.norm(cell[present])
        t = np.eye(3)[int(np.argmin(np.abs(v)))]
        u = np.cross(v, t)
        u /= np.linalg.norm(u)
        cell[missing[0]], cell[missing[1]] = u, np.cross(v, u)
    elif len(missing) == 1:
        a = missing[0]
        b, c = [x for x in range(3) if x != a]
        n = np.cross(cell[b], cell[c])
        cell[a] = n / np.linalg.norm(n)
    return cell


class _Cell:
    """Tiny analogue of `ase.cell.Cell` (array-like with `.array`)."""

    def __init__(self, array):
        self.array = np.array(array, dtype=np.float64).reshape(3, 3)

    def __array__(self, dtype=None, copy=None):
        return self.array if dtype is None else self.array.astype(dtype)

    def __getitem__(self, item):
        return self.array[item]

    @property
    def volume(self):
        return abs(np.linalg.det(self.array))


def _parse_formula(formula: str):
    import re
    out = []
    for sym, num in re.findall(r"([A-Z][a-z]*)(\d*)", formula):
        out.extend([sym] * (int(num) if num else 1))
    return out


class Atoms:
    """Shim with the subset of the `ase.Atoms` API used by the hot path."""

    def __init__(self, symbols=None, positions=None, cell=None, pbc=False, numbers=None,
                 info=None):
        if symbols is not None:
            if isinstance(symbols, str):
                symbols = _parse_formula(symbols)
            numbers = [atomic_numbers[s] for s in symbols]
        self.numbers = np.array(numbers if numbers is not None else [], dtype=np.int64)
        n = len(self.numbers)
        self.positions = (np.zeros((n, 3)) if positions is None
                          else np.array(positions, dtype=np.float64).reshape(n, 3))
        if cell is None:
            cell = np.zeros((3, 3))
        cell = np.array(cell, dtype=np.float64)
        if cell.shape == (3,):
            cell = np.diag(cell)
        self._cell = cell.reshape(3, 3)
        if isinstance(pbc, (bool, np.bool_, int)):
            pbc = [bool(pbc)] * 3
        self.pbc = np.array(pbc, dtype=bool).reshape(3)
        self.info = dict(info or {})
        self.calc = None

    # -- ase.Atoms surface -----------------------------------------------------
    def __len__(self):
        return len(self.numbers)

    @property
    def cell(self):
        return _Cell(self._cell)

    @cell.setter
    def cell(self, value):
        self._cell = np.array(value, dtype=np.float64).reshape(3, 3)

    def set_cell(self, cell, scale_atoms=False):
        new = np.array(cell, dtype=np.float64).reshape(3, 3)
        if scale_atoms:
            m = np.linalg.solve(complete_cell(self._cell), complete_cell(new))
            self.positions = self.positions @ m
        self._cell = new

    def get_cell(self, complete=False):
        return _Cell(complete_cell(self._cell) if complete else self._cell)

    def get_volume(self):
        return float(abs(np.linalg.det(self._cell)))

    def get_positions(self):
        return self.positions.copy()

    def set_positions(self, p):
        self.positions = np.array(p, dtype=np.float64).reshape(len(self), 3)

    def get_pbc(self):
        return self.pbc.copy()

    def get_atomic_numbers(self):
        return self.numbers.copy()

    def get_chemical_symbols(self):
        return [chemical_symbols[z] for z in self.numbers]

    def get_chemical_formula(self, mode="hill"):
        symbols = self.get_chemical_symbols()
        if mode == "reduce":
            # run-length encoding in the given order: 'Pd3O2' vs 'Pd2O2Pd'
            out, i = "", 0
            while i < len(symbols):
                j = i
                while j < len(symbols) and symbols[j] == symbols[i]:
                    j += 1
                out += symbols[i] + (str(j - i) if j - i > 1 else "")
                i = j
            return out
        count = Counter(symbols)
        return "".join(f"{s}{count[s] if count[s] > 1 else ''}" for s in sorted(count))

    def copy(self):
        a = Atoms(numbers=self.numbers.copy(), positions=self.positions.copy(),
                  cell=self._cell.copy(), pbc=self.pbc.copy(),
                  info=_copy.deepcopy(self.info) if self.info else {})
        return a

    def repeat(self, rep):
        if isinstance(rep, int):
            rep = (rep, rep, rep)
        pos, num = [], []
        for x in range(rep[0]):
            for y in range(rep[1]):
                for z in range(rep[2]):
                    pos.append(self.positions + np.array([x, y, z]) @ self._cell)
                    num.append(self.numbers)
        return Atoms(numbers=np.concatenate(num), positions=np.concatenate(pos),
                     cell=self._cell * np.array(rep)[:, None], pbc=self.pbc.copy(),
                     info=_copy.deepcopy(self.info))

    __mul__ = repeat

    # calculator plumbing
    def set_calculator(self, calc):
        self.calc = calc

    def get_potential_energy(self):
        return self.calc.get_potential_energy(self)

    def get_forces(self):
        return self.calc.get_forces(self)

    def get_stress(self, voigt=True):
        return self.calc.get_stress(self, voigt=voigt)


def compare_atoms(a, b, tol=1e-15):
    """System changes between two atoms objects (ASE `compare_atoms`)."""
    if a is None:
        return list(all_changes)
    changes = []
    if len(a) != len(b) or not np.array_equal(np.asarray(a.numbers), np.asarray(b.numbers)):
        return list(all_changes)
    def differ(x, y):
        x, y = np.asarray(x), np.asarray(y)
        if x.shape != y.shape:
            return True
        if np.array_equal(x, y):       # the common case on an MD trajectory is decided here, cheaply
            return False
        return bool((np.abs(x - y) > tol).any())

    if differ(a.positions, b.positions):
        changes.append("positions")
    if differ(a.get_cell(), b.get_cell()):
        changes.append("cell")
    if not np.array_equal(np.asarray(a.pbc), np.asarray(b.pbc)):
        changes.append("pbc")
    return changes


class Calculator:
    """Shim of `ase.calculators.calculator.Calculator` (state bookkeeping only)."""

    implemented_properties: Sequence[str] = []
    default_parameters = {}

    def __init__(self, restart=None, ignore_bad_restart_file=False, label=None, atoms=None,
                 **kwargs):
        self.atoms = None
        self.results = {}
        self.parameters = dict(self.default_parameters)
        self.label = label
        if atoms is not None:
            atoms.calc = self

    def reset(self):
        self.atoms = None
        self.results = {}

    def check_state(self, atoms, tol=1e-15):
        return compare_atoms(self.atoms, atoms, tol)

    def calculate(self, atoms=None, properties=("energy",), system_changes=all_changes):
        if atoms is not None:
            self.atoms = atoms.copy()

    def calculation_required(self, atoms, properties):
        if self.check_state(atoms):
            return True
        return any(p not in self.results for p in properties)

    def get_property(self, name, atoms=None, allow_calculation=True):
        if name not in self.implemented_properties:
            raise PropertyNotImplementedError(f"{name} property not implemented")
        if atoms is None:
            atoms = self.atoms
            system_changes = []
        else:
            system_changes = self.check_state(atoms)
            if system_changes:
                self.reset()
        if name not in self.results:
            if not allow_calculation:
                return None
            self.calculate(atoms, [name], system_changes)
        if name not in self.results:
            raise PropertyNotImplementedError(f"{name} not present in this calculation")
        result = self.results[name]
        if isinstance(result, np.ndarray):
            result = result.copy()
        return result

    def get_potential_energy(self, atoms=None, force_consistent=False):
        return self.get_property("energy", atoms)

    def get_forces(self, atoms=None):
        return self.get_property("forces", atoms)

    def get_stress(self, atoms=None):
        return self.get_property("stress", atoms)


if HAVE_ASE:  # pragma: no cover
    BaseCalculator = _AseCalculator
    all_changes = _ase_all_changes
    PropertyNotImplementedError = _AsePNIE
else:
    BaseCalculator = Calculator
