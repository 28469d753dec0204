#!/usr/bin/env python3
"""
Generates the golden fixtures under tests/golden/ from the reference mounted at
/root/reference. Run in the BUILD container only (the reference never travels
to the GPU box): `python tests/golden/make_golden.py`.

What is produced is DATA: inputs plus expected outputs. Expected outputs come
from (a) data files the reference's own tests hold (copied/sampled verbatim),
(b) the reference's pure-NumPy / pure-Python helper functions, executed here by
compiling only those function definitions out of the reference's files (the
files themselves import TensorFlow / ASE, which are not installed: ordinary
ModuleNotFoundError, nothing was refused), (c) literal numbers asserted in the
reference's tests. No reference source text is stored in this repository.
"""
import ast
import importlib.util
import itertools
import json
import os
import sqlite3
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def functions_from(path, names, namespace):
    """Compile only the named top-level function definitions of a reference file."""
    with open(path) as fp:
        tree = ast.parse(fp.read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    missing = set(names) - {n.name for n in body}
    if missing:
        raise RuntimeError(f"{path}: functions not found: {missing}")
    ns = dict(namespace)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


def module_from(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def read_xyz(path):
    with open(path) as fp:
        lines = fp.read().split("\n")
    n = int(lines[0])
    sym, pos = [], []
    for ln in lines[2:2 + n]:
        t = ln.split()
        sym.append(t[0])
        pos.append([float(x) for x in t[1:4]])
    return sym, np.array(pos)


def main():
    out = {}

    # ---- 1. symmetry-function values from the reference's NumPy test helpers on B28.xyz
    #         (nn/atomic/tests/test_sf.py:159-311; used by its tests at :598-663)
    ns = functions_from(
        f"{REF}/tensoralloy/nn/atomic/tests/test_sf.py",
        ["cutoff_fxn", "get_neighbour_list", "get_radial_fingerprints_v1",
         "get_radial_fingerprints_v2", "get_augular_fingerprints_v1"],
        {"np": np, "product": itertools.product})
    sym, coords = read_xyz(f"{REF}/test_files/B28.xyz")
    rr = np.sqrt(((coords[:, None, :] - coords[None, :, :]) ** 2).sum(-1))
    rc = 6.0
    etas = [0.05, 4.0, 20.0, 80.0]
    omegas = [0.0, 3.2]
    betas, gammas, zetas = [0.005], [1.0, -1.0], [1.0, 4.0]
    g2_v1 = ns["get_radial_fingerprints_v1"](coords, rr, rc, etas)
    g2_v2 = ns["get_radial_fingerprints_v2"](coords, rr, rc, etas, omegas)
    g4 = ns["get_augular_fingerprints_v1"](coords, rr, rc, betas, gammas, zetas)
    np.savez(os.path.join(HERE, "B28_sf.npz"), coords=coords, rc=rc, etas=etas, omegas=omegas,
             betas=betas, gammas=gammas, zetas=zetas, g2_v1=g2_v1, g2_v2=g2_v2, g4=g4)
    out["B28"] = [g2_v1.shape, g2_v2.shape, g4.shape]

    # ---- 2. AMP-generated descriptors of the periodic Pd3O2 slab (data file of the reference,
    #         used at nn/atomic/tests/test_sf.py:666-691)
    g = np.load(f"{REF}/test_files/amp_Pd3O2.npz")["g"]
    np.savez(os.path.join(HERE, "amp_Pd3O2.npz"), g=g)

    # ---- 3. scalar cutoff definitions (nn/tests/test_cutoff.py:24-68)
    ns = functions_from(f"{REF}/tensoralloy/nn/tests/test_cutoff.py",
                        ["cosine_cutoff_simple", "polynomial_cutoff_simple"], {"np": np})
    r = np.linspace(0.0, 10.0, 201)
    cos6 = np.array([ns["cosine_cutoff_simple"](x, 6.0) for x in r])
    pol6 = np.array([ns["polynomial_cutoff_simple"](x, 6.0, 5.0) for x in r])
    np.savez(os.path.join(HERE, "cutoffs.npz"), r=r, rc=6.0, cosine=cos6, polynomial=pol6)

    # ---- 4. k-body term orderings and Szudzik pairing from the reference's utils module
    #         (tensoralloy/utils.py is TensorFlow-free and imports cleanly)
    utils = module_from(f"{REF}/tensoralloy/utils.py", "ref_utils")
    cases = []
    for els in (["A"], ["A", "B"], ["A", "B", "C"], ["C", "A", "B"], ["Ni"], ["Mo", "Ni"],
                ["Pd", "O"], ["Al", "Cu", "Ni", "Mo"]):
        for angular in (False, True):
            for symmetric in (True, False):
                a, k, e = utils.get_kbody_terms(els, angular=angular, symmetric=symmetric)
                cases.append(dict(elements=els, angular=angular, symmetric=symmetric,
                                  all_terms=a, terms_for_element=k, sorted_elements=e))
    x = np.array([1, 3, 5, 8, 9, -2, -7, 0, 4, -1])
    y = np.array([3, 4, 2, 2, 1, 5, -3, 0, -6, -1])
    z = np.array([0, -1, 2, -3, 4, -5, 6, -7, 8, -9])
    sz = dict(x=x.tolist(), y=y.tolist(), z=z.tolist(),
              xy=np.asarray(utils.szudzik_pairing(x, y)).tolist(),
              xyz=np.asarray(utils.szudzik_pairing(x, y, z)).tolist(),
              scalars=[int(utils.szudzik_pairing(int(a), int(b), int(c)))
                       for a, b, c in zip(x, y, z)])
    split = {t: utils.get_elements_from_kbody_term(t)
             for t in ("AlCu", "CuH", "HH", "HHCu", "HHH", "NiMoMo")}
    with open(os.path.join(HERE, "kbody_terms.json"), "w") as fp:
        json.dump(dict(cases=cases, szudzik=sz, split=split), fp, indent=0)

    # ---- 5. VirtualAtomMap outputs from the reference class (transformer/vap.py is NumPy-only)
    vapmod = module_from(f"{REF}/tensoralloy/transformer/vap.py", "ref_vap")
    from collections import Counter
    vcases = []
    for symbols, occ in (("Pd3O2", {"Pd": 4, "O": 5}), ("Pd2O2Pd", {"Pd": 4, "O": 5}),
                         ("Pd3O2", {"Pd": 3, "O": 2}), ("Ni4", {"Ni": 4, "Mo": 1}),
                         ("NiMoNi2Mo", {"Ni": 3, "Mo": 2})):
        import re
        syms = []
        for s, n in re.findall(r"([A-Z][a-z]*)(\d*)", symbols):
            syms.extend([s] * (int(n) if n else 1))
        vap = vapmod.VirtualAtomMap(Counter(occ), syms)
        arr = np.arange(1, len(syms) + 1, dtype=float).reshape(-1, 1) * np.array([[1.0, 10.0, 100.0]])
        fwd = vap.map_array(arr)
        rev = vap.map_array(fwd, reverse=True)
        vcases.append(dict(symbols=syms, max_occurs=occ, max_vap_natoms=int(vap.max_vap_natoms),
                           vap_symbols=list(vap.vap_symbols),
                           atom_masks=vap.atom_masks.astype(int).tolist(),
                           local_to_gsl=[int(vap.local_to_gsl_map[i + 1]) for i in range(len(syms))],
                           forward=fwd.tolist(), reverse=rev.tolist()))
    # literal expectations asserted in transformer/tests/test_vap.py:44-60
    literal = dict(forward_of_1to5=[0, 4, 5, 0, 0, 0, 1, 2, 3, 0],
                   masks=[0, 1, 1, 0, 0, 0, 1, 1, 1, 0], local_to_gsl_1=6)
    with open(os.path.join(HERE, "vap.json"), "w") as fp:
        json.dump(dict(cases=vcases, literal=literal), fp, indent=0)

    # ---- 6. EAM tables: samples of the setfl files the reference asserts its Zjw04 graph against
    #         (nn/eam/tests/test_eam_alloy_nn.py:139-164) and of Zhou's Ni table
    from oracle.eam import read_setfl
    eam = {}
    for tag, fn in (("AlCu", "Zhou_AlCu.alloy.eam"), ("Ni", "zjw04_Ni.alloy.eam")):
        t = read_setfl(f"{REF}/test_files/lammps/{fn}")
        step_r, step_rho = max(1, t["nr"] // 200), max(1, t["nrho"] // 200)
        eam[tag] = dict(elements=t["elements"], dr=t["dr"], drho=t["drho"], rcut=t["rcut"],
                        r_index=list(range(0, t["nr"], step_r)),
                        rho_index=list(range(0, t["nrho"], step_rho)),
                        rho={e: t["rho"][e][::step_r].tolist() for e in t["elements"]},
                        embed={e: t["embed"][e][::step_rho].tolist() for e in t["elements"]},
                        rphi={k: v[::step_r].tolist() for k, v in t["rphi"].items()})
    # literal values asserted in io/tests/test_lammps.py:25-71
    eam["literal"] = dict(F_Al_10=-1.8490865619220642e-01, rphi_CuCu_1=3.8671050028993639)
    adp = read_setfl(f"{REF}/test_files/lammps/AlCu.adp", adp=True)
    eam["AlCu_adp"] = dict(nr=adp["nr"], nrho=adp["nrho"], dr=adp["dr"], drho=adp["drho"],
                           w_AlCu_0_6=adp["w"]["AlCu"][:7].tolist(),
                           u_AlAl_absmax=float(np.abs(adp["u"]["AlAl"]).max()),
                           w_CuCu_absmax=float(np.abs(adp["w"]["CuCu"]).max()))
    with open(os.path.join(HERE, "eam_tables.json"), "w") as fp:
        json.dump(eam, fp)
    # the setfl data file itself (a data file the reference's tests hold), gzip'ed, for the table
    # reader and the spline-tabulated models; and the Al-Cu ADP file thinned to every 10th knot
    # (3 MB otherwise), rewritten in the same layout
    import gzip
    with open(f"{REF}/test_files/lammps/Zhou_AlCu.alloy.eam", "rb") as fi, \
            gzip.GzipFile(os.path.join(HERE, "Zhou_AlCu.alloy.eam.gz"), "wb", mtime=0) as fo:
        fo.write(fi.read())
    k = 10
    with open(f"{REF}/test_files/lammps/AlCu.adp") as fp:
        head = [fp.readline() for _ in range(5)]
    els = adp["elements"]
    adp_lines = [head[0], head[1], "thinned to every 10th knot by tests/golden/make_golden.py\n", head[3],
           f"{adp['nrho'] // k} {adp['drho'] * k!r} {adp['nr'] // k} {adp['dr'] * k!r} {adp['rcut']!r}\n"]
    zs = {"Al": "13 26.982 4.05 fcc\n", "Cu": "29 63.546 3.615 fcc\n"}
    for el in els:
        adp_lines.append(zs[el])
        adp_lines += ["%.16e\n" % v for v in adp["embed"][el][::k]]
        adp_lines += ["%.16e\n" % v for v in adp["rho"][el][::k]]
    order = [(i, j) for i in range(len(els)) for j in range(i + 1)]
    for group in ("rphi", "u", "w"):
        for i, j in order:
            adp_lines += ["%.16e\n" % v for v in adp[group][els[j] + els[i]][::k]]
    with gzip.GzipFile(os.path.join(HERE, "AlCu_thinned.adp.gz"), "wb", mtime=0) as fo:
        fo.write("".join(adp_lines).encode())

    # Agrawal's Be table (the LAMMPS side of potentials/tests/test_agrawal.py:52-58): rho(r) and
    # r phi(r) at every 5th knot; F(rho) in full (it is singular at rho = 0, and the ringing of
    # a natural spline around that knot dies out by a factor 0.27 per interval: with fewer knots it
    # would reach the densities that occur)
    with open(f"{REF}/test_files/lammps/Be_Agrawal.eam.alloy", "rb") as fp:
        raw = fp.read().decode().replace("\r", "").split("\n")
    nrho_b, drho_b, nr_b, dr_b, rc_b = raw[4].split()[:5]
    nrho_b, nr_b, drho_b, dr_b = int(nrho_b), int(nr_b), float(drho_b), float(dr_b)
    tok = " ".join(raw[6:]).split()
    F_b = np.array(tok[:nrho_b], dtype=float)
    rho_b = np.array(tok[nrho_b:nrho_b + nr_b], dtype=float)
    rphi_b = np.array(tok[nrho_b + nr_b:nrho_b + 2 * nr_b], dtype=float)
    k = 5
    be_lines = ["Be, Agrawal et al., Modelling Simul. Mater. Sci. Eng. 2013 (Be_Agrawal.eam.alloy)\n",
                "thinned by tests/golden/make_golden.py (r tables: every 5th knot)\n", "\n", "1 Be\n",
                f"{nrho_b} {drho_b!r} {nr_b // k} {dr_b * k!r} {float(rc_b)!r}\n", raw[5].strip() + "\n"]
    be_lines += ["%.10e\n" % v for v in F_b]
    for arr in (rho_b, rphi_b):
        be_lines += ["%.16e\n" % v for v in arr[::k]]
    with gzip.GzipFile(os.path.join(HERE, "Be_Agrawal_thinned.eam.alloy.gz"), "wb", mtime=0) as fo:
        fo.write("".join(be_lines).encode())

    # ---- 7. Reference-generated Hessian of Zjw04 Ni (nn/constraint/tests/test_fc2.py:29-54)
    #         with the structure it was computed for (test_files/crystals/Ni_sc.cif, P1)
    fc2 = np.load(f"{REF}/test_files/crystals/Ni_fc2.npy")
    frac, a_len = [], None
    with open(f"{REF}/test_files/crystals/Ni_sc.cif") as fp:
        for ln in fp:
            t = ln.split()
            if ln.startswith("_cell_length_a"):
                a_len = float(t[1])
            if len(t) == 7 and t[0] == "Ni":
                frac.append([float(t[3]), float(t[4]), float(t[5])])
    frac = np.array(frac)
    assert len(frac) == fc2.shape[0] == 32
    np.savez(os.path.join(HERE, "Ni_fc2.npz"), fc2=fc2.astype(np.float32), frac=frac,
             cell=np.eye(3) * a_len)
    out["fc2"] = fc2.shape

    # ---- 8. Neighbour statistics the reference cached in snap-Ni.db (io/sqlite.py:234-298)
    from oracle.neighbors import neighbor_list
    con = sqlite3.connect(f"{REF}/tensoralloy/data/datasets/snap-Ni.db")
    meta = json.loads(con.execute("select value from information where name='metadata'").fetchone()[0])
    rows = con.execute("select id, numbers, positions, cell, pbc from systems order by id").fetchall()
    structs = []
    for rid, numbers, positions, cell, pbc in rows:
        numbers = np.frombuffer(numbers, dtype=np.int32)
        pos = np.frombuffer(positions, dtype=np.float64).reshape(-1, 3)
        cel = np.frombuffer(cell, dtype=np.float64).reshape(3, 3)
        structs.append((rid, numbers, pos, cel, [bool(pbc & 1), bool(pbc & 2), bool(pbc & 4)]))
    stats = {}
    keep = {}
    for key, rcv in (("450", 4.5), ("460", 4.6), ("600", 6.0), ("650", 6.5)):
        best = dict(nij=(-1, None), nnl=(-1, None), nijk=(-1, None), ij2k=(-1, None))
        for rid, numbers, pos, cel, pbc in structs:
            i, j, S = neighbor_list(pos, cel, pbc, rcv)
            cnt = np.bincount(i, minlength=len(pos))
            vals = dict(nij=len(i), nnl=int(cnt.max()), nijk=int((cnt * (cnt - 1) // 2).sum()),
                        ij2k=int(cnt.max()) - 1)
            for k, v in vals.items():
                if v > best[k][0]:
                    best[k] = (v, rid)
        ref2 = meta["neighbors"]["2"][key]
        assert best["nij"][0] == ref2["nij_max"], (key, best, ref2)
        assert best["nnl"][0] == ref2["nnl_max"], (key, best, ref2)
        if key in meta["neighbors"]["3"]:
            ref3 = meta["neighbors"]["3"][key]
            assert best["nijk"][0] == ref3["nijk_max"], (key, best, ref3)
            if "ij2k_max" in ref3:
                assert best["ij2k"][0] == ref3["ij2k_max"], (key, best, ref3)
        stats[key] = {k: dict(value=v[0], structure_id=v[1]) for k, v in best.items()}
        for k in best:
            keep[best[k][1]] = True
    arrays = {}
    for rid, numbers, pos, cel, pbc in structs:
        if rid in keep:
            arrays[f"pos_{rid}"] = pos
            arrays[f"cell_{rid}"] = cel
            arrays[f"pbc_{rid}"] = np.array(pbc)
    np.savez(os.path.join(HERE, "snap_Ni_neighbors.npz"), **arrays)
    with open(os.path.join(HERE, "snap_Ni_neighbors.json"), "w") as fp:
        json.dump(dict(n_structures=len(structs), stats=stats,
                       atomic_static_energy_Ni=meta["atomic_static_energy"]["Ni"]), fp, indent=0)
    out["snap"] = stats

    # ---- 9. qm7m molecules with the sizes asserted in tests/test_neighbor.py:20-36
    with open(f"{REF}/test_files/datasets/qm7m/qm7m.xyz") as fp:
        lines = fp.read().split("\n")
    mols, k = [], 0
    while k < len(lines) and lines[k].strip():
        n = int(lines[k])
        syms = [ln.split()[0] for ln in lines[k + 2:k + 2 + n]]
        pos = [[float(x) for x in ln.split()[1:4]] for ln in lines[k + 2:k + 2 + n]]
        mols.append(dict(symbols=syms, positions=pos))
        k += 2 + n
    with open(os.path.join(HERE, "qm7m.json"), "w") as fp:
        json.dump(dict(molecules=mols,
                       expected={"id2": dict(nij=20, nnl=4), "id3": dict(nij=56, nijk=168, nnl=6)},
                       rc=6.5), fp)
    out["qm7m"] = [len(m["symbols"]) for m in mols]

    # ---- 10. snap_Ni_id11.extxyz (BASELINE config 1 structure; labels are DFT, not model output)
    with open(f"{REF}/test_files/snap_Ni_id11.extxyz") as fp:
        txt = fp.read()
    with open(os.path.join(HERE, "snap_Ni_id11.extxyz"), "w") as fp:
        fp.write(txt)
    print(json.dumps(out, default=str, indent=1))


def expand_cif(path):
    """Cell + all sites of a CIF: the `_symmetry_equiv_pos_as_xyz` operators applied to the listed
    sites, duplicates removed modulo 1 (SURVEY section 10 recipe; the reference reads these files through
    ASE / pymatgen, tensoralloy/io/read.py)."""
    import re
    from fractions import Fraction
    with open(path) as fp:
        lines = [ln.strip() for ln in fp if ln.strip() and not ln.startswith("#")]
    val = {}
    for ln in lines:
        t = ln.split()
        if t[0].startswith("_cell_") and len(t) == 2:
            val[t[0]] = float(t[1])
    ops = [m.group(1) for ln in lines for m in [re.match(r"^\d+\s+'([^']+)'$", ln)] if m]
    sites = []
    for ln in lines:
        t = ln.split()
        if len(t) == 7 and re.match(r"^[A-Z][a-z]?$", t[0]):
            sites.append((t[0], [float(t[3]), float(t[4]), float(t[5])]))

    def apply(op, xyz):
        out = []
        for expr in op.split(","):
            expr = expr.strip()
            v = 0.0
            for sign, num, den, var in re.findall(r"([+-]?)(?:(\d+)/(\d+)|([xyz]))", expr):
                term = float(Fraction(int(num), int(den))) if num else xyz["xyz".index(var)]
                v += -term if sign == "-" else term
            out.append(v % 1.0)
        return out

    symbols, frac = [], []
    for sym, xyz in sites:
        for op in ops:
            q = apply(op, xyz)
            if not any(s == sym and np.all(np.abs(((np.array(q) - f) + 0.5) % 1.0 - 0.5) < 1e-4)
                       for s, f in zip(symbols, frac)):
                symbols.append(sym)
                frac.append(np.array(q))
    a, b, c = val["_cell_length_a"], val["_cell_length_b"], val["_cell_length_c"]
    al, be, ga = (np.radians(val[k]) for k in ("_cell_angle_alpha", "_cell_angle_beta", "_cell_angle_gamma"))
    cx = c * np.cos(be)
    cy = c * (np.cos(al) - np.cos(be) * np.cos(ga)) / np.sin(ga)
    cell = np.array([[a, 0, 0], [b * np.cos(ga), b * np.sin(ga), 0], [cx, cy, np.sqrt(c * c - cx * cx - cy * cy)]])
    cell[np.abs(cell) < 1e-12] = 0.0
    return symbols, np.array(frac), cell


def make_graphdef_fixtures():
    """The reference's frozen-graph fixtures (binary model files: data, not source), gzip'ed."""
    import gzip
    for name in ("Ni", "Mo"):
        with open(f"{REF}/test_files/models/{name}.zhou04.pb", "rb") as fi, \
                gzip.GzipFile(os.path.join(HERE, f"{name}.zhou04.pb.gz"), "wb", compresslevel=9, mtime=0) as fo:
            fo.write(fi.read())


def make_nimo_cells():
    """BASELINE config 3 structures: the Ni-Mo conventional cells the reference ships
    (tensoralloy/data/crystals), expanded to explicit sites. Data only."""
    out = {}
    for name in ("Ni4Mo_mp-11507", "Ni3Mo_mp-11506"):
        sym, frac, cell = expand_cif(f"{REF}/tensoralloy/data/crystals/{name}_conventional_standard.cif")
        out[name] = dict(symbols=sym, scaled_positions=frac.round(12).tolist(), cell=cell.round(12).tolist())
        print(name, len(sym), {s: sym.count(s) for s in set(sym)})
    with open(os.path.join(HERE, "NiMo_cells.json"), "w") as fp:
        json.dump(out, fp, indent=0)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "nimo":
        make_nimo_cells()
    elif len(sys.argv) > 1 and sys.argv[1] == "graphdef":
        make_graphdef_fixtures()
    else:
        main()
        make_nimo_cells()
        make_graphdef_fixtures()
