"""
CPU: host-side logic of the product (no GPU compute): term ordering, pairing,
VAP maps, the library's neighbour list, the transformer's feed dict (checked
through the dense-layout oracle), model files, the Atoms / Calculator shims.
"""
import json
import os
from collections import Counter

import numpy as np
import pytest

from tests.conftest import GOLDEN
from tests.helpers import fcc, pd3o2, make_nn, make_eam, oracle_eval


def test_kbody_terms_and_pairing_match_reference():
    from tensoralloy_amd.utils import get_kbody_terms, szudzik_pairing, get_elements_from_kbody_term
    with open(os.path.join(GOLDEN, "kbody_terms.json")) as fp:
        g = json.load(fp)
    for c in g["cases"]:
        a, k, e = get_kbody_terms(c["elements"], angular=c["angular"], symmetric=c["symmetric"])
        assert a == c["all_terms"] and k == c["terms_for_element"] and e == c["sorted_elements"]
    s = g["szudzik"]
    x, y, z = np.array(s["x"]), np.array(s["y"]), np.array(s["z"])
    assert np.asarray(szudzik_pairing(x, y)).tolist() == s["xy"]
    assert np.asarray(szudzik_pairing(x, y, z)).tolist() == s["xyz"]
    assert [szudzik_pairing(int(a), int(b), int(c)) for a, b, c in zip(x, y, z)] == s["scalars"]
    assert np.asarray(szudzik_pairing(np.stack((x, y), 1))).tolist() == s["xy"]
    for term, parts in g["split"].items():
        assert get_elements_from_kbody_term(term) == parts


def test_parameter_grid_order():
    """sklearn ParameterGrid order used by the reference (sf.py:47-51): last key fastest."""
    from tensoralloy_amd.utils import parameter_grid
    g = parameter_grid(eta=[1, 2], omega=[0, 3])
    assert [(d["eta"], d["omega"]) for d in g] == [(1, 0), (1, 3), (2, 0), (2, 3)]
    g = parameter_grid(beta=[5], zeta=[1, 4], gamma=[1, -1])
    assert [(d["gamma"], d["zeta"]) for d in g] == [(1, 1), (1, 4), (-1, 1), (-1, 4)]


def test_vap_matches_reference():
    from tensoralloy_amd.transformer import VirtualAtomMap
    with open(os.path.join(GOLDEN, "vap.json")) as fp:
        g = json.load(fp)
    for c in g["cases"]:
        vap = VirtualAtomMap(Counter(c["max_occurs"]), c["symbols"])
        n = len(c["symbols"])
        assert vap.max_vap_natoms == c["max_vap_natoms"]
        assert list(vap.vap_symbols) == c["vap_symbols"]
        assert vap.atom_masks.astype(int).tolist() == c["atom_masks"]
        assert [vap.local_to_gsl_map[i + 1] for i in range(n)] == c["local_to_gsl"]
        arr = np.arange(1, n + 1, dtype=float).reshape(-1, 1) * np.array([[1.0, 10.0, 100.0]])
        fwd = vap.map_array(arr)
        assert fwd.tolist() == c["forward"]
        assert vap.map_array(fwd, reverse=True).tolist() == c["reverse"]
    # literal expectations of transformer/tests/test_vap.py:44-60
    lit = g["literal"]
    vap = VirtualAtomMap(Counter({"Pd": 4, "O": 5}), ["Pd"] * 3 + ["O"] * 2)
    assert vap.map_array(np.expand_dims([1, 2, 3, 4, 5], 1)).flatten().tolist() == lit["forward_of_1to5"]
    assert vap.atom_masks.astype(int).tolist() == lit["masks"]
    assert vap.local_to_gsl_map[1] == lit["local_to_gsl_1"]
    with pytest.raises(ValueError):
        vap.map_array(np.zeros((4, 3)))
    h = np.arange(10 * 3 * 10 * 3, dtype=float).reshape(10, 3, 10, 3)
    full = vap.reverse_map_hessian(h)
    idx = [vap.local_to_gsl_map[i + 1] for i in range(5)]
    assert full[1 * 3 + 2, 4 * 3 + 0] == h[idx[1], 2, idx[4], 0]
    assert vap.reverse_map_hessian(h, phonopy_format=True)[1, 4, 2, 0] == h[idx[1], 2, idx[4], 0]


def _same_pairs(lib_out, ora_out):
    i, j, s, r = lib_out
    a = sorted(zip(i.tolist(), j.tolist(), map(tuple, s.tolist())))
    b = sorted(zip(ora_out[0].tolist(), ora_out[1].tolist(), map(tuple, ora_out[2].tolist())))
    assert a == b
    assert all(i[r[p]] == j[p] and j[r[p]] == i[p] and (s[r[p]] == -s[p]).all() for p in range(len(i)))


def test_library_neighbor_list_matches_oracle(lib):
    from tensoralloy_amd import _lib
    from oracle.neighbors import neighbor_list
    rng = np.random.RandomState(0)
    a = pd3o2()
    _same_pairs(_lib.neighbor_list([1, 1, 1, 0, 0], a.positions, np.asarray(a.get_cell()), a.pbc, 2, 6.5),
                neighbor_list(a.positions, np.asarray(a.get_cell()), a.pbc, 6.5))
    lat = np.array([[2.479787, 0, 0], [-1.239893, 2.147558, 0], [0, 0, 24.294656]])
    p6 = rng.rand(6, 3) @ lat * 1.3 - 0.5          # tiny cell, atoms outside the box
    _same_pairs(_lib.neighbor_list([0] * 6, p6, lat, [1, 1, 1], 1, 6.0), neighbor_list(p6, lat, [1, 1, 1], 6.0))
    tri = np.array([[8, 0, 0], [2, 7, 0], [1, 1.5, 9.0]])
    pt = rng.rand(40, 3) @ tri * 1.2
    _same_pairs(_lib.neighbor_list(rng.randint(0, 3, 40), pt, tri, [1, 0, 1], 3, 5.0),
                neighbor_list(pt, tri, [1, 0, 1], 5.0))
    mol = rng.rand(28, 3) * 8
    _same_pairs(_lib.neighbor_list([0] * 28, mol, np.zeros((3, 3)), [0, 0, 0], 1, 6.0),
                neighbor_list(mol, np.zeros((3, 3)), [0, 0, 0], 6.0))
    lone = np.zeros((1, 3))
    i, j, s, r = _lib.neighbor_list([0], lone, np.eye(3) * 30, [1, 1, 1], 1, 6.5)
    assert len(i) == 0
    i, j, s, r = _lib.neighbor_list([0], lone, np.eye(3) * 3.0, [1, 1, 1], 1, 6.5)  # self-images only
    assert len(i) > 0 and np.all(i == 0) and np.all(j == 0) and not np.any(np.all(s == 0, axis=1))


def test_library_neighbor_list_reference_statistics(lib):
    """The product's list reproduces the sizes the reference cached in snap-Ni.db."""
    from tensoralloy_amd import _lib
    with open(os.path.join(GOLDEN, "snap_Ni_neighbors.json")) as fp:
        meta = json.load(fp)
    z = np.load(os.path.join(GOLDEN, "snap_Ni_neighbors.npz"))
    for key, rc in (("600", 6.0), ("650", 6.5)):
        for stat in ("nij", "nnl"):
            d = meta["stats"][key][stat]
            rid = d["structure_id"]
            pos = z[f"pos_{rid}"]
            i, j, s, r = _lib.neighbor_list([0] * len(pos), pos, z[f"cell_{rid}"], z[f"pbc_{rid}"], 1, rc)
            got = len(i) if stat == "nij" else int(np.bincount(i, minlength=len(pos)).max())
            assert got == d["value"]


def test_neighbor_list_rejects_bad_input(lib):
    from tensoralloy_amd import _lib
    with pytest.raises(ValueError):
        _lib.neighbor_list([0, 5], np.zeros((2, 3)), np.eye(3), [1, 1, 1], 1, 3.0)   # species range
    with pytest.raises(ValueError):
        _lib.neighbor_list([0], np.zeros((1, 3)), np.zeros((3, 3)) + 1.0, [1, 1, 1], 1, 3.0)  # singular
    with pytest.raises(ValueError):
        _lib.neighbor_list([0], np.full((1, 3), np.nan), np.eye(3), [1, 1, 1], 1, 3.0)


def test_host_neighbor_list_under_address_and_ub_sanitizers(tmp_path):
    """The threaded host list (csrc/ta_neighbor.cpp) compiled for the CPU with
    -fsanitize=address,undefined behind a small driver (tests/native/neighbor_sanitize.cpp): the
    cases of the tests above plus a multi-frame batch (one thread per frame) and the error paths run
    clean, keep the list's invariants (checked inside the driver) and give the oracle's pairs."""
    import shutil
    import subprocess
    from oracle.neighbors import neighbor_list
    if shutil.which("g++") is None:
        pytest.skip("no host compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "neighbor_sanitize")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-pthread", "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "tensoralloy_amd", "csrc"),
           os.path.join(root, "tests", "native", "neighbor_sanitize.cpp"),
           os.path.join(root, "tensoralloy_amd", "csrc", "ta_neighbor.cpp"), "-o", exe]
    built = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert built.returncode == 0, built.stderr[-3000:]

    def run(frames, n_elements, rc):
        lines = [f"{len(frames)} {n_elements} {rc!r}"]
        for species, pos, cell, pbc in frames:
            lines.append(f"{len(pos)} " + " ".join(str(int(bool(x))) for x in pbc) + " " +
                         " ".join(repr(float(x)) for x in np.asarray(cell, float).ravel()))
            for sp, r in zip(species, np.asarray(pos, float).reshape(-1, 3)):
                lines.append(f"{int(sp)} {float(r[0])!r} {float(r[1])!r} {float(r[2])!r}")
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
        env.pop("LD_PRELOAD", None)
        p = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, (p.stdout[-500:], p.stderr[-3000:])
        return p.stdout.split()

    def expect(frames, rc):
        # order-independent checksums of the oracle's list, frames concatenated as the library does
        P = si = sj = ss = trip = nnl = 0
        base = 0
        for species, pos, cell, pbc in frames:
            i, j, S = neighbor_list(pos, cell, pbc, rc)
            i, j = i + base, j + base
            P += len(i)
            si += int(i.sum())
            sj += int((j * (i % 7 + 1)).sum())
            ss += int(((S[:, 0] + 3 * S[:, 1] + 9 * S[:, 2]) * (j % 5 + 1)).sum()) if len(i) else 0
            cnt = np.bincount(i - base, minlength=len(pos))
            trip += int((cnt * (cnt - 1) // 2).sum())
            nnl = max(nnl, int(cnt.max()) if len(pos) else 0)
            base += len(pos)
        return P, trip, nnl, si, sj, ss

    rng = np.random.RandomState(0)
    a = pd3o2()
    lat = np.array([[2.479787, 0, 0], [-1.239893, 2.147558, 0], [0, 0, 24.294656]])
    tri = np.array([[8, 0, 0], [2, 7, 0], [1, 1.5, 9.0]])
    cases = [
        ([([1, 1, 1, 0, 0], a.positions, np.asarray(a.get_cell()), a.pbc)], 2, 6.5),
        ([([0] * 6, rng.rand(6, 3) @ lat * 1.3 - 0.5, lat, [1, 1, 1])], 1, 6.0),     # tiny cell, atoms outside
        ([(rng.randint(0, 3, 40), rng.rand(40, 3) @ tri * 1.2, tri, [1, 0, 1])], 3, 5.0),
        ([([0] * 28, rng.rand(28, 3) * 8, np.zeros((3, 3)), [0, 0, 0])], 1, 6.0),    # molecule, no cell
        ([([0], np.zeros((1, 3)), np.eye(3) * 3.0, [1, 1, 1])], 1, 6.5),             # self-images only
        ([([0], np.zeros((1, 3)), np.eye(3) * 30.0, [1, 1, 1])], 1, 6.5),            # no pairs at all
    ]
    many = []
    for f in range(12):                                                              # threads: frame per thread
        at = fcc(rep=(2, 2, 2), seed=f, jitter=0.1)
        many.append(([0] * len(at), at.positions, np.asarray(at.get_cell()), at.pbc))
    cases.append((many, 1, 6.0))
    cases.append(([], 1, 6.0))                                                       # empty batch
    for frames, nel, rc in cases:
        out = run(frames, nel, rc)
        assert out[0] == "pairs", out
        got = tuple(int(out[k]) for k in (1, 3, 5, 7, 9, 11))
        assert got == expect(frames, rc)
    # error paths: species out of range, singular periodic cell, NaN coordinate
    for frames in ([([0, 5], np.zeros((2, 3)), np.eye(3), [1, 1, 1])],
                   [([0], np.zeros((1, 3)), np.ones((3, 3)), [1, 1, 1])]):
        assert run(frames, 1, 3.0)[0] == "error"


@pytest.mark.parametrize("which", ["pd3o2", "ni"])
def test_feed_dict_through_dense_layout(lib, which):
    """`get_np_feed_dict` -> dense tensors (the reference's graph inputs) -> G2/G4 applied the way
    nn/atomic/sf.py does == packed oracle. Checks keys, dtypes, slot uniqueness, masks, GSL order."""
    from oracle.dense import descriptors_from_dense
    if which == "pd3o2":
        nn, atoms = make_nn(["Pd", "O"], 6.5, True, [8]), pd3o2()
    else:
        nn, atoms = make_nn(["Ni"], 4.6, True, [8]), fcc(rep=(2, 2, 2))
    clf = nn.transformer
    fd = clf.get_np_feed_dict(atoms)
    vap = clf.get_vap_transformer(atoms)
    expected_keys = {"positions", "cell", "volume", "n_atoms_vap", "nnl_max", "atom_masks",
                     "etemperature", "row_splits", "g2.v2g_map", "g2.ilist", "g2.jlist", "g2.n1",
                     "ij2k_max", "g4.v2g_map", "g4.ilist", "g4.jlist", "g4.klist", "g4.n1",
                     "g4.n2", "g4.n3"}
    assert set(fd) == expected_keys
    assert fd["g2.v2g_map"].dtype == np.int32 and fd["g2.v2g_map"].shape[1] == 5
    assert fd["positions"].shape == (vap.max_vap_natoms, 3) and np.all(fd["positions"][0] == 0)
    assert fd["row_splits"].tolist() == [1] + [vap.max_occurs[e] for e in clf.elements]
    for key in ("g2.v2g_map", "g4.v2g_map"):
        m = fd[key]
        assert len({tuple(x) for x in m[:, :4].tolist()}) == len(m)      # no two values share a slot
        assert m[:, 1].min() >= 1                                        # row 0 is the virtual atom
    assert np.allclose(fd["g4.n3"], fd["g4.n2"] - fd["g4.n1"])
    uni = clf.get_descriptors(fd)
    d = nn.descriptor.as_dict()
    dense = descriptors_from_dense(uni, clf.elements, clf.rcut, clf.acut, d["eta"], d["omega"],
                                   d["beta"], d["gamma"], d["zeta"])
    G = oracle_eval(nn, atoms, want_forces=False)["descriptors"]
    syms = np.array(atoms.get_chemical_symbols())
    for el in clf.elements:
        idx = np.where(syms == el)[0]   # GSL keeps the local order inside an element
        assert np.abs(dense[el][:len(idx)] - G[idx]).max() < 1e-10
        assert np.all(uni["atom_masks"][el][:len(idx)] == 1)


def test_transformer_surface():
    from tensoralloy_amd import UniversalTransformer
    clf = UniversalTransformer(["Ni", "Mo"], rcut=6.5, angular=True)
    assert clf.elements == ["Mo", "Ni"] and clf.acut == 6.5 and clf.descriptor == "universal"
    assert clf.kbody_terms_for_element["Ni"] == ["NiNi", "NiMo", "NiMoMo", "NiMoNi", "NiNiNi"]
    d = clf.as_dict()
    assert d == {"class": "UniversalTransformer", "elements": ["Mo", "Ni"], "rcut": 6.5, "acut": 6.5,
                 "angular": True, "periodic": True, "symmetric": True, "use_computed_dists": True}
    d.pop("class")
    assert UniversalTransformer(**d).as_dict()["rcut"] == 6.5
    with pytest.raises(ValueError):
        UniversalTransformer(["Xx"], 6.5)
    a = pd3o2()
    with pytest.raises(ValueError):
        clf.species_indices(a)          # Pd / O are not in this model
    # get_placeholder_features (transformer/base.py:191): the keys get_np_feed_dict fills, in the
    # reference's order, with its dtypes and static shapes (universal.py:728-785)
    for kw in (dict(angular=True), dict(angular=False), dict(angular=True, use_computed_dists=False)):
        t = UniversalTransformer(["Pd", "O"], rcut=4.0, **kw)
        ph = t.get_placeholder_features()
        feed = t.get_np_feed_dict(a)
        assert list(ph) == list(feed)
        for k, spec in ph.items():
            v = np.asarray(feed[k])
            assert spec.name == f"Placeholders/{k}:0" and v.dtype == spec.dtype and v.ndim == len(spec.shape)
            assert all(d is None or d == n for d, n in zip(spec.shape, v.shape)), (k, spec.shape, v.shape)
    assert ph["g4.rijk"].shape == (12, None) and ph["row_splits"].shape == (3,)


def test_model_file_roundtrip(tmp_path):
    from tensoralloy_amd import load_model
    nn = make_nn(["Pd", "O"], 6.5, True, {"Pd": [8, 8], "O": [4]}, minmax=True, resnet=True,
                 static_energy={"Pd": -1.0})
    path = nn.export(str(tmp_path / "model.pb"))
    assert path.endswith(".json") and os.path.exists(str(tmp_path / "model.npz"))
    nn2, clf2, meta = load_model(str(tmp_path / "model.pb"))
    assert nn2.as_dict() == nn.as_dict() and clf2.as_dict() == nn.transformer.as_dict()
    assert meta["Metadata/api"] == "1.1" and meta["Metadata/precision"] == "high"
    assert json.loads(json.dumps(meta["Transformer/params"]))["class"] == "UniversalTransformer"
    for el in nn.elements:
        for (w, b), (w2, b2) in zip(nn.weights[el], nn2.weights[el]):
            assert np.array_equal(w, w2) and (b is None) == (b2 is None)
            if b is not None:
                assert np.array_equal(np.ravel(b), np.ravel(b2))
        assert np.array_equal(nn.minmax[el][0], nn2.minmax[el][0])
    # C ABI description of the loaded model
    desc, keep = nn2.to_desc()
    assert desc.n_elements == 2 and desc.angular == 1 and desc.minmax_scale == 1
    assert [desc.n_layers[k] for k in range(2)] == [2, 3]          # O: [4]+out, Pd: [8,8]+out
    # EAM
    e = make_eam(["Ni", "Mo"], 6.0, adp=True)
    p = e.export(str(tmp_path / "adp"))
    e2, _, _ = load_model(p)
    assert e2.as_dict() == e.as_dict()
    assert np.array_equal(e2.flat_parameters(), e.flat_parameters()) and len(e.flat_parameters()) == 2 * 22 + 3 * 8 + 3 * 8
    with pytest.raises(FileNotFoundError):
        load_model(str(tmp_path / "missing.pb"))
    (tmp_path / "other.pb").write_bytes(b"\x0a\x05\x0a\x03abc")  # a GraphDef, but not a tensoralloy export
    with pytest.raises(ValueError, match="Transformer/params"):
        load_model(str(tmp_path / "other.pb"))
    (tmp_path / "bad.json").write_text(json.dumps({"format": "other"}))
    with pytest.raises(ValueError):
        load_model(str(tmp_path / "bad.json"))


def test_model_argument_checks():
    from tensoralloy_amd import AtomicNN, SymmetryFunction, UniversalTransformer
    from tensoralloy_amd.eam import EamAlloyNN
    with pytest.raises(ValueError):
        AtomicNN(["Ni"], SymmetryFunction(["Ni"]), activation="linear")   # nn/utils.py:73-74
    with pytest.raises(ValueError):
        SymmetryFunction(["Ni"], cutoff_function="tanh")
    with pytest.raises(ValueError):
        EamAlloyNN(["Ni"], custom_potentials="msah11")                     # an eam/fs potential
    with pytest.raises(ValueError, match="no constants"):
        EamAlloyNN(["Ni"], custom_potentials="sutton90").flat_parameters() # AgSutton90 knows Ag only
    ag = EamAlloyNN(["Ag"], custom_potentials="sutton90")
    assert list(ag.flat_parameters()[:2]) == [2.928323832, 2.485883762] and ag.flat_parameters()[21] == 1.0
    with pytest.raises(ValueError, match="mix potentials"):
        EamAlloyNN(["Ag"], custom_potentials={"Ag": {"rho": "sutton90", "embed": "zjw04"}, "AgAg": {"phi": "sutton90"}})
    d = EamAlloyNN(["Mo", "Ni"])                                          # default: all "nn" (alloy.py:110-112)
    assert d.potentials == {"Mo": {"rho": "nn", "embed": "nn"}, "Ni": {"rho": "nn", "embed": "nn"},
                            "MoMo": {"phi": "nn"}, "MoNi": {"phi": "nn"}, "NiNi": {"phi": "nn"}}
    assert d.hidden_sizes["MoNi"] == {"phi": [64, 32]}                     # Defaults.hidden_sizes
    assert d.nn_functions() == [("Mo", "rho"), ("Ni", "rho"), ("Mo", "embed"), ("Ni", "embed"),
                                ("MoMo", "phi"), ("MoNi", "phi"), ("NiNi", "phi")]
    d.attach_transformer(UniversalTransformer(["Mo", "Ni"], rcut=6.0))
    with pytest.raises(ValueError, match="no weights"):
        d.to_desc()
    d.initialize()
    desc, keep = d.to_desc()
    assert desc.n_eam_nets == 7 and [desc.n_layers[k] for k in range(7)] == [3] * 7
    assert [desc.layer_sizes[k] for k in range(4)] == [1, 64, 32, 1]
    assert d.weights["Ni"]["rho"][-1][1] is None                          # no output bias (eam.py:184-190)
    mixed = EamAlloyNN(["Ni"], custom_potentials={"Ni": {"rho": "zjw04"}, "NiNi": {"phi": "zjw04"}},
                       hidden_sizes={"Ni": {"embed": [8]}})
    assert mixed.nn_functions() == [None, ("Ni", "embed"), None] and mixed.hidden_sizes["Ni"]["embed"] == [8]
    with pytest.raises(ValueError, match="one Zjw04 variant"):
        EamAlloyNN(["Ni"], custom_potentials={"Ni": {"rho": "zjw04", "embed": "zjw04xc"},
                                              "NiNi": {"phi": "zjw04"}})
    xc = EamAlloyNN(["Be", "Ni"], custom_potentials="zjw04xc")          # Be := Mo (zjw04.py:436-438)
    assert xc.family == "zjw04xc" and xc.element_parameters("Be") == EamAlloyNN(
        ["Mo"], custom_potentials="zjw04").element_parameters("Mo")
    xcp = EamAlloyNN(["Mo", "Ni"], custom_potentials="zjw04xcp")
    assert xcp.element_parameters("Ni")["rho_e"] == 25.423122           # the refit, zjw04.py:621-626
    flat = xcp.flat_parameters()
    assert len(flat) == 2 * 22 + 3 * 8 and flat[20] == 1.0 and flat[42] == 1.0 and flat[21] == 0.0
    assert list(flat[44:52]) == [0.0] * 8 and flat[52] == 1.0 and flat[53] == 2.235219
    nn = AtomicNN(["Ni"], SymmetryFunction(["Ni"]), hidden_sizes=[4])
    with pytest.raises(ValueError):
        nn.ndim()                                                         # no transformer attached
    nn.attach_transformer(UniversalTransformer(["Ni"], 6.0, angular=True, symmetric=False))
    nn.initialize()
    desc, _ = nn.to_desc()                                                # one element: allowed
    assert desc.eps == 1e-14 and list(nn.descriptor_scale()) == [1.0] * 4 + [2.0] * 4
    nn.precision = "medium"
    assert nn.to_desc()[0].eps == 1e-8
    nn2 = AtomicNN(["Mo", "Ni"], SymmetryFunction(["Mo", "Ni"]), hidden_sizes=[4])
    nn2.attach_transformer(UniversalTransformer(["Mo", "Ni"], 6.0, angular=True, symmetric=False))
    nn2.initialize()
    with pytest.raises(ValueError):
        nn2.to_desc()                                                     # as sf.py:131-132


def test_atoms_and_calculator_shims():
    from tensoralloy_amd.atoms import Atoms, Calculator, compare_atoms, PropertyNotImplementedError
    a = pd3o2()
    assert a.get_chemical_formula(mode="reduce") == "Pd3O2"
    b = Atoms(symbols=["Pd", "Pd", "O", "O", "Pd"], positions=a.positions, cell=np.asarray(a.get_cell()))
    assert b.get_chemical_formula(mode="reduce") == "Pd2O2Pd"
    assert abs(a.get_volume() - 7.78 * 5.50129076 * 15.37532269) < 1e-9
    mol = Atoms(symbols="B2", positions=[[0, 0, 0], [1, 0, 0]])
    assert np.allclose(np.asarray(mol.get_cell(complete=True)), np.eye(3))
    assert compare_atoms(a, a.copy()) == []
    c = a.copy(); c.positions[0, 0] += 0.1
    assert compare_atoms(a, c) == ["positions"]
    assert len(a * (2, 1, 1)) == 10

    class Toy(Calculator):
        implemented_properties = ["energy"]

        def calculate(self, atoms=None, properties=("energy",), system_changes=()):
            Calculator.calculate(self, atoms, properties, system_changes)
            self.results = {"energy": float(len(atoms))}
            self.n = getattr(self, "n", 0) + 1

    t = Toy()
    assert t.get_potential_energy(a) == 5.0 and t.get_potential_energy(a) == 5.0 and t.n == 1
    assert t.get_potential_energy(c) == 5.0 and t.n == 2          # state change -> recalculation
    with pytest.raises(PropertyNotImplementedError):
        t.get_forces(a)


def test_extxyz_config1_structure():
    """BASELINE config 1 input: snap_Ni_id11.extxyz (6 Ni atoms, 2.48 A hexagonal cell)."""
    from tensoralloy_amd.io import read_extxyz
    frames = read_extxyz(os.path.join(GOLDEN, "snap_Ni_id11.extxyz"))
    a = frames[0]
    assert len(a) == 6 and a.get_chemical_symbols() == ["Ni"] * 6 and a.pbc.all()
    assert abs(a.info["energy"] + 33.38981773) < 1e-9
    assert a.info["forces"].shape == (6, 3)
    assert abs(np.asarray(a.get_cell())[2, 2] - 24.294656) < 1e-9


def test_lammps_native_npz_roundtrip(tmp_path):
    """The reference's TF-free interchange format (atomic.py:304-480): keys, dtypes, conventions."""
    from tests.helpers import make_grap_nn
    from tensoralloy_amd import load_model
    nn = make_grap_nn(["W", "Be"], 5.0, [16, 16], moment_tensors=[0, 1, 2, 3], symmetric=True,
                      cutoff="polynomial")
    path = nn.export_to_lammps_native(str(tmp_path / "BeW.npz"))
    z = np.load(path)
    assert z["rmax"] == 5.0 and z["nelt"] == 2 and z["precision"] == 64 and z["use_fnn"] == 0
    assert list(z["numbers"]) == [ord("B"), ord("e"), ord("W"), 0]           # atomic.py:362-369
    assert z["descriptor::method"] == 0 and len(z["descriptor::rl"]) == 10   # pexp
    assert z["nlayers"] == 3 and list(z["layer_sizes"]) == [16, 16, 1]
    assert z["max_moment"] == 3 and z["fctype"] == 1 and z["is_T_symmetric"] == 1 and z["actfn"] == 1
    assert z["weights_0_0"].shape == (nn.ndim(), 16) and z["weights_1_2"].shape == (16,)
    assert z["biases_0_2"].shape == (1,) and abs(z["masses"][0] - 9.0121831) < 1e-6
    nn2, clf2, meta = load_model(path)
    assert nn2.elements == ["Be", "W"] and clf2.rcut == 5.0 and not clf2.angular
    assert nn2.descriptor.as_dict()["moment_tensors"] == [0, 1, 2, 3] and nn2.descriptor.is_T_symmetric
    assert nn2.descriptor.algorithm.as_dict(True)["parameters"] == nn.descriptor.algorithm.as_dict(True)["parameters"]
    for el in nn.elements:
        for (w, b), (w2, b2) in zip(nn.weights[el], nn2.weights[el]):
            assert np.array_equal(np.asarray(w), w2) and np.array_equal(np.ravel(b), b2)
    assert set(meta["Metadata/ops"]) >= {"energy", "forces", "stress"}
    d1, _ = nn.to_desc()
    d2, _ = nn2.to_desc()
    assert d1.kind == d2.kind == 4 and d1.n_grap_params == d2.n_grap_params
    with pytest.raises(ValueError):
        make_nn(["Ni"], 6.0, False, [8]).export_to_lammps_native(str(tmp_path / "sf.npz"))


def test_setfl_readers_against_the_reference_literals(tmp_path):
    """io/tests/test_lammps.py:25-71: values the reference's own reader test asserts for
    `Zhou_AlCu.alloy.eam`; the thinned Al-Cu ADP file keeps every 10th knot of `AlCu.adp`
    (w_AlCu[0] = 0.26740386..., u / w identically zero for Al-Al and Cu-Cu)."""
    import gzip
    from tests.helpers import golden_setfl
    from tensoralloy_amd.io import read_adp_setfl, read_eam_alloy_setfl, natural_spline_coefficients
    fl = read_eam_alloy_setfl(golden_setfl("Zhou_AlCu.alloy.eam", tmp_path))
    assert fl.elements == ["Al", "Cu"] and fl.nr == 2000 and fl.nrho == 2000
    assert fl.dr == 0.003 and fl.drho == 0.05 and fl.rcut == 6.0
    assert fl.embed["Al"].y[10] == -1.8490865619220642e-01
    assert abs(fl.phi["CuCu"].y[1] * fl.phi["CuCu"].x[1] - 3.8671050028993639) < 1e-15
    assert sorted(fl.phi) == ["AlAl", "AlCu", "CuCu"] and fl.pair("phi", "Cu", "Al") is fl.phi["AlCu"]
    assert fl.atomic_masses[0] == 26.98 and fl.lattice_types == ["fcc", "fcc"]
    # the .gz is read directly as well
    import os
    gz = os.path.join(os.path.dirname(__file__), "golden", "Zhou_AlCu.alloy.eam.gz")
    assert np.array_equal(read_eam_alloy_setfl(gz).rho["Cu"].y, fl.rho["Cu"].y)
    with open(os.path.join(os.path.dirname(__file__), "golden", "eam_tables.json")) as fp:
        lit = json.load(fp)["AlCu_adp"]
    adp = read_adp_setfl(golden_setfl("AlCu_thinned.adp", tmp_path))
    assert adp.nr == lit["nr"] // 10 and adp.nrho == lit["nrho"] // 10
    assert abs(adp.dr - 10 * lit["dr"]) < 1e-18 and abs(adp.drho - 10 * lit["drho"]) < 1e-18
    assert abs(adp.quadrupole["AlCu"].y[0] - lit["w_AlCu_0_6"][0]) < 1e-15
    assert np.abs(adp.dipole["AlAl"].y).max() == 0.0 and np.abs(adp.quadrupole["CuCu"].y).max() == 0.0
    # natural cubic spline: interpolates the knots, second derivative zero at both ends,
    # agrees with SciPy's natural spline between the knots
    from scipy.interpolate import CubicSpline
    sp = fl.rho["Al"]
    c = natural_spline_coefficients(sp.x, sp.y)
    assert np.array_equal(c[:, 0], sp.y[:-1]) and c[0, 2] == 0.0
    h = sp.x[1] - sp.x[0]
    assert abs(2 * c[-1, 2] + 6 * c[-1, 3] * h) < 1e-9
    xx = np.random.RandomState(0).rand(500) * sp.x[-1]
    k = np.minimum((xx / h).astype(int), len(sp.x) - 2)
    t = xx - sp.x[k]
    v = c[k, 0] + t * (c[k, 1] + t * (c[k, 2] + t * c[k, 3]))
    assert np.abs(v - CubicSpline(sp.x, sp.y, bc_type="natural")(xx)).max() < 1e-13


def test_spline_potentials_in_the_model_description(tmp_path):
    from tests.helpers import golden_setfl
    from tensoralloy_amd import UniversalTransformer
    from tensoralloy_amd.eam import AdpNN, EamAlloyNN
    path = golden_setfl("Zhou_AlCu.alloy.eam", tmp_path)
    nn = EamAlloyNN.from_setfl(path)
    assert nn.elements == ["Al", "Cu"] and nn.potentials["AlCu"] == {"phi": "spline@" + path}
    nn.attach_transformer(UniversalTransformer(["Al", "Cu"], rcut=6.0))
    desc, keep = nn.to_desc()
    assert desc.n_eam_nets == 7 and [desc.eam_table_n[k] for k in range(7)] == [2000] * 7
    assert abs(desc.eam_table_dx[0] - 0.003) < 1e-15 and abs(desc.eam_table_dx[2] - 0.05) < 1e-15
    assert nn.nn_functions() == [None] * 7
    mixed = EamAlloyNN(["Al", "Cu"], custom_potentials={
        "Al": {"rho": "spline@" + path, "embed": "zjw04"}, "Cu": {"rho": "zjw04", "embed": "zjw04"},
        "AlAl": {"phi": "zjw04"}, "AlCu": {"phi": "spline@" + path}, "CuCu": {"phi": "zjw04"}})
    mixed.attach_transformer(UniversalTransformer(["Al", "Cu"], rcut=6.0))
    desc, keep = mixed.to_desc()
    assert [desc.eam_table_n[k] for k in range(7)] == [2000, 0, 0, 0, 0, 2000, 0]
    with pytest.raises(ValueError, match="no rho table"):
        bad = EamAlloyNN(["Al", "Ni"], custom_potentials={
            "Ni": {"rho": "spline@" + path, "embed": "zjw04"}, "Al": {"rho": "zjw04", "embed": "zjw04"},
            "AlAl": {"phi": "zjw04"}, "AlNi": {"phi": "zjw04"}, "NiNi": {"phi": "zjw04"}})
        bad.attach_transformer(UniversalTransformer(["Al", "Ni"], rcut=6.0))
        bad.to_desc()
    adp = AdpNN.from_setfl(golden_setfl("AlCu_thinned.adp", tmp_path))
    adp.attach_transformer(UniversalTransformer(["Al", "Cu"], rcut=6.0))
    desc, keep = adp.to_desc()
    assert desc.n_eam_nets == 13 and all(desc.eam_table_n[k] == 1000 for k in range(13))


def test_batch_universal_transformer_records():
    """`BatchUniversalTransformer` (universal.py:921-1388): padded records with the reference's
    keys and shapes; decoded and scattered they give, structure by structure, the dense arrays of
    the single-structure transformer (whose layout the oracle tests pin), padded to the data set's
    `nnl_max` / `ij2k_max`."""
    from collections import Counter
    from tensoralloy_amd import UniversalTransformer
    from tensoralloy_amd.transformer import BatchUniversalTransformer
    from tests.test_gpu_sf import _alloy
    frames = [_alloy(["Ni", "Ni", "Mo"], rep=(1, 1, 2), a=3.6), _alloy(["Ni", "Mo"], rep=(1, 1, 1), a=3.7, seed=5)]
    frames[0].info["energy"] = -12.5
    frames[0].info["forces"] = np.arange(3 * len(frames[0]), dtype=float).reshape(-1, 3)
    max_occurs = Counter()
    for a in frames:
        for el, n in Counter(a.get_chemical_symbols()).items():
            max_occurs[el] = max(max_occurs[el], n)
    single = UniversalTransformer(["Mo", "Ni"], rcut=4.0, angular=True)
    feeds = [single.get_np_feed_dict(a) for a in frames]
    nij = max(len(f["g2.ilist"]) for f in feeds)
    nijk = max(len(f["g4.ilist"]) for f in feeds)
    nnl = max(int(f["nnl_max"]) for f in feeds)
    ij2k = max(int(f["ij2k_max"]) for f in feeds)
    clf = BatchUniversalTransformer(max_occurs, rcut=4.0, angular=True, nij_max=nij + 3, nijk_max=nijk + 5,
                                    nnl_max=nnl, ij2k_max=ij2k, batch_size=2, use_forces=True, use_stress=True)
    d = clf.as_dict()
    assert d["class"] == "BatchUniversalTransformer" and d["nij_max"] == nij + 3 and d["use_stress"] is True
    assert clf.as_descriptor_transformer().as_dict() == single.as_dict()
    recs = [clf.encode(a) for a in frames]
    n_vap = sum(max_occurs.values()) + 1
    for a, r in zip(frames, recs):
        assert r["positions"].shape == (n_vap, 3) and r["atom_masks"].shape == (n_vap,)
        assert int(r["n_atoms_vap"]) == len(a)                                  # base.py:411
        assert r["g2.indices"].shape == (nij + 3, 7) and r["g2.indices"].dtype == np.int32
        assert r["g4.indices"].shape == (nijk + 5, 8) and r["g4.shifts"].shape == (nijk + 5, 9)
        assert r["forces"].shape == (n_vap, 3) and r["stress"].shape == (6,)
        assert r["atom_masks"].sum() == len(a)
    assert recs[0]["energy"][0] == -12.5 and recs[1]["energy"][0] == 0.0
    vap0 = clf.get_vap_transformer(frames[0])
    assert np.array_equal(vap0.map_forces(recs[0]["forces"], reverse=True), frames[0].info["forces"])
    batch = clf.batch(recs)
    assert batch["g2.v2g_map"].shape == (2, nij + 3, 5) and batch["g4.klist"].shape == (2, nijk + 5)
    dense = clf.get_descriptors(batch)
    for b, a in enumerate(frames):
        # the same structure through the single-structure transformer, with ITS vap (own counts)
        ref = single.get_descriptors(feeds[b])
        vap_b, vap_s = clf.get_vap_transformer(a), single.get_vap_transformer(a)
        for el in ("Mo", "Ni"):
            n_own = vap_s.max_occurs[el]
            got = dense["radial"][el][0][:, b]                                   # [4, nr, n_el, nnl, 1]
            own = ref["radial"][el][0]
            real = np.flatnonzero(dense["atom_masks"][el][b])
            assert len(real) == Counter(a.get_chemical_symbols())[el]
            k = own.shape[3]
            # atoms of one element keep their relative order in both maps
            assert np.allclose(got[:, :, real, :k], own[:, :, :len(real)], atol=1e-12)
            assert np.abs(got[:, :, :, k:]).max() < 1e-6 if got.shape[3] > k else True
            ga = dense["angular"][el][0][:, b]
            oa = ref["angular"][el][0]
            assert np.allclose(ga[:, :, real, :oa.shape[3], :oa.shape[4]], oa[:, :, :len(real)], atol=1e-12)
            assert n_own >= 1
    back = clf.frames(batch, [a.get_chemical_symbols() for a in frames])
    for a, c in zip(frames, back):
        assert np.allclose(a.positions, c.positions) and np.allclose(np.asarray(a.get_cell()), np.asarray(c.get_cell()))
    with pytest.raises(ValueError, match="more than the declared maximum"):
        BatchUniversalTransformer(max_occurs, rcut=4.0, nij_max=5).encode(frames[0])
    with pytest.raises(ValueError, match="exceed max_occurs"):
        BatchUniversalTransformer(Counter({"Ni": 1, "Mo": 1}), rcut=4.0).encode(frames[0])


def test_graphdef_reader_on_the_reference_fixtures():
    """The reference's own frozen graphs (test_files/models/{Ni,Mo}.zhou04.pb, gzip'ed copies under
    tests/golden/) are read without TensorFlow: transformer JSON, metadata, the Zjw04 constants of
    the frozen shared variables; the loaded model is the `EamAlloyNN(['Ni'], 'zjw04')` the file was
    exported from (constants = zjw04_defaults, zjw04.py:19-152)."""
    from tensoralloy_amd.graphdef import read_graph_model, read_node_ops
    from tensoralloy_amd.model import load_model
    from tensoralloy_amd.eam import ZJW04_DEFAULTS, ZJW04_KEYS
    meta, consts = read_graph_model(os.path.join(GOLDEN, "Ni.zhou04.pb.gz"))
    params = json.loads(meta["Transformer/params"])
    assert params == {"class": "UniversalTransformer", "elements": ["Ni"], "rcut": 6.5, "acut": None,
                      "angular": False, "periodic": True, "symmetric": True, "use_computed_dists": True}
    ops = json.loads(meta["Metadata/ops"])
    assert ops["forces"] == "Output/Forces/forces:0" and ops["hessian"] == "Output/Hessian/hessian:0"
    assert meta["Metadata/precision"] == "high" and meta["Metadata/tf_version"] == "1.15.0"
    assert float(consts["EAM/Shared/Ni/r_eq"]) == 2.488746 and float(consts["EAM/Shared/Ni/Fe"]) == -2.699486
    assert read_node_ops(os.path.join(GOLDEN, "Mo.zhou04.pb.gz"))["Transformer/params"] == "Const"
    for el in ("Ni", "Mo"):
        nn, clf, m = load_model(os.path.join(GOLDEN, f"{el}.zhou04.pb.gz"))
        assert type(nn).__name__ == "EamAlloyNN" and nn.family == "zjw04" and clf.rcut == 6.5
        got = nn.element_parameters(el)
        for key, val in zip(ZJW04_KEYS, ZJW04_DEFAULTS[el]):
            assert got[key] == pytest.approx(val, rel=1e-12), key
        assert ("hessian" in m["Metadata/ops"]) == (el == "Ni")
    with pytest.raises(ValueError):   # not a protobuf at all
        read_graph_model(os.path.join(GOLDEN, "cutoffs.npz"))
