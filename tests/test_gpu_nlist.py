"""
GPU neighbour list (tensoralloy_amd/csrc/ta_nlist.hip) against the oracle's
list (oracle/neighbors.py, itself pinned on the reference's snap-Ni statistics):
the sets of (i, j, S) must be identical -- integer work, bit-exact -- and the
energies / forces must not depend on which builder produced the list.
"""
import numpy as np
import pytest

from oracle import neighbors as onl
from tests.helpers import fcc, make_nn
from tensoralloy_amd import Atoms

pytestmark = pytest.mark.gpu


def _triplets(i, j, S):
    a = np.concatenate([np.asarray(i, np.int64)[:, None], np.asarray(j, np.int64)[:, None],
                        np.asarray(S, np.int64).reshape(-1, 3)], axis=1)
    return a[np.lexsort(a.T[::-1])]


def _device_pairs(nn, atoms_list, expect_device=True):
    from tensoralloy_amd import Engine
    with Engine(nn) as eng:
        info = eng.set_frames(atoms_list)
        assert bool(info.nl_on_device) == expect_device
        i, j, S = eng.pairs()
        return info, _triplets(i, j, S)


def _oracle_pairs(atoms_list, rc):
    out, off = [], 0
    for atoms in atoms_list:
        i, j, S = onl.neighbor_list(atoms.positions, np.asarray(atoms.get_cell(complete=True)), atoms.pbc, rc)
        out.append(_triplets(i + off, j + off, S))
        off += len(atoms)
    a = np.concatenate(out)
    return a[np.lexsort(a.T[::-1])]


def _sheared(rep=(7, 7, 7), seed=5):
    """Triclinic cell, atoms displaced far outside the cell (wrap shifts in S)."""
    atoms = fcc(rep=rep, seed=seed)
    cell = np.asarray(atoms.get_cell(complete=True)).copy()
    cell[1, 0] = 0.3 * cell[0, 0]
    cell[2, 1] = -0.2 * cell[1, 1]
    cell[2, 0] = 0.15 * cell[0, 0]
    frac = atoms.positions @ np.linalg.inv(np.diag(np.diag(cell)))
    rng = np.random.RandomState(seed)
    frac = frac + rng.randint(-2, 3, size=frac.shape)
    return Atoms(symbols=["Ni"] * len(frac), positions=frac @ cell, cell=cell, pbc=True)


def test_periodic_supercell_matches_oracle(lib):
    nn = make_nn(["Ni"], 6.5, True, [16])
    atoms = [fcc(rep=(6, 6, 6), seed=1)]
    info, dev = _device_pairs(nn, atoms)
    ref = _oracle_pairs(atoms, 6.5)
    assert info.n_pairs == len(ref)
    assert np.array_equal(dev, ref)
    counts = np.bincount(ref[:, 0], minlength=len(atoms[0]))
    assert info.nnl_max == counts.max()
    assert info.n_triples == int((counts * (counts - 1) // 2).sum())


def test_triclinic_wrapped_positions(lib):
    nn = make_nn(["Ni"], 6.5, True, [16])
    atoms = [_sheared()]
    info, dev = _device_pairs(nn, atoms)
    assert np.array_equal(dev, _oracle_pairs(atoms, 6.5))
    assert np.abs(dev[:, 2:]).max() >= 2  # shifts really carry the wrap


def test_slab_and_cluster_and_alloy_batch(lib):
    nn = make_nn(["Mo", "Ni"], 6.0, True, [16])
    slab = fcc(rep=(6, 6, 3), seed=2)
    slab.pbc = [True, True, False]
    cell = np.asarray(slab.get_cell(complete=True)).copy()
    cell[2, 2] += 15.0
    slab = Atoms(symbols=["Ni" if k % 3 else "Mo" for k in range(len(slab))], positions=slab.positions,
                 cell=cell, pbc=[True, True, False])
    cluster = fcc(rep=(5, 4, 3), seed=3)
    cluster = Atoms(symbols=["Mo" if k % 2 else "Ni" for k in range(len(cluster))],
                    positions=cluster.positions - 40.0, cell=np.eye(3) * 60.0, pbc=False)
    bulk = fcc(rep=(6, 6, 7), seed=4)
    bulk = Atoms(symbols=["Ni"] * len(bulk), positions=bulk.positions, cell=bulk.get_cell(complete=True),
                 pbc=True)
    frames = [slab, cluster, bulk]
    info, dev = _device_pairs(nn, frames)
    assert np.array_equal(dev, _oracle_pairs(frames, 6.0))


def test_small_cells_self_images_and_fallback(lib):
    nn = make_nn(["Ni"], 6.5, True, [16])
    # 7.05 A cell, rc 6.5: one bin per axis, every neighbour appears in several images
    atoms = [fcc(rep=(2, 2, 2))]
    info, dev = _device_pairs(nn, atoms, expect_device=True)
    assert np.array_equal(dev, _oracle_pairs(atoms, 6.5))
    # two bins along x, one along y and z, sheared
    cell = np.array([[14.5, 0.0, 0.0], [2.0, 7.2, 0.0], [0.5, -1.0, 6.9]])
    rng = np.random.RandomState(3)
    tri = [Atoms(symbols=["Ni"] * 40, positions=rng.rand(40, 3) @ cell * 1.7 - 3.0, cell=cell, pbc=True)]
    info, dev = _device_pairs(nn, tri, expect_device=True)
    assert np.array_equal(dev, _oracle_pairs(tri, 6.5))
    # cells thinner than rc along one, two or three periodic axes stay on the device (round 3: one bin
    # along the thin axis, images -m .. m of it; they took the host builder before): 3.5 A cubic cell
    # (m = 2 everywhere, every atom sees itself 4 and 6 cells away), the 2.48 x 2.48 x 24.3 A cell of
    # BASELINE config 1 (m = 4, 4, 1: 243 (bin, image) combinations), a sheared thin slab with atoms
    # given two cells outside the box, and a batch that mixes thin and thick frames
    thin = [fcc(rep=(1, 2, 2))]
    info, dev = _device_pairs(nn, thin, expect_device=True)
    assert np.array_equal(dev, _oracle_pairs(thin, 6.5))
    tiny = [fcc(rep=(1, 1, 1))]
    info, dev = _device_pairs(nn, tiny, expect_device=True)
    assert np.array_equal(dev, _oracle_pairs(tiny, 6.5))
    import os
    from tensoralloy_amd.io import read_extxyz
    c1 = read_extxyz(os.path.join(os.path.dirname(__file__), "golden", "snap_Ni_id11.extxyz"))[:1]
    info, dev = _device_pairs(nn, c1, expect_device=True)
    assert np.array_equal(dev, _oracle_pairs(c1, 6.5))
    cell = np.array([[3.1, 0.0, 0.0], [1.2, 9.0, 0.0], [0.4, -0.8, 16.0]])
    slab = [Atoms(symbols=["Ni"] * 30, positions=(rng.rand(30, 3) * 3.0 - 1.0) @ cell, cell=cell,
                  pbc=[True, True, False])]
    info, dev = _device_pairs(nn, slab, expect_device=True)
    assert np.array_equal(dev, _oracle_pairs(slab, 6.5))
    mixed = [fcc(rep=(1, 1, 2), seed=4), fcc(rep=(3, 3, 3), seed=5), c1[0]]
    info, dev = _device_pairs(nn, mixed, expect_device=True)
    assert np.array_equal(dev, _oracle_pairs(mixed, 6.5))


def test_results_do_not_depend_on_the_builder(lib, monkeypatch):
    from tensoralloy_amd import Engine
    nn = make_nn(["Mo", "Ni"], 6.5, True, [32, 32])
    base = fcc(rep=(6, 6, 6), seed=9)
    atoms = Atoms(symbols=["Mo" if k % 4 == 0 else "Ni" for k in range(len(base))],
                  positions=base.positions, cell=base.get_cell(complete=True), pbc=True)
    with Engine(nn) as eng:
        dev = eng.evaluate([atoms])[0]
        assert eng.info.nl_on_device == 1
    monkeypatch.setenv("TA_HOST_NL", "1")
    with Engine(nn) as eng:
        host = eng.evaluate([atoms])[0]
        assert eng.info.nl_on_device == 0
    assert abs(dev["energy"] - host["energy"]) < 1e-9
    assert np.abs(dev["forces"] - host["forces"]).max() < 1e-10
    assert np.abs(dev["virial"] - host["virial"]).max() < 1e-8


def test_eam_on_device_list(lib):
    from tests.helpers import make_eam, oracle_eam_eval
    from tensoralloy_amd import Engine
    nn = make_eam(["Cu", "Ni"])
    base = fcc(rep=(6, 6, 6), seed=11, a=3.57)
    atoms = Atoms(symbols=["Cu" if k % 2 else "Ni" for k in range(len(base))],
                  positions=base.positions, cell=base.get_cell(complete=True), pbc=True)
    with Engine(nn) as eng:
        r = eng.evaluate([atoms])[0]
        assert eng.info.nl_on_device == 1
    o = oracle_eam_eval(nn, atoms)
    assert abs(r["energy"] - o["energy"]) < 1e-6
    assert np.abs(r["forces"] - o["forces"]).max() < 1e-5


def test_reference_statistics_through_set_frames(lib):
    """nij / nnl / nijk maxima the reference cached in snap-Ni.db (io/sqlite.py:234-298), for the
    structures that attain them: what `ta_set_frames` reports (device or host builder, whichever
    the cell allows) must be exactly those numbers."""
    import json
    import os
    from tensoralloy_amd import Engine
    golden = os.path.join(os.path.dirname(__file__), "golden")
    with open(os.path.join(golden, "snap_Ni_neighbors.json")) as fp:
        meta = json.load(fp)
    z = np.load(os.path.join(golden, "snap_Ni_neighbors.npz"))
    on_device = 0
    for key, rc in (("450", 4.5), ("460", 4.6), ("600", 6.0), ("650", 6.5)):
        nn = make_nn(["Ni"], rc, True, [8])
        with Engine(nn) as eng:
            for stat, d in meta["stats"][key].items():
                rid = d["structure_id"]
                atoms = Atoms(symbols=["Ni"] * len(z[f"pos_{rid}"]), positions=z[f"pos_{rid}"],
                              cell=z[f"cell_{rid}"], pbc=z[f"pbc_{rid}"])
                info = eng.set_frames([atoms])
                on_device += int(info.nl_on_device)
                got = dict(nij=int(info.n_pairs), nnl=int(info.nnl_max), nijk=int(info.n_triples),
                           ij2k=int(info.nnl_max) - 1)[stat]
                assert got == d["value"], (key, stat, got, d)
    assert on_device > 0  # at least the large cells went through the GPU builder


def test_one_pass_builder_key_order_and_two_pass_fallback(lib, monkeypatch):
    """Round 3: the one-pass builder (ta_nlist.hip::build_pairs_kernel) leaves every centre's neighbours
    sorted by (species, j, S); the two-pass builder takes over on request and beyond 384 neighbours per
    atom. Same set of pairs from both, and the same energies / forces."""
    from tensoralloy_amd import Engine
    nn = make_nn(["Mo", "Ni"], 6.5, True, [16])
    base = _sheared(rep=(6, 6, 6), seed=8)
    sym = ["Mo" if k % 3 == 0 else "Ni" for k in range(len(base))]
    atoms = Atoms(symbols=sym, positions=base.positions, cell=base.get_cell(complete=True), pbc=True)
    sp = np.array([0 if s == "Mo" else 1 for s in sym])
    with Engine(nn) as eng:
        eng.set_frames([atoms])
        i, j, S = eng.pairs()
        one = eng.evaluate([atoms])[0]
    S = np.asarray(S).reshape(-1, 3)
    order = np.lexsort((S[:, 2], S[:, 1], S[:, 0], j, sp[j], i))
    assert np.array_equal(order, np.arange(len(i)))  # already in key order
    ref = _oracle_pairs([atoms], 6.5)
    assert np.array_equal(_triplets(i, j, S), ref)
    monkeypatch.setenv("TA_NL_TWO_PASS", "1")
    with Engine(nn) as eng:
        eng.set_frames([atoms])
        i2, j2, S2 = eng.pairs()
        two = eng.evaluate([atoms])[0]
    assert np.array_equal(_triplets(i2, j2, S2), ref)
    assert not np.array_equal(np.asarray(j2), np.asarray(j))  # traversal order, not key order
    assert abs(one["energy"] - two["energy"]) < 1e-9
    assert np.abs(one["forces"] - two["forces"]).max() < 1e-10
    monkeypatch.delenv("TA_NL_TWO_PASS")
    # 445 neighbours per atom: beyond the one-pass builder's LDS, the two-pass builder runs by itself
    wide = make_nn(["Ni"], 10.5, False, [8])
    big = [fcc(rep=(6, 6, 6), seed=2)]
    info, dev = _device_pairs(wide, big)
    assert info.nnl_max > 384
    assert np.array_equal(dev, _oracle_pairs(big, 10.5))
    # growing lists: the pair arrays are re-sized and the builder runs again
    with Engine(nn) as eng:
        a = eng.set_frames([fcc(rep=(3, 3, 3), seed=1)]).n_pairs
        b = eng.set_frames([atoms]).n_pairs
        assert b > 4 * a
        i3, j3, S3 = eng.pairs()
        assert np.array_equal(_triplets(i3, j3, S3), ref)
