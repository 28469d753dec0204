"""Shared builders for tests: structures and seeded models."""
import numpy as np

from tensoralloy_amd import Atoms, AtomicNN, SymmetryFunction, UniversalTransformer


def fcc(symbol="Ni", a=3.524, rep=(2, 2, 2), jitter=0.05, seed=611):
    base = np.array([[0, 0, 0], [.5, .5, 0], [.5, 0, .5], [0, .5, .5]]) * a
    pts = np.array([base + np.array([x, y, z]) * a
                    for x in range(rep[0]) for y in range(rep[1]) for z in range(rep[2])]).reshape(-1, 3)
    rng = np.random.RandomState(seed)
    pts = pts + rng.normal(0.0, jitter, pts.shape)
    return Atoms(symbols=[symbol] * len(pts), positions=pts,
                 cell=np.diag([a * rep[0], a * rep[1], a * rep[2]]), pbc=True)


def hcp(symbol="Be", a=2.29, c=3.58, rep=(3, 3, 2), jitter=0.1, seed=0):
    cell = np.array([[a, 0, 0], [-a / 2, a * np.sqrt(3) / 2, 0], [0, 0, c]])
    basis = np.array([[0, 0, 0], [1 / 3, 2 / 3, 0.5]])
    pts = np.array([(b + np.array([i, j, k])) @ cell for i in range(rep[0]) for j in range(rep[1])
                    for k in range(rep[2]) for b in basis])
    pts = pts + np.random.RandomState(seed).randn(*pts.shape) * jitter
    return Atoms(symbols=[symbol] * len(pts), positions=pts, cell=cell * np.array(rep)[:, None], pbc=True)


def pd3o2():
    return Atoms(symbols="Pd3O2", pbc=[True, True, False],
                 cell=np.array([[7.78, 0., 0.], [0., 5.50129076, 0.], [0., 0., 15.37532269]]),
                 positions=np.array([[3.89, 0., 8.37532269], [0., 2.75064538, 8.37532269],
                                     [3.89, 2.75064538, 8.37532269], [5.835, 1.37532269, 8.5],
                                     [5.835, 7.12596807, 8.]]))


def make_nn(elements, rcut, angular, hidden, activation="softplus", acut=None, seed=611,
            minmax=False, resnet=False, bias_scale=0.1, cutoff="cosine", sf_kwargs=None,
            static_energy=None, symmetric=True, precision="high"):
    clf = UniversalTransformer(elements, rcut=rcut, acut=acut, angular=angular, symmetric=symmetric)
    sf = SymmetryFunction(elements, cutoff_function=cutoff, **(sf_kwargs or {}))
    nn = AtomicNN(elements, sf, hidden_sizes=hidden, activation=activation,
                  minmax_scale=minmax, use_resnet_dt=resnet,
                  atomic_static_energy=static_energy or {},
                  export_properties=("energy", "forces", "stress"))
    nn.attach_transformer(clf)
    nn.precision = precision
    nn.initialize(seed=seed, bias_scale=bias_scale)
    if minmax:
        rng = np.random.RandomState(seed + 1)
        D = nn.ndim()
        for el in nn.elements:
            nn.minmax[el] = (rng.rand(D) * 0.1, 1.0 + rng.rand(D) * 5.0)
    return nn


def oracle_model(nn):
    """The same model as an oracle `SFModel` (test infrastructure only)."""
    from oracle.sf import SFModel
    clf, sf = nn.transformer, nn.descriptor
    d = sf.as_dict()
    m = SFModel(nn.elements, clf.rcut, acut=clf.acut, angular=clf.angular, eta=d["eta"],
                omega=d["omega"], beta=d["beta"], gamma=d["gamma"], zeta=d["zeta"],
                cutoff_function=d["cutoff_function"], hidden_sizes=nn.hidden_sizes,
                activation=nn._activation, weights=nn.weights, use_resnet_dt=nn._use_resnet_dt,
                minmax=nn.minmax if nn._minmax_scale else None, symmetric=clf.symmetric,
                safe_pow=nn.use_custom_pow)
    return m


def oracle_eval(nn, atoms, want_forces=True):
    from oracle.sf import evaluate
    eps = 1e-8 if getattr(nn, "precision", "high") == "medium" else 1e-14  # precision.py:113-114
    return evaluate(oracle_model(nn), atoms.get_chemical_symbols(), atoms.positions,
                    np.asarray(atoms.get_cell(complete=True)), atoms.pbc, want_forces=want_forces,
                    eps=eps)


def make_eam(elements, rcut=6.5, adp=False, potential="zjw04", parameters=None, hidden_sizes=None,
             activation=None, seed=611, out_scale=0.05):
    """`potential`: a family name, a nested dict, or None = the reference's default: every
    function an "nn" function (alloy.py:110-112, adp.py:120-124)."""
    from tensoralloy_amd.eam import EamAlloyNN, AdpNN
    clf = UniversalTransformer(elements, rcut=rcut, angular=False)
    if adp:
        pots = _adp_pots(elements) if potential == "zjw04" else potential
        nn = AdpNN(elements, custom_potentials=pots, parameters=parameters, hidden_sizes=hidden_sizes,
                   activation=activation)
    else:
        nn = EamAlloyNN(elements, custom_potentials=potential, parameters=parameters,
                        hidden_sizes=hidden_sizes, activation=activation)
    nn.attach_transformer(clf)
    if any(s is not None for s in nn.nn_functions()):
        nn.initialize(seed=seed, bias_scale=0.1)
    # keep randomly initialised nn functions at the scale of physical ones (rho_i of order 1-10,
    # energies of order eV), so that absolute tolerances mean what they mean for real models
    for sec in nn.weights.values():
        for layers in sec.values():
            w, b = layers[-1]
            layers[-1] = (w * out_scale, b)
    return nn


def _adp_pots(elements):
    els = sorted(elements)
    pots = {el: {"rho": "zjw04", "embed": "zjw04"} for el in els}
    for i, a in enumerate(els):
        for b in els[i:]:
            pots[a + b] = {"phi": "zjw04", "dipole": "mishinh", "quadrupole": "mishinh"}
    return pots


def oracle_eam_eval(nn, atoms):
    """Oracle counterpart of an EamAlloyNN / AdpNN (test infrastructure only)."""
    from oracle.eam import evaluate
    m = oracle_eam_model(nn)
    eps = 1e-8 if getattr(nn, "precision", "high") == "medium" else 1e-14
    return evaluate(m, atoms.get_chemical_symbols(), atoms.positions,
                    np.asarray(atoms.get_cell(complete=True)), atoms.pbc, eps=eps)


def oracle_eam_model(nn):
    from oracle.eam import EamModel
    from tensoralloy_amd.eam import AdpNN, ADP_KEYS
    adp = None
    if isinstance(nn, AdpNN):
        adp = {}
        els = nn.elements
        for i, a in enumerate(els):
            for b in els[i:]:
                if nn.pair_parameters(a + b) is not None:
                    adp[a + b] = nn.pair_parameters(a + b)
    nets, tables = {}, {}
    for slot in nn.nn_functions():
        if slot is not None:
            sec, fn = slot
            nets.setdefault(fn, {})[sec] = nn.weights[sec][fn]
    for sec, fn in nn._all_slots():
        if nn.is_spline(sec, fn):
            sp = nn.spline_table(sec, fn)
            tables.setdefault(fn, {})[sec] = (sp.x, sp.y)
    phi_pairs = {}
    els = nn.elements
    for i, a in enumerate(els):
        for b in els[i + 1:]:
            q = nn.phi_parameters(a, b)
            if q is not None:
                phi_pairs[a + b] = q
    if isinstance(nn, AdpNN) and not adp:
        adp = {}
    other = {el: (nn._el_kind[el], nn.other_parameters(el)) for el in nn.elements if nn._el_kind[el] != "zjw"}
    return EamModel(nn.elements, nn.transformer.rcut, other=other,
                    params={el: nn.element_parameters(el) for el in nn.elements if el not in other}, adp=adp,
                    blended_embed=nn.family != "zjw04", phi_pairs=phi_pairs, nets=nets,
                    activation=nn._activation, tables=tables)


PEXP = {"rl": [1.0, 1.15, 1.3, 1.45, 1.6, 1.75, 1.9, 2.05, 2.2, 2.35],
        "pl": [5.0, 4.75, 4.5, 4.25, 4.0, 3.75, 3.5, 3.25, 3.0, 2.75]}  # test_grap.py:52-53


def make_grap_nn(elements, rcut, hidden, algorithm="pexp", parameters=None, moment_tensors=(0, 1, 2),
                 legacy_mode=False, symmetric=False, cutoff="cosine", param_space_method="pair",
                 minmax=False, seed=611, precision="high"):
    from tensoralloy_amd.grap import GenericRadialAtomicPotential
    clf = UniversalTransformer(elements, rcut=rcut, angular=False)
    if algorithm == "nn" and parameters is None:
        parameters = {}
    grap = GenericRadialAtomicPotential(elements, algorithm, parameters=PEXP if parameters is None else parameters,
                                        param_space_method=param_space_method,
                                        moment_tensors=list(moment_tensors), cutoff_function=cutoff,
                                        symmetric=symmetric, legacy_mode=legacy_mode)
    nn = AtomicNN(elements, grap, hidden_sizes=hidden, minmax_scale=minmax,
                  export_properties=("energy", "forces", "stress"))
    nn.attach_transformer(clf)
    nn.precision = precision
    nn.initialize(seed=seed, bias_scale=0.1)
    if algorithm == "nn":
        grap.initialize_filters(seed=seed + 7, bias_scale=0.1)
        w, b = grap.filter_weights[-1]
        grap.filter_weights[-1] = (w * 0.2, b)   # filter values of order 1, like the analytic ones
    if minmax:
        rng = np.random.RandomState(seed + 1)
        D = nn.ndim()
        for el in nn.elements:
            nn.minmax[el] = (rng.rand(D) * 0.1 - 0.05, 1.0 + rng.rand(D) * 5.0)
    return nn


def oracle_grap_model(nn):
    from oracle.grap import GrapModel
    d = nn.descriptor.as_dict()
    filter_net = None
    if d["algorithm"] == "nn":
        a = nn.descriptor.algorithm
        filter_net = dict(layers=nn.descriptor.filter_weights, activation=a.activation,
                          use_resnet_dt=a.use_resnet_dt, h_abck_modifier=a.h_abck_modifier)
    return GrapModel(nn.elements, nn.transformer.rcut, algorithm=d["algorithm"], parameters=d["parameters"],
                     filter_net=filter_net,
                     param_space_method=d["param_space_method"], moment_tensors=d["moment_tensors"],
                     cutoff_function=d["cutoff_function"], symmetric=d["symmetric"],
                     legacy_mode=d["legacy_mode"], weights=nn.weights, activation=nn._activation,
                     use_resnet_dt=nn._use_resnet_dt, minmax=nn.minmax if nn._minmax_scale else None)


def oracle_grap_eval(nn, atoms):
    from oracle.grap import evaluate
    eps = 1e-8 if getattr(nn, "precision", "high") == "medium" else 1e-14
    return evaluate(oracle_grap_model(nn), atoms.get_chemical_symbols(), atoms.positions,
                    np.asarray(atoms.get_cell(complete=True)), atoms.pbc, eps=eps)


def golden_setfl(name, tmp_path):
    """Unpack one of the gzip'ed setfl fixtures (tests/golden/) and return its path."""
    import gzip
    import os
    src = os.path.join(os.path.dirname(__file__), "golden", name + ".gz")
    dst = os.path.join(str(tmp_path), name)
    with gzip.open(src, "rb") as fi, open(dst, "wb") as fo:
        fo.write(fi.read())
    return dst


def nimo_supercell(name="Ni4Mo_mp-11507", rep=(7, 7, 8), jitter=0.05, seed=611):
    """Supercell of one of the Ni-Mo conventional cells the reference ships
    (tensoralloy/data/crystals/*.cif, expanded by tests/golden/make_golden.py into
    tests/golden/NiMo_cells.json): BASELINE config 3 is Ni4Mo 7x7x8 = 3920 atoms."""
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "NiMo_cells.json")) as fp:
        c = json.load(fp)[name]
    cell = np.array(c["cell"])
    frac = np.array(c["scaled_positions"])
    shifts = np.array([[x, y, z] for x in range(rep[0]) for y in range(rep[1]) for z in range(rep[2])])
    pts = ((frac[None, :, :] + shifts[:, None, :]).reshape(-1, 3)) @ cell
    if jitter:
        pts = pts + np.random.RandomState(seed).normal(0.0, jitter, pts.shape)
    return Atoms(symbols=list(c["symbols"]) * len(shifts), positions=pts,
                 cell=cell * np.array(rep)[:, None], pbc=True)
