"""Pin the oracle with the reference's own golden outputs.

Inputs: the reference's test structures / bond files / split XTC trajectories (tests/golden/*.npz, the
lipid-only subset, integer coordinates as stored in the XTC files).  Expected: the YAML files the
reference's integration tests compare against, with the reference's own tolerance (floats within 2e-4,
/root/reference/tests/common/mod.rs:139-150).  Every trig mode of the oracle must reproduce them.
"""
import numpy as np
import pytest

from gorder_amd import structure as st
from oracle import oracle
from golden_util import METHODS, Fixture, aa_setup, cg_setup, expected, ua_setup


@pytest.fixture(scope="module")
def pcpepg(built):
    return Fixture("pcpepg")


@pytest.fixture(scope="module")
def cg(built):
    return Fixture("cg")


def run(tables, fx, frames, trig, n_threads=1):
    eng = oracle.OracleEngine(tables, trig=trig, n_threads=n_threads)
    xyz = fx.xyz[frames][:, :, :]
    eng.submit(np.ascontiguousarray(xyz[:, : tables.n_atoms] if xyz.shape[1] == tables.n_atoms else xyz),
               fx.boxes[frames], frames)
    return eng.finish()


def master_frames(fx, midx, frames):
    return np.ascontiguousarray(fx.xyz[frames][:, midx, :])


@pytest.mark.parametrize("trig", [oracle.TRIG_LIBM, oracle.TRIG_MIRROR, oracle.TRIG_DIRECT])
def test_aa_order_basic(pcpepg, trig):
    # tests_aa.rs:47-77: 51 frames (5 concatenated files), no leaflets
    tables, labels, midx = aa_setup(pcpepg)
    assert [m.name for m in labels] == ["POPE", "POPC", "POPG"]
    assert [m.n_molecules for m in labels] == [131, 128, 15]           # aaorder.rs:413
    assert [len(m.bonds) for m in labels] == [73, 82, 74]               # molecule.rs:525
    frames = pcpepg.window()
    assert len(frames) == 51                                            # tests_aa.rs:5455
    eng = oracle.OracleEngine(tables, trig=trig, n_threads=3)
    eng.submit(master_frames(pcpepg, midx, frames), pcpepg.boxes[frames], frames)
    res = eng.finish()
    assert res.n_frames == 51
    tree = st.results_tree(res, labels, "aa", leaflets=False)
    bad = st.compare_trees(tree, expected("aa_order_basic.yaml"))
    assert not bad, bad[:10]


@pytest.mark.parametrize("method", ["global", "local", "individual"])
def test_aa_order_leaflets(pcpepg, method):
    # tests_aa.rs:289-317: global / local(2.5 nm) / individual all reproduce aa_order_leaflets.yaml
    tables, labels, midx = aa_setup(pcpepg, leaflets=METHODS[method])
    frames = pcpepg.window()
    eng = oracle.OracleEngine(tables, trig=oracle.TRIG_LIBM, n_threads=4)
    eng.submit(master_frames(pcpepg, midx, frames), pcpepg.boxes[frames], frames)
    tree = st.results_tree(eng.finish(), labels, "aa", leaflets=True)
    bad = st.compare_trees(tree, expected("aa_order_leaflets.yaml"))
    assert not bad, bad[:10]


def test_aa_begin_end_step_leaflets(pcpepg):
    # tests_aa.rs:1271-1304: begin 450200 ps, end 450400 ps, step 3, global leaflets -> 4 frames
    tables, labels, midx = aa_setup(pcpepg, leaflets=METHODS["global"], frequency=3)   # real frequency = 1 * step
    frames = pcpepg.window(450_200.0, 450_400.0, 3)
    assert len(frames) == 4
    eng = oracle.OracleEngine(tables, trig=oracle.TRIG_LIBM)
    # SystemTopology::frame advances by step (topology/mod.rs:141-144): 0, 3, 6, 9
    eng.submit(master_frames(pcpepg, midx, frames), pcpepg.boxes[frames], np.arange(4) * 3)
    tree = st.results_tree(eng.finish(), labels, "aa", leaflets=True)
    bad = st.compare_trees(tree, expected("aa_order_begin_end_step.yaml"))
    assert not bad, bad[:10]


@pytest.mark.parametrize("trig", [oracle.TRIG_LIBM, oracle.TRIG_DIRECT])
def test_cg_order_basic(cg, trig):
    tables, labels, midx = cg_setup(cg)
    assert [m.name for m in labels] == ["POPC", "POPE", "POPG"]
    assert [m.n_molecules for m in labels] == [242, 242, 24]            # cgorder.rs counts
    assert [len(m.bonds) for m in labels] == [11, 11, 11]
    frames = cg.window()
    eng = oracle.OracleEngine(tables, trig=trig, n_threads=2)
    eng.submit(master_frames(cg, midx, frames), cg.boxes[frames], frames)
    tree = st.results_tree(eng.finish(), labels, "cg", leaflets=False)
    bad = st.compare_trees(tree, expected("cg_order_basic.yaml"))
    assert not bad, bad[:10]


@pytest.mark.parametrize("method", ["global", "local", "individual"])
def test_cg_order_leaflets(cg, method):
    # tests_cg.rs:180-205
    tables, labels, midx = cg_setup(cg, leaflets=METHODS[method])
    frames = cg.window()
    eng = oracle.OracleEngine(tables, trig=oracle.TRIG_LIBM, n_threads=4)
    eng.submit(master_frames(cg, midx, frames), cg.boxes[frames], frames)
    tree = st.results_tree(eng.finish(), labels, "cg", leaflets=True)
    bad = st.compare_trees(tree, expected("cg_order_leaflets.yaml"))
    assert not bad, bad[:10]


def test_cg_begin_end_step(cg):
    # tests_cg.rs:808-845: 352-358 ns, step 5, global leaflets -> 13 frames
    tables, labels, midx = cg_setup(cg, leaflets=METHODS["global"], frequency=5)
    frames = cg.window(352_000.0, 358_000.0, 5)
    assert len(frames) == 13
    eng = oracle.OracleEngine(tables, trig=oracle.TRIG_LIBM)
    eng.submit(master_frames(cg, midx, frames), cg.boxes[frames], np.arange(13) * 5)
    tree = st.results_tree(eng.finish(), labels, "cg", leaflets=True)
    bad = st.compare_trees(tree, expected("cg_order_begin_end_step.yaml"))
    assert not bad, bad[:10]


@pytest.fixture(scope="module")
def ua(built):
    return Fixture("ua")


def test_ua_order_basic(ua):
    # tests_ua.rs:19-68: pins the virtual-hydrogen construction (uaorder.rs:947-1104) end to end
    tables, labels, midx = ua_setup(ua)
    assert [m.name for m in labels] == ["POPC", "POPS"]
    frames = ua.window()
    eng = oracle.OracleEngine(tables, trig=oracle.TRIG_LIBM, n_threads=4)
    eng.submit(master_frames(ua, midx, frames), ua.boxes[frames], frames)
    res = eng.finish()
    tree = st.results_tree_ua(res, labels, leaflets=False)
    bad = st.compare_trees(tree, expected("ua_order_basic.yaml"))
    assert not bad, bad[:10]


def test_ua_fast_mode_of_the_oracle(ua):
    """GORDER_FLAG_UA_FAST_NORMALISE (include/gorder_hip.h): the oracle's non-libm modes restate the DEVICE's
    tolerance-bounded construction when the flag is in the tables (tests/test_ua_fast_gpu.py compares the two bit for
    bit); the libm mode — the reference's arithmetic — ignores it.  Here, without a GPU: the fast construction still
    reproduces the reference's golden file within its tolerance, every order parameter stays within one 1e-6 tick of the
    libm mode, and single samples move in a bounded fraction of the cases (profiles/r04_ua_fast_fidelity.json)."""
    from gorder_amd.abi import FLAG_UA_FAST_NORMALISE
    tables, labels, midx = ua_setup(ua, leaflets=METHODS["global"])
    frames = ua.window()
    xyz = master_frames(ua, midx, frames)

    def run(trig, flag):
        tables.flags = FLAG_UA_FAST_NORMALISE if flag else 0
        eng = oracle.OracleEngine(tables, trig=trig, n_threads=4)
        eng.submit(xyz, ua.boxes[frames], frames)
        return eng, eng.finish()
    _, libm = run(oracle.TRIG_LIBM, False)
    _, libm_flag = run(oracle.TRIG_LIBM, True)
    np.testing.assert_array_equal(libm.sums, libm_flag.sums)               # the reference-faithful mode ignores the flag
    _, plain = run(oracle.TRIG_DIRECT, False)
    eng, fast = run(oracle.TRIG_DIRECT, True)
    assert not np.array_equal(plain.sums, fast.sums)                        # ... the device-restating mode does not
    np.testing.assert_array_equal(fast.counts, libm.counts)
    assert np.abs(fast.order_ticks() - libm.order_ticks()).max() <= 1
    bad = st.compare_trees(st.results_tree_ua(fast, labels, leaflets=True), expected("ua_order_leaflets.yaml"))
    assert not bad, bad[:10]
    fid = eng.ua_fast_fidelity(xyz[:10], ua.boxes[frames][:10])
    assert fid["samples"] == 10 * int(libm.counts[0].sum()) // len(frames)
    assert fid["fraction_moved"] < 0.12 and fid["fraction_moved_by_more_than_one_tick"] < 0.03
    assert abs(fid["mean_shift_ticks"]) < 0.02 and fid["carbons_sent_to_the_literal_loops"] == 0
    assert fid["fraction_moved_default_path"] < fid["fraction_moved"]


@pytest.mark.parametrize("method,frequency", [("global", 1), ("local", 5), ("individual", 100), ("global", 0)])
def test_ua_order_leaflets(ua, method, frequency):
    # tests_ua.rs:147-200: every method and every frequency reproduces ua_order_leaflets.yaml
    tables, labels, midx = ua_setup(ua, leaflets=METHODS[method], frequency=frequency)
    frames = ua.window()
    eng = oracle.OracleEngine(tables, trig=oracle.TRIG_LIBM, n_threads=4)
    eng.submit(master_frames(ua, midx, frames), ua.boxes[frames], frames)
    tree = st.results_tree_ua(eng.finish(), labels, leaflets=True)
    bad = st.compare_trees(tree, expected("ua_order_leaflets.yaml"))
    assert not bad, bad[:10]


# ---- geometry selection (geometry.rs; tests_aa.rs:3022-3262) ------------------------------------
from gorder_amd.abi import (GEOM_CUBOID, GEOM_CYLINDER, GEOM_SPHERE, GEOMREF_BOX_CENTER, GEOMREF_GROUP,  # noqa: E402
                            GEOMREF_POINT, Geometry)

INF = float("inf")
GEOMETRY_CASES = {
    # name: (heavy-atom selection is the POPC C22/C24/C218 subset?, Geometry, expected file)
    "cuboid_square": (True, Geometry(kind=GEOM_CUBOID, reference=GEOMREF_POINT, point=(8.0, 2.0, 0.0),
                                     xdim=(-2.0, 4.0), ydim=(-4.0, 1.0)), "aa_order_cuboid_square.yaml"),
    "cylinder": (True, Geometry(kind=GEOM_CYLINDER, reference=GEOMREF_POINT, point=(8.0, 2.0, 0.0), radius=2.5,
                                orientation=2), "aa_order_cylinder.yaml"),
    "sphere_static": (False, Geometry(kind=GEOM_SPHERE, reference=GEOMREF_POINT, point=(8.0, 2.0, 4.5), radius=2.5),
                      "aa_order_sphere_static.yaml"),
    "cuboid_patch": (False, Geometry(kind=GEOM_CUBOID, reference=GEOMREF_BOX_CENTER, xdim=(-1.0, 3.0)),
                     "aa_order_cuboid_patch.yaml"),
    "cylinder_x": (False, Geometry(kind=GEOM_CYLINDER, reference=GEOMREF_BOX_CENTER, radius=3.0, span=(-1.0, 3.0),
                                   orientation=0), "aa_order_cylinder_x.yaml"),
    "sphere_center": (False, Geometry(kind=GEOM_SPHERE, reference=GEOMREF_BOX_CENTER, radius=2.5),
                      "aa_order_sphere_center.yaml"),
    # reference = centre of geometry of "resid 1", re-evaluated every frame (tests_aa.rs:3263-3346)
    "cuboid_dynamic": (False, Geometry(kind=GEOM_CUBOID, reference=GEOMREF_GROUP, xdim=(-1.0, 3.0), ydim=(1.0, 4.0),
                                       zdim=(-3.0, 3.0)), "aa_order_cuboid_dynamic.yaml"),
    "cylinder_dynamic": (False, Geometry(kind=GEOM_CYLINDER, reference=GEOMREF_GROUP, radius=2.1, orientation=1),
                         "aa_order_cylinder_dynamic.yaml"),
    "sphere_dynamic": (False, Geometry(kind=GEOM_SPHERE, reference=GEOMREF_GROUP, radius=2.5),
                       "aa_order_sphere_dynamic.yaml"),
}


def geometry_tables(fx, case):
    subset, geom, want = GEOMETRY_CASES[case]
    heavy = None
    if subset:   # "resname POPC and name C22 C24 C218" (tests_aa.rs:3062)
        heavy = np.array(fx.structure.resnames) == "POPC"
        heavy &= fx.name_in("C22", "C24", "C218")
    geom.structure_box = tuple(float(x) for x in fx.structure.box)
    if geom.reference == GEOMREF_GROUP:      # every atom of residue 1 must then be part of the decoded frames
        grp = np.array(fx.structure.resids) == 1
        master = fx.element("carbon") | fx.element("hydrogen") | grp
        remap = -np.ones(fx.structure.n_atoms, dtype=np.int64)
        remap[np.flatnonzero(master)] = np.arange(int(master.sum()))
        geom.group = remap[np.flatnonzero(grp)].astype(np.uint32)
        tables, labels, midx = aa_setup(fx, heavy=heavy, geometry=geom, master=master)
    else:
        tables, labels, midx = aa_setup(fx, heavy=heavy, geometry=geom)
    return tables, labels, midx, want


@pytest.mark.parametrize("case", sorted(GEOMETRY_CASES))
def test_aa_geometry_selection(pcpepg, case):
    tables, labels, midx, want = geometry_tables(pcpepg, case)
    frames = pcpepg.window()
    eng = oracle.OracleEngine(tables, trig=oracle.TRIG_LIBM, n_threads=4)
    eng.submit(master_frames(pcpepg, midx, frames), pcpepg.boxes[frames], frames)
    res = eng.finish()
    assert 0 < res.counts[0].sum() < 51 * tables.n_samples_per_frame      # the shape really filters
    tree = st.results_tree(res, labels, "aa", leaflets=False)
    bad = st.compare_trees(tree, expected(want))
    assert not bad, bad[:10]


def test_unbounded_sphere_is_no_selection(pcpepg):
    # tests_aa.rs:3000-3020: an infinite sphere reproduces aa_order_basic.yaml
    tables, labels, midx = aa_setup(pcpepg, geometry=Geometry(kind=GEOM_SPHERE, reference=GEOMREF_BOX_CENTER, radius=INF))
    frames = pcpepg.window()
    eng = oracle.OracleEngine(tables, trig=oracle.TRIG_LIBM, n_threads=4)
    eng.submit(master_frames(pcpepg, midx, frames), pcpepg.boxes[frames], frames)
    bad = st.compare_trees(st.results_tree(eng.finish(), labels, "aa", leaflets=False), expected("aa_order_basic.yaml"))
    assert not bad, bad[:10]


# ---- dynamic membrane normals (normal.rs:160-199, 421-458) -----------------------------------------
def dynamic_setup(kind, fx):
    """The three reference tests with `DynamicNormal::new(heads, 2.0)`: tests_aa.rs:4774-4807,
    tests_cg.rs:3358-3388, tests_ua.rs:717-743 -> (tables, labels, midx, leaflets?, expected file)."""
    if kind == "aa":
        t, l, m = aa_setup(fx, leaflets=METHODS["individual"], frequency=0,
                           dynamic_normal={"heads": fx.name_in("P"), "radius": 2.0})
        return t, l, m, True, "aa_order_leaflets_dynamic.yaml"
    if kind == "cg":
        t, l, m = cg_setup(fx, leaflets=METHODS["individual"], frequency=0,
                           dynamic_normal={"heads": fx.name_in("PO4"), "radius": 2.0})
        return t, l, m, True, "cg_order_leaflets_dynamic.yaml"
    heads = np.array([n.startswith("P") for n in fx.structure.names])              # name r'^P'
    t, l, m = ua_setup(fx, dynamic_normal={"heads": heads, "radius": 2.0})
    return t, l, m, False, "ua_order_dynamic_normals.yaml"


@pytest.mark.parametrize("kind", ["aa", "cg", "ua"])
def test_dynamic_normals(kind, pcpepg, cg, ua):
    fx = {"aa": pcpepg, "cg": cg, "ua": ua}[kind]
    tables, labels, midx, lf, want = dynamic_setup(kind, fx)
    assert tables.dynamic_normal.enabled and len(tables.dynamic_normal.cloud) == tables.n_molecules_total
    frames = fx.window()
    eng = oracle.OracleEngine(tables, trig=oracle.TRIG_LIBM, n_threads=4)
    eng.submit(master_frames(fx, midx, frames), fx.boxes[frames], frames)
    res = eng.finish()
    normals, npts = eng.normals()
    assert npts.min() >= 3 and np.allclose(np.linalg.norm(normals, axis=1), 1.0, atol=1e-6)
    assert np.abs(normals[:, 2]).mean() > 0.9          # a flat bilayer: local normals stay close to z
    tree = st.results_tree_ua(res, labels, leaflets=lf) if kind == "ua" else st.results_tree(res, labels, kind, leaflets=lf)
    bad = st.compare_trees(tree, expected(want))
    assert not bad, bad[:10]


def test_dynamic_normal_of_a_tilted_plane():
    """membrane_normal_from_cloud on points of a known plane: the normal is the plane's, whatever the sign."""
    rng = np.random.default_rng(3)
    n_true = np.array([0.3, -0.2, 0.933], dtype=np.float64); n_true /= np.linalg.norm(n_true)
    u = np.cross(n_true, [1.0, 0.0, 0.0]); u /= np.linalg.norm(u)
    v = np.cross(n_true, u)
    pts = (5.0 + rng.uniform(-1.2, 1.2, (40, 1)) * u + rng.uniform(-1.2, 1.2, (40, 1)) * v
           + rng.normal(0, 0.01, (40, 1)) * n_true).astype(np.float32)
    pts[0] = 5.0                                        # the reference head
    lib = oracle.load()
    out = np.zeros(4, dtype=np.float32)
    cloud = np.arange(40, dtype=np.uint32)
    box = np.array([10.0, 10.0, 10.0], dtype=np.float32)
    st_ = lib.gorder_oracle_dynamic_normal(pts.ctypes.data, cloud.ctypes.data, 40, 0, 2.0, box.ctypes.data, 1, out.ctypes.data)
    assert st_ == 0 and out[3] == 40
    assert abs(abs(float(np.dot(out[:3].astype(np.float64), n_true))) - 1.0) < 2e-4
    # fewer than three points in reach -> NaN normal, count reported (DynamicNormalError::NotEnoughPoints)
    st_ = lib.gorder_oracle_dynamic_normal(pts.ctypes.data, cloud.ctypes.data, 2, 0, 0.001, box.ctypes.data, 1, out.ctypes.data)
    assert st_ == 0 and out[3] == 1 and np.isnan(out[0])


# ---- exported leaflet assignment (tests_aa.rs:588-722) ------------------------------------------------
def expected_leaflets(labels, frame_row=0):
    """aa_leaflets_every1.yaml: per molecule type one row per assignment frame, 1 = upper, 0 = lower
    -> this repo's encoding (Upper = 0, Lower = 1, lib.rs:416-422) for all molecules in type order."""
    exp = expected("aa_leaflets_every1.yaml")
    return np.concatenate([1 - np.array(exp[m.name][frame_row], dtype=np.uint8) for m in labels])


@pytest.mark.parametrize("method", ["global", "local", "individual"])
def test_aa_leaflet_assignment_per_molecule(pcpepg, method):
    """global, local (2.5 nm) and individual classification all export the same per-molecule assignment
    for every frame; checked here on the first, a middle and the last frame."""
    tables, labels, midx = aa_setup(pcpepg, leaflets=METHODS[method])
    for f in (0, 25, 50):
        eng = oracle.OracleEngine(tables, trig=oracle.TRIG_DIRECT)
        eng.submit(master_frames(pcpepg, midx, [f]), pcpepg.boxes[[f]], [f])
        flags, _, frame = eng.leaflets()
        assert frame == f
        np.testing.assert_array_equal(flags, expected_leaflets(labels, f))


# ---- error estimation (timewise.rs:130-231; tests_aa.rs:2170-2280, tests_cg.rs:1367-1460) ------------
ERROR_CASES = [("aa", False, "aa_order_error.yaml"), ("aa", True, "aa_order_error_leaflets.yaml"),
               ("cg", False, "cg_order_error.yaml"), ("cg", True, "cg_order_error_leaflets.yaml")]


def error_setup(kind, lf, fx):
    setup = aa_setup if kind == "aa" else cg_setup
    return setup(fx, leaflets=METHODS["global"] if lf else None, timewise=True)


@pytest.mark.parametrize("kind,lf,want", ERROR_CASES)
def test_error_estimation(kind, lf, want, pcpepg, cg):
    """EstimateError::default(): 5 blocks over the per-frame partial sums.  Pins the timewise rows, the
    block arithmetic and how aggregates (atoms, molecules, system) inherit their members' rows."""
    fx = pcpepg if kind == "aa" else cg
    tables, labels, midx = error_setup(kind, lf, fx)
    frames = fx.window()
    eng = oracle.OracleEngine(tables, trig=oracle.TRIG_LIBM, n_threads=3)
    eng.submit(master_frames(fx, midx, frames), fx.boxes[frames], frames)
    res = eng.finish()
    tw = eng.timewise(len(frames))
    tree = st.results_tree(res, labels, kind, leaflets=lf, timewise=tw)
    bad = st.compare_trees(tree, expected(want))
    assert not bad, bad[:10]


# ---- ordermaps (ordermap.rs:40-113; tests_ua.rs:351-410) ------------------------------------------------
# (The all-atom and coarse-grained ordermap goldens, tests/files/ordermaps{,_cg}/, were made from pcpepg.xtc / cg.xtc,
# which the reference checkout does not contain; its split/*.xtc twins carry XTC precision 100, so bond midpoints
# are multiples of 0.005 nm: with 0.1-nm tiles 5 % of the samples sit exactly on a tile edge, with 1-nm tiles 1 %
# change tile, and the maps are not reproduced tile for tile whatever the tie rule.  The united-atom maps come from
# tests/files/ua.xtc (precision 1000), which IS there.)
from gorder_amd.abi import OrderMap   # noqa: E402
from golden_util import compare_maps, map_of, read_map   # noqa: E402
from gorder_amd.select import select   # noqa: E402


def ordermap_setup(fx, leaflets=False):
    """`resname POPC and name C50 C20 C13` saturated, `resname POPC and name C24` unsaturated, bin [0.5, 2.0],
    span Auto = (0, box of the structure), xy plane, min_samples 5; with `leaflets` the classifier of
    test_ua_order_maps_leaflets (tests_ua.rs:443): Global, `@membrane`, heads `name r'^P'`."""
    s = fx.structure
    popc = np.array(s.resnames) == "POPC"
    sat = popc & fx.name_in("C50", "C20", "C13")
    unsat = popc & fx.name_in("C24")
    bx = fx.boxes[0]            # ua.tpr's box = the box of the trajectory's first frame (6.28779 nm)
    om = OrderMap(enabled=True, plane=0, span_x=(0.0, float(bx[0, 0])), span_y=(0.0, float(bx[1, 1])), bin=(0.5, 2.0))
    lf = None
    if leaflets:
        lf = {"method": METHODS["global"], "membrane": select(s, "@membrane"), "heads": select(s, "name r'^P'"),
              "methyls": np.zeros(s.n_atoms, dtype=bool), "frequency": 1, "radius": 2.5}
    tables, labels, midx = st.build_tables_ua(s, sat, unsat, np.ones(s.n_atoms, dtype=bool), ordermap=om, leaflets=lf)
    return tables, labels, midx, om


def check_ordermaps(res, labels, om, leaflets=False):
    """Every map the reference's tests compare: per virtual bond, per carbon (its bonds aggregated) and the average
    over all bonds — the `_full` plane (tests_ua.rs:351-410) and, with leaflets, the `_upper` and `_lower` planes as
    well (Map::add_order routes a sample to its molecule's leaflet, ordermap.rs:100-113, bond.rs:199-213;
    tests_ua.rs:418-507), tile for tile with assert_eq_maps' rule (NaN where NaN, else within 2e-4)."""
    (ml,) = labels
    assert ml.name == "POPC"
    n_maps = 0
    for w, plane in enumerate(("full", "upper", "lower") if leaflets else ("full",)):
        slot, allslots = ml.slot0, []
        for c in ml.carbons:
            slots = list(range(slot, slot + c.n_h))
            slot += c.n_h
            allslots += slots
            bad = compare_maps(map_of(res, slots, w, om, 5), read_map(f"ordermap_POPC-{c.name}-{c.rel}_{plane}.dat"))
            assert not bad, (c.name, plane, bad[:5])
            for k, sl in enumerate(slots):
                bad = compare_maps(map_of(res, [sl], w, om, 5),
                                   read_map(f"ordermap_POPC-{c.name}-{c.rel}--POPC-H{k + 1}-{c.rel}_{plane}.dat"))
                assert not bad, (c.name, k, plane, bad[:5])
                n_maps += 1
        bad = compare_maps(map_of(res, allslots, w, om, 5), read_map(f"ordermap_average_{plane}.dat"))
        assert not bad, (plane, bad[:5])
    return n_maps


def test_ua_ordermaps(ua):
    tables, labels, midx, om = ordermap_setup(ua)
    frames = ua.window()
    eng = oracle.OracleEngine(tables, trig=oracle.TRIG_LIBM, n_threads=3)
    eng.submit(master_frames(ua, midx, frames), ua.boxes[frames], frames)
    res = eng.finish()
    assert res.map_sums.shape[2:] == (14, 4)          # GridMap: round(6.288 / 0.5) + 1, round(6.288 / 2) + 1
    assert check_ordermaps(res, labels, om) == 7


@pytest.mark.parametrize("trig", ["libm", "direct"])
def test_ua_ordermaps_with_leaflets(ua, trig):
    # tests_ua.rs:418-507: 36 maps, the upper / lower planes included
    tables, labels, midx, om = ordermap_setup(ua, leaflets=True)
    frames = ua.window()
    eng = oracle.OracleEngine(tables, trig=oracle.TRIG_LIBM if trig == "libm" else oracle.TRIG_DIRECT, n_threads=3)
    eng.submit(master_frames(ua, midx, frames), ua.boxes[frames], frames)
    res = eng.finish()
    assert res.map_sums.shape == (3, res.sums.shape[1], 14, 4)
    assert check_ordermaps(res, labels, om, leaflets=True) == 21
    # a sample lands in exactly one leaflet plane: total = upper + lower, tile by tile (bond.rs:199-213)
    np.testing.assert_array_equal(res.map_counts[0], res.map_counts[1] + res.map_counts[2])
    np.testing.assert_array_equal(res.map_sums[0], res.map_sums[1] + res.map_sums[2])


# ---- single-frame tests of the reference (aaorder.rs:226-464, cgorder.rs:188-351) ------------------------------
def single_frame(kind, fx):
    """The structure file's own coordinates — the .tpr's full-precision f32 positions and box, found in the file by
    tests/golden/make_fixtures.py:tpr_frame — as one frame with global leaflets, and the literal expectation arrays
    of the reference's unit tests (which run on exactly these coordinates)."""
    import json
    import os
    from golden_util import GOLDEN
    z = np.load(os.path.join(GOLDEN, ("pcpepg" if kind == "aa" else "cg") + "_structure_frame.npz"))
    xyz = z["xyz_tpr"].astype(np.float32)
    assert np.abs(xyz - z["ints"].astype(np.float64) * 0.001).max() < 5.01e-4      # the .gro twin prints the same frame
    box = np.zeros((1, 3, 3), dtype=np.float32)
    box[0, 0, 0], box[0, 1, 1], box[0, 2, 2] = z["box_tpr"]
    with open(os.path.join(GOLDEN, "expected", "single_frame_sums.json")) as f:
        want = json.load(f)[kind]
    setup = aa_setup if kind == "aa" else cg_setup
    tables, labels, midx = setup(fx, leaflets=METHODS["global"])
    return tables, labels, np.ascontiguousarray(xyz[midx][None]), box, want


def check_single_frame(kind, res, labels, want, tol=None):
    tol = tol or {"aa": 1.5e-4, "cg": 6e-5}[kind]
    counts = {"aa": ([131, 128, 15], [65, 64, 8], [66, 64, 7]), "cg": ([242, 242, 24], [121, 121, 12], [121, 121, 12])}[kind]
    sign = -1.0 if kind == "aa" else 1.0
    for w, key in enumerate(("total", "upper", "lower")):
        for m, ml in enumerate(labels):
            sl = slice(ml.slot0, ml.slot0 + len(ml.bonds))
            # the leaflet populations are exact: this pins the global classifier (and its Bai-Breen centre)
            assert set(res.counts[w, sl].tolist()) == {counts[w][m]}
            # The reference's own bar is assert_relative_eq!(-real, expected, epsilon = 1e-5) on the f32 sum of up to 242
            # samples (aaorder.rs:394, 456; cgorder.rs).  On the .tpr's f32 coordinates the sums (values 1..130) come
            # out within 1.1e-4 (AA) / 4.6e-5 (CG = 6 f32 ulps of the sum) — with the .gro twins it was 0.2.  An exact
            # f64 evaluation of the same coordinates leaves the same residual, so it is not this arithmetic: it is what
            # ONE f32 ulp of noise per coordinate does to a sum (3e-7 nm on a 0.11 nm bond = 4 ticks per sample), i.e.
            # the reference's expectation arrays were made from coordinates that went through one more f32 rounding
            # somewhere in its .tpr reading path (minitpr / groan_rs, not in the checkout).
            got = (sign * res.sums[w, sl] / 1e6).astype(np.float32)
            err = np.abs(got.astype(np.float64) - np.array(want[key][m], dtype=np.float32).astype(np.float64))
            assert err.max() < tol, (key, m, err.max())


@pytest.mark.parametrize("kind", ["aa", "cg"])
def test_single_frame_leaflet_populations_and_sums(kind, pcpepg, cg):
    fx = pcpepg if kind == "aa" else cg
    tables, labels, xyz, box, want = single_frame(kind, fx)
    eng = oracle.OracleEngine(tables, trig=oracle.TRIG_LIBM)
    eng.submit(xyz, box, [0])
    check_single_frame(kind, eng.finish(), labels, want)


# ---- more of the reference's goldens on the same data: min_samples, windows, inverted shapes, 10-block errors ----
def _geom(kind, **kw):
    g = Geometry(kind=kind, **kw)
    return g


MORE_CASES = {
    # name: (setup kwargs, window (begin, end, step), min_samples, timewise blocks or None, leaflets in the output?)
    "aa_order_limit.yaml": (dict(), (None, None, 1), 2000, None, False),                       # tests_aa.rs:1099-1120
    "aa_order_leaflets_limit.yaml": (dict(leaflets="global"), (None, None, 1), 500, None, True),    # :1123-1146
    "aa_order_step.yaml": (dict(leaflets="global", frequency=5), (None, None, 5), 1, None, True),   # :1300-1345 (real frequency = 1 x step)
    "aa_order_begin_end.yaml": (dict(leaflets="global"), (450_200.0, 450_400.0, 1), 1, None, True),  # :1398-1423
    # (aa_order_cuboid_square_inverted.yaml, tests_aa.rs:3505-3545, is NOT here: the cuboid's face sits at x = 6.0 and
    #  the split/pcpepg*.xtc files are written with XTC precision 100 — bond midpoints are multiples of 5e-3 nm and a
    #  few dozen lie exactly ON the face.  The golden came from the full-precision pcpepg.xtc (not in the checkout):
    #  690 of its 698 values are reproduced, 8 are off by 2-7e-4, one on-face sample each.)
    "aa_order_cylinder_z_inverted.yaml": (dict(geometry=("cylinder_inv",)), (None, None, 1), 1, None, False),    # :3548-3585
    "aa_order_sphere_dynamic_inverted.yaml": (dict(geometry=("sphere_dyn_inv",)), (None, None, 1), 1, None, False),  # :3588-3616
    "aa_order_error_blocks10.yaml": (dict(timewise=True), (None, None, 1), 1, 10, False),       # :2530-2552
    "aa_order_error_limit.yaml": (dict(timewise=True), (None, None, 1), 2000, 5, False),        # :2445-2480
}


def more_setup(fx, name):
    kw, window, min_samples, blocks, lf = MORE_CASES[name]
    kw = dict(kw)
    if "leaflets" in kw:
        kw["leaflets"] = METHODS[kw["leaflets"]]
    master = None
    if "geometry" in kw:
        tag = kw.pop("geometry")[0]
        sbox = tuple(float(x) for x in fx.structure.box)
        if tag == "cuboid_inv":
            kw["geometry"] = Geometry(kind=GEOM_CUBOID, reference=GEOMREF_POINT, point=(8.0, 2.0, 0.0), xdim=(-2.0, 4.0),
                                      ydim=(-4.0, 1.0), invert=True, structure_box=sbox)
        elif tag == "cylinder_inv":
            kw["geometry"] = Geometry(kind=GEOM_CYLINDER, reference=GEOMREF_BOX_CENTER, radius=3.0, orientation=2,
                                      invert=True, structure_box=sbox)
        else:
            grp = np.array(fx.structure.resids) == 1
            master = fx.element("carbon") | fx.element("hydrogen") | grp
            remap = -np.ones(fx.structure.n_atoms, dtype=np.int64)
            remap[np.flatnonzero(master)] = np.arange(int(master.sum()))
            kw["geometry"] = Geometry(kind=GEOM_SPHERE, reference=GEOMREF_GROUP, radius=2.5, invert=True,
                                      group=remap[np.flatnonzero(grp)].astype(np.uint32), structure_box=sbox)
            kw["master"] = master
    tables, labels, midx = aa_setup(fx, **kw)
    frames = fx.window(*window)
    fidx = np.arange(len(frames)) * window[2]          # the reference counts frames of the stepped sequence x step
    return tables, labels, midx, frames, fidx, min_samples, blocks, lf


@pytest.mark.parametrize("name", sorted(MORE_CASES))
def test_more_reference_goldens(pcpepg, name):
    tables, labels, midx, frames, fidx, min_samples, blocks, lf = more_setup(pcpepg, name)
    eng = oracle.OracleEngine(tables, trig=oracle.TRIG_LIBM, n_threads=3)
    eng.submit(master_frames(pcpepg, midx, frames), pcpepg.boxes[frames], fidx)
    res = eng.finish()
    tw = eng.timewise(len(frames)) if blocks else None
    tree = st.results_tree(res, labels, "aa", leaflets=lf, min_samples=min_samples, timewise=tw, n_blocks=blocks or 5)
    bad = st.compare_trees(tree, expected(name))
    assert not bad, bad[:10]


# ---- the coarse-grained counterparts (tests_cg.rs:620-650, 893-916, 2468-2530, 2682-2715) ---------------------
CG_MORE = {
    "cg_order_limit.yaml": (dict(), (None, None, 1), 5000, False),
    "cg_order_begin_end.yaml": (dict(leaflets="global"), (352_000.0, 358_000.0, 1), 1, True),
    "cg_order_cuboid_square.yaml": (dict(geometry=Geometry(kind=GEOM_CUBOID, reference=GEOMREF_BOX_CENTER,
                                                           xdim=(-8.0, -2.0), ydim=(2.0, 8.0))), (None, None, 1), 1, False),
    "cg_order_cylinder.yaml": (dict(geometry=Geometry(kind=GEOM_CYLINDER, reference=GEOMREF_POINT, point=(2.0, 1.0, 0.0),
                                                      radius=3.25, orientation=2)), (None, None, 1), 1, False),
    "cg_order_cylinder_z_inverted.yaml": (dict(geometry=Geometry(kind=GEOM_CYLINDER, reference=GEOMREF_POINT,
                                                                 point=(3.0, 3.0, 3.0), radius=4.0, orientation=2,
                                                                 invert=True)), (None, None, 1), 1, False),
}


def cg_more_setup(fx, name):
    kw, window, min_samples, lf = CG_MORE[name]
    kw = dict(kw)
    if "leaflets" in kw:
        kw["leaflets"] = METHODS[kw["leaflets"]]
    if "geometry" in kw:
        kw["geometry"].structure_box = tuple(float(x) for x in fx.structure.box)
    tables, labels, midx = cg_setup(fx, **kw)
    frames = fx.window(*window)
    return tables, labels, midx, frames, min_samples, lf


@pytest.mark.parametrize("name", sorted(CG_MORE))
def test_more_cg_goldens(cg, name):
    tables, labels, midx, frames, min_samples, lf = cg_more_setup(cg, name)
    eng = oracle.OracleEngine(tables, trig=oracle.TRIG_LIBM, n_threads=3)
    eng.submit(master_frames(cg, midx, frames), cg.boxes[frames], np.arange(len(frames)))
    tree = st.results_tree(eng.finish(), labels, "cg", leaflets=lf, min_samples=min_samples)
    bad = st.compare_trees(tree, expected(name))
    assert not bad, bad[:10]


# ---- NoPBC (pbc.rs:98-253): molecules made whole, no box, handle_pbc(false) — tests_ua.rs:686-714 -------------
@pytest.fixture(scope="module")
def ua_nobox(built):
    return Fixture("ua_nobox")


def test_ua_order_leaflets_without_pbc(ua_nobox):
    tables, labels, midx = ua_setup(ua_nobox, leaflets=METHODS["global"], handle_pbc=False)
    frames = ua_nobox.window()
    eng = oracle.OracleEngine(tables, trig=oracle.TRIG_LIBM, n_threads=4)
    eng.submit(master_frames(ua_nobox, midx, frames), None, frames)
    tree = st.results_tree_ua(eng.finish(), labels, leaflets=True)
    bad = st.compare_trees(tree, expected("ua_order_leaflets_nopbc.yaml"))
    assert not bad, bad[:10]
