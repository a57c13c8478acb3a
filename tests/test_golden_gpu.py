"""The HIP path against the reference's own golden outputs (same fixtures as test_golden_oracle.py)
and, frame for frame, against the oracle on the real membrane data."""
import numpy as np
import pytest

from gorder_amd import HipEngine, abi
from gorder_amd import structure as st
from oracle import oracle
from golden_util import METHODS, Fixture, aa_setup, cg_setup, expected, ua_setup
from leaflet_check import assert_sums_given_device_flags

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pcpepg(built):
    return Fixture("pcpepg")


@pytest.fixture(scope="module")
def cg(built):
    return Fixture("cg")


def gpu_run(tables, fx, midx, frames, frame_index=None, batches=3):
    eng = HipEngine(tables)
    xyz = np.ascontiguousarray(fx.xyz[frames][:, midx, :])
    box = fx.boxes[frames]
    fi = np.asarray(frames if frame_index is None else frame_index)
    edges = np.linspace(0, len(frames), batches + 1).astype(int)
    for a, b in zip(edges[:-1], edges[1:]):
        if b > a:
            eng.submit_host(xyz[a:b], box[a:b], fi[a:b])
    return eng, eng.finish(), xyz, box, fi


def oracle_run(tables, xyz, box, fi, trig):
    o = oracle.OracleEngine(tables, trig=trig, n_threads=4)
    o.submit(xyz, box, fi)
    return o, o.finish()


@pytest.mark.parametrize("flags", [0, abi.FLAG_TRIG_ACOS_COS])
def test_aa_order_basic(pcpepg, flags):
    tables, labels, midx = aa_setup(pcpepg, flags=flags)
    frames = pcpepg.window()
    eng, res, xyz, box, fi = gpu_run(tables, pcpepg, midx, frames)
    assert res.n_frames == 51
    bad = st.compare_trees(st.results_tree(res, labels, "aa", leaflets=False), expected("aa_order_basic.yaml"))
    assert not bad, bad[:10]
    # and bit-exact against the oracle mode that restates the device arithmetic
    _, want = oracle_run(tables, xyz, box, fi, oracle.TRIG_MIRROR if flags else oracle.TRIG_DIRECT)
    np.testing.assert_array_equal(res.sums, want.sums)
    np.testing.assert_array_equal(res.counts, want.counts)
    # within 1e-6 of the reference-faithful (libm) oracle
    _, libm = oracle_run(tables, xyz, box, fi, oracle.TRIG_LIBM)
    assert np.abs(res.order_ticks() - libm.order_ticks()).max() <= 1
    if flags:
        # ... and with the literal flag the SAME integers: the device evaluates acos and cos by glibc's algorithms, so on
        # the reference's own membrane its sums are those of the reference's arithmetic (glibc 2.28 - 2.40 hosts)
        import platform
        if platform.libc_ver()[0] == "glibc" and "2.28" <= platform.libc_ver()[1] <= "2.40":
            np.testing.assert_array_equal(res.sums, libm.sums)


@pytest.mark.parametrize("method", ["global", "local", "individual"])
def test_aa_order_leaflets(pcpepg, method):
    tables, labels, midx = aa_setup(pcpepg, leaflets=METHODS[method])
    frames = pcpepg.window()
    eng, res, xyz, box, fi = gpu_run(tables, pcpepg, midx, frames)
    bad = st.compare_trees(st.results_tree(res, labels, "aa", leaflets=True), expected("aa_order_leaflets.yaml"))
    assert not bad, bad[:10]
    o, want = oracle_run(tables, xyz, box, fi, oracle.TRIG_DIRECT)
    flags, _ = eng.leaflets()
    oflags, odist, _ = o.leaflets()
    diff = flags != oflags
    assert not diff.any() or np.abs(odist[diff]).max() < 1e-4   # only a lipid ON the mid-plane may differ
    # the integer sums are compared in every case: equal to the oracle run with the device's own assignment
    assert_sums_given_device_flags(tables, xyz, box, res, fi)
    if not diff.any():
        np.testing.assert_array_equal(res.sums, want.sums)
        np.testing.assert_array_equal(res.counts, want.counts)


def test_aa_begin_end_step(pcpepg):
    tables, labels, midx = aa_setup(pcpepg, leaflets=METHODS["global"], frequency=3)
    frames = pcpepg.window(450_200.0, 450_400.0, 3)
    eng, res, *_ = gpu_run(tables, pcpepg, midx, frames, frame_index=np.arange(4) * 3, batches=2)
    bad = st.compare_trees(st.results_tree(res, labels, "aa", leaflets=True), expected("aa_order_begin_end_step.yaml"))
    assert not bad, bad[:10]


def test_cg_order_basic(cg):
    tables, labels, midx = cg_setup(cg)
    frames = cg.window()
    eng, res, xyz, box, fi = gpu_run(tables, cg, midx, frames)
    bad = st.compare_trees(st.results_tree(res, labels, "cg", leaflets=False), expected("cg_order_basic.yaml"))
    assert not bad, bad[:10]
    _, want = oracle_run(tables, xyz, box, fi, oracle.TRIG_DIRECT)
    np.testing.assert_array_equal(res.sums, want.sums)


@pytest.mark.parametrize("method", ["global", "local", "individual"])
def test_cg_order_leaflets(cg, method):
    tables, labels, midx = cg_setup(cg, leaflets=METHODS[method])
    frames = cg.window()
    eng, res, xyz, box, fi = gpu_run(tables, cg, midx, frames)
    bad = st.compare_trees(st.results_tree(res, labels, "cg", leaflets=True), expected("cg_order_leaflets.yaml"))
    assert not bad, bad[:10]


def test_cg_begin_end_step(cg):
    tables, labels, midx = cg_setup(cg, leaflets=METHODS["global"], frequency=5)
    frames = cg.window(352_000.0, 358_000.0, 5)
    eng, res, *_ = gpu_run(tables, cg, midx, frames, frame_index=np.arange(13) * 5)
    bad = st.compare_trees(st.results_tree(res, labels, "cg", leaflets=True), expected("cg_order_begin_end_step.yaml"))
    assert not bad, bad[:10]


@pytest.fixture(scope="module")
def ua(built):
    return Fixture("ua")


def test_ua_order_basic(ua):
    tables, labels, midx = ua_setup(ua)
    frames = ua.window()
    eng, res, xyz, box, fi = gpu_run(tables, ua, midx, frames)
    bad = st.compare_trees(st.results_tree_ua(res, labels, leaflets=False), expected("ua_order_basic.yaml"))
    assert not bad, bad[:10]
    _, libm = oracle_run(tables, xyz, box, fi, oracle.TRIG_LIBM)
    np.testing.assert_array_equal(res.counts, libm.counts)
    assert np.abs(res.order_ticks() - libm.order_ticks()).max() <= 1


@pytest.mark.parametrize("method,frequency", [("global", 1), ("local", 5), ("individual", 0)])
def test_ua_order_leaflets(ua, method, frequency):
    tables, labels, midx = ua_setup(ua, leaflets=METHODS[method], frequency=frequency)
    frames = ua.window()
    eng, res, *_ = gpu_run(tables, ua, midx, frames)
    bad = st.compare_trees(st.results_tree_ua(res, labels, leaflets=True), expected("ua_order_leaflets.yaml"))
    assert not bad, bad[:10]


# ---- geometry selection --------------------------------------------------------------------------
from test_golden_oracle import GEOMETRY_CASES, geometry_tables   # noqa: E402


@pytest.mark.parametrize("case", sorted(GEOMETRY_CASES))
def test_aa_geometry_selection(pcpepg, case):
    tables, labels, midx, want = geometry_tables(pcpepg, case)
    frames = pcpepg.window()
    eng, res, xyz, box, fi = gpu_run(tables, pcpepg, midx, frames)
    bad = st.compare_trees(st.results_tree(res, labels, "aa", leaflets=False), expected(want))
    assert not bad, bad[:10]
    _, ref = oracle_run(tables, xyz, box, fi, oracle.TRIG_DIRECT)
    np.testing.assert_array_equal(res.counts, ref.counts)     # the same samples pass the filter
    np.testing.assert_array_equal(res.sums, ref.sums)


# ---- dynamic membrane normals ------------------------------------------------------------------------
from test_golden_oracle import dynamic_setup   # noqa: E402


@pytest.mark.parametrize("kind", ["aa", "cg", "ua"])
def test_dynamic_normals(kind, pcpepg, cg, ua):
    """tests_aa.rs:4774-4807, tests_cg.rs:3358-3388, tests_ua.rs:717-743.  The normals come from an f64
    eigen-decomposition on both sides (nalgebra's f32 SVD is not restatable), device and oracle only
    differ in the order the cloud is summed: normals agree to ~1e-7, order parameters to <= 1 tick."""
    fx = {"aa": pcpepg, "cg": cg, "ua": ua}[kind]
    tables, labels, midx, lf, want = dynamic_setup(kind, fx)
    frames = fx.window()
    eng, res, xyz, box, fi = gpu_run(tables, fx, midx, frames)
    tree = st.results_tree_ua(res, labels, leaflets=lf) if kind == "ua" else st.results_tree(res, labels, kind, leaflets=lf)
    bad = st.compare_trees(tree, expected(want))
    assert not bad, bad[:10]
    o, ref = oracle_run(tables, xyz, box, fi, oracle.TRIG_DIRECT)
    np.testing.assert_array_equal(res.counts, ref.counts)
    assert np.abs(res.order_ticks() - ref.order_ticks()).max() <= 1
    n_gpu, k_gpu = eng.normals()
    n_ref, k_ref = o.normals()
    np.testing.assert_array_equal(k_gpu, k_ref)                    # the same cloud for every molecule
    assert np.abs(n_gpu - n_ref).max() < 1e-6
    _, libm = oracle_run(tables, xyz, box, fi, oracle.TRIG_LIBM)
    assert np.abs(res.order_ticks() - libm.order_ticks()).max() <= 1


# ---- exported leaflet assignment -----------------------------------------------------------------------
from test_golden_oracle import expected_leaflets   # noqa: E402


@pytest.mark.parametrize("method", ["global", "local", "individual"])
def test_aa_leaflet_assignment_per_molecule(pcpepg, method):
    # tests_aa.rs:682-722: every method exports the assignment of aa_leaflets_every1.yaml, frame by frame
    tables, labels, midx = aa_setup(pcpepg, leaflets=METHODS[method])
    eng = HipEngine(tables)
    for f in range(0, 51, 5):
        eng.submit_host(np.ascontiguousarray(pcpepg.xyz[[f]][:, midx, :]), pcpepg.boxes[[f]], [f])
        flags, frame = eng.leaflets()
        assert frame == f
        np.testing.assert_array_equal(flags, expected_leaflets(labels, f))


# ---- error estimation -----------------------------------------------------------------------------------
from test_golden_oracle import ERROR_CASES, error_setup   # noqa: E402


@pytest.mark.parametrize("kind,lf,want", ERROR_CASES)
def test_error_estimation(kind, lf, want, pcpepg, cg):
    fx = pcpepg if kind == "aa" else cg
    tables, labels, midx = error_setup(kind, lf, fx)
    frames = fx.window()
    eng, res, xyz, box, fi = gpu_run(tables, fx, midx, frames)
    tw = eng.timewise(len(frames))
    bad = st.compare_trees(st.results_tree(res, labels, kind, leaflets=lf, timewise=tw), expected(want))
    assert not bad, bad[:10]
    o, ref = oracle_run(tables, xyz, box, fi, oracle.TRIG_DIRECT)
    ws, wc = o.timewise(len(frames))
    np.testing.assert_array_equal(tw[0], ws)        # per-frame rows bit-exact
    np.testing.assert_array_equal(tw[1], wc)


# ---- ordermaps ---------------------------------------------------------------------------------------------
from test_golden_oracle import check_ordermaps, ordermap_setup   # noqa: E402


def test_ua_ordermaps(ua):
    # tests_ua.rs:351-410: the 12 ordermaps of the reference's united-atom test, tile for tile
    tables, labels, midx, om = ordermap_setup(ua)
    frames = ua.window()
    eng, res, xyz, box, fi = gpu_run(tables, ua, midx, frames)
    assert check_ordermaps(res, labels, om) == 7
    _, ref = oracle_run(tables, xyz, box, fi, oracle.TRIG_DIRECT)
    np.testing.assert_array_equal(res.map_counts, ref.map_counts)


@pytest.mark.parametrize("route", ["staged", "direct"])
def test_ua_ordermaps_with_leaflets(ua, route, monkeypatch):
    # tests_ua.rs:418-507 (test_ua_order_maps_leaflets): 36 maps — the `_upper` / `_lower` planes of Map::add_order
    # (ordermap.rs:100-113, bond.rs:199-213) against the reference's own files, through k_map_accumulate's
    # leaflet branch (staged) and through one atomic per sample (GORDER_HIP_MAP_DIRECT=1)
    if route == "direct":
        monkeypatch.setenv("GORDER_HIP_MAP_DIRECT", "1")
    tables, labels, midx, om = ordermap_setup(ua, leaflets=True)
    frames = ua.window()
    eng, res, xyz, box, fi = gpu_run(tables, ua, midx, frames)
    assert eng.plan()["map_staged"] == int(route == "staged")
    assert check_ordermaps(res, labels, om, leaflets=True) == 21
    np.testing.assert_array_equal(res.map_counts[0], res.map_counts[1] + res.map_counts[2])
    np.testing.assert_array_equal(res.map_sums[0], res.map_sums[1] + res.map_sums[2])
    o, ref = oracle_run(tables, xyz, box, fi, oracle.TRIG_DIRECT)
    flags, _ = eng.leaflets()
    oflags, odist, _ = o.leaflets()
    if np.array_equal(flags, oflags):
        np.testing.assert_array_equal(res.map_counts, ref.map_counts)
        np.testing.assert_array_equal(res.map_sums, ref.map_sums)
        np.testing.assert_array_equal(res.sums, ref.sums)
    else:
        np.testing.assert_array_equal(res.map_counts[0], ref.map_counts[0])
        np.testing.assert_array_equal(res.map_sums[0], ref.map_sums[0])
        assert_sums_given_device_flags(tables, xyz, box, res, fi)


# ---- single-frame tests of the reference ------------------------------------------------------------------
from test_golden_oracle import check_single_frame, single_frame   # noqa: E402


@pytest.mark.parametrize("kind", ["aa", "cg"])
def test_single_frame_leaflet_populations_and_sums(kind, pcpepg, cg):
    # aaorder.rs:226-464, cgorder.rs:188-351: leaflet populations 65-64-8 / 66-64-7 and 121-121-12, sums per bond type
    fx = pcpepg if kind == "aa" else cg
    tables, labels, xyz, box, want = single_frame(kind, fx)
    eng = HipEngine(tables)
    eng.submit_host(xyz, box, [0])
    check_single_frame(kind, eng.finish(), labels, want)


# ---- more of the reference's goldens: min_samples, windows, inverted shapes, block counts -------------------------
from test_golden_oracle import MORE_CASES, more_setup   # noqa: E402


@pytest.mark.parametrize("name", sorted(MORE_CASES))
def test_more_reference_goldens(pcpepg, name):
    tables, labels, midx, frames, fidx, min_samples, blocks, lf = more_setup(pcpepg, name)
    eng, res, xyz, box, fi = gpu_run(tables, pcpepg, midx, frames, frame_index=fidx)
    tw = eng.timewise(len(frames)) if blocks else None
    tree = st.results_tree(res, labels, "aa", leaflets=lf, min_samples=min_samples, timewise=tw, n_blocks=blocks or 5)
    bad = st.compare_trees(tree, expected(name))
    assert not bad, bad[:10]


from test_golden_oracle import CG_MORE, cg_more_setup   # noqa: E402


@pytest.mark.parametrize("name", sorted(CG_MORE))
def test_more_cg_goldens(cg, name):
    tables, labels, midx, frames, min_samples, lf = cg_more_setup(cg, name)
    eng, res, xyz, box, fi = gpu_run(tables, cg, midx, frames, frame_index=np.arange(len(frames)))
    bad = st.compare_trees(st.results_tree(res, labels, "cg", leaflets=lf, min_samples=min_samples), expected(name))
    assert not bad, bad[:10]
    _, ref = oracle_run(tables, xyz, box, fi, oracle.TRIG_DIRECT)
    np.testing.assert_array_equal(res.counts, ref.counts)


@pytest.fixture(scope="module")
def ua_nobox(built):
    return Fixture("ua_nobox")


def test_ua_order_leaflets_without_pbc(ua_nobox):
    # tests_ua.rs:686-714: handle_pbc(false) on whole molecules without a box — the NoPBC arms of every operation
    tables, labels, midx = ua_setup(ua_nobox, leaflets=METHODS["global"], handle_pbc=False)
    frames = ua_nobox.window()
    eng = HipEngine(tables)
    xyz = np.ascontiguousarray(ua_nobox.xyz[frames][:, midx, :])
    for a, b in ((0, 20), (20, len(frames))):
        eng.submit_host(xyz[a:b], None, frames[a:b])
    res = eng.finish()
    bad = st.compare_trees(st.results_tree_ua(res, labels, leaflets=True), expected("ua_order_leaflets_nopbc.yaml"))
    assert not bad, bad[:10]
    o = oracle.OracleEngine(tables, trig=oracle.TRIG_DIRECT, n_threads=4)
    o.submit(xyz, None, frames)
    ref = o.finish()
    np.testing.assert_array_equal(res.counts, ref.counts)
    assert np.abs(res.order_ticks() - ref.order_ticks()).max() <= 1
