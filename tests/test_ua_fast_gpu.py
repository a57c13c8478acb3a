"""GORDER_FLAG_UA_FAST_NORMALISE (include/gorder_hip.h): the united-atom hydrogen construction (uaorder.rs:947-1104) with
tolerance-bounded arithmetic.  The oracle's non-libm modes restate the device's fast arithmetic operation for operation
when the flag is in the tables, so sums, counts, per-frame rows and ordermaps must be EQUAL to that mode; against the
reference-faithful libm mode every order parameter stays within one 1e-6 tick (tolerance of north_star: 1e-6); the
reference's own goldens are reproduced within its own tolerance; and the default path is not touched."""
import numpy as np
import pytest

from gorder_amd import HipEngine, abi, synthetic
from gorder_amd import structure as st
from gorder_amd.abi import (FLAG_TRIG_ACOS_COS, FLAG_UA_FAST_NORMALISE, GEOM_CYLINDER, GEOMREF_BOX_CENTER, LEAFLETS_GLOBAL,
                            LEAFLETS_NONE, Geometry, OrderMap)
from oracle import oracle
from golden_util import METHODS, Fixture, expected, ua_setup
from test_extras_gpu import both

pytestmark = pytest.mark.gpu


def fast(system):
    system.tables.flags |= FLAG_UA_FAST_NORMALISE
    return system


@pytest.mark.parametrize("pbc", [True, False])
@pytest.mark.parametrize("leaflets", [LEAFLETS_NONE, LEAFLETS_GLOBAL])
def test_orders_equal_the_oracles_fast_mode(built, leaflets, pbc):
    system = fast(synthetic.ua_membrane(48, leaflets=leaflets, handle_pbc=pbc))
    n = 13
    xyz = system.frames(n, seed=8)
    box = system.box9(n) if pbc else None
    eng, o, got, want = both(system, xyz, box)
    np.testing.assert_array_equal(got.counts, want.counts)
    np.testing.assert_array_equal(got.sums, want.sums)                      # every slot, every kind of carbon
    ref = oracle.OracleEngine(system.tables, trig=oracle.TRIG_LIBM)         # (ignores the flag: the reference's arithmetic)
    ref.submit(xyz, box)
    libm = ref.finish()
    assert np.abs(got.order_ticks() - libm.order_ticks()).max() <= 1
    # the flag does change bits — this is not the default path under another name
    plain = synthetic.ua_membrane(48, leaflets=leaflets, handle_pbc=pbc)
    e2 = HipEngine(plain.tables)
    e2.submit_host(xyz, box, np.arange(n))
    assert not np.array_equal(e2.finish().sums, got.sums)


def test_maps_rows_and_a_shape_equal_the_oracles_fast_mode(built):
    """every variant of the united-atom kernel: maps only (mode 1), rows only (mode 3), everything (mode 2)"""
    om = OrderMap(enabled=True, plane=0, span_x=(0.0, 9.0), span_y=(0.0, 9.0), bin=(0.6, 0.9))
    n = 9
    for kw, geom in ((dict(ordermap=om), None), (dict(timewise=True), None), (dict(ordermap=om, timewise=True), None),
                     (dict(), Geometry(kind=GEOM_CYLINDER, reference=GEOMREF_BOX_CENTER, radius=3.0, orientation=2,
                                       structure_box=(9.0, 9.0, 8.0)))):
        system = fast(synthetic.ua_membrane(32, leaflets=LEAFLETS_GLOBAL, **kw))
        if geom is not None:
            system.tables.geometry = geom
        xyz = system.frames(n, seed=12)
        eng, o, got, want = both(system, xyz, system.box9(n), batches=2)
        np.testing.assert_array_equal(got.counts, want.counts)
        np.testing.assert_array_equal(got.sums, want.sums)
        if "ordermap" in kw:
            np.testing.assert_array_equal(got.map_counts, want.map_counts)
            np.testing.assert_array_equal(got.map_sums, want.map_sums)
        if "timewise" in kw:
            gs, gc = eng.timewise(n)
            ws, wc = o.timewise(n)
            np.testing.assert_array_equal(gs, ws)
            np.testing.assert_array_equal(gc, wc)


def test_atoms_many_box_lengths_away_take_the_literal_loops(built):
    """A molecule made whole far outside the box (an unwrapped trajectory): its hydrogens need more than one box shift,
    which the fast forms hand to the literal-loop evaluation — on the device and in the oracle alike."""
    system = fast(synthetic.ua_membrane(24, leaflets=LEAFLETS_NONE))
    n = 5
    xyz = system.frames(n, seed=21)
    apl = 52
    xyz[:, 3 * apl:4 * apl, 0] += np.float32(3 * system.box[0])        # one lipid three boxes away along x
    xyz[:, 7 * apl:8 * apl, 1] -= np.float32(2 * system.box[1])        # another two boxes below along y
    xyz[:, 9 * apl + 20, 2] += np.float32(4 * system.box[2])           # a single helper atom four boxes up
    eng, o, got, want = both(system, xyz, system.box9(n))
    np.testing.assert_array_equal(got.counts, want.counts)
    np.testing.assert_array_equal(got.sums, want.sums)
    ref = oracle.OracleEngine(system.tables, trig=oracle.TRIG_LIBM)
    ref.submit(xyz, system.box9(n))
    assert np.abs(got.order_ticks() - ref.finish().order_ticks()).max() <= 1


def test_degenerate_carbons_take_the_literal_path(built):
    """two atoms on the same spot: |a|^2 = 0 is outside the fast normalisation's range, the carbon is evaluated like the
    reference does (NaN hydrogen -> tick 0), not with a made-up unit vector"""
    system = fast(synthetic.ua_membrane(8))
    n = 3
    xyz = system.frames(n, seed=2)
    xyz[:, 52 + 15, :] = xyz[:, 52 + 14, :]                            # a carbon on top of its neighbour
    eng, o, got, want = both(system, xyz, system.box9(n))
    np.testing.assert_array_equal(got.counts, want.counts)
    np.testing.assert_array_equal(got.sums, want.sums)


@pytest.mark.parametrize("name,leaflets", [("ua_order_basic.yaml", False), ("ua_order_leaflets.yaml", True)])
def test_reference_goldens_with_the_fast_flag(built, name, leaflets):
    fx = Fixture("ua")
    tables, labels, midx = ua_setup(fx, leaflets=METHODS["global"] if leaflets else None)
    tables.flags |= FLAG_UA_FAST_NORMALISE
    frames = fx.window()
    eng = HipEngine(tables)
    xyz = np.ascontiguousarray(fx.xyz[frames][:, midx, :])
    eng.submit_host(xyz, fx.boxes[frames], frames)
    res = eng.finish()
    bad = st.compare_trees(st.results_tree_ua(res, labels, leaflets=leaflets), expected(name))
    assert not bad, bad[:10]
    o = oracle.OracleEngine(tables, trig=oracle.TRIG_DIRECT, n_threads=4)
    o.submit(xyz, fx.boxes[frames], frames)
    want = o.finish()
    np.testing.assert_array_equal(res.counts, want.counts)
    np.testing.assert_array_equal(res.sums, want.sums)
    libm = oracle.OracleEngine(tables, trig=oracle.TRIG_LIBM, n_threads=4)
    libm.submit(xyz, fx.boxes[frames], frames)
    assert np.abs(res.order_ticks() - libm.finish().order_ticks()).max() <= 1


def test_the_flag_needs_the_default_cosine(built):
    system = synthetic.ua_membrane(8)
    system.tables.flags = FLAG_UA_FAST_NORMALISE | FLAG_TRIG_ACOS_COS
    with pytest.raises(abi.GorderHipError) as ei:
        HipEngine(system.tables)
    assert ei.value.status == abi.ERR_INVALID_ARGUMENT


def test_bond_systems_ignore_the_flag(built):
    system = synthetic.cg_membrane(64, leaflets=LEAFLETS_GLOBAL)
    n = 9
    xyz = system.frames(n, seed=5)
    e1 = HipEngine(system.tables)
    e1.submit_host(xyz, system.box9(n), np.arange(n))
    a = e1.finish()
    system.tables.flags |= FLAG_UA_FAST_NORMALISE
    e2 = HipEngine(system.tables)
    e2.submit_host(xyz, system.box9(n), np.arange(n))
    b = e2.finish()
    np.testing.assert_array_equal(a.sums, b.sums)
