"""GPU parity for the scatter-bound modes: ordermaps (ordermap.rs:100-113), timewise partial sums
(timewise.rs:130-186) and united-atom virtual hydrogens (uaorder.rs:375-437, 947-1104)."""
import numpy as np
import pytest

from gorder_amd import HipEngine, abi, synthetic
from gorder_amd.abi import (GEOM_CUBOID, GEOM_CYLINDER, GEOM_SPHERE, GEOMREF_BOX_CENTER, GEOMREF_GROUP,
                            GEOMREF_POINT, LEAFLETS_GLOBAL, LEAFLETS_INDIVIDUAL, LEAFLETS_NONE, Geometry, OrderMap,
                            UA_CH1_UNSAT, UA_N_H)
from oracle import oracle

pytestmark = pytest.mark.gpu


def both(system, xyz, box, batches=2, trig=None):
    if trig is None:
        trig = oracle.TRIG_MIRROR if (system.tables.flags & abi.FLAG_TRIG_ACOS_COS) else oracle.TRIG_DIRECT
    eng = HipEngine(system.tables)
    o = oracle.OracleEngine(system.tables, trig=trig, n_threads=2)
    edges = np.linspace(0, xyz.shape[0], batches + 1).astype(int)
    for a, b in zip(edges[:-1], edges[1:]):
        fi = np.arange(a, b)
        eng.submit_host(xyz[a:b], None if box is None else box[a:b], fi)
        o.submit(xyz[a:b], None if box is None else box[a:b], fi)
    return eng, o, eng.finish(), o.finish()


@pytest.mark.parametrize("plane", [0, 1, 2])
@pytest.mark.parametrize("leaflets", [LEAFLETS_NONE, LEAFLETS_GLOBAL])
def test_ordermaps(built, plane, leaflets):
    spans = {0: (0, 1), 1: (0, 2), 2: (2, 1)}[plane]     # projection2plane: xy, xz, (z, y)
    system = synthetic.cg_membrane(200, leaflets=leaflets, n_types=2)
    bx = system.box
    om = OrderMap(enabled=True, plane=plane, span_x=(0.0, float(bx[spans[0]])), span_y=(0.0, float(bx[spans[1]])),
                  bin=(0.7, 1.3))
    system.tables.ordermap = om
    xyz = system.frames(9, seed=4)
    eng, o, got, want = both(system, xyz, system.box9(9))
    assert got.map_sums is not None and got.map_sums.shape == want.map_sums.shape
    assert got.map_sums.shape[2] == round(float(bx[spans[0]]) / 0.7) + 1      # GridMap: round(span/bin) + 1 tiles
    np.testing.assert_array_equal(got.sums, want.sums)
    np.testing.assert_array_equal(got.map_counts, want.map_counts)
    np.testing.assert_array_equal(got.map_sums, want.map_sums)
    # a sample lands in at most one tile; the bond position p1 + v/2 is NOT wrapped (bond.rs:422), so a
    # few midpoints beyond the box edge are dropped — by the reference as well (ordermap.rs:100-113)
    tot = got.map_counts.sum(axis=(2, 3))
    assert (tot <= got.counts).all() and tot.sum() > 0.97 * got.counts.sum()


@pytest.mark.parametrize("gather", [False, True])
@pytest.mark.parametrize("leaflets", [LEAFLETS_NONE, LEAFLETS_GLOBAL])
def test_ordermaps_and_timewise_rows_of_bonds_together(built, monkeypatch, gather, leaflets):
    """Staged ordermaps AND per-frame rows: one tiled kernel leaves both kinds of words (k_bonds_tiled_tw<..., MAPS>);
    with GORDER_HIP_TW_GATHER the general k_bonds_extras.  41 frames in two batches against the oracle."""
    if gather:
        monkeypatch.setenv("GORDER_HIP_TW_GATHER", "1")
    system = synthetic.cg_membrane(210, leaflets=leaflets, n_types=2, timewise=True)
    bx = system.box
    system.tables.ordermap = OrderMap(enabled=True, plane=0, span_x=(0.0, float(bx[0])), span_y=(0.0, float(bx[1])), bin=(0.5, 0.7))
    n = 41
    xyz = system.frames(n, seed=14)
    box = system.box9(n)
    eng = HipEngine(system.tables)
    eng.kernel_time()
    o = oracle.OracleEngine(system.tables, trig=oracle.TRIG_DIRECT, n_threads=2)
    for a, b in ((0, 19), (19, n)):
        eng.submit_host(xyz[a:b], box[a:b], np.arange(a, b))
        o.submit(xyz[a:b], box[a:b], np.arange(a, b))
    got, want = eng.finish(), o.finish()
    np.testing.assert_array_equal(got.sums, want.sums)
    np.testing.assert_array_equal(got.counts, want.counts)
    np.testing.assert_array_equal(got.map_counts, want.map_counts)
    np.testing.assert_array_equal(got.map_sums, want.map_sums)
    gs, gc = eng.timewise(n)
    ws, wc = o.timewise(n)
    np.testing.assert_array_equal(gs, ws)
    np.testing.assert_array_equal(gc, wc)
    assert got.map_counts.sum() > 0 and gc[:, 0].sum() > 0
    names = eng.kernel_names()
    assert ("k_bonds_extras" if gather else "k_bonds_tiled_tw") in names and "k_map_accumulate" in names


@pytest.mark.parametrize("gather", [False, True])
@pytest.mark.parametrize("pbc,normal,leaflets", [(True, (0.0, 0.0, 1.0), LEAFLETS_NONE), (True, (0.3, -0.2, 0.9), LEAFLETS_GLOBAL),
                                                 (False, (0.0, 1.0, 0.0), LEAFLETS_NONE), (True, (1.0, 0.0, 0.0), LEAFLETS_INDIVIDUAL)])
def test_timewise_rows_of_bonds_through_both_producers(built, monkeypatch, gather, pbc, normal, leaflets):
    """Per-frame rows and nothing else: out of K1's staging (k_bonds_tiled_tw: the stage's ticks through LDS, a thread per
    slot and frame) — or, with GORDER_HIP_TW_GATHER, out of k_bonds_extras' LDS atomics.  39 frames in two batches (whole
    stages and partial ones), several molecule types, with and without leaflets: the oracle's rows."""
    if gather:
        monkeypatch.setenv("GORDER_HIP_TW_GATHER", "1")
    system = synthetic.cg_membrane(230, leaflets=leaflets, n_types=3, handle_pbc=pbc, normal=normal, timewise=True)
    n = 39
    xyz = system.frames(n, seed=13)
    box = system.box9(n) if pbc else None
    eng = HipEngine(system.tables)
    eng.kernel_time()
    o = oracle.OracleEngine(system.tables, trig=oracle.TRIG_DIRECT, n_threads=2)
    for a, b in ((0, 22), (22, n)):
        eng.submit_host(xyz[a:b], None if box is None else box[a:b], np.arange(a, b))
        o.submit(xyz[a:b], None if box is None else box[a:b], np.arange(a, b))
    got, want = eng.finish(), o.finish()
    np.testing.assert_array_equal(got.sums, want.sums)
    np.testing.assert_array_equal(got.counts, want.counts)
    gs, gc = eng.timewise(n)
    ws, wc = o.timewise(n)
    np.testing.assert_array_equal(gs, ws)
    np.testing.assert_array_equal(gc, wc)
    assert gc[:, 0].sum() > 0 and (leaflets == LEAFLETS_NONE or (gc[:, 1].sum() > 0 and gc[:, 2].sum() > 0))
    assert ("k_bonds_extras" if gather else "k_bonds_tiled_tw") in eng.kernel_names()


@pytest.mark.parametrize("gather", [False, True])
@pytest.mark.parametrize("pbc,normal,leaflets", [(True, (0.0, 0.0, 1.0), LEAFLETS_NONE), (True, (0.3, -0.2, 0.9), LEAFLETS_GLOBAL),
                                                 (False, (0.0, 1.0, 0.0), LEAFLETS_NONE), (True, (1.0, 0.0, 0.0), LEAFLETS_INDIVIDUAL)])
def test_ordermaps_of_bonds_through_both_producers(built, monkeypatch, gather, pbc, normal, leaflets):
    """Maps and nothing else: the samples come out of K1's staging (k_bonds_tiled_maps) — or, with GORDER_HIP_MAPS_GATHER,
    out of k_bonds_extras — and go through k_map_accumulate.  37 frames in one batch (nine whole stages and a partial
    one), periodic or not, the normal an axis or not, with and without leaflets: the oracle's maps tile for tile."""
    if gather:
        monkeypatch.setenv("GORDER_HIP_MAPS_GATHER", "1")
    system = synthetic.cg_membrane(230, leaflets=leaflets, n_types=3, handle_pbc=pbc, normal=normal)
    bx = system.box
    system.tables.ordermap = OrderMap(enabled=True, plane=0, span_x=(0.0, float(bx[0])), span_y=(0.0, float(bx[1])), bin=(0.45, 0.8))
    n = 37
    xyz = system.frames(n, seed=12)
    box = system.box9(n) if pbc else None
    eng = HipEngine(system.tables)
    eng.kernel_time()                       # (switches the event pairs on: the kernels of the batch are then named)
    o = oracle.OracleEngine(system.tables, trig=oracle.TRIG_DIRECT, n_threads=2)
    eng.submit_host(xyz, box, np.arange(n))
    o.submit(xyz, box, np.arange(n))
    got, want = eng.finish(), o.finish()
    np.testing.assert_array_equal(got.sums, want.sums)
    np.testing.assert_array_equal(got.counts, want.counts)
    np.testing.assert_array_equal(got.map_counts, want.map_counts)
    np.testing.assert_array_equal(got.map_sums, want.map_sums)
    assert got.map_counts.sum() > 0
    names = eng.kernel_names()
    assert ("k_bonds_extras" if gather else "k_bonds_tiled_maps") in names and "k_map_accumulate" in names


@pytest.mark.parametrize("leaflets", [LEAFLETS_NONE, LEAFLETS_GLOBAL])
def test_ordermap_words_are_folded_before_they_overflow(built, monkeypatch, leaflets):
    """The kernels add count and tick sum into one packed word per tile (k_fold_maps); with the fold
    limit lowered to a handful of samples the frames go in sub-ranges of 2 with a fold between them, and
    the unpacked maps must still be the oracle's — including negative tick sums next to the count bits."""
    monkeypatch.setenv("GORDER_HIP_MAP_FOLD_LIMIT", "250")
    system = synthetic.cg_membrane(100, leaflets=leaflets)
    bx = system.box
    system.tables.ordermap = OrderMap(enabled=True, plane=0, span_x=(0.0, float(bx[0])), span_y=(0.0, float(bx[1])),
                                      bin=(2.5, 2.5))       # coarse tiles: many samples (and sign changes) per word
    xyz = system.frames(23, seed=8)
    eng, o, got, want = both(system, xyz, system.box9(23), batches=3)
    assert (want.map_sums < 0).any() and (want.map_counts > 100).any()
    np.testing.assert_array_equal(got.map_counts, want.map_counts)
    np.testing.assert_array_equal(got.map_sums, want.map_sums)
    np.testing.assert_array_equal(got.sums, want.sums)


def test_ordermap_manual_span_drops_outside_samples(built):
    system = synthetic.aa_membrane(20)
    system.tables.ordermap = OrderMap(enabled=True, plane=0, span_x=(2.0, 5.0), span_y=(1.0, 4.0), bin=(0.25, 0.5))
    xyz = system.frames(5, seed=2)
    eng, o, got, want = both(system, xyz, system.box9(5), batches=1)
    np.testing.assert_array_equal(got.map_counts, want.map_counts)
    np.testing.assert_array_equal(got.map_sums, want.map_sums)
    assert 0 < got.map_counts[0].sum() < got.counts[0].sum()


@pytest.mark.parametrize("leaflets", [LEAFLETS_NONE, LEAFLETS_INDIVIDUAL])
def test_timewise(built, leaflets):
    system = synthetic.aa_membrane(24, leaflets=leaflets, timewise=True)
    n = 31
    xyz = system.frames(n, seed=6)
    eng, o, got, want = both(system, xyz, system.box9(n), batches=3)
    np.testing.assert_array_equal(got.sums, want.sums)
    gs, gc = eng.timewise(n)
    ws, wc = o.timewise(n)
    np.testing.assert_array_equal(gs, ws)
    np.testing.assert_array_equal(gc, wc)
    # the per-frame rows add up to the running totals, and feed the reference's error estimate
    np.testing.assert_array_equal(gs.sum(axis=0), got.sums)
    np.testing.assert_array_equal(gc.sum(axis=0), got.counts)
    e_gpu = oracle.estimate_error(gs[:, 0, 3], gc[:, 0, 3], 5)
    e_ref = oracle.estimate_error(ws[:, 0, 3], wc[:, 0, 3], 5)
    assert e_gpu == e_ref and e_gpu > 0


def interleaved_ua_types(n_lipids=150):
    """The synthetic united-atom membrane split into two molecule types whose molecules ALTERNATE in the atom order
    (even lipids: every order carbon; odd lipids: the methylene carbons 11..19 and one methyl only), so that the
    molecule index is not monotonic along the atoms — the tile builder groups by molecule ranges."""
    import copy
    system = synthetic.ua_membrane(n_lipids)
    (mt,) = system.tables.molecule_types
    even, odd = np.arange(0, n_lipids, 2), np.arange(1, n_lipids, 2)
    a = copy.copy(mt)
    a.n_molecules, a.name = len(even), "EVEN"
    a.ua_atoms = [(k, np.ascontiguousarray(ix[even])) for k, ix in mt.ua_atoms]
    b = copy.copy(mt)
    b.n_molecules, b.name = len(odd), "ODD"
    b.ua_atoms = [(k, np.ascontiguousarray(ix[odd])) for c, (k, ix) in enumerate(mt.ua_atoms, start=10) if 11 <= c <= 19 or c == 41]
    for t in (a, b):
        t.heads = t.methyls = None
    system.tables = copy.copy(system.tables)
    system.tables.molecule_types = [a, b]
    return system


def test_united_atoms_with_interleaved_molecule_types(built):
    system = interleaved_ua_types()
    assert abi.plan_tables(system.tables)["selfcheck"] == 0
    n = 7
    xyz = system.frames(n, seed=3)
    eng, o, got, want = both(system, xyz, system.box9(n))
    assert got.sums.shape[1] == 62 + (8 * 2 + 1) + 3       # all carbons; 8 CH2 + the saturated CH of 11..19; one methyl
    np.testing.assert_array_equal(got.counts, want.counts)
    assert (got.counts[0] == 75 * n).all()
    assert np.abs(got.order_ticks() - want.order_ticks()).max() <= 1


@pytest.mark.parametrize("lanes", ["4 slots x 16 molecules", "a slot per wave"])
@pytest.mark.parametrize("pbc", [True, False])
@pytest.mark.parametrize("leaflets", [LEAFLETS_NONE, LEAFLETS_GLOBAL])
def test_united_atoms(built, monkeypatch, leaflets, pbc, lanes):
    # (how the carbons of a group of molecules are dealt to the lanes: the default since round 4, and rounds 1-3's order)
    if lanes == "a slot per wave":
        monkeypatch.setenv("GORDER_HIP_UA_SLOT_WAVES", "1")
    system = synthetic.ua_membrane(40, leaflets=leaflets, handle_pbc=pbc)
    n = 11
    xyz = system.frames(n, seed=8)
    box = system.box9(n) if pbc else None
    eng, o, got, want = both(system, xyz, box)
    np.testing.assert_array_equal(got.counts, want.counts)
    # slots of CH3 / CH2 / saturated CH are built from IEEE f32 operations only -> bit-exact;
    # the unsaturated CH uses acos + sin/cos of a data-dependent angle (device vs host libm): <= 1e-6
    slot, exact = 0, []
    for kind, _ in system.tables.molecule_types[0].ua_atoms:
        nh = UA_N_H[int(kind)]
        exact += [kind != UA_CH1_UNSAT] * nh
        slot += nh
    exact = np.array(exact)
    assert exact.size == got.sums.shape[1] == 62
    np.testing.assert_array_equal(got.sums[:, exact], want.sums[:, exact])
    # ... and so is the unsaturated CH since the device evaluates its angle with kernels the oracle restates
    np.testing.assert_array_equal(got.sums[:, ~exact], want.sums[:, ~exact])
    # and everything within 1e-6 of the libm (reference-faithful) oracle
    ref = oracle.OracleEngine(system.tables, trig=oracle.TRIG_LIBM)
    ref.submit(xyz, box)
    assert np.abs(got.order_ticks() - ref.finish().order_ticks()).max() <= 1
    # C-H bond length is 0.109 nm by construction -> order parameters are physical
    s = got.order()[0]
    assert np.all(s >= -0.5 - 1e-6) and np.all(s <= 1.0 + 1e-6)


@pytest.mark.parametrize("leaflets", [LEAFLETS_NONE, LEAFLETS_GLOBAL])
def test_united_atoms_in_the_literal_mode_are_the_references_arithmetic(built, leaflets):
    """GORDER_FLAG_TRIG_ACOS_COS on united atoms: the construction is IEEE f32 operation for operation, the one
    data-dependent angle (unsaturated CH: acos, sin, cos) and P2's acos -> cos go through restatements of glibc's
    algorithms — so on a glibc 2.28 - 2.40 host the device's integers are those of the reference-faithful (libm) oracle."""
    import platform
    system = synthetic.ua_membrane(48, leaflets=leaflets)
    system.tables.flags = abi.FLAG_TRIG_ACOS_COS
    n = 9
    xyz = system.frames(n, seed=14)
    eng, o, got, want = both(system, xyz, system.box9(n))           # (MIRROR mode)
    np.testing.assert_array_equal(got.counts, want.counts)
    np.testing.assert_array_equal(got.sums, want.sums)
    ref = oracle.OracleEngine(system.tables, trig=oracle.TRIG_LIBM)
    ref.submit(xyz, system.box9(n))
    libm = ref.finish()
    assert np.abs(got.order_ticks() - libm.order_ticks()).max() <= 1
    if platform.libc_ver()[0] == "glibc" and "2.28" <= platform.libc_ver()[1] <= "2.40":
        flags, _ = eng.leaflets() if leaflets else (None, None)
        np.testing.assert_array_equal(got.sums[0], libm.sums[0])
        if not leaflets:
            np.testing.assert_array_equal(got.sums, libm.sums)


@pytest.mark.parametrize("lanes,n_lipids", [("4 slots x 16 molecules", 16), ("4 slots x 16 molecules", 100), ("a slot per wave", 70)])
def test_united_atoms_with_maps_and_timewise(built, monkeypatch, lanes, n_lipids):
    if lanes == "a slot per wave":
        monkeypatch.setenv("GORDER_HIP_UA_SLOT_WAVES", "1")
    system = synthetic.ua_membrane(n_lipids, leaflets=LEAFLETS_GLOBAL, timewise=True,
                                   ordermap=OrderMap(enabled=True, plane=0, span_x=(0.0, 9.0), span_y=(0.0, 9.0),
                                                     bin=(1.0, 1.0)))
    n = 7
    xyz = system.frames(n, seed=3)
    eng, o, got, want = both(system, xyz, system.box9(n), batches=2)
    np.testing.assert_array_equal(got.counts, want.counts)
    assert (got.map_counts.sum(axis=(2, 3)) <= got.counts).all()
    np.testing.assert_array_equal(got.map_counts, want.map_counts)
    gs, gc = eng.timewise(n)
    np.testing.assert_array_equal(gc.sum(axis=0), got.counts)
    np.testing.assert_array_equal(gs.sum(axis=0), got.sums)


INF = float("inf")


@pytest.mark.parametrize("pbc", [True, False])
@pytest.mark.parametrize("geom", [
    Geometry(kind=GEOM_CUBOID, reference=GEOMREF_GROUP, xdim=(-3.0, 2.0), ydim=(-INF, INF), zdim=(-1.0, 4.0)),
    Geometry(kind=GEOM_CUBOID, reference=GEOMREF_POINT, point=(9.0, 1.0, 3.0), xdim=(-2.0, 1.5), ydim=(-2.0, 3.0), invert=True),
    Geometry(kind=GEOM_CYLINDER, reference=GEOMREF_GROUP, radius=2.5, span=(-2.0, 2.5), orientation=2),
    Geometry(kind=GEOM_CYLINDER, reference=GEOMREF_POINT, point=(1.0, 11.0, 2.0), radius=2.0, orientation=0, invert=True),
    Geometry(kind=GEOM_SPHERE, reference=GEOMREF_GROUP, radius=3.0),
    Geometry(kind=GEOM_SPHERE, reference=GEOMREF_POINT, point=(0.3, 0.2, 5.0), radius=3.2),
])
def test_geometry_selection(built, geom, pbc):
    """geometry.rs: cuboid / cylinder / sphere, fixed-point and group-centre references, inverted or not,
    with and without PBC; leaflets + timewise on top so that the filtered counts flow everywhere."""
    system = synthetic.cg_membrane(160, leaflets=LEAFLETS_GLOBAL, n_types=2, handle_pbc=pbc, timewise=True)
    geom.structure_box = tuple(float(x) for x in system.box)
    if geom.reference == GEOMREF_GROUP:
        p0 = system.frames(1, seed=0)[0].astype(np.float64)      # a localised group (2-nm blob): its centre is well defined
        d = p0 - p0[0]
        if pbc:
            d -= system.box.astype(np.float64) * np.round(d / system.box.astype(np.float64))
        geom.group = np.flatnonzero(np.linalg.norm(d, axis=1) < 2.0).astype(np.uint32)
    system.tables.geometry = geom
    n = 9
    xyz = system.frames(n, seed=13)
    eng, o, got, want = both(system, xyz, system.box9(n) if pbc else None)
    total = n * system.tables.n_samples_per_frame
    assert 0 < got.counts[0].sum() <= total
    if pbc:
        assert got.counts[0].sum() < total      # every shape above really filters in the periodic box
    np.testing.assert_array_equal(got.counts, want.counts)
    np.testing.assert_array_equal(got.sums, want.sums)
    gs, gc = eng.timewise(n)
    ws, wc = o.timewise(n)
    np.testing.assert_array_equal(gc, wc)
    np.testing.assert_array_equal(gs, ws)


def test_geometry_box_centre_and_united_atoms(built):
    system = synthetic.ua_membrane(24)
    system.tables.geometry = Geometry(kind=GEOM_CYLINDER, reference=GEOMREF_BOX_CENTER, radius=2.5, orientation=2)
    xyz = system.frames(6, seed=1)
    eng, o, got, want = both(system, xyz, system.box9(6))
    np.testing.assert_array_equal(got.counts, want.counts)
    assert 0 < got.counts[0].sum() < 6 * system.tables.n_samples_per_frame
    # box-centre reference without PBC is refused (NoPBC::get_box_center panics, pbc.rs:243-245)
    s2 = synthetic.cg_membrane(20, handle_pbc=False)
    s2.tables.geometry = Geometry(kind=GEOM_SPHERE, reference=GEOMREF_BOX_CENTER, radius=1.0)
    with pytest.raises(abi.GorderHipError) as e:
        HipEngine(s2.tables)
    assert e.value.status == abi.ERR_INVALID_ARGUMENT


# ---- dynamic membrane normals (normal.rs:160-199, 421-458) ------------------------------------------
def _with_dynamic_normals(system, radius):
    from gorder_amd.abi import DynamicNormal
    cloud = []
    for mt in system.tables.molecule_types:
        mt.normal_heads = np.asarray(mt.heads, dtype=np.uint32)
        cloud.append(mt.normal_heads)
    system.tables.dynamic_normal = DynamicNormal(enabled=True, radius=radius, cloud=np.concatenate(cloud))
    return system


@pytest.mark.parametrize("pbc", [True, False])
@pytest.mark.parametrize("kind", ["cg", "ua"])
def test_dynamic_normals(built, kind, pbc):
    if kind == "cg":
        system = synthetic.cg_membrane(300, leaflets=LEAFLETS_INDIVIDUAL, n_types=2, handle_pbc=pbc)
    else:
        system = synthetic.ua_membrane(120, leaflets=LEAFLETS_GLOBAL, handle_pbc=pbc)
    _with_dynamic_normals(system, 2.2)
    n = 9
    xyz = system.frames(n, seed=12)
    box = system.box9(n) if pbc else None
    eng, o, got, want = both(system, xyz, box, batches=2)
    np.testing.assert_array_equal(got.counts, want.counts)
    assert np.abs(got.order_ticks() - want.order_ticks()).max() <= 1
    n_gpu, k_gpu = eng.normals()
    n_ref, k_ref = o.normals()
    np.testing.assert_array_equal(k_gpu, k_ref)
    assert k_ref.min() >= 3 and np.abs(n_gpu - n_ref).max() < 1e-6
    # with two leaflets ~4 nm apart only the head's own leaflet is within 2.2 nm: normals near +-z, and they
    # really differ from the static axis (else this test would not notice a missing normal)
    assert 0.8 < np.abs(n_ref[:, 2]).mean() < 0.99999


def test_dynamic_normals_not_enough_points(built):
    """DynamicNormalError::NotEnoughPoints(n): raised for a molecule whose cloud has < 3 heads — but only if a
    sample of that molecule is accumulated (AA/CG: after the geometry test, bond.rs:424-431)."""
    from gorder_amd import GorderHipError
    system = _with_dynamic_normals(synthetic.cg_membrane(60, leaflets=LEAFLETS_GLOBAL), 0.3)   # heads are ~0.8 nm apart
    xyz = system.frames(3, seed=1)
    eng = HipEngine(system.tables)
    with pytest.raises(GorderHipError) as ei:
        eng.submit_host(xyz, system.box9(3), np.arange(3))
        eng.finish()
    assert ei.value.status == 7 and ei.value.index in (1, 2)        # the cloud size rides in the error index
    o = oracle.OracleEngine(system.tables, trig=oracle.TRIG_DIRECT)
    with pytest.raises(oracle.OracleError) as eo:
        o.submit(xyz, system.box9(3), np.arange(3))
    assert eo.value.status == 7
    # a geometry that contains no sample: nobody asks for a normal, no error
    system.tables.geometry = Geometry(kind=GEOM_SPHERE, reference=GEOMREF_POINT, point=(1.0, 1.0, 0.2), radius=0.1,
                                      structure_box=tuple(float(x) for x in system.box))
    eng = HipEngine(system.tables)
    eng.submit_host(xyz, system.box9(3), np.arange(3))
    assert eng.finish().counts.sum() == 0


def test_ordermaps_reduce_across_ranks(built):
    """Multi-GPU form of Map::add (ordermap.rs:116-138): each rank exports its maps into tensors it owns, the
    host sums them (here: two handles on one GPU stand in for two ranks; bench.py does it with RCCL) — the
    result is bit-identical to one handle that saw every frame."""
    import torch
    system = synthetic.ua_membrane(30, leaflets=LEAFLETS_GLOBAL)
    bx = system.box
    system.tables.ordermap = OrderMap(enabled=True, plane=0, span_x=(0.0, float(bx[0])), span_y=(0.0, float(bx[1])),
                                      bin=(0.9, 1.1))
    n = 14
    xyz, box9 = system.frames(n, seed=21), system.box9(n)
    whole = HipEngine(system.tables)
    whole.submit_host(xyz, box9, np.arange(n))
    want = whole.finish()
    parts = []
    for lo, hi in ((0, 6), (6, n)):
        eng = HipEngine(system.tables)
        eng.submit_host(xyz[lo:hi], box9[lo:hi], np.arange(lo, hi))
        s = torch.zeros(want.map_sums.size, dtype=torch.int64, device="cuda")
        c = torch.zeros_like(s)
        eng.export_maps(s, c)
        parts.append((s, c, eng))
    s = (parts[0][0] + parts[1][0]).cpu().numpy().reshape(want.map_sums.shape)
    c = (parts[0][1] + parts[1][1]).cpu().numpy().reshape(want.map_counts.shape)
    np.testing.assert_array_equal(s, want.map_sums)
    np.testing.assert_array_equal(c.astype(np.uint64), want.map_counts)
    assert c.sum() > 0


@pytest.mark.parametrize("direct", [False, True])
@pytest.mark.parametrize("leaflets", [LEAFLETS_NONE, LEAFLETS_GLOBAL])
def test_united_atom_ordermaps_staged_and_direct(built, monkeypatch, leaflets, direct):
    """United-atom ordermaps go through a staging buffer + slot-major LDS accumulation (k_map_accumulate) by
    default, through one atomic per sample with GORDER_HIP_MAP_DIRECT=1; both must give the oracle's maps, also
    when the frames are cut into sub-ranges of 2 (fold limit) that reuse the staging buffer."""
    if direct:
        monkeypatch.setenv("GORDER_HIP_MAP_DIRECT", "1")
    monkeypatch.setenv("GORDER_HIP_MAP_FOLD_LIMIT", "100")          # 40 molecules -> sub-ranges of 2 frames
    system = synthetic.ua_membrane(40, leaflets=leaflets)
    bx = system.box
    system.tables.ordermap = OrderMap(enabled=True, plane=0, span_x=(0.0, float(bx[0])), span_y=(0.0, float(bx[1])),
                                      bin=(1.5, 1.5))
    n = 11
    xyz = system.frames(n, seed=31)
    eng, o, got, want = both(system, xyz, system.box9(n), batches=2)
    np.testing.assert_array_equal(got.map_counts, want.map_counts)
    ch1u = np.zeros(got.map_sums.shape[1], dtype=bool)              # unsaturated CH: device sincosf vs host libm
    slot = 0
    for kind, _ in system.tables.molecule_types[0].ua_atoms:
        nh = abi.UA_N_H[int(kind)]
        ch1u[slot:slot + nh] = int(kind) == abi.UA_CH1_UNSAT
        slot += nh
    np.testing.assert_array_equal(got.map_sums[:, ~ch1u], want.map_sums[:, ~ch1u])
    np.testing.assert_array_equal(got.map_sums[:, ch1u], want.map_sums[:, ch1u])
    assert got.map_counts.sum() > 0


@pytest.mark.parametrize("kind", ["cg", "ua"])
def test_manual_normals(built, kind):
    """MembraneNormal::Manual (normal.rs:266-300): one vector per frame and molecule, supplied by the host for
    each batch.  z for everybody reproduces the static-axis run bit for bit; arbitrary vectors match the oracle."""
    system = synthetic.cg_membrane(90, leaflets=LEAFLETS_GLOBAL, n_types=2) if kind == "cg" else \
        synthetic.ua_membrane(30, leaflets=LEAFLETS_GLOBAL)
    n, n_mol = 10, system.tables.n_molecules_total
    xyz, box = system.frames(n, seed=41), system.box9(n)
    static = HipEngine(system.tables)
    static.submit_host(xyz, box, np.arange(n))
    want_static = static.finish()
    z = np.zeros((n, n_mol, 3), dtype=np.float32)
    z[:, :, 2] = 1.0
    rng = np.random.default_rng(2)
    tilted = rng.normal(size=(n, n_mol, 3)).astype(np.float32) * 0.4 + z       # not unit length: calc_sch normalises
    for normals, same_as_static in ((z, True), (tilted, False)):
        eng = HipEngine(system.tables)
        o = oracle.OracleEngine(system.tables, trig=oracle.TRIG_DIRECT, n_threads=2)
        for a, b in ((0, 4), (4, n)):
            eng.set_normals(normals[a:b])
            eng.submit_host(xyz[a:b], box[a:b], np.arange(a, b))
            o.set_normals(normals[a:b])
            o.submit(xyz[a:b], box[a:b], np.arange(a, b))
        got, want = eng.finish(), o.finish()
        np.testing.assert_array_equal(got.counts, want.counts)
        if kind == "cg":
            np.testing.assert_array_equal(got.sums, want.sums)
            if same_as_static:
                np.testing.assert_array_equal(got.sums, want_static.sums)
            else:
                assert (got.sums != want_static.sums).any()
        else:           # unsaturated CH hydrogens: device sincosf vs host libm, <= 1 tick
            assert np.abs(got.order_ticks() - want.order_ticks()).max() <= 1
    # the batch size must match
    eng = HipEngine(system.tables)
    eng.set_normals(z[:3])
    with pytest.raises(abi.GorderHipError):
        eng.submit_host(xyz[:4], box[:4], np.arange(4))
