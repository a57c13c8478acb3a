"""BASELINE.json configs[2] and configs[4] at their own size, against the oracle and through size-independent
properties: CGOrder with 3 072 lipids + local-distance leaflets (r = 2.5 nm), and the 1 000 008-bead bilayer
(83 334 lipids, 916 674 bonds per frame, 3 581 tiles, frame offsets beyond 2^32 bytes AND beyond 2^32 floats)."""
import numpy as np
import pytest

from gorder_amd import HipEngine, synthetic
from gorder_amd.abi import LEAFLETS_GLOBAL, LEAFLETS_LOCAL
from oracle import oracle
from leaflet_check import assert_sums_given_device_flags

pytestmark = pytest.mark.gpu


def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def test_cg3k_local_leaflets_at_full_size(built):
    """configs[2]: 3 072 lipids x 12 beads, Local classification with r = 2.5 nm on every frame.  The oracle's
    brute-force cylinders (113 M distance tests per frame) on a handful of frames: flags, distances, sums."""
    torch_cuda()
    system = synthetic.cg_membrane(3072, leaflets=LEAFLETS_LOCAL, radius=2.5)
    assert system.n_atoms == 36864 and system.tables.n_samples_per_frame == 33792
    n = 4
    xyz = system.frames(n, seed=7)
    box = system.box9(n)
    eng = HipEngine(system.tables)
    eng.submit_host(xyz, box)
    got = eng.finish()
    o = oracle.OracleEngine(system.tables, trig=oracle.TRIG_DIRECT, n_threads=1)
    o.submit(xyz, box)
    want = o.finish()
    flags, fr = eng.leaflets()
    oflags, odist, ofr = o.leaflets()
    assert fr == ofr == n - 1
    assert 1400 < int(flags.sum()) < 1700                      # 1 536 lipids per leaflet
    diff = flags != oflags
    assert diff.sum() <= 2 and (not diff.any() or np.abs(odist[diff]).max() < 1e-4)
    np.testing.assert_allclose(eng.leaflet_distances()[~diff], odist[~diff], atol=5e-5)
    np.testing.assert_array_equal(got.sums[0], want.sums[0])
    np.testing.assert_array_equal(got.counts[0], want.counts[0])
    assert_sums_given_device_flags(system.tables, xyz, box, got, max_flag_diffs=4)
    # within 1e-6 of the reference-faithful (libm) arithmetic
    libm = oracle.OracleEngine(system.tables, trig=oracle.TRIG_LIBM, n_threads=1)
    libm.submit(xyz, box)
    lw = libm.finish()
    assert np.abs(got.order_ticks()[0] - lw.order_ticks()[0]).max() <= 1
    # upper / lower against the INDEPENDENT libm run too (not only through the oracle re-run with the device's flags):
    # all three rows when the two assignments agree, otherwise every bond type of the molecule types whose lipids
    # were assigned alike
    lflags, _, _ = libm.leaflets()
    if np.array_equal(flags, lflags):
        np.testing.assert_array_equal(got.counts, lw.counts)
        assert np.abs(got.order_ticks() - lw.order_ticks()).max() <= 1
    else:
        same_counts = (got.counts == lw.counts).all(axis=0)
        assert same_counts.mean() > 0.5
        assert np.abs(got.order_ticks() - lw.order_ticks())[:, same_counts].max() <= 1


@pytest.fixture(scope="module")
def cg1m(built):
    return synthetic.cg_membrane(83334)


def test_cg1m_against_the_oracle(cg1m):
    """configs[4]: every one of the 916 674 bonds of a frame against the oracle, bit for bit, on three frames; with
    global leaflets on two (the membrane group is the whole million-bead frame)."""
    torch_cuda()
    system = cg1m
    assert system.n_atoms == 1000008 and system.tables.n_samples_per_frame == 916674
    n = 3
    xyz = system.frames(n, seed=41)
    box = system.box9(n)
    eng = HipEngine(system.tables)
    eng.submit_host(xyz, box)
    got = eng.finish()
    assert eng.plan()["n_tiles"] == 3581 and eng.plan()["n_direct_items"] == 0
    o = oracle.OracleEngine(system.tables, trig=oracle.TRIG_DIRECT, n_threads=3)
    o.submit(xyz, box)
    want = o.finish()
    np.testing.assert_array_equal(got.sums, want.sums)
    np.testing.assert_array_equal(got.counts, want.counts)
    assert (got.counts[0] == n * 83334).all()
    libm = oracle.OracleEngine(system.tables, trig=oracle.TRIG_LIBM, n_threads=3)
    libm.submit(xyz, box)
    assert np.abs(got.order_ticks()[0] - libm.finish().order_ticks()[0]).max() <= 1
    lsys = synthetic.cg_membrane(83334, leaflets=LEAFLETS_GLOBAL)
    eng = HipEngine(lsys.tables)
    eng.submit_host(xyz[:2], box[:2])
    lgot = eng.finish()
    assert 41000 < int(eng.leaflets()[0].sum()) < 42500
    assert_sums_given_device_flags(lsys.tables, xyz[:2], box[:2], lgot)


def test_cg1m_one_launch_beyond_32_bit_offsets(cg1m):
    """One submit of 1 500 resident frames (18 GB): frame offsets pass 2^32 bytes after frame 357 and 2^32 floats
    after frame 1 431.  Linearity over frame ranges cut on both sides of those marks (each part is a launch of its
    own that starts at offset 0, so the whole can only equal their sum if it read the far frames correctly), sample
    counts, and the oracle on a strided subset that includes frames beyond both marks."""
    torch = torch_cuda()
    system = cg1m
    n = 1500
    d_xyz, d_box = system.frames_device(n, seed=5)
    assert d_xyz.numel() > 2 ** 32 and d_xyz.numel() * 4 > 2 ** 34
    eng = HipEngine(system.tables)
    eng.use_torch_stream()
    eng.submit_device(d_xyz, d_box)
    whole = eng.finish()
    assert whole.n_frames == n and (whole.counts[0] == n * 83334).all()
    parts = 0
    for a, b in ((0, 357), (357, 359), (359, 1431), (1431, 1433), (1433, n)):
        e = HipEngine(system.tables)
        e.use_torch_stream()
        e.submit_device(d_xyz[a:b], d_box[a:b], np.arange(a, b))
        parts = parts + e.finish().sums
    np.testing.assert_array_equal(parts, whole.sums)
    sel = np.array([0, 358, 1432, n - 1])
    sub = d_xyz[torch.from_numpy(sel).cuda()].contiguous()
    e = HipEngine(system.tables)
    e.use_torch_stream()
    e.submit_device(sub, d_box[: len(sel)].contiguous())
    got = e.finish()
    o = oracle.OracleEngine(system.tables, trig=oracle.TRIG_DIRECT, n_threads=4)
    o.submit(sub.cpu().numpy(), system.box9(len(sel)))
    want = o.finish()
    np.testing.assert_array_equal(got.sums, want.sums)
    np.testing.assert_array_equal(got.counts, want.counts)
    s = whole.order()[0]
    assert np.all(s >= -0.5 - 1e-6) and np.all(s <= 1.0 + 1e-6)


# ---- configs[3]: UAOrder + 2-D ordermap scatter at its own size ----------------------------------------------
def _vua(leaflets):
    from gorder_amd.abi import OrderMap
    om = OrderMap(enabled=True, plane=0, span_x=(0.0, 9.0), span_y=(0.0, 9.0), bin=(0.1, 0.1))     # bench.py's ua256-maps
    return synthetic.ua_membrane(256, leaflets=leaflets, ordermap=om)


def _unsaturated_slots(system):
    from gorder_amd import abi
    out, slot = np.zeros(system.tables.n_acc, dtype=bool), 0
    for kind, _ in system.tables.molecule_types[0].ua_atoms:
        nh = abi.UA_N_H[int(kind)]
        out[slot:slot + nh] = int(kind) == abi.UA_CH1_UNSAT
        slot += nh
    return out


@pytest.mark.parametrize("direct", [False, True])
@pytest.mark.parametrize("leaflets", [0, LEAFLETS_GLOBAL])
def test_vua_ordermaps_at_full_size(built, monkeypatch, leaflets, direct):
    """configs[3] (V-UA): 256 united-atom lipids x 62 virtual C-H, 91 x 91 tiles per slot (ordermap.rs:100-113;
    GridMap: round(9.0 / 0.1) + 1 tiles).  One slot's packed map is 66 KB (> 64 KB: the dynamic-LDS branch of
    k_map_accumulate), 132 KB with leaflets (upper + lower planes).  Staged route (default) and one atomic per
    sample (GORDER_HIP_MAP_DIRECT=1), several batches so that the staging buffer is reused, against the oracle:
    sums, counts, map sums and map counts EQUAL in all 62 slots (the oracle's DIRECT mode restates the device's
    arithmetic, the sin / cos / acos kernels of the unsaturated CH included); every order parameter within 1e-6 of
    the libm (reference-faithful) mode."""
    torch_cuda()
    if direct:
        monkeypatch.setenv("GORDER_HIP_MAP_DIRECT", "1")
    system = _vua(leaflets)
    assert system.n_atoms == 13312 and system.tables.n_samples_per_frame == 256 * 62
    n = 6
    xyz, box = system.frames(n, seed=17), system.box9(n)
    eng = HipEngine(system.tables)
    o = oracle.OracleEngine(system.tables, trig=oracle.TRIG_DIRECT, n_threads=2)
    for a, b in ((0, 1), (1, 4), (4, n)):
        eng.submit_host(xyz[a:b], box[a:b], np.arange(a, b))
        o.submit(xyz[a:b], box[a:b], np.arange(a, b))
    got, want = eng.finish(), o.finish()
    assert eng.ordermap_dims() == (91, 91) and got.map_sums.shape == want.map_sums.shape == (3, 62, 91, 91)
    plan = eng.plan()
    assert plan["map_staged"] == int(not direct)                # which route really ran
    assert plan["map_lds_bytes"] == (2 if leaflets else 1) * 91 * 91 * 8 > 64 * 1024
    un = _unsaturated_slots(system)
    assert un.sum() == 2
    np.testing.assert_array_equal(got.counts, want.counts)
    assert (got.counts[0] == 256 * n).all()
    np.testing.assert_array_equal(got.map_counts, want.map_counts)
    np.testing.assert_array_equal(got.sums, want.sums)
    np.testing.assert_array_equal(got.map_sums, want.map_sums)
    assert (got.counts[0, un] == 256 * n).all()
    # every sample of a frame lands in one tile (hydrogens are wrapped into the box, uaorder.rs:979-1104; the bond
    # position H + v/2 may leave it by half a bond: those are dropped by the reference as well)
    tot = got.map_counts[0].sum(axis=(1, 2))
    assert (tot <= got.counts[0]).all() and tot.sum() > 0.97 * got.counts[0].sum()
    if leaflets:
        np.testing.assert_array_equal(got.map_counts[0], got.map_counts[1] + got.map_counts[2])
        np.testing.assert_array_equal(got.map_sums[0], got.map_sums[1] + got.map_sums[2])
        assert got.map_counts[1].sum() > 0 and got.map_counts[2].sum() > 0
    else:
        assert got.map_counts[1:].sum() == 0
    # within 1e-6 of the reference-faithful (libm) arithmetic
    libm = oracle.OracleEngine(system.tables, trig=oracle.TRIG_LIBM, n_threads=2)
    libm.submit(xyz, box)
    assert np.abs(got.order_ticks() - libm.finish().order_ticks()).max() <= 1


def test_vua_maps_many_frames_linearity(built):
    """The same workload on enough frames that k_map_accumulate cuts its frame ranges (600 frames), through
    size-independent properties: two halves add up to the whole (maps included), total = upper + lower,
    every slot counts 256 samples per frame."""
    torch = torch_cuda()
    system = _vua(LEAFLETS_GLOBAL)
    n = 600
    d_xyz, d_box = system.frames_device(n, seed=23)
    whole = HipEngine(system.tables)
    whole.use_torch_stream()
    whole.submit_device(d_xyz, d_box, np.arange(n))
    w = whole.finish()
    assert (w.counts[0] == 256 * n).all()
    parts = []
    for a, b in ((0, 173), (173, n)):
        e = HipEngine(system.tables)
        e.use_torch_stream()
        e.submit_device(d_xyz[a:b], d_box[a:b], np.arange(a, b))
        parts.append(e.finish())
    np.testing.assert_array_equal(parts[0].sums + parts[1].sums, w.sums)
    np.testing.assert_array_equal(parts[0].map_sums + parts[1].map_sums, w.map_sums)
    np.testing.assert_array_equal(parts[0].map_counts + parts[1].map_counts, w.map_counts)
    np.testing.assert_array_equal(w.map_counts[0], w.map_counts[1] + w.map_counts[2])
    np.testing.assert_array_equal(w.map_sums[0], w.map_sums[1] + w.map_sums[2])
    assert w.map_counts[0].sum() > 0.97 * w.counts[0].sum()
    del torch
