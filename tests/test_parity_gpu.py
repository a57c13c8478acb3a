"""GPU parity: the HIP path (through the C ABI) against the oracle on the same seeded inputs.

Bar: sample counts and i64 order sums EQUAL to the oracle mode that restates the device's cosine
evaluation (DIRECT for the default, MIRROR for GORDER_FLAG_TRIG_ACOS_COS); every order parameter
within 1e-6 (north_star tolerance) of the oracle's LIBM mode, which evaluates acos/cos with the host
libm exactly as the Rust reference does.
"""
import numpy as np
import pytest

from gorder_amd import HipEngine, abi, synthetic
from gorder_amd.abi import (LEAFLETS_GLOBAL, LEAFLETS_INDIVIDUAL, LEAFLETS_LOCAL, LEAFLETS_MANUAL,
                            LEAFLETS_NONE, MolType, Tables)
from oracle import oracle
from leaflet_check import assert_sums_given_device_flags

pytestmark = pytest.mark.gpu
TOL = 1e-6   # BASELINE.json north_star: "every per-bond order parameter within 1e-6 of the reference"


def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def run_gpu(system, xyz, box, frame_index=None, host=False, batches=1):
    torch = torch_cuda()
    eng = HipEngine(system.tables)
    eng.use_torch_stream()
    n = xyz.shape[0]
    edges = np.linspace(0, n, batches + 1).astype(int)
    keep = []
    for a, b in zip(edges[:-1], edges[1:]):
        if a == b:
            continue
        fi = None if frame_index is None else frame_index[a:b]
        if fi is None:
            fi = np.arange(a, b)
        if host:
            eng.submit_host(xyz[a:b], None if box is None else box[a:b], fi)
        else:
            dx = torch.from_numpy(xyz[a:b]).cuda()
            db = None if box is None else torch.from_numpy(box[a:b]).cuda()
            keep += [dx, db]
            eng.submit_device(dx, db, fi)
    res = eng.finish()
    return eng, res


import platform   # noqa: E402
LIBM_IS_GLIBC = platform.libc_ver()[0] == "glibc" and "2.28" <= platform.libc_ver()[1] <= "2.40"


def run_oracle(system, xyz, box, frame_index=None, trig=None, n_threads=1):
    if trig is None:   # the oracle mode that restates the device library's cosine evaluation
        trig = oracle.TRIG_MIRROR if (system.tables.flags & abi.FLAG_TRIG_ACOS_COS) else oracle.TRIG_DIRECT
    o = oracle.OracleEngine(system.tables, trig=trig, n_threads=n_threads)
    o.submit(xyz, box, frame_index)
    return o, o.finish()


def assert_within_tolerance(got, ref):
    """|dS| <= 1e-6 for every accumulator.  Compared on the integer means (ticks of 1e-6, order.rs:34-42):
    the f32 the reference finally reports quantises at 1e-6 itself, so 1 tick IS the stated tolerance."""
    a, b = got.order_ticks(), ref.order_ticks()
    nan = np.iinfo(np.int64).min
    assert np.array_equal(a == nan, b == nan)
    ok = a != nan
    if ok.any():
        assert np.abs(a[ok] - b[ok]).max() <= round(TOL * 1e6)
    fa, fb = got.order(), ref.order()
    assert np.array_equal(np.isnan(fa), np.isnan(fb))


def assert_parity(system, xyz, box, frame_index=None, **kw):
    eng, got = run_gpu(system, xyz, box, frame_index, **kw)
    _, want = run_oracle(system, xyz, box, frame_index)
    assert got.n_frames == want.n_frames == xyz.shape[0]
    np.testing.assert_array_equal(got.counts, want.counts)
    np.testing.assert_array_equal(got.sums, want.sums)
    _, libm = run_oracle(system, xyz, box, frame_index, trig=oracle.TRIG_LIBM)
    np.testing.assert_array_equal(got.counts, libm.counts)
    assert_within_tolerance(got, libm)
    if (system.tables.flags & abi.FLAG_TRIG_ACOS_COS) and LIBM_IS_GLIBC and system.tables.leaflets.method in (0, 4):
        # the literal mode computes acos and cos as glibc does: with this libm the device's sums ARE the reference-faithful
        # oracle's, bit for bit (no leaflets or a manual assignment: nothing else could differ)
        np.testing.assert_array_equal(got.sums, libm.sums)
    return eng, got


@pytest.mark.parametrize("flags", [0, abi.FLAG_TRIG_ACOS_COS])
@pytest.mark.parametrize("n_lipids,n_frames", [(256, 12), (7, 9), (64, 1), (31, 50)])
def test_aa_no_leaflets(built, n_lipids, n_frames, flags):
    system = synthetic.aa_membrane(n_lipids)
    system.tables.flags = flags
    xyz = system.frames(n_frames, seed=11)
    assert_parity(system, xyz, system.box9(n_frames))


@pytest.mark.parametrize("flags", [0, abi.FLAG_TRIG_ACOS_COS])
def test_cg_global_leaflets_mixed_types(built, flags):
    system = synthetic.cg_membrane(600, leaflets=LEAFLETS_GLOBAL, n_types=3)
    system.tables.flags = flags
    n = 20
    xyz = system.frames(n, seed=5)
    eng, got = assert_parity(system, xyz, system.box9(n))
    # both leaflets populated, and upper + lower == total
    assert got.counts[1].sum() > 0 and got.counts[2].sum() > 0
    np.testing.assert_array_equal(got.counts[1] + got.counts[2], got.counts[0])
    np.testing.assert_array_equal(got.sums[1] + got.sums[2], got.sums[0])
    o, _ = run_oracle(system, xyz, system.box9(n))
    flags, fr = eng.leaflets()
    oflags, odist, ofr = o.leaflets()
    assert fr == ofr == n - 1
    np.testing.assert_array_equal(flags, oflags)
    np.testing.assert_allclose(eng.leaflet_distances(), odist, atol=2e-5)


@pytest.mark.parametrize("cell_list", ["one kernel", "three kernels", "atoms only"])
@pytest.mark.parametrize("pbc", [True, False])
@pytest.mark.parametrize("radius,n_lipids", [(2.5, 800), (1.2, 300), (30.0, 120), (6.0, 1000)])
def test_local_leaflets(built, monkeypatch, radius, n_lipids, pbc, cell_list):
    """Local classification (leaflets.rs:661-675 + pbc.rs:273-318): cylinder membership is restated with
    identical f32 operations, so flags must agree unless a head sits within 1e-4 nm of its local centre.
    Every way the device builds and walks the cell list: k_local_build (default), bin / scan / scatter (what membranes
    beyond 65 536 atoms take), and without the per-cell sums (every candidate atom by atom)."""
    if cell_list == "three kernels":
        monkeypatch.setenv("GORDER_HIP_LOCAL_THREE_KERNELS", "1")
    if cell_list == "atoms only":
        monkeypatch.setenv("GORDER_HIP_LOCAL_ATOMS_ONLY", "1")
    system = synthetic.cg_membrane(n_lipids, leaflets=LEAFLETS_LOCAL, radius=radius, n_types=2, handle_pbc=pbc)
    n = 7
    xyz = system.frames(n, seed=17)
    if not pbc:
        xyz = np.ascontiguousarray(xyz)
    box = system.box9(n) if pbc else None
    eng, got = run_gpu(system, xyz, box, batches=2)
    o, want = run_oracle(system, xyz, box)
    flags, fr = eng.leaflets()
    oflags, odist, ofr = o.leaflets()
    assert fr == ofr
    diff = flags != oflags
    assert not diff.any() or np.abs(odist[diff]).max() < 1e-4
    np.testing.assert_allclose(eng.leaflet_distances()[~diff], odist[~diff], atol=5e-5)
    assert 0 < flags.sum() < len(flags)
    # sums are compared whatever the flags did: EQUAL to the oracle fed with the device's own assignment, and the
    # assignments differ in at most a few lipids that sit on the mid-plane
    assert_sums_given_device_flags(system.tables, xyz, box, got)
    if not diff.any():
        np.testing.assert_array_equal(got.sums, want.sums)
        np.testing.assert_array_equal(got.counts, want.counts)


@pytest.mark.parametrize("frequency", [0, 1, 5])
def test_leaflet_frequency_and_batching(built, frequency):
    # Frequency::Once / Every(n): non-assignment frames reuse the assignment of floor(g/n)*n
    # (leaflets.rs:1437-1472), also across submit() batches
    system = synthetic.cg_membrane(128, leaflets=LEAFLETS_INDIVIDUAL, frequency=frequency)
    n = 23
    xyz = system.frames(n, seed=9)
    # make lipids flip over time so that the assignment frame matters
    xyz[7:, : 12 * 20, 2] = system.box[2] - xyz[7:, : 12 * 20, 2]
    assert_parity(system, xyz, system.box9(n), batches=4)


def test_flip_and_manual(built):
    system = synthetic.cg_membrane(64, leaflets=LEAFLETS_GLOBAL, flip=True)
    xyz = system.frames(6, seed=2)
    _, flipped = assert_parity(system, xyz, system.box9(6))
    system2 = synthetic.cg_membrane(64, leaflets=LEAFLETS_GLOBAL, flip=False)
    _, plain = assert_parity(system2, xyz, system2.box9(6))
    np.testing.assert_array_equal(flipped.sums[1], plain.sums[2])
    np.testing.assert_array_equal(flipped.counts[2], plain.counts[1])
    # manual flags reproduce the global assignment when fed the same flags
    torch = torch_cuda()
    system3 = synthetic.cg_membrane(64, leaflets=LEAFLETS_MANUAL)
    system3.tables.molecule_types[0].heads = None
    eng = HipEngine(system3.tables)
    flags = (np.arange(64) % 2).astype(np.uint8)
    eng.set_manual_leaflets(flags)
    eng.submit_device(torch.from_numpy(xyz).cuda(), torch.from_numpy(system3.box9(6)).cuda())
    got = eng.finish()
    o = oracle.OracleEngine(system3.tables, trig=oracle.TRIG_DIRECT)
    o.set_manual_leaflets(flags)
    o.submit(xyz, system3.box9(6))
    want = o.finish()
    np.testing.assert_array_equal(got.sums, want.sums)
    np.testing.assert_array_equal(got.counts, want.counts)


def test_no_pbc_and_tilted_normal(built):
    system = synthetic.cg_membrane(100, handle_pbc=False, normal=(0.3, -0.2, 0.9))
    xyz = system.frames(5, seed=4)
    assert_parity(system, xyz, None)
    system = synthetic.cg_membrane(100, handle_pbc=True, normal=(1.0, 0.0, 0.0))
    assert_parity(system, xyz, system.box9(5))


def test_host_submit_matches_device_submit(built):
    system = synthetic.aa_membrane(32)
    xyz = system.frames(10, seed=8)
    _, a = run_gpu(system, xyz, system.box9(10), host=True, batches=3)
    _, b = run_gpu(system, xyz, system.box9(10), host=False)
    np.testing.assert_array_equal(a.sums, b.sums)
    np.testing.assert_array_equal(a.counts, b.counts)


def test_direct_kernel_equals_tiled_kernel(built, monkeypatch):
    system = synthetic.aa_membrane(40)
    xyz = system.frames(9, seed=6)
    _, tiled = run_gpu(system, xyz, system.box9(9))
    monkeypatch.setenv("GORDER_HIP_FORCE_DIRECT", "1")
    eng, direct = run_gpu(system, xyz, system.box9(9))
    assert eng.plan()["n_tiles"] == 0
    np.testing.assert_array_equal(tiled.sums, direct.sums)
    np.testing.assert_array_equal(tiled.counts, direct.counts)


def test_scattered_bonds(built):
    # bonds that cannot share an LDS window take the direct-gather kernel; mixed with tiled ones
    rng = np.random.default_rng(1)
    n_atoms, n_mol = 30000, 50
    far = np.stack([rng.integers(0, 1000, n_mol), rng.integers(20000, 30000, n_mol)], axis=1)
    near = np.stack([np.arange(5000, 5000 + n_mol), np.arange(5001, 5001 + n_mol)], axis=1)
    bonds = np.stack([np.sort(far, axis=1), near]).astype(np.uint32)
    t = Tables(n_atoms=n_atoms, molecule_types=[MolType(n_molecules=n_mol, bonds=bonds)])
    box = np.array([7.0, 8.0, 9.0], dtype=np.float32)
    system = synthetic.System("scatter", t, (rng.random((n_atoms, 3)) * box).astype(np.float32), box, 0.05)
    xyz = system.frames(6, seed=1)
    eng, _ = assert_parity(system, xyz, system.box9(6))
    assert eng.plan()["n_direct_items"] == n_mol


def test_edge_cases(built):
    torch = torch_cuda()
    system = synthetic.cg_membrane(20)
    # zero frames is a no-op
    eng = HipEngine(system.tables)
    eng.submit_host(np.zeros((0, system.n_atoms, 3), np.float32), np.zeros((0, 3, 3), np.float32))
    r = eng.finish()
    assert r.n_frames == 0 and r.counts.sum() == 0 and np.isnan(r.order()).all()
    # zero-length bond: angle() returns 0 -> S = 1 (mod.rs:78-82 with nalgebra's zero-norm rule)
    xyz = system.frames(2, seed=0)
    xyz[:, 1] = xyz[:, 0]
    assert_parity(system, xyz, system.box9(2))
    # atoms exactly half a box apart, on the box faces, outside the box by one image
    xyz = system.frames(3, seed=1)
    xyz[0, 0] = [0.0, 0.0, 0.0]
    xyz[0, 1] = [system.box[0] / 2, system.box[1] / 2, system.box[2] / 2]
    xyz[1, :24] += system.box
    xyz[2, :24] -= system.box
    assert_parity(system, xyz, system.box9(3))


def test_bond_vectors_at_the_edges_of_the_fast_division(built):
    """K1 divides vz^2 by |v|^2 with the Newton core of the IEEE division while |v|^2 lies in [2^-40, 2^40] and falls
    back to the full routine outside (gm_sch_axis): bonds of every magnitude around both guards, along and across
    the normal, tiny and denormal numerators — all bit-exact against the oracle's plain `/`.
    (Against the reference-faithful libm mode only where the squared form has the range: the default arithmetic
    squares the bond vector, so below |v| ~ 1e-19 nm or above ~ 1e19 nm — |v|^2 denormal or infinite — it is not
    the reference's acos/cos of v.n / (|v| |n|) any more; FLAG_TRIG_ACOS_COS is.)"""
    rng = np.random.default_rng(11)

    def stretch(system, xyz, lengths):
        pairs = np.asarray(system.tables.molecule_types[0].bonds).reshape(-1, 2)
        for f in range(xyz.shape[0]):
            for q, (i, j) in enumerate(pairs[rng.permutation(len(pairs))[:400]]):
                u = rng.normal(size=3)
                u /= np.linalg.norm(u)
                if q % 7 == 0:
                    u = np.array([0.0, 0.0, 1.0])                 # along the normal: quotient exactly 1
                if q % 7 == 1:
                    u = np.array([1.0, 0.0, 1e-20]); u /= np.linalg.norm(u)   # numerator far below 2^-103
                if q % 7 == 2:
                    u = np.array([0.6, 0.8, 0.0])                 # numerator 0
                xyz[f, j] = xyz[f, i] + (np.float32(lengths[q % len(lengths)]) * u).astype(np.float32)
        return xyz

    def exact(system, xyz, box):
        _, got = run_gpu(system, xyz, box)
        _, want = run_oracle(system, xyz, box)
        np.testing.assert_array_equal(got.counts, want.counts)
        np.testing.assert_array_equal(got.sums, want.sums)

    system = synthetic.aa_membrane(16)
    around_lower = [2.0 ** -21, 2.0 ** -20 * 0.999, 2.0 ** -20, 2.0 ** -20 * 1.001, 1e-5, 1e-3, 0.1]
    assert_parity(system, stretch(system, system.frames(8, seed=5), around_lower), system.box9(8))
    exact(system, stretch(system, system.frames(8, seed=5), [0.0, 1e-30, 1e-20, 1e-12] + around_lower), system.box9(8))
    nopbc = synthetic.aa_membrane(16, handle_pbc=False)       # far bonds need no periodic image
    around_upper = [2.0 ** 19, 2.0 ** 20 * 0.999, 2.0 ** 20, 2.0 ** 20 * 1.001, 1e10]
    assert_parity(nopbc, stretch(nopbc, nopbc.frames(2, seed=6), around_upper), None)
    exact(nopbc, stretch(nopbc, nopbc.frames(2, seed=6), around_upper + [1e19, 1e25]), None)


def test_division_and_sqrt_cores_equal_the_ieee_operations(built):
    """gm_div_core / gm_sqrt_core against `/` and sqrtf on 2^27 random operand sets inside the guarded ranges,
    bit for bit, on the device (gorder_hip_selftest_arithmetic)."""
    torch_cuda()
    assert abi.selftest_arithmetic(1 << 27, seed=20240213) == (0, 0)
    assert abi.selftest_arithmetic(1 << 24, seed=7) == (0, 0)


def test_device_trig_is_the_hosts_libm(built):
    """gm_acosf_t / gm_sincosf_0pi (gm_math.h) restate glibc's acosf / cosf / sinf: every 257th float of [-1, 1] and of
    [0, pi] through the device (acos also with the Newton cores of division and square root inside) against the oracle's
    restatement — and against the host's libm where that is such a glibc: bit for bit."""
    import platform
    torch_cuda()
    from test_oracle_kat import _trig_domains
    glibc = platform.libc_ver()[0] == "glibc" and "2.28" <= platform.libc_ver()[1] <= "2.40"
    stride = 257
    for fn, lo, hi in _trig_domains():
        n = (hi - lo) // stride + 1
        want = oracle.trig_batch(fn, "mirror", lo, stride, n).view(np.uint32)
        for dev_fn in ([fn] if fn != "acos" else ["acos", "acos_cores"]):
            got = abi.selftest_trig(dev_fn, lo, stride, n).view(np.uint32)
            np.testing.assert_array_equal(got, want)
        if glibc:
            np.testing.assert_array_equal(want, oracle.trig_batch(fn, "libm", lo, stride, n).view(np.uint32))
    # out of range: NaN like (x - x) / (x - x)
    assert np.isnan(abi.selftest_trig("acos", 0x3f800001, 1, 4)).all() and np.isnan(abi.selftest_trig("acos", 0x7fc00000, 1, 1)).all()


def test_errors_mirror_the_reference(built):
    torch = torch_cuda()
    system = synthetic.cg_membrane(20)
    xyz = system.frames(3, seed=0)

    def status_of(xyz, box):
        eng = HipEngine(system.tables)
        eng.submit_host(xyz, box)
        with pytest.raises(abi.GorderHipError) as e:
            eng.finish()
        return e.value

    box = system.box9(3)
    bad = box.copy(); bad[1, 0, 1] = 0.5
    assert status_of(xyz, bad).status == abi.ERR_NOT_ORTHOGONAL_BOX         # common.rs:190-192
    bad = box.copy(); bad[2] = 0.0
    assert status_of(xyz, bad).status == abi.ERR_ZERO_BOX                   # common.rs:194-196
    bad = box.copy(); bad[0] = np.nan
    assert status_of(xyz, bad).status == abi.ERR_UNDEFINED_BOX              # common.rs:187
    x2 = xyz.copy(); x2[1, 13, 0] = np.nan
    e = status_of(x2, box)
    assert e.status == abi.ERR_UNDEFINED_POSITION and e.index == 13         # bond.rs:411-417
    x2 = xyz.copy(); x2[0, 5] = 1e9
    assert status_of(x2, box).status == abi.ERR_BOX_RANGE
    # the oracle agrees on the codes
    for arr, bx, code in ((xyz, bad, abi.ERR_UNDEFINED_BOX),):
        o = oracle.OracleEngine(system.tables)
        with pytest.raises(oracle.OracleError) as oe:
            o.submit(arr, bx)
        assert oe.value.status == code
    # leaflets whose first frame is not an assignment frame need priming
    s2 = synthetic.cg_membrane(20, leaflets=LEAFLETS_GLOBAL, frequency=5)
    eng = HipEngine(s2.tables)
    with pytest.raises(abi.GorderHipError) as e:
        eng.submit_host(xyz, box, np.array([6, 7, 8]))
    assert e.value.status == abi.ERR_LEAFLETS_NOT_PRIMED


@pytest.mark.parametrize("mode", ["tiled", "direct", "timewise", "gather", "ua"])
def test_first_error_in_reference_order(built, monkeypatch, mode):
    """Several undefined atoms in one batch: the status payload is the one the reference's single-threaded walk meets
    first — lowest frame, then bond type, then molecule, first atom before second (bond.rs:406-417; united atoms:
    uaorder.rs:400-437) — whichever wave of which kernel gets there first (64-bit error key + atomicMin)."""
    torch_cuda()
    if mode == "direct":
        monkeypatch.setenv("GORDER_HIP_FORCE_DIRECT", "1")
    if mode == "gather":
        monkeypatch.setenv("GORDER_HIP_KERNEL", "gather")
    if mode == "ua":
        system = synthetic.ua_membrane(24)
        apl = 52
    else:
        system = synthetic.cg_membrane(40, n_types=2, timewise=(mode == "timewise"))
        apl = 12
    n = 9
    xyz = system.frames(n, seed=3)
    box = system.box9(n)
    rng = np.random.default_rng(5)
    for trial in range(6):
        x = xyz.copy()
        # a handful of undefined atoms: several in the first bad frame (different lipids, beads that belong to early
        # and late bond types, a lower atom index that comes LATER in the walk), more in later frames
        f0 = int(rng.integers(1, n - 2))
        lo, hi = (10, 42) if mode == "ua" else (0, apl)      # atoms that the order walk really reads
        for f in (f0, f0, f0, f0 + 1, n - 1):
            x[f, int(rng.integers(0, system.n_atoms // apl)) * apl + int(rng.integers(lo, hi)), 0] = np.nan
        if trial == 0 and mode != "ua":      # bead 11 of lipid 1 (atom 23, last bond type) against bead 0 of lipid 2 (atom 24, first)
            x = xyz.copy()
            x[4, 23, 0] = np.nan
            x[4, 24, 0] = np.nan
            x[6, 0, 0] = np.nan
        o = oracle.OracleEngine(system.tables, trig=oracle.TRIG_DIRECT)
        with pytest.raises(oracle.OracleError) as oe:
            o.submit(x, box)
        for batches in (1, 3):
            eng = HipEngine(system.tables)
            edges = np.linspace(0, n, batches + 1).astype(int)
            with pytest.raises(abi.GorderHipError) as e:
                for a, b in zip(edges[:-1], edges[1:]):
                    eng.submit_host(x[a:b], box[a:b], np.arange(a, b))
                    eng.synchronize()
            assert e.value.status == oe.value.status == abi.ERR_UNDEFINED_POSITION
            assert e.value.index == oe.value.index, (mode, trial, batches)


def test_first_error_across_batches_in_flight(built):
    """Batches queued WITHOUT a synchronisation in between (what gorder_hip_run_trajectory does): the error reported is
    the one of the earliest batch, although a later batch has one in an earlier frame of ITS batch — keys order errors
    inside a batch, k_err_commit latches the first batch that had one (common.rs:248: the first Err ends the iteration) —
    and the frame reported is the frame of the trajectory (SystemTopology::frame), not the frame in the batch."""
    torch_cuda()
    system = synthetic.cg_membrane(40, n_types=2)
    n = 12
    xyz, box = system.frames(n, seed=3), system.box9(n)
    x = xyz.copy()
    x[7, 5 * 12 + 3, 0] = np.nan          # batch 1 (frames 4..7), last frame of the batch (undefined = NaN in x)
    x[8, 2 * 12 + 0, 0] = np.nan          # batch 2 (frames 8..11), FIRST frame of its batch: a smaller key
    o = oracle.OracleEngine(system.tables, trig=oracle.TRIG_DIRECT)
    with pytest.raises(oracle.OracleError) as oe:
        o.submit(x, box, np.arange(100, 100 + 3 * n, 3))
    eng = HipEngine(system.tables)
    with pytest.raises(abi.GorderHipError) as e:
        for a in (0, 4, 8):
            eng.submit_host(x[a:a + 4], box[a:a + 4], np.arange(100 + 3 * a, 100 + 3 * (a + 4), 3))
        eng.finish()
    assert e.value.status == oe.value.status == abi.ERR_UNDEFINED_POSITION
    assert e.value.index == oe.value.index == 5 * 12 + 3
    assert e.value.frame == 100 + 3 * 7
    assert "frame 121 of the trajectory (frame 3 of batch 1)" in str(e.value)
    # an irregular list of frame indices is remembered as such
    eng = HipEngine(system.tables)
    with pytest.raises(abi.GorderHipError) as e:
        eng.submit_host(x[:4], box[:4], np.array([5, 6, 9, 40]))
        eng.submit_host(x[4:8], box[4:8], np.array([41, 50, 51, 77]))
        eng.finish()
    assert e.value.frame == 77 and e.value.index == 5 * 12 + 3
    # after gorder_hip_reset the handle is clean again
    eng.reset()
    eng.submit_host(xyz[:4], box[:4], np.arange(4))
    assert eng.finish().n_frames == 4


def test_priming_replaces_cross_thread_wait(built):
    # rank r starts at frame 6 with Every(5): it needs the assignment of frame 5 (leaflets.rs:1437-1472)
    torch = torch_cuda()
    system = synthetic.cg_membrane(64, leaflets=LEAFLETS_GLOBAL, frequency=5)
    xyz = system.frames(12, seed=3)
    xyz[6:, : 12 * 10, 2] = system.box[2] - xyz[6:, : 12 * 10, 2]
    box = system.box9(12)
    whole, ref = run_oracle(system, xyz, box)
    # shard: frames 6..11 only, primed with frame 5
    eng = HipEngine(system.tables)
    eng.prime_leaflets_device(torch.from_numpy(xyz[5]).cuda(), torch.from_numpy(box[5]).cuda(), 5)
    eng.submit_host(xyz[6:], box[6:], np.arange(6, 12))
    tail = eng.finish()
    eng0 = HipEngine(system.tables)
    eng0.submit_host(xyz[:6], box[:6], np.arange(0, 6))
    head = eng0.finish()
    np.testing.assert_array_equal(head.sums + tail.sums, ref.sums)      # SystemTopology::add
    np.testing.assert_array_equal(head.counts + tail.counts, ref.counts)


def test_frame_count_not_a_multiple_of_the_stage(built, monkeypatch):
    # 43 frames in stages of 4 with few workgroups: partial last stage, several chunks
    monkeypatch.setenv("GORDER_HIP_WG_TARGET", "40")
    system = synthetic.cg_membrane(150, leaflets=LEAFLETS_GLOBAL, frequency=3)
    xyz = system.frames(43, seed=21)
    eng, _ = assert_parity(system, xyz, system.box9(43))
    assert eng.plan()["frames_per_stage"] == 4


def test_launch_geometry_invariance(built):
    # integer accumulation => identical results however the frames are batched (tests_aa.rs:320-368)
    system = synthetic.aa_membrane(24)
    xyz = system.frames(37, seed=12)
    box = system.box9(37)
    _, a = run_gpu(system, xyz, box, batches=1)
    _, b = run_gpu(system, xyz, box, batches=5)
    _, c = run_gpu(system, xyz, box, batches=37)
    for r in (b, c):
        np.testing.assert_array_equal(a.sums, r.sums)
        np.testing.assert_array_equal(a.counts, r.counts)
    _, threaded = run_oracle(system, xyz, box, n_threads=3)
    np.testing.assert_array_equal(a.sums, threaded.sums)


def test_full_size_properties(built):
    """BASELINE config 2 at full width (256 lipids, 25 088 atoms, 16 384 bonds/frame), more frames than
    the oracle is asked to chew: size-independent properties + an oracle check on a frame subset."""
    torch = torch_cuda()
    system = synthetic.aa_membrane(256)
    n = 2000
    d_xyz, d_box = system.frames_device(n, seed=1)
    eng = HipEngine(system.tables)
    eng.use_torch_stream()
    eng.submit_device(d_xyz, d_box)
    full = eng.finish()
    assert full.n_frames == n
    assert (full.counts[0] == n * 256).all()
    # linearity over frame subsets: sum(parts) == whole
    parts = None
    for a, b in ((0, 700), (700, 701), (701, 2000)):
        e = HipEngine(system.tables)
        e.use_torch_stream()
        e.submit_device(d_xyz[a:b].contiguous(), d_box[a:b].contiguous(), np.arange(a, b))
        r = e.finish()
        parts = r.sums.copy() if parts is None else parts + r.sums
    np.testing.assert_array_equal(parts, full.sums)
    # oracle on a strided subset of the very same device frames
    sel = np.arange(0, n, 97)
    sub = d_xyz[torch.from_numpy(sel).cuda()].contiguous()
    e = HipEngine(system.tables)
    e.use_torch_stream()
    e.submit_device(sub, d_box[: len(sel)].contiguous())
    got = e.finish()
    _, want = run_oracle(system, sub.cpu().numpy(), system.box9(len(sel)))
    np.testing.assert_array_equal(got.sums, want.sums)
    np.testing.assert_array_equal(got.counts, want.counts)
    # order parameters are physical
    s = full.order()[0]
    assert np.all(s >= -0.5 - 1e-6) and np.all(s <= 1.0 + 1e-6)


@pytest.mark.parametrize("flags", [0, abi.FLAG_TRIG_ACOS_COS])
def test_reference_known_answer_on_the_device(built, flags):
    """The reference's own unit test of the sample arithmetic (test_calc_sch, mod.rs:94-105): atoms at
    (1.7, 2.1, 9.7) and (1.9, 2.4, 0.8) in a 10-nm box, normal z -> the bond crosses the box face, the
    minimum-image vector is (0.2, 0.3, 1.1) and S = 0.8544775.  One molecule, one bond, through the C ABI."""
    bonds = np.array([[[0, 1]]], dtype=np.uint32)
    tables = Tables(n_atoms=2, molecule_types=[MolType(n_molecules=1, bonds=bonds)], flags=flags)
    xyz = np.array([[[1.7, 2.1, 9.7], [1.9, 2.4, 0.8]]], dtype=np.float32)
    box = np.zeros((1, 3, 3), dtype=np.float32)
    box[0, 0, 0] = box[0, 1, 1] = box[0, 2, 2] = 10.0
    eng = HipEngine(tables)
    eng.submit_host(xyz, box, np.arange(1))
    res = eng.finish()
    assert res.counts[0, 0] == 1
    assert abs(int(res.sums[0, 0]) - 854478) <= 1          # round(0.8544775 * 1e6), within one tick
    np.testing.assert_allclose(res.order()[0, 0], 0.8544775, atol=1.5e-6)


@pytest.mark.parametrize("variant", ["contiguous", "generic", "thick", "subset", "wide"])
def test_global_leaflet_kernels(built, monkeypatch, variant):
    """Global classification has two kernels: the one-pass kernel for a membrane group that is the whole frame
    (with a second read when the membrane is thicker than half the box) and the generic one (index list)."""
    box = None
    if variant == "generic":
        monkeypatch.setenv("GORDER_HIP_LEAFLETS_GENERIC", "1")
    if variant == "wide":
        box = (120.0, 110.0, 10.0)     # in-plane coordinates many box heights away from 0 (the 1M-bead system's shape)
    if variant == "thick":
        box = (8.0, 8.0, 6.2)          # leaflet heads +-2 nm from the mid-plane: 4 nm of membrane in a 6.2-nm box
    system = synthetic.cg_membrane(100, leaflets=LEAFLETS_GLOBAL, n_types=2, box=box)
    if variant == "subset":            # membrane group = every second atom: only the generic kernel can take it
        system.tables.leaflets.membrane = np.arange(0, system.n_atoms, 2, dtype=np.uint32)
    n = 9
    xyz = system.frames(n, seed=23)
    if variant == "thick":             # ... and shifted so that it straddles the periodic boundary
        xyz[:, :, 2] = np.mod(xyz[:, :, 2] + 3.0, 6.2).astype(np.float32)
    eng, got = run_gpu(system, xyz, system.box9(n), batches=2)
    o, want = run_oracle(system, xyz, system.box9(n))
    flags, fr = eng.leaflets()
    oflags, odist, ofr = o.leaflets()
    assert fr == ofr == n - 1
    diff = flags != oflags
    assert not diff.any() or np.abs(odist[diff]).max() < 1e-4
    np.testing.assert_allclose(eng.leaflet_distances()[~diff], odist[~diff], atol=5e-5)
    assert 0 < flags.sum() < len(flags)
    assert_sums_given_device_flags(system.tables, xyz, system.box9(n), got)
    if not diff.any():
        np.testing.assert_array_equal(got.sums, want.sums)
        np.testing.assert_array_equal(got.counts, want.counts)


def test_local_leaflets_membrane_thicker_than_half_the_box(built):
    """The local-centre kernel takes its refinement sum in the first pass when every member's image around the
    head is also its image around the estimate; a 4-nm membrane in a 6.2-nm box, straddling the periodic
    boundary, needs the second pass instead."""
    system = synthetic.cg_membrane(160, leaflets=LEAFLETS_LOCAL, radius=2.0, n_types=2, box=(9.0, 9.0, 6.2))
    n = 6
    xyz = system.frames(n, seed=29)
    xyz[:, :, 2] = np.mod(xyz[:, :, 2] + 3.0, 6.2).astype(np.float32)
    eng, got = run_gpu(system, xyz, system.box9(n), batches=2)
    o, want = run_oracle(system, xyz, system.box9(n))
    flags, _ = eng.leaflets()
    oflags, odist, _ = o.leaflets()
    diff = flags != oflags
    assert not diff.any() or np.abs(odist[diff]).max() < 1e-4
    np.testing.assert_allclose(eng.leaflet_distances()[~diff], odist[~diff], atol=5e-5)
    assert 0 < flags.sum() < len(flags)
    assert_sums_given_device_flags(system.tables, xyz, system.box9(n), got)
    if not diff.any():
        np.testing.assert_array_equal(got.sums, want.sums)


@pytest.mark.parametrize("box_z,amplitude", [(7.0, 0.0), (8.4, 0.6), (10.0, 1.2), (14.0, 2.5), (9.0, 2.0)])
def test_local_leaflets_of_an_undulating_membrane(built, box_z, amplitude):
    """k_local_flags_rows takes the circular mean of the cells wholly inside a head's cylinder as a stand-in for the
    reference's estimate and bounds the distance between the two (kernels_leaflets.h); whether the bound holds (a thin
    water layer or a strongly bent membrane: it does not, and the general passes decide) must not show in the flags."""
    system = synthetic.cg_membrane(400, leaflets=LEAFLETS_LOCAL, radius=2.0, n_types=2, box=(16.0, 16.0, box_z))
    n = 5
    xyz = system.frames(n, seed=31)
    wave = amplitude * np.sin(2 * np.pi * xyz[:, :, 0] / 16.0) * np.cos(2 * np.pi * xyz[:, :, 1] / 16.0)
    xyz[:, :, 2] = (xyz[:, :, 2] + wave + 0.37).astype(np.float32)        # (not wrapped: some atoms may leave the box)
    eng, got = run_gpu(system, xyz, system.box9(n), batches=2)
    o, want = run_oracle(system, xyz, system.box9(n))
    flags, _ = eng.leaflets()
    oflags, odist, _ = o.leaflets()
    diff = flags != oflags
    assert not diff.any() or np.abs(odist[diff]).max() < 1e-4
    np.testing.assert_allclose(eng.leaflet_distances()[~diff], odist[~diff], atol=5e-5)
    assert 0 < flags.sum() < len(flags)
    assert_sums_given_device_flags(system.tables, xyz, system.box9(n), got)
    if not diff.any():
        np.testing.assert_array_equal(got.sums, want.sums)
        np.testing.assert_array_equal(got.counts, want.counts)


def test_local_leaflets_with_a_cell_of_more_than_255_atoms(built):
    """k_local_build keeps an atom's place inside its cell in a byte; a cell that holds more atoms than a byte counts
    makes the whole frame take its places from counters instead (`use_fill`).  340 atoms of the membrane stacked on one
    spot of the plane (their normal coordinates left alone) — and the flags are the oracle's."""
    system = synthetic.cg_membrane(160, leaflets=LEAFLETS_LOCAL, radius=2.0, n_types=2, box=(11.0, 11.0, 10.0))
    n = 3
    xyz = system.frames(n, seed=37)
    rng = np.random.default_rng(5)
    crowd = rng.choice(xyz.shape[1], 340, replace=False)
    xyz[1:, crowd, 0] = (4.11 + rng.normal(0, 0.01, (n - 1, 340))).astype(np.float32)       # (frame 0 stays ordinary)
    xyz[1:, crowd, 1] = (6.52 + rng.normal(0, 0.01, (n - 1, 340))).astype(np.float32)
    eng, got = run_gpu(system, xyz, system.box9(n), batches=1)
    o, want = run_oracle(system, xyz, system.box9(n))
    flags, _ = eng.leaflets()
    oflags, odist, _ = o.leaflets()
    diff = flags != oflags
    assert not diff.any() or np.abs(odist[diff]).max() < 1e-4
    assert_sums_given_device_flags(system.tables, xyz, system.box9(n), got)
    if not diff.any():
        np.testing.assert_array_equal(got.sums, want.sums)
        np.testing.assert_array_equal(got.counts, want.counts)


def test_local_leaflets_on_a_grid_that_fills_the_cell_table(built):
    """Cells of r / 7 with a halo of 2 x 7 columns: 128 rows of 114 + 14 cells are exactly the 16 384 entries the cell
    tables hold, and the end entry of the starts (index 16 384) lies behind the last thread's share of the scan in
    k_local_build — it must be written all the same (the rows kernel reads it for the last cell of the last row)."""
    radius = 1.0
    box = (18.3595, 16.3595, 10.0)

    def cells(L, k=7):                      # kernels_leaflets.h local_axis, in f32
        return int(np.floor(np.float32(L) / (np.float32(radius) / np.float32(k)) * np.float32(0.9999)))
    assert cells(box[0]) == 128 and cells(box[1]) == 114
    system = synthetic.cg_membrane(900, leaflets=LEAFLETS_LOCAL, radius=radius, n_types=2, box=box)
    n = 3
    xyz = system.frames(n, seed=43)
    # lipids in the far corner of the plane: their cylinders cover the last cells of the last rows
    assert ((xyz[0, :, 0] > box[0] - 0.1) & (xyz[0, :, 1] > box[1] - 0.1 - 14 * 0.1431)).any()
    eng, got = run_gpu(system, xyz, system.box9(n), batches=1)
    o, want = run_oracle(system, xyz, system.box9(n))
    flags, _ = eng.leaflets()
    oflags, odist, _ = o.leaflets()
    diff = flags != oflags
    assert not diff.any() or np.abs(odist[diff]).max() < 1e-4
    np.testing.assert_allclose(eng.leaflet_distances()[~diff], odist[~diff], atol=5e-5)
    assert_sums_given_device_flags(system.tables, xyz, system.box9(n), got)
    if not diff.any():
        np.testing.assert_array_equal(got.sums, want.sums)
        np.testing.assert_array_equal(got.counts, want.counts)


def test_independent_handles_interleaved_and_threaded(built):
    """A handle is thread-compatible like one SystemTopology clone (topology/mod.rs:256-278): several handles, each
    with its own stream, fed from different host threads at the same time, do not disturb each other."""
    import threading
    systems = [synthetic.cg_membrane(80 + 40 * k, leaflets=LEAFLETS_GLOBAL if k % 2 else LEAFLETS_NONE, n_types=1 + k % 3)
               for k in range(4)]
    data = [(s.frames(12, seed=50 + k), s.box9(12)) for k, s in enumerate(systems)]
    results = [None] * len(systems)

    def work(k):
        eng = HipEngine(systems[k].tables)
        xyz, box = data[k]
        for a in range(0, 12, 3):
            eng.submit_host(xyz[a:a + 3], box[a:a + 3], np.arange(a, a + 3))
        results[k] = eng.finish()

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(systems))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k, s in enumerate(systems):
        _, want = run_oracle(s, data[k][0], data[k][1])
        np.testing.assert_array_equal(results[k].sums, want.sums)
        np.testing.assert_array_equal(results[k].counts, want.counts)


@pytest.mark.parametrize("method", [LEAFLETS_GLOBAL, LEAFLETS_LOCAL, LEAFLETS_INDIVIDUAL])
@pytest.mark.parametrize("frequency", [4, 0])
def test_primed_shards_equal_one_handle(built, method, frequency):
    """Frame-sharded ranks with a leaflet frequency other than every frame: a shard that starts between
    two assignment frames is primed ON THE DEVICE with the one frame it depends on (gorder_hip_prime_leaflets:
    carry row, flag buffer (re)allocation, the Local method's slab path) — shards cut at non-assignment frames,
    added up, equal the single-handle run bit for bit, flags of the last assignment included."""
    torch = torch_cuda()
    system = synthetic.cg_membrane(96, leaflets=method, frequency=frequency, radius=2.0, n_types=2)
    n = 22
    xyz = system.frames(n, seed=31)
    for f in range(n):                       # lipids change sides over time: which assignment a frame uses matters
        k = 12 * (3 * f)
        xyz[f:, k:k + 36, 2] = system.box[2] - xyz[f:, k:k + 36, 2]
    box = system.box9(n)
    d_xyz, d_box = torch.from_numpy(xyz).cuda(), torch.from_numpy(box).cuda()
    whole = HipEngine(system.tables)
    whole.submit_device(d_xyz, d_box)
    want = whole.finish()
    wflags, wframe = whole.leaflets()
    assert want.counts[1].sum() > 0 and want.counts[2].sum() > 0
    for cuts in ([0, 10, n], [0, 5, 13, n], [0, 1, 2, 3, n]):
        sums, counts, last = 0, 0, None
        for a, b in zip(cuts[:-1], cuts[1:]):
            eng = HipEngine(system.tables)
            assign = 0 if frequency == 0 else (a // frequency) * frequency      # floor(frame / n) * n, leaflets.rs:1437-1472
            if assign != a:
                eng.prime_leaflets_device(d_xyz[assign], d_box[assign], assign)
                f0, fr = eng.leaflets()
                assert fr == assign
            eng.submit_device(d_xyz[a:b].contiguous(), d_box[a:b].contiguous(), np.arange(a, b))
            r = eng.finish()
            assert r.n_frames == b - a
            sums, counts, last = sums + r.sums, counts + r.counts, eng
        np.testing.assert_array_equal(sums, want.sums)
        np.testing.assert_array_equal(counts, want.counts)
        lflags, lframe = last.leaflets()
        assert lframe == wframe
        np.testing.assert_array_equal(lflags, wflags)
    # the priming frame goes through the same checks as an analysed frame
    eng = HipEngine(system.tables)
    with pytest.raises(abi.GorderHipError) as e:
        eng.prime_leaflets_device(d_xyz[0], None, 0)
    assert e.value.status == abi.ERR_INVALID_ARGUMENT
    bad = d_box[0].clone()
    bad[0, 1] = 0.3
    eng.prime_leaflets_device(d_xyz[0], bad, 0)
    with pytest.raises(abi.GorderHipError) as e:
        eng.synchronize()
    assert e.value.status == abi.ERR_NOT_ORTHOGONAL_BOX


def test_library_allreduce_and_reset(built):
    """gorder_hip_allreduce: RCCL called from inside the library over the packed accumulators and the ordermaps — with
    a single-rank communicator (all one GPU can host) the reduce is the identity, which pins the plumbing: RCCL binding,
    communicator helpers, in-place ncclInt64 sums on the handle's stream, maps included.  gorder_hip_reset then gives a
    fresh SystemTopology: the same batch again reproduces the same results, not twice the sums."""
    torch = torch_cuda()
    om = abi.OrderMap(enabled=True, plane=0, span_x=(0.0, 6.0), span_y=(0.0, 6.0), bin=(0.5, 0.5))
    system = synthetic.cg_membrane(90, leaflets=LEAFLETS_GLOBAL, n_types=2, ordermap=om, timewise=True)
    n = 11
    xyz = system.frames(n, seed=12)
    box = system.box9(n)
    plain = HipEngine(system.tables)
    plain.submit_host(xyz, box)
    want = plain.finish()
    eng = HipEngine(system.tables)
    uid = HipEngine.comm_unique_id()
    assert len(uid) == 128
    comm = eng.comm_create(uid, 1, 0)
    eng.submit_host(xyz, box)
    eng.allreduce(comm)
    got = eng.finish()
    for a, b in ((got.sums, want.sums), (got.counts, want.counts), (got.map_sums, want.map_sums),
                 (got.map_counts, want.map_counts)):
        np.testing.assert_array_equal(a, b)
    assert got.n_frames == n
    eng.reset()
    assert eng.finish().n_frames == 0 and eng.finish().counts.sum() == 0
    eng.submit_host(xyz, box)
    eng.allreduce(comm)
    again = eng.finish()
    np.testing.assert_array_equal(again.sums, want.sums)
    np.testing.assert_array_equal(again.map_counts, want.map_counts)
    tw_s, tw_c = eng.timewise(n)
    pw_s, pw_c = plain.timewise(n)
    np.testing.assert_array_equal(tw_s, pw_s)
    np.testing.assert_array_equal(tw_c, pw_c)
    eng.comm_destroy(comm)


@pytest.mark.parametrize("method", ["global", "local", "individual"])
def test_reference_leaflet_kat_on_the_device(built, method):
    """The reference's own leaflet unit tests through the C ABI: atoms 1385 / 11885 of the all-atom membrane are
    Upper / Lower under every classifier (leaflets.rs:1858-1960); distances equal to the oracle's."""
    torch_cuda()
    from test_oracle_kat import _pcpepg_frame, leaflet_kat_tables
    fx, xyz, box = _pcpepg_frame()
    tables = leaflet_kat_tables(fx, {"global": LEAFLETS_GLOBAL, "local": LEAFLETS_LOCAL, "individual": LEAFLETS_INDIVIDUAL}[method])
    eng = HipEngine(tables)
    eng.submit_host(xyz, box, np.arange(1))
    res = eng.finish()
    flags, frame = eng.leaflets()
    assert flags.tolist() == [0, 1] and frame == 0
    assert res.counts[1, 0] == 1 and res.counts[2, 0] == 1
    o = oracle.OracleEngine(tables, trig=oracle.TRIG_DIRECT)
    o.submit(xyz, box, [0])
    _, odist, _ = o.leaflets()
    np.testing.assert_allclose(eng.leaflet_distances(), odist, atol=5e-5)
    np.testing.assert_array_equal(res.sums, o.finish().sums)


def test_global_leaflets_with_unwrapped_coordinates(built):
    """Molecules that have drifted whole box lengths away ('nojump' trajectories): the minimum-image loops of the centre
    and of the head distances take several steps — same flags and sums as the oracle."""
    system = synthetic.cg_membrane(120, leaflets=LEAFLETS_GLOBAL, n_types=2)
    n = 21
    xyz = system.frames(n, seed=3)
    xyz[3:, 12 * 7:12 * 9, 2] += 2 * system.box[2]          # two lipids two boxes up, from frame 3 on
    xyz[10:, 12 * 40:12 * 41, 2] -= 3 * system.box[2]       # one lipid three boxes down
    xyz[:, 12 * 60:12 * 62, 0] += 4 * system.box[0]         # in-plane shifts do not matter to the centre
    box = system.box9(n)
    _, got = run_gpu(system, xyz, box, batches=2)
    assert got.counts[1].sum() > 0 and got.counts[2].sum() > 0
    assert_sums_given_device_flags(system.tables, xyz, box, got)


def test_wide_windows_fit_the_default_lds_budget(built):
    """An atom window of ~860 atoms staged for 4 frames takes 41 KB of LDS per workgroup: inside the 64 KB a launch
    gets without opting in (the widest window, 1024 atoms, takes 48 KB)."""
    rng = np.random.default_rng(4)
    n_atoms, n_mol = 3000, 60
    wide = np.stack([np.arange(n_mol), np.arange(n_mol) + 800], axis=1)
    near = np.stack([np.arange(1500, 1500 + n_mol), np.arange(1501, 1501 + n_mol)], axis=1)
    t = Tables(n_atoms=n_atoms, molecule_types=[MolType(n_molecules=n_mol, bonds=np.stack([wide, near]).astype(np.uint32))])
    box = np.array([7.0, 8.0, 9.0], dtype=np.float32)
    system = synthetic.System("wide", t, (rng.random((n_atoms, 3)) * box).astype(np.float32), box, 0.05)
    xyz = system.frames(19, seed=2)
    eng, _ = assert_parity(system, xyz, system.box9(19))
    plan = eng.plan()
    assert plan["max_window_atoms"] > 700 and plan["n_direct_items"] == 0
    assert plan["frames_per_stage"] == 4 and plan["lds_bytes"] <= 64 * 1024


# ---- k_local_flags_rows decides most heads from a bound on their ring (round 4): the decision and the exact path agree ----
def _sums_equal(a, b):
    np.testing.assert_array_equal(a.sums, b.sums)
    np.testing.assert_array_equal(a.counts, b.counts)


@pytest.mark.parametrize("shape", ["flat", "undulating", "tight box"])
def test_local_leaflets_decided_from_the_bound_are_the_exact_paths(built, monkeypatch, shape):
    """The rows kernel skips a head's ring when prefix sums alone bound the members' mean normal coordinate away from the
    head (kernels_leaflets.h, "a head its ring cannot change"); GORDER_HIP_LOCAL_NO_PRUNE=1 sends every head through the
    lists, the ring loop and the centre.  Upper / lower sums of every frame — all frames but a batch's last take the bound —
    must be EQUAL between the two, and to the oracle's."""
    box = {"flat": None, "undulating": (16.0, 16.0, 12.0), "tight box": (14.0, 14.0, 7.4)}[shape]
    system = synthetic.cg_membrane(900 if shape == "flat" else 500, leaflets=LEAFLETS_LOCAL, radius=2.5, n_types=2, box=box)
    n = 12
    xyz = system.frames(n, seed=53)
    if shape == "undulating":
        wave = 1.5 * np.sin(2 * np.pi * xyz[:, :, 0] / 16.0) * np.cos(2 * np.pi * xyz[:, :, 1] / 16.0)
        xyz[:, :, 2] = (xyz[:, :, 2] + wave).astype(np.float32)
    _, pruned = run_gpu(system, xyz, system.box9(n), batches=2)
    monkeypatch.setenv("GORDER_HIP_LOCAL_NO_PRUNE", "1")
    _, exact = run_gpu(system, xyz, system.box9(n), batches=2)
    monkeypatch.delenv("GORDER_HIP_LOCAL_NO_PRUNE")
    _sums_equal(pruned, exact)
    assert pruned.counts[1].sum() > 0 and pruned.counts[2].sum() > 0
    _, want = run_oracle(system, xyz, system.box9(n))
    _sums_equal(pruned, want)


def test_local_leaflets_with_heads_on_the_local_mid_plane(built, monkeypatch):
    """Heads a millimicron from the centre of their own neighbourhood: the bound cannot decide them (its slack is a few
    thousandths of a nanometre), their waves take the exact path while the others are decided from the sums — and every
    flag is still the oracle's."""
    system = synthetic.cg_membrane(800, leaflets=LEAFLETS_LOCAL, radius=2.5, n_types=2)
    n = 6
    xyz = system.frames(n, seed=59).astype(np.float32)
    box = np.asarray(system.box, dtype=np.float64)
    heads = np.concatenate([np.asarray(m.heads) for m in system.tables.molecule_types])
    rng = np.random.default_rng(61)
    moved = rng.choice(heads, 40, replace=False)
    offs = rng.uniform(4e-4, 2e-3, len(moved)) * rng.choice([-1.0, 1.0], len(moved))
    for f in range(n):
        for _ in range(4):                      # a head moves its own neighbourhood's mean a little: iterate
            p = xyz[f].astype(np.float64)
            for h, off in zip(moved, offs):
                d = p[:, :2] - p[h, :2]
                d -= box[:2] * np.round(d / box[:2])
                members = (d ** 2).sum(1) < 2.5 ** 2
                xyz[f, h, 2] = np.float32(p[members, 2].mean() + off)
    eng, got = run_gpu(system, xyz, system.box9(n), batches=2)
    o, want = run_oracle(system, xyz, system.box9(n))
    flags, _ = eng.leaflets()
    oflags, odist, _ = o.leaflets()
    diff = flags != oflags
    assert not diff.any() or np.abs(odist[diff]).max() < 1e-4
    assert_sums_given_device_flags(system.tables, xyz, system.box9(n), got, max_flag_diffs=8)
    monkeypatch.setenv("GORDER_HIP_LOCAL_NO_PRUNE", "1")
    _, exact = run_gpu(system, xyz, system.box9(n), batches=2)
    _sums_equal(got, exact)


def test_local_leaflets_with_spans_too_long_for_the_16_bit_sums(built, monkeypatch):
    """The edge entries keep the cos / sin sums in 16 bits (differences exact below 2048 records a span): 2 600 atoms on one
    spot make spans longer than that, the heads that see them are not decided from the bound, and nothing shows."""
    system = synthetic.cg_membrane(600, leaflets=LEAFLETS_LOCAL, radius=2.0, n_types=2, box=(16.0, 16.0, 10.0))
    n = 4
    xyz = system.frames(n, seed=67)
    rng = np.random.default_rng(7)
    crowd = rng.choice(xyz.shape[1], 2600, replace=False)
    xyz[:, crowd, 0] = (5.03 + rng.normal(0, 0.01, (n, 2600))).astype(np.float32)
    xyz[:, crowd, 1] = (9.41 + rng.normal(0, 0.01, (n, 2600))).astype(np.float32)
    _, got = run_gpu(system, xyz, system.box9(n), batches=1)
    monkeypatch.setenv("GORDER_HIP_LOCAL_NO_PRUNE", "1")
    _, exact = run_gpu(system, xyz, system.box9(n), batches=1)
    monkeypatch.delenv("GORDER_HIP_LOCAL_NO_PRUNE")
    _sums_equal(got, exact)
    assert_sums_given_device_flags(system.tables, xyz, system.box9(n), got, max_flag_diffs=8)


# ---- k_local_decide: the bound on its own, a lane per head, before the rows kernel (round 4) ----
def _submit_in_batches(system, xyz, box, batches):
    torch = torch_cuda()
    eng = HipEngine(system.tables)
    eng.use_torch_stream()
    edges = np.linspace(0, xyz.shape[0], batches + 1).astype(int)
    keep = []
    for a, b in zip(edges[:-1], edges[1:]):
        dx, db = torch.from_numpy(xyz[a:b]).cuda(), torch.from_numpy(box[a:b]).cuda()
        keep += [dx, db]
        eng.submit_device(dx, db, np.arange(a, b))
        eng.synchronize()                   # (the report of a submit is read at the next one if it has arrived)
    return eng, eng.finish()


def test_local_decide_runs_every_submit_on_a_flat_membrane(built):
    """Every frame but a submit's last (whose distances are wanted) is decided by k_local_decide; it keeps running."""
    system = synthetic.cg_membrane(900, leaflets=LEAFLETS_LOCAL, radius=2.5, n_types=2)
    n = 24
    xyz = system.frames(n, seed=71)
    eng, got = _submit_in_batches(system, xyz, system.box9(n), 4)
    st = eng.local_decide_stats()
    assert st["submits"] == 4 and st["paused"] == 0
    assert st["frames"] == 6 and st["open_frames"] == 1
    _, want = run_oracle(system, xyz, system.box9(n))
    _sums_equal(got, want)


def test_local_decide_pauses_when_it_leaves_the_frames_open(built, monkeypatch):
    """GORDER_HIP_LOCAL_DECIDE_NOTHING=1 makes the kernel leave every head open — what a strongly undulating membrane does —:
    the first report sends the following submits down the rows kernel alone, and the sums stay the oracle's throughout."""
    system = synthetic.cg_membrane(700, leaflets=LEAFLETS_LOCAL, radius=2.5, n_types=2)
    n = 20
    xyz = system.frames(n, seed=73)
    monkeypatch.setenv("GORDER_HIP_LOCAL_DECIDE_NOTHING", "1")
    eng, got = _submit_in_batches(system, xyz, system.box9(n), 5)
    monkeypatch.delenv("GORDER_HIP_LOCAL_DECIDE_NOTHING")
    st = eng.local_decide_stats()
    assert st["submits"] == 1 and st["paused"] == 4
    assert st["open_frames"] == st["frames"] == 4
    eng.reset()                                 # a new run tries the bound again
    assert eng.local_decide_stats() == {"submits": 0, "paused": 0, "open_frames": 0, "frames": 0}
    _, want = run_oracle(system, xyz, system.box9(n))
    _sums_equal(got, want)
    monkeypatch.setenv("GORDER_HIP_LOCAL_NO_DECIDE", "1")
    eng2, got2 = _submit_in_batches(system, xyz, system.box9(n), 5)
    assert eng2.local_decide_stats()["submits"] == 0
    _sums_equal(got, got2)


def test_local_decide_reports_arrive_whenever_they_arrive(built, monkeypatch):
    """Submits queued back to back (no wait in between): the report of one is read by whichever later submit finds it
    there; when that is only changes which kernels run, never the sums."""
    system = synthetic.cg_membrane(700, leaflets=LEAFLETS_LOCAL, radius=2.5, n_types=2)
    n = 32
    xyz = system.frames(n, seed=79)
    monkeypatch.setenv("GORDER_HIP_LOCAL_DECIDE_NOTHING", "1")
    eng, got = run_gpu(system, xyz, system.box9(n), batches=8)
    monkeypatch.delenv("GORDER_HIP_LOCAL_DECIDE_NOTHING")
    st = eng.local_decide_stats()
    assert st["submits"] >= 1 and st["submits"] + st["paused"] == 8
    _, want = run_oracle(system, xyz, system.box9(n))
    _sums_equal(got, want)


# ---- k_local_sums: the table of cell edges from per-cell fixed-point sums, the cell list only for frames left open ----
@pytest.mark.parametrize("case", ["plain", "tall box", "crowded cell", "unwrapped z"])
def test_local_sums_and_the_cell_list_agree(built, monkeypatch, case):
    """k_local_sums + k_local_decide against the same frames through k_local_build + k_local_rowprefix for every frame
    (GORDER_HIP_LOCAL_NO_SUMS=1), and the oracle.  'tall box': atoms more than 8 nm from the middle of the box (the fixed
    point's range) — 'crowded cell': more than 4 095 atoms in one cell (its count's) —: such frames are left open and take
    the cell list; 'unwrapped z': lipids whole boxes away along the normal."""
    box = {"plain": None, "tall box": (16.0, 16.0, 24.0), "crowded cell": (16.0, 16.0, 10.0), "unwrapped z": None}[case]
    system = synthetic.cg_membrane(900 if case == "plain" else 600, leaflets=LEAFLETS_LOCAL, radius=2.0 if case == "crowded cell" else 2.5,
                                   n_types=2, box=box)
    n = 10
    xyz = system.frames(n, seed=83)
    rng = np.random.default_rng(9)
    if case == "tall box":
        xyz[4:, 12 * 5:12 * 6, 2] += 9.5                    # one lipid far up, from frame 4 on (still inside the box)
    if case == "crowded cell":
        crowd = rng.choice(xyz.shape[1], 5000, replace=False)
        xyz[2:7, crowd, 0] = (5.03 + rng.normal(0, 0.004, (5, 5000))).astype(np.float32)
        xyz[2:7, crowd, 1] = (9.41 + rng.normal(0, 0.004, (5, 5000))).astype(np.float32)
    if case == "unwrapped z":
        xyz[3:, 12 * 7:12 * 9, 2] += 2 * system.box[2]
        xyz[6:, 12 * 40:12 * 41, 2] -= 3 * system.box[2]
    eng, got = run_gpu(system, xyz, system.box9(n), batches=2)
    assert eng.local_decide_stats()["submits"] == 2
    monkeypatch.setenv("GORDER_HIP_LOCAL_NO_SUMS", "1")
    _, lists = run_gpu(system, xyz, system.box9(n), batches=2)
    monkeypatch.delenv("GORDER_HIP_LOCAL_NO_SUMS")
    _sums_equal(got, lists)
    assert got.counts[1].sum() > 0 and got.counts[2].sum() > 0
    if case != "crowded cell":              # (5 000 atoms on one spot: the oracle's f32 sums and the device's f64 ones may part)
        _, want = run_oracle(system, xyz, system.box9(n))
        _sums_equal(got, want)
    else:
        assert_sums_given_device_flags(system.tables, xyz, system.box9(n), got, max_flag_diffs=8)
