"""Global leaflets in one read (k_bonds_tiled<..., MOM> + k_spec_resolve / _check / _fixup): from the second batch
on the order kernel routes every molecule by the last assignment and sums the membrane's normal coordinate on the way; the
exact centres come from those sums, the molecules whose side was mispredicted are moved afterwards.  Everything here is
compared with the two-kernel path (GORDER_HIP_NO_SPECULATE=1) — sums and counts EQUAL, exported sides and distances equal —
and with the oracle."""
import numpy as np
import pytest

from gorder_amd import HipEngine, abi, synthetic
from gorder_amd.abi import LEAFLETS_GLOBAL
from oracle import oracle
from leaflet_check import assert_sums_given_device_flags

pytestmark = pytest.mark.gpu


def run(system, xyz, box, batches, monkeypatch=None, speculate=True, frame_index=None):
    if monkeypatch is not None:
        if speculate:
            monkeypatch.delenv("GORDER_HIP_NO_SPECULATE", raising=False)
        else:
            monkeypatch.setenv("GORDER_HIP_NO_SPECULATE", "1")
    eng = HipEngine(system.tables)
    n = xyz.shape[0]
    fi = np.arange(n) if frame_index is None else np.asarray(frame_index)
    edges = np.linspace(0, n, batches + 1).astype(int)
    for a, b in zip(edges[:-1], edges[1:]):
        if b > a:
            eng.submit_host(xyz[a:b], None if box is None else box[a:b], fi[a:b])
    res = eng.finish()
    eng.stats = eng.speculation_stats()
    assert (eng.stats["batches"] > 0) == (speculate and batches > 1) or not eng.stats["enabled"]
    return eng, res


def both(system, xyz, box, batches, monkeypatch):
    e1, spec = run(system, xyz, box, batches, monkeypatch, True)
    e2, plain = run(system, xyz, box, batches, monkeypatch, False)
    np.testing.assert_array_equal(spec.sums, plain.sums)
    np.testing.assert_array_equal(spec.counts, plain.counts)
    f1, r1 = e1.leaflets()
    f2, r2 = e2.leaflets()
    assert r1 == r2
    np.testing.assert_array_equal(f1, f2)
    np.testing.assert_allclose(e1.leaflet_distances(), e2.leaflet_distances(), atol=2e-5)
    monkeypatch.delenv("GORDER_HIP_NO_SPECULATE", raising=False)
    return spec


def oracle_sums(system, xyz, box):
    trig = oracle.TRIG_MIRROR if (system.tables.flags & abi.FLAG_TRIG_ACOS_COS) else oracle.TRIG_DIRECT
    o = oracle.OracleEngine(system.tables, trig=trig, n_threads=4)
    o.submit(xyz, box, np.arange(xyz.shape[0]))
    return o.finish()


@pytest.mark.parametrize("n_frames,batches", [(40, 3), (37, 4), (9, 9)])
@pytest.mark.parametrize("kind", ["aa", "cg", "cg two types"])
def test_speculative_batches_equal_the_two_kernel_path(built, monkeypatch, kind, n_frames, batches):
    if kind == "aa":
        system = synthetic.aa_membrane(48, leaflets=LEAFLETS_GLOBAL)
    else:
        system = synthetic.cg_membrane(300, leaflets=LEAFLETS_GLOBAL, n_types=2 if "two" in kind else 1)
    xyz = system.frames(n_frames, seed=71)
    box = system.box9(n_frames)
    got = both(system, xyz, box, batches, monkeypatch)
    want = oracle_sums(system, xyz, box)
    np.testing.assert_array_equal(got.sums, want.sums)
    np.testing.assert_array_equal(got.counts, want.counts)
    assert got.counts[1].sum() > 0 and got.counts[2].sum() > 0


def test_molecules_that_change_sides_are_moved(built, monkeypatch):
    """Lipids whose head crosses the membrane's centre from one frame to the next (and back): the order kernel routes them by
    the side they had before the batch, k_spec_check finds the frames where that was wrong, k_spec_fixup moves their ticks."""
    system = synthetic.cg_membrane(200, leaflets=LEAFLETS_GLOBAL, n_types=2)
    n = 48
    xyz = system.frames(n, seed=73).astype(np.float32)
    zc = float(system.box[2]) / 2
    rng = np.random.default_rng(3)
    movers = rng.choice(200, 12, replace=False)
    for f in range(n):
        for k, m in enumerate(movers):
            if (f // (2 + k % 5)) % 2:                        # on the other side for a few frames at a time
                a = slice(m * 12, m * 12 + 12)
                xyz[f, a, 2] = (2 * zc - xyz[f, a, 2]).astype(np.float32)
    box = system.box9(n)
    e1, spec = run(system, xyz, box, 4, monkeypatch, True)
    assert e1.stats["batches"] == 3 and e1.stats["moved"] > 50 and e1.stats["exact_frames"] == 0
    got = both(system, xyz, box, 4, monkeypatch)
    assert_sums_given_device_flags(system.tables, xyz, box, got, max_flag_diffs=8)


def test_a_membrane_across_the_periodic_boundary_takes_the_exact_kernel(built, monkeypatch):
    """Shifted by half a box along the normal the membrane's atoms sit at both ends of the box: the plain mean of their
    coordinates is not the centre, k_spec_resolve says so for every frame, and the exact kernel decides (then the handle
    stops speculating)."""
    system = synthetic.cg_membrane(200, leaflets=LEAFLETS_GLOBAL)
    n = 60
    xyz = system.frames(n, seed=79).astype(np.float32)
    L = float(system.box[2])
    xyz[:, :, 2] = np.mod(xyz[:, :, 2] + L / 2, L).astype(np.float32)
    box = system.box9(n)
    e1, _ = run(system, xyz, box, 5, monkeypatch, True)
    assert e1.stats["exact_frames"] >= 12 and not e1.stats["enabled"] and 1 <= e1.stats["batches"] < 4
    got = both(system, xyz, box, 5, monkeypatch)
    want = oracle_sums(system, xyz, box)
    np.testing.assert_array_equal(got.sums, want.sums)
    np.testing.assert_array_equal(got.counts, want.counts)


def test_a_membrane_group_that_is_part_of_the_frame(built, monkeypatch):
    """Water behind the lipids: the membrane group is a range of the frame's atoms, the exact kernel is the index-list one,
    and the tiles' windows end where the group ends."""
    base = synthetic.cg_membrane(150, leaflets=LEAFLETS_GLOBAL, n_types=2)
    n_lip = base.n_atoms
    n_water = 700
    n = 30
    rng = np.random.default_rng(11)
    lip = base.frames(n, seed=83)
    t = base.tables
    t.n_atoms = n_lip + n_water
    water = rng.uniform(0, 1, (n, n_water, 3)).astype(np.float32) * np.asarray(base.box, dtype=np.float32)
    xyz = np.ascontiguousarray(np.concatenate([lip, water], axis=1))
    system = synthetic.System("cg+water", t, np.concatenate([base.base, water[0]]), base.box, jitter=0.0)
    box = base.box9(n)
    e1, _ = run(system, xyz, box, 3, monkeypatch, True)
    assert e1.stats["batches"] == 2 and e1.stats["exact_frames"] == 0
    got = both(system, xyz, box, 3, monkeypatch)
    want = oracle_sums(system, xyz, box)
    np.testing.assert_array_equal(got.sums, want.sums)
    np.testing.assert_array_equal(got.counts, want.counts)


def test_the_literal_cosine_and_flipped_leaflets(built, monkeypatch):
    system = synthetic.aa_membrane(32, leaflets=LEAFLETS_GLOBAL, flip=True)
    system.tables.flags |= abi.FLAG_TRIG_ACOS_COS
    n = 26
    xyz = system.frames(n, seed=89)
    box = system.box9(n)
    got = both(system, xyz, box, 3, monkeypatch)
    want = oracle_sums(system, xyz, box)
    np.testing.assert_array_equal(got.sums, want.sums)


def test_a_non_finite_coordinate_of_a_membrane_atom_outside_every_bond(built, monkeypatch):
    """leaflets.rs:190-192: any non-finite coordinate of a membrane atom makes the centre NaN -> InvalidGlobalMembraneCenter,
    also when no bond touches the atom (a bead of the coarse-grained lipid that only the membrane group holds) and the
    coordinate is not the normal one.  The sums of the speculative kernel cannot vouch for such a frame; the exact kernel
    raises the reference's error."""
    system = synthetic.cg_membrane(100, leaflets=LEAFLETS_GLOBAL)
    mt = system.tables.molecule_types[0]
    mt.bonds = mt.bonds[1:]                      # bead 0 (NC3) of every lipid is in no bond any more
    n = 24
    xyz = system.frames(n, seed=97).astype(np.float32)
    xyz[17, 5 * 12 + 0, 0] = np.inf
    box = system.box9(n)
    codes = []
    for speculate in (True, False):
        eng = None
        with pytest.raises(abi.GorderHipError) as e:
            eng, _ = run(system, xyz, box, 3, monkeypatch, speculate)
        codes.append((e.value.status, e.value.frame if hasattr(e.value, "frame") else None))
    assert codes[0][0] == codes[1][0] == abi.ERR_INVALID_GLOBAL_MEMBRANE_CENTER


def test_a_batch_that_mispredicts_every_molecule(built, monkeypatch):
    """The membrane mirrored about its mid-plane from the second batch on: every molecule of every later frame is on the
    other side than the last assignment says.  The fix-up moves them all (nothing overflows: it keeps no list), the handle
    stops speculating after that batch, and the sums are the two-kernel path's."""
    system = synthetic.cg_membrane(120, leaflets=LEAFLETS_GLOBAL, n_types=2)
    n = 36
    xyz = system.frames(n, seed=101).astype(np.float32)
    zc = float(system.box[2]) / 2
    xyz[9:, :, 2] = (2 * zc - xyz[9:, :, 2]).astype(np.float32)
    box = system.box9(n)
    e1, _ = run(system, xyz, box, 4, monkeypatch, True)
    assert e1.stats["moved"] >= 9 * 120 and not e1.stats["enabled"]
    got = both(system, xyz, box, 4, monkeypatch)
    want = oracle_sums(system, xyz, box)
    np.testing.assert_array_equal(got.sums, want.sums)
    np.testing.assert_array_equal(got.counts, want.counts)


def test_a_first_batch_of_one_frame_and_reset(built, monkeypatch):
    system = synthetic.aa_membrane(24, leaflets=LEAFLETS_GLOBAL)
    n = 21
    xyz = system.frames(n, seed=103)
    box = system.box9(n)
    eng = HipEngine(system.tables)
    for a, b in ((0, 1), (1, 14), (14, 21)):
        eng.submit_host(xyz[a:b], box[a:b], np.arange(a, b))
    first = eng.finish()
    assert eng.speculation_stats()["batches"] == 2
    want = oracle_sums(system, xyz, box)
    np.testing.assert_array_equal(first.sums, want.sums)
    eng.reset()                                   # a new run on the same handle: the first batch assigns exactly again
    assert eng.speculation_stats() == {"batches": 0, "moved": 0, "exact_frames": 0, "enabled": True}
    for a, b in ((0, 8), (8, 21)):
        eng.submit_host(xyz[a:b], box[a:b], np.arange(a, b))
    again = eng.finish()
    np.testing.assert_array_equal(again.sums, want.sums)
    np.testing.assert_array_equal(again.counts, want.counts)
    assert eng.speculation_stats()["batches"] == 1


@pytest.mark.parametrize("kind", ["aa", "cg movers"])
def test_per_frame_rows_in_a_speculative_batch(built, monkeypatch, kind):
    """Error estimation on (timewise.rs:130-186): the rows hold every frame's ticks under total / upper / lower; the tiled
    kernel that writes them sums the membrane as well, and the fix-up moves a mispredicted molecule's ticks in the rows too."""
    if kind == "aa":
        system = synthetic.aa_membrane(40, leaflets=LEAFLETS_GLOBAL, timewise=True)
        n = 30
        xyz = system.frames(n, seed=107)
    else:
        system = synthetic.cg_membrane(150, leaflets=LEAFLETS_GLOBAL, n_types=2, timewise=True)
        n = 40
        xyz = system.frames(n, seed=109).astype(np.float32)
        zc = float(system.box[2]) / 2
        for f in range(n):
            for k, m in enumerate((3, 17, 58, 101, 140)):
                if (f // (2 + k)) % 2:
                    a = slice(m * 12, m * 12 + 12)
                    xyz[f, a, 2] = (2 * zc - xyz[f, a, 2]).astype(np.float32)
    box = system.box9(n)
    e1, spec = run(system, xyz, box, 3, monkeypatch, True)
    assert e1.stats["batches"] == 2 and (kind == "aa" or e1.stats["moved"] > 10)
    tw1 = e1.timewise(n)
    e2, plain = run(system, xyz, box, 3, monkeypatch, False)
    tw2 = e2.timewise(n)
    np.testing.assert_array_equal(spec.sums, plain.sums)
    np.testing.assert_array_equal(spec.counts, plain.counts)
    np.testing.assert_array_equal(tw1[0], tw2[0])
    np.testing.assert_array_equal(tw1[1], tw2[1])
    # and the rows add up to the totals, leaflet by leaflet
    np.testing.assert_array_equal(np.asarray(tw1[0]).sum(axis=0), spec.sums)
    np.testing.assert_array_equal(np.asarray(tw1[1]).sum(axis=0), spec.counts)
