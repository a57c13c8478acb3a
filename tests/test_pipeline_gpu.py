"""Reader -> pinned staging -> copy stream -> kernels: gorder_hip_run_trajectory (the reference's read_trajectory,
common.rs:239-342) on real files, against the reference's goldens and, bit for bit, against the route the other
golden tests take (frames decoded up front, handed over through submit_host)."""
import os

import numpy as np
import pytest

from gorder_amd import HipEngine, abi, xtc
from gorder_amd import structure as st
from golden_util import GOLDEN, METHODS, Fixture, aa_setup, cg_setup, expected, ua_setup

pytestmark = pytest.mark.gpu
CG3 = os.path.join(GOLDEN, "cg3.xtc")


@pytest.fixture(scope="module")
def cg(built):
    return Fixture("cg")


@pytest.fixture(scope="module")
def pcpepg(built):
    return Fixture("pcpepg")


def write_fixture(fx, path, frames=None, precision=None):
    """The fixture's frames as a compressed XTC file written by the repo's own encoder: the coordinates sit on the
    file's grid already, so decoding returns exactly fx.xyz."""
    frames = np.arange(len(fx.times)) if frames is None else np.asarray(frames)
    prec = 100.0 if precision is None else precision      # the split files of the reference carry precision 100
    xtc.write_trajectory(path, fx.xyz[frames], fx.boxes[frames], times=fx.times[frames], precision=prec)


def npz_route(tables, fx, midx, frames, fi, batches=3):
    eng = HipEngine(tables)
    xyz = np.ascontiguousarray(fx.xyz[frames][:, midx, :])
    box = fx.boxes[frames]
    edges = np.linspace(0, len(frames), batches + 1).astype(int)
    for a, b in zip(edges[:-1], edges[1:]):
        if b > a:
            eng.submit_host(xyz[a:b], box[a:b], np.asarray(fi[a:b]))
    return eng, eng.finish()


def assert_same(a, b):
    assert a.n_frames == b.n_frames
    np.testing.assert_array_equal(a.sums, b.sums)
    np.testing.assert_array_equal(a.counts, b.counts)


def test_reference_file_through_the_pipeline(cg):
    """tests/golden/cg3.xtc is the reference's own tests/files/split/cg3.xtc (one frame, the whole system with
    water): the driver decodes the Master group out of it and must reproduce the npz route bit for bit."""
    tables, labels, midx = cg_setup(cg, leaflets=METHODS["global"])
    k = int(np.flatnonzero(cg.times == 354000.0)[0])
    eng = HipEngine(tables)
    stats = eng.run_trajectory([CG3], group=midx, threads=2)
    assert stats["n_frames"] == 1 and stats["n_batches"] == 1
    got = eng.finish()
    _, want = npz_route(tables, cg, midx, np.array([k]), np.array([0]), batches=1)
    assert_same(got, want)


@pytest.mark.parametrize("batch_frames,threads", [(0, 4), (7, 3), (1, 1)])
def test_cg_trajectory_end_to_end(cg, tmp_path, batch_frames, threads):
    """All 101 frames of the CG membrane, written as ONE compressed file by the repo's encoder, through
    reader -> driver -> results_tree: the reference's golden (cg_order_leaflets.yaml) and the npz route bit for bit,
    whatever the batch size and the number of decoder threads."""
    path = str(tmp_path / "cg.xtc")
    write_fixture(cg, path)
    tables, labels, midx = cg_setup(cg, leaflets=METHODS["global"])
    eng = HipEngine(tables)
    stats = eng.run_trajectory([path], group=midx, threads=threads, batch_frames=batch_frames)
    got = eng.finish()
    assert stats["n_frames"] == got.n_frames == 101
    assert stats["bytes_h2d"] == 101 * (len(midx) * 12 + 36)
    bad = st.compare_trees(st.results_tree(got, labels, "cg", leaflets=True), expected("cg_order_leaflets.yaml"))
    assert not bad, bad[:10]
    frames = cg.window()
    _, want = npz_route(tables, cg, midx, frames, frames)
    assert_same(got, want)


def test_window_step_concatenation_and_leaflet_frequency(cg, tmp_path):
    """begin / end / step on the frame times, several files read as one trajectory with the duplicate boundary frames
    dropped (CHANGELOG.md:64), frame indices k * step behind the leaflet frequency (tests_cg.rs:808-845:
    13 frames for 352-358 ns, step 5) — against cg_order_begin_end_step.yaml and the npz route."""
    n = len(cg.times)
    cuts = [0, 21, 40, 41, 77, n]                       # five files; each starts with the last frame of the one before
    paths = []
    for q, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        p = str(tmp_path / f"part{q}.xtc")
        write_fixture(cg, p, frames=np.arange(max(a - 1, 0), b))
        paths.append(p)
    tables, labels, midx = cg_setup(cg, leaflets=METHODS["global"], frequency=5)
    eng = HipEngine(tables)
    stats = eng.run_trajectory(paths, group=midx, begin=352_000.0, end=358_000.0, step=5, threads=3, batch_frames=4)
    got = eng.finish()
    assert stats["n_frames"] == 13
    bad = st.compare_trees(st.results_tree(got, labels, "cg", leaflets=True), expected("cg_order_begin_end_step.yaml"))
    assert not bad, bad[:10]
    frames = cg.window(352_000.0, 358_000.0, 5)
    _, want = npz_route(tables, cg, midx, frames, np.arange(13) * 5)
    assert_same(got, want)


def test_aa_trajectory_end_to_end(pcpepg, tmp_path):
    """The all-atom membrane (51 frames, 3 molecule types): aa_order_leaflets.yaml through the pipeline."""
    path = str(tmp_path / "aa.xtc")
    write_fixture(pcpepg, path)
    tables, labels, midx = aa_setup(pcpepg, leaflets=METHODS["individual"])
    eng = HipEngine(tables)
    stats = eng.run_trajectory([path], group=midx, threads=4, batch_frames=16)
    got = eng.finish()
    assert stats["n_frames"] == 51 and stats["n_batches"] == 4
    bad = st.compare_trees(st.results_tree(got, labels, "aa", leaflets=True), expected("aa_order_leaflets.yaml"))
    assert not bad, bad[:10]
    frames = pcpepg.window()
    _, want = npz_route(tables, pcpepg, midx, frames, frames)
    assert_same(got, want)


def test_errors_end_the_run(cg, tmp_path):
    tables, labels, midx = cg_setup(cg)
    eng = HipEngine(tables)
    with pytest.raises(abi.GorderHipError) as e:                 # a file that is not there
        eng.run_trajectory([str(tmp_path / "missing.xtc")], group=midx)
    assert e.value.status == abi.ERR_INVALID_ARGUMENT
    with pytest.raises(abi.GorderHipError) as e:                 # a group that does not match the tables
        HipEngine(tables).run_trajectory([CG3], group=midx[:-1])
    assert e.value.status == abi.ERR_INVALID_ARGUMENT
    # an analysis error in the middle of the file: the first error ends the run (common.rs:248) with its status
    path = str(tmp_path / "badbox.xtc")
    box = cg.boxes[:30].copy()
    box[11, 0, 1] = 0.25
    xtc.write_trajectory(path, cg.xyz[:30], box, times=cg.times[:30], precision=100.0)
    with pytest.raises(abi.GorderHipError) as e:
        HipEngine(tables).run_trajectory([path], group=midx, batch_frames=4)
    assert e.value.status == abi.ERR_NOT_ORTHOGONAL_BOX


def test_submit_host_is_double_buffered(cg):
    """Many small host batches in a row (the copy of batch k+1 overlaps the kernels of batch k) equal one big one."""
    tables, labels, midx = cg_setup(cg, leaflets=METHODS["local"])
    frames = cg.window()
    _, many = npz_route(tables, cg, midx, frames, frames, batches=37)
    _, one = npz_route(tables, cg, midx, frames, frames, batches=1)
    assert_same(many, one)


@pytest.mark.parametrize("device_decode", [False, True])
def test_frame_shards_add_up_to_the_whole_trajectory(cg, tmp_path, device_decode):
    """SURVEY 8e through the driver: n ranks pass the same files and (i, n); each analyses a contiguous share of the
    frames the window selects, and the shares' accumulators add up to the single-handle result exactly (what
    gorder_hip_allreduce / SystemTopology::reduce then does across GPUs)."""
    tables, labels, midx = cg_setup(cg, leaflets=METHODS["global"])
    a, b = str(tmp_path / "a.xtc"), str(tmp_path / "b.xtc")
    xtc.write_trajectory(a, cg.xyz[:37], cg.boxes[:37], times=cg.times[:37], precision=100.0)
    xtc.write_trajectory(b, cg.xyz[36:], cg.boxes[36:], times=cg.times[36:], precision=100.0)     # duplicate boundary frame
    for kw, total in ((dict(), 101), (dict(begin=float(cg.times[5]), end=float(cg.times[95]), step=3), 31)):
        whole = HipEngine(tables)
        st = whole.run_trajectory([a, b], group=midx, threads=2, device_decode=device_decode, **kw)
        want = whole.finish()
        assert st["n_frames"] == total == want.n_frames
        for n in (2, 3, 7):
            sums = np.zeros_like(want.sums)
            counts = np.zeros_like(want.counts)
            first = 0
            for i in range(n):
                eng = HipEngine(tables)
                st = eng.run_trajectory([a, b], group=midx, threads=2, device_decode=device_decode, batch_frames=8,
                                        shard=(i, n), **kw)
                assert st["shard_frames_total"] == total and st["shard_first"] == first == total * i // n
                assert st["n_frames"] == total * (i + 1) // n - first
                first += st["n_frames"]
                got = eng.finish()
                sums += got.sums
                counts += got.counts
            assert first == total
            np.testing.assert_array_equal(sums, want.sums)
            np.testing.assert_array_equal(counts, want.counts)


@pytest.mark.parametrize("method,frequency", [("global", 5), ("individual", 0), ("local", 3)])
def test_frame_shards_prime_their_leaflets(cg, tmp_path, method, frequency):
    """a leaflet frequency other than every frame: a shard that begins between two assignment frames fetches the one
    frame it depends on itself (the cross-thread wait of leaflets.rs:1529-1565 as one extra frame read); lipids are
    made to change sides over the trajectory so that WHICH assignment a frame uses shows in the result"""
    tables, labels, midx = cg_setup(cg, leaflets=METHODS[method], frequency=frequency)
    xyz = cg.xyz[:40].copy()
    zmid = float(cg.boxes[0][2, 2]) / 2.0
    heads = np.asarray(tables.molecule_types[0].heads)
    per = int(np.diff(np.sort(heads))[0]) if len(heads) > 1 else 12
    for f in range(40):                 # from frame f on, lipid f of the first type is mirrored through the mid-plane
        a0 = int(midx[np.sort(heads)[f]])
        sl = slice(a0, a0 + per)
        xyz[f:, sl, 2] = 2.0 * zmid - xyz[f:, sl, 2]
    path = str(tmp_path / "flip.xtc")
    xtc.write_trajectory(path, xyz, cg.boxes[:40], times=cg.times[:40], precision=100.0)
    whole = HipEngine(tables)
    whole.run_trajectory([path], group=midx, threads=2)
    want = whole.finish()
    every = HipEngine(cg_setup(cg, leaflets=METHODS[method], frequency=1)[0])
    every.run_trajectory([path], group=midx, threads=2)
    assert not np.array_equal(every.finish().sums, want.sums)          # the frequency matters on this trajectory
    for n in (2, 3, 7):
        sums, counts = np.zeros_like(want.sums), np.zeros_like(want.counts)
        for i in range(n):
            eng = HipEngine(tables)
            eng.run_trajectory([path], group=midx, threads=2, device_decode=bool(i % 2), batch_frames=8, shard=(i, n))
            got = eng.finish()
            sums += got.sums
            counts += got.counts
        np.testing.assert_array_equal(sums, want.sums)
        np.testing.assert_array_equal(counts, want.counts)


@pytest.mark.parametrize("device_decode", [False, True])
def test_empty_selections(cg, tmp_path, device_decode):
    """a window that selects nothing, and more shards than frames: zero frames analysed, no error, results all zero"""
    tables, labels, midx = cg_setup(cg, leaflets=METHODS["global"])
    path = str(tmp_path / "c.xtc")
    write_fixture(cg, path, frames=np.arange(5))
    eng = HipEngine(tables)
    st = eng.run_trajectory([path], group=midx, threads=2, device_decode=device_decode, begin=1e9)
    assert st["n_frames"] == 0 and eng.finish().n_frames == 0
    total = 0
    for i in range(8):                      # 5 frames over 8 shards: three of them are empty
        eng = HipEngine(tables)
        st = eng.run_trajectory([path], group=midx, threads=2, device_decode=device_decode, shard=(i, 8))
        assert st["n_frames"] in (0, 1) and st["shard_frames_total"] == 5
        total += eng.finish().n_frames
    assert total == 5
    with pytest.raises(abi.GorderHipError):
        HipEngine(tables).run_trajectory([path], group=midx, shard=(8, 8))
