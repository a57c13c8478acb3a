"""Host-side trajectory reader (include/gorder_xtc.h): XTC decoding, group-partial conversion, the time
window / step / concatenation rules of common.rs:239-304, and TRR.  CPU only."""
import os
import struct

import numpy as np
import pytest

from gorder_amd import xtc
from golden_util import GOLDEN, Fixture

CG3 = os.path.join(GOLDEN, "cg3.xtc")      # tests/files/split/cg3.xtc of the reference: 1 frame, t = 354 000 ps


@pytest.fixture(scope="module")
def cg(built):
    return Fixture("cg")


def test_xtc_frame_equals_the_fixture_integers(cg):
    """The compressed frame decodes to exactly the integers stored in tests/golden/cg.npz (made through the same
    reader from all five split files), for the whole frame and for a group of atoms."""
    xyz, box, t, prec = xtc.read_trajectory([CG3], group=None, return_precision=True)
    assert xyz.shape == (1, 16769, 3) and prec == 100.0 and t[0] == 354000.0
    k = int(np.flatnonzero(cg.times == t[0])[0])
    n = cg.xyz.shape[1]
    np.testing.assert_array_equal(xyz[0, :n], cg.xyz[k])
    np.testing.assert_array_equal(box[0], cg.boxes[k])
    group = np.arange(n - 1, -1, -7, dtype=np.uint32)          # any subset, any order
    part, _, _ = xtc.read_trajectory([CG3], group=group)
    np.testing.assert_array_equal(part[0], cg.xyz[k][group])
    part, _, _ = xtc.read_trajectory([CG3], group=group, threads=4)
    np.testing.assert_array_equal(part[0], cg.xyz[k][group])


def test_window_step_and_concatenation():
    # begin/end are inclusive times in ps (common.rs:239-246)
    assert xtc.read_trajectory([CG3], begin=354000.0, end=354000.0)[0].shape[0] == 1
    assert xtc.read_trajectory([CG3], begin=354000.5)[0].shape[0] == 0
    assert xtc.read_trajectory([CG3], end=353999.0)[0].shape[0] == 0
    # the same file twice: the second copy starts with the time the first ended with -> dropped (CHANGELOG.md:64)
    assert xtc.read_trajectory([CG3, CG3])[0].shape[0] == 1


def test_corrupt_files_are_rejected(tmp_path):
    raw = open(CG3, "rb").read()
    bad_magic = tmp_path / "magic.xtc"
    bad_magic.write_bytes(b"\x00\x00\x07\xcc" + raw[4:])
    with pytest.raises(IOError):
        xtc.read_trajectory([str(bad_magic)])
    cut = tmp_path / "cut.xtc"
    cut.write_bytes(raw[: len(raw) // 2])
    with pytest.raises(IOError):
        xtc.read_trajectory([str(cut)])
    with pytest.raises(IOError):
        xtc.read_trajectory([str(tmp_path / "missing.xtc")])


def _trr_frame(step, t, box, x=None, v=None, double=False):
    """One GROMACS TRR frame (XDR, big-endian) — written here, byte by byte, from the format description."""
    n = len(x if x is not None else v)
    r, rs = (">d", 8) if double else (">f", 4)
    ver = b"GMX_trn_file"
    out = struct.pack(">ii", 1993, len(ver) + 1) + struct.pack(">i", len(ver)) + ver
    sizes = [0, 0, 9 * rs if box is not None else 0, 0, 0, 0, 0, 3 * n * rs if x is not None else 0,
             3 * n * rs if v is not None else 0, 0]
    out += struct.pack(">13i", *sizes, n, step, 0)
    out += struct.pack(r, t) + struct.pack(r, 0.0)
    if box is not None:
        out += b"".join(struct.pack(r, float(b)) for b in np.asarray(box).reshape(9))
    for arr in (x, v):
        if arr is not None:
            out += b"".join(struct.pack(r, float(c)) for c in np.asarray(arr).reshape(-1))
    return out


@pytest.mark.parametrize("double", [False, True])
def test_trr(tmp_path, double):
    rng = np.random.default_rng(5)
    n = 37
    box = np.diag([5.0, 6.0, 7.0])
    frames = [rng.uniform(0, 5, (n, 3)).astype(np.float32) for _ in range(4)]
    vel = rng.normal(size=(n, 3)).astype(np.float32)
    data = b""
    for k, f in enumerate(frames):
        data += _trr_frame(10 * k, 2.5 * k, box, x=f, v=vel if k == 1 else None, double=double)
        if k == 2:                       # a velocities-only frame in between: skipped
            data += _trr_frame(25, 6.0, box, x=None, v=vel, double=double)
    path = tmp_path / "t.trr"
    path.write_bytes(data)
    xyz, b9, t = xtc.read_trajectory([str(path)])
    assert xyz.shape == (4, n, 3)
    np.testing.assert_array_equal(t, np.array([0.0, 2.5, 5.0, 7.5], dtype=np.float32))
    for k in range(4):
        np.testing.assert_array_equal(xyz[k], frames[k])      # f32 survives the f64 round trip exactly
        np.testing.assert_array_equal(b9[k], box.astype(np.float32))
    for threads in (2, 3, 8):            # the multi-threaded window reader returns exactly the same
        again = xtc.read_trajectory([str(path)], threads=threads)
        assert all(np.array_equal(p, q) for p, q in zip((xyz, b9, t), again))
        part2 = xtc.read_trajectory([str(path)], begin=2.0, end=6.0, step=2, threads=threads)
        np.testing.assert_array_equal(part2[2], np.array([2.5], dtype=np.float32))
    group = np.array([5, 0, 36], dtype=np.uint32)
    part, _, tt = xtc.read_trajectory([str(path)], group=group, begin=2.0, end=6.0, step=1)
    np.testing.assert_array_equal(tt, np.array([2.5, 5.0], dtype=np.float32))
    np.testing.assert_array_equal(part[1], frames[2][group])


def test_trr_of_the_reference_when_present(cg):
    """tests/files/split/cg3.trr (t = 355 000 ps) holds the same frame as the XTC data, in full precision
    written from it: the lipid beads equal the fixture frame of that time.  Needs /root/reference."""
    path = "/root/reference/tests/files/split/cg3.trr"
    if not os.path.exists(path):
        pytest.skip("reference checkout not present")
    xyz, box, t = xtc.read_trajectory([path])
    k = int(np.flatnonzero(cg.times == t[0])[0])
    np.testing.assert_array_equal(xyz[0, : cg.xyz.shape[1]], cg.xyz[k])
    np.testing.assert_array_equal(box[0], cg.boxes[k])


# ---- GRO text trajectories (groan_rs GroReader, common.rs:322-333) -----------------------------------------------
def _write_gro(path, frames, boxes, times, decimals=3, triclinic=False):
    w = decimals + 5
    with open(path, "w") as f:
        for x, b, t in zip(frames, boxes, times):
            f.write(f"bilayer t= {t:.5f} step= {int(t * 50)}\n{len(x):5d}\n")
            for i, p in enumerate(x):
                f.write(f"{i % 99999 + 1:5d}{'POPC':<5s}{'C' + str(i % 97):>5s}{(i + 1) % 100000:5d}"
                        + "".join(f"{v:{w}.{decimals}f}" for v in p) + "\n")
            if triclinic:
                f.write(" ".join(f"{v:.5f}" for v in (b[0], b[1], b[2], 0.0, 0.0, 0.25, 0.0, 0.5, 0.75)) + "\n")
            else:
                f.write(f"{b[0]:10.5f}{b[1]:10.5f}{b[2]:10.5f}\n")


@pytest.mark.parametrize("decimals,threads", [(3, 1), (3, 3), (5, 2)])
def test_gro_trajectory(tmp_path, built, decimals, threads):
    rng = np.random.default_rng(5)
    n_frames, n_atoms = 9, 250
    frames = np.round(rng.uniform(-1.0, 12.0, (n_frames, n_atoms, 3)), decimals).astype(np.float32)
    boxes = np.round(rng.uniform(9.0, 11.0, (n_frames, 3)), 5).astype(np.float32)
    times = 100.0 + 20.0 * np.arange(n_frames)
    path = str(tmp_path / "traj.gro")
    _write_gro(path, frames, boxes, times, decimals)
    x, b, t = xtc.read_trajectory([path], threads=threads)
    np.testing.assert_array_equal(x, frames)
    np.testing.assert_array_equal(t, times.astype(np.float32))
    np.testing.assert_array_equal(b[:, [0, 1, 2], [0, 1, 2]], boxes)
    assert (b.reshape(n_frames, 9)[:, [1, 2, 3, 5, 6, 7]] == 0).all()
    # group-partial conversion, time window and step like the binary formats
    grp = np.array([3, 17, 249, 100], dtype=np.uint32)
    x, b, t = xtc.read_trajectory([path], group=grp, begin=140.0, end=240.0, step=2, threads=threads)
    np.testing.assert_array_equal(t, np.array([140.0, 180.0, 220.0], dtype=np.float32))
    np.testing.assert_array_equal(x, frames[[2, 4, 6]][:, grp])


def test_gro_triclinic_box_and_truncation(tmp_path, built):
    frames = np.zeros((2, 4, 3), dtype=np.float32)
    path = str(tmp_path / "tri.gro")
    _write_gro(path, frames, np.array([[5.0, 6.0, 7.0]] * 2), [0.0, 1.0], triclinic=True)
    x, b, t = xtc.read_trajectory([path])
    assert x.shape == (2, 4, 3)
    np.testing.assert_array_equal(b[0], np.array([[5.0, 0.0, 0.0], [0.25, 6.0, 0.0], [0.5, 0.75, 7.0]], dtype=np.float32))
    data = open(path).read()
    cut = str(tmp_path / "cut.gro")
    open(cut, "w").write(data[: len(data) - 60])            # the second frame loses its box line and last atom
    with pytest.raises(IOError):
        xtc.read_trajectory([cut])


def test_writer_round_trip_and_reference_bytes(tmp_path, cg):
    """The repo's own XTC encoder (tooling for the reader -> GPU pipeline tests): what it writes decodes to
    round(x * precision) / precision, for compressed frames with runs, for boxes wide enough to need per-dimension
    bit fields, and for tiny uncompressed frames; and re-encoding a frame of the reference's own file reproduces that
    file byte for byte (same algorithm as the GROMACS writer that produced it)."""
    rng = np.random.default_rng(3)
    # a membrane-like system: neighbours close together (runs of small offsets), wrapped into the box
    n, f = 5000, 4
    base = np.cumsum(rng.normal(0, 0.08, (n, 3)), axis=0) % np.array([9.0, 9.0, 8.0])
    xyz = (base[None] + rng.normal(0, 0.02, (f, n, 3))).astype(np.float32)
    box = np.zeros((f, 3, 3), np.float32)
    box[:, 0, 0], box[:, 1, 1], box[:, 2, 2] = 9.0, 9.0, 8.0
    path = str(tmp_path / "w.xtc")
    for prec in (1000.0, 100.0):
        xtc.write_trajectory(path, xyz, box, times=np.arange(f) * 10.0, precision=prec)
        back, bback, t, p = xtc.read_trajectory([path], return_precision=True)
        assert p == prec and np.array_equal(t, np.arange(f, dtype=np.float32) * 10.0)
        np.testing.assert_array_equal(bback, box)
        want = (np.where(xyz >= 0, np.float32(prec) * xyz + np.float32(0.5), np.float32(prec) * xyz - np.float32(0.5))
                .astype(np.int32).astype(np.float32) * (np.float32(1.0) / np.float32(prec)))
        np.testing.assert_array_equal(back, want)
        assert os.path.getsize(path) < 0.55 * xyz.nbytes            # the small-offset runs do compress
    # coordinates spread over > 2^24 grid steps: per-dimension bit fields instead of one mixed-radix number
    far = (rng.random((2, 300, 3)) * np.array([20000.0, 5.0, 5.0])).astype(np.float32)
    xtc.write_trajectory(path, far, box[:2])
    back = xtc.read_trajectory([path])[0]
    np.testing.assert_allclose(back, far, atol=2e-3, rtol=1e-6)
    # <= 9 atoms: raw floats
    xtc.write_trajectory(path, xyz[:, :7], box)
    np.testing.assert_array_equal(xtc.read_trajectory([path])[0], xyz[:, :7])
    # the reference's frame, re-encoded: identical bytes
    rx, rb, rt, rp = xtc.read_trajectory([CG3], return_precision=True)
    xtc.write_trajectory(path, rx, rb, times=rt, precision=rp)
    ours, theirs = open(path, "rb").read(), open(CG3, "rb").read()
    assert ours[:8] == theirs[:8] and ours[12:] == theirs[12:]    # all but the step number, which the reader does not return
