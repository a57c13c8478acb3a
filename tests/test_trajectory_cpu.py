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
