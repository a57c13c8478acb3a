"""gorder_xtc_pack_window (host half of the device decoder): the frames it selects, their boxes and times are those of
gorder_xtc_read_window; the blob is laid out as include/gorder_xtc.h says.  No GPU: the decoding itself is compared in
tests/test_xtc_device_gpu.py."""
import os

import numpy as np
import pytest

from gorder_amd import xtc

HERE = os.path.dirname(os.path.abspath(__file__))
CG3 = os.path.join(HERE, "golden", "cg3.xtc")


def synthetic(tmp_path, n_atoms=700, n_frames=23, precision=1000.0, span=6.0, seed=3, name="s.xtc"):
    rng = np.random.default_rng(seed)
    base = rng.uniform(0.0, span, size=(n_atoms // 3 + 1, 1, 3)) + rng.normal(0.0, 0.05, size=(n_atoms // 3 + 1, 3, 3))
    x0 = base.reshape(-1, 3)[:n_atoms]
    xyz = (x0[None] + rng.normal(0.0, 0.02, size=(n_frames, n_atoms, 3))).astype(np.float32)
    box = np.zeros((n_frames, 3, 3), np.float32)
    box[:, 0, 0] = box[:, 1, 1] = box[:, 2, 2] = span
    path = str(tmp_path / name)
    xtc.write_trajectory(path, xyz, box, times=np.arange(n_frames, dtype=np.float32) * 10.0, precision=precision)
    return path


def check_layout(w):
    fr = w["frames"]
    assert np.all(fr["offset"] % 64 == 0)
    ends = fr["offset"] + ((fr["n_bytes"].astype(np.uint64) + 63) // 64) * 64 + 64
    assert np.all(ends[:-1] <= fr["offset"][1:]) and ends[-1] <= w["blob"].size
    for f in fr:      # the padding behind a block is zero
        pad = w["blob"][int(f["offset"]) + int(f["n_bytes"]):int(f["offset"]) + ((int(f["n_bytes"]) + 63) // 64) * 64 + 64]
        assert not pad.any()
    assert np.all(fr["n_bytes"] % 4 == 0)


def test_pack_selects_what_read_selects(built, tmp_path):
    path = synthetic(tmp_path)
    for kw in (dict(), dict(begin=30.0, end=170.0, step=3), dict(step=5), dict(begin=1e9)):
        x, b, t = xtc.read_trajectory([path], **kw)
        ws = xtc.pack_trajectory([path], chunk=7, threads=3, **kw)
        n = sum(len(w["time"]) for w in ws)
        assert n == len(t)
        if n:
            np.testing.assert_array_equal(np.concatenate([w["time"] for w in ws]), t)
            np.testing.assert_array_equal(np.concatenate([w["box"] for w in ws]), b)
            for w in ws:
                check_layout(w)
                assert w["n_atoms_file"] == 700 and w["n_stop"] == 700 and w["slot_of"] is None


def test_skip_counts_what_read_returns(built, tmp_path):
    a = synthetic(tmp_path, n_frames=13, name="a.xtc")
    b = synthetic(tmp_path, n_frames=9, seed=8, name="b.xtc")       # times 0..80: overlaps a's, only later ones are new
    for kw in (dict(), dict(begin=25.0, end=95.0, step=2), dict(step=4), dict(begin=1e9)):
        assert xtc.count_frames([a, b], **kw) == len(xtc.read_trajectory([a, b], **kw)[2])
    assert xtc.count_frames([CG3]) == 1


def test_pack_continues_when_the_blob_is_full(built, tmp_path):
    path = synthetic(tmp_path, n_frames=11)
    one = xtc.pack_trajectory([path], chunk=64)
    assert len(one) == 1 and len(one[0]["time"]) == 11
    per_frame = int(one[0]["frames"]["offset"][1])
    small = xtc.pack_trajectory([path], chunk=64, blob_capacity=int(one[0]["frames"]["offset"][3]), threads=2)
    assert len(small[0]["time"]) == 3 and len(small) >= 4 and sum(len(w["time"]) for w in small) == 11
    np.testing.assert_array_equal(np.concatenate([w["time"] for w in small]), one[0]["time"])
    # the same bytes, whatever the windowing
    def blocks(ws):
        return [bytes(w["blob"][int(f["offset"]):int(f["offset"]) + int(f["n_bytes"])]) for w in ws for f in w["frames"]]
    assert blocks(small) == blocks(one)
    with pytest.raises(IOError):      # not even one frame fits
        xtc.pack_trajectory([path], chunk=64, blob_capacity=per_frame // 2)


def test_a_frame_that_does_not_fit_is_met_again(built, tmp_path):
    """the driver's case: a batch's blob is partly full, the next file's FIRST frame does not fit what is left — the call
    refuses (nothing packed) and must leave the selection state untouched, so that the same frame opens the next batch"""
    import ctypes as C
    from gorder_amd.abi import CXtcFrame
    path = synthetic(tmp_path, n_frames=6)
    lib = xtc._lib()
    r = C.c_void_p()
    assert lib.gorder_xtc_open(path.encode(), None, 0, C.byref(r)) == 0
    state, last, used = C.c_uint64(0), C.c_double(float("-inf")), C.c_uint64(0)
    blob = np.empty(1 << 20, np.uint8)
    frames = (CXtcFrame * 8)()
    box, t = np.empty((8, 9), np.float32), np.empty(8, np.float32)
    args = lambda cap: (r, 0.0, -1.0, 2, C.byref(state), C.byref(last), blob.ctypes.data, cap, C.byref(used),
                        C.cast(frames, C.c_void_p), box.ctypes.data, t.ctypes.data, 8, 1)
    assert lib.gorder_xtc_pack_window(*args(100)) == -4           # GORDER_XTC_ERR_NO_SPACE: not one frame fits 100 bytes
    assert state.value == 0 and last.value == float("-inf")
    assert lib.gorder_xtc_pack_window(*args(1 << 20)) == 3        # every second of 6 frames, the first one included
    np.testing.assert_array_equal(t[:3], [0.0, 20.0, 40.0])
    lib.gorder_xtc_close(r)


@pytest.mark.parametrize("step", [1, 2, 3, 4])
@pytest.mark.parametrize("blob_frames", [1.5, 2.5, 3.2])
def test_batches_filled_across_files_select_what_read_window_selects(built, tmp_path, step, blob_frames):
    """gorder_hip_run_trajectory's reader (trajectory_driver.h): ONE blob per batch, filled across file boundaries with
    the capacity that is left; a call that cannot place its first selected frame returns NO_SPACE and the batch ends.
    Frames passed over before that frame (stepped over, or the duplicate frame at a file boundary, CHANGELOG.md:64) must
    not be counted twice when the next batch meets them again: the selection equals gorder_xtc_read_window's."""
    import ctypes as C
    from gorder_amd.abi import CXtcFrame
    rng = np.random.default_rng(3)
    paths, t0 = [], 0.0
    for k, nf in enumerate((4, 6, 3, 5)):           # every file starts with the last time of the one before
        xyz = rng.uniform(0, 6, size=(nf, 700, 3)).astype(np.float32)
        box = np.tile(np.eye(3, dtype=np.float32) * 6.0, (nf, 1, 1))
        p = str(tmp_path / f"part{k}.xtc")
        xtc.write_trajectory(p, xyz, box, times=(t0 + 10.0 * np.arange(nf)).astype(np.float32))
        t0 += 10.0 * (nf - 1)
        paths.append(p)
    want = xtc.read_trajectory(paths, step=step)[2]
    lib = xtc._lib()
    per_frame = int(xtc.pack_trajectory([paths[0]], chunk=4)[0]["frames"]["offset"][1])
    cap = int(per_frame * blob_frames)
    state, last = C.c_uint64(0), C.c_double(float("-inf"))
    got_times, batches = [], []
    f, r = 0, None
    while f < len(paths):
        blob = np.empty(cap, np.uint8)
        frames = (CXtcFrame * 8)()
        box, t = np.empty((8, 9), np.float32), np.empty(8, np.float32)
        n, filled = 0, 0
        while f < len(paths) and n < 8:
            if r is None:
                r = C.c_void_p()
                assert lib.gorder_xtc_open(paths[f].encode(), None, 0, C.byref(r)) == 0
            used = C.c_uint64(0)
            got = lib.gorder_xtc_pack_window_ex(r, 0.0, -1.0, step, C.byref(state), C.byref(last), blob.ctypes.data + filled,
                                                cap - filled, C.byref(used), C.addressof(frames) + n * C.sizeof(CXtcFrame),
                                                box.ctypes.data + 36 * n, t.ctypes.data + 4 * n, 8 - n, 1, None, 0, None)
            if got == -4:
                assert n > 0, "a frame must fit an empty blob"
                break
            assert got >= 0, got
            if got == 0:
                lib.gorder_xtc_close(r)
                r = None
                f += 1
                continue
            n += got
            filled += used.value
        got_times += list(t[:n])
        batches.append(n)
    np.testing.assert_array_equal(np.array(got_times, np.float32), want)
    assert max(batches) <= int(blob_frames) + 1 and len(batches) > 1


def test_a_block_longer_than_the_file_is_a_format_error(built, tmp_path):
    path = synthetic(tmp_path, n_frames=3)
    raw = bytearray(open(path, "rb").read())
    raw[88:92] = (1 << 30).to_bytes(4, "big")          # byte count of the first frame's bit stream
    bad = str(tmp_path / "bad.xtc")
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(IOError, match="-2"):
        xtc.pack_trajectory([bad], chunk=4, blob_capacity=1 << 20)


def test_mapped_copies_equal_pread_copies(built, tmp_path, monkeypatch):
    """The block copies go through a mapping of the file with streaming stores when the destination is 32-byte aligned
    (the driver's pinned blob is); GORDER_XTC_PREAD=1 takes the pread route: the same bytes, the same tables."""
    import ctypes as C
    from gorder_amd.abi import CXtcFrame
    path = synthetic(tmp_path, n_frames=9, n_atoms=1500)
    lib = xtc._lib()

    def pack(aligned):
        r = C.c_void_p()
        assert lib.gorder_xtc_open(path.encode(), None, 0, C.byref(r)) == 0
        raw = np.zeros((1 << 20) + 64, np.uint8)
        shift = (-raw.ctypes.data) % 64 + (0 if aligned else 1)
        blob = raw[shift:shift + (1 << 20)]
        assert (blob.ctypes.data % 32 == 0) == aligned
        state, last, used = C.c_uint64(0), C.c_double(float("-inf")), C.c_uint64(0)
        frames = (CXtcFrame * 16)()
        box, t = np.empty((16, 9), np.float32), np.empty(16, np.float32)
        got = lib.gorder_xtc_pack_window(r, 0.0, -1.0, 1, C.byref(state), C.byref(last), blob.ctypes.data, 1 << 20, C.byref(used),
                                         C.cast(frames, C.c_void_p), box.ctypes.data, t.ctypes.data, 16, 3)
        lib.gorder_xtc_close(r)
        assert got == 9
        return bytes(blob[:used.value]), bytes(frames)[: 9 * C.sizeof(CXtcFrame)]

    mapped = pack(True)
    unaligned = pack(False)            # falls back to pread block by block
    monkeypatch.setenv("GORDER_XTC_PREAD", "1")
    plain = pack(True)
    assert mapped == plain == unaligned


def test_a_file_that_grows_or_shrinks_behind_its_mapping(built, tmp_path):
    """The reader maps the file at its first window.  Frames APPENDED later (a running simulation) lie behind the mapping
    and must still be found (by pread); a file that has become SHORTER must not be touched through the mapping again
    (SIGBUS) — the reader reads what is left by pread and reports the end, or a format error, as a status."""
    import ctypes as C
    import shutil
    from gorder_amd.abi import CXtcFrame
    whole = synthetic(tmp_path, n_frames=9, n_atoms=1500, name="whole.xtc")
    raw = open(whole, "rb").read()
    ws = xtc.pack_trajectory([whole], chunk=64)
    ends = [int(p) for p in np.cumsum([len(raw) // 9] * 9)]          # (all frames of this file have the same size or nearly)
    # exact frame boundaries: the reader tells them (file positions of the packed frames' headers)
    lib = xtc._lib()
    growing = str(tmp_path / "growing.xtc")

    def window(r, want_state):
        raw_blob = np.zeros((1 << 21) + 64, np.uint8)
        blob = raw_blob[(-raw_blob.ctypes.data) % 64:][: 1 << 21]
        used = C.c_uint64(0)
        frames = (CXtcFrame * 16)()
        box, t = np.empty((16, 9), np.float32), np.empty(16, np.float32)
        got = lib.gorder_xtc_pack_window(r, 0.0, -1.0, 1, C.byref(want_state[0]), C.byref(want_state[1]), blob.ctypes.data, 1 << 21,
                                         C.byref(used), C.cast(frames, C.c_void_p), box.ctypes.data, t.ctypes.data, 16, 2)
        return got, t[:max(got, 0)].copy()

    # where frame 5 starts: the size of a file that holds the first five frames
    five = str(tmp_path / "five.xtc")
    x, b, t = xtc.read_trajectory([whole])
    xtc.write_trajectory(five, x[:5], b[:5], times=t[:5], precision=1000.0)
    n5 = os.path.getsize(five)
    assert raw[:n5] == open(five, "rb").read()
    shutil.copy(five, growing)
    r = C.c_void_p()
    assert lib.gorder_xtc_open(growing.encode(), None, 0, C.byref(r)) == 0
    st = (C.c_uint64(0), C.c_double(float("-inf")))
    got, tt = window(r, st)
    assert got == 5 and list(tt) == [0.0, 10.0, 20.0, 30.0, 40.0]     # (the mapping now covers these five frames)
    with open(growing, "ab") as f:                                    # four more frames arrive
        f.write(raw[n5:])
    got, tt = window(r, st)
    assert got == 4 and list(tt) == [50.0, 60.0, 70.0, 80.0]
    assert window(r, st)[0] == 0
    lib.gorder_xtc_close(r)
    # ... and a file cut short after the mapping was made
    shrinking = str(tmp_path / "shrinking.xtc")
    shutil.copy(whole, shrinking)
    r = C.c_void_p()
    assert lib.gorder_xtc_open(shrinking.encode(), None, 0, C.byref(r)) == 0
    st = (C.c_uint64(0), C.c_double(float("-inf")))
    used = C.c_uint64(0)
    frames = (CXtcFrame * 16)()
    box, t2 = np.empty((16, 9), np.float32), np.empty(16, np.float32)
    raw_blob = np.zeros((1 << 21) + 64, np.uint8)
    blob = raw_blob[(-raw_blob.ctypes.data) % 64:][: 1 << 21]
    assert lib.gorder_xtc_pack_window(r, 0.0, -1.0, 1, C.byref(st[0]), C.byref(st[1]), blob.ctypes.data, 1 << 21, C.byref(used),
                                      C.cast(frames, C.c_void_p), box.ctypes.data, t2.ctypes.data, 3, 2) == 3      # maps all nine
    os.truncate(shrinking, n5)                                        # frames 5.. are gone
    got, tt = window(r, st)
    assert got == 2 and list(tt) == [30.0, 40.0]                      # what is left, by pread; no fault
    assert window(r, st)[0] == 0
    lib.gorder_xtc_close(r)


def test_pool_copies_the_same_bytes(built, tmp_path):
    a = synthetic(tmp_path, n_frames=29, name="a.xtc")
    b = synthetic(tmp_path, n_frames=11, seed=5, n_atoms=700, name="b.xtc")
    plain = xtc.pack_trajectory([a, b, CG3] if False else [a, b], chunk=8, threads=3)
    pooled = xtc.pack_trajectory([a, b], chunk=8, threads=3, pool=True)
    assert len(plain) == len(pooled)
    for x, y in zip(plain, pooled):
        assert bytes(x["blob"]) == bytes(y["blob"]) and x["frames"].tobytes() == y["frames"].tobytes()
        np.testing.assert_array_equal(x["time"], y["time"])
        check_layout(y)


def test_pack_concatenation_and_group(built, tmp_path):
    a = synthetic(tmp_path, n_frames=6, name="a.xtc")
    # the second file starts with the last time of the first (a duplicate boundary frame, CHANGELOG.md:64)
    rng = np.random.default_rng(9)
    xyz = rng.uniform(0, 6, size=(4, 700, 3)).astype(np.float32)
    box = np.tile(np.eye(3, dtype=np.float32) * 6.0, (4, 1, 1))
    b = str(tmp_path / "b.xtc")
    xtc.write_trajectory(b, xyz, box, times=np.array([50.0, 60.0, 70.0, 80.0], np.float32))
    group = np.array([5, 17, 300, 311], dtype=np.uint32)
    x, bx, t = xtc.read_trajectory([a, b], group=group)
    ws = xtc.pack_trajectory([a, b], group=group, chunk=4)
    np.testing.assert_array_equal(np.concatenate([w["time"] for w in ws]), t)
    assert len(t) == 9
    w = ws[0]
    assert w["n_stop"] == 312 and w["slot_of"].shape == (700,)
    assert list(np.flatnonzero(w["slot_of"] >= 0)) == [5, 17, 300, 311]


def test_probe(built, tmp_path):
    import ctypes as C
    lib = xtc._lib()
    n, size, first = C.c_uint32(), C.c_uint64(), C.c_uint32()
    assert lib.gorder_xtc_probe(CG3.encode(), C.byref(n), C.byref(size), C.byref(first)) == 1
    assert n.value == 16769 and size.value == os.path.getsize(CG3) == first.value        # one frame: the file IS that frame
    path = synthetic(tmp_path, n_frames=4)
    assert lib.gorder_xtc_probe(path.encode(), C.byref(n), C.byref(size), C.byref(first)) == 1
    assert n.value == 700 and 0 < first.value < size.value == os.path.getsize(path)
    other = str(tmp_path / "not.xtc")
    open(other, "wb").write(b"title\n   12\n" + bytes(100))
    assert lib.gorder_xtc_probe(other.encode(), None, None, None) == 0
    assert lib.gorder_xtc_probe(str(tmp_path / "missing.xtc").encode(), None, None, None) < 0


def test_pack_reference_file(built):
    """tests/golden/cg3.xtc (the reference's tests/files/split/cg3.xtc): header fields as the format defines them."""
    ws = xtc.pack_trajectory([CG3])
    assert len(ws) == 1 and len(ws[0]["time"]) == 1
    f = ws[0]["frames"][0]
    assert f["kind"] == 0 and 9 <= f["smallidx"] < 73 and f["bitsize"] > 0
    assert f["inv_precision"] == np.float32(1.0) / np.float32(1.0 / f["inv_precision"])
    assert int(f["n_bytes"]) + 92 == os.path.getsize(CG3)      # one frame: header + block
    for k in (1, 2):
        s = int(f["sizeint"][k])
        assert int(f[f"recip{k}"]) == (2 ** 64 // s if s > 1 else 2 ** 64 - 1)
