"""k_xtc_decode (XTC frames decompressed on the device, one frame per lane) against the host decoder
gorder_xtc_next, bit for bit — reference files, files of the repo's encoder at several precisions and box sizes
(every field-width path of the format), groups with early stop, corrupt blocks — and the trajectory driver with
device_decode against the host-decode route."""
import os

import numpy as np
import pytest
import torch

from gorder_amd import HipEngine, abi, xtc
from golden_util import GOLDEN, METHODS, Fixture, cg_setup

pytestmark = pytest.mark.gpu
CG3 = os.path.join(GOLDEN, "cg3.xtc")


@pytest.fixture(scope="module")
def cg(built):
    return Fixture("cg")


@pytest.fixture(scope="module")
def engine(built):
    mt = abi.MolType(n_molecules=1, bonds=np.array([[[0, 1]]], dtype=np.uint32))
    return HipEngine(abi.Tables(n_atoms=2, molecule_types=[mt]))


def device_decode(engine, paths, group=None, chunk=64, **kw):
    """-> xyz [F, n_out, 3] decoded by the device from packed windows"""
    out = []
    for w in xtc.pack_trajectory(paths, group=group, chunk=chunk, threads=2, **kw):
        n = len(w["time"])
        n_out = w["n_atoms_file"] if group is None else len(group)
        blob = torch.from_numpy(w["blob"]).cuda()
        frames = torch.from_numpy(w["frames"].view(np.uint8).reshape(-1).copy()).cuda()
        slot = None if w["slot_of"] is None else torch.from_numpy(w["slot_of"]).cuda()
        xyz = torch.full((n, n_out, 3), float("nan"), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        engine.xtc_decode(blob.data_ptr(), blob.numel(), frames.data_ptr(), n, w["n_atoms_file"],
                          0 if slot is None else slot.data_ptr(), w["n_stop"], xyz.data_ptr(), n_out)
        engine.synchronize()
        out.append(xyz.cpu().numpy())
    return np.concatenate(out) if out else np.zeros((0, 0, 3), np.float32)


def same_bits(a, b):
    assert a.shape == b.shape
    np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))


def clustered(rng, n_atoms, n_frames, span, sigma=0.04, jitter=0.02):
    """triplets of atoms close together (water-like: the encoder's runs of small offsets) + slow motion"""
    centres = rng.uniform(0.0, span, size=(n_atoms // 3 + 1, 1, 3))
    x0 = (centres + rng.normal(0.0, sigma, size=(n_atoms // 3 + 1, 3, 3))).reshape(-1, 3)[:n_atoms]
    return (x0[None] + rng.normal(0.0, jitter, size=(n_frames, n_atoms, 3))).astype(np.float32)


def write(tmp_path, name, xyz, span, precision):
    box = np.tile(np.eye(3, dtype=np.float32) * np.float32(span), (xyz.shape[0], 1, 1))
    path = str(tmp_path / name)
    xtc.write_trajectory(path, xyz, box, times=np.arange(xyz.shape[0], dtype=np.float32), precision=precision)
    return path


def test_reference_file(engine):
    host = xtc.read_trajectory([CG3])[0]
    same_bits(device_decode(engine, [CG3]), host)
    group = np.arange(100, 6000, 7, dtype=np.uint32)
    same_bits(device_decode(engine, [CG3], group=group), xtc.read_trajectory([CG3], group=group)[0])


@pytest.mark.parametrize("name", ["pcpepg4.xtc", "multiple_resid_same_name.xtc"])
def test_more_reference_files(engine, name):
    """an all-atom frame with its water (68 375 atoms, long runs of small offsets) and a small multi-frame system, both
    from the reference's tests/files: whole frames and a scattered group"""
    path = os.path.join(GOLDEN, name)
    host = xtc.read_trajectory([path])[0]
    assert host.shape[0] >= 1
    same_bits(device_decode(engine, [path]), host)
    n = host.shape[1]
    group = np.unique(np.random.default_rng(4).integers(0, n, size=max(3, n // 7))).astype(np.uint32)[::-1].copy()
    same_bits(device_decode(engine, [path], group=group), xtc.read_trajectory([path], group=group)[0])


@pytest.mark.parametrize("precision", [10.0, 100.0, 1000.0, 12345.0])
def test_encoder_files(engine, tmp_path, precision):
    rng = np.random.default_rng(int(precision))
    xyz = clustered(rng, 4001, 150, 7.5)                     # 150 frames: three waves, the last one partly idle
    path = write(tmp_path, "p.xtc", xyz, 7.5, precision)
    host = xtc.read_trajectory([path], threads=4)[0]
    same_bits(device_decode(engine, [path], chunk=150), host)
    same_bits(device_decode(engine, [path], chunk=37, begin=10.0, end=120.0, step=3),
              xtc.read_trajectory([path], begin=10.0, end=120.0, step=3)[0])


def test_every_field_width(engine, tmp_path):
    """one mixed-radix number of <= 64 bits (the usual case), of 65..72 bits (edges beyond ~2 million grid steps),
    and three separate fields (an edge beyond 2^24 grid steps); gases (no runs) and dense clusters (long runs);
    smallidx walking up and down"""
    rng = np.random.default_rng(77)
    cases = {
        "wide": (clustered(rng, 900, 70, 5000.0), 5000.0, 1000.0),            # 5e6 steps per edge: 3 x 22.3 bits = 67
        "split": (clustered(rng, 900, 70, 20000.0), 20000.0, 1000.0),         # 2e7 steps > 0xffffff
        "gas": (rng.uniform(0, 30.0, size=(70, 1500, 3)).astype(np.float32), 30.0, 1000.0),
        "dense": (clustered(rng, 3000, 70, 2.0, sigma=0.003, jitter=0.001), 2.0, 1000.0),
        "mixed": (np.concatenate([clustered(rng, 600, 70, 8.0, sigma=0.2), clustered(rng, 600, 70, 8.0, sigma=0.002),
                                  rng.uniform(0, 8.0, size=(70, 300, 3)).astype(np.float32)], axis=1), 8.0, 1000.0),
        "negative": (clustered(rng, 900, 70, 6.0) - 40.0, 6.0, 500.0),        # minint < 0
    }
    for name, (xyz, span, prec) in cases.items():
        path = write(tmp_path, name + ".xtc", xyz, span, prec)
        host = xtc.read_trajectory([path])[0]
        packed = xtc.pack_trajectory([path], chunk=70)[0]["frames"]
        if name == "wide":
            assert packed["bitsize"].min() > 64
        if name == "split":
            assert packed["bitsize"].max() == 0 and packed["bitsizeint"].min() > 0
        same_bits(device_decode(engine, [path], chunk=70), host)


def test_many_shapes(engine, tmp_path):
    """sizes around every boundary of the kernel: atoms per frame just above the raw-float limit and around the flush
    group of 8, frames per launch around a wave of 64, groups whose last atom (= where decoding stops, one atom past
    it for the swap) sits at every position of a flush group, with and without runs of small offsets"""
    rng = np.random.default_rng(2024)
    for k in range(28):
        n_atoms = int(rng.choice([10, 11, 15, 16, 17, 23, 24, 25, 31, 33, 63, 64, 65, 100, 257]))
        n_frames = int(rng.choice([1, 2, 63, 64, 65, 130]))
        span = float(rng.choice([1.5, 6.0, 40.0]))
        xyz = clustered(rng, n_atoms, n_frames, span, sigma=float(rng.choice([0.002, 0.05, 0.5])))
        path = write(tmp_path, f"s{k}.xtc", xyz, span, float(rng.choice([50.0, 1000.0])))
        host = xtc.read_trajectory([path])[0]
        same_bits(device_decode(engine, [path], chunk=n_frames), host)
        last = int(rng.integers(0, n_atoms))
        group = np.unique(np.concatenate([rng.integers(0, last + 1, size=max(1, last // 3)), [last]])).astype(np.uint32)
        rng.shuffle(group)
        same_bits(device_decode(engine, [path], group=group, chunk=n_frames), xtc.read_trajectory([path], group=group)[0])


def test_small_systems_are_raw_floats(engine, tmp_path):
    rng = np.random.default_rng(5)
    xyz = rng.normal(0, 3, size=(9, 7, 3)).astype(np.float32)
    path = write(tmp_path, "tiny.xtc", xyz, 4.0, 1000.0)
    got = device_decode(engine, [path])
    same_bits(got, xtc.read_trajectory([path])[0])
    same_bits(got, xyz)                                       # raw frames are exact
    group = np.array([6, 1], dtype=np.uint32)
    same_bits(device_decode(engine, [path], group=group), xyz[:, group])


def test_group_order_and_early_stop(engine, tmp_path):
    rng = np.random.default_rng(11)
    xyz = clustered(rng, 3000, 80, 6.0)
    path = write(tmp_path, "g.xtc", xyz, 6.0, 1000.0)
    for group in (np.array([2999], np.uint32), np.array([0], np.uint32), rng.permutation(3000)[:500].astype(np.uint32),
                  np.arange(1200, dtype=np.uint32)[::-1].copy()):
        same_bits(device_decode(engine, [path], group=group, chunk=80), xtc.read_trajectory([path], group=group)[0])


def test_corrupt_block_is_reported(built, tmp_path):
    rng = np.random.default_rng(13)
    path = write(tmp_path, "c.xtc", clustered(rng, 2000, 5, 6.0), 6.0, 1000.0)
    w = xtc.pack_trajectory([path])[0]
    for damage in ("truncated", "smallidx", "offset", "bitsize", "bitsizeint", "no bits"):
        mt = abi.MolType(n_molecules=1, bonds=np.array([[[0, 1]]], dtype=np.uint32))
        eng = HipEngine(abi.Tables(n_atoms=2, molecule_types=[mt]))
        fr = w["frames"].copy()
        if damage == "truncated":
            fr["n_bytes"][3] = 400                          # the stream ends long before the last atom
        elif damage == "smallidx":
            fr["smallidx"][3] = 80
        elif damage == "bitsize":                           # widths no writer produces (the table is the caller's):
            fr["bitsize"][3] = 200                          # 63 groups of them would not fit the scan's window
        elif damage == "bitsizeint":
            fr["bitsize"][3] = 0
            fr["bitsizeint"][3] = 255 | (255 << 8) | (255 << 16)
        elif damage == "no bits":
            fr["bitsize"][3] = 0
            fr["bitsizeint"][3] = 0
        else:
            fr["offset"][3] = w["blob"].size                # outside the blob: nothing may be read
        blob = torch.from_numpy(w["blob"]).cuda()
        frames = torch.from_numpy(fr.view(np.uint8).reshape(-1).copy()).cuda()
        out = torch.zeros((5, 2000, 3), dtype=torch.float32, device="cuda")
        eng.xtc_decode(blob.data_ptr(), blob.numel(), frames.data_ptr(), 5, 2000, 0, 2000, out.data_ptr(), 2000)
        with pytest.raises(abi.GorderHipError) as ei:
            eng.synchronize()
        assert ei.value.status == abi.ERR_TRAJECTORY_FORMAT
        assert "frame 3" in str(ei.value)


@pytest.mark.parametrize("batch_frames,threads", [(0, 4), (16, 2), (1, 1)])
def test_driver_device_decode_matches_host_decode(cg, tmp_path, batch_frames, threads):
    """the whole pipeline (pack -> copy -> k_xtc_decode -> analysis) against the host-decode pipeline: 101 CG frames
    in two files with a duplicate boundary frame, a time window and a step"""
    tables, labels, midx = cg_setup(cg, leaflets=METHODS["global"])
    a, b = str(tmp_path / "a.xtc"), str(tmp_path / "b.xtc")
    xtc.write_trajectory(a, cg.xyz[:60], cg.boxes[:60], times=cg.times[:60], precision=100.0)
    xtc.write_trajectory(b, cg.xyz[59:], cg.boxes[59:], times=cg.times[59:], precision=100.0)
    for kw in (dict(), dict(begin=float(cg.times[7]), end=float(cg.times[90]), step=4)):
        host = HipEngine(tables)
        s0 = host.run_trajectory([a, b], group=midx, threads=threads, batch_frames=batch_frames, **kw)
        dev = HipEngine(tables)
        s1 = dev.run_trajectory([a, b], group=midx, threads=threads, batch_frames=batch_frames, device_decode=True, **kw)
        assert s0["device_decode"] == 0 and s1["device_decode"] == 1
        assert s0["n_frames"] == s1["n_frames"] and s1["n_frames"] in (101, 21)
        assert s1["bytes_h2d"] < s0["bytes_h2d"]
        r0, r1 = host.finish(), dev.finish()
        assert r0.n_frames == r1.n_frames
        np.testing.assert_array_equal(r0.sums, r1.sums)
        np.testing.assert_array_equal(r0.counts, r1.counts)


def test_driver_batches_that_end_because_the_blob_is_full(built, tmp_path):
    """frames that compress worse than the blob is sized for (a gas at high precision: 6.5 bytes per atom): a batch
    ends when its blob is full, the frame that did not fit opens the next one — none is lost, none taken twice"""
    rng = np.random.default_rng(21)
    n_atoms, n_frames = 2000, 50
    xyz = rng.uniform(0.0, 30.0, size=(n_frames, n_atoms, 3)).astype(np.float32)
    path = write(tmp_path, "gas.xtc", xyz, 30.0, 5000.0)
    assert os.path.getsize(path) > n_frames * n_atoms * 6.3
    bonds = np.arange(n_atoms, dtype=np.uint32).reshape(1, n_atoms // 2, 2)
    tables = abi.Tables(n_atoms=n_atoms, molecule_types=[abi.MolType(n_molecules=n_atoms // 2, bonds=bonds)])
    res = {}
    for dev in (False, True):
        eng = HipEngine(tables)
        st = eng.run_trajectory([path, path], threads=2, batch_frames=16, device_decode=dev, end=1e9)
        assert st["n_frames"] == 2 * n_frames and st["device_decode"] == int(dev)
        if dev:
            assert st["n_batches"] > (2 * n_frames + 15) // 16                      # batches were cut short by the blob
        res[dev] = eng.finish()
    np.testing.assert_array_equal(res[False].sums, res[True].sums)
    np.testing.assert_array_equal(res[False].counts, res[True].counts)


def _membrane_with_water(tmp_path, n_frames=150, seed=5):
    """a 'membrane' of 600 bonded atoms in front of 4 200 'water' atoms: decoding stops a seventh into every frame"""
    rng = np.random.default_rng(seed)
    n_sel, n_w = 600, 4200
    xyz = np.concatenate([clustered(rng, n_sel, n_frames, 6.0, sigma=0.05), clustered(rng, n_w, n_frames, 6.0, sigma=0.04)], axis=1)
    path = write(tmp_path, "mw.xtc", xyz, 6.0, 1000.0)
    bonds = np.arange(n_sel, dtype=np.uint32).reshape(1, n_sel // 2, 2)
    tables = abi.Tables(n_atoms=n_sel, molecule_types=[abi.MolType(n_molecules=n_sel // 2, bonds=bonds)])
    return path, tables, np.arange(n_sel, dtype=np.uint32)


@pytest.mark.parametrize("mode", ["adaptive", "forced tiny", "forced half", "off"])
def test_driver_copies_only_the_part_of_a_frame_it_needs(built, tmp_path, monkeypatch, mode):
    """The analysed atoms come first in a frame and the decoder stops behind them: after the first batches the driver
    copies only the leading part of every block that the decoder reported it needed (+ margin).  A frame for which
    that was too little is reported SHORT and decoded by the host — forced here with a fixed, far too small part — and
    whatever happens the result is the host route's, bit for bit."""
    path, tables, group = _membrane_with_water(tmp_path)
    host = HipEngine(tables)
    s0 = host.run_trajectory([path] * 6, group=group, threads=2, batch_frames=40)
    want = host.finish()
    if mode == "forced tiny":
        monkeypatch.setenv("GORDER_HIP_PREFIX_Q16", "2000")        # 3 % of a block: every frame comes out short
    if mode == "forced half":
        monkeypatch.setenv("GORDER_HIP_PREFIX_Q16", "32768")       # plenty: the analysed atoms end a seventh into the frame
    if mode == "off":
        monkeypatch.setenv("GORDER_HIP_NO_PREFIX", "1")
    eng = HipEngine(tables)
    s1 = eng.run_trajectory([path] * 6, group=group, threads=2, batch_frames=40, device_decode=True)
    got = eng.finish()
    assert s1["device_decode"] == 1 and s1["n_frames"] == s0["n_frames"] == 900
    np.testing.assert_array_equal(got.sums, want.sums)
    np.testing.assert_array_equal(got.counts, want.counts)
    full = os.path.getsize(path) * 6
    if mode == "forced tiny":
        assert s1["frames_decoded_by_host"] == 900
    elif mode == "forced half":
        assert s1["frames_decoded_by_host"] == 0 and s1["bytes_h2d"] < 0.62 * full
    elif mode == "off":
        assert s1["frames_decoded_by_host"] == 0 and s1["bytes_h2d"] > 0.95 * full
    else:
        assert s1["frames_decoded_by_host"] == 0 and s1["bytes_h2d"] < 0.75 * full      # the first batches travel whole


def test_short_frames_through_the_plain_call_are_errors(engine, tmp_path):
    """gorder_hip_xtc_decode has no list to put a short frame on: a frame packed with too small a part is reported
    like a truncated one"""
    path, tables, group = _membrane_with_water(tmp_path, n_frames=3)
    lib = xtc._lib()
    import ctypes as C
    r = C.c_void_p()
    assert lib.gorder_xtc_open(path.encode(), None, 0, C.byref(r)) == 0
    state, last, used = C.c_uint64(0), C.c_double(float("-inf")), C.c_uint64(0)
    blob = np.zeros(1 << 20, np.uint8)
    frames = (abi.CXtcFrame * 3)()
    box, t, pos = np.empty((3, 9), np.float32), np.empty(3, np.float32), np.zeros(3, np.int64)
    lib.gorder_xtc_pack_window_ex.restype = C.c_int64
    lib.gorder_xtc_pack_window_ex.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_uint32, C.POINTER(C.c_uint64),
                                              C.POINTER(C.c_double), C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    got = lib.gorder_xtc_pack_window_ex(r, 0.0, -1.0, 1, C.byref(state), C.byref(last), blob.ctypes.data, blob.size, C.byref(used),
                                        C.cast(frames, C.c_void_p), box.ctypes.data, t.ctypes.data, 3, 1, None, 1000, pos.ctypes.data)
    lib.gorder_xtc_close(r)
    assert got == 3 and all(f.kind & 2 for f in frames) and pos[0] == 0 and pos[1] > 0
    fr = np.frombuffer(frames, dtype=np.dtype(abi.CXtcFrame)).copy()
    d_blob = torch.from_numpy(blob[:used.value].copy()).cuda()
    d_frames = torch.from_numpy(fr.view(np.uint8).reshape(-1).copy()).cuda()
    out = torch.zeros((3, 4800, 3), dtype=torch.float32, device="cuda")
    mt = abi.MolType(n_molecules=1, bonds=np.array([[[0, 1]]], dtype=np.uint32))
    eng = HipEngine(abi.Tables(n_atoms=2, molecule_types=[mt]))
    eng.xtc_decode(d_blob.data_ptr(), d_blob.numel(), d_frames.data_ptr(), 3, 4800, 0, 4800, out.data_ptr(), 4800)
    with pytest.raises(abi.GorderHipError) as ei:
        eng.synchronize()
    assert ei.value.status == abi.ERR_TRAJECTORY_FORMAT


def test_staging_is_kept_and_released(cg, tmp_path):
    """a handle's second run finds its staging buffers (no setup to speak of), runs of another shape and
    release_staging start over; results do not depend on any of it"""
    tables, labels, midx = cg_setup(cg, leaflets=METHODS["global"])
    path = str(tmp_path / "c.xtc")
    xtc.write_trajectory(path, cg.xyz, cg.boxes, times=cg.times, precision=100.0)
    eng = HipEngine(tables)
    want = None
    for kw in (dict(device_decode=True), dict(device_decode=True), dict(device_decode=False), dict(device_decode=True, batch_frames=9),
               dict(device_decode=True, batch_frames=9), "release", dict(device_decode=True, batch_frames=9)):
        if kw == "release":
            eng.release_staging()
            continue
        eng.reset()
        stats = eng.run_trajectory([path], group=midx, threads=2, **kw)
        res = eng.finish()
        assert stats["n_frames"] == 101
        if want is None:
            want = res.sums.copy()
        np.testing.assert_array_equal(res.sums, want)


def test_driver_falls_back_for_other_formats(cg, tmp_path):
    """device_decode with a reference XTC file whose group is the Master group, and the fallback to the host decoder
    is silent for non-XTC input (nothing to decompress there)"""
    tables, labels, midx = cg_setup(cg, leaflets=METHODS["global"])
    eng = HipEngine(tables)
    stats = eng.run_trajectory([CG3], group=midx, device_decode=True)
    assert stats["n_frames"] == 1 and stats["device_decode"] == 1
    ref = HipEngine(tables)
    ref.run_trajectory([CG3], group=midx)
    np.testing.assert_array_equal(eng.finish().sums, ref.finish().sums)


@pytest.mark.parametrize("damage", ["zeros", "truncated"])
def test_driver_reports_corrupt_files_on_both_routes(cg, tmp_path, damage):
    """a frame whose bit stream is all zeros (every atom at full width: the block is too short for that) and a file cut
    in the middle of a frame: GORDER_ERR_TRAJECTORY_FORMAT whichever side decodes"""
    tables, labels, midx = cg_setup(cg, leaflets=METHODS["global"])
    path = str(tmp_path / "bad.xtc")
    xtc.write_trajectory(path, cg.xyz[:12], cg.boxes[:12], times=cg.times[:12], precision=100.0)
    w = xtc.pack_trajectory([path])[0]
    raw = bytearray(open(path, "rb").read())
    sizes = [92 + int(f["n_bytes"]) for f in w["frames"]]
    start7 = sum(sizes[:7])
    if damage == "zeros":
        raw[start7 + 92:start7 + sizes[7]] = bytes(sizes[7] - 92)
    else:
        raw = raw[:start7 + sizes[7] // 2]
    open(path, "wb").write(bytes(raw))
    for dev in (False, True):
        eng = HipEngine(tables)
        with pytest.raises(abi.GorderHipError) as ei:
            eng.run_trajectory([path], group=midx, threads=2, device_decode=dev)
        assert ei.value.status == abi.ERR_TRAJECTORY_FORMAT, (dev, str(ei.value))
