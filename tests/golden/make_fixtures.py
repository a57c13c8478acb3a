#!/usr/bin/env python3
"""Generate the committed golden fixtures from the reference's own test DATA files.

    python tests/golden/make_fixtures.py        (needs /root/reference; run once, output is committed)

Only data travels: lipid-only subsets of the reference's structure / bond / trajectory fixtures
(re-packed as integer coordinates exactly as stored in the XTC files), the expected-output text files of
its integration tests, and the literal expectation arrays (numbers) of two of its unit tests
(single_frame_kats below).  No reference code is copied.

  pcpepg  tests/files/pcpepg.gro + pcpepg.bnd + split/pcpepg{1..5}.xtc   (tests_aa.rs:47-77)
  cg      tests/files/cg.gro + cg.bnd + split/cg{1..5}.xtc               (tests_cg.rs:46-66)
  ua      tests/files/ua_nobox.pdb (names, CONECT bonds) + ua.xtc        (tests_ua.rs:19-68)
"""
import ctypes as C
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/tests/files"

from gorder_amd import structure as st   # noqa: E402
from gorder_amd import xtc              # noqa: E402

LIPIDS = {"POPC", "POPE", "POPG", "POPS"}


def pack(name, gro, bnd, xtcs, alt_bnd=None, same_frames=None):
    if gro.endswith(".pdb"):
        s = st.read_pdb(os.path.join(REF, gro))     # CONECT records carry the bonds
        adj = s.bonds
        s.box = np.zeros(3, dtype=np.float32)
    else:
        s = st.read_gro(os.path.join(REF, gro))
        adj = st.read_bnd(os.path.join(REF, bnd), s.n_atoms)
    keep = np.array([r in LIPIDS for r in s.resnames])
    idx = np.flatnonzero(keep)
    remap = -np.ones(s.n_atoms, dtype=np.int64)
    remap[idx] = np.arange(len(idx))
    pairs = sorted({(int(remap[a]), int(remap[b])) for a in idx for b in adj[a] if keep[b] and a < b})
    frames, boxes, times, prec = xtc.read_trajectory([os.path.join(REF, x) for x in xtcs], group=idx.astype(np.uint32),
                                                     return_precision=True)
    ints = np.rint(frames.astype(np.float64) * prec).astype(np.int32)
    back = (ints.astype(np.float32) * np.float32(1.0 / np.float32(prec))).astype(np.float32)
    assert np.array_equal(back, frames), "integer round trip must reproduce the decoder's floats exactly"
    assert np.abs(ints).max() < 32768
    extra = {}
    if alt_bnd:          # a second bond definition of the same system (tests_cg.rs:380-430: .bonds() overrides the structure's)
        adj2 = st.read_bnd(os.path.join(REF, alt_bnd), s.n_atoms)
        extra["bonds_alt"] = np.array(sorted({(int(remap[a]), int(remap[b])) for a in idx for b in adj2[a]
                                              if keep[b] and a < b}), dtype=np.int32)
    if same_frames:      # a shorter trajectory of the reference that holds some of the same frames: keep their indices only
        f2, b2, t2 = xtc.read_trajectory([os.path.join(REF, same_frames)], group=idx.astype(np.uint32))
        where = np.searchsorted(times, t2)
        assert np.array_equal(times[where], t2) and np.array_equal(frames[where], f2) and np.array_equal(boxes[where], b2)
        extra["frames_" + os.path.splitext(same_frames)[0]] = where.astype(np.int32)
    out = os.path.join(HERE, name + ".npz")
    np.savez_compressed(out, **extra, resids=s.resids[idx].astype(np.int32), resnames=np.array([s.resnames[i] for i in idx]),
                        names=np.array([s.names[i] for i in idx]), bonds=np.array(pairs, dtype=np.int32),
                        structure_box=s.box, ints=ints.astype(np.int16), precision=np.float32(prec),
                        boxes=boxes.astype(np.float32), times=times.astype(np.float32))
    print(name, "atoms", len(idx), "bonds", len(pairs), "frames", len(times), "t", times[0], times[-1],
          "->", os.path.getsize(out) // 1024, "KiB")


def tpr_frame(tpr_path, gro_positions, gro_box):
    """The full-precision coordinates and box a GROMACS .tpr file holds, WITHOUT parsing the format: the file stores
    them as big-endian f32 arrays, and its .gro twin prints the same frame rounded to 1e-3 nm (box: 1e-5) — so the
    arrays are the (only) places in the file where 3 N consecutive floats agree with the .gro to half a unit of its
    last digit."""
    raw = open(tpr_path, "rb").read()
    n = len(gro_positions)
    want = np.asarray(gro_positions, dtype=np.float64)
    found_x, found_b = [], []
    bw = np.zeros(9)
    bw[[0, 4, 8]] = gro_box[:3]
    for shift in range(4):                      # newer .tpr bodies are not 4-byte aligned
        m = (len(raw) - shift) // 4
        f = np.frombuffer(raw[shift:shift + 4 * m], dtype=">f4")
        with np.errstate(invalid="ignore"):
            f64 = f.astype(np.float64)
            for k in np.flatnonzero(np.abs(f64 - want[0, 0]) < 5.01e-4):
                if k + 3 * n <= m and np.abs(f64[k:k + 3 * n].reshape(n, 3) - want).max() < 5.01e-4:
                    found_x.append(f[k:k + 3 * n].astype(np.float32).reshape(n, 3))
            for k in np.flatnonzero(np.abs(f64 - bw[0]) < 5.01e-6):
                if k + 9 <= m and np.abs(f64[k:k + 9] - bw).max() < 5.01e-6:
                    found_b.append(f[k:k + 9].astype(np.float32))
    assert len(found_x) == 1 and len(found_b) >= 1, (len(found_x), len(found_b))
    assert all(np.array_equal(b, found_b[0]) for b in found_b)
    return found_x[0], found_b[0][[0, 4, 8]]


def single_frame_kats():
    """The literal expectation arrays of the reference's single-frame tests (aaorder.rs:226-464, cgorder.rs:188-351:
    per bond type the SUM of the order parameters of one frame — the structure file's coordinates — total / upper /
    lower) + those coordinates: as the .gro twins of the .tpr files print them (1e-3 nm) and at the .tpr's own full
    f32 precision (tpr_frame), which is what the reference's tests run on."""
    import json
    import re
    out = {}
    for name, rs, gro, tag in (("aa", "aaorder.rs", "pcpepg.gro", "pcpepg"), ("cg", "cgorder.rs", "cg.gro", "cg")):
        src = open(os.path.join(REF, "..", "..", "src", "analysis", rs)).read()
        d = {}
        for fn in ("expected_total_orders", "expected_upper_orders", "expected_lower_orders"):
            body = src[src.index(f"fn {fn}()"):]
            body = body[:body.index("\n    }\n")]
            vecs = re.findall(r"vec!\[(.*?)\]", body, re.S)
            d[fn.replace("expected_", "").replace("_orders", "")] = [
                [float(x) for x in v.replace("\n", " ").split(",") if x.strip()] for v in vecs]
        out[name] = d
        g = st.read_gro(os.path.join(REF, gro))
        keep = np.array([r in LIPIDS for r in g.resnames])
        ints = np.rint(g.positions[keep].astype(np.float64) * 1000).astype(np.int32)
        x32, b32 = tpr_frame(os.path.join(REF, tag + ".tpr"), g.positions, g.box)
        np.savez_compressed(os.path.join(HERE, f"{tag}_structure_frame.npz"), ints=ints.astype(np.int16),
                            box=np.asarray(g.box, dtype=np.float32), xyz_tpr=x32[keep], box_tpr=b32)
    with open(os.path.join(HERE, "expected", "single_frame_sums.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    single_frame_kats()
    pack("pcpepg", "pcpepg.gro", "pcpepg.bnd", [f"split/pcpepg{i}.xtc" for i in range(1, 6)],
         same_frames="pcpepg_selected.xtc")
    pack("cg", "cg.gro", "cg.bnd", [f"split/cg{i}.xtc" for i in range(1, 6)], alt_bnd="cg_redefined.bnd")
    pack("ua", "ua_nobox.pdb", None, ["ua.xtc"])      # tests_ua.rs:19-68 (names + bonds from the PDB twin of ua.tpr)
    pack("ua_nobox", "ua_nobox.pdb", None, ["ua_whole_nobox.xtc"])   # tests_ua.rs:686-714: molecules whole, no box, handle_pbc(false)
    for f in ("aa_order_basic.yaml", "aa_order_begin_end_step.yaml", "aa_order_leaflets.yaml",
              "cg_order_basic.yaml", "cg_order_begin_end_step.yaml", "cg_order_leaflets.yaml",
              "ua_order_basic.yaml", "ua_order_leaflets.yaml",
              "aa_order_cuboid_square.yaml", "aa_order_cylinder.yaml", "aa_order_sphere_static.yaml",
              "aa_order_cuboid_patch.yaml", "aa_order_cylinder_x.yaml", "aa_order_sphere_center.yaml",
              "aa_order_cuboid_dynamic.yaml", "aa_order_cylinder_dynamic.yaml", "aa_order_sphere_dynamic.yaml",
              "aa_order_leaflets_dynamic.yaml", "cg_order_leaflets_dynamic.yaml", "ua_order_dynamic_normals.yaml",
              "aa_leaflets_every1.yaml", "aa_order_error.yaml", "aa_order_error_leaflets.yaml",
              "cg_order_error.yaml", "cg_order_error_leaflets.yaml",
              "aa_order_limit.yaml", "aa_order_leaflets_limit.yaml", "aa_order_step.yaml", "aa_order_begin_end.yaml",
              "aa_order_cylinder_z_inverted.yaml",
              "aa_order_sphere_dynamic_inverted.yaml", "aa_order_error_blocks10.yaml", "aa_order_error_limit.yaml",
              "cg_order_cuboid_square.yaml", "cg_order_cylinder.yaml", "cg_order_cylinder_z_inverted.yaml",
              "cg_order_begin_end.yaml", "cg_order_limit.yaml", "ua_order_leaflets_nopbc.yaml",
              # the wider sweep of tests/golden_cases.py
              "aa_order_small.yaml", "aa_order_leaflets_small.yaml", "aa_order_selected.yaml", "aa_leaflets_once.yaml",
              "aa_leaflets_every5.yaml", "aa_order_error_leaflets_limit.yaml",
              "cg_order_small.yaml", "cg_order_leaflets_small.yaml", "cg_order_leaflets_only_upper.yaml",
              "cg_order_redefined_bonds.yaml", "cg_order_sphere.yaml", "cg_order_error_limit.yaml",
              "cg_order_error_leaflets_limit.yaml", "cg_order_leaflets_limit.yaml",
              "ua_order_basic_saturated.yaml", "ua_order_basic_unsaturated.yaml", "ua_order_begin_end_step.yaml",
              "ua_order_cuboid_point.yaml", "ua_order_cylinder_center.yaml", "ua_order_error.yaml",
              "ua_order_leaflets_error.yaml", "ua_order_leaflets_flipped.yaml", "ua_leaflets_once.yaml",
              "ua_normals.yaml", "ua_order_from_aa.yaml",
              # the CSV twins of some of them, for the writers (tests/test_writers_cpu.py)
              "aa_order_basic.csv", "aa_order_leaflets.csv", "cg_order_basic.csv", "cg_order_leaflets.csv",
              "ua_order_basic.csv", "ua_order_leaflets.csv", "aa_order_error.csv", "cg_order_error_leaflets.csv",
              "aa_order_leaflets_limit.csv",
              "aa_order_basic.tab", "aa_order_leaflets.tab", "cg_order_basic.tab", "cg_order_leaflets.tab",
              "ua_order_basic.tab", "ua_order_leaflets.tab", "aa_order_error.tab", "cg_order_error_leaflets.tab",
              "aa_order_basic_POPC.xvg", "aa_order_leaflets_POPC.xvg", "cg_order_leaflets_POPC.xvg",
              "ua_order_leaflets_POPC.xvg",
              "aa_order_convergence.xvg", "aa_order_leaflets_convergence.xvg", "cg_order_convergence.xvg",
              "aa_order_convergence_s5.xvg",
              # united-atom per-frame rows (tests_ua.rs:509-630) and the writers' other united-atom files
              "ua_order_convergence.xvg", "ua_order_leaflets_convergence.xvg", "ua_order_error.csv", "ua_order_error.tab",
              "ua_order_leaflets_error.csv", "ua_order_leaflets_error.tab", "ua_order_basic_POPC.xvg",
              "ua_order_basic_POPS.xvg", "ua_order_leaflets_POPS.xvg",
              # the remaining text twins of analyses this checkout's data reproduces (round 4)
              "aa_order_basic_POPE.xvg", "aa_order_basic_POPG.xvg", "aa_order_leaflets_POPE.xvg", "aa_order_leaflets_POPG.xvg",
              "aa_order_error_leaflets.csv", "aa_order_error_leaflets.tab", "aa_order_error_leaflets_limit.csv",
              "aa_order_error_leaflets_limit.tab", "aa_order_error_limit.csv", "aa_order_error_limit.tab",
              "aa_order_leaflets_limit.tab", "aa_order_different_hydrogen_numbers.csv", "aa_order_different_hydrogen_numbers.tab",
              "cg_order_basic_POPC.xvg", "cg_order_basic_POPE.xvg", "cg_order_basic_POPG.xvg", "cg_order_leaflets_POPE.xvg",
              "cg_order_leaflets_POPG.xvg", "cg_order_error.csv", "cg_order_error.tab", "cg_order_error_leaflets_limit.csv",
              "cg_order_error_leaflets_limit.tab", "cg_order_error_limit.csv", "cg_order_error_limit.tab",
              "cg_order_leaflets_limit.csv", "cg_order_leaflets_limit.tab", "cg_order_leaflets_convergence.xvg",
              "cg_order_convergence_s5.xvg"):
        src = os.path.join(REF, f)
        if os.path.exists(src):
            shutil.copy(src, os.path.join(HERE, "expected", f))
    # one small trajectory file as it is, for the reader's own tests (1 frame, 16 769 beads, 62 KB)
    shutil.copy(os.path.join(REF, "split", "cg3.xtc"), os.path.join(HERE, "cg3.xtc"))
    # two more of the reference's trajectory files, as data for the decoders (device against host, bit for bit): one
    # all-atom frame with its water (long runs of small offsets) and a small multi-frame system
    shutil.copy(os.path.join(REF, "split", "pcpepg4.xtc"), os.path.join(HERE, "pcpepg4.xtc"))
    shutil.copy(os.path.join(REF, "multiple_resid_same_name.xtc"), os.path.join(HERE, "multiple_resid_same_name.xtc"))
    # the united-atom ordermaps (made from ua.xtc): the 12 `_full` maps compared by tests_ua.rs:351-410 and the 24
    # `_upper` / `_lower` maps of test_ua_order_maps_leaflets (tests_ua.rs:418-507, Global leaflets)
    os.makedirs(os.path.join(HERE, "expected", "ordermaps_ua"), exist_ok=True)
    for f in sorted(os.listdir(os.path.join(REF, "ordermaps_ua"))):
        if f.endswith(("_full.dat", "_upper.dat", "_lower.dat")):
            shutil.copy(os.path.join(REF, "ordermaps_ua", f), os.path.join(HERE, "expected", "ordermaps_ua", f))
