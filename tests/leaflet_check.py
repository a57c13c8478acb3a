"""Unconditional comparison of order sums when the leaflet flags of device and oracle may differ.

A leaflet flag is the SIGN of a distance (leaflets.rs:725-731, 796-800); device and oracle accumulate the membrane
centre in different precisions, so a lipid within ~1e-6 nm of the mid-plane may land on the other side.
Instead of skipping the integer comparison when that happens, this helper
  1. reads the assignment the DEVICE made for every frame (one frame per submit, flags exported after each),
  2. bounds the disagreement with the oracle's own assignment: few molecules, all within `dist_tol` of the mid-plane,
  3. re-runs the oracle with those device flags handed over as a manual assignment (GORDER_LEAFLETS_MANUAL) and
     requires every sum and count — total, upper, lower — to be EQUAL to what the device accumulated.
The "total" rows never depend on the flags and are compared with the plain oracle run as well.
"""
import dataclasses

import numpy as np

from gorder_amd import HipEngine, abi
from gorder_amd.abi import LEAFLETS_MANUAL, Leaflets
from oracle import oracle


def device_flags_per_frame(tables, xyz, box, frame_index):
    """-> flags [F, n_mol] the device applies to each frame, and that engine's Results."""
    eng = HipEngine(tables)
    out = np.zeros((xyz.shape[0], tables.n_molecules_total), dtype=np.uint8)
    for f in range(xyz.shape[0]):
        eng.submit_host(xyz[f:f + 1], None if box is None else box[f:f + 1], frame_index[f:f + 1])
        out[f], _ = eng.leaflets()
    return out, eng.finish()


def oracle_flags_per_frame(tables, xyz, box, frame_index, trig):
    o = oracle.OracleEngine(tables, trig=trig)
    flags = np.zeros((xyz.shape[0], tables.n_molecules_total), dtype=np.uint8)
    dist = np.zeros((xyz.shape[0], tables.n_molecules_total), dtype=np.float32)
    for f in range(xyz.shape[0]):
        o.submit(xyz[f:f + 1], None if box is None else box[f:f + 1], frame_index[f:f + 1])
        flags[f], dist[f], _ = o.leaflets()
    return flags, dist, o.finish()


def assert_sums_given_device_flags(tables, xyz, box, got, frame_index=None, max_flag_diffs=4, dist_tol=1e-4,
                                   trig=None):
    """`got` = Results of the device run under test (any batching).  Returns the number of differing flags."""
    n = xyz.shape[0]
    fi = np.arange(n, dtype=np.uint64) if frame_index is None else np.asarray(frame_index, dtype=np.uint64)
    if trig is None:
        trig = oracle.TRIG_MIRROR if (tables.flags & abi.FLAG_TRIG_ACOS_COS) else oracle.TRIG_DIRECT
    dflags, per_frame = device_flags_per_frame(tables, xyz, box, fi)
    # one frame per submit and the batching under test are the same computation
    np.testing.assert_array_equal(per_frame.sums, got.sums)
    np.testing.assert_array_equal(per_frame.counts, got.counts)
    oflags, odist, plain = oracle_flags_per_frame(tables, xyz, box, fi, trig)
    np.testing.assert_array_equal(got.sums[0], plain.sums[0])        # totals do not depend on any flag
    np.testing.assert_array_equal(got.counts[0], plain.counts[0])
    diff = dflags != oflags
    n_diff = int(diff.sum())
    assert n_diff <= max_flag_diffs, f"{n_diff} leaflet flags differ from the oracle's"
    if n_diff:
        assert np.abs(odist[diff]).max() < dist_tol, "a lipid away from the mid-plane changed leaflet"
    # the oracle with the device's assignment handed over frame by frame: everything EQUAL
    manual = dataclasses.replace(tables, leaflets=Leaflets(method=LEAFLETS_MANUAL, normal_dim=tables.leaflets.normal_dim,
                                                           frequency=1, flip=False))
    o = oracle.OracleEngine(manual, trig=trig)
    for f in range(n):
        o.set_manual_leaflets(dflags[f], int(fi[f]))
        o.submit(xyz[f:f + 1], None if box is None else box[f:f + 1], fi[f:f + 1])
    want = o.finish()
    np.testing.assert_array_equal(got.sums, want.sums)
    np.testing.assert_array_equal(got.counts, want.counts)
    return n_diff
