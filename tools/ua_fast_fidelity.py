#!/usr/bin/env python3
"""What GORDER_FLAG_UA_FAST_NORMALISE costs in fidelity, sample by sample (CPU only; the oracle's FAST mode is the
device's arithmetic bit for bit — tests/test_ua_fast_gpu.py — so nothing here needs a GPU).

    python tools/ua_fast_fidelity.py > profiles/r04_ua_fast_fidelity.json

Every virtual C-H sample of (a) the reference's own united-atom membrane (tests/golden/ua.npz: ua.xtc, 51 frames) and
(b) the synthetic V-UA workload of the benchmark (256 lipids, 91 x 91 ordermap tiles) is evaluated with the reference's
arithmetic (literal construction, libm acosf -> cosf) and with the fast construction + squared cosine; reported: the
histogram of tick differences, the fraction of samples that move, the shift of the mean, how many bond positions change
ordermap tile, and per order parameter (slot mean) the largest difference in ticks."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def means_difference(tables, xyz, box, fidx=None):
    from oracle import oracle
    out = {}
    engines = {}
    for name, trig in (("libm", oracle.TRIG_LIBM), ("fast", oracle.TRIG_DIRECT)):
        e = oracle.OracleEngine(tables, trig=trig, n_threads=8)
        e.submit(xyz, box, fidx)
        engines[name] = e.finish()
    a, b = engines["libm"], engines["fast"]
    assert np.array_equal(a.counts, b.counts)
    d = np.abs(a.order_ticks() - b.order_ticks())
    out["order_parameters"] = int((a.counts > 0).sum())
    out["max_difference_of_an_order_parameter_ticks"] = int(d[a.counts > 0].max())
    out["order_parameters_that_differ_by_one_tick"] = int((d[a.counts > 0] == 1).sum())
    if a.map_sums is not None and a.map_sums.size:
        out["ordermap_tiles_with_another_sample_count"] = int((a.map_counts != b.map_counts).sum())
        out["ordermap_tiles"] = int((a.map_counts > 0).sum())
    return out


def main():
    import __graft_entry__ as g
    g.build()
    from gorder_amd import synthetic
    from gorder_amd.abi import FLAG_UA_FAST_NORMALISE, OrderMap
    from oracle import oracle
    from golden_util import Fixture, METHODS, ua_setup
    report = {"what": __doc__.split("\n\n")[0]}
    # (a) the reference's membrane
    fx = Fixture("ua")
    tables, labels, midx = ua_setup(fx, leaflets=METHODS["global"])
    tables.flags |= FLAG_UA_FAST_NORMALISE
    frames = fx.window()
    xyz = np.ascontiguousarray(fx.xyz[frames][:, midx, :])
    e = oracle.OracleEngine(tables, trig=oracle.TRIG_DIRECT)
    r = e.ua_fast_fidelity(xyz, fx.boxes[frames])
    r.update(means_difference(tables, xyz, fx.boxes[frames], frames))
    report["reference_membrane_ua_xtc_51_frames"] = r
    # (b) the benchmark's synthetic workload with its ordermap
    om = OrderMap(enabled=True, plane=0, span_x=(0.0, 9.0), span_y=(0.0, 9.0), bin=(0.1, 0.1))
    system = synthetic.ua_membrane(256, ordermap=om)
    system.tables.flags |= FLAG_UA_FAST_NORMALISE
    n = 64
    xyz = system.frames(n, seed=5)
    e = oracle.OracleEngine(system.tables, trig=oracle.TRIG_DIRECT)
    r = e.ua_fast_fidelity(xyz, system.box9(n))
    r.update(means_difference(system.tables, xyz, system.box9(n)))
    report["synthetic_v_ua_256_lipids_64_frames_91x91_tiles"] = r
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
