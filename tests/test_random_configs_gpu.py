"""Randomised combinations of every option of the path — kind of system, periodic or not, leaflet method and
frequency, ordermaps, per-frame rows, geometry selection, dynamic normals, cosine mode, batching — each checked
against the oracle.  The single-feature tests pin each option; this one looks for bad interactions."""
import numpy as np
import pytest

from gorder_amd import HipEngine, abi, synthetic
from gorder_amd.abi import (GEOM_CUBOID, GEOM_CYLINDER, GEOM_SPHERE, GEOMREF_BOX_CENTER, GEOMREF_GROUP, GEOMREF_POINT,
                            LEAFLETS_GLOBAL, LEAFLETS_INDIVIDUAL, LEAFLETS_LOCAL, LEAFLETS_NONE, DynamicNormal, Geometry,
                            OrderMap)
from oracle import oracle

pytestmark = pytest.mark.gpu


def make_case(seed):
    rng = np.random.default_rng(seed)
    kind = rng.choice(["aa", "cg", "ua"])
    pbc = bool(rng.random() < 0.75)
    leaflets = int(rng.choice([LEAFLETS_NONE, LEAFLETS_GLOBAL, LEAFLETS_LOCAL, LEAFLETS_INDIVIDUAL]))
    if kind == "ua" and leaflets == LEAFLETS_INDIVIDUAL and rng.random() < 0.5:
        leaflets = LEAFLETS_GLOBAL
    frequency = int(rng.choice([0, 1, 1, 3]))
    timewise = bool(rng.random() < 0.4)
    kw = dict(leaflets=leaflets, frequency=frequency, timewise=timewise, handle_pbc=pbc)
    shape = rng.choice(["default", "default", "wide", "tall"])     # box shapes: flat and wide like the 1M-bead system, or tall
    if shape == "wide":
        kw["box"] = (float(rng.uniform(60, 130)), float(rng.uniform(60, 130)), 10.0 if kind == "cg" else 8.0)
    elif shape == "tall":
        kw["box"] = (float(rng.uniform(9, 14)), float(rng.uniform(9, 14)), float(rng.uniform(25, 40)))
    if kind == "aa":
        system = synthetic.aa_membrane(int(rng.integers(12, 40)), **kw)
    elif kind == "cg":
        system = synthetic.cg_membrane(int(rng.integers(60, 260)), n_types=int(rng.integers(1, 4)), radius=2.0, **kw)
    else:
        system = synthetic.ua_membrane(int(rng.integers(20, 70)), radius=2.0, **kw)
    t = system.tables
    bx = system.box
    if rng.random() < 0.45:
        plane = int(rng.integers(0, 3))
        d0, d1 = {0: (0, 1), 1: (0, 2), 2: (2, 1)}[plane]
        t.ordermap = OrderMap(enabled=True, plane=plane, span_x=(0.0, float(bx[d0])), span_y=(0.0, float(bx[d1])),
                              bin=(float(rng.uniform(0.6, 2.0)), float(rng.uniform(0.6, 2.0))))
    if rng.random() < 0.4:
        gk = int(rng.choice([GEOM_CUBOID, GEOM_CYLINDER, GEOM_SPHERE]))
        ref = int(rng.choice([GEOMREF_POINT, GEOMREF_GROUP] + ([GEOMREF_BOX_CENTER] if pbc else [])))
        g = Geometry(kind=gk, reference=ref, invert=bool(rng.random() < 0.3), radius=float(rng.uniform(2.0, 4.0)),
                     orientation=int(rng.integers(0, 3)), point=tuple(float(x) for x in rng.uniform(0.5, 0.9, 3) * bx),
                     xdim=(-2.0, 2.5), ydim=(-3.0, 1.5), zdim=(-float("inf"), float("inf")), span=(-2.5, 3.0),
                     structure_box=tuple(float(x) for x in bx))
        if ref == GEOMREF_GROUP:
            # a LOCALISED group (the first few molecules, neighbours in space): the circular mean of a group that fills
            # the box uniformly has no direction, and then which periodic images the refinement picks is decided by
            # rounding noise — in the reference as much as here
            p0 = system.frames(1, seed=0)[0].astype(np.float64)
            d = p0 - p0[0]
            if pbc:
                d -= np.asarray(bx, dtype=np.float64) * np.round(d / np.asarray(bx, dtype=np.float64))
            g.group = np.flatnonzero(np.linalg.norm(d, axis=1) < 2.0).astype(np.uint32)      # a 2-nm blob around atom 0
        t.geometry = g
    if rng.random() < 0.35 and all(m.heads is not None for m in t.molecule_types):
        cloud = []
        for m in t.molecule_types:
            m.normal_heads = np.asarray(m.heads, dtype=np.uint32)
            cloud.append(m.normal_heads)
        t.dynamic_normal = DynamicNormal(enabled=True, radius=float(rng.uniform(2.2, 3.0)), cloud=np.concatenate(cloud))
    if rng.random() < 0.3:
        t.flags = abi.FLAG_TRIG_ACOS_COS
    elif kind == "ua" and rng.random() < 0.5:
        t.flags = abi.FLAG_UA_FAST_NORMALISE          # (round 4) the tolerance-bounded construction, restated by the oracle
    n = int(rng.integers(5, 14))
    batches = int(rng.integers(1, 4))
    return system, n, batches, kind


import os  # noqa: E402

N_SEEDS = int(os.environ.get("GORDER_RANDOM_CONFIGS", "150"))     # soak runs: GORDER_RANDOM_CONFIGS=3000


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_configuration(built, seed):
    system, n, batches, kind = make_case(1000 + seed)
    t = system.tables
    xyz = system.frames(n, seed=seed)
    box = system.box9(n) if t.handle_pbc else None
    trig = oracle.TRIG_MIRROR if (t.flags & abi.FLAG_TRIG_ACOS_COS) else oracle.TRIG_DIRECT
    eng = HipEngine(t)
    o = oracle.OracleEngine(t, trig=trig, n_threads=2)
    edges = np.linspace(0, n, batches + 1).astype(int)

    def run(e):
        for a, b in zip(edges[:-1], edges[1:]):
            if b > a:
                e.submit(xyz[a:b], None if box is None else box[a:b], np.arange(a, b)) if e is o else \
                    e.submit_host(xyz[a:b], None if box is None else box[a:b], np.arange(a, b))
        return e.finish()

    from gorder_amd import GorderHipError
    try:
        want = run(o)
    except oracle.OracleError as err:      # e.g. a dynamic-normal cloud with fewer than 3 heads: the same error, please
        with pytest.raises(GorderHipError) as ei:
            run(eng)
        assert ei.value.status == err.status
        return
    got = run(eng)
    lf = t.leaflets.method != LEAFLETS_NONE
    if lf:      # a lipid within 1e-4 nm of its mid-plane may be classified differently (f64 vs f32 centre sums)
        flags, _ = eng.leaflets()
        oflags, odist, _ = o.leaflets()
        diff = flags != oflags
        assert not diff.any() or np.abs(odist[diff]).max() < 1e-4
        if diff.any():
            pytest.skip("a lipid sits on the mid-plane in this random system")
    np.testing.assert_array_equal(got.counts, want.counts)
    # exact integer sums (united atoms too: every angle function is restated by the oracle, in the default, the literal and
    # the fast mode), except with dynamic normals (summation order of the cloud): <= 1 tick there
    loose = bool(t.dynamic_normal.enabled)
    if loose:
        assert np.abs(got.order_ticks() - want.order_ticks()).max() <= 1
    else:
        np.testing.assert_array_equal(got.sums, want.sums)
    if t.ordermap.enabled:
        np.testing.assert_array_equal(got.map_counts, want.map_counts)
        if not loose:
            np.testing.assert_array_equal(got.map_sums, want.map_sums)
    if t.timewise:
        gs, gc = eng.timewise(n)
        ws, wc = o.timewise(n)
        np.testing.assert_array_equal(gc, wc)
        if not loose:
            np.testing.assert_array_equal(gs, ws)
