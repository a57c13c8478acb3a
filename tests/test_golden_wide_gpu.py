"""The HIP path on the wider sweep of the reference's goldens (tests/golden_cases.py, test_golden_wide_oracle.py):
each case against the golden file, and against the oracle on the same frames."""
import numpy as np
import pytest

from gorder_amd import HipEngine
from gorder_amd import structure as st
from oracle import oracle
from golden_cases import CASES, expected_name
from golden_util import Fixture, expected
from test_golden_wide_oracle import EXPORTS, assignment_rows, check_normals, export_setup, normals_setup

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fixtures(built):
    return {k: Fixture(k) for k in ("pcpepg", "cg", "ua")}


@pytest.mark.parametrize("name", sorted(CASES))
def test_wide_reference_goldens(fixtures, name):
    c = CASES[name](fixtures)
    xyz = np.ascontiguousarray(c.fx.xyz[c.frames][:, c.midx, :])
    box = c.fx.boxes[c.frames]
    eng = HipEngine(c.tables)
    half = (len(c.frames) + 1) // 2
    for a, b in ((0, half), (half, len(c.frames))):
        if b > a:
            eng.submit_host(xyz[a:b], box[a:b], c.fidx[a:b])
    res = eng.finish()
    assert res.n_frames == len(c.frames)
    tw = eng.timewise(len(c.frames)) if c.blocks else None
    bad = st.compare_trees(c.tree(res, tw), expected(expected_name(name)))
    assert not bad, bad[:10]
    o = oracle.OracleEngine(c.tables, trig=oracle.TRIG_DIRECT, n_threads=4)
    o.submit(xyz, box, c.fidx)
    ref = o.finish()
    np.testing.assert_array_equal(res.counts, ref.counts)
    if c.kind == "ua":       # the unsaturated-CH hydrogen goes through device sincosf / acosf: one tick on a mean
        assert np.abs(res.order_ticks() - ref.order_ticks()).max() <= 1
    else:
        np.testing.assert_array_equal(res.sums, ref.sums)
    if tw is not None:
        ws, wc = o.timewise(len(c.frames))
        np.testing.assert_array_equal(tw[1], wc)
        if c.kind != "ua":
            np.testing.assert_array_equal(tw[0], ws)


@pytest.mark.parametrize("kind,want,freq,n_rows", EXPORTS)
@pytest.mark.parametrize("method", ["global", "local", "individual"])
def test_exported_leaflet_assignment(fixtures, kind, want, freq, n_rows, method):
    fx, (tables, labels, midx) = export_setup(kind, freq, fixtures, method)
    rows = assignment_rows(expected(want), labels)
    eng = HipEngine(tables)
    for f in range(0, 51, 7):
        fr = np.arange(0 if f == 0 else f - 6, f + 1)
        eng.submit_host(np.ascontiguousarray(fx.xyz[fr][:, midx, :]), fx.boxes[fr], fr)
        flags, frame = eng.leaflets()
        assert frame == (f // freq * freq if freq else 0)
        np.testing.assert_array_equal(flags, rows[frame // freq if freq else 0])


def test_exported_dynamic_normals(fixtures):
    fx, (tables, labels, midx) = normals_setup(fixtures)
    want = expected("ua_normals.yaml")
    eng = HipEngine(tables)
    loose = 0
    for f in range(51):
        eng.submit_host(np.ascontiguousarray(fx.xyz[[f]][:, midx, :]), fx.boxes[[f]], [f])
        loose += check_normals(eng.normals()[0].astype(np.float64), want, labels, f)
    assert loose <= 30


@pytest.mark.parametrize("kind,leaflets,name", [("aa", False, "aa_order_convergence.xvg"), ("aa", True, "aa_order_leaflets_convergence.xvg"),
                                                ("cg", False, "cg_order_convergence.xvg"),
                                                ("ua", False, "ua_order_convergence.xvg"), ("ua", True, "ua_order_leaflets_convergence.xvg")])
def test_convergence_of_the_per_frame_rows(fixtures, kind, leaflets, name):
    """The device's per-frame (timewise) rows reproduce the reference's convergence files frame by frame
    (tests_aa.rs:2580-2630, tests_cg.rs, tests_ua.rs:509-630)."""
    from gorder_amd import writers
    from golden_util import METHODS, aa_setup, cg_setup, ua_setup
    from test_writers_cpu import golden, same_tokens
    fx = fixtures[{"aa": "pcpepg", "cg": "cg", "ua": "ua"}[kind]]
    setup = {"aa": aa_setup, "cg": cg_setup, "ua": ua_setup}[kind]
    tables, labels, midx = setup(fx, leaflets=METHODS["global"] if leaflets else None, timewise=True)
    frames = fx.window()
    eng = HipEngine(tables)
    xyz = np.ascontiguousarray(fx.xyz[frames][:, midx, :])
    for a, b in ((0, 17), (17, 40), (40, len(frames))):
        eng.submit_host(xyz[a:b], fx.boxes[frames][a:b], frames[a:b])
    eng.finish()
    same_tokens(writers.convergence_text(eng.timewise(len(frames)), labels, kind, leaflets), golden(name))
