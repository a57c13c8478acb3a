"""YAML and CSV text in the reference's layouts (gorder_amd/writers.py), compared with its files by its own rule
(tests/common/mod.rs:95-150: the same items line by line, numbers within 2e-4)."""
import os

import numpy as np
import pytest

from gorder_amd import structure as st
from gorder_amd import writers
from oracle import oracle
from golden_util import GOLDEN, METHODS, Fixture, aa_setup, cg_setup, ua_setup


def same_items(got: str, want: str, sep=None, skip=0):
    ga, wa = got.splitlines()[skip:], want.splitlines()[skip:]
    assert len(ga) == len(wa), (len(ga), len(wa))
    for n, (g, w) in enumerate(zip(ga, wa)):
        if sep is None:     # YAML: indentation is structure
            assert len(g) - len(g.lstrip()) == len(w) - len(w.lstrip()), (n, g, w)
        gi, wi = (g.split(sep), w.split(sep))
        assert len(gi) == len(wi), (n, g, w)
        for a, b in zip(gi, wi):
            try:
                fa, fb = np.float32(a), np.float32(b)
            except ValueError:
                assert a == b, (n, g, w)
                continue
            assert (np.isnan(fa) and np.isnan(fb)) or abs(fa - fb) <= np.float32(2e-4), (n, g, w)


def same_tokens(got: str, want: str, skip=1):
    """The reference's rule for .tab / .xvg files: whitespace-separated items line by line."""
    ga, wa = got.splitlines()[skip:], want.splitlines()[skip:]
    assert len(ga) == len(wa), (len(ga), len(wa))
    for n, (g, w) in enumerate(zip(ga, wa)):
        gi, wi = g.split(), w.split()
        assert len(gi) == len(wi), (n, g, w)
        for a, b in zip(gi, wi):
            try:
                fa, fb = np.float32(a), np.float32(b)
            except ValueError:
                assert a == b, (n, g, w)
                continue
            assert (np.isnan(fa) and np.isnan(fb)) or abs(fa - fb) <= np.float32(2e-4), (n, g, w)


def golden(name):
    with open(os.path.join(GOLDEN, "expected", name)) as f:
        return f.read()


CASES = [("aa", False, False, "aa_order_basic"), ("aa", True, False, "aa_order_leaflets"), ("aa", False, True, "aa_order_error"),
         ("cg", False, False, "cg_order_basic"), ("cg", True, False, "cg_order_leaflets"), ("cg", True, True, "cg_order_error_leaflets"),
         ("ua", False, False, "ua_order_basic"), ("ua", True, False, "ua_order_leaflets"),
         ("ua", False, True, "ua_order_error"), ("ua", True, True, "ua_order_leaflets_error")]    # tests_ua.rs:509-630


@pytest.fixture(scope="module")
def fixtures(built):
    return {"aa": Fixture("pcpepg"), "cg": Fixture("cg"), "ua": Fixture("ua")}


@pytest.mark.parametrize("kind,leaflets,errors,name", CASES)
def test_yaml_and_csv_text(fixtures, kind, leaflets, errors, name):
    fx = fixtures[kind]
    setup = {"aa": aa_setup, "cg": cg_setup, "ua": ua_setup}[kind]
    tables, labels, midx = setup(fx, leaflets=METHODS["global"] if leaflets else None, timewise=errors)
    frames = fx.window()
    eng = oracle.OracleEngine(tables, trig=oracle.TRIG_LIBM, n_threads=4)
    eng.submit(np.ascontiguousarray(fx.xyz[frames][:, midx, :]), fx.boxes[frames], frames)
    res = eng.finish()
    tw = eng.timewise(len(frames)) if errors else None
    tree = (st.results_tree_ua(res, labels, leaflets=leaflets, timewise=tw) if kind == "ua"
            else st.results_tree(res, labels, kind, leaflets=leaflets, timewise=tw))
    same_items(writers.yaml_text(tree, header="# made here"), golden(name + ".yaml"), skip=1)
    same_items(writers.csv_text(tree), golden(name + ".csv"), sep=",")
    same_tokens(writers.tab_text(tree), golden(name + ".tab"))
    if not errors and (kind, leaflets) != ("cg", False):
        same_tokens(writers.xvg_text(tree, "POPC", united=kind == "ua"), golden(name + "_POPC.xvg"))
    if kind == "ua" and not errors:
        same_tokens(writers.xvg_text(tree, "POPS", united=True), golden(name + "_POPS.xvg"))


def test_csv_prints_nan_below_min_samples(fixtures):
    fx = fixtures["aa"]
    tables, labels, midx = aa_setup(fx, leaflets=METHODS["global"])
    frames = fx.window()
    eng = oracle.OracleEngine(tables, trig=oracle.TRIG_LIBM, n_threads=4)
    eng.submit(np.ascontiguousarray(fx.xyz[frames][:, midx, :]), fx.boxes[frames], frames)
    tree = st.results_tree(eng.finish(), labels, "aa", leaflets=True, min_samples=500)
    text = writers.csv_text(tree)
    assert "NaN" in text
    same_items(text, golden("aa_order_leaflets_limit.csv"), sep=",")
    assert ".nan" in writers.yaml_text(tree)


@pytest.mark.parametrize("kind,leaflets,name", [("aa", False, "aa_order_convergence.xvg"), ("aa", True, "aa_order_leaflets_convergence.xvg"),
                                                ("cg", False, "cg_order_convergence.xvg"),
                                                ("ua", False, "ua_order_convergence.xvg"), ("ua", True, "ua_order_leaflets_convergence.xvg")])
def test_convergence_of_the_per_frame_rows(fixtures, kind, leaflets, name):
    """tests_aa.rs:2580-2630, tests_cg.rs, tests_ua.rs:509-630: the running averages after every frame — pins the
    per-frame (timewise) rows one frame at a time, not just their block statistics."""
    fx = fixtures[kind]
    setup = {"aa": aa_setup, "cg": cg_setup, "ua": ua_setup}[kind]
    tables, labels, midx = setup(fx, leaflets=METHODS["global"] if leaflets else None, timewise=True)
    frames = fx.window()
    eng = oracle.OracleEngine(tables, trig=oracle.TRIG_LIBM, n_threads=4)
    eng.submit(np.ascontiguousarray(fx.xyz[frames][:, midx, :]), fx.boxes[frames], frames)
    eng.finish()
    text = writers.convergence_text(eng.timewise(len(frames)), labels, kind, leaflets)
    same_tokens(text, golden(name))


def test_convergence_with_a_step(fixtures):
    # tests_aa.rs:2657-2680: every fifth frame; the x column keeps the frame numbers of the trajectory
    fx = fixtures["aa"]
    tables, labels, midx = aa_setup(fx, timewise=True)
    frames = fx.window(None, None, 5)
    eng = oracle.OracleEngine(tables, trig=oracle.TRIG_LIBM, n_threads=4)
    eng.submit(np.ascontiguousarray(fx.xyz[frames][:, midx, :]), fx.boxes[frames], np.arange(len(frames)) * 5)
    eng.finish()
    same_tokens(writers.convergence_text(eng.timewise(len(frames)), labels, "aa", False, step=5), golden("aa_order_convergence_s5.xvg"))


# ---- the remaining text twins of analyses the checkout's data reproduces (round 4) -------------------------------------
def analysed(fx, kind, leaflets=False, errors=False, min_samples=1, heavy=None):
    setup = {"aa": aa_setup, "cg": cg_setup}[kind]
    kw = dict(leaflets=METHODS["global"] if leaflets else None, timewise=errors)
    if heavy is not None:
        kw["heavy"] = heavy
    tables, labels, midx = setup(fx, **kw)
    frames = fx.window()
    eng = oracle.OracleEngine(tables, trig=oracle.TRIG_LIBM, n_threads=4)
    eng.submit(np.ascontiguousarray(fx.xyz[frames][:, midx, :]), fx.boxes[frames], frames)
    res = eng.finish()
    tw = eng.timewise(len(frames)) if errors else None
    return st.results_tree(res, labels, kind, leaflets=leaflets, timewise=tw, min_samples=min_samples)


MORE_TEXT = [   # (kind, leaflets, errors, min_samples, files)   tests_aa.rs / tests_cg.rs: the csv / tab / xvg outputs of the same runs
    ("aa", False, False, 1, ["aa_order_basic_POPE.xvg", "aa_order_basic_POPG.xvg"]),
    ("aa", True, False, 1, ["aa_order_leaflets_POPE.xvg", "aa_order_leaflets_POPG.xvg"]),
    ("aa", True, True, 1, ["aa_order_error_leaflets.csv", "aa_order_error_leaflets.tab"]),
    ("aa", True, True, 500, ["aa_order_error_leaflets_limit.csv", "aa_order_error_leaflets_limit.tab"]),
    ("aa", False, True, 2000, ["aa_order_error_limit.csv", "aa_order_error_limit.tab"]),
    ("aa", True, False, 500, ["aa_order_leaflets_limit.tab"]),
    ("cg", False, False, 1, ["cg_order_basic_POPC.xvg", "cg_order_basic_POPE.xvg", "cg_order_basic_POPG.xvg"]),
    ("cg", True, False, 1, ["cg_order_leaflets_POPE.xvg", "cg_order_leaflets_POPG.xvg"]),
    ("cg", False, True, 1, ["cg_order_error.csv", "cg_order_error.tab"]),
    ("cg", True, True, 2000, ["cg_order_error_leaflets_limit.csv", "cg_order_error_leaflets_limit.tab"]),
    ("cg", False, True, 5000, ["cg_order_error_limit.csv", "cg_order_error_limit.tab"]),
    ("cg", True, False, 2000, ["cg_order_leaflets_limit.csv", "cg_order_leaflets_limit.tab"]),
]


def check_text(tree, name, kind):
    if name.endswith(".csv"):
        same_items(writers.csv_text(tree), golden(name), sep=",")
    elif name.endswith(".tab"):
        same_tokens(writers.tab_text(tree), golden(name))
    else:
        same_tokens(writers.xvg_text(tree, name.rsplit("_", 1)[1][:-4], united=False), golden(name))


@pytest.mark.parametrize("kind,leaflets,errors,min_samples,files", MORE_TEXT, ids=[c[4][0] for c in MORE_TEXT])
def test_more_text_outputs(fixtures, kind, leaflets, errors, min_samples, files):
    tree = analysed(fixtures[kind], kind, leaflets, errors, min_samples)
    for name in files:
        check_text(tree, name, kind)


def test_molecule_types_with_different_numbers_of_hydrogens(fixtures):
    # tests_aa.rs:1043-1095: POPC's two selected carbons carry one hydrogen, POPE's up to three: the table pads the columns
    from gorder_amd.select import select
    fx = fixtures["aa"]
    heavy = select(fx.structure, "(resname POPC and name C29 C210) or (resname POPE and element name carbon)")
    tree = analysed(fx, "aa", leaflets=True, heavy=heavy)
    same_tokens(writers.tab_text(tree), golden("aa_order_different_hydrogen_numbers.tab"))
    same_items(writers.csv_text(tree), golden("aa_order_different_hydrogen_numbers.csv"), sep=",")


@pytest.mark.parametrize("leaflets,step,name", [(True, 1, "cg_order_leaflets_convergence.xvg"), (False, 5, "cg_order_convergence_s5.xvg")])
def test_more_convergence_files(fixtures, leaflets, step, name):
    # tests_cg.rs:1711-1780
    fx = fixtures["cg"]
    tables, labels, midx = cg_setup(fx, leaflets=METHODS["global"] if leaflets else None, timewise=True)
    frames = fx.window(None, None, step)
    eng = oracle.OracleEngine(tables, trig=oracle.TRIG_LIBM, n_threads=4)
    eng.submit(np.ascontiguousarray(fx.xyz[frames][:, midx, :]), fx.boxes[frames], np.arange(len(frames)) * step)
    eng.finish()
    same_tokens(writers.convergence_text(eng.timewise(len(frames)), labels, "cg", leaflets, step=step), golden(name))
