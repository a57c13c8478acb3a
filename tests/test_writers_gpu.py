"""GPU twins of tests/test_writers_cpu.py: the YAML / CSV / TAB / XVG text of the reference's presenters
(presentation/yaml_presenter.rs:80-136, csv_presenter.rs, tab_presenter.rs, xvg_presenter.rs) made from the HIP
path's accumulators — not from the oracle's — and compared with the reference's own files by its own rule
(tests/common/mod.rs:95-150: the same items line by line, numbers within 2e-4)."""
import numpy as np
import pytest

from gorder_amd import HipEngine
from gorder_amd import structure as st
from gorder_amd import writers
from golden_util import METHODS, Fixture, aa_setup, cg_setup, ua_setup
from test_writers_cpu import CASES, MORE_TEXT, check_text, golden, same_items, same_tokens

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fixtures(built):
    return {"aa": Fixture("pcpepg"), "cg": Fixture("cg"), "ua": Fixture("ua")}


def hip_run(tables, fx, midx, frames, frame_index=None, batches=3):
    eng = HipEngine(tables)
    xyz = np.ascontiguousarray(fx.xyz[frames][:, midx, :])
    fi = np.asarray(frames if frame_index is None else frame_index)
    edges = np.linspace(0, len(frames), batches + 1).astype(int)
    for a, b in zip(edges[:-1], edges[1:]):
        if b > a:
            eng.submit_host(xyz[a:b], fx.boxes[frames][a:b], fi[a:b])
    return eng, eng.finish()


@pytest.mark.parametrize("kind,leaflets,errors,name", CASES)
def test_yaml_csv_tab_xvg_text_from_the_hip_path(fixtures, kind, leaflets, errors, name):
    fx = fixtures[kind]
    setup = {"aa": aa_setup, "cg": cg_setup, "ua": ua_setup}[kind]
    tables, labels, midx = setup(fx, leaflets=METHODS["global"] if leaflets else None, timewise=errors)
    frames = fx.window()
    eng, res = hip_run(tables, fx, midx, frames)
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


def test_csv_prints_nan_below_min_samples_from_the_hip_path(fixtures):
    fx = fixtures["aa"]
    tables, labels, midx = aa_setup(fx, leaflets=METHODS["global"])
    eng, res = hip_run(tables, fx, midx, fx.window())
    tree = st.results_tree(res, labels, "aa", leaflets=True, min_samples=500)
    text = writers.csv_text(tree)
    assert "NaN" in text
    same_items(text, golden("aa_order_leaflets_limit.csv"), sep=",")
    assert ".nan" in writers.yaml_text(tree)


def test_convergence_with_a_step_from_the_hip_path(fixtures):
    # tests_aa.rs:2657-2680: every fifth frame; the x column keeps the frame numbers of the trajectory
    fx = fixtures["aa"]
    tables, labels, midx = aa_setup(fx, timewise=True)
    frames = fx.window(None, None, 5)
    eng, _ = hip_run(tables, fx, midx, frames, frame_index=np.arange(len(frames)) * 5, batches=2)
    same_tokens(writers.convergence_text(eng.timewise(len(frames)), labels, "aa", False, step=5),
                golden("aa_order_convergence_s5.xvg"))


# ---- the remaining text twins (round 4), from the HIP path's accumulators -------------------------------------------------
def hip_tree(fx, kind, leaflets=False, errors=False, min_samples=1, heavy=None):
    setup = {"aa": aa_setup, "cg": cg_setup}[kind]
    kw = dict(leaflets=METHODS["global"] if leaflets else None, timewise=errors)
    if heavy is not None:
        kw["heavy"] = heavy
    tables, labels, midx = setup(fx, **kw)
    frames = fx.window()
    eng, res = hip_run(tables, fx, midx, frames)
    tw = eng.timewise(len(frames)) if errors else None
    return st.results_tree(res, labels, kind, leaflets=leaflets, timewise=tw, min_samples=min_samples)


@pytest.mark.parametrize("kind,leaflets,errors,min_samples,files", MORE_TEXT, ids=[c[4][0] for c in MORE_TEXT])
def test_more_text_outputs_from_the_hip_path(fixtures, kind, leaflets, errors, min_samples, files):
    tree = hip_tree(fixtures[kind], kind, leaflets, errors, min_samples)
    for name in files:
        check_text(tree, name, kind)


def test_molecule_types_with_different_numbers_of_hydrogens_from_the_hip_path(fixtures):
    from gorder_amd.select import select
    fx = fixtures["aa"]
    heavy = select(fx.structure, "(resname POPC and name C29 C210) or (resname POPE and element name carbon)")
    tree = hip_tree(fx, "aa", leaflets=True, heavy=heavy)
    same_tokens(writers.tab_text(tree), golden("aa_order_different_hydrogen_numbers.tab"))
    same_items(writers.csv_text(tree), golden("aa_order_different_hydrogen_numbers.csv"), sep=",")


@pytest.mark.parametrize("leaflets,step,name", [(True, 1, "cg_order_leaflets_convergence.xvg"), (False, 5, "cg_order_convergence_s5.xvg")])
def test_more_convergence_files_from_the_hip_path(fixtures, leaflets, step, name):
    fx = fixtures["cg"]
    tables, labels, midx = cg_setup(fx, leaflets=METHODS["global"] if leaflets else None, timewise=True)
    frames = fx.window(None, None, step)
    eng, _ = hip_run(tables, fx, midx, frames, frame_index=np.arange(len(frames)) * step, batches=2)
    same_tokens(writers.convergence_text(eng.timewise(len(frames)), labels, "cg", leaflets, step=step), golden(name))
