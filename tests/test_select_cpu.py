"""The selection evaluator (gorder_amd/select.py) on the queries the reference's own tests use."""
import numpy as np
import pytest

from gorder_amd.select import SelectError, read_ndx, select
from golden_util import Fixture


@pytest.fixture(scope="module")
def aa():
    return Fixture("pcpepg")


@pytest.fixture(scope="module")
def ua():
    return Fixture("ua")


def test_keywords_and_boolean_operators(aa):
    s = aa.structure
    rn, resid = np.array(s.resnames), np.asarray(s.resids)
    everything = select(s, "all")
    assert everything.all() and (select(s, "@membrane") == everything).all()        # the fixture holds the lipids only
    assert (select(s, "@membrane and element name carbon") == aa.element("carbon")).all()
    assert (select(s, "elname hydrogen") == aa.element("hydrogen")).all()
    assert (select(s, "element symbol C") == aa.element("carbon")).all()
    assert (select(s, "not element name carbon") == ~aa.element("carbon")).all()
    assert (select(s, "resname POPC and name C22 C24 C218") == ((rn == "POPC") & aa.name_in("C22", "C24", "C218"))).all()
    assert (select(s, "resname POPC POPE") == np.isin(rn, ["POPC", "POPE"])).all()
    assert select(s, "name P").sum() == 274
    assert (select(s, "name P and not resid 144") == (aa.name_in("P") & (resid != 144))).all()
    assert (select(s, "name P or (resid 144 and name P HA)") ==
            (aa.name_in("P") | ((resid == 144) & aa.name_in("P", "HA")))).all()
    assert (select(s, "(name C218 C316 and not resid 144) or (resid 144 and name C218)") ==
            ((aa.name_in("C218", "C316") & (resid != 144)) | ((resid == 144) & aa.name_in("C218")))).all()
    # `and` binds tighter than `or`, `not` tighter than both; symbols work as well
    assert (select(s, "name P or name N and resname POPE") == (aa.name_in("P") | (aa.name_in("N") & (rn == "POPE")))).all()
    assert (select(s, "! name P && resname POPG || name C218") == ((~aa.name_in("P") & (rn == "POPG")) | aa.name_in("C218"))).all()
    assert (select(s, "resid 1 to 254") == ((resid >= 1) & (resid <= 254))).all()
    assert (select(s, "resid 7 12 30-40") == (np.isin(resid, [7, 12]) | ((resid >= 30) & (resid <= 40)))).all()
    assert (select(s, "serial 1 to 100") == (np.arange(s.n_atoms) < 100)).all()


def test_regular_expressions_and_molwith(ua):
    s = ua.structure
    rn = np.array(s.resnames)
    isc = np.array([n.startswith("C") for n in s.names])
    sat = select(s, "(resname POPC and name r'^C' and not name C15 C34 C24 C25) or "
                    "(resname POPS and name r'^C' and not name C6 C18 C39 C27 C28)")
    want = ((rn == "POPC") & isc & ~ua.name_in("C15", "C34", "C24", "C25")) | \
           ((rn == "POPS") & isc & ~ua.name_in("C6", "C18", "C39", "C27", "C28"))
    assert (sat == want).all()
    assert (select(s, "name r'^P'") == np.array([n.startswith("P") for n in s.names])).all()
    assert (select(s, "name r'^C[0-9]$'") == np.array([len(n) == 2 and n[0] == "C" and n[1].isdigit() for n in s.names])).all()
    # whole molecules holding a selected atom (the reference's helper group, uaorder.rs:215-224)
    one = select(s, "molwith (resid 3 and name r'^P')")
    assert one.sum() == (np.asarray(s.resids) == 3).sum() and (np.asarray(s.resids)[one] == 3).all()


def test_groups_and_index_files(aa, tmp_path):
    s = aa.structure
    path = tmp_path / "index.ndx"
    path.write_text("[ Heads ]\n" + " ".join(str(i + 1) for i in np.flatnonzero(aa.name_in("P"))) +
                    "\n[ Odd Name ]\n1 2 3\n4\n")
    groups = read_ndx(str(path))
    assert (select(s, "Heads", groups) == aa.name_in("P")).all()
    assert (select(s, "Heads and resname POPG", groups) == (aa.name_in("P") & (np.array(s.resnames) == "POPG"))).all()
    assert select(s, "'Odd Name'", groups).sum() == 4
    assert (select(s, "Mask", {"Mask": aa.name_in("N")}) == aa.name_in("N")).all()


@pytest.mark.parametrize("query", ["", "resname", "name P and", "(name P", "name P )", "resid one", "resid 3 to",
                                   "element carbon", "element name unobtainium", "@lipids", "NoSuchGroup",
                                   "name r'['"])
def test_malformed_queries_are_errors(aa, query):
    with pytest.raises(SelectError):
        select(aa.structure, query)
