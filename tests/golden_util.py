"""Shared helpers for the golden-fixture tests (data under tests/golden/, made by make_fixtures.py)."""
import os

import numpy as np
import yaml

from gorder_amd import structure as st
from gorder_amd.select import select
from gorder_amd.abi import LEAFLETS_GLOBAL, LEAFLETS_INDIVIDUAL, LEAFLETS_LOCAL

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Fixture:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.structure = st.Structure(z["resids"], [str(x) for x in z["resnames"]], [str(x) for x in z["names"]],
                                      z["structure_box"])
        n = self.structure.n_atoms
        adj = [[] for _ in range(n)]
        for a, b in z["bonds"]:
            adj[a].append(int(b)); adj[b].append(int(a))
        self.structure.bonds = [sorted(x) for x in adj]
        prec = np.float32(z["precision"])
        # exactly the decoder's arithmetic: int * (1 / precision) in f32
        self.xyz = (z["ints"].astype(np.float32) * np.float32(np.float32(1.0) / prec)).astype(np.float32)
        self.boxes = z["boxes"].astype(np.float32)
        self.times = z["times"]
        self.names = np.array(self.structure.names)
        self.resnames = np.array(self.structure.resnames)
        self.extra = {k: z[k] for k in z.files if k.startswith(("bonds_", "frames_"))}

    def with_bonds(self, key):
        """The same system under another bond definition of the reference (a second .bnd file)."""
        import copy
        other = copy.copy(self)
        other.structure = copy.copy(self.structure)
        adj = [[] for _ in range(self.structure.n_atoms)]
        for a, b in self.extra[key]:
            adj[a].append(int(b)); adj[b].append(int(a))
        other.structure.bonds = [sorted(x) for x in adj]
        return other

    def window(self, begin=None, end=None, step=1):
        """groan_rs time window (common.rs:239-246): frames with begin <= t <= end, every step-th."""
        sel = np.ones(len(self.times), dtype=bool)
        if begin is not None:
            sel &= self.times >= begin
        if end is not None:
            sel &= self.times <= end
        idx = np.flatnonzero(sel)[::step]
        return idx

    def element(self, el):
        s = self.structure
        return np.array([st.guess_element(n, r) == el for n, r in zip(s.names, s.resnames)])

    def name_in(self, *names):
        return np.isin(self.names, names)


def expected(name):
    with open(os.path.join(GOLDEN, "expected", name)) as f:
        return yaml.safe_load(f)


def aa_setup(fx, leaflets=None, frequency=1, heavy=None, **kw):
    """AAOrder '@membrane and element name carbon' / '... hydrogen' (tests_aa.rs:63-66); the fixture
    holds exactly the @membrane lipids."""
    hyd = select(fx.structure, "@membrane and element name hydrogen")
    if heavy is None:
        heavy = select(fx.structure, "@membrane and element name carbon")
    elif isinstance(heavy, str):
        heavy = select(fx.structure, heavy)
    lf = None
    if leaflets is not None:
        lf = {"method": leaflets, "membrane": select(fx.structure, "@membrane"), "heads": select(fx.structure, "name P"),
              "methyls": select(fx.structure, "name C218 C316"), "frequency": frequency, "radius": 2.5}
    return st.build_tables(fx.structure, "aa", heavy, hyd, leaflets=lf, **kw)


def cg_setup(fx, leaflets=None, frequency=1, **kw):
    """CGOrder '@membrane' (tests_cg.rs:194)."""
    beads = select(fx.structure, "@membrane")
    lf = None
    if leaflets is not None:
        lf = {"method": leaflets, "membrane": beads, "heads": select(fx.structure, "name PO4"),
              "methyls": select(fx.structure, "name C4A C4B"), "frequency": frequency, "radius": 2.5}
    return st.build_tables(fx.structure, "cg", beads, leaflets=lf, **kw)


METHODS = {"global": LEAFLETS_GLOBAL, "local": LEAFLETS_LOCAL, "individual": LEAFLETS_INDIVIDUAL}


def ua_setup(fx, leaflets=None, frequency=1, flip=False, sat_only=False, unsat_only=False, **kw):
    """UAOrder selections of tests_ua.rs:41-45 (saturated / unsaturated carbons of POPC and POPS)."""
    s = fx.structure
    sat = select(s, "(resname POPC and name r'^C' and not name C15 C34 C24 C25) or "
                    "(resname POPS and name r'^C' and not name C6 C18 C39 C27 C28)")
    unsat = select(s, "(resname POPC and name C24 C25) or (resname POPS and name C27 C28)")
    allm = select(s, "@membrane")
    lf = None
    if leaflets is not None:
        heads = select(s, "name r'^P'")
        methyls = select(s, "(resname POPC and name CA2 C50) or (resname POPS and name C36 C55)")
        lf = {"method": leaflets, "membrane": allm, "heads": heads, "methyls": methyls, "frequency": frequency,
              "radius": 2.5, "flip": flip}
    if sat_only:
        unsat = np.zeros_like(unsat)
    if unsat_only:
        sat = np.zeros_like(sat)
    return st.build_tables_ua(s, sat, unsat, allm, leaflets=lf, **kw)


def read_map(name, directory="ordermaps_ua"):
    """An ordermap .dat of the reference (ordermap_*.dat: `x y value` lines, NaN for tiles with fewer than
    min_samples samples) -> {(x, y): value}."""
    out = {}
    with open(os.path.join(GOLDEN, "expected", directory, name)) as f:
        for line in f:
            p = line.split()
            if len(p) == 3 and p[0][0] in "0123456789-":
                out[(p[0], p[1])] = float(p[2])
    return out


def map_of(res, slots, w, om, min_samples, sign=-1.0):
    """Finalise one ordermap like the reference's writer: tiles x-major at span_min + k * bin, value = mean of
    the aggregated slots by truncating i64 division, x sign, rounded to 4 decimals; NaN below min_samples."""
    s = res.map_sums[w, slots].sum(axis=0)
    c = res.map_counts[w, slots].sum(axis=0)
    out = {}
    for ix in range(s.shape[0]):
        for iy in range(s.shape[1]):
            key = (f"{om.span_x[0] + ix * om.bin[0]:.4f}", f"{om.span_y[0] + iy * om.bin[1]:.4f}")
            v = st._mean_ticks(int(s[ix, iy]), int(c[ix, iy]), min_samples)
            out[key] = st.round4(sign * v) if v == v else float("nan")
    return out


def compare_maps(got, want, tol=2e-4):
    bad = []
    if set(got) != set(want):
        return [f"tile sets differ: {len(got)} vs {len(want)}"]
    for k, w in want.items():
        g = got[k]
        if (g != g) != (w != w) or (w == w and abs(np.float32(g) - np.float32(w)) > tol):
            bad.append(f"{k}: {g} vs {w}")
    return bad
