"""Harness-level host code around the hot path (minimal forms):

* structure + bonds -> molecule types -> index tables (`Tables`) in the reference's ordering
  (/root/reference/src/analysis/topology/classify.rs:45-314, bond.rs:76-81: bond types sorted by the
  relative indices of their atoms; molecule types in order of first appearance);
* raw accumulators -> the numbers gorder writes: `sum / n` integer mean, sign flip for AA/UA
  (presentation/mod.rs:618-625), per-atom / per-molecule / whole-system aggregation by summing
  accumulators (presentation/converter.rs:324-470), 4-decimal rounding (presentation/mod.rs:496-504).

This is Python because it is test/example plumbing, not the product: the product boundary is the C ABI
that takes the finished index tables.
"""
from __future__ import annotations

from collections import deque
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .abi import (LEAFLETS_GLOBAL, LEAFLETS_INDIVIDUAL, LEAFLETS_LOCAL, LEAFLETS_NONE, Leaflets, MolType,
                  OrderMap, Results, Tables)


@dataclass
class Structure:
    resids: np.ndarray          # [N] int
    resnames: List[str]
    names: List[str]
    box: np.ndarray             # [3] float32 (orthogonal)
    positions: Optional[np.ndarray] = None   # [N,3] float32
    bonds: Optional[List[List[int]]] = None  # adjacency lists

    @property
    def n_atoms(self) -> int:
        return len(self.names)


def read_gro(path: str) -> Structure:
    with open(path) as f:
        f.readline()
        n = int(f.readline())
        resids = np.zeros(n, dtype=np.int64)
        resnames, names = [], []
        pos = np.zeros((n, 3), dtype=np.float32)
        for k in range(n):
            line = f.readline()
            resids[k] = int(line[0:5])
            resnames.append(line[5:10].strip())
            names.append(line[10:15].strip())
            pos[k] = (float(line[20:28]), float(line[28:36]), float(line[36:44]))
        box = np.array([float(x) for x in f.readline().split()[:3]], dtype=np.float32)
    return Structure(resids, resnames, names, box, pos)


def read_bnd(path: str, n_atoms: int) -> List[List[int]]:
    """gorder 'bonds file' (structure.rs:121-167): `atom: bonded atoms...`, 1-based serial numbers."""
    adj: List[set] = [set() for _ in range(n_atoms)]
    with open(path) as f:
        for line in f:
            line = line.split("#")[0].strip()
            if not line:
                continue
            nums = [int(x) for x in line.replace(":", " ").split()]
            a = nums[0] - 1
            for b in nums[1:]:
                adj[a].add(b - 1)
                adj[b - 1].add(a)
    return [sorted(s) for s in adj]


def read_pdb(path: str) -> Structure:
    """ATOM/HETATM records (fixed columns) + CONECT bonds (1-based serials in file order)."""
    resids, resnames, names, pos, serial_of = [], [], [], [], {}
    conect = []
    box = np.zeros(3, dtype=np.float32)
    with open(path) as f:
        for line in f:
            rec = line[:6]
            if rec in ("ATOM  ", "HETATM"):
                serial_of[int(line[6:11])] = len(names)
                names.append(line[12:16].strip())
                resnames.append(line[17:21].strip())
                resids.append(int(line[22:26]))
                pos.append((float(line[30:38]) / 10.0, float(line[38:46]) / 10.0, float(line[46:54]) / 10.0))
            elif rec == "CONECT":
                nums = [int(line[k:k + 5]) for k in range(6, len(line.rstrip()), 5) if line[k:k + 5].strip()]
                conect.append(nums)
            elif rec == "CRYST1":
                box = np.array([float(line[6:15]) / 10, float(line[15:24]) / 10, float(line[24:33]) / 10], np.float32)
            elif rec == "ENDMDL":
                break
        # CONECT records may follow ENDMDL
        for line in f:
            if line[:6] == "CONECT":
                conect.append([int(line[k:k + 5]) for k in range(6, len(line.rstrip()), 5) if line[k:k + 5].strip()])
    adj = [set() for _ in names]
    for nums in conect:
        a = serial_of[nums[0]]
        for b in nums[1:]:
            adj[a].add(serial_of[b]); adj[serial_of[b]].add(a)
    return Structure(np.array(resids), resnames, names, box, np.array(pos, dtype=np.float32), [sorted(x) for x in adj])


def guess_element(name: str, resname: str = "") -> str:
    """Coarse element guess for lipid atoms (groan_rs guesses elements from atom names; only C/H are
    needed by the AA selections used in the fixtures)."""
    n = name.lstrip("0123456789")
    if not n:
        return ""
    if n[0] == "H":
        return "hydrogen"
    if n[0] == "C" and not n.upper().startswith(("CL", "CA2", "CAL")):
        return "carbon"
    return {"N": "nitrogen", "O": "oxygen", "P": "phosphorus", "S": "sulfur"}.get(n[0], "")


@dataclass
class BondLabel:
    rel1: int
    name1: str
    rel2: int
    name2: str
    res1: str = ""      # residue names of the two atoms (the output keys carry these, not the molecule-type name)
    res2: str = ""


@dataclass
class MolLabels:
    name: str
    bonds: List[BondLabel]
    heavy_atoms: List[Tuple[int, str, str]] = field(default_factory=list)    # AA: (relative index, name, residue)
    n_molecules: int = 0
    slot0: int = 0


def _molecule_of(adj, start, visited):
    comp, dq = [start], deque([start])
    visited.add(start)
    while dq:
        a = dq.popleft()
        for b in adj[a]:
            if b not in visited:
                visited.add(b)
                comp.append(b)
                dq.append(b)
    return sorted(comp)


def classify(structure: Structure, group1: np.ndarray, group2: np.ndarray, same_group: bool):
    """BondBasedClassifier (classify.rs:140-314): molecules = connected components reached from the
    atoms of order group 1; molecule types = classes of equal topology, in order of first appearance.
    Returns [(name, [molecule atom lists]), ...] and per type the order-bond list as relative pairs."""
    adj = structure.bonds
    visited: set = set()
    types: List[dict] = []
    for a in np.flatnonzero(group1):
        a = int(a)
        if a in visited:
            continue
        atoms = _molecule_of(adj, a, visited)
        m0 = atoms[0]
        inmol = set(atoms)
        bonds = set()
        for i in atoms:
            for j in adj[i]:
                if j in inmol and i < j:
                    ok = (group1[i] and group2[j]) or (group1[j] and group2[i])
                    if same_group:
                        ok = group1[i] and group1[j]
                    if ok:
                        bonds.add((i - m0, j - m0))
        key = (tuple((structure.resnames[i], structure.names[i], i - m0) for i in atoms), tuple(sorted(bonds)))
        for t in types:
            if t["key"] == key:
                t["mols"].append(atoms)
                break
        else:
            resn = []
            for i in atoms:
                if structure.resnames[i] not in resn:
                    resn.append(structure.resnames[i])
            types.append({"key": key, "name": "-".join(resn), "mols": [atoms], "bonds": sorted(bonds), "m0": m0})
    return _solve_name_conflicts([t for t in types if t["bonds"]])


def _solve_name_conflicts(types: List[dict]) -> List[dict]:
    """classify.rs:267-294: molecule types that share a name but differ in topology are numbered in order of
    appearance (POPE1, POPE2, ...)."""
    counts: Dict[str, int] = {}
    for t in types:
        counts[t["name"]] = counts.get(t["name"], 0) + 1
    seen: Dict[str, int] = {}
    for t in types:
        if counts[t["name"]] > 1:
            seen[t["name"]] = seen.get(t["name"], 0) + 1
            t["name"] = f'{t["name"]}{seen[t["name"]]}'
    return types


def build_tables(structure: Structure, analysis: str, sel1: np.ndarray, sel2: Optional[np.ndarray] = None,
                 master: Optional[np.ndarray] = None, leaflets: Optional[dict] = None, handle_pbc: bool = True,
                 normal=(0.0, 0.0, 1.0), ordermap: Optional[OrderMap] = None, timewise: bool = False,
                 flags: int = 0, geometry=None, dynamic_normal: Optional[dict] = None):
    """analysis: 'aa' (sel1 = heavy atoms, sel2 = hydrogens) or 'cg' (sel1 = beads).
    dynamic_normal: {"heads": mask, "radius": r} — DynamicNormal::new(heads, radius), input/membrane_normal.rs.
    `master`: boolean mask of the atoms present in the coordinate frames handed to the engine (the
    "Master" group, common.rs:92-103); default = union of every selection involved.
    leaflets: {"method": ..., "membrane": mask, "heads": mask, "methyls": mask, "frequency": n, "flip": bool,
               "radius": r}.
    Returns (Tables, [MolLabels], master index array)."""
    same = analysis == "cg"
    if same:
        sel2 = sel1
    types = classify(structure, sel1, sel2, same)
    lf = leaflets or {}
    if master is None:
        master = sel1 | sel2
        for k in ("membrane", "heads", "methyls"):
            if lf.get(k) is not None:
                master = master | lf[k]
        if dynamic_normal is not None:
            master = master | dynamic_normal["heads"]
    midx = np.flatnonzero(master)
    remap = -np.ones(structure.n_atoms, dtype=np.int64)
    remap[midx] = np.arange(len(midx))
    mts, labels = [], []
    slot0 = 0
    for t in types:
        mols = t["mols"]
        n_mol = len(mols)
        bonds = np.zeros((len(t["bonds"]), n_mol, 2), dtype=np.uint32)
        for k, atoms in enumerate(mols):
            m0 = atoms[0]
            for b, (r1, r2) in enumerate(t["bonds"]):
                i, j = remap[m0 + r1], remap[m0 + r2]
                assert i >= 0 and j >= 0, "order atoms must be part of the master group"
                bonds[b, k] = (min(i, j), max(i, j))
        heads = methyls = None
        if lf.get("method", LEAFLETS_NONE) in (LEAFLETS_GLOBAL, LEAFLETS_LOCAL, LEAFLETS_INDIVIDUAL):
            heads = np.zeros(n_mol, dtype=np.uint32)
            for k, atoms in enumerate(mols):
                hs = [a for a in atoms if lf["heads"][a]]
                assert len(hs) == 1, f"molecule type {t['name']}: {len(hs)} head identifiers (need exactly 1)"
                heads[k] = remap[hs[0]]
            if lf["method"] == LEAFLETS_INDIVIDUAL:
                ml = [[remap[a] for a in atoms if lf["methyls"][a]] for atoms in mols]
                assert len(set(len(x) for x in ml)) == 1 and len(ml[0]) > 0
                methyls = np.array(ml, dtype=np.uint32)
        mts.append(MolType(n_molecules=n_mol, bonds=bonds, heads=heads, methyls=methyls, name=t["name"],
                           normal_heads=_normal_heads(mols, dynamic_normal, remap, t["name"])))
        m0 = t["m0"]
        bl = [BondLabel(r1, structure.names[m0 + r1], r2, structure.names[m0 + r2], structure.resnames[m0 + r1],
                        structure.resnames[m0 + r2]) for r1, r2 in t["bonds"]]
        heavy = []
        if analysis == "aa":
            seen = set()
            for r1, r2 in t["bonds"]:
                for r in (r1, r2):
                    if sel1[m0 + r] and r not in seen:
                        seen.add(r)
                        heavy.append((r, structure.names[m0 + r], structure.resnames[m0 + r]))
            heavy.sort()
        labels.append(MolLabels(t["name"], bl, heavy, n_mol, slot0))
        slot0 += len(bl)
    leaf = Leaflets()
    if lf.get("method", LEAFLETS_NONE) != LEAFLETS_NONE:
        mem = None
        if lf["method"] in (LEAFLETS_GLOBAL, LEAFLETS_LOCAL):
            mem = remap[np.flatnonzero(lf["membrane"])].astype(np.uint32)
        leaf = Leaflets(method=lf["method"], normal_dim=lf.get("normal_dim", 2), frequency=lf.get("frequency", 1),
                        flip=lf.get("flip", False), radius=lf.get("radius", 0.0), membrane=mem)
    tables = Tables(n_atoms=len(midx), molecule_types=mts, handle_pbc=handle_pbc, normal=normal, leaflets=leaf,
                    ordermap=ordermap or OrderMap(), timewise=timewise, flags=flags)
    if geometry is not None:
        tables.geometry = geometry
    _set_dynamic_normal(tables, dynamic_normal, remap)
    return tables, labels, midx


def _normal_heads(mols, dynamic_normal, remap, name):
    """get_reference_head for group "NormalHeads" (normal.rs:145-158): exactly one per molecule."""
    if dynamic_normal is None:
        return None
    out = np.zeros(len(mols), dtype=np.uint32)
    for k, atoms in enumerate(mols):
        hs = [a for a in atoms if dynamic_normal["heads"][a]]
        assert len(hs) == 1, f"molecule type {name}: {len(hs)} normal-head identifiers (need exactly 1)"
        out[k] = remap[hs[0]]
    return out


def _set_dynamic_normal(tables, dynamic_normal, remap):
    if dynamic_normal is None:
        return
    from .abi import DynamicNormal
    cloud = remap[np.flatnonzero(dynamic_normal["heads"])]
    assert (cloud >= 0).all()
    tables.dynamic_normal = DynamicNormal(enabled=True, radius=float(dynamic_normal.get("radius", 2.0)),
                                          cloud=cloud.astype(np.uint32))


@dataclass
class UaCarbonLabel:
    rel: int
    name: str
    kind: int
    n_h: int
    res: str = ""


@dataclass
class UaMolLabels:
    name: str
    carbons: List[UaCarbonLabel]
    n_molecules: int = 0
    slot0: int = 0


def build_tables_ua(structure: Structure, saturated: np.ndarray, unsaturated: np.ndarray, master: np.ndarray,
                    leaflets: Optional[dict] = None, handle_pbc: bool = True, normal=(0.0, 0.0, 1.0),
                    ordermap: Optional[OrderMap] = None, timewise: bool = False, flags: int = 0,
                    dynamic_normal: Optional[dict] = None, ignore: Optional[np.ndarray] = None, geometry=None):
    """AtomBasedClassifier + UAOrderAtoms (classify.rs, uaorder.rs:454-665): the number of bonded atoms of
    a carbon decides how many hydrogens are built; helpers = bonded atoms in index order; a methyl's second
    helper is the first neighbour of helper1 that is not the methyl itself (uaorder.rs:609-629).
    `ignore` (uaorder.rs:193-224, 583-587): atoms that do not count as bonded neighbours of a carbon (the explicit
    hydrogens of an all-atom system); they stay part of the molecule, so relative indices count them."""
    from .abi import UA_CH1_SAT, UA_CH1_UNSAT, UA_CH2, UA_CH3, UA_N_H
    adj = structure.bonds
    order = saturated | unsaturated
    visited: set = set()
    types: List[dict] = []
    for a in np.flatnonzero(order):
        a = int(a)
        if a in visited:
            continue
        atoms = _molecule_of(adj, a, visited)
        m0 = atoms[0]
        key = tuple((structure.resnames[i], structure.names[i], i - m0, tuple(j - m0 for j in adj[i])) for i in atoms)
        for t in types:
            if t["key"] == key:
                t["mols"].append(atoms)
                break
        else:
            resn = []
            for i in atoms:
                if structure.resnames[i] not in resn:
                    resn.append(structure.resnames[i])
            types.append({"key": key, "name": "-".join(resn), "mols": [atoms], "m0": m0})
    _solve_name_conflicts(types)
    midx = np.flatnonzero(master)
    remap = -np.ones(structure.n_atoms, dtype=np.int64)
    remap[midx] = np.arange(len(midx))
    lf = leaflets or {}
    mts, labels, slot0 = [], [], 0
    for t in types:
        mols, m0 = t["mols"], t["m0"]
        carbons, ua_atoms = [], []
        for i in mols[0]:
            if not order[i]:
                continue
            bonded = adj[i] if ignore is None else [j for j in adj[i] if not ignore[j]]
            missing = max(0, 4 - len(bonded))
            quad = None
            if saturated[i] and missing == 1:
                kind, quad = UA_CH1_SAT, (bonded[0], bonded[1], bonded[2], i)
            elif saturated[i] and missing == 2:
                kind, quad = UA_CH2, (bonded[0], i, bonded[1], i)
            elif saturated[i] and missing == 3:
                h1 = bonded[0]
                h2 = next((j for j in adj[h1] if j != i), None)
                if h2 is None:
                    continue
                kind, quad = UA_CH3, (h1, i, h2, i)
            elif unsaturated[i] and missing == 2:
                kind, quad = UA_CH1_UNSAT, (bonded[0], i, bonded[1], i)
            else:
                continue
            idx = np.zeros((len(mols), 4), dtype=np.uint32)
            for k, atoms in enumerate(mols):
                off = atoms[0] - m0
                idx[k] = [remap[q + off] for q in quad]
            assert (idx >= 0).all()
            ua_atoms.append((kind, idx))
            carbons.append(UaCarbonLabel(i - m0, structure.names[i], kind, UA_N_H[kind], structure.resnames[i]))
        if not ua_atoms:
            continue
        heads = methyls = None
        if lf.get("method", LEAFLETS_NONE) in (LEAFLETS_GLOBAL, LEAFLETS_LOCAL, LEAFLETS_INDIVIDUAL):
            heads = np.array([remap[[a for a in atoms if lf["heads"][a]][0]] for atoms in mols], dtype=np.uint32)
            if lf["method"] == LEAFLETS_INDIVIDUAL:
                methyls = np.array([[remap[a] for a in atoms if lf["methyls"][a]] for atoms in mols], dtype=np.uint32)
        mts.append(MolType(n_molecules=len(mols), ua_atoms=ua_atoms, heads=heads, methyls=methyls, name=t["name"],
                           normal_heads=_normal_heads(mols, dynamic_normal, remap, t["name"])))
        labels.append(UaMolLabels(t["name"], carbons, len(mols), slot0))
        slot0 += sum(c.n_h for c in carbons)
    leaf = Leaflets()
    if lf.get("method", LEAFLETS_NONE) != LEAFLETS_NONE:
        mem = None
        if lf["method"] in (LEAFLETS_GLOBAL, LEAFLETS_LOCAL):
            mem = remap[np.flatnonzero(lf["membrane"])].astype(np.uint32)
        leaf = Leaflets(method=lf["method"], normal_dim=lf.get("normal_dim", 2), frequency=lf.get("frequency", 1),
                        flip=lf.get("flip", False), radius=lf.get("radius", 0.0), membrane=mem)
    tables = Tables(n_atoms=len(midx), molecule_types=mts, handle_pbc=handle_pbc, normal=normal, leaflets=leaf,
                    ordermap=ordermap or OrderMap(), timewise=timewise, flags=flags)
    if geometry is not None:
        tables.geometry = geometry
    _set_dynamic_normal(tables, dynamic_normal, remap)
    return tables, labels, midx


def _collector(res: Results, leaflets: bool, sign: float, min_samples: int, timewise, n_blocks: int):
    """Value of a set of accumulator slots as the writers print it: mean of the summed accumulators (x sign, 4
    decimals, NaN below min_samples); with timewise = (tw_sums, tw_counts) [frames][3][n_acc] the `estimate_error`
    layout {mean, error}, the error from the members' per-frame rows added up (TimeWiseData merge, timewise.rs:31-76)."""
    which = ["total", "upper", "lower"] if leaflets else ["total"]

    def coll(slots):
        out = {}
        for w, key in enumerate(which):
            s = int(sum(int(res.sums[w, k]) for k in slots))
            n = int(sum(int(res.counts[w, k]) for k in slots))
            v = _mean_ticks(s, n, min_samples)
            mean = round4(sign * v) if v == v else float("nan")
            if timewise is None:
                out[key] = mean
            else:
                ts = np.asarray(timewise[0])[:, w, slots].sum(axis=1)
                tc = np.asarray(timewise[1])[:, w, slots].sum(axis=1)
                e = estimate_error(ts, tc, n_blocks) if mean == mean else float("nan")   # below min_samples: no error either
                out[key] = {"mean": mean, "error": round4(e) if e == e else float("nan")}
        return out
    return coll


def results_tree_ua(res: Results, labels: Sequence[UaMolLabels], leaflets: bool, min_samples: int = 1,
                    timewise=None, n_blocks: int = 5) -> dict:
    """UA YAML shape (uaresults): per carbon `total` (+ upper/lower) and `bonds`: list per virtual hydrogen."""
    coll = _collector(res, leaflets, -1.0, min_samples, timewise, n_blocks)

    tree: Dict[str, object] = {}
    all_slots: List[int] = []
    for ml in labels:
        slot = ml.slot0
        op: Dict[str, object] = {}
        mol_slots = []
        for c in ml.carbons:
            slots = list(range(slot, slot + c.n_h))
            slot += c.n_h
            mol_slots += slots
            entry = dict(coll(slots))
            entry["bonds"] = [coll([k]) for k in slots]
            op[f"{c.res or ml.name} {c.name} ({c.rel})"] = entry
        all_slots += mol_slots
        tree[ml.name] = {"average order": coll(mol_slots), "order parameters": op}
    return {"average order": coll(all_slots), **tree}


# ---- finalisation (presentation layer arithmetic) -------------------------------------------
def _mean_ticks(s: int, n: int, min_samples: int = 1) -> float:
    """AnalysisOrder::calc_order (order.rs:101-107): truncating i64 division, /1e6, as f32."""
    if n < max(1, min_samples):
        return float("nan")
    q = abs(int(s)) // int(n)
    q = -q if s < 0 else q
    return float(np.float32(q / 1e6))


def estimate_error(sums: np.ndarray, counts: np.ndarray, n_blocks: int = 5) -> float:
    """TimeWiseData::estimate_error (timewise.rs:191-231) on per-frame (tick sum, sample count) rows: blocks of
    len // n_blocks frames (the remainder is dropped), per-block mean by the truncating i64 division, sample
    standard deviation in f32 (crate `statistical`); NaN if a block has no samples."""
    n = len(sums)
    if n == 0 or n_blocks < 2 or n // n_blocks == 0:
        return float("nan")
    bs = n // n_blocks
    o = []
    for b in range(n_blocks):
        s_ = int(np.asarray(sums[b * bs:(b + 1) * bs], dtype=np.int64).sum())
        c_ = int(np.asarray(counts[b * bs:(b + 1) * bs], dtype=np.uint64).sum())
        if c_ == 0:
            return float("nan")
        q = abs(s_) // c_
        o.append(np.float32((-q if s_ < 0 else q) / 1e6))
    o = np.array(o, dtype=np.float32)
    mean = np.float32(0)
    for x in o:
        mean = np.float32(mean + x)
    mean = np.float32(mean / np.float32(n_blocks))
    var = np.float32(0)
    for x in o:
        d = np.float32(mean - x)
        var = np.float32(var + np.float32(d * d))
    return float(np.sqrt(np.float32(var / np.float32(n_blocks - 1)), dtype=np.float32))


def round4(x: float) -> float:
    """RoundTo4 (presentation/mod.rs:496-504): (x as f64 * 10000).round() / 10000, half away from zero."""
    if x != x:
        return x
    v = float(x) * 10000.0
    r = np.floor(abs(v) + 0.5) * (1.0 if v >= 0 else -1.0)
    return float(r / 10000.0)


def results_tree(res: Results, labels: Sequence[MolLabels], analysis: str, leaflets: bool, min_samples: int = 1,
                 timewise=None, n_blocks: int = 5) -> dict:
    """Nested dict shaped like gorder's YAML output (aaresults.rs:47-61, cgresults.rs:202-217), values
    rounded to 4 decimals; AA reports -S (presentation/mod.rs:618-625).
    timewise = (tw_sums, tw_counts) [frames][3][n_acc] switches to the `estimate_error` layout: every value
    becomes {mean, error}; an aggregate's error comes from its members' per-frame rows added up
    (TimeWiseData merge, timewise.rs:31-76)."""
    sign = -1.0 if analysis in ("aa", "ua") else 1.0
    coll = _collector(res, leaflets, sign, min_samples, timewise, n_blocks)

    tree: Dict[str, object] = {}
    all_slots: List[int] = []
    for ml in labels:
        slots = list(range(ml.slot0, ml.slot0 + len(ml.bonds)))
        all_slots += slots
        mol: Dict[str, object] = {"average order": coll(slots)}
        op: Dict[str, object] = {}
        if analysis == "aa":
            for rel, name, resn in ml.heavy_atoms:
                mine = [(k, b) for k, b in enumerate(ml.bonds) if b.rel1 == rel or b.rel2 == rel]
                if not mine:
                    continue
                entry = dict(coll([ml.slot0 + k for k, _ in mine]))
                bonds = {}
                for k, b in mine:
                    orel, oname, ores = (b.rel2, b.name2, b.res2) if b.rel1 == rel else (b.rel1, b.name1, b.res1)
                    bonds[f"{ores or ml.name} {oname} ({orel})"] = coll([ml.slot0 + k])
                entry["bonds"] = bonds
                op[f"{resn or ml.name} {name} ({rel})"] = entry
        else:
            for k, b in enumerate(ml.bonds):
                op[f"{b.res1 or ml.name} {b.name1} ({b.rel1}) - {b.res2 or ml.name} {b.name2} ({b.rel2})"] = coll([ml.slot0 + k])
        mol["order parameters"] = op
        tree[ml.name] = mol
    return {"average order": coll(all_slots), **tree}


def compare_trees(got, want, tol=2e-4, path="") -> List[str]:
    """Numeric comparison in the spirit of the reference's own `assert_eq_order`
    (/root/reference/tests/common/mod.rs:35-51, 139-150: floats within 2e-4). Returns mismatches."""
    bad: List[str] = []
    if isinstance(want, dict):
        if not isinstance(got, dict):
            return [f"{path}: expected mapping"]
        for k in want:
            if k not in got:
                bad.append(f"{path}/{k}: missing")
            else:
                bad += compare_trees(got[k], want[k], tol, f"{path}/{k}")
        for k in got:
            if k not in want:
                bad.append(f"{path}/{k}: unexpected")
        return bad
    if isinstance(want, (list, tuple)):
        if not isinstance(got, (list, tuple)) or len(got) != len(want):
            return [f"{path}: expected a list of {len(want)}"]
        for k, (g_, w_) in enumerate(zip(got, want)):
            bad += compare_trees(g_, w_, tol, f"{path}[{k}]")
        return bad
    try:
        g, w = float(got), float(want)
    except (TypeError, ValueError):
        return [] if got == want else [f"{path}: {got!r} != {want!r}"]
    if (g != g) and (w != w):
        return []
    # the reference parses both tokens as f32 and applies approx's `abs_diff <= epsilon` in f32
    if not np.abs(np.float32(g) - np.float32(w)) <= np.float32(tol):
        bad.append(f"{path}: {g} vs {w}")
    return bad
