"""Synthetic membranes of SURVEY.md §8(d): index tables + coordinate generators.

The trajectories behind the reference's published benchmarks are not in the repository
(/root/reference/validation/*/files/README.md), so workloads are generated: same molecule counts,
atoms per lipid and bonds per lipid as the reference's validation systems
(validation/aa_charmm/gorder_nthreads_1/order.yaml: 64 C-H bonds per POPC;
 validation/cg_martini/gorder_nthreads_1/order.yaml:7-28: 11 Martini POPC bonds).

A "system" is (Tables, base frame [N,3] float32, box [3] float32).  Frames are the base frame plus
Gaussian jitter, wrapped into the box so that bonds straddle the periodic faces.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from .abi import (LEAFLETS_GLOBAL, LEAFLETS_INDIVIDUAL, LEAFLETS_LOCAL, LEAFLETS_NONE, Leaflets, MolType,
                  OrderMap, Tables)


@dataclass
class System:
    name: str
    tables: Tables
    base: np.ndarray      # [N, 3] float32
    box: np.ndarray       # [3] float32
    jitter: float = 0.02

    @property
    def n_atoms(self) -> int:
        return self.tables.n_atoms

    @property
    def bytes_per_frame(self) -> int:
        """Algorithmic HBM bytes per frame: every coordinate once + the 3 box edges."""
        return 12 * self.tables.n_atoms + 12

    def box9(self, n_frames: int) -> np.ndarray:
        b = np.zeros((n_frames, 3, 3), dtype=np.float32)
        b[:, 0, 0], b[:, 1, 1], b[:, 2, 2] = self.box
        return b

    def frames(self, n_frames: int, seed: int = 0, first: int = 0) -> np.ndarray:
        """Host frames [n_frames, N, 3]; frame k depends only on (seed, first + k)."""
        out = np.empty((n_frames, self.n_atoms, 3), dtype=np.float32)
        for k in range(n_frames):
            rng = np.random.default_rng([seed, first + k])
            x = self.base + rng.normal(0.0, self.jitter, size=self.base.shape).astype(np.float32)
            out[k] = np.mod(x, self.box).astype(np.float32)
        return out

    def frames_device(self, n_frames: int, seed: int = 0, device="cuda"):
        """Frames generated on the GPU (torch is plumbing: device memory + RNG)."""
        import torch
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        base = torch.from_numpy(self.base).to(device)
        box = torch.from_numpy(self.box).to(device)
        xyz = torch.empty((n_frames, self.n_atoms, 3), dtype=torch.float32, device=device)
        step = max(1, (1 << 26) // max(1, self.n_atoms * 3))
        for s in range(0, n_frames, step):
            e = min(n_frames, s + step)
            noise = torch.randn((e - s, self.n_atoms, 3), generator=g, device=device, dtype=torch.float32)
            xyz[s:e] = torch.remainder(base.unsqueeze(0) + self.jitter * noise, box)
        box9 = torch.zeros((n_frames, 3, 3), dtype=torch.float32, device=device)
        box9[:, 0, 0], box9[:, 1, 1], box9[:, 2, 2] = float(self.box[0]), float(self.box[1]), float(self.box[2])
        return xyz, box9


def _unit_vectors(rng, n):
    v = rng.normal(size=(n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


# ---- all-atom POPC-like lipid: 34 carbons + 64 hydrogens, 64 C-H bond types --------------------
def _aa_template():
    # chain A: C(0H) + 6 CH2 + 2 CH (double bond) + 7 CH2 + CH3 = 17 carbons, 12+2+14+3 = 31 H
    # chain B: C(0H) + 15 CH2 + CH3                         = 17 carbons, 30+3     = 33 H
    h_per_c = [0] + [2] * 6 + [1, 1] + [2] * 7 + [3] + [0] + [2] * 15 + [3]
    assert len(h_per_c) == 34 and sum(h_per_c) == 64
    carbons, bonds, owner = [], [], []
    idx = 0
    for c, nh in enumerate(h_per_c):
        carbons.append(idx)
        ci = idx
        idx += 1
        for _ in range(nh):
            bonds.append((ci, idx))
            owner.append(c)
            idx += 1
    return np.array(carbons), np.array(bonds, dtype=np.uint32), idx, h_per_c


def aa_membrane(n_lipids: int = 256, box=(9.0, 9.0, 8.0), seed: int = 20240213,
                leaflets: int = LEAFLETS_NONE, frequency: int = 1, flip: bool = False,
                ordermap: Optional[OrderMap] = None, timewise: bool = False, handle_pbc: bool = True,
                normal=(0.0, 0.0, 1.0), radius: float = 2.0) -> System:
    """V-AA: n_lipids x 98 selected atoms (34 C + 64 H), 64 bond types."""
    rng = np.random.default_rng(seed)
    carbons, tbonds, apl, h_per_c = _aa_template()
    box = np.asarray(box, dtype=np.float32)
    n_atoms = n_lipids * apl
    base = np.zeros((n_atoms, 3), dtype=np.float64)
    zc = box[2] / 2
    side = int(np.ceil(np.sqrt((n_lipids + 1) // 2)))
    for m in range(n_lipids):
        upper = m % 2 == 0
        g = m // 2
        ox = (g % side + 0.5) * box[0] / side + rng.normal(0, 0.1)
        oy = (g // side + 0.5) * box[1] / side + rng.normal(0, 0.1)
        sgn = 1.0 if upper else -1.0
        a0 = m * apl
        for chain in range(2):
            pos = np.array([ox + 0.25 * chain, oy, zc + sgn * 1.8])
            for k in range(17):
                c = chain * 17 + k
                step = np.array([rng.normal(0, 0.06), rng.normal(0, 0.06), -sgn * 0.1])
                pos = pos + step
                base[a0 + carbons[c]] = pos
        # hydrogens: 0.109 nm from their carbon in a random direction
        u = _unit_vectors(rng, len(tbonds))
        for b, (ci, hi) in enumerate(tbonds):
            base[a0 + hi] = base[a0 + ci] + 0.109 * u[b]
    base = np.mod(base, box).astype(np.float32)
    bonds = np.zeros((len(tbonds), n_lipids, 2), dtype=np.uint32)
    for m in range(n_lipids):
        bonds[:, m, :] = tbonds + m * apl
    heads = (np.arange(n_lipids) * apl + carbons[0]).astype(np.uint32)
    methyls = np.stack([np.arange(n_lipids) * apl + carbons[16], np.arange(n_lipids) * apl + carbons[33]],
                       axis=1).astype(np.uint32)
    lf = Leaflets(method=leaflets, normal_dim=2, frequency=frequency, flip=flip, radius=radius,
                  membrane=np.arange(n_atoms, dtype=np.uint32) if leaflets in (LEAFLETS_GLOBAL, LEAFLETS_LOCAL) else None)
    mt = MolType(n_molecules=n_lipids, bonds=bonds, heads=heads if leaflets else None,
                 methyls=methyls if leaflets == LEAFLETS_INDIVIDUAL else None, name="POPC")
    t = Tables(n_atoms=n_atoms, molecule_types=[mt], handle_pbc=handle_pbc, normal=normal, leaflets=lf,
               ordermap=ordermap or OrderMap(), timewise=timewise)
    return System(f"aa{n_lipids}", t, base, box)


# ---- Martini-3 POPC-like lipid: 12 beads, 11 bond types ---------------------------------------
_CG_BONDS = np.array([(0, 1), (1, 2), (2, 3), (2, 4), (4, 5), (5, 6), (6, 7), (3, 8), (8, 9), (9, 10), (10, 11)],
                     dtype=np.uint32)
# depth of each bead below the head plane, in bond lengths (NC3 PO4 GL1 GL2 C1A D2A C3A C4A C1B C2B C3B C4B)
_CG_DEPTH = np.array([-1.0, 0.0, 1.0, 1.2, 2.0, 3.0, 4.0, 5.0, 2.2, 3.2, 4.2, 5.2])
_CG_SIDE = np.array([0.0, 0.0, 0.0, 0.3, -0.1, -0.1, -0.1, -0.1, 0.4, 0.4, 0.4, 0.4])


def cg_membrane(n_lipids: int = 3072, box=None, seed: int = 7, leaflets: int = LEAFLETS_NONE,
                frequency: int = 1, flip: bool = False, radius: float = 2.5, n_types: int = 1,
                ordermap: Optional[OrderMap] = None, timewise: bool = False, handle_pbc: bool = True,
                normal=(0.0, 0.0, 1.0)) -> System:
    """CG bilayer: n_lipids x 12 beads, 11 bonds each.
    n_types > 1 splits the lipids round-robin into several molecule types (mixed membranes)."""
    rng = np.random.default_rng(seed)
    per_leaflet = (n_lipids + 1) // 2
    nx = int(np.ceil(np.sqrt(per_leaflet)))
    ny = (per_leaflet + nx - 1) // nx
    if box is None:
        box = (nx * 0.8, ny * 0.8, 10.0)
    box = np.asarray(box, dtype=np.float32)
    bl = 0.47
    zc = box[2] / 2
    n_atoms = n_lipids * 12
    base = np.zeros((n_lipids, 12, 3), dtype=np.float64)
    m = np.arange(n_lipids)
    upper = (m % 2 == 0)
    g = m // 2
    ox = (g % nx + 0.5) * box[0] / nx + rng.normal(0, 0.08, n_lipids)
    oy = (g // nx + 0.5) * box[1] / ny + rng.normal(0, 0.08, n_lipids)
    sgn = np.where(upper, 1.0, -1.0)
    for b in range(12):
        base[:, b, 0] = ox + _CG_SIDE[b] + rng.normal(0, 0.05, n_lipids)
        base[:, b, 1] = oy + rng.normal(0, 0.05, n_lipids)
        base[:, b, 2] = zc + sgn * (2.0 - _CG_DEPTH[b] * bl * 0.75) + rng.normal(0, 0.05, n_lipids)
    base = np.mod(base.reshape(n_atoms, 3), box).astype(np.float32)
    mts = []
    for ty in range(n_types):
        ids = m[m % n_types == ty]
        bonds = (_CG_BONDS[:, None, :] + (ids * 12)[None, :, None]).astype(np.uint32)
        heads = (ids * 12 + 1).astype(np.uint32)
        methyls = np.stack([ids * 12 + 7, ids * 12 + 11], axis=1).astype(np.uint32)
        mts.append(MolType(n_molecules=len(ids), bonds=bonds, heads=heads if leaflets else None,
                           methyls=methyls if leaflets == LEAFLETS_INDIVIDUAL else None, name=f"LIP{ty}"))
    lf = Leaflets(method=leaflets, normal_dim=2, frequency=frequency, flip=flip, radius=radius,
                  membrane=np.arange(n_atoms, dtype=np.uint32) if leaflets in (LEAFLETS_GLOBAL, LEAFLETS_LOCAL) else None)
    t = Tables(n_atoms=n_atoms, molecule_types=mts, handle_pbc=handle_pbc, normal=normal, leaflets=lf,
               ordermap=ordermap or OrderMap(), timewise=timewise)
    return System(f"cg{n_lipids}", t, base, box, jitter=0.03)


# ---- united-atom (Berger-like) lipid: 52 atoms, 32 order carbons with virtual hydrogens ----------
def ua_membrane(n_lipids: int = 256, box=(9.0, 9.0, 8.0), seed: int = 11, leaflets: int = LEAFLETS_NONE,
                frequency: int = 1, ordermap: Optional[OrderMap] = None, timewise: bool = False,
                handle_pbc: bool = True, normal=(0.0, 0.0, 1.0), radius: float = 2.0) -> System:
    """V-UA: n_lipids x 52 united atoms; atoms 10..41 are the order carbons:
    26 CH2, 2 CH3, 2 CH1 (double bond) and 2 CH1 (saturated, three helpers) -> 62 virtual C-H bonds."""
    from .abi import UA_CH1_SAT, UA_CH1_UNSAT, UA_CH2, UA_CH3
    rng = np.random.default_rng(seed)
    box = np.asarray(box, dtype=np.float32)
    apl = 52
    n_atoms = n_lipids * apl
    base = np.zeros((n_lipids, apl, 3), dtype=np.float64)
    zc = box[2] / 2
    side = int(np.ceil(np.sqrt((n_lipids + 1) // 2)))
    for m in range(n_lipids):
        sgn = 1.0 if m % 2 == 0 else -1.0
        g = m // 2
        pos = np.array([(g % side + 0.5) * box[0] / side, (g // side + 0.5) * box[1] / side, zc + sgn * 2.2])
        for k in range(apl):
            d = rng.normal(size=3) * 0.6 + np.array([0.0, 0.0, -sgn])
            pos = pos + 0.153 * d / np.linalg.norm(d)
            base[m, k] = pos
    base = np.mod(base.reshape(n_atoms, 3), box).astype(np.float32)
    kinds = {10: UA_CH3, 41: UA_CH3, 20: UA_CH1_UNSAT, 21: UA_CH1_UNSAT, 15: UA_CH1_SAT, 30: UA_CH1_SAT}
    off = np.arange(n_lipids, dtype=np.int64) * apl
    ua_atoms = []
    for c in range(10, 42):
        kind = kinds.get(c, UA_CH2)
        if kind == UA_CH3:           # helper1 = bonded neighbour, helper2 = next atom along the chain
            h1, h2 = (c + 1, c + 2) if c == 10 else (c - 1, c - 2)
            idx = np.stack([off + h1, off + c, off + h2, off + c], axis=1)
        elif kind == UA_CH1_SAT:     # helper1..3, target
            idx = np.stack([off + c - 1, off + c + 1, off + (c + 9) % apl, off + c], axis=1)
        else:
            idx = np.stack([off + c - 1, off + c, off + c + 1, off + c], axis=1)
        ua_atoms.append((kind, idx.astype(np.uint32)))
    heads = (off + 5).astype(np.uint32)
    methyls = np.stack([off + 41, off + 51], axis=1).astype(np.uint32)
    lf = Leaflets(method=leaflets, normal_dim=2, frequency=frequency, radius=radius,
                  membrane=np.arange(n_atoms, dtype=np.uint32) if leaflets in (LEAFLETS_GLOBAL, LEAFLETS_LOCAL) else None)
    mt = MolType(n_molecules=n_lipids, ua_atoms=ua_atoms, heads=heads if leaflets else None,
                 methyls=methyls if leaflets == LEAFLETS_INDIVIDUAL else None, name="POPC")
    t = Tables(n_atoms=n_atoms, molecule_types=[mt], handle_pbc=handle_pbc, normal=normal, leaflets=lf,
               ordermap=ordermap or OrderMap(), timewise=timewise)
    return System(f"ua{n_lipids}", t, base, box, jitter=0.01)
