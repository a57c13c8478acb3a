"""CPU tests of the C-ABI library: it loads, exports every symbol the header declares, its host logic
(the tile planner) is correct, and it FAILS LOUDLY without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from gorder_amd import abi, synthetic
from gorder_amd.abi import LEAFLETS_GLOBAL, MolType, Tables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "gorder_hip.h")).read()
    return sorted(set(re.findall(r"\b(gorder_hip_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built):
    lib = abi.load_library()
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/gorder_hip.h but not exported"
    assert set(names) == set(abi._EXPORTS)


def test_reader_library_exports_every_declared_symbol(built):
    lib = abi.load_library()
    src = open(os.path.join(ROOT, "include", "gorder_xtc.h")).read()
    names = sorted(set(re.findall(r"\b(gorder_xtc_[a-z_]+)\s*\(", src)))
    assert len(names) >= 10
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/gorder_xtc.h but not exported"


def test_strerror(built):
    lib = abi.load_library()
    assert lib.gorder_hip_strerror(0) == b"ok"
    assert b"orthogonal" in lib.gorder_hip_strerror(abi.ERR_NOT_ORTHOGONAL_BOX)
    assert b"no CPU fallback" in lib.gorder_hip_strerror(abi.ERR_NO_DEVICE)


def test_no_gpu_means_loud_failure(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    system = synthetic.cg_membrane(n_lipids=16)
    with pytest.raises(abi.GorderHipError) as e:
        abi.HipEngine(system.tables)
    assert e.value.status == abi.ERR_NO_DEVICE


@pytest.mark.parametrize("make", [
    lambda: synthetic.aa_membrane(256),
    lambda: synthetic.aa_membrane(7),
    lambda: synthetic.cg_membrane(3072, leaflets=LEAFLETS_GLOBAL),
    lambda: synthetic.cg_membrane(333, n_types=3),
])
def test_plan_covers_every_sample_once(built, make):
    system = make()
    plan = abi.plan_tables(system.tables)
    assert plan["selfcheck"] == 0
    assert plan["n_direct_items"] == 0
    assert plan["max_window_atoms"] <= 1024
    n = system.tables.n_samples_per_frame
    assert plan["n_tiles"] >= (n + 255) // 256


def test_plan_scattered_bonds_go_direct(built):
    # bonds between far-apart atoms cannot share an LDS window
    n_atoms = 50000
    rng = np.random.default_rng(0)
    far = np.stack([rng.integers(0, 1000, 40), rng.integers(40000, 50000, 40)], axis=1)
    near = np.stack([np.arange(2000, 2040), np.arange(2001, 2041)], axis=1)
    bonds = np.stack([far, near]).astype(np.uint32)      # 2 bond types x 40 molecules
    t = Tables(n_atoms=n_atoms, molecule_types=[MolType(n_molecules=40, bonds=bonds)])
    plan = abi.plan_tables(t)
    assert plan["selfcheck"] == 0
    assert plan["n_direct_items"] == 40
    assert plan["n_tiles"] >= 1


def test_plan_rejects_bad_tables(built):
    bonds = np.array([[[0, 99]]], dtype=np.uint32)
    t = Tables(n_atoms=10, molecule_types=[MolType(n_molecules=1, bonds=bonds)])
    with pytest.raises(abi.GorderHipError) as e:
        abi.plan_tables(t)
    assert e.value.status == abi.ERR_INVALID_ARGUMENT
    bonds = np.array([[[3, 3]]], dtype=np.uint32)
    t = Tables(n_atoms=10, molecule_types=[MolType(n_molecules=1, bonds=bonds)])
    with pytest.raises(abi.GorderHipError):
        abi.plan_tables(t)


def test_force_direct_env(built, monkeypatch):
    system = synthetic.cg_membrane(64)
    monkeypatch.setenv("GORDER_HIP_FORCE_DIRECT", "1")
    plan = abi.plan_tables(system.tables)
    assert plan["n_tiles"] == 0 and plan["n_direct_items"] == 64 * 11 and plan["selfcheck"] == 0


def test_plan_for_global_leaflets_in_one_read(built):
    """Plan::spec_ok (plan.h): the order kernel can sum the membrane group on the way when the group is one range of atoms
    the tiles' windows cover — extended where needed, the first tile backwards too — and every head lies in it."""
    # the usual case: the group is every atom of the lipids
    assert abi.plan_tables(synthetic.aa_membrane(40, leaflets=LEAFLETS_GLOBAL).tables)["leaflets_one_read"] == 1
    assert abi.plan_tables(synthetic.cg_membrane(300, leaflets=LEAFLETS_GLOBAL, n_types=3).tables)["leaflets_one_read"] == 1
    assert abi.plan_tables(synthetic.cg_membrane(300).tables)["leaflets_one_read"] == 0          # no leaflets
    # the first bead of every lipid in no bond: the first tile's window is moved back to the group's first atom, the
    # others reach forward over the beads between them — and the plan still covers every sample exactly once
    s = synthetic.cg_membrane(200, leaflets=LEAFLETS_GLOBAL)
    mt = s.tables.molecule_types[0]
    mt.bonds = mt.bonds[1:]
    plan = abi.plan_tables(s.tables)
    assert plan["leaflets_one_read"] == 1 and plan["selfcheck"] == 0
    # an index list that is not a range: the two-kernel path
    s = synthetic.cg_membrane(100, leaflets=LEAFLETS_GLOBAL)
    s.tables.leaflets.membrane = np.arange(0, s.n_atoms, 2, dtype=np.uint32)
    assert abi.plan_tables(s.tables)["leaflets_one_read"] == 0
    # a head outside the group
    s = synthetic.cg_membrane(100, leaflets=LEAFLETS_GLOBAL)
    s.tables.leaflets.membrane = np.arange(24, s.n_atoms, dtype=np.uint32)
    assert abi.plan_tables(s.tables)["leaflets_one_read"] == 0
    # water behind the lipids: the group ends before the frame does
    s = synthetic.cg_membrane(100, leaflets=LEAFLETS_GLOBAL)
    s.tables.n_atoms += 500
    plan = abi.plan_tables(s.tables)
    assert plan["leaflets_one_read"] == 1 and plan["selfcheck"] == 0
