"""The multi-GPU reduction (SystemTopology::reduce, /root/reference/src/analysis/topology/mod.rs:256-278) with the HIP
engine — not the oracle — as the per-rank engine: contiguous frame shards, each shard on a handle of its
own that accumulates into a caller-owned int64 tensor, the tensors summed as torch.distributed.all_reduce sums them,
the result compared with ONE handle that saw every frame and with the oracle.  One GPU: the "ranks" are handles of one
process, and — in the second test — two processes launched with a gloo group that really all-reduce."""
import os
import sys

import numpy as np
import pytest

from gorder_amd import HipEngine, synthetic
from gorder_amd.abi import LEAFLETS_GLOBAL, LEAFLETS_INDIVIDUAL, LEAFLETS_LOCAL, OrderMap
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def shard_bounds(n_frames, rank, world):
    edges = np.linspace(0, n_frames, world + 1).astype(int)
    return int(edges[rank]), int(edges[rank + 1])


def flipping_membrane(leaflets, frequency, n=23, **kw):
    system = synthetic.cg_membrane(90, leaflets=leaflets, frequency=frequency, n_types=2, **kw)
    xyz, box = system.frames(n, seed=5), system.box9(n)
    xyz[9:, :120, 2] = system.box[2] - xyz[9:, :120, 2]      # ten lipids change sides: the assignment frame matters
    return system, xyz, box


def unpack(acc, n_acc):
    """The packed block of gorder_hip_accumulators_device: sum_total, sum_upper, cnt_total, cnt_upper, total_frames."""
    a = acc.cpu().numpy()
    st, su, ct, cu = (a[k * n_acc:(k + 1) * n_acc] for k in range(4))
    return np.stack([st, su, st - su]), np.stack([ct, cu, ct - cu]).astype(np.uint64), int(a[4 * n_acc])


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("method,frequency", [(LEAFLETS_GLOBAL, 4), (LEAFLETS_LOCAL, 0), (LEAFLETS_INDIVIDUAL, 1)])
def test_hip_shards_reduce_to_the_whole(built, world, method, frequency):
    import torch
    system, xyz, box = flipping_membrane(method, frequency, timewise=True)
    n = xyz.shape[0]
    whole = HipEngine(system.tables)
    whole.submit_host(xyz, box, np.arange(n))
    want = whole.finish()
    ws, wc = whole.timewise(n)
    total, rows_s, rows_c = None, [], []
    for rank in range(world):
        a, b = shard_bounds(n, rank, world)
        eng = HipEngine(system.tables)
        acc = torch.zeros(eng.accumulator_words(), dtype=torch.int64, device="cuda")
        eng.bind_accumulators(acc)
        assign = 0 if frequency == 0 else (a // frequency) * frequency       # Once: frame 0; Every(n): floor(a / n) n
        if assign != a:           # the shard starts between two assignment frames: prime with the one it depends on
            eng.prime_leaflets_device(torch.from_numpy(xyz[assign]).cuda(), torch.from_numpy(box[assign]).cuda(), assign)
        eng.submit_host(xyz[a:b], box[a:b], np.arange(a, b))
        eng.flush()
        eng.synchronize()
        total = acc.clone() if total is None else total + acc                # what dist.all_reduce(acc, SUM) does
        s, c = eng.timewise(b - a)                                           # per-frame rows are gathered, never summed
        rows_s.append(s); rows_c.append(c)
    sums, counts, frames = unpack(total, system.tables.n_acc)
    assert frames == n == want.n_frames
    np.testing.assert_array_equal(sums, want.sums)
    np.testing.assert_array_equal(counts, want.counts)
    np.testing.assert_array_equal(np.concatenate(rows_s), ws)
    np.testing.assert_array_equal(np.concatenate(rows_c), wc)
    assert want.counts[1].sum() > 0 and want.counts[2].sum() > 0
    # ... and the whole is the oracle's (bit for bit when the flags agree; they do here: nobody sits on the mid-plane)
    o = oracle.OracleEngine(system.tables, trig=oracle.TRIG_DIRECT, n_threads=2)
    o.submit(xyz, box, np.arange(n))
    ref = o.finish()
    np.testing.assert_array_equal(sums, ref.sums)
    np.testing.assert_array_equal(counts, ref.counts)


def test_hip_shards_reduce_ordermaps(built):
    """Map::add (ordermap.rs:116-138) across shards of a united-atom system, through the library's own export."""
    import torch
    system = synthetic.ua_membrane(48, leaflets=LEAFLETS_GLOBAL,
                                   ordermap=OrderMap(enabled=True, plane=0, span_x=(0.0, 9.0), span_y=(0.0, 9.0), bin=(0.6, 0.9)))
    n = 17
    xyz, box = system.frames(n, seed=2), system.box9(n)
    whole = HipEngine(system.tables)
    whole.submit_host(xyz, box, np.arange(n))
    want = whole.finish()
    ms = mc = None
    for rank in range(3):
        a, b = shard_bounds(n, rank, 3)
        eng = HipEngine(system.tables)
        eng.submit_host(xyz[a:b], box[a:b], np.arange(a, b))
        s = torch.zeros(want.map_sums.size, dtype=torch.int64, device="cuda")
        c = torch.zeros_like(s)
        eng.export_maps(s, c)
        ms, mc = (s, c) if ms is None else (ms + s, mc + c)
    np.testing.assert_array_equal(ms.cpu().numpy().reshape(want.map_sums.shape), want.map_sums)
    np.testing.assert_array_equal(mc.cpu().numpy().reshape(want.map_counts.shape).astype(np.uint64), want.map_counts)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)      # two ranks on ONE device: RCCL cannot, gloo can
    torch.cuda.set_device(0)
    system, xyz, box = flipping_membrane(LEAFLETS_GLOBAL, 4)
    n = xyz.shape[0]
    a, b = shard_bounds(n, rank, world)
    eng = HipEngine(system.tables)
    acc = torch.zeros(eng.accumulator_words(), dtype=torch.int64, device="cuda")
    eng.bind_accumulators(acc)
    if a % 4:
        p = (a // 4) * 4
        eng.prime_leaflets_device(torch.from_numpy(xyz[p]).cuda(), torch.from_numpy(box[p]).cuda(), p)
    eng.submit_host(xyz[a:b], box[a:b], np.arange(a, b))
    eng.flush()
    eng.synchronize()
    host = acc.cpu()
    dist.all_reduce(host, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(out, host.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_all_reduce_their_hip_accumulators(built, tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "acc.npy")
    mp.get_context("spawn")
    mp.spawn(_worker, args=(2, 29700 + os.getpid() % 200, out), nprocs=2, join=True)
    system, xyz, box = flipping_membrane(LEAFLETS_GLOBAL, 4)
    whole = HipEngine(system.tables)
    whole.submit_host(xyz, box, np.arange(xyz.shape[0]))
    want = whole.finish()
    import torch
    sums, counts, frames = unpack(torch.from_numpy(np.load(out)), system.tables.n_acc)
    assert frames == xyz.shape[0]
    np.testing.assert_array_equal(sums, want.sums)
    np.testing.assert_array_equal(counts, want.counts)
