"""World-size-2 rehearsal (gloo, CPU) of the multi-GPU path: each rank analyses its own frame shard,
then ONE all-reduce of the packed int64 accumulator block (= SystemTopology::reduce,
/root/reference/src/analysis/topology/mod.rs:256-272).  On the GPU node the per-rank engine is the HIP
library and the backend is nccl (= RCCL); here the per-rank engine is the oracle, the reduction logic,
frame-index bookkeeping and leaflet priming are the same code paths bench.py uses."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shard_bounds(n_frames, rank, world):
    """Contiguous frame ranges per rank."""
    edges = np.linspace(0, n_frames, world + 1).astype(int)
    return int(edges[rank]), int(edges[rank + 1])


def pack(res):
    """{i64 sums[3][n_acc], u64 counts[3][n_acc], total_frames} as one int64 vector."""
    return torch.from_numpy(np.concatenate([res.sums.ravel(), res.counts.astype(np.int64).ravel(),
                                            np.array([res.n_frames], dtype=np.int64)]))


def worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gorder_amd import synthetic
    from gorder_amd.abi import LEAFLETS_GLOBAL
    from oracle import oracle
    system = synthetic.cg_membrane(90, leaflets=LEAFLETS_GLOBAL, frequency=4, n_types=2)
    n = 23
    xyz, box = system.frames(n, seed=5), system.box9(n)
    xyz[9:, :120, 2] = system.box[2] - xyz[9:, :120, 2]      # lipids flip: the assignment frame matters
    a, b = shard_bounds(n, rank, world)
    eng = oracle.OracleEngine(system.tables, trig=oracle.TRIG_LIBM)
    if a % 4 != 0:                                            # prime with the assignment frame before the shard
        p = (a // 4) * 4
        eng.prime_leaflets(xyz[p], box[p], p)
    eng.submit(xyz[a:b], box[a:b], np.arange(a, b))
    t = pack(eng.finish())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if rank == 0:
        ref = oracle.OracleEngine(system.tables, trig=oracle.TRIG_LIBM)
        ref.submit(xyz, box, np.arange(n))
        want = pack(ref.finish())
        np.save(out, np.stack([t.numpy(), want.numpy()]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_frame_sharding_plus_one_allreduce(built, tmp_path, world):
    out = str(tmp_path / "r.npy")
    port = 29600 + world + (os.getpid() % 200)
    mp.spawn(worker, args=(world, port, out), nprocs=world, join=True)
    got, want = np.load(out)
    np.testing.assert_array_equal(got, want)
    assert got[-1] == 23
