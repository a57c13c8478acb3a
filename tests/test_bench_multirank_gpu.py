"""bench.py's N > 1 path, rehearsed on ONE GPU: two ranks launched exactly as the driver launches them
(torch.distributed.run), both on cuda:0 with gloo instead of RCCL (GORDER_BENCH_REHEARSAL=1).  Checks the frame
sharding, the accumulator (and ordermap) reduce and the one-JSON-line contract."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("workload,frames", [("aa256", 600), ("ua256-maps", 200)])
def test_two_ranks_on_one_gpu(built, workload, frames):
    env = dict(os.environ, GORDER_BENCH_REHEARSAL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "2", "--frames", str(frames), "--workload", workload]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                     # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["unit"] == "frames/s"
    # every frame of both ranks' shards (warm-up passes included) arrived in the reduced accumulators
    assert out["sanity"]["frames_accumulated"] == out["sanity"]["expected"] == (3 + 2) * frames * 2
    assert out["sanity"]["counts_ok"]
    assert out["value"] > 0 and "cpu_baseline" not in out      # the CPU baseline is an N = 1 thing
