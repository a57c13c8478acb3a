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


def launch(extra, port):
    env = dict(os.environ, GORDER_BENCH_REHEARSAL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "2"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=420)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                     # rank 0 prints ONE JSON line
    return json.loads(lines[0])


@pytest.mark.parametrize("workload,frames", [("aa256", 600), ("ua256-maps", 200)])
def test_two_ranks_on_one_gpu_weak(built, workload, frames):
    out = launch(["--frames", str(frames), "--workload", workload, "--scaling", "weak"], 29533)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["unit"] == "frames/s"
    # every frame of both ranks' shards (warm-up passes included) arrived in the reduced accumulators
    n_warm = out["config"]["warmup_steps_run"]
    assert n_warm >= 2
    assert out["sanity"]["frames_accumulated"] == out["sanity"]["expected"] == (3 + n_warm) * frames * 2
    assert out["sanity"]["counts_ok"]
    assert out["value"] > 0 and "cpu_baseline" not in out      # the CPU baseline is an N = 1 thing
    assert out["ranks"]["rccl_ranks"] == out["ranks"]["launched"] == 2 and out["ranks"]["backend"] == "gloo"
    assert len(out["ranks"]["kernel_ms_per_step_per_rank"]) == 2 and min(out["ranks"]["kernel_ms_per_step_per_rank"]) > 0


def test_two_ranks_strong_scaling_default(built):
    """`--gpus N` with N > 1 runs the north_star experiment: STRONG scaling of one CG-1M trajectory cut into contiguous
    frame shards, every step = reset + shard + one all-reduce (here 301 frames instead of 10 000: 151 + 150)."""
    out = launch(["--frames", "301"], 29534)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert "1M-bead" in out["config"]["workload"] and out["config"]["frames_per_step"] == 301
    assert out["config"]["frames_per_step_per_gpu"] in (150, 151)
    assert out["sanity"]["frames_accumulated"] == out["sanity"]["expected"] == 301    # the job, not the warm-up
    assert out["sanity"]["counts_ok"] and out["value"] > 0
    assert len(out["roofline"]["frac_per_rank"]) == 2 and out["allreduce_ms"] > 0
    assert out["roofline"]["kernel"] == "k_bonds_tiled"
    # the same sharding from a FILE: every rank ran gorder_hip_run_trajectory on its share, one all-reduce, and the
    # result equals one handle over the whole trajectory
    e2e = out["end_to_end"]
    assert e2e["equal_to_one_handle"] and e2e["frames_of_all_shards"] == e2e["frames"] == 500 * 80
    assert [r["frames"] for r in e2e["per_rank"]] == [20000, 20000] and e2e["per_rank"][1]["first_frame_of_shard"] == 20000
    # the library's own RCCL route (gorder_hip_comm_unique_id / _create / gorder_hip_allreduce) is exercised after the
    # timed region on every multi-rank run.  Two ranks on ONE device is something RCCL refuses: the block must be there,
    # say so, and cost neither the line nor the other blocks (on a real multi-GPU node: ok and equal_to_torch true).
    lib = out["lib_allreduce"]
    assert set(lib) >= {"ok", "equal_to_torch", "ms", "ranks"} and lib["ranks"] == 2
    assert (lib["ok"] and lib["equal_to_torch"] is True and lib["ms"] > 0) or (not lib["ok"] and lib.get("error"))
    assert "extras_timed_out" not in out
    # every kernel group of the step is in the line, and the groups add up to the step
    r = out["roofline"]
    assert abs(sum(k["share_of_step"] for k in r["kernels"]) - 1.0) < 1e-6
    assert abs(sum(k["ms"] for k in r["kernels"]) - r["whole_step_ms"]) < 1e-6 * max(1.0, r["whole_step_ms"])
