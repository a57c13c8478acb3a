"""The line `python bench.py` prints at N = 1 carries what the driver and the reviewer read: the contract's keys, the
roofline and cpu_baseline objects, the end-to-end blocks with both decode routes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_gpu_line(built):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--frames", "2000",
                          "--no-scaling-reference"], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, GORDER_BENCH_LARGE_GB="0.5"))      # (the default run writes 8 GB)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "end_to_end"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["unit"] == "frames/s" and d["higher_is_better"] is True
    assert d["config"]["workload"].startswith("AAOrder") and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0.3 < r["frac"] < 1.0 and d["value"] > 1e6
    assert r["kernel"] == "k_bonds_tiled"                    # what gorder_hip_kernel_time_names reports, not a literal
    assert r["traffic"] is not None and 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.5 and "scaled" in r["traffic_source"]
    # the roofline object describes the step it times: every kernel group, adding up to the whole step
    names = [k["name"] for k in r["kernels"]]
    assert names == ["k_bonds_tiled", "k_batch_end"] and r["timed"] == "k_bonds_tiled + k_batch_end"
    assert abs(sum(k["share_of_step"] for k in r["kernels"]) - 1.0) < 1e-6
    assert abs(sum(k["ms"] for k in r["kernels"]) - r["whole_step_ms"]) < 1e-6
    assert r["whole_step_frac"] <= r["frac"] and r["whole_step_frac"] > 0.9 * r["frac"]
    assert r["whole_step_ms"] <= d["ms_per_step"] * 1.02
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and 1 <= c["cores"] <= c["cores_available"] and c["value"] > 0 and "sample" in c
    assert c["parallel_efficiency"] > 0.5 and len(c["scaling"]) >= 2 and (c["cpu_quota"] is None or c["cores"] <= c["cpu_quota"] + 0.5)
    e = d["end_to_end"]
    assert e["decoded_on"] == "device" and e["host_decode"]["decoded_on"] == "host"
    assert e["value"] > e["host_decode"]["value"] > 0
    # the figure quoted is the one from a trajectory of distinct frames (here 0.5 GB of them; 8 GB by default), with the
    # cold (disk) and hot-file figures beside it
    big = e["large"]
    assert "skipped" in big or (big["warm"]["value"] > 0 and big["cold"]["value"] > 0 and big["parts"] >= 1
                                and e["value"] == big["warm"]["value"] and e["value_is"].startswith("large.warm")
                                and e["hot_file"]["value"] > 0 and 0 < big["warm"]["reader_busy_fraction"] <= 1.0)
    s = e["solvated"]
    assert s["device_decode"]["decoded_on"] == "device" and s["device_decode"]["frames_decoded_by_host_after_all"] == 0
    assert s["device_decode"]["pcie_GBps"] > 0 and s["atoms_analysed"] * 4 == s["atoms_in_file"]


@pytest.mark.parametrize("workload,dominant,others", [
    ("cg3k-local", "k_local_sums", {"k_local_build", "k_local_rowprefix", "k_local_decide", "k_local_flags_rows", "k_local_flags_todo", "k_bonds_tiled", "k_batch_end"}),
    ("aa256-leaflets", None, {"k_leaflets_global_contig", "k_bonds_tiled", "k_batch_end"}),
    ("ua256-maps", "k_ua_extras", {"k_map_accumulate", "k_batch_end"})])
def test_the_roofline_names_the_longest_kernel_group(built, workload, dominant, others):
    """For workloads whose step is more than the order kernel (leaflet kernels, map accumulation) `roofline.kernel` is the
    longest kernel group of the step and `roofline.kernels` lists them all (round 3 reported k_bonds_tiled for
    cg3k-local: 4.7 % of its step)."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--frames", "512",
                          "--workload", workload, "--no-scaling-reference", "--no-cpu-baseline", "--no-end-to-end"],
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    r = d["roofline"]
    names = {k["name"] for k in r["kernels"]}
    assert others <= names, names
    longest = max(r["kernels"], key=lambda k: k["ms"])["name"]
    assert r["kernel"] == longest and (dominant is None or longest == dominant)
    assert abs(sum(k["share_of_step"] for k in r["kernels"]) - 1.0) < 1e-6
    assert r["whole_step_frac"] < r["frac"]
