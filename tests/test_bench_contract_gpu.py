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
                          "--no-scaling-reference"], capture_output=True, text=True, timeout=900)
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
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and 1 <= c["cores"] <= c["cores_available"] and c["value"] > 0 and "sample" in c
    e = d["end_to_end"]
    assert e["decoded_on"] == "device" and e["host_decode"]["decoded_on"] == "host"
    assert e["value"] > e["host_decode"]["value"] > 0
    s = e["solvated"]
    assert s["device_decode"]["decoded_on"] == "device" and s["device_decode"]["frames_decoded_by_host_after_all"] == 0
    assert s["device_decode"]["pcie_GBps"] > 0 and s["atoms_analysed"] * 4 == s["atoms_in_file"]
