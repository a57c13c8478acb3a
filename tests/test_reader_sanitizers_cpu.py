"""xtc_reader.cpp compiled with sanitizers (CPU build only — the GPU pool offers none) and driven through every entry
point of include/gorder_xtc.h by tests/cabi/reader_sanitize.cpp."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("sanitizer", ["address,undefined", "thread"])
def test_reader_is_clean_under_sanitizers(tmp_path, sanitizer):
    exe = str(tmp_path / "reader_sanitize")
    out = str(tmp_path / "out.xtc")
    cmd = ["g++", "-std=c++17", "-g", "-O1", f"-fsanitize={sanitizer}", "-fno-omit-frame-pointer", "-pthread",
           f"-I{os.path.join(ROOT, 'include')}", f'-DGOLDEN="{os.path.join(ROOT, "tests", "golden")}"', f'-DOUTFILE="{out}"',
           os.path.join(ROOT, "tests", "cabi", "reader_sanitize.cpp"), os.path.join(ROOT, "gorder_amd", "csrc", "xtc_reader.cpp"),
           "-o", exe]
    subprocess.check_call(cmd)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.count(" ok (") == 3 and "Sanitizer" not in res.stderr, res.stderr
