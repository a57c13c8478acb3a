"""The C ABI from a plain-C program (tests/cabi/kat.c): no Python, no torch between the client and the library —
the position a Rust host is in.  Here (no GPU) it must refuse loudly; on the GPU box it must give the
reference's known answers."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_client(tmp_path):
    exe = str(tmp_path / "kat")
    lib_dir = os.path.join(ROOT, "gorder_amd")
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", exe,
           os.path.join(ROOT, "tests", "cabi", "kat.c"), "-L", lib_dir, "-lgorder_hip", f"-Wl,-rpath,{lib_dir}", "-lm"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_header_is_plain_c_and_the_library_refuses_without_a_gpu(built, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("this is the no-GPU half")
    exe = build_client(tmp_path)                       # include/gorder_hip.h compiles as C11 with -Wall -Werror
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 77 and "no CPU fallback" in r.stdout


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_plain_c_client_reproduces_the_reference_unit_test(built, tmp_path):
    exe = build_client(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout
