"""The C++ drop-in shim (compat/Compute.hpp) compiles with plain g++ against include/mapn.h and
links to libmapn.so; on a GPU box the reference caller's sequence runs through it."""
import os
import subprocess

import pytest

import mapn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "particles_draw.cpp")
PKG = os.path.dirname(mapn.library_path())


def _build(tmp_path):
    exe = str(tmp_path / "particles_draw")
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(PKG, "compat"), SRC, "-o", exe, "-L", PKG, "-lmapn", f"-Wl,-rpath,{PKG}",
           "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_shim_compiles_and_refuses_without_device(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "--no-device"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_reference_caller_sequence_through_the_shim(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "FAIL" not in r.stdout
