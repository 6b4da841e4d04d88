"""The C++ drop-in shim (compat/Compute.hpp) compiles with plain g++ against include/mapn.h and
links to libmapn.so; on a GPU box the reference caller's sequence runs through it."""
import os
import subprocess

import pytest

import mapn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "particles_draw.cpp")
PKG = os.path.dirname(mapn.library_path())


def _build(tmp_path, src=SRC, hip=False):
    exe = str(tmp_path / os.path.splitext(os.path.basename(src))[0])
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(PKG, "compat"), src, "-o", exe, "-L", PKG, "-lmapn", f"-Wl,-rpath,{PKG}",
           "-Wl,-rpath,/opt/rocm/lib"]
    if hip:      # host-side HIP API only (streams, events, copies): plain g++ against the runtime
        cmd += ["-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-L/opt/rocm/lib", "-lamdhip64", "-Wno-unused-result", "-Wno-deprecated-declarations"]
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


def test_consumer_side_compiles(tmp_path):
    _build(tmp_path, os.path.join(ROOT, "tests", "cpp", "consumer_fence.cpp"), hip=True)


@pytest.mark.gpu
def test_render_side_protocol_with_one_exported_event_and_queued_waits(tmp_path):
    """tests/cpp/consumer_fence.cpp: the exported step-done event is fetched once and used for 150
    frames (it used to go stale after one step / wrap at 64), Simulate queues its wait
    before the consumer's signal exists (Compute.cpp:1012), and a deliberately slow
    consumer's copies are bit-identical to a lockstep reference."""
    exe = _build(tmp_path, os.path.join(ROOT, "tests", "cpp", "consumer_fence.cpp"), hip=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "FAIL" not in r.stdout
