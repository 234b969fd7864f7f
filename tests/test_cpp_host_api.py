"""Builds and runs tests/cpp/host_api_test.cpp against include/gorp.hpp + libgorp_hip.so: the C++ host-side
mirror of the reference API.  Host-only mode on CPU; the -m gpu variant runs the extract calls on the device."""
import os
import subprocess

import pytest

from gorp_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path):
    exe = str(tmp_path / "host_api_test")
    rt = N._load_hip_runtime()._name  # the HIP runtime the Python side would use
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "host_api_test.cpp"),
           "-o", exe, N.LIB_PATH, rt, "-Wl,-rpath," + os.path.dirname(N.LIB_PATH), "-Wl,-rpath," + os.path.dirname(rt),
           "-Wl,--allow-shlib-undefined"]
    subprocess.check_call(cmd)
    return exe


def test_cpp_host_api_host_only(tmp_path):
    exe = build(tmp_path)
    if N.lib().gx_device_count() > 0:
        pytest.skip("a GPU is present: covered by the gpu variant")
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host-only checks ok" in out.stdout


@pytest.mark.gpu
def test_cpp_host_api_on_gpu(tmp_path):
    exe = build(tmp_path)
    out = subprocess.run([exe, "--gpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GPU checks ok" in out.stdout
