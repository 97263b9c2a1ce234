"""The C++ host mirror (include/p3hip.hpp) over the C ABI: compiled with g++ and run as a child process.
CPU box: it must build, link and report "HIP unavailable" (no fallback).  GPU box: the whole demo passes."""
import os
import subprocess

import pytest

from conftest import ROOT

BIN = os.path.join(ROOT, "tools", "_bin", "host_demo")


def _build(p3):
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    libdir = os.path.dirname(p3._lib.LIB_PATH)
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tools", "host_demo.cpp"), "-L" + libdir, "-lp3hip", "-Wl,-rpath," + libdir,
           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64", "-o", BIN]
    subprocess.check_call(cmd)


def test_cpp_host_builds_and_refuses_without_gpu(p3):
    _build(p3)
    ok, _ = p3.is_available()
    if ok:
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "HIP unavailable" in r.stdout


@pytest.mark.gpu
def test_cpp_host_demo_on_gpu(p3):
    _build(p3)
    r = subprocess.run([BIN, "12"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("OK") and "fib_air ok" in r.stdout and "expected error" in r.stdout
