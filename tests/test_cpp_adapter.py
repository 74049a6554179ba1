"""Builds tests/cpp/adapter_test.cpp against the header-only adapters in include/orbslam3_hip/ and runs it:
CPU tier = emulated library, GPU tier = the real liborbhip.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_and_run(libpath, tag, tmp_path):
    exe = str(tmp_path / ("adapter_test_" + tag))
    libdir, libname = os.path.dirname(libpath), os.path.basename(libpath)[3:-3]
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-C", odir], stdout=subprocess.DEVNULL)
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "adapter_test.cpp"),
           "-L", libdir, "-l" + libname, "-L", odir, "-loracle", "-Wl,-rpath," + libdir, "-Wl,-rpath," + odir,
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "adapter_test OK" in out.stdout, out.stdout + out.stderr


def test_cpp_adapters_on_emulated_library(emu_lib, tmp_path):
    import build_emu
    _build_and_run(build_emu.OUT, "emu", tmp_path)


@pytest.mark.gpu
def test_cpp_adapters_on_hip_library(hip_lib, tmp_path):
    from orbhip import _lib
    _build_and_run(_lib.LIB_PATH, "hip", tmp_path)
