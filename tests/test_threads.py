"""Threading contract of the boundary (SURVEY.md 8(b) "Threading"; reference Frame.cc:111-114,1209-1212, LocalMapping.cc:201): tests/cpp/threads_test.cpp
— two extractor threads, a matcher thread, a LocalBundleAdjustment thread and a create / destroy / failing-create thread, every output of every turn
equal to the single-threaded result.  CPU tier: the emulated library, plain and under ThreadSanitizer (host-side races of the library and the
adapters: globals, one-time initialisation, error slots); GPU tier: the real liborbhip.so, 200 turns."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(libpath, tag, tmp_path, flags=()):
    exe = str(tmp_path / ("threads_test_" + tag))
    libdir, libname = os.path.dirname(libpath), os.path.basename(libpath)[3:-3]
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Wno-sign-compare"] + list(flags) + ["-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "threads_test.cpp"), "-L", libdir, "-l" + libname, "-Wl,-rpath," + libdir,
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lpthread", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def _run(exe, args, timeout, env=None):
    out = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout, env=env)
    assert out.returncode == 0 and "threads_test OK" in out.stdout, out.stdout[-3000:] + out.stderr[-6000:]
    return out


def test_threads_on_emulated_library(emu_lib, tmp_path):
    import build_emu
    _run(_build(build_emu.OUT, "emu", tmp_path), (256, 232, 150, 2), 1500)


def test_threads_on_emulated_library_under_tsan(tmp_path):
    """The whole library (product sources + the fiber emulator, its fibers announced through TSAN's fiber API) and the test compiled with
    -fsanitize=thread: any data race between the five host threads — in the library's host code, its one-time initialisations, the adapters'
    error slots — fails the run (halt_on_error)."""
    import build_emu
    lib = build_emu.build(tag="tsan", flags=("-fsanitize=thread", "-O1"))
    exe = _build(lib, "tsan", tmp_path, flags=("-fsanitize=thread",))
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1 history_size=4")
    out = _run(exe, (240, 232, 100, 1, 1, 0), 3000, env=env)      # one LM iteration, no PoseOptimizer: the sanitised emulator pays seconds per launch
    assert "WARNING: ThreadSanitizer" not in out.stderr, out.stderr[-6000:]


@pytest.mark.gpu
def test_threads_on_hip_library(hip_lib, tmp_path):
    from orbhip import _lib
    _run(_build(_lib.LIB_PATH, "hip", tmp_path), (640, 480, 1000, 200), 900)
