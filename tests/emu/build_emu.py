"""Builds tests/emu/build/liborbhip_emu.so: the UNMODIFIED product sources (csrc/*.hip) compiled with g++
against the fiber-based HIP emulator in tests/emu/hip/hip_runtime.h.  TEST INFRASTRUCTURE ONLY: it lets the
CPU-only test tier check kernel *logic* against the oracle; the product never loads this library."""
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EMU = os.path.join(ROOT, "tests", "emu")
OUT = os.path.join(EMU, "build", "liborbhip_emu.so")
SRCS = sorted(glob.glob(os.path.join(ROOT, "awesome-orb-slam3-3dvisioncraft-version_amd", "csrc", "*.hip")))


def build(force=False, defines=(), tag=""):
    global OUT
    out = OUT if not tag else OUT.replace(".so", "_" + tag + ".so")
    return _build(out, force, defines)


def _build(OUT, force, defines):
    deps = SRCS + glob.glob(os.path.join(ROOT, "awesome-orb-slam3-3dvisioncraft-version_amd", "csrc", "*.inc")) + \
        glob.glob(os.path.join(ROOT, "awesome-orb-slam3-3dvisioncraft-version_amd", "csrc", "*.h")) + \
        [os.path.join(EMU, "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "orbhip.h")]
    def fresh():
        return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps)
    if not force and fresh():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    # pytest-xdist workers ask for the same library at the same time: one builds (into a temporary name, renamed when complete), the others wait
    import fcntl
    with open(OUT + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if not force and fresh():
            return OUT
        tmp = OUT + ".tmp%d" % os.getpid()
        cmd = ["g++", "-std=c++17", "-O2", "-g", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-w",
               "-I", EMU, "-I", os.path.join(ROOT, "include")] + ["-D" + d for d in defines]
        for s in SRCS:
            cmd += ["-x", "c++", s]
        cmd += ["-o", tmp]
        subprocess.check_call(cmd)
        os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
