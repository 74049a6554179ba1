"""Builds tests/emu/build/liborbhip_emu.so: the UNMODIFIED product sources (csrc/*.hip) compiled with g++
against the fiber-based HIP emulator in tests/emu/hip/hip_runtime.h.  TEST INFRASTRUCTURE ONLY: it lets the
CPU-only test tier check kernel *logic* against the oracle; the product never loads this library.

Variant libraries (-D switches that force a kernel down one of its rarer paths) share objects with the plain build: a source file is
compiled again only if one of the variant's macro names occurs in it (or in a header, in which case every file is)."""
import glob
import hashlib
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EMU = os.path.join(ROOT, "tests", "emu")
OUT = os.path.join(EMU, "build", "liborbhip_emu.so")
CSRC = os.path.join(ROOT, "awesome-orb-slam3-3dvisioncraft-version_amd", "csrc")
SRCS = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
CXX = ["g++", "-std=c++17", "-O2", "-g", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-w", "-I", EMU, "-I", os.path.join(ROOT, "include")]


def build(force=False, defines=(), tag="", flags=()):
    """flags: extra compiler / linker switches for every object of the variant (e.g. -fsanitize=thread for the TSAN build of the threading test)"""
    out = OUT if not tag else OUT.replace(".so", "_" + tag + ".so")
    return _build(out, force, tuple(defines), tuple(flags))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.inc")) + glob.glob(os.path.join(CSRC, "*.h")) + \
        [os.path.join(EMU, "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "orbhip.h")]


def _mentions(path, names):
    text = open(path, errors="replace").read()
    return any(re.search(r"\b%s\b" % re.escape(n), text) for n in names)


def _object(src, defines, headers, force, flags=()):
    """the object of one source file under `defines` (only those that can reach it), rebuilt when the file or a header is newer"""
    names = [d.split("=")[0] for d in defines]
    everywhere = [n for n in names if any(_mentions(h, [n]) for h in headers)]
    mine = tuple(d for d in defines if d.split("=")[0] in everywhere or _mentions(src, [d.split("=")[0]]))
    key = hashlib.sha1(" ".join(mine + tuple(flags)).encode()).hexdigest()[:10] if (mine or flags) else "plain"
    obj = os.path.join(EMU, "build", "obj", "%s__%s.o" % (os.path.basename(src), key))
    deps = [src] + headers
    if not force and os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in deps):
        return obj
    os.makedirs(os.path.dirname(obj), exist_ok=True)
    tmp = obj + ".tmp%d" % os.getpid()
    subprocess.check_call(CXX + list(flags) + ["-D" + d for d in mine] + ["-x", "c++", "-c", src, "-o", tmp])
    os.replace(tmp, obj)
    return obj


def _build(out, force, defines, flags=()):
    headers = _headers()
    deps = SRCS + headers

    def fresh():
        return os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps)
    if not force and fresh():
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    # pytest-xdist workers ask for the same library (and the same objects) at the same time: one builds, the others wait
    import fcntl
    with open(os.path.join(EMU, "build", "build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if not force and fresh():
            return out
        objs = [_object(s, defines, headers, force, flags) for s in SRCS]
        tmp = out + ".tmp%d" % os.getpid()
        subprocess.check_call(["g++", "-shared"] + list(flags) + ["-o", tmp] + objs)
        os.replace(tmp, out)
    return out


if __name__ == "__main__":
    print(build(force=True))
