#!/usr/bin/env python3
"""Kernel-experiment driver (one parametrised script in place of the 45 one-off tools/exp_* files of rounds 1-5; their findings are in DESIGN_HISTORY.md).

In the BUILD container (hipcc cross-compiles gfx950):
  tools/exp.py build "" "-DFAST_XCD=1" "-DOCT_U=8 -DDESC_WAVES=6"    one liborbhip variant per -D set under exp_so/ (ships to the GPU box with gpurun)
ON THE GPU BOX (through gpurun; ORBHIP_LIB=<variant.so> selects a variant for any sub-command):
  tools/exp.py variants [bench args]          headline / per-kernel times of every exp_so/*.so
  tools/exp.py lm [--windows 256 --kf 100 --fixed 20 --points 20000 --kind mono --threads 1 --reps 2]      LM iterations/s (bench.py's LM leg alone)
  tools/exp.py single                         one 752x480 frame through orbx_extract, us per call
  tools/exp.py host_fed                       where the host-fed step's time goes (full / no D2H / no kernels / copies alone)
  tools/exp.py mixed                          batch sizes alternating on one handle: the FAST pass policy must not change results
  tools/exp.py phases                         -DORBX_PROF builds: per-phase s_memtime shares of k_fast / k_describe
  tools/pmc_passes.sh TAG -- <command>        kernel stats + the separate --pmc passes of any of the above; tools/pmc_summary.py TAG prints per-kernel means"""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p_)
HIPCC = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-I" + os.path.join(ROOT, "include")]


def cmd_build(argv):
    out = os.path.join(ROOT, "exp_so")
    os.makedirs(out, exist_ok=True)
    for f in glob.glob(os.path.join(out, "*.so")):
        os.remove(f)
    src = sorted(glob.glob(os.path.join(ROOT, "awesome-orb-slam3-3dvisioncraft-version_amd", "csrc", "*.hip")))
    for i, v in enumerate(argv or [""]):
        name = "v%02d_%s.so" % (i, "".join(c if c.isalnum() or c in "_-" else "_" for c in "base" + v.replace(" ", "")))
        rc = subprocess.call(HIPCC + src + ["-o", os.path.join(out, name)] + v.split(), stderr=subprocess.DEVNULL)
        print(("built exp_so/%s  [%s]" if rc == 0 else "BUILD FAILED %s [%s]") % (name, v))


def cmd_variants(argv):
    for so in sorted(glob.glob(os.path.join(ROOT, "exp_so", "*.so"))):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--headline-only", "--no-pmc"] + argv, env=dict(os.environ, ORBHIP_LIB=so),
                           capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print(os.path.basename(so), d["value"], d["ms_per_step"], d.get("kernel_ms"))
        except Exception:   # noqa: BLE001
            print(os.path.basename(so), "FAILED", r.stderr[-300:])


def cmd_lm(argv):
    import numpy as np
    import torch
    from orbhip.lba import LbaWindows, synth_window
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=256); ap.add_argument("--reps", type=int, default=2); ap.add_argument("--kf", type=int, default=100)
    ap.add_argument("--fixed", type=int, default=20); ap.add_argument("--points", type=int, default=20000); ap.add_argument("--kind", default="mono")
    ap.add_argument("--threads", type=int, default=1, help="host threads x (windows / threads) windows, one HIP stream each")
    ap.add_argument("--sorted", action="store_true", help="observations of a landmark ordered by pose index (std::map<KeyFrame*> order)")
    a = ap.parse_args(argv)
    dev = torch.device("cuda:0")
    wins, cams = [], None
    for i in range(2):
        w, cams = synth_window(100 + i, a.kf, a.fixed, a.points, 8, a.kind)
        if a.sorted:
            w["edges"] = w["edges"][np.lexsort((w["edges"]["pose"], w["edges"]["point"]))]
        wins.append(w)
    T, per = a.threads, a.windows // a.threads
    to_d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    Ls = [LbaWindows([wins[i % 2] for i in range(per)], cams, to_d) for _ in range(T)]
    p0 = [(L.d["poses"].clone(), L.d["points"].clone()) for L in Ls]
    streams = [torch.cuda.Stream(dev) for _ in range(T)]
    res = [None] * T

    def run(k, reps):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(streams[k]):
            for _ in range(reps):
                Ls[k].d["poses"].copy_(p0[k][0]); Ls[k].d["points"].copy_(p0[k][1])
                res[k] = Ls[k].optimize(5)
            streams[k].synchronize()
    for reps in (1, a.reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        th = [threading.Thread(target=run, args=(k, reps)) for k in range(T)]
        [x.start() for x in th]
        [x.join() for x in th]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    its = sum(float(r[:, 0].sum()) for r in res) * a.reps
    print("lm_iterations_per_s %.1f  ms_per_optimize5 %.3f  trials/window %.2f  chi2[0] %.6f  (%d threads x %d windows, %d KF / %d fixed / %d landmarks, %s)" %
          (its / dt, dt / a.reps * 1e3, float(res[0][:, 3].mean()), float(res[0][0, 1]), T, per, a.kf, a.fixed, a.points, a.kind))


def cmd_single(argv):
    import orbhip
    from orbhip.synth import synth_image
    img = synth_image(5, 752, 480)
    e = orbhip.ORBextractor(1000, 1.2, 8, 20, 7, device=0)
    for _ in range(20):
        e(img, None, (0, 1000))
    t = time.perf_counter()
    for _ in range(200):
        mono, k, d = e(img, None, (0, 1000))
    print("single-frame orbx_extract: %.1f us per call (%d key points)" % ((time.perf_counter() - t) / 200 * 1e6, len(k)))


def cmd_host_fed(argv):
    import torch
    import bench
    B = int(os.environ.get("B", "512"))
    hosts = [bench.make_batch(B, seed0=100000 * i, unique=max(1, B // 2), workers=16) for i in range(3)]
    P = bench.StepPipeline([torch.from_numpy(f_).cuda() for f_ in hosts], 752, 480, 1000, 0, streams=3, frames_host=hosts)
    P.start_streams()
    P.start_host_fed()

    def run(label, steps=40, **knobs):
        for k_ in ("skip_kernels", "skip_d2h"):
            P._hf[k_] = bool(knobs.get(k_))
        for _ in range(4):
            P.host_fed_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            P.host_fed_step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print("%-28s %.3f ms per step  %.0f frames/s" % (label, dt * 1e3, B / dt), flush=True)
    run("full", steps=10); run("full", steps=10); run("full", steps=20); run("full"); run("no d2h", skip_d2h=True); run("no kernels", skip_kernels=True); run("no kernels, no d2h", skip_kernels=True, skip_d2h=True); run("full again")
    print("h2d alone %.3f ms, d2h alone %.3f ms" % (P.host_fed_copy_only(20, "h2d") / 20 * 1e3, P.host_fed_copy_only(20, "d2h") / 20 * 1e3))
    for _ in range(20):
        P.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        P.step()
    torch.cuda.synchronize()
    print("resident step %.3f ms" % ((time.perf_counter() - t0) / 40 * 1e3))


def cmd_mixed(argv):
    import numpy as np
    import torch
    import bench
    import orbhip
    d = torch.from_numpy(bench.make_batch(512)).cuda()
    ex = orbhip.ORBextractor(1000, 1.2, 8, 20, 7, device=0, max_batch=512)
    ref, bad = None, 0
    for it, B in enumerate([512, 64, 512, 128, 64, 96, 512, 65, 512]):
        out = ex.extract_batch(d[:B].contiguous(), (0, 1000))
        torch.cuda.synchronize()
        k, de, c = [t.cpu().numpy().copy() for t in out]
        if ref is None:
            ref = (k, de, c)
        else:
            for b in range(B):
                n = c[b, 0]
                bad += int(n != ref[2][b, 0] or not np.array_equal(de[b, :n], ref[1][b, :n]) or not np.array_equal(k[b, :n].view(np.int32), ref[0][b, :n].view(np.int32)))
        print(it, B, ex.last_fast_passes(), "mismatching frames so far:", bad)
    print("OK" if bad == 0 else "FAIL")


def cmd_kinds(argv):
    """Per-kernel times of a batch made of ONE scene kind each (what a frame of that kind costs), and of the mixed batch."""
    import numpy as np
    import torch
    import bench
    import orbhip
    B = int(argv[0]) if argv else 256
    sets = {}
    for kind in ("textured", "sparse", "low_contrast", "flat", "noise"):
        base = bench.synth_many([(kind, 900 + i, 752, 480) for i in range(16)], 16)
        sets[kind] = bench._pair_up(base, B, np.random.default_rng(5))
    sets["mixed"] = bench.make_mixed_batch(B, seed0=7000, workers=16)[0]
    ex = orbhip.ORBextractor(1000, 1.2, 8, 20, 7, device=0, max_batch=B)
    ex.enable_timing(True)
    for kind, fr in sets.items():
        d = torch.from_numpy(fr).cuda()
        out = None
        for _ in range(3):
            out = ex.extract_batch(d, (0, 1000), out=out)
        torch.cuda.synchronize()
        t = ex.last_timing()
        print("%-13s %s  passes %s  mean kp %.0f" % (kind, {k: round(v * 1e3 / B, 3) for k, v in t.items()}, ex.last_fast_passes(), float(out[2][:, 0].float().mean())), flush=True)
    print("(us per frame)")


def cmd_phases(argv):
    import ctypes as C
    import numpy as np
    import torch
    import bench
    import orbhip
    from orbhip import _lib
    L = _lib.load()
    d = torch.from_numpy(bench.make_batch(512)).cuda()
    ex = orbhip.ORBextractor(1000, 1.2, 8, 20, 7, device=0, max_batch=512)
    ex.enable_timing(True)
    out = ex.extract_batch(d, (0, 1000))
    buf = (C.c_ulonglong * 32)()
    L.orbx_debug_prof(buf, 1)
    ex.extract_batch(d, (0, 1000), out=out)
    L.orbx_debug_prof(buf, 1)
    v = np.array(list(buf), np.float64)[:16].reshape(2, 8)
    names = [["prologue+stage issue", "staging wait (barrier)", "stage1+compaction+stage2", "stage3 score", "barrier", "NMS+retry+list", "emit"],
             ["record+counts", "patch loads->LDS", "barrier1", "row reads+IC_Angle+row pass", "barrier2", "trig+column pass", "barrier3", "rBRIEF+outputs"]]
    for k, kn in enumerate(("k_fast", "k_describe")):
        print(kn, "sum of wave time (ticks): %.3e" % v[k].sum(), ex.last_timing())
        for i, n in enumerate(names[k]):
            print("   %-28s %5.1f %%" % (n, 100 * v[k][i] / v[k].sum()))


if __name__ == "__main__":
    cmds = {n[4:]: f for n, f in globals().items() if n.startswith("cmd_")}
    if len(sys.argv) < 2 or sys.argv[1] not in cmds:
        sys.exit(__doc__)
    cmds[sys.argv[1]](sys.argv[2:])
