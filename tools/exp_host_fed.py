#!/usr/bin/env python3
"""Runs ON THE GPU BOX: where the host-fed step's time goes — the full step, the step without its D2H, without its kernels, and the copies alone."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402

B = int(os.environ.get("B", "512"))
hosts = [bench.make_batch(B, seed0=100000 * i, unique=max(1, B // 2), workers=16) for i in range(3)]
d_sets = [torch.from_numpy(f_).cuda() for f_ in hosts]
P = bench.StepPipeline(d_sets, 752, 480, 1000, 0, streams=3, frames_host=hosts)
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


run("full")
run("no d2h", skip_d2h=True)
run("no kernels", skip_kernels=True)
run("no kernels, no d2h", skip_kernels=True, skip_d2h=True)
run("full again")
print("h2d alone %.3f ms, d2h alone %.3f ms" % (P.host_fed_copy_only(20, "h2d") / 20 * 1e3, P.host_fed_copy_only(20, "d2h") / 20 * 1e3))
for t_ in range(20):
    P.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for t_ in range(40):
    P.step()
torch.cuda.synchronize()
print("resident step %.3f ms" % ((time.perf_counter() - t0) / 40 * 1e3))
