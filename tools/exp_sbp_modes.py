#!/usr/bin/env python3
"""Runs ON THE GPU BOX: device time of orbm_search_by_projection for the six full-size cases of tests/test_fullsize_parity.py (motion model th 15 / 7 / 30,
local map th 1 / 3 / 15-wide) at 512 frames per call — to compare k_sbp_frame with the rounds-1-3 pair (ORBHIP_LIB=<a -DSBP_FUSED_FRAME=0 build>)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orbhip  # noqa: E402
from orbhip.matcher import QUERY_DTYPE, TH_HIGH  # noqa: E402
from test_fullsize_parity import FULL, SBP_FULL  # noqa: E402
from test_matcher_parity import make_queries, scene  # noqa: E402

B = 512
scenes = [scene(**kw) for kw in FULL]
dev = torch.device("cuda", 0)
for name, mode, th, nnratio, ori, stereo, occupied in SBP_FULL:
    rng = np.random.default_rng(7)
    qs = [make_queries(S, mode, th, rng, stereo) for S in scenes]
    cap_k = max(len(S["kb"]) for S in scenes) + 7
    cap_q = max(len(q) for q in qs) + 5
    kps = np.zeros((B, cap_k, 7), np.float32); desc = np.zeros((B, cap_k, 32), np.uint8)
    Q = np.zeros((B, cap_q), QUERY_DTYPE); qd = np.zeros((B, cap_q, 32), np.uint8)
    nk = np.zeros(B, np.int32); nq = np.zeros(B, np.int32)
    ur = np.zeros((B, cap_k), np.float32) if stereo else None
    oc = np.zeros((B, cap_k), np.uint8) if occupied else None
    for b in range(B):
        S, q = scenes[b % 4], qs[b % 4]
        n = len(S["kb"])
        kps[b, :n] = S["kb"].view(np.float32).reshape(-1, 7); desc[b, :n] = S["db"]; nk[b] = n
        Q[b, :len(q)] = q; qd[b, :len(q)] = S["da"]; nq[b] = len(q)
        if ur is not None:
            ur[b, :n] = S["kb"]["x"] - 20.0
        if oc is not None:
            oc[b, :n] = (np.arange(n) % 7 == 0)
    d = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    m = orbhip.ORBmatcher(nnratio, ori)
    dk, dn = d(kps), d(nk)
    grid = scenes[0]["grid"]
    gs, gi = m.grid_build(dk, dn, grid)
    args = (dk, d(desc), dn, gs, gi, d(Q.view(np.uint8).reshape(B, cap_q, 28)), d(qd), d(nq), grid, mode, TH_HIGH)
    kw = dict(u_right=d(ur), occupied0=d(oc))
    work = torch.empty(m._L.orbm_search_workspace_bytes(B, cap_q), dtype=torch.uint8, device=dev)
    out = m.SearchByProjection(*args, work=work, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = m.SearchByProjection(*args, work=work, out=out, **kw)
    e1.record(); torch.cuda.synchronize()
    print("%-28s %.3f ms per 512 frames, mean matches %.1f" % (name, e0.elapsed_time(e1) / 10, float(out[2].float().mean())))
