#!/usr/bin/env python3
"""Randomised parity sweep of the HIP extractor / matcher against the oracle (run on the GPU box:  python tools/fuzz_parity.py [n] [seed0]).
Random image sizes (odd widths included), content mixes, feature counts, pyramid shapes and thresholds; every case must be bit-exact."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import orbhip  # noqa: E402
from orbhip.synth import synth_image  # noqa: E402


def case(rng):
    W = int(rng.integers(200, 1400)); H = int(rng.integers(160, 800))
    if rng.random() < 0.4: W &= ~3   # dense batches (the two-cell-row tile path of k_fast) need 4-byte row strides
    kind = rng.integers(0, 4)
    if kind == 0:
        img = synth_image(int(rng.integers(1 << 30)), W, H, n_rect=int(rng.integers(5, 500)), n_disc=int(rng.integers(0, 300)),
                          noise=float(rng.uniform(0, 6)), contrast=float(rng.uniform(0.2, 1.5)))
    elif kind == 1:
        img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    elif kind == 2:
        img = np.clip(rng.normal(128, rng.uniform(1, 40), (H, W)), 0, 255).astype(np.uint8)
    else:
        base = synth_image(int(rng.integers(1 << 30)), W, H)
        img = np.where(rng.random((H, W)) < 0.02, rng.integers(0, 256, (H, W)), base).astype(np.uint8)
    nf = int(rng.choice([300, 500, 1000, 1500, 2000, 4000]))
    sf = float(rng.choice([1.1, 1.2, 1.2, 1.2, 1.25, 1.4]))
    nl = int(rng.choice([3, 5, 8, 8, 8, 10]))
    ini = int(rng.choice([10, 20, 20, 30])); mn = int(rng.choice([3, 7, 7, 10]))
    if rng.random() < 0.15:   # threshold extremes and parities: the byte-parallel pre-test of k_fast derives its compare constants from min(ini, mn)
        mn = int(rng.choice([0, 1, 2, 4, 5, 6, 8, 9, 11, 16, 33, 64, 127, 128, 200, 253, 254, 255]))
        ini = int(max(mn, rng.choice([0, 1, 12, 20, 40, 100, 254, 255])))
    lap = (int(rng.integers(0, W // 2)), int(rng.integers(W // 2, W + 50))) if rng.random() < 0.5 else (0, 0)
    return img, nf, sf, nl, ini, min(mn, ini), lap


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for i in range(n):
        rng = np.random.default_rng(seed0 + i)
        img, nf, sf, nl, ini, mn, lap = case(rng)
        # keep the smallest level above the extractor's minimum size
        while nl > 1 and min(img.shape) / sf ** (nl - 1) < 70:
            nl -= 1
        k = []
        try:
            try:
                o = O.OrbOracle(nf, sf, nl, ini, mn)
                mono, k, d = o.extract(img, *lap)
                rejected = False
            except ValueError:
                rejected = True   # the reference has undefined behaviour here: the product must refuse, not crash
            try:
                e = orbhip.ORBextractor(nf, sf, nl, ini, mn)
                m2, k2, d2 = e(img, None, lap)
                prod_rejected = False
            except orbhip.OrbHipError:
                prod_rejected = True
            if rejected or prod_rejected:
                ok = rejected == prod_rejected
            else:
                ok = m2 == mono and len(k) == len(k2) and np.array_equal(k.view(np.uint8), k2.view(np.uint8)) and np.array_equal(d, d2)
                if ok and img.shape[1] % 4 == 0:   # (dense batches need 4-byte row strides) also as a batch of 64 copies: batches run k_fast on two-cell-row tiles and, from 64 frames, in two passes; single frames on one-row tiles in one
                    import torch
                    kb, db, cb = e.extract_batch(torch.from_numpy(np.ascontiguousarray(np.stack([img] * 64))).cuda(), lap)
                    kb, db, cb = kb.cpu().numpy(), db.cpu().numpy(), cb.cpu().numpy()
                    for f in (0, 63):
                        nb = int(cb[f, 0])
                        ok = ok and nb == len(k) and int(cb[f, 1]) == mono and np.array_equal(kb[f, :nb].view(np.uint8).reshape(-1), k.view(np.uint8).reshape(-1)) \
                            and np.array_equal(db[f, :nb], d)
        except Exception as ex:  # noqa: BLE001
            ok = False
            print("case %d raised %r" % (seed0 + i, ex))
        print("case %3d  %4dx%-4d nf=%-4d sf=%.2f nl=%-2d th=%d/%d lap=%s  kp=%d  %s" % (seed0 + i, img.shape[1], img.shape[0], nf, sf, nl, ini, mn, lap,
                                                                                     len(k) if ok else -1, "ok" if ok else "MISMATCH"), flush=True)
        bad += not ok
    print("fuzz: %d cases, %d mismatches" % (n, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
