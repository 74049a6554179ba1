"""Pins the oracle's restatement of the OpenCV-delegated steps to their mathematical definitions
(the reference has no tests/golden vectors: SURVEY.md §4, §8(c))."""
import hashlib
import math
import os
import re

import numpy as np
import pytest

import oracle_lib as O
from orbhip.synth import synth_image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATTERN_SHA = "2164181aea6ff9ac426ca512d5130d15e1f6e3cd47b1cbdd568bbe1e55d49023"  # reference ORBextractor.cc:148-406


def _pattern(path):
    txt = open(path).read()
    body = "\n".join(l for l in txt.splitlines() if not l.startswith("//"))
    return [int(x) for x in re.findall(r"-?\d+", body)]


@pytest.mark.parametrize("rel", ["oracle/orb_pattern_oracle.inc", "awesome-orb-slam3-3dvisioncraft-version_amd/csrc/orb_pattern.inc"])
def test_pattern_table_is_reference_data(rel):
    nums = _pattern(os.path.join(ROOT, rel))
    assert len(nums) == 1024
    assert hashlib.sha256(bytes([(n + 256) % 256 for n in nums])).hexdigest() == PATTERN_SHA
    assert max(math.hypot(nums[i], nums[i + 1]) for i in range(0, 1024, 2)) < 18.5  # blurred-patch radius 18 suffices


def test_tables_match_reference_constructor():
    # SURVEY.md §8 derived geometry (ORBextractor.cc:413-467)
    t = O.OrbOracle(1000, 1.2, 8, 20, 7).tables()
    assert t["nfeat"].tolist() == [217, 181, 151, 126, 105, 87, 73, 60]
    assert t["umax"].tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert t["nfeat"].sum() == 1000
    t = O.OrbOracle(1500, 1.2, 8, 20, 7).tables()
    assert t["nfeat"].tolist() == [326, 271, 226, 189, 157, 131, 109, 91]
    np.testing.assert_allclose(t["scale"], 1.2 ** np.arange(8), rtol=1e-6)


def test_level_sizes():
    o = O.OrbOracle()
    o.extract(synth_image(0), 0, 0)
    assert [o.level_size(l) for l in range(8)] == [(752, 480), (627, 400), (522, 333), (435, 278), (363, 231),
                                                    (302, 193), (252, 161), (210, 134)]


def test_cvround_half_even():
    L = O.lib()
    assert [L.oro_cvround(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]


def test_resize_close_to_float_bilinear():
    rng = np.random.default_rng(0)
    src = rng.integers(0, 256, (97, 131), dtype=np.uint8)
    dw, dh = 109, 81
    got = O.resize_linear(src, dw, dh).astype(np.float64)
    sx = (np.arange(dw) + 0.5) * (131 / dw) - 0.5
    sy = (np.arange(dh) + 0.5) * (97 / dh) - 0.5
    x0 = np.clip(np.floor(sx).astype(int), 0, 130); x1 = np.clip(x0 + 1, 0, 130); fx = np.clip(sx - np.floor(sx), 0, 1)
    y0 = np.clip(np.floor(sy).astype(int), 0, 96); y1 = np.clip(y0 + 1, 0, 96); fy = np.clip(sy - np.floor(sy), 0, 1)
    s = src.astype(np.float64)
    ref = (s[y0][:, x0] * (1 - fx) + s[y0][:, x1] * fx) * (1 - fy)[:, None] + (s[y1][:, x0] * (1 - fx) + s[y1][:, x1] * fx) * fy[:, None]
    assert np.abs(got - ref).max() <= 1.0  # 11-bit coefficients + truncating shifts
    const = np.full((50, 60), 77, np.uint8)
    assert (O.resize_linear(const, 50, 41) == 77).all()


def test_gaussian_kernel_and_blur():
    imp = np.zeros((15, 15), np.uint8)
    imp[7, 7] = 255
    out = O.gaussian7(imp).astype(int)
    k = np.array([18, 34, 49, 55, 49, 34, 18])
    ref = (np.outer(k, k) * 255 + 32768) >> 16
    assert np.array_equal(out[4:11, 4:11], ref)
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (40, 53), dtype=np.uint8)
    g = np.exp(-np.arange(-3, 4) ** 2 / 8.0); g /= g.sum()
    pad = np.pad(img.astype(np.float64), 3, mode="reflect")  # numpy 'reflect' == BORDER_REFLECT_101
    tmp = sum(g[i] * pad[:, i:i + 53] for i in range(7))
    ref = sum(g[j] * tmp[j:j + 40, :] for j in range(7))
    assert np.abs(O.gaussian7(img).astype(np.float64) - ref).max() <= 2.7  # sum(k)=257 -> up to +0.78% (2 levels at 255) + rounding
    assert (O.gaussian7(np.full((20, 20), 200, np.uint8)) == ((200 * 257 * 257 + 32768) >> 16)).all()


def _fast_bruteforce(img, th):
    """FAST-9/16 by definition + score = max threshold keeping the corner, 3x3 strict NMS inside the image."""
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    H, W = img.shape
    im = img.astype(int)
    score = np.zeros((H, W), int)
    for y in range(3, H - 3):
        for x in range(3, W - 3):
            d = [im[y, x] - im[y + dy, x + dx] for dx, dy in ring]
            best = -1
            for i in range(16):
                arc = [d[(i + j) % 16] for j in range(9)]
                best = max(best, min(arc), min(-a for a in arc))
            if best > th:
                score[y, x] = best - 1
    out = []
    for y in range(3, H - 3):
        for x in range(3, W - 3):
            s = score[y, x]
            if s > 0 and all(s > score[y + dy, x + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dx or dy)):
                if s >= th:
                    out.append((x, y, s))
    return out


@pytest.mark.parametrize("th", [7, 20])
def test_fast_matches_definition(th):
    img = synth_image(11, 64, 48, n_rect=12, n_disc=6)
    got = [tuple(r) for r in O.fast(img, th).tolist()]
    assert got == _fast_bruteforce(img, th)
    assert len(got) > 0


def test_fast_atan2_accuracy_and_quadrants():
    rng = np.random.default_rng(2)
    for _ in range(2000):
        y, x = rng.normal(size=2) * 1000
        ref = math.degrees(math.atan2(y, x)) % 360
        got = O.fast_atan2(y, x)
        err = abs(got - ref)
        assert min(err, 360 - err) < 0.02
    assert O.fast_atan2(0, 1) == 0.0 and abs(O.fast_atan2(1, 0) - 90) < 1e-3 and abs(O.fast_atan2(0, -1) - 180) < 1e-3


def test_det_sincos_is_correctly_rounded():
    rng = np.random.default_rng(3)
    angs = np.concatenate([rng.uniform(0, 2 * np.pi, 20000), np.arange(0, 360, 0.25) * np.pi / 180]).astype(np.float32)
    bad = 0
    for a in angs:
        s, c = O.sincos(a)
        bad += (np.float32(math.sin(float(a))) != np.float32(s)) + (np.float32(math.cos(float(a))) != np.float32(c))
    assert bad == 0
