"""Seeded synthetic inputs for the hot path.

Images: u8 grayscale, row-major.  Smooth low-frequency background + random rectangles/discs with random
grey levels (corner rich: every pyramid level reaches its feature cap) + Gaussian noise sigma=2.
Pure numpy; used by tests, bench.py and smoke() to build inputs (never part of the timed region).
"""
import numpy as np


def synth_image(seed: int, W: int = 752, H: int = 480, n_rect: int = 400, n_disc: int = 200,
                noise: float = 2.0, contrast: float = 1.0) -> np.ndarray:
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.full((H, W), 110.0, np.float32)
    for _ in range(6):
        fx, fy = rng.uniform(0.5, 3.0, 2) * 2 * np.pi / np.array([W, H])
        ph = rng.uniform(0, 2 * np.pi)
        img += (40.0 / 6) * np.cos(fx * xx + fy * yy + ph).astype(np.float32)
    for _ in range(n_rect):
        cx, cy = rng.uniform(0, W), rng.uniform(0, H)
        hw, hh = rng.uniform(3, 40, 2)
        th = rng.uniform(0, np.pi)
        g = rng.uniform(20, 235)
        c, s = np.cos(th), np.sin(th)
        x0, x1 = int(max(0, cx - 60)), int(min(W, cx + 60))
        y0, y1 = int(max(0, cy - 60)), int(min(H, cy + 60))
        if x1 <= x0 or y1 <= y0:
            continue
        dx = xx[y0:y1, x0:x1] - cx
        dy = yy[y0:y1, x0:x1] - cy
        u = c * dx + s * dy
        v = -s * dx + c * dy
        m = (np.abs(u) <= hw) & (np.abs(v) <= hh)
        sub = img[y0:y1, x0:x1]
        sub[m] = g
    for _ in range(n_disc):
        cx, cy = rng.uniform(0, W), rng.uniform(0, H)
        r = rng.uniform(2, 15)
        g = rng.uniform(20, 235)
        x0, x1 = int(max(0, cx - r - 1)), int(min(W, cx + r + 2))
        y0, y1 = int(max(0, cy - r - 1)), int(min(H, cy + r + 2))
        if x1 <= x0 or y1 <= y0:
            continue
        m = (xx[y0:y1, x0:x1] - cx) ** 2 + (yy[y0:y1, x0:x1] - cy) ** 2 <= r * r
        sub = img[y0:y1, x0:x1]
        sub[m] = g
    img = 110.0 + (img - 110.0) * contrast
    img += rng.normal(0, noise, (H, W)).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def synth_batch(B: int, W: int = 752, H: int = 480, seed0: int = 0, **kw) -> np.ndarray:
    return np.stack([synth_image(seed0 + i, W, H, **kw) for i in range(B)])


def flat_image(W: int = 752, H: int = 480, value: int = 128) -> np.ndarray:
    """Zero-keypoint path."""
    return np.full((H, W), value, np.uint8)


def low_contrast_image(seed: int, W: int = 752, H: int = 480) -> np.ndarray:
    """Forces minThFAST retries in most cells."""
    return synth_image(seed, W, H, contrast=0.12, noise=0.5)
