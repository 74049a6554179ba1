#!/usr/bin/env python3
"""Pinning aid for the oracle's OpenCV-delegated steps (oracle/README.md "Pinning status").

The reference hands five steps of ORBextractor to OpenCV — cv::resize (src/ORBextractor.cc:1171), cv::copyMakeBorder (:1173-1179), cv::FAST (:808-838),
cv::GaussianBlur (:1121) and cv::fastAtan2 (:101) — and ships no vectors; OpenCV is absent from the build image, so oracle/orb_oracle.cpp restates them
and is "parity unpinned".  This tool makes the pin a one-command job for somebody who HAS OpenCV:

  python tools/opencv_crosscheck.py dump  DIR     (here, needs only this repository: writes the oracle's per-stage outputs for the golden images)
  python tools/opencv_crosscheck.py selftest DIR  (here: the checker with the oracle's own primitives behind the cv2 names — every stage must be identical)
  python tools/opencv_crosscheck.py check DIR     (anywhere with `import cv2`: recomputes every stage with real OpenCV FROM THE ORACLE'S INPUT of that stage —
                                                   a divergence is attributed to one stage — and prints the first divergence per stage and case)
  tools/opencv_crosscheck.cpp                     (the same checks in C++ against the OpenCV the reference is built with, plus — with the reference's
                                                   src/ORBextractor.cc compiled in — the reference's own operator() against the oracle's final output)

DIR/<case>/: config.txt, image.pgm, level_<l>.pgm, bordered_<l>.pgm, blurred_<l>.pgm (levels that hold key points), candidates_<l>.txt (x y score per line,
relative to minBorder = 16 like vToDistributeKeys), keypoints.bin (cv::KeyPoint records, 28 B each, output order), descriptors.bin (32 B each),
angles_<l>.txt (x y angle-as-u32 in level coordinates).

Expected to agree: OpenCV 3.2 / 3.3 (plain C++ paths; what the reference targets: README.md:106 "OpenCV ... at least 3.0", the ROS cache points at 3.2).
Known to take another path: OpenCV >= 3.4.1 / 4.x cv::GaussianBlur on CV_8U uses a fixed-point kernel (bit-exact smooth) whose rounding differs from
the float-kernel + saturate_cast path restated here — a build of the reference against those versions is itself expected to give (slightly) different
descriptors than one against 3.2; `check` reports the number of differing blurred pixels and goes on.  cv::resize INTER_LINEAR on CV_8U, cv::FAST,
copyMakeBorder and fastAtan2 have not changed their arithmetic across 3.x / 4.x."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [   # (name, seed, W, H, nfeatures, lap) — the first two are tests/golden/extract_*.npz, the third is the headline geometry
    ("extract_320x240", 1, 320, 240, 300, (0, 1000)),
    ("extract_400x300_lap", 2, 400, 300, 400, (120, 260)),
    ("euroc_752x480", 0, 752, 480, 1000, (0, 1000)),
]
CFG = dict(scale=1.2, nlevels=8, ini=20, min=7)


def write_pgm(path, a):
    a = np.ascontiguousarray(a, np.uint8)
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (a.shape[1], a.shape[0]))
        f.write(a.tobytes())


def read_pgm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P5"
        w, h = [int(v) for v in f.readline().split()]
        assert int(f.readline()) == 255
        return np.frombuffer(f.read(w * h), np.uint8).reshape(h, w)


def dump(out):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import oracle_lib as O
    from orbhip.synth import synth_image
    for name, seed, W, H, nf, lap in CASES:
        d = os.path.join(out, name)
        os.makedirs(d, exist_ok=True)
        img = synth_image(seed, W, H, n_rect=90, n_disc=45) if name != "euroc_752x480" else synth_image(seed)
        o = O.OrbOracle(nf, CFG["scale"], CFG["nlevels"], CFG["ini"], CFG["min"])
        mono, k, desc = o.extract(img, *lap)
        with open(os.path.join(d, "config.txt"), "w") as f:
            f.write("%d %d %d %.9g %d %d %d %d %d %d %d " % (W, H, nf, CFG["scale"], CFG["nlevels"], CFG["ini"], CFG["min"], lap[0], lap[1], mono, len(k)))
            f.write(" ".join(str(int(v)) for v in o.tables()["umax"]) + "\n")       # u_max of IC_Angle's circular patch (ORBextractor.cc:447-468)
        write_pgm(os.path.join(d, "image.pgm"), img)
        for l in range(CFG["nlevels"]):
            write_pgm(os.path.join(d, "level_%d.pgm" % l), o.level_image(l))
            write_pgm(os.path.join(d, "bordered_%d.pgm" % l), o.level_bordered(l))
            b = o.level_blurred(l)
            if b is not None:
                write_pgm(os.path.join(d, "blurred_%d.pgm" % l), b)
            np.savetxt(os.path.join(d, "candidates_%d.txt" % l), o.level_candidates(l), fmt="%d")
            kl, _ = o.level_keypoints(l)
            np.savetxt(os.path.join(d, "angles_%d.txt" % l), np.stack([kl["x"].astype(np.int64), kl["y"].astype(np.int64),
                                                                         kl["angle"].view(np.uint32).astype(np.int64)], 1).reshape(-1, 3), fmt="%d")
        k.tofile(os.path.join(d, "keypoints.bin"))
        desc.tofile(os.path.join(d, "descriptors.bin"))
        print("%s: %d key points, monoIndex %d -> %s" % (name, len(k), mono, d))


def first_diff(a, b):
    if a.shape != b.shape:
        return "shapes differ: %s vs %s" % (a.shape, b.shape)
    idx = np.argwhere(a != b)
    if len(idx) == 0:
        return None
    y, x = idx[0][:2]
    return "%d of %d values differ; first at (x=%d, y=%d): OpenCV %d, oracle %d" % (len(idx), a.size, x, y, int(a[y, x]), int(b[y, x]))


class _OracleAsCv2:
    """`selftest`: the oracle's own primitives behind the cv2 names `check` uses.  Every stage must then come out identical — which proves the
    checker's stage logic (cell loop, coordinates, IC_Angle sums) and leaves the OpenCV functions as the only thing a real run swaps in."""
    __version__ = "none (the oracle's restatements: checker self-test)"
    INTER_LINEAR, BORDER_REFLECT_101 = 1, 4

    def __init__(self):
        for p in (ROOT, os.path.join(ROOT, "tests")):
            sys.path.insert(0, p)
        import oracle_lib as O
        self.O = O

    def resize(self, src, size, interpolation=None):
        return self.O.resize_linear(src, size[0], size[1])

    def copyMakeBorder(self, src, t, b, l, r, kind):
        return np.pad(src, ((t, b), (l, r)), mode="reflect")

    def GaussianBlur(self, src, k, sx, sy, borderType=None):
        return self.O.gaussian7(src)

    def fastAtan2(self, y, x):
        return self.O.fast_atan2(y, x)

    def FastFeatureDetector_create(self, th, nms):
        O = self.O

        class _KP:
            def __init__(self, x, y, r):
                self.pt, self.response = (float(x), float(y)), float(r)

        class _Det:
            def detect(self, roi):
                return [_KP(*t) for t in O.fast(roi, th)]
        return _Det()


def check(out, cv2=None):
    if cv2 is None:
        import cv2
    print("OpenCV", cv2.__version__)
    bad = 0
    for name, *_ in CASES:
        d = os.path.join(out, name)
        cfg = open(os.path.join(d, "config.txt")).read().split()
        W, H, nf, scale, nl, ini, mn, lap0, lap1, mono, nk = cfg[:11]
        umax = [int(v) for v in cfg[11:27]]
        nl, ini, mn = int(nl), int(ini), int(mn)
        lev = [read_pgm(os.path.join(d, "level_%d.pgm" % l)) for l in range(nl)]
        rep = lambda stage, l, msg: print("  %-34s level %d: %s" % (stage, l, msg or "identical"))
        print(name)
        for l in range(nl):
            # cv::resize(mvImagePyramid[level-1], mvImagePyramid[level], sz, 0, 0, INTER_LINEAR)   ORBextractor.cc:1171
            if l:
                m = first_diff(cv2.resize(lev[l - 1], (lev[l].shape[1], lev[l].shape[0]), interpolation=cv2.INTER_LINEAR), lev[l]); bad += m is not None
                rep("cv::resize INTER_LINEAR", l, m)
            # copyMakeBorder(..., EDGE_THRESHOLD x 4, BORDER_REFLECT_101)   :1173-1179
            m = first_diff(cv2.copyMakeBorder(lev[l], 19, 19, 19, 19, cv2.BORDER_REFLECT_101), read_pgm(os.path.join(d, "bordered_%d.pgm" % l))); bad += m is not None
            rep("cv::copyMakeBorder REFLECT_101", l, m)
            # per-cell FAST(iniThFAST), retry at minThFAST, non-max suppression on   :763-855
            cand = np.loadtxt(os.path.join(d, "candidates_%d.txt" % l), dtype=np.int64).reshape(-1, 3)
            got = set()
            h, w = lev[l].shape
            minBX, minBY, maxBX, maxBY = 16, 16, w - 16, h - 16
            fw, fh = float(maxBX - minBX), float(maxBY - minBY)
            nCols, nRows = int(fw / 30), int(fh / 30)
            wCell, hCell = int(np.ceil(fw / nCols)), int(np.ceil(fh / nRows))
            for i in range(nRows):
                iniY = minBY + i * hCell
                maxY = min(iniY + hCell + 6, maxBY)
                if iniY >= maxBY - 3:
                    continue
                for j in range(nCols):
                    iniX = minBX + j * wCell
                    maxX = min(iniX + wCell + 6, maxBX)
                    if iniX >= maxBX - 6:
                        continue
                    roi = np.ascontiguousarray(lev[l][iniY:maxY, iniX:maxX])
                    kp = cv2.FastFeatureDetector_create(ini, True).detect(roi)
                    if not kp:
                        kp = cv2.FastFeatureDetector_create(mn, True).detect(roi)
                    for q in kp:
                        got.add((int(q.pt[0]) + j * wCell, int(q.pt[1]) + i * hCell, int(q.response)))
            want = set(map(tuple, cand.tolist()))
            m = None if got == want else "%d candidates only in OpenCV, %d only in the oracle; e.g. %s" % (len(got - want), len(want - got), sorted(got ^ want)[:3])
            bad += m is not None
            rep("cv::FAST per cell (+ retry)", l, m)
            # GaussianBlur(workingMat, workingMat, Size(7, 7), 2, 2, BORDER_REFLECT_101)   :1121
            pb = os.path.join(d, "blurred_%d.pgm" % l)
            if os.path.exists(pb):
                m = first_diff(cv2.GaussianBlur(lev[l], (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101), read_pgm(pb)); bad += m is not None
                rep("cv::GaussianBlur 7x7 sigma 2", l, (m + "   [OpenCV >= 3.4.1: fixed-point path, see the module docstring]") if m else m)
            # IC_Angle: fastAtan2((float)m_01, (float)m_10)   :75-102
            ang = np.loadtxt(os.path.join(d, "angles_%d.txt" % l), dtype=np.int64).reshape(-1, 3)
            B = cv2.copyMakeBorder(lev[l], 19, 19, 19, 19, cv2.BORDER_REFLECT_101).astype(np.int64)
            nb = 0
            for x, y, bits in ang:
                m01 = m10 = 0
                cy, cx = y + 19, x + 19
                for u in range(-15, 16):
                    m10 += u * B[cy, cx + u]
                for v in range(1, 16):
                    vs = 0
                    for u in range(-umax[v], umax[v] + 1):
                        p, q = B[cy + v, cx + u], B[cy - v, cx + u]
                        vs += p - q
                        m10 += u * (p + q)
                    m01 += v * vs
                a = np.float32(cv2.fastAtan2(float(np.float32(m01)), float(np.float32(m10))))
                nb += int(a.view(np.uint32)) != int(bits)
            bad += nb > 0
            rep("IC_Angle / cv::fastAtan2", l, None if nb == 0 else "%d of %d angles differ in their bits" % (nb, len(ang)))
    print("divergent stages: %d" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    if len(sys.argv) != 3 or sys.argv[1] not in ("dump", "check", "selftest"):
        sys.exit(__doc__)
    if sys.argv[1] == "selftest":
        sys.exit(check(sys.argv[2], _OracleAsCv2()))
    sys.exit(dump(sys.argv[2]) if sys.argv[1] == "dump" else check(sys.argv[2]))
