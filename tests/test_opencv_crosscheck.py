"""tools/opencv_crosscheck.py (the pinning aid for the oracle's OpenCV-delegated steps; it needs real OpenCV to pin anything, which this image does not
have): the CHECKER is proven here — with the oracle's own primitives behind the cv2 names every stage must come out identical, and a planted error
must be reported for exactly the stages that read the damaged file."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "opencv_crosscheck.py")


def test_checker_selftest_and_planted_error(tmp_path):
    d = str(tmp_path)
    subprocess.check_call([sys.executable, TOOL, "dump", d], stdout=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, TOOL, "selftest", d], capture_output=True, text=True)
    assert r.returncode == 0 and "divergent stages: 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("identical") == 3 * (8 * 4 + 7)       # 3 cases x (border, FAST, blur, angle per level + resize for levels 1-7)
    # one pixel of level 2 of the first case changed: level 2's own border / FAST / blur / angle stages and level 3's resize may differ, nothing else
    p = os.path.join(d, "extract_320x240", "level_2.pgm")
    raw = bytearray(open(p, "rb").read())
    raw[-5000] ^= 0x40
    open(p, "wb").write(bytes(raw))
    r = subprocess.run([sys.executable, TOOL, "selftest", d], capture_output=True, text=True)
    assert r.returncode == 1 and "divergent stages: 0" not in r.stdout
    bad = [ln for ln in r.stdout.splitlines() if "level" in ln and "identical" not in ln]
    assert bad and all(" level 2:" in ln or (" level 3:" in ln and "resize" in ln) for ln in bad), bad
    assert any("resize" in ln and " level 2:" in ln for ln in bad)           # level 2 is no longer the resize of level 1
