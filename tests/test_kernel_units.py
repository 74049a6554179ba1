"""k_fast's score functions as plain C++ (tests/cpp/fast_score_test.cpp includes the product source against the HIP emulator header): the one-polarity,
running-minima form fast_S_pk against the straightforward fast_S on two million random and adversarial 7 x 7 patches, and the byte-parallel pre-test
as a necessary condition for a corner at every threshold 0 .. 255."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fast_score_forms_agree():
    out = os.path.join(ROOT, "tests", "emu", "build", "fast_score_test")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-w", "-I", os.path.join(ROOT, "tests", "emu"), "-I", os.path.join(ROOT, "include"), "-x", "c++",
                           os.path.join(ROOT, "tests", "cpp", "fast_score_test.cpp"), "-o", out, "-lpthread"])
    r = subprocess.run([out, "2000000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 mismatches" in r.stdout and " 0 corners dropped" in r.stdout, r.stdout


def test_stereo_cull_median_on_adversarial_sads():
    """k_stereo_cull under the emulator on ties, single / no valid entries, bin boundaries, the largest SAD: the cull threshold is 1.5 * 1.4 * the value
    at rank size / 2 of the valid SADs (tests/cpp/stereo_cull_test.cpp)."""
    out = os.path.join(ROOT, "tests", "emu", "build", "stereo_cull_test")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-w", "-I", os.path.join(ROOT, "tests", "emu"), "-I", os.path.join(ROOT, "include"), "-x", "c++",
                           os.path.join(ROOT, "tests", "cpp", "stereo_cull_test.cpp"), "-o", out, "-lpthread"])
    r = subprocess.run([out], capture_output=True, text=True)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout + r.stderr


def test_describe_arithmetic_forms():
    """k_describe2's round-4 arithmetic (tests/cpp/describe_arith_test.cpp): cvRound by the float adder == lrintf, the sampled point's address from raw
    rint bits with the biases folded into one constant == the plain form, fastAtan2 as selects == its two-branch form."""
    out = os.path.join(ROOT, "tests", "emu", "build", "describe_arith_test")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-w", "-ffp-contract=off", "-I", os.path.join(ROOT, "tests", "emu"), "-I", os.path.join(ROOT, "include"), "-x", "c++",
                           os.path.join(ROOT, "tests", "cpp", "describe_arith_test.cpp"), "-o", out, "-lpthread"])
    r = subprocess.run([out], capture_output=True, text=True)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout + r.stderr
