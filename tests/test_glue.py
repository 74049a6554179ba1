"""The reference-signature glue (integration/ORBmatcher_hip.cc, integration/Optimizer_hip.cc) compiled with -DORBHIP_WITH_ORBSLAM3 against the
minimal mock declarations in tests/cpp/mock_orbslam3 (boundary test infrastructure, see its README) and run: the 5 SearchByProjection overloads,
both SearchByBoW overloads, SearchForInitialization, SearchForTriangulation (pinhole / fisheye / rig), SearchBySim3, both Fuse overloads,
LocalBundleAdjustment(KeyFrame*, bool*, Map*, int&), PoseOptimization(Frame*) and
LocalInertialBA(KeyFrame*, bool*, Map*, bool, bool) (three scenes) must reproduce the oracle / the flattened path / the outcome the scene was built for.
CPU tier = emulated library, GPU tier = the real liborbhip.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_and_run(libpath, tag, tmp_path):
    exe = str(tmp_path / ("glue_test_" + tag))
    libdir, libname = os.path.dirname(libpath), os.path.basename(libpath)[3:-3]
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-C", odir], stdout=subprocess.DEVNULL)
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wno-sign-compare", "-DORBHIP_WITH_ORBSLAM3", "-I", os.path.join(ROOT, "tests", "cpp", "mock_orbslam3"),
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "glue_test.cpp"), os.path.join(ROOT, "integration", "ORBmatcher_hip.cc"),
           os.path.join(ROOT, "integration", "Optimizer_hip.cc"), "-L", libdir, "-l" + libname, "-L", odir, "-loracle", "-Wl,-rpath," + libdir,
           "-Wl,-rpath," + odir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lpthread", "-o", exe]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "glue_test OK" in out.stdout, out.stdout + out.stderr


def test_reference_signature_glue_on_emulated_library(emu_lib, tmp_path):
    import build_emu
    _build_and_run(build_emu.OUT, "emu", tmp_path)


@pytest.mark.gpu
def test_reference_signature_glue_on_hip_library(hip_lib, tmp_path):
    from orbhip import _lib
    _build_and_run(_lib.LIB_PATH, "hip", tmp_path)


def _build_and_run_extractor_cv(libpath, tag, tmp_path):
    """include/orbslam3_hip/ORBextractor.h with -DORBHIP_WITH_OPENCV: the reference's operator()(cv::InputArray, cv::InputArray,
    vector<cv::KeyPoint>&, cv::OutputArray, vector<int>&) compiled against the mock cv:: declarations and run."""
    exe = str(tmp_path / ("extractor_cv_test_" + tag))
    libdir, libname = os.path.dirname(libpath), os.path.basename(libpath)[3:-3]
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wno-sign-compare", "-DORBHIP_WITH_OPENCV", "-I", os.path.join(ROOT, "tests", "cpp", "mock_orbslam3"),
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "extractor_cv_test.cpp"), "-L", libdir, "-l" + libname,
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lpthread", "-o", exe]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "extractor_cv_test OK" in out.stdout, out.stdout + out.stderr


def test_extractor_opencv_signature_on_emulated_library(emu_lib, tmp_path):
    import build_emu
    _build_and_run_extractor_cv(build_emu.OUT, "emu", tmp_path)


@pytest.mark.gpu
def test_extractor_opencv_signature_on_hip_library(hip_lib, tmp_path):
    from orbhip import _lib
    _build_and_run_extractor_cv(_lib.LIB_PATH, "hip", tmp_path)


def _build_and_run_frame_glue(libpath, tag, tmp_path):
    """integration/Frame_hip.cc (Frame::ComputeStereoMatches / UndistortKeyPoints / ComputeStereoFishEyeMatches) with the extractor adapter's
    OpenCV-signature branch standing in for ORB_SLAM3::ORBextractor, against the oracle's restatements of the three loops."""
    exe = str(tmp_path / ("frame_glue_test_" + tag))
    libdir, libname = os.path.dirname(libpath), os.path.basename(libpath)[3:-3]
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-C", odir], stdout=subprocess.DEVNULL)
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wno-sign-compare", "-DORBHIP_WITH_ORBSLAM3", "-DORBHIP_WITH_OPENCV", "-I", os.path.join(ROOT, "tests", "cpp", "mock_orbslam3"),
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "frame_glue_test.cpp"), os.path.join(ROOT, "integration", "Frame_hip.cc"),
           "-L", libdir, "-l" + libname, "-L", odir, "-loracle", "-Wl,-rpath," + libdir, "-Wl,-rpath," + odir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib",
           "-lpthread", "-o", exe]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "frame_glue_test OK" in out.stdout, out.stdout + out.stderr


def test_frame_glue_on_emulated_library(emu_lib, tmp_path):
    import build_emu
    _build_and_run_frame_glue(build_emu.OUT, "emu", tmp_path)


@pytest.mark.gpu
def test_frame_glue_on_hip_library(hip_lib, tmp_path):
    from orbhip import _lib
    _build_and_run_frame_glue(_lib.LIB_PATH, "hip", tmp_path)


def _build_and_run_glue_fault(libpath, tag, tmp_path):
    """tests/cpp/glue_fault_test.cpp: the glue functions replace bodies that never throw — with one C-ABI entry point interposed to fail on demand
    they must return quietly (0 matches / 0 inliers / void), leave frame and map untouched and count the failure; a LocalBundleAdjustment window
    without edges leaves through the reference's own exit."""
    exe = str(tmp_path / ("glue_fault_test_" + tag))
    libdir, libname = os.path.dirname(libpath), os.path.basename(libpath)[3:-3]
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wno-sign-compare", "-DORBHIP_WITH_ORBSLAM3", "-I", os.path.join(ROOT, "tests", "cpp", "mock_orbslam3"),
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "glue_fault_test.cpp"), os.path.join(ROOT, "integration", "ORBmatcher_hip.cc"),
           os.path.join(ROOT, "integration", "Optimizer_hip.cc"), "-L", libdir, "-l" + libname, "-Wl,-rpath," + libdir,
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lpthread", "-ldl", "-o", exe]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "glue_fault_test OK" in out.stdout, out.stdout + out.stderr
    assert out.stderr.count("[orbhip]") == 4, out.stderr   # one report per injected fault


def test_glue_failure_policy_on_emulated_library(emu_lib, tmp_path):
    import build_emu
    _build_and_run_glue_fault(build_emu.OUT, "emu", tmp_path)


@pytest.mark.gpu
def test_glue_failure_policy_on_hip_library(hip_lib, tmp_path):
    from orbhip import _lib
    _build_and_run_glue_fault(_lib.LIB_PATH, "hip", tmp_path)
