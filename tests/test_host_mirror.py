"""The C++ host mirror of the reference interfaces (slam_amd/host): builds, exports the reference's
names, and — on a GPU — passes the reference's own matching tests through that interface."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "slam_amd", "lib")


def _built():
    if not os.path.exists(os.path.join(LIB, "test_matching_hip")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "slam_amd", "csrc")])
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "slam_amd", "host")])
    return os.path.join(LIB, "test_matching_hip")


def test_host_library_exports_reference_names():
    _built()
    out = subprocess.check_output(["nm", "-DC", "--defined-only", os.path.join(LIB, "libslam_host.so")], text=True)
    for name in ["SLAM::matchFeatureInArea(", "ImagePyramid<unsigned char>::ImagePyramid(", "SLAM::Feature::track(", "SLAM::Feature::project(", "SLAM::CameraState::project(", "SLAM::matchLines(",
                 "SLAM::matchPoints(", "SLAM::getSortedKeyPoints(", "slamhip::context()"]:
        assert name in out, name
    # the product library is reached only through the C ABI: no oracle symbols, no oracle dependency
    assert "so_match_feature_in_area" not in out
    deps = subprocess.check_output(["readelf", "-d", os.path.join(LIB, "libslam_host.so")], text=True)
    assert "libslam_hip.so" in deps and "oracle" not in deps


@pytest.mark.gpu
def test_reference_matching_tests_through_cpp_interface():
    exe = _built()
    r = subprocess.run([exe, os.environ.get("SLAM_PROP_TRIALS", "100")], capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, "exit status %d\n" % r.returncode + r.stdout + r.stderr
    assert r.stdout.count("passed") == 9
    assert "resident level" in r.stdout          # LineFeature::track's calls ran on the resident pyramid levels (and were timed)
