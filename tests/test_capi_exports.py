"""CPU-side checks of the C ABI: the library loads and exports exactly what include/slam_hip.h
declares, and argument validation that needs no GPU works."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "slam_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(slam_hip_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_entry_points():
    names = _declared()
    for need in ("slam_hip_create", "slam_hip_pyramid_build_d", "slam_hip_match_batch_d",
                 "slam_hip_match_feature_in_area", "slam_hip_track_batch_d", "slam_hip_fast_detect_d",
                 "slam_hip_get_sorted_keypoints"):
        assert need in names


def test_library_exports_every_declared_symbol():
    from slam_amd import capi
    assert os.path.exists(capi.LIB_PATH), "build with `make -C slam_amd/csrc` (or __graft_entry__.build())"
    L = C.CDLL(capi.LIB_PATH)
    for name in _declared():
        assert hasattr(L, name), "missing export: " + name
    # the Python binding covers the same set
    assert sorted(capi.SIGNATURES) == _declared()
    capi.lib()
    assert capi.lib().slam_hip_version().decode().startswith("slam_hip")


def test_struct_layouts_match_the_header():
    from slam_amd import capi
    assert C.sizeof(capi.MatchJob) == 72
    assert C.sizeof(capi.MatchResult) == 24
    assert C.sizeof(capi.TrackJob) == 72
    assert C.sizeof(capi.TrackResult) == 28
    assert C.sizeof(capi.KeyPoint) == 12
    assert C.sizeof(capi.DescJob) == 32
    assert C.sizeof(capi.DescResult) == 20
    assert C.sizeof(capi.WarpJob) == 80
    assert C.sizeof(capi.DiscJob) == 24
    assert C.sizeof(capi.LineJob) == 56
    assert C.sizeof(capi.EllipseJob) == 48


def test_null_context_is_rejected_without_touching_a_gpu():
    from slam_amd import capi
    L = capi.lib()
    assert L.slam_hip_synchronize(None) == capi.EINVAL
    assert L.slam_hip_match_batch(None, None, None, 0, None, 0, None) == capi.EINVAL
    assert L.slam_hip_fast_scratch_bytes(0, 0, 0, 0) == 0
    assert L.slam_hip_fast_scratch_bytes(1920, 1080, 64, 65536) > 64 * 65536 * 4
