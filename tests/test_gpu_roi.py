"""cv::Mat views: the reference passes ROIs whose row step exceeds their width (matching.cpp:67-68, SLAM.hpp:92-98
take Mat_ by reference), so every host entry point takes (pointer, rows, cols, pitch).  These tests hand the C ABI
views into larger arrays - image, patch, patch mask, area mask, frames of a batch - and compare with the oracle
on contiguous copies."""
import ctypes as C

import numpy as np
import pytest

from slam_amd import api, capi

pytestmark = pytest.mark.gpu


def _view(rng, h, w, pad_r, pad_c, values=256):
    """An h x w view inside a larger random array (the surroundings are different random bytes)."""
    parent = rng.integers(0, values, (h + 2 * pad_r, w + 2 * pad_c), dtype=np.uint8)
    v = parent[pad_r:pad_r + h, pad_c:pad_c + w]
    assert not v.flags["C_CONTIGUOUS"] or pad_c == 0
    return parent, v


def _same_float(a, b):
    return np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32) or (np.isnan(a) and np.isnan(b))


def test_match_feature_in_area_on_views(ctx, oracle):
    rng = np.random.default_rng(404)
    L = capi.lib()
    for it in range(12):
        w, h = int(rng.integers(40, 120)), int(rng.integers(30, 90))
        pw, ph = int(rng.integers(3, 20)), int(rng.integers(3, 16))
        keep_im, im = _view(rng, h, w, 3, int(rng.integers(1, 9)))
        keep_p, patch = _view(rng, ph, pw, 2, int(rng.integers(1, 7)))
        keep_m, pm = _view(rng, ph, pw, 1, int(rng.integers(1, 5)), values=2)
        area = (int(rng.integers(-5, w // 2)), int(rng.integers(-5, h // 2)), int(rng.integers(4, 40)), int(rng.integers(4, 30)))
        keep_a, am = _view(rng, area[3], area[2], 2, int(rng.integers(1, 6)), values=2)
        use_pm, use_am, stride = bool(it & 1), bool(it & 2), int(rng.integers(1, 4))
        ref = oracle.match(im.copy(), patch.copy(), pm.copy() if use_pm else None, area, am.copy() if use_am else None, stride,
                           False, want_score=True)
        scores = np.full(area[2] * area[3], np.nan, np.float32)
        out = capi.MatchResult()
        rc = L.slam_hip_match_feature_in_area(
            ctx.handle, im.ctypes.data, w, h, im.strides[0], patch.ctypes.data, pw, ph, patch.strides[0],
            pm.ctypes.data if use_pm else None, pm.strides[0], area[0], area[1], area[2], area[3],
            am.ctypes.data if use_am else None, am.strides[0], stride, 0, scores.ctypes.data, C.byref(out))
        capi.check(ctx.handle, rc)
        assert (out.x, out.y) == (ref["x"], ref["y"]) and _same_float(out.response, ref["response"])
        rs = ref["score"]
        got = scores[:rs.size].reshape(rs.shape)
        same = (got.view(np.uint32) == rs.view(np.uint32)) | (np.isnan(got) & np.isnan(rs))
        assert same.all()


def test_pyramid_and_fast_on_pitched_frames(ctx, oracle):
    """A batch of frames stored with a row pitch larger than the width and a frame stride larger than pitch*h."""
    rng = np.random.default_rng(405)
    w, h, nfr, pitch = 150, 97, 3, 171
    stride = pitch * h + 333
    buf = rng.integers(0, 256, nfr * stride + 64, dtype=np.uint8)
    frames = [np.lib.stride_tricks.as_strided(buf[f * stride:], (h, w), (pitch, 1)) for f in range(nfr)]
    subs = [1.0, 2.0, 3.0]
    L = capi.lib()
    pyr = C.c_void_p()
    fs = (C.c_float * 3)(*subs)
    capi.check(ctx.handle, L.slam_hip_pyramid_create(ctx.handle, w, h, 3, fs, nfr, C.byref(pyr)))
    try:
        capi.check(ctx.handle, L.slam_hip_pyramid_build(ctx.handle, pyr, buf.ctypes.data, pitch, stride))
        for f in range(nfr):
            levels = oracle.pyramid(frames[f].copy(), subs)
            for l, ref in enumerate(levels):
                got = np.zeros((ref.shape[0], ref.shape[1] + 5), np.uint8)         # download into a pitched host array
                capi.check(ctx.handle, L.slam_hip_pyramid_download(ctx.handle, pyr, l, f, got.ctypes.data, got.strides[0]))
                assert np.array_equal(got[:, :ref.shape[1]], ref), (f, l)
                assert not got[:, ref.shape[1]:].any()
    finally:
        L.slam_hip_pyramid_destroy(ctx.handle, pyr)
    cap = 4096
    out = (capi.KeyPoint * (nfr * cap))()
    counts = (C.c_int32 * nfr)()
    capi.check(ctx.handle, L.slam_hip_fast_detect(ctx.handle, buf.ctypes.data, w, h, pitch, stride, nfr, 20, 1, out, cap, counts))
    for f in range(nfr):
        ref = oracle.fast(frames[f].copy(), 20, True)
        assert counts[f] == len(ref)
        got = np.array([(out[f * cap + i].x, out[f * cap + i].y, out[f * cap + i].response) for i in range(counts[f])], np.float32).reshape(-1, 3)
        assert np.array_equal(got, ref.astype(np.float32))
    kp = (capi.KeyPoint * cap)()
    n = C.c_int32()
    thr = C.c_float(10.0)
    capi.check(ctx.handle, L.slam_hip_get_sorted_keypoints(ctx.handle, frames[1].ctypes.data, w, h, pitch, 60, C.byref(thr), kp, cap, C.byref(n)))
    ref, rthr = oracle.get_sorted_keypoints(frames[1].copy(), 60, 10.0)
    assert n.value == len(ref) and thr.value == rthr
    got = np.array([(kp[i].x, kp[i].y, kp[i].response) for i in range(n.value)], np.float32).reshape(-1, 3)
    assert np.array_equal(got, ref.astype(np.float32))
