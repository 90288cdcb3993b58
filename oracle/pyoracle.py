"""ctypes loader for the CPU oracle (oracle/slam_oracle.c) — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
All image arguments are C-contiguous uint8 numpy arrays (rows = height).
"""
import ctypes as C
import fcntl
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libslam_oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with gcc (seconds).  Safe when several processes call it at once (the ranks of a multi-GPU
    bench): the check and the build run under a file lock, so one process builds and the others find the result."""
    src = [os.path.join(_HERE, f) for f in ("slam_oracle.c", "slam_oracle.h", "Makefile")]
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    with open(os.path.join(os.path.dirname(_SO), ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or not os.path.exists(_SO) or any(
                os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
            subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


class MatchOut(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("response", C.c_float),
                ("dx", C.c_float), ("dy", C.c_float),
                ("aw", C.c_int), ("ah", C.c_int), ("ax", C.c_int), ("ay", C.c_int)]


class TrackOut(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("response", C.c_float),
                ("dx", C.c_float), ("dy", C.c_float), ("levels_run", C.c_int)]


class KeyPoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("response", C.c_float)]


class Projection(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("w", C.c_int), ("h", C.c_int), ("m", C.c_float * 9), ("valid", C.c_int)]


class Ellipse(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("w", C.c_int), ("h", C.c_int), ("ux", C.c_float), ("uy", C.c_float),
                ("q11", C.c_float), ("q12", C.c_float), ("q22", C.c_float), ("valid", C.c_int)]


class PropCase(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("pw", C.c_int), ("ph", C.c_int),
                ("xpatch", C.c_int), ("ypatch", C.c_int),
                ("ax", C.c_int), ("ay", C.c_int), ("aw", C.c_int), ("ah", C.c_int),
                ("im", C.POINTER(C.c_uint8)), ("patch", C.POINTER(C.c_uint8)),
                ("patch_mask", C.POINTER(C.c_uint8)), ("area_mask", C.POINTER(C.c_uint8))]


_u8p = C.POINTER(C.c_uint8)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.so_pyramid_level_size.argtypes = [C.c_int, C.c_int, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.so_pyramid_level_size.restype = None
        L.so_resize_u8_linear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_ssize_t,
                                          C.c_void_p, C.c_int, C.c_int, C.c_ssize_t]
        L.so_resize_u8_linear.restype = C.c_int
        L.so_match_feature_in_area.argtypes = [
            C.c_void_p, C.c_int, C.c_int, C.c_ssize_t,
            C.c_void_p, C.c_int, C.c_int, C.c_ssize_t,
            C.c_void_p, C.c_ssize_t,
            C.c_int, C.c_int, C.c_int, C.c_int,
            C.c_void_p, C.c_ssize_t,
            C.c_int, C.c_int, C.c_void_p, C.POINTER(MatchOut)]
        L.so_match_feature_in_area.restype = C.c_int
        L.so_track.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                               C.POINTER(C.c_float), C.c_int,
                               C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                               C.c_float, C.c_float, C.c_float, C.c_int, C.POINTER(TrackOut)]
        L.so_track.restype = C.c_int
        L.so_track_ex.argtypes = L.so_track.argtypes[:-1] + [C.c_void_p, C.c_void_p, C.POINTER(TrackOut)]
        L.so_track_ex.restype = C.c_int
        L.so_fast_score_map.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_ssize_t, C.c_void_p]
        L.so_fast_score_map.restype = None
        L.so_fast.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_int,
                              C.POINTER(KeyPoint), C.c_int]
        L.so_fast.restype = C.c_int
        L.so_get_sorted_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_ssize_t, C.c_size_t,
                                              C.POINTER(C.c_float), C.POINTER(KeyPoint), C.c_int]
        L.so_get_sorted_keypoints.restype = C.c_int
        L.so_sort_keypoints.argtypes = [C.c_void_p, C.c_int]
        L.so_sort_keypoints.restype = None
        L.so_select_new_points.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_void_p]
        L.so_select_new_points.restype = C.c_int
        L.so_extract_descriptor.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_ssize_t, C.c_float, C.c_float, C.c_int, C.c_int,
                                            C.c_void_p] + [C.POINTER(C.c_int)] * 4
        L.so_extract_descriptor.restype = C.c_int
        L.so_project_setup.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(Projection)]
        L.so_project_setup.restype = C.c_int
        L.so_warp_perspective_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_ssize_t, C.c_void_p, C.c_int,
                                             C.c_void_p, C.c_int, C.c_int, C.c_ssize_t]
        L.so_warp_perspective_u8.restype = C.c_int
        L.so_disc_mask.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.so_disc_mask.restype = C.c_int
        L.so_ellipse_setup.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_float, C.c_float, C.POINTER(Ellipse)]
        L.so_ellipse_setup.restype = C.c_int
        L.so_ellipse_mask.argtypes = [C.POINTER(Ellipse), C.c_void_p]
        L.so_ellipse_mask.restype = C.c_int
        L.so_line_track.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_int,
                                    C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                    C.c_float, C.c_int, C.POINTER(TrackOut)]
        L.so_line_track.restype = C.c_int
        L.so_proptest_srand.argtypes = [C.c_uint]
        L.so_proptest_srand.restype = None
        L.so_proptest_next.argtypes = [C.c_int, C.c_int, C.POINTER(PropCase)]
        L.so_proptest_next.restype = None
        L.so_proptest_free.argtypes = [C.POINTER(PropCase)]
        L.so_proptest_free.restype = None
        L.so_proptest_accept.argtypes = [C.POINTER(PropCase), C.c_int, C.c_int, C.c_float]
        L.so_proptest_accept.restype = C.c_int
        _lib = L
    return _lib


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.ndim == 2
    return a


def pyramid_level_size(w, h, sub):
    dw, dh = C.c_int(), C.c_int()
    lib().so_pyramid_level_size(w, h, C.c_float(sub), C.byref(dw), C.byref(dh))
    return dw.value, dh.value


def resize(src, dw, dh):
    src = _u8(src)
    dst = np.empty((dh, dw), np.uint8)
    rc = lib().so_resize_u8_linear(src.ctypes.data, src.shape[1], src.shape[0], src.strides[0],
                                   dst.ctypes.data, dw, dh, dw)
    if rc != 0:
        raise ValueError("so_resize_u8_linear failed")
    return dst


def pyramid(im, subs):
    """ImagePyramid(im, subs).images as a list of arrays."""
    im = _u8(im)
    h, w = im.shape
    out = []
    for i, s in enumerate(subs):
        if i:
            assert subs[i - 1] < s
        if np.float32(s) == 1:
            out.append(im.copy())
        else:
            dw, dh = pyramid_level_size(w, h, s)
            out.append(resize(im, dw, dh))
    return out


def match(im, patch, patch_mask, area, area_mask, stride, exact=False, want_score=False):
    """matchFeatureInArea.  area = (x, y, w, h).  Returns dict(x, y, response, dx, dy[, score])."""
    im = _u8(im)
    patch = _u8(patch)
    pm = _u8(patch_mask) if patch_mask is not None else None
    am = _u8(area_mask) if area_mask is not None else None
    if pm is not None:
        assert pm.shape == patch.shape
    if am is not None:
        assert am.shape == (area[3], area[2])
    out = MatchOut()
    score = np.full((max(area[3], 1), max(area[2], 1)), np.nan, np.float32) if want_score else None
    rc = lib().so_match_feature_in_area(
        im.ctypes.data, im.shape[1], im.shape[0], im.strides[0],
        patch.ctypes.data, patch.shape[1], patch.shape[0], patch.strides[0],
        pm.ctypes.data if pm is not None else None, pm.strides[0] if pm is not None else 0,
        area[0], area[1], area[2], area[3],
        am.ctypes.data if am is not None else None, am.strides[0] if am is not None else 0,
        stride, int(bool(exact)), score.ctypes.data if want_score else None, C.byref(out))
    if rc != 0:
        raise ValueError("so_match_feature_in_area: bad arguments")
    r = dict(x=out.x, y=out.y, response=out.response, dx=out.dx, dy=out.dy,
             ax=out.ax, ay=out.ay, aw=out.aw, ah=out.ah)
    if want_score:
        r["score"] = score.reshape(-1)[:out.aw * out.ah].reshape(out.ah, out.aw).copy()
    return r


def track(levels, subs, proj, proj_mask, px, py, threshold, stride, coarse_area=None, coarse_mask=None):
    """Feature::track after project(): levels = pyramid images, proj/proj_mask full-res template.
    coarse_area=(x, y, w, h) [+ coarse_mask]: the build's extension, see so_track_ex."""
    levels = [_u8(l) for l in levels]
    n = len(levels)
    ptrs = (C.c_void_p * n)(*[l.ctypes.data for l in levels])
    lw = (C.c_int * n)(*[l.shape[1] for l in levels])
    lh = (C.c_int * n)(*[l.shape[0] for l in levels])
    fs = (C.c_float * n)(*[float(s) for s in subs])
    proj = _u8(proj)
    pm = _u8(proj_mask)
    assert pm.shape == proj.shape
    out = TrackOut()
    ca = (C.c_int * 4)(*coarse_area) if coarse_area is not None else None
    cm = _u8(coarse_mask) if coarse_mask is not None else None
    if cm is not None:
        assert cm.shape == (coarse_area[3], coarse_area[2])
    rc = lib().so_track_ex(ptrs, lw, lh, fs, n, proj.ctypes.data, pm.ctypes.data,
                           proj.shape[1], proj.shape[0], C.c_float(px), C.c_float(py),
                           C.c_float(threshold), stride, ca, cm.ctypes.data if cm is not None else None, C.byref(out))
    if rc != 0:
        raise ValueError("so_track: bad arguments")
    return dict(x=out.x, y=out.y, response=out.response, dx=out.dx, dy=out.dy,
                levels_run=out.levels_run)


def fast_score_map(im):
    im = _u8(im)
    out = np.zeros(im.shape, np.int16)
    lib().so_fast_score_map(im.ctypes.data, im.shape[1], im.shape[0], im.strides[0], out.ctypes.data)
    return out


def _kps(buf, n):
    a = np.frombuffer(buf, dtype=np.float32, count=3 * n).reshape(n, 3).copy()
    return a  # columns x, y, response


def fast(im, threshold, nonmax=True):
    im = _u8(im)
    cap = im.size
    buf = (KeyPoint * max(cap, 1))()
    n = lib().so_fast(im.ctypes.data, im.shape[1], im.shape[0], im.strides[0], int(threshold),
                      int(bool(nonmax)), buf, cap)
    if n < 0:
        raise MemoryError
    return _kps(buf, min(n, cap))


def get_sorted_keypoints(im, n_min_pts, fast_threshold):
    """Returns (keypoints[n,3] sorted, new fastThreshold)."""
    im = _u8(im)
    cap = im.size
    buf = (KeyPoint * max(cap, 1))()
    f = C.c_float(fast_threshold)
    n = lib().so_get_sorted_keypoints(im.ctypes.data, im.shape[1], im.shape[0], im.strides[0],
                                      n_min_pts, C.byref(f), buf, cap)
    if n < 0:
        raise MemoryError
    return _kps(buf, min(n, cap)), f.value


def sort_keypoints(kps):
    """kps[n,3] (x, y, response) -> sorted by response desc, y asc, x asc (new_point.cpp:124)."""
    a = np.ascontiguousarray(kps, dtype=np.float32).copy()
    lib().so_sort_keypoints(a.ctypes.data, len(a))
    return a


def select_new_points(kps, occupied, n, min_dist, rx, ry, w, h):
    """new_point.cpp:137-155 on sorted kps[n,3]; occupied[m,2] = positions of matches and lineMatches.
    Returns the indices of the kept keypoints."""
    a = np.ascontiguousarray(kps, dtype=np.float32)
    occ = np.ascontiguousarray(occupied, dtype=np.float32).reshape(-1, 2)
    sel = np.zeros(max(n, 1), np.int32)
    k = lib().so_select_new_points(a.ctypes.data, len(a), occ.ctypes.data, len(occ), n, min_dist, rx, ry, w, h, sel.ctypes.data)
    return sel[:k].copy()


def extract_descriptor(im, px, py, rx, ry):
    """feature.cpp:42-45: returns (crop array, (x0, y0, w, h))."""
    im = _u8(im)
    out = np.zeros((2 * ry + 2) * (2 * rx + 2), np.uint8)
    r = [C.c_int() for _ in range(4)]
    lib().so_extract_descriptor(im.ctypes.data, im.shape[1], im.shape[0], im.strides[0], px, py, rx, ry, out.ctypes.data,
                                *[C.byref(v) for v in r])
    x0, y0, w, h = [v.value for v in r]
    return out[:w * h].reshape(h, w).copy(), (x0, y0, w, h)


def project_setup(P, B, p3d, dw, dh):
    """feature.cpp:61-94.  P 3x4, B 4x3, p3d 3 (float32).  Returns dict(loc=(x, y, w, h), m=3x3 float32, valid)."""
    P = np.ascontiguousarray(P, np.float32).reshape(3, 4)
    B = np.ascontiguousarray(B, np.float32).reshape(4, 3)
    p = np.ascontiguousarray(p3d, np.float32).reshape(3)
    out = Projection()
    lib().so_project_setup(P.ctypes.data, B.ctypes.data, p.ctypes.data, dw, dh, C.byref(out))
    return dict(loc=(out.x, out.y, out.w, out.h), m=np.array(out.m, np.float32).reshape(3, 3), valid=bool(out.valid))


def warp_perspective(src, M, dw, dh, nearest=False):
    """cv::warpPerspective(src, dst, M, Size(dw, dh), WARP_INVERSE_MAP | INTER_LINEAR or INTER_NEAREST)."""
    src = _u8(src)
    M = np.ascontiguousarray(M, np.float32).reshape(9)
    dst = np.empty((dh, dw), np.uint8)
    rc = lib().so_warp_perspective_u8(src.ctypes.data, src.shape[1], src.shape[0], src.strides[0], M.ctypes.data,
                                      int(bool(nearest)), dst.ctypes.data, dw, dh, dw)
    if rc != 0:
        raise ValueError("so_warp_perspective_u8: bad arguments")
    return dst


def project(descriptor, P, B, p3d):
    """Feature::project (feature.cpp:53-100): returns (proj, mask, location) - empty arrays when the projection is empty."""
    d = _u8(descriptor)
    s = project_setup(P, B, p3d, d.shape[1], d.shape[0])
    if not s["valid"]:
        return np.zeros((0, 0), np.uint8), np.zeros((0, 0), np.uint8), s["loc"]
    _, _, w, h = s["loc"]
    return warp_perspective(d, s["m"], w, h, False), warp_perspective(np.ones_like(d), s["m"], w, h, True), s["loc"]


def disc_mask(w, h, circles):
    """lineFeature.cpp:86-96: zero mask with one filled cv::circle (value 255) per (x, y, radius)."""
    c = np.ascontiguousarray(circles, np.int32).reshape(-1, 3)
    out = np.empty((h, w), np.uint8)
    if lib().so_disc_mask(w, h, c.ctypes.data, len(c), out.ctypes.data) != 0:
        raise ValueError("so_disc_mask: bad arguments")
    return out


def ellipse_window(px, py, cov, gate, res):
    """Gate ellipse of a 2x2 covariance (sxx, sxy, syy) scaled by res: dict(area=(x, y, w, h), mask, q, centre) or
    None when there is no window (degenerate covariance, beyond the 300-pixel cap)."""
    c = np.ascontiguousarray(cov, np.float32).reshape(3)
    e = Ellipse()
    if lib().so_ellipse_setup(C.c_float(px), C.c_float(py), c.ctypes.data, C.c_float(gate), C.c_float(res), C.byref(e)) != 0:
        raise ValueError("so_ellipse_setup: bad arguments")
    if not e.valid:
        return None
    m = np.empty((e.h, e.w), np.uint8)
    lib().so_ellipse_mask(C.byref(e), m.ctypes.data)
    return dict(area=(e.x, e.y, e.w, e.h), mask=m, q=(e.q11, e.q12, e.q22), centre=(e.ux, e.uy))


def line_track(levels, subs, descriptor, area, area_mask, threshold, stride):
    """The searches of LineFeature::track (lineFeature.cpp:120-165); area = (x, y, w, h) on the coarsest level."""
    levels = [_u8(l) for l in levels]
    n = len(levels)
    ptrs = (C.c_void_p * n)(*[l.ctypes.data for l in levels])
    lw = (C.c_int * n)(*[l.shape[1] for l in levels])
    lh = (C.c_int * n)(*[l.shape[0] for l in levels])
    fs = (C.c_float * n)(*[float(s) for s in subs])
    d = _u8(descriptor)
    am = _u8(area_mask)
    assert am.shape == (area[3], area[2])
    out = TrackOut()
    rc = lib().so_line_track(ptrs, lw, lh, fs, n, d.ctypes.data, d.shape[1], d.shape[0], area[0], area[1], area[2], area[3],
                             am.ctypes.data, C.c_float(threshold), stride, C.byref(out))
    if rc != 0:
        raise ValueError("so_line_track: bad arguments")
    return dict(x=out.x, y=out.y, response=out.response, dx=out.dx, dy=out.dy, levels_run=out.levels_run)


class PropTest:
    """Iterator over the trials of tests/test_matching.cpp (same glibc rand() stream)."""

    def __init__(self, use_area_mask, use_patch_mask, seed=1):
        self.uam, self.upm = int(use_area_mask), int(use_patch_mask)
        lib().so_proptest_srand(seed)

    def next(self):
        c = PropCase()
        lib().so_proptest_next(self.uam, self.upm, C.byref(c))

        def arr(p, h, w):
            if not p:
                return None
            return np.ctypeslib.as_array(p, shape=(h, w)).copy()
        case = dict(w=c.w, h=c.h, pw=c.pw, ph=c.ph, xpatch=c.xpatch, ypatch=c.ypatch,
                    area=(c.ax, c.ay, c.aw, c.ah),
                    im=arr(c.im, c.h, c.w), patch=arr(c.patch, c.ph, c.pw),
                    patch_mask=arr(c.patch_mask, c.ph, c.pw),
                    area_mask=arr(c.area_mask, c.ah, c.aw))
        lib().so_proptest_free(C.byref(c))
        return case

    @staticmethod
    def accept(case, mx, my, response):
        c = PropCase()
        c.w, c.h, c.pw, c.ph = case["w"], case["h"], case["pw"], case["ph"]
        c.xpatch, c.ypatch = case["xpatch"], case["ypatch"]
        keep = [np.ascontiguousarray(case["im"]), np.ascontiguousarray(case["patch"])]
        c.im = keep[0].ctypes.data_as(_u8p)
        c.patch = keep[1].ctypes.data_as(_u8p)
        if case["patch_mask"] is not None:
            keep.append(np.ascontiguousarray(case["patch_mask"]))
            c.patch_mask = keep[-1].ctypes.data_as(_u8p)
        return bool(lib().so_proptest_accept(C.byref(c), int(mx), int(my), C.c_float(response)))
