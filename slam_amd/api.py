"""Python mirror of the reference's interface for the tracking hot path, on top of the C ABI.

Names follow the reference: ImagePyramid (imagePyramid.hpp:8-34), match_feature_in_area
(SLAM::matchFeatureInArea, SLAM.hpp:92-98), track (SLAM::Feature::track after project(),
feature.cpp:117-172), fast / get_sorted_keypoints (new_point.cpp:113-127).  Everything runs on the
GPU through libslam_hip.so; numpy is only used to hold host arrays.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import (EllipseJob, KeyPoint, LevelDesc, MatchJob, MatchResult, SlamHipError, TrackJob, TrackResult,
                   check, lib)


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim != 2:
        raise ValueError("expected a 2-D uint8 image")
    return a


class Context:
    """One GPU + one stream (slam_ctx)."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        rc = lib().slam_hip_create(device, C.byref(self._h))
        if rc != capi.OK:
            txt = lib().slam_hip_last_error_string(None)
            raise SlamHipError(rc, txt.decode() if txt else "")
        self.device = device
        if stream is not None:
            self.set_stream(stream)

    @property
    def handle(self):
        return self._h

    def set_stream(self, stream):
        check(self._h, lib().slam_hip_set_stream(self._h, C.c_void_p(stream)))

    def synchronize(self):
        check(self._h, lib().slam_hip_synchronize(self._h))

    def close(self):
        if self._h:
            lib().slam_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class DeviceBuffer:
    """Device memory through the ABI's own helpers (slam_hip_malloc / memcpy), for callers of the `_d` entry points
    that hold no HIP runtime of their own.  Created from a host array (copied) or as nbytes filled with `fill`."""

    def __init__(self, ctx, host=None, nbytes=None, fill=0):
        self.ctx = ctx
        if host is None:
            host = np.full(max(int(nbytes), 1), fill, np.uint8)
        h = np.ascontiguousarray(host)
        self.nbytes = int(h.nbytes)
        self.ptr = C.c_void_p()
        check(ctx.handle, lib().slam_hip_malloc(ctx.handle, max(self.nbytes, 16), C.byref(self.ptr)))
        check(ctx.handle, lib().slam_hip_memcpy_h2d(ctx.handle, self.ptr, h.ctypes.data, self.nbytes))

    def get(self, dtype=np.uint8):
        out = np.zeros(self.nbytes, np.uint8)
        check(self.ctx.handle, lib().slam_hip_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr, self.nbytes))
        return out.view(dtype)

    def free(self):
        if self.ptr:
            lib().slam_hip_free(self.ctx.handle, self.ptr)
            self.ptr = C.c_void_p()


class Blob:
    """Packs patches and masks (rows contiguous) and hands out their byte offsets."""

    def __init__(self):
        self._parts = []
        self._size = 0

    def add(self, arr):
        if arr is None:
            return -1
        a = _u8(arr)
        off = self._size
        self._parts.append(a.reshape(-1))
        self._size += a.size
        return off

    def bytes(self):
        if not self._parts:
            return np.zeros(1, np.uint8)
        return np.concatenate(self._parts)


class ImagePyramid:
    """ImagePyramid<byte>(imFullSize, subsamples) for one frame or a batch (frames[n, h, w])."""

    def __init__(self, ctx, frames, subsamples, device_ptr=None, pitch=None, frame_stride=None, shape=None):
        self.ctx = ctx
        self.subsamples = [float(np.float32(s)) for s in subsamples]
        subs = (C.c_float * len(subsamples))(*self.subsamples)
        self._h = C.c_void_p()
        if device_ptr is not None:
            n, h, w = shape
        else:
            frames = np.ascontiguousarray(frames, dtype=np.uint8)
            if frames.ndim == 2:
                frames = frames[None]
            n, h, w = frames.shape
        self.nframes, self.h, self.w = n, h, w
        check(ctx.handle, lib().slam_hip_pyramid_create(ctx.handle, w, h, len(subsamples), subs, n, C.byref(self._h)))
        try:
            if device_ptr is not None:
                check(ctx.handle, lib().slam_hip_pyramid_build_d(ctx.handle, self._h, C.c_void_p(device_ptr), pitch, frame_stride))
            else:
                self._frames = frames
                check(ctx.handle, lib().slam_hip_pyramid_build(ctx.handle, self._h, frames.ctypes.data, w, w * h))
        except Exception:
            self.close()
            raise

    def rebuild_d(self, device_ptr, pitch, frame_stride):
        check(self.ctx.handle, lib().slam_hip_pyramid_build_d(self.ctx.handle, self._h, C.c_void_p(device_ptr), pitch, frame_stride))

    @property
    def handle(self):
        return self._h

    def nSubs(self):
        return len(self.subsamples)

    def level(self, i):
        d = LevelDesc()
        check(self.ctx.handle, lib().slam_hip_pyramid_level(self._h, i, C.byref(d)))
        return d

    def image(self, level, frame=0):
        d = self.level(level)
        out = np.empty((d.h, d.w), np.uint8)
        check(self.ctx.handle, lib().slam_hip_pyramid_download(self.ctx.handle, self._h, level, frame, out.ctypes.data, d.w))
        return out

    @property
    def images(self):
        return [self.image(i) for i in range(self.nSubs())]

    def close(self):
        if self._h:
            lib().slam_hip_pyramid_destroy(self.ctx.handle, self._h)
            self._h = C.c_void_p()


def match_feature_in_area(ctx, im, patch, patch_mask, area, area_mask, stride, use_exact_area_mask=False,
                          want_score=False):
    """SLAM::matchFeatureInArea.  area = (x, y, w, h).  Returns a dict (x, y, response, dx, dy[, score])."""
    im, patch = _u8(im), _u8(patch)
    pm = _u8(patch_mask) if patch_mask is not None else None
    am = _u8(area_mask) if area_mask is not None else None
    if pm is not None and pm.shape != patch.shape:
        raise ValueError("patch mask shape")
    if am is not None and am.shape != (area[3], area[2]):
        raise ValueError("area mask shape")
    res = MatchResult()
    score = None
    if want_score:
        score = np.full(max(area[2], 0) * max(area[3], 0) + 1, np.nan, np.float32)
    check(ctx.handle, lib().slam_hip_match_feature_in_area(
        ctx.handle, im.ctypes.data, im.shape[1], im.shape[0], im.strides[0],
        patch.ctypes.data, patch.shape[1], patch.shape[0], patch.strides[0],
        pm.ctypes.data if pm is not None else None, pm.strides[0] if pm is not None else 0,
        area[0], area[1], area[2], area[3],
        am.ctypes.data if am is not None else None, am.strides[0] if am is not None else 0,
        stride, int(bool(use_exact_area_mask)), score.ctypes.data if want_score else None, C.byref(res)))
    if res.status != 0:
        raise SlamHipError(res.status, "job rejected on the device")
    out = dict(x=res.x, y=res.y, response=res.response, dx=res.dx, dy=res.dy)
    if want_score:
        h, w = im.shape
        ax, ay = max(area[0], 0), max(area[1], 0)
        aw = min(w - ax, area[2] + area[0] - ax)
        ah = min(h - ay, area[3] + area[1] - ay)
        if aw > 0 and ah > 0 and ax < w and ay < h:
            out["score"] = score[:aw * ah].reshape(ah, aw).copy()
        else:
            out["score"] = np.zeros((0, 0), np.float32)
    return out


def match_batch(ctx, pyr, jobs):
    """jobs: list of dicts(frame, level, patch, patch_mask, area, area_mask, stride, exact)."""
    n = len(jobs)
    blob = Blob()
    arr = (MatchJob * max(n, 1))()
    for i, j in enumerate(jobs):
        p = _u8(j["patch"])
        a = arr[i]
        a.frame, a.level = j.get("frame", 0), j.get("level", 0)
        a.ph, a.pw = p.shape
        a.patch_off = blob.add(p)
        a.pmask_off = blob.add(j.get("patch_mask"))
        a.ax, a.ay, a.aw, a.ah = j["area"]
        a.amask_off = blob.add(j.get("area_mask"))
        a.stride = j.get("stride", 1)
        a.use_exact_area_mask = int(bool(j.get("exact", False)))
        a.score_off = -1
    b = blob.bytes()
    res = (MatchResult * max(n, 1))()
    check(ctx.handle, lib().slam_hip_match_batch(ctx.handle, pyr.handle, arr, n, b.ctypes.data, b.size, res))
    return [dict(x=r.x, y=r.y, response=r.response, dx=r.dx, dy=r.dy, status=r.status) for r in res[:n]]


def track_batch(ctx, pyr, jobs):
    """jobs: list of dicts(frame, proj, proj_mask, px, py, threshold, stride) -> Feature::track results.
    Optional per job: coarse_area=(x, y, w, h) and coarse_mask (or None): the coarsest level is searched in that
    window under that area mask instead of the reference's fixed square (SLAM_HIP_TRACK_COARSE_AREA); or
    window=ellipse_setup(...) on every job: the masks are rasterised on the device (slam_hip_track_ellipse_batch)."""
    n = len(jobs)
    blob = Blob()
    arr = (TrackJob * max(n, 1))()
    win = (EllipseJob * max(n, 1))()
    nwin = 0
    for i, j in enumerate(jobs):
        a = arr[i]
        a.frame = j.get("frame", 0)
        if j["proj"] is None or np.size(j["proj"]) == 0:
            a.projw = a.projh = 0
            a.proj_off = a.mask_off = 0
        else:
            p = _u8(j["proj"])
            m = _u8(j["proj_mask"])
            if m.shape != p.shape:
                raise ValueError("proj mask shape")
            a.projh, a.projw = p.shape
            a.proj_off = blob.add(p)
            a.mask_off = blob.add(m)
        a.px, a.py = float(j["px"]), float(j["py"])
        a.threshold = float(j["threshold"])
        a.stride = int(j["stride"])
        if j.get("coarse_area") is not None:
            a.flags = capi.TRACK_COARSE_AREA
            a.ax, a.ay, a.aw, a.ah = j["coarse_area"]
            cm = j.get("coarse_mask")
            if cm is not None and _u8(cm).shape != (a.ah, a.aw):
                raise ValueError("coarse mask shape")
            a.amask_off = blob.add(cm)
        if j.get("window") is not None:
            win[i] = j["window"]
            nwin += 1
    b = blob.bytes()
    res = (TrackResult * max(n, 1))()
    if nwin:
        if nwin != n:
            raise ValueError("either every job carries an uncertainty window or none does")
        check(ctx.handle, lib().slam_hip_track_ellipse_batch(ctx.handle, pyr.handle, arr, win, n, b.ctypes.data, b.size, res))
    else:
        check(ctx.handle, lib().slam_hip_track_batch(ctx.handle, pyr.handle, arr, n, b.ctypes.data, b.size, res))
    return [dict(x=r.x, y=r.y, response=r.response, dx=r.dx, dy=r.dy, levels_run=r.levels_run, status=r.status)
            for r in res[:n]]


def ellipse_setup(px, py, cov, gate, res):
    """Uncertainty window of a prediction (px, py) with covariance cov = (sxx, sxy, syy) in full-resolution pixels^2:
    the gate ellipse scaled by res = 1 / subsample of the level searched.  Returns an EllipseJob (check .valid);
    pass it as a track job's "window"."""
    c = np.ascontiguousarray(cov, np.float32).reshape(3)
    e = EllipseJob()
    rc = lib().slam_hip_ellipse_setup(C.c_float(px), C.c_float(py), c.ctypes.data, C.c_float(gate), C.c_float(res), C.byref(e))
    if rc != 0:
        raise RuntimeError("slam_hip_ellipse_setup: %d" % rc)
    return e


def ellipse_masks(ctx, windows):
    """Rasterises the area masks of valid windows on the device; returns a list of arrays (None for invalid ones)."""
    n = len(windows)
    arr = (EllipseJob * max(n, 1))()
    off = 0
    for i, w in enumerate(windows):
        arr[i] = w
        arr[i].mask_off = off
        off += w.w * w.h if w.valid else 0
    d_jobs = DeviceBuffer(ctx, host=np.frombuffer(bytes(arr), np.uint8))
    d_blob = DeviceBuffer(ctx, nbytes=max(off, 1))
    d_status = DeviceBuffer(ctx, nbytes=4 * max(n, 1))
    try:
        check(ctx.handle, lib().slam_hip_ellipse_masks_d(ctx.handle, d_jobs.ptr, n, d_blob.ptr, max(off, 1), d_status.ptr, 0))
        ctx.synchronize()
        blob = d_blob.get()
        st = d_status.get(np.int32)
    finally:
        d_jobs.free(); d_blob.free(); d_status.free()
    if n and st[:n].any():
        raise RuntimeError("ellipse_masks: a job was rejected by the device")
    return [blob[a.mask_off:a.mask_off + a.w * a.h].reshape(a.h, a.w).copy() if a.valid else None for a in arr[:n]]


def fast(ctx, frames, threshold, nonmax=True, cap=None):
    """cv::FAST(im, kps, threshold, nonmax) per frame -> list of float32 arrays [n,3] (x, y, response)."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    if frames.ndim == 2:
        frames = frames[None]
    n, h, w = frames.shape
    if cap is None:
        cap = ((w + 1) // 2) * ((h + 1) // 2) if nonmax else w * h
    cap = max(int(cap), 1)
    out = np.zeros((n, cap, 3), np.float32)
    counts = np.zeros(n, np.int32)
    rc = lib().slam_hip_fast_detect(ctx.handle, frames.ctypes.data, w, h, w, w * h, n, int(threshold),
                                    int(bool(nonmax)), out.ctypes.data, cap, counts.ctypes.data)
    check(ctx.handle, rc, allow=(capi.ECAP,))
    res = [out[i, :min(counts[i], cap)].copy() for i in range(n)]
    return res, counts, rc


def get_sorted_keypoints(ctx, im, n_min_pts, fast_threshold, cap=None):
    """SLAM::getSortedKeyPoints(im, nMinPts, out); returns (keypoints[n,3], new fastThreshold)."""
    im = _u8(im)
    h, w = im.shape
    if cap is None:
        cap = ((w + 1) // 2) * ((h + 1) // 2)
    out = np.zeros((max(cap, 1), 3), np.float32)
    f = C.c_float(fast_threshold)
    n = C.c_int32(0)
    check(ctx.handle, lib().slam_hip_get_sorted_keypoints(ctx.handle, im.ctypes.data, w, h, im.strides[0],
                                                          int(n_min_pts), C.byref(f), out.ctypes.data, cap, C.byref(n)))
    return out[:min(n.value, cap)].copy(), f.value


def add_new_points(ctx, im, kps, occupied, n_new, min_dist, rx, ry, already_sorted=False, want_descriptors=True):
    """SLAM::addNewLines' selection (new_point.cpp:137-155) for one frame: kps[n,3] in cv::FAST order
    (or sorted), occupied[m,2].  Returns dict(sel, sorted, descriptors, rects)."""
    im = _u8(im)
    k = np.ascontiguousarray(kps, dtype=np.float32).reshape(-1, 3)
    occ = np.ascontiguousarray(occupied, dtype=np.float32).reshape(-1, 2)
    sel = np.zeros(n_new, np.int32)
    nsel = C.c_int32(0)
    sorted_out = np.zeros_like(k)
    one = (2 * rx + 1) * (2 * ry + 1)
    desc = np.zeros(one * n_new, np.uint8)
    res = (capi.DescResult * n_new)()
    check(ctx.handle, lib().slam_hip_add_new_points(
        ctx.handle, im.ctypes.data, im.shape[1], im.shape[0], im.strides[0], k.ctypes.data, len(k), int(already_sorted),
        occ.ctypes.data if len(occ) else None, len(occ), n_new, min_dist, rx, ry, sel.ctypes.data, C.byref(nsel),
        sorted_out.ctypes.data, desc.ctypes.data if want_descriptors else None, desc.size, res if want_descriptors else None))
    n = nsel.value
    out = dict(sel=sel[:n].copy(), sorted=sorted_out)
    if want_descriptors:
        out["rects"] = [(res[i].x0, res[i].y0, res[i].w, res[i].h) for i in range(n)]
        out["descriptors"] = [desc[i * one:i * one + res[i].w * res[i].h].reshape(res[i].h, res[i].w).copy() for i in range(n)]
    return out


def project_setup(P, B, p3d, dw, dh):
    """feature.cpp:61-94 (host arithmetic of the library): returns dict(loc=(x, y, w, h), m=3x3 float32, valid)."""
    P = np.ascontiguousarray(P, np.float32).reshape(3, 4)
    B = np.ascontiguousarray(B, np.float32).reshape(4, 3)
    p = np.ascontiguousarray(p3d, np.float32).reshape(3)
    loc = np.zeros(4, np.int32)
    m = np.zeros(9, np.float32)
    valid = C.c_int32(0)
    rc = lib().slam_hip_project_setup(P.ctypes.data, B.ctypes.data, p.ctypes.data, dw, dh, loc.ctypes.data, m.ctypes.data,
                                      C.byref(valid))
    if rc != capi.OK:
        raise SlamHipError(rc, "project_setup")
    return dict(loc=tuple(int(v) for v in loc), m=m.reshape(3, 3), valid=bool(valid.value))


def project(ctx, descriptor, P, B, p3d):
    """SLAM::Feature::project (feature.cpp:53-100): returns (proj, mask, location); empty arrays for an empty projection."""
    d = _u8(descriptor)
    P = np.ascontiguousarray(P, np.float32).reshape(3, 4)
    B = np.ascontiguousarray(B, np.float32).reshape(4, 3)
    p = np.ascontiguousarray(p3d, np.float32).reshape(3)
    cap = 500 * 500
    proj = np.zeros(cap, np.uint8)
    mask = np.zeros(cap, np.uint8)
    loc = np.zeros(4, np.int32)
    valid = C.c_int32(0)
    check(ctx.handle, lib().slam_hip_project(ctx.handle, d.ctypes.data, d.shape[1], d.shape[0], d.strides[0], P.ctypes.data,
                                             B.ctypes.data, p.ctypes.data, proj.ctypes.data, mask.ctypes.data, cap,
                                             loc.ctypes.data, C.byref(valid)))
    loc = tuple(int(v) for v in loc)
    if not valid.value:
        return np.zeros((0, 0), np.uint8), np.zeros((0, 0), np.uint8), loc
    n = loc[2] * loc[3]
    return proj[:n].reshape(loc[3], loc[2]).copy(), mask[:n].reshape(loc[3], loc[2]).copy(), loc


def project_track_batch(ctx, pyr, jobs):
    """SLAM::Feature::track as the reference runs it (project, then the level walk; feature.cpp:102-173).
    jobs: list of dicts(frame, descriptor, P, B, p3d, px, py, threshold, stride); the projected templates stay on
    the device."""
    n = len(jobs)
    blob = Blob()
    wj = (capi.WarpJob * max(n, 1))()
    tj = (TrackJob * max(n, 1))()
    for i, j in enumerate(jobs):
        d = _u8(j["descriptor"])
        s = project_setup(j["P"], j["B"], j["p3d"], d.shape[1], d.shape[0])
        w, t = wj[i], tj[i]
        w.desc_off = blob.add(d)
        w.dh, w.dw = d.shape
        w.ow, w.oh = (s["loc"][2], s["loc"][3]) if s["valid"] else (0, 0)
        for k in range(9):
            w.m[k] = float(s["m"].reshape(9)[k])
        t.frame = j.get("frame", 0)
        t.px, t.py = float(j["px"]), float(j["py"])
        t.threshold = float(j["threshold"])
        t.stride = int(j["stride"])
    b = blob.bytes()
    res = (TrackResult * max(n, 1))()
    check(ctx.handle, lib().slam_hip_project_track_batch(ctx.handle, pyr.handle, wj, tj, n, b.ctypes.data, b.size, res))
    return [dict(x=r.x, y=r.y, response=r.response, dx=r.dx, dy=r.dy, levels_run=r.levels_run, status=r.status)
            for r in res[:n]]


def line_track_batch(ctx, pyr, jobs):
    """The searches of SLAM::LineFeature::track (lineFeature.cpp:120-165).  jobs: list of dicts(frame, descriptor,
    area=(x, y, w, h) on the coarsest level, area_mask, threshold, stride)."""
    n = len(jobs)
    blob = Blob()
    arr = (capi.LineJob * max(n, 1))()
    for i, j in enumerate(jobs):
        d, am = _u8(j["descriptor"]), _u8(j["area_mask"])
        a = arr[i]
        a.frame = j.get("frame", 0)
        a.dh, a.dw = d.shape
        a.ax, a.ay, a.aw, a.ah = j["area"]
        if am.shape != (a.ah, a.aw):
            raise ValueError("area mask shape")
        a.desc_off = blob.add(d)
        a.amask_off = blob.add(am)
        a.threshold = float(j["threshold"])
        a.stride = int(j["stride"])
    b = blob.bytes()
    res = (TrackResult * max(n, 1))()
    check(ctx.handle, lib().slam_hip_line_track_batch(ctx.handle, pyr.handle, arr, n, b.ctypes.data, b.size, res))
    return [dict(x=r.x, y=r.y, response=r.response, dx=r.dx, dy=r.dy, levels_run=r.levels_run, status=r.status)
            for r in res[:n]]


def match_lines_batch(ctx, pyr, jobs):
    """SLAM::matchLines' image work (matching.cpp:279-292): like line_track_batch, but each job carries `circles`
    ([k,3] x, y, radius relative to its area) instead of an area mask; the masks are drawn on the device."""
    n = len(jobs)
    blob = Blob()
    arr = (capi.LineJob * max(n, 1))()
    begin = np.zeros(n + 1, np.int32)
    circ = []
    for i, j in enumerate(jobs):
        d = _u8(j["descriptor"])
        a = arr[i]
        a.frame = j.get("frame", 0)
        a.dh, a.dw = d.shape
        a.ax, a.ay, a.aw, a.ah = j["area"]
        a.desc_off = blob.add(d)
        a.amask_off = -1
        a.threshold = float(j["threshold"])
        a.stride = int(j["stride"])
        c = np.ascontiguousarray(j["circles"], np.int32).reshape(-1, 3)
        circ.append(c)
        begin[i + 1] = begin[i] + len(c)
    allc = np.concatenate(circ) if circ else np.zeros((0, 3), np.int32)
    b = blob.bytes()
    res = (TrackResult * max(n, 1))()
    check(ctx.handle, lib().slam_hip_match_lines_batch(ctx.handle, pyr.handle, arr, n, allc.ctypes.data if len(allc) else None,
                                                       len(allc), begin.ctypes.data, b.ctypes.data, b.size, res))
    return [dict(x=r.x, y=r.y, response=r.response, dx=r.dx, dy=r.dy, levels_run=r.levels_run, status=r.status)
            for r in res[:n]]


def debug_walk_deferred(ctx):
    """Jobs the last lean kernel launch of this context left to the general kernel (walks or single searches)."""
    n = C.c_int32(0)
    check(ctx.handle, lib().slam_hip_debug_walk_deferred(ctx.handle, C.byref(n)))
    return n.value


def debug_score_formulas(ctx, cc, si, sp):
    cc = np.ascontiguousarray(cc, np.uint64)
    si = np.ascontiguousarray(si, np.uint64)
    sp = np.ascontiguousarray(sp, np.uint64)
    n = cc.size
    a = np.zeros(n, np.float32)
    b = np.zeros(n, np.float32)
    check(ctx.handle, lib().slam_hip_debug_score_formulas(ctx.handle, cc.ctypes.data, si.ctypes.data, sp.ctypes.data,
                                                          n, a.ctypes.data, b.ctypes.data))
    return a, b


def debug_resize(ctx, src, dw, dh):
    src = _u8(src)
    dst = np.zeros((dh, dw), np.uint8)
    check(ctx.handle, lib().slam_hip_debug_resize(ctx.handle, src.ctypes.data, src.shape[1], src.shape[0], src.strides[0],
                                                  dst.ctypes.data, dw, dh, dw))
    return dst
