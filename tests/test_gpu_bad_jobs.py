"""Jobs whose fields cannot be used (offsets outside the blob, frames or levels that do not exist, strides below 1,
non-finite predictions, sizes beyond the limits) are rejected one by one on the device with status SLAM_HIP_EINVAL and
never executed; the good jobs of the same batch are unaffected."""
import ctypes as C

import numpy as np
import pytest

from slam_amd import api, capi, synth

pytestmark = pytest.mark.gpu


def _same_float(a, b):
    return np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32) or (np.isnan(a) and np.isnan(b))


def test_bad_match_and_track_jobs_are_rejected(ctx, oracle):
    rng = np.random.default_rng(9)
    w, h, subs = 160, 120, [1.0, 2.0]
    im = synth.frame(w, h, 3)
    patch = synth.cut_template(im, 80, 60, 9, 9)
    mask = np.ones_like(patch)
    blob = np.concatenate([patch.reshape(-1), mask.reshape(-1)])
    L = capi.lib()
    pyr = api.ImagePyramid(ctx, im, subs)
    try:
        def mjob(**kw):
            j = capi.MatchJob()
            j.frame, j.level, j.pw, j.ph, j.patch_off, j.pmask_off = 0, 0, 9, 9, 0, 81
            j.ax, j.ay, j.aw, j.ah, j.amask_off, j.stride, j.use_exact_area_mask, j.score_off = 70, 50, 20, 20, -1, 1, 0, -1
            for k, v in kw.items():
                setattr(j, k, v)
            return j
        bad = [dict(frame=1), dict(frame=-1), dict(level=2), dict(level=-1), dict(pw=0), dict(ph=-3), dict(patch_off=100),
               dict(patch_off=-5), dict(pmask_off=82), dict(stride=0), dict(stride=-2), dict(stride=(1 << 20) + 1), dict(stride=0x7fffffff), dict(amask_off=0, aw=20, ah=20),
               dict(pw=1 << 20), dict(aw=1 << 30, ah=1 << 30)]
        empty = [dict(aw=0), dict(ah=-1)]                # an empty area is answered like one that misses the image (matching.cpp:26-30)
        jobs = [mjob()] + [mjob(**b) for b in bad] + [mjob(**b) for b in empty] + [mjob(ax=71)]
        arr = (capi.MatchJob * len(jobs))(*jobs)
        res = (capi.MatchResult * len(jobs))()
        capi.check(ctx.handle, L.slam_hip_match_batch(ctx.handle, pyr.handle, arr, len(jobs), blob.ctypes.data, blob.size, res))
        ref = oracle.match(im, patch, mask, (70, 50, 20, 20), None, 1)
        assert res[0].status == 0 and (res[0].x, res[0].y) == (ref["x"], ref["y"]) == (80, 60) and _same_float(res[0].response, ref["response"])
        assert res[len(jobs) - 1].status == 0 and (res[len(jobs) - 1].x, res[len(jobs) - 1].y) == (80, 60)
        for i, b in enumerate(bad):
            assert res[1 + i].status == capi.EINVAL, b
        for i, b in enumerate(empty):
            e = res[1 + len(bad) + i]
            assert (e.status, e.x, e.y, e.response) == (0, 0, 0, -1.0), b

        def tjob(**kw):
            j = capi.TrackJob()
            j.frame, j.projw, j.projh, j.flags, j.proj_off, j.mask_off = 0, 9, 9, 0, 0, 81
            j.px, j.py, j.threshold, j.stride = 81.0, 59.0, 0.9, 2
            for k, v in kw.items():
                setattr(j, k, v)
            return j
        tbad = [dict(frame=3), dict(frame=-1), dict(proj_off=90), dict(mask_off=-2), dict(stride=0), dict(px=float("nan")),
                dict(py=float("inf")), dict(threshold=float("nan")), dict(projw=1 << 16),
                dict(flags=capi.TRACK_COARSE_AREA, aw=0, ah=5), dict(flags=capi.TRACK_COARSE_AREA, aw=8, ah=8, amask_off=150),
                dict(flags=capi.TRACK_COARSE_AREA, aw=1 << 20, ah=4, amask_off=-1)]
        tjobs = [tjob()] + [tjob(**b) for b in tbad] + [tjob(projw=0)]
        tarr = (capi.TrackJob * len(tjobs))(*tjobs)
        tres = (capi.TrackResult * len(tjobs))()
        capi.check(ctx.handle, L.slam_hip_track_batch(ctx.handle, pyr.handle, tarr, len(tjobs), blob.ctypes.data, blob.size, tres))
        levels = oracle.pyramid(im, subs)
        r = oracle.track(levels, subs, patch, mask, 81.0, 59.0, 0.9, 2)
        assert tres[0].status == 0 and (tres[0].x, tres[0].y, tres[0].levels_run) == (r["x"], r["y"], r["levels_run"])
        for i, b in enumerate(tbad):
            assert tres[1 + i].status == capi.EINVAL and (tres[1 + i].x, tres[1 + i].y) == (-1, -1), b
        last = tres[len(tjobs) - 1]                      # empty projection: not an error (feature.cpp:112-115)
        assert last.status == 0 and (last.x, last.y, last.response) == (-1, -1, -1.0)
    finally:
        pyr.close()


def test_host_entry_points_reject_unusable_arguments(ctx):
    L = capi.lib()
    im = np.zeros((20, 20), np.uint8)
    p = np.ones((5, 5), np.uint8)
    out = capi.MatchResult()

    def call(**kw):
        a = dict(w=20, h=20, pitch=20, pw=5, ph=5, ppitch=5, stride=1)
        a.update(kw)
        return L.slam_hip_match_feature_in_area(ctx.handle, im.ctypes.data, a["w"], a["h"], a["pitch"], p.ctypes.data, a["pw"], a["ph"],
                                                a["ppitch"], None, 0, 2, 2, 8, 8, None, 0, a["stride"], 0, None, C.byref(out))
    assert call() == 0
    for bad in [dict(stride=0), dict(w=0), dict(h=-1), dict(pw=0), dict(pitch=10), dict(ppitch=3)]:
        assert call(**bad) == capi.EINVAL, bad
        assert len(L.slam_hip_last_error_string(ctx.handle)) > 0
    pyr = C.c_void_p()
    subs = (C.c_float * 2)(2.0, 1.0)                     # subsamples must increase
    assert L.slam_hip_pyramid_create(ctx.handle, 64, 48, 2, subs, 1, C.byref(pyr)) == capi.EINVAL
    subs = (C.c_float * 2)(1.0, 2.0)
    assert L.slam_hip_pyramid_create(ctx.handle, 0, 48, 2, subs, 1, C.byref(pyr)) == capi.EINVAL
    assert L.slam_hip_pyramid_create(ctx.handle, 64, 48, 2, subs, 1, C.byref(pyr)) == 0
    try:                                                 # searching a pyramid that was never built is refused
        j = capi.MatchJob()
        res = capi.MatchResult()
        assert L.slam_hip_match_batch(ctx.handle, pyr, C.byref(j), 1, p.ctypes.data, p.size, C.byref(res)) == capi.EINVAL
    finally:
        L.slam_hip_pyramid_destroy(ctx.handle, pyr)
    with api.Context(0) as own:                          # a context of its own: the shared one keeps the shipped LDS split
        assert L.slam_hip_set_match_lds(own.handle, 1000) == capi.EINVAL
        assert L.slam_hip_set_match_lds(own.handle, 65536) == capi.EINVAL      # with the kernels' static LDS it would pass 64 KB
        assert L.slam_hip_set_match_lds(own.handle, 10240) == 0
        assert L.slam_hip_set_walk_lds(own.handle, 100) == capi.EINVAL
        assert L.slam_hip_set_walk_lds(own.handle, 16384) == 0
