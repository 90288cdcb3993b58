"""The lean walk kernel's second build (slam_hip_set_walk_chunks) takes a level whose accumulators, entry lists and tile
do not fit the LDS budget at once in blocks of cell rows and chunks of template rows.  Budget and build only change how
the work is cut: every result equals the oracle's bit for bit, whatever the cut, and templates far beyond the plain
build's reach (81x81: the descriptors of new_image.cpp:80-82 under matchPoints' pyramid {1, 3}, matching.cpp:243-250)
stay on the lean kernel."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from slam_amd import api, synth  # noqa: E402
from conftest import DEFAULT_KERNEL_SPLIT, SOAK_SEED  # noqa: E402

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.float32(a).view(np.uint32)


def _same(g, r):
    return g["status"] == 0 and (g["x"], g["y"], g["levels_run"]) == (r["x"], r["y"], r["levels_run"]) and \
        _bits(g["response"]) == _bits(r["response"]) and _bits(g["dx"]) == _bits(r["dx"]) and _bits(g["dy"]) == _bits(r["dy"])


def _quad_mask(rng, ph, pw):
    """a warped-quadrilateral-like mask: two cut corners (at most two partly kept 4-pixel groups per row)"""
    mask = np.ones((ph, pw), np.uint8)
    k = int(rng.integers(0, min(ph, pw) // 3 + 1))
    for r in range(k):
        mask[r, :k - r] = 0
        mask[ph - 1 - r, pw - (k - r):] = 0
    return mask


def _interior_jobs(rng, f0, shift, shapes, per_shape, margin):
    h, w = f0.shape
    jobs = []
    for (pw, ph) in shapes:
        for _ in range(per_shape):
            cx = int(rng.integers(pw // 2 + margin, w - pw // 2 - margin))
            cy = int(rng.integers(ph // 2 + margin, h - ph // 2 - margin))
            proj = f0[cy - ph // 2:cy - ph // 2 + ph, cx - pw // 2:cx - pw // 2 + pw].copy()
            tx, ty = cx - shift[0], cy - shift[1]
            jobs.append(dict(frame=0, proj=proj, proj_mask=_quad_mask(rng, ph, pw),
                             px=tx + float(rng.integers(-5, 6)) + float(rng.random()) * 0.9,
                             py=ty + float(rng.integers(-5, 6)) + float(rng.random()) * 0.9,
                             threshold=float(rng.choice([0.3, 0.9, 0.98])), stride=3))
    return jobs


@pytest.mark.parametrize("lds", [4096, 5120, 6400, 7680, 10240, 11520, 15360, 24320, 40960, 65536])
@pytest.mark.parametrize("subs", [[1.0, 3.0], [1.0, 2.0, 4.0]])
def test_chunked_walks_equal_the_oracle_at_every_budget(oracle, lds, subs):
    rng = np.random.default_rng(lds + int(subs[1]) + SOAK_SEED)
    w, h, shift = 640, 480, (3, -4)
    f0, f1 = synth.frame(w, h, 31), synth.frame(w, h, 31, shift)
    shapes = [(21, 21), (31, 31), (45, 33), (33, 61), (81, 81), (64, 80), (13, 97), (101, 9)]
    jobs = _interior_jobs(rng, f0, shift, shapes, 3, 45)
    with api.Context(0) as c:
        L = api.capi.lib()
        api.capi.check(c.handle, L.slam_hip_set_walk_lds(c.handle, lds))
        api.capi.check(c.handle, L.slam_hip_set_walk_chunks(c.handle, 1))
        pyr = api.ImagePyramid(c, f1, subs)
        try:
            got = api.track_batch(c, pyr, jobs)
            left = api.debug_walk_deferred(c)
        finally:
            pyr.close()
    levels = oracle.pyramid(f1, subs)
    bad = []
    for k, (j, g) in enumerate(zip(jobs, got)):
        r = oracle.track(levels, subs, j["proj"], j["proj_mask"], j["px"], j["py"], j["threshold"], j["stride"])
        if not _same(g, r):
            bad.append((k, j["proj"].shape, g, r))
    assert not bad, bad[:4]
    if DEFAULT_KERNEL_SPLIT and lds >= 10240:
        assert left == 0, left                         # every walk is interior: none is left to the general kernel


@pytest.mark.parametrize("lds", [4096, 4608, 5248, 6016, 7040])
def test_fit_around_the_peak_of_a_search_in_blocks_and_chunks(oracle, lds):
    """Wide windows (stride 4 under pyramid {1, 3}: 32x32 cells at full resolution) under narrow, tall templates at small
    budgets: several blocks of cell rows and several chunks of template rows, and the sub-pixel fit of the last level
    re-sums the rows around the peak chunk by chunk, with slices picked anew for those few rows."""
    rng = np.random.default_rng(lds + 9 + SOAK_SEED)
    w, h, subs, shift = 480, 400, [1.0, 3.0], (2, -1)
    f0, f1 = synth.frame(w, h, 77), synth.frame(w, h, 77, shift)
    jobs = _interior_jobs(rng, f0, shift, [(19, 33), (17, 29), (23, 41), (9, 57)], 5, 60)
    for j in jobs:
        j["stride"] = 4
    with api.Context(0) as c:
        L = api.capi.lib()
        api.capi.check(c.handle, L.slam_hip_set_walk_lds(c.handle, lds))
        api.capi.check(c.handle, L.slam_hip_set_walk_chunks(c.handle, 1))
        pyr = api.ImagePyramid(c, f1, subs)
        try:
            got = api.track_batch(c, pyr, jobs)
        finally:
            pyr.close()
    levels = oracle.pyramid(f1, subs)
    bad = []
    for k, (j, g) in enumerate(zip(jobs, got)):
        r = oracle.track(levels, subs, j["proj"], j["proj_mask"], j["px"], j["py"], j["threshold"], j["stride"])
        if not _same(g, r):
            bad.append((k, j["proj"].shape, g, r))
    assert not bad, bad[:4]


@pytest.mark.parametrize("lds", [4096, 4352, 5120])
def test_coarse_levels_whose_template_is_taken_in_chunks(oracle, lds):
    """Under a four-level pyramid a 53x55 template is 26x27 at sub = 2: four lattice cells, which the kernel would sum
    straight from the image against the whole staged template - but at these budgets the template is staged in chunks,
    so the entry lists have to do it."""
    rng = np.random.default_rng(lds + 3 + SOAK_SEED)
    w, h, subs, shift = 480, 360, [1.0, 2.0, 4.0, 8.0], (-3, 2)
    f0, f1 = synth.frame(w, h, 78), synth.frame(w, h, 78, shift)
    jobs = _interior_jobs(rng, f0, shift, [(53, 55), (47, 61), (61, 45)], 6, 64)
    with api.Context(0) as c:
        L = api.capi.lib()
        api.capi.check(c.handle, L.slam_hip_set_walk_lds(c.handle, lds))
        api.capi.check(c.handle, L.slam_hip_set_walk_chunks(c.handle, 1))
        pyr = api.ImagePyramid(c, f1, subs)
        try:
            got = api.track_batch(c, pyr, jobs)
        finally:
            pyr.close()
    levels = oracle.pyramid(f1, subs)
    bad = []
    for k, (j, g) in enumerate(zip(jobs, got)):
        r = oracle.track(levels, subs, j["proj"], j["proj_mask"], j["px"], j["py"], j["threshold"], j["stride"])
        if not _same(g, r):
            bad.append((k, j["proj"].shape, g, r))
    assert not bad, bad[:4]


@pytest.mark.parametrize("bands", [1, 2, 3, 5, 8])
def test_small_batches_have_their_last_level_cut_into_bands(oracle, bands, monkeypatch):
    """A small batch under the second build (by batch size; forced here through SLAM_HIP_WALK_BANDS) searches the last
    level of every walk in bands of rows, one wavefront each, and fits around the first of the bands' maxima: the
    oracle's results at every band count - walks that end above the last level, walks over the image border (left to
    the general kernel), single-level pyramids and constant templates included."""
    monkeypatch.setenv("SLAM_HIP_WALK_BANDS", str(bands))
    rng = np.random.default_rng(31 + SOAK_SEED)
    w, h, shift = 640, 480, (3, -2)
    f0, f1 = synth.frame(w, h, 91 + SOAK_SEED), synth.frame(w, h, 91 + SOAK_SEED, shift)
    nbad = 0
    with api.Context(0) as c:
        L = api.capi.lib()
        api.capi.check(c.handle, L.slam_hip_set_walk_lds(c.handle, 12800))
        api.capi.check(c.handle, L.slam_hip_set_walk_chunks(c.handle, 1))
        for subs in ([1.0, 3.0], [1.0], [1.0, 2.0, 4.0]):
            jobs = _interior_jobs(rng, f0, shift, [(81, 81), (45, 33), (21, 21), (64, 80)], 3, 60)
            jobs.append(dict(jobs[0], threshold=1.5))                                     # ends on the coarsest level
            jobs.append(dict(jobs[1], px=3.0, py=h - 2.0))                                # windows over the border
            jobs.append(dict(jobs[2], proj=np.full_like(jobs[2]["proj"], 7)))             # every score equal: the first cell wins
            jobs.append(dict(jobs[3], proj_mask=(rng.random(jobs[3]["proj"].shape) < 0.5).astype(np.uint8)))   # a noise mask
            pyr = api.ImagePyramid(c, f1, subs)
            try:
                got = api.track_batch(c, pyr, jobs)
                left = api.debug_walk_deferred(c)
            finally:
                pyr.close()
            if DEFAULT_KERNEL_SPLIT:
                assert left <= 2, left                     # the walk over the border; the constant template's scores may all be NaN-free
            levels = oracle.pyramid(f1, subs)
            for k, (j, g) in enumerate(zip(jobs, got)):
                r = oracle.track(levels, subs, j["proj"], j["proj_mask"], j["px"], j["py"], j["threshold"], j["stride"])
                if not _same(g, r):
                    nbad += 1
                    print(bands, subs, k, j["proj"].shape, g, r)
    assert nbad == 0


def test_the_plain_build_leaves_large_templates_to_the_general_kernel_and_the_chunked_one_takes_them(oracle):
    rng = np.random.default_rng(5 + SOAK_SEED)
    w, h, subs, shift = 640, 480, [1.0, 3.0], (-2, 5)
    f0, f1 = synth.frame(w, h, 32), synth.frame(w, h, 32, shift)
    jobs = _interior_jobs(rng, f0, shift, [(81, 81)], 6, 50) + _interior_jobs(rng, f0, shift, [(21, 21)], 6, 50)
    levels = oracle.pyramid(f1, subs)
    want = [oracle.track(levels, subs, j["proj"], j["proj_mask"], j["px"], j["py"], j["threshold"], j["stride"]) for j in jobs]
    for chunks, expect_left in [(0, 6), (1, 0)]:
        with api.Context(0) as c:
            L = api.capi.lib()
            api.capi.check(c.handle, L.slam_hip_set_walk_lds(c.handle, 12800))
            api.capi.check(c.handle, L.slam_hip_set_walk_chunks(c.handle, chunks))
            pyr = api.ImagePyramid(c, f1, subs)
            try:
                got = api.track_batch(c, pyr, jobs)
                left = api.debug_walk_deferred(c)
            finally:
                pyr.close()
        assert all(_same(g, r) for g, r in zip(got, want))
        if DEFAULT_KERNEL_SPLIT:
            assert left == expect_left, (chunks, left)
    with api.Context(0) as c:                              # a host job table picks budget and build by itself
        pyr = api.ImagePyramid(c, f1, subs)
        try:
            got = api.track_batch(c, pyr, jobs)
            left = api.debug_walk_deferred(c)
        finally:
            pyr.close()
        assert all(_same(g, r) for g, r in zip(got, want))
        if DEFAULT_KERNEL_SPLIT:
            assert left == 0, left
        assert api.capi.lib().slam_hip_set_walk_chunks(c.handle, 2) == api.capi.EINVAL


def test_chunked_walks_of_random_shapes_against_the_oracle(oracle):
    """Random images, pyramids, template shapes up to 120 pixels a side, masks (quadrilaterals, noise, empty), strides,
    predictions on and beyond the border and thresholds that stop the walk at every level, at random budgets."""
    rng = np.random.default_rng(4242 + SOAK_SEED)
    nbad = 0
    for trial in range(10):
        w, h = int(rng.integers(160, 500)), int(rng.integers(140, 400))
        subs = [[1.0], [1.0, 2.0], [1.0, 3.0], [1.0, 1.5, 2.5], [1.0, 2.0, 4.0, 8.0], [2.0, 5.0]][trial % 6]
        f0 = synth.frame(w, h, 700 + trial)
        f1 = synth.frame(w, h, 700 + trial, (int(rng.integers(-4, 5)), int(rng.integers(-4, 5))))
        jobs = []
        for k in range(32):
            pw, ph = int(rng.integers(1, 121)), int(rng.integers(1, 121))
            pw, ph = min(pw, w // 2), min(ph, h // 2)
            cx, cy = int(rng.integers(pw // 2, w - (pw - pw // 2) + 1)), int(rng.integers(ph // 2, h - (ph - ph // 2) + 1))
            proj = f0[cy - ph // 2:cy - ph // 2 + ph, cx - pw // 2:cx - pw // 2 + pw].copy()
            kind = k % 5
            mask = (np.ones_like(proj) if kind == 0 else rng.integers(0, 2, proj.shape, dtype=np.uint8) if kind == 1
                    else np.zeros_like(proj) if kind == 2 else _quad_mask(rng, ph, pw))
            px = float(cx + rng.integers(-12, 13)) + float(rng.random())
            py = float(cy + rng.integers(-12, 13)) + float(rng.random())
            if k % 9 == 0:
                px, py = float(rng.choice([-30.0, -2.5, 0.0, w - 1.0, w + 25.0])), float(rng.choice([-30.0, 0.4, h - 0.6, h + 40.0]))
            jobs.append(dict(frame=0, proj=proj, proj_mask=mask, px=px, py=py,
                             threshold=float(rng.choice([-5.0, 0.3, 0.9, 0.98, 1.5])), stride=int(rng.integers(1, 6))))
        lds = int(rng.integers(64, 400)) * 64
        with api.Context(0) as c:
            L = api.capi.lib()
            api.capi.check(c.handle, L.slam_hip_set_walk_lds(c.handle, lds))
            api.capi.check(c.handle, L.slam_hip_set_walk_chunks(c.handle, 1))
            pyr = api.ImagePyramid(c, f1, subs)
            try:
                got = api.track_batch(c, pyr, jobs)
            finally:
                pyr.close()
        levels = oracle.pyramid(f1, subs)
        for j, g in zip(jobs, got):
            r = oracle.track(levels, subs, j["proj"], j["proj_mask"], j["px"], j["py"], j["threshold"], j["stride"])
            if not _same(g, r):
                nbad += 1
                print("trial", trial, subs, lds, j["proj"].shape, j["px"], j["py"], j["threshold"], j["stride"], g, r)
    assert nbad == 0
