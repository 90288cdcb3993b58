"""GPU parity of the image work of SLAM::LineFeature::track (lineFeature.cpp:86-165): the disc area mask against
the oracle's restatement of cv::circle, and the searches (coarsest level inside the area mask, finer levels in
small windows) against the oracle's walk.  Parity unpinned: no test of the reference covers LineFeature."""
import ctypes as C

import numpy as np
import pytest

from conftest import SOAK_SEED

from slam_amd import api, capi, synth

DevBuf = api.DeviceBuffer

pytestmark = pytest.mark.gpu


def _disc_masks(ctx, jobs, hint=0):
    """jobs: list of (w, h, circles[k,3]).  One slam_hip_disc_masks_d call; returns [(mask, status)]."""
    n = len(jobs)
    arr = (capi.DiscJob * n)()
    circ, off = [], 0
    for i, (w, h, cs) in enumerate(jobs):
        cs = np.asarray(cs, np.int32).reshape(-1, 3)
        arr[i].mask_off, arr[i].w, arr[i].h = off, w, h
        arr[i].circle_begin, arr[i].circle_count = sum(len(c) for c in circ), len(cs)
        circ.append(cs)
        off += max(w, 0) * max(h, 0)
    allc = np.concatenate(circ) if circ else np.zeros((0, 3), np.int32)
    d_c = DevBuf(ctx, allc if len(allc) else np.zeros(3, np.int32))
    d_j = DevBuf(ctx, np.frombuffer(bytes(arr), np.uint8).copy())
    d_b = DevBuf(ctx, nbytes=off + 64, fill=0x5A)
    d_s = DevBuf(ctx, nbytes=4 * n, fill=0x7F)
    capi.check(ctx.handle, capi.lib().slam_hip_disc_masks_d(ctx.handle, d_c.ptr, len(allc), d_j.ptr, n, d_b.ptr, off, d_s.ptr, hint))
    ctx.synchronize()
    blob, st = d_b.get(), d_s.get(np.int32)
    for b in (d_c, d_j, d_b, d_s):
        b.free()
    assert (blob[off:] == 0x5A).all()
    return [(blob[arr[i].mask_off:arr[i].mask_off + max(w, 0) * max(h, 0)].reshape(max(h, 0), max(w, 0)), int(st[i]))
            for i, (w, h, _) in enumerate(jobs)]


def test_disc_masks_match_cv_circle(ctx, oracle):
    rng = np.random.default_rng(3)
    jobs = []
    for i in range(60):
        w, h = int(rng.integers(1, 301)), int(rng.integers(1, 301))
        k = int(rng.integers(0, 60))
        cs = np.stack([rng.integers(-40, w + 40, k), rng.integers(-40, h + 40, k), rng.integers(0, 45, k)], 1)
        jobs.append((w, h, cs))
    jobs.append((300, 300, [[150, 150, 149], [0, 0, 0], [299, 299, 1], [-500, 10, 505]]))
    jobs.append((64, 48, np.stack([rng.integers(0, 64, 700), rng.integers(0, 48, 700), rng.integers(0, 3, 700)], 1)))   # > one LDS chunk
    jobs.append((17, 9, np.zeros((0, 3), np.int32)))                                                                  # no relevant bin
    for hint in (0, 90000):
        got = _disc_masks(ctx, jobs, hint)
        for (w, h, cs), (m, st) in zip(jobs, got):
            assert st == 0
            assert np.array_equal(m, oracle.disc_mask(w, h, cs)), (w, h, len(cs))
    bad = _disc_masks(ctx, [(8, 8, [[4, 4, -1]]), (8, 8, [[4, 4, 2]])])
    assert bad[0][1] == capi.EINVAL and (bad[0][0] == 0x5A).all()
    assert bad[1][1] == 0 and bad[1][0].sum() == 13 * 255


def _line_jobs(rng, w, h, subs, n, half, frame0, shift):
    """Line features cut from frame0; the search area on the coarsest level is a chain of discs (projected cone bins)
    around the true position in the shifted frame."""
    top = float(subs[-1])
    jobs = []
    for (cx, cy) in synth.feature_grid(w, h, n, half + 40, rng):
        d = synth.cut_template(frame0, cx, cy, 2 * half + 1, 2 * half + 1)
        tx, ty = (cx - shift[0]) / top, (cy - shift[1]) / top
        k = int(rng.integers(3, 9))
        ang = rng.uniform(0, np.pi)
        ts = np.linspace(-1, 1, k) * rng.uniform(4, 30) / top
        cs = np.stack([tx + ts * np.cos(ang) + rng.uniform(-1, 1), ty + ts * np.sin(ang) + rng.uniform(-1, 1),
                       rng.integers(2, 9, k)], 1)
        x0, y0 = int(np.floor((cs[:, 0] - cs[:, 2]).min())), int(np.floor((cs[:, 1] - cs[:, 2]).min()))
        x1, y1 = int(np.ceil((cs[:, 0] + cs[:, 2] + 1).max())), int(np.ceil((cs[:, 1] + cs[:, 2] + 1).max()))
        circ = np.stack([np.round(cs[:, 0]) - x0, np.round(cs[:, 1]) - y0, cs[:, 2]], 1).astype(np.int32)
        jobs.append(dict(frame=0, descriptor=d, area=(x0, y0, x1 - x0, y1 - y0), circles=circ, threshold=0.9, stride=2))
    return jobs


@pytest.mark.parametrize("subs,half,n", [([1.0, 2.0], 20, 24), ([1.0, 2.0, 4.0], 12, 16), ([1.0], 7, 8), ([1.0, 3.0], 40, 6)])
def test_line_track_matches_oracle(ctx, oracle, subs, half, n):
    rng = np.random.default_rng(60 + half)
    w, h = 640, 480
    shift = (4, -3)
    f0, f1 = synth.frame(w, h, 23), synth.frame(w, h, 23, shift)
    jobs = _line_jobs(rng, w, h, subs, n, half, f0, shift)
    for j in jobs:
        j["area_mask"] = oracle.disc_mask(j["area"][2], j["area"][3], j["circles"])
    # one area hanging over the image corner, one entirely outside (matching.cpp:26-30), one below the threshold
    jobs.append(dict(jobs[0], area=(-5, -7, 30, 25), area_mask=oracle.disc_mask(30, 25, [[12, 12, 11]])))
    jobs.append(dict(jobs[0], area=(5000, 10, 9, 9), area_mask=np.full((9, 9), 255, np.uint8)))
    jobs.append(dict(jobs[1], descriptor=255 - jobs[1]["descriptor"][::-1].copy(), threshold=0.999))
    pyr = api.ImagePyramid(ctx, f1, subs)
    try:
        got = api.line_track_batch(ctx, pyr, jobs)
        fused = api.match_lines_batch(ctx, pyr, jobs[:n])          # masks drawn on the device from the circles
    finally:
        pyr.close()
    assert fused == got[:n]
    levels = oracle.pyramid(f1, subs)
    full = 0
    for j, g in zip(jobs, got):
        r = oracle.line_track(levels, subs, j["descriptor"], j["area"], j["area_mask"], j["threshold"], j["stride"])
        assert g["status"] == 0
        assert (g["x"], g["y"], g["levels_run"]) == (r["x"], r["y"], r["levels_run"]), (j["area"], g, r)
        assert np.float32(g["response"]).view(np.uint32) == np.float32(r["response"]).view(np.uint32)
        assert np.float32(g["dx"]).view(np.uint32) == np.float32(r["dx"]).view(np.uint32)
        assert np.float32(g["dy"]).view(np.uint32) == np.float32(r["dy"]).view(np.uint32)
        full += r["levels_run"] == len(subs)
    assert full >= n // 2
    assert got[-2]["response"] == -1.0 and (got[-2]["x"], got[-2]["y"]) == (0, 0)


@pytest.mark.parametrize("bands", [1, 2, 3, 5, 8])
def test_line_walks_do_not_depend_on_the_band_count(oracle, bands, monkeypatch):
    """Small batches have their coarsest search cut into bands of rows, one wavefront each (by batch size; forced here
    through SLAM_HIP_LINE_BANDS): the results - the sub-pixel fit of walks that end on the coarsest level included,
    whose neighbours may sit in another band - equal the oracle's for every band count; areas over the image border
    or lower than the band count are searched whole."""
    monkeypatch.setenv("SLAM_HIP_LINE_BANDS", str(bands))
    rng = np.random.default_rng(160 + SOAK_SEED)
    w, h = 640, 480
    shift = (4, -3)
    f0, f1 = synth.frame(w, h, 29 + SOAK_SEED), synth.frame(w, h, 29 + SOAK_SEED, shift)
    with api.Context(0) as c:
        for subs, half, n in (([1.0, 2.0], 20, 12), ([2.0], 16, 6), ([1.0, 2.0, 4.0], 12, 8)):
            jobs = _line_jobs(rng, w, h, subs, n, half, f0, shift)
            for j in jobs:
                j["area_mask"] = oracle.disc_mask(j["area"][2], j["area"][3], j["circles"])
            jobs.append(dict(jobs[0], area=(-5, -7, 30, 25), area_mask=oracle.disc_mask(30, 25, [[12, 12, 11]])))      # over the corner
            jobs.append(dict(jobs[1], threshold=1.5))                                       # ends on the coarsest level
            jobs.append(dict(jobs[2], area=(jobs[2]["area"][0], jobs[2]["area"][1], 9, 2), area_mask=np.full((2, 9), 255, np.uint8)))  # two rows
            jobs.append(dict(jobs[3], descriptor=np.zeros_like(jobs[3]["descriptor"])))       # every score equal: the first cell wins
            pyr = api.ImagePyramid(c, f1, subs)
            try:
                got = api.line_track_batch(c, pyr, jobs)
            finally:
                pyr.close()
            levels = oracle.pyramid(f1, subs)
            for j, g in zip(jobs, got):
                r = oracle.line_track(levels, subs, j["descriptor"], j["area"], j["area_mask"], j["threshold"], j["stride"])
                assert g["status"] == 0
                assert (g["x"], g["y"], g["levels_run"]) == (r["x"], r["y"], r["levels_run"]), (bands, j["area"], g, r)
                for k in ("response", "dx", "dy"):
                    assert np.float32(g[k]).view(np.uint32) == np.float32(r[k]).view(np.uint32), (bands, k, j["area"], g, r)


def test_golden_projection_line_newpoint_vectors_on_gpu(ctx):
    """The committed vectors for Feature::project + track, the line masks and walks and the new-point selection
    (tests/golden/make_golden.py), reproduced by the HIP path without the oracle in the loop."""
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden as MG
    gold = np.load(os.path.join(here, "golden", "golden_r01.npz"))
    pc = MG.project_case()
    pyr = api.ImagePyramid(ctx, pc["frame"], pc["subs"])
    try:
        res = api.project_track_batch(ctx, pyr, pc["feats"])
    finally:
        pyr.close()
    for i, (f, r) in enumerate(zip(pc["feats"], res)):
        proj, mask, loc = api.project(ctx, f["descriptor"], f["P"], f["B"], f["p3d"])
        assert list(loc) == gold["project/loc"][i].tolist()
        assert [MG.crc(proj), MG.crc(mask)] == gold["project/crc"][i].tolist()
        assert [r["x"], r["y"], r["levels_run"], int(np.float32(r["response"]).view(np.uint32))] == gold["project/track"][i].tolist()
    lc = MG.line_case()
    masks = _disc_masks(ctx, [(j["area"][2], j["area"][3], j["circles"]) for j in lc["jobs"]])
    assert [MG.crc(m) for m, _ in masks] == gold["line/mask_crc"].tolist()
    pyr = api.ImagePyramid(ctx, lc["frame"], lc["subs"])
    try:
        res = api.match_lines_batch(ctx, pyr, lc["jobs"])
    finally:
        pyr.close()
    got = [[r["x"], r["y"], r["levels_run"], int(np.float32(r["response"]).view(np.uint32))] for r in res]
    assert got == gold["line/track"].tolist()
    f = MG.fast_frames()[0]
    kps = api.fast(ctx, f, 20, True)[0][0]
    sel = api.add_new_points(ctx, f, kps, np.array([[100, 100], [320, 180]], np.float32), 40, 30.0, 10, 10, want_descriptors=False)["sel"]
    assert np.array_equal(sel, gold["newpoints/sel"])


def test_line_track_random_shapes_against_oracle(ctx, oracle):
    """Differential test of the line walk over odd image sizes, fractional pyramid factors, non-square descriptors
    (down to 1x1: some levels resize to nothing), areas partly or wholly off the image, empty and full masks,
    strides 1..4 and thresholds that end the walk at every level."""
    rng = np.random.default_rng(777 + SOAK_SEED)
    nbad = 0
    for trial in range(10):
        w, h = int(rng.integers(60, 400)), int(rng.integers(60, 300))
        subs = [[1.0, 2.0], [1.0], [1.0, 3.0], [1.0, 1.5, 2.5], [1.0, 2.0, 4.0]][trial % 5]
        top = subs[-1]
        f0 = synth.frame(w, h, 900 + trial)
        f1 = synth.frame(w, h, 900 + trial, (int(rng.integers(-3, 4)), int(rng.integers(-3, 4))))
        jobs = []
        for k in range(24):
            dw, dh = min(int(rng.integers(1, 70)), w), min(int(rng.integers(1, 70)), h)
            cx, cy = int(rng.integers(dw // 2, w - (dw - dw // 2) + 1)), int(rng.integers(dh // 2, h - (dh - dh // 2) + 1))
            d = f0[cy - dh // 2:cy - dh // 2 + dh, cx - dw // 2:cx - dw // 2 + dw].copy()
            aw, ah = int(rng.integers(1, 50)), int(rng.integers(1, 40))
            ax, ay = int(cx / top) - aw // 2 + int(rng.integers(-6, 7)), int(cy / top) - ah // 2 + int(rng.integers(-6, 7))
            if k % 8 == 0:
                ax, ay = int(rng.choice([-aw - 3, -aw // 2, int(w / top) - 2, int(w / top) + 9])), int(rng.choice([-ah // 2, int(h / top) - 1]))
            kind = k % 3
            am = (np.full((ah, aw), 255, np.uint8) if kind == 0 else np.zeros((ah, aw), np.uint8) if (kind == 1 and k % 2)
                  else oracle.disc_mask(aw, ah, np.stack([rng.integers(0, aw, 4), rng.integers(0, ah, 4), rng.integers(0, 12, 4)], 1)))
            jobs.append(dict(frame=0, descriptor=d, area=(ax, ay, aw, ah), area_mask=am,
                             threshold=float(rng.choice([-5.0, 0.5, 0.9, 1.6])), stride=int(rng.integers(1, 5))))
        pyr = api.ImagePyramid(ctx, f1, subs)
        try:
            got = api.line_track_batch(ctx, pyr, jobs)
        finally:
            pyr.close()
        levels = oracle.pyramid(f1, subs)
        for j, g in zip(jobs, got):
            r = oracle.line_track(levels, subs, j["descriptor"], j["area"], j["area_mask"], j["threshold"], j["stride"])
            same = g["status"] == 0 and (g["x"], g["y"], g["levels_run"]) == (r["x"], r["y"], r["levels_run"])
            for key in ("response", "dx", "dy"):
                a, b = np.float32(g[key]), np.float32(r[key])
                same = same and ((np.isnan(a) and np.isnan(b)) or a.view(np.uint32) == b.view(np.uint32))
            if not same:
                nbad += 1
                print("trial", trial, subs, j["descriptor"].shape, j["area"], j["threshold"], j["stride"], g, r)
    assert nbad == 0


@pytest.mark.parametrize("lds", [4096, 6400, 10240, 12800, 24576, 65536])
def test_line_walks_do_not_depend_on_the_lean_coarse_search(oracle, lds, monkeypatch):
    """Large batches search their coarsest level with the lean core first (line_lean_coarse_kernel: blocks of cell rows,
    chunks of template rows, window energies from box sums of the squared image, numerators without entry lists);
    forced here for small batches, at LDS budgets that move every block and chunk boundary.  Fixed cases (walks that end
    on the coarsest level, areas over the border, constant descriptors) and random shapes - descriptors up to 150 pixels
    wide, for which the box sums do not apply (more than 128 pixel columns under the window) and the entry lists are used."""
    monkeypatch.setenv("SLAM_HIP_LINE_LEAN_MIN", "1")
    monkeypatch.setenv("SLAM_HIP_LINE_BANDS", "1")
    monkeypatch.setenv("SLAM_HIP_LINE_LEAN_LDS_BYTES", str(lds))
    rng = np.random.default_rng(260 + lds + SOAK_SEED)
    w, h = 640, 480
    shift = (4, -3)
    f0, f1 = synth.frame(w, h, 41 + SOAK_SEED), synth.frame(w, h, 41 + SOAK_SEED, shift)
    nbad = 0
    with api.Context(0) as c:
        for subs, half, n in (([1.0, 2.0], 40, 10), ([2.0], 16, 6), ([1.0, 2.0, 4.0], 12, 8), ([1.0], 9, 6), ([1.0, 3.0], 33, 6)):
            top = subs[-1]
            jobs = _line_jobs(rng, w, h, subs, n, half, f0, shift)
            for j in jobs:
                j["area_mask"] = oracle.disc_mask(j["area"][2], j["area"][3], j["circles"])
            jobs.append(dict(jobs[0], area=(-5, -7, 30, 25), area_mask=oracle.disc_mask(30, 25, [[12, 12, 11]])))      # over the corner
            jobs.append(dict(jobs[1], threshold=1.5))                                       # ends on the coarsest level
            jobs.append(dict(jobs[2], area=(jobs[2]["area"][0], jobs[2]["area"][1], 9, 2), area_mask=np.full((2, 9), 255, np.uint8)))
            jobs.append(dict(jobs[3], descriptor=np.zeros_like(jobs[3]["descriptor"])))       # every score equal: the first cell wins
            for k in range(16):                                                                # random shapes
                dw, dh = int(rng.integers(1, 151)), int(rng.integers(1, 100))
                cx, cy = int(rng.integers(dw // 2 + 2, w - dw)), int(rng.integers(dh // 2 + 2, h - dh))
                d = f0[cy - dh // 2:cy - dh // 2 + dh, cx - dw // 2:cx - dw // 2 + dw].copy()
                aw, ah = int(rng.integers(1, 100)), int(rng.integers(1, 70))
                ax, ay = int(cx / top) - aw // 2 + int(rng.integers(-6, 7)), int(cy / top) - ah // 2 + int(rng.integers(-6, 7))
                am = np.full((ah, aw), 255, np.uint8) if k % 2 else (rng.random((ah, aw)) < 0.5).astype(np.uint8) * 255
                jobs.append(dict(frame=0, descriptor=d, area=(ax, ay, aw, ah), area_mask=am,
                                 threshold=float(rng.choice([0.5, 0.9, 1.6])), stride=int(rng.integers(1, 4))))
            pyr = api.ImagePyramid(c, f1, subs)
            try:
                got = api.line_track_batch(c, pyr, jobs)
            finally:
                pyr.close()
            levels = oracle.pyramid(f1, subs)
            for j, g in zip(jobs, got):
                r = oracle.line_track(levels, subs, j["descriptor"], j["area"], j["area_mask"], j["threshold"], j["stride"])
                same = g["status"] == 0 and (g["x"], g["y"], g["levels_run"]) == (r["x"], r["y"], r["levels_run"])
                for key in ("response", "dx", "dy"):
                    a, b = np.float32(g[key]), np.float32(r[key])
                    same = same and ((np.isnan(a) and np.isnan(b)) or a.view(np.uint32) == b.view(np.uint32))
                if not same:
                    nbad += 1
                    print(lds, subs, j["descriptor"].shape, j["area"], j["threshold"], j["stride"], g, r)
    assert nbad == 0
