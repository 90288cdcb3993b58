"""GPU parity of the template projection (SLAM::Feature::project, feature.cpp:53-100) and of
Feature::track run the way the reference runs it (project, then the level walk): device warp against the
oracle's restatement of cv::warpPerspective, byte for byte.  No test of the reference pins these results
(parity unpinned): the oracle is the spec of record."""
import ctypes as C

import numpy as np
import pytest

from conftest import SOAK_SEED

from slam_amd import api, capi, synth

pytestmark = pytest.mark.gpu


DevBuf = api.DeviceBuffer


def _warp_batch(ctx, cases, want_mask=True, hint=0):
    """cases: list of (descriptor, M 3x3, ow, oh).  Returns [(proj, mask, status)] from one slam_hip_warp_batch_d."""
    n = len(cases)
    jobs = (capi.WarpJob * n)()
    parts, off_in, off_out = [], 0, 0
    for i, (d, M, ow, oh) in enumerate(cases):
        j = jobs[i]
        j.desc_off, j.dh, j.dw = off_in, d.shape[0], d.shape[1]
        parts.append(np.ascontiguousarray(d).reshape(-1)); off_in += d.size
        j.ow, j.oh = ow, oh
        for k in range(9):
            j.m[k] = float(np.float32(np.asarray(M).reshape(9)[k]))
        j.proj_off = off_out; off_out += max(ow * oh, 0)
        j.mask_off = off_out if want_mask else -1
        off_out += max(ow * oh, 0) if want_mask else 0
    blob = np.concatenate(parts)
    d_in, d_jobs = DevBuf(ctx, blob), DevBuf(ctx, np.frombuffer(bytes(jobs), np.uint8).copy())
    d_out, d_st = DevBuf(ctx, nbytes=off_out + 64, fill=0xAB), DevBuf(ctx, nbytes=4 * n, fill=0x7F)
    capi.check(ctx.handle, capi.lib().slam_hip_warp_batch_d(ctx.handle, d_in.ptr, blob.size, d_jobs.ptr, n, d_out.ptr,
                                                             off_out, d_st.ptr, hint))
    ctx.synchronize()
    out, st = d_out.get(), d_st.get(np.int32)
    for b in (d_in, d_jobs, d_out, d_st):
        b.free()
    assert (out[off_out:] == 0xAB).all(), "wrote past the output blob"
    res = []
    for i, (d, M, ow, oh) in enumerate(cases):
        j = jobs[i]
        npx = max(ow * oh, 0)
        p = out[j.proj_off:j.proj_off + npx].reshape(max(oh, 0), max(ow, 0))
        m = out[j.mask_off:j.mask_off + npx].reshape(max(oh, 0), max(ow, 0)) if want_mask else None
        res.append((p, m, int(st[i])))
    return res


def _rand_homography(rng, dw, dh, ow, oh, persp=2e-3):
    """Inverse map (output pixel -> descriptor coordinates): rotation, anisotropic scale, shear, shift, perspective."""
    a = rng.uniform(-np.pi, np.pi) if rng.random() < 0.3 else rng.uniform(-0.4, 0.4)
    sx, sy = rng.uniform(0.5, 2.0, 2) * (dw / max(ow, 1), dh / max(oh, 1))
    R = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]) @ np.diag([sx, sy]) @ np.array([[1, rng.uniform(-0.3, 0.3)], [0, 1]])
    M = np.eye(3)
    M[:2, :2] = R
    M[:2, 2] = np.array([dw / 2, dh / 2]) - R @ np.array([ow / 2, oh / 2]) + rng.uniform(-3, 3, 2)
    M[2, :2] = rng.uniform(-persp, persp, 2)
    return (M * rng.uniform(0.2, 5.0)).astype(np.float32)


def test_warp_random_homographies(ctx, oracle):
    rng = np.random.default_rng(11 + SOAK_SEED)
    cases = []
    for i in range(160):
        dw, dh = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        ow, oh = int(rng.integers(1, 120)), int(rng.integers(1, 120))
        d = rng.integers(0, 256, (dh, dw), dtype=np.uint8)
        cases.append((d, _rand_homography(rng, dw, dh, ow, oh), ow, oh))
    # outputs wider than one coordinate block of warpPerspective (block width 64 for tall outputs, 1024/h for flat ones)
    for (ow, oh) in [(500, 500), (500, 3), (341, 3), (342, 3), (65, 16), (64, 40), (129, 17), (1, 500), (300, 1), (1023, 1), (1025, 1)]:
        d = rng.integers(0, 256, (41, 57), dtype=np.uint8)
        cases.append((d, _rand_homography(rng, 57, 41, ow, oh, persp=2e-4), ow, oh))
    got = _warp_batch(ctx, cases, hint=250000)
    for (d, M, ow, oh), (p, m, st) in zip(cases, got):
        assert st == 0
        assert np.array_equal(p, oracle.warp_perspective(d, M, ow, oh, False)), (d.shape, ow, oh)
        assert np.array_equal(m, oracle.warp_perspective(np.ones_like(d), M, ow, oh, True)), (d.shape, ow, oh)
    # the same batch without masks and without the size hint (one workgroup walks a whole output)
    got2 = _warp_batch(ctx, cases[-11:], want_mask=False)
    for (p1, _, _), (p2, m2, st) in zip(got[-11:], got2):
        assert st == 0 and m2 is None and np.array_equal(p1, p2)


def test_warp_large_outputs(ctx, oracle):
    """Outputs of more than 2^20 pixels (the kernel splits the pixel index differently there) and the 4096 limit."""
    rng = np.random.default_rng(13 + SOAK_SEED)
    d = rng.integers(0, 256, (120, 150), dtype=np.uint8)
    cases = [(d, _rand_homography(rng, 150, 120, ow, oh, persp=2e-5), ow, oh) for (ow, oh) in [(1100, 1000), (4096, 300), (257, 4096)]]
    got = _warp_batch(ctx, cases, hint=4096 * 300)
    for (dd, M, ow, oh), (p, m, st) in zip(cases, got):
        assert st == 0
        assert np.array_equal(p, oracle.warp_perspective(dd, M, ow, oh, False)), (ow, oh)
        assert np.array_equal(m, oracle.warp_perspective(np.ones_like(dd), M, ow, oh, True)), (ow, oh)
    r = _warp_batch(ctx, [(d, np.eye(3, dtype=np.float32), 4097, 2)])
    assert r[0][2] == capi.EINVAL


def test_warp_degenerate_maps(ctx, oracle):
    """Zero denominators (W = 0 -> coordinate 0), coordinates beyond the short and int ranges, exact half steps
    (round-half-even), maps that miss the descriptor entirely, and jobs the device must reject."""
    rng = np.random.default_rng(12 + SOAK_SEED)
    d = rng.integers(1, 256, (23, 31), dtype=np.uint8)
    Ms = [
        np.zeros((3, 3)),                                              # W = 0 everywhere: every pixel reads descriptor (0,0)
        [[1, 0, 0], [0, 1, 0], [1, 0, -5]],                            # W = 0 on column 5
        [[1, 0, 0.5], [0, 1, 0.5], [0, 0, 1]],                         # half-pixel shift: (a+b+c+d+2)>>2
        [[1, 0, 1 / 64.], [0, 1, 3 / 64.], [0, 0, 1]],                 # ties of the 1/32 rounding
        [[1, 0, -1], [0, 1, -1], [0, 0, 1]],                           # one row / column outside
        [[1e6, 0, 0], [0, 1e6, 0], [0, 0, 1]],                         # beyond short
        [[1e12, 0, 5], [0, -1e12, 5], [0, 0, 1]],                      # beyond int
        [[1, 0, 1000], [0, 1, 1000], [0, 0, 1]],                       # misses the descriptor
        [[1, 0, 0], [0, 1, 0], [0, 0, 1e-20]],                         # huge W
        [[0.5, 0, 30.999], [0, 0.5, 22.999], [0, 0, 1]],               # last row / column, partly outside taps
    ]
    cases = [(d, np.array(M, np.float32), 40, 30) for M in Ms]
    got = _warp_batch(ctx, cases)
    for (dd, M, ow, oh), (p, m, st) in zip(cases, got):
        assert st == 0
        assert np.array_equal(p, oracle.warp_perspective(dd, M, ow, oh, False)), M
        assert np.array_equal(m, oracle.warp_perspective(np.ones_like(dd), M, ow, oh, True)), M
    assert (got[0][0] == d[0, 0]).all() and (got[7][0] == 0).all() and (got[7][1] == 0).all()
    half = (d[:-1, :-1].astype(int) + d[:-1, 1:] + d[1:, :-1] + d[1:, 1:] + 2) >> 2
    assert np.array_equal(got[2][0][:22, :30], half)
    # rejected jobs: NaN matrix, empty projection is fine (status 0, nothing written)
    bad = [(d, np.full((3, 3), np.nan, np.float32), 8, 8), (d, np.eye(3, dtype=np.float32), 0, 0)]
    r = _warp_batch(ctx, bad)
    assert r[0][2] == capi.EINVAL and (r[0][0] == 0xAB).all()
    assert r[1][2] == 0


def _features(rng, w, h, n, dsize, frame0):
    P = synth.camera(w, h, rng)
    feats = []
    for (cx, cy) in synth.feature_grid(w, h, n, dsize + 30, rng):
        B, p3d, p2d = synth.plane_feature(P, cx, cy, rng)
        feats.append(dict(descriptor=synth.cut_template(frame0, cx, cy, dsize, dsize), P=P, B=B, p3d=p3d, p2d=p2d, c=(cx, cy)))
    return feats


@pytest.mark.parametrize("dsize", [11, 21, 81])
def test_project_matches_oracle(ctx, oracle, dsize):
    rng = np.random.default_rng(20 + dsize)
    w, h = 640, 480
    f0 = synth.frame(w, h, 91)
    nonempty = 0
    for ft in _features(rng, w, h, 24, dsize, f0):
        ref = oracle.project_setup(ft["P"], ft["B"], ft["p3d"], dsize, dsize)
        got = api.project_setup(ft["P"], ft["B"], ft["p3d"], dsize, dsize)
        assert got["loc"] == ref["loc"] and got["valid"] == ref["valid"]
        assert np.array_equal(got["m"].view(np.uint32), ref["m"].view(np.uint32))
        rp, rm, rl = oracle.project(ft["descriptor"], ft["P"], ft["B"], ft["p3d"])
        gp, gm, gl = api.project(ctx, ft["descriptor"], ft["P"], ft["B"], ft["p3d"])
        assert gl == rl and np.array_equal(gp, rp) and np.array_equal(gm, rm)
        assert set(np.unique(gm)) <= {0, 1}
        nonempty += gp.size > 0
    assert nonempty >= 20


def test_project_empty_and_oversize(ctx, oracle):
    """feature.cpp:84-86: a projection wider than 500 pixels, or collapsed to nothing, is empty."""
    P = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]], np.float32)
    d = np.arange(21 * 21, dtype=np.uint8).reshape(21, 21)
    for scale in (30.0, 0.01, 1.0):
        B = np.array([[scale, 0, 0], [0, scale, 0], [0, 0, 0], [0, 0, 1]], np.float32)
        rp, rm, rl = oracle.project(d, P, B, [100, 80, 1])
        gp, gm, gl = api.project(ctx, d, P, B, [100, 80, 1])
        assert gl == rl and gp.shape == rp.shape and np.array_equal(gp, rp) and np.array_equal(gm, rm)
    assert gp.shape == (20, 20) and np.array_equal(gp, d[:20, :20])       # identity homography: the descriptor itself


@pytest.mark.parametrize("subs,dsize,n", [([1.0, 2.0, 4.0], 21, 64), ([1.0, 3.0], 27, 24), ([1.0], 11, 16)])
def test_project_track_batch_matches_oracle(ctx, oracle, subs, dsize, n):
    """Feature::track end to end (feature.cpp:102-173): project on the device, search with the projected template."""
    rng = np.random.default_rng(40 + dsize)
    w, h = 640, 480
    shift = (3, -2)
    f0, f1 = synth.frame(w, h, 17), synth.frame(w, h, 17, shift)
    feats = _features(rng, w, h, n, dsize, f0)
    jobs = []
    for ft in feats:
        px = float(ft["p2d"][0]) - shift[0] + float(rng.integers(-3, 4)) + 0.3
        py = float(ft["p2d"][1]) - shift[1] + float(rng.integers(-3, 4)) - 0.2
        jobs.append(dict(frame=0, descriptor=ft["descriptor"], P=ft["P"], B=ft["B"], p3d=ft["p3d"], px=px, py=py,
                         threshold=0.9, stride=3))
    jobs.append(dict(jobs[0], B=np.zeros((4, 3), np.float32)))           # empty projection: (-1,-1), response -1
    pyr = api.ImagePyramid(ctx, f1, subs)
    try:
        got = api.project_track_batch(ctx, pyr, jobs)
    finally:
        pyr.close()
    levels = oracle.pyramid(f1, subs)
    found = 0
    for j, g in zip(jobs, got):
        proj, mask, _ = oracle.project(j["descriptor"], j["P"], j["B"], j["p3d"])
        if proj.size == 0:
            assert (g["x"], g["y"], g["response"], g["status"]) == (-1, -1, -1.0, 0)
            continue
        r = oracle.track(levels, subs, proj, mask, j["px"], j["py"], j["threshold"], j["stride"])
        assert g["status"] == 0
        assert (g["x"], g["y"], g["levels_run"]) == (r["x"], r["y"], r["levels_run"]), (g, r)
        assert np.float32(g["response"]).view(np.uint32) == np.float32(r["response"]).view(np.uint32)
        found += r["levels_run"] == len(subs)
    assert found >= n // 2
