"""GPU parity of the new-feature steps that follow corner detection (new_point.cpp:124, 137-155;
feature.cpp:42-45): device sort, greedy minimum-distance selection, descriptor crop."""
import ctypes as C

import numpy as np
import pytest

from slam_amd import api, capi, synth

pytestmark = pytest.mark.gpu


DevBuf = api.DeviceBuffer


def _ref_path(oracle, frame, kps_fast_order, occ, n_new, min_dist, rx, ry):
    h, w = frame.shape
    srt = oracle.sort_keypoints(kps_fast_order)
    sel = oracle.select_new_points(srt, occ, n_new, min_dist, rx, ry, w, h)
    descs = [oracle.extract_descriptor(frame, float(srt[i, 0]), float(srt[i, 1]), rx, ry) for i in sel]
    return srt, sel, descs


@pytest.mark.parametrize("thr,n_new,min_dist,r", [(10, 40, 15.0, 10), (25, 200, 6.5, 4), (40, 8, 60.0, 25), (10, 300, 0.0, 0)])
def test_add_new_points_matches_oracle(ctx, oracle, thr, n_new, min_dist, r):
    w, h = 320, 240
    frame = synth.frame(w, h, 70 + thr)
    kps = oracle.fast(frame, thr, True)                       # cv::FAST order
    rng = np.random.default_rng(thr)
    occ = np.stack([rng.uniform(0, w, 12), rng.uniform(0, h, 12)], 1).astype(np.float32)
    occ[:4] = kps[rng.integers(0, len(kps), 4), :2] + np.float32(0.5)      # some right next to corners
    srt, sel, descs = _ref_path(oracle, frame, kps, occ, n_new, min_dist, r, r)
    got = api.add_new_points(ctx, frame, kps, occ, n_new, min_dist, r, r)
    assert np.array_equal(got["sorted"], srt)
    assert np.array_equal(got["sel"], sel)
    assert len(sel) > 0
    for (d, rect), gd, gr in zip(descs, got["descriptors"], got["rects"]):
        assert gr == rect and np.array_equal(gd, d)
        assert d.shape == (2 * r + 1, 2 * r + 1)               # kept points are never cropped (new_point.cpp:148)
    # already sorted input, no occupied positions
    g2 = api.add_new_points(ctx, frame, srt, np.zeros((0, 2), np.float32), n_new, min_dist, r, r, already_sorted=True, want_descriptors=False)
    assert np.array_equal(g2["sel"], oracle.select_new_points(srt, np.zeros((0, 2)), n_new, min_dist, r, r, w, h))


def test_descriptor_crop_at_the_borders(ctx, oracle):
    """newDescriptor crops at the image border (feature.cpp:42-45); half-way coordinates round away from zero."""
    w, h = 97, 61
    frame = synth.frame(w, h, 5)
    pts = [(0.0, 0.0), (2.5, 3.5), (96.0, 60.0), (95.49, 59.5), (48.2, -3.0), (-20.0, 30.0), (50.5, 30.5), (200.0, 30.0)]
    rx, ry = 7, 4
    jobs = (capi.DescJob * len(pts))()
    one = (2 * rx + 1) * (2 * ry + 1)
    for i, (x, y) in enumerate(pts):
        jobs[i].frame, jobs[i].px, jobs[i].py, jobs[i].rx, jobs[i].ry, jobs[i].out_off = 0, x, y, rx, ry, i * one
    d_f = DevBuf(ctx, frame)
    d_j = DevBuf(ctx, np.frombuffer(bytes(jobs), np.uint8).copy())
    d_o = DevBuf(ctx, nbytes=one * len(pts))
    d_r = DevBuf(ctx, nbytes=20 * len(pts))
    L = capi.lib()
    capi.check(ctx.handle, L.slam_hip_extract_descriptors_d(ctx.handle, d_f.ptr, w, h, w, 0, 1, d_j.ptr, len(pts), d_o.ptr,
                                                            d_o.nbytes, d_r.ptr))
    ctx.synchronize()
    out = d_o.get(np.uint8)
    res = d_r.get(np.int32).reshape(-1, 5)
    for b in (d_f, d_j, d_o, d_r):
        b.free()
    for i, (x, y) in enumerate(pts):
        d, rect = oracle.extract_descriptor(frame, x, y, rx, ry)
        assert tuple(res[i, :4]) == rect and res[i, 4] == 0, (i, res[i], rect)
        assert np.array_equal(out[i * one:i * one + rect[2] * rect[3]].reshape(rect[3], rect[2]), d)


def test_batched_device_path_c4(ctx, oracle):
    """FAST -> sort -> select on a batch of frames resident on the device, no host round trip."""
    w, h, nfr, cap, n_new, r = 640, 360, 6, 16384, 64, 8
    frames = np.stack([synth.frame(w, h, 300 + i, (5 * i, 3 * i)) for i in range(nfr)])
    L = capi.lib()
    d_f = DevBuf(ctx, frames)
    sb = L.slam_hip_fast_scratch_bytes(w, h, nfr, cap)
    d_scr = DevBuf(ctx, nbytes=sb)
    d_k = DevBuf(ctx, nbytes=nfr * cap * 12)
    d_s = DevBuf(ctx, nbytes=nfr * cap * 12)
    d_c = DevBuf(ctx, nbytes=nfr * 4)
    rng = np.random.default_rng(9)
    occ_counts = rng.integers(0, 30, nfr)
    occ_off = np.concatenate([[0], np.cumsum(occ_counts)]).astype(np.int32)
    occ = np.stack([rng.uniform(0, w, occ_off[-1]), rng.uniform(0, h, occ_off[-1])], 1).astype(np.float32)
    d_occ = DevBuf(ctx, occ)
    d_off = DevBuf(ctx, occ_off)
    d_sel = DevBuf(ctx, nbytes=nfr * n_new * 4)
    d_n = DevBuf(ctx, nbytes=nfr * 4)
    P = lambda t: t.ptr
    capi.check(ctx.handle, L.slam_hip_fast_detect_d(ctx.handle, P(d_f), w, h, w, w * h, nfr, 20, 1, P(d_k), cap, P(d_c), P(d_scr), sb))
    capi.check(ctx.handle, L.slam_hip_sort_keypoints_d(ctx.handle, P(d_k), P(d_c), nfr, cap, P(d_s)))
    capi.check(ctx.handle, L.slam_hip_select_new_points_d(ctx.handle, P(d_s), P(d_c), nfr, cap, P(d_occ), P(d_off), n_new, 12.0, r, r,
                                                          w, h, P(d_sel), P(d_n)))
    ctx.synchronize()
    cnt = d_c.get(np.int32); srt = d_s.get(np.float32).reshape(nfr, cap, 3)
    sel = d_sel.get(np.int32).reshape(nfr, n_new); nsel = d_n.get(np.int32)
    for b in (d_f, d_scr, d_k, d_s, d_c, d_occ, d_off, d_sel, d_n):
        b.free()
    for f in range(nfr):
        ref_k = oracle.sort_keypoints(oracle.fast(frames[f], 20, True))
        assert cnt[f] == len(ref_k) and np.array_equal(srt[f, :cnt[f]], ref_k)
        ref_sel = oracle.select_new_points(ref_k, occ[occ_off[f]:occ_off[f + 1]], n_new, 12.0, r, r, w, h)
        assert nsel[f] == len(ref_sel) and np.array_equal(sel[f, :nsel[f]], ref_sel)


def test_sort_is_stable_across_run_boundaries(ctx, oracle):
    """Crafted lists around the 2048-key runs one workgroup places: lengths 0, 1, 2047, 2048, 2049 and 10 000, few
    distinct responses (long ties, whose row-major order must survive) and all 256 of them, counts beyond the
    capacity (clamped) - each frame against the oracle's order (response desc, then the input order)."""
    rng = np.random.default_rng(8)
    cap = 10240
    lens = [0, 1, 2047, 2048, 2049, 10000, 6000, 20000]
    nfr = len(lens)
    kps = np.zeros((nfr, cap, 3), np.float32)
    for f, n in enumerate(lens):
        m = min(n, cap)
        hi = [256, 256, 3, 256, 1, 256, 7, 256][f]
        kps[f, :m, 0] = np.arange(m) % 1900            # x, y increasing in row-major order like cv::FAST's output
        kps[f, :m, 1] = np.arange(m) // 1900
        kps[f, :m, 2] = rng.integers(0, hi, m)
    counts = np.array(lens, np.int32)
    L = capi.lib()
    d_in, d_c, d_out = DevBuf(ctx, kps), DevBuf(ctx, counts), DevBuf(ctx, nbytes=kps.nbytes, fill=0xCD)
    capi.check(ctx.handle, L.slam_hip_sort_keypoints_d(ctx.handle, d_in.ptr, d_c.ptr, nfr, cap, d_out.ptr))
    ctx.synchronize()
    got = d_out.get(np.float32).reshape(nfr, cap, 3)
    for b in (d_in, d_c, d_out):
        b.free()
    for f, n in enumerate(lens):
        m = min(n, cap)
        ref = oracle.sort_keypoints(kps[f, :m])
        assert np.array_equal(got[f, :m], ref), f
        assert (got[f, m:].view(np.uint32) == 0xCDCDCDCD).all()        # nothing written beyond the list
