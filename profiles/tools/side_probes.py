#!/usr/bin/env python3
"""Times the kernels around the search on their own, with inputs resident in HBM:
  - the pyramid build (C2: 1024 frames 640x480 {1,2,4}; C3: 256 frames 1280x720 {1,2,4,8})
  - FAST-9-16 + NMS (C4: 64 frames 1920x1080), then the keypoint sort and the new-point selection on its output
  - the template projection (warp) of 65536 21x21 descriptors, alone and chained with the search
  - the line-feature path: disc area masks and the 81x81-descriptor searches of 3 lines per frame, 1024 frames
and prints the achieved fraction of the 8 TB/s HBM peak on the algorithmic bytes.
Diagnostic tool (the contract benchmark is bench.py)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def timed(stream, fn, steps=20, warm=3):
    import torch
    for _ in range(warm):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in evs:
        a.record(stream); fn(); b.record(stream)
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))


def main():
    import torch
    from slam_amd import api, capi, synth
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = api.Context(0, stream=stream.cuda_stream)
    L = capi.lib()
    out = {}

    for name, (w, h, subs, nfr) in {"pyramid_c2": (640, 480, [1.0, 2.0, 4.0], 1024),
                                    "pyramid_c3": (1280, 720, [1.0, 2.0, 4.0, 8.0], 256),
                                    "pyramid_ref_1_3": (640, 480, [1.0, 3.0], 1024)}.items():
        base = synth.frame(w, h, 3)
        d = torch.from_numpy(base).to(dev).unsqueeze(0).repeat(nfr, 1, 1).contiguous()
        pyr = api.ImagePyramid(ctx, None, subs, device_ptr=d.data_ptr(), pitch=w, frame_stride=w * h, shape=(nfr, h, w))
        ms = timed(stream, lambda: pyr.rebuild_d(d.data_ptr(), w, w * h))
        alg = w * h + sum(pyr.level(i).w * pyr.level(i).h for i in range(1, len(subs)))
        gbs = alg * nfr / (ms * 1e-3) / 1e9
        out[name] = {"ms": ms, "frames": nfr, "alg_bytes_per_frame": alg, "GB/s": gbs, "frac_hbm": gbs / 8000.0,
                     "Mpx/s": w * h * nfr / (ms * 1e-3) / 1e6}
        pyr.close()
        del d

    w, h, nfr, cap = 1920, 1080, 64, 65536
    frames = np.stack([synth.frame(w, h, 100 + i) for i in range(4)])
    d = torch.from_numpy(frames).to(dev).repeat(nfr // 4, 1, 1).contiguous()
    sb = L.slam_hip_fast_scratch_bytes(w, h, nfr, cap)
    d_scr = torch.zeros(sb, dtype=torch.uint8, device=dev)
    d_out = torch.zeros(nfr * cap * 12, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(nfr, dtype=torch.int32, device=dev)
    for thr in (10, 40):
        def go():
            capi.check(ctx.handle, L.slam_hip_fast_detect_d(ctx.handle, C.c_void_p(d.data_ptr()), w, h, w, w * h, nfr, thr, 1,
                                                            C.c_void_p(d_out.data_ptr()), cap, C.c_void_p(d_cnt.data_ptr()),
                                                            C.c_void_p(d_scr.data_ptr()), sb))
        ms = timed(stream, go)
        cnt = d_cnt.cpu().numpy()
        alg = w * h * nfr + 12 * int(np.minimum(cnt, cap).sum())
        gbs = alg / (ms * 1e-3) / 1e9
        out["fast_c4_thr%d" % thr] = {"ms": ms, "frames": nfr, "mean_corners_per_frame": float(cnt.mean()), "GB/s": gbs,
                                     "frac_hbm": gbs / 8000.0, "Gpx/s": w * h * nfr / (ms * 1e-3) / 1e9}

    # --- keypoint sort + addNewLines selection on the FAST output above (thr 40), new_image.cpp:80-82 parameters
    d_sorted = torch.zeros(nfr * cap * 12, dtype=torch.uint8, device=dev)
    n_new = 3
    d_sel = torch.zeros(nfr * n_new, dtype=torch.int32, device=dev)
    d_nsel = torch.zeros(nfr, dtype=torch.int32, device=dev)
    P = lambda t: C.c_void_p(t.data_ptr())
    ms = timed(stream, lambda: capi.check(ctx.handle, L.slam_hip_sort_keypoints_d(ctx.handle, P(d_out), P(d_cnt), nfr, cap, P(d_sorted))))
    nk = int(np.minimum(d_cnt.cpu().numpy(), cap).sum())
    out["sort_keypoints_c4_thr40"] = {"ms": ms, "keypoints": nk, "GB/s": 24.0 * nk / (ms * 1e-3) / 1e9, "frac_hbm": 24.0 * nk / (ms * 1e-3) / 8e12}
    ms = timed(stream, lambda: capi.check(ctx.handle, L.slam_hip_select_new_points_d(
        ctx.handle, P(d_sorted), P(d_cnt), nfr, cap, None, None, n_new, 100.0, 40, 40, w, h, P(d_sel), P(d_nsel))))
    out["select_new_points_c4"] = {"ms": ms, "frames": nfr, "kept_per_frame": float(d_nsel.float().mean().item())}
    del d, d_scr, d_out, d_cnt, d_sorted

    # --- template projection: 1024 frames x 64 features, 21x21 descriptors, mild homographies
    w, h, nfr, nfeat, ds = 640, 480, 1024, 64, 21
    rng = np.random.default_rng(7)
    f0 = synth.frame(w, h, 5)
    f1 = synth.frame(w, h, 5, (3, -2))
    Pm = synth.camera(w, h, rng)
    pts = synth.feature_grid(w, h, nfeat, 60, rng)
    wj = (capi.WarpJob * (nfr * nfeat))()
    tj = (capi.TrackJob * (nfr * nfeat))()
    descs, off_in, off_out, alg_warp = [], 0, 0, 0
    for k, (cx, cy) in enumerate(pts):
        B, p3d, p2d = synth.plane_feature(Pm, cx, cy, rng)
        s = api.project_setup(Pm, B, p3d, ds, ds)
        assert s["valid"]
        descs.append(synth.cut_template(f0, cx, cy, ds, ds).reshape(-1))
        for f in range(nfr):
            j, t = wj[f * nfeat + k], tj[f * nfeat + k]
            j.desc_off, j.dw, j.dh, j.ow, j.oh = k * ds * ds, ds, ds, s["loc"][2], s["loc"][3]
            for q in range(9):
                j.m[q] = float(s["m"].reshape(9)[q])
            t.frame, t.projw, t.projh = f, j.ow, j.oh
            t.px, t.py, t.threshold, t.stride = float(p2d[0]) - 3, float(p2d[1]) + 2, 0.98, 3
    for i in range(nfr * nfeat):
        n = wj[i].ow * wj[i].oh
        wj[i].proj_off = tj[i].proj_off = off_out
        wj[i].mask_off = tj[i].mask_off = off_out + n
        off_out += 2 * n
        alg_warp += ds * ds + 2 * n + 80
    d_desc = torch.from_numpy(np.concatenate(descs)).to(dev)
    d_wj = torch.from_numpy(np.frombuffer(bytes(wj), np.uint8).copy()).to(dev)
    d_tj = torch.from_numpy(np.frombuffer(bytes(tj), np.uint8).copy()).to(dev)
    d_blob = torch.zeros(off_out + 64, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(nfr * nfeat * 28, dtype=torch.uint8, device=dev)
    d_fr = torch.from_numpy(f1).to(dev).unsqueeze(0).repeat(nfr, 1, 1).contiguous()
    pyr = api.ImagePyramid(ctx, None, [1.0, 2.0, 4.0], device_ptr=d_fr.data_ptr(), pitch=w, frame_stride=w * h, shape=(nfr, h, w))

    def warp():
        capi.check(ctx.handle, L.slam_hip_warp_batch_d(ctx.handle, P(d_desc), d_desc.numel(), P(d_wj), nfr * nfeat, P(d_blob), off_out, None, 441))

    def track():
        capi.check(ctx.handle, L.slam_hip_track_batch_d(ctx.handle, pyr.handle, P(d_tj), nfr * nfeat, P(d_blob), off_out, P(d_res)))
    ms = timed(stream, warp)
    out["warp_c2_65536x21x21"] = {"ms": ms, "jobs": nfr * nfeat, "alg_bytes": alg_warp, "GB/s": alg_warp / (ms * 1e-3) / 1e9,
                                 "frac_hbm": alg_warp / (ms * 1e-3) / 8e12, "Gpx/s": off_out / 2 / (ms * 1e-3) / 1e9}
    ms_t = timed(stream, track)
    ms_wt = timed(stream, lambda: (warp(), track()))
    res = d_res.cpu().numpy().view(np.int32).reshape(-1, 7)
    out["project_then_track_c2"] = {"ms_track_alone": ms_t, "ms_warp_plus_track": ms_wt,
                                    "tracked_all_levels": float((res[:, 5] == 3).mean())}
    del d_desc, d_wj, d_tj, d_blob, d_res

    # --- C2 at the production template size (new_image.cpp:82: 81x81): 1024 frames x 16 features, masked templates,
    # pyramid {1,2,4}, stride 3; per feature 16 + 4 + 289 cells with 20x20, 40x40 and 81x81 templates
    nf81, ts = 16, 81
    pts81 = synth.feature_grid(w, h, nf81, ts + 10, rng)
    tj = (capi.TrackJob * (nfr * nf81))()
    parts = []
    for k, (cx, cy) in enumerate(pts81):
        proj = synth.cut_template(f0, cx, cy, ts, ts)
        mask = np.ones_like(proj); mask[:6, :9] = 0; mask[-4:, -12:] = 0
        parts += [proj.reshape(-1), mask.reshape(-1)]
        for f in range(nfr):
            t = tj[f * nf81 + k]
            t.frame, t.projw, t.projh = f, ts, ts
            t.proj_off, t.mask_off = 2 * ts * ts * k, 2 * ts * ts * k + ts * ts
            t.px, t.py, t.threshold, t.stride = cx - 3 + 1.0, cy + 2 - 2.0, 0.9, 3
    d_blob = torch.from_numpy(np.concatenate(parts)).to(dev)
    d_tj = torch.from_numpy(np.frombuffer(bytes(tj), np.uint8).copy()).to(dev)
    d_res = torch.zeros(nfr * nf81 * 28, dtype=torch.uint8, device=dev)
    ms = timed(stream, lambda: capi.check(ctx.handle, L.slam_hip_track_batch_d(ctx.handle, pyr.handle, P(d_tj), nfr * nf81, P(d_blob),
                                                                                d_blob.numel(), P(d_res))))
    res = d_res.cpu().numpy().view(np.int32).reshape(-1, 7)
    macs = 2.0 * (16 * 20 * 20 + 4 * 40 * 40 + 289 * 81 * 81) * nfr * nf81
    out["c2_track_16384x81x81"] = {"ms": ms, "jobs": nfr * nf81, "all_levels": float((res[:, 5] == 3).mean()),
                                   "found": float(((res[:, 0] == np.tile([p[0] - 3 for p in pts81], nfr)) & (res[:, 1] == np.tile([p[1] + 2 for p in pts81], nfr))).mean()),
                                   "positions/s": 309.0 * nfr * nf81 / (ms * 1e-3), "u8_mac/s": macs / (ms * 1e-3),
                                   "frac_of_packed_dot_peak": macs / (ms * 1e-3) / 1.4e14}
    pyr.close()
    del d_blob, d_tj, d_res

    # --- line features: 3 lines per frame (new_image.cpp:80: 3 - nLines), 81x81 descriptors, pyramid {1,2}, stride 2
    nline, half = 3, 40
    pyr = api.ImagePyramid(ctx, None, [1.0, 2.0], device_ptr=d_fr.data_ptr(), pitch=w, frame_stride=w * h, shape=(nfr, h, w))
    lj = (capi.LineJob * (nfr * nline))()
    dj = (capi.DiscJob * (nfr * nline))()
    descs, circ, off = [], [], 0
    centres = [(160, 140), (330, 250), (500, 330)]
    for k, (cx, cy) in enumerate(centres):
        descs.append(synth.cut_template(f0, cx, cy, 2 * half + 1, 2 * half + 1).reshape(-1))
    off = sum(d.size for d in descs)
    alg_disc = 0
    for f in range(nfr):
        for k, (cx, cy) in enumerate(centres):
            i = f * nline + k
            tx, ty = (cx - 3) // 2, (cy + 2) // 2
            aw, ah = 60, 40
            cs = [[6 + 6 * b, 8 + 3 * b, 4 + (b + k) % 4] for b in range(8)]
            lj[i].desc_off, lj[i].dw, lj[i].dh = k * (2 * half + 1) ** 2, 2 * half + 1, 2 * half + 1
            lj[i].frame, lj[i].ax, lj[i].ay, lj[i].aw, lj[i].ah = f, tx - 27, ty - 17, aw, ah
            lj[i].amask_off, lj[i].threshold, lj[i].stride = off, 0.9, 2
            dj[i].mask_off, dj[i].w, dj[i].h, dj[i].circle_begin, dj[i].circle_count = off, aw, ah, len(circ), len(cs)
            circ += cs
            off += aw * ah
            alg_disc += aw * ah + 12 * len(cs) + 24
    d_circ = torch.from_numpy(np.asarray(circ, np.int32)).to(dev)
    d_blob = torch.zeros(off + 64, dtype=torch.uint8, device=dev)
    d_blob[:sum(d.size for d in descs)] = torch.from_numpy(np.concatenate(descs)).to(dev)
    d_lj = torch.from_numpy(np.frombuffer(bytes(lj), np.uint8).copy()).to(dev)
    d_dj = torch.from_numpy(np.frombuffer(bytes(dj), np.uint8).copy()).to(dev)
    d_res = torch.zeros(nfr * nline * 28, dtype=torch.uint8, device=dev)
    ms = timed(stream, lambda: capi.check(ctx.handle, L.slam_hip_disc_masks_d(ctx.handle, P(d_circ), len(circ), P(d_dj), nfr * nline,
                                                                               P(d_blob), off, None, 2400)))
    out["disc_masks_3072x60x40"] = {"ms": ms, "alg_bytes": alg_disc, "GB/s": alg_disc / (ms * 1e-3) / 1e9, "frac_hbm": alg_disc / (ms * 1e-3) / 8e12}
    ms = timed(stream, lambda: capi.check(ctx.handle, L.slam_hip_line_track_batch_d(ctx.handle, pyr.handle, P(d_lj), nfr * nline, P(d_blob),
                                                                                     off, P(d_res))))
    res = d_res.cpu().numpy().view(np.int32).reshape(-1, 7)
    # cells per line: the coarsest level evaluates every interior cell of the 60x40 area (no patch mask), level 0 a 5x5 window
    cells = 60 * 40 + 25
    out["line_track_3072x81x81"] = {"ms": ms, "lines": nfr * nline, "refined": float((res[:, 5] == 2).mean()),
                                    "positions/s": cells * nfr * nline / (ms * 1e-3),
                                    "u8_mac/s": (60 * 40 * 40 * 40 + 25 * 81 * 81) * 2.0 * nfr * nline / (ms * 1e-3)}
    pyr.close()
    del d_fr

    # --- C1: 320x240, 16 features, 11x11 templates, one level, 41x41 windows, stride 1, all four mask modes
    # (matchFeatureInArea jobs), 1024 frames
    w, h, nfr, nfeat, ts = 320, 240, 1024, 16, 11
    f1 = synth.frame(w, h, synth.SEED_BASE + 1)
    d_fr = torch.from_numpy(f1).to(dev).unsqueeze(0).repeat(nfr, 1, 1).contiguous()
    pyr = api.ImagePyramid(ctx, None, [1.0], device_ptr=d_fr.data_ptr(), pitch=w, frame_stride=w * h, shape=(nfr, h, w))
    pts = synth.feature_grid(w, h, nfeat, 30, rng)
    mj = (capi.MatchJob * (nfr * nfeat))()
    parts, off = [], 0
    per_feat = []
    for k, (cx, cy) in enumerate(pts):
        patch = synth.cut_template(f1, cx, cy, ts, ts)
        pm = rng.integers(0, 2, (ts, ts), dtype=np.uint8)
        am = synth.ellipse_mask(41, 41, 18.0, 9.0, 0.3 * k)
        o = [off, off + patch.size, off + 2 * patch.size]
        parts += [patch.reshape(-1), pm.reshape(-1), am.reshape(-1)]
        off += 2 * patch.size + am.size
        per_feat.append(o)
    cells = 0
    for f in range(nfr):
        for k, (cx, cy) in enumerate(pts):
            j = mj[f * nfeat + k]
            mode = (f + k) & 3
            j.frame, j.level, j.pw, j.ph = f, 0, ts, ts
            j.patch_off = per_feat[k][0]
            j.pmask_off = per_feat[k][1] if mode & 1 else -1
            j.ax, j.ay, j.aw, j.ah = cx - 20, cy - 20, 41, 41
            j.amask_off = per_feat[k][2] if mode & 2 else -1
            j.stride, j.use_exact_area_mask, j.score_off = 1, 0, -1
            cells += 41 * 41
    d_blob = torch.from_numpy(np.concatenate(parts)).to(dev)
    d_mj = torch.from_numpy(np.frombuffer(bytes(mj), np.uint8).copy()).to(dev)
    d_res = torch.zeros(nfr * nfeat * 24, dtype=torch.uint8, device=dev)
    ms = timed(stream, lambda: capi.check(ctx.handle, L.slam_hip_match_batch_d(ctx.handle, pyr.handle, P(d_mj), nfr * nfeat, P(d_blob), off,
                                                                                None, 0, P(d_res))))
    alg = nfr * nfeat * (51 * 51 + ts * ts + 20)
    out["c1_match_16384x11x11_41x41"] = {"ms": ms, "jobs": nfr * nfeat, "positions/s": cells / (ms * 1e-3), "alg_bytes": alg,
                                         "GB/s": alg / (ms * 1e-3) / 1e9, "frac_hbm": alg / (ms * 1e-3) / 8e12,
                                         "u8_mac/s": cells * 2.0 * ts * ts / (ms * 1e-3)}
    pyr.close()
    del d_fr, d_blob, d_mj, d_res

    # --- C3: 1280x720, 256 features, 31x31 masked templates, pyramid {1,2,4,8}, ellipse windows on the coarsest
    # level (track walk with SLAM_HIP_TRACK_COARSE_AREA), 256 frames
    w, h, nfr, nfeat, ts, subs = 1280, 720, 256, 256, 31, [1.0, 2.0, 4.0, 8.0]
    shift = (5, -4)
    f0, f1 = synth.frame(w, h, synth.SEED_BASE + 33), synth.frame(w, h, synth.SEED_BASE + 33, shift)
    d_fr = torch.from_numpy(f1).to(dev).unsqueeze(0).repeat(nfr, 1, 1).contiguous()
    pyr = api.ImagePyramid(ctx, None, subs, device_ptr=d_fr.data_ptr(), pitch=w, frame_stride=w * h, shape=(nfr, h, w))
    pts = synth.feature_grid(w, h, nfeat, ts + 30, rng)
    tj = (capi.TrackJob * (nfr * nfeat))()
    parts, off, feats = [], 0, []
    for (cx, cy) in pts:
        proj = synth.cut_template(f0, cx, cy, ts, ts)
        mask = np.ones_like(proj); mask[0, :5] = 0; mask[-1, -5:] = 0
        a, b, th = float(rng.uniform(24, 96)) / 8, float(rng.uniform(12, 48)) / 8, float(rng.uniform(0, np.pi))
        ew = 2 * int(np.ceil(max(a, b))) + 1
        am = synth.ellipse_mask(ew, ew, a, b, th)
        px, py = cx - shift[0] + float(rng.integers(-10, 11)) + 0.4, cy - shift[1] + float(rng.integers(-6, 7)) - 0.3
        feats.append((off, off + proj.size, off + 2 * proj.size, ew, px, py, int(am.sum())))
        parts += [proj.reshape(-1), mask.reshape(-1), am.reshape(-1)]
        off += 2 * proj.size + am.size
    for f in range(nfr):
        for k, (po, mo, ao, ew, px, py, _) in enumerate(feats):
            t = tj[f * nfeat + k]
            t.frame, t.projw, t.projh, t.flags = f, ts, ts, capi.TRACK_COARSE_AREA
            t.proj_off, t.mask_off, t.amask_off = po, mo, ao
            t.px, t.py, t.threshold, t.stride = px, py, 0.9, 3
            t.ax, t.ay, t.aw, t.ah = int(round(px / 8)) - ew // 2, int(round(py / 8)) - ew // 2, ew, ew
    d_blob = torch.from_numpy(np.concatenate(parts)).to(dev)
    d_tj = torch.from_numpy(np.frombuffer(bytes(tj), np.uint8).copy()).to(dev)
    d_res = torch.zeros(nfr * nfeat * 28, dtype=torch.uint8, device=dev)
    ms = timed(stream, lambda: capi.check(ctx.handle, L.slam_hip_track_batch_d(ctx.handle, pyr.handle, P(d_tj), nfr * nfeat, P(d_blob), off, P(d_res))))
    res = d_res.cpu().numpy().view(np.int32).reshape(-1, 7)
    ms_p = timed(stream, lambda: pyr.rebuild_d(d_fr.data_ptr(), w, w * h))
    out["c3_track_65536x31x31_ellipse"] = {"ms_track": ms, "ms_pyramid": ms_p, "jobs": nfr * nfeat, "frames": nfr,
                                           "all_levels": float((res[:, 5] == 4).mean()), "frames/s": nfr / ((ms + ms_p) * 1e-3),
                                           "features/s": nfr * nfeat / ((ms + ms_p) * 1e-3)}
    # the same C3 walk with the windows derived from 2x2 covariances (gate 9: semi-axes a, b as above) and the masks
    # rasterised on the device: slam_hip_ellipse_setup (host) + slam_hip_ellipse_masks_d + slam_hip_track_batch_d
    ej = (capi.EllipseJob * (nfr * nfeat))()
    base = 2 * ts * ts * nfeat
    woff, wins = base, []
    for k, (po, mo, ao, ew, px, py, _) in enumerate(feats):
        a, b, th = float(rng.uniform(24, 96)), float(rng.uniform(12, 48)), float(rng.uniform(0, np.pi))
        c, sn = np.cos(th), np.sin(th)
        S = np.array([[c, -sn], [sn, c]]) @ np.diag([a * a / 9, b * b / 9]) @ np.array([[c, sn], [-sn, c]])
        wins.append(api.ellipse_setup(px, py, [S[0, 0], S[0, 1], S[1, 1]], 9.0, 1.0 / 8))
        assert wins[-1].valid
    tparts = []
    for k, (po, mo, ao, ew, px, py, _) in enumerate(feats):
        tparts += [parts[3 * k], parts[3 * k + 1]]
    mask_px = 0
    for f in range(nfr):
        for k, (po, mo, ao, ew, px, py, _) in enumerate(feats):
            i = f * nfeat + k
            ej[i] = wins[k]
            ej[i].mask_off = woff
            t = tj[i]
            t.proj_off, t.mask_off, t.amask_off = 2 * ts * ts * k, 2 * ts * ts * k + ts * ts, woff
            t.ax, t.ay, t.aw, t.ah = wins[k].x, wins[k].y, wins[k].w, wins[k].h
            woff += wins[k].w * wins[k].h
            mask_px += wins[k].w * wins[k].h
    d_blob2 = torch.zeros(woff + 64, dtype=torch.uint8, device=dev)
    d_blob2[:base] = torch.from_numpy(np.concatenate(tparts)).to(dev)
    d_ej = torch.from_numpy(np.frombuffer(bytes(ej), np.uint8).copy()).to(dev)
    d_tj2 = torch.from_numpy(np.frombuffer(bytes(tj), np.uint8).copy()).to(dev)
    max_px = max(wn.w * wn.h for wn in wins)
    ms_e = timed(stream, lambda: capi.check(ctx.handle, L.slam_hip_ellipse_masks_d(ctx.handle, P(d_ej), nfr * nfeat, P(d_blob2), woff, None, max_px)))
    ms_t = timed(stream, lambda: capi.check(ctx.handle, L.slam_hip_track_batch_d(ctx.handle, pyr.handle, P(d_tj2), nfr * nfeat, P(d_blob2), woff, P(d_res))))
    res = d_res.cpu().numpy().view(np.int32).reshape(-1, 7)
    alg_e = mask_px + 48 * nfr * nfeat
    out["c3_covariance_windows_65536"] = {"ms_masks": ms_e, "mask_bytes": mask_px, "GB/s": alg_e / (ms_e * 1e-3) / 1e9,
                                          "frac_hbm": alg_e / (ms_e * 1e-3) / 8e12, "ms_track": ms_t,
                                          "all_levels": float((res[:, 5] == 4).mean()),
                                          "frames/s": nfr / ((ms_e + ms_t + ms_p) * 1e-3)}
    pyr.close()
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
