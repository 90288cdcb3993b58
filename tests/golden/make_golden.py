#!/usr/bin/env python3
"""Generates tests/golden/*.npz: inputs are rebuilt from seeds by slam_amd.synth, expected outputs
come from the CPU oracle (oracle/slam_oracle.c).  The reference itself cannot be run (it needs
OpenCV 2.4 and files missing from its tree), so these vectors pin the ORACLE's behaviour — they
are regression anchors for both the oracle and the HIP path, not reference outputs.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pyoracle as O  # noqa: E402
from slam_amd import synth  # noqa: E402

CONFIGS = {
    # name: (w, h, subs, nfeat, tsize, stride, threshold, seed)
    "c1_320x240": (320, 240, [1.0], 16, 11, 3, 0.98, synth.SEED_BASE + 1),
    "c2_640x480": (640, 480, [1.0, 2.0, 4.0], 64, 21, 3, 0.98, synth.SEED_BASE + 2),
    "c3_1280x720": (1280, 720, [1.0, 2.0, 4.0, 8.0], 32, 31, 3, 0.98, synth.SEED_BASE + 3),
    "ref_640x480_1_3": (640, 480, [1.0, 3.0], 24, 27, 3, 0.98, synth.SEED_BASE + 6),
}


def crc(a):
    import zlib
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def track_case(name):
    """Seeded tracking case: returns dict of inputs (rebuilt identically by the tests)."""
    w, h, subs, nfeat, tsize, stride, thr, seed = CONFIGS[name]
    rng = np.random.default_rng(seed)
    shift = (int(rng.integers(-6, 7)), int(rng.integers(-6, 7)))
    f0 = synth.frame(w, h, seed)
    f1 = synth.frame(w, h, seed, shift)
    pts = synth.feature_grid(w, h, nfeat, tsize + 30, rng)
    jobs = []
    for (cx, cy) in pts:
        proj = synth.cut_template(f0, cx, cy, tsize, tsize)
        mask = np.ones_like(proj)
        k = int(rng.integers(0, tsize // 3 + 1))
        for r in range(k):
            mask[r, :k - r] = 0
            mask[tsize - 1 - r, tsize - (k - r):] = 0
        px = cx - shift[0] + int(rng.integers(-4, 5)) + float(np.float32(rng.random()))
        py = cy - shift[1] + int(rng.integers(-4, 5)) + float(np.float32(rng.random()))
        jobs.append(dict(frame=0, proj=proj, proj_mask=mask, px=px, py=py, threshold=thr, stride=stride))
    return dict(frame=f1, subs=subs, jobs=jobs)


def match_cases(seed=synth.SEED_BASE + 11):
    """C1-style single-level searches in all four mask modes, some windows over the border."""
    rng = np.random.default_rng(seed)
    w, h = 320, 240
    im = synth.frame(w, h, seed)
    cases = []
    for mode in range(4):
        for i in range(8):
            cx, cy = int(rng.integers(0, w)), int(rng.integers(0, h))
            tx, ty = int(np.clip(cx, 5, w - 6)), int(np.clip(cy, 5, h - 6))
            patch = synth.cut_template(im, tx, ty, 11, 11)
            area = (cx - 20, cy - 20, 41, 41)
            pm = rng.integers(0, 2, (11, 11), dtype=np.uint8) if mode & 1 else None
            am = synth.ellipse_mask(41, 41, float(rng.uniform(6, 20)), float(rng.uniform(4, 12)), float(rng.uniform(0, np.pi))) if mode & 2 else None
            cases.append(dict(im=im, patch=patch, patch_mask=pm, area=area, area_mask=am,
                              stride=1 if i % 2 == 0 else 3, exact=bool(i % 3 == 0)))
    return cases


def fast_frames(seed=synth.SEED_BASE + 4):
    return [synth.frame(640, 360, seed + i, (13 * i, 7 * i)) for i in range(2)]


def project_case(seed=synth.SEED_BASE + 21, n=16, dsize=21):
    """Seeded Feature::project / Feature::track inputs: one camera, n slanted planar features."""
    rng = np.random.default_rng(seed)
    w, h, shift = 640, 480, (2, -3)
    f0, f1 = synth.frame(w, h, seed), synth.frame(w, h, seed, shift)
    P = synth.camera(w, h, rng)
    feats = []
    for (cx, cy) in synth.feature_grid(w, h, n, dsize + 30, rng):
        B, p3d, p2d = synth.plane_feature(P, cx, cy, rng)
        feats.append(dict(frame=0, descriptor=synth.cut_template(f0, cx, cy, dsize, dsize), P=P, B=B, p3d=p3d,
                          px=float(p2d[0]) - shift[0] + 1.25, py=float(p2d[1]) - shift[1] - 0.5, threshold=0.9, stride=3))
    return dict(frame=f1, subs=[1.0, 3.0], feats=feats)


def line_case(seed=synth.SEED_BASE + 22, n=8, half=20):
    """Seeded LineFeature::track inputs: descriptors, area rectangles and circle lists on the coarsest level."""
    rng = np.random.default_rng(seed)
    w, h, shift, subs = 640, 480, (-3, 2), [1.0, 2.0]
    f0, f1 = synth.frame(w, h, seed), synth.frame(w, h, seed, shift)
    jobs = []
    for (cx, cy) in synth.feature_grid(w, h, n, half + 40, rng):
        tx, ty = (cx - shift[0]) // 2, (cy - shift[1]) // 2
        k = int(rng.integers(3, 8))
        circ = np.stack([rng.integers(4, 40, k), rng.integers(4, 26, k), rng.integers(1, 7, k)], 1).astype(np.int32)
        circ[0] = (22, 15, 5)                                   # the disc that holds the true position
        jobs.append(dict(frame=0, descriptor=synth.cut_template(f0, cx, cy, 2 * half + 1, 2 * half + 1),
                         area=(int(tx) - 22, int(ty) - 15, 44, 30), circles=circ, threshold=0.9, stride=2))
    return dict(frame=f1, subs=subs, jobs=jobs)


def window_case(seed=synth.SEED_BASE + 23, n=24):
    """Seeded uncertainty windows: predicted point, 2x2 covariance (sxx, sxy, syy), gate, level resolution."""
    rng = np.random.default_rng(seed)
    rows = []
    for _ in range(n):
        a, b, th = rng.uniform(2, 30), rng.uniform(1, 15), rng.uniform(0, np.pi)
        c, s = np.cos(th), np.sin(th)
        S = np.array([[c, -s], [s, c]]) @ np.diag([a * a, b * b]) @ np.array([[c, s], [-s, c]])
        rows.append((rng.uniform(0, 640), rng.uniform(0, 480), S[0, 0], S[0, 1], S[1, 1], rng.choice([4.0, 9.0]), rng.choice([1.0, 0.5, 0.25])))
    return np.array(rows, np.float32)


def main():
    O.build()
    out = {}
    for name in CONFIGS:
        c = track_case(name)
        levels = O.pyramid(c["frame"], c["subs"])
        out[name + "/pyr_crc"] = np.array([crc(l) for l in levels], np.uint32)
        out[name + "/pyr_shape"] = np.array([l.shape for l in levels], np.int32)
        res = [O.track(levels, c["subs"], j["proj"], j["proj_mask"], j["px"], j["py"], j["threshold"], j["stride"])
               for j in c["jobs"]]
        out[name + "/track_xy_levels"] = np.array([(r["x"], r["y"], r["levels_run"]) for r in res], np.int32)
        out[name + "/track_resp_dx_dy"] = np.array([(r["response"], r["dx"], r["dy"]) for r in res], np.float32)
    mc = match_cases()
    res = [O.match(c["im"], c["patch"], c["patch_mask"], c["area"], c["area_mask"], c["stride"], c["exact"], want_score=True) for c in mc]
    out["match/xy"] = np.array([(r["x"], r["y"]) for r in res], np.int32)
    out["match/resp_dx_dy"] = np.array([(r["response"], r["dx"], r["dy"]) for r in res], np.float32)
    out["match/score_crc"] = np.array([crc(r["score"]) for r in res], np.uint32)
    for i, f in enumerate(fast_frames()):
        for thr in (10, 40):
            kp = O.fast(f, thr, True)
            out["fast/%d/t%d" % (i, thr)] = kp.astype(np.float32)
        kps, fthr = O.get_sorted_keypoints(f, 100, 10.0)
        out["gsk/%d/kps" % i] = kps[:300].astype(np.float32)
        out["gsk/%d/n_thr" % i] = np.array([len(kps), fthr], np.float64)
    # hand-checkable resize KATs: a 4x4 ramp by 2 and a 6x6 ramp by 3 (point sample at 3d+1)
    ramp4 = (np.arange(16, dtype=np.uint8) * 10).reshape(4, 4)
    ramp6 = (np.arange(36, dtype=np.uint8) * 5).reshape(6, 6)
    out["kat/resize_4x4_by2"] = O.resize(ramp4, 2, 2)
    out["kat/resize_6x6_by3"] = O.resize(ramp6, 2, 2)
    pc = project_case()
    levels = O.pyramid(pc["frame"], pc["subs"])
    locs, crcs, tr = [], [], []
    for f in pc["feats"]:
        proj, mask, loc = O.project(f["descriptor"], f["P"], f["B"], f["p3d"])
        locs.append(loc); crcs.append((crc(proj), crc(mask)))
        r = O.track(levels, pc["subs"], proj, mask, f["px"], f["py"], f["threshold"], f["stride"])
        tr.append((r["x"], r["y"], r["levels_run"], int(np.float32(r["response"]).view(np.uint32))))
    out["project/loc"] = np.array(locs, np.int32)
    out["project/crc"] = np.array(crcs, np.uint32)
    out["project/track"] = np.array(tr, np.int64)
    lc = line_case()
    levels = O.pyramid(lc["frame"], lc["subs"])
    lt, mc = [], []
    for j in lc["jobs"]:
        am = O.disc_mask(j["area"][2], j["area"][3], j["circles"])
        mc.append(crc(am))
        r = O.line_track(levels, lc["subs"], j["descriptor"], j["area"], am, j["threshold"], j["stride"])
        lt.append((r["x"], r["y"], r["levels_run"], int(np.float32(r["response"]).view(np.uint32))))
    out["line/mask_crc"] = np.array(mc, np.uint32)
    out["line/track"] = np.array(lt, np.int64)
    wc = window_case()
    wr = []
    for row in wc:
        o = O.ellipse_window(row[0], row[1], row[2:5], row[5], row[6])
        wr.append(o["area"] + tuple(int(np.float32(v).view(np.uint32)) for v in o["q"] + o["centre"]) + (crc(o["mask"]),))
    out["window/in"] = wc
    out["window/out"] = np.array(wr, np.int64)
    f = fast_frames()[0]
    srt = O.sort_keypoints(O.fast(f, 20, True))
    out["newpoints/sel"] = O.select_new_points(srt, np.array([[100, 100], [320, 180]], np.float32), 40, 30.0, 10, 10, f.shape[1], f.shape[0])
    np.savez_compressed(os.path.join(HERE, "golden_r01.npz"), **out)
    print("wrote", os.path.join(HERE, "golden_r01.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
