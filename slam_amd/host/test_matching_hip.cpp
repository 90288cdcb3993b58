// The reference's own tests of this path (tests/test_matching.cpp:13-117, driven by tests/tests.cpp:11-17)
// against the HIP implementation through the reference's C++ interface (slam_host.hpp).  The trial
// generator and the acceptance rule are the CPU oracle's restatement of that file (so_proptest_*);
// each result is additionally required to equal the oracle's result bit for bit.
//
//   test_matching_hip [trials]      exit status 0 = every test passed
#include <cmath>
#include <cstdio>
#include <exception>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <cmath>

#include "slam_host.hpp"
#include "slam_oracle.h"

using cv::Point2i;
using cv::Rect;

static bool test_matching0(bool useAreaMask, bool usePatchMask, int nTrials) {
    for (int iTrial = 0; iTrial < nTrials; ++iTrial) {
        so_proptest_case c;
        so_proptest_next(useAreaMask, usePatchMask, &c);
        matb im(c.h, c.w, c.im, (size_t)c.w), patch(c.ph, c.pw, c.patch, (size_t)c.pw);
        matb patchMask, areaMask;
        const matb *ppatchMask = NULL, *pareaMask = NULL;
        if (usePatchMask) { patchMask = matb(c.ph, c.pw, c.patch_mask, (size_t)c.pw); ppatchMask = &patchMask; }
        if (useAreaMask) { areaMask = matb(c.ah, c.aw, c.area_mask, (size_t)c.aw); pareaMask = &areaMask; }
        const Rect areaRect(c.ax, c.ay, c.aw, c.ah);

        // match!
        float response;
        const Point2i m = SLAM::matchFeatureInArea(im, patch, ppatchMask, areaRect, pareaMask, 1, response);

        bool ok = so_proptest_accept(&c, m.x, m.y, response) != 0;
        if (!ok) printf("trial %d: planted (%d,%d) found (%d,%d) response %g: FAIL\n", iTrial, c.xpatch, c.ypatch, m.x, m.y, response);
        so_match_out o;
        so_match_feature_in_area(c.im, c.w, c.h, c.w, c.patch, c.pw, c.ph, c.pw, c.patch_mask, c.pw, c.ax, c.ay, c.aw, c.ah,
                                 c.area_mask, c.aw, 1, 0, NULL, &o);
        if (o.x != m.x || o.y != m.y || std::memcmp(&o.response, &response, 4) != 0) {
            printf("trial %d: HIP (%d,%d,%.9g) differs from the oracle (%d,%d,%.9g)\n", iTrial, m.x, m.y, response, o.x, o.y, o.response);
            ok = false;
        }
        so_proptest_free(&c);
        if (!ok) return false;
    }
    return true;
}

// a pyramid + track + keypoints round through the same interface, checked against the oracle
static bool test_pyramid_track_keypoints() {
    const int w = 320, h = 240;
    matb im(h, w);
    unsigned s = 12345u;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            s = s * 1664525u + 1013904223u;
            const int blocky = ((x / 9) * 37 + (y / 7) * 101) & 0xff;
            im(y, x) = (byte)((blocky * 3 + (int)(s >> 27)) & 0xff);
        }
    std::vector<float> subs;
    subs.push_back(1.f); subs.push_back(2.f); subs.push_back(4.f);
    ImagePyramid<byte> pyramid(im, subs);
    if (pyramid.nSubs() != 3 || pyramid.images.size() != 3) return false;
    uint8_t* lev[3]; int lw[3], lh[3];
    std::vector<std::vector<uint8_t> > store(3);
    for (int i = 0; i < 3; i++) {
        so_pyramid_level_size(w, h, subs[i], &lw[i], &lh[i]);
        store[i].resize((size_t)lw[i] * lh[i]);
        lev[i] = store[i].data();
    }
    so_pyramid_build(im.data, w, h, (ptrdiff_t)im.step, 3, subs.data(), lev);
    for (int i = 0; i < 3; i++) {
        const matb& g = pyramid.images[i];
        if (g.cols != lw[i] || g.rows != lh[i]) { printf("level %d size\n", i); return false; }
        for (int y = 0; y < lh[i]; y++)
            if (std::memcmp(g.ptr(y), lev[i] + (size_t)y * lw[i], lw[i]) != 0) { printf("level %d row %d differs\n", i, y); return false; }
    }
    std::vector<SLAM::Projection> feats;
    for (int k = 0; k < 12; k++) {
        const int cx = 40 + 20 * k, cy = 50 + 11 * k;
        SLAM::Projection p;
        p.proj = im(Rect(cx - 10, cy - 10, 21, 21)).clone();
        p.projMask = matb(21, 21, (byte)1);
        p.projMask(0, 0) = 0; p.projMask(20, 20) = 0;
        p.p2d = cv::Point2f(cx + 2.25f, cy - 1.5f);
        feats.push_back(p);
    }
    std::vector<SLAM::Match> matches;
    std::vector<float> resp;
    SLAM::matchPoints(pyramid, feats, matches, 0.95f, 3, &resp);
    size_t nm = 0;
    for (size_t k = 0; k < feats.size(); k++) {
        so_track_out o;
        so_track(lev, lw, lh, subs.data(), 3, feats[k].proj.data, feats[k].projMask.data, 21, 21, feats[k].p2d.x, feats[k].p2d.y, 0.95f, 3, &o);
        float r1;
        const Point2i one = SLAM::Feature::track(pyramid, feats[k], 0.95f, 3, r1);
        if (one.x != o.x || one.y != o.y || std::memcmp(&r1, &o.response, 4) != 0 || std::memcmp(&resp[k], &o.response, 4) != 0) {
            printf("track %zu: HIP (%d,%d,%.9g) oracle (%d,%d,%.9g)\n", k, one.x, one.y, r1, o.x, o.y, o.response);
            return false;
        }
        if (o.response > 0.95f) {
            if (nm >= matches.size() || matches[nm].iFeature != (int)k || matches[nm].pos.x != (float)o.x || matches[nm].pos.y != (float)o.y) return false;
            nm++;
        }
    }
    if (nm != matches.size()) return false;
    SLAM::Projection empty;
    float re;
    const Point2i pe = SLAM::Feature::track(pyramid, empty, 0.95f, 3, re);
    if (pe.x != -1 || pe.y != -1 || re != -1.f) return false;     // feature.cpp:112-115

    // the same features with gate-ellipse windows on the coarsest level, against the oracle's windows and walk
    std::vector<cv::Vec3f> covs;
    for (size_t k = 0; k < feats.size(); k++) covs.push_back(cv::Vec3f(30.f + 9.f * k, 4.f * ((int)k % 5 - 2), 20.f + 3.f * k));
    covs[3] = cv::Vec3f(1.f, 2.f, 1.f);                            // not positive definite: no window
    std::vector<SLAM::Match> gmatches;
    std::vector<float> gresp;
    SLAM::matchPointsGated(pyramid, feats, covs, 9.f, gmatches, 0.95f, 3, &gresp);
    size_t gm = 0, gfound = 0;
    for (size_t k = 0; k < feats.size(); k++) {
        so_ellipse e;
        const float cov[3] = {covs[k][0], covs[k][1], covs[k][2]};
        so_ellipse_setup(feats[k].p2d.x, feats[k].p2d.y, cov, 9.f, 1.f / subs[2], &e);
        if (!e.valid) {
            if (k != 3 || gresp[k] != -1.f) { printf("gated %zu: expected no window\n", k); return false; }
            continue;
        }
        std::vector<uint8_t> am((size_t)e.w * e.h);
        so_ellipse_mask(&e, am.data());
        const int area[4] = {e.x, e.y, e.w, e.h};
        so_track_out o;
        so_track_ex(lev, lw, lh, subs.data(), 3, feats[k].proj.data, feats[k].projMask.data, 21, 21, feats[k].p2d.x, feats[k].p2d.y, 0.95f, 3,
                    area, am.data(), &o);
        if (std::memcmp(&gresp[k], &o.response, 4) != 0) { printf("gated %zu: HIP %.9g oracle %.9g\n", k, gresp[k], o.response); return false; }
        if (o.response > 0.95f) {
            if (gm >= gmatches.size() || gmatches[gm].iFeature != (int)k || gmatches[gm].pos.x != (float)o.x || gmatches[gm].pos.y != (float)o.y) return false;
            gm++;
            gfound++;
        }
    }
    if (gm != gmatches.size() || gfound < 3) { printf("gated: %zu matches, %zu expected, %zu found\n", gmatches.size(), gm, gfound); return false; }

    SLAM slam;
    std::vector<cv::KeyPoint> kps;
    slam.getSortedKeyPoints(im, 50, kps);
    std::vector<so_keypoint> okp((size_t)w * h / 4 + 1);
    float thr = 10.f;
    const int n = so_get_sorted_keypoints(im.data, w, h, (ptrdiff_t)im.step, 50, &thr, okp.data(), (int)okp.size());
    if ((int)kps.size() != n || slam.fastThreshold != thr) { printf("keypoints: %zu vs %d, thr %g vs %g\n", kps.size(), n, slam.fastThreshold, thr); return false; }
    for (int i = 0; i < n; i++)
        if (kps[i].pt.x != okp[i].x || kps[i].pt.y != okp[i].y || kps[i].response != okp[i].response || kps[i].size != 7.f) return false;

    // new points: the selection of addNewLines against the oracle, with the matches above as occupied positions
    std::vector<SLAM::Match> none;
    std::vector<matb> descr;
    const std::vector<int> picked = SLAM::selectNewPoints(im, kps, matches, none, 25, 14.f, 6, 6, &descr);
    std::vector<float> occ;
    for (size_t i = 0; i < matches.size(); i++) { occ.push_back(matches[i].pos.x); occ.push_back(matches[i].pos.y); }
    std::vector<int> osel(25);
    const int on = so_select_new_points(okp.data(), n, occ.data(), (int)matches.size(), 25, 14.f, 6, 6, w, h, osel.data());
    if ((int)picked.size() != on || descr.size() != picked.size() || on == 0) { printf("selectNewPoints: %zu vs %d\n", picked.size(), on); return false; }
    for (int i = 0; i < on; i++) {
        if (picked[i] != osel[i]) return false;
        uint8_t ref[13 * 13]; int ox, oy, ow, oh;
        so_extract_descriptor(im.data, w, h, (ptrdiff_t)im.step, okp[osel[i]].x, okp[osel[i]].y, 6, 6, ref, &ox, &oy, &ow, &oh);
        if (descr[i].cols != ow || descr[i].rows != oh) return false;
        for (int r = 0; r < oh; r++) if (std::memcmp(descr[i].ptr(r), ref + (size_t)r * ow, ow) != 0) return false;
    }
    return true;
}

// Feature::project and Feature::track with the reference's signatures (feature.cpp:53-173) against the oracle:
// a camera K [R | -R t] with a small rotation, planar features at depth z with slanted tangent directions.
static bool test_project_track() {
    const int w = 640, h = 480;
    matb im0(h, w), im(h, w);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            im0(y, x) = (byte)(((x / 9) * 37 + (y / 7) * 91 + ((x * 7 + y * 13) % 23)) & 0xFF);
            const int xs = x + 2, ys = y - 1;       // the current frame: the scene moved by (-2, +1)
            im(y, x) = (byte)(((xs / 9) * 37 + (ys / 7) * 91 + ((xs * 7 + ys * 13) % 23)) & 0xFF);
        }
    const float f = 600.f, ca = std::cos(0.03f), sa = std::sin(0.03f);
    const float K[3][3] = {{f, 0, 320}, {0, f, 240}, {0, 0, 1}}, R[3][3] = {{ca, 0, sa}, {0, 1, 0}, {-sa, 0, ca}};
    const float t[3] = {0.02f, -0.01f, 0.03f};
    matf P(3, 4), tm(3, 1);
    for (int r = 0; r < 3; r++) {
        float rt = 0;
        for (int c = 0; c < 3; c++) rt += R[r][c] * t[c];
        for (int c = 0; c < 4; c++) {
            float acc = 0;
            for (int k = 0; k < 3; k++) acc += K[r][k] * (c < 3 ? R[k][c] : 0.f);
            P(r, c) = acc;
        }
        tm(r, 0) = t[r];
        (void)rt;
    }
    for (int r = 0; r < 3; r++) {                    // P(:,3) = -K R t
        float acc = 0;
        for (int k = 0; k < 3; k++) acc += P(r, k) * t[k];
        P(r, 3) = -acc;
    }
    const SLAM::CameraState state(P, tm);
    std::vector<float> subs;
    subs.push_back(1.f); subs.push_back(3.f);        // matching.cpp:243-246
    ImagePyramid<byte> pyramid(im, subs);
    uint8_t* lev[2]; int lw[2], lh[2];
    std::vector<std::vector<uint8_t> > store(2);
    for (int i = 0; i < 2; i++) {
        so_pyramid_level_size(w, h, subs[i], &lw[i], &lh[i]);
        store[i].resize((size_t)lw[i] * lh[i]);
        lev[i] = store[i].data();
    }
    so_pyramid_build(im.data, w, h, (ptrdiff_t)im.step, 2, subs.data(), lev);
    std::vector<SLAM::Feature> feats;
    std::vector<matf> p3ds;
    float Pf[12];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) Pf[r * 4 + c] = P(r, c);
    for (int k = 0; k < 14; k++) {
        const int cx = 60 + 38 * k, cy = 70 + 24 * k, half = 13;
        const float z = 2.f + 0.3f * k, px = z / f;
        SLAM::Feature ft;
        ft.iKalman = k;
        ft.descriptor = im0(Rect(cx - half, cy - half, 2 * half + 1, 2 * half + 1)).clone();
        ft.B = matf(4, 3, 0.f);
        ft.B(0, 0) = px * (1.f + 0.02f * k); ft.B(2, 0) = 0.15f * px;       // slanted in depth
        ft.B(1, 1) = px * (1.f - 0.01f * k); ft.B(0, 1) = 0.05f * px; ft.B(2, 1) = -0.1f * px;
        ft.B(3, 2) = 1.f;
        // p3d such that K R (p3d - t) = z (cx, cy, 1): p3d = t + R^T K^-1 (cx, cy, 1) z
        const float ray[3] = {(cx - 320.f) / f * z, (cy - 240.f) / f * z, z};
        matf p3(3, 1);
        for (int r = 0; r < 3; r++) p3(r, 0) = t[r] + R[0][r] * ray[0] + R[1][r] * ray[1] + R[2][r] * ray[2];
        feats.push_back(ft);
        p3ds.push_back(p3);
    }
    feats.push_back(SLAM::Feature());                 // no descriptor: empty projection (feature.cpp:56-59)
    feats.back().B = matf(4, 3, 0.f);
    p3ds.push_back(p3ds[0]);
    std::vector<SLAM::Match> matches;
    std::vector<float> resp;
    SLAM::matchPoints(pyramid, state, feats, p3ds, matches, 0.9f, 3, &resp);
    {   // the reference's member call (SLAM.hpp:106-107) on the same features: it builds the pyramid {1, 3} itself
        SLAM slam;
        slam.features = feats;
        slam.kalman.points = p3ds;                        // iKalman = k above
        slam.features.back().iKalman = (int)p3ds.size() - 1;
        std::vector<SLAM::Match> got;
        got.push_back(SLAM::Match(cv::Point2f(-5.f, -5.f), -1));     // matches are appended, not replaced (matching.cpp:255)
        slam.matchPoints(im, state, got, 0.9f);
        if (got.size() != matches.size() + 1 || got[0].iFeature != -1 || matches.empty()) {
            printf("member matchPoints: %zu matches, expected %zu + 1\n", got.size(), matches.size());
            return false;
        }
        for (size_t k = 0; k < matches.size(); k++)
            if (got[k + 1].iFeature != matches[k].iFeature || got[k + 1].pos.x != matches[k].pos.x || got[k + 1].pos.y != matches[k].pos.y) {
                printf("member matchPoints: match %zu differs\n", k);
                return false;
            }
    }
    size_t nm = 0, tracked = 0;
    for (size_t k = 0; k < feats.size(); k++) {
        const SLAM::Feature& ft = feats[k];
        matb mask; Rect loc;
        const matb proj = ft.project(state, p3ds[k], mask, loc);
        float r1 = 0;
        const Point2i one = ft.track(pyramid, state, p3ds[k], 0.9f, 3, r1);
        if (ft.descriptor.empty()) {
            if (!proj.empty() || one.x != -1 || one.y != -1 || r1 != -1.f || resp[k] != -1.f) { printf("empty feature %zu\n", k); return false; }
            continue;
        }
        float Bf[12], pf[3];
        for (int r = 0; r < 4; r++) for (int c = 0; c < 3; c++) Bf[r * 3 + c] = ft.B(r, c);
        for (int r = 0; r < 3; r++) pf[r] = p3ds[k](r, 0);
        std::vector<uint8_t> op(250000), om(250000);
        so_projection sp;
        so_project(ft.descriptor.data, ft.descriptor.cols, ft.descriptor.rows, (ptrdiff_t)ft.descriptor.step, Pf, Bf, pf, op.data(), om.data(), op.size(), &sp);
        if (!sp.valid || proj.cols != sp.w || proj.rows != sp.h || loc.x != sp.x || loc.y != sp.y || loc.width != sp.w || loc.height != sp.h) {
            printf("project %zu: location (%d,%d,%d,%d) oracle (%d,%d,%d,%d) valid %d\n", k, loc.x, loc.y, loc.width, loc.height, sp.x, sp.y, sp.w, sp.h, sp.valid);
            return false;
        }
        for (int y = 0; y < sp.h; y++)
            if (std::memcmp(proj.ptr(y), op.data() + (size_t)y * sp.w, sp.w) != 0 || std::memcmp(mask.ptr(y), om.data() + (size_t)y * sp.w, sp.w) != 0) {
                printf("project %zu: row %d differs\n", k, y);
                return false;
            }
        const matf p2d = state.project(p3ds[k]);
        so_track_out o;
        so_track(lev, lw, lh, subs.data(), 2, op.data(), om.data(), sp.w, sp.h, p2d(0, 0), p2d(1, 0), 0.9f, 3, &o);
        if (one.x != o.x || one.y != o.y || std::memcmp(&r1, &o.response, 4) != 0 || std::memcmp(&resp[k], &o.response, 4) != 0) {
            printf("track %zu: HIP (%d,%d,%.9g) oracle (%d,%d,%.9g)\n", k, one.x, one.y, r1, o.x, o.y, o.response);
            return false;
        }
        tracked += o.levels_run == 2;
        if (o.response > 0.9f) {
            if (nm >= matches.size() || matches[nm].iFeature != (int)k || matches[nm].pos.x != (float)o.x || matches[nm].pos.y != (float)o.y) return false;
            nm++;
        }
    }
    if (nm != matches.size() || tracked < 7) { printf("matches %zu of %zu, %zu walked both levels\n", nm, matches.size(), tracked); return false; }
    return true;
}

// SLAM::matchLines' image work (lineFeature.cpp:86-165) against the oracle: pyramid {1,2}, stride 2 (matching.cpp:273-277).
static bool test_match_lines() {
    const int w = 640, h = 480;
    matb im0(h, w), im(h, w);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            im0(y, x) = (byte)(((x / 11) * 53 + (y / 6) * 29 + ((x * 5 + y * 3) % 19)) & 0xFF);
            const int xs = x - 3, ys = y + 2;
            im(y, x) = (byte)(((xs / 11) * 53 + (ys / 6) * 29 + ((xs * 5 + ys * 3) % 19)) & 0xFF);
        }
    std::vector<float> subs;
    subs.push_back(1.f); subs.push_back(2.f);
    ImagePyramid<byte> pyramid(im, subs);
    uint8_t* lev[2]; int lw[2], lh[2];
    std::vector<std::vector<uint8_t> > store(2);
    for (int i = 0; i < 2; i++) {
        so_pyramid_level_size(w, h, subs[i], &lw[i], &lh[i]);
        store[i].resize((size_t)lw[i] * lh[i]);
        lev[i] = store[i].data();
    }
    so_pyramid_build(im.data, w, h, (ptrdiff_t)im.step, 2, subs.data(), lev);
    std::vector<SLAM::LineSearch> lines;
    for (int k = 0; k < 10; k++) {
        const int cx = 80 + 50 * k, cy = 90 + 30 * k, half = 20;          // the descriptor of new_image.cpp:80-82 is 81x81; 41x41 here
        SLAM::LineSearch l;
        l.descriptor = im0(Rect(cx - half, cy - half, 2 * half + 1, 2 * half + 1)).clone();
        const int tx = (cx + 3) / 2, ty = (cy - 2) / 2;                    // true position on the coarsest level
        l.areaRect = Rect(tx - 14, ty - 9, 30, 20);
        for (int b = 0; b < 5; b++) { l.binCenters.push_back(Point2i(4 + 5 * b + (k % 3), 5 + 2 * b)); l.binRadii.push_back(2 + (b + k) % 4); }
        lines.push_back(l);
    }
    lines[9].areaRect = Rect(-8, -6, 30, 20);                              // hangs over the image corner
    std::vector<SLAM::Match> matches;
    std::vector<float> resp;
    SLAM::matchLines(pyramid, lines, matches, 0.9f, 2, &resp);
    size_t nm = 0, both = 0;
    for (size_t k = 0; k < lines.size(); k++) {
        const SLAM::LineSearch& l = lines[k];
        std::vector<int> circ;
        for (size_t b = 0; b < l.binCenters.size(); b++) { circ.push_back(l.binCenters[b].x); circ.push_back(l.binCenters[b].y); circ.push_back(l.binRadii[b]); }
        std::vector<uint8_t> mask((size_t)l.areaRect.width * l.areaRect.height);
        so_disc_mask(l.areaRect.width, l.areaRect.height, circ.data(), (int)l.binRadii.size(), mask.data());
        const matb gm = SLAM::lineAreaMask(l);
        for (int y = 0; y < gm.rows; y++)
            if (std::memcmp(gm.ptr(y), mask.data() + (size_t)y * gm.cols, gm.cols) != 0) { printf("line %zu: mask row %d differs\n", k, y); return false; }
        so_track_out o;
        so_line_track(lev, lw, lh, subs.data(), 2, l.descriptor.data, l.descriptor.cols, l.descriptor.rows, l.areaRect.x, l.areaRect.y,
                      l.areaRect.width, l.areaRect.height, mask.data(), 0.9f, 2, &o);
        const bool kept = nm < matches.size() && matches[nm].iFeature == (int)k;
        if (std::memcmp(&resp[k], &o.response, 4) != 0 || kept != (o.response > 0.9f) ||
            (kept && (matches[nm].pos.x != (float)o.x || matches[nm].pos.y != (float)o.y))) {
            printf("line %zu: HIP response %.9g kept %d, oracle (%d,%d,%.9g)\n", k, resp[k], (int)kept, o.x, o.y, o.response);
            return false;
        }
        nm += kept;
        both += o.levels_run == 2;
    }
    if (nm != matches.size() || both < 5) { printf("lines: %zu matches, %zu refined\n", nm, both); return false; }
    return true;
}

// LineFeature::track's own calls (lineFeature.cpp:125-127 and :154-156) made on pyramid.images[i], as the reference's
// unchanged CPU code makes them: they must run on the levels that are resident on the device (no upload of the frame)
// and return what the oracle returns.  Also prints the per-call cost of that route next to the route a foreign image takes.
static bool test_line_calls_on_resident_levels() {
    const int w = 640, h = 480;
    matb im0(h, w), im(h, w);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            im0(y, x) = (byte)(((x / 11) * 53 + (y / 6) * 29 + ((x * 5 + y * 3) % 19)) & 0xFF);
            const int xs = x - 3, ys = y + 2;
            im(y, x) = (byte)(((xs / 11) * 53 + (ys / 6) * 29 + ((xs * 5 + ys * 3) % 19)) & 0xFF);
        }
    std::vector<float> subs;
    subs.push_back(1.f); subs.push_back(2.f);
    ImagePyramid<byte> pyramid(im, subs);
    uint8_t* lev[2]; int lw[2], lh[2];
    std::vector<std::vector<uint8_t> > store(2);
    for (int i = 0; i < 2; i++) {
        so_pyramid_level_size(w, h, subs[i], &lw[i], &lh[i]);
        store[i].resize((size_t)lw[i] * lh[i]);
        lev[i] = store[i].data();
    }
    so_pyramid_build(im.data, w, h, (ptrdiff_t)im.step, 2, subs.data(), lev);
    if (pyramid.levelOf(pyramid.images[1]) != 1 || pyramid.levelOf(pyramid.images[1].clone()) != -1) { printf("levelOf\n"); return false; }
    const int stride = 2;
    for (int k = 0; k < 8; k++) {
        const int cx = 90 + 55 * k, cy = 100 + 32 * k, half = 20;
        const matb descr = im0(Rect(cx - half, cy - half, 2 * half + 1, 2 * half + 1)).clone();
        // lineFeature.cpp:124: descriptor resized to the coarsest level (cv::resize, done by the caller on the CPU)
        const int cw = (int)std::lround(descr.cols * 0.5), ch = (int)std::lround(descr.rows * 0.5);
        matb cdescr(ch, cw);
        so_resize_u8_linear(descr.data, descr.cols, descr.rows, (ptrdiff_t)descr.step, cdescr.data, cw, ch, (ptrdiff_t)cdescr.step);
        const int tx = (cx + 3) / 2, ty = (cy - 2) / 2;
        const Rect area(k == 7 ? -9 : tx - 14, k == 7 ? -7 : ty - 9, 30, 20);       // the last one hangs over the image corner
        matb amask(area.height, area.width);
        std::vector<int> circ;
        for (int b = 0; b < 5; b++) { circ.push_back(4 + 5 * b + (k % 3)); circ.push_back(5 + 2 * b); circ.push_back(2 + (b + k) % 4); }
        so_disc_mask(area.width, area.height, circ.data(), 5, amask.data);
        float r1 = 0.f;
        const Point2i p1 = SLAM::matchFeatureInArea(pyramid.images[1], cdescr, NULL, area, &amask, stride, r1);            // lineFeature.cpp:125-127
        so_match_out o1;
        so_match_feature_in_area(lev[1], lw[1], lh[1], lw[1], cdescr.data, cw, ch, (ptrdiff_t)cdescr.step, NULL, 0, area.x, area.y, area.width,
                                 area.height, amask.data, (ptrdiff_t)amask.step, stride, 0, NULL, &o1);
        if (p1.x != o1.x || p1.y != o1.y || std::memcmp(&r1, &o1.response, 4) != 0) {
            printf("coarse call %d: HIP (%d,%d,%.9g) oracle (%d,%d,%.9g)\n", k, p1.x, p1.y, r1, o1.x, o1.y, o1.response);
            return false;
        }
        const Rect fine(2 * p1.x - 2, 2 * p1.y - 2, 5, 5);                                                                  // lineFeature.cpp:150-152
        float r0 = 0.f;
        const Point2i p0 = SLAM::matchFeatureInArea(pyramid.images[0], descr, NULL, fine, NULL, 1, r0);                      // lineFeature.cpp:154-156
        so_match_out o0;
        so_match_feature_in_area(lev[0], lw[0], lh[0], lw[0], descr.data, descr.cols, descr.rows, (ptrdiff_t)descr.step, NULL, 0, fine.x, fine.y, 5, 5,
                                 NULL, 0, 1, 0, NULL, &o0);
        if (p0.x != o0.x || p0.y != o0.y || std::memcmp(&r0, &o0.response, 4) != 0) {
            printf("fine call %d: HIP (%d,%d,%.9g) oracle (%d,%d,%.9g)\n", k, p0.x, p0.y, r0, o0.x, o0.y, o0.response);
            return false;
        }
    }
    // per-call cost: the resident level against an equal image that belongs to no pyramid (uploaded on every call)
    const matb foreign = pyramid.images[0].clone();
    const matb descr = im0(Rect(300, 200, 41, 41)).clone();
    const Rect win(301, 196, 5, 5);
    float r = 0.f, rf = 0.f;
    const int reps = 200;
    struct timespec t0, t1, t2;
    SLAM::matchFeatureInArea(pyramid.images[0], descr, NULL, win, NULL, 1, r);
    SLAM::matchFeatureInArea(foreign, descr, NULL, win, NULL, 1, rf);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < reps; i++) SLAM::matchFeatureInArea(pyramid.images[0], descr, NULL, win, NULL, 1, r);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    for (int i = 0; i < reps; i++) SLAM::matchFeatureInArea(foreign, descr, NULL, win, NULL, 1, rf);
    clock_gettime(CLOCK_MONOTONIC, &t2);
    const double us_res = ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / reps / 1e3;
    const double us_for = ((t2.tv_sec - t1.tv_sec) * 1e9 + (t2.tv_nsec - t1.tv_nsec)) / reps / 1e3;
    printf("matchFeatureInArea per call, 640x480 level, 41x41 patch, 5x5 window: resident level %.1f us, foreign image %.1f us\n", us_res, us_for);
    return std::memcmp(&r, &rf, 4) == 0;
}

// The mirror never throws: a rejected argument gives the call's sentinel and leaves its code in slamhip::last_error().
static bool test_errors_give_sentinels() {
    bool ok = slamhip::last_error().code == SLAM_HIP_OK;         // none of the tests before this one failed silently
    matb im(60, 80), patch(7, 7), pm(7, 7);
    for (int i = 0; i < im.rows * im.cols; i++) im.data[i] = (byte)(i * 7);
    for (int i = 0; i < 49; i++) { patch.data[i] = (byte)(i * 5); pm.data[i] = 1; }
    pm(3, 3) = 2;                                                // the mask contract is {0, 1}: the library rejects it
    float response = 0.5f;
    const cv::Point2i p = SLAM::matchFeatureInArea(im, patch, &pm, cv::Rect(10, 10, 20, 20), nullptr, 1, response);
    ok = ok && p.x == 0 && p.y == 0 && response == -1.f && slamhip::last_error().code == SLAM_HIP_EINVAL &&
         !slamhip::last_error().message.empty();
    slamhip::clear_error();
    std::vector<float> subs(1, 1.f);
    ImagePyramid<byte> pyr(im, subs);
    ok = ok && pyr.ok();
    SLAM::Projection pr;
    pr.proj = patch; pr.projMask = matb(5, 5); pr.p2d = cv::Point2f(30.f, 30.f);
    response = 0.5f;
    const cv::Point2i q = SLAM::Feature::track(pyr, pr, 0.9f, 1, response);
    ok = ok && q.x == -1 && q.y == -1 && response == -1.f && slamhip::last_error().code == SLAM_HIP_EINVAL;
    slamhip::clear_error();
    std::vector<SLAM::Match> matches(1, SLAM::Match(cv::Point2f(1.f, 1.f), 0));
    std::vector<float> responses;
    SLAM::matchPoints(pyr, std::vector<SLAM::Projection>(3, pr), matches, 0.9f, 1, &responses);
    ok = ok && matches.empty() && responses.size() == 3 && responses[0] == -1.f && slamhip::last_error().code == SLAM_HIP_EINVAL;
    slamhip::clear_error();
    return ok && slamhip::last_error().code == SLAM_HIP_OK;
}

int main(int argc, char* argv[]) {
    const int nTrials = argc > 1 ? std::atoi(argv[1]) : 100;
    int failed = 0;
#define DO_TEST(name, call) do { const bool ok_ = (call); printf("%-28s %s\n", name, ok_ ? "passed" : "FAILED"); fflush(stdout); failed += !ok_; } while (0)
    try {
        // one rand() stream across the four tests, as tests/tests.cpp runs them in one process
        DO_TEST("test_matching", test_matching0(false, false, nTrials));
        DO_TEST("test_matchingArea", test_matching0(true, false, nTrials));
        DO_TEST("test_matchingPatch", test_matching0(false, true, nTrials));
        DO_TEST("test_matchingPatchArea", test_matching0(true, true, nTrials));
        DO_TEST("test_pyramid_track_keypoints", test_pyramid_track_keypoints());
        DO_TEST("test_project_track", test_project_track());
        DO_TEST("test_match_lines", test_match_lines());
        DO_TEST("test_line_calls_resident", test_line_calls_on_resident_levels());
        DO_TEST("test_errors_give_sentinels", test_errors_give_sentinels());
    } catch (const std::exception& e) {
        printf("exception: %s\n", e.what());
        return 2;
    }
    return failed ? 1 : 0;
}
