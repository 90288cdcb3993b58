// The reference's own tests of this path (tests/test_matching.cpp:13-117, driven by tests/tests.cpp:11-17)
// against the HIP implementation through the reference's C++ interface (slam_host.hpp).  The trial
// generator and the acceptance rule are the CPU oracle's restatement of that file (so_proptest_*);
// each result is additionally required to equal the oracle's result bit for bit.
//
//   test_matching_hip [trials]      exit status 0 = every test passed
#include <cstdio>
#include <cstdlib>
#include <cstring>

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
    } catch (const std::exception& e) {
        printf("exception: %s\n", e.what());
        return 2;
    }
    return failed ? 1 : 0;
}
