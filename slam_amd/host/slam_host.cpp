// Implementation of slam_host.hpp over the C ABI (no HIP headers needed: plain C calls).
#include "slam_host.hpp"

#include <cstdlib>
#include <cstring>

namespace slamhip {

// One context per host thread, created on first use and deliberately never destroyed: at process
// exit the HIP runtime may already be shutting down when thread-local destructors run.
slam_ctx* context() {
    static thread_local slam_ctx* ctx = nullptr;
    if (!ctx) {
        const char* d = std::getenv("SLAM_HIP_DEVICE");
        const int dev = d ? std::atoi(d) : 0;
        slam_ctx* c = nullptr;
        const int rc = slam_hip_create(dev, &c);
        if (rc != SLAM_HIP_OK)
            throw std::runtime_error(std::string("slam_hip_create: ") + slam_hip_last_error_string(nullptr));
        ctx = c;
    }
    return ctx;
}

void check(int rc, const char* what) {
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + slam_hip_last_error_string(context()));
}

}  // namespace slamhip

ImagePyramid<byte>::ImagePyramid(const mati& imFullSize, const std::vector<float>& subs) : subsamples(subs) {
    slam_ctx* ctx = slamhip::context();
    const int w = imFullSize.cols, h = imFullSize.rows;
    slamhip::check(slam_hip_pyramid_create(ctx, w, h, (int)subs.size(), subs.data(), 1, &pyr_), "ImagePyramid");
    int rc = slam_hip_pyramid_build(ctx, pyr_, imFullSize.data, imFullSize.step, 0);
    for (size_t i = 0; rc == SLAM_HIP_OK && i < subs.size(); i++) {
        slam_level_desc d;
        rc = slam_hip_pyramid_level(pyr_, (int)i, &d);
        if (rc != SLAM_HIP_OK) break;
        mati lvl(d.h, d.w);
        rc = slam_hip_pyramid_download(ctx, pyr_, (int)i, 0, lvl.data, lvl.step);
        images.push_back(lvl);
    }
    if (rc != SLAM_HIP_OK) {
        slam_hip_pyramid_destroy(ctx, pyr_);
        pyr_ = nullptr;
        slamhip::check(rc, "ImagePyramid");
    }
}

ImagePyramid<byte>::~ImagePyramid() {
    if (pyr_) slam_hip_pyramid_destroy(slamhip::context(), pyr_);
}

cv::Point2i SLAM::matchFeatureInArea(const cv::Mat_<byte>& im, const cv::Mat_<byte>& patch,
                                     const cv::Mat_<byte>* patchMask, const cv::Rect& areaRect,
                                     const matb* areaMask, int stride, float& response, bool useExactAreaMask) {
    slam_match_result r;
    slamhip::check(slam_hip_match_feature_in_area(
                       slamhip::context(), im.data, im.cols, im.rows, im.step, patch.data, patch.cols, patch.rows,
                       patch.step, patchMask ? patchMask->data : nullptr, patchMask ? patchMask->step : 0,
                       areaRect.x, areaRect.y, areaRect.width, areaRect.height,
                       areaMask ? areaMask->data : nullptr, areaMask ? areaMask->step : 0, stride,
                       useExactAreaMask ? 1 : 0, nullptr, &r),
                   "matchFeatureInArea");
    response = r.response;
    return cv::Point2i(r.x, r.y);
}

namespace {

// Packs templates and masks row by row into one blob and fills the job table.
void pack_jobs(const std::vector<SLAM::Projection>& f, float threshold, int stride,
               std::vector<slam_track_job>& jobs, std::vector<uint8_t>& blob) {
    size_t total = 0;
    for (const auto& p : f) total += 2 * (size_t)p.proj.rows * p.proj.cols;
    blob.resize(total ? total : 1);
    jobs.resize(f.size());
    size_t off = 0;
    for (size_t i = 0; i < f.size(); i++) {
        const SLAM::Projection& p = f[i];
        const int pw = p.proj.cols, ph = p.proj.rows;
        if (pw > 0 && ph > 0 && (p.projMask.cols != pw || p.projMask.rows != ph))
            throw std::invalid_argument("Projection: proj and projMask differ in size");
        slam_track_job& j = jobs[i];
        std::memset(&j, 0, sizeof(j));
        j.frame = 0; j.projw = pw; j.projh = ph;
        j.proj_off = (int64_t)off;
        for (int r = 0; r < ph; r++) std::memcpy(&blob[off + (size_t)r * pw], p.proj.ptr(r), pw);
        off += (size_t)pw * ph;
        j.mask_off = (int64_t)off;
        for (int r = 0; r < ph; r++) std::memcpy(&blob[off + (size_t)r * pw], p.projMask.ptr(r), pw);
        off += (size_t)pw * ph;
        j.px = p.p2d.x; j.py = p.p2d.y; j.threshold = threshold; j.stride = stride;
    }
}

}  // namespace

cv::Point2i SLAM::Feature::track(const ImagePyramid<byte>& pyramid, const Projection& p, float threshold, int stride,
                                 float& response) {
    std::vector<slam_track_job> jobs;
    std::vector<uint8_t> blob;
    pack_jobs(std::vector<Projection>(1, p), threshold, stride, jobs, blob);
    slam_track_result r;
    slamhip::check(slam_hip_track_batch(slamhip::context(), pyramid.device(), jobs.data(), 1, blob.data(), blob.size(), &r),
                   "Feature::track");
    if (r.status != SLAM_HIP_OK) throw std::invalid_argument("Feature::track: job rejected by the device");
    response = r.response;
    return cv::Point2i(r.x, r.y);
}

void SLAM::matchPoints(const ImagePyramid<byte>& pyramid, const std::vector<Projection>& features,
                       std::vector<Match>& matches, float threshold, int stride, std::vector<float>* responses) {
    matches.clear();
    if (responses) responses->assign(features.size(), -1.f);
    if (features.empty()) return;
    std::vector<slam_track_job> jobs;
    std::vector<uint8_t> blob;
    pack_jobs(features, threshold, stride, jobs, blob);
    std::vector<slam_track_result> res(features.size());
    slamhip::check(slam_hip_track_batch(slamhip::context(), pyramid.device(), jobs.data(), (int)jobs.size(), blob.data(),
                                        blob.size(), res.data()),
                   "matchPoints");
    for (size_t i = 0; i < res.size(); i++) {
        if (res[i].status != SLAM_HIP_OK) throw std::invalid_argument("matchPoints: job rejected by the device");
        if (responses) (*responses)[i] = res[i].response;
        if (res[i].response > threshold)      // matching.cpp:262
            matches.push_back(Match(cv::Point2f((float)res[i].x, (float)res[i].y), (int)i));
    }
}

std::vector<int> SLAM::selectNewPoints(const matb& im, const std::vector<cv::KeyPoint>& keypoints,
                                       const std::vector<Match>& matches, const std::vector<Match>& lineMatches,
                                       int n, float minDist, int rx, int ry, std::vector<matb>* descriptors) {
    std::vector<int> out;
    if (descriptors) descriptors->clear();
    if (n <= 0 || keypoints.empty()) return out;
    std::vector<slam_keypoint> kps(keypoints.size());
    for (size_t i = 0; i < kps.size(); i++) { kps[i].x = keypoints[i].pt.x; kps[i].y = keypoints[i].pt.y; kps[i].response = keypoints[i].response; }
    std::vector<float> occ;
    for (const Match& m : matches) { occ.push_back(m.pos.x); occ.push_back(m.pos.y); }          // new_point.cpp:139-141
    for (const Match& m : lineMatches) { occ.push_back(m.pos.x); occ.push_back(m.pos.y); }      // new_point.cpp:142-144
    std::vector<int32_t> sel((size_t)n);
    int32_t nsel = 0;
    const size_t one = (size_t)(2 * rx + 1) * (2 * ry + 1);
    std::vector<uint8_t> desc(descriptors ? one * (size_t)n : 0);
    std::vector<slam_desc_result> res(descriptors ? (size_t)n : 0);
    slamhip::check(slam_hip_add_new_points(slamhip::context(), im.data, im.cols, im.rows, im.step, kps.data(), (int)kps.size(), 1,
                                           occ.empty() ? nullptr : occ.data(), (int)(occ.size() / 2), n, minDist, rx, ry,
                                           sel.data(), &nsel, nullptr, descriptors ? desc.data() : nullptr, desc.size(),
                                           descriptors ? res.data() : nullptr),
                   "selectNewPoints");
    out.assign(sel.begin(), sel.begin() + nsel);
    if (descriptors)
        for (int i = 0; i < nsel; i++) {
            matb d(res[i].h, res[i].w);
            for (int r = 0; r < res[i].h; r++) std::memcpy(d.ptr(r), &desc[one * (size_t)i + (size_t)r * res[i].w], (size_t)res[i].w);
            descriptors->push_back(d);
        }
    return out;
}

void SLAM::getSortedKeyPoints(const matb& im, size_t nMinPts, std::vector<cv::KeyPoint>& out) const {
    slam_ctx* ctx = slamhip::context();
    int cap = 1 << 16;
    std::vector<slam_keypoint> kps;
    for (;;) {
        kps.resize(cap);
        int32_t n = 0;
        float thr = fastThreshold;
        const int rc = slam_hip_get_sorted_keypoints(ctx, im.data, im.cols, im.rows, im.step, nMinPts, &thr, kps.data(), cap, &n);
        slamhip::check(rc, "getSortedKeyPoints");
        if (n > cap) { cap = n; continue; }   // the list was cut at cap: run again with room for all
        fastThreshold = thr;
        out.clear();
        out.reserve(n);
        for (int i = 0; i < n && i < cap; i++)
            out.push_back(cv::KeyPoint(kps[i].x, kps[i].y, 7.f, -1.f, kps[i].response, 0, -1));
        return;
    }
}
