// Implementation of slam_host.hpp over the C ABI (no HIP headers needed: plain C calls).
#include "slam_host.hpp"

#include <cstdlib>
#include <cstring>

namespace slamhip {

// One context per host thread, created on first use and deliberately never destroyed: at process
// exit the HIP runtime may already be shutting down when thread-local destructors run.
namespace {
Error& error_slot() {
    static thread_local Error e = {SLAM_HIP_OK, std::string()};
    return e;
}
}  // namespace

const Error& last_error() { return error_slot(); }
void clear_error() { error_slot().code = SLAM_HIP_OK; error_slot().message.clear(); }

bool failed(int rc, const char* what, const char* detail) {
    if (rc >= 0) return false;
    Error& e = error_slot();
    e.code = rc;
    e.message = std::string(what) + ": " + (detail ? detail : "");
    return true;
}

// nullptr when no context can be made (no device, out of memory): every ABI call then answers SLAM_HIP_EINVAL, which
// the callers below turn into their sentinel results.
slam_ctx* context() {
    static thread_local slam_ctx* ctx = nullptr;
    if (!ctx) {
        const char* d = std::getenv("SLAM_HIP_DEVICE");
        const int dev = d ? std::atoi(d) : 0;
        slam_ctx* c = nullptr;
        const int rc = slam_hip_create(dev, &c);
        if (failed(rc, "slam_hip_create", slam_hip_last_error_string(nullptr))) return nullptr;
        ctx = c;
    }
    return ctx;
}

bool failed(int rc, const char* what) {
    if (rc >= 0) return false;
    slam_ctx* c = context();
    return failed(rc, what, c ? slam_hip_last_error_string(c) : "no device context");
}

// Live ImagePyramid mirrors of the calling thread: matchFeatureInArea looks an image up here, so that a search on a
// pyramid level (lineFeature.cpp:125,154; feature.cpp:147 pass pyramid.images[i]) runs on the level that is already
// resident on the device instead of uploading it again.
std::vector<const ImagePyramid<byte>*>& live_pyramids() {
    static thread_local std::vector<const ImagePyramid<byte>*> v;
    return v;
}

}  // namespace slamhip

ImagePyramid<byte>::ImagePyramid(const mati& imFullSize, const std::vector<float>& subs) : subsamples(subs) {
    slam_ctx* ctx = slamhip::context();
    const int w = imFullSize.cols, h = imFullSize.rows;
    if (slamhip::failed(slam_hip_pyramid_create(ctx, w, h, (int)subs.size(), subs.data(), 1, &pyr_), "ImagePyramid")) {
        pyr_ = nullptr;                 // ok() is false, images stays empty
        return;
    }
    int rc = slam_hip_pyramid_build(ctx, pyr_, imFullSize.data, imFullSize.step, 0);
    for (size_t i = 0; rc == SLAM_HIP_OK && i < subs.size(); i++) {
        slam_level_desc d;
        rc = slam_hip_pyramid_level(pyr_, (int)i, &d);
        if (rc != SLAM_HIP_OK) break;
        mati lvl(d.h, d.w);
        rc = slam_hip_pyramid_download(ctx, pyr_, (int)i, 0, lvl.data, lvl.step);
        images.push_back(lvl);
    }
    if (slamhip::failed(rc, "ImagePyramid")) {
        slam_hip_pyramid_destroy(ctx, pyr_);
        pyr_ = nullptr;
        images.clear();
        return;
    }
    slamhip::live_pyramids().push_back(this);
}

ImagePyramid<byte>::~ImagePyramid() {
    std::vector<const ImagePyramid<byte>*>& v = slamhip::live_pyramids();
    for (size_t i = 0; i < v.size(); i++)
        if (v[i] == this) { v.erase(v.begin() + (long)i); break; }
    if (pyr_) slam_hip_pyramid_destroy(slamhip::context(), pyr_);
}

int ImagePyramid<byte>::levelOf(const mati& im) const {
    for (size_t i = 0; i < images.size(); i++)
        if (images[i].data == im.data && images[i].cols == im.cols && images[i].rows == im.rows && images[i].step == im.step)
            return (int)i;
    return -1;
}

cv::Point2i SLAM::matchFeatureInArea(const cv::Mat_<byte>& im, const cv::Mat_<byte>& patch,
                                     const cv::Mat_<byte>* patchMask, const cv::Rect& areaRect,
                                     const matb* areaMask, int stride, float& response, bool useExactAreaMask) {
    slam_match_result r;
    // A whole level of a live pyramid (same buffer, same shape)?  Then search the resident level: the call uploads
    // the patch and the masks only.  The levels are snapshots taken when the pyramid was built; nothing in the
    // reference writes to them afterwards (imagePyramid.hpp:15-30), and writes to `images` would not be seen here.
    // A caller that edits `pyramid.images[i]` in place (the member is public and writable, imagePyramid.hpp:12) sets
    // SLAM_HIP_NO_RESIDENT_LEVELS=1: every call then uploads the image it is given, as the reference would read it.
    static const bool no_resident = [] { const char* e = std::getenv("SLAM_HIP_NO_RESIDENT_LEVELS"); return e && e[0] == '1'; }();
    const std::vector<const ImagePyramid<byte>*>& live = slamhip::live_pyramids();
    for (size_t k = 0; k < live.size() && !no_resident; k++) {
        const int level = live[k]->levelOf(im);
        if (level < 0) continue;
        if (patch.cols < 1 || patch.rows < 1 || stride < 1) break;          // let the generic entry point report it
        const size_t pb = (size_t)patch.cols * patch.rows;
        const size_t ab = areaMask ? (size_t)(areaRect.width > 0 ? areaRect.width : 0) * (size_t)(areaRect.height > 0 ? areaRect.height : 0) : 0;
        if (areaMask && (areaMask->cols != areaRect.width || areaMask->rows != areaRect.height)) break;
        if (patchMask && (patchMask->cols != patch.cols || patchMask->rows != patch.rows)) break;
        std::vector<uint8_t> blob(pb * (patchMask ? 2 : 1) + ab + 1);
        for (int y = 0; y < patch.rows; y++) std::memcpy(&blob[(size_t)y * patch.cols], patch.ptr(y), (size_t)patch.cols);
        if (patchMask) for (int y = 0; y < patch.rows; y++) std::memcpy(&blob[pb + (size_t)y * patch.cols], patchMask->ptr(y), (size_t)patch.cols);
        const size_t aoff = pb * (patchMask ? 2 : 1);
        if (areaMask) for (int y = 0; y < areaMask->rows; y++) std::memcpy(&blob[aoff + (size_t)y * areaMask->cols], areaMask->ptr(y), (size_t)areaMask->cols);
        slam_match_job job;
        std::memset(&job, 0, sizeof(job));
        job.frame = 0; job.level = level; job.pw = patch.cols; job.ph = patch.rows;
        job.patch_off = 0; job.pmask_off = patchMask ? (int64_t)pb : -1;
        job.ax = areaRect.x; job.ay = areaRect.y; job.aw = areaRect.width; job.ah = areaRect.height;
        job.amask_off = areaMask ? (int64_t)aoff : -1;
        job.stride = stride; job.use_exact_area_mask = useExactAreaMask ? 1 : 0;
        job.score_off = -1;
        if (slamhip::failed(slam_hip_match_batch(slamhip::context(), live[k]->device(), &job, 1, blob.data(), blob.size(), &r), "matchFeatureInArea") ||
            slamhip::failed(r.status, "matchFeatureInArea", "job rejected by the device")) {
            response = -1.f;            // the sentinel of matching.cpp:26-30
            return cv::Point2i(0, 0);
        }
        response = r.response;
        return cv::Point2i(r.x, r.y);
    }
    if (slamhip::failed(slam_hip_match_feature_in_area(
                            slamhip::context(), im.data, im.cols, im.rows, im.step, patch.data, patch.cols, patch.rows,
                            patch.step, patchMask ? patchMask->data : nullptr, patchMask ? patchMask->step : 0,
                            areaRect.x, areaRect.y, areaRect.width, areaRect.height,
                            areaMask ? areaMask->data : nullptr, areaMask ? areaMask->step : 0, stride,
                            useExactAreaMask ? 1 : 0, nullptr, &r),
                        "matchFeatureInArea")) {
        response = -1.f;                // the sentinel of matching.cpp:26-30
        return cv::Point2i(0, 0);
    }
    response = r.response;
    return cv::Point2i(r.x, r.y);
}

namespace {

// Packs templates and masks row by row into one blob and fills the job table.  false (error recorded): a projection
// whose template and mask differ in size.
bool pack_jobs(const std::vector<SLAM::Projection>& f, float threshold, int stride,
               std::vector<slam_track_job>& jobs, std::vector<uint8_t>& blob) {
    size_t total = 0;
    for (const auto& p : f) {
        if (p.proj.cols > 0 && p.proj.rows > 0 && (p.projMask.cols != p.proj.cols || p.projMask.rows != p.proj.rows))
            return !slamhip::failed(SLAM_HIP_EINVAL, "Projection", "proj and projMask differ in size");
        total += 2 * (size_t)p.proj.rows * p.proj.cols;
    }
    blob.resize(total ? total : 1);
    jobs.resize(f.size());
    size_t off = 0;
    for (size_t i = 0; i < f.size(); i++) {
        const SLAM::Projection& p = f[i];
        const int pw = p.proj.cols, ph = p.proj.rows;
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
    return true;
}

}  // namespace

cv::Point2i SLAM::Feature::track(const ImagePyramid<byte>& pyramid, const Projection& p, float threshold, int stride,
                                 float& response) {
    std::vector<slam_track_job> jobs;
    std::vector<uint8_t> blob;
    slam_track_result r;
    if (!pack_jobs(std::vector<Projection>(1, p), threshold, stride, jobs, blob) ||
        slamhip::failed(slam_hip_track_batch(slamhip::context(), pyramid.device(), jobs.data(), 1, blob.data(), blob.size(), &r), "Feature::track") ||
        slamhip::failed(r.status, "Feature::track", "job rejected by the device")) {
        response = -1.f;                // the sentinel of feature.cpp:112-115
        return cv::Point2i(-1, -1);
    }
    response = r.response;
    return cv::Point2i(r.x, r.y);
}

SLAM::CameraState::CameraState(const matf& P_, const matf& t_) : P(P_), KR(3, 3), t(t_) {
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) KR(r, c) = P(r, c);      // SLAM.hpp:136
}

// SLAM.hpp:150-158 with OpenCV 2.4's arithmetic: KR*(p - t) is cv::gemm's 3x3 small-matrix case (float),
// P*p its generic case (products summed in double), a /= a(2) multiplies by (float)(1./a(2)).
matf SLAM::CameraState::project(const matf& p) const {
    float a[3];
    if (p.rows == 3) {
        const float d0 = p(0, 0) - t(0, 0), d1 = p(1, 0) - t(1, 0), d2 = p(2, 0) - t(2, 0);
        for (int r = 0; r < 3; r++) a[r] = KR(r, 0) * d0 + KR(r, 1) * d1 + KR(r, 2) * d2;
    } else {
        for (int r = 0; r < 3; r++) {
            double acc = 0.0;
            for (int k = 0; k < 4; k++) acc += (double)P(r, k) * (double)p(k, 0);
            a[r] = (float)acc;
        }
    }
    const float s = (float)(1.0 / (double)a[2]);
    matf out(2, 1);
    out(0, 0) = a[0] * s;
    out(1, 0) = a[1] * s;
    return out;
}

namespace {

// feature.cpp:61-94 through the library's host arithmetic; fills the warp job's matrix and output size.
bool setup_projection(const SLAM::Feature& f, const SLAM::CameraState& state, const matf& p3d, slam_warp_job& job,
                      cv::Rect& location) {
    std::memset(&job, 0, sizeof(job));
    location = cv::Rect(0, 0, 0, 0);
    if (f.descriptor.cols <= 0 || f.descriptor.rows <= 0) return false;              // feature.cpp:56-59
    if (f.B.rows != 4 || f.B.cols != 3 || state.P.rows != 3 || state.P.cols != 4 || p3d.rows != 3)
        return !slamhip::failed(SLAM_HIP_EINVAL, "Feature::project", "B must be 4x3, state.P 3x4, p3d 3x1");
    for (int r = 0; r < 3; r++) f.B(r, 2) = p3d(r, 0);                                // feature.cpp:61
    float P[12], B[12], p[3];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) P[r * 4 + c] = state.P(r, c);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 3; c++) B[r * 3 + c] = f.B(r, c);
    for (int r = 0; r < 3; r++) p[r] = p3d(r, 0);
    int32_t loc[4], valid = 0;
    if (slamhip::failed(slam_hip_project_setup(P, B, p, f.descriptor.cols, f.descriptor.rows, loc, job.m, &valid), "Feature::project", "bad arguments"))
        return false;
    location = cv::Rect(loc[0], loc[1], loc[2], loc[3]);
    if (!valid) return false;
    job.dw = f.descriptor.cols; job.dh = f.descriptor.rows; job.ow = loc[2]; job.oh = loc[3];
    return true;
}

// false (error recorded): the device pass failed or rejected a job; `res` then holds the empty-projection sentinel
// of feature.cpp:112-115 for every feature.
bool track_features(const ImagePyramid<byte>& pyramid, const SLAM::CameraState& state,
                    const SLAM::Feature* const* feats, const matf* const* p3ds, size_t n, float threshold, int stride,
                    std::vector<slam_track_result>& res) {
    std::vector<slam_warp_job> wj(n);
    std::vector<slam_track_job> tj(n);
    std::vector<uint8_t> blob;
    for (size_t i = 0; i < n; i++) {
        cv::Rect loc;
        const SLAM::Feature& f = *feats[i];
        setup_projection(f, state, *p3ds[i], wj[i], loc);       // an empty projection leaves ow = oh = 0
        wj[i].desc_off = (int64_t)blob.size();
        for (int r = 0; r < f.descriptor.rows; r++) blob.insert(blob.end(), f.descriptor.ptr(r), f.descriptor.ptr(r) + f.descriptor.cols);
        const matf p2d = state.project(*p3ds[i]);                // feature.cpp:117
        std::memset(&tj[i], 0, sizeof(tj[i]));
        tj[i].frame = 0; tj[i].px = p2d(0, 0); tj[i].py = p2d(1, 0); tj[i].threshold = threshold; tj[i].stride = stride;
    }
    if (blob.empty()) blob.push_back(0);
    res.resize(n);
    bool bad = slamhip::failed(slam_hip_project_track_batch(slamhip::context(), pyramid.device(), wj.data(), tj.data(), (int)n, blob.data(),
                                                            blob.size(), res.data()),
                               "Feature::track");
    for (size_t i = 0; !bad && i < n; i++) bad = slamhip::failed(res[i].status, "Feature::track", "job rejected by the device");
    if (bad) {
        const slam_track_result none = {-1, -1, -1.f, 0.f, 0.f, 0, SLAM_HIP_EINVAL};
        res.assign(n, none);
    }
    return !bad;
}

}  // namespace

matb SLAM::Feature::project(const CameraState& state, const matf& p3d, matb& mask, cv::Rect& location) const {
    slam_warp_job job;
    if (!setup_projection(*this, state, p3d, job, location)) return matb();
    float P[12], Bm[12], p[3];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) P[r * 4 + c] = state.P(r, c);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 3; c++) Bm[r * 3 + c] = B(r, c);
    for (int r = 0; r < 3; r++) p[r] = p3d(r, 0);
    matb proj(location.height, location.width);
    mask = matb(location.height, location.width);
    int32_t loc[4], valid = 0;
    if (slamhip::failed(slam_hip_project(slamhip::context(), descriptor.data, descriptor.cols, descriptor.rows, descriptor.step, P, Bm, p,
                                         proj.data, mask.data, (size_t)location.width * location.height, loc, &valid),
                        "Feature::project")) {
        mask = matb();
        location = cv::Rect(0, 0, 0, 0);
        return matb();                  // what an empty projection returns (feature.cpp:56-59)
    }
    return proj;
}

cv::Point2i SLAM::Feature::track(const ImagePyramid<byte>& pyramid, const CameraState& state, const matf& p3d,
                                 float threshold, int stride, float& response, cv::Mat*) const {
    const Feature* f = this;
    const matf* p = &p3d;
    std::vector<slam_track_result> res;
    track_features(pyramid, state, &f, &p, 1, threshold, stride, res);
    response = res[0].response;
    return cv::Point2i(res[0].x, res[0].y);
}

void SLAM::matchPoints(const ImagePyramid<byte>& pyramid, const CameraState& state, const std::vector<Feature>& features,
                       const std::vector<matf>& p3ds, std::vector<Match>& matches, float threshold, int stride,
                       std::vector<float>* responses) {
    matches.clear();
    if (responses) responses->assign(features.size(), -1.f);
    if (features.size() != p3ds.size()) { slamhip::failed(SLAM_HIP_EINVAL, "matchPoints", "one p3d per feature"); return; }
    if (features.empty()) return;
    std::vector<const Feature*> fp(features.size());
    std::vector<const matf*> pp(features.size());
    for (size_t i = 0; i < features.size(); i++) { fp[i] = &features[i]; pp[i] = &p3ds[i]; }
    std::vector<slam_track_result> res;
    track_features(pyramid, state, fp.data(), pp.data(), features.size(), threshold, stride, res);
    for (size_t i = 0; i < res.size(); i++) {
        if (responses) (*responses)[i] = res[i].response;
        if (res[i].response > threshold)      // matching.cpp:262
            matches.push_back(Match(cv::Point2f((float)res[i].x, (float)res[i].y), (int)i));
    }
}

void SLAM::matchPoints(const cv::Mat_<byte>& im, const CameraState& state, std::vector<Match>& matches, float threshold) {
    const int stride = 3;                                   // matching.cpp:243
    std::vector<float> subsamples;
    subsamples.push_back(1.f);
    subsamples.push_back(3.f);                              // matching.cpp:244-246
    ImagePyramid<byte> pyramid(im, subsamples);
    if (!pyramid.ok() || features.empty()) return;
    std::vector<matf> p3ds(features.size());
    for (size_t i = 0; i < features.size(); i++) {
        if (features[i].iKalman < 0 || (size_t)features[i].iKalman >= kalman.points.size()) {
            slamhip::failed(SLAM_HIP_EINVAL, "matchPoints", "feature without a point in kalman.points");
            return;
        }
        p3ds[i] = kalman.getPt3d(features[i].iKalman);      // matching.cpp:252
    }
    std::vector<Match> found;
    matchPoints(pyramid, state, features, p3ds, found, threshold, stride, nullptr);
    matches.insert(matches.end(), found.begin(), found.end());
}

void SLAM::matchPoints(const ImagePyramid<byte>& pyramid, const std::vector<Projection>& features,
                       std::vector<Match>& matches, float threshold, int stride, std::vector<float>* responses) {
    matches.clear();
    if (responses) responses->assign(features.size(), -1.f);
    if (features.empty()) return;
    std::vector<slam_track_job> jobs;
    std::vector<uint8_t> blob;
    if (!pack_jobs(features, threshold, stride, jobs, blob)) return;
    std::vector<slam_track_result> res(features.size());
    if (slamhip::failed(slam_hip_track_batch(slamhip::context(), pyramid.device(), jobs.data(), (int)jobs.size(), blob.data(),
                                             blob.size(), res.data()),
                        "matchPoints"))
        return;
    for (size_t i = 0; i < res.size(); i++)
        if (slamhip::failed(res[i].status, "matchPoints", "job rejected by the device")) return;      // no matches at all
    for (size_t i = 0; i < res.size(); i++) {
        if (responses) (*responses)[i] = res[i].response;
        if (res[i].response > threshold)      // matching.cpp:262
            matches.push_back(Match(cv::Point2f((float)res[i].x, (float)res[i].y), (int)i));
    }
}

void SLAM::matchPointsGated(const ImagePyramid<byte>& pyramid, const std::vector<Projection>& features,
                            const std::vector<cv::Vec3f>& covariances, float gate, std::vector<Match>& matches,
                            float threshold, int stride, std::vector<float>* responses) {
    matches.clear();
    if (responses) responses->assign(features.size(), -1.f);
    if (features.size() != covariances.size()) { slamhip::failed(SLAM_HIP_EINVAL, "matchPointsGated", "one covariance per feature"); return; }
    if (features.empty()) return;
    std::vector<slam_track_job> jobs;
    std::vector<uint8_t> blob;
    if (!pack_jobs(features, threshold, stride, jobs, blob)) return;
    const float areaRes = 1.f / pyramid.subsamples[pyramid.nSubs() - 1];
    std::vector<slam_ellipse_job> windows(features.size());
    for (size_t i = 0; i < features.size(); i++) {
        const float cov[3] = {covariances[i][0], covariances[i][1], covariances[i][2]};
        if (slamhip::failed(slam_hip_ellipse_setup(jobs[i].px, jobs[i].py, cov, gate, areaRes, &windows[i]), "matchPointsGated", "bad covariance or gate")) return;
    }
    std::vector<slam_track_result> res(features.size());
    if (slamhip::failed(slam_hip_track_ellipse_batch(slamhip::context(), pyramid.device(), jobs.data(), windows.data(),
                                                     (int)jobs.size(), blob.data(), blob.size(), res.data()),
                        "matchPointsGated"))
        return;
    for (size_t i = 0; i < res.size(); i++)
        if (slamhip::failed(res[i].status, "matchPointsGated", "job rejected by the device")) return;
    for (size_t i = 0; i < res.size(); i++) {
        if (responses) (*responses)[i] = res[i].response;
        if (res[i].response > threshold)
            matches.push_back(Match(cv::Point2f((float)res[i].x, (float)res[i].y), (int)i));
    }
}

void SLAM::matchLines(const ImagePyramid<byte>& pyramid, const std::vector<LineSearch>& lines, std::vector<Match>& matches,
                      float threshold, int stride, std::vector<float>* responses) {
    matches.clear();
    if (responses) responses->assign(lines.size(), -1.f);
    if (lines.empty()) return;
    std::vector<slam_line_job> jobs(lines.size());
    std::vector<int32_t> circles, begin(lines.size() + 1, 0);
    std::vector<uint8_t> blob;
    for (size_t i = 0; i < lines.size(); i++) {
        const LineSearch& l = lines[i];
        if (l.binCenters.size() != l.binRadii.size()) { slamhip::failed(SLAM_HIP_EINVAL, "matchLines", "one radius per bin centre"); return; }
        slam_line_job& j = jobs[i];
        std::memset(&j, 0, sizeof(j));
        j.desc_off = (int64_t)blob.size();
        for (int r = 0; r < l.descriptor.rows; r++) blob.insert(blob.end(), l.descriptor.ptr(r), l.descriptor.ptr(r) + l.descriptor.cols);
        j.amask_off = -1;
        j.dw = l.descriptor.cols; j.dh = l.descriptor.rows;
        j.ax = l.areaRect.x; j.ay = l.areaRect.y; j.aw = l.areaRect.width; j.ah = l.areaRect.height;
        j.threshold = threshold; j.stride = stride;
        for (size_t k = 0; k < l.binCenters.size(); k++) {
            circles.push_back(l.binCenters[k].x); circles.push_back(l.binCenters[k].y); circles.push_back(l.binRadii[k]);
        }
        begin[i + 1] = (int32_t)(circles.size() / 3);
    }
    if (blob.empty()) blob.push_back(0);
    std::vector<slam_track_result> res(lines.size());
    if (slamhip::failed(slam_hip_match_lines_batch(slamhip::context(), pyramid.device(), jobs.data(), (int)jobs.size(),
                                                   circles.empty() ? nullptr : circles.data(), (int)(circles.size() / 3), begin.data(),
                                                   blob.data(), blob.size(), res.data()),
                        "matchLines"))
        return;
    for (size_t i = 0; i < res.size(); i++)
        if (slamhip::failed(res[i].status, "matchLines", "job rejected by the device")) return;
    for (size_t i = 0; i < res.size(); i++) {
        if (responses) (*responses)[i] = res[i].response;
        if (res[i].response > threshold)      // matching.cpp:288
            matches.push_back(Match(cv::Point2f((float)res[i].x, (float)res[i].y), (int)i));
    }
}

matb SLAM::lineAreaMask(const LineSearch& l) {
    if (l.areaRect.width <= 0 || l.areaRect.height <= 0) return matb();
    slam_ctx* ctx = slamhip::context();
    const size_t npix = (size_t)l.areaRect.width * l.areaRect.height, nc = l.binCenters.size();
    std::vector<int32_t> circles;
    for (size_t k = 0; k < nc; k++) { circles.push_back(l.binCenters[k].x); circles.push_back(l.binCenters[k].y); circles.push_back(l.binRadii[k]); }
    slam_disc_job job;
    job.mask_off = 0; job.w = l.areaRect.width; job.h = l.areaRect.height; job.circle_begin = 0; job.circle_count = (int32_t)nc;
    void *d_c = nullptr, *d_j = nullptr, *d_m = nullptr;
    matb out(l.areaRect.height, l.areaRect.width);
    int32_t st = SLAM_HIP_OK;
    int rc = slam_hip_malloc(ctx, 12 * nc + 16, &d_c);
    if (rc == SLAM_HIP_OK) rc = slam_hip_malloc(ctx, sizeof(job) + 16, &d_j);
    if (rc == SLAM_HIP_OK) rc = slam_hip_malloc(ctx, npix, &d_m);
    if (rc == SLAM_HIP_OK && nc) rc = slam_hip_memcpy_h2d(ctx, d_c, circles.data(), 12 * nc);
    if (rc == SLAM_HIP_OK) rc = slam_hip_memcpy_h2d(ctx, d_j, &job, sizeof(job));
    if (rc == SLAM_HIP_OK)
        rc = slam_hip_disc_masks_d(ctx, (const int32_t*)d_c, (int)nc, (const slam_disc_job*)d_j, 1, (uint8_t*)d_m, npix,
                                   (int32_t*)((uint8_t*)d_j + sizeof(job)), (int)npix);
    if (rc == SLAM_HIP_OK) rc = slam_hip_memcpy_d2h(ctx, &st, (uint8_t*)d_j + sizeof(job), 4);
    if (rc == SLAM_HIP_OK && st == SLAM_HIP_OK) rc = slam_hip_memcpy_d2h(ctx, out.data, d_m, npix);
    slam_hip_free(ctx, d_c); slam_hip_free(ctx, d_j); slam_hip_free(ctx, d_m);
    if (slamhip::failed(rc, "lineAreaMask") || slamhip::failed(st, "lineAreaMask", "negative radius")) return matb();
    return out;
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
    if (slamhip::failed(slam_hip_add_new_points(slamhip::context(), im.data, im.cols, im.rows, im.step, kps.data(), (int)kps.size(), 1,
                                                occ.empty() ? nullptr : occ.data(), (int)(occ.size() / 2), n, minDist, rx, ry,
                                                sel.data(), &nsel, nullptr, descriptors ? desc.data() : nullptr, desc.size(),
                                                descriptors ? res.data() : nullptr),
                        "selectNewPoints"))
        return out;                     // no point selected
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
        if (rc != SLAM_HIP_ECAP && slamhip::failed(rc, "getSortedKeyPoints")) { out.clear(); return; }
        if (n > cap) { cap = n; continue; }   // the list was cut at cap: run again with room for all
        fastThreshold = thr;
        out.clear();
        out.reserve(n);
        for (int i = 0; i < n && i < cap; i++)
            out.push_back(cv::KeyPoint(kps[i].x, kps[i].y, 7.f, -1.f, kps[i].response, 0, -1));
        return;
    }
}
