// Host-side mirror of the reference's interfaces on the image hot path, implemented on top of the
// C ABI in include/slam_hip.h (libslam_hip.so).  Names, argument order and sentinel behaviour follow
// the reference so that its call sites compile against this header unchanged:
//   SLAM::matchFeatureInArea       SLAM.hpp:92-98, matching.cpp:9-228
//   ImagePyramid<byte>             imagePyramid.hpp:9-34
//   SLAM::Feature::project         SLAM.hpp:57-58, feature.cpp:53-100
//   SLAM::Feature::track           SLAM.hpp:62-65, feature.cpp:102-173
//   SLAM::matchLines               SLAM.hpp:108-109, matching.cpp:271-310 (the image work of LineFeature::track)
//   SLAM::getSortedKeyPoints       SLAM.hpp:118-119, new_point.cpp:113-127
//   SLAM::matchPoints              SLAM.hpp:106-107, matching.cpp:230-269 (the member call, and forms batched over features)
// Nothing here throws, as nothing in the reference does: a device error (no GPU, failed allocation) or a rejected
// argument gives the call's own sentinel - response -1 and (0,0) from matchFeatureInArea (matching.cpp:26-30), (-1,-1)
// and -1 from Feature::track (feature.cpp:112-115), no matches, no keypoints, an empty Mat - and leaves its code and
// message in slamhip::last_error() of the calling thread.
#pragma once

#include <string>
#include <vector>

#include "slam_cvtypes.hpp"
#include "slam_hip.h"

typedef unsigned char byte;
typedef cv::Mat_<byte> matb;
typedef cv::Mat_<float> matf;

namespace slamhip {
// The calling thread's context on device $SLAM_HIP_DEVICE (default 0), created on first use; nullptr (error recorded)
// when none can be made.
slam_ctx* context();
// The last failure of the calling thread: a SLAM_HIP_E* code (SLAM_HIP_OK: none since clear_error) and what failed.
struct Error { int code; std::string message; };
const Error& last_error();
void clear_error();
// rc < 0: records (rc, "what: detail") as the thread's last error and returns true.  Without `detail` the library's
// slam_hip_last_error_string is used.
bool failed(int rc, const char* what);
bool failed(int rc, const char* what, const char* detail);
}  // namespace slamhip

template <typename imtype> class ImagePyramid;
namespace slamhip {
std::vector<const ImagePyramid<unsigned char>*>& live_pyramids();
}

// imagePyramid.hpp:9-34.  `images` are host copies of the levels (what reference code indexes);
// the levels also stay resident on the device for the search calls below.
template <typename imtype> class ImagePyramid;

template <> class ImagePyramid<byte> {
public:
    typedef cv::Mat_<byte> mati;
    std::vector<mati> images;
    std::vector<float> subsamples;
    ImagePyramid(const mati& imFullSize, const std::vector<float>& subsamples);
    ~ImagePyramid();
    ImagePyramid(const ImagePyramid&) = delete;
    ImagePyramid& operator=(const ImagePyramid&) = delete;
    size_t nSubs() const { return subsamples.size(); }
    // false: the device pyramid could not be made (slamhip::last_error says why); images is then empty
    bool ok() const { return pyr_ != nullptr; }
    const slam_pyr* device() const { return pyr_; }
    // index of the level whose host image `im` is (same buffer and shape), or -1
    int levelOf(const mati& im) const;
private:
    slam_pyr* pyr_ = nullptr;
};

class SLAM {
public:
    // SLAM.hpp:82-86
    struct Match {
        int iFeature;
        cv::Point2f pos;
        Match(const cv::Point2f& pos_, int iFeature_) : iFeature(iFeature_), pos(pos_) {}
    };

    // The members of SLAM::CameraState (SLAM.hpp:26-46) this path reads: P = K [R | -R t] (3x4),
    // KR = P(:, 0:3), t (3x1).  project is SLAM.hpp:150-158.
    struct CameraState {
        matf P, KR, t;
        CameraState() {}
        CameraState(const matf& P_, const matf& t_);      // KR is taken from P
        matf project(const matf& p) const;
    };

    // What Feature::project (feature.cpp:53-100) hands to the search: the warped template, its
    // {0,1} mask and the predicted image point.
    struct Projection {
        matb proj, projMask;
        cv::Point2f p2d;
    };

    struct Feature {
        int iKalman = 0;
        matb descriptor;
        mutable matf B;        // 4x3 (feature.cpp:18-34); project writes p3d into column 2 (feature.cpp:61)

        // feature.cpp:53-100: the matrix steps on the host, both warpPerspective calls on the device.
        // Returns proj and fills mask / location; an empty Mat when the projection is empty or larger
        // than 500 pixels (feature.cpp:56-59, 84-86).
        matb project(const CameraState& state, const matf& p3d, matb& mask, cv::Rect& location) const;
        // feature.cpp:102-173 with the reference's signature: project, then the coarse-to-fine walk, in
        // one device pass (the projected template never visits the host).  disp is ignored.
        cv::Point2i track(const ImagePyramid<byte>& pyramid, const CameraState& state, const matf& p3d,
                          float threshold, int stride, float& response, cv::Mat* disp = nullptr) const;

        // feature.cpp:117-172 for an already projected feature: coarse-to-fine walk over the
        // pyramid levels on the device.  Returns Point2i(trackedPoint); (-1,-1) and response -1
        // when the projection is empty (feature.cpp:112-115).
        static cv::Point2i track(const ImagePyramid<byte>& pyramid, const Projection& p, float threshold,
                                 int stride, float& response);
    };

    // SLAM.hpp:92-98
    static cv::Point2i matchFeatureInArea(const cv::Mat_<byte>& im, const cv::Mat_<byte>& patch,
                                          const cv::Mat_<byte>* patchMask, const cv::Rect& areaRect,
                                          const matb* areaMask, int stride, float& response,
                                          bool useExactAreaMask = false);

    // matching.cpp:230-269 after the projections: every feature of one frame in one batched launch;
    // keeps Match(pos, i) when response > threshold (matching.cpp:254-255).  stride 3 as matching.cpp:243.
    static void matchPoints(const ImagePyramid<byte>& pyramid, const std::vector<Projection>& features,
                            std::vector<Match>& matches, float threshold = .95f, int stride = 3,
                            std::vector<float>* responses = nullptr);

    // matching.cpp:250-261 with the reference's inputs: feature i is projected with p3ds[i]
    // (kalman.getPt3d(i), 3x1) and tracked, all features of the frame in one device pass.
    static void matchPoints(const ImagePyramid<byte>& pyramid, const CameraState& state,
                            const std::vector<Feature>& features, const std::vector<matf>& p3ds,
                            std::vector<Match>& matches, float threshold = .95f, int stride = 3,
                            std::vector<float>* responses = nullptr);

    // matchPoints for predictions that carry a 2x2 covariance (innovation covariance of the filter, in
    // full-resolution pixels^2, stored as {sxx, sxy, syy}): feature i's coarsest level is searched
    // inside the gate ellipse (p - p2d)^T S^-1 (p - p2d) <= gate instead of the fixed square of
    // feature.cpp:134,144-146, the finer levels as in the reference.  A feature without a window
    // (degenerate covariance, window beyond 300 pixels as in lineFeature.cpp:81-84) gets response -1.
    // Not in the reference.
    static void matchPointsGated(const ImagePyramid<byte>& pyramid, const std::vector<Projection>& features,
                                 const std::vector<cv::Vec3f>& covariances, float gate, std::vector<Match>& matches,
                                 float threshold = .95f, int stride = 3, std::vector<float>* responses = nullptr);

    // What LineFeature::track (lineFeature.cpp:22-96) has computed on the CPU from the cone before it
    // touches the image: the descriptor, the area rectangle on the coarsest level and the projected
    // relevant bins as circles relative to that rectangle (lineFeature.cpp:86-96).
    struct LineSearch {
        matb descriptor;
        cv::Rect areaRect;
        std::vector<cv::Point2i> binCenters;
        std::vector<int> binRadii;
    };
    // lineFeature.cpp:87-96: the area mask (0 / 255) of one LineSearch, drawn on the device.
    static matb lineAreaMask(const LineSearch& line);
    // The image work of SLAM::matchLines (matching.cpp:279-292) for all lines of a frame in one device
    // pass: masks drawn, coarsest level searched inside them, finer levels refined (lineFeature.cpp:120-165).
    // Keeps Match(pos, i) when response > threshold; pyramid {1,2} and stride 2 are matching.cpp:273-277.
    static void matchLines(const ImagePyramid<byte>& pyramid, const std::vector<LineSearch>& lines,
                           std::vector<Match>& matches, float threshold = .95f, int stride = 2,
                           std::vector<float>* responses = nullptr);

    // The selection loop of SLAM::addNewLines (new_point.cpp:137-155) and the descriptor crop of the
    // kept points (lineFeature.cpp:7-10): `keypoints` as returned by getSortedKeyPoints; a point is
    // kept when it is at least minDist away from every match, line match and point kept before it and
    // its (2rx+1) x (2ry+1) neighbourhood lies inside the image; at most n are kept.  Returns the
    // indices into `keypoints`; descriptors (optional) receives one crop per kept point.
    static std::vector<int> selectNewPoints(const matb& im, const std::vector<cv::KeyPoint>& keypoints,
                                            const std::vector<Match>& matches, const std::vector<Match>& lineMatches,
                                            int n, float minDist, int rx, int ry,
                                            std::vector<matb>* descriptors = nullptr);

    // SLAM.hpp:118-119; fastThreshold is the mutable member of SLAM.hpp:22.
    void getSortedKeyPoints(const matb& im, size_t nMinPts, std::vector<cv::KeyPoint>& out) const;
    mutable float fastThreshold = 10.f;

    // The reference's own member call, SLAM.hpp:106-107 / matching.cpp:230-269, on the members it reads: `features`
    // and, of the filter (out of scope), the one thing the path asks it for - kalman.getPt3d(i), the 3x1 point of
    // state entry i (kalman.hpp:32-34), kept here as the points the caller's filter last reported.  Builds the
    // pyramid {1, 3} (matching.cpp:243-247), projects and tracks every feature with stride 3 in one device pass and
    // appends Match(pos, i) when response > threshold (matching.cpp:254-255; like the reference it appends to
    // `matches` without clearing it).  The debug overlay (matching.cpp:232-241, 262-268) is not drawn.
    struct PointSource {
        std::vector<matf> points;
        matf getPt3d(int i) const { return points[(size_t)i]; }
    };
    std::vector<Feature> features;
    PointSource kalman;
    void matchPoints(const cv::Mat_<byte>& im, const CameraState& state, std::vector<Match>& matches, float threshold = .95f);
};
