/*
 * CPU oracle for the SLAM image hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded, dependency-free restatement of what the reference computes
 * on its tracking hot path, used only as the checker for the HIP implementation (tests/,
 * __graft_entry__.smoke(), bench.py's cpu_baseline leg).  Nothing under slam_amd/ may
 * include, link or load it.
 *
 * What it follows (paths relative to the reference tree):
 *   matching.cpp:9-228        SLAM::matchFeatureInArea     -> so_match_feature_in_area
 *   imagePyramid.hpp:15-30    ImagePyramid<T> constructor  -> so_pyramid_level_size, so_resize_u8_linear
 *   feature.cpp:117-172       SLAM::Feature::track         -> so_track
 *   new_point.cpp:109-127     SLAM::getSortedKeyPoints     -> so_get_sorted_keypoints
 *   new_point.cpp:137-155     SLAM::addNewLines selection  -> so_select_new_points
 *   feature.cpp:36-51         Feature::newDescriptor crop  -> so_extract_descriptor
 *   feature.cpp:53-100        SLAM::Feature::project       -> so_project_setup, so_warp_perspective_u8
 *   lineFeature.cpp:86-165    LineFeature::track searches  -> so_disc_mask, so_line_track
 *   (build-defined)           gate-ellipse search windows  -> so_ellipse_setup, so_ellipse_mask
 *   tests/test_matching.cpp   property test generator      -> so_proptest_*
 *
 * The arithmetic of cv::resize, cv::matchTemplate, cv::norm / Mat::dot / Mat::mul,
 * cv::minMaxLoc and cv::FAST lives in OpenCV 2.4.x, which is NOT vendored by the reference and
 * is absent from this machine; the version is not pinned by the reference (Makefile:2).  Their
 * published algorithms are restated here from the OpenCV 2.4 sources' documented behaviour.
 *
 * PARITY PINNING: the only results of this path the reference's own tests pin are the four
 * property tests of tests/test_matching.cpp (the planted patch is found); so_proptest_* restates
 * that generator and acceptance rule and the oracle passes all 400 trials.  Everything else
 * (resize, FAST + NMS, getSortedKeyPoints, Feature::track, the exact float scores) has no golden
 * vector in the reference: PARITY UNPINNED for those — the oracle is the spec of record.
 */
#ifndef SLAM_ORACLE_H
#define SLAM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct so_match_out {
    int x, y;        /* argmax position in image coordinates (matching.cpp:225-227) */
    float response;  /* max score (matching.cpp:224) */
    float dx, dy;    /* sub-pixel offset, build-defined extra; never changes x, y, response */
    int aw, ah;      /* cropped area size = size of the score map (matching.cpp:24-25) */
    int ax, ay;      /* cropped area origin */
} so_match_out;

/* imagePyramid.hpp:27  cv::Size(round(w/sub), round(h/sub)) */
void so_pyramid_level_size(int w, int h, float sub, int* dw, int* dh);

/* cv::resize(src, dst, Size(dw,dh)) with the default INTER_LINEAR on 8-bit single channel.
 * Returns 0, or -1 when a size is not positive. */
int so_resize_u8_linear(const uint8_t* src, int sw, int sh, ptrdiff_t spitch,
                        uint8_t* dst, int dw, int dh, ptrdiff_t dpitch);

/* imagePyramid.hpp:15-30: level i = clone when subs[i]==1 else resize from the FULL image.
 * levels[i] must hold so_pyramid_level_size() bytes, pitch = its width.  Returns 0 / -1. */
int so_pyramid_build(const uint8_t* im, int w, int h, ptrdiff_t pitch,
                     int nlev, const float* subs, uint8_t** levels);

/* matching.cpp:9-228.  patch_mask / area_mask may be NULL.  score_out (optional) receives the
 * cropped ah x aw score map, row-major, pitch aw; it must hold area_w*area_h floats.
 * Returns 0, or -1 on arguments the reference cannot be called with (stride<1, empty patch). */
int so_match_feature_in_area(const uint8_t* im, int w, int h, ptrdiff_t ipitch,
                             const uint8_t* patch, int pw, int ph, ptrdiff_t ppitch,
                             const uint8_t* patch_mask, ptrdiff_t mpitch,
                             int area_x, int area_y, int area_w, int area_h,
                             const uint8_t* area_mask, ptrdiff_t ampitch,
                             int stride, int use_exact_area_mask,
                             float* score_out, so_match_out* out);

typedef struct so_track_out {
    int x, y;          /* Point2i(trackedPoint) (feature.cpp:152,172) */
    float response;
    float dx, dy;      /* sub-pixel offset of the last level run, in that level's pixels */
    int levels_run;    /* number of pyramid levels matched before returning */
} so_track_out;

/* feature.cpp:117-172 given the projected template `proj` + `proj_mask` (feature.cpp:109) and the
 * predicted point p2d (feature.cpp:117).  levels[i] / lw[i] / lh[i]: pyramid images (pitch =
 * width).  Returns 0 / -1 (bad arguments). */
int so_track(uint8_t* const* levels, const int* lw, const int* lh, const float* subs, int nlev,
             const uint8_t* proj, const uint8_t* proj_mask, int projw, int projh,
             float px, float py, float threshold, int stride, so_track_out* out);

/* so_track with the coarsest level searched in coarse_area = {x, y, w, h} (that level's pixels) under
 * coarse_mask (pitch w, tested for non-zero, or NULL) instead of the reference's fixed square.  Not in
 * the reference: the build's elliptical-uncertainty windows.  coarse_area == NULL is so_track. */
int so_track_ex(uint8_t* const* levels, const int* lw, const int* lh, const float* subs, int nlev,
                const uint8_t* proj, const uint8_t* proj_mask, int projw, int projh,
                float px, float py, float threshold, int stride,
                const int* coarse_area, const uint8_t* coarse_mask, so_track_out* out);

/* lineFeature.cpp:86-96: the area mask of LineFeature::track, zero-filled w x h with one filled
 * cv::circle of value 255 per (x, y, radius) triple.  Returns 0 / -1. */
int so_disc_mask(int w, int h, const int* circles, int ncircles, uint8_t* mask);

/* lineFeature.cpp:120-165: the searches of LineFeature::track given the coarsest-level area
 * rectangle and its mask (lineFeature.cpp:86-96): descriptor resized per level, coarsest level
 * searched without a patch mask inside the area mask, finer levels in (2*searchRad+1)^2 windows.
 * A level whose resized descriptor would be empty is reported as (-1,-1), response -1. */
int so_line_track(uint8_t* const* levels, const int* lw, const int* lh, const float* subs, int nlev,
                  const uint8_t* descr, int dw, int dh, int ax, int ay, int aw, int ah,
                  const uint8_t* area_mask, float threshold, int stride, so_track_out* out);

typedef struct so_keypoint {
    float x, y;       /* KeyPoint::pt */
    float response;   /* corner score */
} so_keypoint;

/* Threshold-free FAST-9-16 score S(x,y) = max(A_dark, A_bright) - 1 for 3<=x<w-3, 3<=y<h-3
 * (0 elsewhere, and clamped below at 0), as int16 map of pitch w. */
void so_fast_score_map(const uint8_t* im, int w, int h, ptrdiff_t pitch, int16_t* score);

/* cv::FAST(im, kps, threshold, nonmaxSuppression) for the 16-pixel ring.  Writes up to cap
 * keypoints in row-major order; returns the total number found (may exceed cap). */
int so_fast(const uint8_t* im, int w, int h, ptrdiff_t pitch, int threshold, int nonmax,
            so_keypoint* out, int cap);

/* new_point.cpp:113-127.  *fast_threshold is SLAM::fastThreshold (in/out).  Returns the number
 * of keypoints found (written up to cap, sorted by response desc, y asc, x asc). */
int so_get_sorted_keypoints(const uint8_t* im, int w, int h, ptrdiff_t pitch, size_t n_min_pts,
                            float* fast_threshold, so_keypoint* out, int cap);

/* new_point.cpp:124 with the tie order fixed as in so_get_sorted_keypoints. */
void so_sort_keypoints(so_keypoint* kps, int n);

/* new_point.cpp:137-155: walks the sorted keypoints and keeps up to n whose distance to every
 * occupied position (matches and lineMatches, xy pairs), to every point kept so far is >= min_dist
 * and whose (2rx+1) x (2ry+1) neighbourhood lies inside the image.  sel receives indices into kps;
 * returns how many were kept. */
int so_select_new_points(const so_keypoint* kps, int nkps, const float* occ_xy, int nocc, int n,
                         float min_dist, int rx, int ry, int w, int h, int* sel);

/* feature.cpp:42-45 / lineFeature.cpp:7-10: the descriptor crop around pt2d; out (optional) receives
 * the rows packed (pitch = *ow).  An empty crop reports 0 x 0. */
int so_extract_descriptor(const uint8_t* im, int w, int h, ptrdiff_t pitch, float px, float py, int rx, int ry,
                          uint8_t* out, int* ox, int* oy, int* ow, int* oh);

/* feature.cpp:53-100 (SLAM::Feature::project).  PARITY UNPINNED: no test of the reference pins it
 * (tests/test_project.cpp is disabled and asserts nothing); cv::warpPerspective / cv::remap,
 * cv::gemm and cv::invert are restated from OpenCV 2.4's published algorithms. */
typedef struct so_projection {
    int x, y, w, h;    /* `location` (feature.cpp:80-83); w x h is the size of proj and mask */
    float m[9];        /* Am as passed to warpPerspective (feature.cpp:90-94), row-major */
    int valid;         /* 0: project returns an empty Mat (feature.cpp:56-59, 84-86) */
} so_projection;

/* feature.cpp:61-94: P = state.P (3x4), B (4x3, column 2 rows 0-2 are replaced by p3d), row-major. */
int so_project_setup(const float* P, const float* B, const float* p3d, int dw, int dh, so_projection* out);

/* cv::warpPerspective(src, dst, M, Size(dw,dh), WARP_INVERSE_MAP | (nearest ? INTER_NEAREST :
 * INTER_LINEAR)) on 8-bit single channel, constant border 0.  Returns 0 / -1. */
int so_warp_perspective_u8(const uint8_t* src, int sw, int sh, ptrdiff_t spitch, const float* M, int nearest,
                           uint8_t* dst, int dw, int dh, ptrdiff_t dpitch);

/* The whole of feature.cpp:53-100: proj and mask (pitch out->w) must hold cap bytes each. */
int so_project(const uint8_t* descr, int dw, int dh, ptrdiff_t dpitch, const float* P, const float* B, const float* p3d,
               uint8_t* proj, uint8_t* mask, size_t cap, so_projection* out);

/* Uncertainty window of a prediction with a 2x2 covariance cov = {sxx, sxy, syy} (full-resolution
 * pixels^2): gate ellipse scaled by res, its rectangle and quadratic form.  Build-defined: the reference
 * has no live counterpart (SLAM::projectGaussian, new_point.cpp:31-107, is never called). */
typedef struct so_ellipse {
    int x, y, w, h;
    float ux, uy;          /* centre relative to (x, y) */
    float q11, q12, q22;   /* inside <=> q11 dx^2 + 2 q12 dx dy + q22 dy^2 <= 1, evaluated in float, or the pixel
                              nearest the centre */
    int valid;
} so_ellipse;
int so_ellipse_setup(float px, float py, const float* cov, float gate, float res, so_ellipse* out);
/* mask (pitch e->w) = 255 inside, 0 outside.  Returns 0 / -1 (invalid window). */
int so_ellipse_mask(const so_ellipse* e, uint8_t* mask);

/* ---- tests/test_matching.cpp:13-101 restated: generator + acceptance rule ------------------- */
typedef struct so_proptest_case {
    int w, h, pw, ph;
    int xpatch, ypatch;
    int ax, ay, aw, ah;
    uint8_t *im, *patch, *patch_mask, *area_mask;  /* malloc'd; masks NULL when unused */
} so_proptest_case;

void so_proptest_srand(unsigned seed);   /* the reference never seeds: glibc default = srand(1) */
int so_proptest_rand(void);              /* next value of the private glibc-compatible rand() stream */
/* Draws the next trial exactly as tests/test_matching.cpp:15-69 does (same rand() call order). */
void so_proptest_next(int use_area_mask, int use_patch_mask, so_proptest_case* c);
void so_proptest_free(so_proptest_case* c);
/* tests/test_matching.cpp:83-98: 1 = pass, 0 = fail, for a result (mx,my,response). */
int so_proptest_accept(const so_proptest_case* c, int mx, int my, float response);

#ifdef __cplusplus
}
#endif
#endif
