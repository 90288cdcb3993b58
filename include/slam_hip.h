/*
 * slam_hip.h — C ABI of the MI355X (gfx950) implementation of the SLAM tracker's image hot path:
 * batched image pyramid, normalised-cross-correlation template search (matchFeatureInArea and the
 * coarse-to-fine Feature::track walk) and FAST-9-16 corner detection with non-max suppression.
 *
 * Conventions
 *   - Every function returns 0 (SLAM_HIP_OK) or a negative SLAM_HIP_E* code; nothing throws.
 *     slam_hip_last_error_string() gives the text of the last failure on a context.
 *   - Plain pointers and sizes only.  `h_` / unprefixed pointers are host memory, `d_` pointers are
 *     device (HIP) memory on the context's device.  The caller owns every buffer it passes.
 *   - One slam_ctx per (host thread, device); calls on one context are not internally locked.
 *     The `_d` entry points only enqueue work on the context's stream and return; the host-table
 *     entry points upload, run, download and synchronise.
 *   - Images are 8-bit single channel, row-major, `pitch` bytes between rows (a cv::Mat ROI's
 *     `step`), `frame_stride` bytes between the frames of a batch.
 *   - Masks follow the reference's contract: patch masks hold 0 or 1 (feature.cpp:97-98 produces
 *     exactly that); the host-table entry points reject other values with SLAM_HIP_EINVAL, the
 *     `_d` entry points treat any non-zero byte as 1.  Area masks are only tested for non-zero
 *     (matching.cpp:66,86,161,175).
 *
 * Each entry point names the reference interface it replaces (file:line in the reference tree).
 */
#ifndef SLAM_HIP_H
#define SLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLAM_HIP_OK       0
#define SLAM_HIP_EINVAL  (-1)   /* bad argument (null pointer, non-positive size, bad mask value ...) */
#define SLAM_HIP_EHIP    (-2)   /* a HIP runtime call failed */
#define SLAM_HIP_ENOMEM  (-3)   /* host or device allocation failed */
#define SLAM_HIP_ECAP    (-4)   /* an output capacity was exceeded (results truncated) */

#define SLAM_HIP_MAX_LEVELS 8

typedef struct slam_ctx slam_ctx;
typedef struct slam_pyr slam_pyr;

/* ---------------------------------------------------------------------------------------------
 * Context.  The reference has no device boundary; its only state on this path is the SLAM object
 * (SLAM.hpp:14-125).  A context binds one GPU and one HIP stream.
 * ------------------------------------------------------------------------------------------- */
int slam_hip_create(int device, slam_ctx** out);
int slam_hip_destroy(slam_ctx* ctx);
/* Use an existing hipStream_t (e.g. the caller's framework stream); NULL restores the context's own.  Waits for the
 * work queued on the stream used so far: the context's own buffers are ordered by one stream. */
int slam_hip_set_stream(slam_ctx* ctx, void* hip_stream);
int slam_hip_synchronize(slam_ctx* ctx);
/* LDS bytes each search job (one wavefront) may use: a multiple of 64 in [4096, 64512], default
 * 9664 (a CU's 160 KB are handed out in granules of 1280 bytes; with the walk kernel's 576 static bytes a job
 * then takes 8 granules and 16 jobs share a CU, as many as the kernels' registers allow; line walks: 12800,
 * 12 per CU).  Less LDS never adds jobs per CU; more LDS cuts large templates into fewer chunks at the price of
 * fewer jobs per CU.  Results do not depend on it. */
int slam_hip_set_match_lds(slam_ctx* ctx, int bytes);
/* LDS bytes per job of the lean point-walk kernel (the first of the two kernels behind slam_hip_track_batch[_d]): a
 * multiple of 64 in [4096, 65536], default 7680 (6 of the CU's 128 LDS granules of 1280 bytes: 21 walks share a
 * CU; only multiples of 1280 change that number).  A walk any of whose levels needs more - the image tile under the
 * window, the template's entry list and the score cells together - is left to the general kernel, so raise it for
 * templates beyond about 22x22 at full resolution (31x31 templates in 17x17 windows need 10240: 16 walks per CU).  Results do not
 * depend on it. */
int slam_hip_set_walk_lds(slam_ctx* ctx, int bytes);
/* on = 1: the lean point-walk kernel also takes walks whose full-resolution level (stored template, every cell) does
 * not fit the budget above whole: the template is then searched in chunks of rows, the sums of all cells running on
 * in LDS (the reference's live shape, 81x81 descriptors in 24x24 windows, matching.cpp:243-250 with feature.cpp:133-139:
 * 12800 bytes and on = 1).  A build of the kernel with more registers: leave it off (the default) for small templates.
 * The entry points that take a host job table pick budget and build themselves unless this or slam_hip_set_walk_lds
 * was called.  A batch of at most 256 walks under this build has the last level of every walk searched in bands of
 * rows over several wavefronts (one frame's walks: a third of the latency); their records live in a buffer of the
 * context that grows on demand.  Results do not depend on it. */
int slam_hip_set_walk_chunks(slam_ctx* ctx, int on);
/* Text of the last error on ctx (or of the last failed slam_hip_create when ctx is NULL). */
const char* slam_hip_last_error_string(const slam_ctx* ctx);

/* Device-memory helpers for hosts that do not link a HIP runtime themselves (synchronous). */
int slam_hip_malloc(slam_ctx* ctx, size_t bytes, void** d_ptr);
int slam_hip_free(slam_ctx* ctx, void* d_ptr);
int slam_hip_memcpy_h2d(slam_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int slam_hip_memcpy_d2h(slam_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);

/* ---------------------------------------------------------------------------------------------
 * Image pyramid — replaces ImagePyramid<T>::ImagePyramid (imagePyramid.hpp:15-30), batched over
 * nframes frames of equal size.  Level i is a copy of the frame when subs[i] == 1 (here: an
 * alias of the caller's frames, not a copy) and otherwise cv::resize(full, Size(round(w/sub),
 * round(h/sub))) computed directly from the full-resolution frame.  subs must be increasing.
 * ------------------------------------------------------------------------------------------- */
typedef struct slam_level_desc {
    uint8_t* d_data;        /* device pointer to frame 0 of this level */
    int32_t  w, h;
    size_t   pitch;         /* bytes between rows */
    size_t   frame_stride;  /* bytes between frames */
    float    sub;
} slam_level_desc;

int slam_hip_pyramid_create(slam_ctx* ctx, int w, int h, int nlev, const float* subs,
                            int nframes, slam_pyr** out);
int slam_hip_pyramid_destroy(slam_ctx* ctx, slam_pyr* pyr);
/* Build all levels from frames already in device memory (asynchronous).  d_frames must stay
 * valid while the pyramid is used: levels with sub == 1 alias it. */
int slam_hip_pyramid_build_d(slam_ctx* ctx, slam_pyr* pyr, const uint8_t* d_frames,
                             size_t pitch, size_t frame_stride);
/* Build from host frames: uploads into a buffer the pyramid owns, builds, synchronises. */
int slam_hip_pyramid_build(slam_ctx* ctx, slam_pyr* pyr, const uint8_t* h_frames,
                           size_t pitch, size_t frame_stride);
int slam_hip_pyramid_level(const slam_pyr* pyr, int level, slam_level_desc* out);
/* Copy one level of one frame to the host (ImagePyramid::images[level], imagePyramid.hpp:12). */
int slam_hip_pyramid_download(slam_ctx* ctx, const slam_pyr* pyr, int level, int frame,
                              uint8_t* h_dst, size_t dst_pitch);

/* ---------------------------------------------------------------------------------------------
 * Template search — replaces SLAM::matchFeatureInArea (SLAM.hpp:92-98, matching.cpp:9-228).
 * A job is one call of the reference function on one pyramid level of one frame.  Patch, patch
 * mask and area mask live in a byte blob (rows packed: pitch = their width) at the given offsets;
 * an offset of -1 means "absent" (NULL in the reference).
 * ------------------------------------------------------------------------------------------- */
typedef struct slam_match_job {
    int32_t frame, level;
    int32_t pw, ph;                 /* patch size (matching.cpp:36-37) */
    int64_t patch_off;              /* patch,      pitch pw */
    int64_t pmask_off;              /* patchMask,  pitch pw, or -1 */
    int32_t ax, ay, aw, ah;         /* areaRect as passed, uncropped (matching.cpp:12) */
    int64_t amask_off;              /* areaMask,   pitch aw, or -1 */
    int32_t stride;
    int32_t use_exact_area_mask;    /* matching.cpp:15,157 */
    int64_t score_off;              /* float index into the optional score buffer, or -1 */
} slam_match_job;

typedef struct slam_match_result {
    int32_t x, y;       /* returned Point2i (matching.cpp:222-227); (0,0) when the area misses the image or is empty */
    float   response;   /* matching.cpp:224; -1 when the area misses the image (matching.cpp:26-30) */
    float   dx, dy;     /* sub-pixel offset of the peak in [-0.5, 0.5]; not in the reference */
    int32_t status;     /* 0, or SLAM_HIP_EINVAL when the job's fields are unusable */
} slam_match_result;

/* Host tables: jobs, blob and results in host memory.  The pyramid must have been built. */
int slam_hip_match_batch(slam_ctx* ctx, const slam_pyr* pyr, const slam_match_job* jobs, int njobs,
                         const uint8_t* blob, size_t blob_bytes, slam_match_result* out);
/* Device tables, asynchronous, allocates nothing.  Offsets are checked against blob_bytes on the
 * device (a job that reaches outside gets status SLAM_HIP_EINVAL).  d_scores (scores_len floats) may
 * be NULL; when given, each job with score_off >= 0 also writes its cropped score map (row-major,
 * pitch = cropped width) there. */
int slam_hip_match_batch_d(slam_ctx* ctx, const slam_pyr* pyr, const slam_match_job* d_jobs,
                           int njobs, const uint8_t* d_blob, size_t blob_bytes,
                           float* d_scores, size_t scores_len, slam_match_result* d_out);
/* The reference call itself, one image, host pointers with pitches (NULL masks allowed):
 * SLAM::matchFeatureInArea(im, patch, patchMask, areaRect, areaMask, stride, response, exact). */
int slam_hip_match_feature_in_area(slam_ctx* ctx,
                                   const uint8_t* im, int w, int h, size_t im_pitch,
                                   const uint8_t* patch, int pw, int ph, size_t patch_pitch,
                                   const uint8_t* patch_mask, size_t pmask_pitch,
                                   int area_x, int area_y, int area_w, int area_h,
                                   const uint8_t* area_mask, size_t amask_pitch,
                                   int stride, int use_exact_area_mask,
                                   float* h_scores /* optional: area_w*area_h floats */,
                                   slam_match_result* out);

/* ---------------------------------------------------------------------------------------------
 * Coarse-to-fine tracking — replaces the level loop of SLAM::Feature::track (SLAM.hpp:62-65,
 * feature.cpp:117-172), one job per feature, all levels walked on the device.  The job carries
 * the projected template and its mask (the outputs of Feature::project, feature.cpp:109) at full
 * resolution and the predicted point p2d (feature.cpp:117); per level the template is resized on
 * the device (feature.cpp:130-131), the window is placed by the radius rule (feature.cpp:133-146),
 * matched (feature.cpp:147-149) and the walk stops early when response < 0.67*threshold
 * (feature.cpp:151-152).  The pyramid's subs are the `pyramid.subsamples` of the reference.
 * ------------------------------------------------------------------------------------------- */
typedef struct slam_track_job {
    int32_t frame;
    int32_t projw, projh;           /* size of proj; 0 = empty projection (feature.cpp:112-115) */
    int32_t flags;                  /* 0 = the reference's walk; SLAM_HIP_TRACK_COARSE_AREA: see below */
    int64_t proj_off, mask_off;     /* proj and proj_mask in the blob, pitch projw */
    float   px, py;                 /* state.project(p3d) */
    float   threshold;
    int32_t stride;
    /* With SLAM_HIP_TRACK_COARSE_AREA (not in the reference, whose coarsest window is the fixed square of
     * feature.cpp:134,144-146): the coarsest level is searched in this rectangle (that level's pixels)
     * under this area mask (pitch aw, tested for non-zero; -1: no mask) - e.g. the rasterised uncertainty
     * ellipse of the prediction - as matchFeatureInArea(level, totrack, &mask, areaRect, &areaMask, stride);
     * the finer levels follow the reference's rule.  Ignored (may be left zero) without the flag. */
    int64_t amask_off;
    int32_t ax, ay, aw, ah;
} slam_track_job;

#define SLAM_HIP_TRACK_COARSE_AREA 1

typedef struct slam_track_result {
    int32_t x, y;        /* returned Point2i; (-1,-1) for an empty projection */
    float   response;
    float   dx, dy;      /* sub-pixel offset at the last level run, in that level's pixels */
    int32_t levels_run;
    int32_t status;
} slam_track_result;

int slam_hip_track_batch(slam_ctx* ctx, const slam_pyr* pyr, const slam_track_job* jobs, int njobs,
                         const uint8_t* blob, size_t blob_bytes, slam_track_result* out);
int slam_hip_track_batch_d(slam_ctx* ctx, const slam_pyr* pyr, const slam_track_job* d_jobs,
                           int njobs, const uint8_t* d_blob, size_t blob_bytes,
                           slam_track_result* d_out);

/* slam_hip_track_batch_d for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same arguments, same results;
 * the queue of jobs the lean walk kernel hands to the general one lives in a fixed array inside the library's code
 * object (one copy per GPU, about 4 MB, held until the process ends) instead of a buffer the context grows on demand,
 * and the call makes no hipMalloc / hipFree or any other allocating call.  At most 1 048 576 jobs per call
 * (SLAM_HIP_ECAP beyond that).  All calls on one GPU share that array: calls that overlap on one GPU (different
 * streams or contexts) race; calls on one stream are ordered and each clears what it needs first. */
int slam_hip_track_batch_noalloc_d(slam_ctx* ctx, const slam_pyr* pyr, const slam_track_job* d_jobs,
                                   int njobs, const uint8_t* d_blob, size_t blob_bytes,
                                   slam_track_result* d_out);

/* ---------------------------------------------------------------------------------------------
 * Corner detection — replaces cv::FAST(im, keypoints, threshold, nonmaxSuppression) as called by
 * SLAM::getSortedKeyPoints (new_point.cpp:120) and that function itself (SLAM.hpp:118-119,
 * new_point.cpp:113-127).  Keypoints come out in cv::FAST's order (row-major); the remaining
 * cv::KeyPoint fields are constants (size 7, angle -1, octave 0, class_id -1).
 * ------------------------------------------------------------------------------------------- */
typedef struct slam_keypoint {
    float x, y;         /* KeyPoint::pt */
    float response;     /* corner score */
} slam_keypoint;

/* Scratch bytes slam_hip_fast_detect_d needs for this shape (device memory, caller-allocated, 16-byte
 * aligned): about 2 bytes per pixel - a fixed segment of 16-bit keys per (row, 248-column tile) plus
 * two count tables; it does not depend on cap_per_frame. */
size_t slam_hip_fast_scratch_bytes(int w, int h, int nframes, int cap_per_frame);
/* Device frames, asynchronous.  d_out holds nframes*cap_per_frame keypoints, d_counts nframes
 * ints (the number FOUND per frame; when it exceeds cap_per_frame the frame's list is its first
 * cap_per_frame keypoints in row-major order). */
int slam_hip_fast_detect_d(slam_ctx* ctx, const uint8_t* d_frames, int w, int h, size_t pitch,
                           size_t frame_stride, int nframes, int threshold, int nonmax,
                           slam_keypoint* d_out, int cap_per_frame, int32_t* d_counts,
                           void* d_scratch, size_t scratch_bytes);
/* Host frames; returns SLAM_HIP_ECAP (after filling out/counts) if any frame was truncated. */
int slam_hip_fast_detect(slam_ctx* ctx, const uint8_t* h_frames, int w, int h, size_t pitch,
                         size_t frame_stride, int nframes, int threshold, int nonmax,
                         slam_keypoint* out, int cap_per_frame, int32_t* counts);
/* SLAM::getSortedKeyPoints(im, nMinPts, out): *fast_threshold is SLAM::fastThreshold (in/out).
 * Sorted by response descending; ties by y then x ascending (std::sort leaves them unspecified).
 * *n_out = keypoints found; at most cap are written. */
int slam_hip_get_sorted_keypoints(slam_ctx* ctx, const uint8_t* im, int w, int h, size_t pitch,
                                  size_t n_min_pts, float* fast_threshold,
                                  slam_keypoint* out, int cap, int32_t* n_out);

/* ---------------------------------------------------------------------------------------------
 * New features — the steps that follow corner detection in SLAM::addNewLines (SLAM.hpp, new_point.cpp:
 * 129-156): sort by response (new_point.cpp:124), greedy minimum-distance selection (new_point.cpp:
 * 137-155) and the descriptor crop of the kept points (feature.cpp:42-45, lineFeature.cpp:7-10).
 * ------------------------------------------------------------------------------------------- */
/* Per-frame sort of slam_hip_fast_detect_d's output (row-major order) into d_out (same layout, must
 * not alias d_in): response descending, ties by y then x ascending.  Asynchronous. */
int slam_hip_sort_keypoints_d(slam_ctx* ctx, const slam_keypoint* d_in, const int32_t* d_counts,
                              int nframes, int cap_per_frame, slam_keypoint* d_out);
/* new_point.cpp:137-155 per frame on sorted keypoints.  Occupied positions (matches, then
 * lineMatches) are xy float pairs, frame f owning pairs [d_occ_off[f], d_occ_off[f+1]) (both
 * pointers may be NULL: nothing occupied).  d_sel receives nframes x n_new indices into the frame's
 * sorted list, d_nsel the number kept per frame.  Asynchronous, allocates nothing. */
int slam_hip_select_new_points_d(slam_ctx* ctx, const slam_keypoint* d_sorted, const int32_t* d_counts,
                                 int nframes, int cap_per_frame, const float* d_occ_xy,
                                 const int32_t* d_occ_off, int n_new, float min_dist, int rx, int ry,
                                 int w, int h, int32_t* d_sel, int32_t* d_nsel);

typedef struct slam_desc_job {
    int32_t frame;
    float   px, py;         /* pt2d */
    int32_t rx, ry;         /* descriptor half sizes: (2ry+1) x (2rx+1) when not cropped by the border */
    int32_t reserved;
    int64_t out_off;        /* byte offset of the crop in the output buffer (rows packed) */
} slam_desc_job;

typedef struct slam_desc_result {
    int32_t x0, y0, w, h;   /* the crop: rows [max(0,iround(y-ry)), min(H,iround(y+ry+1))), columns likewise */
    int32_t status;
} slam_desc_result;

int slam_hip_extract_descriptors_d(slam_ctx* ctx, const uint8_t* d_frames, int w, int h, size_t pitch,
                                   size_t frame_stride, int nframes, const slam_desc_job* d_jobs, int njobs,
                                   uint8_t* d_out, size_t out_bytes, slam_desc_result* d_res);
/* One frame, host tables: kps in cv::FAST order (already_sorted = 0) or sorted; sel / nsel as above;
 * sorted_out (optional, nkps entries) receives the sorted list; descr_out (optional, n_new crops of
 * (2rx+1)*(2ry+1) bytes) and descr_res the descriptors of the kept points. */
int slam_hip_add_new_points(slam_ctx* ctx, const uint8_t* im, int w, int h, size_t pitch,
                            const slam_keypoint* kps, int nkps, int already_sorted,
                            const float* occ_xy, int nocc, int n_new, float min_dist, int rx, int ry,
                            int32_t* sel, int32_t* nsel, slam_keypoint* sorted_out,
                            uint8_t* descr_out, size_t descr_bytes, slam_desc_result* descr_res);

/* ---------------------------------------------------------------------------------------------
 * Template projection — replaces SLAM::Feature::project (SLAM.hpp:58-60, feature.cpp:53-100): the
 * descriptor is warped by the predicted homography into `proj`, and an all-ones image of the same
 * size into its {0,1} `mask` (cv::warpPerspective with WARP_INVERSE_MAP, INTER_LINEAR and
 * INTER_NEAREST, constant border 0; feature.cpp:96-98).  Outputs are written as packed rows into a
 * byte blob — the same blob slam_hip_track_batch_d reads its templates from, so projected
 * templates never leave the device.
 * ------------------------------------------------------------------------------------------- */
/* The matrix steps of feature.cpp:61-94 (pure host arithmetic, no device work).  P = state.P (3x4),
 * B = Feature::B (4x3; rows 0-2 of column 2 are replaced by p3d as feature.cpp:61 does), row-major.
 * loc = `location` (x, y, width, height), m = Am as handed to warpPerspective (3x3 row-major),
 * *valid = 0 when project returns an empty Mat (feature.cpp:56-59, 84-86). */
int slam_hip_project_setup(const float* P, const float* B, const float* p3d, int dw, int dh,
                           int32_t* loc, float* m, int32_t* valid);

typedef struct slam_warp_job {
    int64_t desc_off;       /* descriptor in the source blob, pitch dw */
    int64_t proj_off;       /* proj in the output blob, pitch ow */
    int64_t mask_off;       /* mask in the output blob, pitch ow; -1: no mask wanted */
    int32_t dw, dh;         /* descriptor size */
    int32_t ow, oh;         /* location.width, location.height; 0 = empty projection, nothing is written */
    float   m[9];           /* Am: maps the homogeneous proj pixel (x, y, 1) into descriptor coordinates */
    int32_t reserved;
} slam_warp_job;

/* Device tables, asynchronous, allocates nothing.  d_status (optional, njobs ints) receives 0 or
 * SLAM_HIP_EINVAL per job (offsets outside the blobs, sizes out of range, non-finite matrix).
 * max_out_pixels: the largest ow*oh of the batch if the caller knows it (spreads large outputs over
 * more workgroups), else 0. */
int slam_hip_warp_batch_d(slam_ctx* ctx, const uint8_t* d_desc, size_t desc_bytes,
                          const slam_warp_job* d_jobs, int njobs, uint8_t* d_out, size_t out_bytes,
                          int32_t* d_status, int max_out_pixels);
/* Feature::project for one feature, host pointers: proj and mask (pitch loc[2]) must hold cap bytes
 * each; SLAM_HIP_ECAP when they are too small. */
int slam_hip_project(slam_ctx* ctx, const uint8_t* descr, int dw, int dh, size_t pitch,
                     const float* P, const float* B, const float* p3d,
                     uint8_t* proj, uint8_t* mask, size_t cap, int32_t* loc, int32_t* valid);
/* Feature::track as the reference runs it (feature.cpp:109-172) for a batch of features, host tables:
 * wjobs[i] (desc_off, dw, dh, ow, oh, m from slam_hip_project_setup; the output offsets are assigned
 * here) is projected on the device and searched with tjobs[i] (frame, px, py, threshold, stride; its
 * template fields are filled in here) without the templates visiting the host. */
int slam_hip_project_track_batch(slam_ctx* ctx, const slam_pyr* pyr, const slam_warp_job* wjobs,
                                 const slam_track_job* tjobs, int njobs, const uint8_t* desc_blob,
                                 size_t desc_bytes, slam_track_result* out);

/* ---------------------------------------------------------------------------------------------
 * Line features — the image work of SLAM::LineFeature::track (SLAM.hpp:68-80, lineFeature.cpp:22-166)
 * as called by SLAM::matchLines (matching.cpp:271-310).  The cone bins are projected by the caller
 * (lineFeature.cpp:41-84; the cone stays on the CPU); what runs here is the area mask of filled discs
 * (lineFeature.cpp:86-96) and the searches (lineFeature.cpp:120-165).
 * ------------------------------------------------------------------------------------------- */
typedef struct slam_disc_job {
    int64_t mask_off;                 /* mask in the blob, pitch w; every byte of it is written (0 or 255) */
    int32_t w, h;                     /* areaRect.width, areaRect.height */
    int32_t circle_begin, circle_count;   /* this job's (x, y, radius) triples in the circle table */
} slam_disc_job;

/* matb areaMask(h, w, 0) followed by cv::circle(areaMask, pt, radius, Scalar(255), -1) per triple
 * (lineFeature.cpp:87-96); centres are relative to the mask.  d_circles holds ncircles int32 triples.
 * d_status (optional) receives 0 / SLAM_HIP_EINVAL per job (negative radius, ranges outside the
 * tables).  max_pixels: largest w*h of the batch if known, else 0.  Asynchronous, allocates nothing. */
int slam_hip_disc_masks_d(slam_ctx* ctx, const int32_t* d_circles, int ncircles, const slam_disc_job* d_jobs,
                          int njobs, uint8_t* d_blob, size_t blob_bytes, int32_t* d_status, int max_pixels);

typedef struct slam_line_job {
    int64_t desc_off;                 /* LineFeature::descriptor in the blob, pitch dw */
    int64_t amask_off;                /* areaMask of the coarsest level in the blob, pitch aw (tested for non-zero) */
    int32_t frame;
    int32_t dw, dh;
    int32_t ax, ay, aw, ah;           /* areaRect on the coarsest level (lineFeature.cpp:86) */
    float   threshold;
    int32_t stride;
    int32_t reserved;
} slam_line_job;

/* The searches of LineFeature::track: result (x, y) = trackedPoint, response, levels_run; sub-pixel
 * offset of the last level searched.  A level whose resized descriptor would be empty gives (-1,-1),
 * response -1 (cv::resize would throw there).  The area-size cap of lineFeature.cpp:81-84 is the
 * caller's (it precedes the mask). */
/* A batch of fewer than 1536 lines (the reference tracks three per frame, new_image.cpp:80) has each line's coarsest
 * search cut into up to 8 bands of rows, one wavefront each (one launch more; same results); the band results live in
 * a buffer of the context that grows when a batch needs more than any before it - the only allocation this entry point
 * can make.  A batch of 1024 lines or more has its coarsest searches done by a leaner kernel first (window energies
 * from box sums of the squared image, lineFeature.cpp:124-127 searches without a patch mask), with the same record per
 * line in the same buffer; same results. */
int slam_hip_line_track_batch_d(slam_ctx* ctx, const slam_pyr* pyr, const slam_line_job* d_jobs,
                                int njobs, const uint8_t* d_blob, size_t blob_bytes,
                                slam_track_result* d_out);
int slam_hip_line_track_batch(slam_ctx* ctx, const slam_pyr* pyr, const slam_line_job* jobs, int njobs,
                              const uint8_t* blob, size_t blob_bytes, slam_track_result* out);
/* The per-frame loop of SLAM::matchLines (matching.cpp:279-292), host tables: job i's area mask is
 * rasterised on the device from circle triples [circle_begin[i], circle_begin[i+1]) (centres relative
 * to the job's area rectangle; circle_begin has njobs+1 entries) and searched there; jobs[i].amask_off
 * is assigned here, jobs[i].desc_off points into desc_blob. */
int slam_hip_match_lines_batch(slam_ctx* ctx, const slam_pyr* pyr, const slam_line_job* jobs, int njobs,
                               const int32_t* circles, int ncircles, const int32_t* circle_begin,
                               const uint8_t* desc_blob, size_t desc_bytes, slam_track_result* out);

/* ---------------------------------------------------------------------------------------------
 * Uncertainty windows — the coarsest-level search window of a feature whose predicted point has a
 * 2x2 covariance (the EKF's innovation covariance): the gate ellipse
 *     { p : (p - p2d)^T S^-1 (p - p2d) <= gate }
 * scaled to the coarsest level, its bounding rectangle and its rasterised area mask, ready for
 * slam_track_job's SLAM_HIP_TRACK_COARSE_AREA fields.  The reference has no live counterpart: its
 * windows are the fixed square of feature.cpp:134,144-146 and the disc unions of lineFeature.cpp:86-96;
 * its density rasteriser SLAM::projectGaussian (new_point.cpp:31-107) is never called.  The rectangle
 * follows the floor / ceil(+1) convention of lineFeature.cpp:60-63 and the 300-pixel cap of
 * lineFeature.cpp:27,81-84.
 * ------------------------------------------------------------------------------------------- */
typedef struct slam_ellipse_job {
    int64_t mask_off;                 /* mask in the blob, pitch w; every byte of it is written (0 or 255) */
    int32_t x, y, w, h;               /* the window, in pixels of the level it was set up for */
    float   ux, uy;                   /* ellipse centre relative to (x, y) */
    float   q11, q12, q22;            /* inside <=> q11*dx*dx + 2*q12*dx*dy + q22*dy*dy <= 1 (float, this order),
                                         or the pixel is the one nearest the centre: (floor(ux+.5), floor(uy+.5)) */
    int32_t valid;                    /* 0: no window (degenerate covariance or larger than the cap) */
} slam_ellipse_job;

#define SLAM_HIP_ELLIPSE_MAX 300      /* widest / tallest window (lineFeature.cpp:27) */

/* Host-only arithmetic, no device work.  cov = {sxx, sxy, syy} in full-resolution pixels^2, gate =
 * squared Mahalanobis radius, res = 1 / subsample of the level searched (lineFeature.cpp:30).  Fills
 * every field of *out except mask_off.  Returns 0, or SLAM_HIP_EINVAL for null pointers; a covariance
 * that is not positive definite, non-finite input or a window beyond the cap gives valid = 0. */
int slam_hip_ellipse_setup(float px, float py, const float* cov, float gate, float res, slam_ellipse_job* out);

/* Rasterises the masks of all valid jobs (invalid ones are skipped).  d_status (optional) receives 0 /
 * SLAM_HIP_EINVAL per job (mask outside the blob, sizes outside 1..SLAM_HIP_ELLIPSE_MAX).
 * max_pixels: largest w*h of the batch if known, else 0.  Asynchronous, allocates nothing. */
int slam_hip_ellipse_masks_d(slam_ctx* ctx, const slam_ellipse_job* d_jobs, int njobs, uint8_t* d_blob,
                             size_t blob_bytes, int32_t* d_status, int max_pixels);

/* Feature::track for a batch of features with uncertainty windows, host tables: windows[i] (from
 * slam_hip_ellipse_setup at the pyramid's coarsest level; mask_off assigned here) becomes job i's
 * coarsest-level area and mask; a job whose window is not valid is reported as (-1,-1), response -1,
 * levels_run 0.  jobs[i].flags / amask_off / ax..ah are overwritten; proj_off / mask_off point into
 * blob. */
int slam_hip_track_ellipse_batch(slam_ctx* ctx, const slam_pyr* pyr, const slam_track_job* jobs,
                                 const slam_ellipse_job* windows, int njobs, const uint8_t* blob,
                                 size_t blob_bytes, slam_track_result* out);

/* ---------------------------------------------------------------------------------------------
 * Diagnostics (used by the parity tests; no reference counterpart).
 * ------------------------------------------------------------------------------------------- */
/* Evaluates the two score formulas of matching.cpp on the device for n triples of exact integer
 * sums (CC = <I,P>, SI = |I|^2, SP = |P|^2): f32_out = cc/(i2*p2) (matching.cpp:70-72),
 * mt_out = matchTemplate CV_TM_CCORR_NORMED normalisation (matching.cpp:152). */
int slam_hip_debug_score_formulas(slam_ctx* ctx, const uint64_t* cc, const uint64_t* si,
                                  const uint64_t* sp, int n, float* f32_out, float* mt_out);
/* cv::resize(src, dst, Size(dw, dh)) (INTER_LINEAR) of one host image on the device. */
int slam_hip_debug_resize(slam_ctx* ctx, const uint8_t* src, int sw, int sh, size_t spitch,
                          uint8_t* dst, int dw, int dh, size_t dpitch);
/* Point walks (slam_hip_track_batch[_d] and the entry points built on it) run in two kernels: a lean one for walks
 * whose every level is an interior window that fits one job's LDS, and the general one for the jobs the first leaves
 * over.  After the stream has drained, *out_deferred = how many jobs of the context's last walk batch took the
 * general kernel (0 when the lean kernel is switched off by SLAM_HIP_NO_LEAN=1 or SLAM_HIP_FORCE_DIRECT=1).
 * Synchronises the stream. */
int slam_hip_debug_walk_deferred(slam_ctx* ctx, int* out_deferred);
/* How the lean walk kernel would take one level of a walk with `budget` bytes of LDS (host arithmetic only, no device
 * needed; the kernel makes the same decision from the same sizes): template tw x th at that level, window aw x ah,
 * lattice step, `resized` != 0 when the level's template is resized from the stored one, `chunks` = the kernel's build
 * (slam_hip_set_walk_chunks).  out[0] = cell rows per block (0: the level is left to the general kernel), out[1] =
 * template rows per chunk, out[2] = LDS bytes of that plan, out[3] = 1 / 2 when the window energies come from box sums
 * (unmasked / masked template), else 0.  No counterpart in the reference: a planning aid for callers of the `_d` forms,
 * who pick budget and build themselves. */
int slam_hip_debug_walk_plan(int tw, int th, int aw, int ah, int step, int resized, int budget, int chunks, int out[4]);
/* Library build identification: "slam_hip <version> gfx950". */
const char* slam_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SLAM_HIP_H */
