// Internal declarations shared by the HIP translation units of libslam_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "slam_hip.h"

struct slam_pyr;

struct slam_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    char err[512] = {0};
    // A CU's 160 KB of LDS are handed out in 128 granules of 1280 bytes: a workgroup that asks for n bytes occupies
    // ceil(n / 1280) of them, so budgets are multiples of 1280 (less any static LDS of the kernel).
    int line_lds_bytes = 12800;        // line walks: 10 granules, 12 jobs per CU (3 per SIMD, as many as the registers allow), large descriptors
    int match_lds_bytes = 9664;        // LDS per search job: with track_kernel's 576 static bytes 8 granules, 16 jobs per CU, as many as the registers allow (SLAM_HIP_MATCH_LDS_KB / slam_hip_set_match_lds)
    int force_direct = 0;   // SLAM_HIP_FORCE_DIRECT=1: search by direct evaluation (cross-check of the tiled core)
    int lean_lds_bytes = 7680;         // LDS per job of the lean walk kernel: 6 granules, 21 jobs per CU
    int lean_lds_user = 0;             // set by the caller (slam_hip_set_walk_lds, SLAM_HIP_WALK_LDS_KB): never sized automatically
    int lean_lds_once = 0;             // budget of the next walk launch only (slam_size_walk_lds), 0 = lean_lds_bytes
    int line_lean_lds_bytes = 12800;   // LDS per line of line_lean_coarse_kernel: 10 granules, 12 lines per CU (SLAM_HIP_LINE_LEAN_LDS_BYTES)
    int line_lean_min_jobs = 1024;     // line batches from this size on search their coarsest level with the lean core (SLAM_HIP_LINE_LEAN_MIN)
    int walk_band_lds_bytes = 65536;   // LDS per wavefront of the banded walk kernels (SLAM_HIP_WALK_BAND_LDS_BYTES): such batches do not fill the chip
    int walk_bands = 0;                // bands per walk of small batches under the second build, 0 = by batch size (SLAM_HIP_WALK_BANDS)
    int walk_chunks = 0;               // lean walk kernel built with template chunks (slam_hip_set_walk_chunks, SLAM_HIP_WALK_CHUNKS)
    int walk_chunks_once = 0;          // the same for the next walk launch only (slam_size_walk_lds)
    int no_lean = 0;                   // SLAM_HIP_NO_LEAN=1: walks and searches skip the lean kernels (cross-check of them)
    int lean_match_lds_bytes = 7680;   // LDS per job of the lean single-search kernel: 6 granules; its registers allow 20 jobs per CU (SLAM_HIP_SEARCH_LDS_KB / _BYTES)
    int lean_match_min_jobs = 64;      // smaller batches of single searches take the general kernel alone (SLAM_HIP_SEARCH_LEAN_MIN)
    // queue of walk jobs the lean kernel leaves to the general one: a counter word, then the job indices
    int* d_queue = nullptr;            // 16 words (slot counter, -, last launch's count), then the list of queued jobs (index + 1, 0 = empty)
    size_t queue_jobs = 0;
    int queue_last = 0;                // queue of the last launch: 0 = d_queue, -1 = the device's no-alloc scratch
    // band results of the coarsest level of small line-walk batches (line_coarse_kernel), grown on demand
    void* d_bands = nullptr;
    size_t bands_bytes = 0;
    int line_bands = 0;                // SLAM_HIP_LINE_BANDS: bands per line (1 = never cut), 0 = by batch size
    // one-level pyramid reused by slam_hip_match_feature_in_area while the image size stays the same
    slam_pyr* mfa_pyr = nullptr;
    int mfa_w = 0, mfa_h = 0;
    // pinned host staging for image uploads (slam_upload_image)
    uint8_t* h_pin = nullptr;
    size_t pin_bytes = 0;
    // growable device staging buffers used by the host-table entry points
    void* d_stage[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t stage_bytes[4] = {0, 0, 0, 0};
};

// One level of a pyramid as the kernels see it (passed by value inside slam_pyr_dev).
struct slam_level_dev {
    const uint8_t* data;
    int w, h;
    long long pitch, frame_stride;
    float sub;
    // derived on the host when the pyramid is created
    float inv_sub;       // 1 / sub
    float rad;           // Feature::track's search radius at this level (feature.cpp:134,136), before the level-0 stride factor
    int pow2;            // sub is a power of two: x / sub == x * inv_sub exactly
};

struct slam_pyr_dev {
    int nlev;
    int nframes;
    slam_level_dev lev[SLAM_HIP_MAX_LEVELS];
};

// Tile shape of the pyramid kernel for one resized level.
struct slam_pyr_plan {
    int xt = 0, ncolt = 0, rb = 0, nband = 0, span = 0, lds = 0;
    int box = 0;    // 2, 4 or 8: exact integer reduction with regular taps (box-filter path)
};

struct slam_pyr {
    int device = 0;
    int w = 0, h = 0, nlev = 0, nframes = 0;
    float subs[SLAM_HIP_MAX_LEVELS] = {0};
    int lw[SLAM_HIP_MAX_LEVELS] = {0}, lh[SLAM_HIP_MAX_LEVELS] = {0};
    size_t lpitch[SLAM_HIP_MAX_LEVELS] = {0}, lstride[SLAM_HIP_MAX_LEVELS] = {0};
    uint8_t* d_level[SLAM_HIP_MAX_LEVELS] = {nullptr};   // owned buffers (nullptr for aliased levels)
    // bilinear tap tables per resized level, device memory: xofs[dw], xcoef[dw] (a0 | a1<<16),
    // yofs[dh], ycoef[dh]
    int32_t* d_xofs[SLAM_HIP_MAX_LEVELS] = {nullptr};
    uint32_t* d_xcoef[SLAM_HIP_MAX_LEVELS] = {nullptr};
    int32_t* d_yofs[SLAM_HIP_MAX_LEVELS] = {nullptr};
    uint32_t* d_ycoef[SLAM_HIP_MAX_LEVELS] = {nullptr};
    slam_pyr_plan plan[SLAM_HIP_MAX_LEVELS];
    uint8_t* d_upload = nullptr;     // owned copy of host frames (slam_hip_pyramid_build)
    size_t upload_bytes = 0;
    const uint8_t* d_frames = nullptr;   // current level-0 source
    size_t pitch0 = 0, stride0 = 0;
    bool built = false;
    slam_pyr_dev dev{};
};

int slam_set_error(slam_ctx* ctx, int code, const char* fmt, ...);
int slam_stage_reserve(slam_ctx* ctx, int slot, size_t bytes);
int slam_queue_reserve(slam_ctx* ctx, int njobs);
// Queues the upload of `rows` rows of `width` bytes (host pitch spitch) to device rows of pitch dpitch.  A 2-D copy from
// pageable memory is carried out row by row by the runtime (about 9 us per row); here the rows are packed into the
// context's pinned buffer first and leave in one transfer.  The pinned buffer is reused by the next call: callers
// synchronise the stream before they return (all host-image entry points do).  slot_off: byte offset inside the
// pinned buffer for this image, so that one call can queue several.
int slam_upload_image(slam_ctx* ctx, uint8_t* d_dst, size_t dpitch, const uint8_t* h_src, size_t spitch, size_t width, size_t rows,
                      size_t slot_off);
int slam_track_scratch_init(slam_ctx* ctx);
void slam_size_walk_lds(slam_ctx* ctx, const slam_pyr* pyr, const slam_track_job* jobs, int njobs);
// clear_first: clear the queue's counters with a memset before the launch (a queue several contexts share)
int slam_launch_track(slam_ctx* ctx, const slam_pyr* pyr, const slam_track_job* d_jobs, int njobs, const uint8_t* d_blob,
                      size_t blob_bytes, slam_track_result* d_out, int* queue, bool clear_first);

#define SLAM_HIP_CHECK(ctx, call)                                                              \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return slam_set_error((ctx), SLAM_HIP_EHIP, "%s failed: %s (%s:%d)", #call,        \
                                  hipGetErrorString(e__), __FILE__, __LINE__);                 \
    } while (0)

// Host-table entry points queue asynchronous copies from caller memory or from local vectors and synchronise before
// they return; this guard makes every early (error) return wait for the stream as well, so that no copy is still
// reading a buffer that is about to go out of scope.
struct slam_stream_drain {
    slam_ctx* ctx;
    explicit slam_stream_drain(slam_ctx* c) : ctx(c) {}
    ~slam_stream_drain() { (void)hipStreamSynchronize(ctx->stream); }
    slam_stream_drain(const slam_stream_drain&) = delete;
    slam_stream_drain& operator=(const slam_stream_drain&) = delete;
};

// Bilinear tap tables of cv::resize(INTER_LINEAR) for 8-bit images, computed on the host in
// double/float exactly as OpenCV 2.4's resize() does (11-bit fixed-point coefficients).
void slam_resize_tables(int ssize, int dsize, std::vector<int32_t>& ofs, std::vector<uint32_t>& coef,
                        bool clamp_fraction);

// kernels' host launchers (defined in the .hip files)
int slam_launch_pyramid(slam_ctx* ctx, slam_pyr* pyr);
int slam_pyr_plan_level(const std::vector<int32_t>& xofs, int sw, int dw, int dh, slam_pyr_plan* out);
int slam_pyr_detect_box(const std::vector<int32_t>& xofs, const std::vector<uint32_t>& xcoef,
                        const std::vector<int32_t>& yofs, const std::vector<uint32_t>& ycoef,
                        int sw, int sh, int dw, int dh);

static_assert(sizeof(slam_match_job) == 72 && sizeof(slam_match_result) == 24, "ABI struct layout");
static_assert(sizeof(slam_track_job) == 72 && sizeof(slam_track_result) == 28, "ABI struct layout");
static_assert(sizeof(slam_keypoint) == 12, "ABI struct layout");
static_assert(sizeof(slam_desc_job) == 32 && sizeof(slam_desc_result) == 20, "ABI struct layout");
static_assert(sizeof(slam_warp_job) == 80, "ABI struct layout");
static_assert(sizeof(slam_disc_job) == 24 && sizeof(slam_line_job) == 56 && sizeof(slam_ellipse_job) == 48, "ABI struct layout");
