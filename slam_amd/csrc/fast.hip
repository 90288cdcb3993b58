// FAST-9-16 corner score + non-max suppression (cv::FAST as called at new_point.cpp:120) and
// SLAM::getSortedKeyPoints (new_point.cpp:113-127).
//
// Kernel 1 (fast_band_kernel): one workgroup per band of R image rows of one frame.  Phase A
// writes the stored corner score (score for corners, 0 otherwise) of rows -1..R of the band into
// an LDS tile; phase B applies the 3x3 strict non-max test and compacts the survivors IN ROW-MAJOR
// ORDER (each wavefront owns R/4 consecutive rows and compacts with ballots) into an LDS key list;
// the band's keys are then appended to the frame's key pool.
// Kernel 2 (fast_gather_kernel): one workgroup per frame concatenates the bands in band order,
// which gives cv::FAST's row-major keypoint order, and converts keys to keypoints.
#include <algorithm>
#include <cmath>

#include "slam_internal.h"

namespace {

constexpr int FAST_WG = 256;
constexpr int FAST_LDS_BUDGET = 80 * 1024;

struct FastPlan {
    int R;            // rows per band (multiple of 4)
    int nbands;
    int keycap_wave;  // LDS key capacity of one wavefront
    size_t lds_bytes;
};

bool fast_plan(int w, int h, int nonmax, FastPlan& p) {
    // LDS: (R+2)*w score bytes + 4 wave lists of keycap_wave keys (4 bytes each)
    for (int R = 32; R >= 4; R -= 4) {
        const int rows_w = R / 4;
        const int kc = nonmax ? ((w + 1) / 2) * ((rows_w + 1) / 2) : w * rows_w;
        const size_t lds = (size_t)(R + 2) * w + (size_t)4 * kc * 4 + 64;
        if (lds <= (size_t)FAST_LDS_BUDGET) {
            p.R = R; p.nbands = (h + R - 1) / R; p.keycap_wave = kc; p.lds_bytes = (lds + 15) / 16 * 16;
            return true;
        }
    }
    return false;
}

// per-frame scratch layout (all 4-byte words): [0] pool counter, [1..3] pad, then band_off[nbands],
// band_cnt[nbands], then the key pool of `cap` keys
__host__ __device__ inline size_t fast_frame_words(int nbands, int cap) {
    return (size_t)4 + 2 * (size_t)((nbands + 3) / 4 * 4) + (size_t)((cap + 3) / 4 * 4);
}

__device__ __forceinline__ int ring_px(const uint8_t* p, long long pitch, int k) {
    // (dx,dy) ring of cv::FAST, k = 0..15
    constexpr int DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
    constexpr int DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
    return p[(long long)DY[k] * pitch + DX[k]];
}

__device__ __forceinline__ unsigned rot16(unsigned m, int s) { return ((m >> s) | (m << (16 - s))) & 0xffffu; }

// Stored score of pixel p at `threshold`: 0 unless 9 contiguous ring pixels are all brighter than
// v+t or all darker than v-t; then S = max(A_dark, A_bright) - 1 with A = max over the 16 arcs of
// the minimum of (v - ring) resp. (ring - v).  `raw` returns S+1 semantics for the no-NMS mode:
// out_corner says whether the pixel is a corner at all.
__device__ __forceinline__ int fast_stored_score(const uint8_t* p, long long pitch, int t, bool& corner) {
    const int v = p[0];
    int d[16];
    unsigned mb = 0, md = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        d[k] = v - ring_px(p, pitch, k);
        md |= (unsigned)(d[k] > t) << k;
        mb |= (unsigned)(d[k] < -t) << k;
    }
    auto run9 = [](unsigned m) {
        const unsigned m2 = m & rot16(m, 1);
        const unsigned m4 = m2 & rot16(m2, 2);
        const unsigned m8 = m4 & rot16(m4, 4);
        return (m8 & rot16(m, 8)) != 0;
    };
    corner = run9(md) || run9(mb);
    if (!corner) return 0;
    int ad = -256, ab = -256;
#pragma unroll
    for (int s = 0; s < 16; s++) {
        int mn = d[s], mx = d[s];
#pragma unroll
        for (int j = 1; j < 9; j++) { const int q = d[(s + j) & 15]; mn = min(mn, q); mx = max(mx, q); }
        ad = max(ad, mn); ab = max(ab, -mx);
    }
    return max(t, max(ad, ab)) - 1;   // cornerScore<16>: seeded with the threshold
}

__global__ __launch_bounds__(FAST_WG) void fast_band_kernel(
    const uint8_t* __restrict__ frames, int w, int h, long long pitch, long long frame_stride,
    int threshold, int nonmax, int R, int keycap_wave, int nbands, int cap, unsigned* __restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint8_t* sc = smem;                                        // (R+2) x w stored scores
    const size_t sc_bytes = ((size_t)(R + 2) * w + 15) / 16 * 16;
    unsigned* keys = reinterpret_cast<unsigned*>(smem + sc_bytes);   // 4 x keycap_wave
    __shared__ int wave_cnt[4];
    __shared__ int pool_off;

    const int band = blockIdx.x, frame = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint8_t* im = frames + (long long)frame * frame_stride;
    const int y0 = band * R;                                   // first row of the band
    unsigned* fs = scratch + (size_t)frame * fast_frame_words(nbands, cap);

    // phase A: stored score of band rows -1 .. R (tile row r <-> image row y0 - 1 + r)
    const int ntile = (R + 2) * w;
    for (int i = tid; i < ntile; i += FAST_WG) {
        const int r = i / w, x = i - r * w;
        const int y = y0 - 1 + r;
        int s = 0;
        if (y >= 3 && y < h - 3 && x >= 3 && x < w - 3) {
            bool corner;
            s = fast_stored_score(im + (long long)y * pitch + x, pitch, threshold, corner);
            if (!nonmax) s = corner ? (s + 1) : 0;             // keep "is a corner" distinguishable from score 0
        }
        sc[i] = (uint8_t)s;
    }
    __syncthreads();

    // phase B: each wavefront walks its R/4 rows in row-major order, 64 pixels at a time
    const int rows_w = R / 4;
    unsigned* mykeys = keys + (size_t)wid * keycap_wave;
    int cnt = 0;
    for (int rr = 0; rr < rows_w; rr++) {
        const int ry = wid * rows_w + rr;                      // row inside the band
        const int y = y0 + ry;
        if (y < 3 || y >= h - 3) continue;                     // wave-uniform
        const uint8_t* row = sc + (size_t)(ry + 1) * w;
        for (int xb = 3; xb < w - 3; xb += 64) {
            const int x = xb + lane;
            bool keep = false;
            int s = 0;
            if (x < w - 3) {
                s = row[x];
                if (nonmax) {
                    keep = s > 0 || false;
                    // a corner with stored score 0 can never be strictly greater than its
                    // neighbours' stored scores (>= 0), so s > 0 is necessary
                    keep = keep && s > row[x - 1] && s > row[x + 1] &&
                           s > row[x - w - 1] && s > row[x - w] && s > row[x - w + 1] &&
                           s > row[x + w - 1] && s > row[x + w] && s > row[x + w + 1];
                } else {
                    keep = s > 0;
                    s -= 1;
                }
            }
            const unsigned long long bal = __ballot(keep);
            if (keep) {
                const int pos = cnt + __popcll(bal & ((1ull << lane) - 1ull));
                if (pos < keycap_wave) mykeys[pos] = ((unsigned)(ry * w + x) << 8) | (unsigned)s;
            }
            cnt += __popcll(bal);
        }
    }
    if (lane == 0) wave_cnt[wid] = min(cnt, keycap_wave);
    __syncthreads();
    const int c0 = wave_cnt[0], c1 = wave_cnt[1], c2 = wave_cnt[2], c3 = wave_cnt[3];
    const int total = c0 + c1 + c2 + c3;
    if (tid == 0) {
        int off = 0;
        if (total > 0) off = (int)atomicAdd(&fs[0], (unsigned)total);
        pool_off = off;
        const int nb4 = (nbands + 3) / 4 * 4;
        fs[4 + band] = (unsigned)off;
        fs[4 + nb4 + band] = (unsigned)total;
    }
    __syncthreads();
    const int off = pool_off;
    unsigned* pool = fs + 4 + 2 * ((nbands + 3) / 4 * 4);
    const int wofs = (wid == 0) ? 0 : (wid == 1) ? c0 : (wid == 2) ? c0 + c1 : c0 + c1 + c2;
    const int mycnt = wave_cnt[wid];
    for (int i = lane; i < mycnt; i += 64) {
        const int dst = off + wofs + i;
        if (dst < cap) pool[dst] = mykeys[i];
    }
}

__global__ __launch_bounds__(FAST_WG) void fast_gather_kernel(int w, int R, int nbands, int cap,
                                                              const unsigned* __restrict__ scratch,
                                                              slam_keypoint* __restrict__ out,
                                                              int* __restrict__ counts) {
    const int frame = blockIdx.x, tid = threadIdx.x;
    const unsigned* fs = scratch + (size_t)frame * fast_frame_words(nbands, cap);
    const int nb4 = (nbands + 3) / 4 * 4;
    const unsigned* band_off = fs + 4;
    const unsigned* band_cnt = fs + 4 + nb4;
    const unsigned* pool = fs + 4 + 2 * nb4;
    slam_keypoint* o = out + (size_t)frame * cap;
    int base = 0;   // every thread walks the bands; counts are tiny compared with the copies
    for (int b = 0; b < nbands; b++) {
        const int cnt = (int)band_cnt[b], off = (int)band_off[b];
        for (int i = tid; i < cnt; i += FAST_WG) {
            const int dst = base + i;
            if (dst < cap && off + i < cap) {
                const unsigned key = pool[off + i];
                const unsigned lin = key >> 8;
                const int ry = (int)(lin / (unsigned)w), x = (int)(lin - (unsigned)ry * (unsigned)w);
                slam_keypoint kp;
                kp.x = (float)x; kp.y = (float)(b * R + ry); kp.response = (float)(key & 0xffu);
                o[dst] = kp;
            }
        }
        base += cnt;
    }
    if (tid == 0) counts[frame] = base;
}

}  // namespace

extern "C" size_t slam_hip_fast_scratch_bytes(int w, int h, int nframes, int cap_per_frame) {
    FastPlan p;
    if (w < 1 || h < 1 || nframes < 1 || cap_per_frame < 1) return 0;
    // the no-NMS plan has the smaller bands, i.e. the larger band tables: size for it
    if (!fast_plan(w, h, 0, p)) return 0;
    return fast_frame_words(p.nbands, cap_per_frame) * 4 * (size_t)nframes;
}

extern "C" int slam_hip_fast_detect_d(slam_ctx* ctx, const uint8_t* d_frames, int w, int h, size_t pitch,
                                      size_t frame_stride, int nframes, int threshold, int nonmax,
                                      slam_keypoint* d_out, int cap_per_frame, int32_t* d_counts,
                                      void* d_scratch, size_t scratch_bytes) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (!d_frames || !d_out || !d_counts || !d_scratch || w < 1 || h < 1 || nframes < 1 || cap_per_frame < 1 ||
        pitch < (size_t)w || (nframes > 1 && frame_stride < pitch * (size_t)h) || w > 16384 || h > (1 << 20) ||
        cap_per_frame > (1 << 26))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "fast_detect: bad arguments");
    FastPlan p;
    if (!fast_plan(w, h, nonmax != 0, p))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "fast_detect: image too wide (%d) for the band kernel", w);
    const size_t need = fast_frame_words(p.nbands, cap_per_frame) * 4 * (size_t)nframes;
    if (scratch_bytes < need || ((uintptr_t)d_scratch & 15))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "fast_detect: scratch too small or misaligned (%zu < %zu)", scratch_bytes, need);
    threshold = std::min(std::max(threshold, 0), 255);          // cv::FAST clamps the threshold
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // zero the pool counters (first word of every frame record); the records are 16-byte multiples
    SLAM_HIP_CHECK(ctx, hipMemset2DAsync(d_scratch, fast_frame_words(p.nbands, cap_per_frame) * 4, 0, 16, (size_t)nframes, ctx->stream));
    hipLaunchKernelGGL(fast_band_kernel, dim3(p.nbands, nframes), dim3(FAST_WG), p.lds_bytes, ctx->stream,
                       d_frames, w, h, (long long)pitch, (long long)frame_stride, threshold, nonmax ? 1 : 0,
                       p.R, p.keycap_wave, p.nbands, cap_per_frame, (unsigned*)d_scratch);
    SLAM_HIP_CHECK(ctx, hipGetLastError());
    hipLaunchKernelGGL(fast_gather_kernel, dim3(nframes), dim3(FAST_WG), 0, ctx->stream, w, p.R, p.nbands,
                       cap_per_frame, (const unsigned*)d_scratch, d_out, d_counts);
    SLAM_HIP_CHECK(ctx, hipGetLastError());
    return SLAM_HIP_OK;
}

// upload + detect + download for frames that are already resident at d_frames
static int fast_detect_resident(slam_ctx* ctx, const uint8_t* d_frames, int w, int h, size_t dpitch, size_t dstride,
                                int nframes, int threshold, int nonmax, slam_keypoint* out, int cap, int32_t* counts) {
    const size_t sb = slam_hip_fast_scratch_bytes(w, h, nframes, cap);
    if (!sb) return slam_set_error(ctx, SLAM_HIP_EINVAL, "fast_detect: unsupported shape %dx%d", w, h);
    const size_t ob = sizeof(slam_keypoint) * (size_t)cap * nframes, cb = 4 * (size_t)nframes;
    int rc;
    if ((rc = slam_stage_reserve(ctx, 1, sb)) || (rc = slam_stage_reserve(ctx, 2, ob)) || (rc = slam_stage_reserve(ctx, 3, cb)))
        return rc;
    rc = slam_hip_fast_detect_d(ctx, d_frames, w, h, dpitch, dstride, nframes, threshold, nonmax,
                                (slam_keypoint*)ctx->d_stage[2], cap, (int32_t*)ctx->d_stage[3],
                                ctx->d_stage[1], ctx->stage_bytes[1]);
    if (rc != SLAM_HIP_OK) return rc;
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(counts, ctx->d_stage[3], cb, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    bool trunc = false;
    for (int f = 0; f < nframes; f++) {
        const int n = std::min(counts[f], cap);
        trunc = trunc || counts[f] > cap;
        if (n > 0)
            SLAM_HIP_CHECK(ctx, hipMemcpyAsync(out + (size_t)f * cap, (slam_keypoint*)ctx->d_stage[2] + (size_t)f * cap,
                                               sizeof(slam_keypoint) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    }
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (trunc) return slam_set_error(ctx, SLAM_HIP_ECAP, "fast_detect: more keypoints than cap_per_frame=%d", cap);
    return SLAM_HIP_OK;
}

static int fast_upload(slam_ctx* ctx, const uint8_t* h_frames, int w, int h, size_t pitch, size_t frame_stride,
                       int nframes, size_t& dpitch, size_t& dstride) {
    dpitch = ((size_t)w + 63) / 64 * 64;
    dstride = dpitch * (size_t)h;
    int rc = slam_stage_reserve(ctx, 0, dstride * (size_t)nframes);
    if (rc != SLAM_HIP_OK) return rc;
    for (int f = 0; f < nframes; f++)
        SLAM_HIP_CHECK(ctx, hipMemcpy2DAsync((uint8_t*)ctx->d_stage[0] + (size_t)f * dstride, dpitch,
                                             h_frames + (size_t)f * frame_stride, pitch, (size_t)w, (size_t)h,
                                             hipMemcpyHostToDevice, ctx->stream));
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_fast_detect(slam_ctx* ctx, const uint8_t* h_frames, int w, int h, size_t pitch,
                                    size_t frame_stride, int nframes, int threshold, int nonmax,
                                    slam_keypoint* out, int cap_per_frame, int32_t* counts) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (!h_frames || !out || !counts || w < 1 || h < 1 || nframes < 1 || cap_per_frame < 1 || pitch < (size_t)w ||
        (nframes > 1 && frame_stride < pitch * (size_t)h))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "fast_detect: bad arguments");
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    size_t dpitch, dstride;
    int rc = fast_upload(ctx, h_frames, w, h, pitch, frame_stride, nframes, dpitch, dstride);
    if (rc != SLAM_HIP_OK) return rc;
    return fast_detect_resident(ctx, (const uint8_t*)ctx->d_stage[0], w, h, dpitch, dstride, nframes,
                                threshold, nonmax, out, cap_per_frame, counts);
}

extern "C" int slam_hip_get_sorted_keypoints(slam_ctx* ctx, const uint8_t* im, int w, int h, size_t pitch,
                                             size_t n_min_pts, float* fast_threshold,
                                             slam_keypoint* out, int cap, int32_t* n_out) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (!im || !fast_threshold || !out || !n_out || w < 1 || h < 1 || cap < 1 || pitch < (size_t)w)
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "get_sorted_keypoints: bad arguments");
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    size_t dpitch, dstride;
    int rc = fast_upload(ctx, im, w, h, pitch, pitch * (size_t)h, 1, dpitch, dstride);
    if (rc != SLAM_HIP_OK) return rc;
    // non-max-suppressed corners are never 8-adjacent: at most ceil(w/2)*ceil(h/2) of them
    const size_t kmax = (size_t)((w + 1) / 2) * (size_t)((h + 1) / 2);
    std::vector<slam_keypoint> kps(kmax ? kmax : 1);
    float f = *fast_threshold;
    size_t n = 0;
    while (n < n_min_pts) {                                   // new_point.cpp:116-122
        const int t = (int)f;                                 // FAST(im, out, fastThreshold, true): float -> int
        int32_t cnt = 0;
        rc = fast_detect_resident(ctx, (const uint8_t*)ctx->d_stage[0], w, h, dpitch, dstride, 1, t, 1,
                                  kps.data(), (int)kps.size(), &cnt);
        if (rc != SLAM_HIP_OK) return rc;
        n = (size_t)cnt;
        f *= 0.5f;
        if (t <= 0) break;   // the reference never leaves this loop on a corner-poor image; stop after t = 0
    }
    f *= 2.f;                                                 // new_point.cpp:123
    // new_point.cpp:124 sorts by response only with an unstable sort; ties are fixed as y, x ascending
    std::sort(kps.begin(), kps.begin() + n, [](const slam_keypoint& a, const slam_keypoint& b) {
        if (a.response != b.response) return a.response > b.response;
        if (a.y != b.y) return a.y < b.y;
        return a.x < b.x;
    });
    if (n > 2 * n_min_pts) f = kps[(size_t)((double)n_min_pts * 1.9)].response;   // new_point.cpp:125-126
    *fast_threshold = f;
    *n_out = (int32_t)n;
    for (size_t i = 0; i < n && i < (size_t)cap; i++) out[i] = kps[i];
    return SLAM_HIP_OK;
}
