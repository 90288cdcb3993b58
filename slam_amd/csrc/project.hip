// Template projection — SLAM::Feature::project (feature.cpp:53-100): the small float matrix steps on
// the host (feature.cpp:61-94), the two cv::warpPerspective calls (feature.cpp:96-98) batched on the
// device, written straight into the byte blob the search kernels read their templates from.
#include <algorithm>
#include <cmath>

#include "slam_internal.h"

namespace {

constexpr int WARP_WG = 256;

// Destination columns per block of OpenCV 2.4's warpPerspective (imgwarp.cpp): bh0 = min(16, h),
// bw0 = min(1024 / bh0, w).  The per-row constants are formed at a block's first column, so the block
// width is part of the arithmetic.
__device__ __forceinline__ int warp_block_w(int ow, int oh) { return min(1024 / min(16, oh), ow); }

__device__ __forceinline__ int sat_s16(int v) { return min(max(v, -32768), 32767); }

// One workgroup per (job, slice of the output): every lane produces pixels of proj
// (WARP_INVERSE_MAP | INTER_LINEAR) and of mask (INTER_NEAREST of an all-ones image), both with the
// constant border 0.  All coordinate arithmetic is OpenCV's, in double, one IEEE operation per step:
//   X0 = M0*x + M1*y + M2 at the block's first column x, then (X0 + M0*x1) * W,
//   W = 32 / (W0 + M6*x1) (linear) or 1 / (...) (nearest), 0 when the denominator is 0,
//   clamped to the int range and rounded half-to-even; linear keeps 5 fractional bits per axis and
//   weights the four taps with the 15-bit table (32-fx)(32-fy)*32 ..., result (sum + 16384) >> 15.
__global__ __launch_bounds__(WARP_WG) void warp_kernel(const uint8_t* __restrict__ desc, unsigned long long desc_bytes,
                                                       const slam_warp_job* __restrict__ jobs, int njobs,
                                                       uint8_t* __restrict__ out, unsigned long long out_bytes,
                                                       int32_t* __restrict__ status) {
    const int j = blockIdx.x;
    if (j >= njobs) return;
    const slam_warp_job job = jobs[j];
    const int ow = job.ow, oh = job.oh, dw = job.dw, dh = job.dh;
    if (ow == 0 || oh == 0) {                     // empty projection (feature.cpp:84-86): nothing to write
        if (blockIdx.y == 0 && threadIdx.x == 0 && status) status[j] = SLAM_HIP_OK;
        return;
    }
    const unsigned long long npix = (unsigned long long)ow * (unsigned long long)oh;
    bool ok = ow > 0 && oh > 0 && ow <= 4096 && oh <= 4096 && dw >= 1 && dh >= 1 && dw <= 32767 && dh <= 32767 &&
              job.desc_off >= 0 && (unsigned long long)job.desc_off + (unsigned long long)dw * dh <= desc_bytes &&
              job.proj_off >= 0 && (unsigned long long)job.proj_off + npix <= out_bytes &&
              (job.mask_off == -1 || (job.mask_off >= 0 && (unsigned long long)job.mask_off + npix <= out_bytes));
    for (int k = 0; k < 9; k++) ok = ok && fabsf(job.m[k]) < 1e30f;
    if (!ok) {
        if (blockIdx.y == 0 && threadIdx.x == 0 && status) status[j] = SLAM_HIP_EINVAL;
        return;
    }
    if (blockIdx.y == 0 && threadIdx.x == 0 && status) status[j] = SLAM_HIP_OK;
    const double M0 = job.m[0], M1 = job.m[1], M2 = job.m[2], M3 = job.m[3], M4 = job.m[4], M5 = job.m[5],
                 M6 = job.m[6], M7 = job.m[7], M8 = job.m[8];
    const int bw0 = warp_block_w(ow, oh);
    const uint8_t* S = desc + job.desc_off;
    uint8_t* P = out + job.proj_off;
    uint8_t* K = job.mask_off >= 0 ? out + job.mask_off : nullptr;
    const int total = (int)npix;
    const int nthr = blockDim.x;                  // 64 for batches of small outputs, WARP_WG otherwise
    // floor(n / d) through one reciprocal: exact while n < 2^20 ((n + 0.5) / d stays 0.5/d away from an integer, far
    // more than the float error at these magnitudes); larger outputs take the integer division
    const bool small = total < (1 << 20);
    const float row = __builtin_amdgcn_rcpf((float)ow), rbw = __builtin_amdgcn_rcpf((float)bw0);
    for (int i = blockIdx.y * nthr + threadIdx.x; i < total; i += gridDim.y * nthr) {
        const int y = small ? (int)(((float)i + 0.5f) * row) : i / ow, x = i - y * ow;
        const int xb = (int)(((float)x + 0.5f) * rbw) * bw0, x1 = x - xb;
        const double X0 = M0 * xb + M1 * y + M2;
        const double Y0 = M3 * xb + M4 * y + M5;
        const double W0 = M6 * xb + M7 * y + M8;
        const double Wd = W0 + M6 * x1;
        const double nx = X0 + M0 * x1, ny = Y0 + M3 * x1;
        const double lo = -2147483648.0, hi = 2147483647.0;
        // One division serves both maps: 32/Wd == 32 * (1/Wd) bit for bit, a scaling by a power of two being
        // exact (|Wd| is far from the overflow and underflow ranges: the matrix entries are floats below 1e30
        // and the coordinates are below 4096).
        const double inv = Wd != 0.0 ? 1.0 / Wd : 0.0;
        {
            const double W = 32.0 * inv;
            double fX = nx * W, fY = ny * W;
            fX = fX < hi ? fX : hi; fX = fX > lo ? fX : lo;       // std::max(INT_MIN, std::min(INT_MAX, v))
            fY = fY < hi ? fY : hi; fY = fY > lo ? fY : lo;
            const int X = __double2int_rn(fX), Y = __double2int_rn(fY);
            const int sx = sat_s16(X >> 5), sy = sat_s16(Y >> 5);
            const int ax = X & 31, ay = Y & 31;
            const bool x0in = (unsigned)sx < (unsigned)dw, x1in = (unsigned)(sx + 1) < (unsigned)dw;
            const bool y0in = (unsigned)sy < (unsigned)dh, y1in = (unsigned)(sy + 1) < (unsigned)dh;
            const int v0 = (x0in && y0in) ? S[sy * dw + sx] : 0;
            const int v1 = (x1in && y0in) ? S[sy * dw + sx + 1] : 0;
            const int v2 = (x0in && y1in) ? S[(sy + 1) * dw + sx] : 0;
            const int v3 = (x1in && y1in) ? S[(sy + 1) * dw + sx + 1] : 0;
            const int top = v0 * (32 - ax) + v1 * ax, bot = v2 * (32 - ax) + v3 * ax;
            P[i] = (uint8_t)(((top * (32 - ay) + bot * ay) * 32 + (1 << 14)) >> 15);
        }
        if (K) {
            double fX = nx * inv, fY = ny * inv;
            fX = fX < hi ? fX : hi; fX = fX > lo ? fX : lo;
            fY = fY < hi ? fY : hi; fY = fY > lo ? fY : lo;
            const int sx = sat_s16(__double2int_rn(fX)), sy = sat_s16(__double2int_rn(fY));
            K[i] = ((unsigned)sx < (unsigned)dw && (unsigned)sy < (unsigned)dh) ? 1 : 0;
        }
    }
}

}  // namespace

// feature.cpp:61-94 with OpenCV 2.4's arithmetic for each matrix step:
//   P*B (3x4 by 4x3): cv::gemm's generic CV_32F path, products summed in double over k, rounded to float;
//   A*p and Am*p (3x3 by 3x1): gemm's small-matrix path, a0*b0 + a1*b1 + a2*b2 in float;
//   row /= s: convertTo(row, -1, 1./s), i.e. times (float)(1./(double)s) in float;
//   A.inv(): cv::invert's closed 3x3 form, cofactors and determinant in double, entries rounded to float.
extern "C" int slam_hip_project_setup(const float* P, const float* B, const float* p3d, int dw, int dh,
                                      int32_t* loc, float* m, int32_t* valid) {
    if (!P || !B || !p3d || !loc || !m || !valid) return SLAM_HIP_EINVAL;
    *valid = 0;
    loc[0] = loc[1] = loc[2] = loc[3] = 0;
    for (int k = 0; k < 9; k++) m[k] = 0.f;
    if (dw <= 0 || dh <= 0) return SLAM_HIP_OK;                  // feature.cpp:56-59
    float Bp[4][3];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 3; c++) Bp[r][c] = B[r * 3 + c];
    for (int r = 0; r < 3; r++) Bp[r][2] = p3d[r];               // feature.cpp:61
    float A[3][3];
    for (int r = 0; r < 3; r++)                                  // feature.cpp:62
        for (int c = 0; c < 3; c++) {
            double acc = 0.0;
            for (int k = 0; k < 4; k++) acc += (double)P[r * 4 + k] * (double)Bp[k][c];
            A[r][c] = (float)acc;
        }
    const int dx = dw / 2, dy = dh / 2;                          // feature.cpp:63
    const int sgn[4][2] = {{-1, -1}, {-1, 1}, {1, 1}, {1, -1}};  // feature.cpp:64
    int lim[4] = {1000000, -1000000, 1000000, -1000000};         // xmin, xmax, ymin, ymax (feature.cpp:73)
    for (int i = 0; i < 4; i++) {                                // feature.cpp:67-79
        const float gx = (float)(sgn[i][0] * dx), gy = (float)(sgn[i][1] * dy);
        float c[3];
        for (int r = 0; r < 3; r++) c[r] = A[r][0] * gx + A[r][1] * gy + A[r][2] * 1.0f;
        const float s = (float)(1.0 / (double)c[2]);
        const float u = c[0] * s, v = c[1] * s;
        if (!(std::fabs(u) < 1e9f) || !(std::fabs(v) < 1e9f)) return SLAM_HIP_OK;   // no int holds it: empty
        lim[0] = std::min(lim[0], (int)std::ceil((double)u)); lim[1] = std::max(lim[1], (int)std::floor((double)u));
        lim[2] = std::min(lim[2], (int)std::ceil((double)v)); lim[3] = std::max(lim[3], (int)std::floor((double)v));
    }
    loc[0] = lim[0]; loc[1] = lim[2];                            // feature.cpp:80-83
    loc[2] = std::max(0, lim[1] - lim[0]); loc[3] = std::max(0, lim[3] - lim[2]);
    if (loc[2] == 0 || loc[3] == 0 || loc[2] > 500 || loc[3] > 500) return SLAM_HIP_OK;   // feature.cpp:84-86
    double a[3][3];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) a[r][c] = (double)A[r][c];
    double det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                 a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
    float Am[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};          // feature.cpp:90
    if (det != 0.0) {
        det = 1.0 / det;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) {
                // entry (r,c) of the inverse = cofactor (c,r) / det, with the index order of cv::invert
                const int r1 = (c + 1) % 3, r2 = (c + 2) % 3, c1 = (r + 1) % 3, c2 = (r + 2) % 3;
                Am[r][c] = (float)((a[r1][c1] * a[r2][c2] - a[r1][c2] * a[r2][c1]) * det);
            }
    }
    const float ox = (float)lim[0], oy = (float)lim[2];          // feature.cpp:91-92
    float col[3];
    for (int r = 0; r < 3; r++) col[r] = Am[r][0] * ox + Am[r][1] * oy + Am[r][2] * 1.0f;
    for (int r = 0; r < 3; r++) Am[r][2] = col[r];
    for (int c = 0; c < 3; c++) Am[0][c] = Am[0][c] + Am[2][c] * (float)dx;   // feature.cpp:93
    for (int c = 0; c < 3; c++) Am[1][c] = Am[1][c] + Am[2][c] * (float)dy;   // feature.cpp:94
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            if (!(std::fabs(Am[r][c]) < 1e30f)) { loc[2] = loc[3] = 0; return SLAM_HIP_OK; }
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) m[r * 3 + c] = Am[r][c];
    *valid = 1;
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_warp_batch_d(slam_ctx* ctx, const uint8_t* d_desc, size_t desc_bytes,
                                     const slam_warp_job* d_jobs, int njobs, uint8_t* d_out, size_t out_bytes,
                                     int32_t* d_status, int max_out_pixels) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (njobs < 0 || (njobs > 0 && (!d_desc || !d_jobs || !d_out)))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "warp_batch: null table or negative count");
    if (njobs == 0) return SLAM_HIP_OK;
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // slices of 16 pixels per lane; a job larger than the hint is still completed (grid-stride loop).
    // Outputs known to be small get one wavefront each: a 21x21 projection fills 7 passes of 64 lanes
    // but only 1.7 passes of 256.
    const int nthr = (max_out_pixels > 0 && max_out_pixels <= 1024) ? 64 : WARP_WG;
    int slices = max_out_pixels > 0 ? (max_out_pixels + nthr * 16 - 1) / (nthr * 16) : 1;
    slices = std::min(std::max(slices, 1), 64);
    hipLaunchKernelGGL(warp_kernel, dim3(njobs, slices), dim3(nthr), 0, ctx->stream, d_desc,
                       (unsigned long long)desc_bytes, d_jobs, njobs, d_out, (unsigned long long)out_bytes, d_status);
    SLAM_HIP_CHECK(ctx, hipGetLastError());
    return SLAM_HIP_OK;
}

// Packs the outputs of all jobs (proj then mask, rows packed) into one blob layout; returns its size.
static size_t assign_outputs(slam_warp_job* jobs, int njobs, int* max_pix) {
    size_t off = 0;
    *max_pix = 0;
    for (int i = 0; i < njobs; i++) {
        const bool live = jobs[i].ow > 0 && jobs[i].oh > 0 && jobs[i].ow <= 4096 && jobs[i].oh <= 4096;
        const size_t n = live ? (size_t)jobs[i].ow * jobs[i].oh : 0;
        jobs[i].proj_off = (int64_t)off; off += n;
        jobs[i].mask_off = (int64_t)off; off += n;
        *max_pix = std::max(*max_pix, (int)n);
    }
    return off;
}

extern "C" int slam_hip_project_track_batch(slam_ctx* ctx, const slam_pyr* pyr, const slam_warp_job* wjobs,
                                            const slam_track_job* tjobs, int njobs, const uint8_t* desc_blob,
                                            size_t desc_bytes, slam_track_result* out) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (njobs < 0 || (njobs > 0 && (!wjobs || !tjobs || !desc_blob || !out)))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "project_track_batch: null table or negative count");
    if (!pyr) return slam_set_error(ctx, SLAM_HIP_EINVAL, "project_track_batch: no pyramid");
    if (njobs == 0) return SLAM_HIP_OK;
    std::vector<slam_warp_job> wj(wjobs, wjobs + njobs);
    std::vector<slam_track_job> tj(tjobs, tjobs + njobs);
    int max_pix = 0;
    const size_t out_bytes = assign_outputs(wj.data(), njobs, &max_pix);
    for (int i = 0; i < njobs; i++) {           // the templates of the search are the outputs of the warp
        const bool live = wj[i].ow > 0 && wj[i].oh > 0;
        tj[i].projw = live ? wj[i].ow : 0; tj[i].projh = live ? wj[i].oh : 0;
        tj[i].proj_off = wj[i].proj_off; tj[i].mask_off = wj[i].mask_off;
    }
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    slam_stream_drain drain(ctx);        // error returns below wait for the copies queued so far
    const size_t wb = sizeof(slam_warp_job) * (size_t)njobs, tb = sizeof(slam_track_job) * (size_t)njobs;
    const size_t rb = sizeof(slam_track_result) * (size_t)njobs, sb = 4 * (size_t)njobs;
    const size_t o_t = (wb + 255) / 256 * 256, o_r = o_t + (tb + 255) / 256 * 256, o_s = o_r + (rb + 255) / 256 * 256;
    int rc;
    if ((rc = slam_stage_reserve(ctx, 0, o_s + sb)) || (rc = slam_stage_reserve(ctx, 1, desc_bytes + 16)) ||
        (rc = slam_stage_reserve(ctx, 2, out_bytes + 16)))
        return rc;
    uint8_t* t0 = (uint8_t*)ctx->d_stage[0];
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(t0, wj.data(), wb, hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(t0 + o_t, tj.data(), tb, hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_stage[1], desc_blob, desc_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = slam_hip_warp_batch_d(ctx, (const uint8_t*)ctx->d_stage[1], desc_bytes, (const slam_warp_job*)t0, njobs,
                               (uint8_t*)ctx->d_stage[2], out_bytes, (int32_t*)(t0 + o_s), max_pix);
    if (rc != SLAM_HIP_OK) return rc;
    slam_size_walk_lds(ctx, pyr, tj.data(), njobs);
    rc = slam_hip_track_batch_d(ctx, pyr, (const slam_track_job*)(t0 + o_t), njobs, (const uint8_t*)ctx->d_stage[2],
                                out_bytes ? out_bytes : 1, (slam_track_result*)(t0 + o_r));
    if (rc != SLAM_HIP_OK) return rc;
    std::vector<int32_t> st((size_t)njobs);
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(out, t0 + o_r, rb, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(st.data(), t0 + o_s, sb, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < njobs; i++)
        if (st[i] != SLAM_HIP_OK) out[i].status = st[i];
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_project(slam_ctx* ctx, const uint8_t* descr, int dw, int dh, size_t pitch,
                                const float* P, const float* B, const float* p3d,
                                uint8_t* proj, uint8_t* mask, size_t cap, int32_t* loc, int32_t* valid) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (!descr || dw < 1 || dh < 1 || dw > 32767 || dh > 32767 || pitch < (size_t)dw || !proj || !mask || !loc || !valid)
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "project: bad arguments");
    slam_warp_job job;
    memset(&job, 0, sizeof(job));
    int rc = slam_hip_project_setup(P, B, p3d, dw, dh, loc, job.m, valid);
    if (rc != SLAM_HIP_OK) return slam_set_error(ctx, rc, "project: bad matrices");
    if (!*valid) return SLAM_HIP_OK;
    const size_t npix = (size_t)loc[2] * loc[3];
    if (npix > cap) return slam_set_error(ctx, SLAM_HIP_ECAP, "project: %d x %d outputs need %zu bytes each", loc[2], loc[3], npix);
    job.dw = dw; job.dh = dh; job.ow = loc[2]; job.oh = loc[3];
    job.desc_off = 0; job.proj_off = 0; job.mask_off = (int64_t)npix;
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    slam_stream_drain drain(ctx);        // error returns below wait for the copies queued so far
    const size_t db = (size_t)dw * dh;
    const size_t o_desc = 256, o_out = o_desc + (db + 255) / 256 * 256;
    if ((rc = slam_stage_reserve(ctx, 0, o_out + 2 * npix + 16))) return rc;
    uint8_t* base = (uint8_t*)ctx->d_stage[0];
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(base, &job, sizeof(job), hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipMemcpy2DAsync(base + o_desc, (size_t)dw, descr, pitch, (size_t)dw, (size_t)dh, hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));      // job is a stack object
    rc = slam_hip_warp_batch_d(ctx, base + o_desc, db, (const slam_warp_job*)base, 1, base + o_out, 2 * npix,
                               (int32_t*)(base + 128), (int)npix);
    if (rc != SLAM_HIP_OK) return rc;
    int32_t st = 0;
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(proj, base + o_out, npix, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(mask, base + o_out + npix, npix, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(&st, base + 128, 4, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (st != SLAM_HIP_OK) return slam_set_error(ctx, st, "project: job rejected by the device");
    return SLAM_HIP_OK;
}

// ---- area mask of LineFeature::track (lineFeature.cpp:86-96) ----------------------------------
namespace {

// One workgroup per (job, slice of the mask).  cv::circle with thickness -1 walks the midpoint
// circle (dx, dy) keeping err = dx^2 + dy^2 - r^2 <= 0 and fills the rows cy +- dy over cx +- dx and
// cy +- dx over cx +- dy: the filled set is exactly { (x, y) : (x-cx)^2 + (y-cy)^2 <= r^2 }, so a
// pixel is 255 when it lies in one of the job's discs and 0 otherwise.
__global__ __launch_bounds__(WARP_WG) void disc_mask_kernel(const int32_t* __restrict__ circles, int ncircles,
                                                            const slam_disc_job* __restrict__ jobs, int njobs,
                                                            uint8_t* __restrict__ blob, unsigned long long blob_bytes,
                                                            int32_t* __restrict__ status) {
    constexpr int CHUNK = 256;
    __shared__ int cx[CHUNK], cy[CHUNK];
    __shared__ long long r2[CHUNK];
    __shared__ int bad;
    const int j = blockIdx.x;
    if (j >= njobs) return;
    const slam_disc_job job = jobs[j];
    const bool lead = blockIdx.y == 0 && threadIdx.x == 0;
    const unsigned long long npix = (unsigned long long)(unsigned)max(job.w, 0) * (unsigned)max(job.h, 0);
    const bool ok = job.w >= 1 && job.h >= 1 && job.w <= (1 << 14) && job.h <= (1 << 14) && job.mask_off >= 0 &&
                    (unsigned long long)job.mask_off + npix <= blob_bytes && job.circle_begin >= 0 && job.circle_count >= 0 &&
                    (long long)job.circle_begin + job.circle_count <= (long long)ncircles;
    if (!ok) { if (lead && status) status[j] = SLAM_HIP_EINVAL; return; }
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    for (int c = threadIdx.x; c < job.circle_count; c += WARP_WG)
        if (circles[3 * (job.circle_begin + c) + 2] < 0) bad = 1;          // cv::circle asserts radius >= 0
    __syncthreads();
    if (bad) { if (lead && status) status[j] = SLAM_HIP_EINVAL; return; }
    if (lead && status) status[j] = SLAM_HIP_OK;
    uint8_t* M = blob + job.mask_off;
    const int total = (int)npix;
    const int span = gridDim.y * WARP_WG;
    const int iters = (total + span - 1) / span;                            // the same for every lane: barriers inside
    const bool small = total + span < (1 << 20);                            // then (i + 0.5) / w through a reciprocal is exact
    const float rw = __builtin_amdgcn_rcpf((float)job.w);
    for (int it = 0; it < iters; it++) {
        const int i = it * span + blockIdx.y * WARP_WG + threadIdx.x;
        const int y = small ? (int)(((float)i + 0.5f) * rw) : i / job.w, x = i - y * job.w;
        bool in = false;
        for (int c0 = 0; c0 < job.circle_count; c0 += CHUNK) {
            const int nc = min(CHUNK, job.circle_count - c0);
            __syncthreads();
            if ((int)threadIdx.x < nc) {
                const int32_t* c = circles + 3 * (size_t)(job.circle_begin + c0 + threadIdx.x);
                cx[threadIdx.x] = c[0]; cy[threadIdx.x] = c[1]; r2[threadIdx.x] = (long long)c[2] * c[2];
            }
            __syncthreads();
            for (int c = 0; c < nc && !in; c++) {
                const long long ddx = (long long)x - cx[c], ddy = (long long)y - cy[c];
                in = ddx * ddx + ddy * ddy <= r2[c];
            }
        }
        if (i < total) M[i] = in ? 255 : 0;
    }
}

}  // namespace

extern "C" int slam_hip_disc_masks_d(slam_ctx* ctx, const int32_t* d_circles, int ncircles, const slam_disc_job* d_jobs,
                                     int njobs, uint8_t* d_blob, size_t blob_bytes, int32_t* d_status, int max_pixels) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (njobs < 0 || ncircles < 0 || (njobs > 0 && (!d_jobs || !d_blob)) || (ncircles > 0 && !d_circles))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "disc_masks: null table or negative count");
    if (njobs == 0) return SLAM_HIP_OK;
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    int slices = max_pixels > 0 ? (max_pixels + WARP_WG * 16 - 1) / (WARP_WG * 16) : 1;
    slices = std::min(std::max(slices, 1), 64);
    hipLaunchKernelGGL(disc_mask_kernel, dim3(njobs, slices), dim3(WARP_WG), 0, ctx->stream, d_circles, ncircles, d_jobs,
                       njobs, d_blob, (unsigned long long)blob_bytes, d_status);
    SLAM_HIP_CHECK(ctx, hipGetLastError());
    return SLAM_HIP_OK;
}

// SLAM::matchLines' image work for one frame batch, host tables (matching.cpp:279-292 around
// lineFeature.cpp:86-165): the area masks are rasterised on the device from the projected cone bins and
// searched there; only the descriptors, the circle table and the results cross the bus.
extern "C" int slam_hip_match_lines_batch(slam_ctx* ctx, const slam_pyr* pyr, const slam_line_job* jobs, int njobs,
                                          const int32_t* circles, int ncircles, const int32_t* circle_begin,
                                          const uint8_t* desc_blob, size_t desc_bytes, slam_track_result* out) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (njobs < 0 || ncircles < 0 || (njobs > 0 && (!jobs || !circle_begin || !desc_blob || !out)) || (ncircles > 0 && !circles))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "match_lines_batch: null table or negative count");
    if (!pyr) return slam_set_error(ctx, SLAM_HIP_EINVAL, "match_lines_batch: no pyramid");
    if (njobs == 0) return SLAM_HIP_OK;
    std::vector<slam_line_job> lj(jobs, jobs + njobs);
    std::vector<slam_disc_job> dj((size_t)njobs);
    size_t off = (desc_bytes + 255) / 256 * 256;
    int max_pix = 0;
    for (int i = 0; i < njobs; i++) {
        if (circle_begin[i] < 0 || circle_begin[i + 1] < circle_begin[i] || circle_begin[i + 1] > ncircles)
            return slam_set_error(ctx, SLAM_HIP_EINVAL, "match_lines_batch: job %d: circle range", i);
        const bool live = lj[i].aw >= 1 && lj[i].ah >= 1 && lj[i].aw <= (1 << 12) && lj[i].ah <= (1 << 12);
        const size_t n = live ? (size_t)lj[i].aw * lj[i].ah : 0;
        dj[i].mask_off = (int64_t)off; dj[i].w = lj[i].aw; dj[i].h = lj[i].ah;
        dj[i].circle_begin = circle_begin[i]; dj[i].circle_count = circle_begin[i + 1] - circle_begin[i];
        lj[i].amask_off = (int64_t)off;
        off += n;
        max_pix = std::max(max_pix, (int)n);
    }
    for (int c = 0; c < ncircles; c++)
        if (circles[3 * c + 2] < 0) return slam_set_error(ctx, SLAM_HIP_EINVAL, "match_lines_batch: circle %d has a negative radius", c);
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    slam_stream_drain drain(ctx);        // error returns below wait for the copies queued so far
    const size_t lb = sizeof(slam_line_job) * (size_t)njobs, db = sizeof(slam_disc_job) * (size_t)njobs;
    const size_t rb = sizeof(slam_track_result) * (size_t)njobs, cb = 12 * (size_t)ncircles;
    const size_t o_d = (lb + 255) / 256 * 256, o_r = o_d + (db + 255) / 256 * 256, o_c = o_r + (rb + 255) / 256 * 256;
    int rc;
    if ((rc = slam_stage_reserve(ctx, 0, o_c + cb + 16)) || (rc = slam_stage_reserve(ctx, 1, off + 16))) return rc;
    uint8_t* t0 = (uint8_t*)ctx->d_stage[0];
    uint8_t* blob = (uint8_t*)ctx->d_stage[1];
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(t0, lj.data(), lb, hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(t0 + o_d, dj.data(), db, hipMemcpyHostToDevice, ctx->stream));
    if (cb) SLAM_HIP_CHECK(ctx, hipMemcpyAsync(t0 + o_c, circles, cb, hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(blob, desc_blob, desc_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = slam_hip_disc_masks_d(ctx, (const int32_t*)(t0 + o_c), ncircles, (const slam_disc_job*)(t0 + o_d), njobs, blob, off,
                               nullptr, max_pix);
    if (rc != SLAM_HIP_OK) return rc;
    rc = slam_hip_line_track_batch_d(ctx, pyr, (const slam_line_job*)t0, njobs, blob, off ? off : 1, (slam_track_result*)(t0 + o_r));
    if (rc != SLAM_HIP_OK) return rc;
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(out, t0 + o_r, rb, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_HIP_OK;
}

// ---- uncertainty windows: gate ellipse of a 2x2 covariance -> rectangle + area mask --------------
extern "C" int slam_hip_ellipse_setup(float px, float py, const float* cov, float gate, float res, slam_ellipse_job* out) {
    if (!cov || !out) return SLAM_HIP_EINVAL;
    std::memset(out, 0, sizeof(*out));
    const double sxx = cov[0], sxy = cov[1], syy = cov[2];
    if (!std::isfinite(px) || !std::isfinite(py) || !std::isfinite(sxx) || !std::isfinite(sxy) || !std::isfinite(syy) ||
        !std::isfinite(gate) || !std::isfinite(res))
        return SLAM_HIP_OK;
    const double det = sxx * syy - sxy * sxy;
    if (!(gate > 0.f) || !(res > 0.f) || !(sxx > 0.0) || !(syy > 0.0) || !(det > 0.0)) return SLAM_HIP_OK;
    const double cx = (double)px * (double)res, cy = (double)py * (double)res;
    const double hx = std::sqrt((double)gate * sxx) * (double)res, hy = std::sqrt((double)gate * syy) * (double)res;
    const double lim = 1e9;
    if (!(std::fabs(cx) + hx < lim) || !(std::fabs(cy) + hy < lim)) return SLAM_HIP_OK;
    const int x0 = (int)std::floor(cx - hx), y0 = (int)std::floor(cy - hy);
    const int x1 = (int)std::ceil(cx + hx) + 1, y1 = (int)std::ceil(cy + hy) + 1;
    if (x1 - x0 > SLAM_HIP_ELLIPSE_MAX || y1 - y0 > SLAM_HIP_ELLIPSE_MAX) return SLAM_HIP_OK;   // lineFeature.cpp:81-84
    const double s = (double)gate * det * (double)res * (double)res;
    out->x = x0; out->y = y0; out->w = x1 - x0; out->h = y1 - y0;
    out->ux = (float)(cx - (double)x0); out->uy = (float)(cy - (double)y0);
    out->q11 = (float)(syy / s); out->q12 = (float)(-sxy / s); out->q22 = (float)(sxx / s);
    out->valid = 1;
    return SLAM_HIP_OK;
}

namespace {

__global__ __launch_bounds__(WARP_WG) void ellipse_mask_kernel(const slam_ellipse_job* __restrict__ jobs, int njobs,
                                                               uint8_t* __restrict__ blob, unsigned long long blob_bytes,
                                                               int32_t* __restrict__ status) {
    const int j = blockIdx.x;
    if (j >= njobs) return;
    const slam_ellipse_job job = jobs[j];
    const bool lead = blockIdx.y == 0 && threadIdx.x == 0;
    if (!job.valid) { if (lead && status) status[j] = SLAM_HIP_OK; return; }
    const bool ok = job.w >= 1 && job.h >= 1 && job.w <= SLAM_HIP_ELLIPSE_MAX && job.h <= SLAM_HIP_ELLIPSE_MAX && job.mask_off >= 0 &&
                    (unsigned long long)job.mask_off + (unsigned long long)(job.w * job.h) <= blob_bytes;
    if (lead && status) status[j] = ok ? SLAM_HIP_OK : SLAM_HIP_EINVAL;
    if (!ok) return;
    uint8_t* M = blob + job.mask_off;
    const int total = job.w * job.h;
    const float q12x2 = __fadd_rn(job.q12, job.q12);
    // the pixel nearest the centre always belongs to the window, however small the ellipse
    const int nx = (int)floorf(__fadd_rn(job.ux, 0.5f)), ny = (int)floorf(__fadd_rn(job.uy, 0.5f));
    const float rw = __builtin_amdgcn_rcpf((float)job.w);     // total <= 300 * 300: the reciprocal quotient is exact
    for (int i = blockIdx.y * WARP_WG + threadIdx.x; i < total; i += gridDim.y * WARP_WG) {
        const int y = (int)(((float)i + 0.5f) * rw), x = i - y * job.w;
        const float dx = __fsub_rn((float)x, job.ux), dy = __fsub_rn((float)y, job.uy);
        const float a = __fmul_rn(__fmul_rn(job.q11, dx), dx);
        const float b = __fmul_rn(__fmul_rn(q12x2, dx), dy);
        const float c = __fmul_rn(__fmul_rn(job.q22, dy), dy);
        M[i] = (__fadd_rn(__fadd_rn(a, b), c) <= 1.f || (x == nx && y == ny)) ? 255 : 0;
    }
}

bool bytes_are_01(const uint8_t* m, size_t n) {
    uint8_t acc = 0;
    for (size_t i = 0; i < n; i++) acc |= m[i];
    return (acc & 0xFE) == 0;
}

}  // namespace

extern "C" int slam_hip_ellipse_masks_d(slam_ctx* ctx, const slam_ellipse_job* d_jobs, int njobs, uint8_t* d_blob,
                                        size_t blob_bytes, int32_t* d_status, int max_pixels) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (njobs < 0 || (njobs > 0 && (!d_jobs || !d_blob)))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "ellipse_masks: null table or negative count");
    if (njobs == 0) return SLAM_HIP_OK;
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    int slices = max_pixels > 0 ? (max_pixels + WARP_WG * 8 - 1) / (WARP_WG * 8) : 4;
    slices = std::min(std::max(slices, 1), 64);
    hipLaunchKernelGGL(ellipse_mask_kernel, dim3(njobs, slices), dim3(WARP_WG), 0, ctx->stream, d_jobs, njobs, d_blob,
                       (unsigned long long)blob_bytes, d_status);
    SLAM_HIP_CHECK(ctx, hipGetLastError());
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_track_ellipse_batch(slam_ctx* ctx, const slam_pyr* pyr, const slam_track_job* jobs,
                                            const slam_ellipse_job* windows, int njobs, const uint8_t* blob,
                                            size_t blob_bytes, slam_track_result* out) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (njobs < 0 || (njobs > 0 && (!jobs || !windows || !blob || !out)))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "track_ellipse_batch: null table or negative count");
    if (!pyr) return slam_set_error(ctx, SLAM_HIP_EINVAL, "track_ellipse_batch: no pyramid");
    if (njobs == 0) return SLAM_HIP_OK;
    std::vector<slam_track_job> tj(jobs, jobs + njobs);
    std::vector<slam_ellipse_job> ej(windows, windows + njobs);
    size_t off = (blob_bytes + 255) / 256 * 256;
    int max_pix = 0;
    for (int i = 0; i < njobs; i++) {
        slam_track_job& t = tj[i];
        if (t.projw >= 1 && t.projh >= 1 && t.mask_off >= 0 &&
            (unsigned long long)t.mask_off + (unsigned long long)t.projw * t.projh <= blob_bytes &&
            !bytes_are_01(blob + t.mask_off, (size_t)t.projw * t.projh))
            return slam_set_error(ctx, SLAM_HIP_EINVAL, "track_ellipse_batch: job %d: mask values must be 0 or 1", i);
        slam_ellipse_job& e = ej[i];
        const bool live = e.valid && e.w >= 1 && e.h >= 1 && e.w <= SLAM_HIP_ELLIPSE_MAX && e.h <= SLAM_HIP_ELLIPSE_MAX;
        if (!live) {                                   // no window: reported like an empty projection
            e.valid = 0; t.projw = 0; t.projh = 0; t.flags = 0;
            continue;
        }
        e.mask_off = (int64_t)off;
        t.flags = SLAM_HIP_TRACK_COARSE_AREA; t.amask_off = (int64_t)off;
        t.ax = e.x; t.ay = e.y; t.aw = e.w; t.ah = e.h;
        off += (size_t)e.w * e.h;
        max_pix = std::max(max_pix, e.w * e.h);
    }
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    slam_stream_drain drain(ctx);        // error returns below wait for the copies queued so far
    const size_t tb = sizeof(slam_track_job) * (size_t)njobs, eb = sizeof(slam_ellipse_job) * (size_t)njobs;
    const size_t rb = sizeof(slam_track_result) * (size_t)njobs;
    const size_t o_e = (tb + 255) / 256 * 256, o_r = o_e + (eb + 255) / 256 * 256;
    int rc;
    if ((rc = slam_stage_reserve(ctx, 0, o_r + rb + 16)) || (rc = slam_stage_reserve(ctx, 1, off + 16))) return rc;
    uint8_t* t0 = (uint8_t*)ctx->d_stage[0];
    uint8_t* dblob = (uint8_t*)ctx->d_stage[1];
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(t0, tj.data(), tb, hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(t0 + o_e, ej.data(), eb, hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(dblob, blob, blob_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = slam_hip_ellipse_masks_d(ctx, (const slam_ellipse_job*)(t0 + o_e), njobs, dblob, off, nullptr, max_pix);
    if (rc != SLAM_HIP_OK) return rc;
    slam_size_walk_lds(ctx, pyr, tj.data(), njobs);
    rc = slam_hip_track_batch_d(ctx, pyr, (const slam_track_job*)t0, njobs, dblob, off, (slam_track_result*)(t0 + o_r));
    if (rc != SLAM_HIP_OK) return rc;
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(out, t0 + o_r, rb, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_HIP_OK;
}
