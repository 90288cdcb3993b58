// New-feature path that follows FAST on a frame batch, all on the device:
//   sort_kernel     sorts each frame's keypoints by response (new_point.cpp:124; ties by y, x)
//   select_kernel   the greedy minimum-distance selection of SLAM::addNewLines (new_point.cpp:137-155)
//   descr_kernel    the descriptor crop of Feature::newDescriptor / the LineFeature constructor
//                   (feature.cpp:42-45, lineFeature.cpp:7-10)
#include <algorithm>
#include <cmath>

#include "slam_internal.h"

namespace {

constexpr int NP_WG = 256;

constexpr int NP_SUPER = 2048;      // keys one workgroup places

// Stable counting sort by response.  cv::FAST's output is row-major, i.e. already ordered by (y, x),
// and responses are integers in [0, 255]: a stable sort by descending response therefore yields the
// order (response desc, y asc, x asc).  A frame's list is cut into runs of NP_SUPER keys, one workgroup each
// (no scratch memory, no ordering between workgroups): the workgroup first counts, over the whole list, the
// keys of every response value and those of them that precede its run - that fixes where each value's keys
// of this run start - and then walks its run in groups of 256 keys: per group each wavefront ranks its keys
// among equal responses (one ballot per distinct value), publishes per-wavefront bucket counts, and the
// destination is
//   start of the value + keys of that value in earlier groups + in earlier wavefronts + rank.
__global__ __launch_bounds__(NP_WG) void sort_kernel(const slam_keypoint* __restrict__ in, const int* __restrict__ counts,
                                                     int cap, slam_keypoint* __restrict__ out) {
    __shared__ unsigned base[256];          // next free output slot of each response value
    __shared__ unsigned before[256];        // keys of each value ahead of this run
    __shared__ unsigned wcnt[4][256];       // keys per (wavefront, response) in the current group
    const int frame = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n = min(counts[frame], cap);
    const int run0 = blockIdx.y * NP_SUPER, run1 = min(n, run0 + NP_SUPER);
    if (run0 >= n) return;
    const slam_keypoint* src = in + (size_t)frame * cap;
    slam_keypoint* dst = out + (size_t)frame * cap;
    base[tid] = 0u; before[tid] = 0u;
    for (int k = 0; k < 4; k++) wcnt[k][tid] = 0u;
    __syncthreads();
    for (int i = tid; i < n; i += NP_WG) {
        const int bkt = min(max((int)src[i].response, 0), 255);
        atomicAdd(&base[bkt], 1u);
        if (i < run0) atomicAdd(&before[bkt], 1u);
    }
    __syncthreads();
    if (tid == 0) {                         // descending response: bucket 255 first
        unsigned run = 0;
        for (int b = 255; b >= 0; b--) { const unsigned c = base[b]; base[b] = run + before[b]; run += c; }
    }
    __syncthreads();
    for (int g0 = run0; g0 < run1; g0 += NP_WG) {
        const int i = g0 + tid;
        const bool valid = i < run1;
        slam_keypoint kp = {0.f, 0.f, 0.f};
        if (valid) kp = src[i];
        const int b = min(max((int)kp.response, 0), 255);
        unsigned rank = 0;
        unsigned long long todo = __ballot(valid);
        while (todo) {                      // wave-uniform loop over the distinct responses of these 64 keys
            const int leader = __ffsll((long long)todo) - 1;
            const int lb = __builtin_amdgcn_readlane(b, leader);
            const unsigned long long same = __ballot(valid && b == lb);
            if (valid && b == lb) rank = (unsigned)__popcll(same & ((1ull << lane) - 1ull));
            if (lane == leader) wcnt[wid][lb] = (unsigned)__popcll(same);
            todo &= ~same;
        }
        __syncthreads();
        if (valid) {
            unsigned d = base[b] + rank;
            for (int k = 0; k < wid; k++) d += wcnt[k][b];
            dst[d] = kp;
        }
        __syncthreads();
        {                                   // thread = bucket: advance the bases, clear the counts
            base[tid] += wcnt[0][tid] + wcnt[1][tid] + wcnt[2][tid] + wcnt[3][tid];
            wcnt[0][tid] = 0u; wcnt[1][tid] = 0u; wcnt[2][tid] = 0u; wcnt[3][tid] = 0u;
        }
        __syncthreads();
    }
}

// cv::norm(a - b) < minDist with the reference's types: float difference, double norm.
__device__ __forceinline__ bool closer_than(float ax, float ay, float bx, float by, float min_dist) {
    const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by);
    const double d2 = __dadd_rn(__dmul_rn((double)dx, (double)dx), __dmul_rn((double)dy, (double)dy));
    return __dsqrt_rn(d2) < (double)min_dist;
}

// One wavefront per frame walks the sorted keypoints in order (the selection is sequential by
// nature: a kept point blocks later ones); the distance tests against the occupied positions and
// the points kept so far are spread over the 64 lanes.
__global__ __launch_bounds__(64) void select_kernel(const slam_keypoint* __restrict__ kps, const int* __restrict__ counts,
                                                    int cap, const float* __restrict__ occ_xy, const int* __restrict__ occ_off,
                                                    int n_new, float min_dist, int rx, int ry, int w, int h,
                                                    int* __restrict__ sel, int* __restrict__ nsel_out) {
    extern __shared__ float kept[];         // xy of the points kept so far (n_new pairs)
    const int frame = blockIdx.x, lane = threadIdx.x;
    const int n = min(counts[frame], cap);
    const slam_keypoint* kp = kps + (size_t)frame * cap;
    const int o0 = occ_off ? occ_off[frame] : 0, o1 = occ_off ? occ_off[frame + 1] : 0;
    const float* occ = occ_xy + 2 * (size_t)o0;
    const int nocc = o1 - o0;
    int* s = sel + (size_t)frame * n_new;
    int nsel = 0;
    for (int i = 0; i < n && nsel < n_new; i++) {
        const float px = kp[i].x, py = kp[i].y;
        bool bad = (__fsub_rn(py, (float)ry) < 0.f) || (__fsub_rn(px, (float)rx) < 0.f) ||
                   (__fadd_rn(py, (float)ry) >= (float)h) || (__fadd_rn(px, (float)rx) >= (float)w);   // new_point.cpp:148-149
        for (int j = lane; j < nocc && !bad; j += 64) bad = closer_than(occ[2 * j], occ[2 * j + 1], px, py, min_dist);
        for (int j = lane; j < nsel && !bad; j += 64) bad = closer_than(kept[2 * j], kept[2 * j + 1], px, py, min_dist);
        if (__ballot(bad) != 0ull) continue;
        if (lane == 0) { kept[2 * nsel] = px; kept[2 * nsel + 1] = py; s[nsel] = i; }
        nsel++;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (lane == 0) nsel_out[frame] = nsel;
}

__device__ __forceinline__ int iround_f(float a) { return (int)round((double)a); }   // common.hpp:62-66

// One wavefront per descriptor: the crop rectangle, then a row-by-row byte copy.
__global__ __launch_bounds__(64) void descr_kernel(const uint8_t* __restrict__ frames, int w, int h, long long pitch,
                                                   long long frame_stride, int nframes, const slam_desc_job* __restrict__ jobs,
                                                   int njobs, uint8_t* __restrict__ out, unsigned long long out_bytes,
                                                   slam_desc_result* __restrict__ res) {
    const int j = blockIdx.x, lane = threadIdx.x;
    if (j >= njobs) return;
    const slam_desc_job job = jobs[j];
    slam_desc_result r = {0, 0, 0, 0, SLAM_HIP_OK};
    const bool ok = job.frame >= 0 && job.frame < nframes && job.rx >= 0 && job.ry >= 0 && job.rx <= 4096 && job.ry <= 4096 &&
                    fabsf(job.px) < 1e7f && fabsf(job.py) < 1e7f && job.out_off >= 0;
    if (!ok) { r.status = SLAM_HIP_EINVAL; if (lane == 0) res[j] = r; return; }
    // feature.cpp:42-45
    const int y0 = max(0, iround_f(__fsub_rn(job.py, (float)job.ry))), y1 = min(h, iround_f(__fadd_rn(__fadd_rn(job.py, (float)job.ry), 1.f)));
    const int x0 = max(0, iround_f(__fsub_rn(job.px, (float)job.rx))), x1 = min(w, iround_f(__fadd_rn(__fadd_rn(job.px, (float)job.rx), 1.f)));
    const int cw = max(x1 - x0, 0), ch = max(y1 - y0, 0);
    if (cw == 0 || ch == 0) { if (lane == 0) res[j] = r; return; }
    if ((unsigned long long)job.out_off + (unsigned long long)cw * ch > out_bytes) {
        r.status = SLAM_HIP_EINVAL; if (lane == 0) res[j] = r; return;
    }
    r.x0 = x0; r.y0 = y0; r.w = cw; r.h = ch;
    const uint8_t* src = frames + (long long)job.frame * frame_stride + (long long)y0 * pitch + x0;
    uint8_t* dst = out + job.out_off;
    const int total = cw * ch;
    const float rcw = __builtin_amdgcn_rcpf((float)cw);
    for (int i = lane; i < total; i += 64) {
        int yy = (int)(((float)i + 0.5f) * rcw);
        if (total > (1 << 20)) yy = i / cw;                 // beyond the float quotient's exact range
        const int xx = i - yy * cw;
        dst[i] = src[(long long)yy * pitch + xx];
    }
    if (lane == 0) res[j] = r;
}

}  // namespace

extern "C" int slam_hip_sort_keypoints_d(slam_ctx* ctx, const slam_keypoint* d_in, const int32_t* d_counts,
                                         int nframes, int cap_per_frame, slam_keypoint* d_out) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (!d_in || !d_counts || !d_out || d_in == d_out || nframes < 1 || cap_per_frame < 1)
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "sort_keypoints: bad arguments");
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(sort_kernel, dim3(nframes, (cap_per_frame + NP_SUPER - 1) / NP_SUPER), dim3(NP_WG), 0, ctx->stream, d_in,
                       d_counts, cap_per_frame, d_out);
    SLAM_HIP_CHECK(ctx, hipGetLastError());
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_select_new_points_d(slam_ctx* ctx, const slam_keypoint* d_sorted, const int32_t* d_counts,
                                            int nframes, int cap_per_frame, const float* d_occ_xy,
                                            const int32_t* d_occ_off, int n_new, float min_dist, int rx, int ry,
                                            int w, int h, int32_t* d_sel, int32_t* d_nsel) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (!d_sorted || !d_counts || !d_sel || !d_nsel || nframes < 1 || cap_per_frame < 1 || n_new < 1 || n_new > 4096 ||
        rx < 0 || ry < 0 || w < 1 || h < 1 || !(min_dist == min_dist) || (d_occ_off && !d_occ_xy))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "select_new_points: bad arguments");
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(select_kernel, dim3(nframes), dim3(64), (size_t)n_new * 8, ctx->stream, d_sorted, d_counts, cap_per_frame,
                       d_occ_xy, d_occ_off, n_new, min_dist, rx, ry, w, h, d_sel, d_nsel);
    SLAM_HIP_CHECK(ctx, hipGetLastError());
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_extract_descriptors_d(slam_ctx* ctx, const uint8_t* d_frames, int w, int h, size_t pitch,
                                              size_t frame_stride, int nframes, const slam_desc_job* d_jobs, int njobs,
                                              uint8_t* d_out, size_t out_bytes, slam_desc_result* d_res) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (njobs < 0 || !d_frames || w < 1 || h < 1 || pitch < (size_t)w || nframes < 1 ||
        (nframes > 1 && frame_stride < pitch * (size_t)h) || (njobs > 0 && (!d_jobs || !d_out || !d_res)))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "extract_descriptors: bad arguments");
    if (njobs == 0) return SLAM_HIP_OK;
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(descr_kernel, dim3(njobs), dim3(64), 0, ctx->stream, d_frames, w, h, (long long)pitch,
                       (long long)frame_stride, nframes, d_jobs, njobs, d_out, (unsigned long long)out_bytes, d_res);
    SLAM_HIP_CHECK(ctx, hipGetLastError());
    return SLAM_HIP_OK;
}

// Host-table form of the whole step for one frame: sorted keypoints in, kept indices and their
// descriptors out (SLAM::addNewLines, new_point.cpp:129-156, without the LineFeature bookkeeping).
extern "C" int slam_hip_add_new_points(slam_ctx* ctx, const uint8_t* im, int w, int h, size_t pitch,
                                       const slam_keypoint* kps, int nkps, int already_sorted,
                                       const float* occ_xy, int nocc, int n_new, float min_dist, int rx, int ry,
                                       int32_t* sel, int32_t* nsel, slam_keypoint* sorted_out,
                                       uint8_t* descr_out, size_t descr_bytes, slam_desc_result* descr_res) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (!im || w < 1 || h < 1 || pitch < (size_t)w || nkps < 0 || (nkps > 0 && !kps) || nocc < 0 || (nocc > 0 && !occ_xy) ||
        n_new < 1 || n_new > 4096 || rx < 0 || ry < 0 || !sel || !nsel)
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "add_new_points: bad arguments");
    *nsel = 0;
    if (nkps == 0) return SLAM_HIP_OK;
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    slam_stream_drain drain(ctx);        // error returns below wait for the copies queued so far
    const size_t kb = sizeof(slam_keypoint) * (size_t)nkps;
    const size_t off_sorted = (kb + 255) / 256 * 256;
    const size_t off_occ = off_sorted + (kb + 255) / 256 * 256;
    const size_t off_cnt = off_occ + ((size_t)nocc * 8 + 255) / 256 * 256 + 256;
    const size_t off_sel = off_cnt + 256;
    const size_t off_jobs = off_sel + ((size_t)n_new * 4 + 255) / 256 * 256;
    const size_t off_res = off_jobs + ((size_t)n_new * sizeof(slam_desc_job) + 255) / 256 * 256;
    const size_t off_desc = off_res + ((size_t)n_new * sizeof(slam_desc_result) + 255) / 256 * 256;
    const size_t one = (size_t)(2 * rx + 1) * (2 * ry + 1);
    const size_t desc_need = one * (size_t)n_new;
    const size_t dpitch = ((size_t)w + 63) / 64 * 64;
    const size_t off_im = off_desc + (desc_need + 255) / 256 * 256;
    int rc = slam_stage_reserve(ctx, 0, off_im + dpitch * (size_t)h);
    if (rc != SLAM_HIP_OK) return rc;
    uint8_t* base = (uint8_t*)ctx->d_stage[0];
    int32_t hdr[3] = {nkps, 0, nocc};            // counts[0]; occ_off[0..1]
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(base, kps, kb, hipMemcpyHostToDevice, ctx->stream));
    if (nocc) SLAM_HIP_CHECK(ctx, hipMemcpyAsync(base + off_occ, occ_xy, (size_t)nocc * 8, hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(base + off_cnt, hdr, sizeof(hdr), hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));      // hdr is a stack object
    const slam_keypoint* d_sorted = (const slam_keypoint*)base;
    if (!already_sorted) {
        rc = slam_hip_sort_keypoints_d(ctx, (const slam_keypoint*)base, (const int32_t*)(base + off_cnt), 1, nkps,
                                       (slam_keypoint*)(base + off_sorted));
        if (rc != SLAM_HIP_OK) return rc;
        d_sorted = (const slam_keypoint*)(base + off_sorted);
    }
    rc = slam_hip_select_new_points_d(ctx, d_sorted, (const int32_t*)(base + off_cnt), 1, nkps, (const float*)(base + off_occ),
                                      (const int32_t*)(base + off_cnt) + 1, n_new, min_dist, rx, ry, w, h,
                                      (int32_t*)(base + off_sel), (int32_t*)(base + off_cnt) + 8);
    if (rc != SLAM_HIP_OK) return rc;
    std::vector<slam_keypoint> hs((size_t)nkps);
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(hs.data(), d_sorted, kb, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(nsel, (int32_t*)(base + off_cnt) + 8, 4, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (*nsel > 0) SLAM_HIP_CHECK(ctx, hipMemcpy(sel, base + off_sel, (size_t)*nsel * 4, hipMemcpyDeviceToHost));
    if (sorted_out) std::copy(hs.begin(), hs.end(), sorted_out);
    if (descr_out && descr_res && *nsel > 0) {
        if (descr_bytes < one * (size_t)*nsel) return slam_set_error(ctx, SLAM_HIP_EINVAL, "add_new_points: descriptor buffer too small");
        std::vector<slam_desc_job> jobs((size_t)*nsel);
        for (int i = 0; i < *nsel; i++) {
            jobs[i].frame = 0; jobs[i].px = hs[sel[i]].x; jobs[i].py = hs[sel[i]].y; jobs[i].rx = rx; jobs[i].ry = ry;
            jobs[i].reserved = 0; jobs[i].out_off = (int64_t)(one * (size_t)i);
        }
        SLAM_HIP_CHECK(ctx, hipMemcpy2DAsync(base + off_im, dpitch, im, pitch, (size_t)w, (size_t)h, hipMemcpyHostToDevice, ctx->stream));
        SLAM_HIP_CHECK(ctx, hipMemcpyAsync(base + off_jobs, jobs.data(), jobs.size() * sizeof(slam_desc_job), hipMemcpyHostToDevice, ctx->stream));
        SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        rc = slam_hip_extract_descriptors_d(ctx, base + off_im, w, h, dpitch, 0, 1, (const slam_desc_job*)(base + off_jobs), *nsel,
                                            base + off_desc, desc_need, (slam_desc_result*)(base + off_res));
        if (rc != SLAM_HIP_OK) return rc;
        SLAM_HIP_CHECK(ctx, hipMemcpyAsync(descr_out, base + off_desc, one * (size_t)*nsel, hipMemcpyDeviceToHost, ctx->stream));
        SLAM_HIP_CHECK(ctx, hipMemcpyAsync(descr_res, base + off_res, sizeof(slam_desc_result) * (size_t)*nsel, hipMemcpyDeviceToHost, ctx->stream));
        SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return SLAM_HIP_OK;
}
