// Batched image pyramid: every level is a bilinear resize of the FULL-resolution frame
// (imagePyramid.hpp:27), levels with sub == 1 alias the input (imagePyramid.hpp:24-25).
#include <cmath>

#include "slam_internal.h"

namespace {

// One thread produces 4 horizontally adjacent output pixels of one frame and stores them as one
// dword.  Tap tables (source offset + two 11-bit weights per output index) are read-only and
// shared by all frames.  Arithmetic is cv::resize's 8-bit INTER_LINEAR path:
//   H(r)  = S[r][sx]*a0 + S[r][sx+1]*a1                    (int32)
//   out   = (((b0*(H(r0)>>4))>>16) + ((b1*(H(r1)>>4))>>16) + 2) >> 2
__global__ __launch_bounds__(256) void pyr_resize_kernel(
    const uint8_t* __restrict__ src, int sw, int sh, long long spitch, long long sstride,
    uint8_t* __restrict__ dst, int dw, int dh, long long dpitch, long long dstride,
    const int32_t* __restrict__ xofs, const uint32_t* __restrict__ xcoef,
    const int32_t* __restrict__ yofs, const uint32_t* __restrict__ ycoef) {
    const int gx = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;   // first of 4 output columns
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int frame = blockIdx.z;
    if (gx >= dw || dy >= dh) return;
    const uint8_t* S = src + (long long)frame * sstride;
    const int sy = yofs[dy];
    const uint32_t yc = ycoef[dy];
    const int b0 = (int)(int16_t)(yc & 0xffff), b1 = (int)(int16_t)(yc >> 16);
    const int r0 = min(max(sy, 0), sh - 1), r1 = min(max(sy + 1, 0), sh - 1);
    const uint8_t* S0 = S + (long long)r0 * spitch;
    const uint8_t* S1 = S + (long long)r1 * spitch;
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int dx = min(gx + i, dw - 1);
        const int sx = xofs[dx];
        const uint32_t xc = xcoef[dx];
        const int a0 = (int)(int16_t)(xc & 0xffff), a1 = (int)(int16_t)(xc >> 16);
        const int sx1 = min(sx + 1, sw - 1);
        const int h0 = (int)S0[sx] * a0 + (int)S0[sx1] * a1;
        const int h1 = (int)S1[sx] * a0 + (int)S1[sx1] * a1;
        const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        packed |= (uint32_t)(v & 0xff) << (8 * i);
    }
    // rows are padded to a multiple of 4 bytes, so the last dword of a row may be stored whole
    uint8_t* D = dst + (long long)frame * dstride + (long long)dy * dpitch + gx;
    *reinterpret_cast<uint32_t*>(D) = packed;
}

}  // namespace

int slam_launch_pyramid(slam_ctx* ctx, slam_pyr* pyr) {
    for (int l = 0; l < pyr->nlev; l++) {
        if (!pyr->d_level[l]) continue;   // aliased level
        dim3 grid((pyr->lw[l] + 255) / 256, (pyr->lh[l] + 3) / 4, pyr->nframes);
        hipLaunchKernelGGL(pyr_resize_kernel, grid, dim3(256), 0, ctx->stream,
                           pyr->d_frames, pyr->w, pyr->h, (long long)pyr->pitch0, (long long)pyr->stride0,
                           pyr->d_level[l], pyr->lw[l], pyr->lh[l], (long long)pyr->lpitch[l],
                           (long long)pyr->lstride[l], pyr->d_xofs[l], pyr->d_xcoef[l],
                           pyr->d_yofs[l], pyr->d_ycoef[l]);
        SLAM_HIP_CHECK(ctx, hipGetLastError());
    }
    return SLAM_HIP_OK;
}

static void pyr_refresh_dev(slam_pyr* p) {
    p->dev.nlev = p->nlev;
    p->dev.nframes = p->nframes;
    for (int l = 0; l < p->nlev; l++) {
        slam_level_dev& d = p->dev.lev[l];
        d.w = p->lw[l]; d.h = p->lh[l]; d.sub = p->subs[l];
        if (p->d_level[l]) {
            d.data = p->d_level[l]; d.pitch = (long long)p->lpitch[l]; d.frame_stride = (long long)p->lstride[l];
        } else {
            d.data = p->d_frames; d.pitch = (long long)p->pitch0; d.frame_stride = (long long)p->stride0;
        }
    }
}

extern "C" int slam_hip_pyramid_create(slam_ctx* ctx, int w, int h, int nlev, const float* subs,
                                       int nframes, slam_pyr** out) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (!out || !subs || w < 1 || h < 1 || nlev < 1 || nlev > SLAM_HIP_MAX_LEVELS || nframes < 1)
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "pyramid_create: bad arguments (w=%d h=%d nlev=%d nframes=%d)", w, h, nlev, nframes);
    *out = nullptr;
    for (int i = 0; i < nlev; i++) {
        if (!(subs[i] > 0.f)) return slam_set_error(ctx, SLAM_HIP_EINVAL, "pyramid_create: subsample %g", (double)subs[i]);
        // imagePyramid.hpp:20-21: assert(subsamples[i-1] < subsamples[i])
        if (i && !(subs[i - 1] < subs[i])) return slam_set_error(ctx, SLAM_HIP_EINVAL, "pyramid_create: subsamples must increase");
    }
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    slam_pyr* p = new (std::nothrow) slam_pyr();
    if (!p) return slam_set_error(ctx, SLAM_HIP_ENOMEM, "pyramid_create: out of host memory");
    p->device = ctx->device; p->w = w; p->h = h; p->nlev = nlev; p->nframes = nframes;
    int rc = SLAM_HIP_OK;
    for (int l = 0; l < nlev && rc == SLAM_HIP_OK; l++) {
        const float sub = subs[l];
        p->subs[l] = sub;
        if (sub == 1.f) { p->lw[l] = w; p->lh[l] = h; continue; }
        // imagePyramid.hpp:27: Size(round(w/sub), round(h/sub)); int/float -> float, C round()
        const float fw = (float)w / sub, fh = (float)h / sub;
        const int dw = (int)std::round((double)fw), dh = (int)std::round((double)fh);
        if (dw < 1 || dh < 1) { rc = slam_set_error(ctx, SLAM_HIP_EINVAL, "pyramid_create: level %d is empty", l); break; }
        p->lw[l] = dw; p->lh[l] = dh;
        p->lpitch[l] = ((size_t)dw + 63) / 64 * 64;
        p->lstride[l] = p->lpitch[l] * (size_t)dh;
        std::vector<int32_t> xo, yo;
        std::vector<uint32_t> xc, yc;
        slam_resize_tables(w, dw, xo, xc, true);
        slam_resize_tables(h, dh, yo, yc, false);
        void* ptr = nullptr;
        auto alloc_copy = [&](const void* src, size_t bytes, void** dptr) -> int {
            hipError_t e = hipMalloc(dptr, bytes);
            if (e != hipSuccess) { (void)hipGetLastError(); return slam_set_error(ctx, SLAM_HIP_ENOMEM, "pyramid_create: hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
            if (src) {
                e = hipMemcpy(*dptr, src, bytes, hipMemcpyHostToDevice);
                if (e != hipSuccess) return slam_set_error(ctx, SLAM_HIP_EHIP, "pyramid_create: hipMemcpy: %s", hipGetErrorString(e));
            }
            return SLAM_HIP_OK;
        };
        if ((rc = alloc_copy(nullptr, p->lstride[l] * (size_t)nframes, &ptr)) != SLAM_HIP_OK) break;
        p->d_level[l] = (uint8_t*)ptr;
        if ((rc = alloc_copy(xo.data(), xo.size() * 4, &ptr)) != SLAM_HIP_OK) break;
        p->d_xofs[l] = (int32_t*)ptr;
        if ((rc = alloc_copy(xc.data(), xc.size() * 4, &ptr)) != SLAM_HIP_OK) break;
        p->d_xcoef[l] = (uint32_t*)ptr;
        if ((rc = alloc_copy(yo.data(), yo.size() * 4, &ptr)) != SLAM_HIP_OK) break;
        p->d_yofs[l] = (int32_t*)ptr;
        if ((rc = alloc_copy(yc.data(), yc.size() * 4, &ptr)) != SLAM_HIP_OK) break;
        p->d_ycoef[l] = (uint32_t*)ptr;
    }
    if (rc != SLAM_HIP_OK) { slam_hip_pyramid_destroy(ctx, p); return rc; }
    *out = p;
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_pyramid_destroy(slam_ctx* ctx, slam_pyr* p) {
    if (!p) return SLAM_HIP_OK;
    if (ctx) { (void)hipSetDevice(ctx->device); (void)hipStreamSynchronize(ctx->stream); }
    for (int l = 0; l < SLAM_HIP_MAX_LEVELS; l++) {
        if (p->d_level[l]) (void)hipFree(p->d_level[l]);
        if (p->d_xofs[l]) (void)hipFree(p->d_xofs[l]);
        if (p->d_xcoef[l]) (void)hipFree(p->d_xcoef[l]);
        if (p->d_yofs[l]) (void)hipFree(p->d_yofs[l]);
        if (p->d_ycoef[l]) (void)hipFree(p->d_ycoef[l]);
    }
    if (p->d_upload) (void)hipFree(p->d_upload);
    delete p;
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_pyramid_build_d(slam_ctx* ctx, slam_pyr* p, const uint8_t* d_frames,
                                        size_t pitch, size_t frame_stride) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (!p || !d_frames || pitch < (size_t)p->w || (p->nframes > 1 && frame_stride < pitch * (size_t)p->h))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "pyramid_build: bad frame layout (pitch=%zu stride=%zu)", pitch, frame_stride);
    if (p->device != ctx->device) return slam_set_error(ctx, SLAM_HIP_EINVAL, "pyramid_build: pyramid belongs to device %d", p->device);
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    p->d_frames = d_frames; p->pitch0 = pitch; p->stride0 = frame_stride;
    pyr_refresh_dev(p);
    int rc = slam_launch_pyramid(ctx, p);
    if (rc != SLAM_HIP_OK) return rc;
    p->built = true;
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_pyramid_build(slam_ctx* ctx, slam_pyr* p, const uint8_t* h_frames,
                                      size_t pitch, size_t frame_stride) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (!p || !h_frames || pitch < (size_t)p->w || (p->nframes > 1 && frame_stride < pitch * (size_t)p->h))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "pyramid_build: bad frame layout (pitch=%zu stride=%zu)", pitch, frame_stride);
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // device copy: rows packed to a 64-byte multiple
    const size_t dpitch = ((size_t)p->w + 63) / 64 * 64;
    const size_t dstride = dpitch * (size_t)p->h;
    const size_t need = dstride * (size_t)p->nframes;
    if (need > p->upload_bytes) {
        SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (p->d_upload) { SLAM_HIP_CHECK(ctx, hipFree(p->d_upload)); p->d_upload = nullptr; p->upload_bytes = 0; }
        hipError_t e = hipMalloc((void**)&p->d_upload, need);
        if (e != hipSuccess) { (void)hipGetLastError(); return slam_set_error(ctx, SLAM_HIP_ENOMEM, "pyramid_build: hipMalloc(%zu): %s", need, hipGetErrorString(e)); }
        p->upload_bytes = need;
    }
    for (int f = 0; f < p->nframes; f++)
        SLAM_HIP_CHECK(ctx, hipMemcpy2DAsync(p->d_upload + (size_t)f * dstride, dpitch,
                                             h_frames + (size_t)f * frame_stride, pitch,
                                             (size_t)p->w, (size_t)p->h, hipMemcpyHostToDevice, ctx->stream));
    int rc = slam_hip_pyramid_build_d(ctx, p, p->d_upload, dpitch, dstride);
    if (rc != SLAM_HIP_OK) return rc;
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_pyramid_level(const slam_pyr* p, int level, slam_level_desc* out) {
    if (!p || !out || level < 0 || level >= p->nlev) return SLAM_HIP_EINVAL;
    out->w = p->lw[level]; out->h = p->lh[level]; out->sub = p->subs[level];
    if (p->d_level[level]) {
        out->d_data = p->d_level[level]; out->pitch = p->lpitch[level]; out->frame_stride = p->lstride[level];
    } else {
        out->d_data = const_cast<uint8_t*>(p->d_frames); out->pitch = p->pitch0; out->frame_stride = p->stride0;
    }
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_pyramid_download(slam_ctx* ctx, const slam_pyr* p, int level, int frame,
                                         uint8_t* h_dst, size_t dst_pitch) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (!p || !h_dst || level < 0 || level >= p->nlev || frame < 0 || frame >= p->nframes || !p->built)
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "pyramid_download: bad arguments or pyramid not built");
    slam_level_desc d;
    slam_hip_pyramid_level(p, level, &d);
    if (dst_pitch < (size_t)d.w) return slam_set_error(ctx, SLAM_HIP_EINVAL, "pyramid_download: dst_pitch < width");
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    SLAM_HIP_CHECK(ctx, hipMemcpy2DAsync(h_dst, dst_pitch, d.d_data + (size_t)frame * d.frame_stride, d.pitch,
                                         (size_t)d.w, (size_t)d.h, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_HIP_OK;
}

// ---- debug: one-off resize of a host image (tests the table + kernel arithmetic on any size) ----
extern "C" int slam_hip_debug_resize(slam_ctx* ctx, const uint8_t* src, int sw, int sh, size_t spitch,
                                     uint8_t* dst, int dw, int dh, size_t dpitch) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (!src || !dst || sw < 1 || sh < 1 || dw < 1 || dh < 1 || spitch < (size_t)sw || dpitch < (size_t)dw)
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "debug_resize: bad arguments");
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    std::vector<int32_t> xo, yo;
    std::vector<uint32_t> xc, yc;
    slam_resize_tables(sw, dw, xo, xc, true);
    slam_resize_tables(sh, dh, yo, yc, false);
    const size_t dp = ((size_t)dw + 63) / 64 * 64;
    const size_t off_src = 0, off_dst = ((size_t)sw * sh + 63) / 64 * 64, off_t = off_dst + dp * dh;
    const size_t tbytes = ((size_t)dw + dh) * 8;
    const size_t off_tab = (off_t + 15) / 16 * 16;
    int rc = slam_stage_reserve(ctx, 0, off_tab + tbytes + 64);
    if (rc != SLAM_HIP_OK) return rc;
    uint8_t* base = (uint8_t*)ctx->d_stage[0];
    SLAM_HIP_CHECK(ctx, hipMemcpy2DAsync(base + off_src, sw, src, spitch, sw, sh, hipMemcpyHostToDevice, ctx->stream));
    int32_t* d_xo = (int32_t*)(base + off_tab);
    uint32_t* d_xc = (uint32_t*)(d_xo + dw);
    int32_t* d_yo = (int32_t*)(d_xc + dw);
    uint32_t* d_yc = (uint32_t*)(d_yo + dh);
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(d_xo, xo.data(), dw * 4, hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(d_xc, xc.data(), dw * 4, hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(d_yo, yo.data(), dh * 4, hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(d_yc, yc.data(), dh * 4, hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));   // the host vectors go out of scope below
    dim3 grid((dw + 255) / 256, (dh + 3) / 4, 1);
    hipLaunchKernelGGL(pyr_resize_kernel, grid, dim3(256), 0, ctx->stream,
                       base + off_src, sw, sh, (long long)sw, 0LL, base + off_dst, dw, dh, (long long)dp, 0LL,
                       d_xo, d_xc, d_yo, d_yc);
    SLAM_HIP_CHECK(ctx, hipGetLastError());
    SLAM_HIP_CHECK(ctx, hipMemcpy2DAsync(dst, dpitch, base + off_dst, dp, dw, dh, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_HIP_OK;
}
