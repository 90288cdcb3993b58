// Batched image pyramid: every level is a bilinear resize of the FULL-resolution frame
// (imagePyramid.hpp:27), levels with sub == 1 alias the input (imagePyramid.hpp:24-25).
#include <algorithm>
#include <cmath>

#include "slam_internal.h"

namespace {

// One launch builds every resized level of every frame.  A workgroup owns one tile of one level of
// one frame: `rb` output rows x `xt` output columns.  It stages the 2*rb source rows the tile needs
// (clamped row indices, dword-coalesced, only the column span under the tile) and the tile's tap
// tables into LDS, then each thread produces 4 horizontally adjacent output pixels per item and
// stores them as one dword.  Arithmetic is cv::resize's 8-bit INTER_LINEAR path:
//   H(r)  = S[r][sx]*a0 + S[r][sx+1]*a1                    (int32)
//   out   = (((b0*(H(r0)>>4))>>16) + ((b1*(H(r1)>>4))>>16) + 2) >> 2
// Workgroup ids interleave 8 frames (id % 8 picks the frame within a group of 8): with the usual
// round-robin placement of workgroups over the 8 XCDs, all tiles of all levels of a frame then run
// on one XCD and the second level's reads of the frame hit that XCD's L2.
struct PyrLevelArgs {
    uint8_t* dst;
    long long dpitch, dstride;
    const int32_t* xofs; const uint32_t* xcoef;
    const int32_t* yofs; const uint32_t* ycoef;
    int dw, dh;
    int xt, ncolt;      // output columns per tile (multiple of 4), tiles per row
    int rb, nband;      // output rows per tile, tiles per column
    int span;           // bytes of one staged source row in LDS (multiple of 4)
    int unit0;          // first work unit of this level within a frame
    int box;            // 0, or s in {2,4,8}: every tap pair is (s*d + s/2 - 1, +1) with equal weights
};

struct PyrArgs {
    const uint8_t* src;
    long long spitch, sstride;
    int sw, sh, nframes, nlev, units;
    PyrLevelArgs lev[SLAM_HIP_MAX_LEVELS];
};

// Exact integer reduction by S in {2,4,8} when every weight is 1/2: the resize formula collapses to
//   out = (S[r][c] + S[r][c+1] + S[r+1][c] + S[r+1][c+1] + 2) >> 2,  r = S*dy + S/2 - 1, c = S*dx + S/2 - 1
// (with all four 11-bit weights equal to 1024 the two >>16 steps are exact).  Each thread produces 4
// adjacent output pixels from 4*S contiguous source bytes of two rows; xn is a multiple of 4 and the
// source is 16-byte aligned (checked on the host).
// The four source bytes of one output pixel gathered into a dword (v_perm_b32: selector bytes 0..3 pick from lo,
// 4..7 from hi), summed with the rounding term by one v_sad_u8 against zero, and shifted: (a + b + c + d + 2) >> 2.
__device__ __forceinline__ uint32_t box_px(uint32_t hi, uint32_t lo, uint32_t sel) {
    return __builtin_amdgcn_sad_u8(__builtin_amdgcn_perm(hi, lo, sel), 0u, 2u) >> 2;
}

template <int S>
__device__ __forceinline__ void box_tile(const uint8_t* __restrict__ src, long long spitch, uint8_t* __restrict__ dst,
                                         long long dpitch, int xb, int xn, int yb, int yn) {
    const int ncg = xn >> 2;
    const int items = yn * ncg;
    const float rncg = 1.0f / (float)ncg;
#pragma unroll 2
    for (int item = threadIdx.x; item < items; item += 256) {
        const int row = (int)(((float)item + 0.5f) * rncg);           // items <= 2^13: exact
        const int cg = item - row * ncg;
        const uint8_t* r0 = src + (long long)(S * (yb + row) + S / 2 - 1) * spitch + (long long)S * (xb + 4 * cg);
        const uint8_t* r1 = r0 + spitch;
        uint32_t packed;
        if (S == 2) {
            // output pixel j: bytes 2j, 2j + 1 of both rows' 8 bytes
            const uint2 u = *reinterpret_cast<const uint2*>(r0), v = *reinterpret_cast<const uint2*>(r1);
            packed = box_px(v.x, u.x, 0x05040100u) | (box_px(v.x, u.x, 0x07060302u) << 8) |
                     (box_px(v.y, u.y, 0x05040100u) << 16) | (box_px(v.y, u.y, 0x07060302u) << 24);
        } else if (S == 4) {
            // output pixel j: bytes 1, 2 of dword j of both rows
            const uint4 u = *reinterpret_cast<const uint4*>(r0), v = *reinterpret_cast<const uint4*>(r1);
            packed = box_px(v.x, u.x, 0x06050201u) | (box_px(v.y, u.y, 0x06050201u) << 8) |
                     (box_px(v.z, u.z, 0x06050201u) << 16) | (box_px(v.w, u.w, 0x06050201u) << 24);
        } else {
            // output pixel j: byte 3 of dword 2j and byte 0 of dword 2j + 1 of both rows
            const uint4 u0 = reinterpret_cast<const uint4*>(r0)[0], u1 = reinterpret_cast<const uint4*>(r0)[1];
            const uint4 v0 = reinterpret_cast<const uint4*>(r1)[0], v1 = reinterpret_cast<const uint4*>(r1)[1];
            const uint32_t a0 = __builtin_amdgcn_perm(u0.y, u0.x, 0x0c0c0403u), b0 = __builtin_amdgcn_perm(v0.y, v0.x, 0x0c0c0403u);
            const uint32_t a1 = __builtin_amdgcn_perm(u0.w, u0.z, 0x0c0c0403u), b1 = __builtin_amdgcn_perm(v0.w, v0.z, 0x0c0c0403u);
            const uint32_t a2 = __builtin_amdgcn_perm(u1.y, u1.x, 0x0c0c0403u), b2 = __builtin_amdgcn_perm(v1.y, v1.x, 0x0c0c0403u);
            const uint32_t a3 = __builtin_amdgcn_perm(u1.w, u1.z, 0x0c0c0403u), b3 = __builtin_amdgcn_perm(v1.w, v1.z, 0x0c0c0403u);
            packed = box_px(b0, a0, 0x05040100u) | (box_px(b1, a1, 0x05040100u) << 8) |
                     (box_px(b2, a2, 0x05040100u) << 16) | (box_px(b3, a3, 0x05040100u) << 24);
        }
        *reinterpret_cast<uint32_t*>(dst + (long long)(yb + row) * dpitch + xb + 4 * cg) = packed;
    }
}

template <bool ALIGNED>
__global__ __launch_bounds__(256, ALIGNED ? 8 : 5) void pyr_build_kernel(PyrArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pyr_lds[];
    const unsigned per = 8u * (unsigned)a.units;
    const unsigned fg = blockIdx.x / per, rem = blockIdx.x - fg * per;
    const int unit = (int)(rem >> 3);
    const int frame = (int)(fg * 8u + (rem & 7u));
    if (frame >= a.nframes) return;
    int l = 0;
    for (int k = 1; k < a.nlev; k++)
        if (unit >= a.lev[k].unit0) l = k;
    const PyrLevelArgs& L = a.lev[l];
    const int local = unit - L.unit0;
    const int band = local / L.ncolt, colt = local - band * L.ncolt;
    const int xb = colt * L.xt, xn = min(L.xt, L.dw - xb);
    const int yb = band * L.rb, yn = min(L.rb, L.dh - yb);
    const int tid = threadIdx.x;
    if (L.box) {        // exact 2x / 4x / 8x reduction: straight from global memory, no LDS
        const uint8_t* S = a.src + (long long)frame * a.sstride;
        uint8_t* D = L.dst + (long long)frame * L.dstride;
        if (L.box == 2) box_tile<2>(S, a.spitch, D, L.dpitch, xb, xn, yb, yn);
        else if (L.box == 4) box_tile<4>(S, a.spitch, D, L.dpitch, xb, xn, yb, yn);
        else box_tile<8>(S, a.spitch, D, L.dpitch, xb, xn, yb, yn);
        return;
    }
    const int span = L.span;
    const int cx0 = L.xofs[xb] & ~3;                                   // first staged source column
    const int ndw = min(span, ((a.sw + 3) & ~3) - cx0) >> 2;           // dwords staged per row

    unsigned char* rows = pyr_lds;
    uint2* xtab = reinterpret_cast<uint2*>(pyr_lds + ((2 * L.rb * span + 7) & ~7));
    uint32_t* ytab = reinterpret_cast<uint32_t*>(xtab + ((L.xt + 3) & ~3));

    int* rowidx = reinterpret_cast<int*>(ytab + L.rb);                  // clamped source row of each staged row

    const int xpad = (xn + 3) & ~3;
    for (int i = tid; i < xpad; i += 256) {
        const int dx = min(xb + i, L.dw - 1);
        const int sx = L.xofs[dx];
        const unsigned o = (unsigned)(sx - cx0), o1 = (unsigned)(min(sx + 1, a.sw - 1) - cx0);
        xtab[i] = make_uint2(o | (o1 << 16), L.xcoef[dx]);
    }
    if (tid < yn) ytab[tid] = L.ycoef[yb + tid];
    if (tid < 2 * yn) rowidx[tid] = min(max(L.yofs[yb + (tid >> 1)] + (tid & 1), 0), a.sh - 1);
    __syncthreads();

    // source rows -> LDS.  Every load of a batch is issued (unconditionally, on a clamped index) before
    // the first LDS write, so a thread waits for one memory round trip per batch, not one per dword.
    const uint8_t* S = a.src + (long long)frame * a.sstride;
    const int total = 2 * yn * ndw;
    const float rndw = 1.0f / (float)ndw;
    constexpr int NB = 12;
    for (int base = tid; base < total; base += 256 * NB) {
        uint32_t v[NB];
        int dsto[NB];
#pragma unroll
        for (int u = 0; u < NB; u++) {
            const int idx = min(base + u * 256, total - 1);
            const int k = (int)(((float)idx + 0.5f) * rndw);               // idx < 2^16: exact
            const int c = idx - k * ndw;
            const uint8_t* R = S + (long long)rowidx[k] * a.spitch + cx0;
            dsto[u] = k * span + 4 * c;
            if (ALIGNED) {
                v[u] = *reinterpret_cast<const uint32_t*>(R + 4 * c);
            } else {                                                       // any pitch / base: byte loads, clamped
                const int lim = a.sw - 1 - cx0;
                v[u] = (uint32_t)R[min(4 * c, lim)] | ((uint32_t)R[min(4 * c + 1, lim)] << 8) |
                       ((uint32_t)R[min(4 * c + 2, lim)] << 16) | ((uint32_t)R[min(4 * c + 3, lim)] << 24);
            }
        }
#pragma unroll
        for (int u = 0; u < NB; u++)
            if (base + u * 256 < total) *reinterpret_cast<uint32_t*>(rows + dsto[u]) = v[u];
    }
    __syncthreads();

    const int ncg = xpad >> 2;
    const int items = yn * ncg;
    const float rncg = 1.0f / (float)ncg;
    uint8_t* D = L.dst + (long long)frame * L.dstride + (long long)yb * L.dpitch + xb;
    for (int item = tid; item < items; item += 256) {
        const int row = (int)(((float)item + 0.5f) * rncg);           // items <= 2048: exact
        const int cg = item - row * ncg;
        const uint32_t yc = ytab[row];
        const int b0 = (int)(int16_t)(yc & 0xffff), b1 = (int)(int16_t)(yc >> 16);
        const unsigned char* R0 = rows + (2 * row) * span;
        const unsigned char* R1 = R0 + span;
        uint32_t packed = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint2 e = xtab[cg * 4 + i];
            const unsigned o = e.x & 0xffffu, o1 = e.x >> 16;
            const int a0 = (int)(int16_t)(e.y & 0xffff), a1 = (int)(int16_t)(e.y >> 16);
            const int h0 = (int)R0[o] * a0 + (int)R0[o1] * a1;
            const int h1 = (int)R1[o] * a0 + (int)R1[o1] * a1;
            const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            packed |= (uint32_t)(v & 0xff) << (8 * i);
        }
        // rows are padded to a multiple of 4 bytes, so the last dword of a row may be stored whole
        *reinterpret_cast<uint32_t*>(D + (long long)row * L.dpitch + cg * 4) = packed;
    }
}

const int PYR_LDS_BUDGET = 40 * 1024;

int pyr_lds_bytes(int rb, int xt, int span) {
    return ((2 * rb * span + 7) & ~7) + ((xt + 3) & ~3) * 8 + rb * 4 + 2 * rb * 4;   // rows, xtab, ytab, rowidx
}

// Largest staged row span (bytes, multiple of 4) over the column tiles of width xt.
int pyr_span(const std::vector<int32_t>& xofs, int sw, int dw, int xt) {
    int span = 4;
    for (int xb = 0; xb < dw; xb += xt) {
        const int xe = std::min(dw, xb + xt);
        const int cx0 = xofs[xb] & ~3;
        const int last = std::min(sw - 1, xofs[xe - 1] + 1);
        span = std::max(span, ((last + 4) & ~3) - cx0);
    }
    return span;
}

}  // namespace

// s in {2,4,8} when the level is an exact reduction by s whose tap tables are regular (offset
// s*d + s/2 - 1, both weights 1024) and whose width allows whole-dword groups; else 0.
int slam_pyr_detect_box(const std::vector<int32_t>& xofs, const std::vector<uint32_t>& xcoef,
                        const std::vector<int32_t>& yofs, const std::vector<uint32_t>& ycoef,
                        int sw, int sh, int dw, int dh) {
    if (dw < 4 || (dw & 3) || sw % dw || sh % dh || sw / dw != sh / dh) return 0;
    const int s = sw / dw;
    if (s != 2 && s != 4 && s != 8) return 0;
    const uint32_t half = 1024u | (1024u << 16);
    for (int d = 0; d < dw; d++) if (xofs[d] != s * d + s / 2 - 1 || xcoef[d] != half) return 0;
    for (int d = 0; d < dh; d++) if (yofs[d] != s * d + s / 2 - 1 || ycoef[d] != half) return 0;
    return s;
}

// Tile shape of one resized level: as many output rows / columns per workgroup as fit the LDS budget.
int slam_pyr_plan_level(const std::vector<int32_t>& xofs, int sw, int dw, int dh, slam_pyr_plan* out) {
    int xt = (std::min(dw, 1024) + 3) & ~3, rb = 8;
    int span = pyr_span(xofs, sw, dw, xt);
    while (pyr_lds_bytes(rb, xt, span) > PYR_LDS_BUDGET || span > 65532) {
        if (rb > 1 && span <= 65532) rb >>= 1;
        else if (xt > 4) { xt = ((xt >> 1) + 3) & ~3; span = pyr_span(xofs, sw, dw, xt); }
        else return SLAM_HIP_EINVAL;
    }
    out->xt = xt; out->ncolt = (dw + xt - 1) / xt;
    out->rb = rb; out->nband = (dh + rb - 1) / rb;
    out->span = span; out->lds = pyr_lds_bytes(rb, xt, span);
    return SLAM_HIP_OK;
}

namespace {

struct PyrLevelHost {
    uint8_t* dst; size_t dpitch, dstride; int dw, dh;
    const int32_t* xofs; const uint32_t* xcoef; const int32_t* yofs; const uint32_t* ycoef;
    slam_pyr_plan plan;
};

int pyr_launch(slam_ctx* ctx, const uint8_t* src, int sw, int sh, size_t spitch, size_t sstride, int nframes,
               const PyrLevelHost* lev, int nlev) {
    if (nlev == 0) return SLAM_HIP_OK;
    PyrArgs a{};
    a.src = src; a.spitch = (long long)spitch; a.sstride = (long long)sstride;
    a.sw = sw; a.sh = sh; a.nframes = nframes; a.nlev = nlev;
    int units = 0, lds = 0;
    for (int l = 0; l < nlev; l++) {
        PyrLevelArgs& L = a.lev[l];
        L.dst = lev[l].dst; L.dpitch = (long long)lev[l].dpitch; L.dstride = (long long)lev[l].dstride;
        L.xofs = lev[l].xofs; L.xcoef = lev[l].xcoef; L.yofs = lev[l].yofs; L.ycoef = lev[l].ycoef;
        L.dw = lev[l].dw; L.dh = lev[l].dh;
        L.xt = lev[l].plan.xt; L.ncolt = lev[l].plan.ncolt; L.rb = lev[l].plan.rb; L.nband = lev[l].plan.nband;
        L.span = lev[l].plan.span; L.unit0 = units;
        L.box = (lev[l].plan.box && ((((uintptr_t)src | spitch | sstride) & 15) == 0)) ? lev[l].plan.box : 0;
        units += L.ncolt * L.nband;
        if (!L.box) lds = std::max(lds, lev[l].plan.lds);     // box levels stage nothing and need no LDS
    }
    a.units = units;
    const long long groups = ((long long)nframes + 7) / 8;
    const long long grid = groups * 8 * units;
    if (grid > 0x7fffffffLL) return slam_set_error(ctx, SLAM_HIP_EINVAL, "pyramid: batch too large for one launch (%lld workgroups)", grid);
    const bool aligned = (((uintptr_t)src | spitch | sstride) & 3) == 0;
    if (aligned) hipLaunchKernelGGL(pyr_build_kernel<true>, dim3((unsigned)grid), dim3(256), lds, ctx->stream, a);
    else hipLaunchKernelGGL(pyr_build_kernel<false>, dim3((unsigned)grid), dim3(256), lds, ctx->stream, a);
    SLAM_HIP_CHECK(ctx, hipGetLastError());
    return SLAM_HIP_OK;
}

}  // namespace

int slam_launch_pyramid(slam_ctx* ctx, slam_pyr* pyr) {
    PyrLevelHost lev[SLAM_HIP_MAX_LEVELS];
    int n = 0;
    for (int l = 0; l < pyr->nlev; l++) {
        if (!pyr->d_level[l]) continue;   // aliased level
        lev[n++] = PyrLevelHost{pyr->d_level[l], pyr->lpitch[l], pyr->lstride[l], pyr->lw[l], pyr->lh[l],
                                pyr->d_xofs[l], pyr->d_xcoef[l], pyr->d_yofs[l], pyr->d_ycoef[l], pyr->plan[l]};
    }
    return pyr_launch(ctx, pyr->d_frames, pyr->w, pyr->h, pyr->pitch0, pyr->stride0, pyr->nframes, lev, n);
}

static void pyr_refresh_dev(slam_pyr* p) {
    p->dev.nlev = p->nlev;
    p->dev.nframes = p->nframes;
    for (int l = 0; l < p->nlev; l++) {
        slam_level_dev& d = p->dev.lev[l];
        d.w = p->lw[l]; d.h = p->lh[l]; d.sub = p->subs[l];
        const float sub = p->subs[l];
        int ex = 0;
        d.pow2 = (std::frexp(sub, &ex) == 0.5f && ex > -20 && ex < 20) ? 1 : 0;
        d.inv_sub = 1.f / sub;
        // feature.cpp:119,134: fullResSearchRad / sub on the coarsest level; :136: 1.3 * sub[i+1] / sub[i] below it
        // (a double expression stored to float)
        if (l == p->nlev - 1) d.rad = 20.f / sub;
        else d.rad = (float)((1.3 * (double)p->subs[l + 1]) / (double)sub);
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
        if (slam_pyr_plan_level(xo, w, dw, dh, &p->plan[l]) != SLAM_HIP_OK) {
            rc = slam_set_error(ctx, SLAM_HIP_EINVAL, "pyramid_create: level %d cannot be tiled", l); break;
        }
        p->plan[l].box = slam_pyr_detect_box(xo, xc, yo, yc, w, h, dw, dh);
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
    for (int f = 0; f < p->nframes; f++) {
        const int rcu = slam_upload_image(ctx, p->d_upload + (size_t)f * dstride, dpitch, h_frames + (size_t)f * frame_stride, pitch,
                                          (size_t)p->w, (size_t)p->h, (size_t)f * dstride);
        if (rcu != SLAM_HIP_OK) { (void)hipStreamSynchronize(ctx->stream); return rcu; }
    }
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
    PyrLevelHost lev{base + off_dst, dp, 0, dw, dh, d_xo, d_xc, d_yo, d_yc, {}};
    if (slam_pyr_plan_level(xo, sw, dw, dh, &lev.plan) != SLAM_HIP_OK)
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "debug_resize: cannot be tiled");
    lev.plan.box = slam_pyr_detect_box(xo, xc, yo, yc, sw, sh, dw, dh);
    rc = pyr_launch(ctx, base + off_src, sw, sh, (size_t)sw, 0, 1, &lev, 1);
    if (rc != SLAM_HIP_OK) return rc;
    SLAM_HIP_CHECK(ctx, hipMemcpy2DAsync(dst, dpitch, base + off_dst, dp, dw, dh, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_HIP_OK;
}
