// Device-side pieces shared by the template-search kernels: job geometry (matching.cpp:17-49),
// the score-map cell classification (which cells the reference's loops visit and with which
// formula), the two score formulas, template access (direct or resized on the fly) and the
// workgroup argmax.  gfx950 only; wavefront = 64 lanes.
#pragma once

#include "slam_internal.h"

#define SLAM_WG 64           // one wavefront per job: no cross-wave synchronisation, many jobs per CU
#define SLAM_MAX_TMPL 512    // max template width/height of a track job (the reference caps projections at 500)

// Diagnostic build only (-DSLAM_STAMPS, `make stamps`): lane 0 of every job records s_memtime at
// phase boundaries into a buffer of its own; the product build compiles STAMP() to nothing.
#ifdef SLAM_STAMPS
#define SLAM_STAMP_SLOTS 64
__device__ unsigned long long* g_slam_stamps = nullptr;
__device__ __forceinline__ void slam_stamp(int& slot, int id) {
    if (g_slam_stamps && threadIdx.x == 0 && slot < SLAM_STAMP_SLOTS) {
        unsigned long long t;
#ifdef SLAM_STAMPS_REALTIME       // the 100 MHz counter all CUs share (launch timelines) instead of the CU's shader clock (phase costs)
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
#else
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
#endif
        g_slam_stamps[(size_t)blockIdx.x * SLAM_STAMP_SLOTS + slot] = (t << 8) | (unsigned)id;
    }
    slot++;
}
#define STAMP(id) slam_stamp(stamp_slot, (id))
#define STAMP_DECL int stamp_slot = 0
#define STAMP_ARG , int& stamp_slot
#define STAMP_PASS , stamp_slot
#define STAMP_LEVEL(i) ((void)0)
#elif defined(SLAM_DIAG_EXIT_AT)
// diagnostic builds that end every wave at one phase boundary of one pyramid level: the instruction counters of such
// a build, minus those of the build that ends one boundary earlier, are the cost of the phase in between (never shipped)
#ifndef SLAM_DIAG_EXIT_LEVEL
#define SLAM_DIAG_EXIT_LEVEL -1
#endif
#define STAMP(id) do { if ((id) == SLAM_DIAG_EXIT_AT && diag_lvl == SLAM_DIAG_EXIT_LEVEL) __builtin_amdgcn_endpgm(); } while (0)
#define STAMP_DECL int diag_lvl = -1
#define STAMP_ARG , int diag_lvl
#define STAMP_PASS , diag_lvl
#define STAMP_LEVEL(i) diag_lvl = (i)
#else
#define STAMP(id) ((void)0)
#define STAMP_DECL ((void)0)
#define STAMP_ARG
#define STAMP_PASS
#define STAMP_LEVEL(i) ((void)0)
#endif

namespace slamdev {

// The lane index as a value the optimiser cannot see through: what a phase derives from it (LDS addresses, float
// copies, neighbour offsets) is then computed inside that phase instead of being hoisted out of the level loop of a
// walk into registers that stay live, and spill, across all of it.
__device__ __forceinline__ int lane_here() {
    int l = (int)threadIdx.x;
    asm volatile("" : "+v"(l));
    return l;
}

// ---- score formulas -------------------------------------------------------------------------
// Exact integer sums in, IEEE single/double operations out; every operation is one correctly
// rounded step, in the order of matching.cpp:70-72 (float cc, i2, p2; cc / (i2*p2)).
__device__ __forceinline__ float score_f32(unsigned long long CC, unsigned long long SI, unsigned long long SP) {
    const float cc = (float)(double)CC;
    const float i2 = (float)__dsqrt_rn((double)SI);
    const float p2 = (float)__dsqrt_rn((double)SP);
    const float den = __fmul_rn(i2, p2);
    return __fdiv_rn(cc, den);
}

// cv::matchTemplate(CV_TM_CCORR_NORMED) normalisation (matching.cpp:152): double arithmetic,
// |num| < t -> num/t, |num| < 1.125 t -> +-1, else 0.
__device__ __forceinline__ float score_mt(unsigned long long CC, unsigned long long SI, unsigned long long SP) {
    double num = (double)CC;
    const double tn = __dsqrt_rn((double)SP);
    const double t = __dmul_rn(__dsqrt_rn((double)SI), tn);
    if (fabs(num) < t) num = __ddiv_rn(num, t);
    else if (fabs(num) < __dmul_rn(t, 1.125)) num = num > 0 ? 1.0 : -1.0;
    else num = 0.0;
    return (float)num;
}

// The same two formulas with the template-side factor computed once per job:
//   p2f = (float)sqrt((double)SP) for the f32 formula, tn = sqrt((double)SP) for matchTemplate.
struct Norms { float p2f; double tn; };

__device__ __forceinline__ Norms make_norms(unsigned long long SP) {
    Norms n;
    n.tn = __dsqrt_rn((double)SP);
    n.p2f = (float)n.tn;
    return n;
}

__device__ __forceinline__ float score_f32_n(unsigned long long CC, unsigned long long SI, float p2f) {
    const float cc = (float)(double)CC;
    const float i2 = (float)__dsqrt_rn((double)SI);
    return __fdiv_rn(cc, __fmul_rn(i2, p2f));
}

__device__ __forceinline__ float score_mt_n(unsigned long long CC, unsigned long long SI, double tn) {
    double num = (double)CC;
    const double t = __dmul_rn(__dsqrt_rn((double)SI), tn);
    if (fabs(num) < t) num = __ddiv_rn(num, t);
    else if (fabs(num) < __dmul_rn(t, 1.125)) num = num > 0 ? 1.0 : -1.0;
    else num = 0.0;
    return (float)num;
}

// exact unsigned division by a small runtime constant d via its reciprocal rcp = ceil(2^32/d);
// valid while n*d < 2^32
__device__ __forceinline__ unsigned rcp_u32(unsigned d) { return d == 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d); }
__device__ __forceinline__ unsigned div_rcp(unsigned n, unsigned d, unsigned rcp) { return d == 1 ? n : __umulhi(n, rcp); }

// floor(n / d) for 0 <= n < 2^20, d >= 1 through one v_rcp_f32 (1 ulp): (n + 0.5) / d is at least
// 0.5/d away from an integer, far more than the float rounding error at these magnitudes.
__device__ __forceinline__ int small_div(int n, int d) {
    return (int)(((float)n + 0.5f) * __builtin_amdgcn_rcpf((float)d));
}
// the same for wave-uniform operands: the quotient goes back to a scalar register so that the
// integer arithmetic that follows stays on the scalar unit
__device__ __forceinline__ int small_div_u(int n, int d) { return __builtin_amdgcn_readfirstlane(small_div(n, d)); }

// ---- geometry -------------------------------------------------------------------------------
struct Geo {
    const uint8_t* im;        // level image of the job's frame
    int w, h;
    long long ipitch;
    const uint8_t* amask;     // areaMask0: already offset to the cropped origin, or nullptr
    long long ampitch;
    int ax, ay, aw, ah;       // cropped area (matching.cpp:19-25)
    int pw, ph;
    int x0, y0;               // image position of patch pixel (0,0) for cell (0,0) (matching.cpp:38-39)
    int xl0, xl1, yl0, yl1;   // cells whose whole patch is inside the image (matching.cpp:44-49)
    int stride;
    float rstride;            // 1 / stride (v_rcp_f32): n % stride without an integer division
    int yf;                   // first multiple of stride >= yl0
    bool masked;              // patchMask given
    bool exact;               // useExactAreaMask
    bool reject;              // area misses the image (matching.cpp:26-30)
};

__device__ __forceinline__ Geo make_geo(const uint8_t* im, int w, int h, long long ipitch,
                                        int pw, int ph, int ax0, int ay0, int aw0, int ah0,
                                        const uint8_t* amask, int stride, bool masked, bool exact) {
    Geo g;
    g.im = im; g.w = w; g.h = h; g.ipitch = ipitch; g.pw = pw; g.ph = ph;
    g.stride = stride; g.masked = masked; g.exact = exact;
    const int ax = max(ax0, 0), ay = max(ay0, 0);
    // 64-bit so that absurd rectangles cannot overflow
    const long long awl = min((long long)w - ax, (long long)aw0 + ax0 - ax);
    const long long ahl = min((long long)h - ay, (long long)ah0 + ay0 - ay);
    g.ax = ax; g.ay = ay;
    g.reject = (ax >= w) || (ay >= h) || (ax + awl <= 0) || (ay + ahl <= 0) || awl <= 0 || ahl <= 0;
    g.aw = g.reject ? 0 : (int)awl;
    g.ah = g.reject ? 0 : (int)ahl;
    g.amask = amask ? amask + (long long)(ay - ay0) * aw0 + (ax - ax0) : nullptr;
    g.ampitch = aw0;
    g.x0 = ax - pw / 2;
    g.y0 = ay - ph / 2;
    const int x0ua = max(g.x0, 0), x1ua = min(w, g.x0 + g.aw + pw - 1);
    const int y0ua = max(g.y0, 0), y1ua = min(h, g.y0 + g.ah + ph - 1);
    g.xl0 = x0ua - g.x0; g.xl1 = x1ua - g.x0 - pw + 1;
    g.yl0 = y0ua - g.y0; g.yl1 = y1ua - g.y0 - ph + 1;
    g.rstride = __builtin_amdgcn_rcpf((float)stride);
    g.yf = (stride > (1 << 19)) ? ((g.yl0 + stride - 1) / stride * stride)
                                : (int)(((float)(g.yl0 + stride - 1) + 0.5f) * g.rstride) * stride;
    return g;
}

// n % g.stride for 0 <= n <= 2^20 (see small_div for the exactness of the float quotient)
__device__ __forceinline__ int mod_stride(const Geo& g, int n) {
    return n - (int)(((float)n + 0.5f) * g.rstride) * g.stride;
}

// row of the (wave-uniform) linear cell index idx = y*aw + x
__device__ __forceinline__ int row_of_idx(const Geo& g, int idx) {
    if ((long long)g.aw * g.ah <= (1 << 20)) return small_div_u(idx, g.aw);
    return __builtin_amdgcn_readfirstlane(idx / g.aw);
}

enum CellKind { CELL_NONE = 0, CELL_F32_FULL = 1, CELL_MT = 2, CELL_F32_CROP = 3 };

// Which formula (if any) the reference's loops apply to score cell (x,y):
//  - patchMask modes: interior lattice from (xl0,yl0) step stride (matching.cpp:64-73,110-118);
//  - no patchMask: every interior cell through matchTemplate (matching.cpp:151-153), reset to -1
//    where the area mask is 0 only with useExactAreaMask (matching.cpp:157-163);
//  - border lattice from (0,0) step stride with the jump over the interior block
//    (matching.cpp:76-104 and its three copies).
__device__ __forceinline__ int classify_cell(const Geo& g, int x, int y) {
    const bool inx = (x >= g.xl0) && (x < g.xl1);
    const bool interior = inx && (y >= g.yl0) && (y < g.yl1);
    const bool am = !g.amask || g.amask[(long long)y * g.ampitch + x] != 0;
    if (interior) {
        if (g.masked) {
            if (g.stride != 1 && (mod_stride(g, x - g.xl0) != 0 || mod_stride(g, y - g.yl0) != 0)) return CELL_NONE;
            return am ? CELL_F32_FULL : CELL_NONE;
        }
        if (g.exact && !am) return CELL_NONE;
        return CELL_MT;
    }
    if (g.stride != 1) {
        if (mod_stride(g, x) != 0) return CELL_NONE;
        bool on;
        if (inx && g.yf < g.yl1)   // this column's walk hits the interior block at yf and jumps to yl1
            on = (y < g.yl0) ? (mod_stride(g, y) == 0) : ((y >= g.yl1) && (mod_stride(g, y - g.yl1) == 0));
        else
            on = mod_stride(g, y) == 0;
        if (!on) return CELL_NONE;
    }
    if (!am) return CELL_NONE;
    const int dl = max(0, -g.x0 - x), dr = max(0, g.x0 + x + g.pw - g.w);
    const int dt = max(0, -g.y0 - y), db = max(0, g.y0 + y + g.ph - g.h);
    return (dt + db < g.ph && dl + dr < g.pw) ? CELL_F32_CROP : CELL_NONE;
}

// ---- template access ------------------------------------------------------------------------
// A level template is either the job's patch as stored (sub == 1) or cv::resize of it to
// (tw, th) evaluated on demand from tap tables in LDS (feature.cpp:130-131).
struct Tmpl {
    const uint8_t* patch;     // source patch, pitch srcw
    const uint8_t* mask;      // source mask (0 / non-zero), or nullptr
    int srcw, srch;
    int tw, th;               // size at this level
    bool resized;
    double scale_x, scale_y;  // cv::resize's scale = 1 / ((double)dsize / ssize), when resized
    const uint8_t* limit;     // end of the buffer patch and mask live in (no byte at or past it is read)
};

// Four consecutive bytes at p (any alignment) as one little-endian dword; bytes at or beyond
// `limit` read as 0.
__device__ __forceinline__ unsigned load4(const uint8_t* p, const uint8_t* limit) {
    unsigned v = 0;
    if (p + 4 <= limit) {
        __builtin_memcpy(&v, p, 4);
    } else {
#pragma unroll
        for (int b = 0; b < 4; b++)
            if (p + b < limit) v |= (unsigned)p[b] << (8 * b);
    }
    return v;
}

// cv::resize's scale factor for one axis
__device__ __forceinline__ double resize_scale(int ssize, int dsize) {
    return __ddiv_rn(1.0, __ddiv_rn((double)dsize, (double)ssize));
}

// One tap of cv::resize(INTER_LINEAR) for destination index d, with the same double/float steps
// as OpenCV's resize(): f = (float)((d+0.5)*scale - 0.5); s = floor(f); f -= s; weights =
// round-half-even((1-f)*2048), (f*2048) saturated to int16.  Columns reset the fraction at the
// image edges (clamp_fraction); rows keep it and clamp the row index when it is used.
__device__ __forceinline__ void resize_tap(int d, double scale, int ssize, bool clamp_fraction, int& ofs, int& c0, int& c1) {
    const double t = __dmul_rn((double)d + 0.5, scale);
    float f = (float)__dadd_rn(t, -0.5);
    int s = (int)floorf(f);
    f = __fsub_rn(f, (float)s);
    if (clamp_fraction) {
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    }
    c0 = min(max(__float2int_rn(__fmul_rn(__fsub_rn(1.f, f), 2048.f)), -32768), 32767);
    c1 = min(max(__float2int_rn(__fmul_rn(f, 2048.f)), -32768), 32767);
    ofs = s;
}

// All products fit 24-bit operands (pixels < 2^8, weights <= 2^11, row sums >> 4 < 2^15, template sides < 2^23), so
// the full-rate 24-bit multiplier gives the same integers as 32-bit arithmetic.
struct ResizeSrc { int p00, p01, p10, p11; };

// the four source pixels of one resized pixel (rows clamped when used, second column clamped at the edge)
__device__ __forceinline__ ResizeSrc resize_fetch(const uint8_t* S, int sw, int sh, int sx, int sy) {
    const int r0 = min(max(sy, 0), sh - 1), r1 = min(max(sy + 1, 0), sh - 1);
    const int sx1 = min(sx + 1, sw - 1);
    const unsigned o0 = __umul24((unsigned)r0, (unsigned)sw), o1 = __umul24((unsigned)r1, (unsigned)sw);
    ResizeSrc r;
    r.p00 = S[o0 + (unsigned)sx]; r.p01 = S[o0 + (unsigned)sx1];
    r.p10 = S[o1 + (unsigned)sx]; r.p11 = S[o1 + (unsigned)sx1];
    return r;
}

__device__ __forceinline__ int resize_blend(ResizeSrc r, int a0, int a1, int b0, int b1, bool binarize) {
    if (binarize) { r.p00 = r.p00 != 0; r.p01 = r.p01 != 0; r.p10 = r.p10 != 0; r.p11 = r.p11 != 0; }
    const int h0 = __mul24(r.p00, a0) + __mul24(r.p01, a1);
    const int h1 = __mul24(r.p10, a0) + __mul24(r.p11, a1);
    return (((__mul24(b0, h0 >> 4) >> 16) + (__mul24(b1, h1 >> 4) >> 16) + 2) >> 2) & 0xff;
}

__device__ __forceinline__ int resize_px(const uint8_t* S, int sw, int sh, int sx, int a0, int a1, int sy, int b0, int b1, bool binarize) {
    return resize_blend(resize_fetch(S, sw, sh, sx, sy), a0, a1, b0, b1, binarize);
}

// patch value and mask bit (0/1) of template pixel (row u, column v)
__device__ __forceinline__ void tmpl_fetch(const Tmpl& t, int u, int v, unsigned& p, unsigned& m) {
    if (!t.resized) {
        p = t.patch[(long long)u * t.srcw + v];
        m = t.mask ? (t.mask[(long long)u * t.srcw + v] != 0) : 1u;
    } else {
        int sx, a0, a1, sy, b0, b1;
        resize_tap(v, t.scale_x, t.srcw, true, sx, a0, a1);
        resize_tap(u, t.scale_y, t.srch, false, sy, b0, b1);
        p = (unsigned)resize_px(t.patch, t.srcw, t.srch, sx, a0, a1, sy, b0, b1, false);
        m = t.mask ? (unsigned)(resize_px(t.mask, t.srcw, t.srch, sx, a0, a1, sy, b0, b1, true) != 0) : 1u;
    }
}

// ---- sums over one cell ---------------------------------------------------------------------
struct Sums { unsigned long long cc, si, sp; };

// Sum over template rows [r0,r1) x columns [c0,c1) for the cell whose patch origin is image pixel
// (ix, iy); the caller guarantees the rectangle lies inside the image.  `lane`/`nlanes` split the
// pixels among cooperating lanes (each returns its partial sums).
__device__ __forceinline__ Sums cell_sums_partial(const Geo& g, const Tmpl& t, int ix, int iy,
                                                  int c0, int c1, int r0, int r1, int lane, int nlanes) {
    Sums s = {0ull, 0ull, 0ull};
    const int cw = c1 - c0;
    const int n = cw * (r1 - r0);
    for (int i = lane; i < n; i += nlanes) {
        const int u = r0 + i / cw, v = c0 + i % cw;
        unsigned p, m;
        tmpl_fetch(t, u, v, p, m);
        const unsigned I = g.im[(long long)(iy + u) * g.ipitch + (ix + v)] * m;
        const unsigned P = p * m;
        s.cc += I * P; s.si += I * I; s.sp += P * P;
    }
    return s;
}

// Inclusive scan step helpers on the 64 lanes with DPP moves (no LDS traffic): four row_shr steps
// inside each row of 16 lanes, then row_bcast:15 and row_bcast:31 carry the row totals upwards; lane
// 63 ends up with the reduction over all lanes.  Lanes without a source keep `identity`.
#define SLAM_DPP_ROW_SHR(n) (0x110 + (n))
#define SLAM_DPP_BCAST15 0x142
#define SLAM_DPP_BCAST31 0x143

__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, SLAM_DPP_ROW_SHR(1), 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, SLAM_DPP_ROW_SHR(2), 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, SLAM_DPP_ROW_SHR(4), 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, SLAM_DPP_ROW_SHR(8), 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, SLAM_DPP_BCAST15, 0xa, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, SLAM_DPP_BCAST31, 0xc, 0xf, false);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// Sum over the wavefront of per-lane values below 2^48 (three 16-bit digits: each digit sum stays
// below 2^22); the result is wave-uniform.
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
    const unsigned d0 = (unsigned)v & 0xffffu, d1 = (unsigned)(v >> 16) & 0xffffu, d2 = (unsigned)(v >> 32) & 0xffffu;
    return (unsigned long long)wave_sum_u32(d0) + ((unsigned long long)wave_sum_u32(d1) << 16) +
           ((unsigned long long)wave_sum_u32(d2) << 32);
}

// ---- argmax ---------------------------------------------------------------------------------
// cv::minMaxLoc semantics: strict '>' from -FLT_MAX in row-major order, so among equal maxima
// the smallest linear index wins and NaN never wins.
struct Best { float v; int idx; };

__device__ __forceinline__ void best_update(Best& b, float v, int idx) {
    if (v > b.v || (v == b.v && idx < b.idx)) { b.v = v; b.idx = idx; }
}

// Wave-wide argmax, wave-uniform result.  (score, index) is ordered as one 64-bit key: the score's
// bits made monotonic in the high word, the inverted index in the low word, so the maximum key is
// the largest score and, among equal scores, the smallest index.  Scores are never NaN here
// (best_update does not store them); lanes that saw nothing hold (-FLT_MAX, INT_MAX).
__device__ __forceinline__ Best best_wave(Best b) {
    const unsigned fb = (unsigned)__float_as_int(b.v);
    unsigned hi = (fb & 0x80000000u) ? ~fb : (fb | 0x80000000u);
    unsigned lo = ~(unsigned)b.idx;
#pragma unroll
    for (int step = 0; step < 6; step++) {
        unsigned oh, ol;
        switch (step) {
            case 0: oh = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, SLAM_DPP_ROW_SHR(1), 0xf, 0xf, false);
                    ol = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, SLAM_DPP_ROW_SHR(1), 0xf, 0xf, false); break;
            case 1: oh = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, SLAM_DPP_ROW_SHR(2), 0xf, 0xf, false);
                    ol = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, SLAM_DPP_ROW_SHR(2), 0xf, 0xf, false); break;
            case 2: oh = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, SLAM_DPP_ROW_SHR(4), 0xf, 0xf, false);
                    ol = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, SLAM_DPP_ROW_SHR(4), 0xf, 0xf, false); break;
            case 3: oh = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, SLAM_DPP_ROW_SHR(8), 0xf, 0xf, false);
                    ol = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, SLAM_DPP_ROW_SHR(8), 0xf, 0xf, false); break;
            case 4: oh = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, SLAM_DPP_BCAST15, 0xa, 0xf, false);
                    ol = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, SLAM_DPP_BCAST15, 0xa, 0xf, false); break;
            default: oh = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, SLAM_DPP_BCAST31, 0xc, 0xf, false);
                     ol = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, SLAM_DPP_BCAST31, 0xc, 0xf, false); break;
        }
        const bool take = oh > hi || (oh == hi && ol > lo);      // key 0 (no source lane) never wins
        hi = take ? oh : hi; lo = take ? ol : lo;
    }
    hi = (unsigned)__builtin_amdgcn_readlane((int)hi, 63);
    lo = (unsigned)__builtin_amdgcn_readlane((int)lo, 63);
    Best r;
    r.v = __int_as_float((int)((hi & 0x80000000u) ? (hi & 0x7fffffffu) : ~hi));
    r.idx = (int)~lo;
    return r;
}

}  // namespace slamdev
