// LDS-tiled packed-u8 search core: one wavefront (64 lanes) per job.
//
// Data layout in LDS (dynamic; the launch gives lds_total bytes, split by LdsPlan):
//   acc   : per-cell sums {CC, SI[, SPcrop]} of the current block (u32, or u64 for templates of
//           more than 66051 pixels), accumulated with LDS atomics
//   tmpl  : the current chunk of template rows, one 8-byte entry per 4-pixel group:
//           low dword  P0   = patch*mask, 4 pixels
//           high dword mask = 0xFF per kept pixel (0 where masked out or beyond the patch width)
//   tile  : image rows under all cells of the current block, zero outside the image; tile column 0
//           is a multiple of 4 in IMAGE columns so rows are fetched as aligned dwords
//   vtile : 0xFF where the tile pixel is inside the image (only for blocks with cropped cells)
//
// Work split: the block's cells are cut into strips of 4 consecutive cells of one row; a task is one
// strip x one slice of the chunk's template rows, and the 64 lanes walk the task list.  Per 4-pixel
// group a lane forms each cell's 4 image pixels with v_alignbyte_b32, masks them, and feeds two
// v_dot4_u32_u8 (CC += <I,P0>, SI += <Im,Im>); blocks with cropped cells add SPcrop += <P0&valid,P0>.
// Every sum is an exact integer, so the scores are bit-identical to a scalar evaluation.
//
// The code is written once (no per-mode copies outside the inner loop): the search kernels are
// instruction-cache bound if each mode gets its own copy of the staging and scoring code.
#pragma once

#include <cfloat>
#include <climits>

#include "match_common.h"

#define FAST_LDS_DEFAULT (16 * 1024)
#define FAST_J 4
#ifndef FAST_SLIDE_MIN_G
#define FAST_SLIDE_MIN_G 7            // template groups per row from which the sliding window energy of unmasked strips pays
#endif
#define FAST_MAX_G 128        // template width up to 512 pixels; wider ones use direct evaluation

namespace slamdev {

typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
typedef __attribute__((address_space(3))) unsigned char lds_u8;

struct CoreOut { int x, y; float response, dx, dy; };

// Split of the job's LDS: half for the accumulators, 3/16 for the template chunk, 5/16 for the tile.
struct LdsPlan {
    int acc_bytes, tmpl_bytes, tile_bytes;
};

__device__ __forceinline__ LdsPlan make_lds_plan(int lds_total) {
    LdsPlan p;
    p.acc_bytes = (lds_total / 2) & ~15;
    p.tmpl_bytes = (lds_total * 3 / 16) & ~15;
    p.tile_bytes = (lds_total - p.acc_bytes - p.tmpl_bytes) & ~15;
    return p;
}

struct BlockArgs {
    int xb, yb, strips, bh;
    bool border;
};

// One 4-pixel group of one template row for the 4 cells of a strip.  MASKED: the window is masked before its energy
// is summed; without a patch mask only the last group of a row (columns beyond the template width) needs that.
template <bool BORDER, bool MASKED>
__device__ __forceinline__ void strip_group(unsigned d0, unsigned dn, unsigned v0, unsigned vn, unsigned long long pm,
                                            unsigned (&cc)[4], unsigned (&si)[4], unsigned (&sp)[4]) {
    const unsigned P0 = (unsigned)pm, M = (unsigned)(pm >> 32);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned w = (k == 0) ? d0 : __builtin_amdgcn_alignbyte(dn, d0, k);
        const unsigned wm = MASKED ? (w & M) : w;
        cc[k] = __builtin_amdgcn_udot4(w, P0, cc[k], false);      // P0 is already masked
        si[k] = __builtin_amdgcn_udot4(wm, wm, si[k], false);
        if (BORDER) {
            const unsigned vm = (k == 0) ? v0 : __builtin_amdgcn_alignbyte(vn, v0, k);
            sp[k] = __builtin_amdgcn_udot4(P0 & vm, P0, sp[k], false);
        }
    }
}

// Accumulate template rows [ub, ue) (chunk-relative) for the 4 cells of one strip.  Tile row r
// holds image row (first image row of the tile) + r and the cell row `row` is relative to the
// block, so template row u of that cell lies in tile row row + u.  ALLKEPT: no patch mask, so every
// group but the last of a row keeps all four pixels.  SLIDE (with ALLKEPT, wide templates): the window energy of
// those all-kept groups is summed for the strip's first cell only; cell k = 1..3 sees the same bytes shifted by k, so
// its energy is the first cell's minus the squares of the k bytes that leave on the left (the row's first dword) plus
// those of the k bytes that enter on the right (the dword after the last all-kept group) - 12 instructions per
// template row instead of 3 dots per group.  All of it modulo 2^32, like the sums themselves.
template <bool BORDER, bool ALLKEPT, bool SLIDE>
__device__ __forceinline__ void strip_accumulate(const lds_u32* tile, const lds_u32* vtile, int tp,
                                                 const lds_u64* tmpl, int G, int ub, int ue, int row, int strip,
                                                 unsigned (&cc)[4], unsigned (&si)[4], unsigned (&sp)[4]) {
    static_assert(!SLIDE || (ALLKEPT && !BORDER), "the sliding energy is for unmasked interior strips");
    unsigned s0 = 0, out1 = 0, out2 = 0, out3 = 0;        // SLIDE: first cell's energy, squares that left cells 1..3
    for (int u = ub; u < ue; ++u) {
        const int tofs = (row + u) * tp + strip;
        const lds_u32* tr = tile + tofs;
        const lds_u32* vr = vtile + tofs;
        const lds_u64* pr = tmpl + u * G;
        unsigned d0 = tr[0];
        unsigned v0 = BORDER ? vr[0] : 0u;
        const int Gfull = ALLKEPT ? G - 1 : G;
        if (SLIDE) {
            const unsigned lo = d0;
#pragma unroll 2
            for (int g = 0; g < Gfull; ++g) {
                const unsigned dn = tr[g + 1];
                const unsigned P0 = (unsigned)pr[g];
                cc[0] = __builtin_amdgcn_udot4(d0, P0, cc[0], false);
                cc[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(dn, d0, 1), P0, cc[1], false);
                cc[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(dn, d0, 2), P0, cc[2], false);
                cc[3] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(dn, d0, 3), P0, cc[3], false);
                s0 = __builtin_amdgcn_udot4(d0, d0, s0, false);
                d0 = dn;
            }
            // d0: the dword after the all-kept groups (the row's first dword again when there are none: nothing moves)
            si[1] = __builtin_amdgcn_udot4(d0 & 0x000000FFu, d0, si[1], false); out1 = __builtin_amdgcn_udot4(lo & 0x000000FFu, lo, out1, false);
            si[2] = __builtin_amdgcn_udot4(d0 & 0x0000FFFFu, d0, si[2], false); out2 = __builtin_amdgcn_udot4(lo & 0x0000FFFFu, lo, out2, false);
            si[3] = __builtin_amdgcn_udot4(d0 & 0x00FFFFFFu, d0, si[3], false); out3 = __builtin_amdgcn_udot4(lo & 0x00FFFFFFu, lo, out3, false);
        } else {
#pragma unroll 2
            for (int g = 0; g < Gfull; ++g) {
                const unsigned dn = tr[g + 1];
                const unsigned vn = BORDER ? vr[g + 1] : 0u;
                strip_group<BORDER, !ALLKEPT>(d0, dn, v0, vn, pr[g], cc, si, sp);
                d0 = dn; v0 = vn;
            }
        }
        if (ALLKEPT) {
            const unsigned dn = tr[G];
            const unsigned vn = BORDER ? vr[G] : 0u;
            strip_group<BORDER, true>(d0, dn, v0, vn, pr[G - 1], cc, si, sp);
        }
    }
    if (SLIDE) { si[0] += s0; si[1] += s0 - out1; si[2] += s0 - out2; si[3] += s0 - out3; }
}

__device__ __forceinline__ unsigned long long acc_read(const lds_u8* acc, int i, bool big) {
    if (big) return reinterpret_cast<const lds_u64*>(acc)[i];
    return reinterpret_cast<const lds_u32*>(acc)[i];
}

// Score-map value of block cell (x,y) from the block's accumulators (-1 for cells the reference
// never visits).  One sqrt(double) of the window energy serves all three formulas.
// `on_lattice`: the caller enumerates only cells of the interior lattice of a block without border
// cells, so the lattice and border tests of classify_cell are already decided.
__device__ __forceinline__ float cell_from_acc(const Geo& g, const lds_u8* acc, const BlockArgs& ba, bool big,
                                               int x, int y, const Norms& nm, bool on_lattice = false) {
    int kind;
    if (on_lattice) {
        const bool am = !g.amask || g.amask[(long long)y * g.ampitch + x] != 0;
        kind = g.masked ? (am ? CELL_F32_FULL : CELL_NONE) : ((g.exact && !am) ? CELL_NONE : CELL_MT);
    } else {
        kind = classify_cell(g, x, y);
    }
    if (kind == CELL_NONE) return -1.f;                    // matching.cpp:53
    const int aw = ba.border ? 3 : 2;
    const int c = ((y - ba.yb) * ba.strips) * FAST_J + (x - ba.xb);      // (row*strips + strip)*4 + k
    const unsigned long long CC = acc_read(acc, c * aw + 0, big);
    const unsigned long long SI = acc_read(acc, c * aw + 1, big);
    const double si = __dsqrt_rn((double)SI);
    if (kind == CELL_MT) {                                 // cv::matchTemplate(CV_TM_CCORR_NORMED), matching.cpp:152
        double num = (double)CC;
        const double t = __dmul_rn(si, nm.tn);
        if (fabs(num) < t) num = __ddiv_rn(num, t);
        else if (fabs(num) < __dmul_rn(t, 1.125)) num = num > 0 ? 1.0 : -1.0;
        else num = 0.0;
        return (float)num;
    }
    float p2 = nm.p2f;                                     // matching.cpp:63,109
    if (kind == CELL_F32_CROP) p2 = (float)__dsqrt_rn((double)acc_read(acc, c * aw + 2, big));   // matching.cpp:99
    return __fdiv_rn((float)(double)CC, __fmul_rn((float)si, p2));       // matching.cpp:70-72
}

// Single-precision estimate of cell_from_acc for an enumerated cell of a block without border cells and with 32-bit
// sums: -1 for a cell the reference never visits, -FLT_MAX where the exact score is NaN (it never wins), else
// CC / (sqrt(SI) * |P0|) up to a relative error below 1e-6 (matchTemplate's value is the same quotient, capped at 1).
__device__ __forceinline__ float cell_estimate(const Geo& g, const lds_u8* acc, const BlockArgs& ba, int x, int y, float rn) {
    const bool am = !g.amask || g.amask[(long long)y * g.ampitch + x] != 0;
    if (!am && (g.masked || g.exact)) return -1.f;
    const int c = ((y - ba.yb) * ba.strips) * FAST_J + (x - ba.xb);
    const unsigned CC = reinterpret_cast<const lds_u32*>(acc)[c * 2], SI = reinterpret_cast<const lds_u32*>(acc)[c * 2 + 1];
    if (SI == 0u) return g.masked ? -FLT_MAX : 0.f;
    const float e = (float)CC * __builtin_amdgcn_rsqf((float)SI) * rn;
    return g.masked ? e : fminf(e, 1.f);
}

// Lane -> (row, column) walk over a rows x width rectangle without per-element divisions:
// when width <= 64 a lane keeps its column and steps rows by 64/width, else rows are done one at a
// time with the lanes striding over the columns.
struct Walk2D {
    int r0, c0, rstep, cstep;
};

__device__ __forceinline__ Walk2D make_walk(int width, int rows) {
    Walk2D w;
    const int lane = lane_here();
    if (width <= SLAM_WG) {
        const float rw = __builtin_amdgcn_rcpf((float)width);
        const int per = (int)(64.5f * rw);                               // rows handled per pass
        const int lr = (int)(((float)lane + 0.5f) * rw);
        w.c0 = lane - lr * width;
        w.r0 = lr < per ? lr : rows;                       // surplus lanes do nothing
        w.rstep = per;
        w.cstep = width;
    } else {
        w.r0 = 0; w.c0 = lane; w.rstep = 1; w.cstep = SLAM_WG;
    }
    return w;
}

// Stages template rows [u0, u0 + ur) into LDS as {P0 = patch*mask, mask = 0xFF per kept pixel} entries of 8 bytes per
// 4-pixel group (G groups per row), resizing the template on the fly when the level needs it
// (feature.cpp:130-131), and adds the rows' |P0|^2 to sp_part.  `taps_region` (taps_bytes of LDS that nothing
// else uses until this returns) holds the tap tables of the resize.  Ends with the staged rows NOT yet visible
// to other lanes: the caller synchronises.
__device__ __forceinline__ void stage_template(const Geo& g, const Tmpl& t, lds_u64* tmpl, lds_u64* taps_region, int taps_bytes,
                                               int u0, int ur, int G, unsigned long long& sp_part STAMP_ARG) {
    const int lane = lane_here();
    {
    // ---- stage the template chunk: {P0, mask} per 4-pixel group -----------------------------
    const int ntap = G * 4 + ur;                       // column taps then row taps, 8 bytes each
    if (t.resized && ntap * 8 <= taps_bytes) {
        // cv::resize of the template on the fly (feature.cpp:130-131).  The tap tables of this
        // chunk go to LDS first (the tile region is still free), then every lane produces one
        // template pixel at a time: 4 source bytes of the patch, 4 of the mask, fixed-point blend.
        lds_u64* taps = taps_region;
        for (int i = lane; i < ntap; i += SLAM_WG) {
            int o, c0, c1;
            if (i < G * 4) resize_tap(min(i, g.pw - 1), t.scale_x, t.srcw, true, o, c0, c1);
            else resize_tap(u0 + (i - G * 4), t.scale_y, t.srch, false, o, c0, c1);
            taps[i] = ((unsigned long long)(((unsigned)c0 & 0xffffu) | ((unsigned)c1 << 16)) << 32) | (unsigned)o;
        }
        STAMP(46);
        __syncthreads();
        STAMP(44);
        lds_u8* tb = reinterpret_cast<lds_u8*>(tmpl);
        const int wpx = G * 4;
        const float rw = __builtin_amdgcn_rcpf((float)wpx);
        // four pixels per lane and pass: their 32 source bytes are all requested before the first blend, so a
        // pass costs one memory round trip (without a mask the patch stands in for it and the result is ignored)
        constexpr int NB = 4;
        const int n = ur * wpx;
        const uint8_t* msrc = t.mask ? t.mask : t.patch;
        const unsigned nomask = t.mask ? 0u : 1u;
        for (int i0 = lane; i0 < n; i0 += NB * SLAM_WG) {
            const int nq = min(NB, (n - (i0 - lane) + SLAM_WG - 1) >> 6);   // pixels per lane in this pass (wave-uniform)
            ResizeSrc sp[NB], sm[NB];
            unsigned long long tx[NB], ty[NB];
            int e[NB];
            bool pad[NB];
#pragma unroll
            for (int q = 0; q < NB; q++) {
                if (q < nq) {
                    const int i = min(i0 + q * SLAM_WG, n - 1);
                    const int u = (int)(((float)i + 0.5f) * rw), v = i - __mul24(u, wpx);
                    tx[q] = taps[min(v, g.pw - 1)]; ty[q] = taps[wpx + u];
                    pad[q] = v >= g.pw;                                  // padding columns of the last group
                    e[q] = (__mul24(u, G) + (v >> 2)) * 8 + (v & 3);     // byte of the {P0, mask} entry
                    sm[q] = resize_fetch(msrc, t.srcw, t.srch, (int)(unsigned)tx[q], (int)(unsigned)ty[q]);
                    sp[q] = resize_fetch(t.patch, t.srcw, t.srch, (int)(unsigned)tx[q], (int)(unsigned)ty[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < NB; q++) {
                if (q < nq) {
                    const int a0 = (short)(tx[q] >> 32), a1 = (short)(tx[q] >> 48), b0 = (short)(ty[q] >> 32), b1 = (short)(ty[q] >> 48);
                    unsigned m = (unsigned)(resize_blend(sm[q], a0, a1, b0, b1, true) != 0) | nomask;
                    unsigned p = (unsigned)resize_blend(sp[q], a0, a1, b0, b1, false) * m;
                    if (pad[q]) { p = 0; m = 0; }
                    if (i0 + q * SLAM_WG < n) {
                        sp_part += p * p;
                        tb[e[q]] = (unsigned char)p;
                        tb[e[q] + 4] = m ? 0xFF : 0x00;
                    }
                }
            }
        }
        __syncthreads();                               // the taps are dead before the tile is staged over them
    } else {
        const Walk2D w = make_walk(G, ur);
        for (int gg = w.c0; gg < G; gg += w.cstep) {
            const int nb = min(4, g.pw - gg * 4);      // pixels of this group inside the template
            const unsigned keep = nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u);
            if (!t.resized) {
                for (int u = w.r0; u < ur; u += w.rstep) {
                    const long long o = (long long)(u0 + u) * t.srcw + gg * 4;
                    const unsigned P = load4(t.patch + o, t.limit);
                    const unsigned Mraw = t.mask ? load4(t.mask + o, t.limit) : 0x01010101u;
                    // per byte: non-zero -> 0xFF
                    const unsigned nz = (((Mraw & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | Mraw) & 0x80808080u;
                    const unsigned M = ((nz >> 7) * 0xFFu) & keep;
                    const unsigned P0 = P & M;
                    sp_part += __builtin_amdgcn_udot4(P0, P0, 0u, false);
                    tmpl[u * G + gg] = ((unsigned long long)M << 32) | P0;
                }
            } else {
                int sx[4], a0[4], a1[4];               // the column taps do not depend on the row
#pragma unroll
                for (int b = 0; b < 4; b++) resize_tap(min(gg * 4 + b, g.pw - 1), t.scale_x, t.srcw, true, sx[b], a0[b], a1[b]);
                for (int u = w.r0; u < ur; u += w.rstep) {
                    int sy, b0, b1;
                    resize_tap(u0 + u, t.scale_y, t.srch, false, sy, b0, b1);
                    unsigned P0 = 0, M = 0;
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const unsigned m = t.mask ? (unsigned)(resize_px(t.mask, t.srcw, t.srch, sx[b], a0[b], a1[b], sy, b0, b1, true) != 0) : 1u;
                        const unsigned p = (unsigned)resize_px(t.patch, t.srcw, t.srch, sx[b], a0[b], a1[b], sy, b0, b1, false) * m;
                        P0 |= p << (8 * b);
                        M |= (m ? 0xFFu : 0u) << (8 * b);
                    }
                    P0 &= keep; M &= keep;
                    sp_part += __builtin_amdgcn_udot4(P0, P0, 0u, false);
                    tmpl[u * G + gg] = ((unsigned long long)M << 32) | P0;
                }
            }
        }
    }
    }
}

// One block of cells [xb, xb + 4*strips) x [yb, yb+bh) (xb may be < 0: columns left of 0 and right
// of aw are padding): sums into acc, then (update) scores into the running argmax / score map.
__device__ __forceinline__ void fast_block(const Geo& g, const Tmpl& t, float* scores, lds_u8* lds, const LdsPlan& lp,
                                           const BlockArgs& ba, int UR, int S, int G, bool big, bool first_block,
                                           bool update, unsigned long long& spfull, Norms& nm, Best& best STAMP_ARG) {
    const int lane = threadIdx.x;
    const int AW = ba.border ? 3 : 2;
    lds_u8* acc = lds;
    lds_u64* tmpl = reinterpret_cast<lds_u64*>(lds + lp.acc_bytes);
    lds_u32* tile = reinterpret_cast<lds_u32*>(lds + lp.acc_bytes + lp.tmpl_bytes);
    const int strips = ba.strips, bh = ba.bh, xb = ba.xb, yb = ba.yb;
    const int tp = (strips + G) | 1;                       // odd dword pitch: rows spread over the banks
    const int trows_cap = small_div_u((lp.tile_bytes >> 2) >> (ba.border ? 1 : 0), tp);
    lds_u32* vtile = tile + trows_cap * tp;
    const int units = strips * bh;
    const int ntask = units * S;
    const int ix0 = g.x0 + xb;                             // image column of tile column 0 (multiple of 4)
    const bool aligned = ((reinterpret_cast<unsigned long long>(g.im) | (unsigned long long)g.ipitch) & 3ull) == 0;
    unsigned long long sp_part = 0;

    __syncthreads();                                       // the previous block's readers of acc are done
    for (int i = lane; i < units * FAST_J * AW * (big ? 2 : 1); i += SLAM_WG) reinterpret_cast<lds_u32*>(acc)[i] = 0;

    for (int u0 = 0; u0 < g.ph; u0 += UR) {
        const int ur = min(UR, g.ph - u0);
        __syncthreads();                                   // previous chunk's readers are done
        STAMP(10);
        stage_template(g, t, tmpl, reinterpret_cast<lds_u64*>(tile), lp.tile_bytes, u0, ur, G, sp_part STAMP_PASS);
        STAMP(11);
        // ---- stage the image tile (and its validity) --------------------------------------------
        {
            const int trows = bh + ur - 1;
            const int iy0 = g.y0 + yb + u0;
            if (aligned && !ba.border && ix0 >= 0 && ix0 + 4 * tp <= g.w && iy0 >= 0 && iy0 + trows <= g.h) {
                // the whole tile lies inside the image: plain dword copies, four loads in flight per lane
                const uint8_t* base = g.im + (long long)iy0 * g.ipitch + ix0;
                const int n = trows * tp;
                const float rtp = __builtin_amdgcn_rcpf((float)tp);
                for (int i0 = lane; i0 < n; i0 += 4 * SLAM_WG) {
                    unsigned v[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int i = min(i0 + q * SLAM_WG, n - 1);
                        const int r = (int)(((float)i + 0.5f) * rtp), c = i - r * tp;
                        v[q] = *reinterpret_cast<const unsigned*>(base + (long long)r * g.ipitch + 4 * c);
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (i0 + q * SLAM_WG < n) tile[i0 + q * SLAM_WG] = v[q];
                }
            } else {
            const Walk2D w = make_walk(tp, trows);
            for (int r = w.r0; r < trows; r += w.rstep) {
                const int iy = iy0 + r;
                const bool rowin = iy >= 0 && iy < g.h;
                const uint8_t* srow = g.im + (long long)iy * g.ipitch;
                for (int c = w.c0; c < tp; c += w.cstep) {
                    const int ix = ix0 + 4 * c;
                    unsigned px = 0, vv = 0;
                    if (rowin && ix + 3 >= 0 && ix < g.w) {
                        const uint8_t* src = srow + ix;
                        if (ix >= 0 && ix + 3 < g.w) {
                            vv = 0xFFFFFFFFu;
                            if (aligned) px = *reinterpret_cast<const unsigned*>(src);
                            else px = (unsigned)src[0] | ((unsigned)src[1] << 8) | ((unsigned)src[2] << 16) | ((unsigned)src[3] << 24);
                        } else {
#pragma unroll
                            for (int b = 0; b < 4; b++)
                                if (ix + b >= 0 && ix + b < g.w) { px |= (unsigned)src[b] << (8 * b); vv |= 0xFFu << (8 * b); }
                        }
                    }
                    tile[r * tp + c] = px;
                    if (ba.border) vtile[r * tp + c] = vv;
                }
            }
            }
        }
        __syncthreads();
        STAMP(12);
        // ---- accumulate: the lanes walk the (strip, row-slice) tasks -----------------------------
#ifndef DIAG_SKIP_ACC
        for (int task = lane; task < ntask; task += SLAM_WG) {
            const int unit = small_div(task, S), slice = task - unit * S;
            const int row = small_div(unit, strips), strip = unit - row * strips;
            const int ub = small_div(slice * ur, S), ue = small_div((slice + 1) * ur, S);
            unsigned cc[4] = {0, 0, 0, 0}, si[4] = {0, 0, 0, 0}, sp[4] = {0, 0, 0, 0};
            if (ba.border) strip_accumulate<true, false, false>(tile, vtile, tp, tmpl, G, ub, ue, row, strip, cc, si, sp);
            else if (g.masked) strip_accumulate<false, false, false>(tile, vtile, tp, tmpl, G, ub, ue, row, strip, cc, si, sp);
            else if (G >= FAST_SLIDE_MIN_G) strip_accumulate<false, true, true>(tile, vtile, tp, tmpl, G, ub, ue, row, strip, cc, si, sp);
            else strip_accumulate<false, true, false>(tile, vtile, tp, tmpl, G, ub, ue, row, strip, cc, si, sp);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int a = (unit * FAST_J + k) * AW;
                if (big) {
                    lds_u64* a64 = reinterpret_cast<lds_u64*>(acc);
                    __hip_atomic_fetch_add(a64 + a, (unsigned long long)cc[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(a64 + a + 1, (unsigned long long)si[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (ba.border) __hip_atomic_fetch_add(a64 + a + 2, (unsigned long long)sp[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    lds_u32* a32 = reinterpret_cast<lds_u32*>(acc);
                    __hip_atomic_fetch_add(a32 + a, cc[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(a32 + a + 1, si[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (ba.border) __hip_atomic_fetch_add(a32 + a + 2, sp[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
#endif
    }
    STAMP(13);
    if (first_block) {                                     // |patch0|^2 over the whole template
        spfull = wave_sum_u64(sp_part);
        nm = make_norms(spfull);
    }
    __syncthreads();

    // ---- scores of the block's cells, running argmax ---------------------------------------------
#ifndef DIAG_SKIP_FIN
    if (update) {
        const int bx0 = max(xb, 0), bx1 = min(xb + strips * FAST_J, g.aw);      // the block's columns inside the area
        int nx, ny, cx0, cy0, step;
        // in a block without border cells every enumerated cell is decided without classify_cell:
        // all cells for stride 1 or no patch mask (matching.cpp:151-153), else the interior lattice
        const bool decided = !ba.border && (g.stride == 1 || !g.masked || !scores);
        if (!scores && g.stride > 1 && g.masked && !ba.border) {
            // the whole block lies in the interior: only the lattice from (xl0, yl0) is ever visited
            // (matching.cpp:64-73); the -1 of the other cells is accounted for by the caller
            step = g.stride;
            const int i0 = small_div_u(bx0 - g.xl0 + step - 1, step), i1 = small_div_u(bx1 - 1 - g.xl0, step);
            const int j0 = small_div_u(yb - g.yl0 + step - 1, step), j1 = small_div_u(yb + bh - 1 - g.yl0, step);
            nx = i1 - i0 + 1; ny = j1 - j0 + 1;
            cx0 = g.xl0 + i0 * step; cy0 = g.yl0 + j0 * step;
        } else {
            step = 1; nx = bx1 - bx0; ny = bh; cx0 = bx0; cy0 = yb;
        }
        if (nx > 0 && ny > 0) {
            const float rnx = __builtin_amdgcn_rcpf((float)nx);
            // Only the maximum is wanted here, and the exact formulas (double square root, IEEE division) cost far
            // more than the sums' two loads: a first pass estimates every score in single precision (relative
            // error below 1e-6), and only cells whose estimate comes within 1e-5 of the largest one - the only
            // ones that can hold the maximum or tie with it - are scored exactly.
            const bool prune = decided && !scores && !big && nx * ny > SLAM_WG && nm.tn > 0.0;
            const float rn = __builtin_amdgcn_rcpf(nm.p2f);
            float keep_from = -FLT_MAX;
            if (prune) {
                float m = -FLT_MAX;
                for (int i = lane; i < nx * ny; i += SLAM_WG) {
                    const int r = (int)(((float)i + 0.5f) * rnx), c = i - r * nx;
                    m = fmaxf(m, cell_estimate(g, acc, ba, cx0 + c * step, cy0 + r * step, rn));
                }
#pragma unroll
                for (int d = 1; d < SLAM_WG; d <<= 1) m = fmaxf(m, __shfl_xor(m, d));
                keep_from = m - 1e-5f * fmaxf(m, 1.f);
            }
            for (int i = lane; i < nx * ny; i += SLAM_WG) {
                const int r = (int)(((float)i + 0.5f) * rnx), c = i - r * nx;
                const int x = cx0 + c * step, y = cy0 + r * step;
                if (prune && !(cell_estimate(g, acc, ba, x, y, rn) >= keep_from)) continue;
                const float sc = cell_from_acc(g, acc, ba, big, x, y, nm, decided);
                const int idx = y * g.aw + x;
                if (scores) scores[idx] = sc;
                best_update(best, sc, idx);
            }
        }
    }
#endif
    STAMP(14);
}

// First cell (row-major) the reference's loops never visit, as an argmax candidate with the score
// map's initial value -1 (matching.cpp:53); idx == INT_MAX when every cell is visited.  Cold path.
__device__ __noinline__ Best first_unvisited(Geo g) {
    Best b = {-FLT_MAX, INT_MAX};
    const int n = g.aw * g.ah;
    for (int base = 0; base < n && b.idx == INT_MAX; base += SLAM_WG) {
        const int i = base + (int)threadIdx.x;
        Best c = {-FLT_MAX, INT_MAX};
        if (i < n) {
            const int y = i / g.aw, x = i - y * g.aw;
            if (classify_cell(g, x, y) == CELL_NONE) { c.v = -1.f; c.idx = i; }
        }
        b = best_wave(c);
    }
    return b;
}

// True when the tiled core can run this job with the LDS it was given: the template must be at most
// 4*FAST_MAX_G wide and three tile rows (the peak's neighbourhood) must fit next to one template row
// in the cropped (two-tile) case.  Otherwise the caller uses direct evaluation.
__device__ __forceinline__ bool fast_supported(const Geo& g, const LdsPlan& lp) {
    const int G = (g.pw + 3) >> 2;
    return G <= FAST_MAX_G && lp.tile_bytes / 8 >= 3 * ((16 + G) | 1) && lp.tmpl_bytes / 8 >= G;
}

// Row slices per strip: the S in [1, 8] (S <= rows) that minimises passes x rows-per-slice when
// units x S tasks are dealt to the 64 lanes.
__device__ __forceinline__ int pick_slices(int units, int rows) {
    int best_s = 1, best_c = 1 << 30;
#pragma unroll
    for (int s = 1; s <= 8; s++) {
        if (s > rows) break;
        const int passes = (units * s + SLAM_WG - 1) >> 6;
        const int c = passes * ((rows + s - 1) / s + 1);   // +1: the per-task flush
        if (c < best_c) { best_c = c; best_s = s; }
    }
    return best_s;
}

// What is left in the accumulators after fast_search: the block that holds the peak's neighbours.
struct FastState {
    bool valid;
    bool big;
    BlockArgs ba;
};

// Searches the whole score map block by block.  On return `best` is the wave-uniform argmax and,
// when a refinement was asked for (or the map was a single block), fs describes a block in the accumulators that
// contains the peak and its four neighbours (for the sub-pixel fit).
__device__ __forceinline__ void fast_search(const Geo& g, const Tmpl& t, float* scores, lds_u8* lds, const LdsPlan& lp,
                                            bool refine_always, double refine_below, unsigned long long& spfull,
                                            Norms& nm, Best& best, FastState& fs STAMP_ARG) {
    const int G = (g.pw + 3) >> 2;
    const bool big = (long long)g.pw * g.ph > 66051;       // 66051 * 255^2 < 2^32: beyond it sums need 64 bits
    const int o = g.x0 & 3;                                // cells left of 0 that align the tile to image dwords
    const int BW = min((g.aw + o + 3) & ~3, 64);
    const int strips_max = BW >> 2;
    // does any cell's patch leave the image?  Then blocks may need the validity tile and a third sum.
    const bool any_border = g.xl0 > 0 || g.yl0 > 0 || g.xl1 < g.aw || g.yl1 < g.ah;
    const int tp_max = (strips_max + G) | 1;
    int TR = max(1, small_div_u((lp.tile_bytes >> 2) >> (any_border ? 1 : 0), tp_max));
    const int cap_rows = max(1, small_div_u(small_div_u(lp.acc_bytes >> (big ? 3 : 2), any_border ? 3 : 2), BW));
    int UR = max(1, min(min(g.ph, small_div_u(lp.tmpl_bytes / 8, G)), TR));
    int BH = min(g.ah, cap_rows);
    if (BH + UR - 1 > TR) {
        // prefer whole-template chunks: shrink the block first, then the chunk
        if (TR - UR + 1 >= 4 || UR == 1) BH = max(1, min(BH, TR - UR + 1));
        else { UR = max(1, TR / 2); BH = max(1, min(BH, TR - UR + 1)); }
    }
    // A window that needs several row blocks under the fixed split gets the job's LDS re-split to its own shapes:
    // the template chunk keeps what it uses, and accumulators and tile share the rest so that a block is as tall
    // as both allow (small templates leave most of the fixed template and tile shares idle).
    LdsPlan lq = lp;
    if (BH < g.ah) {
        const int total = lp.acc_bytes + lp.tmpl_bytes + lp.tile_bytes;
        const int A = BW * (big ? 8 : 4) * (any_border ? 3 : 2);          // accumulator bytes per block row
        const int T = tp_max * 4 * (any_border ? 2 : 1);                  // tile (and validity) bytes per tile row
        const int tb = (UR * G * 8 + 15) & ~15;
        const int rest = total - tb - 32;
        const int bh2 = small_div_u(max(rest - (UR - 1) * T, 0), A + T);
        if (bh2 > BH) {
            BH = min(g.ah, bh2);
            lq.tmpl_bytes = tb;
            lq.acc_bytes = (BH * A + 15) & ~15;
            lq.tile_bytes = (total - lq.tmpl_bytes - lq.acc_bytes) & ~15;
            TR = max(1, small_div_u((lq.tile_bytes >> 2) >> (any_border ? 1 : 0), tp_max));
        }
    }
    if (BH < g.ah) {                                       // several row blocks: equal heights instead of a thin last one
        const int nblk = small_div_u(g.ah + BH - 1, BH);
        BH = small_div_u(g.ah + nblk - 1, nblk);
    }
    fs.valid = false; fs.big = big;
    STAMP(4);
    bool first = true, refine = false;
    int nblocks = 0, yb = 0, xb = -o;
    // The split travels through the block loop as one scalar (both sizes are multiples of 16 below 64 KB); it is
    // unpacked per block, which keeps three wave-uniform values out of the registers that live across a block.
    unsigned split = (unsigned)(lq.acc_bytes >> 4) | ((unsigned)(lq.tmpl_bytes >> 4) << 16);
    while (true) {
        asm volatile("" : "+v"(split));
        split = __builtin_amdgcn_readfirstlane(split);
        LdsPlan lb;
        lb.acc_bytes = (int)(split & 0xffffu) << 4;
        lb.tmpl_bytes = (int)(split >> 16) << 4;
        lb.tile_bytes = (lp.acc_bytes + lp.tmpl_bytes + lp.tile_bytes - lb.acc_bytes - lb.tmpl_bytes) & ~15;
        BlockArgs ba;
        if (!refine) {
            if (yb >= g.ah) {                              // scan finished: the argmax is known
                best = best_wave(best);
                best.idx = __builtin_amdgcn_readfirstlane(best.idx);
                best.v = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(best.v)));
                // Visited cells score >= 0 or NaN.  If none scored, the maximum is the -1 of the first
                // cell the reference never visits; lattice-only blocks skipped those, so look it up.
                if (!scores && g.stride > 1 && g.masked && !(best.v >= 0.f)) best = first_unvisited(g);
                if (nblocks == 1) { fs.valid = true; break; }
                if (best.idx == INT_MAX || !(refine_always || (double)best.v < refine_below)) break;
                // multi-block map: recompute the 3x3 cells around the peak into the accumulators
                const int ly = row_of_idx(g, best.idx), lx = best.idx - ly * g.aw;
                refine = true;
                ba.xb = ((max(lx - 1, 0) + o) & ~3) - o;
                ba.strips = ((min(lx + 1, g.aw - 1) - ba.xb) >> 2) + 1;
                ba.yb = max(ly - 1, 0);
                ba.bh = min(ly + 1, g.ah - 1) - ba.yb + 1;
            } else {
                ba.xb = xb; ba.yb = yb;
                ba.strips = (min(BW, g.aw - xb) + 3) >> 2;
                ba.bh = min(BH, g.ah - yb);
                xb += BW;
                if (xb >= g.aw) { xb = -o; yb += BH; }
            }
        }
        ba.border = max(ba.xb, 0) < g.xl0 || min(ba.xb + ba.strips * FAST_J, g.aw) > g.xl1 || ba.yb < g.yl0 || ba.yb + ba.bh > g.yl1;
        const int URb = refine ? max(1, min(UR, TR - ba.bh + 1)) : UR;     // the 3-row refine block must fit the tile
        const int S = pick_slices(ba.strips * ba.bh, URb);
        STAMP(5);
        fast_block(g, t, refine ? nullptr : scores, lds, lb, ba, URb, S, G, big, first, !refine, spfull, nm, best STAMP_PASS);
        first = false;
        nblocks++;
        fs.ba = ba;
        if (refine) { fs.valid = true; break; }
    }
}

// ---- small levels: one lane group per cell ------------------------------------------------------
// The coarse levels of a walk score a handful of cells with a small template; the block machinery above
// (strips, row slices, LDS accumulators, block planning) costs far more than their arithmetic.  When every
// cell's patch lies inside the image, the level visits at most 64 cells and template, tile and tap tables fit
// the job's LDS, the level is searched here instead: template and image tile are staged as above, each visited
// cell gets P = 2^k lanes that split the template rows, partial sums meet by butterfly shuffles, and every
// lane of a cell evaluates the cell's score.  Same integer sums, same formulas, same argmax rule.
struct MiniPlan { int nx, ny, step, parts, G, tp, trows, o; };

__device__ __forceinline__ bool mini_supported(const Geo& g, const Tmpl& t, int lds_total, MiniPlan& mp) {
    const int G = (g.pw + 3) >> 2;
    const bool interior = g.xl0 == 0 && g.yl0 == 0 && g.xl1 == g.aw && g.yl1 == g.ah;
    if (!interior || g.aw > 64 || g.ah > 64 || G > 16 || g.ph > 64) return false;
    const int step = g.masked ? g.stride : 1;          // without a patch mask every interior cell is visited (matching.cpp:151-153)
    const int nx = step == 1 ? g.aw : small_div_u(g.aw + step - 1, step);
    const int ny = step == 1 ? g.ah : small_div_u(g.ah + step - 1, step);
    const int ncell = nx * ny;
    if (ncell > 64) return false;
    const bool aligned = ((reinterpret_cast<unsigned long long>(g.im) | (unsigned long long)g.ipitch) & 3ull) == 0;
    const int o = g.x0 & 3;
    const int tp = ((o + g.aw + 4 * G + 3) >> 2) | 1;  // dwords per tile row: every cell's G dwords plus the one after them
    const int trows = g.ah + g.ph - 1;
    const int tile_bytes = trows * tp * 4, tmpl_bytes = g.ph * G * 8, tap_bytes = t.resized ? (G * 4 + g.ph) * 8 : 0;
    if (!aligned || (g.x0 - o) + 4 * tp > g.w) return false;   // the tile's dword rows must lie inside the image rows
    if (tmpl_bytes + max(tile_bytes, tap_bytes) > lds_total) return false;
    mp.nx = nx; mp.ny = ny; mp.step = step; mp.G = G; mp.tp = tp; mp.trows = trows; mp.o = o;
    mp.parts = ncell <= 4 ? 16 : ncell <= 8 ? 8 : ncell <= 16 ? 4 : ncell <= 32 ? 2 : 1;
    return true;
}

// parabola through the peak s and its two neighbours on one axis (not in the reference); 0 when a neighbour
// is missing, unvisited (-1) or NaN, or the three values are not strictly concave
__device__ __forceinline__ float subpixel_axis(float s, float a, float b, bool oka, bool okb) {
    if (!(oka && okb && a != -1.f && b != -1.f && a == a && b == b)) return 0.f;
    const float d2 = __fsub_rn(__fadd_rn(a, b), __fmul_rn(2.0f, s));
    if (!(d2 < 0.f)) return 0.f;
    const float q = __fdiv_rn(__fsub_rn(a, b), __fmul_rn(2.0f, d2));
    return q < -0.5f ? -0.5f : (q > 0.5f ? 0.5f : q);
}

__device__ __forceinline__ CoreOut mini_level(const Geo& g, const Tmpl& t, lds_u8* lds, const MiniPlan& mp,
                                              bool refine_always, double refine_below STAMP_ARG) {
    const int lane = lane_here();
    const int G = mp.G, tp = mp.tp;
    lds_u64* tmpl = reinterpret_cast<lds_u64*>(lds);
    lds_u32* tile = reinterpret_cast<lds_u32*>(lds + g.ph * G * 8);
    unsigned long long sp_part = 0;
    STAMP(45);
    __syncthreads();                                   // earlier readers of this LDS are done
    stage_template(g, t, tmpl, reinterpret_cast<lds_u64*>(tile), t.resized ? (G * 4 + g.ph) * 8 : 0, 0, g.ph, G, sp_part STAMP_PASS);
    STAMP(40);
    {   // image tile: rows y0 .. y0 + ah + ph - 2, dword columns from (x0 & ~3); all inside the image
        const uint8_t* base = g.im + (long long)g.y0 * g.ipitch + (g.x0 - mp.o);
        const int n = mp.trows * tp;
        const float rtp = __builtin_amdgcn_rcpf((float)tp);
        for (int i0 = lane; i0 < n; i0 += 4 * SLAM_WG) {
            const int nq = min(4, (n - (i0 - lane) + SLAM_WG - 1) >> 6);    // dwords per lane in this pass (wave-uniform)
            unsigned v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (q < nq) {
                    const int i = min(i0 + q * SLAM_WG, n - 1);
                    const int r = (int)(((float)i + 0.5f) * rtp), c = i - r * tp;
                    v[q] = *reinterpret_cast<const unsigned*>(base + (long long)r * g.ipitch + 4 * c);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (q < nq && i0 + q * SLAM_WG < n) tile[i0 + q * SLAM_WG] = v[q];
        }
    }
    // at most 66051 template pixels: the whole sum of squares fits 32 bits
    const unsigned long long spfull = wave_sum_u32((unsigned)sp_part);
    const Norms nm = make_norms(spfull);
    __syncthreads();
    STAMP(41);

    // lane -> (cell, part)
    const int P = mp.parts, ncell = mp.nx * mp.ny;
    const int pshift = P == 16 ? 4 : P == 8 ? 3 : P == 4 ? 2 : P == 2 ? 1 : 0;
    const int cell = lane >> pshift, part = lane & (P - 1);
    const bool live = cell < ncell;
    const int cj = live ? small_div(cell, mp.nx) : 0, ci = live ? cell - cj * mp.nx : 0;
    const int x = ci * mp.step, y = cj * mp.step;
    unsigned cc = 0, si = 0;
    if (live) {
        const int bx = mp.o + x;
        const int k = bx & 3;
        for (int u = part; u < g.ph; u += P) {
            const lds_u32* tr = tile + (y + u) * tp + (bx >> 2);
            const lds_u64* pr = tmpl + u * G;
            unsigned d0 = tr[0];
            for (int gq = 0; gq < G; ++gq) {
                const unsigned dn = tr[gq + 1];
                const unsigned long long pm = pr[gq];
                const unsigned P0 = (unsigned)pm, M = (unsigned)(pm >> 32);
                const unsigned w = __builtin_amdgcn_alignbyte(dn, d0, (unsigned)k);
                const unsigned wm = w & M;
                cc = __builtin_amdgcn_udot4(w, P0, cc, false);
                si = __builtin_amdgcn_udot4(wm, wm, si, false);
                d0 = dn;
            }
        }
    }
    for (int m = 1; m < P; m <<= 1) {                  // butterfly over the parts of a cell: every part ends with the sums
        cc += (unsigned)__shfl_xor((int)cc, m, 64);
        si += (unsigned)__shfl_xor((int)si, m, 64);
    }
    STAMP(42);
    float sc = -1.f;                                   // matching.cpp:53
    if (live) {
        const bool am = !g.amask || g.amask[(long long)y * g.ampitch + x] != 0;
        if (g.masked) { if (am) sc = score_f32_n(cc, si, nm.p2f); }
        else if (!(g.exact && !am)) sc = score_mt_n(cc, si, nm.tn);
    }
    Best b = {-FLT_MAX, INT_MAX};
    if (live && part == 0) best_update(b, sc, y * g.aw + x);
    b = best_wave(b);
    b.idx = __builtin_amdgcn_readfirstlane(b.idx);
    b.v = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(b.v)));
    // Visited cells score >= 0 or NaN.  If none scored, the maximum is the -1 of the first cell the
    // reference never visits (off the lattice): look it up.
    if (g.stride > 1 && g.masked && !(b.v >= 0.f)) b = first_unvisited(g);
    STAMP(43);
    CoreOut o;
    o.dx = 0.f; o.dy = 0.f;
    if (b.idx == INT_MAX) {                            // nothing compared greater than -FLT_MAX: every cell is NaN
        o.x = -1 + g.ax; o.y = -1 + g.ay; o.response = 0.f;
        return o;
    }
    const int ly = row_of_idx(g, b.idx), lx = b.idx - ly * g.aw;
    o.x = lx + g.ax; o.y = ly + g.ay; o.response = b.v;
    if (!(refine_always || (double)b.v < refine_below)) return o;
    // the four neighbours of the peak: a neighbour on the lattice sits in another lane group, one off the
    // lattice was never visited (-1)
    float nb[4]; bool in[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int qx = lx + (q == 0 ? -1 : q == 1 ? 1 : 0), qy = ly + (q == 2 ? -1 : q == 3 ? 1 : 0);
        in[q] = qx >= 0 && qy >= 0 && qx < g.aw && qy < g.ah;
        float v = -1.f;
        if (in[q]) {
            int src = -1;
            if (mp.step == 1) src = (qy * mp.nx + qx) << pshift;
            else if (mod_stride(g, qx) == 0 && mod_stride(g, qy) == 0) src = (small_div_u(qy, mp.step) * mp.nx + small_div_u(qx, mp.step)) << pshift;
            const float got = __shfl(sc, src < 0 ? 0 : src, 64);
            if (src >= 0) v = got;
        }
        nb[q] = v;
    }
    o.dx = subpixel_axis(b.v, nb[0], nb[1], in[0], in[1]);
    o.dy = subpixel_axis(b.v, nb[2], nb[3], in[2], in[3]);
    return o;
}

}  // namespace slamdev
