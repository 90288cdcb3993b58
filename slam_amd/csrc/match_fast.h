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
// Work split: the block's cells are cut into strips of J (4 or 8) consecutive cells of one row; a
// task is one strip x one slice of the chunk's template rows, and the 64 lanes walk the task list.
// Per 4-pixel group a lane forms, for each of its J cells, the cell's 4 image pixels with one
// v_perm_b32 (byte alignment and patch mask folded into one selector) and feeds two v_dot4_u32_u8
// (CC += <Im,P0>, SI += <Im,Im>); blocks with cropped cells add SPcrop += <P0 & valid, P0>.
// Every sum is an exact integer, so the scores are bit-identical to a scalar evaluation.
#pragma once

#include "match_common.h"

#define FAST_LDS_DEFAULT (16 * 1024)
#define FAST_LDS_MIN (4 * 1024)
#define FAST_LDS_MAX (64 * 1024)
#define FAST_MAX_G 96         // template width up to 384 pixels; wider ones use direct evaluation

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

// What the tiled core leaves behind for the sub-pixel fit when the whole score map was one block.
struct FastState {
    bool single_block;
    bool border, big;
    int J, xb, yb, strips, bh;
};

__device__ __forceinline__ unsigned bfi32(unsigned mask, unsigned a, unsigned b) { return (a & mask) | (b & ~mask); }

// Accumulate template rows [ub, ue) (chunk-relative) for the J cells of one strip.  Tile row r
// holds image row (first image row of the tile) + r and the cell row `row` is relative to the
// block, so template row u of that cell lies in tile row row + u.
template <int J, bool BORDER>
__device__ __forceinline__ void strip_accumulate(const lds_u32* tile, const lds_u32* vtile, int tp,
                                                 const lds_u64* tmpl, int G, int ub, int ue, int row, int strip,
                                                 unsigned (&cc)[J], unsigned (&si)[J], unsigned (&sp)[J]) {
    constexpr int JD = J / 4;
    for (int u = ub; u < ue; ++u) {
        const int tofs = (row + u) * tp + strip * JD;
        const lds_u32* tr = tile + tofs;
        const lds_u32* vr = vtile + tofs;
        const lds_u64* pr = tmpl + u * G;
        unsigned d0 = tr[0], d1 = (JD == 2) ? tr[1] : 0u;
        unsigned v0 = 0, v1 = 0;
        if (BORDER) { v0 = vr[0]; v1 = (JD == 2) ? vr[1] : 0u; }
#pragma unroll 2
        for (int g = 0; g < G; ++g) {
            const unsigned dn = tr[g + JD];
            const unsigned long long pm = pr[g];
            const unsigned P0 = (unsigned)pm, M = (unsigned)(pm >> 32);
            unsigned sel[4];
#pragma unroll
            for (int k = 0; k < 4; k++) sel[k] = bfi32(M, 0x03020100u + 0x01010101u * (unsigned)k, 0x0C0C0C0Cu);
            unsigned vn = 0;
            if (BORDER) vn = vr[g + JD];
#pragma unroll
            for (int k = 0; k < J; k++) {
                const unsigned hi = (JD == 1) ? dn : (k < 4 ? d1 : dn);
                const unsigned lo = (JD == 1) ? d0 : (k < 4 ? d0 : d1);
                const unsigned w = __builtin_amdgcn_perm(hi, lo, sel[k & 3]);
                cc[k] = __builtin_amdgcn_udot4(w, P0, cc[k], false);
                si[k] = __builtin_amdgcn_udot4(w, w, si[k], false);
                if (BORDER) {
                    const unsigned vhi = (JD == 1) ? vn : (k < 4 ? v1 : vn);
                    const unsigned vlo = (JD == 1) ? v0 : (k < 4 ? v0 : v1);
                    const unsigned vm = __builtin_amdgcn_perm(vhi, vlo, sel[k & 3]);
                    sp[k] = __builtin_amdgcn_udot4(P0 & vm, P0, sp[k], false);
                }
            }
            if (JD == 1) { d0 = dn; v0 = vn; } else { d0 = d1; d1 = dn; v0 = v1; v1 = vn; }
        }
    }
}

__device__ __forceinline__ float finalize_cell(const Geo& g, int x, int y, unsigned long long CC,
                                               unsigned long long SI, unsigned long long SPc, const Norms& nm) {
    const int kind = classify_cell(g, x, y);
    if (kind == CELL_NONE) return -1.f;                    // matching.cpp:53
    if (kind == CELL_MT) return score_mt_n(CC, SI, nm.tn);
    if (kind == CELL_F32_FULL) return score_f32_n(CC, SI, nm.p2f);
    return score_f32(CC, SI, SPc);
}

template <bool BIG>
__device__ __forceinline__ unsigned long long acc_read(const lds_u8* acc, int i) {
    if (BIG) return reinterpret_cast<const lds_u64*>(acc)[i];
    return reinterpret_cast<const lds_u32*>(acc)[i];
}

// Score of block cell (x,y) from the block's accumulators (x,y must be cells of that block).
template <int J, bool BORDER, bool BIG>
__device__ __forceinline__ float cell_from_acc(const Geo& g, const lds_u8* acc, int xb, int yb, int strips,
                                               int x, int y, const Norms& nm) {
    constexpr int AW = BORDER ? 3 : 2;
    const int c = ((y - yb) * strips) * J + (x - xb);      // (row*strips + strip)*J + k
    const unsigned long long CC = acc_read<BIG>(acc, c * AW + 0);
    const unsigned long long SI = acc_read<BIG>(acc, c * AW + 1);
    const unsigned long long SP = BORDER ? acc_read<BIG>(acc, c * AW + 2) : 0ull;
    return finalize_cell(g, x, y, CC, SI, SP, nm);
}

// Lane -> (row, column) walk over a rows x width rectangle without per-element divisions:
// when width <= 64 a lane keeps its column and steps rows by 64/width, else rows are done one at a
// time with the lanes striding over the columns.
struct Walk2D {
    int r0, c0, rstep, cstep;
};

__device__ __forceinline__ Walk2D make_walk(int width, int rows) {
    Walk2D w;
    const int lane = threadIdx.x;
    if (width <= SLAM_WG) {
        const int per = SLAM_WG / width;                   // rows handled per pass
        const int lr = lane / width;
        w.c0 = lane - lr * width;
        w.r0 = lr < per ? lr : rows;                       // surplus lanes do nothing
        w.rstep = per;
        w.cstep = width;
    } else {
        w.r0 = 0; w.c0 = lane; w.rstep = 1; w.cstep = SLAM_WG;
    }
    return w;
}

// One block of cells [xb, xb + strips*J) x [yb, yb+bh) (xb may be < 0: columns left of 0 and right
// of aw are padding).  Adds the block's scores to the running argmax.
template <int J, bool BORDER, bool BIG>
__device__ __forceinline__ void fast_block(const Geo& g, const Tmpl& t, float* scores, lds_u8* lds, const LdsPlan& lp,
                                           int xb, int yb, int strips, int bh, int UR, int S, int G,
                                           bool first_block, unsigned long long& spfull, Norms& nm, Best& best STAMP_ARG) {
    const int lane = threadIdx.x;
    constexpr int JD = J / 4;
    constexpr int AW = BORDER ? 3 : 2;
    lds_u8* acc = lds;
    lds_u64* tmpl = reinterpret_cast<lds_u64*>(lds + lp.acc_bytes);
    lds_u32* tile = reinterpret_cast<lds_u32*>(lds + lp.acc_bytes + lp.tmpl_bytes);
    const int tp = (strips * JD + G) | 1;                  // odd dword pitch: rows spread over the banks
    const int trows_max = (lp.tile_bytes / 4 / (BORDER ? 2 : 1)) / tp;
    lds_u32* vtile = tile + trows_max * tp;
    const int width = strips * J;                          // cells per block row, <= 64
    const int units = strips * bh;
    const int ntask = units * S;
    const int ncell = units * J;
    const int ix0 = g.x0 + xb;                             // image column of tile column 0 (multiple of 4)
    const bool aligned = ((reinterpret_cast<unsigned long long>(g.im) | (unsigned long long)g.ipitch) & 3ull) == 0;
    const unsigned rcpS = rcp_u32((unsigned)S), rcpStrips = rcp_u32((unsigned)strips);
    unsigned long long sp_part = 0;

    __syncthreads();                                       // the previous block's finalize has read acc
    for (int i = lane; i < ncell * AW * (BIG ? 2 : 1); i += SLAM_WG) reinterpret_cast<lds_u32*>(acc)[i] = 0;

    for (int u0 = 0; u0 < g.ph; u0 += UR) {
        const int ur = min(UR, g.ph - u0);
        __syncthreads();                                   // previous chunk's readers are done
        STAMP(10);
        // ---- stage the template chunk: {P0, mask} per 4-pixel group -----------------------------
        {
            const Walk2D w = make_walk(G, ur);
            for (int u = w.r0; u < ur; u += w.rstep)
                for (int gg = w.c0; gg < G; gg += w.cstep) {
                    unsigned P0 = 0, M = 0;
                    if (!t.resized) {
                        const long long o = (long long)(u0 + u) * t.srcw + gg * 4;
#pragma unroll
                        for (int b = 0; b < 4; b++)
                            if (gg * 4 + b < g.pw) {
                                unsigned p = t.patch[o + b];
                                const unsigned m = t.mask ? (unsigned)(t.mask[o + b] != 0) : 1u;
                                p *= m;
                                P0 |= p << (8 * b);
                                M |= (m ? 0xFFu : 0u) << (8 * b);
                                if (first_block) sp_part += p * p;
                            }
                    } else {
                        int sy, b0, b1;
                        resize_tap(u0 + u, t.scale_y, t.srch, false, sy, b0, b1);
#pragma unroll
                        for (int b = 0; b < 4; b++)
                            if (gg * 4 + b < g.pw) {
                                int sx, a0, a1;
                                resize_tap(gg * 4 + b, t.scale_x, t.srcw, true, sx, a0, a1);
                                unsigned p = (unsigned)resize_px(t.patch, t.srcw, t.srch, sx, a0, a1, sy, b0, b1, false);
                                const unsigned m = t.mask ? (unsigned)(resize_px(t.mask, t.srcw, t.srch, sx, a0, a1, sy, b0, b1, true) != 0) : 1u;
                                p *= m;
                                P0 |= p << (8 * b);
                                M |= (m ? 0xFFu : 0u) << (8 * b);
                                if (first_block) sp_part += p * p;
                            }
                    }
                    tmpl[u * G + gg] = ((unsigned long long)M << 32) | P0;
                }
        }
        STAMP(11);
        // ---- stage the image tile (and its validity) --------------------------------------------
        {
            const int trows = bh + ur - 1;
            const int iy0 = g.y0 + yb + u0;
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
                    if (BORDER) vtile[r * tp + c] = vv;
                }
            }
        }
        __syncthreads();
        STAMP(12);
        // ---- accumulate: the lanes walk the (strip, row-slice) tasks -----------------------------
        for (int task = lane; task < ntask; task += SLAM_WG) {
            const int unit = (int)div_rcp((unsigned)task, (unsigned)S, rcpS), slice = task - unit * S;
            const int row = (int)div_rcp((unsigned)unit, (unsigned)strips, rcpStrips), strip = unit - row * strips;
            const int ub = (slice * ur) / S, ue = ((slice + 1) * ur) / S;
            unsigned cc[J], si[J], sp[J];
#pragma unroll
            for (int k = 0; k < J; k++) { cc[k] = 0; si[k] = 0; sp[k] = 0; }
            strip_accumulate<J, BORDER>(tile, vtile, tp, tmpl, G, ub, ue, row, strip, cc, si, sp);
#pragma unroll
            for (int k = 0; k < J; k++) {
                const int a = (unit * J + k) * AW;
                if (BIG) {
                    lds_u64* a64 = reinterpret_cast<lds_u64*>(acc);
                    __hip_atomic_fetch_add(a64 + a, (unsigned long long)cc[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(a64 + a + 1, (unsigned long long)si[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (BORDER) __hip_atomic_fetch_add(a64 + a + 2, (unsigned long long)sp[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    lds_u32* a32 = reinterpret_cast<lds_u32*>(acc);
                    __hip_atomic_fetch_add(a32 + a, cc[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(a32 + a + 1, si[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (BORDER) __hip_atomic_fetch_add(a32 + a + 2, sp[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
    STAMP(13);
    if (first_block) {                                     // |patch0|^2 over the whole template
        spfull = wave_sum_u64(sp_part);
        nm = make_norms(spfull);
    }
    __syncthreads();

    // ---- finalize: scores of the block's cells, running argmax -----------------------------------
    {
        const Walk2D w = make_walk(width, bh);
        const int x = xb + w.c0;
        if (x >= 0 && x < g.aw)
            for (int r = w.r0; r < bh; r += w.rstep) {
                const int y = yb + r;
                const float sc = cell_from_acc<J, BORDER, BIG>(g, acc, xb, yb, strips, x, y, nm);
                const int idx = y * g.aw + x;
                if (scores) scores[idx] = sc;
                best_update(best, sc, idx);
            }
    }
    STAMP(14);
}

// True when the tiled core can run this job; otherwise the caller uses direct evaluation.
__device__ __forceinline__ bool fast_supported(const Geo& g) {
    return (g.pw + 3) / 4 <= FAST_MAX_G;
}

// Row slices per strip: the S in [1, smax] that best fills the 64 lanes with strip x slice tasks.
__device__ __forceinline__ int pick_slices(int units, int smax) {
    int best_s = 1, best_u = 0;
    for (int s = 1; s <= smax; s++) {
        const int n = units * s;
        const int u = n * 1000 / ((n + SLAM_WG - 1) / SLAM_WG * SLAM_WG);   // lane utilisation, per mille
        if (u > best_u + 30) { best_u = u; best_s = s; }
    }
    return best_s;
}

template <int J>
__device__ __forceinline__ void fast_blocks(const Geo& g, const Tmpl& t, float* scores, lds_u8* lds, const LdsPlan& lp,
                                            unsigned long long& spfull, Norms& nm, Best& best, FastState& fs STAMP_ARG) {
    constexpr int JD = J / 4;
    const int G = (g.pw + 3) / 4;
    const bool big = (long long)g.pw * g.ph > 66051;       // 66051 * 255^2 < 2^32: beyond it sums need 64 bits
    const int o = ((g.x0 % 4) + 4) & 3;                    // cells left of 0 that align the tile to image dwords
    const int BW = min((g.aw + o + J - 1) / J * J, 64);
    const int strips_max = BW / J;
    // does any cell's patch leave the image?  Then blocks may need the validity tile and a third sum.
    const bool any_border = g.xl0 > 0 || g.yl0 > 0 || g.xl1 < g.aw || g.yl1 < g.ah;
    // one plan per band of rows, sized for the widest block
    const int tp_max = (strips_max * JD + G) | 1;
    const int TR = max(1, (lp.tile_bytes / 4 / (any_border ? 2 : 1)) / tp_max);
    const int cap_cells = lp.acc_bytes / ((any_border ? 3 : 2) * (big ? 8 : 4));
    bool first = true;
    int nblocks = 0;
    for (int yb = 0; yb < g.ah;) {
        int UR = max(1, min(min(g.ph, lp.tmpl_bytes / 8 / G), TR));
        int BH = max(1, min(g.ah - yb, cap_cells / (strips_max * J)));
        if (BH + UR - 1 > TR) {
            // prefer whole-template chunks: shrink the block first, then the chunk
            if (TR - UR + 1 >= 4 || UR == 1) BH = max(1, min(BH, TR - UR + 1));
            else { UR = max(1, TR / 2); BH = max(1, min(BH, TR - UR + 1)); }
        }
        for (int xb = -o; xb < g.aw; xb += BW) {
            const int bw = min(BW, g.aw - xb);
            const int strips = (bw + J - 1) / J;
            const int cx0 = max(xb, 0), cx1 = xb + bw;
            const bool border = cx0 < g.xl0 || cx1 > g.xl1 || yb < g.yl0 || yb + BH > g.yl1;
            const int S = pick_slices(strips * BH, min(8, UR));
            if (border) {
                if (big) fast_block<J, true, true>(g, t, scores, lds, lp, xb, yb, strips, BH, UR, S, G, first, spfull, nm, best STAMP_PASS);
                else fast_block<J, true, false>(g, t, scores, lds, lp, xb, yb, strips, BH, UR, S, G, first, spfull, nm, best STAMP_PASS);
            } else {
                if (big) fast_block<J, false, true>(g, t, scores, lds, lp, xb, yb, strips, BH, UR, S, G, first, spfull, nm, best STAMP_PASS);
                else fast_block<J, false, false>(g, t, scores, lds, lp, xb, yb, strips, BH, UR, S, G, first, spfull, nm, best STAMP_PASS);
            }
            first = false;
            nblocks++;
            fs.border = border; fs.big = big; fs.J = J; fs.xb = xb; fs.yb = yb; fs.strips = strips; fs.bh = BH;
        }
        yb += BH;
    }
    fs.single_block = nblocks == 1;
}

// VALU operations per useful cell and 4-pixel group for strips of J cells (4 selector ops + 3 per cell)
__device__ __forceinline__ int fast_cost(const Geo& g, int J) {
    const int o = ((g.x0 % 4) + 4) & 3;
    const int strips = (g.aw + o + J - 1) / J;
    return (4 + 3 * J) * strips;
}

// Score of a cell of the (single) block that is still held in the accumulators.
__device__ __forceinline__ float fast_cell_score(const Geo& g, const lds_u8* lds, const FastState& fs,
                                                 int x, int y, const Norms& nm) {
    if (fs.J == 8) {
        if (fs.border) return fs.big ? cell_from_acc<8, true, true>(g, lds, fs.xb, fs.yb, fs.strips, x, y, nm)
                                     : cell_from_acc<8, true, false>(g, lds, fs.xb, fs.yb, fs.strips, x, y, nm);
        return fs.big ? cell_from_acc<8, false, true>(g, lds, fs.xb, fs.yb, fs.strips, x, y, nm)
                      : cell_from_acc<8, false, false>(g, lds, fs.xb, fs.yb, fs.strips, x, y, nm);
    }
    if (fs.border) return fs.big ? cell_from_acc<4, true, true>(g, lds, fs.xb, fs.yb, fs.strips, x, y, nm)
                                 : cell_from_acc<4, true, false>(g, lds, fs.xb, fs.yb, fs.strips, x, y, nm);
    return fs.big ? cell_from_acc<4, false, true>(g, lds, fs.xb, fs.yb, fs.strips, x, y, nm)
                  : cell_from_acc<4, false, false>(g, lds, fs.xb, fs.yb, fs.strips, x, y, nm);
}

}  // namespace slamdev
