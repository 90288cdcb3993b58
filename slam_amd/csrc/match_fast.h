// LDS-tiled packed-u8 search core (one workgroup of 256 lanes per job).
//
// Data layout in LDS (dynamic, FAST_LDS_BYTES):
//   acc   : per-cell partial sums (CC, SI, SPcrop) as u32, used only when a cell's template rows
//           are split over several lanes
//   tmpl  : the current chunk of template rows, per 4-pixel group {P0 dword, mask dword}:
//           P0 = patch*mask packed 4 pixels/dword, mask = 0xFF per kept pixel (0 beyond pw)
//   tile  : image rows under all cells of the current block, zero outside the image, 4-byte
//           aligned to IMAGE columns so that rows are fetched as dwords when the image is aligned
//   vtile : 0xFF where the tile pixel is inside the image (only for blocks with cropped cells)
//
// Work split: the block's cells are cut into strips of J (4 or 8) consecutive cells of one row; a
// lane owns one strip and one slice of the chunk's template rows.  Per 4-pixel group it forms,
// for each of its J cells, the cell's 4 image pixels with one v_perm_b32 (byte alignment and the
// patch mask in one selector) and feeds two v_dot4_u32_u8 (CC += <Im,P0>, SI += <Im,Im>); blocks
// with cropped cells add SPcrop += <P0 & valid, P0>.  All sums are exact integers.
#pragma once

#include "match_common.h"

#define FAST_LDS_BYTES (24 * 1024)
#define FAST_ACC_BYTES (6 * 1024)
#define FAST_TMPL_BYTES (6 * 1024)
#define FAST_TILE_BYTES (12 * 1024)
#define FAST_MAX_G 256        // template width up to 1024 pixels

namespace slamdev {

struct CoreOut { int x, y; float response, dx, dy; };

struct CoreShared {
    unsigned long long red64[SLAM_WG / 64];
    float bestv[SLAM_WG / 64];
    int besti[SLAM_WG / 64];
    unsigned long long spfull;
    float nb[4];
    int nbok[4];
    CoreOut out;
};

__device__ __forceinline__ unsigned bfi32(unsigned mask, unsigned a, unsigned b) { return (a & mask) | (b & ~mask); }

// Accumulate template rows [ub, ue) (chunk-relative) for the J cells of one strip.
// tile row r holds image row (first image row of the tile) + r; the cell row `row` is relative to
// the block, so template row u of that cell lies in tile row row + u.
template <int J, bool BORDER>
__device__ __forceinline__ void strip_accumulate(const unsigned* __restrict__ tile, const unsigned* __restrict__ vtile,
                                                 int tp, const uint2* __restrict__ tmpl, int G, int ub, int ue,
                                                 int row, int strip, unsigned (&cc)[J], unsigned (&si)[J], unsigned (&sp)[J]) {
    constexpr int JD = J / 4;
    for (int u = ub; u < ue; ++u) {
        const int tofs = (row + u) * tp + strip * JD;
        const unsigned* tr = tile + tofs;
        const unsigned* vr = vtile + tofs;
        const uint2* pr = tmpl + u * G;
        unsigned d0 = tr[0], d1 = (JD == 2) ? tr[1] : 0u;
        unsigned v0 = 0, v1 = 0;
        if (BORDER) { v0 = vr[0]; v1 = (JD == 2) ? vr[1] : 0u; }
#pragma unroll 2
        for (int g = 0; g < G; ++g) {
            const unsigned dn = tr[g + JD];
            const uint2 pm = pr[g];
            const unsigned P0 = pm.x, M = pm.y;
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
                                               unsigned long long SI, unsigned long long SPc, unsigned long long spfull) {
    const int kind = classify_cell(g, x, y);
    if (kind == CELL_NONE) return -1.f;                    // matching.cpp:53
    if (kind == CELL_MT) return score_mt(CC, SI, spfull);
    if (kind == CELL_F32_FULL) return score_f32(CC, SI, spfull);
    return score_f32(CC, SI, SPc);
}

// One block of cells [xb, xb+BW) x [yb, yb+bh) (xb may be < 0: columns left of 0 and right of aw
// are padding).  J cells per strip.
template <int J, bool BORDER, bool BIG>
__device__ __forceinline__ void fast_block(const Geo& g, const Tmpl& t, float* scores, CoreShared& sm,
                                           unsigned char* lds, int xb, int yb, int strips, int bh, int UR, int S,
                                           int G, bool first_block, unsigned long long& sp_part, Best& best) {
    const int tid = threadIdx.x;
    unsigned* acc = reinterpret_cast<unsigned*>(lds);
    uint2* tmpl = reinterpret_cast<uint2*>(lds + FAST_ACC_BYTES);
    unsigned* tile = reinterpret_cast<unsigned*>(lds + FAST_ACC_BYTES + FAST_TMPL_BYTES);
    constexpr int JD = J / 4;
    const int tp = (strips * JD + G) | 1;                  // odd dword pitch: rows spread over the banks
    const int trows_max = (FAST_TILE_BYTES / 4 / (BORDER ? 2 : 1)) / tp;
    unsigned* vtile = tile + trows_max * tp;
    const int units = strips * bh;
    const int ntask = units * S;
    const int ix0 = g.x0 + xb;                             // image column of tile column 0 (multiple of 4)
    const bool aligned = ((reinterpret_cast<unsigned long long>(g.im) | (unsigned long long)g.ipitch) & 3ull) == 0;

    // this lane's task
    const bool active = tid < ntask;
    const int unit = active ? tid / S : 0;
    const int slice = active ? tid - unit * S : 0;
    const int row = unit / strips, strip = unit - row * strips;

    unsigned cc[J], si[J], sp[J];
    unsigned long long tcc[BIG ? J : 1], tsi[BIG ? J : 1], tsp[BIG ? J : 1];
#pragma unroll
    for (int k = 0; k < J; k++) { cc[k] = 0; si[k] = 0; sp[k] = 0; }
    if (BIG) {
#pragma unroll
        for (int k = 0; k < J; k++) { tcc[k] = 0; tsi[k] = 0; tsp[k] = 0; }
    }
    __syncthreads();                                       // the previous block's finalize has read acc
    if (S > 1) {
        for (int i = tid; i < units * J * 3; i += SLAM_WG) acc[i] = 0;
    }

    for (int u0 = 0; u0 < g.ph; u0 += UR) {
        const int ur = min(UR, g.ph - u0);
        __syncthreads();                                   // previous chunk's readers are done
        // ---- stage the template chunk: {P0, mask} per 4-pixel group -----------------------------
        for (int i = tid; i < ur * G; i += SLAM_WG) {
            const int u = i / G, gg = i - u * G;
            unsigned P0 = 0, M = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int v = gg * 4 + b;
                if (v < g.pw) {
                    unsigned p, m;
                    tmpl_fetch(t, u0 + u, v, p, m);
                    p *= m;
                    P0 |= p << (8 * b);
                    M |= (m ? 0xFFu : 0u) << (8 * b);
                    if (first_block) sp_part += p * p;
                }
            }
            tmpl[i] = make_uint2(P0, M);
        }
        // ---- stage the image tile (and its validity) --------------------------------------------
        const int trows = bh + ur - 1;
        const int iy0 = g.y0 + yb + u0;
        for (int i = tid; i < trows * tp; i += SLAM_WG) {
            const int r = i / tp, c = i - r * tp;
            const int iy = iy0 + r, ix = ix0 + 4 * c;
            unsigned px = 0, vv = 0;
            if (iy >= 0 && iy < g.h && ix + 3 >= 0 && ix < g.w) {
                const uint8_t* src = g.im + (long long)iy * g.ipitch + ix;
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
            tile[i] = px;
            if (BORDER) vtile[i] = vv;
        }
        __syncthreads();
        // ---- accumulate ---------------------------------------------------------------------------
        if (active) {
            const int ub = (slice * ur) / S, ue = ((slice + 1) * ur) / S;
            strip_accumulate<J, BORDER>(tile, vtile, tp, tmpl, G, ub, ue, row, strip, cc, si, sp);
        }
        if (BIG) {
#pragma unroll
            for (int k = 0; k < J; k++) { tcc[k] += cc[k]; tsi[k] += si[k]; tsp[k] += sp[k]; cc[k] = 0; si[k] = 0; sp[k] = 0; }
        }
    }

    // |patch0|^2 over the whole template, once per job
    if (first_block) {
        const unsigned long long s = wave_sum_u64(sp_part);
        if ((tid & 63) == 0) sm.red64[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) sm.spfull = sm.red64[0] + sm.red64[1] + sm.red64[2] + sm.red64[3];
    }
    if (S > 1) {
        if (active) {
#pragma unroll
            for (int k = 0; k < J; k++) {
                atomicAdd(&acc[(unit * J + k) * 3 + 0], cc[k]);
                atomicAdd(&acc[(unit * J + k) * 3 + 1], si[k]);
                if (BORDER) atomicAdd(&acc[(unit * J + k) * 3 + 2], sp[k]);
            }
        }
    }
    __syncthreads();
    const unsigned long long spfull = sm.spfull;

    // ---- finalize: scores of the block's cells, running argmax -----------------------------------
    if (S > 1) {
        for (int c = tid; c < units * J; c += SLAM_WG) {
            const int un = c / J, k = c - un * J;
            const int r = un / strips, st = un - r * strips;
            const int x = xb + st * J + k, y = yb + r;
            if (x < 0 || x >= g.aw) continue;
            const float sc = finalize_cell(g, x, y, acc[c * 3 + 0], acc[c * 3 + 1], acc[c * 3 + 2], spfull);
            const int idx = y * g.aw + x;
            if (scores) scores[idx] = sc;
            best_update(best, sc, idx);
        }
    } else if (active) {
#pragma unroll
        for (int k = 0; k < J; k++) {
            const int x = xb + strip * J + k, y = yb + row;
            if (x < 0 || x >= g.aw) continue;
            const unsigned long long CC = BIG ? tcc[k] : (unsigned long long)cc[k];
            const unsigned long long SI = BIG ? tsi[k] : (unsigned long long)si[k];
            const unsigned long long SP = BIG ? tsp[k] : (unsigned long long)sp[k];
            const float sc = finalize_cell(g, x, y, CC, SI, SP, spfull);
            const int idx = y * g.aw + x;
            if (scores) scores[idx] = sc;
            best_update(best, sc, idx);
        }
    }
}

// True when the tiled core can run this job; otherwise the caller uses direct evaluation.
__device__ __forceinline__ bool fast_supported(const Geo& g) {
    return (g.pw + 3) / 4 <= FAST_MAX_G && (long long)g.pw * g.ph <= 66051;   // 66051 * 255^2 < 2^32: u32 sums are exact
}

template <int J>
__device__ __forceinline__ void fast_blocks(const Geo& g, const Tmpl& t, float* scores, CoreShared& sm,
                                            unsigned char* lds, Best& best) {
    constexpr int JD = J / 4;
    const int G = (g.pw + 3) / 4;
    const int o = ((g.x0 % 4) + 4) & 3;                    // cells left of 0 that align the tile to image dwords
    const int BW = min((g.aw + o + J - 1) / J * J, 64);
    const int strips_max = BW / J;
    // one plan per band of rows, sized for the widest block and for the two-tile (cropped) case
    const int tp_max = (strips_max * JD + G) | 1;
    const int TR = max(1, (FAST_TILE_BYTES / 4 / 2) / tp_max);
    unsigned long long sp_part = 0;
    bool first = true;
    for (int yb = 0; yb < g.ah;) {
        int UR = min(min(g.ph, FAST_TMPL_BYTES / 8 / G), TR);
        int BH = min(g.ah - yb, SLAM_WG / strips_max);
        if (BH + UR - 1 > TR) {
            // prefer whole-template chunks: shrink the block first, then the chunk
            if (TR - UR + 1 >= 4 || UR == 1) BH = max(1, TR - UR + 1);
            else { UR = max(1, TR / 2); BH = max(1, min(BH, TR - UR + 1)); }
        }
        for (int xb = -o; xb < g.aw; xb += BW) {
            const int bw = min(BW, g.aw - xb);
            const int strips = (bw + J - 1) / J;
            const int cx0 = max(xb, 0), cx1 = xb + bw;
            // does the block hold cells whose patch leaves the image?
            const bool border = cx0 < g.xl0 || cx1 > g.xl1 || yb < g.yl0 || yb + BH > g.yl1;
            const int units = strips * BH;
            int S = max(1, min(min(8, SLAM_WG / units), UR));
            if (units * J * 12 > FAST_ACC_BYTES) S = 1;
            if (border) fast_block<J, true, false>(g, t, scores, sm, lds, xb, yb, strips, BH, UR, S, G, first, sp_part, best);
            else fast_block<J, false, false>(g, t, scores, sm, lds, xb, yb, strips, BH, UR, S, G, first, sp_part, best);
            first = false;
        }
        yb += BH;
    }
}

// ops per useful cell and 4-pixel group for strips of J cells (4 selector ops + 3 per cell)
__device__ __forceinline__ int fast_cost(const Geo& g, int J) {
    const int o = ((g.x0 % 4) + 4) & 3;
    const int strips = (g.aw + o + J - 1) / J;
    return (4 + 3 * J) * strips;
}

}  // namespace slamdev
