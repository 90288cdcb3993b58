// FAST-9-16 corner score + non-max suppression (cv::FAST as called at new_point.cpp:120) and
// SLAM::getSortedKeyPoints (new_point.cpp:113-127).
//
// Kernel 1 (fast_tile_kernel): one workgroup (4 wavefronts) per tile of 248 x 30 output pixels of one
//   frame.  The tile's pixels (+ 4-pixel halo) are staged into LDS once.
//   Phase 1: each lane tests 4 adjacent pixels with the opposite-pair test (a 9-long arc contains one
//   pixel of every opposite ring pair, so "all 8 pairs have a brighter / darker member" is necessary),
//   8 pixel rows per wavefront on shared window rows, the even pairs first and the odd ones only on
//   dense rows; the candidates of the 8 rows are listed with one wave scan.
//   Phase 2: each lane scores two listed candidates at once (exact, threshold-free); the stored score
//   (score for corners, 0 otherwise) of tile rows -1..30 and columns -4..251 goes into an LDS score tile.
//   Phase 3: each wavefront applies the strict 3x3 non-max test to its output rows; the tile's survivors
//   are written in row-major order into the tile's area of the frame's key table, the rows' counts
//   into a count table.  No atomics: every tile has a fixed place.
// Kernel 2 (fast_scan_kernel): one workgroup per frame turns the (row, tile) counts into offsets in
//   row-major order - cv::FAST's keypoint order - and reports the frame's total.
// Kernel 3 (fast_scatter_kernel): writes every tile's keys to their places in the output as keypoints.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "slam_internal.h"

namespace {

constexpr int FAST_WG = 256;
constexpr int FAST_TW = 248;                    // output columns per tile
constexpr int FAST_CW = 256;                    // columns with a stored score: 4 halo columns each side (4 pixels per lane)
constexpr int FAST_PPITCH = FAST_CW + 8;        // pixel tile pitch: 4 more columns each side for the ring
constexpr int FAST_TH = 30;                     // output rows per tile
constexpr int FAST_SR = FAST_TH + 2;            // score rows: one halo row above and below; 8 per wavefront
constexpr int FAST_PR = FAST_TH + 8;            // pixel rows
#ifndef FAST_DENSE_LANES
#define FAST_DENSE_LANES 24                     // lanes of a row with a candidate after half the pair test: beyond, the other half runs
#endif
constexpr int FAST_LIST = 600;                  // candidate slots per wavefront; a row adds at most 2 * FAST_CW (one per pixel and polarity).  With 600 the
                                                // kernel's LDS is 23024 bytes = 18 of the CU's 128 granules of 1280 bytes: 7 workgroups per CU (768: 20 granules, 6)

struct FastPlan {
    int ntx, nty;     // tiles per row / column of tiles
};

bool fast_plan(int w, int h, FastPlan& p) {
    if (w < 1 || h < 1 || w > (1 << 16) || h > (1 << 20)) return false;
    p.ntx = (w + FAST_TW - 1) / FAST_TW;
    p.nty = (h + FAST_TH - 1) / FAST_TH;
    return true;
}

// Per-frame scratch (4-byte words): rowcnt and rowoff, 32 words per tile each (one per tile row, tile-major, so a tile
// writes and reads whole 128-byte lines), then the key table: FAST_TW * FAST_TH slots per tile of 16-bit keys
// ((column in tile << 8) | stored score); a tile's keys fill its slots from the first, in row-major order, so the bytes
// written and read back are the keys themselves rather than one part-filled line per row.
__host__ __device__ inline size_t fast_tab_words(int h, int ntx) { return (size_t)((h + FAST_TH - 1) / FAST_TH) * ntx * 32; }
__host__ __device__ inline size_t fast_frame_words(int h, int ntx) {
    return 2 * fast_tab_words(h, ntx) + (size_t)((h + FAST_TH - 1) / FAST_TH) * ntx * (FAST_TW * FAST_TH / 2);
}

// With 8 frames or more in a launch, all tiles of a frame run on one XCD (workgroups are dealt to the 8 XCDs round-robin
// by linear index), so the halo rows and the 128-byte lines that neighbouring tiles share are fetched from HBM once and
// found in that XCD's L2 afterwards.  The grid is then one-dimensional: 8 * ceil(nframes / 8) * tiles_per_frame.
struct FastTile { int tx, ty, frame; };
__device__ __forceinline__ FastTile fast_tile_of_block(int ntx, int nty, int nframes) {
    FastTile f;
    if (gridDim.y == 1 && gridDim.z == 1 && nframes >= 8) {
        const unsigned b = blockIdx.x, per = (unsigned)(ntx * nty), g = b >> 3;
        const unsigned round = g / per, tile = g - round * per;
        f.frame = (int)(round * 8 + (b & 7));
        f.ty = (int)(tile / (unsigned)ntx);
        f.tx = (int)(tile - (unsigned)f.ty * ntx);
    } else {
        f.tx = blockIdx.x; f.ty = blockIdx.y; f.frame = blockIdx.z;
    }
    return f;
}

typedef __attribute__((address_space(3))) unsigned char fl_u8;
typedef __attribute__((address_space(3))) unsigned short fl_u16;
typedef __attribute__((address_space(3))) unsigned fl_u32;

// (dx, dy) of ring position k of cv::FAST's 16-pixel circle
__device__ __forceinline__ constexpr int ring_dx(int k) {
    constexpr int DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
    return DX[k];
}
__device__ __forceinline__ constexpr int ring_dy(int k) {
    constexpr int DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
    return DY[k];
}

typedef short fast_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short fast_u2 __attribute__((ext_vector_type(2)));

// Two of a lane's four pixels' ring neighbours as 16-bit halves: window bytes (4 + dx + odd) and
// (6 + dx + odd) of the 12-byte row window {a, b, c}, picked with one v_perm_b32.
__device__ __forceinline__ fast_u2 ring_pair(unsigned a, unsigned b, unsigned c, int dx, int odd) {
    const int i0 = 4 + dx + odd, i2 = i0 + 2;
    unsigned r;
    if (i2 <= 7) r = __builtin_amdgcn_perm(b, a, 0x0c000c00u | (unsigned)i0 | ((unsigned)i2 << 16));
    else r = __builtin_amdgcn_perm(c, b, 0x0c000c00u | (unsigned)(i0 - 4) | ((unsigned)(i2 - 4) << 16));
    return __builtin_bit_cast(fast_u2, r);
}

__device__ __forceinline__ unsigned pk_sub_sat_u16(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned pk_min_u16(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned pk_mad_u16(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// Three-operand packed minimum / maximum exist on gfx950 for half floats only.  16-bit values below 0x400 are positive
// subnormal half floats, which order like their bit patterns, so on such values these are integer min3 / max3.
__device__ __forceinline__ unsigned pk_min3_h(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned pk_max3_h(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// cv::FAST's cornerScore "A" value of two candidates at once, one per 16-bit half: max over the 16 arcs of 9 contiguous
// ring pixels of the arc minimum of (ring - v) for a candidate of the bright polarity, of (v - ring) for the dark one.
// xm holds 0xFF in the halves of dark candidates.  The differences are kept with a bias of 256 (1..511: below 0x400, see
// pk_min3_h): arc minima as min3 of min3 (9 = 3 x 3), their maximum by a max3 tree.
__device__ __forceinline__ fast_s2 fast_arc_scores2(const fl_u8* p1, const fl_u8* p2, int pitch, unsigned xm) {
    // difference + 256 = ring * s + (256 - v * s) per half, s = +1 (bright) or -1 (dark), modulo 2^16: one packed
    // multiply-add per ring pixel pair.  sg holds 1 or 0xFFFF in the halves.
    const unsigned V = (unsigned)p1[0] | ((unsigned)p2[0] << 16);
    const unsigned sg = 0x00010001u | xm | (xm << 8);      // xm is 0xFF in the halves of dark candidates
    const unsigned W = pk_mad_u16(V, sg ^ 0xFFFEFFFEu, 0x01000100u);     // 256 - v * s: multiply by -s (1 <-> 0xFFFF)
    unsigned e[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int o = ring_dy(k) * pitch + ring_dx(k);
        e[k] = pk_mad_u16((unsigned)p1[o] | ((unsigned)p2[o] << 16), sg, W);
    }
    unsigned m3[16], m9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) m3[k] = pk_min3_h(e[k], e[(k + 1) & 15], e[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) m9[k] = pk_min3_h(m3[k], m3[(k + 3) & 15], m3[(k + 6) & 15]);
    unsigned g[5];
#pragma unroll
    for (int k = 0; k < 5; k++) g[k] = pk_max3_h(m9[3 * k], m9[3 * k + 1], m9[3 * k + 2]);
    const unsigned best = pk_max_u16(pk_max3_h(g[0], g[1], g[2]), pk_max3_h(g[3], g[4], m9[15]));
    return fast_s2{(short)((int)(best & 0xffffu) - 256), (short)((int)(best >> 16) - 256)};
}

// Opposite-pair test of a lane's 4 adjacent pixels, two pixels per operation (pixels j = 0 and 2 as 16-bit halves, then
// pixels 1 and 3).  A 9-long arc holds a member of every opposite ring pair, so for a bright arc every pair has a member
// brighter than v+t (min over pairs of the pair maximum > v+t), for a dark arc every pair has a member darker than v-t
// (max over pairs of the pair minimum < v-t).  The test runs on precomputed shifted pairs: S[d][s + 3] holds bytes
// (4 + s, 6 + s) of window row d (s = -3..4) as 16-bit halves, i.e. ring_pair(.., dx, o) with s = dx + o.  A window row's shifted pairs are shared by the pixel rows
// and ring positions that meet on it (dy = -1, 0, +1 use the same four, dy = +-3 and dy = +-2 overlap in two), so the
// 8 rows of a wavefront need about 100 byte picks instead of 272.  `q`: the pixel row, whose window rows are q .. q + 6.
template <int Q>
__device__ __forceinline__ void fast_pair_test_shared(const fast_u2 (&S)[FAST_SR / 4 + 6][8], int t, unsigned inside, unsigned& flags) {
    // Pixel values (0..255) in 16-bit halves are also positive subnormal half floats, which order like their bit
    // patterns: the three-operand packed minimum / maximum of gfx950 (half-float only) halves the reduction trees.
    // Flags of the lane's four pixels in one register: bright pixel j at bit (j & 1) + 16 * (j >> 1), dark two bits
    // higher.  Saturating differences are non-zero exactly when the comparison holds; min(.., 1) makes them flags.
    // (Written as instructions: left to itself the compiler turns each flag into a compare and a select.)
    const unsigned T2 = (unsigned)t * 0x10001u, ONE = 0x10001u;     // no carry between the halves: pixel + t < 2^16
    unsigned mn_of_mx[2], mx_of_mn[2], v[2];
    // the even pairs first (the compass points and the diagonals) ...
    flags = 0;
#pragma unroll
    for (int o = 0; o < 2; o++) {
        unsigned mx[4], mn[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const fast_u2 r0 = S[Q + 3 + ring_dy(2 * k)][ring_dx(2 * k) + o + 3];
            const fast_u2 r1 = S[Q + 3 + ring_dy(2 * k + 8)][ring_dx(2 * k + 8) + o + 3];
            mx[k] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(r0, r1));
            mn[k] = __builtin_bit_cast(unsigned, __builtin_elementwise_min(r0, r1));
        }
        mn_of_mx[o] = pk_min_u16(pk_min3_h(mx[0], mx[1], mx[2]), mx[3]);
        mx_of_mn[o] = pk_max_u16(pk_max3_h(mn[0], mn[1], mn[2]), mn[3]);
        v[o] = __builtin_bit_cast(unsigned, S[Q + 3][o + 3]);
        const unsigned nb = pk_min_u16(pk_sub_sat_u16(mn_of_mx[o], v[o] + T2), ONE);
        const unsigned nd = pk_min_u16(pk_sub_sat_u16(v[o], mx_of_mn[o] + T2), ONE);
        flags |= (nb << o) | (nd << (2 + o));
    }
    flags &= inside;
    // ... and the odd pairs only where the even ones leave many candidates (wave-uniform): a candidate that the full
    // test would have dropped costs one exact score, about three times a pixel's share of this test, so the second half
    // pays on dense rows and is skipped on sparse ones.  Either way the flags are a superset of the corners.
    if (__popcll(__ballot(flags != 0u)) > FAST_DENSE_LANES) {
        unsigned f2 = 0;
#pragma unroll
        for (int o = 0; o < 2; o++) {
            unsigned mx[4], mn[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const fast_u2 r0 = S[Q + 3 + ring_dy(2 * k + 1)][ring_dx(2 * k + 1) + o + 3];
                const fast_u2 r1 = S[Q + 3 + ring_dy(2 * k + 9)][ring_dx(2 * k + 9) + o + 3];
                mx[k] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(r0, r1));
                mn[k] = __builtin_bit_cast(unsigned, __builtin_elementwise_min(r0, r1));
            }
            const unsigned a = pk_min3_h(pk_min3_h(mx[0], mx[1], mx[2]), mx[3], mn_of_mx[o]);
            const unsigned b = pk_max3_h(pk_max3_h(mn[0], mn[1], mn[2]), mn[3], mx_of_mn[o]);
            const unsigned nb = pk_min_u16(pk_sub_sat_u16(a, v[o] + T2), ONE);
            const unsigned nd = pk_min_u16(pk_sub_sat_u16(v[o], b + T2), ONE);
            f2 |= (nb << o) | (nd << (2 + o));
        }
        flags &= f2;
    }
}

template <int Q>
__device__ __forceinline__ void fast_rows_shared(fast_u2 (&S)[FAST_SR / 4 + 6][8], const fl_u8* px, int ry0, int lane, int y_first, int h,
                                                 int t, bool skip, unsigned inside, unsigned& cand_lo, unsigned& cand_hi) {
    if constexpr (Q < FAST_SR / 4) {
        // window rows 0..6 before the first pixel row, then one new row per pixel row
#pragma unroll
        for (int d = (Q == 0 ? 0 : Q + 6); d <= Q + 6; d++) {
            const fl_u32* row = reinterpret_cast<const fl_u32*>(px + (ry0 + d) * FAST_PPITCH) + lane;
            const unsigned a = row[0], b = row[1], c = row[2];
#pragma unroll
            for (int sft = -3; sft <= 4; sft++) S[d][sft + 3] = ring_pair(a, b, c, sft, 0);
        }
        const int y = y_first + Q;
        unsigned flags = 0;
        if (y >= 3 && y < h - 3 && !skip) fast_pair_test_shared<Q>(S, t, inside, flags);      // wave-uniform condition
        // four rows per register: row Q's flags sit 4 * (Q & 3) bits up, in bits 0..15 and 16..31 alike
        if (Q < 4) cand_lo |= flags << (4 * (Q & 3));
        else cand_hi |= flags << (4 * (Q & 3));
        fast_rows_shared<Q + 1>(S, px, ry0, lane, y_first, h, t, skip, inside, cand_lo, cand_hi);
    }
}

// Inclusive prefix sum over the 64 lanes with DPP moves: four row_shr steps inside each row of 16
// lanes, then row_bcast:15 / row_bcast:31 carry the row totals upwards.
__device__ __forceinline__ unsigned wave_scan_incl(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}


__global__ __launch_bounds__(FAST_WG, 7) void fast_tile_kernel(
    const uint8_t* __restrict__ frames, int w, int h, long long pitch, long long frame_stride,
    int threshold, int nonmax, int ntx, int nty, int nframes, unsigned* __restrict__ scratch) {
    __shared__ __attribute__((aligned(16))) unsigned char px_s[FAST_PR * FAST_PPITCH];    // pixels
    __shared__ __attribute__((aligned(16))) unsigned char sc_s[FAST_SR * FAST_CW];        // stored scores
    __shared__ unsigned short lists_s[4 * FAST_LIST];
    fl_u8* px = (fl_u8*)px_s;
    fl_u8* sc = (fl_u8*)sc_s;
    fl_u16* lists = (fl_u16*)lists_s;

    const FastTile bt = fast_tile_of_block(ntx, nty, nframes);
    const int tx = bt.tx, ty = bt.ty, frame = bt.frame;
    if (frame >= nframes) return;                               // the last round of 8 frames may be short
    // the wavefront's index as a scalar: row numbers, row addresses and the tests on them then stay on the scalar unit
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint8_t* im = frames + (long long)frame * frame_stride;
    const int cx0 = tx * FAST_TW, y0 = ty * FAST_TH;            // first output column / row of the tile
    unsigned* fs = scratch + (size_t)frame * fast_frame_words(h, ntx);
    const bool aligned = ((reinterpret_cast<unsigned long long>(frames) | (unsigned long long)pitch | (unsigned long long)frame_stride) & 3ull) == 0;
    const int t = threshold;
    // diagnostic runs (SLAM_HIP_FAST_DIAG, never set in production) leave out a phase; run-time bits, so that the
    // compiler cannot drop the phases that feed the skipped one: 1 = no pair test, 2 = no scoring, 4 = no suppression
    const int diag = nonmax >> 8;
    nonmax &= 1;

    {   // zero the score tile
        fl_u32* z = (fl_u32*)sc;
#pragma unroll
        for (int i = 0; i < FAST_SR * FAST_CW / 4 / FAST_WG; i++) z[tid + i * FAST_WG] = 0u;
    }
    // ---- stage pixels: tile row r <-> image row y0 - 4 + r, tile byte c <-> image column cx0 - 8 + c.  A wavefront
    // takes every fourth row; lane l carries dword l of the row and lanes 0 and 1 also dwords 64 and 65.  Column
    // bounds depend on the lane only and are settled before the row loop; the loads of all of a wavefront's rows are
    // issued before the first LDS write (one memory round trip).
    {
        constexpr int RW = (FAST_PR + 3) / 4;                    // rows per wavefront
        constexpr int DW = FAST_PPITCH / 4;                      // dwords per tile row (66)
        const int xa = cx0 - 8 + 4 * lane, xb = xa + 256;        // image column of the lane's dword, and of its second one
        const bool in_a = aligned && xa >= 0 && xa + 3 < w, in_b = aligned && lane < DW - 64 && xb >= 0 && xb + 3 < w;
        const bool edge_a = !in_a && xa + 3 >= 0 && xa < w, edge_b = !in_b && lane < DW - 64 && xb + 3 >= 0 && xb < w;
        unsigned va[RW], vb[RW];
        if (aligned && (w & 3) == 0) {
            // the usual case - a dword lies inside the image or outside it, never across its edge: every lane loads
            // from a column clamped into the row and picks the value or zero afterwards, without per-lane branches
            const unsigned xac = in_a ? (unsigned)xa : 0u, xbc = in_b ? (unsigned)xb : 0u;
#pragma unroll
            for (int q = 0; q < RW; q++) {
                const int r = wid + 4 * q, y = y0 - 4 + r;
                va[q] = 0; vb[q] = 0;
                if (r < FAST_PR && y >= 0 && y < h) {            // wave-uniform
                    const uint8_t* src = im + (long long)y * pitch;
                    va[q] = *reinterpret_cast<const unsigned*>(src + xac);
                    vb[q] = *reinterpret_cast<const unsigned*>(src + xbc);
                }
            }
#pragma unroll
            for (int q = 0; q < RW; q++) {
                va[q] = in_a ? va[q] : 0u;
                vb[q] = in_b ? vb[q] : 0u;
            }
        } else
#pragma unroll
        for (int q = 0; q < RW; q++) {
            const int r = wid + 4 * q, y = y0 - 4 + r;
            va[q] = 0; vb[q] = 0;
            if (r < FAST_PR && y >= 0 && y < h) {                // wave-uniform
                const uint8_t* src = im + (long long)y * pitch;
                if (in_a) va[q] = *reinterpret_cast<const unsigned*>(src + (unsigned)xa);      // in_a: xa >= 0
                else if (edge_a) {
#pragma unroll
                    for (int b2 = 0; b2 < 4; b2++)
                        if (xa + b2 >= 0 && xa + b2 < w) va[q] |= (unsigned)src[xa + b2] << (8 * b2);
                }
                if (in_b) vb[q] = *reinterpret_cast<const unsigned*>(src + (unsigned)xb);
                else if (edge_b) {
#pragma unroll
                    for (int b2 = 0; b2 < 4; b2++)
                        if (xb + b2 >= 0 && xb + b2 < w) vb[q] |= (unsigned)src[xb + b2] << (8 * b2);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < RW; q++) {
            const int r = wid + 4 * q;
            if (r < FAST_PR) {
                reinterpret_cast<fl_u32*>(px)[r * DW + lane] = va[q];
                if (lane < DW - 64) reinterpret_cast<fl_u32*>(px)[r * DW + 64 + lane] = vb[q];
            }
        }
    }
    __syncthreads();

    // ---- phase 1: the wavefront's 8 score rows; score row ry <-> image row y0 - 1 + ry <-> pixel row ry + 3;
    // score column c <-> image column cx0 - 4 + c <-> pixel byte c + 4
    constexpr int rows_per_wave = FAST_SR / 4;
    fl_u16* mylist = lists + wid * FAST_LIST;
    // Phase 1 proper: the pair test of the wavefront's 8 score rows.  The 14 window rows they span are read once (8
    // score rows share them); every lane keeps 8 flags per row (fast_pair_test_shared: pixel j bright at bit
    // (j & 1) + 16 * (j >> 1), dark two bits higher), four rows to a register, row q shifted up by 4 * (q & 3).
    unsigned cand_lo = 0, cand_hi = 0;                           // rows 0..3 and 4..7
    {
        const int ry0 = wid * rows_per_wave;
        const int x0 = cx0 - 4 + 4 * lane;                       // image column of the first of the lane's 4 pixels
        unsigned inside = 0;                                     // only pixels with a full ring inside the image are tested
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (x0 + j >= 3 && x0 + j < w - 3) inside |= 5u << ((j & 1) + 16 * (j >> 1));     // both polarities
        fast_u2 S[rows_per_wave + 6][8];
        fast_rows_shared<0>(S, px, ry0, lane, y0 - 1 + ry0, h, t, (diag & 1) != 0, inside, cand_lo, cand_hi);
    }
    // Candidates are listed as (dark << 15) | (row << 8) | score column (the order inside the list is irrelevant: every
    // entry is scored on its own) and scored from the list (phase 2), two per lane.
    auto score_list = [&](int n) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (diag & 2) n = 0;
        // ---- phase 2: exact score of the listed candidates, two per lane (one in each 16-bit half of the arithmetic)
        for (int i = lane; 2 * i < n; i += 64) {
            const int e1 = mylist[2 * i], e2 = mylist[min(2 * i + 1, n - 1)];      // an odd count: the last entry twice
            const int c1 = e1 & 255, ry1 = wid * rows_per_wave + ((e1 >> 8) & 0x7f);
            const int c2 = e2 & 255, ry2 = wid * rows_per_wave + ((e2 >> 8) & 0x7f);
            const unsigned xm = ((e1 & 0x8000) ? 0xFFu : 0u) | ((e2 & 0x8000) ? 0xFF0000u : 0u);
            const fast_s2 a = fast_arc_scores2(px + (ry1 + 3) * FAST_PPITCH + 4 + c1, px + (ry2 + 3) * FAST_PPITCH + 4 + c2, FAST_PPITCH, xm);
            // a pixel cannot hold a bright and a dark arc of 9 at once: of its (at most two) entries one at most stores
            if (a.x > t) sc[ry1 * FAST_CW + c1] = (unsigned char)(nonmax ? a.x - 1 : a.x);   // no NMS: score + 1 marks a corner
            if (a.y > t) sc[ry2 * FAST_CW + c2] = (unsigned char)(nonmax ? a.y - 1 : a.y);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the list is reused by the next rows
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // One scan lists the candidates of all 8 rows when they fit the list - the usual case; otherwise one scan per half
    // (4 rows), and row by row only where even a half does not fit.  Every lane writes its own entries.
    auto list_bits = [&](unsigned bb, int row0, int pos) {       // bb: four rows of flags (fast_rows_shared), first row row0
        while (bb) {
            const int bi = __builtin_ctz(bb);
            bb &= bb - 1u;
            const int rr = row0 + ((bi & 15) >> 2), j = (bi & 1) + 2 * (bi >> 4);
            mylist[pos++] = (unsigned short)((((unsigned)bi & 2u) << 14) | ((unsigned)rr << 8) | (unsigned)(4 * lane + j));
        }
        return pos;
    };
    const unsigned mine8 = (unsigned)(__popc(cand_lo) + __popc(cand_hi));
    if (__ballot(mine8 != 0u) != 0ull) {                         // else: no candidate in the 8 rows, nothing to list
        const unsigned incl8 = wave_scan_incl(mine8);
        const int tot8 = __builtin_amdgcn_readlane((int)incl8, 63);
        if (tot8 <= FAST_LIST) {
            list_bits(cand_hi, 4, list_bits(cand_lo, 0, (int)(incl8 - mine8)));
            score_list(tot8);
        } else {
            for (int half = 0; half < 2; half++) {
                const unsigned bits = half ? cand_hi : cand_lo;
                const unsigned mine4 = (unsigned)__popc(bits);
                if (__ballot(mine4 != 0u) == 0ull) continue;
                const unsigned incl4 = wave_scan_incl(mine4);
                const int tot4 = __builtin_amdgcn_readlane((int)incl4, 63);
                if (tot4 <= FAST_LIST) {
                    list_bits(bits, 4 * half, (int)(incl4 - mine4));
                    score_list(tot4);
                    continue;
                }
                // dense rows: row by row, the list scored whenever another row (at most 2 * FAST_CW entries) might not fit
                int n = 0;
                for (int rq = 0; rq <= 4; rq++) {
                    if (n > 0 && (rq == 4 || n + 2 * FAST_CW > FAST_LIST)) { score_list(n); n = 0; }
                    if (rq == 4) break;
                    const int rr = 4 * half + rq;
                    const unsigned cbd = (bits >> (4 * rq)) & 0x000F000Fu;
                    if (__ballot(cbd != 0u) != 0ull) {
                        const unsigned mine = (unsigned)__popc(cbd);
                        const unsigned incl = wave_scan_incl(mine);
                        int pos = n + (int)(incl - mine);
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int bit = (j & 1) + 16 * (j >> 1);
                            if ((cbd >> bit) & 1u) mylist[pos++] = (unsigned short)((rr << 8) | (4 * lane + j));
                            if ((cbd >> (bit + 2)) & 1u) mylist[pos++] = (unsigned short)(0x8000u | (rr << 8) | (4 * lane + j));
                        }
                        n += __builtin_amdgcn_readlane((int)incl, 63);
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- phase 3: non-max suppression.  Pass A finds the kept pixels of the wavefront's rows (4 bits per lane and row,
    // 8 rows in one register) and the rows' counts; the wavefronts' totals give each its first slot in the tile's key
    // area; pass B writes the keys there in row-major order.
    unsigned* rowcnt = fs + (size_t)(ty * ntx + tx) * 32;
    fl_u32* wtot = (fl_u32*)lists;                               // the candidate lists are done with
    fl_u32* rowtot = wtot + 4;
    unsigned short* tkeys = reinterpret_cast<unsigned short*>(fs + 2 * fast_tab_words(h, ntx)) +
                            (size_t)(ty * ntx + tx) * (FAST_TW * FAST_TH);
    const int r_lo = (FAST_TH * wid) >> 2, r_hi = (FAST_TH * (wid + 1)) >> 2;   // this wavefront's output rows (7 or 8)
    unsigned keepbits = 0;
    int wave_total = 0;
    for (int ro = r_lo; ro < r_hi; ro++) {
        const int y = y0 + ro;
        int total = 0;
        if (y >= 3 && y < h - 3 && !(diag & 4)) {
            const fl_u8* row = sc + (ro + 1) * FAST_CW;          // score row of image row y
            const unsigned c4 = (lane >= 1 && lane <= FAST_TW / 4) ? *reinterpret_cast<const fl_u32*>(row + 4 * lane) : 0u;
            if (__ballot(c4 != 0u) != 0ull) {
                unsigned keep = 0;
                if (c4 != 0u) {
                    if (nonmax) {
                        unsigned U[3], M[3], D[3];
#pragma unroll
                        for (int q = 0; q < 3; q++) {
                            U[q] = *reinterpret_cast<const fl_u32*>(row - FAST_CW + 4 * lane - 4 + 4 * q);
                            M[q] = *reinterpret_cast<const fl_u32*>(row + 4 * lane - 4 + 4 * q);
                            D[q] = *reinterpret_cast<const fl_u32*>(row + FAST_CW + 4 * lane - 4 + 4 * q);
                        }
                        // two pixels per operation, as in phase 1: pixels j = o and o + 2 as 16-bit halves, shifted pairs
                        // shared between the neighbours (s = dx + o); a score is kept when it exceeds the largest of its
                        // 8 neighbours (a zero score never does).  Scores are below 0x400: see pk_max3_h.
                        unsigned Us[4], Ms[4], Ds[4];
#pragma unroll
                        for (int sft = -1; sft <= 2; sft++) {
                            Us[sft + 1] = __builtin_bit_cast(unsigned, ring_pair(U[0], U[1], U[2], sft, 0));
                            Ms[sft + 1] = __builtin_bit_cast(unsigned, ring_pair(M[0], M[1], M[2], sft, 0));
                            Ds[sft + 1] = __builtin_bit_cast(unsigned, ring_pair(D[0], D[1], D[2], sft, 0));
                        }
#pragma unroll
                        for (int o = 0; o < 2; o++) {
                            const unsigned nbu = pk_max3_h(Us[o], Us[o + 1], Us[o + 2]);
                            const unsigned nbd = pk_max3_h(Ds[o], Ds[o + 1], Ds[o + 2]);
                            const unsigned nb = pk_max3_h(nbu, nbd, pk_max_u16(Ms[o], Ms[o + 2]));
                            const unsigned hit = pk_sub_sat_u16(Ms[o + 1], nb);
                            if (hit & 0xffffu) keep |= 1u << o;
                            if (hit >> 16) keep |= 4u << o;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if ((c4 >> (8 * j)) & 0xffu) keep |= 1u << j;
                    }
                }
                keepbits |= keep << (4 * (ro - r_lo));
                total = __popcll(__ballot(keep & 1u)) + __popcll(__ballot(keep & 2u)) + __popcll(__ballot(keep & 4u)) + __popcll(__ballot(keep & 8u));
            }
        }
        wave_total += total;
        if (lane == 0) rowtot[ro] = (unsigned)total;
    }
    if (lane == 0) wtot[wid] = (unsigned)wave_total;
    __syncthreads();
    if (wid == 0 && lane < FAST_TH) rowcnt[lane] = rowtot[lane];
    if (wave_total == 0) return;                                 // wave-uniform; no barrier follows
    int slot = 0;
    for (int k = 0; k < wid; k++) slot += (int)wtot[k];
    for (int ro = r_lo; ro < r_hi; ro++) {
        const unsigned keep = (keepbits >> (4 * (ro - r_lo))) & 15u;
        if (__ballot(keep != 0u) == 0ull) continue;
        // column order inside the row = lane-major, then j
        int before = 0, total = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned long long bal = __ballot((keep >> j) & 1u);
            before += __popcll(bal & ((1ull << lane) - 1ull));
            total += __popcll(bal);
        }
        if (keep) {
            const unsigned c4 = *reinterpret_cast<const fl_u32*>(sc + (ro + 1) * FAST_CW + 4 * lane);
            int pos = slot + before;
#pragma unroll
            for (int j = 0; j < 4; j++)
                if ((keep >> j) & 1u) {
                    const unsigned sv = (c4 >> (8 * j)) & 0xffu;
                    tkeys[pos++] = (unsigned short)(((unsigned)(4 * lane - 4 + j) << 8) | (nonmax ? sv : sv - 1u));
                }
        }
        slot += total;
    }
}

// One workgroup per frame: exclusive prefix sums of the (row, tile) counts in row-major order (row y of tile column tx
// is word (y / FAST_TH * ntx + tx) * 32 + y % FAST_TH of the tables).
__global__ __launch_bounds__(FAST_WG) void fast_scan_kernel(int h, int ntx, unsigned* __restrict__ scratch, int* __restrict__ counts) {
    __shared__ unsigned part[4];
    __shared__ unsigned carry;
    const int frame = blockIdx.x, tid = threadIdx.x;
    unsigned* fs = scratch + (size_t)frame * fast_frame_words(h, ntx);
    const unsigned* rowcnt = fs;
    unsigned* rowoff = fs + fast_tab_words(h, ntx);
    const int n = h * ntx;
    if (tid == 0) carry = 0u;
    __syncthreads();
    for (int base = 0; base < n; base += FAST_WG * 4) {
        const int i0 = base + tid * 4;
        unsigned c[4];
        int at[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int i = i0 + q, y = i / ntx, tx = i - y * ntx, ty = y / FAST_TH;
            at[q] = (ty * ntx + tx) * 32 + (y - ty * FAST_TH);
            c[q] = i < n ? rowcnt[at[q]] : 0u;
        }
        const unsigned mine = c[0] + c[1] + c[2] + c[3];
        const unsigned incl = wave_scan_incl(mine);
        if ((tid & 63) == 63) part[tid >> 6] = incl;
        __syncthreads();
        unsigned off = carry + incl - mine;
        for (int k = 0; k < (tid >> 6); k++) off += part[k];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (i0 + q < n) rowoff[at[q]] = off;
            off += c[q];
        }
        __syncthreads();
        if (tid == 0) carry += part[0] + part[1] + part[2] + part[3];
        __syncthreads();
    }
    if (tid == 0) counts[frame] = (int)carry;
}

// One wavefront per tile: lane r looks up row r's segment; the tile's keys are then numbered across its
// rows and dealt to the lanes 64 at a time (a binary search over the rows' running totals finds the
// segment of key number i), so the lanes stay busy however few keys a single row holds.
__global__ __launch_bounds__(64) void fast_scatter_kernel(int w, int h, int ntx, int nty, int nframes, int cap,
                                                          const unsigned* __restrict__ scratch, slam_keypoint* __restrict__ out) {
    __shared__ int start_s[32], off_s[32];
    const FastTile bt = fast_tile_of_block(ntx, nty, nframes);
    const int tx = bt.tx, ty = bt.ty, frame = bt.frame;
    if (frame >= nframes) return;
    const int lane = threadIdx.x;
    const unsigned* fs = scratch + (size_t)frame * fast_frame_words(h, ntx);
    const unsigned* rowcnt = fs;
    const unsigned* rowoff = fs + fast_tab_words(h, ntx);
    const unsigned short* keys = reinterpret_cast<const unsigned short*>(fs + 2 * fast_tab_words(h, ntx));
    slam_keypoint* o = out + (size_t)frame * cap;
    const int y_l = ty * FAST_TH + lane;
    int n_l = 0, off_l = 0;
    if (lane < FAST_TH && y_l < h) { n_l = (int)rowcnt[(ty * ntx + tx) * 32 + lane]; off_l = (int)rowoff[(ty * ntx + tx) * 32 + lane]; }
    const int incl = (int)wave_scan_incl((unsigned)n_l);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    if (total == 0) return;
    if (lane < 32) { start_s[lane] = incl - n_l; off_s[lane] = off_l; }   // rows 30 and 31 start at `total`
    __syncthreads();
    for (int i = lane; i < total; i += 64) {
        int r = 0;                                   // last row whose first key number is <= i
#pragma unroll
        for (int step = 16; step >= 1; step >>= 1)
            if (start_s[r + step] <= i) r += step;
        const long long dst = (long long)(unsigned)off_s[r] + (i - start_s[r]);
        if (dst < cap) {
            const int y = ty * FAST_TH + r;
            const unsigned key = keys[(size_t)(ty * ntx + tx) * (FAST_TW * FAST_TH) + i];
            slam_keypoint kp;
            kp.x = (float)(tx * FAST_TW + (int)(key >> 8)); kp.y = (float)y; kp.response = (float)(key & 0xffu);
            o[dst] = kp;
        }
    }
}

}  // namespace

static int fast_diag_bits() {
    static const int bits = [] { const char* e = getenv("SLAM_HIP_FAST_DIAG"); return e ? (atoi(e) & 7) : 0; }();
    return bits;
}

extern "C" size_t slam_hip_fast_scratch_bytes(int w, int h, int nframes, int cap_per_frame) {
    FastPlan p;
    if (w < 1 || h < 1 || nframes < 1 || cap_per_frame < 1) return 0;
    if (!fast_plan(w, h, p)) return 0;
    return fast_frame_words(h, p.ntx) * 4 * (size_t)nframes;     // independent of the output capacity
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
    if (!fast_plan(w, h, p))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "fast_detect: unsupported image size %d x %d", w, h);
    if (nframes > 65535 || p.nty > 65535)
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "fast_detect: at most 65535 frames and tile rows per call");
    const size_t need = fast_frame_words(h, p.ntx) * 4 * (size_t)nframes;
    if (scratch_bytes < need || ((uintptr_t)d_scratch & 15))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "fast_detect: scratch too small or misaligned (%zu < %zu)", scratch_bytes, need);
    threshold = std::min(std::max(threshold, 0), 255);          // cv::FAST clamps the threshold
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // every word of the tables that is read is written by the kernel before it: nothing to clear
    // 8 frames or more: one-dimensional grid, a frame per XCD (fast_tile_of_block)
    const long long flat = (long long)((nframes + 7) / 8 * 8) * p.ntx * p.nty;
    const dim3 grid = (nframes >= 8 && flat < (1ll << 24)) ? dim3((unsigned)flat) : dim3(p.ntx, p.nty, nframes);
    hipLaunchKernelGGL(fast_tile_kernel, grid, dim3(FAST_WG), 0, ctx->stream,
                       d_frames, w, h, (long long)pitch, (long long)frame_stride, threshold, (nonmax ? 1 : 0) | (fast_diag_bits() << 8),
                       p.ntx, p.nty, nframes, (unsigned*)d_scratch);
    SLAM_HIP_CHECK(ctx, hipGetLastError());
    hipLaunchKernelGGL(fast_scan_kernel, dim3(nframes), dim3(FAST_WG), 0, ctx->stream, h, p.ntx, (unsigned*)d_scratch, d_counts);
    SLAM_HIP_CHECK(ctx, hipGetLastError());
    hipLaunchKernelGGL(fast_scatter_kernel, grid, dim3(64), 0, ctx->stream, w, h, p.ntx, p.nty, nframes,
                       cap_per_frame, (const unsigned*)d_scratch, d_out);
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
    for (int f = 0; f < nframes; f++) {
        rc = slam_upload_image(ctx, (uint8_t*)ctx->d_stage[0] + (size_t)f * dstride, dpitch, h_frames + (size_t)f * frame_stride, pitch,
                               (size_t)w, (size_t)h, (size_t)f * dstride);
        if (rc != SLAM_HIP_OK) return rc;
    }
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
    slam_stream_drain drain(ctx);        // error returns below wait for the copies queued so far
    size_t dpitch, dstride;
    int rc = fast_upload(ctx, h_frames, w, h, pitch, frame_stride, nframes, dpitch, dstride);
    if (rc != SLAM_HIP_OK) return rc;
    return fast_detect_resident(ctx, (const uint8_t*)ctx->d_stage[0], w, h, dpitch, dstride, nframes,
                                threshold, nonmax, out, cap_per_frame, counts);
}

// One FAST pass over the resident image followed by the device sort (new_point.cpp:120,124): the frame's corners,
// sorted by response descending (ties y, x ascending), arrive in kps; *found = their number.
static int fast_sorted_resident(slam_ctx* ctx, const uint8_t* d_frame, int w, int h, size_t dpitch, size_t dstride,
                                int threshold, std::vector<slam_keypoint>& kps, size_t* found) {
    const int cap = (int)kps.size();
    const size_t sb = slam_hip_fast_scratch_bytes(w, h, 1, cap);
    if (!sb) return slam_set_error(ctx, SLAM_HIP_EINVAL, "get_sorted_keypoints: unsupported shape %dx%d", w, h);
    const size_t ob = sizeof(slam_keypoint) * (size_t)cap;          // slot 2: the row-major list, then the sorted one
    int rc;
    if ((rc = slam_stage_reserve(ctx, 1, sb)) || (rc = slam_stage_reserve(ctx, 2, 2 * ob)) || (rc = slam_stage_reserve(ctx, 3, 4)))
        return rc;
    slam_keypoint* d_list = (slam_keypoint*)ctx->d_stage[2];
    slam_keypoint* d_sorted = d_list + cap;
    rc = slam_hip_fast_detect_d(ctx, d_frame, w, h, dpitch, dstride, 1, threshold, 1, d_list, cap, (int32_t*)ctx->d_stage[3],
                                ctx->d_stage[1], ctx->stage_bytes[1]);
    if (rc != SLAM_HIP_OK) return rc;
    rc = slam_hip_sort_keypoints_d(ctx, d_list, (const int32_t*)ctx->d_stage[3], 1, cap, d_sorted);
    if (rc != SLAM_HIP_OK) return rc;
    int32_t cnt = 0;
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(&cnt, ctx->d_stage[3], 4, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (cnt > cap) return slam_set_error(ctx, SLAM_HIP_ECAP, "get_sorted_keypoints: %d corners, room for %d", cnt, cap);
    if (cnt > 0) {
        SLAM_HIP_CHECK(ctx, hipMemcpyAsync(kps.data(), d_sorted, sizeof(slam_keypoint) * (size_t)cnt, hipMemcpyDeviceToHost, ctx->stream));
        SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    *found = (size_t)cnt;
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_get_sorted_keypoints(slam_ctx* ctx, const uint8_t* im, int w, int h, size_t pitch,
                                             size_t n_min_pts, float* fast_threshold,
                                             slam_keypoint* out, int cap, int32_t* n_out) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (!im || !fast_threshold || !out || !n_out || w < 1 || h < 1 || cap < 1 || pitch < (size_t)w)
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "get_sorted_keypoints: bad arguments");
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    slam_stream_drain drain(ctx);        // error returns below wait for the copies queued so far
    float f = *fast_threshold;
    size_t n = 0;
    std::vector<slam_keypoint> kps;
    // new_point.cpp:116-122: while (out.size() < nMinPts) { FAST(im, out, fastThreshold, true); fastThreshold *= .5; }
    // With nMinPts == 0 the loop body never runs: no keypoints, and the threshold is only doubled below.
    if (n_min_pts > 0) {
        size_t dpitch, dstride;
        int rc = fast_upload(ctx, im, w, h, pitch, pitch * (size_t)h, 1, dpitch, dstride);
        if (rc != SLAM_HIP_OK) return rc;
        // non-max-suppressed corners are never 8-adjacent: at most ceil(w/2)*ceil(h/2) of them
        const size_t kmax = (size_t)((w + 1) / 2) * (size_t)((h + 1) / 2);
        kps.resize(kmax ? kmax : 1);
        // The first pass runs at the current threshold and usually ends the loop.  When it does not, ONE more pass at
        // threshold 0 serves every later iteration: a corner's score does not depend on the threshold it was detected
        // with, and a pixel survives non-max suppression at threshold t >= 1 exactly when it survives at threshold 0 and
        // its score is at least t (a neighbour that is no corner at t scores below t, hence below the pixel), so the
        // list of iteration t is the threshold-0 list without the scores below t - and, both being sorted by
        // (response descending, y, x) on the device (new_point.cpp:124 sorts by response only with an unstable sort;
        // ties are fixed as y, x ascending), a prefix of it.
        const int t = (int)f;                                 // FAST(im, out, fastThreshold, true): float -> int
        rc = fast_sorted_resident(ctx, (const uint8_t*)ctx->d_stage[0], w, h, dpitch, dstride, t, kps, &n);
        if (rc != SLAM_HIP_OK) return rc;
        f *= 0.5f;
        if (n < n_min_pts && t > 0) {       // the reference never leaves this loop on a corner-poor image; here it stops after t = 0
            size_t n0 = 0;
            rc = fast_sorted_resident(ctx, (const uint8_t*)ctx->d_stage[0], w, h, dpitch, dstride, 0, kps, &n0);
            if (rc != SLAM_HIP_OK) return rc;
            for (;;) {
                const int ti = (int)f;
                // keypoints with response >= ti: a prefix of the descending list
                n = (size_t)(std::partition_point(kps.begin(), kps.begin() + n0, [ti](const slam_keypoint& k) { return k.response >= (float)ti; }) - kps.begin());
                f *= 0.5f;
                if (n >= n_min_pts || ti <= 0) break;
            }
        }
    }
    f *= 2.f;                                                 // new_point.cpp:123
    if (n > 2 * n_min_pts) f = kps[(size_t)((double)n_min_pts * 1.9)].response;   // new_point.cpp:125-126
    *fast_threshold = f;
    *n_out = (int32_t)n;
    for (size_t i = 0; i < n && i < (size_t)cap; i++) out[i] = kps[i];
    return SLAM_HIP_OK;
}
