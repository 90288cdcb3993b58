// Lean search of one window: a pyramid level of the point walk (SLAM::Feature::track, feature.cpp:124-152) or a single
// SLAM::matchFeatureInArea call (matching.cpp:9-228), in the common case in which every candidate patch of the window
// lies inside the image and the template is small.  Same integer sums, same formulas and the same argmax rule as the
// general core (match_fast.h); what differs is the data layout, chosen so that the inner loop is nearly nothing but v_dot4:
//
//   tile   : the image rows under all cells, one 8-byte entry per dword position p of a row holding the byte windows
//            that start at bytes 4p (W0) and 4p+3 (W3); the two windows in between are one v_perm_b32 each, so a strip
//            of 4 adjacent cells gets its 4 windows from one ds_read_b64
//   lists  : the template's 4-pixel groups as a list of {P0 = patch*mask, tile offset} entries; groups whose mask is
//            all zero are dropped, groups that keep all four pixels come first (their window needs no masking before
//            its energy is summed), the others follow with their mask in a parallel array.  Both parts are padded to
//            a multiple of the slice count, so every lane runs the same trip counts
//   acc    : {CC, SI} per cell, u32, LDS atomics
//
// A lane owns a unit (a strip of 4 adjacent cells at stride 1, one cell on a coarser lattice) and one of S slices of
// the entry lists.  Anything this path does not cover is left to the general kernel: the caller queues the job.
#pragma once

#include "match_fast.h"

namespace slamdev {

typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4u32 lds_v4u32;

// (int)round((double)f) for a float f: C round() (half away from zero) of the same real value; the float is
// split exactly into trunc and fraction, so single precision gives the identical integer.
__device__ __forceinline__ int round_f2i(float f) {
    const float tr = truncf(f);
    const float fr = __fsub_rn(f, tr);                    // exact: |fr| < 1 and tr share f's exponent range
    return (int)tr + (fr >= 0.5f ? 1 : 0) - (fr <= -0.5f ? 1 : 0);
}

__device__ __forceinline__ bool span_ok(long long off, long long len, unsigned long long total) {
    return off >= 0 && len >= 0 && (unsigned long long)off <= total && (unsigned long long)len <= total - (unsigned long long)off;
}

// a value every lane holds alike, moved to scalar registers
__device__ __forceinline__ int uni_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uni_f32(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// Maximum / minimum over the wavefront with DPP moves (see wave_sum_u32); the result is wave-uniform.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, SLAM_DPP_ROW_SHR(1), 0xf, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, SLAM_DPP_ROW_SHR(2), 0xf, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, SLAM_DPP_ROW_SHR(4), 0xf, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, SLAM_DPP_ROW_SHR(8), 0xf, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, SLAM_DPP_BCAST15, 0xa, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, SLAM_DPP_BCAST31, 0xc, 0xf, false));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, SLAM_DPP_ROW_SHR(1), 0xf, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, SLAM_DPP_ROW_SHR(2), 0xf, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, SLAM_DPP_ROW_SHR(4), 0xf, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, SLAM_DPP_ROW_SHR(8), 0xf, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, SLAM_DPP_BCAST15, 0xa, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, SLAM_DPP_BCAST31, 0xc, 0xf, false));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// inclusive prefix sum over the lanes (Hillis-Steele inside the rows of 16 lanes, then the row totals)
__device__ __forceinline__ unsigned wave_scan_u32(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, SLAM_DPP_ROW_SHR(1), 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, SLAM_DPP_ROW_SHR(2), 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, SLAM_DPP_ROW_SHR(4), 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, SLAM_DPP_ROW_SHR(8), 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, SLAM_DPP_BCAST15, 0xa, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, SLAM_DPP_BCAST31, 0xc, 0xf, false);
    return v;
}

struct LeanLevel {           // wave-uniform description of one level's search
    const uint8_t* im;       // level image of the job's frame (4-byte aligned, pitch a multiple of 4)
    unsigned ipitch;
    int w, h;
    int tw, th;              // template size at this level
    int x0, y0;              // image position of template pixel (0,0) for cell (0,0); the window is interior:
    int aw, ah;              //   x0 >= 0, y0 >= 0, x0 + aw + tw - 1 <= w, y0 + ah + th - 1 <= h
    int step;                // lattice step: cells (i*step, j*step) are visited (matching.cpp:64-66 with xl0 = yl0 = 0)
    const uint8_t* amask;    // area mask (pitch aw) or nullptr: cells whose byte is 0 are not visited (matching.cpp:66)
    const uint8_t* patch;    // full-resolution template and mask, pitch srcw
    const uint8_t* mask;     //   nullptr: no patch mask, every template pixel counts
    const uint8_t* limit;    // end of the buffer they live in
    int srcw, srch;
    bool resized;            // template resized to (tw, th) on the fly (feature.cpp:130-131)
    bool mt;                 // scores by cv::matchTemplate's normalisation (matching.cpp:151-153: the modes without a patch mask)
    int own0, own1;          // chunk-capable builds: only cells of rows own0 <= y < own1 compete for the maximum (a band of a window
                             //   searched with a row of its neighbours on either side: the fit around a peak on the band's edge reads them)
    double scx, scy;
};

struct LeanOut { int x, y; float response, dx, dy; bool ok; };

#define LEAN_MAX_S 8

// diagnostic build (-DSLAM_LEAN_DEBUG): job 0 says why a walk was left to the general kernel
#ifdef SLAM_LEAN_DEBUG
#define LEAN_REJECT(code) do { if (blockIdx.x == 0 && threadIdx.x == 0) printf("lean reject %d\n", (code)); } while (0)
#else
#define LEAN_REJECT(code) ((void)0)
#endif

// Slice count S and the per-slice trip counts: na all-kept entries per slice (the nA - na*S left over join the masked
// part), nb masked entries per slice after padding.  The cost model is instruction counts of the loops below; every
// division is by a compile-time constant.
struct LeanSlices { int S, na, nb; };

__device__ __forceinline__ LeanSlices lean_pick_slices(int nunits, int nA, int nB, bool strips) {
    const int ca = strips ? 12 : 5, cb = strips ? 16 : 6, cf = strips ? 30 : 16;
    LeanSlices best = {1, nA, nB};
    int best_c = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < 6; q++) {
        constexpr int cand[6] = {1, 2, 3, 4, 6, 8};          // at most LEAN_MAX_S
        const int s = cand[q];
        const int passes = (nunits * s + SLAM_WG - 1) >> 6;
        const int a = nA / s;
        const int b = (nA - a * s + nB + s - 1) / s;
        const int c = passes * (a * ca + b * cb + cf);
        if (c < best_c) { best_c = c; best.S = s; best.na = a; best.nb = b; }
    }
    return best;
}

// Entries the lists hold at most: the template's groups, the padding of the masked part to a multiple of the slice count
// (at most S - 1 null entries) and the entries the loops read ahead of their use (four past the all-kept part of the
// last slice, two past the masked part).
__host__ __device__ __forceinline__ int lean_cap(int NG) { return NG + LEAN_MAX_S + 4; }

// Entries of the mask array: only the masked part of the lists reads it.  A point walk's template mask is a warped
// quadrilateral (feature.cpp:96-98): a row has at most two partly kept groups, and a last group cut by the template's
// width, so the masked part holds at most 3 * th groups, the all-kept entries left over by the slicing (at most S - 1),
// its padding (at most S - 1) and two entries read ahead.  A walk whose mask has more (noise) is turned away when its
// lists are built and searched by the general kernel.  Single searches keep room for every group.
__host__ __device__ __forceinline__ int lean_mcap(int NG, int th, bool walk) {
    const int cap = lean_cap(NG), m = 3 * th + 2 * LEAN_MAX_S + 4;
    return walk && NG > 32 && m < cap ? m : cap;           // templates of at most 32 groups mask every entry (lean_level)
}

// dwords per template row of the resident P0 array of lean_level's box mode: rows of three groups and more start on
// 16-byte boundaries (four groups' dwords are read at once); with fewer the rows stay packed, which keeps the in-place
// move from the staged {P0, M} pairs (8 bytes a group) behind its own reads
__host__ __device__ __forceinline__ int lean_pc_pitch(int G) { return G >= 3 ? (G + 3) & ~3 : G; }

// Which searches of the chunk-capable builds take their window energies from box sums of the squared image (see
// lean_level): 0 none; 1 templates without a mask (single searches: the P0 dwords of the whole template stay resident);
// 2 masked templates of walks, stored at this level's resolution and large enough for the box sums to pay (their cost
// grows with the cell rows, the cost of the dots they replace with strips x template groups).  (In the plain walk build
// the same scheme, with the boxes summed out of the LDS tile, measured slower on the templates that build is for: a C2
// walk 4347 instead of 4104 VALU instructions - 17 cell rows of prefix sums cost more than the 670 dots they replace -
// and its mere presence in the kernel cost the C2 path 3.6 %; 31x31 templates gained 1 - 2 %.)
#ifndef LEAN_BOX_MIN_WORK
#define LEAN_BOX_MIN_WORK 2048
#endif
#define LEAN_BOX_NONE 0
#define LEAN_BOX_UNMASKED 1
#define LEAN_BOX_MASKED 2
__host__ __device__ __forceinline__ int lean_box_mode(bool search, bool has_mask, int step, bool resized, int tw, int th, int aw) {
    if (step != 1 || aw + tw - 1 > 2 * SLAM_WG) return LEAN_BOX_NONE;
    if (!has_mask) return search ? LEAN_BOX_UNMASKED : LEAN_BOX_NONE;
    const int G = (tw + 3) >> 2, cols = (aw + 3) >> 2;
    return !search && !resized && th * G * cols >= LEAN_BOX_MIN_WORK ? LEAN_BOX_MASKED : LEAN_BOX_NONE;
}

// Entries of the list of groups that drop pixels inside the template (masked box sums): every group at worst, padding
// and read-ahead
__host__ __device__ __forceinline__ int lean_ccap(int NG) { return NG + LEAN_MAX_S + 4; }

// LDS bytes a level needs (INT_MAX: a shape this path does not take), with the window's column phase x0 & 3 at its
// worst, so that the answer depends on sizes only and a walk can be turned away before its first level is searched.
__host__ __device__ __forceinline__ int lean_need(int tw, int th, int aw, int ah, int step, int nx, int ny, bool resized) {
    const int G = (tw + 3) >> 2, NG = th * G;
    const bool strips = step == 1;
    const int cols = (aw + 6) >> 2;
    const int tp = (cols + G - 1) | 1;
    const int trows = ah + th - 1;
    const int nacc = strips ? cols * ah * 4 : nx * ny;
    const int cap = lean_cap(NG);
    const int tile_bytes = (trows * tp * 8 + 15) & ~15;
    const int raw_bytes = NG * 8 + (resized ? (G * 4 + th) * 8 : 0);
    // sums fit 32 bits up to 66051 template pixels; a lane's window positions index the tile with 16 bits to spare
    if (tw > 256 || th > 256 || tw * th > 66051 || tp > 64 || nx * ny > 4096) return INT_MAX;
    return ((nacc * 8 + 15) & ~15) + cap * 8 + ((lean_mcap(NG, th, true) * 4 + 15) & ~15) + (tile_bytes > raw_bytes ? tile_bytes : raw_bytes);
}

// How the chunk-capable builds of the kernels (CHUNKS in lean_level) take a search whose accumulators, lists and tile do
// not fit `budget` bytes at once: the cells in blocks of bh unit rows (tile, sums and scores per block; the first maximum
// in row-major order carries over), the template in chunks of R rows (lists and tile per chunk - the tile then holds the
// image rows under these template rows for the block's cells - while the block's sums run on in the accumulators).  Both
// are equal parts up to rounding.  Of the block counts 1..8 the one with the least staging work is taken (tile entries
// and template groups staged over the whole search); bh = 0: not covered.  `walk`: the mask array's size rule (lean_mcap).
#define LEAN_MIN_CHUNK 4
struct LeanPlan { int bh, R; };

__host__ __device__ __forceinline__ int lean_div(int n, int d) {
#ifdef __HIP_DEVICE_COMPILE__
    return small_div(n, d);
#else
    return n / d;
#endif
}

__host__ __device__ __forceinline__ int lean_plan_bytes(int G, int tp, int upr, int cpu, int rstep, int rows_all, int bh, int R, bool resized, bool walk,
                                                        int th, int box) {
    const int NGc = R * G;
    const int fit_rows = rows_all < 3 ? rows_all : 3;      // the three rows around the peak are summed at once for the fit
    const int acc = ((bh > fit_rows ? bh : fit_rows) * upr * cpu * 8 + 15) & ~15;
    int tile = (((bh - 1) * rstep + R) * tp * 8 + 15) & ~15;
    if (tile < 2 * SLAM_WG * 4) tile = 2 * SLAM_WG * 4;    // the box sums' prefix array lives there before the tile
    if (box == LEAN_BOX_MASKED)                            // the chunk's P0 dwords, its complement list (every group at worst), the tile
        return acc + (((R + 1) * lean_pc_pitch(G) * 4 + 15) & ~15) + lean_ccap(NGc) * 8 + tile;
    if (box) {
        // no lists: the whole template's P0 dwords stay resident, staged (8 bytes a group, taps behind them) where the
        // tile goes afterwards
        const int NG = th * G, keep = ((th + 1) * lean_pc_pitch(G) * 4 + 15) & ~15;      // (+ an all-zero row: lean_accumulate_rows2)
        const int stage = NG * 8 + (resized ? (G * 4 + th) * 8 : 0);
        return acc + (keep + tile > stage ? keep + tile : stage);
    }
    const int raw = NGc * 8 + (resized ? (G * 4 + R) * 8 : 0);
    return acc + lean_cap(NGc) * 8 + ((lean_mcap(NGc, R, walk) * 4 + 15) & ~15) + (tile > raw ? tile : raw);
}

__host__ __device__ __forceinline__ LeanPlan lean_plan(int tw, int th, int aw, int ah, int step, int nx, int ny, bool resized, bool walk, int budget,
                                                       int box = LEAN_BOX_NONE) {
    LeanPlan best = {0, 0};
    const int G = (tw + 3) >> 2;
    const bool strips = step == 1;
    const int cols = (aw + 3) >> 2;                        // (these builds shift the window to the tile's first byte)
    const int tp = (cols + G - 1) | 1;
    const int rows_all = strips ? ah : ny, rstep = strips ? 1 : step, upr = strips ? cols : nx, cpu = strips ? 4 : 1;
    // sums fit 32 bits up to 66051 template pixels; a lane's window positions index the tile with 16 bits to spare
    if (tw > 256 || th > 256 || tw * th > 66051 || tp > 64 || nx * ny > 65535 || budget > (1 << 19)) return best;
    int best_cost = 0x7fffffff, last_bh = 0;
    for (int nb = 1; nb <= 8; nb++) {
        const int bh = lean_div(rows_all + nb - 1, nb);
        if (bh == last_bh) continue;
        last_bh = bh;
        if (bh < (rows_all < 3 ? rows_all : 3)) break;
        int R = th;
        if (lean_plan_bytes(G, tp, upr, cpu, rstep, rows_all, bh, th, resized, walk, th, box) > budget) {
            const int room = budget - lean_plan_bytes(G, tp, upr, cpu, rstep, rows_all, bh, 0, resized, walk, th, box);
            const int per_row = box == LEAN_BOX_MASKED ? lean_pc_pitch(G) * 4 + G * 8 + tp * 8 : box ? tp * 8
                                : G * 8 + (walk ? 12 : G * 4) + tp * 8;           // list entries, mask entries, a tile row
            if (room < per_row * LEAN_MIN_CHUNK) continue;
            R = lean_div(room, per_row);
            if (R > th - 1) R = th - 1;
            while (R >= LEAN_MIN_CHUNK && lean_plan_bytes(G, tp, upr, cpu, rstep, rows_all, bh, R, resized, walk, th, box) > budget) R--;   // roundings of the regions
            if (R < LEAN_MIN_CHUNK) continue;
            const int n = lean_div(th + R - 1, R);
            R = lean_div(th + n - 1, n);
        }
        const int nch = R == th ? 1 : lean_div(th + R - 1, R);
        const int nbl = lean_div(rows_all + bh - 1, bh);
        const int cost = nbl * nch * (((bh - 1) * rstep + R) * tp + (box == LEAN_BOX_UNMASKED ? 0 : R * G * (resized ? 16 : 2)) + 64);
        if (cost < best_cost) { best_cost = cost; best.bh = bh; best.R = R; }
        if (nb == 1 && R == th) break;                     // everything at once
    }
    return best;
}

// Accumulators of a block of nu units with NC cells each: the {CC, SI} pair of cell k of unit u is pair k * nu + u, so
// that the lanes of a flush - consecutive units - fall on consecutive pairs (unit-major pairs, 32 bytes apart, met in
// 4 of the banks only: the flushes then made a third of all LDS cycles of a walk conflict cycles).  Pair of the cell in
// tile column c (= o + x), row y, of a stride-1 search: strips of 4 columns run down the rows (unit = strip * rows + row).
__device__ __forceinline__ int lean_strip_cell(int c, int y, int rows, int nu) { return (c & 3) * nu + (c >> 2) * rows + y; }

// the score of a cell from its sums - the F32 formula (matching.cpp:70-72) or matchTemplate's (matching.cpp:152); the
// caller has checked that the cell is visited
__device__ __forceinline__ float lean_score(const lds_u32* acc, int ai, const Norms& nm, bool mt) {
    return mt ? score_mt_n(acc[2 * ai], acc[2 * ai + 1], nm.tn) : score_f32_n(acc[2 * ai], acc[2 * ai + 1], nm.p2f);
}

// One entry of the lists for the NC cells of a unit: cc += <window, P0>, si += <window & M, window & M>.  A tile entry
// holds the byte windows W0 (bytes 0..3 from its position) and W3 (bytes 3..6); W1 and W2 are picked out of the two.
template <int NC, bool MASKED>
__device__ __forceinline__ void lean_step(const lds_u8* wp, unsigned P0, unsigned M, unsigned sel, unsigned (&cc)[NC], unsigned (&si)[NC]) {
    if (NC == 4) {
        const unsigned long long w03 = *reinterpret_cast<const lds_u64*>(wp);
        const unsigned w0 = (unsigned)w03, w3 = (unsigned)(w03 >> 32);
        const unsigned w1 = __builtin_amdgcn_perm(w3, w0, 0x05030201u), w2 = __builtin_amdgcn_perm(w3, w0, 0x06050302u);
        const unsigned m0 = MASKED ? (w0 & M) : w0, m1 = MASKED ? (w1 & M) : w1, m2 = MASKED ? (w2 & M) : w2, m3 = MASKED ? (w3 & M) : w3;
        cc[0] = __builtin_amdgcn_udot4(w0, P0, cc[0], false); si[0] = __builtin_amdgcn_udot4(m0, m0, si[0], false);
        cc[1 % NC] = __builtin_amdgcn_udot4(w1, P0, cc[1 % NC], false); si[1 % NC] = __builtin_amdgcn_udot4(m1, m1, si[1 % NC], false);
        cc[2 % NC] = __builtin_amdgcn_udot4(w2, P0, cc[2 % NC], false); si[2 % NC] = __builtin_amdgcn_udot4(m2, m2, si[2 % NC], false);
        cc[3 % NC] = __builtin_amdgcn_udot4(w3, P0, cc[3 % NC], false); si[3 % NC] = __builtin_amdgcn_udot4(m3, m3, si[3 % NC], false);
    } else {
        // the cell's window starts k = 0..3 bytes into the entry (k = 0 and 3 are stored, 1 and 2 picked): k is the same
        // for every entry of a unit, so the caller passes the entry-aligned base and the byte selector (lean_window_sel)
        const unsigned long long w03 = *reinterpret_cast<const lds_u64*>(wp);
        const unsigned w0 = (unsigned)w03, w3 = (unsigned)(w03 >> 32);
        const unsigned w = __builtin_amdgcn_perm(w3, w0, sel);
        const unsigned m0 = MASKED ? (w & M) : w;
        cc[0] = __builtin_amdgcn_udot4(w, P0, cc[0], false); si[0] = __builtin_amdgcn_udot4(m0, m0, si[0], false);
    }
}

// v_perm_b32 selector of the byte window that starts k = 0..3 bytes into a tile entry {W0, W3}: bytes k .. k + 3 of the
// seven the entry holds (W3's bytes 0..3 are the entry's bytes 3..6, selector values 4..7)
__device__ __forceinline__ unsigned lean_window_sel(unsigned k) {
    const unsigned a = k & 1u ? 0x05030201u : 0x03020100u, b = k & 1u ? 0x07060504u : 0x06050302u;
    return k & 2u ? b : a;
}

// Tasks are dealt slice-major (task = slice * nunits + unit): neighbouring lanes then hold neighbouring units of one
// slice, so an entry read is one broadcast address and the window reads of a lane group fall on 16-byte tile entries an
// odd number of entries apart (distinct banks).  Entries are fetched four ahead of their use; the lists are padded so that reading
// up to four entries past a slice stays inside them.
template <int NC>
__device__ __forceinline__ void lean_accumulate(const lds_u8* tile, const lds_u64* list, const lds_u32* marr, lds_u32* acc,
                                                int nunits, int S, int na, int nb, int nA1, int div_units, int tp, int o, int step) {
    const int lane = lane_here();
    const int ntask = nunits * S;
    const float rnu = __builtin_amdgcn_rcpf((float)nunits), rdu = __builtin_amdgcn_rcpf((float)div_units);
    for (int t0 = 0; t0 < ntask; t0 += SLAM_WG) {
        const int task = min(t0 + lane, ntask - 1);        // surplus lanes repeat the last task and do not flush
        const int slice = (int)(((float)task + 0.5f) * rnu), unit = task - slice * nunits;
        // strips run down the columns (unit = strip * rows + row): neighbouring lanes are tp entries apart, and tp is odd
        const int uq = (int)(((float)unit + 0.5f) * rdu), um = unit - uq * div_units;
        unsigned tb;                                       // byte offset of the unit's first window in the tile
        if (NC == 4) tb = (unsigned)(um * tp + uq) * 8u;
        unsigned sel = 0;
        if (NC != 4) { const int bx = o + um * step; tb = (unsigned)((uq * step) * tp + (bx >> 2)) * 8u; sel = lean_window_sel((unsigned)bx & 3u); }
        const lds_u8* tl = tile + tb;
        unsigned cc[NC], si[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) { cc[k] = 0; si[k] = 0; }
        {   // groups that keep all four pixels
            const lds_u64* ea = list + slice * na;
            unsigned long long q0 = ea[0], q1 = ea[1], q2 = ea[2], q3 = ea[3];
            int e = 0;
            for (; e + 4 <= na; e += 4) {
                const unsigned long long c0 = q0, c1 = q1, c2 = q2, c3 = q3;
                q0 = ea[e + 4]; q1 = ea[e + 5]; q2 = ea[e + 6]; q3 = ea[e + 7];
                lean_step<NC, false>(tl + (unsigned)(c0 >> 32), (unsigned)c0, 0u, sel, cc, si);
                lean_step<NC, false>(tl + (unsigned)(c1 >> 32), (unsigned)c1, 0u, sel, cc, si);
                lean_step<NC, false>(tl + (unsigned)(c2 >> 32), (unsigned)c2, 0u, sel, cc, si);
                lean_step<NC, false>(tl + (unsigned)(c3 >> 32), (unsigned)c3, 0u, sel, cc, si);
            }
            if (e < na) lean_step<NC, false>(tl + (unsigned)(q0 >> 32), (unsigned)q0, 0u, sel, cc, si);
            if (e + 1 < na) lean_step<NC, false>(tl + (unsigned)(q1 >> 32), (unsigned)q1, 0u, sel, cc, si);
            if (e + 2 < na) lean_step<NC, false>(tl + (unsigned)(q2 >> 32), (unsigned)q2, 0u, sel, cc, si);
        }
        {   // partly masked groups (and the padding entries, mask 0)
            const lds_u64* eb = list + nA1 + slice * nb;
            const lds_u32* mb = marr + slice * nb;          // marr: masks of the entries from nA1 on
            unsigned long long q0 = eb[0], q1 = eb[1];
            unsigned m0 = mb[0], m1 = mb[1];
            int e = 0;
            for (; e + 2 <= nb; e += 2) {
                const unsigned long long c0 = q0, c1 = q1;
                const unsigned k0 = m0, k1 = m1;
                q0 = eb[e + 2]; q1 = eb[e + 3]; m0 = mb[e + 2]; m1 = mb[e + 3];
                lean_step<NC, true>(tl + (unsigned)(c0 >> 32), (unsigned)c0, k0, sel, cc, si);
                lean_step<NC, true>(tl + (unsigned)(c1 >> 32), (unsigned)c1, k1, sel, cc, si);
            }
            if (e < nb) lean_step<NC, true>(tl + (unsigned)(q0 >> 32), (unsigned)q0, m0, sel, cc, si);
        }
        if (t0 + lane < ntask) {
#pragma unroll
            for (int k = 0; k < NC; k++) {
                __hip_atomic_fetch_add(acc + (k * nunits + unit) * 2, cc[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(acc + (k * nunits + unit) * 2 + 1, si[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
}

// Numerators of a block of strips under rows u0 .. u0 + Rc - 1 of a template without a mask (the box mode of lean_level:
// no entry lists - `pc` holds the P0 dword of every group of the chunk's rows, row after row).  Tasks are (slice, unit),
// slice-major; a slice takes the chunk's rows s, s + S, ... and walks their groups in order: one window read, two
// v_perm and four v_dot4 per group, the group's offset an immediate.
__device__ __forceinline__ int lean_pick_row_slices(int nunits, int Rc, int G) {
    int best = 1, best_c = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < 6; q++) {
        constexpr int cand[6] = {1, 2, 3, 4, 6, 8};
        const int s = cand[q];
        const int passes = (nunits * s + SLAM_WG - 1) >> 6;
        const int c = passes * (((Rc + s - 1) / s) * (G * 6 + 4) + 20);
        if (c < best_c) { best_c = c; best = s; }
    }
    return best;
}

__device__ __forceinline__ void lean_accumulate_rows(const lds_u8* tile, const lds_u32* pc, lds_u32* acc, int nunits, int S, int Rc, int G,
                                                     int rows, int tp) {
    const int lane = lane_here();
    const int ntask = nunits * S;
    const float rnu = __builtin_amdgcn_rcpf((float)nunits), rdu = __builtin_amdgcn_rcpf((float)rows);
    for (int t0 = 0; t0 < ntask; t0 += SLAM_WG) {
        const int task = min(t0 + lane, ntask - 1);        // surplus lanes repeat the last task and do not flush
        const int slice = (int)(((float)task + 0.5f) * rnu), unit = task - slice * nunits;
        const int uq = (int)(((float)unit + 0.5f) * rdu), um = unit - uq * rows;      // strip, row (unit = strip * rows + row)
        const lds_u8* tl = tile + (unsigned)((um + slice) * tp + uq) * 8u;
        const int Gp = lean_pc_pitch(G);
        const lds_u32* pr = pc + slice * Gp;
        unsigned cc[4] = {0u, 0u, 0u, 0u};
        auto group = [&](unsigned P0, const lds_u8* wp) {
            const unsigned long long w03 = *reinterpret_cast<const lds_u64*>(wp);
            const unsigned w0 = (unsigned)w03, w3 = (unsigned)(w03 >> 32);
            const unsigned w1 = __builtin_amdgcn_perm(w3, w0, 0x05030201u), w2 = __builtin_amdgcn_perm(w3, w0, 0x06050302u);
            cc[0] = __builtin_amdgcn_udot4(w0, P0, cc[0], false);
            cc[1] = __builtin_amdgcn_udot4(w1, P0, cc[1], false);
            cc[2] = __builtin_amdgcn_udot4(w2, P0, cc[2], false);
            cc[3] = __builtin_amdgcn_udot4(w3, P0, cc[3], false);
        };
        for (int u = slice; u < Rc; u += S) {
            int g = 0;
            for (; g + 4 <= G; g += 4) {                   // the P0 dwords of four groups in one read
                const v4u32 p4 = *reinterpret_cast<const lds_v4u32*>(pr + g);
                group(p4.x, tl + g * 8); group(p4.y, tl + g * 8 + 8); group(p4.z, tl + g * 8 + 16); group(p4.w, tl + g * 8 + 24);
            }
            if (G & 2) {
                const unsigned long long p2 = *reinterpret_cast<const lds_u64*>(pr + g);
                group((unsigned)p2, tl + g * 8); group((unsigned)(p2 >> 32), tl + g * 8 + 8);
                g += 2;
            }
            if (G & 1) group(pr[g], tl + g * 8);
            tl += (unsigned)(S * tp) * 8u;
            pr += S * Gp;
        }
        if (t0 + lane < ntask) {
#pragma unroll
            for (int k = 0; k < 4; k++)
                __hip_atomic_fetch_add(acc + (k * nunits + unit) * 2, cc[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// The same numerators with two cell rows per unit: the window read of tile row 2m + t serves template row t for cell
// row 2m and template row t - 1 for cell row 2m + 1 - ten VALU and one and a half LDS instructions per eight dots where
// a unit of one row takes twelve and two and a half.  Tasks are (slice, strip, row pair); a slice takes the template
// rows [a, b) of the chunk and with them the tile rows a .. b: on the first only the upper cell row counts, on the last
// only the lower one (the other gets the all-zero row `zrow` of P0 dwords).  An odd last row pairs with nothing.
__device__ __forceinline__ void lean_accumulate_rows2(const lds_u8* tile, const lds_u32* pc, const lds_u32* zrow, lds_u32* acc, int nunits,
                                                      int Rc, int G, int rows, int tp) {
    const int lane = lane_here();
    const int npairs = (rows + 1) >> 1, upr = nunits / rows;                       // (wave-uniform: scalar division)
    const int nu2 = upr * npairs;
    int S = nu2 >= 33 ? 1 : min(min(LEAN_MAX_S, SLAM_WG / nu2), max(Rc >> 2, 1));
    const int per = (Rc + S - 1) / S;                                              // template rows per slice
    S = (Rc + per - 1) / per;
    const int ntask = nu2 * S;
    const int Gp = lean_pc_pitch(G);
    const float rnu = __builtin_amdgcn_rcpf((float)nu2), rdp = __builtin_amdgcn_rcpf((float)npairs);
    for (int t0 = 0; t0 < ntask; t0 += SLAM_WG) {
        const int task = min(t0 + lane, ntask - 1);        // surplus lanes repeat the last task and do not flush
        const int slice = (int)(((float)task + 0.5f) * rnu), unit2 = task - slice * nu2;
        const int uq = (int)(((float)unit2 + 0.5f) * rdp), m = unit2 - uq * npairs;  // strip, row pair
        const bool has1 = 2 * m + 1 < rows;
        const int a = slice * per, b = min(a + per, Rc);
        const lds_u8* tl = tile + (unsigned)((2 * m + a) * tp + uq) * 8u;
        const lds_u32* pr = pc + a * Gp;
        unsigned c0[4] = {0u, 0u, 0u, 0u}, c1[4] = {0u, 0u, 0u, 0u};
        auto group = [&](unsigned P0, unsigned P1, const lds_u8* wp) {
            const unsigned long long w03 = *reinterpret_cast<const lds_u64*>(wp);
            const unsigned w0 = (unsigned)w03, w3 = (unsigned)(w03 >> 32);
            const unsigned w1 = __builtin_amdgcn_perm(w3, w0, 0x05030201u), w2 = __builtin_amdgcn_perm(w3, w0, 0x06050302u);
            c0[0] = __builtin_amdgcn_udot4(w0, P0, c0[0], false); c1[0] = __builtin_amdgcn_udot4(w0, P1, c1[0], false);
            c0[1] = __builtin_amdgcn_udot4(w1, P0, c0[1], false); c1[1] = __builtin_amdgcn_udot4(w1, P1, c1[1], false);
            c0[2] = __builtin_amdgcn_udot4(w2, P0, c0[2], false); c1[2] = __builtin_amdgcn_udot4(w2, P1, c1[2], false);
            c0[3] = __builtin_amdgcn_udot4(w3, P0, c0[3], false); c1[3] = __builtin_amdgcn_udot4(w3, P1, c1[3], false);
        };
        const int tend = has1 ? b : b - 1;                 // (an odd last row: the tile holds no row under its missing partner)
        for (int t = a; t <= tend; t++) {
            const lds_u32* p0 = t < b ? pr : zrow;
            const lds_u32* p1 = (t > a && has1) ? pr - Gp : zrow;
            int g = 0;
            for (; g + 4 <= G; g += 4) {                   // the P0 dwords of four groups in one read, for either cell row
                const v4u32 q0 = *reinterpret_cast<const lds_v4u32*>(p0 + g), q1 = *reinterpret_cast<const lds_v4u32*>(p1 + g);
                group(q0.x, q1.x, tl + g * 8); group(q0.y, q1.y, tl + g * 8 + 8); group(q0.z, q1.z, tl + g * 8 + 16); group(q0.w, q1.w, tl + g * 8 + 24);
            }
            if (G & 2) {
                const unsigned long long q0 = *reinterpret_cast<const lds_u64*>(p0 + g), q1 = *reinterpret_cast<const lds_u64*>(p1 + g);
                group((unsigned)q0, (unsigned)q1, tl + g * 8); group((unsigned)(q0 >> 32), (unsigned)(q1 >> 32), tl + g * 8 + 8);
                g += 2;
            }
            if (G & 1) group(p0[g], p1[g], tl + g * 8);
            tl += (unsigned)tp * 8u;
            pr += Gp;
        }
        if (t0 + lane < ntask) {
            const int unit = uq * rows + 2 * m;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                __hip_atomic_fetch_add(acc + (k * nunits + unit) * 2, c0[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (has1) __hip_atomic_fetch_add(acc + (k * nunits + unit + 1) * 2, c1[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
}

// What the box sums count too much under a masked template: for every group that drops pixels, the energy of the dropped
// ones - si -= <window & C, window & C> with C the group's complement mask inside the template - over a list of
// {C, tile offset} entries padded with null entries to S slices of equal length (tasks as in lean_accumulate).
__device__ __forceinline__ void lean_accumulate_dropped(const lds_u8* tile, const lds_u64* clist, lds_u32* acc, int nunits, int nc, int rows, int tp) {
    if (nc == 0) return;
    const int lane = lane_here();
    int S = 1, per = nc, best_c = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < 6; q++) {
        constexpr int cand[6] = {1, 2, 3, 4, 6, 8};        // at most LEAN_MAX_S
        const int s = cand[q];
        const int n = (nc + s - 1) / s;
        const int c = ((nunits * s + SLAM_WG - 1) >> 6) * (n * 12 + 20);
        if (c < best_c) { best_c = c; S = s; per = n; }
    }
    const int ntask = nunits * S;
    const float rnu = __builtin_amdgcn_rcpf((float)nunits), rdu = __builtin_amdgcn_rcpf((float)rows);
    for (int t0 = 0; t0 < ntask; t0 += SLAM_WG) {
        const int task = min(t0 + lane, ntask - 1);        // surplus lanes repeat the last task and do not flush
        const int slice = (int)(((float)task + 0.5f) * rnu), unit = task - slice * nunits;
        const int uq = (int)(((float)unit + 0.5f) * rdu), um = unit - uq * rows;
        const lds_u8* tl = tile + (unsigned)(um * tp + uq) * 8u;
        const lds_u64* e = clist + slice * per;
        unsigned si[4] = {0u, 0u, 0u, 0u};
        unsigned long long q0 = e[0], q1 = e[1];
        for (int j = 0; j < per; j += 2) {
            const unsigned long long c0 = q0, c1 = q1;
            q0 = e[j + 2]; q1 = e[j + 3];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const unsigned long long c = h ? c1 : c0;
                if (h && j + 1 >= per) break;
                const unsigned cm = (unsigned)c;
                const unsigned long long w03 = *reinterpret_cast<const lds_u64*>(tl + (unsigned)(c >> 32));
                const unsigned w0 = (unsigned)w03, w3 = (unsigned)(w03 >> 32);
                const unsigned w1 = __builtin_amdgcn_perm(w3, w0, 0x05030201u), w2 = __builtin_amdgcn_perm(w3, w0, 0x06050302u);
                const unsigned m0 = w0 & cm, m1 = w1 & cm, m2 = w2 & cm, m3 = w3 & cm;
                si[0] = __builtin_amdgcn_udot4(m0, m0, si[0], false); si[1] = __builtin_amdgcn_udot4(m1, m1, si[1], false);
                si[2] = __builtin_amdgcn_udot4(m2, m2, si[2], false); si[3] = __builtin_amdgcn_udot4(m3, m3, si[3], false);
            }
        }
        if (t0 + lane < ntask) {
#pragma unroll
            for (int k = 0; k < 4; k++)
                __hip_atomic_fetch_add(acc + (k * nunits + unit) * 2 + 1, 0u - si[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// One search.  SEARCH = false: a level of a point walk - a patch mask is present, scores by the F32 formula, and the caller
// has checked (lean_need) that the whole window fits the LDS at once.  SEARCH = true: a single matchFeatureInArea call -
// the patch mask may be absent, the formula is L.mt's, and a window whose sums do not fit the LDS at once is searched
// in blocks of rows (the lists are built once; tile, sums and scores per block; the first maximum in row-major order
// carries over from block to block).  `refine_*`: also report the sub-pixel offset of the peak (the level that ends a
// walk does; single searches always do).  Returns ok = false when the search is not covered here (nothing has been
// written anywhere).
template <bool SEARCH, bool CHUNKS = false>
__device__ __forceinline__ LeanOut lean_level(const LeanLevel& L, lds_u8* lds, int lds_total, bool refine_always, double refine_below STAMP_ARG) {
    LeanOut out;
    out.ok = false; out.x = 0; out.y = 0; out.response = 0.f; out.dx = 0.f; out.dy = 0.f;
    const int lane = lane_here();
    const bool mt = SEARCH && L.mt;
    const bool has_mask = !SEARCH || L.mask != nullptr;
    const int G = (L.tw + 3) >> 2, NG = L.th * G;
    // Column phase of the window in the tile.  The plain builds stage whole image dwords, so cell x's window starts o + x
    // bytes into a tile row and a strip may hold cells of two image dwords; the chunk-capable builds shift the bytes
    // while staging (the tile starts at the window's first byte), which saves the strip a misaligned window needs extra.
    const int osrc = L.x0 & 3;
    const int o = CHUNKS ? 0 : osrc;
    const bool strips = L.step == 1;
    const int nx = strips ? L.aw : small_div_u(L.aw + L.step - 1, L.step), ny = strips ? L.ah : small_div_u(L.ah + L.step - 1, L.step);
    const int cols = (o + L.aw + 3) >> 2;                  // dword columns that hold a cell's first pixel
    const int tp = (cols + G - 1) | 1;                     // window positions per tile row, odd: see lean_accumulate
    const int rows_all = strips ? L.ah : ny;               // rows of units: cell rows at stride 1, lattice rows otherwise
    const int rstep = strips ? 1 : L.step;                 // image rows from one unit row to the next
    const int upr = strips ? cols : nx;                    // units per row
    const int cpu = strips ? 4 : 1;                        // cells per unit
    // Without a patch mask a cell's window energy is the sum of the squared image over the template's rectangle, whatever
    // the template holds: the chunk-capable builds take it from box sums - every lane keeps the energy of two pixel
    // columns over the th rows under the current cell row, slides it down row by row (and on from block to block), and
    // a prefix sum over the lanes gives every cell's tw columns as a difference.  The numerators then need no entry
    // lists either: every group counts with all four bytes, so a task walks template rows and groups in order, the
    // P0 dwords of the whole template staying resident.
    const int npix = L.aw + L.tw - 1;                       // pixel columns under the window
    // Under a masked template (walks, from a size on where it pays: lean_box_mode) the box sums count the dropped pixels
    // too: their energy is taken off again by dots over the groups that drop any (lean_accumulate_dropped), a short list
    // for the masks of warped quadrilaterals; the numerators run in row order as well, over the P0 = patch & mask dwords.
    const int boxmode = CHUNKS ? lean_box_mode(SEARCH, has_mask, L.step, L.resized, L.tw, L.th, L.aw) : LEAN_BOX_NONE;
    const bool box = boxmode != LEAN_BOX_NONE, boxm = boxmode == LEAN_BOX_MASKED;
    int R = L.th, nchunk = 1;                              // template rows per chunk, chunks (lean_plan)
    int bh = rows_all, nblk = 1;                           // unit rows per block, blocks
    if (CHUNKS) {
        const LeanPlan pl = lean_plan(L.tw, L.th, L.aw, L.ah, L.step, nx, ny, L.resized, !SEARCH, lds_total, boxmode);
        bh = uni_i32(pl.bh); R = uni_i32(pl.R);
        if (bh <= 0) { LEAN_REJECT(100); return out; }
        nblk = small_div_u(rows_all + bh - 1, bh);
        if (R < L.th) nchunk = small_div_u(L.th + R - 1, R);
    } else if (!SEARCH) {
        if (lean_need(L.tw, L.th, L.aw, L.ah, L.step, nx, ny, L.resized) > lds_total) { LEAN_REJECT(100); return out; }
    }
    const bool chunked = CHUNKS && nchunk > 1;
    const int NGc = R * G;                                 // groups the lists hold at most
    const int cap = lean_cap(NGc);                         // room for the padding entries and for reads a few entries ahead
    // (box: no lists - the P0 dwords of the whole template, or of a chunk and its list of groups that drop pixels, take their place)
    const int list_bytes = boxm ? ((R + 1) * lean_pc_pitch(G) * 4 + 15) & ~15 : box ? ((L.th + 1) * lean_pc_pitch(G) * 4 + 15) & ~15 : cap * 8;
    const int marr_bytes = boxm ? lean_ccap(NGc) * 8 : box ? 0 : (lean_mcap(NGc, R, !SEARCH) * 4 + 15) & ~15;
    if (SEARCH && !CHUNKS) {
        // sums fit 32 bits up to 66051 template pixels; a lane's window positions index the tile with 16 bits to spare
        if (L.tw > 256 || L.th > 256 || L.tw * L.th > 66051 || tp > 64) { LEAN_REJECT(101); return out; }
        const int raw_bytes = NG * 8 + (L.resized ? (G * 4 + L.th) * 8 : 0);
        const int fixed = list_bytes + marr_bytes + 32;    // + rounding of the two regions around them
        const int per_row = upr * cpu * 8 + rstep * tp * 8;
        const int room = lds_total - fixed - (L.th - rstep) * tp * 8;
        int r = room > 0 ? small_div_u(room, per_row) : 0;
        const int room2 = lds_total - fixed - raw_bytes;   // the template is staged where the tile goes afterwards
        const int r2 = room2 > 0 ? small_div_u(room2, upr * cpu * 8) : 0;
        r = min(min(r, r2), rows_all);
        if (r < min(3, rows_all) || lds_total > (1 << 19)) { LEAN_REJECT(102); return out; }      // the fit needs three rows at once
        nblk = small_div_u(rows_all + r - 1, r);
        bh = small_div_u(rows_all + nblk - 1, nblk);
    }
    // tile rows are fetched as aligned dwords: a needed byte must not sit in a partial last dword of the image row
    if (L.x0 + L.aw + L.tw - 1 > (L.w & ~3)) { LEAN_REJECT(2); return out; }
    // room for the three rows around the peak too, when blocks are lower than that (they fit: r above is at least 3)
    const int acc_bytes = (max(bh, min(3, rows_all)) * upr * cpu * 8 + 15) & ~15;
    lds_u32* acc = reinterpret_cast<lds_u32*>(lds);
    lds_u64* list = reinterpret_cast<lds_u64*>(lds + acc_bytes);
    lds_u32* marr = reinterpret_cast<lds_u32*>(lds + acc_bytes + list_bytes);       // masks of list entries nA1, nA1 + 1, ...: the all-kept part has none
    lds_u8* tile = lds + acc_bytes + list_bytes + marr_bytes;
    // {P0, M} per group in (row, group) order, until the tile is staged (box: of the whole template, from where the P0 dwords stay)
    lds_u64* raw = box ? reinterpret_cast<lds_u64*>(list) : reinterpret_cast<lds_u64*>(tile);
    lds_u64* taps = raw + (box ? NG : NGc);
    lds_u32* pc = reinterpret_cast<lds_u32*>(list);        // box: P0 of every group of the template (masked: of the chunk)
    lds_u64* clist = reinterpret_cast<lds_u64*>(marr);     // masked box: {complement mask, tile offset} of the groups that drop pixels
    int ncomp = 0;                                         //   and their number in the current chunk
    const lds_u32* zrow = pc + (boxm ? R : L.th) * lean_pc_pitch(G);     // box: a row of zero P0 dwords behind the stored rows

    STAMP(50);
    // A template of at most 32 groups (the coarse levels of a walk) is not sorted into the two lists: compacting it
    // costs more instructions than masking every group of so short a list does.
    // (decided here, from sizes alone: the direct search of a few lattice cells further down needs no entry lists either)
    const int dshift = nx * ny <= 4 ? 4 : nx * ny <= 8 ? 3 : nx * ny <= 16 ? 2 : 1;       // 2^dshift lanes per cell
    // (it reads the staged pairs of the whole template: not where the template is taken in chunks)
    const bool direct = !SEARCH && !strips && nx * ny <= 32 && ((L.th + (1 << dshift) - 1) >> dshift) * G <= 16 &&
                        L.x0 + (nx - 1) * L.step + 4 * G <= L.w && !chunked && nblk == 1;
    const bool tiny = direct || (NG <= 32 && !chunked);
    unsigned sp_part = 0;
    const float rG = __builtin_amdgcn_rcpf((float)G);
    Norms nm;
    nm.p2f = 0.f; nm.tn = 0.0;
    Best b = {-FLT_MAX, INT_MAX};
    int S = 1, na = 0, nb = 0, nA1 = 0;                     // slices; all-kept and masked entries per slice; first masked entry
    const float rnx = __builtin_amdgcn_rcpf((float)nx);
    int r0 = 0, rows_b = rows_all;                          // the block of cell rows whose sums the accumulators hold
    int u0 = 0, Rc = L.th;                                  // the template rows the lists and the tile hold
    int lx = 0, ly = 0;
    float bv = 0.f;
    auto own = [&](int y) { return !CHUNKS || (y >= L.own0 && y < L.own1); };
    unsigned box_e0 = 0, box_e1 = 0;                        // energies of pixel columns 2 * lane and 2 * lane + 1
    int box_y = -2;                                         // the cell row they belong to
    // Template rows -> {P0, M} pairs -> entry lists (or, for a handful of lattice cells, the sums themselves).  The plain
    // builds run this once, ahead of the blocks; the chunk-capable ones whenever the template rows change.
    // Returns 0: lists in place; 1: every cell has been scored (the direct search); -1: not covered.
    auto stage = [&](int blk, int ch, int nunits) -> int {
        __syncthreads();                                   // the previous level's, block's or chunk's readers of this LDS are done
        // ---- template: {P0, M} per group, |P0|^2, counts of all-kept and partly kept groups ------------------------
        const int ut = box && !boxm ? 0 : u0, Rt = box && !boxm ? L.th : Rc; // the template rows staged here
        const int NGb = Rt * G;                            // groups of these template rows
        int nA = 0, nB = 0;
        if (CHUNKS && boxm) {
            // masked box sums (stored template): P0 straight into the chunk's row-ordered array, the groups that drop
            // pixels inside the template onto the complement list
            const int ccap = lean_ccap(NGc);
            const int Gp = lean_pc_pitch(G);
            int baseC = 0;
#pragma unroll 4
            for (int i0 = 0; i0 < NGb; i0 += SLAM_WG) {             // (unrolled: the loads of four passes in flight at once)
                const int i = i0 + lane;
                unsigned cm = 0;
                const int u = (int)(((float)min(i, NGb - 1) + 0.5f) * rG), gq = min(i, NGb - 1) - u * G;
                if (i < NGb) {
                    const long long so = (long long)(ut + u) * L.srcw + gq * 4;
                    const unsigned P = load4(L.patch + so, L.limit);
                    const unsigned Mraw = load4(L.mask + so, L.limit);
                    const int nbk = min(4, L.tw - gq * 4);
                    const unsigned keep = nbk >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nbk)) - 1u);
                    const unsigned nz = (((Mraw & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | Mraw) & 0x80808080u;   // per byte: non-zero -> 0x80
                    const unsigned M = ((nz >> 7) * 0xFFu) & keep;
                    const unsigned P0 = P & M;
                    cm = ~M & keep;
                    pc[u * Gp + gq] = P0;
                    if (blk == 0) sp_part += __builtin_amdgcn_udot4(P0, P0, 0u, false);
                }
                const unsigned long long bc = __ballot(cm != 0u);
                const int pos = baseC + __builtin_popcountll(bc & ((1ull << lane) - 1ull));
                if (cm != 0u && pos < ccap) clist[pos] = ((unsigned long long)((unsigned)(u * tp + gq) * 8u) << 32) | cm;
                baseC += __builtin_popcountll(bc);
            }
            // a mask that drops pixels in more groups than a warped quadrilateral does (the plain build's list has room for those only)
            if (baseC + LEAN_MAX_S + 3 > ccap) { LEAN_REJECT(4); return -1; }
            for (int p2 = baseC + lane; p2 < baseC + LEAN_MAX_S; p2 += SLAM_WG) clist[p2] = 0ull;       // padding: adds nothing
            for (int p2 = lane; p2 < Gp; p2 += SLAM_WG) pc[R * Gp + p2] = 0u;
            ncomp = baseC;
            if (blk == 0 && ch == nchunk - 1) nm = make_norms(wave_sum_u32(sp_part));       // |P0|^2 of the whole template
            STAMP(52);
            return 0;
        }
        if (!L.resized) {
        for (int i0 = 0; i0 < NGb; i0 += SLAM_WG) {
            const int i = i0 + lane;
            unsigned P0 = 0, M = 0;
            if (i < NGb) {
                const int u = (int)(((float)i + 0.5f) * rG), gq = i - u * G;
                const long long so = (long long)(ut + u) * L.srcw + gq * 4;
                const unsigned P = load4(L.patch + so, L.limit);
                const unsigned Mraw = has_mask ? load4(L.mask + so, L.limit) : 0x01010101u;
                const int nb = min(4, L.tw - gq * 4);
                const unsigned keep = nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u);
                const unsigned nz = (((Mraw & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | Mraw) & 0x80808080u;   // per byte: non-zero -> 0x80
                M = ((nz >> 7) * 0xFFu) & keep;
                P0 = P & M;
                raw[i] = ((unsigned long long)M << 32) | P0;
                if (!CHUNKS || blk == 0) sp_part += __builtin_amdgcn_udot4(P0, P0, 0u, false);
            }
            nA += __builtin_popcountll(__ballot(M == 0xFFFFFFFFu));
            nB += __builtin_popcountll(__ballot(M != 0u && M != 0xFFFFFFFFu));
        }
    } else {
        // cv::resize of template and mask on the fly (feature.cpp:130-131): tap tables first, then one pixel per lane
        // and step - 4 source bytes of the patch, 4 of the mask, fixed-point blend - into the raw bytes
        const int wpx = G * 4, ntap = wpx + Rt;
        for (int i = lane; i < ntap; i += SLAM_WG) {
            int so, c0, c1;
            if (i < wpx) resize_tap(min(i, L.tw - 1), L.scx, L.srcw, true, so, c0, c1);
            else resize_tap(ut + i - wpx, L.scy, L.srch, false, so, c0, c1);
            taps[i] = ((unsigned long long)(((unsigned)c0 & 0xffffu) | ((unsigned)c1 << 16)) << 32) | (unsigned)so;
        }
        __syncthreads();
        lds_u8* rb = reinterpret_cast<lds_u8*>(raw);
        const float rw = __builtin_amdgcn_rcpf((float)wpx);
        const int n = Rt * wpx;
        for (int i0 = 0; i0 < n; i0 += SLAM_WG) {
            const int ii = i0 + lane;
            const bool live = ii < n;
            const int i = min(ii, n - 1);
            const int u = (int)(((float)i + 0.5f) * rw), v = i - u * wpx;
            const unsigned long long txq = taps[min(v, L.tw - 1)], tyq = taps[wpx + u];
            const bool pad = v >= L.tw;
            const int eo = (u * G + (v >> 2)) * 8 + (v & 3);
            const ResizeSrc spx = resize_fetch(L.patch, L.srcw, L.srch, (int)(unsigned)txq, (int)(unsigned)tyq);
            const ResizeSrc smx = has_mask ? resize_fetch(L.mask, L.srcw, L.srch, (int)(unsigned)txq, (int)(unsigned)tyq) : spx;
            const int a0 = (short)(txq >> 32), a1 = (short)(txq >> 48), b0 = (short)(tyq >> 32), b1 = (short)(tyq >> 48);
            unsigned m = has_mask ? (unsigned)(resize_blend(smx, a0, a1, b0, b1, true) != 0) : 1u;
            unsigned p = (unsigned)resize_blend(spx, a0, a1, b0, b1, false) * m;
            if (pad) { p = 0; m = 0; }
            if (live) {
                if (!CHUNKS || blk == 0) sp_part += p * p;
                rb[eo] = (unsigned char)p;
                rb[eo + 4] = m ? 0xFF : 0x00;
            }
        }
        __syncthreads();
        if (!tiny)
            for (int i0 = 0; i0 < NGb; i0 += SLAM_WG) {
                const int i = i0 + lane;
                const unsigned M = i < NGb ? (unsigned)(raw[i] >> 32) : 0u;
                nA += __builtin_popcountll(__ballot(M == 0xFFFFFFFFu));
                nB += __builtin_popcountll(__ballot(M != 0u && M != 0xFFFFFFFFu));
            }
    }
    if (tiny) { nA = 0; nB = NG; }
    if (!CHUNKS || box || (blk == 0 && ch == nchunk - 1)) nm = make_norms(wave_sum_u32(sp_part));       // |P0|^2 of the whole template
    STAMP(51);
    if (CHUNKS && box) {
        // the P0 dwords move to the front of the staged pairs (a lane's write lands on bytes read in this pass or before)
        __syncthreads();
        const int Gp = lean_pc_pitch(G);
        for (int i0 = 0; i0 < NG; i0 += SLAM_WG) {
            const int i = i0 + lane;
            const unsigned P0 = i < NG ? (unsigned)raw[i] : 0u;
            __syncthreads();
            if (i < NG) { const int u = (int)(((float)i + 0.5f) * rG), gq = i - u * G; pc[u * Gp + gq] = P0; }
        }
        for (int p2 = lane; p2 < Gp; p2 += SLAM_WG) pc[L.th * Gp + p2] = 0u;      // (behind every staged pair that was still to be read)
    } else {

    // ---- a handful of lattice cells under a small template (the coarse levels of a walk) --------------------------------
    // At most 32 cells and at most 16 template groups per lane: tile, lists and LDS accumulators would cost more than the sums themselves.
    // Every cell gets P = 2^k lanes that split the template rows; a lane reads its windows straight from the image
    // (unaligned dwords, served by the caches: the cells overlap) against the {P0, M} pairs staged above, the partial
    // sums of a cell meet by DPP butterflies, and the cell's first lane scores it.  Same integer sums, same formula, same
    // argmax rule as below.  (The last group's dword may reach up to 3 bytes past the template: it must stay inside
    // the image row.  Off the lattice there is no sub-pixel fit - the neighbours are unvisited - so none is needed here.)
    LeanSlices lsl = direct ? LeanSlices{1, 0, 0} : lean_pick_slices(chunked ? nunits : bh * upr, nA, nB, strips);
    // (the chunk-capable builds pick slices per chunk and block: where the padding of the masked part to a multiple of
    //  the slice count is what overflows the mask array, one slice - no padding - still fits)
    if (CHUNKS && !SEARCH && !direct && lsl.nb * lsl.S + 2 > lean_mcap(NGc, R, true)) lsl = LeanSlices{1, nA, nB};
    S = lsl.S; na = lsl.na; nA1 = na * S;                   // all-kept entries per slice; the rest of them joins the masked part
    nb = lsl.nb;
    const int nB1 = nb * S;
    // a mask with more partly kept groups than a warped quadrilateral has (never true for single searches)
    if (!SEARCH && !direct && nB1 + 2 > lean_mcap(NGc, R, true)) { LEAN_REJECT(4); return -1; }
    __syncthreads();                                        // raw is complete (direct path: written above by other lanes)
    if (direct) {
        const int ncell = nx * ny, pshift = dshift;
        const int cell = lane >> pshift, part = lane & ((1 << pshift) - 1);
        const bool live = cell < ncell;
        const int cy = live ? (int)(((float)cell + 0.5f) * __builtin_amdgcn_rcpf((float)nx)) : 0, cx = live ? cell - cy * nx : 0;
        const int x = cx * L.step, y = cy * L.step;
        unsigned cc = 0, si = 0;
        if (live) {
            const uint8_t* cellp = L.im + (unsigned long long)(unsigned)(L.y0 + y) * L.ipitch + (unsigned)(L.x0 + x);
            for (int u = part; u < L.th; u += 1 << pshift) {
                const uint8_t* row = cellp + (unsigned)u * L.ipitch;
                const lds_u64* pr = raw + u * G;
                for (int gq = 0; gq < G; gq++) {
                    unsigned w;
                    __builtin_memcpy(&w, row + 4 * gq, 4);
                    const unsigned long long pm = pr[gq];
                    const unsigned wm = w & (unsigned)(pm >> 32);
                    cc = __builtin_amdgcn_udot4(w, (unsigned)pm, cc, false);
                    si = __builtin_amdgcn_udot4(wm, wm, si, false);
                }
            }
        }
        // the parts of a cell are 2, 4, 8 or 16 neighbouring lanes inside a row of 16: pairs, quads, half rows, rows
        cc += (unsigned)__builtin_amdgcn_update_dpp(0, (int)cc, 0xB1, 0xf, 0xf, false);
        si += (unsigned)__builtin_amdgcn_update_dpp(0, (int)si, 0xB1, 0xf, 0xf, false);
        if (pshift >= 2) {
            cc += (unsigned)__builtin_amdgcn_update_dpp(0, (int)cc, 0x4E, 0xf, 0xf, false);
            si += (unsigned)__builtin_amdgcn_update_dpp(0, (int)si, 0x4E, 0xf, 0xf, false);
        }
        if (pshift >= 3) {
            cc += (unsigned)__builtin_amdgcn_update_dpp(0, (int)cc, 0x141, 0xf, 0xf, false);      // row_half_mirror
            si += (unsigned)__builtin_amdgcn_update_dpp(0, (int)si, 0x141, 0xf, 0xf, false);
        }
        if (pshift >= 4) {
            cc += (unsigned)__builtin_amdgcn_update_dpp(0, (int)cc, 0x140, 0xf, 0xf, false);      // row_mirror
            si += (unsigned)__builtin_amdgcn_update_dpp(0, (int)si, 0x140, 0xf, 0xf, false);
        }
        if (live && part == 0 && !(L.amask && L.amask[y * L.aw + x] == 0))
            best_update(b, score_f32_n(cc, si, nm.p2f), (y << 16) | x);
        return 1;                                           // straight to the argmax
    } else if (tiny) {                                      // every group a masked entry, in template order
        if (lane < NG) {
            const unsigned long long r = raw[lane];
            const int u = (int)(((float)lane + 0.5f) * rG), gq = lane - u * G;
            list[lane] = ((unsigned long long)((unsigned)(u * tp + gq) * 8u) << 32) | (unsigned)r;
            marr[lane] = (unsigned)(r >> 32);               // nA1 = 0 here
        }
        for (int p = NG + lane; p < nA1 + nB1; p += SLAM_WG) { list[p] = 0ull; marr[p] = 0u; }
    } else {
        int baseA = 0, baseB = nA;
        for (int i0 = 0; i0 < NGb; i0 += SLAM_WG) {
            const int i = i0 + lane;
            unsigned P0 = 0, M = 0;
            if (i < NGb) { const unsigned long long r = raw[i]; P0 = (unsigned)r; M = (unsigned)(r >> 32); }
            const bool isA = M == 0xFFFFFFFFu, isB = M != 0u && !isA;
            const unsigned long long ba = __ballot(isA), bb = __ballot(isB);
            const int u = (int)(((float)i + 0.5f) * rG), gq = i - u * G;
            const unsigned long long en = ((unsigned long long)((unsigned)(u * tp + gq) * 8u) << 32) | P0;
            const unsigned long long below = (1ull << lane) - 1ull;
            // the raw region is read above and the lists written below live in different LDS ranges
            if (isA) { const int p = baseA + __builtin_popcountll(ba & below); list[p] = en; if (p >= nA1) marr[p - nA1] = M; }
            if (isB) { const int p = baseB + __builtin_popcountll(bb & below); list[p] = en; marr[p - nA1] = M; }
            baseA += __builtin_popcountll(ba); baseB += __builtin_popcountll(bb);
        }
        for (int p = nA + nB + lane; p < nA1 + nB1; p += SLAM_WG) { list[p] = 0ull; marr[p - nA1] = 0u; }   // padding: adds nothing
    }
    }
    STAMP(52);
        return 0;
    };
    int blk0 = 0;
    if (!CHUNKS) {
        const int st = stage(0, 0, 0);
        if (st < 0) return out;
        if (st == 1) blk0 = nblk;
    }
    // ---- blocks of cell rows, and within a block chunks of template rows: lists where they change, tile, sums; scores
    //      when a block's sums are complete; after the last block the argmax and, for the fit, the rows around it
    for (int blk = blk0; blk <= nblk; blk++) {
        if (blk == nblk) {
            // Scores kept by best_update are >= 0 (never NaN): as unsigned integers their bits order like the values.  The
            // wave's maximum first, then the smallest cell index among the lanes that hold it (cv::minMaxLoc: first
            // maximum in row-major order, NaN never wins).
            const unsigned bkey = wave_max_u32(b.idx != INT_MAX ? __float_as_uint(b.v) + 1u : 0u);
            const int bidx = (int)wave_min_u32((b.idx != INT_MAX && __float_as_uint(b.v) + 1u == bkey) ? (unsigned)b.idx : 0xFFFFFFFFu);
            bv = __uint_as_float(bkey - 1u);
            STAMP(55);
            // Visited cells score >= 0 or NaN.  If none scored (every visited cell NaN, or no cell visited), the maximum is
            // a -1 of the untouched score map or there is none at all: the general kernel knows those cases.
            if (bkey == 0u) { LEAN_REJECT(3); return out; }
            lx = bidx & 0xffff; ly = bidx >> 16;
            out.ok = true;
            out.x = lx + L.x0 + L.tw / 2; out.y = ly + L.y0 + L.th / 2;      // + areaRect origin (matching.cpp:225-226)
            out.response = bv;
            if (!(refine_always || (double)bv < refine_below)) return out;
            if (!strips) return out;                       // off the lattice the four neighbours are unvisited: no fit
            if (nblk == 1) break;                          // the sums of the only block are still in place
            r0 = max(ly - 1, 0); rows_b = min(ly + 1, L.ah - 1) - r0 + 1;
        } else {
            r0 = blk * bh; rows_b = min(bh, rows_all - r0);
        }
        const int nunits = rows_b * upr;
      int ch = 0;
      do {                                                  // (one pass unless the template is taken in chunks)
        if (chunked) { u0 = ch * R; Rc = min(R, L.th - u0); }
      if (CHUNKS && ((blk == 0 && ch == 0) || (chunked && (!box || boxm)))) {
        const int st = stage(blk, ch, nunits);
        if (st < 0) { out.ok = false; return out; }        // (also from the rows around the peak: the position found so far is dropped with it)
        if (st == 1) break;
      }
        __syncthreads();                                    // lists complete; the previous block's readers of tile and sums done
        if (ch == 0)
            for (int i = lane; i < nunits * cpu * 2; i += SLAM_WG) acc[i] = 0;
        if (CHUNKS && box && ch == 0) {
            lds_u32* inc = reinterpret_cast<lds_u32*>(tile);    // inclusive prefix sums of the column energies (the tile is staged afterwards)
            const bool in0 = 2 * lane < npix, in1 = 2 * lane + 1 < npix;
            const uint8_t* c0 = L.im + (unsigned long long)(unsigned)L.y0 * L.ipitch + (unsigned)(L.x0 + min(2 * lane, npix - 1));
            const uint8_t* c1 = L.im + (unsigned long long)(unsigned)L.y0 * L.ipitch + (unsigned)(L.x0 + min(2 * lane + 1, npix - 1));
            for (int cy = 0; cy < rows_b; cy++) {
                const int y = r0 + cy;
                if (box_y == y - 1) {                       // one row down: the row above leaves, the row below enters
                    const unsigned a0 = c0[(unsigned)(y - 1) * L.ipitch], a1 = c1[(unsigned)(y - 1) * L.ipitch];
                    const unsigned b0 = c0[(unsigned)(y - 1 + L.th) * L.ipitch], b1 = c1[(unsigned)(y - 1 + L.th) * L.ipitch];
                    box_e0 += b0 * b0 - a0 * a0; box_e1 += b1 * b1 - a1 * a1;
                } else if (box_y != y) {                    // from scratch: the first block, or the rows around the peak
                    box_e0 = 0; box_e1 = 0;
#pragma unroll 8
                    for (int r = 0; r < L.th; r++) {
                        const unsigned a0 = c0[(unsigned)(y + r) * L.ipitch], a1 = c1[(unsigned)(y + r) * L.ipitch];
                        box_e0 += a0 * a0; box_e1 += a1 * a1;
                    }
                }
                box_y = y;
                const unsigned e0 = in0 ? box_e0 : 0u, e1 = in1 ? box_e1 : 0u;
                const unsigned pre = wave_scan_u32(e0 + e1);
                __syncthreads();                            // the previous row's readers of inc are done
                inc[2 * lane] = pre - e1; inc[2 * lane + 1] = pre;
                __syncthreads();
                for (int x = lane; x < L.aw; x += SLAM_WG) {
                    const unsigned hi = inc[x + L.tw - 1], lo = x > 0 ? inc[x - 1] : 0u;
                    acc[2 * lean_strip_cell(x, cy, rows_b, nunits) + 1] = hi - lo;
                }
            }
            __syncthreads();                                // inc is read before the tile overwrites it
        }
        {   // image tile: byte windows W0 and W3 of every dword position, for the image rows under template rows u0 .. u0 + Rc - 1
            const int trows = (rows_b - 1) * rstep + Rc;
            const int lg = 32 - __builtin_clz((unsigned)tp);   // 2^lg > tp lanes per tile row
            const int c = lane & ((1 << lg) - 1), lr = lane >> lg, rpp = SLAM_WG >> lg;
            if (c < tp) {
                // Bytes past the image row's last whole dword are only ever read for padding cells and padding template
                // columns (whose sums are dropped or masked to zero), and rows past the tile by nobody: such loads are
                // clamped onto valid dwords instead of being predicated.
                const unsigned lastd = (unsigned)(L.w >> 2) - 1u;
                const unsigned dcol = (unsigned)((L.x0 - osrc) >> 2) + (unsigned)c;
                const unsigned g0 = min(dcol, lastd) * 4u, g1 = min(dcol + 1u, lastd) * 4u, g2 = min(dcol + 2u, lastd) * 4u;
                const bool low = osrc == 0;                 // bytes 3..6 from the window's start begin in the first dword
                const uint8_t* src = L.im + (unsigned long long)(L.y0 + r0 * rstep + u0) * L.ipitch;
                const int nit = (trows + rpp - 1) >> (6 - lg);   // wave-uniform
                for (int it = 0; it < nit; it += 4) {
                    unsigned d0[4], d1[4], d2[4], ro[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        ro[q] = (unsigned)min(lr + (it + q) * rpp, trows - 1);
                        d0[q] = *reinterpret_cast<const unsigned*>(src + ro[q] * L.ipitch + g0);
                        d1[q] = *reinterpret_cast<const unsigned*>(src + ro[q] * L.ipitch + g1);
                        if (CHUNKS) d2[q] = *reinterpret_cast<const unsigned*>(src + ro[q] * L.ipitch + g2);
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        unsigned w0 = d0[q], w3;
                        if (CHUNKS) {
                            w0 = __builtin_amdgcn_alignbyte(d1[q], d0[q], (unsigned)osrc);
                            w3 = __builtin_amdgcn_alignbyte(low ? d1[q] : d2[q], low ? d0[q] : d1[q], (unsigned)(osrc + 3) & 3u);
                        } else {
                            w3 = __builtin_amdgcn_alignbyte(d1[q], d0[q], 3);
                        }
                        *reinterpret_cast<lds_u64*>(tile + (ro[q] * (unsigned)tp + (unsigned)c) * 8u) = ((unsigned long long)w3 << 32) | w0;
                    }
                }
            }
        }
        __syncthreads();
        STAMP(53);
        if (CHUNKS && box) {
            // two cell rows per unit where a chunk has template rows enough to carry the extra tile row a task then walks
            // (a 60x40 line area under 20 of a 40x40 template's rows: 0.404 -> 0.365 ms per 3072 lines; chunks of 6 rows of
            // an 81x81 template: 1.98 -> 2.43 ms per 16384 walks, hence the bound)
            if (Rc >= 12 && rows_b >= 2) lean_accumulate_rows2(tile, boxm ? pc : pc + u0 * lean_pc_pitch(G), zrow, acc, nunits, Rc, G, rows_b, tp);
            else lean_accumulate_rows(tile, boxm ? pc : pc + u0 * lean_pc_pitch(G), acc, nunits, lean_pick_row_slices(nunits, Rc, G), Rc, G, rows_b, tp);
            if (boxm) lean_accumulate_dropped(tile, clist, acc, nunits, ncomp, rows_b, tp);
        }
        else if (strips) lean_accumulate<4>(tile, list, marr, acc, nunits, S, na, nb, nA1, rows_b, tp, o, 1);
        else lean_accumulate<1>(tile, list, marr, acc, nunits, S, na, nb, nA1, nx, tp, o, L.step);
        __syncthreads();
        STAMP(54);
        ch++;
      } while (CHUNKS && ch < nchunk);
        if (direct) { blk = nblk - 1; continue; }
        if (blk == nblk) break;                            // the rows around the peak: no scores, the fit reads the sums

        // ---- scores of the block's cells ---------------------------------------------------------------------------
        const int ncell = nx * rows_b;
        constexpr int NQ = 8;
        if (ncell <= SLAM_WG) {                            // one cell per lane
            if (lane < ncell) {
                const int cy = (int)(((float)lane + 0.5f) * rnx), cx = lane - cy * nx;
                const int x = cx * L.step, y = (r0 + cy) * L.step;
                if (!(L.amask && L.amask[y * L.aw + x] == 0) && own(y))
                    best_update(b, lean_score(acc, strips ? lean_strip_cell(o + x, cy, rows_b, nunits) : lane, nm, mt), (y << 16) | x);
            }
        } else if (ncell <= NQ * SLAM_WG && nm.tn > 0.0) {
            // Single-precision estimates first (relative error below 1e-6): only cells within 1e-5 of the largest estimate
            // can hold the maximum or tie with it, and only those are scored with the exact formula.  -FLT_MAX marks cells
            // that are not visited or whose exact score is NaN (an all-zero window under the F32 formula).
            const float rn = __builtin_amdgcn_rcpf(nm.p2f);
            if constexpr (SEARCH) {
                // the single-search kernel runs at 96 registers (5 wavefronts per SIMD): a cell's accumulator index and key
                // are formed again for the few cells scored exactly instead of being kept for all of them
                float est[NQ], m = -FLT_MAX;
                // (cell q * 64 + lane: accumulator index and the key (y << 16) | x)
                auto cell_of = [&](int i, int& ai, int& key) {
                    const int cy = (int)(((float)i + 0.5f) * rnx), cx = i - cy * nx;
                    const int x = cx * L.step, y = (r0 + cy) * L.step;
                    ai = strips ? lean_strip_cell(o + x, cy, rows_b, nunits) : i;
                    key = (y << 16) | x;
                    return !(L.amask && L.amask[y * L.aw + x] == 0) && own(y);
                };
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    est[q] = -FLT_MAX;
                    if (q * SLAM_WG < ncell) {
                        const int i = q * SLAM_WG + lane;
                        int ai, key;
                        if (i < ncell && cell_of(i, ai, key)) {
                            const unsigned CC = acc[2 * ai], SI = acc[2 * ai + 1];
                            if (SI != 0u) est[q] = (float)CC * __builtin_amdgcn_rsqf((float)SI) * rn;
                            else if (mt) est[q] = 0.f;           // matchTemplate scores an all-zero window 0, the F32 formula NaN
                        }
                        m = fmaxf(m, est[q]);
                    }
                }
                // estimates are >= 0 or the marker, so their bit patterns order like the values once the marker is mapped to 0
                m = __uint_as_float(wave_max_u32(m >= 0.f ? __float_as_uint(m) : 0u));
                const float keep_from = m - 1e-5f * fmaxf(m, 1.f);
#pragma unroll
                for (int q = 0; q < NQ; q++)
                    if (q * SLAM_WG < ncell && est[q] != -FLT_MAX && est[q] >= keep_from) {
                        int ai, key;
                        cell_of(q * SLAM_WG + lane, ai, key);
                        best_update(b, lean_score(acc, ai, nm, mt), key);
                    }
            } else {
                float est[NQ], m = -FLT_MAX;
                int ai[NQ], key[NQ];
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    est[q] = -FLT_MAX; ai[q] = 0; key[q] = 0;
                    if (q * SLAM_WG < ncell) {
                        const int i = q * SLAM_WG + lane;
                        const int cy = (int)(((float)i + 0.5f) * rnx), cx = i - cy * nx;
                        const int x = cx * L.step, y = (r0 + cy) * L.step;
                        if (i < ncell && !(L.amask && L.amask[y * L.aw + x] == 0) && own(y)) {
                            ai[q] = strips ? lean_strip_cell(o + x, cy, rows_b, nunits) : i;
                            key[q] = (y << 16) | x;
                            const unsigned CC = acc[2 * ai[q]], SI = acc[2 * ai[q] + 1];
                            if (SI != 0u) est[q] = (float)CC * __builtin_amdgcn_rsqf((float)SI) * rn;
                            else if (mt) est[q] = 0.f;           // matchTemplate scores an all-zero window 0, the F32 formula NaN
                        }
                        m = fmaxf(m, est[q]);
                    }
                }
                // estimates are >= 0 or the marker, so their bit patterns order like the values once the marker is mapped to 0
                m = __uint_as_float(wave_max_u32(m >= 0.f ? __float_as_uint(m) : 0u));
                const float keep_from = m - 1e-5f * fmaxf(m, 1.f);
#pragma unroll
                for (int q = 0; q < NQ; q++)
                    if (q * SLAM_WG < ncell && est[q] != -FLT_MAX && est[q] >= keep_from) best_update(b, lean_score(acc, ai[q], nm, mt), key[q]);
            }
        } else if (nm.tn > 0.0) {
            // The same with more cells than registers.  A lane keeps its best estimate with its cell, and its second
            // best: when no lane's second best comes within the margin of the wave's maximum - the usual case - only
            // the lanes' best cells can qualify and one pass over the cells is enough; otherwise the estimates are
            // formed a second time to pick every cell that gets the exact formula.
            const float rn = __builtin_amdgcn_rcpf(nm.p2f);
            float m1 = -FLT_MAX, m2 = -FLT_MAX;
            int i1 = 0;
            for (int i = lane; i < ncell; i += SLAM_WG) {
                const int cy = (int)(((float)i + 0.5f) * rnx), cx = i - cy * nx;
                const int x = cx * L.step, y = (r0 + cy) * L.step;
                if ((L.amask && L.amask[y * L.aw + x] == 0) || !own(y)) continue;
                const int ai = strips ? lean_strip_cell(o + x, cy, rows_b, nunits) : i;
                const unsigned CC = acc[2 * ai], SI = acc[2 * ai + 1];
                float est = -FLT_MAX;
                if (SI != 0u) est = (float)CC * __builtin_amdgcn_rsqf((float)SI) * rn;
                else if (mt) est = 0.f;
                if (est > m1) { m2 = m1; m1 = est; i1 = i; }
                else m2 = fmaxf(m2, est);                  // a tie with the best counts as a second candidate
            }
            const float m = __uint_as_float(wave_max_u32(m1 >= 0.f ? __float_as_uint(m1) : 0u));
            const float keep_from = m - 1e-5f * fmaxf(m, 1.f);
            if (__ballot(m2 != -FLT_MAX && m2 >= keep_from) == 0ull) {
                if (m1 != -FLT_MAX && m1 >= keep_from) {
                    const int cy = (int)(((float)i1 + 0.5f) * rnx), cx = i1 - cy * nx;
                    const int x = cx * L.step, y = (r0 + cy) * L.step;
                    best_update(b, lean_score(acc, strips ? lean_strip_cell(o + x, cy, rows_b, nunits) : i1, nm, mt), (y << 16) | x);
                }
            } else {
                for (int i = lane; i < ncell; i += SLAM_WG) {
                    const int cy = (int)(((float)i + 0.5f) * rnx), cx = i - cy * nx;
                    const int x = cx * L.step, y = (r0 + cy) * L.step;
                    if ((L.amask && L.amask[y * L.aw + x] == 0) || !own(y)) continue;
                    const int ai = strips ? lean_strip_cell(o + x, cy, rows_b, nunits) : i;
                    const unsigned CC = acc[2 * ai], SI = acc[2 * ai + 1];
                    float est = -FLT_MAX;
                    if (SI != 0u) est = (float)CC * __builtin_amdgcn_rsqf((float)SI) * rn;
                    else if (mt) est = 0.f;
                    if (est != -FLT_MAX && est >= keep_from) best_update(b, lean_score(acc, ai, nm, mt), (y << 16) | x);
                }
            }
        } else {
            for (int i = lane; i < ncell; i += SLAM_WG) {
                const int cy = (int)(((float)i + 0.5f) * rnx), cx = i - cy * nx;
                const int x = cx * L.step, y = (r0 + cy) * L.step;
                if ((L.amask && L.amask[y * L.aw + x] == 0) || !own(y)) continue;
                best_update(b, lean_score(acc, strips ? lean_strip_cell(o + x, cy, rows_b, nunits) : i, nm, mt), (y << 16) | x);
            }
        }
    }
    // sub-pixel fit through the four neighbours of the peak: lanes 0..3 score one each from the sums of rows r0 .. r0 + rows_b - 1
    const int n = lane & 3;
    const int qx = lx + (n == 0 ? -1 : n == 1 ? 1 : 0), qy = ly + (n == 2 ? -1 : n == 3 ? 1 : 0);
    const bool in = qx >= 0 && qy >= 0 && qx < L.aw && qy < L.ah;
    float v = -1.f;
    if (in && !(L.amask && L.amask[qy * L.aw + qx] == 0)) v = lean_score(acc, lean_strip_cell(o + qx, qy - r0, rows_b, rows_b * upr), nm, mt);
    const float sl = __shfl(v, 0, 64), sr = __shfl(v, 1, 64), su = __shfl(v, 2, 64), sd = __shfl(v, 3, 64);
    const bool okl = __shfl((int)in, 0, 64) != 0, okr = __shfl((int)in, 1, 64) != 0;
    const bool oku = __shfl((int)in, 2, 64) != 0, okd = __shfl((int)in, 3, 64) != 0;
    out.dx = subpixel_axis(bv, sl, sr, okl, okr);
    out.dy = subpixel_axis(bv, su, sd, oku, okd);
    return out;
}

}  // namespace slamdev
