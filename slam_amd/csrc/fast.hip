// FAST-9-16 corner score + non-max suppression (cv::FAST as called at new_point.cpp:120) and
// SLAM::getSortedKeyPoints (new_point.cpp:113-127).
//
// Kernel 1 (fast_band_kernel): one workgroup (4 wavefronts) per band of R image rows of one frame.
//   Phase 1+2, per 256-pixel chunk of the band: the chunk's pixels (+ halo) are staged into LDS; each
//   lane tests 4 adjacent pixels with the opposite-pair test (a 9-long arc contains one pixel of
//   every opposite ring pair, so "all 8 pairs have a brighter / darker member" is necessary), the
//   wavefront compacts the candidates and computes the exact score only for those, writing the
//   stored score (score for corners, 0 otherwise) of band rows -1..R into an LDS score tile.
//   Phase 3: each wavefront walks its rows of the score tile in row-major order, applies the strict
//   3x3 non-max test to the non-zero entries and compacts survivors with ballots; a first pass
//   counts, the band reserves a slice of the frame's key pool, a second pass writes the keys.
// Kernel 2 (fast_gather_kernel): one workgroup per frame concatenates the bands in band order,
//   which gives cv::FAST's row-major keypoint order, and converts keys to keypoints.
#include <algorithm>
#include <cmath>

#include "slam_internal.h"

namespace {

constexpr int FAST_WG = 256;
constexpr int FAST_CHUNK = 256;                 // pixels per chunk: 4 per lane
constexpr int FAST_PPITCH = FAST_CHUNK + 8;     // pixel tile pitch: 4 halo columns each side
#ifndef FAST_LDS_BUDGET_KB
#define FAST_LDS_BUDGET_KB 40
#endif
constexpr int FAST_LDS_BUDGET = FAST_LDS_BUDGET_KB * 1024;
constexpr int FAST_CL_CAP = 1024;               // corners a band can list for the sparse non-max pass

struct FastPlan {
    int R;            // rows per band; R + 2 score rows are a multiple of 4 (one share per wavefront)
    int nbands;
    int spitch;       // score tile pitch (bytes): 4 + w + 4, rounded up to 4
    size_t lds_bytes;
};

__host__ __device__ inline size_t fast_lds_bytes(int R, int spitch) {
    // score tile + pixel chunk + candidate lists (4 wavefronts x (R+2)/4 score rows x 256 slots x 2 bytes);
    // phase 3 reuses the pixel chunk + lists as one keep-flag byte per 4 pixels (R * spitch / 4 bytes)
    // a list of the band's corners (FAST_CL_CAP dwords) follows
    const size_t work = (size_t)(R + 8) * FAST_PPITCH + (size_t)(R + 2) * 256 * 2;
    const size_t flags = (size_t)R * (spitch / 4);
    return (size_t)(R + 2) * spitch + ((work > flags ? work : flags) + 15) / 16 * 16 + 16 + (size_t)FAST_CL_CAP * 4;
}

bool fast_plan(int w, int h, FastPlan& p) {
    const int spitch = (w + 8 + 3) & ~3;
    for (int R = 30; R >= 2; R -= 4) {
        if (fast_lds_bytes(R, spitch) <= (size_t)FAST_LDS_BUDGET || R == 2) {
            if (fast_lds_bytes(R, spitch) > 64 * 1024) return false;   // dynamic LDS of a plain launch
            p.R = R; p.nbands = (h + R - 1) / R; p.spitch = spitch; p.lds_bytes = fast_lds_bytes(R, spitch);
            return true;
        }
    }
    return false;
}

// per-frame scratch layout (all 4-byte words): [0] pool counter, [1..3] pad, then band_off[nbands],
// band_cnt[nbands], then the key pool of `cap` keys
__host__ __device__ inline size_t fast_frame_words(int nbands, int cap) {
    return (size_t)4 + 2 * (size_t)((nbands + 3) / 4 * 4) + (size_t)((cap + 3) / 4 * 4);
}

typedef __attribute__((address_space(3))) unsigned char fl_u8;
typedef __attribute__((address_space(3))) unsigned short fl_u16;
typedef __attribute__((address_space(3))) unsigned fl_u32;

// byte i (0..11) of the 12-byte window {a, b, c}
__device__ __forceinline__ int win_byte(unsigned a, unsigned b, unsigned c, int i) {
    const unsigned wd = i < 4 ? a : (i < 8 ? b : c);
    return (int)((wd >> (8 * (i & 3))) & 0xffu);
}

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

// max over the 16 arcs of 9 contiguous ring pixels of the arc minimum of (v - ring) [low half] and
// of (ring - v) [high half]: the two "A" values of cv::FAST's cornerScore, by window doubling.
__device__ __forceinline__ void fast_arc_scores(const fl_u8* p, int pitch, int& a_dark, int& a_bright) {
    const int v = p[0];
    fast_s2 e[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int r = p[ring_dy(k) * pitch + ring_dx(k)];
        e[k] = fast_s2{(short)(v - r), (short)(r - v)};
    }
    fast_s2 m2[16], m4[16], m8[16];
#pragma unroll
    for (int k = 0; k < 16; k++) m2[k] = __builtin_elementwise_min(e[k], e[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) m4[k] = __builtin_elementwise_min(m2[k], m2[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) m8[k] = __builtin_elementwise_min(m4[k], m4[(k + 4) & 15]);
    fast_s2 best = fast_s2{(short)-512, (short)-512};
#pragma unroll
    for (int k = 0; k < 16; k++) best = __builtin_elementwise_max(best, __builtin_elementwise_min(m8[k], e[(k + 8) & 15]));
    a_dark = best.x; a_bright = best.y;
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

__global__ __launch_bounds__(FAST_WG) void fast_band_kernel(
    const uint8_t* __restrict__ frames, int w, int h, long long pitch, long long frame_stride,
    int threshold, int nonmax, int R, int spitch, int nbands, int cap, unsigned* __restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int wave_cnt[4];
    __shared__ int pool_off;
    fl_u8* sc = (fl_u8*)smem;                                    // (R+2) x spitch stored scores; column x at byte x + 4
    fl_u8* px = sc + (size_t)(R + 2) * spitch;                   // (R+8) x FAST_PPITCH pixels of the current chunk
    fl_u16* lists = (fl_u16*)(px + (size_t)(R + 8) * FAST_PPITCH);   // 4 x ((R+2)/4 rows x 256) candidate slots
    const size_t work_bytes = (size_t)(R + 8) * FAST_PPITCH + (size_t)(R + 2) * 256 * 2, flag_bytes = (size_t)R * (spitch / 4);
    fl_u32* ccount = (fl_u32*)(px + ((work_bytes > flag_bytes ? work_bytes : flag_bytes) + 15) / 16 * 16);
    fl_u32* clist = ccount + 4;                                   // corners of band rows: (band row << 16) | column

    const int band = blockIdx.x, frame = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint8_t* im = frames + (long long)frame * frame_stride;
    const int y0 = band * R;                                     // first image row of the band
    unsigned* fs = scratch + (size_t)frame * fast_frame_words(nbands, cap);
    const bool aligned = ((reinterpret_cast<unsigned long long>(frames) | (unsigned long long)pitch | (unsigned long long)frame_stride) & 3ull) == 0;
    const int t = threshold;

    if (tid == 0) *ccount = 0u;
    {   // zero the score tile
        fl_u32* z = (fl_u32*)sc;
        const int n = ((R + 2) * spitch) >> 2;
        for (int i = tid; i < n; i += FAST_WG) z[i] = 0u;
    }

    const int srows = R + 2;                                     // score rows: image rows y0-1 .. y0+R
    const int rows_per_wave = srows >> 2;
    fl_u16* mylist = lists + wid * (rows_per_wave * 256);
    for (int cx = 0; cx < w; cx += FAST_CHUNK) {
        __syncthreads();                                         // previous chunk's readers are done (and the zeroing)
        // ---- stage pixels: tile row r <-> image row y0 - 4 + r, tile byte c <-> image column cx - 4 + c
        // (loads of a batch are all issued before the first LDS write: one memory round trip per batch)
        constexpr int NB = 6;
        const int ndw = (R + 8) * (FAST_PPITCH / 4);
        for (int base = tid; base < ndw; base += FAST_WG * NB) {
            unsigned v[NB];
#pragma unroll
            for (int u = 0; u < NB; u++) {
                const int i = base + u * FAST_WG;
                v[u] = 0;
                if (i < ndw) {
                    const int r = i / (FAST_PPITCH / 4), c = i - r * (FAST_PPITCH / 4);
                    const int y = y0 - 4 + r, x = cx - 4 + 4 * c;
                    if (y >= 0 && y < h && x + 3 >= 0 && x < w) {
                        const uint8_t* src = im + (long long)y * pitch + x;
                        if (aligned && x >= 0 && x + 3 < w) v[u] = *reinterpret_cast<const unsigned*>(src);
                        else {
#pragma unroll
                            for (int b = 0; b < 4; b++)
                                if (x + b >= 0 && x + b < w) v[u] |= (unsigned)src[b] << (8 * b);
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < NB; u++) {
                const int i = base + u * FAST_WG;
                if (i < ndw) reinterpret_cast<fl_u32*>(px)[i] = v[u];
            }
        }
        __syncthreads();
        // ---- phase 1 + 2: each wavefront takes rows_per_wave score rows of this chunk
        int n = 0;                                               // candidates of this wavefront in this chunk
        for (int rr = 0; rr < rows_per_wave; rr++) {
            const int ry = wid * rows_per_wave + rr;             // score row; image row y0 - 1 + ry; pixel tile row ry + 3
            const int y = y0 - 1 + ry;
            if (y < 3 || y >= h - 3) continue;                   // wave-uniform
            const int x0 = cx + 4 * lane;                        // first of the lane's 4 pixels
            unsigned cand = 0;
#ifdef FAST_DIAG_SKIP_P1
            if (x0 < -w) {
#else
            if (x0 < w) {
#endif
                // window rows dy = -3..3: dwords at tile bytes 4*lane, +4, +8 (image columns x0-4 .. x0+7)
                unsigned A[7], B[7], Cw[7];
#pragma unroll
                for (int d = 0; d < 7; d++) {
                    const fl_u32* row = reinterpret_cast<const fl_u32*>(px + (ry + d) * FAST_PPITCH) + lane;
                    A[d] = row[0]; B[d] = row[1]; Cw[d] = row[2];
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int v = win_byte(A[3], B[3], Cw[3], 4 + j);
                    // every opposite pair must have a member brighter than v+t (resp. darker than v-t):
                    // min over pairs of the pair maximum > v+t, max over pairs of the pair minimum < v-t
                    int mn_of_mx = 255, mx_of_mn = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const int r0 = win_byte(A[3 + ring_dy(k)], B[3 + ring_dy(k)], Cw[3 + ring_dy(k)], 4 + j + ring_dx(k));
                        const int r1 = win_byte(A[3 + ring_dy(k + 8)], B[3 + ring_dy(k + 8)], Cw[3 + ring_dy(k + 8)], 4 + j + ring_dx(k + 8));
                        mn_of_mx = min(mn_of_mx, max(r0, r1));
                        mx_of_mn = max(mx_of_mn, min(r0, r1));
                    }
                    const bool bright = mn_of_mx > v + t, dark = mx_of_mn < v - t;
                    const int x = x0 + j;
                    if ((bright || dark) && x >= 3 && x < w - 3) cand |= 1u << j;
                }
            }
            // append the candidates of these 256 pixels to the wavefront's list: (score row << 8) | column
            if (__ballot(cand != 0u) != 0ull) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const bool c = (cand >> j) & 1u;
                    const unsigned long long bal = __ballot(c);
                    if (c) mylist[n + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)((rr << 8) | (4 * lane + j));
                    n += __popcll(bal);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef FAST_DIAG_SKIP_P2
        n = 0;
#endif
        for (int i = lane; i < n; i += 64) {
            const int e = mylist[i];
            const int lx = e & 255, ry = wid * rows_per_wave + (e >> 8);
            int ad, ab;
            fast_arc_scores(px + (ry + 3) * FAST_PPITCH + 4 + lx, FAST_PPITCH, ad, ab);
            const int a = max(ad, ab);
            if (a > t) {
                sc[ry * spitch + 4 + cx + lx] = (unsigned char)(nonmax ? a - 1 : a);   // no NMS: score + 1 marks a corner
                if (ry >= 1 && ry <= R) {                        // halo rows only serve as neighbours
                    const unsigned pos = __hip_atomic_fetch_add(ccount, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (pos < (unsigned)FAST_CL_CAP) clist[pos] = ((unsigned)(ry - 1) << 16) | (unsigned)(cx + lx);
                }
            }
        }
    }
    __syncthreads();

    // ---- phase 3: non-max suppression and ordered compaction; pass 0 counts, pass 1 writes
    const int r_lo = (R * wid) >> 2, r_hi = (R * (wid + 1)) >> 2;   // this wavefront's band rows
    unsigned* pool = fs + 4 + 2 * ((nbands + 3) / 4 * 4);
    const int ncorner = (int)*ccount;
    const int wpr = (w + 31) >> 5;                               // bitmap words per row
    if (ncorner <= FAST_CL_CAP && (size_t)R * wpr * 4 <= work_bytes) {
        // Sparse band: one lane per listed corner decides the non-max test and sets its bit in a
        // band bitmap (over the pixel chunk, free now); the bitmap is then read out in row-major order.
        fl_u32* bmp = (fl_u32*)px;
        for (int i = tid; i < R * wpr; i += FAST_WG) bmp[i] = 0u;
        __syncthreads();
        for (int i = tid; i < ncorner; i += FAST_WG) {
            const unsigned e = clist[i];
            const int ry = (int)(e >> 16), x = (int)(e & 0xffffu);
            const int y = y0 + ry;
            if (y < 3 || y >= h - 3) continue;
            const fl_u8* c = sc + (ry + 1) * spitch + 4 + x;
            bool keep = true;
            if (nonmax) {
                const int s0 = c[0];
                int nb = max(max((int)c[-1], (int)c[1]), max((int)c[-spitch], (int)c[spitch]));
                nb = max(nb, max(max((int)c[-spitch - 1], (int)c[-spitch + 1]), max((int)c[spitch - 1], (int)c[spitch + 1])));
                keep = s0 > nb;
            }
            if (keep) __hip_atomic_fetch_or(bmp + ry * wpr + (x >> 5), 1u << (x & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
        int cnt = 0;
        for (int ry = r_lo; ry < r_hi; ry++)
            for (int wb = 0; wb < wpr; wb += 64)
                if (wb + lane < wpr) cnt += __popc(bmp[ry * wpr + wb + lane]);
        {
            const unsigned tot = wave_scan_incl((unsigned)cnt);
            if (lane == 63) wave_cnt[wid] = (int)tot;
        }
        __syncthreads();
        const int c0 = wave_cnt[0], c1 = wave_cnt[1], c2 = wave_cnt[2], c3 = wave_cnt[3];
        const int total = c0 + c1 + c2 + c3;
        if (tid == 0) {
            int off = 0;
            if (total > 0) off = (int)atomicAdd(&fs[0], (unsigned)total);
            pool_off = off;
            const int nb4 = (nbands + 3) / 4 * 4;
            fs[4 + band] = (unsigned)off;
            fs[4 + nb4 + band] = (unsigned)total;
        }
        __syncthreads();
        if (total == 0) return;
        int pos0 = pool_off + ((wid == 0) ? 0 : (wid == 1) ? c0 : (wid == 2) ? c0 + c1 : c0 + c1 + c2);
        for (int ry = r_lo; ry < r_hi; ry++)
            for (int wb = 0; wb < wpr; wb += 64) {
                unsigned m = (wb + lane < wpr) ? bmp[ry * wpr + wb + lane] : 0u;
                const unsigned incl = wave_scan_incl((unsigned)__popc(m));
                int pos = pos0 + (int)incl - __popc(m);
                pos0 += (int)__builtin_amdgcn_readlane((int)incl, 63);
                while (m) {
                    const int b = __ffs((int)m) - 1;
                    m &= m - 1u;
                    const int x = ((wb + lane) << 5) + b;
                    const unsigned s = sc[(ry + 1) * spitch + 4 + x];
                    if (pos < cap) pool[pos] = ((unsigned)(ry * w + x) << 8) | (nonmax ? s : s - 1u);
                    pos++;
                }
            }
        return;
    }
    fl_u8* flags = px;                                           // the pixel chunk and the lists are free now
    const int fpitch = spitch >> 2;
    int base = 0;
#ifdef FAST_DIAG_SKIP_P3
    if (w > 0) return;
#endif
    for (int pass = 0; pass < 2; pass++) {
        int cnt = 0;
        for (int ry = r_lo; ry < r_hi; ry++) {
            const int y = y0 + ry;
            if (y < 3 || y >= h - 3) continue;                   // wave-uniform
            const fl_u8* row = sc + (ry + 1) * spitch + 4;       // score row of image row y, column 0
            for (int xb = 0; xb < w; xb += FAST_CHUNK) {
                const int x0 = xb + 4 * lane;
                fl_u8* flag = flags + ry * fpitch + (x0 >> 2);   // pass 0 leaves the lane's 4 keep bits here
                unsigned c4 = 0, keep = 0;
                if (pass == 1) {
                    if (x0 < w) keep = *flag;
                    if (__ballot(keep != 0u) == 0ull) continue;
                    if (keep) c4 = *reinterpret_cast<const fl_u32*>(row + x0);
                } else {
                    if (x0 < w) c4 = *reinterpret_cast<const fl_u32*>(row + x0);
                    if (__ballot(c4 != 0u) == 0ull) {
                        if (x0 < w) *flag = 0;
                        continue;
                    }
                    if (c4 != 0u) {
                        if (nonmax) {
                            unsigned U[3], M[3], D[3];
#pragma unroll
                            for (int q = 0; q < 3; q++) {
                                U[q] = *reinterpret_cast<const fl_u32*>(row - spitch + x0 - 4 + 4 * q);
                                M[q] = *reinterpret_cast<const fl_u32*>(row + x0 - 4 + 4 * q);
                                D[q] = *reinterpret_cast<const fl_u32*>(row + spitch + x0 - 4 + 4 * q);
                            }
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const int s = win_byte(M[0], M[1], M[2], 4 + j);
                                int nb = 0;                      // largest of the 8 neighbours
#pragma unroll
                                for (int dx = -1; dx <= 1; dx++) {
                                    nb = max(nb, max(win_byte(U[0], U[1], U[2], 4 + j + dx), win_byte(D[0], D[1], D[2], 4 + j + dx)));
                                    if (dx != 0) nb = max(nb, win_byte(M[0], M[1], M[2], 4 + j + dx));
                                }
                                if (s > nb) keep |= 1u << j;
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; j++)
                                if ((c4 >> (8 * j)) & 0xffu) keep |= 1u << j;
                        }
                    }
                    if (x0 < w) *flag = (unsigned char)keep;
                }
                // row-major order inside the chunk = lane-major, then j
                int before = 0, total = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const unsigned long long bal = __ballot((keep >> j) & 1u);
                    before += __popcll(bal & ((1ull << lane) - 1ull));
                    total += __popcll(bal);
                }
                if (pass == 1 && keep) {
                    int pos = base + cnt + before;
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if ((keep >> j) & 1u) {
                            const unsigned s = (c4 >> (8 * j)) & 0xffu;
                            if (pos < cap) pool[pos] = ((unsigned)(ry * w + x0 + j) << 8) | (nonmax ? s : s - 1u);
                            pos++;
                        }
                }
                cnt += total;
            }
        }
        if (pass == 0) {
            if (lane == 0) wave_cnt[wid] = cnt;
            __syncthreads();
            const int c0 = wave_cnt[0], c1 = wave_cnt[1], c2 = wave_cnt[2], c3 = wave_cnt[3];
            const int total = c0 + c1 + c2 + c3;
            if (tid == 0) {
                int off = 0;
                if (total > 0) off = (int)atomicAdd(&fs[0], (unsigned)total);
                pool_off = off;
                const int nb4 = (nbands + 3) / 4 * 4;
                fs[4 + band] = (unsigned)off;
                fs[4 + nb4 + band] = (unsigned)total;
            }
            __syncthreads();
            base = pool_off + ((wid == 0) ? 0 : (wid == 1) ? c0 : (wid == 2) ? c0 + c1 : c0 + c1 + c2);
            if (total == 0) break;
        }
    }
}

// One workgroup per (band, frame): finds the band's place in the frame's output by summing the
// counts of the bands before it, then converts and copies its keys.
__global__ __launch_bounds__(FAST_WG) void fast_gather_kernel(int w, int R, int nbands, int cap,
                                                              const unsigned* __restrict__ scratch,
                                                              slam_keypoint* __restrict__ out,
                                                              int* __restrict__ counts) {
    __shared__ int part[4];
    const int band = blockIdx.x, frame = blockIdx.y, tid = threadIdx.x;
    const unsigned* fs = scratch + (size_t)frame * fast_frame_words(nbands, cap);
    const int nb4 = (nbands + 3) / 4 * 4;
    const unsigned* band_off = fs + 4;
    const unsigned* band_cnt = fs + 4 + nb4;
    const unsigned* pool = fs + 4 + 2 * nb4;
    slam_keypoint* o = out + (size_t)frame * cap;
    unsigned mine = 0;
    for (int b = tid; b < band; b += FAST_WG) mine += band_cnt[b];
    const unsigned tot = wave_scan_incl(mine);
    if ((tid & 63) == 63) part[tid >> 6] = (int)tot;
    __syncthreads();
    const int base = part[0] + part[1] + part[2] + part[3];
    const int cnt = (int)band_cnt[band], off = (int)band_off[band];
    for (int i = tid; i < cnt; i += FAST_WG) {
        const int dst = base + i;
        if (dst < cap && off + i < cap) {
            const unsigned key = pool[off + i];
            const unsigned lin = key >> 8;
            const int ry = (int)(lin / (unsigned)w), x = (int)(lin - (unsigned)ry * (unsigned)w);
            slam_keypoint kp;
            kp.x = (float)x; kp.y = (float)(band * R + ry); kp.response = (float)(key & 0xffu);
            o[dst] = kp;
        }
    }
    if (band == nbands - 1 && tid == 0) counts[frame] = base + cnt;
}

}  // namespace

extern "C" size_t slam_hip_fast_scratch_bytes(int w, int h, int nframes, int cap_per_frame) {
    FastPlan p;
    if (w < 1 || h < 1 || nframes < 1 || cap_per_frame < 1) return 0;
    if (!fast_plan(w, h, p)) return 0;
    return fast_frame_words(p.nbands, cap_per_frame) * 4 * (size_t)nframes;
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
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "fast_detect: image too wide (%d) for the band kernel", w);
    const size_t need = fast_frame_words(p.nbands, cap_per_frame) * 4 * (size_t)nframes;
    if (scratch_bytes < need || ((uintptr_t)d_scratch & 15))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "fast_detect: scratch too small or misaligned (%zu < %zu)", scratch_bytes, need);
    threshold = std::min(std::max(threshold, 0), 255);          // cv::FAST clamps the threshold
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // zero the pool counters (first word of every frame record); the records are 16-byte multiples
    SLAM_HIP_CHECK(ctx, hipMemset2DAsync(d_scratch, fast_frame_words(p.nbands, cap_per_frame) * 4, 0, 16, (size_t)nframes, ctx->stream));
    hipLaunchKernelGGL(fast_band_kernel, dim3(p.nbands, nframes), dim3(FAST_WG), p.lds_bytes, ctx->stream,
                       d_frames, w, h, (long long)pitch, (long long)frame_stride, threshold, nonmax ? 1 : 0,
                       p.R, p.spitch, p.nbands, cap_per_frame, (unsigned*)d_scratch);
    SLAM_HIP_CHECK(ctx, hipGetLastError());
    hipLaunchKernelGGL(fast_gather_kernel, dim3(p.nbands, nframes), dim3(FAST_WG), 0, ctx->stream, w, p.R, p.nbands,
                       cap_per_frame, (const unsigned*)d_scratch, d_out, d_counts);
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
    for (int f = 0; f < nframes; f++)
        SLAM_HIP_CHECK(ctx, hipMemcpy2DAsync((uint8_t*)ctx->d_stage[0] + (size_t)f * dstride, dpitch,
                                             h_frames + (size_t)f * frame_stride, pitch, (size_t)w, (size_t)h,
                                             hipMemcpyHostToDevice, ctx->stream));
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
    size_t dpitch, dstride;
    int rc = fast_upload(ctx, h_frames, w, h, pitch, frame_stride, nframes, dpitch, dstride);
    if (rc != SLAM_HIP_OK) return rc;
    return fast_detect_resident(ctx, (const uint8_t*)ctx->d_stage[0], w, h, dpitch, dstride, nframes,
                                threshold, nonmax, out, cap_per_frame, counts);
}

extern "C" int slam_hip_get_sorted_keypoints(slam_ctx* ctx, const uint8_t* im, int w, int h, size_t pitch,
                                             size_t n_min_pts, float* fast_threshold,
                                             slam_keypoint* out, int cap, int32_t* n_out) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (!im || !fast_threshold || !out || !n_out || w < 1 || h < 1 || cap < 1 || pitch < (size_t)w)
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "get_sorted_keypoints: bad arguments");
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    size_t dpitch, dstride;
    int rc = fast_upload(ctx, im, w, h, pitch, pitch * (size_t)h, 1, dpitch, dstride);
    if (rc != SLAM_HIP_OK) return rc;
    // non-max-suppressed corners are never 8-adjacent: at most ceil(w/2)*ceil(h/2) of them
    const size_t kmax = (size_t)((w + 1) / 2) * (size_t)((h + 1) / 2);
    std::vector<slam_keypoint> kps(kmax ? kmax : 1);
    float f = *fast_threshold;
    size_t n = 0;
    while (n < n_min_pts) {                                   // new_point.cpp:116-122
        const int t = (int)f;                                 // FAST(im, out, fastThreshold, true): float -> int
        int32_t cnt = 0;
        rc = fast_detect_resident(ctx, (const uint8_t*)ctx->d_stage[0], w, h, dpitch, dstride, 1, t, 1,
                                  kps.data(), (int)kps.size(), &cnt);
        if (rc != SLAM_HIP_OK) return rc;
        n = (size_t)cnt;
        f *= 0.5f;
        if (t <= 0) break;   // the reference never leaves this loop on a corner-poor image; stop after t = 0
    }
    f *= 2.f;                                                 // new_point.cpp:123
    // new_point.cpp:124 sorts by response only with an unstable sort; ties are fixed as y, x ascending
    std::sort(kps.begin(), kps.begin() + n, [](const slam_keypoint& a, const slam_keypoint& b) {
        if (a.response != b.response) return a.response > b.response;
        if (a.y != b.y) return a.y < b.y;
        return a.x < b.x;
    });
    if (n > 2 * n_min_pts) f = kps[(size_t)((double)n_min_pts * 1.9)].response;   // new_point.cpp:125-126
    *fast_threshold = f;
    *n_out = (int32_t)n;
    for (size_t i = 0; i < n && i < (size_t)cap; i++) out[i] = kps[i];
    return SLAM_HIP_OK;
}
