// Template search kernels: one wavefront (workgroup of 64 lanes) per job.
//   match_kernel  = SLAM::matchFeatureInArea (matching.cpp:9-228) for a table of jobs
//   track_kernel  = the level loop of SLAM::Feature::track (feature.cpp:117-172), walking the
//                   pyramid levels of one feature inside one workgroup (each level's window depends
//                   on the previous level's result)
#include <cfloat>
#include <climits>
#include <mutex>

#include "match_lean.h"

using namespace slamdev;

// Minimum wavefronts per SIMD the register allocation must allow (caps VGPRs at 512 / this).  Four waves (128
// VGPRs, a few spilled to scratch) beat three (168, none spilled) wherever a launch has more jobs than the chip has
// wave slots: C2 walks by 3-5 %, 81x81 walks by 8-16 %.  Line walks come three to a frame (new_image.cpp:80): a
// batch rarely fills the slots, so that kernel keeps the spill-free allocation.
#ifndef SLAM_MIN_WAVES
#define SLAM_MIN_WAVES 4
#endif
#ifndef SLAM_MIN_WAVES_CHUNKS
#define SLAM_MIN_WAVES_CHUNKS 4       // the lean walk kernel's build with blocks and chunks (track_lean_kernel<true>)
#endif
#ifndef SLAM_MIN_WAVES_LINE
#define SLAM_MIN_WAVES_LINE 3
#endif

namespace {

static_assert(SLAM_WG == 64, "the search kernels assume one wavefront per workgroup");

// Direct evaluation of every score cell the reference visits: one lane per cell, looping over the
// template pixels.  Cold path: templates too wide for the tiled core (or for the LDS it was given)
// and the in-GPU cross-check of the tiled core (SLAM_HIP_FORCE_DIRECT=1).  Not inlined, so that it
// does not weigh on the register allocation and code size of the tiled path.
struct DirectOut { unsigned long long spfull; float v; int idx; };   // returned in registers

__device__ __noinline__ DirectOut direct_cells(Geo g, Tmpl t, float* scores) {
    const int lane = threadIdx.x;
    unsigned long long spfull;
    {   // |patch0|^2 over the whole template (matching.cpp:63,109; template norm of matchTemplate)
        unsigned long long sp = 0;
        const int n = t.tw * t.th;
        for (int i = lane; i < n; i += SLAM_WG) {
            unsigned p, m;
            tmpl_fetch(t, i / t.tw, i % t.tw, p, m);
            p *= m;
            sp += p * p;
        }
        spfull = wave_sum_u64(sp);
    }
    Best b = {-FLT_MAX, INT_MAX};
    const int ncell = g.aw * g.ah;
    for (int c = lane; c < ncell; c += SLAM_WG) {
        const int x = c % g.aw, y = c / g.aw;
        const int kind = classify_cell(g, x, y);
        float sc = -1.f;                                   // matching.cpp:53
        if (kind != CELL_NONE) {
            int dl = 0, dr = 0, dt = 0, db = 0;
            if (kind == CELL_F32_CROP) {
                dl = max(0, -g.x0 - x); dr = max(0, g.x0 + x + g.pw - g.w);
                dt = max(0, -g.y0 - y); db = max(0, g.y0 + y + g.ph - g.h);
            }
            const Sums s = cell_sums_partial(g, t, g.x0 + x, g.y0 + y, dl, g.pw - dr, dt, g.ph - db, 0, 1);
            sc = (kind == CELL_MT) ? score_mt(s.cc, s.si, spfull)
               : (kind == CELL_F32_FULL) ? score_f32(s.cc, s.si, spfull)
                                         : score_f32(s.cc, s.si, s.sp);
        }
        if (scores) scores[c] = sc;
        best_update(b, sc, c);
    }
    b = best_wave(b);
    DirectOut o = {spfull, b.v, b.idx};
    return o;
}

// Score-map value of cell (nx, ny) recomputed by the 64 lanes together (-1 for unvisited cells).
// Cold path (sub-pixel fit after a direct search).
__device__ __noinline__ float cell_score_coop(Geo g, Tmpl t, int nx, int ny, unsigned long long spfull) {
    const int kind = classify_cell(g, nx, ny);
    if (kind == CELL_NONE) return -1.f;
    int dl = 0, dr = 0, dt = 0, db = 0;
    if (kind == CELL_F32_CROP) {
        dl = max(0, -g.x0 - nx); dr = max(0, g.x0 + nx + g.pw - g.w);
        dt = max(0, -g.y0 - ny); db = max(0, g.y0 + ny + g.ph - g.h);
    }
    Sums s = cell_sums_partial(g, t, g.x0 + nx, g.y0 + ny, dl, g.pw - dr, dt, g.ph - db, (int)threadIdx.x, SLAM_WG);
    s.cc = wave_sum_u64(s.cc); s.si = wave_sum_u64(s.si); s.sp = wave_sum_u64(s.sp);
    return (kind == CELL_MT) ? score_mt(s.cc, s.si, spfull)
         : (kind == CELL_F32_FULL) ? score_f32(s.cc, s.si, spfull)
                                   : score_f32(s.cc, s.si, s.sp);
}

// SLAM::matchFeatureInArea for one job: score cells (tiled packed-u8 core, or direct evaluation),
// the cv::minMaxLoc argmax (matching.cpp:221-227) and, when asked for (always, or when the response
// is below refine_below), the sub-pixel offset of the peak: a parabola through the two horizontal /
// vertical neighbours of the argmax cell (not in the reference).  The result is wave-uniform.
__device__ __forceinline__ CoreOut match_core(const Geo& g, const Tmpl& t, float* scores, lds_u8* lds, const LdsPlan& lp,
                                              bool force_direct, bool refine_always, double refine_below STAMP_ARG) {
    Best b = {-FLT_MAX, INT_MAX};
    unsigned long long spfull = 0;
    Norms nm = make_norms(0);
    FastState fs;
    fs.valid = false;
    STAMP(3);
    if (!force_direct && !scores) {                    // small all-interior level: one lane group per cell
        MiniPlan mp;
        if (mini_supported(g, t, lp.acc_bytes + lp.tmpl_bytes + lp.tile_bytes, mp)) return mini_level(g, t, lds, mp, refine_always, refine_below STAMP_PASS);
    }
    const bool use_fast = !force_direct && fast_supported(g, lp);
    if (use_fast) fast_search(g, t, scores, lds, lp, refine_always, refine_below, spfull, nm, b, fs STAMP_PASS);
    else {
        const DirectOut d = direct_cells(g, t, scores);
        spfull = d.spfull; b.v = d.v; b.idx = d.idx;
    }
    STAMP(20);
    CoreOut o;
    o.dx = 0.f; o.dy = 0.f;
    if (b.idx == INT_MAX) {      // nothing compared greater than -FLT_MAX: every cell is NaN
        o.x = -1 + g.ax; o.y = -1 + g.ay; o.response = 0.f;
        return o;
    }
    const int ly = row_of_idx(g, b.idx), lx = b.idx - ly * g.aw;
    o.x = lx + g.ax; o.y = ly + g.ay; o.response = b.v;
    if (!(refine_always || (double)b.v < refine_below)) return o;

    // the four neighbours of the peak: lanes 0..3 each fetch one
    const int n = lane_here() & 3;
    const int nx = lx + (n == 0 ? -1 : n == 1 ? 1 : 0);
    const int ny = ly + (n == 2 ? -1 : n == 3 ? 1 : 0);
    const bool in = nx >= 0 && ny >= 0 && nx < g.aw && ny < g.ah;
    float v = -1.f;
    if (use_fast) {
        if (in) v = cell_from_acc(g, lds, fs.ba, fs.big, nx, ny, nm);
    } else {
        for (int k = 0; k < 4; k++) {                      // cooperative: every lane takes part in each cell
            const int kx = lx + (k == 0 ? -1 : k == 1 ? 1 : 0), ky = ly + (k == 2 ? -1 : k == 3 ? 1 : 0);
            if (kx >= 0 && ky >= 0 && kx < g.aw && ky < g.ah) {
                const float s = cell_score_coop(g, t, kx, ky, spfull);
                if (n == k) v = s;
            }
        }
    }
    const float sl = __shfl(v, 0, 64), sr = __shfl(v, 1, 64), su = __shfl(v, 2, 64), sd = __shfl(v, 3, 64);
    const bool okl = __shfl((int)in, 0, 64) != 0, okr = __shfl((int)in, 1, 64) != 0;
    const bool oku = __shfl((int)in, 2, 64) != 0, okd = __shfl((int)in, 3, 64) != 0;
    const float s = b.v;
    if (okl && okr && sl != -1.f && sr != -1.f && sl == sl && sr == sr) {
        const float d2 = __fsub_rn(__fadd_rn(sl, sr), __fmul_rn(2.0f, s));
        if (d2 < 0.f) {
            const float q = __fdiv_rn(__fsub_rn(sl, sr), __fmul_rn(2.0f, d2));
            o.dx = q < -0.5f ? -0.5f : (q > 0.5f ? 0.5f : q);
        }
    }
    if (oku && okd && su != -1.f && sd != -1.f && su == su && sd == sd) {
        const float d2 = __fsub_rn(__fadd_rn(su, sd), __fmul_rn(2.0f, s));
        if (d2 < 0.f) {
            const float q = __fdiv_rn(__fsub_rn(su, sd), __fmul_rn(2.0f, d2));
            o.dy = q < -0.5f ? -0.5f : (q > 0.5f ? 0.5f : q);
        }
    }
    return o;
}

__device__ __forceinline__ long long uni_i64(long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ int lane_i32(int v, int i) { return __builtin_amdgcn_readlane(v, i); }
__device__ __forceinline__ float lane_f32(float v, int i) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i)); }
__device__ __forceinline__ long long lane_i64(long long v, int i) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, i);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), i);
    return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double lane_f64(double v, int i) { return __longlong_as_double(lane_i64(__double_as_longlong(v), i)); }

// SLAM::matchFeatureInArea (matching.cpp:9-228) for job j, general form.
__device__ __forceinline__ void match_job(const slam_pyr_dev& pyr, const slam_match_job* __restrict__ jobs, int j,
                                          const uint8_t* __restrict__ blob, unsigned long long blob_bytes, float* scores,
                                          unsigned long long scores_len, slam_match_result* __restrict__ out, int force_direct,
                                          lds_u8* lds, const LdsPlan& plan) {
    const slam_match_job job = jobs[j];
    bool ok = job.frame >= 0 && job.frame < pyr.nframes && job.level >= 0 && job.level < pyr.nlev &&
              job.pw >= 1 && job.ph >= 1 && job.pw <= 32767 && job.ph <= 32767 && job.stride >= 1 && job.stride <= (1 << 20) &&
              job.aw <= (1 << 20) && job.ah <= (1 << 20) &&
              job.ax > -(1 << 28) && job.ax < (1 << 28) && job.ay > -(1 << 28) && job.ay < (1 << 28);
    const long long pbytes = (long long)job.pw * job.ph;
    ok = ok && span_ok(job.patch_off, pbytes, blob_bytes);
    if (job.pmask_off != -1) ok = ok && span_ok(job.pmask_off, pbytes, blob_bytes);
    if (job.amask_off != -1) ok = ok && job.aw >= 1 && job.ah >= 1 && span_ok(job.amask_off, (long long)job.aw * job.ah, blob_bytes);
    if (!ok) {
        if (threadIdx.x == 0) { slam_match_result r = {0, 0, -1.f, 0.f, 0.f, SLAM_HIP_EINVAL}; out[j] = r; }
        return;
    }
    int lw = 0, lh = 0; long long lp = 0, ls = 0; const uint8_t* ld = nullptr;
#pragma unroll
    for (int l = 0; l < SLAM_HIP_MAX_LEVELS; l++)
        if (l == job.level) { lw = pyr.lev[l].w; lh = pyr.lev[l].h; lp = pyr.lev[l].pitch; ls = pyr.lev[l].frame_stride; ld = pyr.lev[l].data; }
    const uint8_t* im = ld + (long long)job.frame * ls;
    const Geo g = make_geo(im, lw, lh, lp, job.pw, job.ph, job.ax, job.ay, job.aw, job.ah,
                           job.amask_off != -1 ? blob + job.amask_off : nullptr, job.stride,
                           job.pmask_off != -1, job.use_exact_area_mask != 0);
    if (g.reject) {   // matching.cpp:26-30
        if (threadIdx.x == 0) { slam_match_result r = {0, 0, -1.f, 0.f, 0.f, SLAM_HIP_OK}; out[j] = r; }
        return;
    }
    float* sc = nullptr;
    if (scores && job.score_off >= 0) {
        if (!span_ok(job.score_off, (long long)g.aw * g.ah, scores_len)) {
            if (threadIdx.x == 0) { slam_match_result r = {0, 0, -1.f, 0.f, 0.f, SLAM_HIP_EINVAL}; out[j] = r; }
            return;
        }
        sc = scores + job.score_off;
    }
    Tmpl t;
    t.patch = blob + job.patch_off;
    t.mask = job.pmask_off != -1 ? blob + job.pmask_off : nullptr;
    t.srcw = job.pw; t.srch = job.ph; t.tw = job.pw; t.th = job.ph; t.resized = false;
    t.scale_x = 1.0; t.scale_y = 1.0;
    t.limit = blob + blob_bytes;
    STAMP_DECL;
    const CoreOut o = match_core(g, t, sc, lds, plan, force_direct != 0, true, 0.0 STAMP_PASS);
    if (threadIdx.x == 0) { slam_match_result r = {o.x, o.y, o.response, o.dx, o.dy, SLAM_HIP_OK}; out[j] = r; }
}

#define SLAM_DEFER_LIST 16

// Queue of the jobs a lean kernel leaves to the general one.  Word 0 is the lean kernel's slot counter, word 2 keeps the
// count of the last launch (diagnostics); the list starts at word SLAM_DEFER_LIST and holds job index + 1, zero meaning
// "no entry".  Every launch leaves the counter and the list at zero: the general kernel never reads the counter - its
// workgroup b takes the slots b, b + grid, ... until it meets an empty one (the lean kernel fills slots from 0 upwards
// without gaps) and empties every slot it takes, and its first workgroup resets the counter, which nobody else reads
// before the next lean kernel starts.  So a launch depends on no host-side state, a captured launch replays as often as
// wanted, and a launch with an empty queue costs each workgroup one load.
__device__ __forceinline__ void queue_begin(int* queue) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { queue[2] = queue[0]; queue[0] = 0; }
}
// the job in slot q (wave-uniform), or -1 when the slot is empty; the slot is left empty
__device__ __forceinline__ int queue_take(int* queue, int q) {
    const int e = __builtin_amdgcn_readfirstlane(queue[SLAM_DEFER_LIST + q]);
    if (e != 0 && threadIdx.x == 0) queue[SLAM_DEFER_LIST + q] = 0;
    return e - 1;
}
__device__ __forceinline__ void queue_put(int* queue, int j) { queue[SLAM_DEFER_LIST + atomicAdd(queue, 1)] = j + 1; }

// General single-search kernel.  queue == nullptr: workgroup b takes job b.  Otherwise the workgroups share out the jobs
// the lean kernel queued (a later launch on the same stream: its counter and list are complete and visible).
__global__ __launch_bounds__(SLAM_WG, SLAM_MIN_WAVES) void match_kernel(slam_pyr_dev pyr, const slam_match_job* __restrict__ jobs,
                                                        int njobs, const uint8_t* __restrict__ blob,
                                                        unsigned long long blob_bytes, float* scores,
                                                        unsigned long long scores_len,
                                                        slam_match_result* __restrict__ out, int force_direct, int lds_total,
                                                        int* queue) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    lds_u8* lds = (lds_u8*)dyn_lds;
    const LdsPlan plan = make_lds_plan(lds_total);
    if (!queue) {
        const int j = blockIdx.x;
        if (j < njobs) match_job(pyr, jobs, j, blob, blob_bytes, scores, scores_len, out, force_direct, lds, plan);
        return;
    }
    queue_begin(queue);
    for (int q = blockIdx.x; q < njobs; q += gridDim.x) {
        const int j = queue_take(queue, q);
        if (j < 0) break;
        __syncthreads();                                   // the previous job's last readers of the dynamic LDS
        if (j < njobs) match_job(pyr, jobs, j, blob, blob_bytes, scores, scores_len, out, force_direct, lds, plan);
    }
}

// Lean single-search kernel: one matchFeatureInArea call whose candidate patches all lie inside the image, without a
// score map, small enough for lean_level (match_lean.h).  With a patch mask the cells of the stride lattice inside
// the area mask are scored by the F32 formula (matching.cpp:64-73); without one every cell is scored by
// matchTemplate's normalisation whatever the stride (matching.cpp:151-153), and an area mask removes cells only
// when it is exact (matching.cpp:157-163).  Any other job is appended to the queue for match_kernel; nothing is
// written to `out` for it here.
// The lean single-search kernel is built for 5 wavefronts per SIMD (96 registers, a handful spilled): with 7680 bytes of
// LDS per search (6 granules, slam_internal.h) 20 searches share a CU instead of 16, which pays for the spills and
// for the lower row blocks (C1: 0.245 -> 0.228 ms per 16384 searches; at 4 wavefronts and 10240 bytes 0.245, at 5 and 10240 0.261).
#ifndef SLAM_MIN_WAVES_LEAN_SEARCH
#define SLAM_MIN_WAVES_LEAN_SEARCH 5
#endif
__global__ __launch_bounds__(SLAM_WG, SLAM_MIN_WAVES_LEAN_SEARCH) void match_lean_kernel(slam_pyr_dev pyr, const slam_match_job* __restrict__ jobs,
                                                        int njobs, const uint8_t* __restrict__ blob,
                                                        unsigned long long blob_bytes, const float* scores,
                                                        slam_match_result* __restrict__ out, int lds_total, int* __restrict__ queue) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    lds_u8* lds = (lds_u8*)dyn_lds;
    const int j = blockIdx.x;
    if (j >= njobs) return;
    const slam_match_job job = jobs[j];
    STAMP_DECL;
    bool ok = job.frame >= 0 && job.frame < pyr.nframes && job.level >= 0 && job.level < pyr.nlev &&
              job.pw >= 1 && job.ph >= 1 && job.pw <= 255 && job.ph <= 255 && job.stride >= 1 && job.stride <= 1024 &&
              job.aw >= 1 && job.ah >= 1 && job.aw <= 255 && job.ah <= 255 &&
              job.ax > -(1 << 28) && job.ax < (1 << 28) && job.ay > -(1 << 28) && job.ay < (1 << 28) &&
              !(scores && job.score_off >= 0);
    const long long pbytes = (long long)job.pw * job.ph;
    ok = ok && span_ok(job.patch_off, pbytes, blob_bytes);
    if (job.pmask_off != -1) ok = ok && span_ok(job.pmask_off, pbytes, blob_bytes);
    if (job.amask_off != -1) ok = ok && span_ok(job.amask_off, (long long)job.aw * job.ah, blob_bytes);
    LeanOut o;
    o.ok = false;
    if (ok) {
        int lw = 0, lh = 0; long long lp = 0, ls = 0; const uint8_t* ld = nullptr;
#pragma unroll
        for (int l = 0; l < SLAM_HIP_MAX_LEVELS; l++)
            if (l == job.level) { lw = pyr.lev[l].w; lh = pyr.lev[l].h; lp = pyr.lev[l].pitch; ls = pyr.lev[l].frame_stride; ld = pyr.lev[l].data; }
        LeanLevel L;
        L.im = ld + (long long)job.frame * ls;
        L.ipitch = (unsigned)lp; L.w = lw; L.h = lh;
        L.tw = job.pw; L.th = job.ph; L.srcw = job.pw; L.srch = job.ph;
        L.resized = false; L.scx = 1.0; L.scy = 1.0;
        L.patch = blob + job.patch_off; L.limit = blob + blob_bytes;
        L.mask = job.pmask_off != -1 ? blob + job.pmask_off : nullptr;
        L.mt = job.pmask_off == -1;
        L.own0 = 0; L.own1 = INT_MAX;
        L.x0 = job.ax - job.pw / 2; L.y0 = job.ay - job.ph / 2;                       // matching.cpp:38-39
        L.aw = job.aw; L.ah = job.ah;
        L.step = L.mt ? 1 : job.stride;
        L.amask = (job.amask_off != -1 && (!L.mt || job.use_exact_area_mask != 0)) ? blob + job.amask_off : nullptr;
        // every candidate patch inside the image (no crop, no border cells: matching.cpp:19-25,44-49), image rows that
        // can be fetched as aligned dwords
        ok = L.x0 >= 0 && L.y0 >= 0 && L.x0 + L.aw + L.tw - 1 <= L.w && L.y0 + L.ah + L.th - 1 <= L.h && lp <= 0x7fffffffLL &&
             ((reinterpret_cast<unsigned long long>(L.im) | (unsigned long long)lp) & 3ull) == 0;
        if (ok) o = lean_level<true>(L, lds, lds_total, true, 0.0 STAMP_PASS);
    }
    if (threadIdx.x == 0) {
        if (o.ok) { slam_match_result r = {o.x, o.y, o.response, o.dx, o.dy, SLAM_HIP_OK}; out[j] = r; }
        else queue_put(queue, j);
    }
}

// per-level constants of a point walk, one LDS entry per pyramid level
struct TrackLevel { float sub, rad; int lw, lh, tw, th, rs, pad; long long lp, ls, ld; double scx, scy; };

// The level loop of SLAM::Feature::track (feature.cpp:117-172) for job j, general form: any window, any template
// size the tiled core or direct evaluation can take.
__device__ __forceinline__ void track_job(const slam_pyr_dev& pyr, const slam_track_job* __restrict__ jobs, int j,
                                          const uint8_t* __restrict__ blob, unsigned long long blob_bytes,
                                          slam_track_result* __restrict__ out, int force_direct, lds_u8* lds,
                                          const LdsPlan& plan, TrackLevel* lc) {
    const slam_track_job job = jobs[j];
    const int tid = threadIdx.x;
    if (job.projw <= 0 || job.projh <= 0) {   // feature.cpp:112-115
        if (tid == 0) { slam_track_result r = {-1, -1, -1.f, 0.f, 0.f, 0, SLAM_HIP_OK}; out[j] = r; }
        return;
    }
    const long long pbytes = (long long)job.projw * job.projh;
    bool ok = job.frame >= 0 && job.frame < pyr.nframes && job.projw <= SLAM_MAX_TMPL && job.projh <= SLAM_MAX_TMPL &&
              job.stride >= 1 && job.stride <= 1024 &&
              span_ok(job.proj_off, pbytes, blob_bytes) && span_ok(job.mask_off, pbytes, blob_bytes) &&
              fabsf(job.px) < 1e7f && fabsf(job.py) < 1e7f && job.threshold == job.threshold;
    const bool coarse_area = (job.flags & SLAM_HIP_TRACK_COARSE_AREA) != 0;
    if (coarse_area)
        ok = ok && job.aw >= 1 && job.ah >= 1 && job.aw <= (1 << 12) && job.ah <= (1 << 12) &&
             job.ax > -(1 << 28) && job.ax < (1 << 28) && job.ay > -(1 << 28) && job.ay < (1 << 28) &&
             (job.amask_off == -1 || span_ok(job.amask_off, (long long)job.aw * job.ah, blob_bytes));
    if (!ok) {
        if (tid == 0) { slam_track_result r = {-1, -1, -1.f, 0.f, 0.f, 0, SLAM_HIP_EINVAL}; out[j] = r; }
        return;
    }
    Tmpl t;
    t.patch = blob + job.proj_off; t.mask = blob + job.mask_off;
    t.srcw = job.projw; t.srch = job.projh;
    t.scale_x = 1.0; t.scale_y = 1.0;
    t.limit = blob + blob_bytes;

    STAMP_DECL;
    STAMP(1);
    // Per-level constants do not depend on the walk's results: lane l computes those of level l once
    // (template size and cv::resize scales, search radius, window size) and parks them in LDS; the level
    // loop reads entry i back into scalar registers.
    if (tid < pyr.nlev) {
        TrackLevel c;
        float subn = 1.f;
#pragma unroll
        for (int l = 0; l < SLAM_HIP_MAX_LEVELS; l++) {
            if (l == tid) { c.sub = pyr.lev[l].sub; c.lw = pyr.lev[l].w; c.lh = pyr.lev[l].h; c.lp = pyr.lev[l].pitch; c.ls = pyr.lev[l].frame_stride; c.ld = reinterpret_cast<long long>(pyr.lev[l].data); }
            if (l == tid + 1) subn = pyr.lev[l].sub;
        }
        STAMP(8);
        c.tw = job.projw; c.th = job.projh;      // feature.cpp:126-128
        c.scx = 1.0; c.scy = 1.0;
        if (c.sub != 1.f) {                      // feature.cpp:130-131, Size(projw/sub, projh/sub) truncates
            c.tw = (int)__fdiv_rn((float)job.projw, c.sub);
            c.th = (int)__fdiv_rn((float)job.projh, c.sub);
            if (c.tw > 0 && c.th > 0) {
                c.scx = resize_scale(job.projw, c.tw);
                c.scy = resize_scale(job.projh, c.th);
            }
        }
        STAMP(7);
        if (tid == pyr.nlev - 1) {
            c.rad = __fdiv_rn(20.f, c.sub);      // feature.cpp:119,134
        } else {                                 // feature.cpp:136-138: double expression stored to float
            c.rad = (float)__ddiv_rn(__dmul_rn(1.3, (double)subn), (double)c.sub);
            if (tid == 0) c.rad = __fmul_rn(c.rad, (float)job.stride);
        }
        c.rs = round_f2i(__fadd_rn(__fmul_rn(2.f, c.rad), 1.f));   // feature.cpp:146
        c.pad = 0;
        lc[tid] = c;
    }
    const double stop_below = __dmul_rn(0.67, (double)job.threshold);           // feature.cpp:151
    __syncthreads();
    STAMP(6);

    float tx = job.px, ty = job.py;             // trackedPoint (feature.cpp:122)
    float response = 0.f, sdx = 0.f, sdy = 0.f;
    int levels_run = 0;
    bool empty_level = false;
    for (int i = pyr.nlev - 1; i >= 0; --i) {
        const TrackLevel c = lc[i];
        const float sub = uni_f32(c.sub), searchRad = uni_f32(c.rad);
        const int lw = uni_i32(c.lw), lh = uni_i32(c.lh), rs = uni_i32(c.rs);
        const long long lp = uni_i64(c.lp), ls = uni_i64(c.ls);
        const uint8_t* ld = reinterpret_cast<const uint8_t*>(uni_i64(c.ld));
        t.tw = uni_i32(c.tw); t.th = uni_i32(c.th);
        t.resized = sub != 1.f;
        if (t.tw <= 0 || t.th <= 0) { empty_level = true; break; }
        t.scale_x = __longlong_as_double(uni_i64(__double_as_longlong(c.scx)));
        t.scale_y = __longlong_as_double(uni_i64(__double_as_longlong(c.scy)));
        // feature.cpp:144-146
        int rx = __builtin_amdgcn_readfirstlane(round_f2i(__fsub_rn(__fdiv_rn(tx, sub), searchRad)));
        int ry = __builtin_amdgcn_readfirstlane(round_f2i(__fsub_rn(__fdiv_rn(ty, sub), searchRad)));
        int rw = rs, rh = rs;
        const uint8_t* am = nullptr;
        if (coarse_area && i == pyr.nlev - 1) {                  // caller's window and area mask on the coarsest level
            rx = job.ax; ry = job.ay; rw = job.aw; rh = job.ah;
            if (job.amask_off != -1) am = blob + job.amask_off;
        }
        const Geo g = make_geo(ld + (long long)job.frame * ls, lw, lh, lp, t.tw, t.th, rx, ry, rw, rh,
                               am, (i == 0) ? 1 : job.stride, true, false);   // feature.cpp:147-149
        int mx = 0, my = 0;
        sdx = 0.f; sdy = 0.f;
        levels_run++;
        STAMP(2);
        if (g.reject) { response = -1.f; }       // matching.cpp:26-30
        else {
            // feature.cpp:151-152: the level that ends the walk also reports the sub-pixel fit
            const CoreOut o = match_core(g, t, nullptr, lds, plan, force_direct != 0, i == 0, stop_below STAMP_PASS);
            mx = o.x; my = o.y; response = o.response; sdx = o.dx; sdy = o.dy;
            STAMP(21);
        }
        tx = __fmul_rn((float)mx, sub);          // feature.cpp:150
        ty = __fmul_rn((float)my, sub);
        if ((double)response < stop_below) break;   // feature.cpp:151-152
    }
    STAMP(30);
    if (tid == 0) {
        slam_track_result r;
        if (empty_level) { r.x = -1; r.y = -1; r.response = -1.f; r.dx = 0.f; r.dy = 0.f; r.levels_run = levels_run; }
        else { r.x = (int)tx; r.y = (int)ty; r.response = response; r.dx = sdx; r.dy = sdy; r.levels_run = levels_run; }
        r.status = SLAM_HIP_OK;
        out[j] = r;
    }
}

// General walk kernel.  queue == nullptr: workgroup b walks job b.  Otherwise the workgroups share out the jobs the
// lean kernel queued (a later launch on the same stream: its counter and list are complete and visible).
__global__ __launch_bounds__(SLAM_WG, SLAM_MIN_WAVES) void track_kernel(slam_pyr_dev pyr, const slam_track_job* __restrict__ jobs,
                                                        int njobs, const uint8_t* __restrict__ blob,
                                                        unsigned long long blob_bytes,
                                                        slam_track_result* __restrict__ out, int force_direct, int lds_total,
                                                        int* queue) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    __shared__ TrackLevel lc[SLAM_HIP_MAX_LEVELS];
    lds_u8* lds = (lds_u8*)dyn_lds;
    const LdsPlan plan = make_lds_plan(lds_total);
    if (!queue) {
        const int j = blockIdx.x;
        if (j < njobs) track_job(pyr, jobs, j, blob, blob_bytes, out, force_direct, lds, plan, lc);
        return;
    }
    queue_begin(queue);                                    // see match_kernel
    for (int q = blockIdx.x; q < njobs; q += gridDim.x) {
        const int j = queue_take(queue, q);
        if (j < 0) break;
        __syncthreads();                                   // the previous job's last readers of lc and the dynamic LDS
        if (j < njobs) track_job(pyr, jobs, j, blob, blob_bytes, out, force_direct, lds, plan, lc);
    }
}

// Lean walk kernel: the level loop of SLAM::Feature::track (feature.cpp:117-172) for walks whose every level is an
// interior window small enough for lean_level (match_lean.h).  A job it does not cover - checked level by level, as
// the windows depend on the results - is appended to the queue and walked again from the start by track_kernel;
// nothing is written to `out` for it here.
//
// A small batch of walks under large templates (one frame of the reference's own: 64 features with 81x81 descriptors,
// new_image.cpp:80-82) leaves most of the chip idle while one wavefront per walk searches the full-resolution level.
// For such batches that level is cut into `nb` bands of cell rows, one wavefront each (WALK_BAND: every band's
// wavefront walks the levels above by itself - they are a few per cent of the work and give every band the same
// window - and searches its rows of the last level together with one row of its neighbours on either side: only its
// own rows compete for the maximum, the neighbours' are there for the sub-pixel fit of a peak on the band's edge;
// track_band_finish_kernel takes the first of the bands' maxima in band order, the first maximum in row-major order).
enum { WALK_WHOLE = 0, WALK_BAND = 1 };
struct WalkBand {
    float tx, ty;            // trackedPoint after the walk's last level, from this band's peak - or where the walk ended above it
    int x, y;                // the band's peak at the last level, level coordinates
    float response, dx, dy;
    int levels_run;
    int state;               // 0: not covered (the general kernel walks the job), 1: band searched, 2: the walk ended above the last level
    int pad[3];
};

template <bool CHUNKS, int MODE = WALK_WHOLE>
__device__ __forceinline__ bool lean_walk(const slam_pyr_dev& pyr, const slam_track_job& job, const uint8_t* __restrict__ blob,
                                          unsigned long long blob_bytes, lds_u8* lds, int lds_total, slam_track_result& r,
                                          WalkBand* wb = nullptr, int band = 0, int nb = 1) {
    STAMP_DECL;
    STAMP(1);
    const long long pbytes = (long long)job.projw * job.projh;
    bool ok = job.projw >= 1 && job.projh >= 1 && job.projw <= 255 && job.projh <= 255 &&
              job.frame >= 0 && job.frame < pyr.nframes && job.stride >= 1 && job.stride <= 1024 &&
              span_ok(job.proj_off, pbytes, blob_bytes) && span_ok(job.mask_off, pbytes, blob_bytes) &&
              fabsf(job.px) < 1e7f && fabsf(job.py) < 1e7f && job.threshold == job.threshold;
    const bool coarse_area = (job.flags & SLAM_HIP_TRACK_COARSE_AREA) != 0;
    if (coarse_area)
        ok = ok && job.aw >= 1 && job.ah >= 1 && job.aw <= 255 && job.ah <= 255 &&
             job.ax > -(1 << 28) && job.ax < (1 << 28) && job.ay > -(1 << 28) && job.ay < (1 << 28) &&
             (job.amask_off == -1 || span_ok(job.amask_off, (long long)job.aw * job.ah, blob_bytes));
    if (!ok) { LEAN_REJECT(10); return false; }
    LeanLevel L;
    L.patch = blob + job.proj_off; L.mask = blob + job.mask_off; L.limit = blob + blob_bytes;
    L.srcw = job.projw; L.srch = job.projh;
    L.mt = false;                               // walks carry a patch mask: the F32 formula (matching.cpp:70-72)
    L.own0 = 0; L.own1 = INT_MAX;
    const double stop_below = __dmul_rn(0.67, (double)job.threshold);           // feature.cpp:151
    float tx = job.px, ty = job.py;             // trackedPoint (feature.cpp:122)
    float response = 0.f, sdx = 0.f, sdy = 0.f;
    int levels_run = 0;
    const int nlev = pyr.nlev;
    // Template size and cv::resize scale of every level do not depend on the walk's results: lane l computes the
    // width and x scale of level l, lane 8 + l its height and y scale (feature.cpp:126-131); the level loop reads them
    // back with readlane.
    int lvl_t; double lvl_sc;
    {
        const int lane = lane_here();
        const int q = min(lane & 7, nlev - 1);
        const slam_level_dev& lq = pyr.lev[q];
        const float sub = lq.sub;
        const int src = (lane & 8) ? job.projh : job.projw;
        lvl_t = src; lvl_sc = 1.0;
        int oth = (lane & 8) ? job.projw : job.projh;             // the level's other template side
        if (sub != 1.f) {                       // Size(projw/sub, projh/sub) truncates
            lvl_t = (int)(lq.pow2 ? __fmul_rn((float)src, lq.inv_sub) : __fdiv_rn((float)src, sub));
            oth = (int)(lq.pow2 ? __fmul_rn((float)oth, lq.inv_sub) : __fdiv_rn((float)oth, sub));
            if (lvl_t > 0) lvl_sc = resize_scale(src, lvl_t);
        }
        // Does every level fit this kernel's LDS?  Decided here from the sizes alone (lanes 0..nlev-1, one level each), so
        // that a walk the general kernel must take is handed over before any of its levels has been searched.
        bool fits = true;
        if (lane < nlev && lvl_t > 0 && oth > 0) {
            float rad = lq.rad;
            if (q == 0 && nlev > 1) rad = __fmul_rn(rad, (float)job.stride);
            int aw = round_f2i(__fadd_rn(__fmul_rn(2.f, rad), 1.f)), ah = aw;
            if (coarse_area && q == nlev - 1) { aw = job.aw; ah = job.ah; }
            const int step = q == 0 ? 1 : job.stride;
            const int nx = step == 1 ? aw : small_div(aw + step - 1, step), ny = step == 1 ? ah : small_div(ah + step - 1, step);
            fits = aw >= 1 && ah >= 1 && aw <= 255 && ah <= 255 &&
                   (CHUNKS ? lean_plan(lvl_t, oth, aw, ah, step, nx, ny, sub != 1.f, true, lds_total, lean_box_mode(false, true, step, sub != 1.f, lvl_t, oth, aw)).bh > 0
                           : lean_need(lvl_t, oth, aw, ah, step, nx, ny, sub != 1.f) <= lds_total);
        }
        if (__ballot(!fits) != 0ull) { LEAN_REJECT(13); return false; }
    }
    for (int i = nlev - 1; i >= 0; --i) {
        STAMP_LEVEL(i);
        STAMP(2);
        const slam_level_dev& lv = pyr.lev[i];
        const float sub = lv.sub;
        const bool p2 = lv.pow2 != 0;           // dividing by a power of two and multiplying by its inverse are both exact
        L.tw = __builtin_amdgcn_readlane(lvl_t, i); L.th = __builtin_amdgcn_readlane(lvl_t, 8 + i);
        L.resized = sub != 1.f;
        if (L.tw <= 0 || L.th <= 0) { LEAN_REJECT(11); return false; }
        {
            const long long bx = __double_as_longlong(lvl_sc);
            const unsigned xl = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bx, i), xh = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)bx >> 32), i);
            const unsigned yl = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bx, 8 + i), yh = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)bx >> 32), 8 + i);
            L.scx = __longlong_as_double((long long)(((unsigned long long)xh << 32) | xl));
            L.scy = __longlong_as_double((long long)(((unsigned long long)yh << 32) | yl));
        }
        float rad = lv.rad;                     // feature.cpp:134 (coarsest) or :136 (finer)
        if (i == 0 && nlev > 1) rad = __fmul_rn(rad, (float)job.stride);        // feature.cpp:137-138
        const int rs = uni_i32(round_f2i(__fadd_rn(__fmul_rn(2.f, rad), 1.f)));  // feature.cpp:146
        // feature.cpp:144-146
        const float qx = p2 ? __fmul_rn(tx, lv.inv_sub) : __fdiv_rn(tx, sub), qy = p2 ? __fmul_rn(ty, lv.inv_sub) : __fdiv_rn(ty, sub);
        int rx = uni_i32(round_f2i(__fsub_rn(qx, rad))), ry = uni_i32(round_f2i(__fsub_rn(qy, rad)));
        int rw = rs, rh = rs;
        L.amask = nullptr;
        if (coarse_area && i == nlev - 1) {     // caller's window and area mask on the coarsest level
            rx = job.ax; ry = job.ay; rw = job.aw; rh = job.ah;
            if (job.amask_off != -1) L.amask = blob + job.amask_off;
        }
        L.im = lv.data + (long long)job.frame * lv.frame_stride;
        L.ipitch = (unsigned)lv.pitch; L.w = lv.w; L.h = lv.h;
        L.x0 = rx - L.tw / 2; L.y0 = ry - L.th / 2;                                // matching.cpp:38-39
        L.aw = rw; L.ah = rh;
        L.step = (i == 0) ? 1 : job.stride;                                         // feature.cpp:147-149
        // every candidate patch inside the image (no crop, no border cells: matching.cpp:19-25,44-49), and image rows
        // that can be fetched as aligned dwords
        if (rw < 1 || rh < 1 || rw > 255 || rh > 255 || L.x0 < 0 || L.y0 < 0 || L.x0 + rw + L.tw - 1 > L.w || L.y0 + rh + L.th - 1 > L.h ||
            lv.pitch > 0x7fffffffLL || ((reinterpret_cast<unsigned long long>(L.im) | (unsigned long long)lv.pitch) & 3ull) != 0) {
            LEAN_REJECT(12 + 1000 * i); return false;
        }
        if (MODE == WALK_BAND && i == 0) {                 // this band's rows of the last level, and a row either side for the fit
            if (rh < nb) { LEAN_REJECT(14); return false; }
            const int y0b = small_div_u(band * rh, nb), y1b = small_div_u((band + 1) * rh, nb);
            const int r0 = max(y0b - 1, 0), r1 = min(y1b + 1, rh);
            if (L.amask) L.amask += r0 * rw;
            L.y0 += r0; L.ah = r1 - r0;
            L.own0 = y0b - r0; L.own1 = y1b - r0;
            const LeanOut o = lean_level<false, CHUNKS>(L, lds, lds_total, true, stop_below STAMP_PASS);
            if (!o.ok) return false;
            wb->tx = __fmul_rn((float)o.x, sub); wb->ty = __fmul_rn((float)o.y, sub);      // feature.cpp:150
            wb->x = o.x; wb->y = o.y; wb->response = o.response; wb->dx = o.dx; wb->dy = o.dy;
            wb->levels_run = levels_run + 1; wb->state = 1;
            return true;
        }
        // feature.cpp:151-152: the level that ends the walk also reports the sub-pixel fit
        const LeanOut o = lean_level<false, CHUNKS>(L, lds, lds_total, i == 0, stop_below STAMP_PASS);
        if (!o.ok) return false;
        levels_run++;
        response = o.response; sdx = o.dx; sdy = o.dy;
        tx = __fmul_rn((float)o.x, sub);         // feature.cpp:150
        ty = __fmul_rn((float)o.y, sub);
        if ((double)response < stop_below) break;   // feature.cpp:151-152
    }
    STAMP(30);
    if (MODE == WALK_BAND) {                               // the walk ended above the last level: every band holds its result
        wb->tx = tx; wb->ty = ty; wb->x = 0; wb->y = 0; wb->response = response; wb->dx = sdx; wb->dy = sdy;
        wb->levels_run = levels_run; wb->state = 2;
        return true;
    }
    r.x = (int)tx; r.y = (int)ty; r.response = response; r.dx = sdx; r.dy = sdy; r.levels_run = levels_run;
    r.status = SLAM_HIP_OK;
    return true;
}

template <bool CHUNKS>
__global__ __launch_bounds__(SLAM_WG, CHUNKS ? SLAM_MIN_WAVES_CHUNKS : SLAM_MIN_WAVES) void track_lean_kernel(slam_pyr_dev pyr, const slam_track_job* __restrict__ jobs,
                                                        int njobs, const uint8_t* __restrict__ blob,
                                                        unsigned long long blob_bytes,
                                                        slam_track_result* __restrict__ out, int lds_total, int* __restrict__ queue) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    lds_u8* lds = (lds_u8*)dyn_lds;
    // Workgroups are dealt round-robin over the 8 XCDs, each with an L2 of its own.  Workgroup b takes job
    // (b % 8) * ceil(njobs / 8) + b / 8, so that every XCD walks one contiguous eighth of the job table: the features
    // of a frame - whose image tiles share 128-byte lines, most of all on the coarse levels - meet in one L2 back to
    // back.  Placement is a matter of speed only: any assignment of jobs to workgroups gives the same results.
    const int per = (njobs + 7) >> 3;
    const int j = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if (j >= njobs) return;
    const slam_track_job job = jobs[j];
    slam_track_result r;
    const bool done = lean_walk<CHUNKS>(pyr, job, blob, blob_bytes, lds, lds_total, r);
    if (threadIdx.x == 0) {
        if (done) out[j] = r;
        else queue_put(queue, j);
    }
}

// Band b of walk j (workgroup j * nb + b): the levels above the last one, then the band's rows of the last level.
__global__ __launch_bounds__(SLAM_WG, SLAM_MIN_WAVES_CHUNKS) void track_band_kernel(slam_pyr_dev pyr, const slam_track_job* __restrict__ jobs,
                                                        int njobs, const uint8_t* __restrict__ blob, unsigned long long blob_bytes,
                                                        WalkBand* __restrict__ bands, int nb, int lds_total) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    lds_u8* lds = (lds_u8*)dyn_lds;
    const int j = (int)blockIdx.x / nb, band = (int)blockIdx.x - j * nb;
    if (j >= njobs) return;
    const slam_track_job job = jobs[j];
    slam_track_result r;
    WalkBand w;
    w.tx = 0.f; w.ty = 0.f; w.x = 0; w.y = 0; w.response = 0.f; w.dx = 0.f; w.dy = 0.f; w.levels_run = 0; w.state = 0;
    w.pad[0] = w.pad[1] = w.pad[2] = 0;
    if (!lean_walk<true, WALK_BAND>(pyr, job, blob, blob_bytes, lds, lds_total, r, &w, band, nb)) w.state = 0;
    if (threadIdx.x == 0) bands[(size_t)j * nb + band] = w;
}

// The bands of walk j put together: its result, or its index on the queue of the general kernel when a band was not covered.
__global__ void track_band_finish_kernel(const WalkBand* __restrict__ bands, int njobs, int nb, slam_track_result* __restrict__ out,
                                         int* __restrict__ queue) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= njobs) return;
    const WalkBand* bj = bands + (size_t)j * nb;
    WalkBand best = bj[0];
    bool all = best.state != 0;
    for (int k = 1; k < nb; k++) {
        const WalkBand bk = bj[k];
        all = all && bk.state == best.state;
        if (bk.state == 1 && bk.response > best.response) best = bk;      // strict: the first band holding the maximum keeps it
    }
    if (all) {
        slam_track_result r;
        r.x = (int)best.tx; r.y = (int)best.ty; r.response = best.response; r.dx = best.dx; r.dy = best.dy;
        r.levels_run = best.levels_run; r.status = SLAM_HIP_OK;
        out[j] = r;
    } else queue_put(queue, j);
}

// A small batch of line walks (the reference tracks three lines per frame, new_image.cpp:80) leaves most of the chip
// idle while one wavefront per line searches the coarsest level - 2400 cells under a 40x40 template for an 81x81
// descriptor.  For such batches the coarsest search (lineFeature.cpp:124-127) is cut into `nb` bands of rows, one
// wavefront each (line_coarse_kernel), and line_track_kernel takes the first of the bands' maxima in band order - the
// first maximum in row-major order, what cv::minMaxLoc returns (matching.cpp:221-227).  Only areas whose candidate
// patches all lie inside the image are cut (every cell of such an area is scored by matchTemplate's rule whatever the
// stride or the area mask, matching.cpp:151-153, so a band is searched exactly as part of the whole); any other area
// is marked invalid here and searched whole by line_track_kernel.
struct LineBand { int x, y; float response; int valid; };

// job checks shared by the two line kernels
__device__ __forceinline__ bool line_job_ok(const slam_pyr_dev& pyr, const slam_line_job& job, unsigned long long blob_bytes) {
    return job.frame >= 0 && job.frame < pyr.nframes && job.dw >= 1 && job.dh >= 1 && job.dw <= SLAM_MAX_TMPL &&
           job.dh <= SLAM_MAX_TMPL && job.stride >= 1 && job.stride <= 1024 && job.aw >= 1 && job.ah >= 1 &&
           job.aw <= (1 << 12) && job.ah <= (1 << 12) &&
           job.ax > -(1 << 28) && job.ax < (1 << 28) && job.ay > -(1 << 28) && job.ay < (1 << 28) &&
           span_ok(job.desc_off, (long long)job.dw * job.dh, blob_bytes) &&
           span_ok(job.amask_off, (long long)job.aw * job.ah, blob_bytes) && job.threshold == job.threshold;
}

__global__ __launch_bounds__(SLAM_WG, SLAM_MIN_WAVES_LINE) void line_coarse_kernel(slam_pyr_dev pyr, const slam_line_job* __restrict__ jobs,
                                                        int njobs, const uint8_t* __restrict__ blob, unsigned long long blob_bytes,
                                                        LineBand* __restrict__ bands, int nb, int force_direct, int lds_total) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    lds_u8* lds = (lds_u8*)dyn_lds;
    const LdsPlan plan = make_lds_plan(lds_total);
    const int j = blockIdx.x, band = blockIdx.y;
    if (j >= njobs || band >= nb) return;
    const slam_line_job job = jobs[j];
    LineBand out = {0, 0, -1.f, 0};
    if (line_job_ok(pyr, job, blob_bytes)) {
        const int i = pyr.nlev - 1;
        float sub = 1.f; int lw = 0, lh = 0; long long lp = 0, ls = 0; const uint8_t* ld = nullptr;
#pragma unroll
        for (int l = 0; l < SLAM_HIP_MAX_LEVELS; l++)
            if (l == i) { sub = pyr.lev[l].sub; lw = pyr.lev[l].w; lh = pyr.lev[l].h; lp = pyr.lev[l].pitch; ls = pyr.lev[l].frame_stride; ld = pyr.lev[l].data; }
        Tmpl t;
        t.patch = blob + job.desc_off; t.mask = nullptr;
        t.srcw = job.dw; t.srch = job.dh;
        t.limit = blob + blob_bytes;
        const float areaRes = __fdiv_rn(1.f, sub);                                  // lineFeature.cpp:30
        t.tw = __builtin_amdgcn_readfirstlane(round_f2i(__fmul_rn((float)job.dw, areaRes)));   // lineFeature.cpp:124
        t.th = __builtin_amdgcn_readfirstlane(round_f2i(__fmul_rn((float)job.dh, areaRes)));
        if (t.tw > 0 && t.th > 0 && t.tw <= 2 * SLAM_MAX_TMPL && t.th <= 2 * SLAM_MAX_TMPL) {
            t.resized = !(t.tw == job.dw && t.th == job.dh);
            t.scale_x = 1.0; t.scale_y = 1.0;
            if (t.resized) { t.scale_x = resize_scale(job.dw, t.tw); t.scale_y = resize_scale(job.dh, t.th); }
            const uint8_t* im = ld + (long long)job.frame * ls;
            const uint8_t* am = blob + job.amask_off;
            const Geo g = make_geo(im, lw, lh, lp, t.tw, t.th, job.ax, job.ay, job.aw, job.ah, am, job.stride, false, false);
            const bool interior = !g.reject && g.xl0 == 0 && g.yl0 == 0 && g.xl1 == g.aw && g.yl1 == g.ah &&
                                  g.aw == job.aw && g.ah == job.ah && g.ah >= nb;
            if (interior) {
                const int y0 = small_div_u(band * g.ah, nb), y1 = small_div_u((band + 1) * g.ah, nb);
                const Geo gb = make_geo(im, lw, lh, lp, t.tw, t.th, job.ax, job.ay + y0, job.aw, y1 - y0,
                                        am + (long long)y0 * job.aw, job.stride, false, false);
                STAMP_DECL;
                const CoreOut o = match_core(gb, t, nullptr, lds, plan, force_direct != 0, false, 0.0 STAMP_PASS);
                out.x = o.x; out.y = o.y; out.response = o.response; out.valid = 1;
            }
        }
    }
    if (threadIdx.x == 0) bands[(size_t)j * nb + band] = out;
}

// The coarsest search of every line of a large batch (lineFeature.cpp:124-127) through lean_level's build with blocks of
// cell rows and chunks of template rows: one wavefront per line, the same band record as line_coarse_kernel (one band:
// the whole area).  Covered: areas whose candidate patches all lie inside the image (every cell is then scored by
// matchTemplate's rule whatever the stride or the area mask, matching.cpp:151-153); any other area, and any search
// lean_level turns away, is marked invalid and searched by line_track_kernel itself.
#ifndef SLAM_MIN_WAVES_LINE_LEAN
#define SLAM_MIN_WAVES_LINE_LEAN 4
#endif
__global__ __launch_bounds__(SLAM_WG, SLAM_MIN_WAVES_LINE_LEAN) void line_lean_coarse_kernel(slam_pyr_dev pyr, const slam_line_job* __restrict__ jobs,
                                                        int njobs, const uint8_t* __restrict__ blob, unsigned long long blob_bytes,
                                                        LineBand* __restrict__ bands, int lds_total) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    lds_u8* lds = (lds_u8*)dyn_lds;
    const int j = blockIdx.x;
    if (j >= njobs) return;
    const slam_line_job job = jobs[j];
    LineBand out = {0, 0, -1.f, 0};
    STAMP_DECL;
    if (line_job_ok(pyr, job, blob_bytes) && job.dw <= 255 && job.dh <= 255 && job.aw <= 255 && job.ah <= 255) {
        const int i = pyr.nlev - 1;
        float sub = 1.f; int lw = 0, lh = 0; long long lp = 0, ls = 0; const uint8_t* ld = nullptr;
#pragma unroll
        for (int l = 0; l < SLAM_HIP_MAX_LEVELS; l++)
            if (l == i) { sub = pyr.lev[l].sub; lw = pyr.lev[l].w; lh = pyr.lev[l].h; lp = pyr.lev[l].pitch; ls = pyr.lev[l].frame_stride; ld = pyr.lev[l].data; }
        const float areaRes = __fdiv_rn(1.f, sub);                                  // lineFeature.cpp:30
        LeanLevel L;
        L.tw = __builtin_amdgcn_readfirstlane(round_f2i(__fmul_rn((float)job.dw, areaRes)));   // lineFeature.cpp:124
        L.th = __builtin_amdgcn_readfirstlane(round_f2i(__fmul_rn((float)job.dh, areaRes)));
        if (L.tw > 0 && L.th > 0 && L.tw <= 255 && L.th <= 255) {
            L.im = ld + (long long)job.frame * ls;
            L.ipitch = (unsigned)lp; L.w = lw; L.h = lh;
            L.srcw = job.dw; L.srch = job.dh;
            L.resized = !(L.tw == job.dw && L.th == job.dh);  // cv::resize to the same size copies the image
            L.scx = 1.0; L.scy = 1.0;
            if (L.resized) { L.scx = resize_scale(job.dw, L.tw); L.scy = resize_scale(job.dh, L.th); }
            L.patch = blob + job.desc_off; L.mask = nullptr; L.limit = blob + blob_bytes;
            L.mt = true;
            L.own0 = 0; L.own1 = INT_MAX;
            L.x0 = job.ax - L.tw / 2; L.y0 = job.ay - L.th / 2;                       // matching.cpp:38-39
            L.aw = job.aw; L.ah = job.ah;
            L.step = 1;
            L.amask = nullptr;
            const bool ok = L.x0 >= 0 && L.y0 >= 0 && L.x0 + L.aw + L.tw - 1 <= L.w && L.y0 + L.ah + L.th - 1 <= L.h && lp <= 0x7fffffffLL &&
                            ((reinterpret_cast<unsigned long long>(L.im) | (unsigned long long)lp) & 3ull) == 0;
            if (ok) {
                const LeanOut o = lean_level<true, true>(L, lds, lds_total, false, 0.0 STAMP_PASS);
                if (o.ok) { out.x = o.x; out.y = o.y; out.response = o.response; out.valid = 1; }
            }
        }
    }
    if (threadIdx.x == 0) bands[j] = out;
}

// The searches of SLAM::LineFeature::track (lineFeature.cpp:120-165) for one line feature: the
// descriptor resized to the coarsest level and searched without a patch mask inside the area mask
// (lineFeature.cpp:124-127), then, while the response stays above 0.67*threshold, each finer level in
// a (2*searchRad+1)^2 window around the previous result (lineFeature.cpp:144-160).
__global__ __launch_bounds__(SLAM_WG, SLAM_MIN_WAVES_LINE) void line_track_kernel(slam_pyr_dev pyr, const slam_line_job* __restrict__ jobs,
                                                        int njobs, const uint8_t* __restrict__ blob,
                                                        unsigned long long blob_bytes,
                                                        slam_track_result* __restrict__ out, int force_direct, int lds_total,
                                                        const LineBand* __restrict__ bands, int nb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    lds_u8* lds = (lds_u8*)dyn_lds;
    const LdsPlan plan = make_lds_plan(lds_total);
    const int j = blockIdx.x;
    if (j >= njobs) return;
    const slam_line_job job = jobs[j];
    const int tid = threadIdx.x;
    const bool ok = line_job_ok(pyr, job, blob_bytes);
    if (!ok) {
        if (tid == 0) { slam_track_result r = {-1, -1, -1.f, 0.f, 0.f, 0, SLAM_HIP_EINVAL}; out[j] = r; }
        return;
    }
    Tmpl t;
    t.patch = blob + job.desc_off; t.mask = nullptr;
    t.srcw = job.dw; t.srch = job.dh;
    t.limit = blob + blob_bytes;
    const double keep_above = __dmul_rn((double)job.threshold, 0.67);   // lineFeature.cpp:144,158
    const float descw = (float)job.dw, desch = (float)job.dh;           // lineFeature.cpp:122

    int tpx = -1, tpy = -1, levels_run = 0;
    float response = -1.f, sdx = 0.f, sdy = 0.f;
    bool failed = false;
    float lastsub = 1.f;
    STAMP_DECL;
    for (int i = pyr.nlev - 1; i >= 0 && !failed; --i) {
        float sub = 1.f; int lw = 0, lh = 0; long long lp = 0, ls = 0; const uint8_t* ld = nullptr;
#pragma unroll
        for (int l = 0; l < SLAM_HIP_MAX_LEVELS; l++)
            if (l == i) { sub = pyr.lev[l].sub; lw = pyr.lev[l].w; lh = pyr.lev[l].h; lp = pyr.lev[l].pitch; ls = pyr.lev[l].frame_stride; ld = pyr.lev[l].data; }
        const bool coarsest = i == pyr.nlev - 1;
        int ax, ay, aw, ah, stride;
        const uint8_t* am = nullptr;
        float areaRes = 1.f;
        if (coarsest) {
            areaRes = __fdiv_rn(1.f, sub);                                         // lineFeature.cpp:30
            t.tw = round_f2i(__fmul_rn(descw, areaRes));                  // lineFeature.cpp:124
            t.th = round_f2i(__fmul_rn(desch, areaRes));
            ax = job.ax; ay = job.ay; aw = job.aw; ah = job.ah; am = blob + job.amask_off; stride = job.stride;
        } else {
            stride = (i == 0) ? 1 : job.stride;                                    // lineFeature.cpp:148-150
            const int searchRad = (int)__fdiv_rn(__fmul_rn(__fdiv_rn(sub, lastsub), (float)job.stride), (float)stride);
            ax = (int)__fsub_rn(__fdiv_rn((float)tpx, sub), (float)searchRad);    // lineFeature.cpp:151-152
            ay = (int)__fsub_rn(__fdiv_rn((float)tpy, sub), (float)searchRad);
            aw = ah = 2 * searchRad + 1;
            t.tw = (int)__fdiv_rn(descw, sub);                                     // lineFeature.cpp:153
            t.th = (int)__fdiv_rn(desch, sub);
        }
        ax = __builtin_amdgcn_readfirstlane(ax); ay = __builtin_amdgcn_readfirstlane(ay);
        t.tw = __builtin_amdgcn_readfirstlane(t.tw); t.th = __builtin_amdgcn_readfirstlane(t.th);
        if (t.tw <= 0 || t.th <= 0 || t.tw > 2 * SLAM_MAX_TMPL || t.th > 2 * SLAM_MAX_TMPL || aw < 1) { failed = true; break; }
        t.resized = !(t.tw == job.dw && t.th == job.dh);      // cv::resize to the same size copies the image
        t.scale_x = 1.0; t.scale_y = 1.0;
        if (t.resized) { t.scale_x = resize_scale(job.dw, t.tw); t.scale_y = resize_scale(job.dh, t.th); }
        const Geo g = make_geo(ld + (long long)job.frame * ls, lw, lh, lp, t.tw, t.th, ax, ay, aw, ah, am, stride, false, false);
        int mx = 0, my = 0;
        sdx = 0.f; sdy = 0.f;
        levels_run++;
        // the coarsest level searched beforehand: in bands by line_coarse_kernel (a small batch; all bands valid, or none)
        // or whole by line_lean_coarse_kernel (one band)
        bool banded = false;
        if (coarsest && bands && !g.reject) {
            const LineBand* bj = bands + (size_t)j * nb;
            banded = true;
            LineBand best = {0, 0, -FLT_MAX, 0};
            for (int k = 0; k < nb; k++) {                    // wave-uniform loads
                const LineBand bk = bj[k];
                banded = banded && bk.valid != 0;
                if (bk.response > best.response) best = bk;   // strict: the first band holding the maximum keeps it
            }
            if (banded) {
                mx = __builtin_amdgcn_readfirstlane(best.x); my = __builtin_amdgcn_readfirstlane(best.y);
                response = uni_f32(best.response);
                // If the walk ends here it reports this level's sub-pixel fit: the peak's neighbours may lie in another
                // band, so the (at most) three rows around it are searched once more as an area of their own - the
                // peak is their first maximum too, and their cells are the area's cells.
                if (pyr.nlev == 1 || !((double)response > keep_above)) {
                    const int ly = my - g.ay, r0 = max(ly - 1, 0), r1 = min(ly + 1, g.ah - 1);
                    const Geo g3 = make_geo(ld + (long long)job.frame * ls, lw, lh, lp, t.tw, t.th, ax, ay + r0, aw, r1 - r0 + 1,
                                            am + (long long)r0 * aw, stride, false, false);
                    const CoreOut o = match_core(g3, t, nullptr, lds, plan, force_direct != 0, true, 0.0 STAMP_PASS);
                    sdx = o.dx; sdy = o.dy;
                }
            }
        }
        if (banded) {}
        else if (g.reject) response = -1.f;                   // matching.cpp:26-30
        else {
            const CoreOut o = match_core(g, t, nullptr, lds, plan, force_direct != 0, true, 0.0 STAMP_PASS);
            mx = o.x; my = o.y; response = o.response; sdx = o.dx; sdy = o.dy;
        }
        if (coarsest) {                                       // lineFeature.cpp:130: Point2i *= double rounds to nearest even
            const double inv = __ddiv_rn(1.0, (double)areaRes);
            tpx = __double2int_rn(__dmul_rn((double)mx, inv));
            tpy = __double2int_rn(__dmul_rn((double)my, inv));
            if (!((double)response > keep_above)) break;      // lineFeature.cpp:144
        } else {                                              // lineFeature.cpp:157: Point2i *= float
            tpx = __float2int_rn(__fmul_rn((float)mx, sub));
            tpy = __float2int_rn(__fmul_rn((float)my, sub));
            if ((double)response < keep_above) break;         // lineFeature.cpp:158-159
        }
        lastsub = sub;
    }
    if (tid == 0) {
        slam_track_result r;
        if (failed) { r.x = -1; r.y = -1; r.response = -1.f; r.dx = 0.f; r.dy = 0.f; }
        else { r.x = tpx; r.y = tpy; r.response = response; r.dx = sdx; r.dy = sdy; }
        r.levels_run = levels_run;
        r.status = SLAM_HIP_OK;
        out[j] = r;
    }
}

__global__ void score_formula_kernel(const unsigned long long* cc, const unsigned long long* si,
                                     const unsigned long long* sp, int n, float* f32o, float* mto) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f32o[i] = score_f32(cc[i], si[i], sp[i]);
    mto[i] = score_mt(cc[i], si[i], sp[i]);
}

// After the launch of the general kernel behind a lean one.  The general kernel is what returns the queue to zero; if
// it could not be launched the lean kernel's entries would stay behind, so the queue is cleared here.
int slam_queue_launched(slam_ctx* ctx, int* queue, int njobs, hipError_t e) {
    if (e == hipSuccess) return SLAM_HIP_OK;
    (void)hipMemsetAsync(queue, 0, (SLAM_DEFER_LIST + (size_t)njobs) * sizeof(int), ctx->stream);
    return slam_set_error(ctx, SLAM_HIP_EHIP, "launch of the general kernel failed: %s", hipGetErrorString(e));
}

// Room for the band records of a small batch (line walks, point walks) in the context's buffer, grown on demand like
// the walk queue.  On failure the buffer is gone and the caller goes without bands.
int slam_bands_reserve(slam_ctx* ctx, size_t need) {
    if (need <= ctx->bands_bytes) return SLAM_HIP_OK;
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_bands) { SLAM_HIP_CHECK(ctx, hipFree(ctx->d_bands)); ctx->d_bands = nullptr; ctx->bands_bytes = 0; }
    const size_t want = need * 2 + 4096;
    if (hipMalloc(&ctx->d_bands, want) != hipSuccess) {
        (void)hipGetLastError();
        ctx->d_bands = nullptr;
        return SLAM_HIP_ENOMEM;
    }
    ctx->bands_bytes = want;
    return SLAM_HIP_OK;
}

int check_pyr(slam_ctx* ctx, const slam_pyr* pyr, const char* who) {
    if (!pyr || !pyr->built) return slam_set_error(ctx, SLAM_HIP_EINVAL, "%s: pyramid not built", who);
    if (pyr->device != ctx->device) return slam_set_error(ctx, SLAM_HIP_EINVAL, "%s: pyramid belongs to device %d", who, pyr->device);
    return SLAM_HIP_OK;
}

}  // namespace

// Queue storage of slam_hip_track_batch_noalloc_d: device memory that exists as long as the code object is loaded, one
// copy per GPU.  Host code never uses this name as an address; it asks the runtime for the device address once per
// device (slam_track_scratch_init) and passes that pointer to the kernels like any other queue.
#define SLAM_NOALLOC_MAX_JOBS (1 << 20)
static __device__ __align__(256) int g_walk_scratch[SLAM_DEFER_LIST + SLAM_NOALLOC_MAX_JOBS];

#define SLAM_MAX_DEVICES 64
static int* g_walk_scratch_ptr[SLAM_MAX_DEVICES] = {nullptr};
static std::mutex g_walk_scratch_mutex;      // contexts are created from several host threads (one per thread and device)

// Called by slam_hip_create with the context's device current.
int slam_track_scratch_init(slam_ctx* ctx) {
    if (ctx->device < 0 || ctx->device >= SLAM_MAX_DEVICES) return SLAM_HIP_OK;      // such a device simply has no no-alloc entry point
    std::lock_guard<std::mutex> lock(g_walk_scratch_mutex);
    if (g_walk_scratch_ptr[ctx->device]) return SLAM_HIP_OK;
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_walk_scratch));
    if (e != hipSuccess || !p) {
        (void)hipGetLastError();
        return slam_set_error(ctx, SLAM_HIP_EHIP, "hipGetSymbolAddress(walk scratch) on device %d: %s", ctx->device, hipGetErrorString(e));
    }
    g_walk_scratch_ptr[ctx->device] = (int*)p;
    return SLAM_HIP_OK;
}

#ifdef SLAM_STAMPS
extern "C" int slam_hip_debug_set_stamps(void* d_ptr) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_slam_stamps), &d_ptr, sizeof(void*)) == hipSuccess ? 0 : -2;
}
#endif

extern "C" int slam_hip_match_batch_d(slam_ctx* ctx, const slam_pyr* pyr, const slam_match_job* d_jobs,
                                      int njobs, const uint8_t* d_blob, size_t blob_bytes,
                                      float* d_scores, size_t scores_len, slam_match_result* d_out) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (njobs < 0 || (njobs > 0 && (!d_jobs || !d_blob || !d_out)))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "match_batch: null table or negative count");
    int rc = check_pyr(ctx, pyr, "match_batch");
    if (rc != SLAM_HIP_OK) return rc;
    if (njobs == 0) return SLAM_HIP_OK;
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // Small batches (single reference calls among them) go straight to the general kernel: the lean kernel's queue
    // costs a clear and a second launch, which only a batch repays.
    if (ctx->force_direct || ctx->no_lean || njobs < ctx->lean_match_min_jobs) {
        hipLaunchKernelGGL(match_kernel, dim3(njobs), dim3(SLAM_WG), ctx->match_lds_bytes, ctx->stream,
                           pyr->dev, d_jobs, njobs, d_blob, (unsigned long long)blob_bytes, d_scores,
                           (unsigned long long)scores_len, d_out, ctx->force_direct, ctx->match_lds_bytes, (int*)nullptr);
        SLAM_HIP_CHECK(ctx, hipGetLastError());
        return SLAM_HIP_OK;
    }
    rc = slam_queue_reserve(ctx, njobs);
    if (rc != SLAM_HIP_OK) return rc;
    int* queue = ctx->d_queue;
    ctx->queue_last = 0;
    hipLaunchKernelGGL(match_lean_kernel, dim3(njobs), dim3(SLAM_WG), ctx->lean_match_lds_bytes, ctx->stream,
                       pyr->dev, d_jobs, njobs, d_blob, (unsigned long long)blob_bytes, (const float*)d_scores, d_out,
                       ctx->lean_match_lds_bytes, queue);
    SLAM_HIP_CHECK(ctx, hipGetLastError());                // nothing was queued: the counters are still zero
    const int grid = njobs < 4096 ? njobs : 4096;
    hipLaunchKernelGGL(match_kernel, dim3(grid), dim3(SLAM_WG), ctx->match_lds_bytes, ctx->stream,
                       pyr->dev, d_jobs, njobs, d_blob, (unsigned long long)blob_bytes, d_scores,
                       (unsigned long long)scores_len, d_out, 0, ctx->match_lds_bytes, queue);
    return slam_queue_launched(ctx, queue, njobs, hipGetLastError());
}

static int mask_is_01(const uint8_t* m, size_t n) {
    for (size_t i = 0; i < n; i++) if (m[i] > 1) return 0;
    return 1;
}

static int match_batch_host(slam_ctx* ctx, const slam_pyr* pyr, const slam_match_job* jobs, int njobs,
                            const uint8_t* blob, size_t blob_bytes, float* h_scores, size_t scores_len,
                            slam_match_result* out) {
    if (njobs < 0 || (njobs > 0 && (!jobs || !blob || !out)))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "match_batch: null table or negative count");
    int rc = check_pyr(ctx, pyr, "match_batch");
    if (rc != SLAM_HIP_OK) return rc;
    if (njobs == 0) return SLAM_HIP_OK;
    for (int i = 0; i < njobs; i++) {   // the reference's mask contract: patch masks hold 0 or 1
        const slam_match_job& jb = jobs[i];
        if (jb.pmask_off != -1 && jb.pw >= 1 && jb.ph >= 1 && jb.pmask_off >= 0 &&
            (unsigned long long)jb.pmask_off + (unsigned long long)jb.pw * jb.ph <= blob_bytes &&
            !mask_is_01(blob + jb.pmask_off, (size_t)jb.pw * jb.ph))
            return slam_set_error(ctx, SLAM_HIP_EINVAL, "match_batch: job %d: patch mask values must be 0 or 1", i);
    }
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    slam_stream_drain drain(ctx);        // error returns below wait for the copies queued so far
    const size_t jb = sizeof(slam_match_job) * (size_t)njobs, rb = sizeof(slam_match_result) * (size_t)njobs;
    if ((rc = slam_stage_reserve(ctx, 0, jb)) || (rc = slam_stage_reserve(ctx, 1, blob_bytes + 16)) ||
        (rc = slam_stage_reserve(ctx, 2, rb)) || (h_scores && (rc = slam_stage_reserve(ctx, 3, scores_len * 4))))
        return rc;
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_stage[0], jobs, jb, hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_stage[1], blob, blob_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = slam_hip_match_batch_d(ctx, pyr, (const slam_match_job*)ctx->d_stage[0], njobs,
                                (const uint8_t*)ctx->d_stage[1], blob_bytes,
                                h_scores ? (float*)ctx->d_stage[3] : nullptr, scores_len,
                                (slam_match_result*)ctx->d_stage[2]);
    if (rc != SLAM_HIP_OK) return rc;
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(out, ctx->d_stage[2], rb, hipMemcpyDeviceToHost, ctx->stream));
    if (h_scores) SLAM_HIP_CHECK(ctx, hipMemcpyAsync(h_scores, ctx->d_stage[3], scores_len * 4, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_match_batch(slam_ctx* ctx, const slam_pyr* pyr, const slam_match_job* jobs, int njobs,
                                    const uint8_t* blob, size_t blob_bytes, slam_match_result* out) {
    if (!ctx) return SLAM_HIP_EINVAL;
    return match_batch_host(ctx, pyr, jobs, njobs, blob, blob_bytes, nullptr, 0, out);
}

extern "C" int slam_hip_match_feature_in_area(slam_ctx* ctx,
                                              const uint8_t* im, int w, int h, size_t im_pitch,
                                              const uint8_t* patch, int pw, int ph, size_t patch_pitch,
                                              const uint8_t* patch_mask, size_t pmask_pitch,
                                              int area_x, int area_y, int area_w, int area_h,
                                              const uint8_t* area_mask, size_t amask_pitch,
                                              int stride, int use_exact_area_mask,
                                              float* h_scores, slam_match_result* out) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (!im || !patch || !out || w < 1 || h < 1 || pw < 1 || ph < 1 || stride < 1 ||
        im_pitch < (size_t)w || patch_pitch < (size_t)pw || (patch_mask && pmask_pitch < (size_t)pw) ||
        (area_mask && (area_w < 1 || area_h < 1 || amask_pitch < (size_t)area_w)))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "match_feature_in_area: bad arguments");
    // One-level pyramid kept in the context and reused while the image size stays the same: a call costs the upload of
    // the image and of the job, no device allocation.
    int rc = SLAM_HIP_OK;
    if (!ctx->mfa_pyr || ctx->mfa_w != w || ctx->mfa_h != h) {
        if (ctx->mfa_pyr) { slam_hip_pyramid_destroy(ctx, ctx->mfa_pyr); ctx->mfa_pyr = nullptr; }
        const float one = 1.f;
        rc = slam_hip_pyramid_create(ctx, w, h, 1, &one, 1, &ctx->mfa_pyr);
        if (rc != SLAM_HIP_OK) { ctx->mfa_pyr = nullptr; return rc; }
        ctx->mfa_w = w; ctx->mfa_h = h;
    }
    slam_pyr* pyr = ctx->mfa_pyr;
    rc = slam_hip_pyramid_build(ctx, pyr, im, im_pitch, im_pitch * (size_t)h);
    if (rc != SLAM_HIP_OK) return rc;
    const size_t pb = (size_t)pw * ph, ab = area_mask ? (size_t)area_w * area_h : 0;
    std::vector<uint8_t> blob(pb * (patch_mask ? 2 : 1) + ab);
    for (int r = 0; r < ph; r++) memcpy(&blob[(size_t)r * pw], patch + (size_t)r * patch_pitch, (size_t)pw);
    if (patch_mask) for (int r = 0; r < ph; r++) memcpy(&blob[pb + (size_t)r * pw], patch_mask + (size_t)r * pmask_pitch, (size_t)pw);
    const size_t aoff = pb * (patch_mask ? 2 : 1);
    if (area_mask) for (int r = 0; r < area_h; r++) memcpy(&blob[aoff + (size_t)r * area_w], area_mask + (size_t)r * amask_pitch, (size_t)area_w);
    slam_match_job job;
    job.frame = 0; job.level = 0; job.pw = pw; job.ph = ph;
    job.patch_off = 0; job.pmask_off = patch_mask ? (int64_t)pb : -1;
    job.ax = area_x; job.ay = area_y; job.aw = area_w; job.ah = area_h;
    job.amask_off = area_mask ? (int64_t)aoff : -1;
    job.stride = stride; job.use_exact_area_mask = use_exact_area_mask;
    job.score_off = h_scores ? 0 : -1;
    const size_t slen = h_scores ? (size_t)(area_w > 0 ? area_w : 0) * (size_t)(area_h > 0 ? area_h : 0) : 0;
    // match_batch_host synchronises before it returns: the stack job and the local blob outlive their copies
    return match_batch_host(ctx, pyr, &job, 1, blob.data(), blob.size(), slen ? h_scores : nullptr, slen, out);
}

extern "C" int slam_hip_track_batch_d(slam_ctx* ctx, const slam_pyr* pyr, const slam_track_job* d_jobs,
                                      int njobs, const uint8_t* d_blob, size_t blob_bytes,
                                      slam_track_result* d_out) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (njobs < 0 || (njobs > 0 && (!d_jobs || !d_blob || !d_out)))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "track_batch: null table or negative count");
    int rc = check_pyr(ctx, pyr, "track_batch");
    if (rc != SLAM_HIP_OK) return rc;
    if (njobs == 0) return SLAM_HIP_OK;
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (ctx->force_direct || ctx->no_lean) {               // diagnostic settings: every job through the general kernel
        ctx->lean_lds_once = 0; ctx->walk_chunks_once = 0;
        hipLaunchKernelGGL(track_kernel, dim3(njobs), dim3(SLAM_WG), ctx->match_lds_bytes,
                           ctx->stream, pyr->dev, d_jobs, njobs, d_blob, (unsigned long long)blob_bytes, d_out,
                           ctx->force_direct, ctx->match_lds_bytes, (int*)nullptr);
        SLAM_HIP_CHECK(ctx, hipGetLastError());
        return SLAM_HIP_OK;
    }
    int rc2 = slam_queue_reserve(ctx, njobs);
    if (rc2 != SLAM_HIP_OK) { ctx->lean_lds_once = 0; ctx->walk_chunks_once = 0; return rc2; }
    ctx->queue_last = 0;
    return slam_launch_track(ctx, pyr, d_jobs, njobs, d_blob, blob_bytes, d_out, ctx->d_queue, false);
}

// The entry points that take their job table from the host see the shapes before the launch: they size the lean walk
// kernel's LDS to the largest need among the walks it could take (the host-side replica of lean_walk's fit check;
// sizing only - the device decides), within [default, 24 KB], for that launch alone.  Walks with a level beyond 24 KB
// that the kernel's other build can take in blocks of cell rows and chunks of template rows (lean_plan) select that
// build, with the first budget of a short list at which every chunk has SLAM_AUTO_CHUNK_ROWS rows or more.  A
// budget or build set by the caller (slam_hip_set_walk_lds, slam_hip_set_walk_chunks, SLAM_HIP_WALK_LDS_KB) is left alone.
#define SLAM_AUTO_CHUNK_ROWS 10
void slam_size_walk_lds(slam_ctx* ctx, const slam_pyr* pyr, const slam_track_job* jobs, int njobs) {
    ctx->lean_lds_once = 0;
    ctx->walk_chunks_once = 0;
    if (ctx->lean_lds_user || !pyr) return;
    const int nlev = pyr->dev.nlev;
    struct Shape { int tw, th, aw, ah, step, nx, ny; bool resized; };
    int need = 0, need_chunked = 0;
    for (int i = 0; i < njobs; i++) {
        const slam_track_job& jb = jobs[i];
        if (jb.projw < 1 || jb.projh < 1 || jb.projw > 255 || jb.projh > 255 || jb.stride < 1 || jb.stride > 1024) continue;
        const bool coarse_area = (jb.flags & SLAM_HIP_TRACK_COARSE_AREA) != 0;
        Shape sh[SLAM_HIP_MAX_LEVELS];
        int worst = 0;
        for (int q = 0; q < nlev; q++) {
            const slam_level_dev& lv = pyr->dev.lev[q];
            const int tw = lv.sub == 1.f ? jb.projw : (int)((float)jb.projw / lv.sub);
            const int th = lv.sub == 1.f ? jb.projh : (int)((float)jb.projh / lv.sub);
            if (tw < 1 || th < 1) { worst = INT_MAX; break; }
            float rad = lv.rad;
            if (q == 0 && nlev > 1) rad = rad * (float)jb.stride;
            int aw = (int)roundf(2.f * rad + 1.f), ah = aw;
            if (coarse_area && q == nlev - 1) { aw = jb.aw; ah = jb.ah; }
            if (aw < 1 || ah < 1 || aw > 255 || ah > 255) { worst = INT_MAX; break; }
            const int step = q == 0 ? 1 : jb.stride;
            const int nx = (aw + step - 1) / step, ny = (ah + step - 1) / step;
            sh[q] = Shape{tw, th, aw, ah, step, nx, ny, lv.sub != 1.f};
            const int nd = lean_need(tw, th, aw, ah, step, nx, ny, lv.sub != 1.f);
            worst = std::max(worst, nd);
        }
        if (worst == INT_MAX) continue;                    // a shape the lean kernels do not take at any budget
        if (worst <= 24576) { need = std::max(need, worst); continue; }
        // budgets in the order of preference measured on the reference's live shape (81x81 descriptors, 24x24 windows):
        // 10 granules (12 walks per CU) before 12 and 8
        static const int order[] = {12800, 15360, 10240, 17920, 20480, 24320};
        for (int b : order) {
            if (b < need_chunked || b < ctx->lean_lds_bytes) continue;
            bool all = true;
            for (int q = 0; q < nlev && all; q++) {
                const Shape& h = sh[q];
                const LeanPlan pl = lean_plan(h.tw, h.th, h.aw, h.ah, h.step, h.nx, h.ny, h.resized, true, b,
                                              lean_box_mode(false, true, h.step, h.resized, h.tw, h.th, h.aw));
                all = pl.bh > 0 && pl.R >= std::min(SLAM_AUTO_CHUNK_ROWS, h.th);
            }
            if (all) { need_chunked = std::max(need_chunked, b); break; }
        }                                                  // (none found: the walk goes to the general kernel)
    }
    if (need_chunked > 0) { need = std::max(need, need_chunked); ctx->walk_chunks_once = 1; }
    const int rounded = (need + 1279) / 1280 * 1280;        // whole LDS granules (slam_internal.h)
    if (rounded > ctx->lean_lds_bytes) ctx->lean_lds_once = rounded;
}

// Lean kernel over all jobs, then the general kernel over whatever it queued.  `queue`: device words, SLAM_DEFER_LIST
// + njobs of them.  Allocates nothing.
int slam_launch_track(slam_ctx* ctx, const slam_pyr* pyr, const slam_track_job* d_jobs, int njobs, const uint8_t* d_blob,
                      size_t blob_bytes, slam_track_result* d_out, int* queue, bool clear_first) {
    const int lds = ctx->lean_lds_once ? ctx->lean_lds_once : ctx->lean_lds_bytes;      // a host-table entry point may have sized this launch
    const bool chunks = ctx->walk_chunks_once ? true : ctx->walk_chunks != 0;
    ctx->lean_lds_once = 0;
    ctx->walk_chunks_once = 0;
    // a queue shared by every context of the device (the no-alloc entry point) is cleared first: another context's
    // failed or overlapping launch must not leave its count or its entries behind
    if (clear_first) SLAM_HIP_CHECK(ctx, hipMemsetAsync(queue, 0, (SLAM_DEFER_LIST + (size_t)njobs) * sizeof(int), ctx->stream));
    // A small batch under the second build (large templates) has the last level of every walk cut into bands of rows
    // (track_band_kernel + track_band_finish_kernel); not where nothing may be allocated (the bands' records live in a
    // buffer of the context that grows on demand).
    int nb = 1;
    if (chunks && !clear_first) {
        nb = ctx->walk_bands > 0 ? ctx->walk_bands : (njobs <= 256 ? 2048 / njobs : 1);
        nb = nb < 1 ? 1 : (nb > 8 ? 8 : nb);
        if (nb > 1 && slam_bands_reserve(ctx, (size_t)njobs * nb * sizeof(WalkBand)) != SLAM_HIP_OK) nb = 1;
    }
    if (nb > 1) {
        // (a batch this small does not fill the chip: LDS per wavefront is no concern, fewer chunks are what counts)
        const int bl = ctx->walk_band_lds_bytes > lds ? ctx->walk_band_lds_bytes : lds;
        hipLaunchKernelGGL(track_band_kernel, dim3(njobs * nb), dim3(SLAM_WG), bl, ctx->stream, pyr->dev, d_jobs, njobs,
                           d_blob, (unsigned long long)blob_bytes, (WalkBand*)ctx->d_bands, nb, bl);
        SLAM_HIP_CHECK(ctx, hipGetLastError());
        hipLaunchKernelGGL(track_band_finish_kernel, dim3((njobs + SLAM_WG - 1) / SLAM_WG), dim3(SLAM_WG), 0, ctx->stream,
                           (const WalkBand*)ctx->d_bands, njobs, nb, d_out, queue);
    } else
    // Two builds of the lean kernel: the plain one takes walks whose every level fits the budget whole; the other also
    // takes full-resolution levels with templates too large for that, in blocks of cell rows and chunks of template rows (lean_plan), at
    // the price of more registers - the caller, or slam_size_walk_lds for a host job table, picks it for such shapes.
    if (chunks)
        hipLaunchKernelGGL(track_lean_kernel<true>, dim3(8 * ((njobs + 7) / 8)), dim3(SLAM_WG), lds, ctx->stream, pyr->dev, d_jobs, njobs,
                           d_blob, (unsigned long long)blob_bytes, d_out, lds, queue);
    else
        hipLaunchKernelGGL(track_lean_kernel<false>, dim3(8 * ((njobs + 7) / 8)), dim3(SLAM_WG), lds, ctx->stream, pyr->dev, d_jobs, njobs,
                           d_blob, (unsigned long long)blob_bytes, d_out, lds, queue);
    SLAM_HIP_CHECK(ctx, hipGetLastError());                // nothing was queued: the counters are still zero
    // enough workgroups to fill the chip if everything was queued, few enough to start and end in microseconds if nothing was
    const int grid = njobs < 4096 ? njobs : 4096;
    hipLaunchKernelGGL(track_kernel, dim3(grid), dim3(SLAM_WG), ctx->match_lds_bytes,
                       ctx->stream, pyr->dev, d_jobs, njobs, d_blob, (unsigned long long)blob_bytes, d_out,
                       0, ctx->match_lds_bytes, queue);
    return slam_queue_launched(ctx, queue, njobs, hipGetLastError());
}

extern "C" int slam_hip_track_batch_noalloc_d(slam_ctx* ctx, const slam_pyr* pyr, const slam_track_job* d_jobs,
                                              int njobs, const uint8_t* d_blob, size_t blob_bytes,
                                              slam_track_result* d_out) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (njobs < 0 || (njobs > 0 && (!d_jobs || !d_blob || !d_out)))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "track_batch_noalloc: null table or negative count");
    if (njobs > SLAM_NOALLOC_MAX_JOBS)
        return slam_set_error(ctx, SLAM_HIP_ECAP, "track_batch_noalloc: %d jobs, at most %d per call", njobs, SLAM_NOALLOC_MAX_JOBS);
    int rc = check_pyr(ctx, pyr, "track_batch_noalloc");
    if (rc != SLAM_HIP_OK) return rc;
    if (njobs == 0) return SLAM_HIP_OK;
    int* q = (ctx->device >= 0 && ctx->device < SLAM_MAX_DEVICES) ? g_walk_scratch_ptr[ctx->device] : nullptr;
    if (!q) return slam_set_error(ctx, SLAM_HIP_EINVAL, "track_batch_noalloc: no scratch on device %d", ctx->device);
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (ctx->force_direct || ctx->no_lean) {               // diagnostic settings: every job through the general kernel, no queue
        ctx->lean_lds_once = 0; ctx->walk_chunks_once = 0;
        hipLaunchKernelGGL(track_kernel, dim3(njobs), dim3(SLAM_WG), ctx->match_lds_bytes,
                           ctx->stream, pyr->dev, d_jobs, njobs, d_blob, (unsigned long long)blob_bytes, d_out,
                           ctx->force_direct, ctx->match_lds_bytes, (int*)nullptr);
        SLAM_HIP_CHECK(ctx, hipGetLastError());
        return SLAM_HIP_OK;
    }
    ctx->queue_last = -1;
    return slam_launch_track(ctx, pyr, d_jobs, njobs, d_blob, blob_bytes, d_out, q, true);
}

extern "C" int slam_hip_track_batch(slam_ctx* ctx, const slam_pyr* pyr, const slam_track_job* jobs, int njobs,
                                    const uint8_t* blob, size_t blob_bytes, slam_track_result* out) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (njobs < 0 || (njobs > 0 && (!jobs || !blob || !out)))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "track_batch: null table or negative count");
    int rc = check_pyr(ctx, pyr, "track_batch");
    if (rc != SLAM_HIP_OK) return rc;
    if (njobs == 0) return SLAM_HIP_OK;
    for (int i = 0; i < njobs; i++) {
        const slam_track_job& jb = jobs[i];
        if (jb.projw >= 1 && jb.projh >= 1 && jb.mask_off >= 0 &&
            (unsigned long long)jb.mask_off + (unsigned long long)jb.projw * jb.projh <= blob_bytes &&
            !mask_is_01(blob + jb.mask_off, (size_t)jb.projw * jb.projh))
            return slam_set_error(ctx, SLAM_HIP_EINVAL, "track_batch: job %d: mask values must be 0 or 1", i);
    }
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    slam_stream_drain drain(ctx);        // error returns below wait for the copies queued so far
    const size_t jb = sizeof(slam_track_job) * (size_t)njobs, rb = sizeof(slam_track_result) * (size_t)njobs;
    if ((rc = slam_stage_reserve(ctx, 0, jb)) || (rc = slam_stage_reserve(ctx, 1, blob_bytes + 16)) ||
        (rc = slam_stage_reserve(ctx, 2, rb)))
        return rc;
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_stage[0], jobs, jb, hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_stage[1], blob, blob_bytes, hipMemcpyHostToDevice, ctx->stream));
    slam_size_walk_lds(ctx, pyr, jobs, njobs);
    rc = slam_hip_track_batch_d(ctx, pyr, (const slam_track_job*)ctx->d_stage[0], njobs,
                                (const uint8_t*)ctx->d_stage[1], blob_bytes, (slam_track_result*)ctx->d_stage[2]);
    if (rc != SLAM_HIP_OK) return rc;
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(out, ctx->d_stage[2], rb, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_line_track_batch_d(slam_ctx* ctx, const slam_pyr* pyr, const slam_line_job* d_jobs,
                                           int njobs, const uint8_t* d_blob, size_t blob_bytes,
                                           slam_track_result* d_out) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (njobs < 0 || (njobs > 0 && (!d_jobs || !d_blob || !d_out)))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "line_track_batch: null table or negative count");
    int rc = check_pyr(ctx, pyr, "line_track_batch");
    if (rc != SLAM_HIP_OK) return rc;
    if (njobs == 0) return SLAM_HIP_OK;
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // A batch too small to fill the chip with one wavefront per line (the kernel's registers allow three per SIMD:
    // 3072 at once) has its coarsest searches cut into bands of rows, up to 8 per line (line_coarse_kernel).
    int nb = ctx->line_bands > 0 ? ctx->line_bands : 3072 / njobs;
    nb = nb < 1 ? 1 : (nb > 8 ? 8 : nb);
    // A large batch has them searched by the lean core first (line_lean_coarse_kernel): one band per line.
    const bool lean = nb == 1 && !ctx->force_direct && !ctx->no_lean && njobs >= ctx->line_lean_min_jobs;
    LineBand* d_bands = nullptr;
    if (nb > 1 || lean) {
        if (slam_bands_reserve(ctx, (size_t)njobs * nb * sizeof(LineBand)) != SLAM_HIP_OK) nb = 1;     // no room for the bands: one wavefront per line
        d_bands = (LineBand*)ctx->d_bands;
    }
    if (lean && d_bands) {
        hipLaunchKernelGGL(line_lean_coarse_kernel, dim3(njobs), dim3(SLAM_WG), ctx->line_lean_lds_bytes, ctx->stream, pyr->dev, d_jobs,
                           njobs, d_blob, (unsigned long long)blob_bytes, d_bands, ctx->line_lean_lds_bytes);
        SLAM_HIP_CHECK(ctx, hipGetLastError());
    } else if (nb > 1) {
        hipLaunchKernelGGL(line_coarse_kernel, dim3(njobs, nb), dim3(SLAM_WG), ctx->line_lds_bytes, ctx->stream, pyr->dev, d_jobs,
                           njobs, d_blob, (unsigned long long)blob_bytes, d_bands, nb, ctx->force_direct, ctx->line_lds_bytes);
        SLAM_HIP_CHECK(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(line_track_kernel, dim3(njobs), dim3(SLAM_WG), ctx->line_lds_bytes,
                       ctx->stream, pyr->dev, d_jobs, njobs, d_blob, (unsigned long long)blob_bytes, d_out,
                       ctx->force_direct, ctx->line_lds_bytes, (const LineBand*)d_bands, nb);
    SLAM_HIP_CHECK(ctx, hipGetLastError());
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_line_track_batch(slam_ctx* ctx, const slam_pyr* pyr, const slam_line_job* jobs, int njobs,
                                         const uint8_t* blob, size_t blob_bytes, slam_track_result* out) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (njobs < 0 || (njobs > 0 && (!jobs || !blob || !out)))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "line_track_batch: null table or negative count");
    int rc = check_pyr(ctx, pyr, "line_track_batch");
    if (rc != SLAM_HIP_OK) return rc;
    if (njobs == 0) return SLAM_HIP_OK;
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    slam_stream_drain drain(ctx);        // error returns below wait for the copies queued so far
    const size_t jb = sizeof(slam_line_job) * (size_t)njobs, rb = sizeof(slam_track_result) * (size_t)njobs;
    if ((rc = slam_stage_reserve(ctx, 0, jb)) || (rc = slam_stage_reserve(ctx, 1, blob_bytes + 16)) ||
        (rc = slam_stage_reserve(ctx, 2, rb)))
        return rc;
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_stage[0], jobs, jb, hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_stage[1], blob, blob_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = slam_hip_line_track_batch_d(ctx, pyr, (const slam_line_job*)ctx->d_stage[0], njobs,
                                     (const uint8_t*)ctx->d_stage[1], blob_bytes, (slam_track_result*)ctx->d_stage[2]);
    if (rc != SLAM_HIP_OK) return rc;
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(out, ctx->d_stage[2], rb, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_debug_walk_plan(int tw, int th, int aw, int ah, int step, int resized, int budget, int chunks, int out[4]) {
    if (!out || tw < 1 || th < 1 || aw < 1 || ah < 1 || step < 1 || tw > 255 || th > 255 || aw > 255 || ah > 255 || budget < 0)
        return SLAM_HIP_EINVAL;
    const int nx = (aw + step - 1) / step, ny = (ah + step - 1) / step;
    out[0] = out[1] = out[2] = out[3] = 0;
    if (!chunks) {
        const int need = lean_need(tw, th, aw, ah, step, nx, ny, resized != 0);
        if (need != INT_MAX && need <= budget) { out[0] = step == 1 ? ah : ny; out[1] = th; out[2] = need; }
        return SLAM_HIP_OK;
    }
    const int box = lean_box_mode(false, true, step, resized != 0, tw, th, aw);
    const LeanPlan pl = lean_plan(tw, th, aw, ah, step, nx, ny, resized != 0, true, budget, box);
    if (pl.bh > 0) {
        const int G = (tw + 3) >> 2, cols = (aw + 3) >> 2, tp = (cols + G - 1) | 1;
        const bool strips = step == 1;
        out[0] = pl.bh; out[1] = pl.R; out[3] = box;
        out[2] = lean_plan_bytes(G, tp, strips ? cols : nx, strips ? 4 : 1, strips ? 1 : step, strips ? ah : ny, pl.bh, pl.R, resized != 0, true, th, box);
    }
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_debug_walk_deferred(slam_ctx* ctx, int* out_deferred) {
    if (!ctx || !out_deferred) return SLAM_HIP_EINVAL;
    *out_deferred = 0;
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->force_direct || ctx->no_lean) return SLAM_HIP_OK;
    const int* queue = nullptr;                             // of the context's last launch with a queue
    if (ctx->queue_last >= 0) queue = ctx->d_queue;
    else if (ctx->device >= 0 && ctx->device < SLAM_MAX_DEVICES) queue = g_walk_scratch_ptr[ctx->device];
    if (!queue) return SLAM_HIP_OK;
    SLAM_HIP_CHECK(ctx, hipMemcpy(out_deferred, queue + 2, sizeof(int), hipMemcpyDeviceToHost));   // word 2: the last launch's count
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_debug_score_formulas(slam_ctx* ctx, const uint64_t* cc, const uint64_t* si,
                                             const uint64_t* sp, int n, float* f32_out, float* mt_out) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (n < 0 || (n > 0 && (!cc || !si || !sp || !f32_out || !mt_out)))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "debug_score_formulas: bad arguments");
    if (n == 0) return SLAM_HIP_OK;
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    slam_stream_drain drain(ctx);        // error returns below wait for the copies queued so far
    const size_t b8 = (size_t)n * 8, b4 = (size_t)n * 4;
    int rc = slam_stage_reserve(ctx, 0, 3 * b8 + 2 * b4);
    if (rc != SLAM_HIP_OK) return rc;
    uint8_t* base = (uint8_t*)ctx->d_stage[0];
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(base, cc, b8, hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(base + b8, si, b8, hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(base + 2 * b8, sp, b8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(score_formula_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream,
                       (const unsigned long long*)base, (const unsigned long long*)(base + b8),
                       (const unsigned long long*)(base + 2 * b8), n, (float*)(base + 3 * b8), (float*)(base + 3 * b8 + b4));
    SLAM_HIP_CHECK(ctx, hipGetLastError());
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(f32_out, base + 3 * b8, b4, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(mt_out, base + 3 * b8 + b4, b4, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_HIP_OK;
}
