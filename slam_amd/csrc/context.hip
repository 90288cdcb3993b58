// Context, error reporting, device-memory helpers and host-side resize tap tables.
#include <cmath>
#include <cstdlib>

#include "slam_internal.h"

static char g_create_err[512] = "";

int slam_set_error(slam_ctx* ctx, int code, const char* fmt, ...) {
    char* dst = ctx ? ctx->err : g_create_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* slam_hip_version(void) { return "slam_hip 0.1 gfx950"; }

extern "C" const char* slam_hip_last_error_string(const slam_ctx* ctx) {
    return ctx ? ctx->err : g_create_err;
}

extern "C" int slam_hip_create(int device, slam_ctx** out) {
    if (!out) return slam_set_error(nullptr, SLAM_HIP_EINVAL, "slam_hip_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return slam_set_error(nullptr, SLAM_HIP_EHIP, "slam_hip_create: no HIP device (%s)",
                              hipGetErrorString(e));
    if (device < 0 || device >= ndev)
        return slam_set_error(nullptr, SLAM_HIP_EINVAL, "slam_hip_create: device %d of %d", device, ndev);
    e = hipSetDevice(device);
    if (e != hipSuccess)
        return slam_set_error(nullptr, SLAM_HIP_EHIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    slam_ctx* c = new (std::nothrow) slam_ctx();
    if (!c) return slam_set_error(nullptr, SLAM_HIP_ENOMEM, "slam_hip_create: out of host memory");
    c->device = device;
    e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return slam_set_error(nullptr, SLAM_HIP_EHIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    c->stream = c->own_stream;
    const char* fd = getenv("SLAM_HIP_FORCE_DIRECT");
    c->force_direct = (fd && fd[0] == '1') ? 1 : 0;
    const char* nl = getenv("SLAM_HIP_NO_LEAN");
    c->no_lean = (nl && nl[0] == '1') ? 1 : 0;
    const char* wk = getenv("SLAM_HIP_WALK_LDS_KB");
    if (wk && atoi(wk) >= 4 && atoi(wk) <= 64) { c->lean_lds_bytes = atoi(wk) * 1024; c->lean_lds_user = 1; }
    const char* wb = getenv("SLAM_HIP_WALK_LDS_BYTES");
    if (wb && atoi(wb) >= 4096 && atoi(wb) <= 65536) { c->lean_lds_bytes = atoi(wb) & ~63; c->lean_lds_user = 1; }
    const char* ll = getenv("SLAM_HIP_LINE_LEAN_LDS_BYTES");
    if (ll && atoi(ll) >= 4096 && atoi(ll) <= 65536) c->line_lean_lds_bytes = atoi(ll) & ~63;
    const char* lm = getenv("SLAM_HIP_LINE_LEAN_MIN");
    if (lm && atoi(lm) >= 1) c->line_lean_min_jobs = atoi(lm);
    const char* wbl = getenv("SLAM_HIP_WALK_BAND_LDS_BYTES");
    if (wbl && atoi(wbl) >= 4096 && atoi(wbl) <= 65536) c->walk_band_lds_bytes = atoi(wbl) & ~63;
    const char* wbn = getenv("SLAM_HIP_WALK_BANDS");
    if (wbn && atoi(wbn) >= 1 && atoi(wbn) <= 8) c->walk_bands = atoi(wbn);
    const char* wc = getenv("SLAM_HIP_WALK_CHUNKS");
    if (wc && (wc[0] == '0' || wc[0] == '1')) c->walk_chunks = wc[0] - '0';
    const char* sk = getenv("SLAM_HIP_SEARCH_LDS_KB");
    if (sk && atoi(sk) >= 4 && atoi(sk) <= 64) c->lean_match_lds_bytes = atoi(sk) * 1024;
    const char* sb = getenv("SLAM_HIP_SEARCH_LDS_BYTES");
    if (sb && atoi(sb) >= 4096 && atoi(sb) <= 65536) c->lean_match_lds_bytes = atoi(sb) & ~63;
    const char* sm = getenv("SLAM_HIP_SEARCH_LEAN_MIN");
    if (sm && atoi(sm) >= 1) c->lean_match_min_jobs = atoi(sm);
    const char* lk = getenv("SLAM_HIP_MATCH_LDS_KB");
    if (lk && atoi(lk) >= 4 && atoi(lk) <= 63) c->match_lds_bytes = c->line_lds_bytes = atoi(lk) * 1024;
    const char* lbn = getenv("SLAM_HIP_LINE_BANDS");
    if (lbn && atoi(lbn) >= 1 && atoi(lbn) <= 8) c->line_bands = atoi(lbn);
    const int rc = slam_track_scratch_init(c);             // this device's copy of the no-alloc walk queue
    if (rc != SLAM_HIP_OK) {
        strncpy(g_create_err, c->err, sizeof(g_create_err) - 1);
        (void)hipStreamDestroy(c->own_stream);
        delete c;
        return rc;
    }
    *out = c;
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_destroy(slam_ctx* ctx) {
    if (!ctx) return SLAM_HIP_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < 4; i++)
        if (ctx->d_stage[i]) (void)hipFree(ctx->d_stage[i]);
    if (ctx->mfa_pyr) { slam_hip_pyramid_destroy(ctx, ctx->mfa_pyr); ctx->mfa_pyr = nullptr; }
    if (ctx->d_queue) (void)hipFree(ctx->d_queue);
    if (ctx->d_bands) (void)hipFree(ctx->d_bands);
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_set_stream(slam_ctx* ctx, void* hip_stream) {
    if (!ctx) return SLAM_HIP_EINVAL;
    hipStream_t next = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    if (next == ctx->stream) return SLAM_HIP_OK;
    // the context's queue, staging and pinned buffers are ordered by one stream: work on the old one ends first
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = next;
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_set_match_lds(slam_ctx* ctx, int bytes) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (bytes < 4096 || bytes > 64512 || (bytes & 63))        // the walk kernel adds 576 B of static LDS: 64 KB per workgroup in all
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "set_match_lds: %d not a multiple of 64 in [4096, 64512]", bytes);
    ctx->match_lds_bytes = bytes;
    ctx->line_lds_bytes = bytes;
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_set_walk_lds(slam_ctx* ctx, int bytes) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (bytes < 4096 || bytes > 65536 || (bytes & 63))
        return slam_set_error(ctx, SLAM_HIP_EINVAL, "set_walk_lds: %d not a multiple of 64 in [4096, 65536]", bytes);
    ctx->lean_lds_bytes = bytes;
    ctx->lean_lds_user = 1;
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_set_walk_chunks(slam_ctx* ctx, int on) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (on != 0 && on != 1) return slam_set_error(ctx, SLAM_HIP_EINVAL, "set_walk_chunks: %d is neither 0 nor 1", on);
    ctx->walk_chunks = on;
    ctx->lean_lds_user = 1;            // the caller picks budget and kernel: no sizing per launch
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_synchronize(slam_ctx* ctx) {
    if (!ctx) return SLAM_HIP_EINVAL;
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_malloc(slam_ctx* ctx, size_t bytes, void** d_ptr) {
    if (!ctx || !d_ptr) return SLAM_HIP_EINVAL;
    *d_ptr = nullptr;
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 1);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return slam_set_error(ctx, SLAM_HIP_ENOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    }
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_free(slam_ctx* ctx, void* d_ptr) {
    if (!ctx) return SLAM_HIP_EINVAL;
    if (!d_ptr) return SLAM_HIP_OK;
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    SLAM_HIP_CHECK(ctx, hipFree(d_ptr));
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_memcpy_h2d(slam_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    if (!ctx || (bytes && (!d_dst || !h_src))) return SLAM_HIP_EINVAL;
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_HIP_OK;
}

extern "C" int slam_hip_memcpy_d2h(slam_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
    if (!ctx || (bytes && (!h_dst || !d_src))) return SLAM_HIP_EINVAL;
    SLAM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SLAM_HIP_OK;
}

int slam_stage_reserve(slam_ctx* ctx, int slot, size_t bytes) {
    if (bytes <= ctx->stage_bytes[slot]) return SLAM_HIP_OK;
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_stage[slot]) {
        SLAM_HIP_CHECK(ctx, hipFree(ctx->d_stage[slot]));
        ctx->d_stage[slot] = nullptr;
        ctx->stage_bytes[slot] = 0;
    }
    size_t want = bytes + bytes / 2 + 4096;
    hipError_t e = hipMalloc(&ctx->d_stage[slot], want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return slam_set_error(ctx, SLAM_HIP_ENOMEM, "staging hipMalloc(%zu): %s", want, hipGetErrorString(e));
    }
    ctx->stage_bytes[slot] = want;
    return SLAM_HIP_OK;
}

#define SLAM_PIN_MAX (64u << 20)      // larger batches fall back to the runtime's 2-D copy

int slam_upload_image(slam_ctx* ctx, uint8_t* d_dst, size_t dpitch, const uint8_t* h_src, size_t spitch, size_t width, size_t rows,
                      size_t slot_off) {
    const size_t bytes = dpitch * rows;
    if (rows == 0 || width == 0) return SLAM_HIP_OK;
    if (slot_off + bytes > SLAM_PIN_MAX) {
        SLAM_HIP_CHECK(ctx, hipMemcpy2DAsync(d_dst, dpitch, h_src, spitch, width, rows, hipMemcpyHostToDevice, ctx->stream));
        return SLAM_HIP_OK;
    }
    if (slot_off + bytes > ctx->pin_bytes) {
        if (slot_off != 0) {          // never move the buffer while earlier images of this call are still leaving it
            SLAM_HIP_CHECK(ctx, hipMemcpy2DAsync(d_dst, dpitch, h_src, spitch, width, rows, hipMemcpyHostToDevice, ctx->stream));
            return SLAM_HIP_OK;
        }
        SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->h_pin) { (void)hipHostFree(ctx->h_pin); ctx->h_pin = nullptr; ctx->pin_bytes = 0; }
        size_t want = bytes * 4 < SLAM_PIN_MAX ? bytes * 4 : SLAM_PIN_MAX;
        if (want < (1u << 20)) want = 1u << 20;
        if (hipHostMalloc((void**)&ctx->h_pin, want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            ctx->h_pin = nullptr;
            SLAM_HIP_CHECK(ctx, hipMemcpy2DAsync(d_dst, dpitch, h_src, spitch, width, rows, hipMemcpyHostToDevice, ctx->stream));
            return SLAM_HIP_OK;
        }
        ctx->pin_bytes = want;
    }
    uint8_t* stage = ctx->h_pin + slot_off;
    if (spitch == dpitch) memcpy(stage, h_src, spitch * (rows - 1) + width);
    else for (size_t r = 0; r < rows; r++) memcpy(stage + r * dpitch, h_src + r * spitch, width);
    SLAM_HIP_CHECK(ctx, hipMemcpyAsync(d_dst, stage, dpitch * (rows - 1) + width, hipMemcpyHostToDevice, ctx->stream));
    return SLAM_HIP_OK;
}

// The deferred-job queue of the walk kernels holds a counter block and one word per job of the largest batch seen.
int slam_queue_reserve(slam_ctx* ctx, int njobs) {
    if ((size_t)njobs <= ctx->queue_jobs) return SLAM_HIP_OK;
    SLAM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_queue) { SLAM_HIP_CHECK(ctx, hipFree(ctx->d_queue)); ctx->d_queue = nullptr; ctx->queue_jobs = 0; }
    const size_t want = (size_t)njobs + (size_t)njobs / 2 + 65536;
    hipError_t e = hipMalloc((void**)&ctx->d_queue, (want + 16) * sizeof(int));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return slam_set_error(ctx, SLAM_HIP_ENOMEM, "walk queue hipMalloc(%zu words): %s", want + 16, hipGetErrorString(e));
    }
    ctx->queue_jobs = want;
    ctx->queue_last = 0;
    SLAM_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_queue, 0, (want + 16) * sizeof(int), ctx->stream));     // counter and list start at zero; every launch leaves them so
    return SLAM_HIP_OK;
}

// cv::resize(INTER_LINEAR), CV_8U: source offset and the two 11-bit fixed-point weights of every
// destination index.  Columns reset the fraction at the image edges (clamp_fraction), rows keep it
// and clamp the row index when it is used.  All arithmetic follows OpenCV 2.4's resize():
// scale = 1 / ((double)dsize / ssize); f = (float)((d + 0.5) * scale - 0.5); s = floor(f); f -= s;
// weights = saturate_cast<short>(cvRound(w * 2048)) with round-half-even.
void slam_resize_tables(int ssize, int dsize, std::vector<int32_t>& ofs, std::vector<uint32_t>& coef,
                        bool clamp_fraction) {
    ofs.resize(dsize);
    coef.resize(dsize);
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    for (int d = 0; d < dsize; d++) {
        volatile double t = (d + 0.5) * scale;   // two separately rounded operations
        float f = (float)(t - 0.5);
        int s = (int)std::floor(f);
        f -= (float)s;
        if (clamp_fraction) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
        long c0 = std::lrint((double)((1.f - f) * 2048.f));
        long c1 = std::lrint((double)(f * 2048.f));
        c0 = c0 < -32768 ? -32768 : (c0 > 32767 ? 32767 : c0);
        c1 = c1 < -32768 ? -32768 : (c1 > 32767 ? 32767 : c1);
        ofs[d] = s;
        coef[d] = (uint32_t)(uint16_t)(int16_t)c0 | ((uint32_t)(uint16_t)(int16_t)c1 << 16);
    }
}
