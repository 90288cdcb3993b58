/*
 * CPU oracle for the SLAM image hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * See slam_oracle.h for scope, the reference lines each function follows and the parity
 * pinning statement.  Plain C99, single thread, no dependencies.
 *
 * Build without FP contraction (-ffp-contract=off): every float/double operation below is meant
 * to be one IEEE-754 round-to-nearest operation, in the order written.
 */
#include "slam_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                              */
/* ------------------------------------------------------------------------------------------ */

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* OpenCV cvFloor / cvRound (round-half-even under the default rounding mode) / saturate_cast */
static inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
static inline int cv_round(double v) { return (int)lrint(v); }
static inline short sat_s16(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

/* common.hpp:62-66 and imagePyramid.hpp:27: C round(), half away from zero, on a double */
static inline int c_round_to_int(double v) { return (int)round(v); }

/* ------------------------------------------------------------------------------------------ */
/* cv::resize, INTER_LINEAR, CV_8UC1  (used by imagePyramid.hpp:27, feature.cpp:130-131)        */
/* ------------------------------------------------------------------------------------------ */

#define SO_COEF_BITS 11
#define SO_COEF_SCALE (1 << SO_COEF_BITS)

void so_pyramid_level_size(int w, int h, float sub, int* dw, int* dh) {
    /* imagePyramid.hpp:27: round(w/sub): int/float -> float, promoted to double for round() */
    float fw = (float)w / sub;
    float fh = (float)h / sub;
    *dw = c_round_to_int((double)fw);
    *dh = c_round_to_int((double)fh);
}

int so_resize_u8_linear(const uint8_t* src, int sw, int sh, ptrdiff_t spitch,
                        uint8_t* dst, int dw, int dh, ptrdiff_t dpitch) {
    if (sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0) return -1;
    /* resize(): inv_scale = dsize/ssize, scale = 1/inv_scale, both double */
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;

    int* xofs = (int*)malloc(sizeof(int) * (size_t)dw);
    short* ialpha = (short*)malloc(sizeof(short) * 2 * (size_t)dw);
    int* rows[2];
    rows[0] = (int*)malloc(sizeof(int) * (size_t)dw);
    rows[1] = (int*)malloc(sizeof(int) * (size_t)dw);
    if (!xofs || !ialpha || !rows[0] || !rows[1]) { free(xofs); free(ialpha); free(rows[0]); free(rows[1]); return -1; }

    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[2 * dx] = sat_s16(cv_round((1.f - fx) * SO_COEF_SCALE));
        ialpha[2 * dx + 1] = sat_s16(cv_round(fx * SO_COEF_SCALE));
    }

    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        const short b0 = sat_s16(cv_round((1.f - fy) * SO_COEF_SCALE));
        const short b1 = sat_s16(cv_round(fy * SO_COEF_SCALE));
        for (int k = 0; k < 2; k++) {
            int r = sy + k;
            r = r < 0 ? 0 : (r > sh - 1 ? sh - 1 : r);
            const uint8_t* S = src + (ptrdiff_t)r * spitch;
            int* D = rows[k];
            for (int dx = 0; dx < dw; dx++) {
                const int sx = xofs[dx];
                if (sx >= sw - 1)
                    D[dx] = S[sw - 1] * SO_COEF_SCALE;       /* HResizeLinear tail: S[sx]*ONE */
                else
                    D[dx] = S[sx] * ialpha[2 * dx] + S[sx + 1] * ialpha[2 * dx + 1];
            }
        }
        uint8_t* out = dst + (ptrdiff_t)dy * dpitch;
        for (int dx = 0; dx < dw; dx++)   /* VResizeLinear<uchar,int,short,FixedPtCast<..,22>> */
            out[dx] = (uint8_t)((((b0 * (rows[0][dx] >> 4)) >> 16) + ((b1 * (rows[1][dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(ialpha); free(rows[0]); free(rows[1]);
    return 0;
}

int so_pyramid_build(const uint8_t* im, int w, int h, ptrdiff_t pitch,
                     int nlev, const float* subs, uint8_t** levels) {
    for (int i = 0; i < nlev; i++) {
        if (i != 0 && !(subs[i - 1] < subs[i])) return -1;    /* imagePyramid.hpp:20-21 assert */
        const float sub = subs[i];
        if (sub == 1) {                                        /* imagePyramid.hpp:24-25 clone */
            for (int y = 0; y < h; y++) memcpy(levels[i] + (size_t)y * w, im + (ptrdiff_t)y * pitch, (size_t)w);
        } else {
            int dw, dh;
            so_pyramid_level_size(w, h, sub, &dw, &dh);
            if (so_resize_u8_linear(im, w, h, pitch, levels[i], dw, dh, dw) != 0) return -1;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* matchFeatureInArea (matching.cpp:9-228)                                                    */
/* ------------------------------------------------------------------------------------------ */

static inline unsigned sat_mul_u8(unsigned a, unsigned m) {   /* Mat::mul on CV_8U saturates */
    unsigned v = a * m;
    return v > 255u ? 255u : v;
}

/* Exact integer sums over a rows x cols rectangle:
 *   Im = I.mul(M) (or I when M == NULL);  CC = Im.dot(P0);  SI = |Im|^2;  SP = |P0|^2
 * P0 is the already-masked patch (matching.cpp:58), or the plain patch in the unmasked modes. */
static void rect_sums(const uint8_t* I, ptrdiff_t ip, const uint8_t* P0, ptrdiff_t pp,
                      const uint8_t* M, ptrdiff_t mp, int rows, int cols,
                      uint64_t* CC, uint64_t* SI, uint64_t* SP) {
    uint64_t cc = 0, si = 0, sp = 0;
    for (int r = 0; r < rows; r++) {
        const uint8_t* ir = I + (ptrdiff_t)r * ip;
        const uint8_t* pr = P0 + (ptrdiff_t)r * pp;
        uint32_t c = 0, s = 0, q = 0;
        if (M) {
            const uint8_t* mr = M + (ptrdiff_t)r * mp;
            for (int x = 0; x < cols; x++) {
                const unsigned v = sat_mul_u8(ir[x], mr[x]);
                c += v * pr[x]; s += v * v; q += (unsigned)pr[x] * pr[x];
            }
        } else {
            for (int x = 0; x < cols; x++) {
                const unsigned v = ir[x];
                c += v * pr[x]; s += v * v; q += (unsigned)pr[x] * pr[x];
            }
        }
        cc += c; si += s; sp += q;
    }
    *CC = cc; *SI = si; *SP = sp;
}

/* matching.cpp:70-72 / :97-100: float cc = dot; float i2 = norm; float p2 = norm; cc/(i2*p2).
 * cv::norm(NORM_L2) and Mat::dot on CV_8U accumulate exactly and return double. */
static inline float score_f32(uint64_t CC, uint64_t SI, uint64_t SP) {
    const float cc = (float)(double)CC;
    const float i2 = (float)sqrt((double)SI);
    const float p2 = (float)sqrt((double)SP);
    const float den = i2 * p2;
    return cc / den;
}

/* cv::matchTemplate(CV_TM_CCORR_NORMED), one channel (matching.cpp:152).  The numerator is the
 * exact integer correlation (OpenCV computes it with a float DFT; its rounding noise is not
 * reproducible and not restated) and the template norm is sqrt of the exact sum of squares. */
static inline float score_mt(uint64_t CC, uint64_t SI, uint64_t SP) {
    double num = (double)CC;
    const double tn = sqrt((double)SP);
    const double t = sqrt((double)SI) * tn;
    if (fabs(num) < t) num /= t;
    else if (fabs(num) < t * 1.125) num = num > 0 ? 1 : -1;
    else num = 0;
    return (float)num;
}

int so_match_feature_in_area(const uint8_t* im, int w, int h, ptrdiff_t ipitch,
                             const uint8_t* patch, int pw, int ph, ptrdiff_t ppitch,
                             const uint8_t* patch_mask, ptrdiff_t mpitch,
                             int area_x, int area_y, int area_w, int area_h,
                             const uint8_t* area_mask, ptrdiff_t ampitch,
                             int stride, int use_exact_area_mask,
                             float* score_out, so_match_out* out) {
    memset(out, 0, sizeof(*out));
    if (stride < 1 || pw < 1 || ph < 1 || w < 1 || h < 1) return -1;

    /* matching.cpp:17-30: crop the area to the image */
    int ax = imax(area_x, 0), ay = imax(area_y, 0);
    const int aw = imin(w - ax, area_w + area_x - ax);
    const int ah = imin(h - ay, area_h + area_y - ay);
    out->ax = ax; out->ay = ay; out->aw = aw; out->ah = ah;
    if (ax >= w || ay >= h || ax + aw <= 0 || ay + ah <= 0 || aw <= 0 || ah <= 0) {
        /* (aw<=0 / ah<=0 with the origin inside the image would hand OpenCV an empty Mat; it is
         * folded into the reference's own reject branch) */
        out->response = -1.f; out->x = 0; out->y = 0; out->aw = 0; out->ah = 0;
        return 0;
    }
    /* matching.cpp:31-35 */
    const uint8_t* amask0 = area_mask ? area_mask + (ptrdiff_t)(ay - area_y) * ampitch + (ax - area_x) : NULL;
    /* matching.cpp:36-39 */
    const int x0 = ax - pw / 2, y0 = ay - ph / 2;
    /* matching.cpp:44-49 */
    const int x0ua = imax(x0, 0), x1ua = imin(w, x0 + aw + pw - 1);
    const int y0ua = imax(y0, 0), y1ua = imin(h, y0 + ah + ph - 1);
    const int xl0 = x0ua - x0, xl1 = x1ua - x0 - pw + 1;
    const int yl0 = y0ua - y0, yl1 = y1ua - y0 - ph + 1;

    float* score = score_out ? score_out : (float*)malloc(sizeof(float) * (size_t)aw * ah);
    if (!score) return -1;
    for (size_t i = 0; i < (size_t)aw * ah; i++) score[i] = -1.f;   /* matching.cpp:53 */
#define SCORE(y, x) score[(size_t)(y) * aw + (x)]
#define AMASK(y, x) amask0[(ptrdiff_t)(y) * ampitch + (x)]

    uint8_t* patch0 = NULL;   /* patch.mul(patchMask), matching.cpp:58 */
    if (patch_mask) {
        patch0 = (uint8_t*)malloc((size_t)pw * ph);
        if (!patch0) { if (!score_out) free(score); return -1; }
        for (int r = 0; r < ph; r++)
            for (int c = 0; c < pw; c++)
                patch0[(size_t)r * pw + c] = (uint8_t)sat_mul_u8(patch[(ptrdiff_t)r * ppitch + c], patch_mask[(ptrdiff_t)r * mpitch + c]);
    }
    const uint8_t* P0 = patch_mask ? patch0 : patch;
    const ptrdiff_t p0pitch = patch_mask ? pw : ppitch;
    uint64_t CC, SI, SP, SPfull;
    {   /* norm(patch0) (matching.cpp:63,109) / template norm of matchTemplate */
        uint64_t d0, d1;
        rect_sums(P0, p0pitch, P0, p0pitch, NULL, 0, ph, pw, &d0, &d1, &SPfull);
    }

    if (patch_mask) {
        /* interior lattice, matching.cpp:62-74 / :108-119 */
        for (int x = xl0; x < xl1; x += stride)
            for (int y = yl0; y < yl1; y += stride)
                if (!area_mask || AMASK(y, x)) {
                    rect_sums(im + (ptrdiff_t)(y0 + y) * ipitch + (x0 + x), ipitch, P0, p0pitch,
                              patch_mask, mpitch, ph, pw, &CC, &SI, &SP);
                    SCORE(y, x) = score_f32(CC, SI, SPfull);
                }
    } else {
        /* matching.cpp:151-153: every interior cell, stride ignored */
        if (xl1 - xl0 > 0 && yl1 - yl0 > 0)
            for (int y = yl0; y < yl1; y++)
                for (int x = xl0; x < xl1; x++) {
                    rect_sums(im + (ptrdiff_t)(y0 + y) * ipitch + (x0 + x), ipitch, P0, p0pitch,
                              NULL, 0, ph, pw, &CC, &SI, &SP);
                    SCORE(y, x) = score_mt(CC, SI, SPfull);
                }
        /* matching.cpp:157-163 */
        if (area_mask && use_exact_area_mask)
            for (int x = xl0; x < xl1; ++x)
                for (int y = yl0; y < yl1; ++y)
                    if (!AMASK(y, x)) SCORE(y, x) = -1.f;
    }

    /* border lattice, matching.cpp:76-104 / :121-147 / :165-190 / :194-217 (identical shape) */
    for (int x = 0; x < aw; x += stride) {
        const int dl = imax(0, -x0 - x);
        const int dr = imax(0, x0 + x + pw - w);
        for (int y = 0; y < ah; y += stride) {
            if (yl0 <= y && y < yl1 && xl0 <= x && x < xl1) {
                if (yl1 >= ah) break;
                else y = yl1;
            }
            if (area_mask && !AMASK(y, x)) continue;
            const int dt = imax(0, -y0 - y);
            const int db = imax(0, y0 + y + ph - h);
            if (dt + db < ph && dr + dl < pw) {
                rect_sums(im + (ptrdiff_t)(y0 + y + dt) * ipitch + (x0 + x + dl), ipitch,
                          P0 + (ptrdiff_t)dt * p0pitch + dl, p0pitch,
                          patch_mask ? patch_mask + (ptrdiff_t)dt * mpitch + dl : NULL, mpitch,
                          ph - dt - db, pw - dl - dr, &CC, &SI, &SP);
                SCORE(y, x) = score_f32(CC, SI, SP);   /* p2 recomputed on the crop (:99) */
            }
        }
    }

    /* cv::minMaxLoc(score, NULL, &max, NULL, &loc), matching.cpp:221-227: strict '>' from
     * -FLT_MAX in row-major order; NaN never compares greater.  A map in which nothing compares
     * (all NaN) reports max = 0 at (-1,-1), as cv::minMaxIdx does when no index was found. */
    float best = -FLT_MAX;
    long besti = -1;
    for (long i = 0; i < (long)aw * ah; i++)
        if (score[i] > best) { best = score[i]; besti = i; }
    int lx, ly;
    if (besti < 0) { lx = -1; ly = -1; out->response = 0.f; }
    else { lx = (int)(besti % aw); ly = (int)(besti / aw); out->response = best; }
    out->x = lx + ax;
    out->y = ly + ay;

    /* sub-pixel refinement: NOT in the reference; parabola through the two neighbours, f32. */
    out->dx = 0.f; out->dy = 0.f;
    if (besti >= 0) {
        const float s = best;
        if (lx - 1 >= 0 && lx + 1 < aw) {
            const float sl = SCORE(ly, lx - 1), sr = SCORE(ly, lx + 1);
            if (sl != -1.f && sr != -1.f && sl == sl && sr == sr) {
                const float d2 = (sl + sr) - 2.0f * s;
                if (d2 < 0.f) {
                    float o = (sl - sr) / (2.0f * d2);
                    out->dx = o < -0.5f ? -0.5f : (o > 0.5f ? 0.5f : o);
                }
            }
        }
        if (ly - 1 >= 0 && ly + 1 < ah) {
            const float su = SCORE(ly - 1, lx), sd = SCORE(ly + 1, lx);
            if (su != -1.f && sd != -1.f && su == su && sd == sd) {
                const float d2 = (su + sd) - 2.0f * s;
                if (d2 < 0.f) {
                    float o = (su - sd) / (2.0f * d2);
                    out->dy = o < -0.5f ? -0.5f : (o > 0.5f ? 0.5f : o);
                }
            }
        }
    }
#undef SCORE
#undef AMASK
    free(patch0);
    if (!score_out) free(score);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Feature::track (feature.cpp:117-172), after project()                                      */
/* ------------------------------------------------------------------------------------------ */

int so_track(uint8_t* const* levels, const int* lw, const int* lh, const float* subs, int nlev,
             const uint8_t* proj, const uint8_t* proj_mask, int projw, int projh,
             float px, float py, float threshold, int stride, so_track_out* out) {
    return so_track_ex(levels, lw, lh, subs, nlev, proj, proj_mask, projw, projh, px, py, threshold, stride, NULL, NULL, out);
}

/* The same walk; with coarse_area (x, y, w, h in the coarsest level's pixels) that level is searched
 * in the given window under coarse_mask (pitch w, or NULL) instead of the square of feature.cpp:134,
 * 144-146 - an extension of the build (elliptical uncertainty windows), not reference behaviour. */
int so_track_ex(uint8_t* const* levels, const int* lw, const int* lh, const float* subs, int nlev,
                const uint8_t* proj, const uint8_t* proj_mask, int projw, int projh,
                float px, float py, float threshold, int stride,
                const int* coarse_area, const uint8_t* coarse_mask, so_track_out* out) {
    memset(out, 0, sizeof(*out));
    if (nlev < 1 || stride < 1) return -1;
    if (projw <= 0 || projh <= 0) {                 /* feature.cpp:112-115 */
        out->response = -1.f; out->x = -1; out->y = -1;
        return 0;
    }
    const float fullResSearchRad = 20;              /* feature.cpp:119 */
    float tx = px, ty = py;                         /* trackedPoint, feature.cpp:122 */
    float response = 0.f, searchRad;
    uint8_t* tbuf = (uint8_t*)malloc((size_t)projw * projh);
    uint8_t* mbuf = (uint8_t*)malloc((size_t)projw * projh);
    if (!tbuf || !mbuf) { free(tbuf); free(mbuf); return -1; }
    for (int iSub = nlev - 1; iSub >= 0; --iSub) {
        const float sub = subs[iSub];
        const uint8_t *totrack, *totrack_mask;
        int tw, th;
        if (sub == 1) {                             /* feature.cpp:126-128 */
            totrack = proj; totrack_mask = proj_mask; tw = projw; th = projh;
        } else {                                    /* feature.cpp:130-131: Size(projw/sub, projh/sub) truncates */
            tw = (int)((float)projw / sub);
            th = (int)((float)projh / sub);
            if (tw <= 0 || th <= 0) {
                /* cv::resize would throw on an empty size; reported like an empty projection */
                out->response = -1.f; out->x = -1; out->y = -1; out->dx = out->dy = 0.f;
                free(tbuf); free(mbuf);
                return 0;
            }
            so_resize_u8_linear(proj, projw, projh, projw, tbuf, tw, th, tw);
            so_resize_u8_linear(proj_mask, projw, projh, projw, mbuf, tw, th, tw);
            totrack = tbuf; totrack_mask = mbuf;
        }
        if (iSub == nlev - 1) {
            searchRad = fullResSearchRad / sub;     /* feature.cpp:134 */
        } else {
            searchRad = (float)(1.3 * subs[iSub + 1] / sub);   /* feature.cpp:136, double expr */
            if (iSub == 0) searchRad *= stride;     /* feature.cpp:137-138 */
        }
        /* feature.cpp:144-146: round() on float expressions */
        int rx = (int)round((double)(tx / sub - searchRad));
        int ry = (int)round((double)(ty / sub - searchRad));
        int rw = (int)round((double)(2 * searchRad + 1)), rh = rw;
        const uint8_t* am = NULL;
        if (coarse_area && iSub == nlev - 1) {
            rx = coarse_area[0]; ry = coarse_area[1]; rw = coarse_area[2]; rh = coarse_area[3];
            am = coarse_mask;
        }
        so_match_out m;
        so_match_feature_in_area(levels[iSub], lw[iSub], lh[iSub], lw[iSub], totrack, tw, th, tw,
                                 totrack_mask, tw, rx, ry, rw, rh, am, rw,
                                 (iSub == 0) ? 1 : stride, 0, NULL, &m);   /* feature.cpp:147-149 */
        response = m.response;
        tx = (float)m.x; ty = (float)m.y;
        tx *= sub; ty *= sub;                       /* feature.cpp:150 */
        out->levels_run++;
        out->dx = m.dx; out->dy = m.dy;
        if ((double)response < 0.67 * (double)threshold) break;   /* feature.cpp:151-152 */
    }
    free(tbuf); free(mbuf);
    out->x = (int)tx; out->y = (int)ty;            /* Point2i(float,float) truncates */
    out->response = response;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* LineFeature::track (lineFeature.cpp:86-165): the disc mask and the level walk              */
/* ------------------------------------------------------------------------------------------ */

/* cv::circle(img, center, radius, Scalar(255), -1) on CV_8UC1: thickness < 0 and line type 8 select
 * OpenCV 2.4's Circle(img, center, radius, color, fill = 1) (drawing.cpp), the midpoint walk
 * (dx, dy) from (radius, 0) with err = dx^2 + dy^2 - radius^2 that fills the rows center.y +- dy over
 * center.x +- dx and the rows center.y +- dx over center.x +- dy, clipped to the image. */
static void hline255(uint8_t* row, int xl, int xr, int w) {
    if (xl < 0) xl = 0;
    if (xr > w - 1) xr = w - 1;
    for (int x = xl; x <= xr; x++) row[x] = 255;
}

static void cv_circle_filled(uint8_t* img, int w, int h, ptrdiff_t pitch, int cx, int cy, int radius) {
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        const int y11 = cy - dy, y12 = cy + dy, y21 = cy - dx, y22 = cy + dx;
        const int x11 = cx - dx, x12 = cx + dx, x21 = cx - dy, x22 = cx + dy;
        if (x11 < w && x12 >= 0 && y21 < h && y22 >= 0) {
            if ((unsigned)y11 < (unsigned)h) hline255(img + (ptrdiff_t)y11 * pitch, x11, x12, w);
            if ((unsigned)y12 < (unsigned)h) hline255(img + (ptrdiff_t)y12 * pitch, x11, x12, w);
            if (x21 < w && x22 >= 0) {
                if ((unsigned)y21 < (unsigned)h) hline255(img + (ptrdiff_t)y21 * pitch, x21, x22, w);
                if ((unsigned)y22 < (unsigned)h) hline255(img + (ptrdiff_t)y22 * pitch, x21, x22, w);
            }
        }
        dy++;
        err += plus;
        plus += 2;
        const int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
}

int so_disc_mask(int w, int h, const int* circles, int ncircles, uint8_t* mask) {
    if (w <= 0 || h <= 0 || ncircles < 0) return -1;
    memset(mask, 0, (size_t)w * h);                           /* lineFeature.cpp:87 */
    for (int i = 0; i < ncircles; i++) {                      /* lineFeature.cpp:88-96 */
        if (circles[3 * i + 2] < 0) return -1;                /* cv::circle asserts radius >= 0 */
        cv_circle_filled(mask, w, h, w, circles[3 * i], circles[3 * i + 1], circles[3 * i + 2]);
    }
    return 0;
}

int so_line_track(uint8_t* const* levels, const int* lw, const int* lh, const float* subs, int nlev,
                  const uint8_t* descr, int dw, int dh, int ax, int ay, int aw, int ah,
                  const uint8_t* area_mask, float threshold, int stride, so_track_out* out) {
    memset(out, 0, sizeof(*out));
    if (nlev < 1 || stride < 1 || dw <= 0 || dh <= 0 || aw <= 0 || ah <= 0) return -1;
    uint8_t* tbuf = (uint8_t*)malloc((size_t)dw * dh * 4 + 16);
    if (!tbuf) return -1;
    const float areaRes = 1.f / subs[nlev - 1];               /* lineFeature.cpp:30 */
    const float desch = (float)dh, descw = (float)dw;         /* lineFeature.cpp:122 */
    /* lineFeature.cpp:124: Size(round(descw*areaRes), round(desch*areaRes)) */
    int tw = (int)round((double)(descw * areaRes)), th = (int)round((double)(desch * areaRes));
    int tpx = -1, tpy = -1;
    float response = -1.f;
    if (tw <= 0 || th <= 0 || (size_t)tw * th > (size_t)dw * dh * 4) goto fail;   /* cv::resize would throw: reported as a failure */
    so_resize_u8_linear(descr, dw, dh, dw, tbuf, tw, th, tw);
    {
        so_match_out m;                                       /* lineFeature.cpp:125-127 */
        so_match_feature_in_area(levels[nlev - 1], lw[nlev - 1], lh[nlev - 1], lw[nlev - 1], tbuf, tw, th, tw, NULL, 0,
                                 ax, ay, aw, ah, area_mask, aw, stride, 0, NULL, &m);
        response = m.response;
        out->levels_run = 1; out->dx = m.dx; out->dy = m.dy;
        const double inv = 1. / (double)areaRes;              /* lineFeature.cpp:130: Point2i *= double rounds (cvRound) */
        tpx = cv_round((double)m.x * inv);
        tpy = cv_round((double)m.y * inv);
    }
    if ((double)response > (double)threshold * 0.67) {        /* lineFeature.cpp:144 */
        for (int iSub = nlev - 2; iSub >= 0; --iSub) {
            const float lastsub = subs[iSub + 1], sub = subs[iSub];
            const int laststride = stride, newstride = (iSub == 0) ? 1 : stride;
            const int searchRad = (int)(sub / lastsub * (float)laststride / (float)newstride);   /* lineFeature.cpp:150 */
            /* lineFeature.cpp:151-152: Rect(int/float - int, ...) converts the float expressions to int by truncation */
            const int rx = (int)((float)tpx / sub - (float)searchRad), ry = (int)((float)tpy / sub - (float)searchRad);
            const int rs = 2 * searchRad + 1;
            tw = (int)(descw / sub); th = (int)(desch / sub); /* lineFeature.cpp:153 */
            if (tw <= 0 || th <= 0) goto fail;
            so_resize_u8_linear(descr, dw, dh, dw, tbuf, tw, th, tw);
            so_match_out m;                                   /* lineFeature.cpp:154-156 */
            so_match_feature_in_area(levels[iSub], lw[iSub], lh[iSub], lw[iSub], tbuf, tw, th, tw, NULL, 0,
                                     rx, ry, rs, rs, NULL, 0, newstride, 0, NULL, &m);
            response = m.response;
            out->levels_run++; out->dx = m.dx; out->dy = m.dy;
            tpx = cv_round((double)((float)m.x * sub));       /* lineFeature.cpp:157: Point2i *= float rounds (cvRound) */
            tpy = cv_round((double)((float)m.y * sub));
            if ((double)response < (double)threshold * 0.67) break;   /* lineFeature.cpp:158-159 */
        }
    }
    free(tbuf);
    out->x = tpx; out->y = tpy; out->response = response;
    return 0;
fail:
    free(tbuf);
    out->x = -1; out->y = -1; out->response = -1.f; out->dx = out->dy = 0.f;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* cv::FAST, 16-pixel ring, 9 contiguous (new_point.cpp:120)                                  */
/* ------------------------------------------------------------------------------------------ */

static const int so_ring[16][2] = {   /* (dx, dy) */
    {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

/* A_dark = max over the 16 arcs of 9 of min(v - ring), A_bright the same on (ring - v). */
static void fast_arcs(const uint8_t* p, ptrdiff_t pitch, int* a_dark, int* a_bright) {
    int d[16];
    const int v = p[0];
    for (int k = 0; k < 16; k++) d[k] = v - p[(ptrdiff_t)so_ring[k][1] * pitch + so_ring[k][0]];
    int ad = -256, ab = -256;
    for (int s = 0; s < 16; s++) {
        int mn = 256, mx = -256;
        for (int j = 0; j < 9; j++) {
            const int q = d[(s + j) & 15];
            if (q < mn) mn = q;
            if (q > mx) mx = q;
        }
        if (mn > ad) ad = mn;
        if (-mx > ab) ab = -mx;
    }
    *a_dark = ad; *a_bright = ab;
}

void so_fast_score_map(const uint8_t* im, int w, int h, ptrdiff_t pitch, int16_t* score) {
    memset(score, 0, sizeof(int16_t) * (size_t)w * h);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            int ad, ab;
            fast_arcs(im + (ptrdiff_t)y * pitch + x, pitch, &ad, &ab);
            const int s = imax(ad, ab) - 1;
            score[(size_t)y * w + x] = (int16_t)(s < 0 ? 0 : s);
        }
}

int so_fast(const uint8_t* im, int w, int h, ptrdiff_t pitch, int threshold, int nonmax,
            so_keypoint* out, int cap) {
    threshold = imin(imax(threshold, 0), 255);
    if (w < 7 || h < 7) return 0;
    /* per-pixel stored score as cv::FAST keeps it: cornerScore for corners, 0 elsewhere (uchar) */
    uint8_t* sc = (uint8_t*)calloc((size_t)w * h, 1);
    uint8_t* is = (uint8_t*)calloc((size_t)w * h, 1);
    if (!sc || !is) { free(sc); free(is); return -1; }
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            int ad, ab;
            fast_arcs(im + (ptrdiff_t)y * pitch + x, pitch, &ad, &ab);
            if (ad > threshold || ab > threshold) {          /* 9 contiguous |diff| > threshold */
                is[(size_t)y * w + x] = 1;
                /* cornerScore<16>: a0 starts at threshold, result = max(t, arcs) - 1 */
                sc[(size_t)y * w + x] = (uint8_t)(imax(threshold, imax(ad, ab)) - 1);
            }
        }
    int n = 0;
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const size_t i = (size_t)y * w + x;
            if (!is[i]) continue;
            const int s = sc[i];
            if (nonmax) {
                if (!(s > sc[i - 1] && s > sc[i + 1] &&
                      s > sc[i - w - 1] && s > sc[i - w] && s > sc[i - w + 1] &&
                      s > sc[i + w - 1] && s > sc[i + w] && s > sc[i + w + 1]))
                    continue;
            }
            if (n < cap) { out[n].x = (float)x; out[n].y = (float)y; out[n].response = (float)s; }
            n++;
        }
    free(sc); free(is);
    return n;
}

static int kp_cmp(const void* a, const void* b) {
    const so_keypoint* p = (const so_keypoint*)a;
    const so_keypoint* q = (const so_keypoint*)b;
    if (p->response != q->response) return p->response > q->response ? -1 : 1;  /* new_point.cpp:109-111 */
    if (p->y != q->y) return p->y < q->y ? -1 : 1;      /* std::sort is unstable: ties fixed as y, x ascending */
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    return 0;
}

int so_get_sorted_keypoints(const uint8_t* im, int w, int h, ptrdiff_t pitch, size_t n_min_pts,
                            float* fast_threshold, so_keypoint* out, int cap) {
    const size_t maxk = (size_t)w * h;
    so_keypoint* tmp = (so_keypoint*)malloc(sizeof(so_keypoint) * (maxk ? maxk : 1));
    if (!tmp) return -1;
    float f = *fast_threshold;
    size_t n = 0;
    while (n < n_min_pts) {                             /* new_point.cpp:116-122 */
        const int t = (int)f;                           /* FAST(..., int threshold, ...) */
        const int r = so_fast(im, w, h, pitch, t, 1, tmp, (int)maxk);
        n = r < 0 ? 0 : (size_t)r;
        f *= 0.5f;
        if (t <= 0) break;   /* the reference loops forever on a corner-poor image; stop after t = 0 */
    }
    f *= 2.f;                                           /* new_point.cpp:123 */
    qsort(tmp, n, sizeof(so_keypoint), kp_cmp);         /* new_point.cpp:124 */
    if (n > 2 * n_min_pts)                              /* new_point.cpp:125-126 */
        f = tmp[(size_t)((double)n_min_pts * 1.9)].response;
    *fast_threshold = f;
    for (size_t i = 0; i < n && i < (size_t)cap; i++) out[i] = tmp[i];
    free(tmp);
    return (int)n;
}

/* ------------------------------------------------------------------------------------------ */
/* new_point.cpp:129-156 (selection loop of addNewLines), feature.cpp:36-51 (descriptor crop)  */
/* ------------------------------------------------------------------------------------------ */

/* cv::norm(Point2f a - Point2f b): the difference is formed in float, the norm in double. */
static double pt_dist(float ax, float ay, float bx, float by) {
    const float dx = ax - bx, dy = ay - by;
    return sqrt((double)dx * dx + (double)dy * dy);
}

void so_sort_keypoints(so_keypoint* kps, int n) { qsort(kps, (size_t)n, sizeof(so_keypoint), kp_cmp); }

int so_select_new_points(const so_keypoint* kps, int nkps, const float* occ_xy, int nocc, int n,
                         float min_dist, int rx, int ry, int w, int h, int* sel) {
    int nsel = 0;
    for (int i = 0; i < nkps && n > 0; i++) {                 /* new_point.cpp:137 */
        const float px = kps[i].x, py = kps[i].y;
        int bad = 0;
        for (int j = 0; j < nocc && !bad; j++)                /* new_point.cpp:139-144: matches, then lineMatches */
            if (pt_dist(occ_xy[2 * j], occ_xy[2 * j + 1], px, py) < min_dist) bad = 1;
        for (int j = 0; j < nsel && !bad; j++)                /* new_point.cpp:145-147 */
            if (pt_dist(kps[sel[j]].x, kps[sel[j]].y, px, py) < min_dist) bad = 1;
        if (!bad && ((py - ry < 0) || (px - rx < 0) || (py + ry >= h) || (px + rx >= w))) bad = 1;   /* :148-150 */
        if (bad) continue;
        sel[nsel++] = i;                                      /* new_point.cpp:151-153 */
        --n;
    }
    return nsel;
}

static int iround_f(float a) { return (int)round((double)a); }   /* common.hpp:62-66 */

int so_extract_descriptor(const uint8_t* im, int w, int h, ptrdiff_t pitch, float px, float py, int rx, int ry,
                          uint8_t* out, int* ox, int* oy, int* ow, int* oh) {
    /* feature.cpp:42-45 (and lineFeature.cpp:7-10): im(Range(max(0, iround(y-ry)), min(h, iround(y+ry+1))), Range(...x...)) */
    const int y0 = imax(0, iround_f(py - (float)ry)), y1 = imin(h, iround_f(py + (float)ry + 1.f));
    const int x0 = imax(0, iround_f(px - (float)rx)), x1 = imin(w, iround_f(px + (float)rx + 1.f));
    *ox = x0; *oy = y0; *ow = x1 > x0 ? x1 - x0 : 0; *oh = y1 > y0 ? y1 - y0 : 0;
    if (*ow == 0 || *oh == 0) { *ox = 0; *oy = 0; *ow = 0; *oh = 0; return 0; }
    if (out)
        for (int y = y0; y < y1; y++) memcpy(out + (size_t)(y - y0) * (size_t)(x1 - x0), im + (ptrdiff_t)y * pitch + x0, (size_t)(x1 - x0));
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* feature.cpp:53-100  SLAM::Feature::project                                                  */
/* ------------------------------------------------------------------------------------------ */

/* The float matrix steps of feature.cpp:61-94 with OpenCV 2.4's arithmetic:
 *   P*B (3x4 by 4x3)        generic cv::gemm for CV_32F: products accumulated in double over k = 0..3,
 *                           the sum rounded to float;
 *   A*p, Am*p (3x3 by 3x1)  gemm's small-matrix case: a0*b0 + a1*b1 + a2*b2 evaluated in float;
 *   row /= s                convertTo(row, -1, 1./s): each element times (float)(1./(double)s), in float;
 *   A.inv()                 cv::invert's closed form for 3x3 CV_32F: cofactors and determinant in
 *                           double, each entry rounded to float; a zero determinant gives the zero matrix. */
int so_project_setup(const float* P, const float* B, const float* p3d, int dw, int dh, so_projection* out) {
    memset(out, 0, sizeof(*out));
    if (dw <= 0 || dh <= 0) return 0;                                 /* feature.cpp:56-59 */
    float Bm[12];
    memcpy(Bm, B, sizeof(Bm));
    Bm[0 * 3 + 2] = p3d[0]; Bm[1 * 3 + 2] = p3d[1]; Bm[2 * 3 + 2] = p3d[2];   /* feature.cpp:61 */
    float A[9];
    for (int i = 0; i < 3; i++)                                       /* feature.cpp:62 */
        for (int j = 0; j < 3; j++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += (double)P[i * 4 + k] * (double)Bm[k * 3 + j];
            A[i * 3 + j] = (float)s;
        }
    const int dx = dw / 2, dy = dh / 2;                               /* feature.cpp:63 */
    const float gc[4][2] = {{(float)-dx, (float)-dy}, {(float)-dx, (float)dy}, {(float)dx, (float)dy}, {(float)dx, (float)-dy}};
    int xmin = 1000000, xmax = -1000000, ymin = 1000000, ymax = -1000000;
    for (int i = 0; i < 4; i++) {                                     /* feature.cpp:67-79 */
        float c[3];
        for (int r = 0; r < 3; r++) c[r] = A[r * 3 + 0] * gc[i][0] + A[r * 3 + 1] * gc[i][1] + A[r * 3 + 2] * 1.0f;
        const float inv = (float)(1.0 / (double)c[2]);
        const float cx = c[0] * inv, cy = c[1] * inv;
        if (!(fabsf(cx) < 1e9f) || !(fabsf(cy) < 1e9f)) return 0;     /* not representable as int: treated as empty */
        xmin = imin(xmin, (int)ceil((double)cx)); xmax = imax(xmax, (int)floor((double)cx));
        ymin = imin(ymin, (int)ceil((double)cy)); ymax = imax(ymax, (int)floor((double)cy));
    }
    out->x = xmin; out->y = ymin;                                     /* feature.cpp:80-83 */
    out->w = imax(0, xmax - xmin); out->h = imax(0, ymax - ymin);
    if (out->w == 0 || out->h == 0 || out->w > 500 || out->h > 500) return 0;   /* feature.cpp:84-86 */
    /* feature.cpp:90  Am = A.inv() */
    const double a00 = A[0], a01 = A[1], a02 = A[2], a10 = A[3], a11 = A[4], a12 = A[5], a20 = A[6], a21 = A[7], a22 = A[8];
    double d = a00 * (a11 * a22 - a12 * a21) - a01 * (a10 * a22 - a12 * a20) + a02 * (a10 * a21 - a11 * a20);
    float Am[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (d != 0.0) {
        d = 1.0 / d;
        Am[0] = (float)((a11 * a22 - a12 * a21) * d); Am[1] = (float)((a02 * a21 - a01 * a22) * d); Am[2] = (float)((a01 * a12 - a02 * a11) * d);
        Am[3] = (float)((a12 * a20 - a10 * a22) * d); Am[4] = (float)((a00 * a22 - a02 * a20) * d); Am[5] = (float)((a02 * a10 - a00 * a12) * d);
        Am[6] = (float)((a10 * a21 - a11 * a20) * d); Am[7] = (float)((a01 * a20 - a00 * a21) * d); Am[8] = (float)((a00 * a11 - a01 * a10) * d);
    }
    /* feature.cpp:91-92  Am.col(2) = Am * (xmin, ymin, 1) */
    const float fx = (float)xmin, fy = (float)ymin;
    float c2[3];
    for (int r = 0; r < 3; r++) c2[r] = Am[r * 3 + 0] * fx + Am[r * 3 + 1] * fy + Am[r * 3 + 2] * 1.0f;
    for (int r = 0; r < 3; r++) Am[r * 3 + 2] = c2[r];
    /* feature.cpp:93-94  Am.row(0) += dx * Am.row(2); Am.row(1) += dy * Am.row(2) */
    for (int j = 0; j < 3; j++) Am[0 * 3 + j] = Am[0 * 3 + j] + Am[2 * 3 + j] * (float)dx;
    for (int j = 0; j < 3; j++) Am[1 * 3 + j] = Am[1 * 3 + j] + Am[2 * 3 + j] * (float)dy;
    for (int j = 0; j < 9; j++) if (!(fabsf(Am[j]) < 1e30f)) { out->w = 0; out->h = 0; return 0; }   /* non-finite map: empty */
    memcpy(out->m, Am, sizeof(Am));
    out->valid = 1;
    return 0;
}

/* cv::warpPerspective(src, dst, M, Size(dw, dh), WARP_INVERSE_MAP | interpolation) for CV_8UC1 with
 * the default BORDER_CONSTANT, value 0 (feature.cpp:96-98), OpenCV 2.4 imgwarp.cpp:
 *   - M is converted to double; the destination is walked in blocks of bw0 columns
 *     (bh0 = min(16, h); bw0 = min(1024/bh0, w)); per row of a block X0 = M0*x + M1*y + M2 (x = block
 *     start) and per pixel (X0 + M0*x1) * W with W = 1/(W0 + M6*x1) (nearest) or 32/(...) (linear);
 *     a zero denominator gives W = 0; the products are clamped to the int range and rounded
 *     half-to-even (cvRound);
 *   - nearest: remap takes src(sat_s16(Y), sat_s16(X)) or 0 outside;
 *   - linear: integer part sat_s16(X >> 5), 5 fractional bits per axis, the four taps weighted by
 *     (32-fx)(32-fy)*32 etc. (the 15-bit table; its (0,0) entry is stored as 32767,0,0,1, which yields
 *     the same byte as 32768,0,0,0 for every tap combination), taps outside the image read 0,
 *     result (sum + 16384) >> 15. */
int so_warp_perspective_u8(const uint8_t* src, int sw, int sh, ptrdiff_t spitch, const float* Mf, int nearest,
                           uint8_t* dst, int dw, int dh, ptrdiff_t dpitch) {
    if (sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0) return -1;
    double M[9];
    for (int i = 0; i < 9; i++) M[i] = (double)Mf[i];
    int bh0 = imin(16, dh);
    const int bw0 = imin(1024 / bh0, dw);
    for (int y = 0; y < dh; y++) {
        for (int x = 0; x < dw; x += bw0) {
            const int bw = imin(bw0, dw - x);
            const double X0 = M[0] * x + M[1] * y + M[2];
            const double Y0 = M[3] * x + M[4] * y + M[5];
            const double W0 = M[6] * x + M[7] * y + M[8];
            for (int x1 = 0; x1 < bw; x1++) {
                double W = W0 + M[6] * x1;
                if (nearest) W = W != 0.0 ? 1. / W : 0.0; else W = W != 0.0 ? 32. / W : 0.0;
                const double fX = fmax((double)INT32_MIN, fmin((double)INT32_MAX, (X0 + M[0] * x1) * W));
                const double fY = fmax((double)INT32_MIN, fmin((double)INT32_MAX, (Y0 + M[3] * x1) * W));
                const int X = cv_round(fX), Y = cv_round(fY);
                uint8_t* D = dst + (ptrdiff_t)y * dpitch + x + x1;
                if (nearest) {
                    const int sx = sat_s16(X), sy = sat_s16(Y);
                    *D = ((unsigned)sx < (unsigned)sw && (unsigned)sy < (unsigned)sh) ? src[(ptrdiff_t)sy * spitch + sx] : 0;
                } else {
                    const int sx = sat_s16(X >> 5), sy = sat_s16(Y >> 5);
                    const int ax = X & 31, ay = Y & 31;
                    const int w0 = (32 - ax) * (32 - ay) * 32, w1 = ax * (32 - ay) * 32, w2 = (32 - ax) * ay * 32, w3 = ax * ay * 32;
                    const int in_x0 = (unsigned)sx < (unsigned)sw, in_x1 = (unsigned)(sx + 1) < (unsigned)sw;
                    const int in_y0 = (unsigned)sy < (unsigned)sh, in_y1 = (unsigned)(sy + 1) < (unsigned)sh;
                    const int v0 = (in_x0 && in_y0) ? src[(ptrdiff_t)sy * spitch + sx] : 0;
                    const int v1 = (in_x1 && in_y0) ? src[(ptrdiff_t)sy * spitch + sx + 1] : 0;
                    const int v2 = (in_x0 && in_y1) ? src[(ptrdiff_t)(sy + 1) * spitch + sx] : 0;
                    const int v3 = (in_x1 && in_y1) ? src[(ptrdiff_t)(sy + 1) * spitch + sx + 1] : 0;
                    *D = (uint8_t)((v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3 + (1 << 14)) >> 15);
                }
            }
        }
    }
    return 0;
}

int so_project(const uint8_t* descr, int dw, int dh, ptrdiff_t dpitch, const float* P, const float* B, const float* p3d,
               uint8_t* proj, uint8_t* mask, size_t cap, so_projection* out) {
    so_project_setup(P, B, p3d, dw, dh, out);
    if (!out->valid) return 0;
    if ((size_t)out->w * (size_t)out->h > cap) return -1;
    so_warp_perspective_u8(descr, dw, dh, dpitch, out->m, 0, proj, out->w, out->h, out->w);     /* feature.cpp:96 */
    uint8_t* ones = (uint8_t*)malloc((size_t)dw * dh);
    if (!ones) return -1;
    memset(ones, 1, (size_t)dw * dh);
    so_warp_perspective_u8(ones, dw, dh, dw, out->m, 1, mask, out->w, out->h, out->w);            /* feature.cpp:97-98 */
    free(ones);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* tests/test_matching.cpp:13-101 restated                                                    */
/* ------------------------------------------------------------------------------------------ */
/* ---- uncertainty windows (build-defined; no live counterpart in the reference) -------------------
 * Gate ellipse (p - p2d)^T S^-1 (p - p2d) <= gate of a 2x2 covariance S, scaled by res to the level
 * searched.  The rectangle uses the floor / ceil(+1) convention of lineFeature.cpp:60-63 and the
 * 300-pixel cap of lineFeature.cpp:27,81-84. */
int so_ellipse_setup(float px, float py, const float* cov, float gate, float res, so_ellipse* out) {
    if (!cov || !out) return -1;
    memset(out, 0, sizeof(*out));
    double in[7] = {px, py, cov[0], cov[1], cov[2], gate, res};
    for (int i = 0; i < 7; i++)
        if (isnan(in[i]) || isinf(in[i])) return 0;
    double a = cov[0], b = cov[1], d = cov[2];
    double det = a * d - b * b;
    if (gate <= 0.f || res <= 0.f || a <= 0.0 || d <= 0.0 || det <= 0.0) return 0;
    double r = res, cx = (double)px * r, cy = (double)py * r;
    double hx = sqrt((double)gate * a) * r, hy = sqrt((double)gate * d) * r;
    if (!(fabs(cx) + hx < 1e9 && fabs(cy) + hy < 1e9)) return 0;
    int xmin = (int)floor(cx - hx), ymin = (int)floor(cy - hy);
    int xmax = (int)ceil(cx + hx) + 1, ymax = (int)ceil(cy + hy) + 1;
    if (xmax - xmin > 300 || ymax - ymin > 300) return 0;
    double s = (double)gate * det * r * r;
    out->x = xmin; out->y = ymin; out->w = xmax - xmin; out->h = ymax - ymin;
    out->ux = (float)(cx - xmin); out->uy = (float)(cy - ymin);
    out->q11 = (float)(d / s); out->q12 = (float)(-b / s); out->q22 = (float)(a / s);
    out->valid = 1;
    return 0;
}

int so_ellipse_mask(const so_ellipse* e, uint8_t* mask) {
    if (!e || !mask || !e->valid || e->w < 1 || e->h < 1) return -1;
    const float q12x2 = e->q12 + e->q12;
    const int cxn = (int)floorf(e->ux + 0.5f), cyn = (int)floorf(e->uy + 0.5f);   /* nearest pixel: always inside */
    for (int y = 0; y < e->h; y++)
        for (int x = 0; x < e->w; x++) {
            const float dx = (float)x - e->ux, dy = (float)y - e->uy;
            const float t0 = (e->q11 * dx) * dx;
            const float t1 = (q12x2 * dx) * dy;
            const float t2 = (e->q22 * dy) * dy;
            const float t = (t0 + t1) + t2;
            mask[(size_t)y * e->w + x] = (t <= 1.f || (x == cxn && y == cyn)) ? 255 : 0;
        }
    return 0;
}


/* glibc's rand() (the TYPE_3 additive feedback generator r[i] = r[i-3] + r[i-31], seeded through
 * the 16807 Lehmer generator, first 310 outputs discarded), kept in private state: the trial stream
 * of tests/test_matching.cpp is the unseeded (seed 1) glibc stream, and a private copy cannot be
 * disturbed by other users of rand() in the process (runtime libraries, other threads). */
static uint32_t pt_r[34];
static int pt_f = 3, pt_b = 0, pt_seeded = 0;

void so_proptest_srand(unsigned seed) {
    int32_t word = seed ? (int32_t)seed : 1;
    pt_r[0] = (uint32_t)word;
    for (int i = 1; i < 31; i++) {
        const long hi = word / 127773, lo = word % 127773;
        long w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        word = (int32_t)w;
        pt_r[i] = (uint32_t)word;
    }
    pt_f = 3; pt_b = 0; pt_seeded = 1;
    for (int i = 0; i < 310; i++) (void)so_proptest_rand();
}

int so_proptest_rand(void) {
    if (!pt_seeded) so_proptest_srand(1);
    pt_r[pt_f] += pt_r[pt_b];
    const uint32_t out = pt_r[pt_f] >> 1;
    if (++pt_f >= 31) pt_f = 0;
    if (++pt_b >= 31) pt_b = 0;
    return (int)out;
}

void so_proptest_next(int use_area_mask, int use_patch_mask, so_proptest_case* c) {
    memset(c, 0, sizeof(*c));
    /* tests/test_matching.cpp:15-17.  The order in which the two rand() calls of one constructor
     * call are evaluated is unspecified in C++; g++ (the reference's compiler, Makefile:1)
     * evaluates call arguments right to left, which is what is done here. */
    const int hh = so_proptest_rand() % 1000 + 1;
    const int ww = so_proptest_rand() % 1000 + 1;
    c->w = ww; c->h = hh;
    const int ph = imin(so_proptest_rand() % imin(200, hh) + 1, hh);
    const int pw = imin(so_proptest_rand() % imin(200, ww) + 1, ww);
    c->pw = pw; c->ph = ph;
    c->im = (uint8_t*)malloc((size_t)ww * hh);
    c->patch = (uint8_t*)malloc((size_t)pw * ph);
    /* :21-23, int arithmetic wraps, % truncates toward zero, store to uchar keeps the low byte */
    for (int i = 0; i < ww; ++i)
        for (int j = 0; j < hh; ++j) {
            const int32_t v = (int32_t)(15485867u * (uint32_t)i + 15486883u * (uint32_t)j);
            c->im[(size_t)j * ww + i] = (uint8_t)(v % 7907);
        }
    /* :26-40 */
    const int leftpatch = pw / 2, toppatch = ph / 2;
    const int xpatch = so_proptest_rand() % ww;
    const int ypatch = so_proptest_rand() % hh;
    c->xpatch = xpatch; c->ypatch = ypatch;
    const int ptop = imax(ypatch - toppatch, 0), pleft = imax(xpatch - leftpatch, 0);
    const int pbottom = imin(ypatch - toppatch + ph, hh);
    const int pright = imin(xpatch - leftpatch + pw, ww);
    const int dptop = ptop - (ypatch - toppatch), dpleft = pleft - (xpatch - leftpatch);
    for (int i = 0; i < pw; ++i)
        for (int j = 0; j < ph; ++j)
            c->patch[(size_t)j * pw + i] = (uint8_t)(so_proptest_rand() % 255);
    for (int y = ptop; y < pbottom; ++y)
        for (int x = pleft; x < pright; ++x)
            c->patch[(size_t)(dptop + y - ptop) * pw + (dpleft + x - pleft)] = c->im[(size_t)y * ww + x];
    /* :44-46 */
    const int x0area = xpatch - (so_proptest_rand() % 100), y0area = ypatch - (so_proptest_rand() % 100);
    const int x1area = xpatch + (so_proptest_rand() % 100) + 1, y1area = ypatch + (so_proptest_rand() % 100) + 1;
    c->ax = x0area; c->ay = y0area; c->aw = x1area - x0area; c->ah = y1area - y0area;
    /* :49-58 */
    if (use_area_mask) {
        c->area_mask = (uint8_t*)malloc((size_t)c->aw * c->ah);
        for (int i = 0; i < c->aw; ++i)
            for (int j = 0; j < c->ah; ++j)
                c->area_mask[(size_t)j * c->aw + i] = ((so_proptest_rand() % 255) > 128) ? 1 : 0;
        c->area_mask[(size_t)(ypatch - y0area) * c->aw + (xpatch - x0area)] = 1;
    }
    /* :61-69 */
    if (use_patch_mask) {
        c->patch_mask = (uint8_t*)malloc((size_t)pw * ph);
        for (int i = 0; i < pw; ++i)
            for (int j = 0; j < ph; ++j)
                c->patch_mask[(size_t)j * pw + i] = ((so_proptest_rand() % 255) > 128) ? 1 : 0;
    }
}

void so_proptest_free(so_proptest_case* c) {
    free(c->im); free(c->patch); free(c->patch_mask); free(c->area_mask);
    memset(c, 0, sizeof(*c));
}

int so_proptest_accept(const so_proptest_case* c, int mx, int my, float response) {
    if (c->xpatch == mx && c->ypatch == my) return 1;                         /* :83 */
    if (0 > mx || 0 > my || mx >= c->w || my >= c->h || (double)response <= 0.99) return 0;   /* :86-88 */
    const int leftpatch = c->pw / 2, toppatch = c->ph / 2;
    for (int i = 0; i < c->pw; ++i)                                           /* :89-97 */
        for (int j = 0; j < c->ph; ++j)
            if (!c->patch_mask || c->patch_mask[(size_t)j * c->pw + i]) {
                const int yim = j + my - toppatch, xim = i + mx - leftpatch;
                if (xim >= 0 && yim >= 0 && xim < c->w && yim < c->h)
                    if (c->patch[(size_t)j * c->pw + i] != c->im[(size_t)yim * c->w + xim]) return 0;
            }
    return 1;
}
