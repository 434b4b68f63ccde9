/*
 * oracle/align_oracle.c -- CPU restatement of the alignment APPLY step of shinestacker's
 * align_images (reference src/shinestacker/algorithms/align.py:238-251, ALIGN_RIGID):
 *   img_warp = cv2.warpAffine(img_0, M, (w, h), borderMode, borderValue)        :242-244
 *   mask     = cv2.warpAffine(ones_u8, M, (w, h), BORDER_CONSTANT, 0)            :245-247
 *   blurred  = cv2.GaussianBlur(img_warp, (21, 21), sigmaX=border_blur)          :249
 *   img_warp[mask == 0] = blurred[mask == 0]                                     :251
 * TEST INFRASTRUCTURE ONLY (see pyramid_oracle.c).
 *
 * PARITY STATUS: "parity unpinned" -- OpenCV is absent here and the reference's tests hold no
 * pixel values for this step.  Restated from OpenCV's imgwarp.cpp [from memory]:
 *  - M (src->dst) is inverted in double (no WARP_INVERSE_MAP);
 *  - fixed-point source coordinates: AB_BITS = 10, INTER_BITS = 5:
 *      X0 = cvRound((iM01*y + iM02)*1024) + 16, adelta[x] = cvRound(iM00*x*1024),
 *      X = (X0 + adelta[x]) >> 5;  sx = X >> 5, fx = X & 31   (same for Y);
 *  - bilinear weights from the 32x32 table of products (1-f/32 | f/32):
 *      8-bit : 15-bit integer weights w*32768 (exact multiples of 32; the w = 1.0 entry saturates
 *              to 32767 and its missing unit goes to the diagonal tap), result (sum + 16384) >> 15;
 *      16-bit: float weights, sum in tap order, rounded half-to-even, saturated;
 *  - border: a tap outside the image takes the replicated edge pixel (REPLICATE) or the constant;
 *    a pixel whose four taps are all outside takes the constant directly;
 *  - the all-ones uint8 mask warped with constant 0 is therefore 1 iff the in-image 15-bit
 *    weights sum to >= 16384.
 *  - GaussianBlur on 8- and 16-bit images: OpenCV's bit-exact FIXED-POINT separable path (smooth.dispatch.cpp
 *    "running bit-exact version", fixedpoint.inl.hpp) [from memory, parity unpinned]:
 *      kernel   getGaussianKernelBitExact: t_i = exp(x_i^2 * (-0.125 / sigma^2)) with x_i = 2 i - (n - 1), normalised by
 *               1 / sum in double (OpenCV uses its softdouble exp; a last-bit difference of exp cannot survive the
 *               quantisation below except on an exact rounding tie), then getGaussianKernelFixedPoint_ED: 8 (8-bit
 *               images, ufixedpoint16) or 16 (16-bit images, ufixedpoint32) fractional bits, outer taps rounded half to
 *               even from the outside in with the rounding error carried to the next tap (error diffusion), mirrored,
 *               and the CENTRE tap takes what is left so that the taps sum to exactly 1.0;
 *      rows     R = sum_x k_x * src (8.8 / 16.16 fixed point: exact, the sum cannot overflow because sum k = 1.0);
 *      columns  S = sum_y k_y * R   (16.16 / 32.32), result (S + half) >> 16 / 32 -- round half up, saturated;
 *      border   REFLECT101 (cv2's BORDER_DEFAULT).  Integer sums: the order of the taps does not matter.
 *    Only pixels with mask == 0 (out-of-frame filler) use it.  (Rounds 1-2 evaluated a float32 blur here; OpenCV's
 *    IPP-backed builds may take yet another path for some sizes -- oracle/probe_cv2.py compares on a box with OpenCV.)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define ORC_API __attribute__((visibility("default")))

static inline int cv_round(double v) {
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (-2147483647 - 1);
    return (int)lrint(v);
}
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int r101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * n - 2 - i;
    }
    return i;
}

ORC_API void orc_invert_affine(const double* M, double* iM) {
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0.0 ? 1.0 / D : 0.0;
    double A11 = M[4] * D, A22 = M[0] * D;
    iM[0] = A11;
    iM[1] = M[1] * (-D);
    iM[3] = M[3] * (-D);
    iM[4] = A22;
    iM[2] = -iM[0] * M[2] - iM[1] * M[5];
    iM[5] = -iM[3] * M[2] - iM[4] * M[5];
}

/* 15-bit bilinear weights of table entry (fy, fx) */
static inline void wtab_i(int fx, int fy, int* iw) {
    iw[0] = (32 - fy) * (32 - fx) * 32;
    iw[1] = (32 - fy) * fx * 32;
    iw[2] = fy * (32 - fx) * 32;
    iw[3] = fy * fx * 32;
    if (fx == 0 && fy == 0) { iw[0] = 32767; iw[3] = 1; } /* short saturation + OpenCV's fix-up */
}

/* one destination pixel from the source position (X, Y) in 1/32 pixel: the remap core shared by warpAffine and
 * warpPerspective.  dtype: 0 = u8, 1 = u16.  mode: 0 = constant(border[c]), 1 = replicate. */
static void sample_px(const void* src_, void* dst_, uint8_t* valid, int h, int w, int dtype, int mode, const double* border,
                      int y, int x, int X, int Y) {
    const uint8_t* s8 = (const uint8_t*)src_;
    const uint16_t* s16 = (const uint16_t*)src_;
    uint8_t* d8 = (uint8_t*)dst_;
    uint16_t* d16 = (uint16_t*)dst_;
    int sx = X >> 5, sy = Y >> 5, fx = X & 31, fy = Y & 31;
    int in00 = sx >= 0 && sx < w && sy >= 0 && sy < h;
    int in01 = sx + 1 >= 0 && sx + 1 < w && sy >= 0 && sy < h;
    int in10 = sx >= 0 && sx < w && sy + 1 >= 0 && sy + 1 < h;
    int in11 = sx + 1 >= 0 && sx + 1 < w && sy + 1 >= 0 && sy + 1 < h;
    int iw[4];
    wtab_i(fx, fy, iw);
    if (valid) {
        int s = (in00 ? iw[0] : 0) + (in01 ? iw[1] : 0) + (in10 ? iw[2] : 0) + (in11 ? iw[3] : 0);
        valid[(size_t)y * w + x] = (uint8_t)(((s + 16384) >> 15) != 0);
    }
    int all_out = !(in00 || in01 || in10 || in11);
    int x0 = clampi(sx, 0, w - 1), x1 = clampi(sx + 1, 0, w - 1);
    int y0 = clampi(sy, 0, h - 1), y1 = clampi(sy + 1, 0, h - 1);
    for (int c = 0; c < 3; ++c) {
        size_t o = ((size_t)y * w + x) * 3 + c;
        double bv = border[c];
        if (dtype == 0) {
            int cb = clampi(cv_round(bv), 0, 255);
            int v00, v01, v10, v11;
            if (mode == 1) {
                v00 = s8[((size_t)y0 * w + x0) * 3 + c]; v01 = s8[((size_t)y0 * w + x1) * 3 + c];
                v10 = s8[((size_t)y1 * w + x0) * 3 + c]; v11 = s8[((size_t)y1 * w + x1) * 3 + c];
            } else {
                v00 = in00 ? s8[((size_t)sy * w + sx) * 3 + c] : cb;
                v01 = in01 ? s8[((size_t)sy * w + sx + 1) * 3 + c] : cb;
                v10 = in10 ? s8[((size_t)(sy + 1) * w + sx) * 3 + c] : cb;
                v11 = in11 ? s8[((size_t)(sy + 1) * w + sx + 1) * 3 + c] : cb;
            }
            int r = (mode == 0 && all_out) ? cb
                    : clampi((v00 * iw[0] + v01 * iw[1] + v10 * iw[2] + v11 * iw[3] + 16384) >> 15, 0, 255);
            d8[o] = (uint8_t)r;
        } else {
            int cb = clampi(cv_round(bv), 0, 65535);
            float v00, v01, v10, v11;
            if (mode == 1) {
                v00 = s16[((size_t)y0 * w + x0) * 3 + c]; v01 = s16[((size_t)y0 * w + x1) * 3 + c];
                v10 = s16[((size_t)y1 * w + x0) * 3 + c]; v11 = s16[((size_t)y1 * w + x1) * 3 + c];
            } else {
                v00 = in00 ? s16[((size_t)sy * w + sx) * 3 + c] : (float)cb;
                v01 = in01 ? s16[((size_t)sy * w + sx + 1) * 3 + c] : (float)cb;
                v10 = in10 ? s16[((size_t)(sy + 1) * w + sx) * 3 + c] : (float)cb;
                v11 = in11 ? s16[((size_t)(sy + 1) * w + sx + 1) * 3 + c] : (float)cb;
            }
            float wx1 = fx * (1.0f / 32), wx0 = 1.0f - wx1, wy1 = fy * (1.0f / 32), wy0 = 1.0f - wy1;
            volatile float p0 = v00 * (wy0 * wx0), p1 = v01 * (wy0 * wx1);
            volatile float p2 = v10 * (wy1 * wx0), p3 = v11 * (wy1 * wx1);
            volatile float s = p0 + p1;
            s = s + p2;
            s = s + p3;
            int r = (mode == 0 && all_out) ? cb : clampi((int)lrintf(s), 0, 65535);
            d16[o] = (uint16_t)r;
        }
    }
}

/* cv2.warpAffine.  valid (may be NULL): h*w bytes, the warped all-ones mask. */
ORC_API void orc_warp_affine(const void* src_, void* dst_, uint8_t* valid, int h, int w, int dtype,
                             const double* M, int mode, const double* border) {
    double iM[6];
    orc_invert_affine(M, iM);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        int X0 = cv_round((iM[1] * y + iM[2]) * 1024.0) + 16;
        int Y0 = cv_round((iM[4] * y + iM[5]) * 1024.0) + 16;
        for (int x = 0; x < w; ++x) {
            int X = (X0 + cv_round(iM[0] * x * 1024.0)) >> 5;
            int Y = (Y0 + cv_round(iM[3] * x * 1024.0)) >> 5;
            sample_px(src_, dst_, valid, h, w, dtype, mode, border, y, x, X, Y);
        }
    }
}

/* cv2.warpPerspective (align.py:231-237) [from memory of imgwarp.cpp's WarpPerspectiveInvoker]: M (3x3, src->dst) is
 * inverted (cv::invert of a 3x3 double matrix: cofactors times 1 / det); the image is walked in blocks of
 * bw0 = min(1024 / min(16, h), w) columns, and for the block starting at column bx:
 *     X0 = M0*bx + M1*y + M2,  Y0 = M3*bx + M4*y + M5,  W0 = M6*bx + M7*y + M8
 *     W = W0 + M6*x1;  W = W ? 32 / W : 0;  X = cvRound(clamp((X0 + M0*x1) * W)),  Y likewise      (x1 = x - bx)
 * in double (no fused multiply-add: the translation unit is built with -ffp-contract=off); then the same remap core. */
ORC_API void orc_invert_3x3(const double* m, double* o) {
    double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    d = d != 0.0 ? 1.0 / d : 0.0;
    o[0] = (m[4] * m[8] - m[5] * m[7]) * d; o[1] = (m[2] * m[7] - m[1] * m[8]) * d; o[2] = (m[1] * m[5] - m[2] * m[4]) * d;
    o[3] = (m[5] * m[6] - m[3] * m[8]) * d; o[4] = (m[0] * m[8] - m[2] * m[6]) * d; o[5] = (m[2] * m[3] - m[0] * m[5]) * d;
    o[6] = (m[3] * m[7] - m[4] * m[6]) * d; o[7] = (m[1] * m[6] - m[0] * m[7]) * d; o[8] = (m[0] * m[4] - m[1] * m[3]) * d;
}

ORC_API void orc_warp_perspective(const void* src_, void* dst_, uint8_t* valid, int h, int w, int dtype,
                                  const double* M9, int mode, const double* border) {
    double M[9];
    orc_invert_3x3(M9, M);
    const int bh0 = h < 16 ? h : 16;
    const int bw0 = 1024 / bh0 < w ? 1024 / bh0 : w;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int bx = 0; bx < w; bx += bw0) {
            const double X0 = M[0] * bx + M[1] * y + M[2], Y0 = M[3] * bx + M[4] * y + M[5], W0 = M[6] * bx + M[7] * y + M[8];
            for (int x1 = 0; x1 < bw0 && bx + x1 < w; ++x1) {
                double W = W0 + M[6] * x1;
                W = W != 0.0 ? 32.0 / W : 0.0;
                const double fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + M[0] * x1) * W));
                const double fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + M[3] * x1) * W));
                sample_px(src_, dst_, valid, h, w, dtype, mode, border, y, bx + x1, cv_round(fX), cv_round(fY));
            }
        }
}

ORC_API void orc_gauss_kernel_f32(int ksize, double sigma, float* k) {
    /* cv::getGaussianKernel(ksize, sigma, CV_32F) for sigma > 0 [from memory] */
    double sum = 0.0, *t = (double*)malloc(sizeof(double) * ksize);
    double scale2x = -0.5 / (sigma * sigma);
    for (int i = 0; i < ksize; ++i) {
        double x = i - (ksize - 1) * 0.5;
        t[i] = exp(scale2x * x * x);
        sum += t[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < ksize; ++i) k[i] = (float)(t[i] * sum);
    free(t);
}

/* cv2's fixed-point Gaussian taps (see header): `bits` fractional bits, k[ksize], sum(k) == 1 << bits */
ORC_API void orc_gauss_kernel_fixed(int ksize, double sigma, int bits, uint32_t* k) {
    const int n = ksize, n2 = (n - 1) / 2;
    double* t = (double*)malloc(sizeof(double) * (n2 + 1));
    if (sigma <= 0) sigma = n * 0.15 + 0.35;
    const double scale2x = -0.125 / (sigma * sigma);
    double sum = 0.0;
    for (int i = 0, x = 1 - n; i < n2; ++i, x += 2) {
        t[i] = exp((double)(x * x) * scale2x);
        sum += t[i];
    }
    sum *= 2.0;
    sum += 1.0;
    const double mul1 = 1.0 / sum;
    const double fixed_1 = (double)(1u << bits);
    int64_t acc = 0;
    double carry = 0.0;
    for (int i = 0; i < n2; ++i) {
        const double adj = t[i] * mul1 * fixed_1 + carry;
        const int64_t v = (int64_t)nearbyint(adj);   /* cvRound: half to even */
        carry = adj - (double)v;
        k[i] = k[n - 1 - i] = (uint32_t)v;
        acc += 2 * v;
    }
    k[n2] = (uint32_t)(((int64_t)1 << bits) - acc);
    free(t);
}

/* out = valid ? warp : GaussianBlur(warp)   (cv2's fixed-point blur: see header) */
ORC_API void orc_border_blur_composite(const void* warp_, const uint8_t* valid, void* out_, int h, int w,
                                       int dtype, int ksize, double sigma) {
    const int bits = dtype == 0 ? 8 : 16;
    uint32_t* k = (uint32_t*)malloc(sizeof(uint32_t) * ksize);
    orc_gauss_kernel_fixed(ksize, sigma, bits, k);
    const int r = ksize / 2;
    const uint8_t* s8 = (const uint8_t*)warp_;
    const uint16_t* s16 = (const uint16_t*)warp_;
    uint8_t* d8 = (uint8_t*)out_;
    uint16_t* d16 = (uint16_t*)out_;
    const int maxv = dtype == 0 ? 255 : 65535;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < 3; ++c) {
                size_t o = ((size_t)y * w + x) * 3 + c;
                if (valid[(size_t)y * w + x]) {
                    if (dtype == 0) d8[o] = s8[o]; else d16[o] = s16[o];
                    continue;
                }
                uint64_t acc = 0;
                for (int dy = 0; dy < ksize; ++dy) {
                    int yy = r101(y + dy - r, h);
                    uint64_t row = 0;
                    for (int dx = 0; dx < ksize; ++dx) {
                        int xx = r101(x + dx - r, w);
                        uint64_t v = dtype == 0 ? s8[((size_t)yy * w + xx) * 3 + c] : s16[((size_t)yy * w + xx) * 3 + c];
                        row += k[dx] * v;
                    }
                    acc += k[dy] * row;
                }
                uint64_t rr = (acc + ((uint64_t)1 << (2 * bits - 1))) >> (2 * bits);
                if (rr > (uint64_t)maxv) rr = (uint64_t)maxv;
                if (dtype == 0) d8[o] = (uint8_t)rr; else d16[o] = (uint16_t)rr;
            }
    free(k);
}
