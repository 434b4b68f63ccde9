// kernels_align.hpp -- the alignment APPLY step of align_images (reference
// src/shinestacker/algorithms/align.py:238-251, ALIGN_RIGID): cv2.warpAffine with
// replicate / constant border, the warped all-ones mask, and the blurred-border composite.
// Arithmetic follows oracle/align_oracle.c operation by operation (see its header for the
// OpenCV semantics restated from memory and the parity status).
#pragma once
#include "common.hpp"

namespace mi {

struct AffineArgs {
    double iM[6];      // inverted transform (dst -> src), double
    int h, w;
    int mode;          // 0 constant, 1 replicate
    int border[3];     // constant border value per channel, already rounded / saturated
};

__device__ __forceinline__ int cv_round_d(double v) {
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (-2147483647 - 1);
    return (int)rint(v);  // round half to even, as cvRound / lrint
}

template <typename T>
__global__ void warp_affine_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                   uint8_t* __restrict__ valid, AffineArgs a) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int h = a.h, w = a.w;
    if (x >= w || y >= h) return;
    const int X0 = cv_round_d((a.iM[1] * y + a.iM[2]) * 1024.0) + 16;
    const int Y0 = cv_round_d((a.iM[4] * y + a.iM[5]) * 1024.0) + 16;
    const int X = (X0 + cv_round_d(a.iM[0] * x * 1024.0)) >> 5;
    const int Y = (Y0 + cv_round_d(a.iM[3] * x * 1024.0)) >> 5;
    const int sx = X >> 5, sy = Y >> 5, fx = X & 31, fy = Y & 31;
    const bool inx0 = sx >= 0 && sx < w, inx1 = sx + 1 >= 0 && sx + 1 < w;
    const bool iny0 = sy >= 0 && sy < h, iny1 = sy + 1 >= 0 && sy + 1 < h;
    const bool in00 = inx0 && iny0, in01 = inx1 && iny0, in10 = inx0 && iny1, in11 = inx1 && iny1;
    int iw0 = (32 - fy) * (32 - fx) * 32, iw1 = (32 - fy) * fx * 32, iw2 = fy * (32 - fx) * 32, iw3 = fy * fx * 32;
    if (fx == 0 && fy == 0) { iw0 = 32767; iw3 = 1; }
    if (valid) {
        const int s = (in00 ? iw0 : 0) + (in01 ? iw1 : 0) + (in10 ? iw2 : 0) + (in11 ? iw3 : 0);
        valid[(size_t)y * w + x] = (uint8_t)(((s + 16384) >> 15) != 0);
    }
    const bool all_out = !(in00 || in01 || in10 || in11);
    const int x0 = min(max(sx, 0), w - 1), x1 = min(max(sx + 1, 0), w - 1);
    const int y0 = min(max(sy, 0), h - 1), y1 = min(max(sy + 1, 0), h - 1);
    const bool rep = a.mode == 1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const size_t o = ((size_t)y * w + x) * 3 + c;
        const int cb = a.border[c];
        // replicate: clamped taps; constant: in-image taps or the border value
        const T t00 = (rep || in00) ? src[((size_t)(rep ? y0 : sy) * w + (rep ? x0 : sx)) * 3 + c] : (T)cb;
        const T t01 = (rep || in01) ? src[((size_t)(rep ? y0 : sy) * w + (rep ? x1 : sx + 1)) * 3 + c] : (T)cb;
        const T t10 = (rep || in10) ? src[((size_t)(rep ? y1 : sy + 1) * w + (rep ? x0 : sx)) * 3 + c] : (T)cb;
        const T t11 = (rep || in11) ? src[((size_t)(rep ? y1 : sy + 1) * w + (rep ? x1 : sx + 1)) * 3 + c] : (T)cb;
        int r;
        if constexpr (sizeof(T) == 1) {
            r = ((int)t00 * iw0 + (int)t01 * iw1 + (int)t10 * iw2 + (int)t11 * iw3 + 16384) >> 15;
            r = min(max(r, 0), 255);
        } else {
            const float wx1 = fx * (1.0f / 32), wx0 = 1.0f - wx1, wy1 = fy * (1.0f / 32), wy0 = 1.0f - wy1;
            const float p0 = (float)t00 * (wy0 * wx0), p1 = (float)t01 * (wy0 * wx1);
            const float p2 = (float)t10 * (wy1 * wx0), p3 = (float)t11 * (wy1 * wx1);
            float s = p0 + p1;
            s = s + p2;
            s = s + p3;
            r = min(max((int)rintf(s), 0), 65535);
        }
        if (!rep && all_out) r = cb;
        dst[o] = (T)r;
    }
}

struct GaussArgs {
    float k[32];
    int ksize;
};

// out = valid ? warp : gaussian_blur(warp); blur = horizontal pass then vertical pass in float32,
// taps in index order, REFLECT101, round-half-even + saturate (align_oracle.c).
template <typename T>
__global__ void border_blur_composite_kernel(const T* __restrict__ warp, const uint8_t* __restrict__ valid,
                                             T* __restrict__ out, int h, int w, GaussArgs g) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    const size_t px = (size_t)y * w + x;
    if (valid[px]) {
        out[px * 3 + 0] = warp[px * 3 + 0];
        out[px * 3 + 1] = warp[px * 3 + 1];
        out[px * 3 + 2] = warp[px * 3 + 2];
        return;
    }
    const int r = g.ksize / 2;
    const int maxv = sizeof(T) == 1 ? 255 : 65535;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int dy = 0; dy < g.ksize; ++dy) {
        const int yy = r101_loop(y + dy - r, h);
        float row[3] = {0.f, 0.f, 0.f};
        for (int dx = 0; dx < g.ksize; ++dx) {
            const int xx = r101_loop(x + dx - r, w);
            const T* p = warp + ((size_t)yy * w + xx) * 3;
            const float k = g.k[dx];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float pr = k * (float)p[c];
                row[c] = row[c] + pr;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float q = g.k[dy] * row[c];
            acc[c] = acc[c] + q;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) out[px * 3 + c] = (T)min(max((int)rintf(acc[c]), 0), maxv);
}

}  // namespace mi
