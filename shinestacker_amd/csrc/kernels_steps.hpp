// kernels_steps.hpp -- the reference's PyramidStack step methods as standalone device operations (mi_pyr_step):
// convolve (pyramid.py:24-25), reduce_layer (:27-32), expand_layer (:34-46), fuse_laplacian (:48-55) and one collapse
// step (:57-63), each on whole images of 1 or 3 interleaved channels, in the reference's own evaluation order (the
// row-major 25-tap chain, REFLECT101; the same device functions the one-thread-per-output implementation is made of).
// Plain kernels: these entry points exist for callers and tests that drive the algorithm one method at a time, the fused
// level kernels (kernels_tiled.hpp / kernels_sep.hpp) are what a stack runs.
#pragma once
#include "common.hpp"
#include "kernels_tiled.hpp"

namespace mi {

// cv2.filter2D(image, -1, K, BORDER_REFLECT101) for C interleaved channels; STEP 2 = the decimated form convolve()[::2, ::2]
template <int C, bool FMA, int STEP>
__global__ void step_convolve(const float* __restrict__ src, int h, int w, float* __restrict__ dst, int ho, int wo, K25 K) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= ho || j >= wo) return;
    float s[C];
#pragma unroll
    for (int c = 0; c < C; ++c) s[c] = 0.f;
#pragma unroll
    for (int ty = 0; ty < 5; ++ty) {
        const float* row = src + (size_t)r101(STEP * i + ty - 2, h) * w * C;
#pragma unroll
        for (int tx = 0; tx < 5; ++tx) {
            const float* p = row + (size_t)r101(STEP * j + tx - 2, w) * C;
            const float k = K.k[ty * 5 + tx];
#pragma unroll
            for (int c = 0; c < C; ++c) s[c] = mac<FMA>(k, p[c], s[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) dst[((size_t)i * wo + j) * C + c] = s[c];
}

// expand_layer: the 2 hs x 2 ws zero-stuffed image through the same filter, times 4 (the taps that fall on stuffed zeros
// are skipped: a zero product leaves the chain's value as it is)
template <int C, bool FMA>
__global__ void step_expand(const float* __restrict__ src, int hs, int ws, float* __restrict__ dst, K25 K) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    const int H2 = 2 * hs, W2 = 2 * ws;
    if (y >= H2 || x >= W2) return;
    float s[C];
#pragma unroll
    for (int c = 0; c < C; ++c) s[c] = 0.f;
#pragma unroll
    for (int ty = 0; ty < 5; ++ty) {
        const int yy = r101(y + ty - 2, H2);
        if (yy & 1) continue;
#pragma unroll
        for (int tx = 0; tx < 5; ++tx) {
            const int xx = r101(x + tx - 2, W2);
            if (xx & 1) continue;
            const float* p = src + ((size_t)(yy >> 1) * ws + (xx >> 1)) * C;
            const float k = K.k[ty * 5 + tx];
#pragma unroll
            for (int c = 0; c < C; ++c) s[c] = mac<FMA>(k, p[c], s[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) dst[((size_t)y * W2 + x) * C + c] = 4.0f * s[c];
}

// np.square(cv2.cvtColor(lap, BGR2GRAY)) of n stacked H x W x 3 images
template <bool FMA>
__global__ void step_gray_sq(const float* __restrict__ lap, size_t npix, float* __restrict__ q) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float g = gray_of<FMA>(lap[3 * i], lap[3 * i + 1], lap[3 * i + 2]);
    q[i] = g * g;
}

// best = np.argmax(energies, axis=0) (first maximum); fused = sum_i where(best == i, lap_i, 0)  (-0 -> +0)
__global__ void step_fuse(const float* __restrict__ e, const float* __restrict__ laps, int n, size_t npix, float* __restrict__ fused) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float be = e[p];
    int bi = 0;
    for (int i = 1; i < n; ++i) {
        const float v = e[(size_t)i * npix + p];
        if (v > be) { be = v; bi = i; }
    }
    const float* l = laps + ((size_t)bi * npix + p) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) fused[3 * p + c] = 0.0f + l[c];
}

// one step of collapse: expanded[:h, :w] + layer
__global__ void step_add_crop(const float* __restrict__ up, int wu, const float* __restrict__ layer, int h, int w, int c,
                              float* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (y >= h || x >= w) return;
    for (int k = 0; k < c; ++k) out[((size_t)y * w + x) * c + k] = up[((size_t)y * wu + x) * c + k] + layer[((size_t)y * w + x) * c + k];
}

// np.clip(np.abs(img), 0, maxv)
__global__ void step_clip_abs(const float* __restrict__ in, size_t n, float maxv, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = fabsf(in[i]);
    out[i] = v > maxv ? maxv : v;
}

// host side: one of the three image-to-image operations (the op codes are the public MI_PYR_* of mi355stack.h)
template <int C, bool FMA>
inline void pyr_step_launch(int op, hipStream_t st, const float* in, int h, int w, float* out, const K25& K) {
    const dim3 blk(64, 4);
    if (op == 0)
        hipLaunchKernelGGL((step_convolve<C, FMA, 1>), dim3(cdiv(w, 64), cdiv(h, 4)), blk, 0, st, in, h, w, out, h, w, K);
    else if (op == 1)
        hipLaunchKernelGGL((step_convolve<C, FMA, 2>), dim3(cdiv((w + 1) / 2, 64), cdiv((h + 1) / 2, 4)), blk, 0, st, in, h, w, out,
                           (h + 1) / 2, (w + 1) / 2, K);
    else
        hipLaunchKernelGGL((step_expand<C, FMA>), dim3(cdiv(2 * w, 64), cdiv(2 * h, 4)), blk, 0, st, in, h, w, out, K);
}

}  // namespace mi
