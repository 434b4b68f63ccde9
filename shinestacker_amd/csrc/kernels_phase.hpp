// kernels_phase.hpp -- translation between two gray planes by PHASE CORRELATION (north_star: "the ECC/phase-correlation
// warp-affine alignment loop"): the coarse initialiser of the ECC estimator (kernels_ecc.hpp).  The reference has no
// counterpart in its own code (its estimate is OpenCV's feature matcher + RANSAC, algorithms/align.py:90-151); the recipe
// is cv2.phaseCorrelate's [from memory]: Hann window, 2-D DFT of both planes, normalised cross-power spectrum
// R = A conj(B) / |A conj(B)|, inverse DFT, arg-max of the correlation surface and the weighted centroid of the 5 x 5
// window around it for the sub-pixel part, response = that window's sum over P Q (1 for a pure shift, ~0 for unrelated planes).  Validated against a NumPy
// float64 statement of the same recipe (tests/test_gpu_ecc.py) and by the shifts it recovers.
//
// The planes are small (a pyramid level of at most 512 pixels per side, zero-padded to powers of two): one workgroup
// transforms one line of up to 1024 complex points in LDS (radix-2, bit-reversed load), rows then columns.
#pragma once
#include "common.hpp"

namespace mi {

constexpr int PC_MAX_N = 1024;   // longest line one workgroup transforms

// plane (h x w floats, row stride w) -> complex P x Q (row stride Q), Hann-windowed, zero-padded
__global__ void pc_prepare(const float* __restrict__ src, int h, int w, float2* __restrict__ dst, int P, int Q) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= Q || y >= P) return;
    float v = 0.f;
    if (x < w && y < h) {
        // cv2.createHanningWindow: 0.5 (1 - cos(2 pi i / (n - 1)))
        const float wy = h > 1 ? 0.5f * (1.0f - cospif(2.0f * (float)y / (float)(h - 1))) : 1.0f;
        const float wx = w > 1 ? 0.5f * (1.0f - cospif(2.0f * (float)x / (float)(w - 1))) : 1.0f;
        v = src[(size_t)y * w + x] * (wy * wx);
    }
    dst[(size_t)y * Q + x] = make_float2(v, 0.f);
}

// in-place DFT of `nlines` lines of n = 2^log2n complex points: element i of line l at data[l * line_stride + i * elem_stride].
// blockDim.x = n / 2 threads, blockIdx.x = line.  INVERSE: conjugate twiddles, no 1 / n scaling.
template <bool INVERSE>
__global__ void pc_fft_lines(float2* __restrict__ data, int n, int log2n, size_t elem_stride, size_t line_stride) {
    __shared__ float2 s[PC_MAX_N];
    float2* line = data + (size_t)blockIdx.x * line_stride;
    const int t = threadIdx.x;
    for (int i = t; i < n; i += blockDim.x) s[__brev((unsigned)i) >> (32 - log2n)] = line[(size_t)i * elem_stride];
    __syncthreads();
    for (int sft = 0; sft < log2n; ++sft) {
        const int half = 1 << sft;
        const int k = t & (half - 1);                 // position inside the butterfly group
        const int i0 = ((t >> sft) << (sft + 1)) + k, i1 = i0 + half;
        float sn, cs;
        sincospif((INVERSE ? 1.0f : -1.0f) * (float)k / (float)half, &sn, &cs);   // e^{-+ i pi k / half}
        const float2 a = s[i0], b = s[i1];
        const float2 tw = make_float2(b.x * cs - b.y * sn, b.x * sn + b.y * cs);
        s[i0] = make_float2(a.x + tw.x, a.y + tw.y);
        s[i1] = make_float2(a.x - tw.x, a.y - tw.y);
        __syncthreads();
    }
    for (int i = t; i < n; i += blockDim.x) line[(size_t)i * elem_stride] = s[i];
}

// R = A conj(B) / |A conj(B)| in place of A
__global__ void pc_cross_power(float2* __restrict__ a, const float2* __restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 x = a[i], y = b[i];
    const float re = x.x * y.x + x.y * y.y, im = x.y * y.x - x.x * y.y;
    const float mag = sqrtf(re * re + im * im);
    const float inv = mag > 1e-20f ? 1.0f / mag : 0.0f;
    a[i] = make_float2(re * inv, im * inv);
}

// arg-max of the real part (first maximum in raster order), weighted centroid of the 5 x 5 window around it (wrapping),
// response = window sum / (P Q).  One workgroup of 1024 threads.  out[0] = dx, out[1] = dy (the peak
// position as a signed shift: positions beyond half the size wrap to negative), out[2] = response.
__global__ __launch_bounds__(1024) void pc_peak(const float2* __restrict__ c, int P, int Q, double* __restrict__ out) {
    __shared__ float sv[1024];
    __shared__ int si[1024];
    const int t = threadIdx.x, n = P * Q;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = t; i < n; i += 1024) {
        const float v = c[i].x;
        if (v > best) { best = v; bi = i; }
    }
    sv[t] = best; si[t] = bi;
    __syncthreads();
    for (int d = 512; d > 0; d >>= 1) {
        if (t < d) {
            const float v2 = sv[t + d];
            const int i2 = si[t + d];
            if (v2 > sv[t] || (v2 == sv[t] && i2 < si[t])) { sv[t] = v2; si[t] = i2; }
        }
        __syncthreads();
    }
    if (t == 0) {
        const int py = si[0] / Q, px = si[0] - py * Q;
        double m = 0.0, mx = 0.0, my = 0.0;
        for (int dy = -2; dy <= 2; ++dy)
            for (int dx = -2; dx <= 2; ++dx) {
                const int yy = (py + dy + P) % P, xx = (px + dx + Q) % Q;
                const double v = (double)c[(size_t)yy * Q + xx].x;
                m += v; mx += v * (double)(px + dx); my += v * (double)(py + dy);
            }
        double cx = m != 0.0 ? mx / m : (double)px, cy = m != 0.0 ? my / m : (double)py;
        if (cx > 0.5 * Q) cx -= Q;
        if (cy > 0.5 * P) cy -= P;
        out[0] = cx;
        out[1] = cy;
        out[2] = m / ((double)P * (double)Q);   // a perfect match is a delta of height P Q (unnormalised inverse DFT)
    }
}

}  // namespace mi
