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

// one destination pixel: the 3 channel values and the mask bit
template <typename T>
__device__ __forceinline__ void warp_pixel(const T* __restrict__ src, const AffineArgs& a, int x, int X0, int Y0,
                                           int out[3], int& ok) {
    const int h = a.h, w = a.w;
    const int X = (X0 + cv_round_d(a.iM[0] * x * 1024.0)) >> 5;
    const int Y = (Y0 + cv_round_d(a.iM[3] * x * 1024.0)) >> 5;
    const int sx = X >> 5, sy = Y >> 5, fx = X & 31, fy = Y & 31;
    const bool inx0 = sx >= 0 && sx < w, inx1 = sx + 1 >= 0 && sx + 1 < w;
    const bool iny0 = sy >= 0 && sy < h, iny1 = sy + 1 >= 0 && sy + 1 < h;
    const bool in00 = inx0 && iny0, in01 = inx1 && iny0, in10 = inx0 && iny1, in11 = inx1 && iny1;
    int iw0 = (32 - fy) * (32 - fx) * 32, iw1 = (32 - fy) * fx * 32, iw2 = fy * (32 - fx) * 32, iw3 = fy * fx * 32;
    if (fx == 0 && fy == 0) { iw0 = 32767; iw3 = 1; }
    {
        const int s = (in00 ? iw0 : 0) + (in01 ? iw1 : 0) + (in10 ? iw2 : 0) + (in11 ? iw3 : 0);
        ok = ((s + 16384) >> 15) != 0;
    }
    const bool all_out = !(in00 || in01 || in10 || in11);
    const int x0 = min(max(sx, 0), w - 1), x1 = min(max(sx + 1, 0), w - 1);
    const int y0 = min(max(sy, 0), h - 1), y1 = min(max(sy + 1, 0), h - 1);
    const bool rep = a.mode == 1;
    // replicate: clamped taps; constant: in-image taps or the border value.  A tap's three channels
    // come in with one (unaligned) 32-bit load -- the byte loads were the bottleneck (12 per pixel
    // through the texture-address unit); the very last pixel of the image is read one byte early so
    // the load never leaves the buffer.
    const size_t last = (size_t)h * w - 1;
    auto tap = [&](int yy, int xx, int t[3]) {
        const size_t pi = (size_t)yy * w + xx;
        if constexpr (sizeof(T) == 1) {
            uint32_t u;
            if (pi != last) {
                __builtin_memcpy(&u, src + pi * 3, 4);
            } else {
                __builtin_memcpy(&u, src + pi * 3 - 1, 4);
                u >>= 8;
            }
            t[0] = u & 255u; t[1] = (u >> 8) & 255u; t[2] = (u >> 16) & 255u;
        } else {
            uint32_t u;
            uint16_t v2;
            __builtin_memcpy(&u, src + pi * 3, 4);
            __builtin_memcpy(&v2, src + pi * 3 + 2, 2);
            t[0] = u & 65535u; t[1] = u >> 16; t[2] = v2;
        }
    };
    int q00[3], q01[3], q10[3], q11[3];
    tap(rep ? y0 : min(max(sy, 0), h - 1), rep ? x0 : min(max(sx, 0), w - 1), q00);
    tap(rep ? y0 : min(max(sy, 0), h - 1), rep ? x1 : min(max(sx + 1, 0), w - 1), q01);
    tap(rep ? y1 : min(max(sy + 1, 0), h - 1), rep ? x0 : min(max(sx, 0), w - 1), q10);
    tap(rep ? y1 : min(max(sy + 1, 0), h - 1), rep ? x1 : min(max(sx + 1, 0), w - 1), q11);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int cb = a.border[c];
        const T t00 = (rep || in00) ? (T)q00[c] : (T)cb;
        const T t01 = (rep || in01) ? (T)q01[c] : (T)cb;
        const T t10 = (rep || in10) ? (T)q10[c] : (T)cb;
        const T t11 = (rep || in11) ? (T)q11[c] : (T)cb;
        int r;
        if constexpr (sizeof(T) == 1) {
            r = ((int)t00 * iw0 + (int)t01 * iw1 + (int)t10 * iw2 + (int)t11 * iw3 + 16384) >> 15;
            r = min(max(r, 0), 255);
        } else {
            const float wx1 = fx * (1.0f / 32), wx0 = 1.0f - wx1, wy1 = fy * (1.0f / 32), wy0 = 1.0f - wy1;
            const float q0 = (float)t00 * (wy0 * wx0), q1 = (float)t01 * (wy0 * wx1);
            const float q2 = (float)t10 * (wy1 * wx0), q3 = (float)t11 * (wy1 * wx1);
            float s = q0 + q1;
            s = s + q2;
            s = s + q3;
            r = min(max((int)rintf(s), 0), 65535);
        }
        if (!rep && all_out) r = cb;
        out[c] = r;
    }
}

// The same pixel when all four taps are known to lie inside the image (the caller checked the thread's first and last
// pixel; an affine map keeps the ones between them between): no clamping, no per-tap in-image flags, mask = 1 -- the
// kernel is instruction-bound, and that bookkeeping was a third of its instructions.  Same arithmetic, same results.
template <typename T>
__device__ __forceinline__ void warp_pixel_inside(const T* __restrict__ src, const AffineArgs& a, int x, int X0, int Y0,
                                                  int out[3]) {
    const int w = a.w;
    const int X = (X0 + cv_round_d(a.iM[0] * x * 1024.0)) >> 5;
    const int Y = (Y0 + cv_round_d(a.iM[3] * x * 1024.0)) >> 5;
    const int sx = X >> 5, sy = Y >> 5, fx = X & 31, fy = Y & 31;
    int iw0 = (32 - fy) * (32 - fx) * 32, iw1 = (32 - fy) * fx * 32, iw2 = fy * (32 - fx) * 32, iw3 = fy * fx * 32;
    if (fx == 0 && fy == 0) { iw0 = 32767; iw3 = 1; }
    const T* p0 = src + ((size_t)sy * w + sx) * 3;
    const T* p1 = p0 + (size_t)w * 3;
    int q[4][3];
    if constexpr (sizeof(T) == 1) {
        uint32_t u[4];
        __builtin_memcpy(&u[0], p0, 4);     __builtin_memcpy(&u[1], p0 + 3, 4);
        __builtin_memcpy(&u[2], p1, 4);     __builtin_memcpy(&u[3], p1 + 3, 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) { q[t][0] = u[t] & 255u; q[t][1] = (u[t] >> 8) & 255u; q[t][2] = (u[t] >> 16) & 255u; }
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const T* p = (t < 2 ? p0 : p1) + (t & 1) * 3;
            uint32_t u;
            uint16_t v2;
            __builtin_memcpy(&u, p, 4);
            __builtin_memcpy(&v2, p + 2, 2);
            q[t][0] = u & 65535u; q[t][1] = u >> 16; q[t][2] = v2;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int r;
        if constexpr (sizeof(T) == 1) {
            r = (q[0][c] * iw0 + q[1][c] * iw1 + q[2][c] * iw2 + q[3][c] * iw3 + 16384) >> 15;
            r = min(max(r, 0), 255);
        } else {
            const float wx1 = fx * (1.0f / 32), wx0 = 1.0f - wx1, wy1 = fy * (1.0f / 32), wy0 = 1.0f - wy1;
            const float q0 = (float)q[0][c] * (wy0 * wx0), q1 = (float)q[1][c] * (wy0 * wx1);
            const float q2 = (float)q[2][c] * (wy1 * wx0), q3 = (float)q[3][c] * (wy1 * wx1);
            float s = q0 + q1;
            s = s + q2;
            s = s + q3;
            r = min(max((int)rintf(s), 0), 65535);
        }
        out[c] = r;
    }
}

// Four consecutive destination pixels per thread: 12 (uint8) / 24 (uint16) contiguous output bytes
// and the 4 mask bytes leave as whole dwords when the row start allows it (VEC: w % 4 == 0 and
// 4-byte aligned images).
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void warp_affine_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                                          uint8_t* __restrict__ valid, AffineArgs a) {
    const int xq = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int h = a.h, w = a.w;
    if (xq >= w || y >= h) return;
    const int X0 = cv_round_d((a.iM[1] * y + a.iM[2]) * 1024.0) + 16;
    const int Y0 = cv_round_d((a.iM[4] * y + a.iM[5]) * 1024.0) + 16;
    int v[4][3], ok[4];
    // source positions of the thread's first and last pixel: both (with their +1 taps, and not touching the image's
    // very last pixel, whose 4-byte tap load would leave the buffer) inside the image -> the fast path for all four
    bool inside = xq + 3 < w;
    if (inside) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int x = xq + 3 * e;
            const int sx = ((X0 + cv_round_d(a.iM[0] * x * 1024.0)) >> 5) >> 5;
            const int sy = ((Y0 + cv_round_d(a.iM[3] * x * 1024.0)) >> 5) >> 5;
            inside = inside && sx >= 0 && sx + 1 < w && sy >= 0 && sy + 1 < h - 1;
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        ok[p] = 0;
        v[p][0] = v[p][1] = v[p][2] = 0;
        if (inside) {
            warp_pixel_inside<T>(src, a, xq + p, X0, Y0, v[p]);
            ok[p] = 1;
        } else if (xq + p < w) {
            warp_pixel<T>(src, a, xq + p, X0, Y0, v[p], ok[p]);
        }
    }
    const size_t px = (size_t)y * w + xq;
    if constexpr (VEC) {
        if constexpr (sizeof(T) == 1) {
            uint32_t o[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                o[d] = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) o[d] |= (uint32_t)v[(4 * d + b) / 3][(4 * d + b) % 3] << (8 * b);
            }
            uint32_t* d4 = reinterpret_cast<uint32_t*>(dst + px * 3);
            d4[0] = o[0]; d4[1] = o[1]; d4[2] = o[2];
        } else {
            uint32_t* d4 = reinterpret_cast<uint32_t*>(dst + px * 3);
#pragma unroll
            for (int d = 0; d < 6; ++d)
                d4[d] = (uint32_t)v[(2 * d) / 3][(2 * d) % 3] | ((uint32_t)v[(2 * d + 1) / 3][(2 * d + 1) % 3] << 16);
        }
        if (valid)
            *reinterpret_cast<uint32_t*>(valid + px) =
                (uint32_t)ok[0] | ((uint32_t)ok[1] << 8) | ((uint32_t)ok[2] << 16) | ((uint32_t)ok[3] << 24);
    } else {
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (xq + p < w) {
                dst[(px + p) * 3 + 0] = (T)v[p][0];
                dst[(px + p) * 3 + 1] = (T)v[p][1];
                dst[(px + p) * 3 + 2] = (T)v[p][2];
                if (valid) valid[px + p] = (uint8_t)ok[p];
            }
    }
}

struct GaussArgs {
    float k[32];
    int ksize;
};

// the three channels of pixel `pi` as floats: one (unaligned) 32-bit load for 8-bit images -- three byte
// loads per tap kept the texture-address unit busier than the arithmetic -- the very last pixel of the
// image is read one byte early so the load never leaves the buffer; 16-bit: a 32-bit and a 16-bit load
template <typename T>
__device__ __forceinline__ void load_px3(const T* __restrict__ img, size_t pi, size_t last, float out[3]) {
    if constexpr (sizeof(T) == 1) {
        uint32_t u;
        if (pi != last) {
            __builtin_memcpy(&u, img + pi * 3, 4);
        } else {
            __builtin_memcpy(&u, img + pi * 3 - 1, 4);
            u >>= 8;
        }
        out[0] = (float)(u & 255u); out[1] = (float)((u >> 8) & 255u); out[2] = (float)((u >> 16) & 255u);
    } else {
        uint32_t u;
        uint16_t v2;
        __builtin_memcpy(&u, img + pi * 3, 4);
        __builtin_memcpy(&v2, img + pi * 3 + 2, 2);
        out[0] = (float)(u & 65535u); out[1] = (float)(u >> 16); out[2] = (float)v2;
    }
}

// the same pixel kept raw (8-bit: one dword, 16-bit: two), decoded later
template <typename T>
__device__ __forceinline__ uint2 load_px3_raw(const T* __restrict__ img, size_t pi, size_t last) {
    uint2 v = {0u, 0u};
    if constexpr (sizeof(T) == 1) {
        if (pi != last) {
            __builtin_memcpy(&v.x, img + pi * 3, 4);
        } else {
            __builtin_memcpy(&v.x, img + pi * 3 - 1, 4);
            v.x >>= 8;
        }
    } else {
        uint16_t v2;
        __builtin_memcpy(&v.x, img + pi * 3, 4);
        __builtin_memcpy(&v2, img + pi * 3 + 2, 2);
        v.y = v2;
    }
    return v;
}
template <typename T>
__device__ __forceinline__ void decode_px3(uint2 v, float out[3]) {
    if constexpr (sizeof(T) == 1) {
        out[0] = (float)(v.x & 255u); out[1] = (float)((v.x >> 8) & 255u); out[2] = (float)((v.x >> 16) & 255u);
    } else {
        out[0] = (float)(v.x & 65535u); out[1] = (float)(v.x >> 16); out[2] = (float)v.y;
    }
}

// gaussian_blur(img) at one pixel: horizontal pass then vertical pass in float32, taps in index
// order, REFLECT101, round-half-even + saturate (align_oracle.c)
// `kof(i)` returns tap i of the Gaussian: the callers keep the taps in a VGPR (lane i holds tap i) and read them
// with v_readlane -- indexing the kernel-argument array with a loop counter costs a scalar load and an lgkmcnt
// wait per tap, ~0.15 us each, 441 of them per pixel
template <typename T, typename KOf>
__device__ __forceinline__ void blur_at(const T* __restrict__ img, int h, int w, int y, int x, const GaussArgs& g,
                                        int out[3], KOf kof) {
    const int r = g.ksize / 2;
    const int maxv = sizeof(T) == 1 ? 255 : 65535;
    const size_t last = (size_t)h * w - 1;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int dy = 0; dy < g.ksize; ++dy) {
        const int yy = r101_loop(y + dy - r, h);
        float row[3] = {0.f, 0.f, 0.f};
        // the window row's loads are issued together (groups of 8), not one per dependent multiply-add: a wave that
        // comes here alone -- a run that crosses a row end -- otherwise pays 441 memory latencies in a row (~75 us)
        for (int d0 = 0; d0 < g.ksize; d0 += 8) {
            uint2 raw[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int dx = min(d0 + u, g.ksize - 1);
                raw[u] = load_px3_raw<T>(img, (size_t)yy * w + r101_loop(x + dx - r, w), last);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (d0 + u < g.ksize) {
                    float p[3];
                    decode_px3<T>(raw[u], p);
                    const float k = kof(d0 + u);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float pr = k * p[c];
                        row[c] = row[c] + pr;
                    }
                }
            }
        }
        const float kd = kof(dy);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float q = kd * row[c];
            acc[c] = acc[c] + q;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = min(max((int)rintf(acc[c]), 0), maxv);
}

// out = valid ? warp : gaussian_blur(warp), in place on `img` in two sparse passes: the pixels whose
// mask is 0 (a frame along the borders: a few columns at the sides, wedges where the frame rotated)
// get their blurred value computed from the untouched image into `side` (same index), then copied back.
//
// Pass 0 (collect), one wave per 64 consecutive pixels (four waves per workgroup).  A wave with no masked pixel leaves at once.
// A wave with many does one pixel per lane (blur_at).  A wave with few (the side columns: 1-3 lanes)
// works through them 64 / ksize at a time instead of idling 60 lanes for 441 taps: ksize lanes per
// pixel, lane r does the horizontal pass of window row r (taps in index order), the first lane of the
// group adds the ksize row results in row order (shuffles, no LDS) -- the same float32 operations in
// the same order as align_oracle.c either way.
template <typename T>
__global__ __launch_bounds__(256) void border_blur_collect(const T* __restrict__ img, const uint8_t* __restrict__ valid,
                                                           T* __restrict__ side, int h, int w, GaussArgs g) {
    // four independent waves per workgroup (one 64-pixel chunk each): a quarter of the workgroups to dispatch --
    // with one-wave workgroups the launch of the 375 000 of a 24 MP frame alone took 80 us
    const int lane = threadIdx.x & 63;
    const size_t n = (size_t)h * w;
    const size_t base = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    const size_t pi = base + lane;
    const bool masked = pi < n && valid[pi] == 0;
    const unsigned long long ballot = __ballot(masked);
    if (ballot == 0) return;
    const int nm = __popcll(ballot);
    const int ks = g.ksize, r = ks / 2;
    const int maxv = sizeof(T) == 1 ? 255 : 65535;
    const int per = 64 / ks;  // pixels per cooperative round
    const float kreg = g.k[lane & 31];   // lane i holds tap i (ksize <= 31)
    auto kof = [&](int i) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(kreg), i)); };
    if (nm > 4 * per) {
        // A run of masked pixels (a row along the top or bottom edge): the wave shares each window row through LDS: 64 + 2r pixels are loaded once (two loads per lane instead of
        // ksize per lane, and the reflection map per loaded pixel instead of per tap) and every lane sums its ksize
        // neighbours from there -- the operations of blur_at in the same order, a third of the instructions.
        __shared__ float sRow[4][64 + 2 * 10][3];
        const int y_first = (int)(base / w), x_first = (int)(base - (size_t)y_first * w);
        if (r <= 10) {
            float (*s)[3] = sRow[threadIdx.x >> 6];
            // a chunk that crosses a row end is two runs (the tail of row y_first, the head of the next row): the
            // procedure runs once per run, with lane 0 at a virtual column xs of that row
            const int nruns = x_first + 63 >= w ? 2 : 1;
            for (int run = 0; run < nruns; ++run) {
                const int yr = y_first + run;
                if (yr >= h) break;
                const int xs = run == 0 ? x_first : x_first - w;
                const bool mine = masked && (run == 0 ? lane < w - x_first : lane >= w - x_first);
                float acc[3] = {0.f, 0.f, 0.f};
                // all window rows are requested before the first one is used: only a few hundred such waves exist
                // per frame, so nothing else hides the latency of 21 load round trips in a row
                uint2 pre[21][2];
#pragma unroll
                for (int dy = 0; dy < 21; ++dy) {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = lane + 64 * jj;
                        pre[dy][jj] = uint2{0u, 0u};
                        if (dy < ks && j < 64 + 2 * r)
                            pre[dy][jj] = load_px3_raw<T>(img, (size_t)r101_loop(yr + dy - r, h) * w +
                                                                   r101_loop(xs + j - r, w), n - 1);
                    }
                }
#pragma unroll
                for (int dy = 0; dy < 21; ++dy) {
                    if (dy < ks) {
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int j = lane + 64 * jj;
                            if (j < 64 + 2 * r) {
                                float p[3];
                                decode_px3<T>(pre[dy][jj], p);
                                s[j][0] = p[0]; s[j][1] = p[1]; s[j][2] = p[2];
                            }
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes have landed
                        float row[3] = {0.f, 0.f, 0.f};
                        // seven taps' LDS reads are in flight at a time (a loop that waits for every read is a chain
                        // of 1 323 LDS latencies per pixel)
                        for (int d0 = 0; d0 < ks; d0 += 7) {
                            float v[7][3];
#pragma unroll
                            for (int u = 0; u < 7; ++u) {
                                const int jx = lane + min(d0 + u, ks - 1);
                                v[u][0] = s[jx][0]; v[u][1] = s[jx][1]; v[u][2] = s[jx][2];
                            }
#pragma unroll
                            for (int u = 0; u < 7; ++u) {
                                if (d0 + u < ks) {
                                    const float k = kof(d0 + u);
#pragma unroll
                                    for (int c = 0; c < 3; ++c) {
                                        const float pr = k * v[u][c];
                                        row[c] = row[c] + pr;
                                    }
                                }
                            }
                        }
                        const float kd = kof(dy);
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float q = kd * row[c];
                            acc[c] = acc[c] + q;
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next row overwrites
                    }
                }
                if (mine) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) side[pi * 3 + c] = (T)min(max((int)rintf(acc[c]), 0), maxv);
                }
            }
            return;
        }
        if (masked) {
            const int y = (int)(pi / w), x = (int)(pi - (size_t)y * w);
            int o[3];
            blur_at<T>(img, h, w, y, x, g, o, kof);
            side[pi * 3 + 0] = (T)o[0]; side[pi * 3 + 1] = (T)o[1]; side[pi * 3 + 2] = (T)o[2];
        }
        return;
    }
    const int slot = lane / ks, wr = lane - slot * ks;
    unsigned long long rest = ballot;
    while (rest) {   // wave-uniform
        // the next `per` masked lanes: group `slot` takes the slot-th of them
        unsigned long long pick = rest;
        int src_lane = -1;
        for (int k = 0; k < per; ++k) {
            if (!pick) break;
            const int b = __ffsll((long long)pick) - 1;
            if (k == slot) src_lane = b;
            pick &= pick - 1;
        }
        rest = pick;
        const bool act = slot < per && src_lane >= 0;
        const size_t qi = base + (act ? src_lane : 0);
        float q[3] = {0.f, 0.f, 0.f};
        if (act) {
            const int y = (int)(qi / w), x = (int)(qi - (size_t)y * w);
            const int yy = r101_loop(y + wr - r, h);
            float row[3] = {0.f, 0.f, 0.f};
            for (int dx = 0; dx < ks; ++dx) {
                const int xx = r101_loop(x + dx - r, w);
                float p[3];
                load_px3<T>(img, (size_t)yy * w + xx, n - 1, p);
                const float k = kof(dx);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float pr = k * p[c];
                    row[c] = row[c] + pr;
                }
            }
            const float kw = __shfl(kreg, wr, 64);   // tap of this lane's window row
#pragma unroll
            for (int c = 0; c < 3; ++c) q[c] = kw * row[c];
        }
        // vertical pass: the group's first lane adds the row results in row order
        float acc[3] = {0.f, 0.f, 0.f};
        for (int dy = 0; dy < ks; ++dy) {
            const int from = min(slot * ks + dy, 63);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = __shfl(q[c], from, 64);
                acc[c] = acc[c] + v;
            }
        }
        if (act && wr == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) side[qi * 3 + c] = (T)min(max((int)rintf(acc[c]), 0), maxv);
        }
    }
}

// Pass 1 (scatter): masked pixels take their blurred value from `side`.
template <typename T>
__global__ __launch_bounds__(256) void border_blur_scatter(T* __restrict__ img, const uint8_t* __restrict__ valid,
                                                           const T* __restrict__ side, int h, int w) {
    const size_t n = (size_t)h * w;
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i0 >= n) return;
    uint32_t m[4];
    if (i0 + 16 <= n && (reinterpret_cast<uintptr_t>(valid) & 15) == 0) {
        const uint4 q = *reinterpret_cast<const uint4*>(valid + i0);
        m[0] = q.x; m[1] = q.y; m[2] = q.z; m[3] = q.w;
        if (m[0] == 0x01010101u && m[1] == 0x01010101u && m[2] == 0x01010101u && m[3] == 0x01010101u)
            return;  // all 16 valid: the common case
    } else {
        for (int k = 0; k < 4; ++k) {
            m[k] = 0;
            for (int b = 0; b < 4; ++b) {
                const size_t i = i0 + 4 * k + b;
                m[k] |= (uint32_t)(i < n ? valid[i] : 1) << (8 * b);
            }
        }
    }
    for (int k = 0; k < 16; ++k) {
        if ((m[k >> 2] >> (8 * (k & 3))) & 0xffu) continue;
        const size_t i = i0 + k;
        img[i * 3 + 0] = side[i * 3 + 0]; img[i * 3 + 1] = side[i * 3 + 1]; img[i * 3 + 2] = side[i * 3 + 2];
    }
}

}  // namespace mi
