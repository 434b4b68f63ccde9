// kernels_depthmap.hpp -- DepthMapStack (reference algorithms/depth_map.py:10-123; SURVEY.md 8(f)
// rank 4), frame at a time; F = float (float_type 'float-32') or double ('float-64': gray / energy planes and
// the image pyramids in float64; the bilateral filter still works on float32 copies and returns float32, so
// with smoothing the weights W are float32 in both modes, depth_map.py:46-51):
//
//   pass 1 (at push)   gray -> energy: |Sobel_x| + |Sobel_y| (:28-34) or |Laplacian(GaussianBlur)| (:36-41),
//                      running global maximum (:88)
//   finish             energies / max (:90), bilateral smoothing (:43-52), focus map: e / sum_i e (:55-57)
//                      or softmax((e - max_i e) / T) (:58-61);
//   pass 2             per frame: Gaussian pyramids of the frame and of its weight (pyrDown), Laplacian
//                      pyramid of the frame (pyrUp), weighted accumulation (:94-112); collapse, clip, cast
//                      (:117-123)
//
// Every kernel is one thread per output with the operation order of oracle/depth_map_oracle.py (the
// OpenCV primitives restated there: parity unpinned, stated in DESIGN.md).  With the default parameters
// the energies are exact whatever the order (integer gray levels, dyadic 5-tap Gaussian, integer
// derivative kernels, float64 accumulation as cv2's CV_64F output implies).
#pragma once
#include "common.hpp"
#include "kernels_balance.hpp"

namespace mi {

typedef float dm_v2f __attribute__((ext_vector_type(2)));

template <typename F>
struct DmTapsT {
    F k[32];  // symmetric Gaussian in the image's type (cv2.getGaussianKernel(ksize, 0, CV_32F / CV_64F))
    int ksize;
};
struct DmK2 {
    double k[15 * 15];  // 2-D Laplacian aperture (integers), row-major
    int ksize;
};

// v >= 0: the bit patterns order like the values
__device__ __forceinline__ void atomic_max_pos(float* addr, float v) {
    atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_min_pos(float* addr, float v) {
    atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_pos(double* addr, double v) {
    atomicMax(reinterpret_cast<unsigned long long*>(addr), (unsigned long long)__double_as_longlong(v));
}
__device__ __forceinline__ void atomic_min_pos(double* addr, double v) {
    atomicMin(reinterpret_cast<unsigned long long*>(addr), (unsigned long long)__double_as_longlong(v));
}
template <typename F>
__device__ __forceinline__ F wave_max(F v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const F u = __shfl_xor(v, o);
        v = u > v ? u : v;
    }
    return v;
}
template <typename F>
__device__ __forceinline__ F wave_min(F v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const F u = __shfl_xor(v, o);
        v = u < v ? u : v;
    }
    return v;
}

// img_bw (utils.py:46-47): integer BGR2GRAY, then np.array(..., dtype=float32) (:77)
template <typename T, typename F>
__global__ __launch_bounds__(256) void dm_gray(const T* __restrict__ img, size_t npix, F* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    out[i] = (F)bgr2gray_int(img[3 * i], img[3 * i + 1], img[3 * i + 2]);
}

// one axis of cv2.GaussianBlur: k[c]*S[0] + sum_j k[c+j]*(S[-j] + S[j]) in the image's type
template <bool ROWS, typename F>
__global__ __launch_bounds__(256) void dm_blur(const F* __restrict__ src, int h, int w, F* __restrict__ dst,
                                               DmTapsT<F> t) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int r = t.ksize / 2;
    auto at = [&](int o) {
        return ROWS ? src[(size_t)y * w + r101_loop(x + o, w)] : src[(size_t)r101_loop(y + o, h) * w + x];
    };
    F acc = t.k[r] * at(0);
    for (int j = 1; j <= r; ++j) {
        const F pr = t.k[r + j] * (at(-j) + at(j));
        acc = acc + pr;
    }
    dst[(size_t)y * w + x] = acc;
}

// workgroup maximum -> one atomic per workgroup, and none when the global value is already as large (the
// running maximum only grows, so a stale read can cost an extra atomic but never lose one)
template <typename F>
__device__ __forceinline__ void block_max_to(F v, F* gmax) {
    __shared__ F sm[16];
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) v = sm[i] > v ? sm[i] : v;
        if (v > __hip_atomic_load(gmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomic_max_pos(gmax, v);
    }
}
template <typename F>
__device__ __forceinline__ void block_min_to(F v, F* gmin) {
    __shared__ F sm[16];
    v = wave_min(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) v = sm[i] < v ? sm[i] : v;
        if (v < __hip_atomic_load(gmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomic_min_pos(gmin, v);
    }
}

// |cv2.Laplacian(blurred, CV_64F, ksize)| -> float32, and the running global maximum.  KS = the aperture as a
// compile-time constant (taps unrolled, kernel in registers) or 0 for any size; pixels whose window lies inside the
// image skip the reflection maps.
// KS > 0: a thread produces DM_LAP_ROWS vertically adjacent pixels from one (KS + ROWS - 1) x KS register patch -- every
// source value is loaded once per thread instead of once per output; each output is still the row-major chain over its
// own window, so the values do not change.  Launch over ceil(h / (4 * DM_LAP_ROWS)) block rows.
constexpr int DM_LAP_ROWS = 4;
template <int KS, typename F>
__global__ __launch_bounds__(256) void dm_laplacian_rows(const F* __restrict__ src, int h, int w,
                                                         F* __restrict__ out, F* __restrict__ gmax, DmK2 K) {
    static_assert(KS > 0, "compile-time aperture");
    constexpr int R = KS / 2, PH = KS + DM_LAP_ROWS - 1;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * DM_LAP_ROWS;
    F emax = 0;
    if (x < w && y0 < h) {
        const bool inside = x >= R && x + R < w && y0 >= R && y0 + DM_LAP_ROWS - 1 + R < h;
        F p[PH][KS];
#pragma unroll
        for (int i = 0; i < PH; ++i) {
            const F* row = src + (size_t)(inside ? y0 + i - R : r101_loop(y0 + i - R, h)) * w;
#pragma unroll
            for (int j = 0; j < KS; ++j) p[i][j] = row[inside ? x + j - R : r101_loop(x + j - R, w)];
        }
#pragma unroll
        for (int q = 0; q < DM_LAP_ROWS; ++q) {
            if (y0 + q >= h) break;
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < KS; ++i)
#pragma unroll
                for (int j = 0; j < KS; ++j) {
                    const double k = K.k[i * KS + j];
                    if (k == 0.0) continue;
                    const double pr = k * (double)p[q + i][j];
                    s = s + pr;
                }
            const F e = (F)fabs(s);
            out[(size_t)(y0 + q) * w + x] = e;
            emax = e > emax ? e : emax;
        }
    }
    block_max_to(emax, gmax);
}

// dm_gray -> dm_blur<rows> -> dm_blur<cols> -> dm_laplacian_rows<5> of one frame in ONE pass (float32 planes, 5-tap blur, 5 x 5
// aperture: the defaults): the four kernels read and wrote three float planes between them (330 us per 24 MP frame for
// one 72 MB read and one 96 MB write).  A workgroup owns a 64 x 32 tile: gray patch (halo 4) -> row blur (halo 2 in x) ->
// column blur (halo 2) -> Laplacian, through two LDS images.  The gray patch is filled with gray(r101(y), r101(x)) at EVERY
// position, also those outside the image; the blurs then run over the whole patch without any border logic and still
// produce, at a position outside the image, the value the separate kernels have at its mirror position inside: a blur's
// window around the mirrored position is the mirror image of the window around the position, its taps are symmetric and
// each pair (S[-j] + S[j]) is summed first -- the same products in the same order, float addition being commutative.
// Every in-image value therefore carries the bits of the separate kernels (GPU test), whose order it keeps:
// k[2] S[0] + k[3] (S[-1] + S[1]) + k[4] (S[-2] + S[2]), the Laplacian as the row-major float64 chain over its non-zero taps.
// per-frame (min, max) slots of the raw energies: (+inf, 0)
__global__ void dm_fmm_reset(float* __restrict__ fmm, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { fmm[2 * i] = INFINITY; fmm[2 * i + 1] = 0.f; }
}
struct DmK25 {
    double k[25];
};
template <typename T>
__global__ __launch_bounds__(256) void dm_energy_lap5(const T* __restrict__ img, int h, int w, float* __restrict__ out,
                                                      float* __restrict__ gmax, DmTapsT<float> taps, DmK25 K,
                                                      float* __restrict__ fmm) {
    constexpr int TW = 64, TH = 32;
    constexpr int GW = TW + 8, GH = TH + 8, GS = GW + 4;      // gray patch: x0-4 .., y0-4 ..; rows 16-byte aligned
    constexpr int RW = TW + 4, RS = RW + 1;                   // row blur: x0-2 .., same rows as the gray patch
    constexpr int CH = TH + 4;                                // column blur: y0-2 ..
    __shared__ __attribute__((aligned(16))) float sG[GH * GS];   // later: the column blur [CH][RS]
    __shared__ float sR[GH * RS];
    float* sC = sG;
    static_assert(CH * RS <= GH * GS, "the column blur aliases the gray patch");
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, tid = threadIdx.x;
    const float kc = taps.k[2], k1 = taps.k[3], k2 = taps.k[4];
    const bool interior = x0 >= 4 && y0 >= 4 && x0 + TW + 4 <= w && y0 + TH + 4 <= h;
    // ---- gray patch, four entries per thread and step
    for (int g = tid; g < GH * (GW / 4); g += 256) {
        const int r = g / (GW / 4), c = 4 * (g - r * (GW / 4));
        const int y = y0 - 4 + r, x = x0 - 4 + c;
        float o[4];
        if (interior) {
            T px[12];
            __builtin_memcpy(px, img + ((size_t)y * w + x) * 3, sizeof px);
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = (float)bgr2gray_int(px[3 * q], px[3 * q + 1], px[3 * q + 2]);
        } else {
            const T* row = img + (size_t)r101_loop(y, h) * w * 3;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const T* p = row + (size_t)r101_loop(x + q, w) * 3;
                o[q] = (float)bgr2gray_int(p[0], p[1], p[2]);
            }
        }
        *reinterpret_cast<float4*>(sG + r * GS + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
    __syncthreads();
    // ---- row blur: 40 rows x 17 groups of four columns (column c reads gray c .. c+4)
    for (int g = tid; g < GH * (RW / 4); g += 256) {
        const int r = g / (RW / 4), c = 4 * (g - r * (RW / 4));
        const float4 a = *reinterpret_cast<const float4*>(sG + r * GS + c);
        const float4 b = *reinterpret_cast<const float4*>(sG + r * GS + c + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float acc = kc * v[q + 2];
            const float p1 = k1 * (v[q + 1] + v[q + 3]);
            acc = acc + p1;
            const float p2 = k2 * (v[q] + v[q + 4]);
            acc = acc + p2;
            sR[r * RS + c + q] = acc;
        }
    }
    __syncthreads();
    // ---- column blur: 68 columns x 3 runs of 12 rows, a five-row window sliding down the column
    if (tid < 3 * RW) {
        const int run = tid / RW, c = tid - run * RW, r0 = 12 * run;
        const float* col = sR + r0 * RS + c;
        float v0 = col[0], v1 = col[RS], v2 = col[2 * RS], v3 = col[3 * RS];
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const float v4 = col[(j + 4) * RS];
            float acc = kc * v2;
            const float p1 = k1 * (v1 + v3);
            acc = acc + p1;
            const float p2 = k2 * (v0 + v4);
            acc = acc + p2;
            sC[(r0 + j) * RS + c] = acc;
            v0 = v1; v1 = v2; v2 = v3; v3 = v4;
        }
    }
    __syncthreads();
    // ---- Laplacian: a thread walks 8 rows of one column, the 5 x 5 window sliding down (output (yl, xl) reads column-blur
    // rows yl .. yl+4, columns xl .. xl+4)
    float emax = 0.f, emin = INFINITY;
    {
        const int xl = tid & 63, yb = 8 * (tid >> 6);
        const float* base = sC + yb * RS + xl;
        float win[5][5];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) win[i + 1][j] = base[i * RS + j];
        const int x = x0 + xl;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) win[i][j] = win[i + 1][j];
#pragma unroll
            for (int j = 0; j < 5; ++j) win[4][j] = base[(q + 4) * RS + j];
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const double k = K.k[i * 5 + j];
                    if (k == 0.0) continue;
                    const double pr = k * (double)win[i][j];
                    s = s + pr;
                }
            const int y = y0 + yb + q;
            if (x < w && y < h) {
                const float e = (float)fabs(s);
                out[(size_t)y * w + x] = e;
                emax = e > emax ? e : emax;
                emin = e < emin ? e : emin;
            }
        }
    }
    block_max_to(emax, gmax);
    // the frame's own minimum / maximum (fmm[0] preset to +inf, fmm[1] to 0: dm_fmm_reset): what dm_normalise used to find in a
    // pass of its own -- min / max of e / m are min / max of e, divided (a division by m > 0 is monotone under rounding)
    if (fmm) {
        __syncthreads();
        block_max_to(emax, fmm + 1);
        __syncthreads();
        block_min_to(emin, fmm);
    }
}

template <int KS, typename F>
__global__ __launch_bounds__(256) void dm_laplacian(const F* __restrict__ src, int h, int w,
                                                    F* __restrict__ out, F* __restrict__ gmax, DmK2 K) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    F e = 0;
    if (x < w && y < h) {
        const int ks = KS ? KS : K.ksize, r = ks / 2;
        const bool inside = x >= r && y >= r && x + r < w && y + r < h;
        double s = 0.0;
        for (int i = 0; i < ks; ++i) {
            const F* row = src + (size_t)(inside ? y + i - r : r101_loop(y + i - r, h)) * w;
            for (int j = 0; j < ks; ++j) {
                const double k = K.k[i * ks + j];
                if (k == 0.0) continue;
                const double pr = k * (double)row[inside ? x + j - r : r101_loop(x + j - r, w)];
                s = s + pr;
            }
        }
        e = (F)fabs(s);
        out[(size_t)y * w + x] = e;
    }
    block_max_to(e, gmax);
}

// |Sobel_x| + |Sobel_y| (3x3, CV_64F) -> the energy plane's type
template <typename F>
__global__ __launch_bounds__(256) void dm_sobel(const F* __restrict__ src, int h, int w, F* __restrict__ out,
                                                F* __restrict__ gmax) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    F e = 0;
    if (x < w && y < h) {
        double p[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const F* row = src + (size_t)r101_loop(y + i - 1, h) * w;
#pragma unroll
            for (int j = 0; j < 3; ++j) p[i][j] = (double)row[r101_loop(x + j - 1, w)];
        }
        // row-major over the non-zero taps of outer([1 2 1], [-1 0 1]) and outer([-1 0 1], [1 2 1])
        double gx = 0.0 + -1.0 * p[0][0];
        gx = gx + 1.0 * p[0][2];
        gx = gx + -2.0 * p[1][0];
        gx = gx + 2.0 * p[1][2];
        gx = gx + -1.0 * p[2][0];
        gx = gx + 1.0 * p[2][2];
        double gy = 0.0 + -1.0 * p[0][0];
        gy = gy + -2.0 * p[0][1];
        gy = gy + -1.0 * p[0][2];
        gy = gy + 1.0 * p[2][0];
        gy = gy + 2.0 * p[2][1];
        gy = gy + 1.0 * p[2][2];
        e = (F)(fabs(gx) + fabs(gy));
        out[(size_t)y * w + x] = e;
    }
    block_max_to(e, gmax);
}

// energies / max_energy (:90, a division in the plane's type) in place, and the plane's own min / max (the
// bilateral filter scales its range table with them).  mm[0] = min, mm[1] = max, preset to +inf bits / 0.
// A zero *gmax leaves the values as they are (used to take min / max only).
template <typename F>
__global__ __launch_bounds__(256) void dm_normalise(F* __restrict__ e, size_t n, const F* __restrict__ gmax,
                                                    F* __restrict__ mm) {
    const F m = *gmax;
    F hi = 0, lo = INFINITY;
    // four elements per step (one 16 / 32-byte access each way, four independent divisions), then the tail
    typedef F V4 __attribute__((ext_vector_type(4)));
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        V4 v = reinterpret_cast<const V4*>(e)[i];
        if (m > 0) {
            v = V4{v.x / m, v.y / m, v.z / m, v.w / m};
            reinterpret_cast<V4*>(e)[i] = v;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            hi = v[k] > hi ? v[k] : hi;
            lo = v[k] < lo ? v[k] : lo;
        }
    }
    for (size_t i = 4 * n4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        F v = e[i];
        if (m > 0) {
            v = v / m;
            e[i] = v;
        }
        hi = v > hi ? v : hi;
        lo = v < lo ? v : lo;
    }
    block_max_to(hi, mm + 1);
    block_min_to(lo, mm);
}

// energy_map[i].astype(np.float32) (:48)
__global__ __launch_bounds__(256) void dm_to_f32(const double* __restrict__ src, size_t n, float* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}

constexpr int DM_LUT_BINS = 4096;
// dm_bilateral<NP>: NP pixels of a thread go through the taps together (rows tq, tq + 4, ...: independent chains that hide
// the latency of the patch read and of the range-table gather); the tile is 4 NP rows of 64 pixels.  LDS of a launch:
inline size_t dm_bilateral_lds(int np, int radius) {
    return sizeof(float) * (2 * (DM_LUT_BINS + 1) + (size_t)(4 * np + 2 * radius) * (64 + 2 * radius));
}

// expLUT of cv2.bilateralFilter (float32 images).  bp[0] = scale_index, bp[1] = 1 if the image is constant.
__global__ __launch_bounds__(1024) void dm_bilateral_lut(const float* __restrict__ mm, double color_coeff,
                                                         float* __restrict__ lut, float* __restrict__ bp,
                                                         const float* __restrict__ norm = nullptr) {
    // norm: the global maximum the plane has NOT been divided by yet (dm_bilateral<.., NORM> divides as it stages):
    // mm then holds the raw plane's min / max, and the normalised plane's are their quotients
    float lo = mm[0], hi = mm[1];
    if (norm && *norm > 0) { lo = lo / *norm; hi = hi / *norm; }
    const bool flat = fabs((double)lo - (double)hi) < (double)1.1920928955078125e-07f;
    const float len = (float)((double)hi - (double)lo);
    const float scale_index = (float)DM_LUT_BINS / len;
    if (threadIdx.x == 0) {
        bp[0] = scale_index;
        bp[1] = flat ? 1.f : 0.f;
    }
    if (flat) return;
    auto entry = [&](int i) {
        const double val = (double)((float)i / scale_index);
        return (float)exp(val * val * color_coeff);
    };
    for (int i = threadIdx.x; i < DM_LUT_BINS + 2; i += 1024)
        lut[i] = (i == 0 || entry(i - 1) > 0.f) ? entry(i) : 0.f;   // the table stops at its first zero
}

struct DmBilateral {
    const float* src;
    float* dst;
    int h, w, radius, ntaps;
    const int2* taps;     // ntaps x (dy * patch_width + dx, bits of the space weight), raster order of the disc
    const float* lut;
    const float* bp;
    float* acc;           // AVERAGE: running sum of the smoothed energies; MAX: running maximum
    int mode;             // 0 = sum, 1 = max
    int first;
    const float* norm;    // not null: src is divided by *norm (if > 0) as it is staged (the dm_normalise pass folded in)
};

// cv2.bilateralFilter(e, d, 25, 25) on a float32 plane; the smoothed plane also goes into the running
// sum / maximum over frames (np.sum / np.max over axis 0 add the planes in frame order).
// 16 x 64 pixels per workgroup; the patch (radius <= 15) and the range table live in LDS.
// RC: the radius as a compile-time constant (the patch row length then is one too, and the rows of a thread's pixels become
// immediate offsets of its LDS reads instead of an address addition per pixel and tap), or 0 for any radius
template <int NP, int RC>
__global__ __launch_bounds__(256) void dm_bilateral(DmBilateral a) {
    constexpr int TH = 4 * NP, TW = 64;   // NP pixels (rows tq + 4 k) per thread
    // dynamic LDS (dm_bilateral_lds): the range table as (lut[i], lut[i+1] - lut[i]) pairs, then the (TH + 2 r) x (TW + 2 r) patch
    extern __shared__ __attribute__((aligned(16))) float dm_bil_lds[];
    float* sL = dm_bil_lds;
    float* sP = dm_bil_lds + 2 * (DM_LUT_BINS + 1);
    const int t = threadIdx.x, r = RC > 0 ? RC : a.radius;
    const int y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
    const bool flat = a.bp[1] != 0.f;
    const float nm = a.norm ? *a.norm : 0.f;
    const int pw = TW + 2 * r, ph = TH + 2 * r;
    if (!flat) {
        for (int i = t; i < DM_LUT_BINS + 1; i += 256) {
            const float l0 = a.lut[i], l1 = a.lut[i + 1];
            sL[2 * i] = l0;
            sL[2 * i + 1] = l1 - l0;
        }
        for (int i = t; i < ph * pw; i += 256) {
            const int py = i / pw, px = i - py * pw;
            float v = a.src[(size_t)r101_loop(y0 + py - r, a.h) * a.w + r101_loop(x0 + px - r, a.w)];
            if (nm > 0) v = v / nm;
            sP[i] = v;
        }
    }
    __syncthreads();
    const float scale_index = a.bp[0];
    const int tx = t & 63, tq = t >> 6;
    // the thread's four pixels (rows tq, tq+4, tq+8, tq+12 of column tx) go through the taps together: four
    // independent chains hide the LDS latency of the patch read and of the range-table gather
    const float* c = sP + (tq + r) * pw + (tx + r);
    float v0[NP], sum[NP], wsum[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) { v0[k] = c[4 * k * pw]; sum[k] = 0.f; wsum[k] = 0.f; }
    if (!flat) {
#pragma unroll 2
        for (int n = 0; n < a.ntaps; ++n) {
            const int2 tap = a.taps[n];                // (dy * pw + dx, bits of the space weight): a scalar load
            const float swn = __int_as_float(tap.y);
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const float val = c[4 * k * pw + tap.x];
                const float alpha = fabsf(val - v0[k]) * scale_index;   // 0 <= alpha <= DM_LUT_BINS
                const int idx = (int)alpha;                               // floor of a non-negative value
                const float fr = __builtin_amdgcn_fractf(alpha);          // alpha - floor(alpha), exact
                const dm_v2f l = *reinterpret_cast<const dm_v2f*>(sL + 2 * idx);
                const float ad = fr * l.y;
                const float cw = l.x + ad;
                const float wk = swn * cw;
                const float vw = val * wk;
                sum[k] = sum[k] + vw;
                wsum[k] = wsum[k] + wk;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int y = y0 + tq + 4 * k, x = x0 + tx;
        if (y >= a.h || x >= a.w) continue;
        const size_t pi = (size_t)y * a.w + x;
        float res;
        if (flat) {
            res = a.src[pi];
            if (nm > 0) res = res / nm;
        } else res = sum[k] / wsum[k];
        a.dst[pi] = res;
        if (a.mode == 0) a.acc[pi] = a.first ? 0.f + res : a.acc[pi] + res;
        else a.acc[pi] = a.first ? res : fmaxf(a.acc[pi], res);
    }
}

// smooth_size <= 0: no smoothing, only the running sum / maximum (W = the energy planes' type)
template <typename W>
__global__ __launch_bounds__(256) void dm_accumulate(const W* __restrict__ e, size_t n, W* __restrict__ acc,
                                                     int mode, int first) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const W v = e[i];
    if (mode == 0) acc[i] = first ? (W)0 + v : acc[i] + v;
    else acc[i] = first ? v : (acc[i] > v ? acc[i] : v);
}

// MAX map (:59-60): relative = exp((e - max_e) / T) in place, running sum of the relatives.  float32: the
// correctly rounded exp (through the double one); float64: the device library's exp (the oracle's is correctly
// rounded -- the last bit may differ, which the parity tolerance covers)
template <typename W>
__global__ __launch_bounds__(256) void dm_relative(W* __restrict__ e, const W* __restrict__ mx, size_t n,
                                                   W temperature, W* __restrict__ tot, int first) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const W d = e[i] - mx[i];
    const W q = d / temperature;
    const W rel = (W)exp((double)q);
    e[i] = rel;
    tot[i] = first ? (W)0 + rel : tot[i] + rel;
}

// weights (:57, :61): e / total; the AVERAGE map leaves pixels whose total is 0 undefined in the
// reference (np.divide(..., where=) without out=) -- 0 here
template <typename W>
__global__ __launch_bounds__(256) void dm_weight(const W* __restrict__ e, const W* __restrict__ tot, size_t n,
                                                 int guard_zero, W* __restrict__ wgt) {
    // four elements per thread (the launch still covers n threads: the upper three quarters leave at once)
    typedef W V4 __attribute__((ext_vector_type(4)));
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, n4 = n / 4;
    if (i < n4) {
        const V4 t = reinterpret_cast<const V4*>(tot)[i], ev = reinterpret_cast<const V4*>(e)[i];
        V4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (guard_zero && t[k] == 0) ? (W)0 : ev[k] / t[k];
        reinterpret_cast<V4*>(wgt)[i] = o;
    } else if (i - n4 < n - 4 * n4) {
        const size_t j = 4 * n4 + (i - n4);
        const W t = tot[j];
        wgt[j] = (guard_zero && t == 0) ? (W)0 : e[j] / t;
    }
}

// cv2.pyrDown with C interleaved channels, arithmetic in F: rows s[2x]*6 + (s[2x-1] + s[2x+1])*4 + s[2x-2] +
// s[2x+2], the same down the columns, * 1/256.  TSrc = the frame's integer type for level 0.
template <typename TSrc, int C, typename F>
__global__ __launch_bounds__(256) void dm_pyrdown(const TSrc* __restrict__ src, int h, int w, F* __restrict__ dst,
                                                  int ho, int wo) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= wo || y >= ho) return;
    int xs[5], ys[5];
#pragma unroll
    for (int o = 0; o < 5; ++o) {
        xs[o] = r101_loop(2 * x + o - 2, w);
        ys[o] = r101_loop(2 * y + o - 2, h);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        F rowv[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const TSrc* row = src + (size_t)ys[i] * w * C + c;
            const F m2 = (F)row[xs[0] * C], m1 = (F)row[xs[1] * C], c0 = (F)row[xs[2] * C],
                    p1 = (F)row[xs[3] * C], p2 = (F)row[xs[4] * C];
            F s = c0 * (F)6;
            const F pr = (m1 + p1) * (F)4;
            s = s + pr;
            s = s + m2;
            rowv[i] = s + p2;
        }
        F s = rowv[2] * (F)6;
        const F pr = (rowv[1] + rowv[3]) * (F)4;
        s = s + pr;
        s = s + rowv[0];
        s = s + rowv[4];
        dst[((size_t)y * wo + x) * C + c] = s * (F)(1.0 / 256.0);
    }
}

// dm_pyrdown through LDS: a workgroup stages the source patch of a 64 x TH output tile (reflected border), runs the row
// formula into a second LDS image and the column formula from there -- the same operations per output, every source value
// read from HBM once instead of up to 25 times.  TH shrinks with the element size so that both images fit 64 KB of LDS;
// 0 = does not fit (3-channel double): the caller keeps dm_pyrdown.
template <typename TSrc, int C, typename F>
constexpr int dm_pyrdown_tile_rows() {
    return sizeof(TSrc) * C <= 6 && sizeof(F) * C <= 12 ? 16 : (sizeof(TSrc) * C <= 12 && sizeof(F) * C <= 12 ? 8 : (sizeof(F) * C <= 8 ? 8 : 0));
}
template <typename TSrc, int C, typename F, int TH>
__global__ __launch_bounds__(256) void dm_pyrdown_tile(const TSrc* __restrict__ src, int h, int w, F* __restrict__ dst,
                                                       int ho, int wo) {
    constexpr int TW = 64, IW = 2 * (TW - 1) + 5, IH = 2 * (TH - 1) + 5;
    constexpr int ND = (IW * C * (int)sizeof(TSrc) + 3) / 4;   // dwords per staged row (padded: rows start on a dword)
    __shared__ uint32_t s_in_dw[IH * ND];
    __shared__ F s_row[IH * TW * C];
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, tid = threadIdx.x;
    // Interior tiles (the patch, plus the padding bytes of its rows, inside the image): the patch rows are contiguous in
    // memory -- copied a dword at a time (unaligned global loads) instead of element by element through two reflection
    // loops and an index division each, which was more than half of this kernel's instructions.
    const bool interior = 2 * x0 - 2 >= 0 && 2 * y0 - 2 >= 0 && 2 * x0 - 2 + IW + 2 <= w && 2 * y0 - 2 + IH <= h;
    if (interior) {
        const char* base = (const char*)(src + ((size_t)(2 * y0 - 2) * w + (2 * x0 - 2)) * C);
        const size_t pitch = (size_t)w * C * sizeof(TSrc);
        for (int i = tid; i < IH * ND; i += 256) {
            const int r = i / ND, q = i - r * ND;
            uint32_t v;
            __builtin_memcpy(&v, base + (size_t)r * pitch + 4 * q, 4);
            s_in_dw[i] = v;
        }
    } else {
        for (int i = tid; i < IH * IW; i += 256) {
            const int r = i / IW, q = i - r * IW;
            const TSrc* px = src + ((size_t)r101_loop(2 * y0 - 2 + r, h) * w + r101_loop(2 * x0 - 2 + q, w)) * C;
            TSrc* d = reinterpret_cast<TSrc*>(s_in_dw + r * ND) + q * C;
#pragma unroll
            for (int c = 0; c < C; ++c) d[c] = px[c];
        }
    }
    __syncthreads();
    for (int i = tid; i < IH * TW; i += 256) {
        const int r = i / TW, x = i - r * TW;
        const TSrc* p = reinterpret_cast<const TSrc*>(s_in_dw + r * ND) + 2 * x * C;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const F m2 = (F)p[c], m1 = (F)p[C + c], c0 = (F)p[2 * C + c], p1 = (F)p[3 * C + c], p2 = (F)p[4 * C + c];
            F s = c0 * (F)6;
            const F pr = (m1 + p1) * (F)4;
            s = s + pr;
            s = s + m2;
            s_row[i * C + c] = s + p2;
        }
    }
    __syncthreads();
    for (int i = tid; i < TH * TW; i += 256) {
        const int yl = i / TW, x = i - yl * TW;
        if (y0 + yl >= ho || x0 + x >= wo) continue;
        const F* p = s_row + ((2 * yl) * TW + x) * C;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            F s = p[2 * TW * C + c] * (F)6;
            const F pr = (p[TW * C + c] + p[3 * TW * C + c]) * (F)4;
            s = s + pr;
            s = s + p[c];
            s = s + p[4 * TW * C + c];
            dst[((size_t)(y0 + yl) * wo + x0 + x) * C + c] = s * (F)(1.0 / 256.0);
        }
    }
}

// one axis of cv2.pyrUp, unnormalised: sample i of a destination of nd samples from n source samples
template <typename F, typename A>
__device__ __forceinline__ F up_axis(int n, int nd, int i, A at) {
    if (i >= 2 * n) i = 2 * n - 1;  // an odd destination repeats its last sample
    const int s = i >> 1;
    if (i & 1) {
        if (s == n - 1) return at(s) * (F)8;
        return (at(s) + at(s + 1)) * (F)4;
    }
    if (n == 1) return at(0) * (F)8;
    if (s == 0) {
        const F a = at(0) * (F)6, b = at(1) * (F)2;
        return a + b;
    }
    if (s == n - 1) {
        const F b = at(s) * (F)7;
        return at(s - 1) + b;
    }
    const F m = at(s) * (F)6;
    const F l = at(s - 1) + m;
    return l + at(s + 1);
}

// cv2.pyrUp(src, dstsize=(wd, hd)) at destination (y, x), channel c: columns first, then rows, * 1/64
template <int C, typename F>
__device__ __forceinline__ F pyrup_at(const F* __restrict__ src, int hs, int ws, int hd, int wd, int y, int x, int c) {
    const F v = up_axis<F>(hs, hd, y, [&](int r) {
        const F* row = src + (size_t)r * ws * C + c;
        return up_axis<F>(ws, wd, x, [&](int q) { return row[q * C]; });
    });
    return v * (F)(1.0 / 64.0);
}

// pyrUp for the 2 x 2 destination quad (2i .. 2i+1, 2j .. 2j+1) of a 3-channel image: the 3 x 3 source patch around (i, j)
// is loaded once (rows / columns clamped into the source: the clamped entries are exactly the ones the edge rules of
// up_axis never read), the column pass runs once per patch row and destination column, the row pass per destination pixel --
// the same operations pyrup_at performs per pixel.  up[q][c]: q = 2 * (row parity) + (column parity).
template <typename F>
__device__ __forceinline__ F pick3(F a0, F a1, F a2, int k) { return k == 0 ? a0 : (k == 1 ? a1 : a2); }   // no indexed registers

template <typename F>
__device__ __forceinline__ void pyrup_quad3(const F* __restrict__ src, int hs, int ws, int hd, int wd, int i, int j,
                                            F up[4][3]) {
    F patch[3][3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int r = min(max(i - 1 + a, 0), hs - 1);
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const F* px = src + ((size_t)r * ws + min(max(j - 1 + b, 0), ws - 1)) * 3;
            patch[a][b][0] = px[0]; patch[a][b][1] = px[1]; patch[a][b][2] = px[2];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        F hx[3][2];   // column pass of patch row a at destination columns 2j, 2j+1
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                hx[a][e] = up_axis<F>(ws, wd, 2 * j + e, [&](int q) { return pick3(patch[a][0][c], patch[a][1][c], patch[a][2][c], q - j + 1); });
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const F v = up_axis<F>(hs, hd, 2 * i + d, [&](int r) { return pick3(hx[0][e], hx[1][e], hx[2][e], r - i + 1); });
                up[2 * d + e][c] = v * (F)(1.0 / 64.0);
            }
    }
}

// Laplacian level (fine - pyrUp(coarse)) times the weight plane of that level, accumulated over frames (:104-110).
// One lane per 2 x 2 quad (launch over ceil(w / 2) x ceil(h / 2)).
template <typename TFine, typename F, typename W>
__global__ __launch_bounds__(256) void dm_lap_blend_quad(const TFine* __restrict__ fine, int h, int w,
                                                         const F* __restrict__ coarse, int hc, int wc,
                                                         const W* __restrict__ wgt, F* __restrict__ blend, int first) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (2 * j >= w || 2 * i >= h) return;
    F up[4][3];
    pyrup_quad3<F>(coarse, hc, wc, h, w, i, j, up);
    if (2 * j + 1 < w) {
        // both pixels of a quad row lie side by side: the pair's six values, its two weights and its six sums as one
        // wide access each instead of one per element (the lanes of a wave then cover a row contiguously)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int y = 2 * i + d;
            if (y >= h) continue;
            const size_t p = (size_t)y * w + 2 * j;
            TFine fv[6];
            W wv[2];
            F bv[6];
            __builtin_memcpy(fv, fine + p * 3, sizeof fv);
            __builtin_memcpy(wv, wgt + p, sizeof wv);
            if (!first) __builtin_memcpy(bv, blend + p * 3, sizeof bv);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const F lap = (F)fv[3 * e + c] - up[2 * d + e][c];
                    const F cur = lap * (F)wv[e];
                    bv[3 * e + c] = first ? cur : bv[3 * e + c] + cur;
                }
            __builtin_memcpy(blend + p * 3, bv, sizeof bv);
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int y = 2 * i + (q >> 1), x = 2 * j + (q & 1);
        if (y >= h || x >= w) continue;
        const size_t p = (size_t)y * w + x;
        const F wv = (F)wgt[p];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const F lap = (F)fine[p * 3 + c] - up[q][c];
            const F cur = lap * wv;
            blend[p * 3 + c] = first ? cur : blend[p * 3 + c] + cur;
        }
    }
}

template <typename F>
__global__ __launch_bounds__(256) void dm_collapse_quad(const F* __restrict__ coarse, int hc, int wc,
                                                        const F* __restrict__ blend, int h, int w, F* __restrict__ out) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (2 * j >= w || 2 * i >= h) return;
    F up[4][3];
    pyrup_quad3<F>(coarse, hc, wc, h, w, i, j, up);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int y = 2 * i + (q >> 1), x = 2 * j + (q & 1);
        if (y >= h || x >= w) continue;
        const size_t p = (size_t)y * w + x;
#pragma unroll
        for (int c = 0; c < 3; ++c) out[p * 3 + c] = up[q][c] + blend[p * 3 + c];
    }
}

template <typename TFine, typename F, typename W>
__global__ __launch_bounds__(256) void dm_lap_blend(const TFine* __restrict__ fine, int h, int w,
                                                    const F* __restrict__ coarse, int hc, int wc,
                                                    const W* __restrict__ wgt, F* __restrict__ blend, int first) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const size_t p = (size_t)y * w + x;
    const F wv = (F)wgt[p];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const F lap = (F)fine[p * 3 + c] - pyrup_at<3, F>(coarse, hc, wc, h, w, y, x, c);
        const F cur = lap * wv;
        blend[p * 3 + c] = first ? cur : blend[p * 3 + c] + cur;
    }
}

// coarsest level: the Gaussian level itself times its weight plane
template <typename TFine, typename F, typename W>
__global__ __launch_bounds__(256) void dm_top_blend(const TFine* __restrict__ top, size_t npix,
                                                    const W* __restrict__ wgt, F* __restrict__ blend, int first) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    const F wv = (F)wgt[p];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const F cur = (F)top[p * 3 + c] * wv;
        blend[p * 3 + c] = first ? cur : blend[p * 3 + c] + cur;
    }
}

// result = pyrUp(result) + blended level (:119-121)
template <typename F>
__global__ __launch_bounds__(256) void dm_collapse(const F* __restrict__ coarse, int hc, int wc,
                                                   const F* __restrict__ blend, int h, int w, F* __restrict__ out) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const size_t p = (size_t)y * w + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[p * 3 + c] = pyrup_at<3, F>(coarse, hc, wc, h, w, y, x, c) + blend[p * 3 + c];
}

// np.clip(np.absolute(result), 0, n_values).astype(dtype) (:122-123)
template <typename TOut, typename F>
__global__ __launch_bounds__(256) void dm_finalize(const F* __restrict__ img, size_t n, F maxv,
                                                   TOut* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    F v = img[i] < 0 ? -img[i] : img[i];
    v = v < 0 ? (F)0 : (v > maxv ? maxv : v);
    out[i] = (TOut)v;
}

}  // namespace mi
