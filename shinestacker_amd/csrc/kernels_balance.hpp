// kernels_balance.hpp -- the two data-parallel steps of BalanceFrames (reference
// algorithms/balance.py): the histogram of a (sub-sampled, optionally
// circular-masked) frame -- balance.py:158-180 calc_hist_1ch, per BGR channel (RGBCorrection
// :264-266) or of the luminance (LumiCorrection :235-236) -- and the look-up-table apply
// (cv2.LUT / np.take, balance.py:30-32, per channel :34-50).  The LUT itself (LINEAR / GAMMA /
// MATCH_HIST: bisect, interp1d on 256 or 65536 entries) stays on the host in NumPy/SciPy, exactly
// the reference's calls (shinestacker_amd/balance.py).
//
// Integer work: bit-exact by construction.  Two OpenCV primitives are restated from memory (OpenCV
// is not in this image -- parity unpinned for them, stated in DESIGN.md):
//   * cv2.cvtColor(BGR2GRAY) on 8/16-bit: (B*1868 + G*9617 + R*4899 + 2^13) >> 14;
//   * cv2.resize(INTER_AREA) by an integer factor s: mean of the s x s block, (sum+2)>>2 for s = 2,
//     otherwise round-half-even of sum * (1.f / s^2).
#pragma once
#include "common.hpp"

namespace mi {

struct HistArgs {
    const void* img;   // H x W x 3, uint8 / uint16
    int h, w;          // full-resolution size
    int hs, ws;        // size of the sub-sampled image the histogram is taken of
    int s;             // sub-sampling factor (1 = none)
    int fast;          // img[::s, ::s] instead of the area mean
    int gray;          // luminance histogram (1 channel) instead of B, G, R (3 channels)
    int masked;        // keep (x - ws/2)^2 + (y - hs/2)^2 <= r2 only (balance.py:165-175)
    double cx, cy, r2;
    uint32_t* counts;  // [nch][nbins], zeroed by the caller
};

__device__ __forceinline__ uint32_t bgr2gray_int(uint32_t b, uint32_t g, uint32_t r) {
    return (b * 1868u + g * 9617u + r * 4899u + (1u << 13)) >> 14;
}

// value of channel c (or of the luminance) of sub-sampled pixel (sy, sx)
template <typename T>
__device__ __forceinline__ void sub_pixel(const HistArgs& a, int sy, int sx, uint32_t out[3]) {
    const T* img = (const T*)a.img;
    const int nch = a.gray ? 1 : 3;
    if (a.s == 1 || a.fast) {
        const T* p = img + ((size_t)sy * a.s * a.w + (size_t)sx * a.s) * 3;
        if (a.gray) out[0] = bgr2gray_int(p[0], p[1], p[2]);
        else { out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; }
        return;
    }
    // cv2.resize(INTER_AREA) by the integer factor [from memory, as align.img_subsample restates it]: a last row / column
    // of blocks may hang over the image edge (the output size is round-half-even(dim / s)) and averages the pixels it has
    const int ny = min(a.s, a.h - sy * a.s), nx = min(a.s, a.w - sx * a.s);
    uint32_t sum[3] = {0, 0, 0};
    for (int dy = 0; dy < ny; ++dy) {
        const T* row = img + ((size_t)(sy * a.s + dy) * a.w + (size_t)sx * a.s) * 3;
        for (int dx = 0; dx < nx; ++dx) {
            const T* p = row + dx * 3;
            if (a.gray) sum[0] += bgr2gray_int(p[0], p[1], p[2]);
            else { sum[0] += p[0]; sum[1] += p[1]; sum[2] += p[2]; }
        }
    }
    const float scale = 1.0f / (float)(a.s * a.s);
    for (int c = 0; c < nch; ++c) {
        if (ny * nx == a.s * a.s) out[c] = a.s == 2 ? (sum[c] + 2u) >> 2 : (uint32_t)__float2int_rn((float)sum[c] * scale);
        else out[c] = (uint32_t)__float2int_rn((float)sum[c] / (float)(ny * nx));
    }
}

// 8-bit: per-workgroup histogram in LDS, one flush per bin and workgroup
__global__ __launch_bounds__(256) void hist_u8(HistArgs a) {
    __shared__ uint32_t sh[3 * 256];
    for (int i = threadIdx.x; i < 3 * 256; i += blockDim.x) sh[i] = 0;
    __syncthreads();
    const size_t total = (size_t)a.hs * a.ws;
    const int nch = a.gray ? 1 : 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int sy = (int)(i / a.ws), sx = (int)(i - (size_t)sy * a.ws);
        if (a.masked) {
            const double dx = (double)sx - a.cx, dy = (double)sy - a.cy;
            if (!(dx * dx + dy * dy <= a.r2)) continue;
        }
        uint32_t v[3];
        sub_pixel<uint8_t>(a, sy, sx, v);
        for (int c = 0; c < nch; ++c) atomicAdd(&sh[c * 256 + (v[c] & 255u)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nch * 256; i += blockDim.x)
        if (sh[i]) atomicAdd(&a.counts[i], sh[i]);
}

// 16-bit: 65536 bins per channel do not fit LDS; atomics on the L2-resident table
__global__ __launch_bounds__(256) void hist_u16(HistArgs a) {
    const size_t total = (size_t)a.hs * a.ws;
    const int nch = a.gray ? 1 : 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int sy = (int)(i / a.ws), sx = (int)(i - (size_t)sy * a.ws);
        if (a.masked) {
            const double dx = (double)sx - a.cx, dy = (double)sy - a.cy;
            if (!(dx * dx + dy * dy <= a.r2)) continue;
        }
        uint32_t v[3];
        sub_pixel<uint16_t>(a, sy, sx, v);
        for (int c = 0; c < nch; ++c) atomicAdd(&a.counts[(size_t)c * 65536 + (v[c] & 65535u)], 1u);
    }
}

// dst[p][c] = lut[nlut == 1 ? 0 : c][src[p][c]];  n = number of pixel*channel elements
__global__ __launch_bounds__(256) void lut_apply_u8(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t n,
                                                    const uint8_t* __restrict__ lut, int nlut) {
    __shared__ uint8_t sl[3 * 256];
    for (int i = threadIdx.x; i < 3 * 256; i += blockDim.x) sl[i] = lut[nlut == 1 ? (i & 255) : i];
    __syncthreads();
    // 12 bytes (4 pixels) per thread and step: three dword loads, channel of byte k is k % 3
    const size_t nq = n / 12;
    const uint32_t* s4 = (const uint32_t*)src;
    uint32_t* d4 = (uint32_t*)dst;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (size_t)gridDim.x * blockDim.x) {
        uint32_t in[3] = {s4[3 * q], s4[3 * q + 1], s4[3 * q + 2]}, out[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            uint32_t o = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int c = (4 * d + b) % 3;
                o |= (uint32_t)sl[c * 256 + ((in[d] >> (8 * b)) & 255u)] << (8 * b);
            }
            out[d] = o;
        }
        d4[3 * q] = out[0]; d4[3 * q + 1] = out[1]; d4[3 * q + 2] = out[2];
    }
    // tail (n is a multiple of 3; fewer than 12 elements left)
    if (blockIdx.x == 0)
        for (size_t i = nq * 12 + threadIdx.x; i < n; i += blockDim.x) dst[i] = sl[(i % 3) * 256 + src[i]];
}

__global__ __launch_bounds__(256) void lut_apply_u16(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, size_t n,
                                                     const uint16_t* __restrict__ lut, int nlut) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = nlut == 1 ? 0 : (int)(i % 3);
        dst[i] = lut[(size_t)c * 65536 + src[i]];
    }
}

// ---------------------------------------------------------------- LINEAR correction tables built on the device
// LinearMap (balance.py:87-105) without the host round trip: mean = sum(i * h[i]) / sum(h[i]) over [lo, hi) -- exact
// integers, one float64 division, as np.average does --, ratio = reference mean / mean, table[i] = trunc(clip(i * ratio, 0,
// vmax)) in float64.  One workgroup per table; tables below `first_channel` are the identity (the hue channel of HSV / HLS).
// An empty histogram (np.average raises there) gives ratio 1.
struct LinRef { double mean[3]; };

template <typename T>
__global__ __launch_bounds__(256) void lut_linear_build(const uint32_t* __restrict__ counts, int nbins, int lo, int hi,
                                                        int first_channel, LinRef ref, T* __restrict__ lut,
                                                        double* __restrict__ corr_out) {
    const int t = blockIdx.x, tid = threadIdx.x;
    T* out = lut + (size_t)t * nbins;
    if (t < first_channel) {
        for (int i = tid; i < nbins; i += 256) out[i] = (T)i;
        return;
    }
    const uint32_t* h = counts + (size_t)t * nbins;
    unsigned long long s0 = 0, s1 = 0;
    for (int i = lo + tid; i < hi; i += 256) {
        const unsigned long long c = h[i];
        s0 += c;
        s1 += c * (unsigned long long)i;
    }
    __shared__ unsigned long long r0[256], r1[256];
    __shared__ double s_ratio;
    r0[tid] = s0; r1[tid] = s1;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) { r0[tid] += r0[tid + st]; r1[tid] += r1[tid + st]; }
        __syncthreads();
    }
    if (tid == 0) {
        const double ratio = r0[0] ? ref.mean[t - first_channel] / ((double)r1[0] / (double)r0[0]) : 1.0;
        s_ratio = ratio;
        if (corr_out) corr_out[t - first_channel] = ratio;
    }
    __syncthreads();
    const double ratio = s_ratio, vmax = (double)(nbins - 1);
    for (int i = tid; i < nbins; i += 256) {
        double v = (double)i * ratio;
        v = v < 0.0 ? 0.0 : (v > vmax ? vmax : v);
        out[i] = (T)v;
    }
}

// ---------------------------------------------------------------- 8-bit BGR <-> HSV / HLS
// cv2.cvtColor(COLOR_BGR2HSV / HSV2BGR / BGR2HLS / HLS2BGR) on uint8 images: the pre- and post-processing of the
// reference's SVCorrection / LSCorrection (balance.py:340-363; cv2 has no 16-bit form of these conversions, the
// reference raises there).  OpenCV's color_hsv arithmetic restated [from memory -- parity unpinned]: BGR2HSV in
// integers with 12-bit reciprocal tables (H in [0, 180)), the other three through float32, one rounding per written
// operation (the translation unit is built with fp-contract off), round-half-even + saturate at the end.
// oracle/oracle.py (bgr2hsv_u8 ...) states the same operations in NumPy float32: equal bit for bit.
enum { CVT_BGR2HSV = 0, CVT_HSV2BGR = 1, CVT_BGR2HLS = 2, CVT_HLS2BGR = 3 };

__device__ __forceinline__ int sat_u8(float v) {
    const float r = rintf(v);
    return (int)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
}

__device__ __forceinline__ void sector_pick(const float tab[4], int sector, float& b, float& g, float& r) {
    // sector_data[6][3] = {1,3,0},{1,0,2},{3,0,1},{0,2,1},{0,1,3},{2,1,0}
    const int ib = (0x200311 >> (4 * sector)) & 15;
    const int ig = (0x112003 >> (4 * sector)) & 15;
    const int ir = (0x031120 >> (4 * sector)) & 15;
    auto pick = [&](int i) { return i == 0 ? tab[0] : (i == 1 ? tab[1] : (i == 2 ? tab[2] : tab[3])); };
    b = pick(ib); g = pick(ig); r = pick(ir);
}

template <int CODE>
__global__ __launch_bounds__(256) void cvt_color_u8(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t npix) {
    __shared__ int sdiv[256], hdiv[256];
    if constexpr (CODE == CVT_BGR2HSV) {
        const int i = threadIdx.x;   // blockDim.x == 256
        sdiv[i] = i ? (int)rint((double)(255 << 12) / (1.0 * i)) : 0;
        hdiv[i] = i ? (int)rint((double)(180 << 12) / (6.0 * i)) : 0;
        __syncthreads();
    }
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
        const int c0 = src[p * 3], c1 = src[p * 3 + 1], c2 = src[p * 3 + 2];
        int o0, o1, o2;
        if constexpr (CODE == CVT_BGR2HSV) {
            const int b = c0, g = c1, r = c2;
            const int v = max(max(b, g), r), vmin = min(min(b, g), r), diff = v - vmin;
            const int s = (diff * sdiv[v] + (1 << 11)) >> 12;
            int h = v == r ? g - b : (v == g ? b - r + 2 * diff : r - g + 4 * diff);
            h = (h * hdiv[diff] + (1 << 11)) >> 12;
            h += h < 0 ? 180 : 0;
            o0 = h & 255; o1 = s & 255; o2 = v;
        } else if constexpr (CODE == CVT_HSV2BGR) {
            const float h0 = (float)c0, s = (float)c1 * (1.0f / 255.0f), v = (float)c2 * (1.0f / 255.0f);
            float hh = h0 * (6.0f / 180.0f);
            int sector = (int)floorf(hh);
            hh = hh - (float)sector;
            if ((unsigned)sector >= 6u) { sector = 0; hh = 0.f; }
            const float tab[4] = {v, v * (1.0f - s), v * (1.0f - s * hh), v * (1.0f - s * (1.0f - hh))};
            float b, g, r;
            sector_pick(tab, sector, b, g, r);
            if (s == 0.f) b = g = r = v;
            o0 = sat_u8(b * 255.0f); o1 = sat_u8(g * 255.0f); o2 = sat_u8(r * 255.0f);
        } else if constexpr (CODE == CVT_BGR2HLS) {
            const float b = (float)c0 * (1.0f / 255.0f), g = (float)c1 * (1.0f / 255.0f), r = (float)c2 * (1.0f / 255.0f);
            const float vmax = fmaxf(fmaxf(r, g), b), vmin = fminf(fminf(r, g), b);
            const float diff = vmax - vmin, l = (vmax + vmin) * 0.5f;
            float h = 0.f, s = 0.f;
            if (diff > 1.1920929e-07f) {
                s = l < 0.5f ? diff / (vmax + vmin) : diff / (2.0f - vmax - vmin);
                const float d60 = 60.0f / diff;
                if (vmax == r) h = (g - b) * d60;
                else if (vmax == g) h = (b - r) * d60 + 120.0f;
                else h = (r - g) * d60 + 240.0f;
                if (h < 0.f) h = h + 360.0f;
            }
            h = h * (180.0f / 360.0f);
            o0 = sat_u8(h); o1 = sat_u8(l * 255.0f); o2 = sat_u8(s * 255.0f);
        } else {
            const float h0 = (float)c0, l = (float)c1 * (1.0f / 255.0f), s = (float)c2 * (1.0f / 255.0f);
            const float p2 = l <= 0.5f ? l * (1.0f + s) : l + s - l * s;
            const float p1 = 2.0f * l - p2;
            float hh = h0 * (6.0f / 180.0f);
            int sector = (int)floorf(hh);
            hh = hh - (float)sector;
            if ((unsigned)sector >= 6u) { sector = 0; hh = 0.f; }
            const float tab[4] = {p2, p1, p1 + (p2 - p1) * (1.0f - hh), p1 + (p2 - p1) * hh};
            float b, g, r;
            sector_pick(tab, sector, b, g, r);
            if (s == 0.f) b = g = r = l;
            o0 = sat_u8(b * 255.0f); o1 = sat_u8(g * 255.0f); o2 = sat_u8(r * 255.0f);
        }
        dst[p * 3] = (uint8_t)o0; dst[p * 3 + 1] = (uint8_t)o1; dst[p * 3 + 2] = (uint8_t)o2;
    }
}

}  // namespace mi
