// kernels_f64.hpp -- the float_type='float-64' variant of the pyramid path
// (base_stack_algo.py:14-17: self.float_type = np.float64).  What changes against float-32
// (pyramid.py line by line):
//   * :126  pyramid images are float64; cv2.filter2D on CV_64F keeps the float64 generating kernel
//           np.outer(k, k) and accumulates in double -> reduce / expand / collapse chains in double;
//   * :49   energies stay float32: cvtColor(lap.astype(np.float32)), square and filter in float32;
//   * :51-54 fused Laplacians are float64;
//   * :68-69, :77  p = counts.astype(float64) / counts.sum(), entropy = float64(-sum(level * log(p)))
//           with NumPy's pairwise float64 sum;  :85 deviation in float64;
//   * :106-111 fused base float64; :57-64 collapse float64; :179 truncating cast.
// One thread per output, global-memory taps (the structure of kernels_simple.hpp): float-64 is a
// precision option nobody benchmarks -- the reference's examples and tests all run float-32 -- so it
// gets the simple formulation, one frame at a time.
// log(p): the host fills the table (logl rounded to double; NumPy's float64 log is CPU-dispatch
// dependent, so the oracle uses the same definition -- see oracle/ref_import.py).
#pragma once
#include "common.hpp"

namespace mi {

struct K25d {
    double k[25];
};

template <bool FMA>
__device__ __forceinline__ double macd(double k, double x, double s) {
    if constexpr (FMA) return __builtin_fma(k, x, s);
    else {
        double p = k * x;  // contraction is off for this TU
        return s + p;
    }
}

template <typename TSrc, bool FMA>
__global__ void reduce_f64(const TSrc* __restrict__ g, int h, int w, double* __restrict__ out, int ho, int wo,
                           K25d K) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int i = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= ho || j >= wo) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int ty = 0; ty < 5; ++ty) {
        const TSrc* row = g + (size_t)r101(2 * i + ty - 2, h) * w * 3;
#pragma unroll
        for (int tx = 0; tx < 5; ++tx) {
            const TSrc* p = row + (size_t)r101(2 * j + tx - 2, w) * 3;
            const double k = K.k[ty * 5 + tx];
            s0 = macd<FMA>(k, (double)p[0], s0);
            s1 = macd<FMA>(k, (double)p[1], s1);
            s2 = macd<FMA>(k, (double)p[2], s2);
        }
    }
    double* o = out + ((size_t)i * wo + j) * 3;
    o[0] = s0; o[1] = s1; o[2] = s2;
}

template <bool FMA>
__device__ __forceinline__ void expand_at_f64(const double* __restrict__ src, int hs, int ws, const K25d& K, int y,
                                              int x, double& e0, double& e1, double& e2) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    const int H2 = 2 * hs, W2 = 2 * ws;
#pragma unroll
    for (int ty = 0; ty < 5; ++ty) {
        int yy = r101(y + ty - 2, H2);
        if (yy & 1) continue;
#pragma unroll
        for (int tx = 0; tx < 5; ++tx) {
            int xx = r101(x + tx - 2, W2);
            if (xx & 1) continue;
            const double* p = src + ((size_t)(yy >> 1) * ws + (xx >> 1)) * 3;
            const double k = K.k[ty * 5 + tx];
            s0 = macd<FMA>(k, p[0], s0);
            s1 = macd<FMA>(k, p[1], s1);
            s2 = macd<FMA>(k, p[2], s2);
        }
    }
    e0 = 4.0 * s0; e1 = 4.0 * s1; e2 = 4.0 * s2;
}

template <typename TSrc, bool FMA>
__global__ void lapq_f64(const TSrc* __restrict__ g, int h, int w, const double* __restrict__ gn, int hs, int ws,
                         double* __restrict__ lap, float* __restrict__ q, K25d K) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (y >= h || x >= w) return;
    double e0, e1, e2;
    expand_at_f64<FMA>(gn, hs, ws, K, y, x, e0, e1, e2);
    const size_t p = (size_t)y * w + x;
    const double l0 = (double)g[p * 3 + 0] - e0, l1 = (double)g[p * 3 + 1] - e1, l2 = (double)g[p * 3 + 2] - e2;
    lap[p * 3 + 0] = l0; lap[p * 3 + 1] = l1; lap[p * 3 + 2] = l2;
    const float gr = gray_of<FMA>((float)l0, (float)l1, (float)l2);  // lap.astype(np.float32), pyramid.py:49
    q[p] = gr * gr;
}

template <bool FMA>
__global__ void select_f64(const float* __restrict__ q, const double* __restrict__ lap, int h, int w, int frame_idx,
                           int first, float* __restrict__ best_e, double* __restrict__ best_lap,
                           int32_t* __restrict__ best_idx, K25 K) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (y >= h || x >= w) return;
    float s = 0.f;
#pragma unroll
    for (int ty = 0; ty < 5; ++ty) {
        const float* row = q + (size_t)r101(y + ty - 2, h) * w;
#pragma unroll
        for (int tx = 0; tx < 5; ++tx) s = mac<FMA>(K.k[ty * 5 + tx], row[r101(x + tx - 2, w)], s);
    }
    const size_t p = (size_t)y * w + x;
    if (first || s > best_e[p]) {
        best_e[p] = s;
        best_idx[p] = frame_idx;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double lv = lap[p * 3 + c];
            best_lap[p * 3 + c] = (lv == 0.0) ? 0.0 : lv;  // -0 -> +0, as the np.where sum gives
        }
    }
}

template <bool FMA>
__global__ void base_gray_hist_f64(const double* __restrict__ base, int npix, int nlevels, int32_t* __restrict__ lev,
                                   uint32_t* __restrict__ cnt) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float gr = gray_of<FMA>((float)base[3 * i], (float)base[3 * i + 1], (float)base[3 * i + 2]);
    int l = (int)gr;  // .astype(uint8/uint16): truncation
    l = l < 0 ? 0 : (l >= nlevels ? nlevels - 1 : l);
    lev[i] = l;
    atomicAdd(&cnt[l], 1u);
}

// NumPy's float64 add.reduce order (pairwise, 8 accumulators) for n <= 128
template <typename F>
__device__ __forceinline__ double np_sum_d(int n, F elem) {
    if (n < 8) {
        double res = -0.0;
        for (int i = 0; i < n; ++i) res += elem(i);
        return res;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = elem(j);
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += elem(i + j);
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += elem(i);
    return res;
}

__global__ void base_feat_select_f64(const int32_t* __restrict__ lev, const double* __restrict__ logp,
                                     const double* __restrict__ base, int hb, int wb, int pad, int frame_idx,
                                     int first, double* __restrict__ best_ent, double* __restrict__ best_dev,
                                     int32_t* __restrict__ idx_e, int32_t* __restrict__ idx_d,
                                     double* __restrict__ base_e, double* __restrict__ base_d) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (y >= hb || x >= wb) return;
    const int win = 2 * pad + 1, n = win * win;
    auto level_at = [&](int t) {
        int dy = t / win - pad, dx = t % win - pad;
        return lev[(size_t)r101_loop(y + dy, hb) * wb + r101_loop(x + dx, wb)];
    };
    const double ent = -1.0 * np_sum_d(n, [&](int t) {
                           const int l = level_at(t);
                           return (double)l * logp[l];
                       });
    // np.average(area): integer levels summed in float64 (exact), divided once
    const double mean = np_sum_d(n, [&](int t) { return (double)level_at(t); }) / (double)n;
    const double dev = np_sum_d(n, [&](int t) {
                           const double d = (double)level_at(t) - mean;
                           return d * d;
                       }) / (double)n;
    const size_t p = (size_t)y * wb + x;
    if (first || ent > best_ent[p]) {
        best_ent[p] = ent;
        idx_e[p] = frame_idx;
        for (int c = 0; c < 3; ++c) base_e[p * 3 + c] = base[p * 3 + c];
    }
    if (first || dev > best_dev[p]) {
        best_dev[p] = dev;
        idx_d[p] = frame_idx;
        for (int c = 0; c < 3; ++c) base_d[p * 3 + c] = base[p * 3 + c];
    }
}

__global__ void base_fuse_f64(const double* __restrict__ base_e, const double* __restrict__ base_d, size_t n,
                              double* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0 + base_e[i];
    s = s + base_d[i];
    out[i] = s / 2.0;
}

template <bool FMA>
__global__ void collapse_f64(const double* __restrict__ up, int hs, int ws, const double* __restrict__ lap, int h,
                             int w, double* __restrict__ out, K25d K) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (y >= h || x >= w) return;
    double e0, e1, e2;
    expand_at_f64<FMA>(up, hs, ws, K, y, x, e0, e1, e2);
    const size_t p = ((size_t)y * w + x) * 3;
    out[p + 0] = e0 + lap[p + 0];
    out[p + 1] = e1 + lap[p + 1];
    out[p + 2] = e2 + lap[p + 2];
}

template <typename TOut>
__global__ void finalize_cast_f64(const double* __restrict__ img, size_t n, double maxv, double* __restrict__ clipped,
                                  TOut* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = fabs(img[i]);
    v = v > maxv ? maxv : v;
    if (clipped) clipped[i] = v;
    out[i] = (TOut)v;
}

template <typename TIn>
__global__ void frame_to_f64(const TIn* __restrict__ src, size_t n, double* __restrict__ dst) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (double)src[i];
}

}  // namespace mi
