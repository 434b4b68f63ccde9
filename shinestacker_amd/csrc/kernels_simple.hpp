// kernels_simple.hpp -- one-thread-per-output kernels with global-memory taps.
// They are the on-GPU cross-check (MI_IMPL_SIMPLE) for the LDS-tiled production
// kernels, and the implementation of everything that is O(1) in the frame
// count (base level, collapse, finalise, cross-GPU combine).
//
// Arithmetic contract (identical in every implementation and in oracle/):
//   * 5x5 stencils: taps in row-major order, s = 0, s = mac(k, x, s) per tap
//     (reference algorithms/pyramid.py:24-25 -> cv2.filter2D, REFLECT101);
//   * reduce  = that stencil at even coordinates (pyramid.py:27-32);
//   * expand  = the stencil on the zero-stuffed 2h x 2w grid with the zero taps
//     skipped (exact: they add +0), times 4 (pyramid.py:34-46);
//   * laplacian = G_l - expand(G_{l+1})[:h,:w] (pyramid.py:133-138);
//   * energy = stencil(gray(lap)^2), selection = first maximum over frames
//     (pyramid.py:48-55), winner's lap with -0 -> +0.
#pragma once
#include "common.hpp"

namespace mi {

// ---------------------------------------------------------------- reduce
template <typename TIn, bool FMA>
__global__ void reduce_simple(const TIn* __restrict__ g, int h, int w, float* __restrict__ out,
                              int ho, int wo, K25 K) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int i = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= ho || j >= wo) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int ty = 0; ty < 5; ++ty) {
        const TIn* row = g + (size_t)r101(2 * i + ty - 2, h) * w * 3;
#pragma unroll
        for (int tx = 0; tx < 5; ++tx) {
            const TIn* p = row + (size_t)r101(2 * j + tx - 2, w) * 3;
            float k = K.k[ty * 5 + tx];
            s0 = mac<FMA>(k, to_f32(p[0]), s0);
            s1 = mac<FMA>(k, to_f32(p[1]), s1);
            s2 = mac<FMA>(k, to_f32(p[2]), s2);
        }
    }
    float* o = out + ((size_t)i * wo + j) * 3;
    o[0] = s0;
    o[1] = s1;
    o[2] = s2;
}

// expand_layer(src)[y, x, :] for an hs x ws x 3 source
template <bool FMA>
__device__ __forceinline__ void expand_at(const float* __restrict__ src, int hs, int ws,
                                          const K25& K, int y, int x, float& e0, float& e1,
                                          float& e2) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    const int H2 = 2 * hs, W2 = 2 * ws;
#pragma unroll
    for (int ty = 0; ty < 5; ++ty) {
        int yy = r101(y + ty - 2, H2);
        if (yy & 1) continue;
#pragma unroll
        for (int tx = 0; tx < 5; ++tx) {
            int xx = r101(x + tx - 2, W2);
            if (xx & 1) continue;
            const float* p = src + ((size_t)(yy >> 1) * ws + (xx >> 1)) * 3;
            float k = K.k[ty * 5 + tx];
            s0 = mac<FMA>(k, p[0], s0);
            s1 = mac<FMA>(k, p[1], s1);
            s2 = mac<FMA>(k, p[2], s2);
        }
    }
    e0 = 4.0f * s0;
    e1 = 4.0f * s1;
    e2 = 4.0f * s2;
}

// ---------------------------------------------------------------- laplacian + gray^2
template <typename TIn, bool FMA>
__global__ void lapq_simple(const TIn* __restrict__ g, int h, int w, const float* __restrict__ gn,
                            int hs, int ws, float* __restrict__ lap, float* __restrict__ q,
                            K25 K) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (y >= h || x >= w) return;
    float e0, e1, e2;
    expand_at<FMA>(gn, hs, ws, K, y, x, e0, e1, e2);
    size_t p = (size_t)y * w + x;
    float l0 = to_f32(g[p * 3 + 0]) - e0;
    float l1 = to_f32(g[p * 3 + 1]) - e1;
    float l2 = to_f32(g[p * 3 + 2]) - e2;
    lap[p * 3 + 0] = l0;
    lap[p * 3 + 1] = l1;
    lap[p * 3 + 2] = l2;
    float gr = gray_of<FMA>(l0, l1, l2);
    q[p] = gr * gr;
}

// ---------------------------------------------------------------- energy + running first-max
template <bool FMA>
__global__ void select_simple(const float* __restrict__ q, const float* __restrict__ lap, int h,
                              int w, int frame_idx, int first, float* __restrict__ best_e,
                              float* __restrict__ best_lap, int32_t* __restrict__ best_idx,
                              K25 K) {
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
    size_t p = (size_t)y * w + x;
    if (first || s > best_e[p]) {
        best_e[p] = s;
        best_idx[p] = frame_idx;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float lv = lap[p * 3 + c];
            best_lap[p * 3 + c] = (lv == 0.0f) ? 0.0f : lv;
        }
    }
}

// ---------------------------------------------------------------- base level (pyramid.py:66-111)
template <bool FMA>
__global__ void base_gray_hist(const float* __restrict__ base, int npix, int nlevels,
                               int32_t* __restrict__ lev, uint32_t* __restrict__ cnt) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    float gr = gray_of<FMA>(base[3 * i], base[3 * i + 1], base[3 * i + 2]);
    int l = (int)gr;  // .astype(uint8/uint16): truncation
    l = l < 0 ? 0 : (l >= nlevels ? nlevels - 1 : l);
    lev[i] = l;
    atomicAdd(&cnt[l], 1u);
}

// log table: p = float32(float64(float32(count)) / float64(npix)); logp = float32(log(float64(p)))
__global__ void base_logp(const uint32_t* __restrict__ cnt, int nlevels, int npix,
                          float* __restrict__ logp) {
    int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nlevels) return;
    uint32_t c = cnt[l];
    float v = 0.f;
    if (c) {
        float p = (float)((double)(float)c / (double)npix);
        v = (float)log((double)p);
    }
    logp[l] = v;
}

// NumPy's float32 add.reduce order (pairwise, 8 accumulators) for n <= 128
template <typename F>
__device__ __forceinline__ float np_sum(int n, F elem) {
    if (n < 8) {
        float res = -0.0f;
        for (int i = 0; i < n; ++i) res += elem(i);
        return res;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = elem(j);
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += elem(i + j);
    }
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += elem(i);
    return res;
}

// entropy / deviation features of one frame's base + running first-max selection.
// The winner's base pixel is copied next to the running maxima so the final
// (img[best_e] + img[best_d]) / 2 needs no per-frame storage.
__global__ void base_feat_select(const int32_t* __restrict__ lev, const float* __restrict__ logp,
                                 const float* __restrict__ base, int hb, int wb, int pad,
                                 int frame_idx, int first, float* __restrict__ best_ent,
                                 float* __restrict__ best_dev, int32_t* __restrict__ idx_e,
                                 int32_t* __restrict__ idx_d, float* __restrict__ base_e,
                                 float* __restrict__ base_d) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (y >= hb || x >= wb) return;
    const int win = 2 * pad + 1, n = win * win;
    auto level_at = [&](int t) {
        int dy = t / win - pad, dx = t % win - pad;
        return lev[(size_t)r101_loop(y + dy, hb) * wb + r101_loop(x + dx, wb)];
    };
    float ent = -1.0f * np_sum(n, [&](int t) {
                    int l = level_at(t);
                    return (float)l * logp[l];
                });
    double isum = 0.0;
    for (int t = 0; t < n; ++t) isum += (double)level_at(t);
    float mean = (float)(isum / (double)n);
    float dev = np_sum(n, [&](int t) {
                    float d = (float)level_at(t) - mean;
                    return d * d;
                }) / (float)n;
    size_t p = (size_t)y * wb + x;
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

// batch version, two launches for `nframes` consecutive frames: the features of every (frame, pixel) in parallel
// -- one thread per pixel walking the frames left a 63 x 94 base with 93 waves and 0.66 ms per 32-frame batch, all of
// it latency -- then the per-pixel first-max scan over the frames in index order.
__global__ void base_feat_batch(const int32_t* __restrict__ lev, const float* __restrict__ logp, int nlevels, int hb,
                                int wb, int pad, float* __restrict__ feat) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int f = blockIdx.z;
    if (y >= hb || x >= wb) return;
    const int npix = hb * wb;
    const int32_t* lv = lev + (size_t)f * npix;
    const float* lp = logp + (size_t)f * nlevels;
    const int win = 2 * pad + 1, n = win * win;
    auto level_at = [&](int t) {
        int dy = t / win - pad, dx = t % win - pad;
        return lv[(size_t)r101_loop(y + dy, hb) * wb + r101_loop(x + dx, wb)];
    };
    float ent = -1.0f * np_sum(n, [&](int t) {
                    int l = level_at(t);
                    return (float)l * lp[l];
                });
    double isum = 0.0;
    for (int t = 0; t < n; ++t) isum += (double)level_at(t);
    float mean = (float)(isum / (double)n);
    float dev = np_sum(n, [&](int t) {
                    float d = (float)level_at(t) - mean;
                    return d * d;
                }) / (float)n;
    float* o = feat + ((size_t)f * npix + (size_t)y * wb + x) * 2;
    o[0] = ent;
    o[1] = dev;
}

// np_sum for a compile-time n >= 8, every index a constant (the elements may live in registers)
template <int N, typename F>
__device__ __forceinline__ float np_sum_c(F elem) {
    static_assert(N >= 8 && N <= 128, "pairwise block of NumPy's add.reduce");
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = elem(j);
#pragma unroll
    for (int i = 8; i < N - (N % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += elem(i + j);
    }
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
    for (int i = N - (N % 8); i < N; ++i) res += elem(i);
    return res;
}

// base_feat_batch for a compile-time window (PAD = 2: the default 5 x 5): the window's levels and their log-probability
// terms are gathered ONCE into registers -- the generic kernel recomputes two reflections, a division and a modulo per tap
// in each of its three passes (3000 instructions per pixel: 118 us for 256 bases of 63 x 94) -- then the same three sums.
template <int PAD>
__global__ void base_feat_batch_c(const int32_t* __restrict__ lev, const float* __restrict__ logp, int nlevels, int hb,
                                  int wb, float* __restrict__ feat) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y, f = blockIdx.z;
    if (y >= hb || x >= wb) return;
    constexpr int WIN = 2 * PAD + 1, N = WIN * WIN;
    const int npix = hb * wb;
    const int32_t* lv = lev + (size_t)f * npix;
    const float* lp = logp + (size_t)f * nlevels;
    int xs[WIN], ys[WIN];
#pragma unroll
    for (int o = 0; o < WIN; ++o) {
        xs[o] = r101_loop(x + o - PAD, wb);
        ys[o] = r101_loop(y + o - PAD, hb) * wb;
    }
    float lf[N], term[N];
#pragma unroll
    for (int t = 0; t < N; ++t) {
        const int l = lv[ys[t / WIN] + xs[t % WIN]];
        lf[t] = (float)l;
        term[t] = (float)l * lp[l];
    }
    const float ent = -1.0f * np_sum_c<N>([&](int t) { return term[t]; });
    double isum = 0.0;
#pragma unroll
    for (int t = 0; t < N; ++t) isum += (double)(int)lf[t];
    const float mean = (float)(isum / (double)N);
    const float dev = np_sum_c<N>([&](int t) {
                          const float d = lf[t] - mean;
                          return d * d;
                      }) / (float)N;
    float* o = feat + ((size_t)f * npix + (size_t)y * wb + x) * 2;
    o[0] = ent;
    o[1] = dev;
}

__global__ void base_select_batch(const float* __restrict__ feat, const float* __restrict__ bases, size_t base_stride,
                                  int nframes, int npix, int frame_idx0, int first, float* __restrict__ best_ent,
                                  float* __restrict__ best_dev, int32_t* __restrict__ idx_e,
                                  int32_t* __restrict__ idx_d, float* __restrict__ base_e,
                                  float* __restrict__ base_d) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float be = first ? -INFINITY : best_ent[p], bd = first ? -INFINITY : best_dev[p];
    int ie = first ? -1 : idx_e[p], id = first ? -1 : idx_d[p];
    int fe = -1, fd = -1;  // frame of this batch that currently holds the maximum
    for (int f = 0; f < nframes; ++f) {
        const float ent = feat[((size_t)f * npix + p) * 2], dev = feat[((size_t)f * npix + p) * 2 + 1];
        // the very first frame of a stack wins unconditionally (as `first ||` does in the
        // single-frame kernel): -inf start values give exactly that for finite features
        if (ent > be || (first && f == 0)) { be = ent; ie = frame_idx0 + f; fe = f; }
        if (dev > bd || (first && f == 0)) { bd = dev; id = frame_idx0 + f; fd = f; }
    }
    best_ent[p] = be;
    best_dev[p] = bd;
    idx_e[p] = ie;
    idx_d[p] = id;
    if (fe >= 0) {
        const float* b = bases + (size_t)fe * base_stride + (size_t)p * 3;
        base_e[p * 3 + 0] = b[0]; base_e[p * 3 + 1] = b[1]; base_e[p * 3 + 2] = b[2];
    }
    if (fd >= 0) {
        const float* b = bases + (size_t)fd * base_stride + (size_t)p * 3;
        base_d[p * 3 + 0] = b[0]; base_d[p * 3 + 1] = b[1]; base_d[p * 3 + 2] = b[2];
    }
}

// The same scan with the frames of a pixel split over SEG neighbouring lanes (one thread per pixel walking 256 frames left
// 93 waves and 72 us of pure load latency): every lane scans its contiguous share with the strict '>', then the shares are
// merged in frame order, again with the strict '>' -- the earliest frame holding the maximum wins, as in the serial scan.
template <int SEG>
__global__ __launch_bounds__(256) void base_select_batch_seg(const float* __restrict__ feat, const float* __restrict__ bases,
                                                             size_t base_stride, int nframes, int npix, int frame_idx0, int first,
                                                             float* __restrict__ best_ent, float* __restrict__ best_dev,
                                                             int32_t* __restrict__ idx_e, int32_t* __restrict__ idx_d,
                                                             float* __restrict__ base_e, float* __restrict__ base_d) {
    const int t = blockIdx.x * 256 + threadIdx.x, p = t / SEG, seg = t % SEG;
    const bool live = p < npix;
    const int per = (nframes + SEG - 1) / SEG, f_lo = seg * per, f_hi = min(nframes, f_lo + per);
    float be = -INFINITY, bd = -INFINITY;
    int fe = -1, fd = -1;
    if (live)
        for (int f = f_lo; f < f_hi; ++f) {
            const float ent = feat[((size_t)f * npix + p) * 2], dev = feat[((size_t)f * npix + p) * 2 + 1];
            if (ent > be || (first && f == 0)) { be = ent; fe = f; }
            if (dev > bd || (first && f == 0)) { bd = dev; fd = f; }
        }
    // merge the SEG shares (lanes lane0 .. lane0 + SEG - 1 of the wave) in frame order
    const int lane0 = (threadIdx.x & 63) - seg;
    float me = -INFINITY, md = -INFINITY;
    int ge = -1, gd = -1;
#pragma unroll
    for (int q = 0; q < SEG; ++q) {
        const float e2 = __shfl(be, lane0 + q), d2 = __shfl(bd, lane0 + q);
        const int fe2 = __shfl(fe, lane0 + q), fd2 = __shfl(fd, lane0 + q);
        if (fe2 >= 0 && (e2 > me || (first && fe2 == 0))) { me = e2; ge = fe2; }
        if (fd2 >= 0 && (d2 > md || (first && fd2 == 0))) { md = d2; gd = fd2; }
    }
    if (!live || seg != 0) return;
    // against the running state of earlier batches (a fresh stack has none)
    if (!first) {
        const float pe = best_ent[p], pd = best_dev[p];
        if (!(ge >= 0 && me > pe)) ge = -1;
        if (!(gd >= 0 && md > pd)) gd = -1;
    }
    if (ge >= 0) {
        best_ent[p] = me;
        idx_e[p] = frame_idx0 + ge;
        const float* b = bases + (size_t)ge * base_stride + (size_t)p * 3;
        base_e[p * 3 + 0] = b[0]; base_e[p * 3 + 1] = b[1]; base_e[p * 3 + 2] = b[2];
    }
    if (gd >= 0) {
        best_dev[p] = md;
        idx_d[p] = frame_idx0 + gd;
        const float* b = bases + (size_t)gd * base_stride + (size_t)p * 3;
        base_d[p * 3 + 0] = b[0]; base_d[p * 3 + 1] = b[1]; base_d[p * 3 + 2] = b[2];
    }
}

__global__ void base_fuse(const float* __restrict__ base_e, const float* __restrict__ base_d,
                          size_t n, float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.0f + base_e[i];  // zeros + where(best_e == f, img, 0) ...
    s = s + base_d[i];           // ... + where(best_d == f, img, 0); a+b is commutative
    out[i] = s / 2.0f;
}

// ---------------------------------------------------------------- collapse (pyramid.py:57-64)
template <bool FMA>
__global__ void collapse_simple(const float* __restrict__ up, int hs, int ws,
                                const float* __restrict__ lap, int h, int w,
                                float* __restrict__ out, K25 K) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (y >= h || x >= w) return;
    float e0, e1, e2;
    expand_at<FMA>(up, hs, ws, K, y, x, e0, e1, e2);
    size_t p = ((size_t)y * w + x) * 3;
    out[p + 0] = e0 + lap[p + 0];
    out[p + 1] = e1 + lap[p + 1];
    out[p + 2] = e2 + lap[p + 2];
}

// the finest collapse step fused with clip(abs()) and the cast (pyramid.py:62-64, :179): the collapsed float image
// never goes to HBM (mi_stack_get_level(MI_TAP_COLLAPSED) rebuilds it on request)
template <bool FMA, typename TOut>
__global__ void collapse_final(const float* __restrict__ up, int hs, int ws, const float* __restrict__ lap, int h,
                               int w, float maxv, TOut* __restrict__ out, K25 K) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (y >= h || x >= w) return;
    float e[3];
    expand_at<FMA>(up, hs, ws, K, y, x, e[0], e[1], e[2]);
    const size_t p = ((size_t)y * w + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = fabsf(e[c] + lap[p + c]);
        v = v > maxv ? maxv : v;
        out[p + c] = (TOut)v;
    }
}

// clip(abs(img), 0, max) then .astype(dtype) (truncation), pyramid.py:64, :179
template <typename TOut>
__global__ void finalize_cast(const float* __restrict__ img, size_t n, float maxv,
                              float* __restrict__ clipped, TOut* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = fabsf(img[i]);
    v = v > maxv ? maxv : v;
    if (clipped) clipped[i] = v;
    out[i] = (TOut)v;
}

// ---------------------------------------------------------------- cross-GPU combine
// candidates in ascending global frame order: strict '>' keeps the first maximum.
__global__ void combine_select(int n, const float* __restrict__ cand_e,
                               const float* __restrict__ cand_lap,
                               const int32_t* __restrict__ cand_idx, size_t npix,
                               float* __restrict__ out_e, float* __restrict__ out_lap,
                               int32_t* __restrict__ out_idx) {
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float be = cand_e[p];
    int bi = 0;
    for (int r = 1; r < n; ++r) {
        float e = cand_e[(size_t)r * npix + p];
        if (e > be) {
            be = e;
            bi = r;
        }
    }
    out_e[p] = be;
    const float* l = cand_lap + ((size_t)bi * npix + p) * 3;
    out_lap[p * 3 + 0] = l[0];
    out_lap[p * 3 + 1] = l[1];
    out_lap[p * 3 + 2] = l[2];
    if (cand_idx && out_idx) out_idx[p] = cand_idx[(size_t)bi * npix + p];
}

// ---- "winners only" form of the combine: the ranks exchange energies, agree on the winning rank of every pixel, and
// each rank sends just the payload rows it won, packed in pixel order, straight to the collapsing rank.
// Blocks of CB_PX pixels; pos(p) = offsets[block][rank of p] + (pixels of the same rank before p inside the block).
constexpr int CB_PX = 1024, CB_MAXR = 16;

__global__ void combine_winner(int n, const float* __restrict__ cand_e, size_t npix, uint8_t* __restrict__ win) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float be = cand_e[p];
    int bi = 0;
    for (int r = 1; r < n; ++r) {
        const float e = cand_e[(size_t)r * npix + p];
        if (e > be) { be = e; bi = r; }   // strict: the lowest rank (= lowest global frame index) keeps a tie
    }
    win[p] = (uint8_t)bi;
}

// The same when the ranks hold INTERLEAVED frames (rank r: frames r, r + W, ...): the rank order is no longer the frame order, so a
// tie between ranks goes to the candidate with the lower global frame index -- np.argmax's first maximum (pyramid.py:51).
__global__ void combine_winner_idx(int n, const float* __restrict__ cand_e, const int32_t* __restrict__ cand_idx, size_t npix,
                                   uint8_t* __restrict__ win) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float be = cand_e[p];
    int32_t bf = cand_idx[p];
    int bi = 0;
    for (int r = 1; r < n; ++r) {
        const float e = cand_e[(size_t)r * npix + p];
        const int32_t f = cand_idx[(size_t)r * npix + p];
        if (e > be || (e == be && f < bf)) { be = e; bf = f; bi = r; }
    }
    win[p] = (uint8_t)bi;
}

// stored winner index (first + consecutive frame number) -> global index (first + number * stride), in place
__global__ void idx_export(int32_t* __restrict__ idx, size_t n, int first, int stride) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int32_t v = idx[p];
    if (v >= first) idx[p] = first + (v - first) * stride;
}

// Blocks of CB_PX = 1024 pixels, one workgroup of 256 threads each, a thread = 4 consecutive pixels (one 32-bit load of the
// winner map).  Inside a block the position of a pixel among the pixels of ITS rank comes from one packed prefix scan:
// the thread's per-rank counts (0..4) sit in 16-bit fields of 64-bit words (4 ranks per word), scanned across the wave by
// shuffles and across the four waves through LDS -- one barrier per block, no per-rank ballot loops (round 3: four passes
// of 256 pixels with three barriers and 2 x n_ranks ballots each; 1.7 ms of kernels per combine at 32 Mpx).
constexpr int CB_WORDS = CB_MAXR / 4;
struct CbCounts { unsigned long long w[CB_WORDS]; };

__device__ __forceinline__ unsigned long long shfl_up64(unsigned long long v, int d) {
    const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, d), hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), d);
    return ((unsigned long long)hi << 32) | lo;
}
// per-thread winner bytes (r[4], -1 = outside the image) -> exclusive prefix (over the threads of the block, in pixel
// order) and block totals of the per-rank counts; NW = words in use = ceil(n_ranks / 4)
template <int NW>
__device__ __forceinline__ void cb_block_scan(const int* r, CbCounts& excl, CbCounts& total, unsigned long long (*sh)[CB_WORDS]) {
    CbCounts c{};
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (r[j] >= 0) c.w[r[j] >> 2] += 1ull << (16 * (r[j] & 3));
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    CbCounts inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const unsigned long long up = shfl_up64(inc.w[k], d);
            if (lane >= d) inc.w[k] += up;
        }
    }
    if (lane == 63)
#pragma unroll
        for (int k = 0; k < NW; ++k) sh[wv][k] = inc.w[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        unsigned long long before = 0, all = 0;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const unsigned long long t = sh[v][k];
            if (v < wv) before += t;
            all += t;
        }
        excl.w[k] = before + inc.w[k] - c.w[k];
        total.w[k] = all;
    }
}
__device__ __forceinline__ uint32_t cb_field(const CbCounts& c, int r) { return (uint32_t)(c.w[r >> 2] >> (16 * (r & 3))) & 0xffffu; }
__device__ __forceinline__ void cb_load4(const uint8_t* __restrict__ win, size_t npix, size_t p0, int* r) {
    if (p0 + 3 < npix && (p0 & 3) == 0) {
        const uint32_t v = *reinterpret_cast<const uint32_t*>(win + p0);
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = (int)((v >> (8 * j)) & 0xffu);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = p0 + j < npix ? (int)win[p0 + j] : -1;
    }
}

// per block and rank: number of pixels that rank won
template <int NW>
__global__ __launch_bounds__(256) void combine_count(const uint8_t* __restrict__ win, size_t npix, int n_ranks, uint32_t* __restrict__ counts) {
    __shared__ unsigned long long sh[4][CB_WORDS];
    int r[4];
    cb_load4(win, npix, (size_t)blockIdx.x * CB_PX + 4 * threadIdx.x, r);
    CbCounts excl, total;
    cb_block_scan<NW>(r, excl, total, sh);
    if ((int)threadIdx.x < n_ranks) counts[(size_t)blockIdx.x * n_ranks + threadIdx.x] = cb_field(total, threadIdx.x);
}

// exclusive scan of one rank's column down the blocks (in place: counts -> offsets) + its total; one workgroup PER RANK
__global__ __launch_bounds__(1024) void combine_scan(uint32_t* __restrict__ counts, int nblocks, int n_ranks, unsigned long long* __restrict__ totals) {
    __shared__ unsigned long long part[1024];
    const int t = threadIdx.x, nt = blockDim.x, s = blockIdx.x;
    const int per = (nblocks + nt - 1) / nt;
    unsigned long long sum = 0;
    for (int k = 0; k < per; ++k) {
        const int b = t * per + k;
        if (b < nblocks) sum += counts[(size_t)b * n_ranks + s];
    }
    part[t] = sum;
    __syncthreads();
    // Hillis-Steele inclusive scan over the 1024 partial sums
    for (int d = 1; d < nt; d <<= 1) {
        const unsigned long long v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    if (t == nt - 1) totals[s] = part[t];
    unsigned long long run = part[t] - sum;
    for (int k = 0; k < per; ++k) {
        const int b = t * per + k;
        if (b < nblocks) {
            const uint32_t v = counts[(size_t)b * n_ranks + s];
            counts[(size_t)b * n_ranks + s] = (uint32_t)run;
            run += v;
        }
    }
}

// PACK: out[pos(p)] = src[p] for the pixels `rank` won; !PACK: dst[p] = bufs[rank of p][pos(p)] for every pixel `rank`
// (the receiver) did NOT win itself.  Rows of `width` floats (12-byte moves for width 3).
template <bool PACK, int NW>
__global__ __launch_bounds__(256) void combine_move(const uint8_t* __restrict__ win, size_t npix, int n_ranks, int rank,
                                                    const uint32_t* __restrict__ offsets, const float* __restrict__ src,
                                                    const float* const* __restrict__ bufs, int width, float* __restrict__ dst) {
    __shared__ unsigned long long sh[4][CB_WORDS];
    __shared__ uint32_t base[CB_MAXR];
    if ((int)threadIdx.x < n_ranks) base[threadIdx.x] = offsets[(size_t)blockIdx.x * n_ranks + threadIdx.x];
    const size_t p0 = (size_t)blockIdx.x * CB_PX + 4 * threadIdx.x;
    int r[4];
    cb_load4(win, npix, p0, r);
    CbCounts excl, total;
    cb_block_scan<NW>(r, excl, total, sh);     // (its barrier also publishes `base`)
    uint32_t seen[4] = {0, 0, 0, 0};           // pixels of the same rank earlier in this thread's four
#pragma unroll
    for (int j = 1; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < j; ++i) seen[j] += (r[i] == r[j]) ? 1u : 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rr = r[j];
        if (rr < 0 || (PACK ? rr != rank : rr == rank)) continue;
        const size_t pos = (size_t)base[rr] + cb_field(excl, rr) + seen[j], p = p0 + j;
        if (width == 3) {
            struct __attribute__((packed, aligned(4))) Row3 { float v[3]; };
            if (PACK) *reinterpret_cast<Row3*>(dst + pos * 3) = *reinterpret_cast<const Row3*>(src + p * 3);
            else *reinterpret_cast<Row3*>(dst + p * 3) = *reinterpret_cast<const Row3*>(bufs[rr] + pos * 3);
        } else if (PACK) {
            for (int c = 0; c < width; ++c) dst[pos * width + c] = src[p * width + c];
        } else {
            const float* b = bufs[rr];
            for (int c = 0; c < width; ++c) dst[p * width + c] = b[pos * width + c];
        }
    }
}

// ---------------------------------------------------------------- frame -> float32 (stacks without
// Laplacian levels: the frame itself is the base, pyramid.py:126 img.astype(float_type))
template <typename TIn>
__global__ void frame_to_f32(const TIn* __restrict__ src, size_t n, float* __restrict__ dst) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = to_f32(src[i]);
}

// ---------------------------------------------------------------- synthetic frames
template <typename T>
__global__ void synth_frames(T* __restrict__ out, int H, int W, int f0, int nf, int N,
                             uint32_t seed, int scale) {
    size_t per = (size_t)H * W * 3;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per * nf) return;
    int fi = (int)(i / per);
    size_t r = i - (size_t)fi * per;
    int c = (int)(r % 3);
    size_t px = r / 3;
    int x = (int)(px % W), y = (int)(px / W);
    int f = f0 + fi;
    uint32_t hsh = lowbias32(seed ^ ((uint32_t)f * 0x9E3779B1U) ^ ((uint32_t)y * 0x85EBCA77U) ^
                             ((uint32_t)x * 0xC2B2AE3DU) ^ (uint32_t)c);
    int noise = (int)(hsh >> 24) - 128;
    int band = (int)(((int64_t)y * N) / H);
    int d = band - f;
    d = d < 0 ? -d : d;
    int amp = 64 >> (d < 6 ? d : 6);
    int base = ((3 * x + 5 * y + 17 * c) & 127) + 64;
    int v = base + ((noise * amp) >> 7);
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    out[i] = (T)(v * scale);
}

}  // namespace mi
