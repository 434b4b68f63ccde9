// kernels_tiled.hpp -- LDS-tiled fused level kernel (production path, gfx950).
//
// One launch handles ONE pyramid level of a BATCH of frames.  Each 256-thread
// workgroup owns a TH x TW tile of level l and walks the batch frame by frame:
//
//   stage   G_l patch (tile + 6 halo) of frame b      global -> LDS  (next frame's
//           patch is already in flight in registers while this one is computed)
//   reduce  G_{l+1} patch (tile/2 + 2 halo) = 5x5 stencil at even coordinates
//           LDS -> LDS, tile centre also stored to global (input of the next level)
//   lapq    Laplacian = G_l - 4*expand(G_{l+1}) on tile + 2 halo, Q = gray(lap)^2
//           LDS -> LDS (Q) / registers (the thread's own 2x2 quad of lap)
//   energy  E = 5x5 stencil of Q; running first-max (E, idx, lap) in REGISTERS
//
// and writes the running state once per batch.  Per frame the kernel reads G_l
// once and writes G_{l+1} once; the Laplacian, Q and E never touch HBM, and the
// selection state costs 20 B/pixel per BATCH instead of per frame.
//
// LDS images are pixel-interleaved exactly like global memory (BGRBGR...), so
// staging is a linear copy and every thread works on all three channels of its
// pixels (which the gray conversion needs anyway).
//
// Arithmetic is identical, operation by operation, to kernels_simple.hpp and to
// oracle/: taps of one output are applied in row-major order; interleaving the
// chains of different outputs does not change any of them.
#pragma once
#include "common.hpp"

namespace mi {

struct LevelArgs {
    const void* src;        // level-l images of the batch (TIn for l == 0, else f32)
    size_t src_stride;      // bytes between consecutive frames
    float* gnext;           // level-(l+1) images of the batch (written)
    size_t gnext_stride;    // floats between consecutive frames
    int nframes;
    int h, w, hn, wn;
    int tiles_x, tiles_y;
    float* best_e;          // running state of level l
    float* best_lap;
    int32_t* best_idx;
    int first;              // state holds nothing yet
    int frame_idx0;         // global index of the batch's first frame
    K25 K;
};

template <int TH_, int TW_>
struct TileGeom {
    static constexpr int TH = TH_, TW = TW_;
    static constexpr int NT = 256;                    // threads per workgroup
    static constexpr int GH = TH + 12, GW = TW + 12;  // G_l patch (pixels)
    static constexpr int GS = GW * 3;                 // row stride (floats)
    static constexpr int NH = TH / 2 + 4, NW = TW / 2 + 4;
    static constexpr int NS = NW * 3;
    static constexpr int QH = TH + 4, QW = TW + 4;
    static constexpr int QS = QW;
    static constexpr int G_ELEMS = GH * GS;
    static constexpr int NPRE = (G_ELEMS + NT - 1) / NT;  // staged elements per thread
    static constexpr int NQ = (TH / 2) * (TW / 2) / NT;   // owned 2x2 quads per thread
    static constexpr int LDS_FLOATS = GH * GS + NH * NS + QH * QS;
    static_assert((TH / 2) * (TW / 2) % NT == 0, "tile must split into whole quads per thread");
};

// clamp(reflect101(v)) -- cells whose overshoot exceeds 2 are never consumed by a
// valid output; the clamp only keeps their addresses inside the arrays.
__device__ __forceinline__ int map_clamp(int v, int n) {
    int r = v < 0 ? -v : v;
    r = r >= n ? 2 * n - 2 - r : r;
    return r < 0 ? 0 : (r >= n ? n - 1 : r);
}
// index map of the expand source (REFLECT101 acts on the zero-stuffed grid):
// V[-1] = G[1], V[n] = G[n-1]
__device__ __forceinline__ int map_expand_src(int i, int n) {
    int r = i < 0 ? -i : (i >= n ? 2 * n - 1 - i : i);
    return r < 0 ? 0 : (r >= n ? n - 1 : r);
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <typename TIn, bool FMA, int TH, int TW>
__global__ __launch_bounds__(256) void level_fused(LevelArgs a) {
    using G = TileGeom<TH, TW>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sG = smem;
    float* sN = sG + G::GH * G::GS;
    float* sQ = sN + G::NH * G::NS;

    // XCD-aware tile order: block b runs on XCD b % 8; give every XCD one
    // contiguous band of tiles so neighbouring tiles share an L2.
    const int ntiles = a.tiles_x * a.tiles_y;
    const int per_xcd = (ntiles + 7) >> 3;
    const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= ntiles) return;
    const int tyi = tile / a.tiles_x, txi = tile - tyi * a.tiles_x;
    const int y0 = tyi * TH, x0 = txi * TW;
    const int tid = threadIdx.x;
    const int h = a.h, w = a.w, hn = a.hn, wn = a.wn;
    const bool interior = (y0 >= 6) && (x0 >= 6) && (y0 + TH + 6 <= h) && (x0 + TW + 6 <= w);

    // ---- per-thread constants of the staging copy: global element offsets
    int goff[G::NPRE];
#pragma unroll
    for (int n = 0; n < G::NPRE; ++n) {
        int e = tid + n * G::NT;
        int r = e / G::GS, k = e - r * G::GS;
        int col = k / 3, c = k - col * 3;
        int gy = map_clamp(y0 - 6 + r, h), gx = map_clamp(x0 - 6 + col, w);
        goff[n] = (e < G::G_ELEMS) ? (gy * w + gx) * 3 + c : -1;
    }

    // ---- running state of the thread's own quads
    float bE[G::NQ][4], bL[G::NQ][4][3];
    int bI[G::NQ][4];
    int qoy[G::NQ], qox[G::NQ];
#pragma unroll
    for (int q = 0; q < G::NQ; ++q) {
        int qi = tid + q * G::NT;
        qoy[q] = qi / (TW / 2);
        qox[q] = qi - qoy[q] * (TW / 2);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int y = y0 + 2 * qoy[q] + (p >> 1), x = x0 + 2 * qox[q] + (p & 1);
            bool valid = y < h && x < w;
            if (!a.first && valid) {
                size_t px = (size_t)y * w + x;
                bE[q][p] = a.best_e[px];
                bI[q][p] = a.best_idx[px];
                bL[q][p][0] = a.best_lap[px * 3 + 0];
                bL[q][p][1] = a.best_lap[px * 3 + 1];
                bL[q][p][2] = a.best_lap[px * 3 + 2];
            } else {
                bE[q][p] = -1.0f;  // every energy is >= 0: the first frame always wins
                bI[q][p] = -1;
                bL[q][p][0] = bL[q][p][1] = bL[q][p][2] = 0.f;
            }
        }
    }

    float pre[G::NPRE];
    auto prefetch = [&](int b) {
        const TIn* fr = (const TIn*)((const char*)a.src + (size_t)b * a.src_stride);
#pragma unroll
        for (int n = 0; n < G::NPRE; ++n) pre[n] = goff[n] >= 0 ? to_f32(fr[goff[n]]) : 0.f;
    };
    prefetch(0);

    for (int b = 0; b < a.nframes; ++b) {
        // ---------------- stage
#pragma unroll
        for (int n = 0; n < G::NPRE; ++n) {
            int e = tid + n * G::NT;
            if (e < G::G_ELEMS) sG[e] = pre[n];
        }
        __syncthreads();
        if (b + 1 < a.nframes) prefetch(b + 1);

        // ---------------- reduce: 2x2 output blocks, all three channels
        {
            constexpr int BY = G::NH / 2, BX = G::NW / 2;
            for (int it = tid; it < BY * BX; it += G::NT) {
                int by = it / BX, bx = it - by * BX;
                int ri = 2 * by, rj = 2 * bx;  // local sN coordinates of the block
                float acc[2][2][3];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int v = 0; v < 2; ++v) acc[u][v][0] = acc[u][v][1] = acc[u][v][2] = 0.f;
                if (interior) {
                    // input rows 2ri .. 2ri+6, pixels 2rj .. 2rj+6 (21 floats per row)
#pragma unroll
                    for (int rr = 0; rr < 7; ++rr) {
                        const float* row = sG + (2 * ri + rr) * G::GS + 2 * rj * 3;
                        float v[21];
#pragma unroll
                        for (int t = 0; t < 21; ++t) v[t] = row[t];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            int ty = rr - 2 * u;  // tap row of output row u
                            if (ty < 0 || ty > 4) continue;
#pragma unroll
                            for (int vv = 0; vv < 2; ++vv)
#pragma unroll
                                for (int tx = 0; tx < 5; ++tx) {
                                    float k = a.K.k[ty * 5 + tx];
#pragma unroll
                                    for (int c = 0; c < 3; ++c)
                                        acc[u][vv][c] = mac<FMA>(k, v[(2 * vv + tx) * 3 + c], acc[u][vv][c]);
                                }
                        }
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int vv = 0; vv < 2; ++vv) {
                            int im = map_expand_src(y0 / 2 - 2 + ri + u, hn);
                            int jm = map_expand_src(x0 / 2 - 2 + rj + vv, wn);
                            for (int ty = 0; ty < 5; ++ty) {
                                int r = clampi(2 * im - 2 + ty - (y0 - 6), 0, G::GH - 1);
                                // rows beyond the image already hold reflected data (staging)
                                for (int tx = 0; tx < 5; ++tx) {
                                    int cc = clampi(2 * jm - 2 + tx - (x0 - 6), 0, G::GW - 1);
                                    float k = a.K.k[ty * 5 + tx];
                                    const float* p = sG + r * G::GS + cc * 3;
#pragma unroll
                                    for (int c = 0; c < 3; ++c) acc[u][vv][c] = mac<FMA>(k, p[c], acc[u][vv][c]);
                                }
                            }
                        }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int vv = 0; vv < 2; ++vv)
#pragma unroll
                        for (int c = 0; c < 3; ++c) sN[(ri + u) * G::NS + (rj + vv) * 3 + c] = acc[u][vv][c];
            }
        }
        __syncthreads();

        // ---------------- store the tile centre of G_{l+1}
        {
            float* gout = a.gnext + (size_t)b * a.gnext_stride;
            const int i0 = y0 / 2, j0 = x0 / 2;
            constexpr int CW = (TW / 2) * 3;
            for (int e = tid; e < (TH / 2) * CW; e += G::NT) {
                int r = e / CW, k = e - r * CW;
                int i = i0 + r, j = j0 + k / 3;
                if (i < hn && j < wn) gout[((size_t)i * wn + j0) * 3 + k] = sN[(r + 2) * G::NS + 6 + k];
            }
        }

        // ---------------- laplacian + Q on (TH+4) x (TW+4), as 2x2 quads
        float myLap[G::NQ][4][3];
        {
            constexpr int QY = TH / 2 + 2, QX = TW / 2 + 2;
            auto do_quad = [&](int qy, int qx, float (*keep)[3]) {
                // rows: even cell then odd cell of the quad, mapped into the image
                int ye, yo, xe, xo;
                if (interior) {
                    ye = y0 - 2 + 2 * qy; yo = ye + 1;
                    xe = x0 - 2 + 2 * qx; xo = xe + 1;
                } else {
                    ye = map_clamp(y0 - 2 + 2 * qy, h); yo = map_clamp(y0 - 1 + 2 * qy, h);
                    xe = map_clamp(x0 - 2 + 2 * qx, w); xo = map_clamp(x0 - 1 + 2 * qx, w);
                }
                // expand-source rows/cols (local sN coordinates)
                int re = (ye >> 1) - (y0 / 2 - 2), ro = ((yo - 1) >> 1) - (y0 / 2 - 2);
                int ce = (xe >> 1) - (x0 / 2 - 2), co = ((xo - 1) >> 1) - (x0 / 2 - 2);
                if (!interior) {
                    re = clampi(re, 1, G::NH - 2); ro = clampi(ro, 0, G::NH - 2);
                    ce = clampi(ce, 1, G::NW - 2); co = clampi(co, 0, G::NW - 2);
                }
                // G_l cells (local sG coordinates)
                int gre = ye - (y0 - 6), gro = yo - (y0 - 6), gce = xe - (x0 - 6), gco = xo - (x0 - 6);
                if (!interior) {
                    gre = clampi(gre, 0, G::GH - 1); gro = clampi(gro, 0, G::GH - 1);
                    gce = clampi(gce, 0, G::GW - 1); gco = clampi(gco, 0, G::GW - 1);
                }
                float see[3] = {0, 0, 0}, seo[3] = {0, 0, 0}, soe[3] = {0, 0, 0}, soo[3] = {0, 0, 0};
                const K25& K = a.K;
                // (even row, even col): taps ty in {0,2,4} x tx in {0,2,4}
                // (even row, odd  col): ty in {0,2,4} x tx in {1,3}
#pragma unroll
                for (int ar = 0; ar < 3; ++ar) {
                    const float* nrow = sN + (re - 1 + ar) * G::NS;
#pragma unroll
                    for (int ac = 0; ac < 3; ++ac) {
                        float k = K.k[(2 * ar) * 5 + 2 * ac];
                        const float* p = nrow + (ce - 1 + ac) * 3;
#pragma unroll
                        for (int c = 0; c < 3; ++c) see[c] = mac<FMA>(k, p[c], see[c]);
                    }
#pragma unroll
                    for (int ac = 0; ac < 2; ++ac) {
                        float k = K.k[(2 * ar) * 5 + 2 * ac + 1];
                        const float* p = nrow + (co + ac) * 3;
#pragma unroll
                        for (int c = 0; c < 3; ++c) seo[c] = mac<FMA>(k, p[c], seo[c]);
                    }
                }
                // (odd row, *): ty in {1,3}
#pragma unroll
                for (int ar = 0; ar < 2; ++ar) {
                    const float* nrow = sN + (ro + ar) * G::NS;
#pragma unroll
                    for (int ac = 0; ac < 3; ++ac) {
                        float k = K.k[(2 * ar + 1) * 5 + 2 * ac];
                        const float* p = nrow + (ce - 1 + ac) * 3;
#pragma unroll
                        for (int c = 0; c < 3; ++c) soe[c] = mac<FMA>(k, p[c], soe[c]);
                    }
#pragma unroll
                    for (int ac = 0; ac < 2; ++ac) {
                        float k = K.k[(2 * ar + 1) * 5 + 2 * ac + 1];
                        const float* p = nrow + (co + ac) * 3;
#pragma unroll
                        for (int c = 0; c < 3; ++c) soo[c] = mac<FMA>(k, p[c], soo[c]);
                    }
                }
                const float* gee = sG + gre * G::GS + gce * 3;
                const float* geo = sG + gre * G::GS + gco * 3;
                const float* goe = sG + gro * G::GS + gce * 3;
                const float* goo = sG + gro * G::GS + gco * 3;
                float l[4][3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    l[0][c] = gee[c] - 4.0f * see[c];
                    l[1][c] = geo[c] - 4.0f * seo[c];
                    l[2][c] = goe[c] - 4.0f * soe[c];
                    l[3][c] = goo[c] - 4.0f * soo[c];
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float gr = gray_of<FMA>(l[p][0], l[p][1], l[p][2]);
                    sQ[(2 * qy + (p >> 1)) * G::QS + 2 * qx + (p & 1)] = gr * gr;
                    if (keep) {
                        keep[p][0] = l[p][0];
                        keep[p][1] = l[p][1];
                        keep[p][2] = l[p][2];
                    }
                }
            };
            // own quads (tile interior) ...
#pragma unroll
            for (int q = 0; q < G::NQ; ++q) do_quad(qoy[q] + 1, qox[q] + 1, myLap[q]);
            // ... and the halo ring, spread over the first threads
            constexpr int RING = QY * QX - (TH / 2) * (TW / 2);
            for (int it = tid; it < RING; it += G::NT) {
                int qy, qx;
                if (it < QX) { qy = 0; qx = it; }
                else if (it < 2 * QX) { qy = QY - 1; qx = it - QX; }
                else {
                    int s = it - 2 * QX;  // left/right columns, rows 1..QY-2
                    qy = 1 + (s >> 1);
                    qx = (s & 1) ? QX - 1 : 0;
                }
                do_quad(qy, qx, nullptr);
            }
        }
        __syncthreads();

        // ---------------- energy of the own quads + running first-max
        {
            const int fidx = a.frame_idx0 + b;
#pragma unroll
            for (int q = 0; q < G::NQ; ++q) {
                float e[4] = {0.f, 0.f, 0.f, 0.f};
                const float* base = sQ + (2 * qoy[q]) * G::QS + 2 * qox[q];
#pragma unroll
                for (int rr = 0; rr < 6; ++rr) {
                    float v[6];
#pragma unroll
                    for (int t = 0; t < 6; ++t) v[t] = base[rr * G::QS + t];
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy) {
                        int ty = rr - dy;
                        if (ty < 0 || ty > 4) continue;
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                            for (int tx = 0; tx < 5; ++tx)
                                e[dy * 2 + dx] = mac<FMA>(a.K.k[ty * 5 + tx], v[dx + tx], e[dy * 2 + dx]);
                    }
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    if (e[p] > bE[q][p]) {
                        bE[q][p] = e[p];
                        bI[q][p] = fidx;
                        bL[q][p][0] = myLap[q][p][0] + 0.0f;  // -0 -> +0 (np.where sum)
                        bL[q][p][1] = myLap[q][p][1] + 0.0f;
                        bL[q][p][2] = myLap[q][p][2] + 0.0f;
                    }
                }
            }
        }
        // no barrier needed here: the next writes to sG happen after every thread
        // passed the barrier above (sG/sN are only read before it), and sQ is
        // rewritten only after the next iteration's two barriers.
    }

    // ---- write the running state back
#pragma unroll
    for (int q = 0; q < G::NQ; ++q)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int y = y0 + 2 * qoy[q] + (p >> 1), x = x0 + 2 * qox[q] + (p & 1);
            if (y < h && x < w) {
                size_t px = (size_t)y * w + x;
                a.best_e[px] = bE[q][p];
                a.best_idx[px] = bI[q][p];
                a.best_lap[px * 3 + 0] = bL[q][p][0];
                a.best_lap[px * 3 + 1] = bL[q][p][1];
                a.best_lap[px * 3 + 2] = bL[q][p][2];
            }
        }
}

// ---------------------------------------------------------------- batched base level
template <bool FMA>
__global__ void base_gray_hist_batch(const float* __restrict__ bases, size_t base_stride, int npix,
                                     int nlevels, int32_t* __restrict__ lev,
                                     uint32_t* __restrict__ cnt) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int f = blockIdx.y;
    if (i >= npix) return;
    const float* base = bases + (size_t)f * base_stride;
    float gr = gray_of<FMA>(base[3 * i], base[3 * i + 1], base[3 * i + 2]);
    int l = (int)gr;
    l = l < 0 ? 0 : (l >= nlevels ? nlevels - 1 : l);
    lev[(size_t)f * npix + i] = l;
    atomicAdd(&cnt[(size_t)f * nlevels + l], 1u);
}

__global__ void base_logp_batch(const uint32_t* __restrict__ cnt, int nlevels, int npix,
                                float* __restrict__ logp) {
    int l = blockIdx.x * blockDim.x + threadIdx.x;
    int f = blockIdx.y;
    if (l >= nlevels) return;
    uint32_t c = cnt[(size_t)f * nlevels + l];
    float v = 0.f;
    if (c) {
        float p = (float)((double)(float)c / (double)npix);
        v = (float)log((double)p);
    }
    logp[(size_t)f * nlevels + l] = v;
}

}  // namespace mi
