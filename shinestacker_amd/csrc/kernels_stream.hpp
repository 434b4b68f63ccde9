// kernels_stream.hpp -- register-streaming level kernel for the INTERIOR of a pyramid level.
//
// One wave (64 lanes, no workgroup barriers) owns a strip of 256 columns -- lane m holds the 4
// pixels x0 + 4m .. x0 + 4m + 3 of the current row in registers, = 2 pixels of G_{l+1} -- and
// marches down a segment of rows, frame after frame of the batch.  Nothing of the stencil data
// path goes through LDS or HBM:
//
//   * horizontal neighbours come from the adjacent lanes by DPP wave shifts (v_mov_b32_dpp
//     wave_shr:1 / wave_shl:1): 9 per G row, 6 per G_{l+1} row, 4 per Q row;
//   * vertical reuse is by PARTIAL ACCUMULATORS instead of row windows: an arriving row adds
//     tap-row t to every output row that sees it as its t-th row.  Rows arrive in order, so each
//     output is still the row-major fma chain from 0 of the reference-order 5x5 filter --
//     bit-identical to kernels_tiled.hpp / kernels_simple.hpp / oracle -- with 3 (reduce) or
//     6 (energy) live accumulators per pixel instead of a 5-row window;
//   * two outputs that take the SAME source value with different coefficients (vertically
//     adjacent output rows) share one v_pk_fma_f32 (source broadcast by op_sel), which halves
//     the VALU instruction count of the stencils (packed fp32 is the only way to the 78 Tfma/s
//     vector peak of gfx950; plain v_fma_f32 tops out at 39).
//
// Per step i (two rows of G_l in, one row of G_{l+1} out, two rows of selection):
//   rows 2i+1, 2i+2 of G_l arrive (prefetched one step ahead)
//   reduce    N[i] complete (accumulators A = N[i], B = N[i+1], C = N[i+2])     -> gnext
//   expand    rows 2i-2 (even) and 2i-1 (odd) from the window N[i-2], N[i-1], N[i]
//   lap, Q    lap = G - 4*expand with the own pixels of G rows 2i-2, 2i-1 (LDS delay line,
//             lane-private), Q = gray(lap)^2
//   energy    Q rows 2i-2, 2i-1 feed the accumulator pairs (E[2i-4],E[2i-3]), (E[2i-2],E[2i-1]),
//             (E[2i],E[2i+1]); the first pair is complete
//   select    E rows 2i-4, 2i-3 against the running state in HBM (E read once per frame,
//             state written only where the frame wins; lap of the winners from the lane-private
//             LDS delay line of the previous step)
//
// Coordinates never leave the image (the launch covers [iy0, iy1) x [ix0, ix1), at least 8
// pixels away from every edge); the frame of border pixels is done by level_fused<INTERIOR=false>.
#pragma once
#include "kernels_tiled.hpp"

namespace mi {

constexpr int ST_UW = 240;  // useful pixels per strip: lanes 2..61

__device__ __forceinline__ float lane_prev(float v) {  // lane m <- lane m-1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_next(float v) {  // lane m <- lane m+1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

// two chains advance one tap each on the SAME source value:
// (acc.x, acc.y) <- (k.x*x + acc.x, k.y*x + acc.y), one v_pk_fma_f32 (source broadcast by op_sel)
template <bool FMA>
__device__ __forceinline__ v2f mac2(v2f k, float x, v2f acc) {
    const v2f xx = {x, x};
    if constexpr (FMA) return __builtin_elementwise_fma(k, xx, acc);
    else {
        const v2f p = k * xx;  // contraction is off for this TU
        return acc + p;
    }
}

typedef float v4f_u __attribute__((ext_vector_type(4), aligned(4)));
typedef float v2f_u __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t v3u __attribute__((ext_vector_type(3)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));

// the 12 values (4 pixels x BGR) a lane holds of one row, as loaded
template <typename TIn> struct RawRow;
template <> struct RawRow<float> { v4f q[3]; };
template <> struct RawRow<uint8_t> { uint32_t d[3]; };
template <> struct RawRow<uint16_t> { uint32_t d[6]; };

// VEC: the row starts 16-byte (f32) / 4-byte (u8, u16) aligned for every lane
template <bool VEC>
__device__ __forceinline__ void load_raw(const float* p, RawRow<float>& r) {
    if constexpr (VEC) {
        const v4f* q = (const v4f*)p;
        r.q[0] = __builtin_nontemporal_load(q);
        r.q[1] = __builtin_nontemporal_load(q + 1);
        r.q[2] = __builtin_nontemporal_load(q + 2);
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) r.q[k] = v4f{p[4 * k], p[4 * k + 1], p[4 * k + 2], p[4 * k + 3]};
    }
}
template <bool VEC>
__device__ __forceinline__ void load_raw(const uint8_t* p, RawRow<uint8_t>& r) {
    if constexpr (VEC) {
        const uint32_t* q = (const uint32_t*)p;
#pragma unroll
        for (int k = 0; k < 3; ++k) r.d[k] = __builtin_nontemporal_load(q + k);
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            r.d[k] = (uint32_t)p[4 * k] | ((uint32_t)p[4 * k + 1] << 8) | ((uint32_t)p[4 * k + 2] << 16) |
                     ((uint32_t)p[4 * k + 3] << 24);
    }
}
template <bool VEC>
__device__ __forceinline__ void load_raw(const uint16_t* p, RawRow<uint16_t>& r) {
    if constexpr (VEC) {
        const uint32_t* q = (const uint32_t*)p;
#pragma unroll
        for (int k = 0; k < 6; ++k) r.d[k] = __builtin_nontemporal_load(q + k);
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) r.d[k] = (uint32_t)p[2 * k] | ((uint32_t)p[2 * k + 1] << 16);
    }
}
__device__ __forceinline__ void unpack(const RawRow<float>& r, float* g) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { g[4 * k] = r.q[k].x; g[4 * k + 1] = r.q[k].y; g[4 * k + 2] = r.q[k].z; g[4 * k + 3] = r.q[k].w; }
}
__device__ __forceinline__ void unpack(const RawRow<uint8_t>& r, float* g) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int b = 0; b < 4; ++b) g[4 * k + b] = (float)((r.d[k] >> (8 * b)) & 0xffu);
}
__device__ __forceinline__ void unpack(const RawRow<uint16_t>& r, float* g) {
#pragma unroll
    for (int k = 0; k < 6; ++k) { g[2 * k] = (float)(r.d[k] & 0xffffu); g[2 * k + 1] = (float)(r.d[k] >> 16); }
}

struct StreamGeom {
    int nstrips, nsegs, seg;
};

// block = one wave.  blockIdx.x = segment * nstrips + strip.
template <typename TIn, bool FMA, bool VEC, int PF>
__global__ __launch_bounds__(64, 2) void level_stream(LevelArgs a, StreamGeom sg) {
    __shared__ v4f sGd[4][3][64];  // own pixels of the last G rows (slot = row & 3)
    __shared__ v4f sLd[2][3][64];  // own pixels of the lap rows of the previous step
    const int lane = threadIdx.x;
    const int strip = blockIdx.x % sg.nstrips, segi = blockIdx.x / sg.nstrips;
    const int w = a.w, wn = a.wn;
    const int ys = a.iy0 + segi * sg.seg;
    const int ye = min(ys + sg.seg, a.iy1);
    int x0 = a.ix0 + strip * ST_UW - 8 + 4 * lane;
    const bool useful = lane >= 2 && lane <= 61 && x0 < a.ix1;
    x0 = min(x0, w - 4);  // lanes hanging over the image re-read its last pixels (never consumed)
    const int j0 = x0 >> 1;
    const K6 K = a.K;
    const int i0 = (ys >> 1) - 4, i1 = (ye >> 1) + 1;  // steps

    for (int f = 0; f < a.nframes; ++f) {
        const TIn* src = (const TIn*)((const char*)a.src + (size_t)f * a.src_stride) + (size_t)x0 * 3;
        float* gout = a.gnext + (size_t)f * a.gnext_stride;
        const int fidx = a.frame_idx0 + f;
        const bool fresh = a.first && f == 0;

        v2f AB[2][3];
        float C[2][3];
        float Nw[3][4][3];
        v2f EP[3][4];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int c = 0; c < 3; ++c) { AB[p][c] = v2f{0.f, 0.f}; C[p][c] = 0.f; }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                Nw[r][q][0] = Nw[r][q][1] = Nw[r][q][2] = 0.f;
                EP[r][q] = v2f{0.f, 0.f};
            }
        // the delay lines start with finite values (warm-up results are discarded, but must not trap)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int k = 0; k < 3; ++k) sGd[s][k][lane] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int k = 0; k < 3; ++k) sLd[s][k][lane] = v4f{0.f, 0.f, 0.f, 0.f};

        // rows 2i+1, 2i+2 of steps i .. i+PF-1 in flight, and the running energies of the rows
        // steps i .. i+PF-1 select (rows 2i-4, 2i-3): loaded PF steps before their use, without
        // control flow around the loads (rows clamped into the image / the segment; validity is
        // decided at the selection), so the wait for them sits at their use
        RawRow<TIn> rq[PF][2];
        v4f eq[PF][2];
        auto issue = [&](int step, RawRow<TIn>* r2, v4f* e2) {
            const int ic = min(step, i1);
            if (!(a.ablate & 16)) {
                load_raw<VEC>(src + (size_t)(2 * ic + 1) * w * 3, r2[0]);
                load_raw<VEC>(src + (size_t)(2 * ic + 2) * w * 3, r2[1]);
            }
            if (!(a.ablate & 64))
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int y = min(max(2 * step - 4 + r, ys), ye - 1);
                const float* pe = a.best_e + (size_t)y * w + x0;
                if constexpr (VEC) e2[r] = *(const v4f*)pe;
                else e2[r] = v4f{pe[0], pe[1], pe[2], pe[3]};
            }
        };
#pragma unroll
        for (int k = 0; k < PF; ++k) issue(i0 + k, rq[k], eq[k]);

        for (int i = i0; i <= ((a.ablate & 128) ? i0 + 1 : i1); ++i) {
            float g[2][12];
            unpack(rq[0][0], g[0]);
            unpack(rq[0][1], g[1]);
            const v4f eo[2] = {eq[0][0], eq[0][1]};
#pragma unroll
            for (int k = 0; k + 1 < PF; ++k) {
                rq[k][0] = rq[k + 1][0]; rq[k][1] = rq[k + 1][1];
                eq[k][0] = eq[k + 1][0]; eq[k][1] = eq[k + 1][1];
            }
            issue(i + PF, rq[PF - 1], eq[PF - 1]);

            // ---------------- reduce
            if (!(a.ablate & 1))
#pragma unroll
            for (int r = 0; r < 2; ++r) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float s[7];
                    s[0] = lane_prev(g[r][2 * 3 + c]);
                    s[1] = lane_prev(g[r][3 * 3 + c]);
#pragma unroll
                    for (int p = 0; p < 4; ++p) s[2 + p] = g[r][p * 3 + c];
                    s[6] = lane_next(g[r][c]);
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int tx = 0; tx < 5; ++tx) {
                            const float v = s[2 * p + tx];
                            if (r == 0) {  // odd row 2i+1: tap row 3 of A = N[i], tap row 1 of B = N[i+1]
                                AB[p][c] = mac2<FMA>(v2f{K(3, tx), K(1, tx)}, v, AB[p][c]);
                            } else {       // even row 2i+2: tap rows 4, 2, 0 of A, B, C = N[i+2]
                                AB[p][c] = mac2<FMA>(v2f{K(4, tx), K(2, tx)}, v, AB[p][c]);
                                C[p][c] = mac<FMA>(K(0, tx), v, C[p][c]);
                            }
                        }
                }
            }
            float Nn[2][3];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    Nn[p][c] = AB[p][c].x;
                    AB[p][c] = v2f{AB[p][c].y, C[p][c]};
                    C[p][c] = 0.f;
                }
            if (useful && 2 * i >= ys && 2 * i < ye && !(a.ablate & 2)) {
                float* o = gout + ((size_t)i * wn + j0) * 3;
                if constexpr (VEC) {
                    // VEC implies w % 4 == 0, hence wn even and 8-byte aligned rows
                    *(v2f*)o = v2f{Nn[0][0], Nn[0][1]};
                    *(v2f*)(o + 2) = v2f{Nn[0][2], Nn[1][0]};
                    *(v2f*)(o + 4) = v2f{Nn[1][1], Nn[1][2]};
                } else {
#pragma unroll
                    for (int k = 0; k < 6; ++k) o[k] = Nn[k / 3][k % 3];
                }
            }
            // ---------------- window of G_{l+1} rows i-2, i-1, i with the neighbours' columns
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 3; ++c) { Nw[0][q][c] = Nw[1][q][c]; Nw[1][q][c] = Nw[2][q][c]; }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                Nw[2][0][c] = lane_prev(Nn[1][c]);
                Nw[2][1][c] = Nn[0][c];
                Nw[2][2][c] = Nn[1][c];
                Nw[2][3][c] = lane_next(Nn[0][c]);
            }
            // ---------------- expand rows 2i-2 (.x) and 2i-1 (.y), Laplacian, Q
            float lap[2][4][3], Q[2][4];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int p = 0; p < 4; ++p) { lap[r][p][0] = lap[r][p][1] = lap[r][p][2] = 0.f; Q[r][p] = 0.f; }
            if (!(a.ablate & 4)) {
                const int sa = (2 * i - 2) & 3, sb = (2 * i - 1) & 3;
                v4f ga[3], gb[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) { ga[k] = sGd[sa][k][lane]; gb[k] = sGd[sb][k][lane]; }
                float gd[2][12];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    gd[0][4 * k] = ga[k].x; gd[0][4 * k + 1] = ga[k].y; gd[0][4 * k + 2] = ga[k].z; gd[0][4 * k + 3] = ga[k].w;
                    gd[1][4 * k] = gb[k].x; gd[1][4 * k + 1] = gb[k].y; gd[1][4 * k + 2] = gb[k].z; gd[1][4 * k + 3] = gb[k].w;
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const bool oddx = p & 1;
                    const int col0 = oddx ? (p + 1) / 2 : p / 2;  // first window column of the taps
                    const int ntap = oddx ? 2 : 3;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        v2f S = {0.f, 0.f};
#pragma unroll
                        for (int ac = 0; ac < ntap; ++ac) {
                            const int tx = oddx ? 2 * ac + 1 : 2 * ac;
                            S.x = mac<FMA>(K(0, tx), Nw[0][col0 + ac][c], S.x);
                        }
#pragma unroll
                        for (int ac = 0; ac < ntap; ++ac) {
                            const int tx = oddx ? 2 * ac + 1 : 2 * ac;
                            S = mac2<FMA>(v2f{K(2, tx), K(1, tx)}, Nw[1][col0 + ac][c], S);
                        }
#pragma unroll
                        for (int ac = 0; ac < ntap; ++ac) {
                            const int tx = oddx ? 2 * ac + 1 : 2 * ac;
                            S = mac2<FMA>(v2f{K(4, tx), K(3, tx)}, Nw[2][col0 + ac][c], S);
                        }
                        // g - 4*s: the product is exact, so one fused op rounds identically
                        lap[0][p][c] = __builtin_fmaf(-4.0f, S.x, gd[0][p * 3 + c]);
                        lap[1][p][c] = __builtin_fmaf(-4.0f, S.y, gd[1][p * 3 + c]);
                    }
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const float gr = gray_of<FMA>(lap[r][p][0], lap[r][p][1], lap[r][p][2]);
                        Q[r][p] = gr * gr;
                    }
                }
                // rows 2i+1, 2i+2 enter the delay line (slot of 2i+2 = slot of 2i-2, read above)
                const int sc = (2 * i + 1) & 3, sd = (2 * i + 2) & 3;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    sGd[sc][k][lane] = v4f{g[0][4 * k], g[0][4 * k + 1], g[0][4 * k + 2], g[0][4 * k + 3]};
                    sGd[sd][k][lane] = v4f{g[1][4 * k], g[1][4 * k + 1], g[1][4 * k + 2], g[1][4 * k + 3]};
                }
            }
            // ---------------- energy: Q rows 2i-2 (r = 0) and 2i-1 (r = 1)
            if (!(a.ablate & 8))
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                float s[8];
                s[0] = lane_prev(Q[r][2]);
                s[1] = lane_prev(Q[r][3]);
#pragma unroll
                for (int p = 0; p < 4; ++p) s[2 + p] = Q[r][p];
                s[6] = lane_next(Q[r][0]);
                s[7] = lane_next(Q[r][1]);
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int tx = 0; tx < 5; ++tx) {
                        const float v = s[p + tx];
                        if (r == 0) {
                            EP[0][p] = mac2<FMA>(v2f{K(4, tx), K(3, tx)}, v, EP[0][p]);
                            EP[1][p] = mac2<FMA>(v2f{K(2, tx), K(1, tx)}, v, EP[1][p]);
                            EP[2][p].x = mac<FMA>(K(0, tx), v, EP[2][p].x);
                        } else {
                            EP[0][p].y = mac<FMA>(K(4, tx), v, EP[0][p].y);
                            EP[1][p] = mac2<FMA>(v2f{K(3, tx), K(2, tx)}, v, EP[1][p]);
                            EP[2][p] = mac2<FMA>(v2f{K(1, tx), K(0, tx)}, v, EP[2][p]);
                        }
                    }
            }
            // ---------------- select rows 2i-4, 2i-3 (complete in EP[0]) against the running state
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int y = 2 * i - 4 + r;
                if (useful && y >= ys && y < ye) {
                    const float e[4] = {r ? EP[0][0].y : EP[0][0].x, r ? EP[0][1].y : EP[0][1].x,
                                        r ? EP[0][2].y : EP[0][2].x, r ? EP[0][3].y : EP[0][3].x};
                    const float o[4] = {eo[r].x, eo[r].y, eo[r].z, eo[r].w};
                    // every energy is >= 0: with no state yet the first frame always wins
                    const bool wn0 = fresh || e[0] > o[0], wn1 = fresh || e[1] > o[1], wn2 = fresh || e[2] > o[2],
                               wn3 = fresh || e[3] > o[3];
                    if ((wn0 || wn1 || wn2 || wn3) && !(a.ablate & 32)) {
                        const v4f l0 = sLd[r][0][lane], l1 = sLd[r][1][lane], l2 = sLd[r][2][lane];
                        const float lv[12] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w, l2.x, l2.y, l2.z, l2.w};
                        const size_t px = (size_t)y * w + x0;
                        const bool wins[4] = {wn0, wn1, wn2, wn3};
#pragma unroll
                        for (int p = 0; p < 4; ++p)
                            if (wins[p]) {
                                a.best_e[px + p] = e[p];
                                a.best_idx[px + p] = fidx;
                                // winner's lap with -0 -> +0, as the np.where sum gives
                                a.best_lap[(px + p) * 3 + 0] = lv[p * 3 + 0] + 0.0f;
                                a.best_lap[(px + p) * 3 + 1] = lv[p * 3 + 1] + 0.0f;
                                a.best_lap[(px + p) * 3 + 2] = lv[p * 3 + 2] + 0.0f;
                            }
                    }
                }
            }
            // the lap rows of this step wait one step for their energies
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    sLd[r][k][lane] = v4f{lap[r][(4 * k) / 3][(4 * k) % 3], lap[r][(4 * k + 1) / 3][(4 * k + 1) % 3],
                                          lap[r][(4 * k + 2) / 3][(4 * k + 2) % 3], lap[r][(4 * k + 3) / 3][(4 * k + 3) % 3]};
#pragma unroll
            for (int p = 0; p < 4; ++p) { EP[0][p] = EP[1][p]; EP[1][p] = EP[2][p]; EP[2][p] = v2f{0.f, 0.f}; }
        }
    }
}

}  // namespace mi
