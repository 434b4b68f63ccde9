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
// Two instantiations per input type: INTERIOR tiles (no coordinate can leave the
// image: no reflection logic at all, launched over 8x8-tile super-blocks so the
// tiles an XCD runs concurrently share their halos in that XCD's L2) and BORDER
// tiles (every coordinate goes through the REFLECT101 maps; a thin frame of
// tiles plus all tiles of the small levels).
//
// Arithmetic is identical, operation by operation, to kernels_simple.hpp and to
// oracle/: taps of one output are applied in row-major order; interleaving the
// chains of different outputs does not change any of them.
#pragma once
#include "common.hpp"

namespace mi {

// The 25 taps float32(k[i]*k[j]) of the symmetric generating kernel
// [a0 a1 a2 a1 a0] take only 6 distinct values.
struct K6 {
    float c[6];  // (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
    __host__ __device__ __forceinline__ float operator()(int ty, int tx) const {
        int a = ty > 2 ? 4 - ty : ty, b = tx > 2 ? 4 - tx : tx;
        int lo = a < b ? a : b, hi = a < b ? b : a;
        return c[lo == 0 ? hi : (lo == 1 ? 2 + hi : 5)];
    }
};

struct LevelArgs {
    const void* src;        // level-l images of the batch (TIn for l == 0, else f32)
    size_t src_stride;      // bytes between consecutive frames
    float* gnext;           // level-(l+1) images of the batch (written)
    size_t gnext_stride;    // floats between consecutive frames
    int nframes;
    int h, w, hn, wn;
    // Interior region in pixels, [iy0, iy1) x [ix0, ix1): a multiple of both the interior
    // and the border kernel's tile size, at least 6 pixels away from every image edge.  The
    // interior kernel tiles it (as SB x SB super-blocks); the border kernel tiles the rest.
    int iy0, iy1, ix0, ix1;
    float* best_e;          // running state of level l
    float* best_lap;
    int32_t* best_idx;
    int first;              // state holds nothing yet
    int frame_idx0;         // global index of the batch's first frame
    int ablate;             // debug: bit mask of phases to skip (timing studies only)
    K6 K;
};

constexpr int SB = 8;  // super-block edge in tiles (64 tiles = one XCD's concurrent set)

// tile configurations (compile-time; tools/tune.sh builds variants with -D):
// level 0 -- 75% of all pixels -- and the coarser levels are tuned separately
#ifndef MI_TILE0_H
#define MI_TILE0_H 32
#endif
#ifndef MI_TILE0_W
#define MI_TILE0_W 64
#endif
#ifndef MI_TILE0_NT
#define MI_TILE0_NT 512
#endif
#ifndef MI_TILE_H
#define MI_TILE_H 32
#endif
#ifndef MI_TILE_W
#define MI_TILE_W 32
#endif
#ifndef MI_TILE_NT
#define MI_TILE_NT 256
#endif
#ifndef MI_TILE_PAD
#define MI_TILE_PAD 1
#endif
#ifndef MI_REDUCE_RU
#define MI_REDUCE_RU 2
#endif

template <int TH_, int TW_, int NT_, bool PAD_>
struct TileGeom {
    static constexpr int TH = TH_, TW = TW_;
    static constexpr int NT = NT_;                    // threads per workgroup
    static constexpr bool PAD = PAD_;                 // bank-conflict padding of the LDS images
    static constexpr int GH = TH + 12, GW = TW + 12;  // G_l patch (pixels)
    static constexpr int GD = GW * 3;                 // data floats per patch row
    // sG row stride: multiple of 4 (ds_read_b128 alignment) and = 28 (mod 32), so that the
    // reduce phase's 16-lane b128 groups -- lanes step 12 dwords along a row and wrap to the
    // next output row (2 patch rows, 2*GS = 56 mod 64) after NW/2 items -- hit 16 distinct
    // 4-bank slots
    static constexpr int GS = PAD ? ((GD - 28 + 31) / 32) * 32 + 28 : GD;
    static constexpr int NH = TH / 2 + 4, NW = TW / 2 + 4;
    // sN row stride.  Narrow tiles (16 quads per row): = 16 (mod 32), so the two quad rows of a
    // 32-lane group fall on disjoint banks.  Wide tiles (>= 32 quads per row: a 32-lane group
    // is one row, conflict-free at any stride): = 6 (mod 32), which spreads the halo ring's
    // side columns -- lanes stepping DOWN the patch -- over the banks (tools/lds_banks.py:
    // 324 -> 135 cycles per tile for the ring's reads at 32x64).
    static constexpr int NS = !PAD ? NW * 3 + 2  // +2: ring columns / quad-row pairs (lds_banks.py)
                              : (TW / 2 >= 32 ? ((NW * 3 - 6 + 31) / 32) * 32 + 6
                                              : ((NW * 3 - 16 + 31) / 32) * 32 + 16);
    static constexpr int QH = TH + 4, QW = TW + 4;
    // Q row stride.  Narrow tiles: 2*QS = 32 (mod 64) so the two quad rows a 32-lane group
    // reads with ds_read_b64 fall into disjoint bank halves.  Wide tiles: = 18 (mod 32) for the
    // ring's column-wise ds_write_b64 (44 -> 16 cycles per tile).
    static constexpr int QS = !PAD ? QW
                              : (TW / 2 >= 32 ? ((QW - 18 + 31) / 32) * 32 + 18
                                              : ((QW + 15) / 32) * 32 + 16);
    static constexpr int NQ = (TH / 2) * (TW / 2) / NT;   // owned 2x2 quads per thread
    static constexpr int LDS_FLOATS = GH * GS + NH * NS + QH * QS;
    static_assert((TH / 2) * (TW / 2) % NT == 0, "tile must split into whole quads per thread");
    static_assert(GS % 4 == 0 && GS >= GD && NS >= NW * 3 && QS >= QW && GD % 4 == 0, "layout");
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

// native vector types: one IR load/store each, so the access width is what we wrote
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// full-rate 24-bit multiply for LDS/pixel index arithmetic (v_mul_lo_u32 is quarter rate)
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }

// two consecutive elements -> two floats; `p` is aligned to 2 elements
__device__ __forceinline__ void load_pair(const float* p, float& a, float& b) {
    v2f v = *reinterpret_cast<const v2f*>(p);
    a = v.x;
    b = v.y;
}
__device__ __forceinline__ void load_pair(const uint16_t* p, float& a, float& b) {
    uint32_t v = *reinterpret_cast<const uint32_t*>(p);
    a = (float)(v & 0xffffu);
    b = (float)(v >> 16);
}
__device__ __forceinline__ void load_pair(const uint8_t* p, float& a, float& b) {
    uint16_t v = *reinterpret_cast<const uint16_t*>(p);
    a = (float)(v & 0xffu);
    b = (float)(v >> 8);
}

// EPL consecutive elements -> floats.  u8 / u16: one (possibly unaligned) 4- / 8-byte load --
// global memory accesses need no alignment on gfx950/amdhsa; f32: see load_pair.
template <int EPL, typename TIn>
__device__ __forceinline__ void load_elems(const TIn* p, float* out, bool aligned) {
    if constexpr (sizeof(TIn) == 1) {
        static_assert(EPL == 4, "u8 staging moves 4 elements per lane");
        uint32_t v;
        __builtin_memcpy(&v, p, 4);
        out[0] = (float)(v & 0xffu);
        out[1] = (float)((v >> 8) & 0xffu);
        out[2] = (float)((v >> 16) & 0xffu);
        out[3] = (float)(v >> 24);
    } else if constexpr (sizeof(TIn) == 2) {
        static_assert(EPL == 4, "u16 staging moves 4 elements per lane");
        uint64_t v;
        __builtin_memcpy(&v, p, 8);
        out[0] = (float)(uint32_t)(v & 0xffffu);
        out[1] = (float)(uint32_t)((v >> 16) & 0xffffu);
        out[2] = (float)(uint32_t)((v >> 32) & 0xffffu);
        out[3] = (float)(uint32_t)(v >> 48);
    } else {
        static_assert(EPL == 2, "f32 staging moves 2 elements per lane");
        if (aligned) load_pair(p, out[0], out[1]);
        else { out[0] = to_f32(p[0]); out[1] = to_f32(p[1]); }
    }
}

// keeps the compiler from hoisting LDS loads across this point (bounds live ranges)
#define MI_LDS_FENCE() asm volatile("" ::: "memory")

// 16-byte LDS load that stays a ds_read_b128: the empty asm pins the four lanes of
// the result, so the compiler cannot re-split the access into narrower reads (it
// otherwise does, to pre-pair operands for v_pk_fma_f32, and the narrow reads at a
// 12-dword lane stride are 4-way bank conflicted).
__device__ __forceinline__ v4f lds_load4(const float* p) {
    return *reinterpret_cast<const v4f*>(p);
}
// global access as uniform base + 32-bit lane byte offset (one VGPR of addressing)
template <typename T>
__device__ __forceinline__ void gstore32(T* base, uint32_t byte_off, T v) {
    *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
__device__ __forceinline__ v2f lds_load2(const float* p) {
    return *reinterpret_cast<const v2f*>(p);
}
__device__ __forceinline__ void lds_store2(float* p, float a, float b) {
    v2f v = {a, b};
    *reinterpret_cast<v2f*>(p) = v;
}
__device__ __forceinline__ void lds_store4(float* p, float a, float b, float c, float d) {
    v4f v = {a, b, c, d};
    *reinterpret_cast<v4f*>(p) = v;
}

template <typename TIn, bool FMA, bool INTERIOR, int TH, int TW, int NT, bool PAD>
__global__ __launch_bounds__(NT) void level_fused(LevelArgs a) {
    using G = TileGeom<TH, TW, NT, PAD>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sG = smem;
    float* sN = sG + G::GH * G::GS;
    float* sQ = sN + G::NH * G::NS;
    const int tid = threadIdx.x;
    const int h = a.h, w = a.w, hn = a.hn, wn = a.wn;

    // ---- which tile?
    int tyi, txi;
    if constexpr (INTERIOR) {
        // Block b runs on XCD b % 8.  Super-block S = 8 * (slot / 64) + xcd: the 64 tiles an
        // XCD works on at a time form one 8x8 square, whose inner halos hit that XCD's L2.
        const int ty_lo = a.iy0 / TH, ty_hi = a.iy1 / TH, tx_lo = a.ix0 / TW, tx_hi = a.ix1 / TW;
        const int sb_x = (tx_hi - tx_lo + SB - 1) / SB;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int S = (slot >> 6) * 8 + xcd, within = slot & 63;
        const int sby = S / sb_x, sbx = S - sby * sb_x;
        tyi = ty_lo + sby * SB + (within >> 3);
        txi = tx_lo + sbx * SB + (within & 7);
        if (tyi >= ty_hi || txi >= tx_hi) return;
    } else {
        // everything outside the interior rectangle: the tile rows above and below it, then
        // the side columns of the rows it spans
        const int tiles_x = (w + TW - 1) / TW, tiles_y = (h + TH - 1) / TH;
        const int ty_lo = a.iy0 / TH, ty_hi = a.iy1 / TH, tx_lo = a.ix0 / TW, tx_hi = a.ix1 / TW;
        const int nyi = ty_hi - ty_lo, nxi = tx_hi - tx_lo;
        int t = blockIdx.x;
        const int top = ty_lo * tiles_x;
        const int bot = (tiles_y - ty_hi) * tiles_x;
        if (nyi <= 0 || nxi <= 0) {  // no interior at all: plain raster
            tyi = t / tiles_x;
            txi = t - tyi * tiles_x;
        } else if (t < top) {
            tyi = t / tiles_x;
            txi = t - tyi * tiles_x;
        } else if (t < top + bot) {
            t -= top;
            tyi = ty_hi + t / tiles_x;
            txi = t - (t / tiles_x) * tiles_x;
        } else {
            t -= top + bot;
            const int side = tiles_x - nxi;  // side tiles per interior row
            tyi = ty_lo + t / side;
            int k = t - (t / side) * side;
            txi = k < tx_lo ? k : tx_hi + (k - tx_lo);
        }
        if (tyi >= tiles_y || txi >= tiles_x) return;
    }
    const int y0 = tyi * TH, x0 = txi * TW;

    // ---- running state of the thread's own quads: (E, idx, lap) live in registers for the
    // whole batch; strict '>' keeps the first maximum.
    float bE[G::NQ][4], bL[G::NQ][4][3];
    int bI[G::NQ][4];
#pragma unroll
    for (int q = 0; q < G::NQ; ++q) {
        const int qi = tid + q * G::NT, oy = qi / (TW / 2), ox = qi % (TW / 2);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int y = y0 + 2 * oy + (p >> 1), x = x0 + 2 * ox + (p & 1);
            const bool valid = INTERIOR || (y < h && x < w);
            if (!a.first && valid) {
                const size_t px = (size_t)y * w + x;
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

    // ---- staging: pass n copies patch rows [n*RPP, (n+1)*RPP); a thread always handles the
    // same EPL consecutive elements of "its" row, so global and LDS offsets advance by constants.
    constexpr int EPL = sizeof(TIn) <= 2 ? 4 : 2;           // elements per lane and load
    static_assert(G::GD % EPL == 0, "patch row must split into whole loads");
    constexpr int PR = G::GD / EPL;                           // loads per patch row
    constexpr int RPP = G::NT / PR;                           // patch rows staged per pass
    constexpr int NPRE = (G::GH + RPP - 1) / RPP;             // passes = loads per thread
    static_assert(RPP >= 1, "workgroup too small for one patch row per pass");
    float pre[NPRE][EPL];
    const bool pair_ok = (w & 1) == 0;  // f32: every patch row then starts on an even element
    const int srow = tid / PR, sk = tid - srow * PR;
    const bool sact = tid < RPP * PR;
    // interior only: byte offsets inside a frame (uniform base + 32-bit lane offset)
    const uint32_t goff0 = (uint32_t)(((y0 - 6 + srow) * w + (x0 - 6)) * 3 + EPL * sk) * (uint32_t)sizeof(TIn);
    const uint32_t gstep = (uint32_t)(RPP * w * 3) * (uint32_t)sizeof(TIn);
    auto prefetch = [&](int b) {
        const char* frb = (const char*)a.src + (size_t)b * a.src_stride;
        const TIn* fr = (const TIn*)frb;
        if (!sact) return;
        if constexpr (INTERIOR) {
#pragma unroll
            for (int n = 0; n < NPRE; ++n)
                if ((n + 1) * RPP <= G::GH || srow + n * RPP < G::GH)
                    load_elems<EPL>((const TIn*)(frb + (size_t)n * gstep + goff0), pre[n], pair_ok);
        } else {
#pragma unroll
            for (int n = 0; n < NPRE; ++n) {
                const int r = srow + n * RPP;
                if (r < G::GH) {
                    const int gy = map_clamp(y0 - 6 + r, h);
#pragma unroll
                    for (int e = 0; e < EPL; ++e) {
                        int k = EPL * sk + e, col = k / 3, c = k - col * 3;
                        int gx = map_clamp(x0 - 6 + col, w);
                        pre[n][e] = to_f32(fr[((size_t)gy * w + gx) * 3 + c]);
                    }
                }
            }
        }
    };
    prefetch(0);

    const K6 K = a.K;
    // ---- per-thread work coordinates, packed (hi << 16 | lo); -1 = no item.  They are
    // laundered at the top of every iteration: the compiler then rebuilds the few addresses it
    // needs per frame (a multiply-add each) instead of either re-deriving them from the thread
    // id (integer divisions) or hoisting dozens of loop-invariant addresses into registers
    // (which cost a wave of occupancy).
    constexpr int RED_BX = G::NW / 2;
    constexpr int RED_RU = INTERIOR ? MI_REDUCE_RU : 1;
    constexpr int RED_N = (G::NH / RED_RU) * RED_BX;
    constexpr int RED_ITEMS = (RED_N + G::NT - 1) / G::NT;
    constexpr int QY = TH / 2 + 2, QX = TW / 2 + 2;
    constexpr int RING = QY * QX - (TH / 2) * (TW / 2);
    constexpr int RING_ITEMS = (RING + G::NT - 1) / G::NT;
    constexpr int GN_CW = (TW / 2) * 3;
    constexpr int GN_C4 = GN_CW / 4;
    constexpr int GN_N4 = (TH / 2) * GN_C4;
    constexpr int GN_ITEMS4 = (GN_N4 + G::NT - 1) / G::NT;
    int c_red[RED_ITEMS], c_ring[RING_ITEMS], c_own[G::NQ], c_gn[GN_ITEMS4];
#pragma unroll
    for (int k = 0; k < RED_ITEMS; ++k) {
        const int it = tid + k * G::NT, rb = it / RED_BX;
        c_red[k] = it < RED_N ? (rb << 16) | (it - rb * RED_BX) : -1;
    }
#pragma unroll
    for (int k = 0; k < RING_ITEMS; ++k) {
        const int it = tid + k * G::NT;
        int qy, qx;
        if (it < QX) { qy = 0; qx = it; }
        else if (it < 2 * QX) { qy = QY - 1; qx = it - QX; }
        else {
            const int sidx = it - 2 * QX;  // left/right columns, rows 1..QY-2
            qy = 1 + (sidx >> 1);
            qx = (sidx & 1) ? QX - 1 : 0;
        }
        c_ring[k] = it < RING ? (qy << 16) | qx : -1;
    }
#pragma unroll
    for (int q = 0; q < G::NQ; ++q) {
        const int qi = tid + q * G::NT, oy = qi / (TW / 2);
        c_own[q] = (oy << 16) | (qi - oy * (TW / 2));
    }
#pragma unroll
    for (int k = 0; k < GN_ITEMS4; ++k) {
        const int e = tid + k * G::NT, r = e / GN_C4;
        c_gn[k] = e < GN_N4 ? (r << 16) | ((e - r * GN_C4) * 4) : -1;
    }
    for (int b = 0; b < a.nframes; ++b) {
        int ltid = tid;
        asm volatile("" : "+v"(ltid));  // used by the border variants only
#pragma unroll
        for (int k = 0; k < RED_ITEMS; ++k) asm volatile("" : "+v"(c_red[k]));
#pragma unroll
        for (int k = 0; k < RING_ITEMS; ++k) asm volatile("" : "+v"(c_ring[k]));
#pragma unroll
        for (int q = 0; q < G::NQ; ++q) asm volatile("" : "+v"(c_own[q]));
#pragma unroll
        for (int k = 0; k < GN_ITEMS4; ++k) asm volatile("" : "+v"(c_gn[k]));
        // ---------------- stage
        if (sact) {
#pragma unroll
            for (int n = 0; n < NPRE; ++n)
                if ((n + 1) * RPP <= G::GH || srow + n * RPP < G::GH) {
                    float* d = sG + mul24(srow + n * RPP, G::GS) + EPL * sk;
                    if constexpr (EPL == 4) lds_store4(d, pre[n][0], pre[n][1], pre[n][2], pre[n][3]);
                    else lds_store2(d, pre[n][0], pre[n][1]);
                }
        }
        __syncthreads();
        if (b + 1 < a.nframes && !(a.ablate & 16)) prefetch(b + 1);

        // ---------------- reduce: items of RU output rows x 2 output pixels x 3 channels
        if (!(a.ablate & 1)) {
            constexpr int RU = RED_RU;  // output rows per item
            static_assert(G::NH % RU == 0, "reduce rows per item must divide the patch height");
#pragma unroll
            for (int kk = 0; kk < RED_ITEMS; ++kk) {
                if (c_red[kk] < 0) continue;
                const int rb = c_red[kk] >> 16, bx = c_red[kk] & 0xffff, ri = rb * RU, rj = 2 * bx;
                float acc[RU][2][3];
#pragma unroll
                for (int u = 0; u < RU; ++u)
#pragma unroll
                    for (int vv = 0; vv < 2; ++vv) acc[u][vv][0] = acc[u][vv][1] = acc[u][vv][2] = 0.f;
                if constexpr (INTERIOR) {
                    // input rows 2ri .. 2ri+2RU+2, pixels 2rj .. 2rj+6: 21 floats, 16-byte aligned.
                    // Row rr+1 is loaded while row rr is consumed (two register sets); input row
                    // rr is tap row rr-2u of output row u.
                    constexpr int NR = 2 * RU + 3;
                    const float* p0 = sG + mul24(2 * ri, G::GS) + 2 * rj * 3;
                    v4f rq[2][5];
                    float rl[2];
                    auto load_row = [&](int rr, int s) {
#pragma unroll
                        for (int t = 0; t < 5; ++t) rq[s][t] = lds_load4(p0 + rr * G::GS + 4 * t);
                        rl[s] = p0[rr * G::GS + 20];
                    };
                    load_row(0, 0);
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr) {
                        const int s = rr & 1;
                        if (rr + 1 < NR) load_row(rr + 1, s ^ 1);
                        MI_LDS_FENCE();
                        float v[21];
#pragma unroll
                        for (int t = 0; t < 5; ++t) {
                            v[4 * t] = rq[s][t].x; v[4 * t + 1] = rq[s][t].y;
                            v[4 * t + 2] = rq[s][t].z; v[4 * t + 3] = rq[s][t].w;
                        }
                        v[20] = rl[s];
#pragma unroll
                        for (int u = 0; u < RU; ++u) {
                            const int ty = rr - 2 * u;
                            if (ty < 0 || ty > 4) continue;
#pragma unroll
                            for (int vv = 0; vv < 2; ++vv)
#pragma unroll
                                for (int tx = 0; tx < 5; ++tx) {
                                    const float k = K(ty, tx);
#pragma unroll
                                    for (int c = 0; c < 3; ++c)
                                        acc[u][vv][c] = mac<FMA>(k, v[(2 * vv + tx) * 3 + c], acc[u][vv][c]);
                                }
                        }
                    }
                } else {
                    const int im = map_expand_src(y0 / 2 - 2 + ri, hn);
#pragma unroll
                    for (int vv = 0; vv < 2; ++vv) {
                        const int jm = map_expand_src(x0 / 2 - 2 + rj + vv, wn);
                        for (int ty = 0; ty < 5; ++ty) {
                            // rows/cols beyond the image already hold reflected data (staging)
                            const int r = clampi(2 * im - 2 + ty - (y0 - 6), 0, G::GH - 1);
#pragma unroll
                            for (int tx = 0; tx < 5; ++tx) {
                                const int cc = clampi(2 * jm - 2 + tx - (x0 - 6), 0, G::GW - 1);
                                const float k = K(ty, tx);
                                const float* p = sG + r * G::GS + cc * 3;
#pragma unroll
                                for (int c = 0; c < 3; ++c) acc[0][vv][c] = mac<FMA>(k, p[c], acc[0][vv][c]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    float* o = sN + mul24(ri + u, G::NS) + rj * 3;  // 6 floats, 8-byte aligned
                    lds_store2(o, acc[u][0][0], acc[u][0][1]);
                    lds_store2(o + 2, acc[u][0][2], acc[u][1][0]);
                    lds_store2(o + 4, acc[u][1][1], acc[u][1][2]);
                }
            }
        }
        __syncthreads();

        // ---------------- store the tile centre of G_{l+1}: (TH/2) rows of (TW/2)*3 floats
        if (!(a.ablate & 2)) {
            float* gout = a.gnext + (size_t)b * a.gnext_stride;
            const int i0 = y0 / 2, j0 = x0 / 2;
            constexpr int CW = (TW / 2) * 3;
            if (INTERIOR && (wn & 3) == 0 && (CW & 3) == 0) {
                // rows start 16-byte aligned in global memory: one float4 per lane
#pragma unroll
                for (int kk = 0; kk < GN_ITEMS4; ++kk) {
                    if (c_gn[kk] < 0) continue;
                    const int r = c_gn[kk] >> 16, k = c_gn[kk] & 0xffff;
                    const float* sp = sN + mul24(r + 2, G::NS) + 6 + k;  // 8-byte aligned
                    const v2f lo = lds_load2(sp), hi = lds_load2(sp + 2);
                    v4f v = {lo.x, lo.y, hi.x, hi.y};
                    *reinterpret_cast<v4f*>(reinterpret_cast<char*>(gout) +
                                            (uint32_t)(mul24(mul24(i0 + r, wn) + j0, 3) + k) * 4u) = v;
                }
            } else {
                for (int e = ltid; e < (TH / 2) * CW; e += G::NT) {
                    const int r = e / CW, k = e - r * CW;
                    const int i = i0 + r, j = j0 + k / 3;
                    if (INTERIOR || (i < hn && j < wn))
                        gstore32(gout, (uint32_t)(mul24(mul24(i, wn) + j0, 3) + k) * 4u,
                                 sN[mul24(r + 2, G::NS) + 6 + k]);
                }
            }
        }

        // ---------------- laplacian + Q on (TH+4) x (TW+4), as 2x2 quads
        float myLap[G::NQ][4][3];
#pragma unroll
        for (int q = 0; q < G::NQ; ++q)
#pragma unroll
            for (int p = 0; p < 4; ++p) myLap[q][p][0] = myLap[q][p][1] = myLap[q][p][2] = 0.f;
        if (!(a.ablate & 4)) {
            auto do_quad = [&](int qy, int qx, float (*keep)[3]) {
                // local sN rows/cols of the expand source, local sG rows/cols of the cells
                int re, ro, ce, co, gre, gro, gce, gco;
                if constexpr (INTERIOR) {
                    re = ro = qy + 1;
                    ce = co = qx + 1;
                    gre = 2 * qy + 4; gro = gre + 1;
                    gce = 2 * qx + 4; gco = gce + 1;
                } else {
                    const int ye = map_clamp(y0 - 2 + 2 * qy, h), yo = map_clamp(y0 - 1 + 2 * qy, h);
                    const int xe = map_clamp(x0 - 2 + 2 * qx, w), xo = map_clamp(x0 - 1 + 2 * qx, w);
                    re = clampi((ye >> 1) - (y0 / 2 - 2), 1, G::NH - 2);
                    ro = clampi(((yo - 1) >> 1) - (y0 / 2 - 2), 0, G::NH - 2);
                    ce = clampi((xe >> 1) - (x0 / 2 - 2), 1, G::NW - 2);
                    co = clampi(((xo - 1) >> 1) - (x0 / 2 - 2), 0, G::NW - 2);
                    gre = clampi(ye - (y0 - 6), 0, G::GH - 1); gro = clampi(yo - (y0 - 6), 0, G::GH - 1);
                    gce = clampi(xe - (x0 - 6), 0, G::GW - 1); gco = clampi(xo - (x0 - 6), 0, G::GW - 1);
                }
                float see[3] = {0, 0, 0}, seo[3] = {0, 0, 0}, soe[3] = {0, 0, 0}, soo[3] = {0, 0, 0};
                // (even row, even col): taps ty in {0,2,4} x tx in {0,2,4};  (even, odd): tx in {1,3}
#pragma unroll
                for (int ar = 0; ar < 3; ++ar) {
                    const float* nrow = sN + mul24(re - 1 + ar, G::NS);
#pragma unroll
                    for (int ac = 0; ac < 3; ++ac) {
                        const float k = K(2 * ar, 2 * ac);
                        const float* p = nrow + (ce - 1 + ac) * 3;
#pragma unroll
                        for (int c = 0; c < 3; ++c) see[c] = mac<FMA>(k, p[c], see[c]);
                    }
#pragma unroll
                    for (int ac = 0; ac < 2; ++ac) {
                        const float k = K(2 * ar, 2 * ac + 1);
                        const float* p = nrow + (co + ac) * 3;
#pragma unroll
                        for (int c = 0; c < 3; ++c) seo[c] = mac<FMA>(k, p[c], seo[c]);
                    }
                }
                // (odd row, *): ty in {1,3}
#pragma unroll
                for (int ar = 0; ar < 2; ++ar) {
                    const float* nrow = sN + mul24(ro + ar, G::NS);
#pragma unroll
                    for (int ac = 0; ac < 3; ++ac) {
                        const float k = K(2 * ar + 1, 2 * ac);
                        const float* p = nrow + (ce - 1 + ac) * 3;
#pragma unroll
                        for (int c = 0; c < 3; ++c) soe[c] = mac<FMA>(k, p[c], soe[c]);
                    }
#pragma unroll
                    for (int ac = 0; ac < 2; ++ac) {
                        const float k = K(2 * ar + 1, 2 * ac + 1);
                        const float* p = nrow + (co + ac) * 3;
#pragma unroll
                        for (int c = 0; c < 3; ++c) soo[c] = mac<FMA>(k, p[c], soo[c]);
                    }
                }
                const float* gee = sG + mul24(gre, G::GS) + gce * 3;
                const float* geo = sG + mul24(gre, G::GS) + gco * 3;
                const float* goe = sG + mul24(gro, G::GS) + gce * 3;
                const float* goo = sG + mul24(gro, G::GS) + gco * 3;
                float l[4][3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    // g - 4*s: the product is exact, so one fused op rounds identically
                    l[0][c] = __builtin_fmaf(-4.0f, see[c], gee[c]);
                    l[1][c] = __builtin_fmaf(-4.0f, seo[c], geo[c]);
                    l[2][c] = __builtin_fmaf(-4.0f, soe[c], goe[c]);
                    l[3][c] = __builtin_fmaf(-4.0f, soo[c], goo[c]);
                }
                float qv[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float gr = gray_of<FMA>(l[p][0], l[p][1], l[p][2]);
                    qv[p] = gr * gr;
                    if (keep) {
                        keep[p][0] = l[p][0];
                        keep[p][1] = l[p][1];
                        keep[p][2] = l[p][2];
                    }
                }
                float* qo = sQ + mul24(2 * qy, G::QS) + 2 * qx;
                lds_store2(qo, qv[0], qv[1]);
                lds_store2(qo + G::QS, qv[2], qv[3]);
            };
            // own quads (tile interior) ...
#pragma unroll
            for (int q = 0; q < G::NQ; ++q)
                do_quad((c_own[q] >> 16) + 1, (c_own[q] & 0xffff) + 1, myLap[q]);
            // ... and the halo ring, spread over the first threads
#pragma unroll
            for (int kk = 0; kk < RING_ITEMS; ++kk)
                if (c_ring[kk] >= 0) do_quad(c_ring[kk] >> 16, c_ring[kk] & 0xffff, nullptr);
        }
        __syncthreads();

        // ---------------- energy of the own quads + running first-max
        if (!(a.ablate & 8)) {
            const int fidx = a.frame_idx0 + b;
#pragma unroll
            for (int q = 0; q < G::NQ; ++q) {
                float e[4] = {0.f, 0.f, 0.f, 0.f};
                const int oy = c_own[q] >> 16, ox = c_own[q] & 0xffff;
                const float* base = sQ + mul24(2 * oy, G::QS) + 2 * ox;  // 8-byte aligned
#pragma unroll
                for (int rr = 0; rr < 6; ++rr) {
                    const float* rp = base + rr * G::QS;
                    v2f v0 = lds_load2(rp), v1 = lds_load2(rp + 2), v2 = lds_load2(rp + 4);
                    const float v[6] = {v0.x, v0.y, v1.x, v1.y, v2.x, v2.y};
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy) {
                        const int ty = rr - dy;
                        if (ty < 0 || ty > 4) continue;
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                            for (int tx = 0; tx < 5; ++tx)
                                e[dy * 2 + dx] = mac<FMA>(K(ty, tx), v[dx + tx], e[dy * 2 + dx]);
                    }
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const bool win = e[p] > bE[q][p];
                    bE[q][p] = win ? e[p] : bE[q][p];
                    bI[q][p] = win ? fidx : bI[q][p];
                    bL[q][p][0] = win ? myLap[q][p][0] : bL[q][p][0];
                    bL[q][p][1] = win ? myLap[q][p][1] : bL[q][p][1];
                    bL[q][p][2] = win ? myLap[q][p][2] : bL[q][p][2];
                }
            }
        }
        // no barrier needed here: the next writes to sG happen after every thread
        // passed the barrier above (sG/sN are only read before it), and sQ is
        // rewritten only after the next iteration's two barriers.
    }

    // ---- write the running state back (winner's lap with -0 -> +0, as the np.where sum gives)
#pragma unroll
    for (int q = 0; q < G::NQ; ++q) {
        const int qi = tid + q * G::NT, oy = qi / (TW / 2), ox = qi % (TW / 2);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int y = y0 + 2 * oy + (p >> 1), x = x0 + 2 * ox + (p & 1);
            if (INTERIOR || (y < h && x < w)) {
                const size_t px = (size_t)y * w + x;
                a.best_e[px] = bE[q][p];
                a.best_idx[px] = bI[q][p];
                a.best_lap[px * 3 + 0] = bL[q][p][0] + 0.0f;
                a.best_lap[px * 3 + 1] = bL[q][p][1] + 0.0f;
                a.best_lap[px * 3 + 2] = bL[q][p][2] + 0.0f;
            }
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
