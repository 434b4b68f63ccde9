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
//   energy  E = 5x5 stencil of Q; running first-max (E, idx) in REGISTERS
//
// and writes the running maxima once per launch.  Per frame the kernel reads G_l
// once and writes G_{l+1} once; the Laplacian, Q and E never touch HBM, and the
// selection state costs 8 B/pixel per LAUNCH instead of per frame.  The winner's
// three-channel Laplacian is filled in once per level and batch (exact_payload):
// round 3 -- until then the kernel carried it in registers (12 more VGPRs, twelve
// selects per pixel and frame, 20 B/pixel of state per launch).
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
#ifdef MI_PHASE_CLOCK
    unsigned long long* dbg;  // [16 intervals] cycle sums over all waves (timing studies only)
#endif
    K6 K;
    float k1d[3];           // MI_ARITH_SEPARABLE: float32 of the 1-D generating kernel (k0, k1, k2)
    float rk[4];            // MI_ARITH_SEPARABLE, reduce: taps (w0, w1, w2) and the final scale (red_taps, kernels_sep.hpp)
    int mfma_ok;            // the integer (MFMA) form of the level-0 reduce may be used: small non-negative integer taps
    // MI_ARITH_SEPARABLE, frame chunks: blockIdx.y = c works on frames [c * chunk_frames, (c + 1) * chunk_frames) of the
    // batch.  Chunk 0 continues the running state; chunk c > 0 starts from nothing and leaves its (max, arg-max) in
    // part_e / part_idx [(c - 1) * part_stride + pixel]; merge_chunks folds them into the running state in chunk order.
    int chunk_frames;
    float* part_e;
    int32_t* part_idx;
    size_t part_stride;
    // Interior launch: super-block of workgroup group g (64 consecutive workgroups of one XCD) = sb_order[g], g = 8 * round
    // + XCD; 0xFFFF = none.  Built on the host so that every XCD gets the same number of TILES (partial super-blocks at
    // the right / bottom edge make `round * 8 + XCD` uneven: 5 % between the XCDs at 24 MP).  nullptr: g itself.
    // (The separable kernel only: the exact kernel's 32 x 64 tiling of 24 MP is even as it is, measured no gain.)
    const uint16_t* sb_order;
    // MI_ARITH_SEPARABLE, level pairs (kernels_sep.hpp, "PAIR"): the level-l kernel also reduces its G_{l+1} patch to the
    // tile's G_{l+2} pixels and hands level l+1 -- whose energy path needs ONE channel -- the gray of G_{l+1}; the three-
    // channel G_{l+1} is stored for one frame of the launch only (`g1_keep`, frame number inside the launch; < 0: none),
    // into `gnext` (then a single image: the debug tap).
    float* gray1;           // gray(G_{l+1}) of the batch, hn x wn floats per frame
    size_t gray1_stride;    // floats between frames
    float* g2;              // G_{l+2} of the batch (written)
    size_t g2_stride;
    int hn2, wn2;
    int g1_keep;
    // the pair's tile-by-tile payload pass (kernels_sep.hpp "PL"): best_idx / best_lap = the first level's, idx1 / lap1 = the
    // second level's; tile_flag[tile] = 1: more distinct winners than the pass walks -- left to the per-quad kernels
    const int32_t* idx1;
    float* lap1;
    uint8_t* tile_flag;
};

// reduce item of lane `tid` for the 32x64 / 512-thread tile with GS = 236 (generated by the search in
// tools/lds_banks.py; 0xFFFF = no item)
__device__ static const uint16_t RED_MAP_32x64[192] = {
    0x0000, 0x000b, 0x0006, 0x0001, 0x0010, 0x0107, 0x0102, 0x0011, 0x0108, 0x0103, 0x010e, 0x0109,
    0x000c, 0x0007, 0x0002, 0x000d, 0x0104, 0x010f, 0x010a, 0x0105, 0x0008, 0x0003, 0x000e, 0x0009,
    0x0004, 0x000f, 0x000a, 0x0005, 0x0100, 0x010b, 0x0106, 0x0101, 0x010c, 0x0203, 0x020e, 0x010d,
    0x0208, 0x030f, 0x030a, 0x0209, 0x0300, 0x030b, 0x0306, 0x0301, 0x0204, 0x020f, 0x020a, 0x0205,
    0x0210, 0x0307, 0x0302, 0x0211, 0x0200, 0x020b, 0x0206, 0x0201, 0x0110, 0x0207, 0x0202, 0x0111,
    0x020c, 0x0303, 0x030e, 0x020d, 0x0304, 0x040b, 0x0406, 0x0305, 0x0400, 0x0507, 0x0502, 0x0401,
    0x040c, 0x0503, 0x050e, 0x040d, 0x0310, 0x0407, 0x0402, 0x0311, 0x0408, 0x050f, 0x050a, 0x0409,
    0x030c, 0x0403, 0x040e, 0x030d, 0x0308, 0x040f, 0x040a, 0x0309, 0x0404, 0x050b, 0x0506, 0x0405,
    0x0410, 0x0603, 0x060e, 0x0411, 0x050c, 0x070f, 0x070a, 0x050d, 0x0604, 0x070b, 0x0706, 0x0605,
    0x0508, 0x060f, 0x060a, 0x0509, 0x0600, 0x0707, 0x0702, 0x0601, 0x0504, 0x060b, 0x0606, 0x0505,
    0x0500, 0x0607, 0x0602, 0x0501, 0x0510, 0x0703, 0x070e, 0x0511, 0x0608, 0x080b, 0x0806, 0x0609,
    0x0704, 0x0907, 0x0902, 0x0705, 0x0710, 0x0903, 0x090e, 0x0711, 0x0700, 0x0807, 0x0802, 0x0701,
    0x070c, 0x090f, 0x090a, 0x070d, 0x0610, 0x0803, 0x080e, 0x0611, 0x060c, 0x080f, 0x080a, 0x060d,
    0x0708, 0x090b, 0x0906, 0x0709, 0x0800, 0x0801, 0x080c, 0x080d, 0x0810, 0x0811, 0x0908, 0x0909,
    0x0904, 0x0905, 0x0900, 0x0901, 0x0808, 0x0809, 0x0804, 0x0805, 0x090d, 0x0911, 0xffff, 0xffff,
    0x090c, 0x0910, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff};

constexpr int SB = 8;  // super-block edge in tiles (64 tiles = one XCD's concurrent set)
// super-block of the separable kernels, tiles in x / y (study knobs; the product SEP_SBW * SEP_SBH consecutive workgroups of an
// XCD share one super-block -- see docs/studies.md "super-block shape")
#ifndef MI_SEP_SBW
#define MI_SEP_SBW 8
#endif
#ifndef MI_SEP_SBH
#define MI_SEP_SBH 8
#endif
constexpr int SEP_SBW = MI_SEP_SBW, SEP_SBH = MI_SEP_SBH;

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
#ifndef MI_F32_EPL4
#define MI_F32_EPL4 1
#endif
// wave scheduling inside the interior workgroup: 1 = reduce waves prefetch late, ring quads on the last lanes
#ifndef MI_SCHED
#define MI_SCHED 1
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
    // The level-0 tile (32x64, 512 threads, 2 output rows per reduce item: 10 x 18 items on 3 waves)
    // assigns the reduce items to lanes through a table instead of linearly: a 16-lane b128 group
    // reads at 4*GS*rb + 12*bx, i.e. 4-bank slot (GS*rb + 3*bx) mod 16, and the table gives every
    // hardware lane group one item of each slot (tools/lds_banks.py: 70 LDS cycles per tile and tap
    // column instead of 115, ideal 60; rocprof had 4/5 of the kernel's bank-conflict cycles here).
    static constexpr bool TABLE_REDUCE = PAD && TW == 64 && TH == 32 && NT == 512 && GS == 236;
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
    } else if constexpr (EPL == 4) {
        // one (possibly unaligned) 16-byte load
        v4f v;
        __builtin_memcpy(&v, p, 16);
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    } else {
        static_assert(EPL == 2, "f32 staging moves 2 or 4 elements per lane");
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
// 8 bytes at a 4-byte aligned LDS address (ds_read2_b32): the two floats land in an aligned register pair
__device__ __forceinline__ v2f lds_load2u(const float* p) {
    v2f v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// Two of the reference's tap chains advanced by ONE packed instruction: acc.x and acc.y are the accumulators of two outputs
// that consume the SAME source element at this step, each with its own tap weight (k.x, k.y) -- e.g. the outputs of two
// pyramid rows that share an input row.  Each half is mac<FMA>'s operation on its own operands, so every chain keeps the
// reference's order and rounding.  The element is half `sel` of an aligned register pair `x`, picked with op_sel (no
// move to build a splat); k is a uniform pair (SGPRs).
template <bool FMA>
__device__ __forceinline__ v2f mac2_shared(v2f k, v2f x, int sel, v2f acc) {
    if constexpr (FMA) {
        if (sel == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(k), "v"(x));
        else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(k), "v"(x));
        return acc;
    } else {
        v2f pr;
        if (sel == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(pr) : "s"(k), "v"(x));
        else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(pr) : "s"(k), "v"(x));
        return acc + pr;   // contraction is off for this TU
    }
}
__device__ __forceinline__ float half_of(v2f x, int sel) { return sel ? x.y : x.x; }

#ifdef MI_PHASE_CLOCK
#define MI_TICK(i)                                                   \
    do {                                                             \
        const unsigned int _t = (unsigned int)clock64();             \
        pc_acc[i] += _t - pc_last;                                   \
        pc_last = _t;                                                \
    } while (0)
#else
#define MI_TICK(i) do { } while (0)
#endif

template <typename TIn, bool FMA, bool INTERIOR, int TH, int TW, int NT, bool PAD>
__device__ __forceinline__ void level_fused_body(const LevelArgs& a) {
    using G = TileGeom<TH, TW, NT, PAD>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sG = smem;
    float* sN = sG + G::GH * G::GS;
    float* sQ = sN + G::NH * G::NS;
    const int tid = threadIdx.x;
    const int h = a.h, w = a.w, hn = a.hn, wn = a.wn;

    // ---- which tile?
    int tyi, txi, y0, x0;
    if constexpr (INTERIOR) {
        // Block b runs on XCD b % 8.  Super-block S = 8 * (slot / 64) + xcd: the 64 tiles an
        // XCD works on at a time form one 8x8 square, whose inner halos hit that XCD's L2.
        // The interior tiling starts at (iy0, ix0), which is aligned to the BORDER tiling only (the
        // image's left border frame is then one 32-pixel border tile wide instead of one interior tile).
        const int nty = (a.iy1 - a.iy0) / TH, ntx = (a.ix1 - a.ix0) / TW;
        const int sb_x = (ntx + SB - 1) / SB;
        const int xcd = (blockIdx.x + 8 - (blockIdx.y & 7)) & 7, slot = blockIdx.x >> 3;   // chunks rotate the XCD
        const int S = (slot >> 6) * 8 + xcd, within = slot & 63;
        const int sby = S / sb_x, sbx = S - sby * sb_x;
        tyi = sby * SB + (within >> 3);
        txi = sbx * SB + (within & 7);
        if (tyi >= nty || txi >= ntx) return;
        y0 = a.iy0 + tyi * TH;
        x0 = a.ix0 + txi * TW;
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
        y0 = tyi * TH;
        x0 = txi * TW;
    }

    // ---- which frames?  (levels with few tiles split the batch into chunks along blockIdx.y, see LevelArgs)
    const int ck = blockIdx.y, f_lo = ck * a.chunk_frames;
    const int nfr = min(a.nframes - f_lo, a.chunk_frames);
    const bool fresh = ck > 0 || a.first;
    float* const st_e = ck ? a.part_e + (size_t)(ck - 1) * a.part_stride : a.best_e;
    int32_t* const st_i = ck ? a.part_idx + (size_t)(ck - 1) * a.part_stride : a.best_idx;
    const char* const src0 = (const char*)a.src + (size_t)f_lo * a.src_stride;
    float* const gnext0 = a.gnext + (size_t)f_lo * a.gnext_stride;

    // ---- running state of the thread's own quads: (max energy, its frame) live in registers for the
    // whole launch; strict '>' keeps the first maximum.  The running arg-max is not read back: -1 = "no
    // frame of this launch has won (yet)", and only pixels a frame of this launch won are written.
    float bE[G::NQ][4];
    int bI[G::NQ][4];
#pragma unroll
    for (int q = 0; q < G::NQ; ++q) {
        const int qi = tid + q * G::NT, oy = qi / (TW / 2), ox = qi % (TW / 2);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int y = y0 + 2 * oy + (p >> 1), x = x0 + 2 * ox + (p & 1);
            const bool valid = INTERIOR || (y < h && x < w);
            bE[q][p] = (!fresh && valid) ? st_e[(size_t)y * w + x] : -__builtin_inff();  // the first frame always wins (a > 0.5: negative taps, negative energies)
            bI[q][p] = -1;
        }
    }

    // ---- staging: pass n copies patch rows [n*RPP, (n+1)*RPP); a thread always handles the
    // same EPL consecutive elements of "its" row, so global and LDS offsets advance by constants.
    constexpr int EPL = (sizeof(TIn) <= 2 || MI_F32_EPL4) ? 4 : 2;   // elements per lane and load
    static_assert(G::GD % EPL == 0, "patch row must split into whole loads");
    constexpr int PR = G::GD / EPL;                           // loads per patch row
    constexpr int RPP = G::NT / PR;                           // patch rows staged per pass
    constexpr int NPRE = (G::GH + RPP - 1) / RPP;             // passes = loads per thread
    static_assert(RPP >= 1, "workgroup too small for one patch row per pass");
    float pre[NPRE][EPL];
    // 8/16-bit interior tiles keep the prefetched elements RAW (one / two dwords per load instead of
    // four floats) across the frame loop and convert at the staging store: 18 / 12 fewer live VGPRs
    constexpr bool RAW = INTERIOR && sizeof(TIn) <= 2;
    constexpr int RAWW = sizeof(TIn) == 1 ? 1 : 2;
    uint32_t praw[RAW ? NPRE : 1][RAWW];
    const bool pair_ok = (w & 1) == 0;  // f32: every patch row then starts on an even element
    const int srow = tid / PR, sk = tid - srow * PR;
    const bool sact = tid < RPP * PR;
    // interior only: byte offsets inside a frame (uniform base + 32-bit lane offset)
    const uint32_t goff0 = (uint32_t)(((y0 - 6 + srow) * w + (x0 - 6)) * 3 + EPL * sk) * (uint32_t)sizeof(TIn);
    const uint32_t gstep = (uint32_t)(RPP * w * 3) * (uint32_t)sizeof(TIn);
    auto prefetch = [&](int b) {
        const char* frb = src0 + (size_t)b * a.src_stride;
        const TIn* fr = (const TIn*)frb;
        if (!sact) return;
        if constexpr (INTERIOR) {
#pragma unroll
            for (int n = 0; n < NPRE; ++n)
                if ((n + 1) * RPP <= G::GH || srow + n * RPP < G::GH) {
                    if constexpr (RAW) __builtin_memcpy(praw[n], frb + (size_t)n * gstep + goff0, 4 * RAWW);
                    else load_elems<EPL>((const TIn*)(frb + (size_t)n * gstep + goff0), pre[n], pair_ok);
                }
        } else {
            // the lane's EPL elements lie in columns col_lo .. col_hi of the patch: inside the image they
            // are contiguous in memory (the row itself may be a reflected one) and come in with one load
            const int col_lo = (EPL * sk) / 3, col_hi = (EPL * sk + EPL - 1) / 3;
            const bool inside = x0 - 6 + col_lo >= 0 && x0 - 6 + col_hi < w;
#pragma unroll
            for (int n = 0; n < NPRE; ++n) {
                const int r = srow + n * RPP;
                if (r < G::GH) {
                    const int gy = map_clamp(y0 - 6 + r, h);
                    if (inside) {
                        load_elems<EPL>(fr + ((size_t)gy * w + (x0 - 6)) * 3 + EPL * sk, pre[n], false);
                    } else {
#pragma unroll
                        for (int e = 0; e < EPL; ++e) {
                            int k = EPL * sk + e, col = k / 3, c = k - col * 3;
                            int gx = map_clamp(x0 - 6 + col, w);
                            pre[n][e] = to_f32(fr[((size_t)gy * w + gx) * 3 + c]);
                        }
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
        if constexpr (INTERIOR && G::TABLE_REDUCE && RED_RU == 2 && RED_ITEMS == 1) {
            static_assert(RED_BX == 18 && G::NH / RED_RU == 10, "the reduce item table is for the 32x64 tile");
            const int e = it < 192 ? RED_MAP_32x64[it] : 0xFFFF;   // (row pair << 8) | column pair
            if (!MI_ABL(1024)) c_red[k] = e != 0xFFFF ? ((e >> 8) << 16) | (e & 0xff) : -1;
        }
    }
    // waves that own reduce items (lanes below RED_N, or the table's 192 lanes)
    const bool red_wave = (tid >> 6) < (((G::TABLE_REDUCE && INTERIOR ? 192 : (RED_N < G::NT ? RED_N : G::NT)) + 63) >> 6);
#pragma unroll
    for (int k = 0; k < RING_ITEMS; ++k) {
        // the halo ring goes to the LAST lanes: the first waves carry the reduce items and issue their loads late
        const int it = (MI_SCHED && INTERIOR ? G::NT - 1 - tid : tid) + k * G::NT;
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
#ifdef MI_PHASE_CLOCK
    unsigned int pc_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, pc_last = (unsigned int)clock64();
#endif
    for (int b = 0; b < nfr; ++b) {
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
                    if constexpr (RAW) {
                        if constexpr (sizeof(TIn) == 1) {
                            const uint32_t v = praw[n][0];
                            lds_store4(d, (float)(v & 0xffu), (float)((v >> 8) & 0xffu), (float)((v >> 16) & 0xffu),
                                       (float)(v >> 24));
                        } else {
                            const uint32_t v0 = praw[n][0], v1 = praw[n][1];
                            lds_store4(d, (float)(v0 & 0xffffu), (float)(v0 >> 16), (float)(v1 & 0xffffu),
                                       (float)(v1 >> 16));
                        }
                    } else if constexpr (EPL == 4) lds_store4(d, pre[n][0], pre[n][1], pre[n][2], pre[n][3]);
                    else lds_store2(d, pre[n][0], pre[n][1]);
                }
        }
        MI_TICK(0);   // stage (waits for the prefetched loads)
        __syncthreads();
        MI_TICK(1);   // barrier 1
        // Issuing the next frame's loads stalls a wave for about as long as a stencil phase takes (the memory
        // pipeline is kept full), so the waves that own reduce items -- the critical path to the next barrier --
        // issue theirs after it; the others, idle until then, issue now.
        const bool pf_late = MI_SCHED && INTERIOR && red_wave;
        if (!pf_late && b + 1 < nfr && !MI_ABL(16)) prefetch(b + 1);
        MI_TICK(2);   // prefetch issue

        // ---------------- reduce: items of RU output rows x 2 output pixels x 3 channels
        if (!MI_ABL(1)) {
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
                if constexpr (INTERIOR && RU == 2) {
                    // input rows 2ri .. 2ri+6, pixels 2rj .. 2rj+6: 21 floats (+1 spare), 16-byte aligned.
                    // Row rr+1 is loaded while row rr is consumed (two register sets).  Input row rr is tap row rr of
                    // output row 0 and tap row rr-2 of output row 1: rows 2..4 feed both, with one packed instruction
                    // per (tap column, pixel, channel) -- 35 instead of 50 instructions per chain pair.
                    constexpr int NR = 7;
                    const float* p0 = sG + mul24(2 * ri, G::GS) + 2 * rj * 3;
                    v4f rq[2][5];
                    v2f rl[2];
                    auto load_row = [&](int rr, int s) {
#pragma unroll
                        for (int t = 0; t < 5; ++t) rq[s][t] = lds_load4(p0 + rr * G::GS + 4 * t);
                        rl[s] = lds_load2(p0 + rr * G::GS + 20);
                    };
                    v2f acc2[2][3];   // [pixel][channel] = (output row 0, output row 1)
#pragma unroll
                    for (int vv = 0; vv < 2; ++vv)
#pragma unroll
                        for (int c = 0; c < 3; ++c) acc2[vv][c] = v2f{0.f, 0.f};
                    load_row(0, 0);
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr) {
                        const int s = rr & 1;
                        if (rr + 1 < NR) load_row(rr + 1, s ^ 1);
                        MI_LDS_FENCE();
                        v2f P[11];
#pragma unroll
                        for (int t = 0; t < 5; ++t) {
                            P[2 * t] = rq[s][t].xy;
                            P[2 * t + 1] = rq[s][t].zw;
                        }
                        P[10] = rl[s];
#pragma unroll
                        for (int vv = 0; vv < 2; ++vv)
#pragma unroll
                            for (int tx = 0; tx < 5; ++tx)
#pragma unroll
                                for (int c = 0; c < 3; ++c) {
                                    const int e = (2 * vv + tx) * 3 + c;
                                    if (rr >= 2 && rr <= 4)
                                        acc2[vv][c] = mac2_shared<FMA>(v2f{K(rr, tx), K(rr - 2, tx)}, P[e >> 1], e & 1, acc2[vv][c]);
                                    else if (rr < 2)
                                        acc2[vv][c].x = mac<FMA>(K(rr, tx), half_of(P[e >> 1], e & 1), acc2[vv][c].x);
                                    else
                                        acc2[vv][c].y = mac<FMA>(K(rr - 2, tx), half_of(P[e >> 1], e & 1), acc2[vv][c].y);
                                }
                    }
#pragma unroll
                    for (int vv = 0; vv < 2; ++vv)
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            acc[0][vv][c] = acc2[vv][c].x;
                            acc[1][vv][c] = acc2[vv][c].y;
                        }
                } else if constexpr (INTERIOR) {
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
                    {
                        // the common case: both cells of the item map to neighbouring columns and their
                        // joint 5 x 7 pixel window lies inside the patch -- read it like the interior
                        // path does, 8 bytes at a time (rows and windows start on even floats)
                        const int jm0 = map_expand_src(x0 / 2 - 2 + rj, wn), jm1 = map_expand_src(x0 / 2 - 2 + rj + 1, wn);
                        const int r0 = 2 * im - 2 - (y0 - 6), c0 = 2 * jm0 - 2 - (x0 - 6);
                        if ((G::GS & 1) == 0 && jm1 == jm0 + 1 && r0 >= 0 && r0 + 4 < G::GH && c0 >= 0 && c0 + 6 < G::GW) {
                            const float* p = sG + mul24(r0, G::GS) + c0 * 3;
#pragma unroll
                            for (int ty = 0; ty < 5; ++ty) {
                                float v[22];
#pragma unroll
                                for (int t = 0; t < 11; ++t) {
                                    const v2f q = lds_load2(p + ty * G::GS + 2 * t);   // 21 floats + 1 spare
                                    v[2 * t] = q.x; v[2 * t + 1] = q.y;
                                }
#pragma unroll
                                for (int vv = 0; vv < 2; ++vv)
#pragma unroll
                                    for (int tx = 0; tx < 5; ++tx) {
                                        const float k = K(ty, tx);
#pragma unroll
                                        for (int c = 0; c < 3; ++c)
                                            acc[0][vv][c] = mac<FMA>(k, v[(2 * vv + tx) * 3 + c], acc[0][vv][c]);
                                    }
                            }
                            goto reduce_store;
                        }
                    }
#pragma unroll
                    for (int vv = 0; vv < 2; ++vv) {
                        const int jm = map_expand_src(x0 / 2 - 2 + rj + vv, wn);
                        // window of the (mapped) cell in patch coordinates; the patch already holds
                        // reflected data, so a window that lies inside it needs no per-tap mapping
                        const int r0 = 2 * im - 2 - (y0 - 6), c0 = 2 * jm - 2 - (x0 - 6);
                        if (r0 >= 0 && r0 + 4 < G::GH && c0 >= 0 && c0 + 4 < G::GW) {
                            const float* p = sG + mul24(r0, G::GS) + c0 * 3;
#pragma unroll
                            for (int ty = 0; ty < 5; ++ty)
#pragma unroll
                                for (int tx = 0; tx < 5; ++tx) {
                                    const float k = K(ty, tx);
#pragma unroll
                                    for (int c = 0; c < 3; ++c)
                                        acc[0][vv][c] = mac<FMA>(k, p[ty * G::GS + tx * 3 + c], acc[0][vv][c]);
                                }
                            continue;
                        }
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
            reduce_store:
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    float* o = sN + mul24(ri + u, G::NS) + rj * 3;  // 6 floats, 8-byte aligned
                    lds_store2(o, acc[u][0][0], acc[u][0][1]);
                    lds_store2(o + 2, acc[u][0][2], acc[u][1][0]);
                    lds_store2(o + 4, acc[u][1][1], acc[u][1][2]);
                }
            }
        }
        MI_TICK(3);   // reduce
        __syncthreads();
        MI_TICK(4);   // barrier 2
        if (pf_late && b + 1 < nfr && !MI_ABL(16)) prefetch(b + 1);

        // ---------------- store the tile centre of G_{l+1}: (TH/2) rows of (TW/2)*3 floats
        if (!MI_ABL(2)) {
            float* gout = gnext0 + (size_t)b * a.gnext_stride;
            const int i0 = y0 / 2, j0 = x0 / 2;
            constexpr int CW = (TW / 2) * 3;
            // whole tile centre inside G_{l+1} (always so for interior tiles, mostly so for border ones)
            const bool whole = INTERIOR || (i0 + TH / 2 <= hn && j0 + TW / 2 <= wn);
            if (whole && (wn & 3) == 0 && (CW & 3) == 0) {
                // rows start 16-byte aligned in global memory: one float4 per lane
#pragma unroll
                for (int kk = 0; kk < GN_ITEMS4; ++kk) {
                    if (c_gn[kk] < 0) continue;
                    const int r = c_gn[kk] >> 16, k = c_gn[kk] & 0xffff;
                    const float* sp = sN + mul24(r + 2, G::NS) + 6 + k;  // 8-byte aligned
                    const v2f lo = lds_load2(sp), hi = lds_load2(sp + 2);
                    v4f v = {lo.x, lo.y, hi.x, hi.y};
                    *reinterpret_cast<v4f*>(reinterpret_cast<char*>(gout) +
                                            ((uint32_t)(mul24(i0 + r, wn) + j0) * 3u + (uint32_t)k) * 4u) = v;
                }
            } else {
                for (int e = ltid; e < (TH / 2) * CW; e += G::NT) {
                    const int r = e / CW, k = e - r * CW;
                    const int i = i0 + r, j = j0 + k / 3;
                    if (INTERIOR || (i < hn && j < wn))
                        gstore32(gout, ((uint32_t)(mul24(i, wn) + j0) * 3u + (uint32_t)k) * 4u,
                                 sN[mul24(r + 2, G::NS) + 6 + k]);
                }
            }
        }

        MI_TICK(5);   // G_{l+1} store
        // ---------------- laplacian + Q on (TH+4) x (TW+4), as 2x2 quads
        if (!MI_ABL(4)) {
            auto do_quad = [&](int qy, int qx) {
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
                if constexpr (INTERIOR) {
                    // The even output row takes tap rows 0, 2, 4 from G_{l+1} rows qy, qy+1, qy+2; the odd one tap rows
                    // 1, 3 from rows qy+1, qy+2: on those two rows one packed instruction advances the even-row and
                    // the odd-row chain of the same column phase (15 instead of 25 instructions per channel).  A row's
                    // window (pixels qx .. qx+2: nine floats, 4-byte aligned) comes in as five register pairs.
                    v2f A[3], B[3];   // (see, soe) and (seo, soo) per channel
#pragma unroll
                    for (int c = 0; c < 3; ++c) A[c] = B[c] = v2f{0.f, 0.f};
#pragma unroll
                    for (int ar = 0; ar < 3; ++ar) {
                        const float* wp = sN + mul24(qy + ar, G::NS) + 3 * qx;
                        v2f W[5];
#pragma unroll
                        for (int t = 0; t < 5; ++t) W[t] = lds_load2u(wp + 2 * t);   // the tenth float is not used
#pragma unroll
                        for (int ac = 0; ac < 3; ++ac)
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                const int e = 3 * ac + c;
                                if (ar == 0) A[c].x = mac<FMA>(K(0, 2 * ac), half_of(W[e >> 1], e & 1), A[c].x);
                                else A[c] = mac2_shared<FMA>(v2f{K(2 * ar, 2 * ac), K(2 * ar - 1, 2 * ac)}, W[e >> 1], e & 1, A[c]);
                            }
#pragma unroll
                        for (int ac = 0; ac < 2; ++ac)
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                const int e = 3 * (1 + ac) + c;
                                if (ar == 0) B[c].x = mac<FMA>(K(0, 2 * ac + 1), half_of(W[e >> 1], e & 1), B[c].x);
                                else B[c] = mac2_shared<FMA>(v2f{K(2 * ar, 2 * ac + 1), K(2 * ar - 1, 2 * ac + 1)}, W[e >> 1], e & 1, B[c]);
                            }
                    }
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        see[c] = A[c].x; soe[c] = A[c].y;
                        seo[c] = B[c].x; soo[c] = B[c].y;
                    }
                } else {
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
                }
                float* qo = sQ + mul24(2 * qy, G::QS) + 2 * qx;
                lds_store2(qo, qv[0], qv[1]);
                lds_store2(qo + G::QS, qv[2], qv[3]);
            };
            // own quads (tile interior) ...
#pragma unroll
            for (int q = 0; q < G::NQ; ++q)
                do_quad((c_own[q] >> 16) + 1, (c_own[q] & 0xffff) + 1);
            // ... and the halo ring, spread over the first threads
#pragma unroll
            for (int kk = 0; kk < RING_ITEMS; ++kk)
                if (c_ring[kk] >= 0) do_quad(c_ring[kk] >> 16, c_ring[kk] & 0xffff);
        }
        MI_TICK(6);   // laplacian + Q
        __syncthreads();
        MI_TICK(7);   // barrier 3

        // ---------------- energy of the own quads + running first-max
        if (!MI_ABL(8)) {
            const int fidx = a.frame_idx0 + f_lo + b;
#pragma unroll
            for (int q = 0; q < G::NQ; ++q) {
                const int oy = c_own[q] >> 16, ox = c_own[q] & 0xffff;
                const float* base = sQ + mul24(2 * oy, G::QS) + 2 * ox;  // 8-byte aligned
                // Q row rr is tap row rr of the quad's upper pixels and tap row rr-1 of its lower ones: rows 1..4 advance
                // both chains of a column with one packed instruction (60 instead of 100 per quad)
                v2f E[2] = {v2f{0.f, 0.f}, v2f{0.f, 0.f}};   // [dx] = (upper, lower)
#pragma unroll
                for (int rr = 0; rr < 6; ++rr) {
                    const float* rp = base + rr * G::QS;
                    v2f V[3];
                    V[0] = lds_load2(rp);
                    if MI_ABL(2048) { V[1] = V[0]; V[2] = V[0]; }   // probe: a third of the energy phase's LDS reads
                    else { V[1] = lds_load2(rp + 2); V[2] = lds_load2(rp + 4); }
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                        for (int tx = 0; tx < 5; ++tx) {
                            const int e1 = dx + tx;
                            if (rr == 0) E[dx].x = mac<FMA>(K(0, tx), half_of(V[e1 >> 1], e1 & 1), E[dx].x);
                            else if (rr == 5) E[dx].y = mac<FMA>(K(4, tx), half_of(V[e1 >> 1], e1 & 1), E[dx].y);
                            else E[dx] = mac2_shared<FMA>(v2f{K(rr, tx), K(rr - 1, tx)}, V[e1 >> 1], e1 & 1, E[dx]);
                        }
                }
                const float e[4] = {E[0].x, E[1].x, E[0].y, E[1].y};
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const bool win = e[p] > bE[q][p];
                    bE[q][p] = win ? e[p] : bE[q][p];
                    bI[q][p] = win ? fidx : bI[q][p];
                }
            }
        }
        MI_TICK(8);   // energy + select
        // no barrier needed here: the next writes to sG happen after every thread
        // passed the barrier above (sG/sN are only read before it), and sQ is
        // rewritten only after the next iteration's two barriers.
    }

#ifdef MI_PHASE_CLOCK
    if (INTERIOR && a.dbg && (tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) atomicAdd(a.dbg + (tid >> 6) * 16 + i, (unsigned long long)pc_acc[i]);
        atomicAdd(a.dbg + (tid >> 6) * 16 + 15, 1ull);
    }
#endif
    // ---- write the running maxima back: only pixels a frame of this launch won (a fresh state: all of them)
#pragma unroll
    for (int q = 0; q < G::NQ; ++q) {
        const int qi = tid + q * G::NT, oy = qi / (TW / 2), ox = qi % (TW / 2);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int y = y0 + 2 * oy + (p >> 1), x = x0 + 2 * ox + (p & 1);
            if ((INTERIOR || (y < h && x < w)) && bI[q][p] >= 0) {
                const size_t px = (size_t)y * w + x;
                st_e[px] = bE[q][p];
                st_i[px] = bI[q][p];
            }
        }
    }
}

// ================================================================================================
// Winner's Laplacian of the frames of one batch, reference evaluation order (the level kernel keeps only the running
// maximum and its frame): one lane per 2x2 quad of level l.  A pixel whose arg-max is a frame of this batch gets
// lap = fma(-4, expand-stencil(G_{l+1}), G_l) of that frame -- the polyphase taps of the 25-tap chain in row-major
// order, exactly as level_fused's lapq phase and kernels_simple.hpp evaluate it -- with -0 -> +0 as the reference's
// np.where sum gives (pyramid.py:52-54).
struct __attribute__((packed, aligned(4))) Pix3 { float v[3]; };   // one pixel: a 12-byte load / store
template <typename TIn, bool FMA>
__global__ void exact_payload(const void* __restrict__ src, size_t src_stride, const float* __restrict__ gnext,
                              size_t gnext_stride, int nframes, int h, int w, int hn, int wn,
                              const int32_t* __restrict__ best_idx, int frame_idx0, float* __restrict__ best_lap, K6 K) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y * blockDim.y + threadIdx.y;
    if (2 * i >= h || 2 * j >= w) return;
    int fr[4];
    unsigned pending = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int y = 2 * i + (p >> 1), x = 2 * j + (p & 1);
        fr[p] = -1;
        if (y < h && x < w) {
            const int f = best_idx[(size_t)y * w + x] - frame_idx0;
            if (f >= 0 && f < nframes) { fr[p] = f; pending |= 1u << p; }
        }
    }
    if (!pending) return;
    const int ri[3] = {map_expand_src(i - 1, hn), i, map_expand_src(i + 1, hn)};
    const int cj[3] = {map_expand_src(j - 1, wn), j, map_expand_src(j + 1, wn)};
    while (pending) {
        const int f = (pending & 1u) ? fr[0] : (pending & 2u) ? fr[1] : (pending & 4u) ? fr[2] : fr[3];
        const float* gn = gnext + (size_t)f * gnext_stride;
        const TIn* g = (const TIn*)((const char*)src + (size_t)f * src_stride);
        Pix3 N[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) N[r][q] = *(const Pix3*)(gn + ((size_t)ri[r] * wn + cj[q]) * 3);
        float e[4][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float see = 0.f, seo = 0.f, soe = 0.f, soo = 0.f;
            // (even row, even col): taps ty in {0,2,4} x tx in {0,2,4};  (even, odd): tx in {1,3}
#pragma unroll
            for (int ar = 0; ar < 3; ++ar) {
#pragma unroll
                for (int ac = 0; ac < 3; ++ac) see = mac<FMA>(K(2 * ar, 2 * ac), N[ar][ac].v[c], see);
#pragma unroll
                for (int ac = 0; ac < 2; ++ac) seo = mac<FMA>(K(2 * ar, 2 * ac + 1), N[ar][1 + ac].v[c], seo);
            }
            // (odd row, *): ty in {1,3}
#pragma unroll
            for (int ar = 0; ar < 2; ++ar) {
#pragma unroll
                for (int ac = 0; ac < 3; ++ac) soe = mac<FMA>(K(2 * ar + 1, 2 * ac), N[1 + ar][ac].v[c], soe);
#pragma unroll
                for (int ac = 0; ac < 2; ++ac) soo = mac<FMA>(K(2 * ar + 1, 2 * ac + 1), N[1 + ar][1 + ac].v[c], soo);
            }
            e[0][c] = see; e[1][c] = seo; e[2][c] = soe; e[3][c] = soo;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (!((pending >> p) & 1u) || fr[p] != f) continue;
            pending &= ~(1u << p);
            const size_t px = (size_t)(2 * i + (p >> 1)) * w + 2 * j + (p & 1);
            Pix3 o;
#pragma unroll
            for (int c = 0; c < 3; ++c)   // g - 4*s: the product is exact, so one fused op rounds identically
                o.v[c] = __builtin_fmaf(-4.0f, e[p][c], to_f32(g[px * 3 + c])) + 0.0f;
            *(Pix3*)(best_lap + px * 3) = o;
        }
    }
}

// collapse step in the reference's evaluation order (pyramid.py:57-64), one lane per 2x2 quad of the output: the 3x3
// patch of `up` that expands to the quad is loaded once (12-byte pixels), the four polyphase sums are the 25-tap chain's
// non-stuffed taps in row-major order -- the same chains as kernels_simple.hpp's expand_at -- then out = 4*s + lap.
// TOut != float: the finest step, fused with clip(abs()) and the truncating cast (pyramid.py:64, :179).
template <bool FMA, typename TOut>
__global__ void collapse_exact_quad(const float* __restrict__ up, int hs, int ws, const float* __restrict__ lap, int h, int w,
                                    float maxv, TOut* __restrict__ out, K25 K) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y * blockDim.y + threadIdx.y;
    if (2 * i >= h || 2 * j >= w) return;
    const int ri[3] = {map_expand_src(i - 1, hs), map_expand_src(i, hs), map_expand_src(i + 1, hs)};
    const int cj[3] = {map_expand_src(j - 1, ws), map_expand_src(j, ws), map_expand_src(j + 1, ws)};
    Pix3 N[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) N[r][q] = *(const Pix3*)(up + ((size_t)ri[r] * ws + cj[q]) * 3);
    float e[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float see = 0.f, seo = 0.f, soe = 0.f, soo = 0.f;
#pragma unroll
        for (int ar = 0; ar < 3; ++ar) {
#pragma unroll
            for (int ac = 0; ac < 3; ++ac) see = mac<FMA>(K.k[(2 * ar) * 5 + 2 * ac], N[ar][ac].v[c], see);
#pragma unroll
            for (int ac = 0; ac < 2; ++ac) seo = mac<FMA>(K.k[(2 * ar) * 5 + 2 * ac + 1], N[ar][1 + ac].v[c], seo);
        }
#pragma unroll
        for (int ar = 0; ar < 2; ++ar) {
#pragma unroll
            for (int ac = 0; ac < 3; ++ac) soe = mac<FMA>(K.k[(2 * ar + 1) * 5 + 2 * ac], N[1 + ar][ac].v[c], soe);
#pragma unroll
            for (int ac = 0; ac < 2; ++ac) soo = mac<FMA>(K.k[(2 * ar + 1) * 5 + 2 * ac + 1], N[1 + ar][1 + ac].v[c], soo);
        }
        e[0][c] = 4.0f * see; e[1][c] = 4.0f * seo; e[2][c] = 4.0f * soe; e[3][c] = 4.0f * soo;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int y = 2 * i + (p >> 1), x = 2 * j + (p & 1);
        if (y >= h || x >= w) continue;
        const size_t px = ((size_t)y * w + x) * 3;
        const Pix3 lv = *(const Pix3*)(lap + px);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = e[p][c] + lv.v[c];
            if constexpr (sizeof(TOut) != 4) {
                v = fabsf(v);
                v = v > maxv ? maxv : v;
            }
            out[px + c] = (TOut)v;
        }
    }
}

// The kernels proper.  A coarser level that runs on level 0's tile configuration gets its own name, so that profiles
// (rocprofv3 --stats aggregates by kernel name) keep the level-0 launches apart.
template <typename TIn, bool FMA, bool INTERIOR, int TH, int TW, int NT, bool PAD>
__global__ __launch_bounds__(NT) void level_fused(LevelArgs a) {
    level_fused_body<TIn, FMA, INTERIOR, TH, TW, NT, PAD>(a);
}
template <typename TIn, bool FMA, bool INTERIOR, int TH, int TW, int NT, bool PAD>
__global__ __launch_bounds__(NT) void level_fused_coarse(LevelArgs a) {
    level_fused_body<TIn, FMA, INTERIOR, TH, TW, NT, PAD>(a);
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
    // The base level is a heavily smoothed image: the pixels of a wave fall into a handful of bins.  One atomic per
    // distinct bin of the wave (leader lane adds the wave's count) instead of 64 atomics on the same few addresses.
    bool todo = true;
    while (todo) {
        const int l0 = __builtin_amdgcn_readfirstlane(l);
        const bool same = l == l0;
        const unsigned long long m = __ballot(same);
        if (same) {
            if ((int)__lane_id() == __builtin_ctzll(m)) atomicAdd(&cnt[(size_t)f * nlevels + l0], (uint32_t)__builtin_popcountll(m));
            todo = false;
        }
    }
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
