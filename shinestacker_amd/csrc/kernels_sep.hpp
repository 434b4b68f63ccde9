// kernels_sep.hpp -- MI_ARITH_SEPARABLE: the LDS-staged 5-tap separable / polyphase Burt-Adelson form of the
// fused level kernel (gfx950).  Same job as kernels_tiled.hpp's level_fused (one launch = one pyramid level of
// a batch of frames, running first-max state in registers for the whole batch), different arithmetic: every 5x5
// stencil of the reference (algorithms/pyramid.py:24-46, cv2.filter2D with the outer-product kernel) is evaluated
// as two 1-D passes with the float32 generating kernel [k0 k1 k2 k1 k0] in its symmetric form
//
//     s5(a, b, c, d, e) = fma(k0, a + e, fma(k1, b + d, k2 * c))
//
//   reduce   V = s5r down the rows at even rows, G_{l+1} = rs * s5r along the rows at even columns   (pyramid.py:27-32)
//            s5r(a, b, c, d, e) = fma(w1, b + d, fma(w0, a + e, w2 * c)) with (w0, w1, w2, rs) = red_taps(gen_kernel):
//            when 20 k is integral -- the reference's default 0.4 gives (1, 5, 8) -- the taps are those INTEGERS and
//            rs = float32(1 / 400): every product and every partial sum of integer-valued input (8 / 16-bit frames, fp32
//            frames holding such values) is exact, the one rounding of the sum is that of the exact integer sum S, and
//            G_{l+1} = fl(fl(S) * rs) -- two roundings instead of ten, and an integer pipeline (level_sep's MFMA form for
//            8-bit frames) reproduces it bit for bit.  Otherwise (w0, w1, w2) = float32(k) and rs = 1 (round 5).
//   expand   X = along the rows:  even column  fma(2k0, N[j-1] + N[j+1], 2k2 * N[j]),  odd  2k1 * (N[j] + N[j+1])
//            then the same down the rows (the zero-stuffed grid's zero taps skipped)           (pyramid.py:34-46)
//   lap      G_l - expand(G_{l+1})                                                             (pyramid.py:133-138)
//   energy   gray is linear, so gray(lap) = gray(G_l) - expand(gray(G_{l+1})): the energy path runs on ONE channel:
//            Q = (gray(G_l) - expand(gray(G_{l+1})))^2, HB = s5 along the rows of Q, E = s5 down the rows of HB
//            (pyramid.py:49-50); the three-channel Laplacian is only needed for the winning frame of a pixel
//            and is filled in once per batch (sep_payload)
//
// This is NOT bit-identical to the exact-order mode (kernels_tiled.hpp: the C library's default and the audit mode
// `arith="exact"`; THIS arithmetic is what the Python entry points -- PyramidStack() and pipeline.py -- run by default):
// coefficients agree with a float64 evaluation within the forward-error bound of a 25-term float32 dot product, and the
// per-pixel arg-max can flip at near ties.  oracle/separable_oracle.c restates this arithmetic operation by operation
// (bit-exact parity target of this file); tests/test_sep_tolerance.py holds the tolerance against float64.
//
// Workgroup = 512 threads, tile 28 x 56 pixels of level l; per frame, four barrier phases:
//   P0 stage   G_l patch (tile + 6 halo = 40 x 68 px) registers -> LDS (prefetched one frame ahead)
//   P1 v-red   V (18 x 68 px): lane = one float4 column group, 7 ds_read_b128 -> 2 rows, packed fp32
//   P2 h-red   G_{l+1} patch (18 x 32): lane = one pixel, 32 lanes per row; the tile centre goes to global memory
//              from registers; gray of the pixel, its row neighbours by DPP wave shifts, and the lane writes the
//              horizontally expanded gray X (18 rows x 64 columns) -- G_{l+1} itself never sits in LDS
//   P3 lapq    lane = one 2x2 quad of the 32 x 64 (tile + 2 halo, + 2 dummy columns) region: vertical expand from
//              3 X rows, gray of the staged G_l, Q; the row blur of Q by DPP wave shifts -> HB in LDS
//   P4 select  lane's own quad: column blur of HB (6 ds_read_b64, packed), strict '>' against the running max
// Running state per pixel in registers: (max energy, its frame).
#pragma once
#include "common.hpp"
#include "kernels_tiled.hpp"

namespace mi {

// tile height (compile-time; tools build variants with -DMI_SEP_TH=..): 28 -> 512 threads, 3 workgroups per CU
#ifndef MI_SEP_TH
#define MI_SEP_TH 28
#endif
// threads per workgroup for fp32 input (all levels >= 1 and fp32 frames): 512 = one quad per lane; 576 adds a ninth wave so
// that the 18 x 32 items of P2 fit one round (with 512 the last two rows are a second round on wave 0 alone)
#ifndef MI_SEP_PF_SPLIT
#define MI_SEP_PF_SPLIT 0
#endif
#ifndef MI_SEP_NT_F32
#define MI_SEP_NT_F32 ((MI_SEP_TH / 2 + 2) * 32)
#endif
#ifndef MI_SEP_NT_INT
#define MI_SEP_NT_INT ((MI_SEP_TH / 2 + 2) * 32)
#endif
template <typename TIn> constexpr int sep_nt() { return sizeof(TIn) == 4 ? MI_SEP_NT_F32 : MI_SEP_NT_INT; }
// non-temporal G_{l+1} stores (written once, read by the next level's launch much later)
#ifndef MI_SEP_NT_STORE
#define MI_SEP_NT_STORE 1
#endif

// fp32 interior tiles: stage the G_l patch by LDS-DMA (`buffer_load_dwordx4 ... lds`: HBM -> LDS without a VGPR round trip,
// 1 KB per wave instruction) instead of prefetching into registers and writing them out.  The lane's quad takes its gray
// of G_l in P1 (registers), so the staged patch is dead after P1 and the next frame's DMA runs beside P2-P4.
// Bit-identical (all parity tests pass with it), 60 instead of 73 VGPRs -- and NOT faster: interleaved A/B on three boxes
// gave +1.0 %, -1.1 %, -1.4 % on the job (docs/studies.md, round 4): the kernel moves its bytes at the fabric's rate
// either way.  Off by default; -DMI_SEP_DMA=1 builds it.
// study knobs for the 8 / 16-bit instantiations (VALU-bound): waves per SIMD the compiler must leave room for (8 = 64
// VGPRs, 4 workgroups per CU; 6 = 85 VGPRs, 3 workgroups) and whether the per-frame addresses are laundered (rebuilt every
// frame) or may be hoisted into registers
#ifndef MI_SEP_INT_WAVES
#define MI_SEP_INT_WAVES 8
#endif
// MI_SEP_TOUCH n (study): touch one dword of every 128-byte line of the patch of frame b + n while frame b is worked on -- a
// software prefetch into L2 / the Infinity Cache that costs one register, so that the real loads of that frame (issued
// one frame ahead, 16 registers) find their lines on the way instead of in DRAM.  0 = off.
#ifndef MI_SEP_TOUCH
#define MI_SEP_TOUCH 0
#endif
// V rows a lane of P1 produces for 8 / 16-bit interior tiles (one float4 column group each): R rows read 2 R + 3 patch rows
// -- 3.5 conversions and LDS reads per output at R = 2, 2.75 at 4, 2.5 at 6 -- on correspondingly fewer lanes (the integer
// kernels are VALU-bound: what counts is the number of wave instructions, not how many waves share them).  When R does not
// divide the NH rows the last group starts at row NH - R: the rows it shares with the group before are written twice with
// the same bits.
#ifndef MI_SEP_P1_ROWS
#define MI_SEP_P1_ROWS 4
#endif
#ifndef MI_SEP_LAUNDER
#define MI_SEP_LAUNDER 1
#endif
// MF: interior tiles fetch their patch rows as 16-byte pieces (one / two loads per lane instead of four).  Measured slower
// (u8 level-0 launch 0.854 -> 0.901 ms: the pieces start 2 bytes off a dword boundary); off.
#ifndef MI_SEP_MF_WIDE
#define MI_SEP_MF_WIDE 0
#endif
#ifndef MI_SEP_MF_LAUNDER
#define MI_SEP_MF_LAUNDER 1
#endif
#ifndef MI_SEP_DMA
#define MI_SEP_DMA 0
#endif
// Level 0 of 8-bit frames with integer reduce taps (red_taps: the default generating kernel): the 5 x 5 reduce as exact
// integer arithmetic on the matrix pipe (v_mfma_i32_16x16x64_i8 on the staged bytes) instead of P1 + P2's VALU work -- see
// level_sep_body, "MF".  Bit-identical to the VALU form (all of tests/test_gpu_separable.py and test_gpu_parity.py pass with
// it) and NOT faster (round 5, docs/studies.md): 154 instead of 197 VALU instructions per wave and frame and one barrier
// less, but more LDS bank conflicts, and the kernel is bound by the sum of its phases (ablating the matrix instructions
// themselves changes nothing; the phase around them costs what P1 + P2 cost).  8-bit level-0 launch, interleaved A/B:
// 0.833 ms (VALU form) vs 0.853 ms (tile height 28, wave 0 takes a ninth row pair) and 0.854 ms (tile height 24, 17 % more
// workgroups; the VALU form at that height: 0.936 ms).  16-bit frames (two byte planes, ten matrix instructions per row
// pair): 0.962 -> 1.154 ms, never dispatched.  Off; -DMI_SEP_MFMA=1 builds it.
#ifndef MI_SEP_MFMA
#define MI_SEP_MFMA 0
#endif
#ifndef MI_SEP_MF_TH
#define MI_SEP_MF_TH 28
#endif
constexpr int SEP_MF_TH = MI_SEP_MF_TH, SEP_MF_NT = (MI_SEP_MF_TH / 2 + 2) * 32;

// the pair's tile-by-tile payload pass walks at most this many distinct winners per tile (level_sep_body, "PL")
constexpr int SEP_PL_MAXF = 32;
template <int TH_, int NT_>
struct SepGeom {
    static constexpr int TH = TH_, TW = 56, NT = NT_;
    static constexpr int GH = TH + 12, GW = TW + 12;          // G_l patch, pixels
    static constexpr int GD = GW * 3, GS = GD;                // dense rows: chunk id * 4 is the LDS offset
    static constexpr int NCH = GH * (GD / 4);                 // 16-byte chunks of the patch
    static constexpr int NPRE = (NCH + NT - 1) / NT;          // chunks (loads) per thread
    static constexpr int NH = TH / 2 + 4, NW = TW / 2 + 4;    // G_{l+1} patch: 18 x 32
    static constexpr int VS = GD;
    static constexpr int XW = 2 * NW, XS = XW;                // expanded GRAY columns x0-4 .. x0+TW+4, one float each
    static constexpr int QY = TH / 2 + 2, QL = NW;            // quad rows x lanes per quad row
    static constexpr int HBH = TH + 4, HBS = XW;              // HB rows y0-2 .. y0+TH+1, columns as X
    static constexpr int LDS_FLOATS = GH * GS + NH * VS + NH * XS;
    // Interior tiles of 8- and 16-bit frames keep the staged patch in its INPUT type (one / two dwords per 4-element chunk
    // instead of four): 27.5 / 35.6 KB of LDS instead of 51.9 and 64 VGPRs -> four workgroups per CU instead of three; P1
    // and P3 unpack on the fly.  Measured: level 0 of a 256 x 24 MP 8-bit stack 15.2 -> 14.5 ms (round 2); 16-bit frames
    // followed in round 3, once buffer addressing had brought that instantiation from 70 to 64 VGPRs.
    static constexpr int lds_floats(int esize, bool interior) {
        return (interior && esize <= 2 ? GH * (GD / 4) * esize : GH * GS) + NH * VS + NH * XS;
    }
    // MF (integer reduce on the matrix pipe): the staged patch as BYTE PLANES at a 208-byte row pitch (52 dwords: the MFMA
    // operand reads want 8-byte alignment) -- one plane for 8-bit frames, low bytes | high bytes for 16-bit frames --, then X,
    // HB (its own array: V does not exist) and the three weight operands (64 lanes x 16 bytes each)
    static constexpr int MF_CPR = 52, MF_PLANE = GH * MF_CPR;   // dwords
    static constexpr int lds_floats_mf(int esize) { return esize * MF_PLANE + NH * XS + HBH * HBS + 3 * 64 * 4; }
    static_assert(GD % 4 == 0 && NW == 32, "tile width is fixed by the 32-lane quad rows");
    static_assert(QY * QL <= NT, "one quad per lane (a ninth wave, if any, only stages, reduces and carries P2 items)");
    static_assert(HBH * HBS <= NH * VS, "HB aliases V");
    static_assert((NH / 2) * (GD / 4) <= NT && NH % 2 == 0 && NT % 64 == 0, "phase items");
};

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) Px3 { float v[3]; };   // one pixel: a 12-byte load / store

__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f s5(v2f a, v2f b, v2f c, v2f d, v2f e, float k0, float k1, float k2) {
    const v2f t0 = a + e, t1 = b + d;
    v2f m = c * k2;
    m = pk_fma((v2f)k1, t1, m);
    return pk_fma((v2f)k0, t0, m);
}
__device__ __forceinline__ float s5(float a, float b, float c, float d, float e, float k0, float k1, float k2) {
    const float t0 = a + e, t1 = b + d;
    return __builtin_fmaf(k0, t0, __builtin_fmaf(k1, t1, k2 * c));
}
// the reduce's tap order (see the header): the centre and outer taps first, the inner pair last -- with integer taps the last
// fma is then the only operation that can round (16-bit input: w0 (a + e) + w2 c <= 10 * 20 * 65535 < 2^24)
__device__ __forceinline__ v2f s5r(v2f a, v2f b, v2f c, v2f d, v2f e, float w0, float w1, float w2) {
    const v2f t0 = a + e, t1 = b + d;
    v2f m = c * w2;
    m = pk_fma((v2f)w0, t0, m);
    return pk_fma((v2f)w1, t1, m);
}
__device__ __forceinline__ float s5r(float a, float b, float c, float d, float e, float w0, float w1, float w2) {
    const float t0 = a + e, t1 = b + d;
    return __builtin_fmaf(w1, t1, __builtin_fmaf(w0, t0, w2 * c));
}
// expand, one dimension: even position from (left, centre, right), odd position from (centre, right)
__device__ __forceinline__ float ex_even(float l, float c, float r, float ce, float cc) {
    return __builtin_fmaf(ce, l + r, cc * c);
}
__device__ __forceinline__ float ex_odd(float c, float r, float co) { return co * (c + r); }
__device__ __forceinline__ v2f ex_even(v2f l, v2f c, v2f r, float ce, float cc) { return pk_fma((v2f)ce, l + r, c * cc); }
__device__ __forceinline__ v2f ex_odd(v2f c, v2f r, float co) { return (c + r) * co; }
// expand_layer(N)[y, x] for an hs x ws source given as N(row, col) (indices already inside the source), (y, x)
// inside the 2hs x 2ws grid: along the rows first, then down the rows
template <typename F>
__device__ __forceinline__ float expand_sep_of(F N, int hs, int ws, int y, int x, float ce, float cc, float co) {
    const int i = y >> 1, j = x >> 1;
    auto M = [&](int r, int q) { return N(map_expand_src(r, hs), map_expand_src(q, ws)); };
    auto X = [&](int r) { return (x & 1) ? ex_odd(M(r, j), M(r, j + 1), co) : ex_even(M(r, j - 1), M(r, j), M(r, j + 1), ce, cc); };
    return (y & 1) ? ex_odd(X(i), X(i + 1), co) : ex_even(X(i - 1), X(i), X(i + 1), ce, cc);
}

// one pixel of a level-l image as floats
template <typename TIn>
__device__ __forceinline__ void load_px3(const TIn* p, float o[3]) {
    if constexpr (sizeof(TIn) == 4) {
        const Px3 v = *(const Px3*)p;
        o[0] = v.v[0]; o[1] = v.v[1]; o[2] = v.v[2];
    } else {
        o[0] = to_f32(p[0]); o[1] = to_f32(p[1]); o[2] = to_f32(p[2]);
    }
}
// two grays at once, each component in gray_of<true>'s order
__device__ __forceinline__ v2f gray_of2(v2f b, v2f g, v2f r) {
    return pk_fma(r, (v2f)0.299f, pk_fma(g, (v2f)0.587f, b * 0.114f));
}

// n / D for 0 <= n < LIM with one 24-bit multiply and a shift (the compiler's n / 51 is a 32-bit mul_hi plus a 64-bit
// mad, both quarter rate); the magic is checked at compile time over the whole range
template <int D, int LIM>
struct SmallDiv {
    static constexpr uint32_t M = 65536 / D + 1;
    static constexpr bool ok() {
        for (int n = 0; n < LIM; ++n)
            if ((int)(((uint32_t)n * M) >> 16) != n / D) return false;
        return (uint64_t)LIM * M < (1u << 24);
    }
    static_assert(ok(), "magic does not cover the range");
    static __device__ __forceinline__ int div(int n) { return (int)(__umul24((uint32_t)n, M) >> 16); }
};
// 12 x: shift-and-add (the compiler picks the quarter-rate v_mul_lo_u32 for the constant)
__device__ __forceinline__ uint32_t times12(uint32_t x) {
    uint32_t x3;
    asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(x3) : "v"(x));
    return x3 << 2;
}

// 8-byte LDS load that stays a ds_read_b64 (2 LDS cycles per wave): the machine load/store optimiser otherwise pairs
// neighbouring ones into ds_read2_b64, which moves the same 16 bytes per lane in 8 cycles
__device__ __forceinline__ v2f lds_load2s(const float* p) {
    typedef const volatile v2f __attribute__((address_space(3))) * lds_ptr;   // volatile accesses are never paired
    return *(lds_ptr)(size_t)(uint32_t)(uintptr_t)p;                                 // generic LDS address: low 32 bits = offset
}

// ---- LDS-DMA (gfx950: 16 bytes per lane).  Inline assembly on purpose: with the builtin the compiler drains vmcnt at every
// workgroup barrier behind it (it cannot tell which LDS reads the transfer feeds); here the kernel places the wait itself.
__device__ __forceinline__ v4u make_rsrc_words(const void* base, uint32_t bytes) {   // raw buffer, as make_rsrc
    const uint64_t ad = (uint64_t)(uintptr_t)base;
    return v4u{(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ad),
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ad >> 32)) & 0xffffu, bytes, 0x00020000u};
}
// lane i of the wave: 16 bytes from buffer offset `voff` to LDS byte address lds_base + 16 i (lds_base wave-uniform)
__device__ __forceinline__ void lds_dma16(v4u rsrc, uint32_t voff, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :: "s"(lds_base), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void wait_vmem_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// value of lane-1 / lane+1 across the wave; lanes without a source get 0
__device__ __forceinline__ float dpp_wave_prev(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_wave_next(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// Buffer addressing for the per-frame streams: a 128-bit resource (uniform: base of the frame + its size, rebuilt per frame
// with scalar instructions) + ONE 32-bit lane offset per access -- no 64-bit lane addresses in the frame loop, and
// accesses outside [0, size) return zero instead of faulting (an edge tile's don't-care chunks may point anywhere).
typedef __amdgpu_buffer_rsrc_t BufRsrc;
__device__ __forceinline__ BufRsrc make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);   // gfx9 raw buffer
}

// 4 consecutive input elements as floats; raw form kept in registers across the frame loop for 8/16-bit input
template <typename TIn> struct PreChunk;
template <> struct PreChunk<float> {
    v4f v;
    __device__ __forceinline__ void store_raw(uint32_t*) const {}                       // never used: fp32 stages as fp32
    typedef uint32_t raw_t;
    static __device__ __forceinline__ raw_t ld_raw(const uint32_t*) { return 0u; }
    static __device__ __forceinline__ v4f cvt_raw(raw_t) { return v4f{}; }
    static __device__ __forceinline__ v4f unpack_raw(const uint32_t*) { return v4f{}; }
    static __device__ __forceinline__ void unpack6(const uint32_t*, int, v2f*) {}
    __device__ __forceinline__ void store_mf(uint32_t*, int) const {}
    __device__ __forceinline__ void set_raw(v4u, int) {}
    static __device__ __forceinline__ void store_mf16(const PreChunk*, uint32_t*, int) {}
    static __device__ __forceinline__ void unpack6_mf(const uint32_t*, int, int, v2f*) {}
    __device__ __forceinline__ void load(const char* p) { __builtin_memcpy(&v, p, 16); }
    __device__ __forceinline__ void load(BufRsrc r, uint32_t off) {
        v = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    }
    __device__ __forceinline__ void set(float a, float b, float c, float d) { v = v4f{a, b, c, d}; }
    __device__ __forceinline__ v4f get() const { return v; }
    __device__ __forceinline__ void rot2() { v = v4f{v.z, v.w, v.x, v.y}; }   // rotate by two elements
};
template <> struct PreChunk<uint8_t> {
    uint32_t v;
    // raw LDS form: one dword = four elements
    __device__ __forceinline__ void store_raw(uint32_t* q) const { *q = v; }
    typedef uint32_t raw_t;   // four elements as staged
    static __device__ __forceinline__ raw_t ld_raw(const uint32_t* q) {
        return *(const volatile uint32_t __attribute__((address_space(3)))*)(size_t)(uint32_t)(uintptr_t)q;
    }
    static __device__ __forceinline__ v4f cvt_raw(raw_t w) {
        return v4f{(float)(w & 0xffu), (float)((w >> 8) & 0xffu), (float)((w >> 16) & 0xffu), (float)(w >> 24)};
    }
    static __device__ __forceinline__ v4f unpack_raw(const uint32_t* q) { return cvt_raw(ld_raw(q)); }
    // six consecutive elements (two pixels) starting at element `e0` (even) of the row at `row`, channel by channel:
    // out[c] = (pixel 0, pixel 1) of channel c.  v_alignbyte takes the byte offset from the low two bits of e0.
    static __device__ __forceinline__ void unpack6(const uint32_t* row, int e0, v2f* out) {
        const uint32_t* q = row + (e0 >> 2);
        const uint32_t q0 = q[0], q1 = q[1];
        const uint32_t lo = __builtin_amdgcn_alignbyte(q1, q0, (uint32_t)e0), hi = __builtin_amdgcn_alignbyte(0u, q1, (uint32_t)e0);
        out[0] = v2f{(float)(lo & 0xffu), (float)(lo >> 24)};
        out[1] = v2f{(float)((lo >> 8) & 0xffu), (float)(hi & 0xffu)};
        out[2] = v2f{(float)((lo >> 16) & 0xffu), (float)((hi >> 8) & 0xffu)};
    }
    // MF form: the chunk's dword in the (single) byte plane, every byte as x ^ 0x80 = the signed byte x - 128 the MFMA takes
    __device__ __forceinline__ void store_mf(uint32_t* q, int) const { *q = v ^ 0x80808080u; }
    __device__ __forceinline__ void set_raw(v4u t, int i) { v = t[i]; }   // dword i of a 16-byte piece
    // four chunks = one 16-byte piece, one ds_write_b128
    static __device__ __forceinline__ void store_mf16(const PreChunk* c, uint32_t* q, int) {
        *reinterpret_cast<v4u*>(q) = v4u{c[0].v ^ 0x80808080u, c[1].v ^ 0x80808080u, c[2].v ^ 0x80808080u, c[3].v ^ 0x80808080u};
    }
    static __device__ __forceinline__ void unpack6_mf(const uint32_t* row, int, int e0, v2f* out) {
        const uint32_t* q = row + (e0 >> 2);
        const uint32_t q0 = q[0], q1 = q[1];
        const uint32_t lo = __builtin_amdgcn_alignbyte(q1, q0, (uint32_t)e0) ^ 0x80808080u;
        const uint32_t hi = __builtin_amdgcn_alignbyte(0u, q1, (uint32_t)e0) ^ 0x00008080u;
        out[0] = v2f{(float)(lo & 0xffu), (float)(lo >> 24)};
        out[1] = v2f{(float)((lo >> 8) & 0xffu), (float)(hi & 0xffu)};
        out[2] = v2f{(float)((lo >> 16) & 0xffu), (float)((hi >> 8) & 0xffu)};
    }
    __device__ __forceinline__ void load(const char* p) { __builtin_memcpy(&v, p, 4); }
    __device__ __forceinline__ void load(BufRsrc r, uint32_t off) { v = __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0); }
    __device__ __forceinline__ void rot2() { v = (v >> 16) | (v << 16); }
    __device__ __forceinline__ void set(float a, float b, float c, float d) {
        v = (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24);
    }
    __device__ __forceinline__ v4f get() const {
        return v4f{(float)(v & 0xffu), (float)((v >> 8) & 0xffu), (float)((v >> 16) & 0xffu), (float)(v >> 24)};
    }
};
template <> struct PreChunk<uint16_t> {
    uint32_t v0, v1;
    // raw LDS form: two dwords = four elements
    __device__ __forceinline__ void store_raw(uint32_t* q) const { *reinterpret_cast<v2u*>(q) = v2u{v0, v1}; }
    typedef v2u raw_t;
    static __device__ __forceinline__ raw_t ld_raw(const uint32_t* q) {
        return *(const volatile v2u __attribute__((address_space(3)))*)(size_t)(uint32_t)(uintptr_t)q;
    }
    static __device__ __forceinline__ v4f cvt_raw(raw_t w) {
        return v4f{(float)(w.x & 0xffffu), (float)(w.x >> 16), (float)(w.y & 0xffffu), (float)(w.y >> 16)};
    }
    static __device__ __forceinline__ v4f unpack_raw(const uint32_t* q) { return cvt_raw(ld_raw(q)); }
    // six consecutive elements (two pixels) starting at element `e0` (even) of the row at `row`, channel by channel
    static __device__ __forceinline__ void unpack6(const uint32_t* row, int e0, v2f* out) {
        const uint32_t* q = row + (e0 >> 1);
        const uint32_t q0 = q[0], q1 = q[1], q2 = q[2];
        out[0] = v2f{(float)(q0 & 0xffffu), (float)(q1 >> 16)};
        out[1] = v2f{(float)(q0 >> 16), (float)(q2 & 0xffffu)};
        out[2] = v2f{(float)(q1 & 0xffffu), (float)(q2 >> 16)};
    }
    // MF form: the four low bytes -> plane 0, the four high bytes -> plane 1 (`plane` dwords further), each as x ^ 0x80
    __device__ __forceinline__ void store_mf(uint32_t* q, int plane) const {
        q[0] = __builtin_amdgcn_perm(v1, v0, 0x06040200u) ^ 0x80808080u;
        q[plane] = __builtin_amdgcn_perm(v1, v0, 0x07050301u) ^ 0x80808080u;
    }
    __device__ __forceinline__ void set_raw(v4u t, int i) { v0 = t[2 * i]; v1 = t[2 * i + 1]; }   // chunk i of a 16-byte piece
    // two chunks = one 16-byte piece: 8 bytes into each plane
    static __device__ __forceinline__ void store_mf16(const PreChunk* c, uint32_t* q, int plane) {
        *reinterpret_cast<v2u*>(q) = v2u{__builtin_amdgcn_perm(c[0].v1, c[0].v0, 0x06040200u) ^ 0x80808080u,
                                         __builtin_amdgcn_perm(c[1].v1, c[1].v0, 0x06040200u) ^ 0x80808080u};
        *reinterpret_cast<v2u*>(q + plane) = v2u{__builtin_amdgcn_perm(c[0].v1, c[0].v0, 0x07050301u) ^ 0x80808080u,
                                                 __builtin_amdgcn_perm(c[1].v1, c[1].v0, 0x07050301u) ^ 0x80808080u};
    }
    // six consecutive elements from the two byte planes: bytes realigned per plane, re-paired into 16-bit values
    static __device__ __forceinline__ void unpack6_mf(const uint32_t* row, int plane, int e0, v2f* out) {
        const uint32_t* q = row + (e0 >> 2);
        const uint32_t l0 = q[0], l1 = q[1], h0 = q[plane], h1 = q[plane + 1];
        const uint32_t La = __builtin_amdgcn_alignbyte(l1, l0, (uint32_t)e0) ^ 0x80808080u;
        const uint32_t Lb = __builtin_amdgcn_alignbyte(0u, l1, (uint32_t)e0) ^ 0x00008080u;
        const uint32_t Ha = __builtin_amdgcn_alignbyte(h1, h0, (uint32_t)e0) ^ 0x80808080u;
        const uint32_t Hb = __builtin_amdgcn_alignbyte(0u, h1, (uint32_t)e0) ^ 0x00008080u;
        const uint32_t p01 = __builtin_amdgcn_perm(Ha, La, 0x05010400u), p23 = __builtin_amdgcn_perm(Ha, La, 0x07030602u);
        const uint32_t p45 = __builtin_amdgcn_perm(Hb, Lb, 0x05010400u);
        out[0] = v2f{(float)(p01 & 0xffffu), (float)(p23 >> 16)};
        out[1] = v2f{(float)(p01 >> 16), (float)(p45 & 0xffffu)};
        out[2] = v2f{(float)(p23 & 0xffffu), (float)(p45 >> 16)};
    }
    __device__ __forceinline__ void load(const char* p) {
        uint64_t t;
        __builtin_memcpy(&t, p, 8);
        v0 = (uint32_t)t;
        v1 = (uint32_t)(t >> 32);
    }
    __device__ __forceinline__ void load(BufRsrc r, uint32_t off) {
        const v2u t = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
        v0 = t.x;
        v1 = t.y;
    }
    __device__ __forceinline__ void rot2() { const uint32_t t = v0; v0 = v1; v1 = t; }
    __device__ __forceinline__ void set(float a, float b, float c, float d) {
        v0 = (uint32_t)a | ((uint32_t)b << 16);
        v1 = (uint32_t)c | ((uint32_t)d << 16);
    }
    __device__ __forceinline__ v4f get() const {
        return v4f{(float)(v0 & 0xffffu), (float)(v0 >> 16), (float)(v1 & 0xffffu), (float)(v1 >> 16)};
    }
};

typedef int v4i __attribute__((ext_vector_type(4)));

// Which tile does this workgroup own?  (interior: 8x8-tile super-blocks, one per XCD at a time -- see kernels_tiled.hpp;
// border: every tile of the TH x TW grid outside the interior rectangle.)  false: none.
template <bool INTERIOR, int TH, int TW>
__device__ __forceinline__ bool sep_tile_origin(const LevelArgs& a, int& y0, int& x0) {
    const int h = a.h, w = a.w;
    if constexpr (INTERIOR) {
        const int nty = (a.iy1 - a.iy0) / TH, ntx = (a.ix1 - a.ix0) / TW;
        const int sb_x = (ntx + SEP_SBW - 1) / SEP_SBW;
        // frame chunks (blockIdx.y) rotate the XCD a super-block runs on: a small level has fewer super-blocks than
        // the GPU has XCDs, and its chunks would otherwise all queue up on the same few
        const int xcd = (blockIdx.x + 8 - (blockIdx.y & 7)) & 7, slot = blockIdx.x >> 3;
        constexpr int SBN = SEP_SBW * SEP_SBH;
        int S = (slot / SBN) * 8 + xcd;
        const int within = slot % SBN;
        if (a.sb_order) {
            S = a.sb_order[S];
            if (S == 0xFFFF) return false;
        }
        const int sby = S / sb_x, sbx = S - sby * sb_x;
        const int tyi = sby * SEP_SBH + within / SEP_SBW, txi = sbx * SEP_SBW + within % SEP_SBW;
        if (tyi >= nty || txi >= ntx) return false;
        y0 = a.iy0 + tyi * TH;
        x0 = a.ix0 + txi * TW;
        if (y0 >= h || x0 >= w) return false;
    } else {
        // every tile of the TH x TW grid that is not inside the interior rectangle: rows above and below it,
        // then the side columns of the rows it spans
        const int tiles_x = (w + TW - 1) / TW, tiles_y = (h + TH - 1) / TH;
        const int ty_lo = a.iy0 / TH, ty_hi = a.iy1 / TH, tx_lo = a.ix0 / TW, tx_hi = a.ix1 / TW;
        const int nyi = ty_hi - ty_lo, nxi = tx_hi - tx_lo;
        int t = blockIdx.x, tyi, txi;
        const int top = ty_lo * tiles_x, bot = (tiles_y - ty_hi) * tiles_x;
        if (nyi <= 0 || nxi <= 0 || t < top) {
            tyi = t / tiles_x;
            txi = t - tyi * tiles_x;
        } else if (t < top + bot) {
            t -= top;
            tyi = ty_hi + t / tiles_x;
            txi = t - (t / tiles_x) * tiles_x;
        } else {
            t -= top + bot;
            const int side = tiles_x - nxi;
            tyi = ty_lo + t / side;
            const int k = t - (t / side) * side;
            txi = k < tx_lo ? k : tx_hi + (k - tx_lo);
        }
        if (tyi >= tiles_y || txi >= tiles_x) return false;
        y0 = tyi * TH;
        x0 = txi * TW;
    }

    return true;
}


template <typename TIn, bool INTERIOR, int TH, int NT, bool MF_ = false, bool PAIR_ = false, bool PL_ = false>
__device__ __forceinline__ void level_sep_body(const LevelArgs& a) {
    using G = SepGeom<TH, NT>;
    constexpr int TW = G::TW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr bool RAW = INTERIOR && sizeof(TIn) <= 2;   // staged patch kept in the input type (see SepGeom::lds_floats)
    constexpr bool DMA = MI_SEP_DMA && INTERIOR && sizeof(TIn) == 4;   // patch staged by LDS-DMA (see MI_SEP_DMA)
    // PAIR (round 6): levels l and l + 1 as a pair.  The 18 x 32 patch of G_{l+1} this tile computes anyway is exactly the
    // support of the tile's 7 x 14 pixels of G_{l+2} (rows 2m - 2 .. 2m + 2 of the patch's 17 leading rows), and the energy
    // path of level l + 1 needs gray(G_{l+1}) only: the kernel keeps the three-channel patch in LDS (planar, sN -- where the
    // staged G_l patch was: the lane's quad takes its gray of G_l in P1, so that patch is dead after P1), reduces it to
    // G_{l+2} (V2 beside P3, H2 + the store beside P4; REFLECT101 on G_{l+1} through explicit index maps: a patch's
    // natural twins beyond an even far edge are those of the zero-stuffed expand grid, not of the reduce) and writes
    // gray(G_{l+1}) -- 4 bytes per pixel instead of 12 -- for level_sep_e.  G_{l+1} itself never reaches HBM (one frame per
    // launch excepted: LevelArgs::g1_keep, the debug tap); the payload passes recompute the winners' (sep_payload_pair).
    constexpr bool PAIR = PAIR_;
    static_assert(!PAIR || (!MF_ && !DMA && TH % 4 == 0 && G::NH * G::NW * 3 <= G::GH * (G::GD / 4)), "PAIR geometry");
    // PL (round 6): the PAYLOAD pass of a pair, tile by tile.  The workgroup collects the distinct winners of its tile (level
    // l: the tile's pixels; level l + 1: the tile's 14 x 28 pixels of it) -- at most SEP_PL_MAXF, else the tile is flagged and
    // left to the per-quad kernels (sep_payload_pair0 / 1) -- and walks THOSE frames: P0 - P2 as ever give the frame's
    // G_{l+1} patch (sN), from which the lanes fill in lap_l = G_l - expand(G_{l+1}) of the pixels that frame won and
    // lap_{l+1} = G_{l+1} - expand(G_{l+2}) likewise.  A tile whose pixels have one winner costs one tile-frame of P0 - P2.
    constexpr bool PL = PL_;
    static_assert(!PL || (!PAIR && !MF_ && !DMA), "PL geometry");
    constexpr int N2H = TH / 4, N2W = TW / 4, NPL1 = G::NH * G::NW;   // G_{l+2} pixels of the tile; floats per plane of sN
    // fp32 interior tiles: the G_{l+1} patch goes where the staged G_l patch was, whose last reader -- the gray of the lane's quad
    // of G_l -- moves from P3 to P1 (registers).  8 / 16-bit tiles keep that gray in P3 (their P1 runs on half the waves: it is
    // the critical path of its phase) and give the patch its own LDS (8-bit: 34.4 KB, still four workgroups per CU).
    constexpr bool SN_ALIAS = PAIR && INTERIOR && sizeof(TIn) == 4;
    constexpr bool GQ1 = DMA || SN_ALIAS;   // gray of the lane's quad of G_l taken in P1 (registers)
    constexpr int RD = (int)sizeof(TIn);                 // dwords per 4-element chunk in that form
    // MF: the reduce of 8 / 16-bit frames on the matrix pipe.  P1 + P2 become one phase: wave v computes the G_{l+1} patch
    // rows 2v, 2v + 1 (16 rows = 8 waves) with five v_mfma_i32_16x16x64_i8 per byte plane -- one per tap row -- whose DATA
    // operand (B, 64 x 16) is read straight from the staged bytes: column n = a 64-byte window of a patch row, 24 bytes (four
    // output pixels) further along the row per window, 8 windows per row, two rows; and whose WEIGHT operand (A, 16 x 64)
    // holds kv[t] * kh[tau] at byte 6 p + 3 tau + c of the window in row 4 p + c (output pixel p of the window, channel c;
    // row 4 p + 3 is empty).  The result layout then hands lane 16 p + n the three channel sums of ONE G_{l+1} pixel.  The
    // staged bytes carry x ^ 0x80 (= x - 128 as the signed byte the instruction takes) and the accumulator starts at
    // 128 * 400.  16-bit frames: the same on the plane of low bytes and on the plane of high bytes, S = 256 S_hi + S_lo.
    // S is the exact integer sum and G_{l+1} = float(S) * rs: bit-identical to the float evaluation of red_taps' integer
    // taps (header; float(S) rounds once, to nearest even, exactly where the float chain's last fma does).
    static_assert(!MF_ || (INTERIOR && sizeof(TIn) <= 2 && G::NH % 2 == 0 && NT % 64 == 0), "MF geometry");
    constexpr bool MF = MF_;
    constexpr int NPL = MF ? (int)sizeof(TIn) : 0;    // byte planes
    float* sG = smem;
    uint32_t* sGr = reinterpret_cast<uint32_t*>(smem);
    float* sV = smem + (MF ? 0 : G::lds_floats((int)sizeof(TIn), INTERIOR) - G::NH * G::VS - G::NH * G::XS);
    float* sX = MF ? smem + NPL * G::MF_PLANE : sV + G::NH * G::VS;
    float* sHB = MF ? sX + G::NH * G::XS : sV;         // (not MF: V is dead once P2 has read it)
    uint32_t* sW = reinterpret_cast<uint32_t*>(sHB + G::HBH * G::HBS);   // MF: weight operands, [3][64] x 16 bytes
    // PAIR: G_{l+1} patch, [3][NH][NW] (interior tiles: over the dead G_l patch; border tiles and PL: their own array)
    float* sN = (SN_ALIAS || (PL && INTERIOR)) ? smem : smem + (INTERIOR ? G::lds_floats((int)sizeof(TIn), true) : G::LDS_FLOATS);
    // PL: [0..7] winner bit map, [8] count, [16..] frame list -- behind everything else (interior tiles: the patch is over G_l's)
    uint32_t* sFL = reinterpret_cast<uint32_t*>(smem + (INTERIOR ? G::lds_floats((int)sizeof(TIn), true) : G::LDS_FLOATS + 3 * G::NH * G::NW));
    float* sV2 = sV + G::HBH * G::HBS;                   // PAIR: column sums of the G_{l+2} reduce, [3][N2H][NW] (V's tail)
    static_assert(!PAIR || G::HBH * G::HBS + 3 * N2H * G::NW <= G::NH * G::VS, "V2 fits behind HB");
    const int tid = threadIdx.x;
    const int h = a.h, w = a.w, hn = a.hn, wn = a.wn;

    int y0, x0;
    if (!sep_tile_origin<INTERIOR, TH, TW>(a, y0, x0)) return;

    const float k0 = a.k1d[0], k1 = a.k1d[1], k2 = a.k1d[2];
    const float w0 = a.rk[0], w1 = a.rk[1], w2 = a.rk[2], rs = a.rk[3];   // reduce taps and final scale (red_taps)
    const float ce = 2.0f * k0, cc = 2.0f * k2, co = 2.0f * k1;   // expand taps (the reference's 4 * K, per dimension)
    if constexpr (MF) {
        // weight operands: class 0 = tap rows 0 and 4 (kv = w0), 1 = rows 1 and 3 (w1), 2 = row 2 (w2); lane (row s = lane & 15,
        // k-group g = lane >> 4) holds the bytes k = 16 g .. 16 g + 15 of row s.  Visible after the first frame's barrier.
        if (tid < 192) {
            const int cls = tid >> 6, ln = tid & 63, sl = ln & 15, kg = ln >> 4, p = sl >> 2, c = sl & 3;
            const int wi[3] = {(int)w0, (int)w1, (int)w2};
            const int kv = wi[cls];
            uint32_t wd[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int d = 16 * kg + i - 6 * p - c;
                int wt = 0;
#pragma unroll
                for (int tau = 0; tau < 5; ++tau)
                    if (c < 3 && d == 3 * tau) wt = kv * wi[tau < 3 ? tau : 4 - tau];
                wd[i >> 2] |= (uint32_t)(wt & 0xff) << (8 * (i & 3));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) sW[4 * tid + i] = wd[i];
        }
    }

    // ---- which frames?  (levels with few tiles split the batch into chunks along blockIdx.y, see LevelArgs)
    const int ck = blockIdx.y, f_lo = ck * a.chunk_frames;
    int nfr = min(a.nframes - f_lo, a.chunk_frames);
    const bool fresh = ck > 0 || a.first;
    float* const st_e = ck ? a.part_e + (size_t)(ck - 1) * a.part_stride : a.best_e;
    int32_t* const st_i = ck ? a.part_idx + (size_t)(ck - 1) * a.part_stride : a.best_idx;
    const char* const src0 = (const char*)a.src + (size_t)f_lo * a.src_stride;
    // the frame the loop's step b works on: b itself, or (PL) the b-th distinct winner of the tile
    auto fid = [&](int b) { return PL ? __builtin_amdgcn_readfirstlane((int)sFL[16 + b]) : b; };
    float* const gnext0 = PAIR ? a.gnext : a.gnext + (size_t)f_lo * a.gnext_stride;   // (PAIR: one image, see g1_keep)
    float* const gray1_0 = PAIR ? a.gray1 + (size_t)f_lo * a.gray1_stride : nullptr;
    float* const g2_0 = PAIR ? a.g2 + (size_t)f_lo * a.g2_stride : nullptr;

    // ---- the lane's quad: rows y0-2+2qy+{0,1}, columns x0-4+2ql+{0,1}; owned = inside the tile.  Running state of
    // the quad = (max energy, its frame); the winner's Laplacian is filled in after the batch (sep_payload).
    const int qy = tid >> 5, ql = tid & 31;
    const bool own_tile = qy >= 1 && qy < G::QY - 1 && ql >= 2 && ql < G::QL - 2;
    const int oy = y0 - 2 + 2 * qy, ox = x0 - 4 + 2 * ql;
    float bE[4];
    int bI[4];
    int f1 = -1;   // PL: the winner (frame of this batch, or -1) of the lane's pixel of level l + 1
    if constexpr (PL) {
        // bI[p] = the winner of the quad's pixel p as a frame of this batch (-1: none of them / outside the image)
        if (tid < 16) sFL[tid] = 0u;
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int y = oy + (p >> 1), x = ox + (p & 1);
            int fr = -1;
            if (own_tile && y < h && x < w) {
                fr = a.best_idx[(size_t)y * w + x] - a.frame_idx0;
                if (fr < 0 || fr >= a.nframes) fr = -1;
                else atomicOr(&sFL[fr >> 5], 1u << (fr & 31));
            }
            bI[p] = fr;
            bE[p] = 0.f;
        }
        if (tid < (TH / 2) * (TW / 2)) {
            const int r1 = SmallDiv<TW / 2, NT>::div(tid), c1 = tid - r1 * (TW / 2);
            const int i1 = y0 / 2 + r1, j1 = x0 / 2 + c1;
            if (i1 < hn && j1 < wn) {
                f1 = a.idx1[(size_t)i1 * wn + j1] - a.frame_idx0;
                if (f1 < 0 || f1 >= a.nframes) f1 = -1;
                else atomicOr(&sFL[f1 >> 5], 1u << (f1 & 31));
            }
        }
        __syncthreads();
        if (tid < 256) {
            const uint32_t wd = sFL[tid >> 5], bit = 1u << (tid & 31);
            if (wd & bit) {
                int pos = __builtin_popcount(wd & (bit - 1u));
                for (int k = 0; k < (tid >> 5); ++k) pos += __builtin_popcount(sFL[k]);
                if (pos < SEP_PL_MAXF) sFL[16 + pos] = (uint32_t)tid;
            }
        }
        if (tid == 0) {
            int n = 0;
            for (int k = 0; k < 8; ++k) n += __builtin_popcount(sFL[k]);
            sFL[8] = (uint32_t)n;
        }
        __syncthreads();
        nfr = __builtin_amdgcn_readfirstlane((int)sFL[8]);
        if (nfr > SEP_PL_MAXF) {   // too many winners: the per-quad kernels take this tile
            if (tid == 0) a.tile_flag[(y0 / TH) * ((w + TW - 1) / TW) + x0 / TW] = 1;
            return;
        }
        if (nfr == 0) return;
    } else {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int y = oy + (p >> 1), x = ox + (p & 1);
        const bool valid = own_tile && y < h && x < w;   // edge tiles may overhang the image
        if (!fresh && valid) {
            const size_t px = (size_t)y * w + x;
            bE[p] = st_e[px];
            bI[p] = -1;      // the running arg-max is not read: -1 = "no frame of this launch has won (yet)"
        } else {
            bE[p] = -__builtin_inff();   // the first frame always wins (energies can be negative: a generating kernel with a > 0.5 has negative taps)
            bI[p] = -1;
        }
    }
    }

    // ---- staging: chunk id = tid + n * NT covers patch row id / 51, floats 4 * (id % 51) .. +3
    //
    // The INTERIOR instantiation serves interior tiles AND edge tiles (`edge`, uniform per workgroup).  An edge tile puts
    // the REFLECT101 image of its surroundings into the staged patch and then runs the interior code unchanged: the cells
    // of G_{l+1}, X and Q that code computes beyond an image edge are then the bit-identical twins of the cells
    // REFLECT101 prescribes (on the zero-stuffed grid for the expand, on Q for the blur) -- s5 and the expand taps are
    // symmetric and float addition commutes -- PROVIDED the far edge the tile touches has an even size (the mirror about
    // row n-1 = 2(n/2)-1 then maps G_{l+1} row n/2+k onto row n/2-1-k; an odd size does not, and those tiles stay with the
    // general border instantiation: the host draws the line, launch_level_sep).  Pinned on the CPU by
    // tests/test_oracle.py::test_sep_natural_extension and on the GPU by the odd/even size cases of test_gpu_separable.py.
    // Staging an edge tile: mirrored ROWS arrive through the row part of the chunk offsets; mirrored COLUMNS are copied
    // inside LDS after the staging barrier (6 columns per side; edge tiles pay one more barrier per frame).  Patch rows
    // start 2 elements off the 4-element chunk grid of an image row, so the chunk that straddles an image edge holds 2
    // in-image elements: it is loaded 2 elements further inside the row and stored rotated by 2.
    constexpr int CPR = MF ? G::MF_CPR : G::GD / 4;   // chunks per patch row (MF: one more, for a 208-byte LDS row pitch)
    constexpr int NCH = G::GH * CPR, NPRE = (NCH + NT - 1) / NT;
    constexpr int W16 = 13 * (int)sizeof(TIn), NW16 = G::GH * W16, NPW = (NW16 + NT - 1) / NT;   // MF: 16-byte pieces (row, patch, per lane)
    constexpr int CP16 = 4 / (int)sizeof(TIn);                                                     // chunks per piece
    PreChunk<TIn> pre[NPRE];
    uint32_t goff[NPRE];          // interior / edge: byte offset of the chunk inside a frame
    bool edge = false;
    int colL = -1, colR = -1;        // edge: the straddling chunk column at the left / right image edge (-1: none)
    int peR = -1;                    // edge: patch column of the image's last column when the patch crosses it
    if constexpr (INTERIOR) {
        edge = y0 < 6 || x0 < 6 || y0 + TH + 6 > h || x0 + TW + 6 > w;
        const int rel = (w - (x0 - 6)) * 3;               // elements from the patch row start to the image row end
        if (edge) {
            if (x0 < 6) colL = 4;                          // elements 16 .. 19: columns -1 | 0
            if (rel < G::GD) { peR = w - 1 - (x0 - 6); if ((rel & 3) == 2) colR = (rel - 2) >> 2; }
        }
        if (MF && MI_SEP_MF_WIDE && !edge) {
            // MF, tiles that mirror nothing: the patch row as 16-byte pieces (four / two chunks), one / two loads per lane
            // instead of four (the 208-byte row pitch makes the pieces 16-byte aligned in LDS)
#pragma unroll
            for (int m = 0; m < NPW; ++m) {
                const int id = tid + m * NT, row = id / W16, c16 = id - row * W16;
                goff[m] = (uint32_t)(((y0 - 6 + row) * w + (x0 - 6)) * 3 * (int)sizeof(TIn) + 16 * c16);
            }
        } else
#pragma unroll
        for (int n = 0; n < NPRE; ++n) {
            const int id = tid + n * NT, row = id / CPR, col = id - row * CPR;
            int e0 = 4 * col;                              // first element of the chunk, relative to the patch row start
            int gy = y0 - 6 + row;
            if (edge) {
                gy = map_clamp(gy, h);
                if (col == colL) e0 += 2;
                else if (col == colR) e0 -= 2;
                e0 = clampi(e0, -(x0 - 6) * 3, rel - 4);   // chunks outside the image: anything inside the row
            }
            goff[n] = (uint32_t)((gy * w + (x0 - 6)) * 3 + e0) * (uint32_t)sizeof(TIn);
        }
    }
    // PAIR: does the G_{l+2} reduce of this tile touch an edge of G_{l+1} (rows / columns through REFLECT101, pixels of G_{l+2}
    // outside its image)?  Uniform per workgroup; the tiles that do not take the unmapped fast paths of V2 / H2.
    const bool e2 = !INTERIOR || y0 < 4 || x0 < 4 || y0 / 2 + TH / 2 >= hn || x0 / 2 + TW / 2 >= wn;
    const uint32_t frame_bytes = (uint32_t)h * (uint32_t)w * 3u * (uint32_t)sizeof(TIn);
    // chunks [n0, n1) of frame b
    auto prefetch = [&](int b, int n0, int n1) {
        const char* frb = src0 + (size_t)fid(b) * a.src_stride;
        const BufRsrc rs = make_rsrc(frb, frame_bytes);
        if constexpr (MF) {
            if (MI_SEP_MF_WIDE && !edge) {
#pragma unroll
                for (int m = 0; m < NPW; ++m) {
                    if ((m + 1) * NT > NW16 && tid + m * NT >= NW16) continue;
                    const v4u t = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rs, goff[m], 0, 0));
                    pre[CP16 * m].set_raw(t, 0);
                    if constexpr (CP16 >= 2) pre[CP16 * m + 1].set_raw(t, 1);
                    if constexpr (CP16 == 4) { pre[4 * m + 2].set_raw(t, 2); pre[4 * m + 3].set_raw(t, 3); }
                }
                return;
            }
        }
#pragma unroll
        for (int n = 0; n < NPRE; ++n) {
            if (n < n0 || n >= n1) continue;
            const int id = tid + n * NT;
            if ((n + 1) * NT > NCH && id >= NCH) continue;
            if constexpr (INTERIOR) {
                pre[n].load(rs, goff[n]);
            } else {
                const TIn* fr = (const TIn*)frb;
                const int row = id / CPR, col = id - row * CPR;
                const int gy = map_clamp(y0 - 6 + row, h);
                const int c_lo = (4 * col) / 3, c_hi = (4 * col + 3) / 3;
                if (x0 - 6 + c_lo >= 0 && x0 - 6 + c_hi < w) {
                    pre[n].load((const char*)(fr + ((size_t)gy * w + (x0 - 6)) * 3 + 4 * col));
                } else {
                    float e[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int f = 4 * col + k, pc = f / 3, c = f - pc * 3;
                        e[k] = to_f32(fr[((size_t)gy * w + map_clamp(x0 - 6 + pc, w)) * 3 + c]);
                    }
                    pre[n].set(e[0], e[1], e[2], e[3]);
                }
            }
        }
    };
    // LDS-DMA form of the same: chunk id = tid + n * NT lands at sG + 4 * id, i.e. wave-uniform base + 16 * lane
    const uint32_t dma_base = (uint32_t)(uintptr_t)sG + 1024u * (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
    auto dma_issue = [&](int b) {
        if constexpr (DMA) {
            const v4u rs = make_rsrc_words(src0 + (size_t)b * a.src_stride, frame_bytes);
#pragma unroll
            for (int n = 0; n < NPRE; ++n) {
                if ((n + 1) * NT > NCH && tid + n * NT >= NCH) continue;
                lds_dma16(rs, goff[n], dma_base + 16u * (uint32_t)(n * NT));
            }
        }
    };
    if constexpr (DMA) dma_issue(0);
    else prefetch(0, 0, NPRE);
    constexpr bool TOUCH = MI_SEP_TOUCH > 0 && INTERIOR && !DMA;
    constexpr int TOUCH_PER_ROW = (G::GD * (int)sizeof(TIn) + 127) / 128 + 1;   // lines a patch row can straddle
    uint32_t touch_off = 0, touch_val = 0, touch_acc = 0;
    const bool touch_on = TOUCH && !edge && tid < G::GH * TOUCH_PER_ROW;
    if (touch_on) {
        const int row = tid / TOUCH_PER_ROW, j = tid - row * TOUCH_PER_ROW;
        touch_off = (uint32_t)(((y0 - 6 + row) * w + (x0 - 6)) * 3) * (uint32_t)sizeof(TIn) +
                    (uint32_t)min(128 * j, G::GD * (int)sizeof(TIn) - 4);
        touch_off &= ~3u;
    }
    v2f gq_e = {0.f, 0.f}, gq_o = {0.f, 0.f};   // DMA: gray of the lane's quad (rows 2qy+4, +5 of the patch), taken in P1
#ifdef MI_PHASE_CLOCK
    unsigned int pc_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pc_last = (unsigned int)clock64();
#endif

    for (int b = 0; b < nfr; ++b) {
        int lt = tid;
#if MI_SEP_LAUNDER
        if (!MF || MI_SEP_MF_LAUNDER) asm volatile("" : "+v"(lt));   // per-frame addresses are rebuilt from this, not hoisted out of the loop
#endif
        // ---------------- P0: stage
        if constexpr (DMA) {
            wait_vmem_all();   // this wave's share of the patch has landed (and the previous frame's G_{l+1} stores)
            if (edge) {
                // the chunk that straddles an image edge was fetched 2 elements further in: rotate it in place, then the
                // mirrored columns as below
#pragma unroll
                for (int n = 0; n < NPRE; ++n) {
                    const int id = lt + n * NT;
                    if ((n + 1) * NT > NCH && id >= NCH) continue;
                    const int col = id - (id / CPR) * CPR;
                    if (col == colL || col == colR) {
                        const v4f c = lds_load4(sG + 4 * id);
                        lds_store4(sG + 4 * id, c.z, c.w, c.x, c.y);
                    }
                }
                __syncthreads();
                for (int e = lt; e < G::GH * 18; e += NT) {
                    const int r = e / 18, k = e - r * 18, pc = k / 3, c = k - pc * 3;
                    float* rowp = sG + mul24(r, G::GS);
                    if (colL >= 0) rowp[3 * pc + c] = rowp[3 * (12 - pc) + c];
                    if (peR >= 0 && peR + 1 + pc < G::GW) rowp[3 * (peR + 1 + pc) + c] = rowp[3 * (peR - 1 - pc) + c];
                }
            }
        } else if (INTERIOR && edge) {
#pragma unroll
            for (int n = 0; n < NPRE; ++n) {
                const int id = lt + n * NT;
                if ((n + 1) * NT > NCH && id >= NCH) continue;
                const int col = id - (id / CPR) * CPR;
                PreChunk<TIn> c = pre[n];
                if (col == colL || col == colR) c.rot2();
                if constexpr (MF) c.store_mf(sGr + id, G::MF_PLANE);
                else if constexpr (RAW) c.store_raw(sGr + RD * id);
                else *reinterpret_cast<v4f*>(sG + 4 * id) = c.get();
            }
            __syncthreads();
            // mirrored columns: patch column pc <- 12 - pc on the left, pc <- 2 peR - pc on the right (6 columns each)
            for (int e = lt; e < G::GH * 18; e += NT) {
                const int r = e / 18, k = e - r * 18, pc = k / 3, c = k - pc * 3;
                if constexpr (MF) {
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) {   // byte planes: an element is one byte of each
                        uint8_t* rowp = reinterpret_cast<uint8_t*>(sGr + pl * G::MF_PLANE) + mul24(r, 4 * CPR);
                        if (colL >= 0) rowp[3 * pc + c] = rowp[3 * (12 - pc) + c];
                        if (peR >= 0 && peR + 1 + pc < G::GW) rowp[3 * (peR + 1 + pc) + c] = rowp[3 * (peR - 1 - pc) + c];
                    }
                } else if constexpr (RAW) {
                    TIn* rowp = reinterpret_cast<TIn*>(sGr) + mul24(r, 4 * CPR);   // a raw row = CPR chunks of 4 elements
                    if (colL >= 0) rowp[3 * pc + c] = rowp[3 * (12 - pc) + c];
                    if (peR >= 0 && peR + 1 + pc < G::GW) rowp[3 * (peR + 1 + pc) + c] = rowp[3 * (peR - 1 - pc) + c];
                } else {
                    float* rowp = sG + mul24(r, G::GS);
                    if (colL >= 0) rowp[3 * pc + c] = rowp[3 * (12 - pc) + c];
                    if (peR >= 0 && peR + 1 + pc < G::GW) rowp[3 * (peR + 1 + pc) + c] = rowp[3 * (peR - 1 - pc) + c];
                }
            }
        } else if (MF && MI_SEP_MF_WIDE && !edge) {
#pragma unroll
            for (int m = 0; m < NPW; ++m) {
                const int id = lt + m * NT;
                if ((m + 1) * NT > NW16 && id >= NW16) continue;
                PreChunk<TIn>::store_mf16(pre + CP16 * m, sGr + CP16 * id, G::MF_PLANE);   // piece id = chunks CP16 id ..
            }
        } else {
#pragma unroll
            for (int n = 0; n < NPRE; ++n) {
                if ((n + 1) * NT > NCH && lt + n * NT >= NCH) continue;
                if constexpr (MF) pre[n].store_mf(sGr + (lt + n * NT), G::MF_PLANE);
                else if constexpr (RAW) pre[n].store_raw(sGr + RD * (lt + n * NT));
                else *reinterpret_cast<v4f*>(sG + 4 * (lt + n * NT)) = pre[n].get();
            }
        }
        MI_TICK(0);   // stage (waits for the prefetched loads)
        __syncthreads();
        MI_TICK(1);   // barrier 1
        // the next frame's loads: all here (0), or spread over the phases (MI_SEP_PF_SPLIT) so that the eight waves do
        // not queue up at the texture-address unit together (an issue stalls while its queue is full)
        constexpr int PF_A = MI_SEP_PF_SPLIT == 0 ? NPRE : MI_SEP_PF_SPLIT == 1 ? (NPRE + 1) / 2 : 1;
        if (!DMA && b + 1 < nfr && !MI_ABL(16)) prefetch(b + 1, 0, PF_A);
        if constexpr (TOUCH) {
            touch_acc ^= touch_val;   // last frame's touch has long returned: keeps the load alive for the compiler
            if (touch_on && b + MI_SEP_TOUCH < nfr)
                touch_val = __builtin_amdgcn_raw_buffer_load_b32(make_rsrc(src0 + (size_t)(b + MI_SEP_TOUCH) * a.src_stride, frame_bytes),
                                                                 touch_off, 0, 0);
        }
        MI_TICK(2);   // prefetch issue
        const BufRsrc gn_rs = make_rsrc(PAIR ? gnext0 : gnext0 + (size_t)b * a.gnext_stride, (uint32_t)hn * (uint32_t)wn * 12u);
        const BufRsrc gy_rs = make_rsrc(PAIR ? gray1_0 + (size_t)b * a.gray1_stride : nullptr, (uint32_t)hn * (uint32_t)wn * 4u);
        const BufRsrc g2_rs = make_rsrc(PAIR ? g2_0 + (size_t)b * a.g2_stride : nullptr, (uint32_t)a.hn2 * (uint32_t)a.wn2 * 12u);
        const bool keep3 = PAIR && f_lo + b == a.g1_keep;   // this frame's three-channel G_{l+1} goes to `gnext` (the tap)

        if constexpr (!MF) {
        // ---------------- P1: vertical reduce, P1R V rows x one float4 column group per lane
        constexpr int P1R = (RAW && MI_SEP_P1_ROWS > 2 && G::NH >= MI_SEP_P1_ROWS) ? MI_SEP_P1_ROWS : 2;
        if constexpr (P1R > 2) {
            constexpr int P1G = (G::NH + P1R - 1) / P1R;
            if (lt < P1G * CPR && !MI_ABL(1)) {
                const int rg = SmallDiv<CPR, NT>::div(lt), g = lt - mul24(rg, CPR);
                const int v0 = min(P1R * rg, G::NH - P1R);
                const uint32_t* p = sGr + RD * (mul24(2 * v0, CPR) + g);
                // rows in their raw form one output ahead (2 / 4 registers per row), the floats in a five-row window that
                // slides down two rows per output: the 64-register budget of these kernels holds
                typedef PreChunk<TIn> PC;
                auto ld = [&](int t) { return PC::ld_raw(p + t * (RD * CPR)); };
                const typename PC::raw_t q0 = ld(0), q1 = ld(1), q2 = ld(2);
                typename PC::raw_t n3 = ld(3), n4 = ld(4);
                float* d = sV + mul24(v0, G::VS) + 4 * g;
                v4f r0 = PC::cvt_raw(q0), r1 = PC::cvt_raw(q1), r2 = PC::cvt_raw(q2);
#pragma unroll
                for (int u = 0; u < P1R; ++u) {
                    typename PC::raw_t f3 = n3, f4 = n4;
                    if (u < P1R - 1) { f3 = ld(2 * u + 5); f4 = ld(2 * u + 6); }
                    const v4f r3 = PC::cvt_raw(n3), r4 = PC::cvt_raw(n4);
                    const v2f lo = s5r(r0.xy, r1.xy, r2.xy, r3.xy, r4.xy, w0, w1, w2);
                    const v2f hi = s5r(r0.zw, r1.zw, r2.zw, r3.zw, r4.zw, w0, w1, w2);
                    lds_store4(d + u * G::VS, lo.x, lo.y, hi.x, hi.y);
                    r0 = r2; r1 = r3; r2 = r4;
                    n3 = f3; n4 = f4;
                }
            }
        } else
        if (lt < (G::NH / 2) * CPR && !MI_ABL(1)) {
            const int rp = SmallDiv<CPR, NT>::div(lt), g = lt - mul24(rp, CPR);
            v2f a0, a1, b0, b1;   // (row 2rp | 2rp+1) x (floats 4g, 4g+1 | 4g+2, 4g+3)
            if constexpr (INTERIOR) {
                v4f r[7];
                if constexpr (RAW) {
                    const uint32_t* p = sGr + RD * (mul24(4 * rp, CPR) + g);
#pragma unroll
                    for (int t = 0; t < 7; ++t) r[t] = PreChunk<TIn>::unpack_raw(p + t * (RD * CPR));
                } else {
                    const float* p = sG + mul24(4 * rp, G::GS) + 4 * g;
#pragma unroll
                    for (int t = 0; t < 7; ++t) r[t] = lds_load4(p + t * G::GS);
                }
                a0 = s5r(r[0].xy, r[1].xy, r[2].xy, r[3].xy, r[4].xy, w0, w1, w2);
                a1 = s5r(r[0].zw, r[1].zw, r[2].zw, r[3].zw, r[4].zw, w0, w1, w2);
                b0 = s5r(r[2].xy, r[3].xy, r[4].xy, r[5].xy, r[6].xy, w0, w1, w2);
                b1 = s5r(r[2].zw, r[3].zw, r[4].zw, r[5].zw, r[6].zw, w0, w1, w2);
            } else {
                // a cell outside G_{l+1} is computed at its mirror position (REFLECT101 acts on the zero-stuffed
                // grid: V[-1] = G[1], V[n] = G[n-1]); the staged patch already holds reflected rows
                v2f o[2][2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int im = map_expand_src(y0 / 2 - 2 + 2 * rp + u, hn);
                    const int r0 = 2 * im - 2 - (y0 - 6);
                    v4f r[5];
#pragma unroll
                    for (int t = 0; t < 5; ++t) r[t] = lds_load4(sG + mul24(clampi(r0 + t, 0, G::GH - 1), G::GS) + 4 * g);
                    o[u][0] = s5r(r[0].xy, r[1].xy, r[2].xy, r[3].xy, r[4].xy, w0, w1, w2);
                    o[u][1] = s5r(r[0].zw, r[1].zw, r[2].zw, r[3].zw, r[4].zw, w0, w1, w2);
                }
                a0 = o[0][0]; a1 = o[0][1]; b0 = o[1][0]; b1 = o[1][1];
            }
            float* d = sV + mul24(2 * rp, G::VS) + 4 * g;
            lds_store4(d, a0.x, a0.y, a1.x, a1.y);
            lds_store4(d + G::VS, b0.x, b0.y, b1.x, b1.y);
        }
        if constexpr (GQ1) {
            // gray of the lane's quad of G_l (what P3 subtracts the expanded gray from): the last read of the staged patch
            if (lt < G::QY * G::QL) {
                v2f ge[3], go[3];
                if constexpr (RAW) {
                    const uint32_t* rowp = sGr + mul24(2 * (lt >> 5) + 4, RD * CPR);
                    PreChunk<TIn>::unpack6(rowp, 6 * (lt & 31) + 6, ge);
                    PreChunk<TIn>::unpack6(rowp + RD * CPR, 6 * (lt & 31) + 6, go);
                    gq_e = gray_of2(ge[0], ge[1], ge[2]);
                    gq_o = gray_of2(go[0], go[1], go[2]);
                } else {
                    const float* gp = sG + mul24(2 * (lt >> 5) + 4, G::GS) + 6 * (lt & 31) + 6;
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        ge[t] = lds_load2s(gp + 2 * t);
                        go[t] = lds_load2s(gp + G::GS + 2 * t);
                    }
                    gq_e = v2f{gray_of<true>(ge[0].x, ge[0].y, ge[1].x), gray_of<true>(ge[1].y, ge[2].x, ge[2].y)};
                    gq_o = v2f{gray_of<true>(go[0].x, go[0].y, go[1].x), gray_of<true>(go[1].y, go[2].x, go[2].y)};
                }
            }
        }
        if (!DMA && MI_SEP_PF_SPLIT == 1 && b + 1 < nfr && !MI_ABL(16)) prefetch(b + 1, PF_A, NPRE);
        if (!DMA && MI_SEP_PF_SPLIT == 2 && b + 1 < nfr && !MI_ABL(16)) prefetch(b + 1, 1, 2);
        MI_TICK(3);   // P1
        __syncthreads();
        MI_TICK(4);   // barrier 2
        }   // !MF
        if (DMA && b + 1 < nfr && !MI_ABL(16)) dma_issue(b + 1);   // the patch is dead: the next frame streams in beside P2-P4

        // ---------------- P2: horizontal reduce (one G_{l+1} pixel per lane, 32 lanes per patch row), G_{l+1} store, gray,
        // horizontal expand of the gray -> X.  18 rows x 32 = 576 items: the last two rows are a second round on wave 0.
        if constexpr (MF) { if (!MI_ABL(2)) {
            // ---------------- MF: the whole reduce of this wave's two G_{l+1} rows on the matrix pipe (see the top of the function)
            // (a task = two rows; tile height 28 has nine of them for eight waves: wave 0 takes the last one too)
            const int ln = lt & 63, n = ln & 15, kg = ln >> 4;
            for (int task = lt >> 6; task < G::NH / 2; task += NT / 64) {
            const int r = 2 * task + (n >> 3), jp = 4 * (n & 7) + kg;    // the pixel this lane ends up with
            typedef const volatile v4i __attribute__((address_space(3))) * lds_v4i;
            typedef const volatile v2u __attribute__((address_space(3))) * lds_v2u;
            const uint32_t wbase = (uint32_t)(uintptr_t)sW + 16u * (uint32_t)ln;
            const v4i A0 = *(lds_v4i)(size_t)wbase, A1 = *(lds_v4i)(size_t)(wbase + 1024u), A2 = *(lds_v4i)(size_t)(wbase + 2048u);
            // data: window n of patch row 2 r + t, bytes 16 kg .. 16 kg + 15 (two 8-byte reads: windows are 8-byte aligned)
            const uint32_t dbase = (uint32_t)(uintptr_t)sGr + (uint32_t)(mul24(2 * r, 4 * CPR) + 24 * (n & 7) + 16 * kg);
            v4i S4 = {0, 0, 0, 0};
#pragma unroll
            for (int pl = NPL - 1; pl >= 0; --pl) {
                v4i acc = {128 * 400, 128 * 400, 128 * 400, 128 * 400};
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    const uint32_t ad = dbase + (uint32_t)(4 * (pl * G::MF_PLANE + t * CPR));
                    const v2u d0 = *(lds_v2u)(size_t)ad, d1 = *(lds_v2u)(size_t)(ad + 8u);
                    const v4i B = {(int)d0.x, (int)d0.y, (int)d1.x, (int)d1.y};
                    if (MI_ABL(64)) acc += B;   // study: the data reads without the matrix instruction
                    else acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(t == 2 ? A2 : (t == 1 || t == 3) ? A1 : A0, B, acc, 0, 0, 0);
                }
                S4 = pl == NPL - 1 ? acc : (S4 << 8) + acc;   // 16-bit frames: 256 * (high-byte sum) + low-byte sum
            }
            float nn[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) nn[c] = (float)S4[c] * rs;
            {
                const int i = y0 / 2 - 2 + r, j = x0 / 2 - 2 + jp;
                bool st = r >= 2 && r < G::NH - 2 && jp >= 2 && jp < G::NW - 2 && !MI_ABL(32);
                st = st && i < hn && j < wn;   // edge tiles overhang the image
                if (st) {
                    typedef uint32_t v3u __attribute__((ext_vector_type(3)));
                    const v3u pv = {__builtin_bit_cast(uint32_t, nn[0]), __builtin_bit_cast(uint32_t, nn[1]),
                                    __builtin_bit_cast(uint32_t, nn[2])};
                    __builtin_amdgcn_raw_buffer_store_b96(pv, gn_rs, times12((uint32_t)(mul24(i, wn) + j)), 0, MI_SEP_NT_STORE ? 2 : 0);
                }
            }
            // gray of the pixel; its neighbours along the row sit 16 lanes away (next pixel of the window) or in the
            // neighbouring window's lane group: two bpermutes (LDS crossbar, no memory).  Pixels 0 / 31 of a row get a wrong
            // neighbour: X columns 0, 1, 62, 63 feed only the quads nobody owns or blurs from.
            const float g = gray_of<true>(nn[0], nn[1], nn[2]);
            const int lprev = kg > 0 ? ln - 16 : ln + 47, lnext = kg < 3 ? ln + 16 : ln - 47;
            float gl, gr;
            if (MI_ABL(128)) { gl = dpp_wave_prev(g); gr = dpp_wave_next(g); }   // study: (wrong) neighbours without the bpermutes
            else {
                gl = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * lprev, __builtin_bit_cast(int, g)));
                gr = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * lnext, __builtin_bit_cast(int, g)));
            }
            lds_store2(sX + mul24(r, G::XS) + 2 * jp, ex_even(gl, g, gr, ce, cc), ex_odd(g, gr, co));
            }   // task
        } } else {
#pragma unroll
        for (int rnd = 0; rnd < (G::NH * 32 + NT - 1) / NT; ++rnd) {
            const int it = lt + rnd * NT;
            if (it >= G::NH * 32 || MI_ABL(2)) break;   // uniform per wave: NT and the item count are multiples of 64
            const int r = it >> 5, jp = it & 31;
            float n[3];
            if constexpr (INTERIOR) {
                const float* p = sV + mul24(r, G::VS) + 6 * jp;   // patch pixels 2j' .. 2j'+4: 15 floats, 8-byte aligned
                float v[16];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const v2f q = lds_load2s(p + 2 * t);
                    v[2 * t] = q.x; v[2 * t + 1] = q.y;
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) n[c] = s5r(v[c], v[3 + c], v[6 + c], v[9 + c], v[12 + c], w0, w1, w2) * rs;
            } else {
                const int jm = map_expand_src(x0 / 2 - 2 + jp, wn);
                const int c0 = 2 * jm - 2 - (x0 - 6);
                const float* vr = sV + mul24(r, G::VS);
                float t5[5][3];
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    const float* q = vr + 3 * clampi(c0 + t, 0, G::GW - 1);
                    t5[t][0] = q[0]; t5[t][1] = q[1]; t5[t][2] = q[2];
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) n[c] = s5r(t5[0][c], t5[1][c], t5[2][c], t5[3][c], t5[4][c], w0, w1, w2) * rs;
            }
            if constexpr (PL) {   // the patch is all the payload phase wants
                float* np = sN + it;
                np[0] = n[0]; np[NPL1] = n[1]; np[2 * NPL1] = n[2];
                continue;
            }
            // gray of the pixel, its neighbours along the row by DPP, expanded columns 2j' (even) and 2j'+1 (odd)
            const float g = gray_of<true>(n[0], n[1], n[2]);
            // tile centre of G_{l+1} -> global (input of the next level; PAIR: its gray, and the pixel into the LDS patch)
            {
                const int i = y0 / 2 - 2 + r, j = x0 / 2 - 2 + jp;
                bool st = r >= 2 && r < G::NH - 2 && jp >= 2 && jp < G::NW - 2 && !MI_ABL(32);
                st = st && i < hn && j < wn;   // edge / border tiles overhang the image
                if constexpr (PAIR) {
                    float* np = sN + it;       // planar [3][NH][NW]: item `it` = pixel (r, jp)
                    np[0] = n[0]; np[NPL1] = n[1]; np[2 * NPL1] = n[2];
                    if (st)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, g), gy_rs,
                                                              (uint32_t)(mul24(i, wn) + j) << 2, 0, MI_SEP_NT_STORE ? 2 : 0);
                    st = st && keep3;
                }
                if ((!PAIR || keep3) && st) {   // (PAIR: keep3 is wave-uniform -- the whole block is skipped)
                    typedef uint32_t v3u __attribute__((ext_vector_type(3)));
                    const v3u pv = {__builtin_bit_cast(uint32_t, n[0]), __builtin_bit_cast(uint32_t, n[1]),
                                    __builtin_bit_cast(uint32_t, n[2])};
                    // one 12-byte buffer store, non-temporal (aux bit 1 = NT on gfx94x / gfx950): written once, read by
                    // the next level's launch much later
                    __builtin_amdgcn_raw_buffer_store_b96(pv, gn_rs, times12((uint32_t)(mul24(i, wn) + j)), 0, MI_SEP_NT_STORE ? 2 : 0);
                }
            }
            const float gl = dpp_wave_prev(g), gr = dpp_wave_next(g);
            lds_store2(sX + mul24(r, G::XS) + 2 * jp, ex_even(gl, g, gr, ce, cc), ex_odd(g, gr, co));
        }
        }   // !MF
        if (!DMA && MI_SEP_PF_SPLIT == 2 && b + 1 < nfr && !MI_ABL(16)) prefetch(b + 1, 2, 3);
        MI_TICK(5);   // P2
        __syncthreads();
        MI_TICK(6);   // barrier 3

        if constexpr (PL) {
            // ---------------- PL: the Laplacians of the pixels this frame won, from the G_{l+1} patch
            const int f = fid(b);
            const TIn* gfr = (const TIn*)(src0 + (size_t)f * a.src_stride);
            if (own_tile && (bI[0] == f || bI[1] == f || bI[2] == f || bI[3] == f)) {
                const int qy3 = lt >> 5, ql3 = lt & 31;
                float e[4][3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float* np = sN + c * NPL1 + mul24(qy3, G::NW) + ql3 - 1;   // G_{l+1} rows i-1 .. i+1, columns j-1 .. j+1
                    float xe[3], xo[3];
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const float n0 = np[r * G::NW], n1 = np[r * G::NW + 1], n2 = np[r * G::NW + 2];
                        xe[r] = ex_even(n0, n1, n2, ce, cc);
                        xo[r] = ex_odd(n1, n2, co);
                    }
                    e[0][c] = ex_even(xe[0], xe[1], xe[2], ce, cc);
                    e[1][c] = ex_even(xo[0], xo[1], xo[2], ce, cc);
                    e[2][c] = ex_odd(xe[1], xe[2], co);
                    e[3][c] = ex_odd(xo[1], xo[2], co);
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    if (bI[p] != f) continue;
                    const size_t px = (size_t)(oy + (p >> 1)) * w + ox + (p & 1);
                    float gv[3];
                    load_px3(gfr + px * 3, gv);
                    Px3 o;
#pragma unroll
                    for (int c = 0; c < 3; ++c) o.v[c] = (gv[c] - e[p][c]) + 0.0f;   // -0 -> +0 (pyramid.py:52-54)
                    *(Px3*)(a.best_lap + px * 3) = o;
                }
            }
            if (f1 == f) {   // (lanes below (TH / 2) * (TW / 2) only)
                const int r1 = SmallDiv<TW / 2, NT>::div(lt), c1 = lt - r1 * (TW / 2);
                const int i1 = y0 / 2 + r1, j1 = x0 / 2 + c1;
                const float* g2f = a.g2 + (size_t)f * a.g2_stride;
                const int hn2 = a.hn2, wn2 = a.wn2;
                Px3 o;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float g1v = sN[c * NPL1 + mul24(r1 + 2, G::NW) + c1 + 2];
                    const float ev = expand_sep_of([&](int r, int k) { return g2f[((size_t)r * wn2 + k) * 3 + c]; }, hn2, wn2, i1, j1, ce, cc, co);
                    o.v[c] = (g1v - ev) + 0.0f;
                }
                *(Px3*)(a.lap1 + ((size_t)i1 * wn + j1) * 3) = o;
            }
            if constexpr (INTERIOR) __syncthreads();   // the patch lies over G_l's: the next step's staging must not overtake these reads
            continue;
        }

        // ---------------- P3: vertical expand of the gray, gray Laplacian, Q, row blur of Q -> HB
        if (lt < G::QY * G::QL && !MI_ABL(4)) {   // (uniform per wave)
            const int qy3 = lt >> 5, ql3 = lt & 31;
            float q[4];
            if constexpr (INTERIOR) {
                const float* xr = sX + mul24(qy3, G::XS) + 2 * ql3;           // X rows qy, qy+1, qy+2
                const v2f xa = lds_load2s(xr), xb = lds_load2s(xr + G::XS), xc = lds_load2s(xr + 2 * G::XS);
                const v2f ev = ex_even(xa, xb, xc, ce, cc), od = ex_odd(xb, xc, co);
                v2f gge, ggo;   // gray of patch rows 2qy+4, +5; columns 2ql+2, +3
                if constexpr (GQ1) {
                    gge = gq_e;
                    ggo = gq_o;
                } else if constexpr (MF) {
                    v2f ge[3], go[3];
                    const uint32_t* rowp = sGr + mul24(2 * qy3 + 4, CPR);
                    PreChunk<TIn>::unpack6_mf(rowp, G::MF_PLANE, 6 * ql3 + 6, ge);
                    PreChunk<TIn>::unpack6_mf(rowp + CPR, G::MF_PLANE, 6 * ql3 + 6, go);
                    gge = gray_of2(ge[0], ge[1], ge[2]);
                    ggo = gray_of2(go[0], go[1], go[2]);
                } else if constexpr (RAW) {
                    // unpacked channel by channel ((pixel, pixel) pairs: the conversions write where they like), so the
                    // two grays of a row are one packed chain
                    v2f ge[3], go[3];
                    const uint32_t* rowp = sGr + mul24(2 * qy3 + 4, RD * CPR);
                    PreChunk<TIn>::unpack6(rowp, 6 * ql3 + 6, ge);
                    PreChunk<TIn>::unpack6(rowp + RD * CPR, 6 * ql3 + 6, go);
                    gge = gray_of2(ge[0], ge[1], ge[2]);
                    ggo = gray_of2(go[0], go[1], go[2]);
                } else {
                    // fp32 arrives as three 8-byte loads per row (pairs in memory order; regrouping them by channel
                    // costs more moves than the packed chain saves)
                    v2f ge[3], go[3];
                    const float* gp = sG + mul24(2 * qy3 + 4, G::GS) + 6 * ql3 + 6;
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        ge[t] = lds_load2s(gp + 2 * t);
                        go[t] = lds_load2s(gp + G::GS + 2 * t);
                    }
                    gge = v2f{gray_of<true>(ge[0].x, ge[0].y, ge[1].x), gray_of<true>(ge[1].y, ge[2].x, ge[2].y)};
                    ggo = v2f{gray_of<true>(go[0].x, go[0].y, go[1].x), gray_of<true>(go[1].y, go[2].x, go[2].y)};
                }
                const v2f le = gge - ev, lo = ggo - od;
                const v2f qe = le * le, qo = lo * lo;
                q[0] = qe.x; q[1] = qe.y; q[2] = qo.x; q[3] = qo.y;
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    // the pixel at its mirror position (REFLECT101 of Q); parity is preserved by the mirror
                    const int ym = map_clamp(y0 - 2 + 2 * qy3 + (p >> 1), h), xm = map_clamp(x0 - 4 + 2 * ql3 + (p & 1), w);
                    const int ri = clampi((ym >> 1) - (y0 / 2 - 2), 1, G::NH - 2);   // X row of G_{l+1} row ym / 2
                    const int ec = clampi(xm - (x0 - 4), 0, G::XW - 1);             // X column
                    const float* xq = sX + mul24(ri, G::XS) + ec;
                    const float* gq = sG + mul24(clampi(ym - (y0 - 6), 0, G::GH - 1), G::GS) + 3 * clampi(xm - (x0 - 6), 0, G::GW - 1);
                    const float e = (p >> 1) == 0 ? ex_even(xq[-G::XS], xq[0], xq[G::XS], ce, cc) : ex_odd(xq[0], xq[G::XS], co);
                    const float l = gray_of<true>(gq[0], gq[1], gq[2]) - e;
                    q[p] = l * l;
                }
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const float q0 = q[2 * rr], q1 = q[2 * rr + 1];
                const float l0 = dpp_wave_prev(q0), l1 = dpp_wave_prev(q1);
                const float r0 = dpp_wave_next(q0), r1 = dpp_wave_next(q1);
                const float hb0 = s5(l0, l1, q0, q1, r0, k0, k1, k2);
                const float hb1 = s5(l1, q0, q1, r0, r1, k0, k1, k2);
                lds_store2(sHB + mul24(2 * qy3 + rr, G::HBS) + 2 * ql3, hb0, hb1);
            }
        }
        if constexpr (PAIR) {
            // V2: column sums of the G_{l+2} reduce, one float4 of a patch row per lane (3 channels x N2H rows x NW / 4 groups),
            // on the last waves (wave 0 carries P2's second round); G_{l+1} rows through REFLECT101
            constexpr int V2N = 3 * N2H * (G::NW / 4), V2L0 = (NT - V2N) & ~63;
            static_assert(V2N <= NT, "V2 items");
            if (lt >= V2L0 && lt < V2L0 + V2N) {
                const int item = lt - V2L0, c = item / (N2H * (G::NW / 4)), rem = item - c * (N2H * (G::NW / 4));
                const int m = rem / (G::NW / 4), ch = rem - m * (G::NW / 4);
                v4f rr[5];
                if (!e2) {   // (uniform) no row of the patch is mapped: rows 2m .. 2m + 4
                    const float* p = sN + c * NPL1 + mul24(2 * m, G::NW) + 4 * ch;
#pragma unroll
                    for (int t = 0; t < 5; ++t) rr[t] = lds_load4(p + t * G::NW);
                } else {
                    const int mg = y0 / 4 + m, pr0 = y0 / 2 - 2;
#pragma unroll
                    for (int t = 0; t < 5; ++t) {
                        const int pr = clampi(r101(2 * mg - 2 + t, hn) - pr0, 0, G::NH - 1);   // (rows of G_{l+2} outside the image: anything)
                        rr[t] = lds_load4(sN + c * NPL1 + mul24(pr, G::NW) + 4 * ch);
                    }
                }
                const v2f lo = s5r(rr[0].xy, rr[1].xy, rr[2].xy, rr[3].xy, rr[4].xy, w0, w1, w2);
                const v2f hi = s5r(rr[0].zw, rr[1].zw, rr[2].zw, rr[3].zw, rr[4].zw, w0, w1, w2);
                lds_store4(sV2 + (c * N2H + m) * G::NW + 4 * ch, lo.x, lo.y, hi.x, hi.y);
            }
        }
        if (!DMA && MI_SEP_PF_SPLIT == 2 && b + 1 < nfr && !MI_ABL(16)) prefetch(b + 1, 3, NPRE);
        MI_TICK(7);   // P3
        __syncthreads();
        MI_TICK(8);   // barrier 4

        // ---------------- P4: column blur of HB for the own quad + running first-max
        if (own_tile && !MI_ABL(8)) {
            const int qy4 = lt >> 5, ql4 = lt & 31;
            const float* hp = sHB + mul24(2 * qy4 - 2, G::HBS) + 2 * ql4;
            v2f hb[6];
#pragma unroll
            for (int t = 0; t < 6; ++t) hb[t] = lds_load2s(hp + t * G::HBS);
            const v2f e0 = s5(hb[0], hb[1], hb[2], hb[3], hb[4], k0, k1, k2);
            const v2f e1 = s5(hb[1], hb[2], hb[3], hb[4], hb[5], k0, k1, k2);
            const float e[4] = {e0.x, e0.y, e1.x, e1.y};
            const int fidx = a.frame_idx0 + f_lo + b;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const bool win = e[p] > bE[p];
                bE[p] = win ? e[p] : bE[p];
                bI[p] = win ? fidx : bI[p];
            }
        }
        if constexpr (PAIR) {
            // H2: the tile's N2H x N2W pixels of G_{l+2}, one per lane on the first and the last wave (their quad rows are
            // halo: half idle in P4); G_{l+1} columns through REFLECT101
            constexpr int H2N = N2H * N2W, H2W = (H2N + 1) / 2;
            static_assert(H2W <= 64 || MI_SEP_TH != 28, "H2 items");   // (taller-tile study builds never launch the pair kernels)
            const int wv = lt >> 6, wl = lt & 63;
            if ((wv == 0 || wv == NT / 64 - 1) && wl < H2W && (wv == 0 ? wl : H2W + wl) < H2N) {
                const int item = wv == 0 ? wl : H2W + wl, m = item / N2W, k = item - m * N2W;
                const int mg = y0 / 4 + m, ng = x0 / 4 + k;
                float o[3];
                bool ok = true;
                if (!e2) {   // (uniform) no column is mapped: columns 2k .. 2k + 4 of the V2 row, three 8-byte reads
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float* vr = sV2 + (c * N2H + m) * G::NW + 2 * k;
                        const v2f q0 = lds_load2s(vr), q1 = lds_load2s(vr + 2), q2 = lds_load2s(vr + 4);
                        o[c] = s5r(q0.x, q0.y, q1.x, q1.y, q2.x, w0, w1, w2) * rs;
                    }
                } else {
                    ok = mg < a.hn2 && ng < a.wn2;
                    const int pc0 = x0 / 2 - 2;
                    int pc[5];
#pragma unroll
                    for (int t = 0; t < 5; ++t) pc[t] = clampi(r101(2 * ng - 2 + t, wn) - pc0, 0, G::NW - 1);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float* vr = sV2 + (c * N2H + m) * G::NW;
                        o[c] = s5r(vr[pc[0]], vr[pc[1]], vr[pc[2]], vr[pc[3]], vr[pc[4]], w0, w1, w2) * rs;
                    }
                }
                if (ok) {
                    typedef uint32_t v3u __attribute__((ext_vector_type(3)));
                    const v3u pv = {__builtin_bit_cast(uint32_t, o[0]), __builtin_bit_cast(uint32_t, o[1]),
                                    __builtin_bit_cast(uint32_t, o[2])};
                    __builtin_amdgcn_raw_buffer_store_b96(pv, g2_rs, times12((uint32_t)(mul24(mg, a.wn2) + ng)), 0, MI_SEP_NT_STORE ? 2 : 0);
                }
            }
        }
        MI_TICK(9);   // P4
        // no barrier here: sG is rewritten after P3's reads (barrier above), V/HB after the next frame's first
        // barrier, X after its second
    }
#ifdef MI_PHASE_CLOCK
    if (INTERIOR && a.dbg && (tid & 63) == 0 && blockIdx.y == 0) {
#pragma unroll
        for (int i = 0; i < 10; ++i) atomicAdd(a.dbg + (tid >> 6) * 16 + i, (unsigned long long)pc_acc[i]);
        atomicAdd(a.dbg + (tid >> 6) * 16 + 15, 1ull);
    }
#endif

    // ---- write the running maxima back
    if constexpr (PL) return;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int y = oy + (p >> 1), x = ox + (p & 1);
        if (own_tile && y < h && x < w) {
            const size_t px = (size_t)y * w + x;
            if (bI[p] >= 0) {   // only pixels a frame of this launch won are written back (a fresh state: all of them)
                st_e[px] = bE[p];
                st_i[px] = bI[p];
            }
        }
    }
    if constexpr (TOUCH) {   // never true: the touched dwords have a use
        if ((touch_acc ^ touch_val) == 0x9e3779b9u && a.nframes < 0) st_i[0] = (int32_t)touch_acc;
    }
}

// The kernels proper; the coarser levels get their own name so that profiles (rocprofv3 --stats aggregates by
// kernel name) keep the level-0 launches apart.
// border instantiation: minimum waves per SIMD the compiler has to leave room for (caps its VGPRs; 1 = no cap)
#ifndef MI_SEP_BD_WAVES
#define MI_SEP_BD_WAVES 1
#endif
template <typename TIn, bool INTERIOR, int TH, int NT>
__global__ __launch_bounds__(NT, INTERIOR ? (sizeof(TIn) <= 2 ? MI_SEP_INT_WAVES : (NT > 512 ? 7 : 1)) : MI_SEP_BD_WAVES) void level_sep(LevelArgs a) {
    level_sep_body<TIn, INTERIOR, TH, NT>(a);
}
// level 0 of 8 / 16-bit frames, reduce on the matrix pipe (MI_SEP_MFMA; tile height SEP_MF_TH)
template <typename TIn, int TH, int NT>
__global__ __launch_bounds__(NT, MI_SEP_INT_WAVES) void level_sep_mf(LevelArgs a) {
    level_sep_body<TIn, true, TH, NT, true>(a);
}
template <typename TIn, bool INTERIOR, int TH, int NT>
__global__ __launch_bounds__(NT, INTERIOR && NT > 512 ? 7 : 1) void level_sep_coarse(LevelArgs a) {
    level_sep_body<TIn, INTERIOR, TH, NT>(a);
}

// level pairs (level_sep_body, "PAIR"): the first level of a pair
template <typename TIn, bool INTERIOR, int TH, int NT>
__global__ __launch_bounds__(NT, INTERIOR ? (sizeof(TIn) <= 2 ? MI_SEP_INT_WAVES : 1) : MI_SEP_BD_WAVES) void level_sep_pair(LevelArgs a) {
    level_sep_body<TIn, INTERIOR, TH, NT, false, true>(a);
}

// the pair's payload pass, tile by tile (level_sep_body, "PL")
template <typename TIn, bool INTERIOR, int TH, int NT>
__global__ __launch_bounds__(NT, 1) void level_sep_pl(LevelArgs a) {
    level_sep_body<TIn, INTERIOR, TH, NT, false, false, true>(a);
}
template <typename TIn, int TH, int NT>
constexpr int sep_pl_lds_floats(bool interior) {
    using G = SepGeom<TH, NT>;
    return (interior ? G::lds_floats((int)sizeof(TIn), true) : G::LDS_FLOATS + 3 * G::NH * G::NW) + 16 + SEP_PL_MAXF;
}

// ================================================================================================
// level_sep_e -- the second level of a pair: ENERGY + running first-max only.  Its Gaussian image never exists in HBM: the
// first level's kernel left gray(G_l) (LevelArgs::src here: h x w floats per frame) and G_{l+1} (LevelArgs::gnext, READ here),
// and that is all the energy path needs (gray is linear: Q = (gray(G_l) - expand(gray(G_{l+1})))^2, kernels_sep.hpp header).
// Same tile, lane roles and arithmetic as level_sep's P2 (from the gray on) .. P4, so the running state is bit-identical to
// what level_sep computes from the three-channel images; per frame 4 bytes per pixel are read instead of 12, nothing is
// written, two barriers instead of four:
//   P0  stage the gray patch (tile + 2 rows / + 4 columns of halo: 32 x 64 floats, one 16-byte chunk per lane, prefetched one
//       frame ahead) and, one G_{l+1} pixel per lane (18 x 32, a 12-byte load, expand-source index map), the horizontally
//       expanded gray X
//   P3, P4  as level_sep
// Edge tiles: the INTERIOR instantiation stages the patch through REFLECT101 (element by element) and runs the interior code
// -- twins as in level_sep, same condition (even far edge; the host draws the same line); the general instantiation maps
// every pixel.
template <bool INTERIOR, int TH, int NT>
__device__ __forceinline__ void level_sep_e_body(const LevelArgs& a) {
    using G = SepGeom<TH, NT>;
    constexpr int TW = G::TW;
    constexpr int YH = TH + 4, YW = G::XW;   // gray patch: rows y0-2 .. y0+TH+1, columns x0-4 .. x0+TW+7 (X's and HB's columns)
    static_assert(YH * (YW / 4) <= NT && G::NH * 32 <= 2 * NT, "one chunk of the gray patch and at most two G_{l+1} pixels per lane");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sY = smem;
    float* sX = sY + YH * YW;
    float* sHB = sX + G::NH * G::XS;
    const int tid = threadIdx.x;
    const int h = a.h, w = a.w, hn = a.hn, wn = a.wn;
    int y0, x0;
    if (!sep_tile_origin<INTERIOR, TH, TW>(a, y0, x0)) return;
    const float k0 = a.k1d[0], k1 = a.k1d[1], k2 = a.k1d[2];
    const float ce = 2.0f * k0, cc = 2.0f * k2, co = 2.0f * k1;

    const int ck = blockIdx.y, f_lo = ck * a.chunk_frames;
    const int nfr = min(a.nframes - f_lo, a.chunk_frames);
    const bool fresh = ck > 0 || a.first;
    float* const st_e = ck ? a.part_e + (size_t)(ck - 1) * a.part_stride : a.best_e;
    int32_t* const st_i = ck ? a.part_idx + (size_t)(ck - 1) * a.part_stride : a.best_idx;
    const char* const src0 = (const char*)a.src + (size_t)f_lo * a.src_stride;
    const float* const gn0 = a.gnext + (size_t)f_lo * a.gnext_stride;

    const int qy = tid >> 5, ql = tid & 31;
    const bool own_tile = qy >= 1 && qy < G::QY - 1 && ql >= 2 && ql < G::QL - 2;
    const int oy = y0 - 2 + 2 * qy, ox = x0 - 4 + 2 * ql;
    float bE[4];
    int bI[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int y = oy + (p >> 1), x = ox + (p & 1);
        const bool valid = own_tile && y < h && x < w;
        bE[p] = (!fresh && valid) ? st_e[(size_t)y * w + x] : -__builtin_inff();   // (the first frame always wins, whatever the sign of its energy)
        bI[p] = -1;
    }

    // ---- staging offsets
    const bool edge = !INTERIOR || y0 < 2 || x0 < 4 || y0 + TH + 2 > h || x0 + TW + 8 > w;
    uint32_t yoff[4];
    {
        const int row = tid >> 4, c4 = 4 * (tid & 15);
        if (!edge) yoff[0] = yoff[1] = yoff[2] = yoff[3] = (uint32_t)((y0 - 2 + row) * w + (x0 - 4) + c4) * 4u;
        else {
            const int gy = map_clamp(y0 - 2 + row, h);
#pragma unroll
            for (int e = 0; e < 4; ++e) yoff[e] = (uint32_t)(gy * w + map_clamp(x0 - 4 + c4 + e, w)) * 4u;
        }
    }
    uint32_t noff[2];
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
        const int it = tid + rnd * NT, r = it >> 5, jp = it & 31;
        noff[rnd] = (uint32_t)(map_expand_src(y0 / 2 - 2 + r, hn) * wn + map_expand_src(x0 / 2 - 2 + jp, wn)) * 12u;
    }
    const uint32_t gray_bytes = (uint32_t)h * (uint32_t)w * 4u;
    v4f preY;
    // G_{l+1} pixels: 12-byte loads through a uniform base + a 32-bit lane offset (always inside the image: the index map
    // clamps).  NOT __builtin_amdgcn_raw_buffer_load_b96: this compiler (clang 19, ROCm 7.2) returns element 0 in all three
    // lanes of its result.
    Px3 preN[2];
    auto prefetch = [&](int b) {
        const BufRsrc ry = make_rsrc(src0 + (size_t)b * a.src_stride, gray_bytes);
        const char* gnb = (const char*)(gn0 + (size_t)b * a.gnext_stride);
        if (!edge) preY = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(ry, yoff[0], 0, 0));
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) preY[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, yoff[e], 0, 0));
        }
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd)
            if (tid + rnd * NT < G::NH * 32) preN[rnd] = *(const Px3*)(gnb + noff[rnd]);
    };
    prefetch(0);

    for (int b = 0; b < nfr; ++b) {
        int lt = tid;
#if MI_SEP_LAUNDER
        asm volatile("" : "+v"(lt));
#endif
        // ---------------- P0: the gray patch, and X from the G_{l+1} pixels (one per lane; rows 16, 17: a second round on wave 0)
        *reinterpret_cast<v4f*>(sY + 4 * lt) = preY;
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) {
            const int it = lt + rnd * NT;
            if (it >= G::NH * 32) break;   // uniform per wave
            const int r = it >> 5, jp = it & 31;
            const float g = gray_of<true>(preN[rnd].v[0], preN[rnd].v[1], preN[rnd].v[2]);
            const float gl = dpp_wave_prev(g), gr = dpp_wave_next(g);
            lds_store2(sX + mul24(r, G::XS) + 2 * jp, ex_even(gl, g, gr, ce, cc), ex_odd(g, gr, co));
        }
        __syncthreads();
        if (b + 1 < nfr) prefetch(b + 1);

        // ---------------- P3: vertical expand of the gray, gray Laplacian, Q, row blur of Q -> HB
        if (lt < G::QY * G::QL) {
            const int qy3 = lt >> 5, ql3 = lt & 31;
            float q[4];
            if constexpr (INTERIOR) {
                const float* xr = sX + mul24(qy3, G::XS) + 2 * ql3;
                const v2f xa = lds_load2s(xr), xb = lds_load2s(xr + G::XS), xc = lds_load2s(xr + 2 * G::XS);
                const v2f ev = ex_even(xa, xb, xc, ce, cc), od = ex_odd(xb, xc, co);
                const float* yp = sY + mul24(2 * qy3, YW) + 2 * ql3;
                const v2f gge = lds_load2s(yp), ggo = lds_load2s(yp + YW);
                const v2f le = gge - ev, lo = ggo - od;
                const v2f qe = le * le, qo = lo * lo;
                q[0] = qe.x; q[1] = qe.y; q[2] = qo.x; q[3] = qo.y;
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    // the pixel at its mirror position (REFLECT101 of Q); parity is preserved by the mirror
                    const int ym = map_clamp(y0 - 2 + 2 * qy3 + (p >> 1), h), xm = map_clamp(x0 - 4 + 2 * ql3 + (p & 1), w);
                    const int ri = clampi((ym >> 1) - (y0 / 2 - 2), 1, G::NH - 2);
                    const int ec = clampi(xm - (x0 - 4), 0, G::XW - 1);
                    const float* xq = sX + mul24(ri, G::XS) + ec;
                    const float gv = sY[mul24(clampi(ym - (y0 - 2), 0, YH - 1), YW) + clampi(xm - (x0 - 4), 0, YW - 1)];
                    const float e = (p >> 1) == 0 ? ex_even(xq[-G::XS], xq[0], xq[G::XS], ce, cc) : ex_odd(xq[0], xq[G::XS], co);
                    const float l = gv - e;
                    q[p] = l * l;
                }
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const float q0 = q[2 * rr], q1 = q[2 * rr + 1];
                const float l0 = dpp_wave_prev(q0), l1 = dpp_wave_prev(q1);
                const float r0 = dpp_wave_next(q0), r1 = dpp_wave_next(q1);
                const float hb0 = s5(l0, l1, q0, q1, r0, k0, k1, k2);
                const float hb1 = s5(l1, q0, q1, r0, r1, k0, k1, k2);
                lds_store2(sHB + mul24(2 * qy3 + rr, G::HBS) + 2 * ql3, hb0, hb1);
            }
        }
        __syncthreads();

        // ---------------- P4: column blur of HB for the own quad + running first-max
        if (own_tile) {
            const int qy4 = lt >> 5, ql4 = lt & 31;
            const float* hp = sHB + mul24(2 * qy4 - 2, G::HBS) + 2 * ql4;
            v2f hb[6];
#pragma unroll
            for (int t = 0; t < 6; ++t) hb[t] = lds_load2s(hp + t * G::HBS);
            const v2f e0 = s5(hb[0], hb[1], hb[2], hb[3], hb[4], k0, k1, k2);
            const v2f e1 = s5(hb[1], hb[2], hb[3], hb[4], hb[5], k0, k1, k2);
            const float e[4] = {e0.x, e0.y, e1.x, e1.y};
            const int fidx = a.frame_idx0 + f_lo + b;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const bool win = e[p] > bE[p];
                bE[p] = win ? e[p] : bE[p];
                bI[p] = win ? fidx : bI[p];
            }
        }
        // (sY and sX are rewritten after P3's reads -- the barrier above; HB after the next frame's first barrier)
    }

#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int y = oy + (p >> 1), x = ox + (p & 1);
        if (own_tile && y < h && x < w && bI[p] >= 0) {
            const size_t px = (size_t)y * w + x;
            st_e[px] = bE[p];
            st_i[px] = bI[p];
        }
    }
}
template <bool INTERIOR, int TH, int NT>
__global__ __launch_bounds__(NT, INTERIOR ? 8 : 1) void level_sep_e(LevelArgs a) {
    level_sep_e_body<INTERIOR, TH, NT>(a);
}
template <int TH, int NT>
constexpr int sep_e_lds_floats() { return (TH + 4) * SepGeom<TH, NT>::XW + SepGeom<TH, NT>::NH * SepGeom<TH, NT>::XS + SepGeom<TH, NT>::HBH * SepGeom<TH, NT>::HBS; }

// Fold the chunks' partial (max, arg-max) into the running state, in chunk order with a strict '>': the earliest frame
// holding the maximum stays the winner, as if the frames had been visited one after the other.
__global__ void merge_chunks(float* __restrict__ best_e, int32_t* __restrict__ best_idx, const float* __restrict__ part_e,
                             const int32_t* __restrict__ part_idx, size_t part_stride, int nparts, size_t npx) {
    const size_t px = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (px >= npx) return;
    float e = best_e[px];
    int32_t i = best_idx[px];
    for (int c = 0; c < nparts; ++c) {
        const float pe = part_e[(size_t)c * part_stride + px];
        const int32_t pi = part_idx[(size_t)c * part_stride + px];
        const bool win = pe > e;
        e = win ? pe : e;
        i = win ? pi : i;
    }
    best_e[px] = e;
    best_idx[px] = i;
}

// ================================================================================================
// Winner's Laplacian of the frames of one batch (the level kernel keeps only the running maximum and its frame):
// one lane per 2x2 quad of level l; a pixel whose arg-max is a frame of this batch gets
// lap = G_l - expand(G_{l+1}) of that frame, with -0 -> +0 as the reference's np.where sum gives (pyramid.py:52-54).
// A level that ran in frame chunks (LevelArgs::part_e) is folded here first, pixel by pixel, exactly as merge_chunks does
// (chunk order, strict '>'): `nparts` > 0 -- one launch and one dependent-launch gap less per level and batch.
template <typename TIn>
__global__ void sep_payload(const void* __restrict__ src, size_t src_stride, const float* __restrict__ gnext,
                            size_t gnext_stride, int nframes, int h, int w, int hn, int wn,
                            int32_t* __restrict__ best_idx, int frame_idx0, float* __restrict__ best_lap, float k0,
                            float k1, float k2, float* __restrict__ best_e, const float* __restrict__ part_e,
                            const int32_t* __restrict__ part_idx, size_t part_stride, int nparts) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y * blockDim.y + threadIdx.y;
    if (2 * i >= h || 2 * j >= w) return;
    const float ce = 2.0f * k0, cc = 2.0f * k2, co = 2.0f * k1;
    int fr[4];
    bool any = false;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int y = 2 * i + (p >> 1), x = 2 * j + (p & 1);
        fr[p] = -1;
        if (y < h && x < w) {
            const size_t px = (size_t)y * w + x;
            int32_t bi = best_idx[px];
            if (nparts > 0) {
                float e = best_e[px];
                bool changed = false;
                for (int c = 0; c < nparts; ++c) {
                    const float pe = part_e[(size_t)c * part_stride + px];
                    if (pe > e) { e = pe; bi = part_idx[(size_t)c * part_stride + px]; changed = true; }
                }
                if (changed) { best_e[px] = e; best_idx[px] = bi; }
            }
            const int f = bi - frame_idx0;
            if (f >= 0 && f < nframes) { fr[p] = f; any = true; }
        }
    }
    if (!any) return;
    const int ri[3] = {map_expand_src(i - 1, hn), i, map_expand_src(i + 1, hn)};
    const int cj[3] = {map_expand_src(j - 1, wn), j, map_expand_src(j + 1, wn)};
    // The four pixels of a quad mostly have the same winner: the 3x3 patch of G_{l+1} that expands to the quad is loaded
    // once per DISTINCT frame (12-byte pixel loads), expanded for the four positions, and stored for the pixels that frame won.
    unsigned pending = (fr[0] >= 0 ? 1u : 0u) | (fr[1] >= 0 ? 2u : 0u) | (fr[2] >= 0 ? 4u : 0u) | (fr[3] >= 0 ? 8u : 0u);
    while (pending) {
        const int f = (pending & 1u) ? fr[0] : (pending & 2u) ? fr[1] : (pending & 4u) ? fr[2] : fr[3];
        const float* gn = gnext + (size_t)f * gnext_stride;
        const TIn* g = (const TIn*)((const char*)src + (size_t)f * src_stride);
        Px3 N[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) N[r][q] = *(const Px3*)(gn + ((size_t)ri[r] * wn + cj[q]) * 3);
        float e[4][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float xe[3], xo[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                xe[r] = ex_even(N[r][0].v[c], N[r][1].v[c], N[r][2].v[c], ce, cc);
                xo[r] = ex_odd(N[r][1].v[c], N[r][2].v[c], co);
            }
            e[0][c] = ex_even(xe[0], xe[1], xe[2], ce, cc);
            e[1][c] = ex_even(xo[0], xo[1], xo[2], ce, cc);
            e[2][c] = ex_odd(xe[1], xe[2], co);
            e[3][c] = ex_odd(xo[1], xo[2], co);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (!((pending >> p) & 1u) || fr[p] != f) continue;
            pending &= ~(1u << p);
            const size_t px = (size_t)(2 * i + (p >> 1)) * w + 2 * j + (p & 1);
            Px3 o;
            if constexpr (sizeof(TIn) == 4) {
                const Px3 gv = *(const Px3*)((const float*)g + px * 3);
#pragma unroll
                for (int c = 0; c < 3; ++c) o.v[c] = (gv.v[c] - e[p][c]) + 0.0f;
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) o.v[c] = (to_f32(g[px * 3 + c]) - e[p][c]) + 0.0f;
            }
            *(Px3*)(best_lap + px * 3) = o;
        }
    }
}

// ================================================================================================
// Payload passes of a level pair (level_sep_body "PAIR" + level_sep_e): the three-channel G_{l+1} of the batch was never
// stored, so the winners' pixels of it are recomputed from G_l -- the same reduce (s5r down the rows, then along the rows, * rs,
// REFLECT101), hence the same bits.  Once per level and batch; the loads mostly hit L1 / L2 (neighbouring quads share windows
// when they share the winner).

// G_{l+1}[i, j] (inside its image) of the frame at `g` (level l, h x w): one pixel, 25 loads
template <typename TIn>
__device__ __forceinline__ void reduce_px(const TIn* __restrict__ g, int h, int w, int i, int j, float w0, float w1, float w2,
                                          float rs, float o[3]) {
    float V[5][3];
    int rows[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) rows[u] = r101(2 * i - 2 + u, h);
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int x = r101(2 * j - 2 + t, w);
        float p[5][3];
#pragma unroll
        for (int u = 0; u < 5; ++u) load_px3(g + ((size_t)rows[u] * w + x) * 3, p[u]);
#pragma unroll
        for (int c = 0; c < 3; ++c) V[t][c] = s5r(p[0][c], p[1][c], p[2][c], p[3][c], p[4][c], w0, w1, w2);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = s5r(V[0][c], V[1][c], V[2][c], V[3][c], V[4][c], w0, w1, w2) * rs;
}

// First level of a pair: lap_l = G_l - expand(G_{l+1}), the 3 x 3 patch of G_{l+1} that expands to the quad recomputed from
// the winner's G_l (away from the image edges: its 9 x 9 window column by column, 81 pixel loads; else pixel by pixel).
// Chunk partials are folded first, as sep_payload does.
template <typename TIn>
__global__ __launch_bounds__(256, 3) void sep_payload_pair0(const void* __restrict__ src, size_t src_stride, int nframes, int h, int w,
                                                         int hn, int wn, int32_t* __restrict__ best_idx, int frame_idx0,
                                                         float* __restrict__ best_lap, float k0, float k1, float k2, float w0,
                                                         float w1, float w2, float rs, float* __restrict__ best_e,
                                                         const float* __restrict__ part_e, const int32_t* __restrict__ part_idx,
                                                         size_t part_stride, int nparts, const uint8_t* __restrict__ tile_flag,
                                                         int tile_h, int tile_w) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y * blockDim.y + threadIdx.y;
    if (2 * i >= h || 2 * j >= w) return;
    // behind the tile-by-tile pass (level_sep_pl): only the tiles it flagged
    if (tile_flag && !tile_flag[((2 * i) / tile_h) * ((w + tile_w - 1) / tile_w) + (2 * j) / tile_w]) return;
    const float ce = 2.0f * k0, cc = 2.0f * k2, co = 2.0f * k1;
    int fr[4];
    bool any = false;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int y = 2 * i + (p >> 1), x = 2 * j + (p & 1);
        fr[p] = -1;
        if (y < h && x < w) {
            const size_t px = (size_t)y * w + x;
            int32_t bi = best_idx[px];
            if (nparts > 0) {
                float e = best_e[px];
                bool changed = false;
                for (int c = 0; c < nparts; ++c) {
                    const float pe = part_e[(size_t)c * part_stride + px];
                    if (pe > e) { e = pe; bi = part_idx[(size_t)c * part_stride + px]; changed = true; }
                }
                if (changed) { best_e[px] = e; best_idx[px] = bi; }
            }
            const int f = bi - frame_idx0;
            if (f >= 0 && f < nframes) { fr[p] = f; any = true; }
        }
    }
    if (!any) return;
    const int ri[3] = {map_expand_src(i - 1, hn), i, map_expand_src(i + 1, hn)};
    const int cj[3] = {map_expand_src(j - 1, wn), j, map_expand_src(j + 1, wn)};
    const bool inner = i >= 2 && j >= 2 && 2 * i + 4 < h && 2 * j + 4 < w;   // (then i + 1 < hn, j + 1 < wn: nothing is mapped)
    unsigned pending = (fr[0] >= 0 ? 1u : 0u) | (fr[1] >= 0 ? 2u : 0u) | (fr[2] >= 0 ? 4u : 0u) | (fr[3] >= 0 ? 8u : 0u);
    while (pending) {
        const int f = (pending & 1u) ? fr[0] : (pending & 2u) ? fr[1] : (pending & 4u) ? fr[2] : fr[3];
        const TIn* g = (const TIn*)((const char*)src + (size_t)f * src_stride);
        float N[3][3][3];
        if (inner) {
            float V[3][9][3];
            const TIn* c0 = g + ((size_t)(2 * i - 4) * w + (2 * j - 4)) * 3;
#pragma unroll
            for (int x = 0; x < 9; ++x) {
                float p[9][3];
#pragma unroll
                for (int u = 0; u < 9; ++u) load_px3(c0 + ((size_t)u * w + x) * 3, p[u]);
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        V[r][x][c] = s5r(p[2 * r][c], p[2 * r + 1][c], p[2 * r + 2][c], p[2 * r + 3][c], p[2 * r + 4][c], w0, w1, w2);
            }
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        N[r][q][c] = s5r(V[r][2 * q][c], V[r][2 * q + 1][c], V[r][2 * q + 2][c], V[r][2 * q + 3][c], V[r][2 * q + 4][c], w0, w1, w2) * rs;
        } else {
#pragma unroll 1
            for (int rq = 0; rq < 9; ++rq) {   // (image edges only: kept rolled)
                const int r = rq / 3, q = rq - 3 * r;
                float o[3];
                reduce_px(g, h, w, r == 0 ? ri[0] : r == 1 ? ri[1] : ri[2], q == 0 ? cj[0] : q == 1 ? cj[1] : cj[2], w0, w1, w2, rs, o);
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                    for (int qq = 0; qq < 3; ++qq)
                        if (rr == r && qq == q) { N[rr][qq][0] = o[0]; N[rr][qq][1] = o[1]; N[rr][qq][2] = o[2]; }
            }
        }
        float e[4][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float xe[3], xo[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                xe[r] = ex_even(N[r][0][c], N[r][1][c], N[r][2][c], ce, cc);
                xo[r] = ex_odd(N[r][1][c], N[r][2][c], co);
            }
            e[0][c] = ex_even(xe[0], xe[1], xe[2], ce, cc);
            e[1][c] = ex_even(xo[0], xo[1], xo[2], ce, cc);
            e[2][c] = ex_odd(xe[1], xe[2], co);
            e[3][c] = ex_odd(xo[1], xo[2], co);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (!((pending >> p) & 1u) || fr[p] != f) continue;
            pending &= ~(1u << p);
            const size_t px = (size_t)(2 * i + (p >> 1)) * w + 2 * j + (p & 1);
            float gv[3];
            load_px3(g + px * 3, gv);
            Px3 o;
#pragma unroll
            for (int c = 0; c < 3; ++c) o.v[c] = (gv[c] - e[p][c]) + 0.0f;
            *(Px3*)(best_lap + px * 3) = o;
        }
    }
}

// Second level of a pair (level l + 1 of h1 x w1 pixels; (h, w) = level l, the stored one): lap = G_{l+1} - expand(G_{l+2}) with
// the winner's pixel of G_{l+1} recomputed from its G_l (`src`), G_{l+2} (`g2`, h2 x w2) as stored by the pair's first kernel.
template <typename TIn>
__global__ __launch_bounds__(256, 4) void sep_payload_pair1(const void* __restrict__ src, size_t src_stride, const float* __restrict__ g2,
                                                         size_t g2_stride, int nframes, int h, int w, int h1, int w1, int h2, int w2,
                                                         int32_t* __restrict__ best_idx, int frame_idx0,
                                                         float* __restrict__ best_lap, float k0, float k1, float k2, float w0,
                                                         float w1r, float w2r, float rs, float* __restrict__ best_e,
                                                         const float* __restrict__ part_e, const int32_t* __restrict__ part_idx,
                                                         size_t part_stride, int nparts, const uint8_t* __restrict__ tile_flag,
                                                         int tile_h, int tile_w) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y * blockDim.y + threadIdx.y;
    if (2 * i >= h1 || 2 * j >= w1) return;
    if (tile_flag && !tile_flag[((4 * i) / tile_h) * ((w + tile_w - 1) / tile_w) + (4 * j) / tile_w]) return;
    const float ce = 2.0f * k0, cc = 2.0f * k2, co = 2.0f * k1;
    int fr[4];
    bool any = false;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int y = 2 * i + (p >> 1), x = 2 * j + (p & 1);
        fr[p] = -1;
        if (y < h1 && x < w1) {
            const size_t px = (size_t)y * w1 + x;
            int32_t bi = best_idx[px];
            if (nparts > 0) {
                float e = best_e[px];
                bool changed = false;
                for (int c = 0; c < nparts; ++c) {
                    const float pe = part_e[(size_t)c * part_stride + px];
                    if (pe > e) { e = pe; bi = part_idx[(size_t)c * part_stride + px]; changed = true; }
                }
                if (changed) { best_e[px] = e; best_idx[px] = bi; }
            }
            const int f = bi - frame_idx0;
            if (f >= 0 && f < nframes) { fr[p] = f; any = true; }
        }
    }
    if (!any) return;
    const int ri[3] = {map_expand_src(i - 1, h2), i, map_expand_src(i + 1, h2)};
    const int cj[3] = {map_expand_src(j - 1, w2), j, map_expand_src(j + 1, w2)};
    const bool inner = i >= 1 && j >= 1 && 4 * i + 4 < h && 4 * j + 4 < w && 2 * i + 1 < h1 && 2 * j + 1 < w1;
    unsigned pending = (fr[0] >= 0 ? 1u : 0u) | (fr[1] >= 0 ? 2u : 0u) | (fr[2] >= 0 ? 4u : 0u) | (fr[3] >= 0 ? 8u : 0u);
    while (pending) {
        const int f = (pending & 1u) ? fr[0] : (pending & 2u) ? fr[1] : (pending & 4u) ? fr[2] : fr[3];
        const float* gn = g2 + (size_t)f * g2_stride;
        const TIn* g = (const TIn*)((const char*)src + (size_t)f * src_stride);
        Px3 N[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) N[r][q] = *(const Px3*)(gn + ((size_t)ri[r] * w2 + cj[q]) * 3);
        float e[4][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float xe[3], xo[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                xe[r] = ex_even(N[r][0].v[c], N[r][1].v[c], N[r][2].v[c], ce, cc);
                xo[r] = ex_odd(N[r][1].v[c], N[r][2].v[c], co);
            }
            e[0][c] = ex_even(xe[0], xe[1], xe[2], ce, cc);
            e[1][c] = ex_even(xo[0], xo[1], xo[2], ce, cc);
            e[2][c] = ex_odd(xe[1], xe[2], co);
            e[3][c] = ex_odd(xo[1], xo[2], co);
        }
        if (inner && pending == 15u && fr[1] == f && fr[2] == f && fr[3] == f) {
            // the whole quad has one winner: its 7 x 7 window of G_l column by column (49 pixel loads instead of 100)
            pending = 0u;
            float V[2][7][3];
            const TIn* c0 = g + ((size_t)(4 * i - 2) * w + (4 * j - 2)) * 3;
#pragma unroll
            for (int x = 0; x < 7; ++x) {
                float p[7][3];
#pragma unroll
                for (int u = 0; u < 7; ++u) load_px3(c0 + ((size_t)u * w + x) * 3, p[u]);
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        V[r][x][c] = s5r(p[2 * r][c], p[2 * r + 1][c], p[2 * r + 2][c], p[2 * r + 3][c], p[2 * r + 4][c], w0, w1r, w2r);
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int r = p >> 1, q = p & 1;
                Px3 o;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float gv = s5r(V[r][2 * q][c], V[r][2 * q + 1][c], V[r][2 * q + 2][c], V[r][2 * q + 3][c], V[r][2 * q + 4][c], w0, w1r, w2r) * rs;
                    o.v[c] = (gv - e[p][c]) + 0.0f;
                }
                *(Px3*)(best_lap + ((size_t)(2 * i + r) * w1 + 2 * j + q) * 3) = o;
            }
            break;
        }
#pragma unroll 1
        for (int p = 0; p < 4; ++p) {   // (mixed winners, image edges: kept rolled)
            const int fp = p == 0 ? fr[0] : p == 1 ? fr[1] : p == 2 ? fr[2] : fr[3];
            if (!((pending >> p) & 1u) || fp != f) continue;
            pending &= ~(1u << p);
            const int y = 2 * i + (p >> 1), x = 2 * j + (p & 1);
            float gv[3];
            reduce_px(g, h, w, y, x, w0, w1r, w2r, rs, gv);
            Px3 o;
#pragma unroll
            for (int c = 0; c < 3; ++c) o.v[c] = (gv[c] - (p == 0 ? e[0][c] : p == 1 ? e[1][c] : p == 2 ? e[2][c] : e[3][c])) + 0.0f;
            *(Px3*)(best_lap + ((size_t)y * w1 + x) * 3) = o;
        }
    }
}

// ================================================================================================
// One-thread-per-output kernels of the same arithmetic: the on-GPU cross-check (MI_IMPL_SIMPLE) and the
// once-per-stack collapse.

// V(i, x) of a source row pair: s5 down the rows 2i-2 .. 2i+2 (REFLECT101), one channel
template <typename TIn>
__device__ __forceinline__ float sep_v(const TIn* __restrict__ g, int h, int w, int i, int x, int c, float w0, float w1,
                                       float w2) {
    float v[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) v[t] = to_f32(g[((size_t)r101(2 * i - 2 + t, h) * w + x) * 3 + c]);
    return s5r(v[0], v[1], v[2], v[3], v[4], w0, w1, w2);
}

template <typename TIn>
__global__ void reduce_sep_simple(const TIn* __restrict__ g, int h, int w, float* __restrict__ out, int ho, int wo,
                                  float w0, float w1, float w2, float rs) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= ho || j >= wo) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) v[t] = sep_v(g, h, w, i, r101(2 * j - 2 + t, w), c, w0, w1, w2);
        out[((size_t)i * wo + j) * 3 + c] = s5r(v[0], v[1], v[2], v[3], v[4], w0, w1, w2) * rs;
    }
}

// Laplacian (the payload) and Q = (gray(G_l) - expand(gray(G_{l+1})))^2 (the energy path)
template <typename TIn>
__global__ void lapq_sep_simple(const TIn* __restrict__ g, int h, int w, const float* __restrict__ gn, int hs, int ws,
                                float* __restrict__ lap, float* __restrict__ q, float k0, float k1, float k2) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (y >= h || x >= w) return;
    const float ce = 2.0f * k0, cc = 2.0f * k2, co = 2.0f * k1;
    const size_t p = (size_t)y * w + x;
    float gl[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        gl[c] = to_f32(g[p * 3 + c]);
        lap[p * 3 + c] = gl[c] - expand_sep_of([&](int r, int k) { return gn[((size_t)r * ws + k) * 3 + c]; }, hs, ws, y, x, ce, cc, co);
    }
    const float eg = expand_sep_of([&](int r, int k) {
        const float* s = gn + ((size_t)r * ws + k) * 3;
        return gray_of<true>(s[0], s[1], s[2]);
    }, hs, ws, y, x, ce, cc, co);
    const float l = gray_of<true>(gl[0], gl[1], gl[2]) - eg;
    q[p] = l * l;
}

__global__ void select_sep_simple(const float* __restrict__ q, const float* __restrict__ lap, int h, int w, int frame_idx,
                                  int first, float* __restrict__ best_e, float* __restrict__ best_lap,
                                  int32_t* __restrict__ best_idx, float k0, float k1, float k2) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (y >= h || x >= w) return;
    float hb[5];
#pragma unroll
    for (int ty = 0; ty < 5; ++ty) {
        const float* row = q + (size_t)r101(y + ty - 2, h) * w;
        hb[ty] = s5(row[r101(x - 2, w)], row[r101(x - 1, w)], row[x], row[r101(x + 1, w)], row[r101(x + 2, w)], k0, k1, k2);
    }
    const float s = s5(hb[0], hb[1], hb[2], hb[3], hb[4], k0, k1, k2);
    const size_t p = (size_t)y * w + x;
    if (first || s > best_e[p]) {
        best_e[p] = s;
        best_idx[p] = frame_idx;
#pragma unroll
        for (int c = 0; c < 3; ++c) best_lap[p * 3 + c] = lap[p * 3 + c] + 0.0f;
    }
}

// collapse step (pyramid.py:57-64): out = expand(up) + lap; TOut != float: the finest step, fused with
// clip(abs()) and the truncating cast (pyramid.py:64, :179).  One lane per 2x2 quad of the output (launch over
// ceil(w/2) x ceil(h/2)): the 3x3 patch of `up` that expands to the quad is loaded once, as 12-byte pixels.
template <typename TOut>
__global__ void collapse_sep(const float* __restrict__ up, int hs, int ws, const float* __restrict__ lap, int h, int w,
                             float maxv, TOut* __restrict__ out, float k0, float k1, float k2) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y * blockDim.y + threadIdx.y;
    if (2 * i >= h || 2 * j >= w) return;
    const float ce = 2.0f * k0, cc = 2.0f * k2, co = 2.0f * k1;
    const int ri[3] = {map_expand_src(i - 1, hs), map_expand_src(i, hs), map_expand_src(i + 1, hs)};
    const int cj[3] = {map_expand_src(j - 1, ws), map_expand_src(j, ws), map_expand_src(j + 1, ws)};
    Px3 N[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) N[r][q] = *(const Px3*)(up + ((size_t)ri[r] * ws + cj[q]) * 3);
    float e[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float xe[3], xo[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            xe[r] = ex_even(N[r][0].v[c], N[r][1].v[c], N[r][2].v[c], ce, cc);
            xo[r] = ex_odd(N[r][1].v[c], N[r][2].v[c], co);
        }
        e[0][c] = ex_even(xe[0], xe[1], xe[2], ce, cc);
        e[1][c] = ex_even(xo[0], xo[1], xo[2], ce, cc);
        e[2][c] = ex_odd(xe[1], xe[2], co);
        e[3][c] = ex_odd(xo[1], xo[2], co);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int y = 2 * i + (p >> 1), x = 2 * j + (p & 1);
        if (y >= h || x >= w) continue;
        const size_t px = ((size_t)y * w + x) * 3;
        const Px3 lv = *(const Px3*)(lap + px);
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

}  // namespace mi
