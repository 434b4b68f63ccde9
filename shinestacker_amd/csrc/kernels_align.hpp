// kernels_align.hpp -- the alignment APPLY step of align_images (reference
// src/shinestacker/algorithms/align.py:238-251, ALIGN_RIGID): cv2.warpAffine with
// replicate / constant border, the warped all-ones mask, and the blurred-border composite.
// Arithmetic follows oracle/align_oracle.c operation by operation (see its header for the
// OpenCV semantics restated from memory and the parity status).
#pragma once
#include "common.hpp"
#include "kernels_tiled.hpp"   // v4f, mul24

namespace mi {

struct AffineArgs {
    double iM[6];      // inverted transform (dst -> src), double
    int h, w;
    int mode;          // 0 constant, 1 replicate
    int border[3];     // constant border value per channel, already rounded / saturated
};

// lane value x uniform value as ONE 24-bit multiply (left to itself the compiler re-associates the row offset into
// v_mul_lo_u32, a quarter-rate instruction).  Only for operands that ordinary VALU instructions produced: the hazard
// recogniser does not look inside inline asm, and a v_dot* result needs three wait states before a VALU may read it.
__device__ __forceinline__ int mul24_vs(int v, int s) {
    int d;
    asm("v_mul_i32_i24 %0, %1, %2" : "=v"(d) : "s"(s), "v"(v));
    return d;
}

__device__ __forceinline__ int cv_round_d(double v) {
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (-2147483647 - 1);
    return (int)rint(v);  // round half to even, as cvRound / lrint
}

// OpenCV's warpAffine rounds its coordinate terms in double ONCE per column (adelta / bdelta) and once per row; so does
// this kernel, into a table the warp kernel reads: [ad: w][bd: w][X0: h][Y0: h] -- no double arithmetic per tile and row
// in the warp itself (it was a third of the tiled kernel's instruction time).
// `clr` (optional): `nclr` dwords of the border blur's tile scratch (counter + bitmap) to zero for the warp kernel behind this
// one -- a separate memset was a launch of its own per frame.
__global__ __launch_bounds__(256) void warp_coord_tables(AffineArgs a, int* __restrict__ tab, uint32_t* __restrict__ clr, int nclr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = i; k < nclr; k += gridDim.x * blockDim.x) clr[k] = 0u;
    if (i < a.w) {
        tab[i] = cv_round_d(a.iM[0] * i * 1024.0);
        tab[a.w + i] = cv_round_d(a.iM[3] * i * 1024.0);
    }
    if (i < a.h) {
        tab[2 * a.w + i] = cv_round_d((a.iM[1] * i + a.iM[2]) * 1024.0) + 16;
        tab[2 * a.w + a.h + i] = cv_round_d((a.iM[4] * i + a.iM[5]) * 1024.0) + 16;
    }
}

// one destination pixel: the 3 channel values and the mask bit
// (X, Y): the source position in 1/32 pixel
template <typename T>
__device__ __forceinline__ void warp_pixel_xy(const T* __restrict__ src, const AffineArgs& a, int X, int Y,
                                              int out[3], int& ok) {
    const int h = a.h, w = a.w;
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
__device__ __forceinline__ void warp_pixel_inside(const T* __restrict__ src, const AffineArgs& a, int adx, int bdx, int X0, int Y0,
                                                  int out[3]) {
    const int w = a.w;
    const int X = (X0 + adx) >> 5;
    const int Y = (Y0 + bdx) >> 5;
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

// ---- cv2.warpPerspective (ALIGN_HOMOGRAPHY, align.py:231-237): the same interpolation, the source position from the
// projective map.  OpenCV's WarpPerspectiveInvoker [from memory] walks blocks of bw0 = min(1024 / min(16, h), w) columns:
//   X0 = M0*bx + M1*y + M2, Y0 = M3*bx + M4*y + M5, W0 = M6*bx + M7*y + M8   (bx = the block's first column)
//   W = W0 + M6*x1;  W = W ? 32 / W : 0;  X = cvRound(clamp((X0 + M0*x1) * W)),  Y likewise   (x1 = x - bx)
// all in double, no fused multiply-add (oracle/align_oracle.c restates the same).  One pixel per thread, gathers from
// global memory: homographies are rare in focus stacking; this path is about having the option, not about speed.
struct PerspArgs {
    double iM[9];   // inverted 3x3 (dst -> src)
    int bw0;
};

template <typename T>
__global__ __launch_bounds__(256) void warp_perspective_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                                               uint8_t* __restrict__ valid, AffineArgs a, PerspArgs pa) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= a.w || y >= a.h) return;
    const int bx = (x / pa.bw0) * pa.bw0, x1 = x - bx;
    const double* M = pa.iM;
    const double X0 = M[0] * bx + M[1] * y + M[2], Y0 = M[3] * bx + M[4] * y + M[5], W0 = M[6] * bx + M[7] * y + M[8];
    double W = W0 + M[6] * x1;
    W = W != 0.0 ? 32.0 / W : 0.0;
    const double fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + M[0] * x1) * W));
    const double fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + M[3] * x1) * W));
    int v[3], ok;
    warp_pixel_xy<T>(src, a, cv_round_d(fX), cv_round_d(fY), v, ok);
    const size_t px = (size_t)y * a.w + x;
    dst[px * 3 + 0] = (T)v[0]; dst[px * 3 + 1] = (T)v[1]; dst[px * 3 + 2] = (T)v[2];
    if (valid) valid[px] = (uint8_t)ok;
}

// ---- The warp kernel.  A per-pixel gather spends its time in the texture-address unit (four unaligned 4-byte
// gathers per pixel: 99 us per 24 MP frame) and in double-precision coordinate arithmetic (OpenCV rounds
// iM[0]*x*1024 and iM[3]*x*1024 in double per COLUMN -- its adelta / bdelta tables -- and the row terms per ROW).
// Here a workgroup owns a TH x 256 tile of the destination: the tile's source bounding box (exact: the rounded
// terms are monotone, so the extremes sit at the tile's corners) is staged into LDS with coalesced dword loads,
// every thread keeps the column terms of its 4 columns for all its rows, and the taps come out of LDS (three dword
// reads and two v_alignbyte per source row and pixel pair).  Same integer / float arithmetic, same results.
// Tiles whose bounding box leaves the image, or does not fit the LDS budget (large rotations), take the per-pixel
// path.
constexpr int WT_W = 256;
#ifndef MI_WARP_TH_U8
#define MI_WARP_TH_U8 32
#endif
#ifndef MI_WARP_LDS_DWORDS
#define MI_WARP_LDS_DWORDS 10240
#endif
template <typename T> struct WarpTile { static constexpr int TH = sizeof(T) == 1 ? MI_WARP_TH_U8 : 16; };
constexpr int WT_SPLIT = 4;   // workgroups per outer-ring tile (divides the rows per thread: 8 / 4)
constexpr int WT_LDS_DWORDS = MI_WARP_LDS_DWORDS;   // 40 KB: four workgroups per CU

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void warp_affine_tiled(const T* __restrict__ src, T* __restrict__ dst,
                                                         uint8_t* __restrict__ valid, AffineArgs a,
                                                         uint32_t* __restrict__ tile_bitmap, uint32_t* __restrict__ tile_list,
                                                         int blur_tiles_x,
                                                         const int* __restrict__ tab, int gx, int gy) {
    extern __shared__ uint32_t s_src[];
    constexpr int BPP = 3 * (int)sizeof(T), TH = WarpTile<T>::TH, RPT = TH / 4;   // rows per thread
    const int h = a.h, w = a.w;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);   // the wave index as a scalar: row terms come in through scalar loads
    // Which tile, which of its rows?  The grid is one-dimensional: the OUTER RING of tiles first, each as WT_SPLIT
    // workgroups that take a quarter of every thread's rows, then the inner tiles one workgroup each.  A ring tile's source
    // window usually leaves the image, so it takes the per-pixel path below -- dependent gathers, ten times an inner
    // tile's duration: dispatched last and whole (the raster order's bottom row) they ran on alone for half the kernel's
    // time (75 us per 24 MP frame, of which 35 us with the GPU nearly empty).
    static_assert(RPT % WT_SPLIT == 0, "ring tiles split their rows evenly");
    int tx, ty, k_lo = 0, k_hi = RPT;
    {
        const bool ring = gx >= 3 && gy >= 3;
        const int nring = ring ? 2 * gx + 2 * (gy - 2) : 0;
        int id = blockIdx.x;
        if (id < nring * WT_SPLIT) {
            int r = id / WT_SPLIT;
            const int sub = id - r * WT_SPLIT;
            k_lo = sub * (RPT / WT_SPLIT);
            k_hi = k_lo + RPT / WT_SPLIT;
            if (r < gx) { ty = 0; tx = r; }
            else if (r < 2 * gx) { ty = gy - 1; tx = r - gx; }
            else { r -= 2 * gx; ty = 1 + (r >> 1); tx = (r & 1) ? gx - 1 : 0; }
        } else if (ring) {
            // inner rows in raster order over the FULL width (their two ring tiles are done above: those workgroups leave):
            // workgroup -> XCD stays `tile column mod 8` when the tile columns are a multiple of 8 (6000 px: 24), so an
            // XCD's L2 sees vertical stripes of tiles, whose source rows overlap
            id -= nring * WT_SPLIT;
            ty = id / gx;
            tx = id - ty * gx;
            ty += 1;
            if (tx == 0 || tx == gx - 1) return;
        } else {
            ty = id / gx;
            tx = id - ty * gx;
        }
    }
    const int x_t = tx * WT_W, y_t = ty * TH;
    const int xq = x_t + 4 * lane;
    const int* const tad = tab;           // column terms (warp_coord_tables)
    const int* const tbd = tab + w;
    const int* const tx0 = tab + 2 * w;   // row terms, + 16
    const int* const ty0 = tab + 2 * w + h;
    int ad[4], bd[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int x = min(xq + p, w - 1);
        ad[p] = tad[x];
        bd[p] = tbd[x];
    }
    // source bounding box of the tile (first taps; the +1 taps are added below)
    int sxmin = 0x7fffffff, sxmax = -0x7fffffff, symin = 0x7fffffff, symax = -0x7fffffff;
    {
        const int cxs[2] = {x_t, min(x_t + WT_W, w) - 1}, cys[2] = {y_t, min(y_t + TH, h) - 1};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int X0 = tx0[cys[i]], Y0 = ty0[cys[i]];   // (uniform: scalar loads)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int sx = ((X0 + tad[cxs[j]]) >> 5) >> 5;
                const int sy = ((Y0 + tbd[cxs[j]]) >> 5) >> 5;
                sxmin = min(sxmin, sx); sxmax = max(sxmax, sx);
                symin = min(symin, sy); symax = max(symax, sy);
            }
        }
    }
    const int nc = sxmax - sxmin + 2, nr = symax - symin + 2;
    // dwords per staged row: alignment shift (<= 3 bytes) + the pixels, rounded up to whole 16-byte chunks
    const int pitch = (((nc * BPP + 3 + 3) >> 2) + 3) & ~3;
    const size_t row_bytes = (size_t)w * BPP;
    const size_t start0 = ((size_t)max(symin, 0) * w + max(sxmin, 0)) * BPP;
    bool tiled = sxmin >= 0 && symin >= 0 && sxmax + 1 < w && symax + 1 < h && (long long)nr * pitch <= WT_LDS_DWORDS;
    if (tiled)   // the 16-byte reads of the last staged row must end inside the buffer
        tiled = start0 + (size_t)(nr - 1) * row_bytes + (size_t)pitch * 4 <= (size_t)h * w * BPP;
    if (tiled) {   // uniform over the workgroup
        // staging: wave wv takes rows wv, wv+4, ...; a lane one 16-byte chunk of a row, read from the row's first BYTE on
        // (global loads need no alignment here), so that byte b of every staged row is byte b of the source window and
        // the taps' LDS addresses need no per-row alignment term.  All loads of a pass are issued before the first LDS
        // store: one pass of latency per 12 rows instead of one per load.
        const char* g8 = reinterpret_cast<const char*>(src);
        const int cpr = pitch >> 2;
        constexpr int PASS = 12;
        for (int j = lane; j < cpr; j += 64) {
            for (int rb = wv; rb < nr; rb += 4 * PASS) {
                v4f buf[PASS];
#pragma unroll
                for (int i = 0; i < PASS; ++i) {
                    const int r = rb + 4 * i;
                    if (r < nr) {
                        __builtin_memcpy(&buf[i], g8 + start0 + (size_t)r * row_bytes + 16 * (size_t)j, 16);
                    }
                }
#pragma unroll
                for (int i = 0; i < PASS; ++i) {
                    const int r = rb + 4 * i;
                    if (r < nr) *reinterpret_cast<v4f*>(s_src + mul24(r, pitch) + 4 * j) = buf[i];
                }
            }
        }
        __syncthreads();
        const int pitch4 = pitch * 4;
        auto trow = [&](int k) __attribute__((always_inline)) {
            const int y = y_t + wvu * RPT + k;
            if (y >= h) return;
            const int X0 = tx0[y], Y0 = ty0[y];   // (uniform per wave: scalar loads)
            int v[4][3];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int X = (X0 + ad[p]) >> 5, Y = (Y0 + bd[p]) >> 5;
                const int sx = X >> 5, sy = Y >> 5, fx = X & 31, fy = Y & 31;
                const int r0 = sy - symin;
                if constexpr (sizeof(T) == 1) {
                    // 8-bit: cv2's weights are iw = wy * wx * 32 (wy, wx in 0..32), so
                    //   (sum q * iw + 16384) >> 15  ==  (sum q * wy * wx + 512) >> 10
                    // exactly (the fx = fy = 0 table entry 32767 / 1 gives q00 either way), and the sum splits into a row
                    // step and a column step in integers.  Row step: the pixel pair of a source row is 6 bytes q0c0 q0c1
                    // q0c2 q1c0 q1c1 q1c2; shifted by c bytes, channel c's two taps sit in bytes 0 and 3 of a dword and one
                    // v_dot4_u32_u8 with the weight bytes (32 - fx, 0, 0, fx) is q0 * wx0 + q1 * wx1 -- no unpacking.
                    const uint32_t Wx = (uint32_t)(32 - fx) | ((uint32_t)fx << 24);
                    uint32_t hrow[2][3];
                    const uint32_t cb = (uint32_t)(3 * (sx - sxmin));   // byte column inside the staged row
                    const char* sp0 = reinterpret_cast<const char*>(s_src) + mul24_vs(r0, pitch4) + (cb & ~3u);
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        const uint32_t* sp = reinterpret_cast<const uint32_t*>(sp0 + rr * pitch4);
                        const uint32_t d0 = sp[0], d1 = sp[1], d2 = sp[2];
                        const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, cb), hi = __builtin_amdgcn_alignbyte(d2, d1, cb);
                        const uint32_t l1 = __builtin_amdgcn_alignbyte(hi, lo, 1u), l2 = __builtin_amdgcn_alignbyte(hi, lo, 2u);
                        hrow[rr][0] = __builtin_amdgcn_udot4(lo, Wx, 0u, false);
                        hrow[rr][1] = __builtin_amdgcn_udot4(l1, Wx, 0u, false);
                        hrow[rr][2] = __builtin_amdgcn_udot4(l2, Wx, 0u, false);
                    }
                    const uint32_t wy0 = (uint32_t)(32 - fy), wy1 = (uint32_t)fy;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        uint32_t S = __umul24(hrow[1][c], wy1) + (__umul24(hrow[0][c], wy0) + 512u);
                        asm("" : "+v"(S));   // keeps the two 24-bit multiply-adds apart from the byte packing below
                        v[p][c] = (int)(S >> 10);   // <= 255
                    }
                } else {
                    // 16-bit: cv2 interpolates in float (weights from the 1/32 table); six elements per source row
                    int q[4][3];
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        const uint32_t A = (uint32_t)(BPP * (sx - sxmin));   // byte column inside the staged row
                        const uint32_t* sp = s_src + mul24_vs(r0 + rr, pitch) + (A >> 2);
                        const uint32_t d0 = sp[0], d1 = sp[1], d2 = sp[2], d3 = sp[3];
                        const uint32_t w0 = __builtin_amdgcn_alignbyte(d1, d0, A), w1 = __builtin_amdgcn_alignbyte(d2, d1, A),
                                       w2 = __builtin_amdgcn_alignbyte(d3, d2, A);
                        q[2 * rr][0] = w0 & 65535u; q[2 * rr][1] = w0 >> 16; q[2 * rr][2] = w1 & 65535u;
                        q[2 * rr + 1][0] = w1 >> 16; q[2 * rr + 1][1] = w2 & 65535u; q[2 * rr + 1][2] = w2 >> 16;
                    }
                    const float wx1 = fx * (1.0f / 32), wx0 = 1.0f - wx1, wy1 = fy * (1.0f / 32), wy0 = 1.0f - wy1;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float q0 = (float)q[0][c] * (wy0 * wx0), q1 = (float)q[1][c] * (wy0 * wx1);
                        const float q2 = (float)q[2][c] * (wy1 * wx0), q3 = (float)q[3][c] * (wy1 * wx1);
                        float sacc = q0 + q1;
                        sacc = sacc + q2;
                        sacc = sacc + q3;
                        v[p][c] = min(max((int)rintf(sacc), 0), 65535);
                    }
                }
            }
            const size_t px = (size_t)y * w + xq;
            if (VEC) {
                if (xq < w) {
                    uint32_t* d4 = reinterpret_cast<uint32_t*>(dst + px * 3);
                    if constexpr (sizeof(T) == 1) {
#pragma unroll
                        for (int d = 0; d < 3; ++d) {
                            uint32_t o = 0;
#pragma unroll
                            for (int b = 0; b < 4; ++b) o |= (uint32_t)v[(4 * d + b) / 3][(4 * d + b) % 3] << (8 * b);
                            d4[d] = o;
                        }
                    } else {
#pragma unroll
                        for (int d = 0; d < 6; ++d)
                            d4[d] = (uint32_t)v[(2 * d) / 3][(2 * d) % 3] | ((uint32_t)v[(2 * d + 1) / 3][(2 * d + 1) % 3] << 16);
                    }
                    if (valid) *reinterpret_cast<uint32_t*>(valid + px) = 0x01010101u;
                }
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if (xq + p < w) {
                        dst[(px + p) * 3 + 0] = (T)v[p][0];
                        dst[(px + p) * 3 + 1] = (T)v[p][1];
                        dst[(px + p) * 3 + 2] = (T)v[p][2];
                        if (valid) valid[px + p] = 1;
                    }
            }
        };
        if (k_hi - k_lo == RPT) {
#pragma unroll 2
            for (int k = 0; k < RPT; ++k) trow(k);
        } else {
#pragma unroll
            for (int k = 0; k < RPT / WT_SPLIT; ++k) trow(k_lo + k);
        }
        return;
    }
    // ---- per-pixel path (border tiles, large rotations): gathers from global memory, in-image flags per tap
    bool bad = false;   // some pixel of this thread has no full in-image footprint: its blur tile goes on the list
    auto row = [&](int k) __attribute__((always_inline)) {
        const int y = y_t + wvu * RPT + k;
        if (y >= h || xq >= w) return;
        const int X0 = tx0[y], Y0 = ty0[y];
        int v[4][3], ok[4];
        bool inside = xq + 3 < w;
        if (inside) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int sx = ((X0 + ad[3 * e]) >> 5) >> 5, sy = ((Y0 + bd[3 * e]) >> 5) >> 5;
                inside = inside && sx >= 0 && sx + 1 < w && sy >= 0 && sy + 1 < h - 1;
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            ok[p] = 0;
            v[p][0] = v[p][1] = v[p][2] = 0;
            if (inside) {
                warp_pixel_inside<T>(src, a, ad[p], bd[p], X0, Y0, v[p]);
                ok[p] = 1;
            } else if (xq + p < w) {
                warp_pixel_xy<T>(src, a, (X0 + ad[p]) >> 5, (Y0 + bd[p]) >> 5, v[p], ok[p]);
            }
        }
        const size_t px = (size_t)y * w + xq;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (xq + p < w) {
                dst[(px + p) * 3 + 0] = (T)v[p][0];
                dst[(px + p) * 3 + 1] = (T)v[p][1];
                dst[(px + p) * 3 + 2] = (T)v[p][2];
                if (valid) valid[px + p] = (uint8_t)ok[p];
                bad = bad || !ok[p];
            }
    };
    // constant trip counts (the compiler overlaps the rows' gathers): all of a thread's rows, or a ring workgroup's share
    if (k_hi - k_lo == RPT) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) row(k);
    } else {
#pragma unroll
        for (int k = 0; k < RPT / WT_SPLIT; ++k) row(k_lo + k);
    }
    // the border-blur pass works on 32 x 64 tiles that hold masked pixels: the warp marks them here (fire-and-forget
    // atomics from the few threads concerned) instead of a separate scan of the whole mask (38 us per 24 MP frame).
    // A thread's rows lie in one blur-tile row (TH divides BT_H = 32), its four pixels in one blur-tile column.
    // One atomic per 16 lanes (= one blur-tile column), not per thread: same-address atomics serialise in the L2, and a
    // ring tile under a rotation has hundreds of threads with masked pixels.
    if (tile_bitmap) {
        const unsigned long long b = __ballot(bad);
        if ((lane & 15) == 0 && ((b >> (lane & 48)) & 0xffffull) != 0) {
            const int bit = (y_t / 32) * blur_tiles_x + (xq >> 6);
            // the first marker of a tile also appends it to the blur pass's list (count = the dword 16 in front of the bitmap:
            // warp_scratch's layout; the list's order does not matter) -- round 5: tile_bitmap_to_list was a launch of its own
            const uint32_t old = atomicOr(&tile_bitmap[bit >> 5], 1u << (bit & 31));
            if (!((old >> (bit & 31)) & 1u)) tile_list[atomicAdd(tile_bitmap - 16, 1u)] = (uint32_t)bit;
        }
    }
}

// cv2.GaussianBlur on 8- / 16-bit images = OpenCV's bit-exact fixed-point path (oracle/align_oracle.c header): integer taps
// with 8 / 16 fractional bits that sum to exactly 1.0
struct GaussArgs {
    uint32_t k[32];
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
__device__ __forceinline__ void decode_px3u(uint2 v, uint32_t out[3]) {
    if constexpr (sizeof(T) == 1) {
        out[0] = v.x & 255u; out[1] = (v.x >> 8) & 255u; out[2] = (v.x >> 16) & 255u;
    } else {
        out[0] = v.x & 65535u; out[1] = v.x >> 16; out[2] = v.y;
    }
}
template <typename T>
__device__ __forceinline__ void decode_px3(uint2 v, float out[3]) {
    if constexpr (sizeof(T) == 1) {
        out[0] = (float)(v.x & 255u); out[1] = (float)((v.x >> 8) & 255u); out[2] = (float)((v.x >> 16) & 255u);
    } else {
        out[0] = (float)(v.x & 65535u); out[1] = (float)(v.x >> 16); out[2] = (float)v.y;
    }
}

// out = valid ? warp : gaussian_blur(warp), in place on `img`.  The pixels whose mask is 0 form a frame along the
// borders (a few columns at the sides, wedges where the frame rotated: some ten thousand of the 24 million pixels of a
// frame that moved by a few pixels).  Their 21 x 21 windows overlap almost completely, so the separable blur is
// evaluated densely, but only on the 32 x 64 tiles that contain a masked pixel:
//   scan     tiles with a masked pixel are marked in a bitmap (16 mask bytes per lane and step), the bitmap becomes a list;
//   blur     one workgroup per listed tile: the tile + r halo of the untouched image -> LDS (raw pixels), the horizontal
//            pass for every row of the patch -> LDS (fixed point, planar), the vertical pass for the tile's masked
//            pixels -> `side` (same index as the image);
//   scatter  the masked pixels of the listed tiles take their value from `side`.
// Arithmetic = align_oracle.c = OpenCV's bit-exact fixed-point GaussianBlur for 8- / 16-bit images: integer taps with 8 /
// 16 fractional bits (sum exactly 1.0), exact integer row and column sums, REFLECT101, round half up + saturate.  (Rounds
// 1-2 evaluated a float32 blur here -- a known deviation from cv2 on 8-bit frames, closed in round 3.  Round 1 also
// walked every 64-pixel chunk of the mask inside the blur pass and evaluated each masked pixel's 441 taps on its own:
// 80-160 us per 24 MP frame; this: ~30.)
constexpr int BT_H = 32, BT_W = 64;

__global__ __launch_bounds__(256) void mask_scan_tiles(const uint8_t* __restrict__ valid, int h, int w, int tiles_x,
                                                       uint32_t* __restrict__ bitmap) {
    // a fixed grid walks the mask 16 bytes per lane and step (one workgroup per 1024 pixels would be 23 000 workgroups
    // for a 24 MP frame: the dispatch alone took longer than reading the 24 MB)
    const size_t npix = (size_t)h * w;
    const size_t nvec = (npix + 15) / 16;
    const bool aligned = (reinterpret_cast<uintptr_t>(valid) & 15) == 0;
    for (size_t v0 = (size_t)blockIdx.x * blockDim.x; v0 < nvec; v0 += (size_t)gridDim.x * blockDim.x) {
        const size_t v = v0 + threadIdx.x, i0 = v * 16;
        uint32_t q[4] = {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u};
        if (v < nvec) {
            if (aligned && i0 + 16 <= npix) {
                const uint4 t = *reinterpret_cast<const uint4*>(valid + i0);
                q[0] = t.x; q[1] = t.y; q[2] = t.z; q[3] = t.w;
            } else {
                for (int b = 0; b < 16; ++b)
                    if (i0 + b < npix && valid[i0 + b] == 0) q[b >> 2] &= ~(0xffu << (8 * (b & 3)));
            }
        }
        const bool any = q[0] != 0x01010101u || q[1] != 0x01010101u || q[2] != 0x01010101u || q[3] != 0x01010101u;
        if (__ballot(any) == 0) continue;   // wave-uniform: the common case
        // tiles of the lane's first and last masked pixel (16 consecutive pixels span at most two tiles of one row,
        // or a row end)
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            int tile = -1;
            if (any) {
                for (int b = 0; b < 16; ++b) {
                    const int bb = pass ? 15 - b : b;
                    if (((q[bb >> 2] >> (8 * (bb & 3))) & 0xffu) == 0) {
                        const uint32_t pi = (uint32_t)(i0 + bb);
                        const uint32_t y = pi / (uint32_t)w, x = pi - y * (uint32_t)w;
                        tile = (int)((y / BT_H) * (uint32_t)tiles_x + x / BT_W);
                        break;
                    }
                }
            }
            unsigned long long rest = __ballot(tile >= 0);
            while (rest) {   // wave-uniform: one (fire-and-forget) atomic per distinct tile of the wave
                const int leader = __ffsll((long long)rest) - 1;
                const int t = __builtin_amdgcn_readlane(tile, leader);
                if ((int)(threadIdx.x & 63) == leader) atomicOr(&bitmap[t >> 5], 1u << (t & 31));
                rest &= ~__ballot(tile == t);
            }
        }
    }
}

// bitmap of tiles -> list (one workgroup; the order of the list does not matter)
__global__ __launch_bounds__(1024) void tile_bitmap_to_list(const uint32_t* __restrict__ bitmap, int nwords,
                                                            uint32_t* __restrict__ cnt, uint32_t* __restrict__ list) {
    for (int i = threadIdx.x; i < nwords; i += blockDim.x) {
        uint32_t m = bitmap[i];
        if (!m) continue;
        uint32_t pos = atomicAdd(cnt, (uint32_t)__popc(m));
        while (m) {
            const int b = __ffs((int)m) - 1;
            list[pos++] = (uint32_t)(i * 32 + b);
            m &= m - 1;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void border_blur_tiles(const T* __restrict__ img, const uint8_t* __restrict__ valid,
                                                         T* __restrict__ side, int h, int w, int tiles_x, GaussArgs g,
                                                         const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ list) {
    extern __shared__ uint32_t s_blur[];
    __shared__ int s_box[4];            // rows [0] .. [1], columns [2] .. [3] of the tile that hold a masked pixel
    __shared__ int s_cols[BT_W + 1];    // the masked columns, compact; [BT_W] = how many
    const int ks = g.ksize, r = ks / 2;
    const int PH = BT_H + 2 * r, PW = BT_W + 2 * r;
    constexpr int RAW = sizeof(T) == 1 ? 1 : 2;            // dwords per staged pixel
    uint32_t* sP = s_blur;                                  // PH x PW raw pixels
    uint32_t* sH = s_blur + PH * PW * RAW;   // [3][PH][BT_W] horizontal pass: 8.8 / 16.16 fixed point
    constexpr int BITS = sizeof(T) == 1 ? 8 : 16;           // fractional bits of the taps
    typedef typename std::conditional<sizeof(T) == 1, uint32_t, uint64_t>::type Acc;   // column sums: 16.16 / 32.32
    const uint32_t maxv = sizeof(T) == 1 ? 255u : 65535u;
    const size_t last = (size_t)h * w - 1;
    const int lane = threadIdx.x & 63;
    const uint32_t kreg = g.k[lane & 31];   // lane i holds tap i (ksize <= 31): v_readlane instead of a scalar load per tap
    auto kof = [&](int i) { return (uint32_t)__builtin_amdgcn_readlane((int)kreg, i); };
    const uint32_t n = *cnt;
    for (uint32_t e = blockIdx.x; e < n; e += gridDim.x) {
        const int t = (int)list[e];
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int y0 = ty * BT_H, x0 = tx * BT_W;
        // ---- which pixels of the tile are masked: the lane's 8 (rows yy0 + 4j, one column), and the tile's bounding box
        // of them -- a side-column tile needs the horizontal pass for 4 of its 64 columns, a top-edge tile for 23 of
        // the patch's 52 rows
        if (threadIdx.x < 4) s_box[threadIdx.x] = (threadIdx.x & 1) ? -1 : 1 << 20;
        __syncthreads();
        const int xx = threadIdx.x & 63, yy0 = threadIdx.x >> 6;
        uint32_t mymask = 0;
#pragma unroll
        for (int j = 0; j < BT_H / 4; ++j) {
            const int y = y0 + yy0 + 4 * j, x = x0 + xx;
            if (y < h && x < w && valid[(size_t)y * w + x] == 0) mymask |= 1u << j;
        }
        if (mymask) {
            atomicMin(&s_box[0], yy0 + 4 * (__ffs((int)mymask) - 1));
            atomicMax(&s_box[1], yy0 + 4 * (31 - __clz((int)mymask)));
            atomicMin(&s_box[2], xx);
            atomicMax(&s_box[3], xx);
        }
        {   // compact list of the masked columns (wave 0 sees the flags of all four waves through LDS)
            if (threadIdx.x < BT_W + 1) s_cols[threadIdx.x] = 0;
            __syncthreads();
            if (mymask) s_cols[xx] = 1;    // benign race: every writer stores 1
            __syncthreads();
            if (threadIdx.x < 64) {
                const unsigned long long m = __ballot(s_cols[lane] != 0);
                __builtin_amdgcn_s_waitcnt(0);
                if ((m >> lane) & 1ull) s_cols[__popcll(m & ((1ull << lane) - 1ull))] = lane;
                if (lane == 0) s_cols[BT_W] = __popcll(m);
            }
        }
        __syncthreads();
        const int ra = s_box[0], rb = s_box[1], ca = s_box[2], cb = s_box[3], ncols = s_cols[BT_W];
        if (rb >= 0) {   // uniform (a listed tile always has a masked pixel; guard anyway)
            const int nrows = rb - ra + 1 + 2 * r, ncp = cb - ca + 1 + 2 * r;   // patch rectangle: rows ra.., columns ca..
            // ---- stage that rectangle of the patch (REFLECT101 of the image), raw
            for (int i0 = threadIdx.x; i0 < nrows * ncp; i0 += 256 * 8) {
                uint2 raw[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = min(i0 + 256 * u, nrows * ncp - 1);
                    const int py = ra + i / ncp, px = ca + i % ncp;
                    raw[u] = load_px3_raw<T>(img, (size_t)r101_loop(y0 - r + py, h) * w + r101_loop(x0 - r + px, w), last);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + 256 * u;
                    if (i < nrows * ncp) {
                        const int py = ra + i / ncp, px = ca + i % ncp;
                        sP[(py * PW + px) * RAW] = raw[u].x;
                        if (RAW == 2) sP[(py * PW + px) * RAW + RAW - 1] = raw[u].y;
                    }
                }
            }
            __syncthreads();
            // ---- horizontal pass: the needed rows of the patch x the masked columns
            for (int i = threadIdx.x; i < nrows * ncols; i += 256) {
                const int py = ra + i / ncols, cx = s_cols[i % ncols];
                const uint32_t* p = sP + (py * PW + cx) * RAW;
                uint32_t row[3] = {0u, 0u, 0u};      // exact: sum k = 1.0, so the sums stay below 2^16 / 2^32
                for (int d0 = 0; d0 < ks; d0 += 7) {   // seven taps' LDS reads in flight at a time
                    uint2 rw[7];
#pragma unroll
                    for (int u = 0; u < 7; ++u) {
                        const int dx = min(d0 + u, ks - 1);
                        rw[u].x = p[dx * RAW];
                        rw[u].y = RAW == 2 ? p[dx * RAW + RAW - 1] : 0u;
                    }
#pragma unroll
                    for (int u = 0; u < 7; ++u) {
                        if (d0 + u < ks) {
                            uint32_t v[3];
                            decode_px3u<T>(rw[u], v);
                            const uint32_t k = kof(d0 + u);   // v_readlane ignores EXEC: fine in divergent code
#pragma unroll
                            for (int c = 0; c < 3; ++c) row[c] += k * v[c];
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) sH[(c * PH + py) * BT_W + cx] = row[c];
            }
            __syncthreads();
            // ---- vertical pass for the lane's masked pixels
#pragma unroll 1
            for (int j = 0; j < BT_H / 4; ++j) {
                if (!((mymask >> j) & 1u)) continue;
                const int yy = yy0 + 4 * j;
                Acc acc[3] = {0, 0, 0};
                for (int dy = 0; dy < ks; ++dy) {
                    const uint32_t kd = kof(dy);
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[c] += (Acc)kd * (Acc)sH[(c * PH + yy + dy) * BT_W + xx];
                }
                const size_t pi = (size_t)(y0 + yy) * w + (x0 + xx);
#pragma unroll
                for (int c = 0; c < 3; ++c) {   // round half up, saturate (ufixedpoint -> integer)
                    const Acc rr = (acc[c] + ((Acc)1 << (2 * BITS - 1))) >> (2 * BITS);
                    side[pi * 3 + c] = (T)(rr > (Acc)maxv ? (Acc)maxv : rr);
                }
            }
        }
        __syncthreads();   // the next tile's passes overwrite the LDS arrays
    }
}

template <typename T>
__global__ __launch_bounds__(256) void border_blur_scatter(T* __restrict__ img, const uint8_t* __restrict__ valid,
                                                           const T* __restrict__ side, int h, int w, int tiles_x,
                                                           const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ list) {
    const uint32_t n = *cnt;
    for (uint32_t e = blockIdx.x; e < n; e += gridDim.x) {
        const int t = (int)list[e];
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        for (int i = threadIdx.x; i < BT_H * BT_W; i += 256) {
            const int y = ty * BT_H + (i >> 6), x = tx * BT_W + (i & 63);
            if (y < h && x < w) {
                const size_t pi = (size_t)y * w + x;
                if (valid[pi] == 0) {
                    img[pi * 3 + 0] = side[pi * 3 + 0]; img[pi * 3 + 1] = side[pi * 3 + 1]; img[pi * 3 + 2] = side[pi * 3 + 2];
                }
            }
        }
    }
}

}  // namespace mi
