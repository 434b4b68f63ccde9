// kernels_ecc.hpp -- GPU transform estimator for frame alignment: Enhanced Correlation
// Coefficient (ECC) maximisation [Evangelidis & Psarakis, PAMI 2008] of a 4-DoF similarity
// (the motion model of the reference's default ALIGN_RIGID = cv2.estimateAffinePartial2D,
// algorithms/align.py:141-148), coarse-to-fine on a Gaussian pyramid.
//
// This is a NEW capability (north_star names it; the reference removed its ECC refinement in
// v0.1.4, CHANGELOG.md:206): there is no reference output to match, it is validated against
// ground-truth transforms with the reference author's tolerances
// (tests/test_0031_align_precision.py:62-65: angle < 0.005 deg, shift < 0.2 px, scale < 1e-4).
//
// Warp model, in coordinates centred on the image centre c:   W(x) = c + [a -b; b a](x - c) + t
// maps a pixel of the reference frame to the matching position in the moving frame, so the
// matrix cv2.warpAffine wants (moving -> reference) is the inverse of W.
#pragma once
#include "common.hpp"

namespace mi {

// gray (float) of a BGR image with the reference's sub-sampling (utils.py:79-86 img_subsample) folded in: the fast
// branch img[::s, ::s], or (area != 0) the integer-factor cv2.resize(INTER_AREA) -- the mean of the s x s block per
// channel, rounded to the image's integer type (half up; blocks that hang over the right / bottom edge average the
// pixels they have).  Any fixed positive combination of the channels works for registration.  out is h x w, the source
// is src_h x src_w pixels.
template <typename T>
__global__ void ecc_gray(const T* __restrict__ img, int src_h, int src_w, int h, int w, int s, int area,
                         float* __restrict__ out) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    const T* p = img + ((size_t)y * s * src_w + (size_t)x * s) * 3;
    float c[3];
    if (!area || s == 1) {
        c[0] = (float)p[0]; c[1] = (float)p[1]; c[2] = (float)p[2];
    } else {
        const int ny = min(s, src_h - y * s), nx = min(s, src_w - x * s);
        uint32_t sum[3] = {0u, 0u, 0u};
        for (int dy = 0; dy < ny; ++dy) {
            const T* q = p + (size_t)dy * src_w * 3;
            for (int dx = 0; dx < nx; ++dx) {
                sum[0] += q[3 * dx]; sum[1] += q[3 * dx + 1]; sum[2] += q[3 * dx + 2];
            }
        }
        // cv2.resize(INTER_AREA) by an integer factor [from memory, as align.img_subsample restates it]: whole blocks
        // (sum + 2) >> 2 for s == 2, else the float32 product sum * (1 / s^2) rounded half to even; the partial blocks
        // of a last row / column float32 sum / count, rounded half to even
        const int n = ny * nx;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (n == s * s) c[k] = s == 2 ? (float)((sum[k] + 2u) >> 2) : (float)__float2int_rn((float)sum[k] * (1.0f / (float)(s * s)));
            else c[k] = (float)__float2int_rn((float)sum[k] / (float)n);
        }
    }
    out[(size_t)y * w + x] = 0.114f * c[0] + 0.587f * c[1] + 0.299f * c[2];
}

// 5x5 binomial blur ([1 4 6 4 1]/16 separable, replicate border) then 2x decimation
__global__ void ecc_blur_down(const float* __restrict__ src, int h, int w, float* __restrict__ dst, int ho,
                              int wo, int decimate) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= wo || y >= ho) return;
    const float k[5] = {1.f / 16, 4.f / 16, 6.f / 16, 4.f / 16, 1.f / 16};
    const int sx = decimate ? 2 * x : x, sy = decimate ? 2 * y : y;
    float acc = 0.f;
    for (int dy = -2; dy <= 2; ++dy) {
        const int yy = min(max(sy + dy, 0), h - 1);
        float row = 0.f;
        for (int dx = -2; dx <= 2; ++dx) row += k[dx + 2] * src[(size_t)yy * w + min(max(sx + dx, 0), w - 1)];
        acc += k[dy + 2] * row;
    }
    dst[(size_t)y * wo + x] = acc;
}

// The two kernels above, restructured for bandwidth (same arithmetic in the same order: bit-identical output).
//
// ecc_gray_s2: the common sub-sampling factor 2; a thread produces four outputs of one row from 8 consecutive source
// pixels (24 / 48 bytes, loaded as three / six 8-byte words) of one row (fast) or two rows (area), instead of
// byte-wide loads per channel and output.  The launch covers ceil(w / 4) x h threads; threads whose 8 source pixels
// are not all inside the row use the scalar form.
// four consecutive gray values (x0 .. x0+3 of row y) into o[]; entries at x >= w are left alone
template <typename T, bool AREA>
__device__ __forceinline__ void ecc_gray4(const T* __restrict__ img, int src_h, int src_w, int w, int x0, int y, float o[4]) {
    const bool rows_ok = !AREA || 2 * y + 1 < src_h;
    if (x0 + 3 < w && 2 * x0 + 7 < src_w && rows_ok) {
        constexpr int NW = 24 * (int)sizeof(T) / 8;   // 8-byte words per row segment
        uint64_t r0[NW], r1[NW];
        const T* p0 = img + ((size_t)2 * y * src_w + 2 * x0) * 3;
#pragma unroll
        for (int k = 0; k < NW; ++k) __builtin_memcpy(&r0[k], (const char*)p0 + 8 * k, 8);
        if constexpr (AREA) {
            const T* p1 = p0 + (size_t)src_w * 3;
#pragma unroll
            for (int k = 0; k < NW; ++k) __builtin_memcpy(&r1[k], (const char*)p1 + 8 * k, 8);
        }
        auto elem = [&](const uint64_t* r, int e) -> uint32_t {   // element e (0..23) of the row segment
            constexpr int PER = 8 / (int)sizeof(T);
            return (uint32_t)(r[e / PER] >> (8 * (int)sizeof(T) * (e % PER))) & (sizeof(T) == 1 ? 0xffu : 0xffffu);
        };
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float c[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if constexpr (AREA) {
                    const uint32_t sum = elem(r0, 6 * q + k) + elem(r0, 6 * q + 3 + k) + elem(r1, 6 * q + k) + elem(r1, 6 * q + 3 + k);
                    c[k] = (float)((sum + 2u) >> 2);
                } else {
                    c[k] = (float)elem(r0, 6 * q + k);
                }
            }
            o[q] = 0.114f * c[0] + 0.587f * c[1] + 0.299f * c[2];
        }
        return;
    }
    for (int x = x0; x < min(x0 + 4, w); ++x) {   // row / image ends: the scalar form of ecc_gray
        const T* p = img + ((size_t)y * 2 * src_w + (size_t)x * 2) * 3;
        float c[3];
        if (!AREA) {
            c[0] = (float)p[0]; c[1] = (float)p[1]; c[2] = (float)p[2];
        } else {
            const int ny = min(2, src_h - y * 2), nx = min(2, src_w - x * 2);
            uint32_t sum[3] = {0u, 0u, 0u};
            for (int dy = 0; dy < ny; ++dy)
                for (int dx = 0; dx < nx; ++dx)
                    for (int k = 0; k < 3; ++k) sum[k] += p[((size_t)dy * src_w + dx) * 3 + k];
            const int n = ny * nx;
            for (int k = 0; k < 3; ++k)
                c[k] = n == 4 ? (float)((sum[k] + 2u) >> 2) : (float)__float2int_rn((float)sum[k] / (float)n);
        }
        o[x - x0] = 0.114f * c[0] + 0.587f * c[1] + 0.299f * c[2];
    }
}

template <typename T, bool AREA>
__global__ __launch_bounds__(256) void ecc_gray_s2(const T* __restrict__ img, int src_h, int src_w, int h, int w,
                                                   float* __restrict__ out) {
    const int x0 = 4 * (blockIdx.x * blockDim.x + threadIdx.x), y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x0 >= w || y >= h) return;
    float o[4];
    ecc_gray4<T, AREA>(img, src_h, src_w, w, x0, y, o);
    float* d = out + (size_t)y * w + x0;
    for (int q = 0; q < min(4, w - x0); ++q) d[q] = o[q];
}

// four consecutive gray values of an INTERIOR tile (every source pixel inside the frame), 8-bit frames, area rule: the
// channel sums of a 2 x 2 source block straight from the loaded words -- v_dot4_u32_u8 with a 0 / 1 weight per byte adds the
// bytes of one channel that a word holds to the running sum (which starts at the rounding constant 2): at most four
// instructions per sum instead of four byte extractions and three adds.  Same integers, same floats as ecc_gray4.
__device__ __forceinline__ void ecc_gray4_u8_area(const uint32_t r0[6], const uint32_t r1[6], float o[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int A = 6 * q + k, B = A + 3;                     // byte offsets of the pair's two pixels, channel k
            const int a = A >> 2, b = B >> 2;
            const uint32_t wa = 1u << (8 * (A & 3)), wb = 1u << (8 * (B & 3));
            uint32_t sum = 2u;
            if (a == b) {
                sum = __builtin_amdgcn_udot4(r0[a], wa | wb, sum, false);
                sum = __builtin_amdgcn_udot4(r1[a], wa | wb, sum, false);
            } else {
                sum = __builtin_amdgcn_udot4(r0[a], wa, sum, false);
                sum = __builtin_amdgcn_udot4(r0[b], wb, sum, false);
                sum = __builtin_amdgcn_udot4(r1[a], wa, sum, false);
                sum = __builtin_amdgcn_udot4(r1[b], wb, sum, false);
            }
            c[k] = (float)(sum >> 2);
        }
        o[q] = 0.114f * c[0] + 0.587f * c[1] + 0.299f * c[2];
    }
}

// ecc_pyramid2: sub-sampled gray, level 0 (= 5 x 5 binomial blur of it) and level 1 (= blur + 2x decimation of level 0) of
// one frame in ONE pass over it: the three kernels above / below read and wrote the gray image and level 0 twice each
// (100 us per 24 MP frame for what is one 72 MB read and 30 MB of writes).  A workgroup owns a 64 x 32 tile of level 0 and
// the 32 x 16 tile of level 1 under it: gray patch (halo 4) -> row pass -> level-0 patch (halo 2: what level 1's taps
// reach) -> row pass -> level-1 tile, all through LDS.  Replicate borders = clamped reads of in-image patch entries, the
// sums in ecc_blur_tile's order: the pyramids are bit-identical to the separate kernels' (GPU test).
//
// Round 4: the kernel was VALU-bound (~1350 instructions per thread: an index division, five clamps and five address
// computations per output and pass).  A tile whose patches lie inside the image -- all but the frame's rim -- takes the
// INTERIOR path: no clamps, four outputs per thread from two 16-byte LDS reads in the row passes, a register window sliding
// down a column in the column passes, the 8-bit area sums by v_dot4 (above).  Every sum keeps its order (the leading
// `0.f + k[0] * v` of the generic form is exact: the products are >= +0), so both paths give the same bits.
constexpr int ECC_P2_GS = 76;   // gray patch row stride: a multiple of 4 (16-byte LDS accesses of the interior path)

template <typename T, bool AREA>
__device__ __forceinline__ void ecc_pyramid2_interior(const T* __restrict__ img, int src_h, int src_w, int w,
                                                      float* __restrict__ L0, int w1, float* __restrict__ L1,
                                                      float* sG, float* sR, float* sP, int x0, int y0, int tid) {
    constexpr int TW = 64, TH = 32;
    constexpr int GW = TW + 8, GH = TH + 8, GS = ECC_P2_GS;
    constexpr int PW = TW + 4, PH = TH + 4, PS = PW + 1;
    constexpr float k0 = 1.f / 16, k1 = 4.f / 16, k2 = 6.f / 16;
    // ---- gray patch: 40 rows x 18 groups of four
    if constexpr (AREA && sizeof(T) == 1) {
        // a thread's three groups: all six 24-byte loads issued before the first sum (the third group of the last 48
        // threads does not exist: they load their second one again and drop it)
        constexpr int NG = GH * (GW / 4), NR = (NG + 255) / 256;
        uint32_t r0[NR][6], r1[NR][6];
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            const int g = min(tid + 256 * n, NG - 1);
            const int r = g / (GW / 4), c = 4 * (g - r * (GW / 4));
            const uint8_t* p0 = (const uint8_t*)img + ((size_t)2 * (y0 - 4 + r) * src_w + 2 * (x0 - 4 + c)) * 3;
            __builtin_memcpy(r0[n], p0, 24);
            __builtin_memcpy(r1[n], p0 + (size_t)src_w * 3, 24);
        }
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            const int g = tid + 256 * n;
            if (g < NG) {
                const int r = g / (GW / 4), c = 4 * (g - r * (GW / 4));
                float o[4];
                ecc_gray4_u8_area(r0[n], r1[n], o);
                *reinterpret_cast<float4*>(sG + r * GS + c) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    } else {
        for (int g = tid; g < GH * (GW / 4); g += 256) {
            const int r = g / (GW / 4), c = 4 * (g - r * (GW / 4));
            float o[4];
            ecc_gray4<T, AREA>(img, src_h, src_w, w, x0 - 4 + c, y0 - 4 + r, o);
            *reinterpret_cast<float4*>(sG + r * GS + c) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    __syncthreads();
    // ---- level 0, row pass: 40 rows x 17 groups of four columns of the level-0 patch (patch column c reads gray c .. c+4)
    for (int g = tid; g < GH * (PW / 4); g += 256) {
        const int r = g / (PW / 4), c = 4 * (g - r * (PW / 4));
        const float4 a = *reinterpret_cast<const float4*>(sG + r * GS + c);
        const float4 b = *reinterpret_cast<const float4*>(sG + r * GS + c + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float row = k0 * v[q];
            row += k1 * v[q + 1];
            row += k2 * v[q + 2];
            row += k1 * v[q + 3];
            row += k0 * v[q + 4];
            sR[r * PS + c + q] = row;
        }
    }
    __syncthreads();
    // ---- level 0, column pass: 68 columns x 3 runs of 12 rows; a run slides a 5-row window down its column
    if (tid < 3 * PW) {
        const int run = tid / PW, c = tid - run * PW;
        const int r0 = 12 * run;                      // patch rows r0 .. r0+11 read row-pass rows r0 .. r0+15
        const float* col = sR + r0 * PS + c;
        float v0 = col[0], v1 = col[PS], v2 = col[2 * PS], v3 = col[3 * PS];
        const bool cin = c >= 2 && c < PW - 2;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const float v4 = col[(j + 4) * PS];
            float acc = k0 * v0;
            acc += k1 * v1;
            acc += k2 * v2;
            acc += k1 * v3;
            acc += k0 * v4;
            const int r = r0 + j;
            sP[r * PS + c] = acc;
            if (cin && r >= 2 && r < PH - 2) L0[(size_t)(y0 - 2 + r) * w + (x0 - 2 + c)] = acc;
            v0 = v1; v1 = v2; v2 = v3; v3 = v4;
        }
    }
    __syncthreads();
    // ---- level 1, row pass (decimating): 36 rows x 32 columns; column j reads patch columns 2j .. 2j+4
    float* sR1 = sR;
    constexpr int R1S = TW / 2 + 1;
    for (int i = tid; i < PH * (TW / 2); i += 256) {
        const int r = i >> 5, j = i & 31;
        const float* p = sP + r * PS + 2 * j;
        float row = k0 * p[0];
        row += k1 * p[1];
        row += k2 * p[2];
        row += k1 * p[3];
        row += k0 * p[4];
        sR1[r * R1S + j] = row;
    }
    __syncthreads();
    // ---- level 1, column pass: 16 x 32 outputs, two per thread (rows yl and yl + 8)
    {
        const int j = tid & 31, yl = tid >> 5;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int y = yl + 8 * e;
            const float* p = sR1 + (2 * y) * R1S + j;
            float acc = k0 * p[0];
            acc += k1 * p[R1S];
            acc += k2 * p[2 * R1S];
            acc += k1 * p[3 * R1S];
            acc += k0 * p[4 * R1S];
            L1[(size_t)(y0 / 2 + y) * w1 + (x0 / 2 + j)] = acc;
        }
    }
}

template <typename T, bool AREA>
__device__ __forceinline__ void ecc_pyramid2_body(const T* __restrict__ img, int src_h, int src_w, int h, int w,
                                                  float* __restrict__ L0, int h1, int w1, float* __restrict__ L1) {
    constexpr int TW = 64, TH = 32;
    constexpr int GW = TW + 8, GH = TH + 8, GS = ECC_P2_GS;   // gray patch: level coordinates x0-4 .., y0-4 ..
    constexpr int PW = TW + 4, PH = TH + 4, PS = PW + 1;      // level-0 patch: x0-2 .., y0-2 ..
    // two LDS images (23 KB: six workgroups per CU): the gray patch, whose space the level-0 patch takes once the row pass
    // has consumed it, and the row pass of level 0, whose space the row pass of level 1 takes
    __shared__ __attribute__((aligned(16))) float sG[GH * GS];
    __shared__ float sR[GH * PS];
    float* sP = sG;
    static_assert(PH * PS <= GH * GS, "the level-0 patch aliases the gray patch");
    static_assert(PH * (TW / 2 + 1) <= GH * PS, "level 1's row pass aliases level 0's");
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, tid = threadIdx.x;
    // interior: the gray patch and the source pixels under it inside the frame (the level-1 tile then is, too)
    if (x0 >= 4 && y0 >= 4 && x0 + TW + 4 <= w && y0 + TH + 4 <= h && 2 * (x0 + TW + 4) <= src_w && 2 * (y0 + TH + 4) <= src_h) {
        ecc_pyramid2_interior<T, AREA>(img, src_h, src_w, w, L0, w1, L1, sG, sR, sP, x0, y0, tid);
        return;
    }
    const float k[5] = {1.f / 16, 4.f / 16, 6.f / 16, 4.f / 16, 1.f / 16};
    // ---- gray patch, four entries per thread and step (the patch starts on a multiple of 4)
    for (int g = tid; g < GH * (GW / 4); g += 256) {
        const int r = g / (GW / 4), c = 4 * (g - r * (GW / 4));
        const int y = y0 - 4 + r, x = x0 - 4 + c;
        if (y < 0 || y >= h || x < 0 || x >= w) continue;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        ecc_gray4<T, AREA>(img, src_h, src_w, w, x, y, o);
#pragma unroll
        for (int q = 0; q < 4; ++q) sG[r * GS + c + q] = o[q];
    }
    __syncthreads();
    // ---- level 0, row pass: the in-image rows of the gray patch x the in-image columns of the level-0 patch
    for (int i = tid; i < GH * PW; i += 256) {
        const int r = i / PW, c = i - r * PW;
        const int y = y0 - 4 + r, xx = x0 - 2 + c;
        if (y < 0 || y >= h || xx < 0 || xx >= w) continue;
        float row = 0.f;
#pragma unroll
        for (int t = 0; t < 5; ++t) row += k[t] * sG[r * GS + (min(max(xx - 2 + t, 0), w - 1) - (x0 - 4))];
        sR[r * PS + c] = row;
    }
    __syncthreads();
    // ---- level 0, column pass -> patch; the tile itself also goes to memory
    for (int i = tid; i < PH * PW; i += 256) {
        const int r = i / PW, c = i - r * PW;
        const int yy = y0 - 2 + r, xx = x0 - 2 + c;
        if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 5; ++t) acc += k[t] * sR[(min(max(yy - 2 + t, 0), h - 1) - (y0 - 4)) * PS + c];
        sP[r * PS + c] = acc;
        if (r >= 2 && r < PH - 2 && c >= 2 && c < PW - 2) L0[(size_t)yy * w + xx] = acc;
    }
    __syncthreads();
    // ---- level 1, row pass (decimating): every in-image row of the level-0 patch x the tile's 32 columns
    float* sR1 = sR;
    constexpr int R1S = TW / 2 + 1;
    for (int i = tid; i < PH * (TW / 2); i += 256) {
        const int r = i / (TW / 2), j = i - r * (TW / 2);
        const int yy = y0 - 2 + r, xj = x0 / 2 + j;
        if (yy < 0 || yy >= h || xj >= w1) continue;
        float row = 0.f;
#pragma unroll
        for (int t = 0; t < 5; ++t) row += k[t] * sP[r * PS + (min(max(2 * xj - 2 + t, 0), w - 1) - (x0 - 2))];
        sR1[r * R1S + j] = row;
    }
    __syncthreads();
    // ---- level 1, column pass
    for (int i = tid; i < (TH / 2) * (TW / 2); i += 256) {
        const int yl = i / (TW / 2), j = i - yl * (TW / 2);
        const int yi = y0 / 2 + yl, xj = x0 / 2 + j;
        if (yi >= h1 || xj >= w1) continue;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 5; ++t) acc += k[t] * sR1[(min(max(2 * yi - 2 + t, 0), h - 1) - (y0 - 2)) * R1S + j];
        L1[(size_t)yi * w1 + xj] = acc;
    }
}

template <typename T, bool AREA>
__global__ __launch_bounds__(256) void ecc_pyramid2(const T* __restrict__ img, int src_h, int src_w, int h, int w,
                                                    float* __restrict__ L0, int h1, int w1, float* __restrict__ L1) {
    ecc_pyramid2_body<T, AREA>(img, src_h, src_w, h, w, L0, h1, w1, L1);
}

// the frames of a batch in ONE launch (blockIdx.z = frame k: its source anywhere in memory, its levels in slot k of the
// batch's level images): a single frame's 2961 workgroups are 1.4 rounds of the chip, a batch of 16 fills it 23 times over
struct EccFramePtrs { const void* p[128]; };
template <typename T, bool AREA>
__global__ __launch_bounds__(256) void ecc_pyramid2_batch(EccFramePtrs fr, int src_h, int src_w, int h, int w,
                                                          float* __restrict__ L0, int h1, int w1, float* __restrict__ L1) {
    const int k = blockIdx.z;
    ecc_pyramid2_body<T, AREA>((const T*)fr.p[k], src_h, src_w, h, w, L0 + (size_t)k * h * w, h1, w1, L1 + (size_t)k * h1 * w1);
}

// ecc_blur_tile: ecc_blur_down through LDS -- a workgroup stages the source patch of a 64 x 16 output tile (replicate
// border), runs the 5-tap row pass into a second LDS image and the 5-tap column pass from there; every source value is
// read from HBM once instead of up to 25 times through the texture units.
// blockIdx.z = image of a batch (consecutive images of h x w / ho x wo floats).
template <int DEC>
__global__ __launch_bounds__(256) void ecc_blur_tile(const float* __restrict__ src, int h, int w, float* __restrict__ dst,
                                                     int ho, int wo) {
    src += (size_t)blockIdx.z * h * w;
    dst += (size_t)blockIdx.z * ho * wo;
    constexpr int TW = 64, TH = 16, S = DEC ? 2 : 1;
    constexpr int IW = S * (TW - 1) + 5, IH = S * (TH - 1) + 5, IS = IW | 1;
    __shared__ float s_in[IH * IS];
    __shared__ float s_row[IH * TW];
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, tid = threadIdx.x;
    const float k[5] = {1.f / 16, 4.f / 16, 6.f / 16, 4.f / 16, 1.f / 16};
    for (int i = tid; i < IH * IW; i += 256) {
        const int r = i / IW, c = i - r * IW;
        const int yy = min(max(S * y0 - 2 + r, 0), h - 1), xx = min(max(S * x0 - 2 + c, 0), w - 1);
        s_in[r * IS + c] = src[(size_t)yy * w + xx];
    }
    __syncthreads();
    for (int i = tid; i < IH * TW; i += 256) {
        const int r = i / TW, x = i - r * TW;
        const float* p = s_in + r * IS + S * x;
        float row = 0.f;
#pragma unroll
        for (int t = 0; t < 5; ++t) row += k[t] * p[t];
        s_row[i] = row;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TH * TW / 256; ++j) {
        const int i = tid + j * 256, yl = i / TW, x = i - yl * TW;
        if (y0 + yl >= ho || x0 + x >= wo) continue;
        const float* p = s_row + (S * yl) * TW + x;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 5; ++t) acc += k[t] * p[t * TW];
        dst[(size_t)(y0 + yl) * wo + x0 + x] = acc;
    }
}

__device__ __forceinline__ float bilerp(const float* __restrict__ im, int w, int x0, int y0, float fx, float fy) {
    const float* p = im + (size_t)y0 * w + x0;
    const float a = p[0] + fx * (p[1] - p[0]);
    const float b = p[w] + fx * (p[w + 1] - p[w]);
    return a + fy * (b - a);
}

constexpr int ECC_NSUM = 28;  // n, Siw, Sir, Siw2, Sir2, Siwir, SJ[4], SJiw[4], SJir[4], H[10]

struct EccParams {
    double a, b, tx, ty;  // similarity about the image centre
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

constexpr int ECC_MAX_BLOCKS = 1024;
constexpr int ECC_MAXF = 128;   // moving frames per batched launch (blockIdx.y)

// The Gauss-Newton state of one moving frame, resident in device memory: the accumulation kernel reads the parameters
// from it and the block that finishes the frame's reduction UPDATES it (the 4 x 4 solves, the step, the convergence test)
// -- an iteration is one launch and no host round trip; the host enqueues a few iterations at a time and looks at the
// `active` flags only then.  (Round 2 solved on the host: ~110 synchronisations per batch were what an estimate of 128
// frames spent most of its 58 ms on.)
struct EccState {
    double a, b, tx, ty;     // similarity about the image centre of the current level
    double T0, T1;           // translation in origin coordinates (carried from level to level)
    double rho, last_rho;
    int iters, failed, active;
    int tslot;               // the frame's template: -1 = the handle's reference, s >= 0 = the pyramid of the batch's frame s
                             // (mi_aligner_estimate_pairs: every frame against a neighbour of the same batch)
};

// H d = r for the symmetric 4 x 4 H (10 values, row-major upper triangle): column-scaled Gaussian elimination with
// partial pivoting, in double.  false: not positive on the diagonal / singular.
__device__ inline bool ecc_solve4(const double Hs[10], const double r[4], double d[4]) {
    double A[4][5], Hm[4][4], sc[4];
    int k = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = i; j < 4; ++j) Hm[i][j] = Hm[j][i] = Hs[k++];
    for (int i = 0; i < 4; ++i) {
        if (!(Hm[i][i] > 0)) return false;
        sc[i] = 1.0 / sqrt(Hm[i][i]);
    }
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 4; ++j) A[i][j] = Hm[i][j] * sc[i] * sc[j];
        A[i][4] = r[i] * sc[i];
    }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int i = c + 1; i < 4; ++i)
            if (fabs(A[i][c]) > fabs(A[piv][c])) piv = i;
        if (fabs(A[piv][c]) < 1e-12) return false;
        if (piv != c)
            for (int j = 0; j < 5; ++j) { const double t = A[c][j]; A[c][j] = A[piv][j]; A[piv][j] = t; }
        for (int i = c + 1; i < 4; ++i) {
            const double f = A[i][c] / A[c][c];
            for (int j = c; j < 5; ++j) A[i][j] -= f * A[c][j];
        }
    }
    for (int i = 3; i >= 0; --i) {
        double v = A[i][4];
        for (int j = i + 1; j < 4; ++j) v -= A[i][j] * d[j];
        d[i] = v / A[i][i];
    }
    for (int i = 0; i < 4; ++i) d[i] *= sc[i];
    return true;
}

// One forward-additive ECC step from the 28 sums of a frame (Evangelidis & Psarakis; the same arithmetic the host ran in
// round 2): correlation, the two projections, lambda, the parameter update, the stopping rule.
// `reach`: distance of the level's corners from its centre (a rotation / scale step moves them the most).
__device__ inline void ecc_update(EccState& f, const double* S, double reach, double eps) {
    ++f.iters;
    const double cnt = S[0];
    if (cnt < 64) { f.failed = 1; f.active = 0; return; }   // the images do not overlap
    const double mw = S[1] / cnt, mr = S[2] / cnt;
    const double wn2 = S[3] - cnt * mw * mw, rn2 = S[4] - cnt * mr * mr, corr = S[5] - cnt * mw * mr;
    if (!(wn2 > 0) || !(rn2 > 0)) { f.failed = 1; f.active = 0; return; }   // constant image
    f.rho = corr / sqrt(wn2 * rn2);
    double ip[4], tp[4], Hi_ip[4];
    for (int q = 0; q < 4; ++q) {
        ip[q] = S[10 + q] - mw * S[6 + q];
        tp[q] = S[14 + q] - mr * S[6 + q];
    }
    if (!ecc_solve4(&S[18], ip, Hi_ip)) { f.active = 0; return; }
    double ipH = 0, tpH = 0;
    for (int q = 0; q < 4; ++q) { ipH += ip[q] * Hi_ip[q]; tpH += tp[q] * Hi_ip[q]; }
    const double lam_n = wn2 - ipH, lam_d = corr - tpH;
    if (!(lam_d > 0)) { f.active = 0; return; }   // the algorithm stopped before its convergence
    const double lam = lam_n / lam_d;
    double ep[4], dp[4];
    for (int q = 0; q < 4; ++q) ep[q] = lam * tp[q] - ip[q];
    if (!ecc_solve4(&S[18], ep, dp)) { f.active = 0; return; }
    f.a += dp[0]; f.b += dp[1]; f.tx += dp[2]; f.ty += dp[3];
    // converged when the update moves no pixel of this level by more than 0.002 px, or when rho stalls
    const double move = (fabs(dp[0]) + fabs(dp[1])) * reach + fabs(dp[2]) + fabs(dp[3]);
    if (move < 2e-3 || fabs(f.rho - f.last_rho) < eps) f.active = 0;
    f.last_rho = f.rho;
}

// level transitions of the whole batch: origin coordinates u = A x + T with A = [a -b; b a]  <->  centred parameters
// t = T - c + A c of the level about to be solved; T doubles on the way to the next finer level
__global__ void ecc_level_begin(EccState* __restrict__ st, int n, double cx, double cy) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    EccState& f = st[k];
    f.tx = f.T0 - cx + (f.a * cx - f.b * cy);
    f.ty = f.T1 - cy + (f.b * cx + f.a * cy);
    f.last_rho = -2.0;
    f.active = !f.failed;
}
__global__ void ecc_level_end(EccState* __restrict__ st, int n, double cx, double cy, int finer) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    EccState& f = st[k];
    f.T0 = f.tx + cx - (f.a * cx - f.b * cy);
    f.T1 = f.ty + cy - (f.b * cx + f.a * cy);
    if (finer) { f.T0 *= 2.0; f.T1 *= 2.0; }
}

// One Gauss-Newton iteration for up to ECC_MAXF moving frames against one template (blockIdx.y = frame; frames whose
// `active` flag is 0 return at once): every reference pixel samples the moving image (and its gradients) at W(x) and adds
// its terms to 28 sums (double).  `step`: pixel stride (sub-sampling of the sum at the finest levels).  Deterministic
// two-stage reduction in one launch: each block writes its 28 partial sums to partial[frame][block][28]; the block that
// draws the frame's last ticket adds the partials in a fixed order, takes the step (ecc_update) and re-arms the ticket.
// img: frame f at base + f * fstride.  The image gradients (central differences) are taken on the fly
// from the 4x4 neighbourhood of the sample point -- 12 loads from one array instead of 4 from each of
// three, and no gradient images to build or keep.
__global__ __launch_bounds__(256) void ecc_accumulate(const float* __restrict__ tmpl, const float* __restrict__ img,
                               size_t fstride, int h, int w, EccState* __restrict__ state, int step,
                               double* __restrict__ partial, unsigned int* __restrict__ ticket, double reach, double eps) {
    const int f = blockIdx.y;
    if (!state[f].active) return;
    const EccParams p = {state[f].a, state[f].b, state[f].tx, state[f].ty};
    if (state[f].tslot >= 0) tmpl = img + (size_t)state[f].tslot * fstride;
    img += (size_t)f * fstride;
    partial += (size_t)f * ECC_MAX_BLOCKS * ECC_NSUM;
    ticket += f;
    double acc[ECC_NSUM];
#pragma unroll
    for (int i = 0; i < ECC_NSUM; ++i) acc[i] = 0.0;
    const float cx = 0.5f * (w - 1), cy = 0.5f * (h - 1);
    const float a = (float)p.a, b = (float)p.b, tx = (float)p.tx, ty = (float)p.ty;
    const int nx = (w + step - 1) / step, ny = (h + step - 1) / step;
    const unsigned total = (unsigned)nx * ny;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int y = (int)(i / nx) * step, x = (int)(i % nx) * step;
        const float xc = x - cx, yc = y - cy;
        const float u = cx + a * xc - b * yc + tx, v = cy + b * xc + a * yc + ty;
        const int x0 = (int)floorf(u), y0 = (int)floorf(v);
        if (x0 < 1 || y0 < 1 || x0 >= w - 2 || y0 >= h - 2) continue;
        const float fx = u - x0, fy = v - y0;
        // rows y0-1 .. y0+2, columns x0-1 .. x0+2 (all inside the image by the test above)
        const float* r1 = img + (size_t)y0 * w + x0;   // (y0, x0)
        const float* r0 = r1 - w;
        const float* r2 = r1 + w;
        const float* r3 = r2 + w;
        const float a_m = r1[-1], a_0 = r1[0], a_1 = r1[1], a_2 = r1[2];
        const float b_m = r2[-1], b_0 = r2[0], b_1 = r2[1], b_2 = r2[2];
        const float t_0 = r0[0], t_1 = r0[1], u_0 = r3[0], u_1 = r3[1];
        const float iw = (a_0 + fx * (a_1 - a_0)) + fy * ((b_0 + fx * (b_1 - b_0)) - (a_0 + fx * (a_1 - a_0)));
        // gx at the four corners: 0.5 * (right - left); gy: 0.5 * (below - above)
        const float gxa0 = 0.5f * (a_1 - a_m), gxa1 = 0.5f * (a_2 - a_0), gxb0 = 0.5f * (b_1 - b_m), gxb1 = 0.5f * (b_2 - b_0);
        const float gya0 = 0.5f * (b_0 - t_0), gya1 = 0.5f * (b_1 - t_1), gyb0 = 0.5f * (u_0 - a_0), gyb1 = 0.5f * (u_1 - a_1);
        const float gxt = gxa0 + fx * (gxa1 - gxa0), gxb = gxb0 + fx * (gxb1 - gxb0);
        const float gyt = gya0 + fx * (gya1 - gya0), gyb = gyb0 + fx * (gyb1 - gyb0);
        const float dgx = gxt + fy * (gxb - gxt), dgy = gyt + fy * (gyb - gyt);
        const float ir = tmpl[(size_t)y * w + x];
        const float J[4] = {dgx * xc + dgy * yc, -dgx * yc + dgy * xc, dgx, dgy};
        acc[0] += 1.0;
        acc[1] += iw;
        acc[2] += ir;
        acc[3] += (double)iw * iw;
        acc[4] += (double)ir * ir;
        acc[5] += (double)iw * ir;
        int hk = 18;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[6 + k] += J[k];
            acc[10 + k] += (double)J[k] * iw;
            acc[14 + k] += (double)J[k] * ir;
#pragma unroll
            for (int m = k; m < 4; ++m) acc[hk++] += (double)J[k] * J[m];
        }
    }
    // wave reduction in registers, 4 waves through LDS
    __shared__ double red[4][ECC_NSUM];
    __shared__ bool last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int s = 0; s < ECC_NSUM; ++s) {
        const double v = wave_sum(acc[s]);
        if (lane == 0) red[wave][s] = v;
    }
    __syncthreads();
    if (threadIdx.x < ECC_NSUM)
        partial[(size_t)blockIdx.x * ECC_NSUM + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    // 28 sums x 9 interleaved slices of the block list on 252 threads (one thread per sum walked up to 1024 partials
    // one load latency at a time: ~220 us of the finest level's 340 us), then the slices in slice order: still one
    // fixed summation order, independent of which block came last
    constexpr int SLICES = 9;
    __shared__ double slice[SLICES][ECC_NSUM];
    if (threadIdx.x < ECC_NSUM * SLICES) {
        const int sidx = threadIdx.x % ECC_NSUM, part = threadIdx.x / ECC_NSUM;
        double t = 0.0;
#pragma unroll 4
        for (unsigned bk = part; bk < gridDim.x; bk += SLICES)
            t += __builtin_nontemporal_load(&partial[(size_t)bk * ECC_NSUM + sidx]);
        slice[part][sidx] = t;
    }
    __syncthreads();
    __shared__ double tot28[ECC_NSUM];
    if (threadIdx.x < ECC_NSUM) {
        double t = 0.0;
#pragma unroll
        for (int part = 0; part < SLICES; ++part) t += slice[part][threadIdx.x];
        tot28[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // every block of this frame has read the parameters (they all drew their tickets): the step may overwrite them
        EccState fs = state[f];
        ecc_update(fs, tot28, reach, eps);
        state[f] = fs;
        *ticket = 0;
    }
}

// ================================================================================================
// ALIGN_HOMOGRAPHY (align.py:138-140 cv2.findHomography): 8-DoF refinement of the similarity the loop above finds.
// The projective terms of a focus stack are tiny and ill-conditioned on the coarse levels, so the homography is
// estimated at the finest level only, by the same forward-additive ECC iteration (cv2.findTransformECC's
// MOTION_HOMOGRAPHY form), starting from the converged similarity.  Parameters in NORMALISED centred coordinates
// (xn = (x - cx) / R, R = half the diagonal: every parameter is O(1)):
//     xn' = (h0 xn + h1 yn + h2) / (h6 xn + h7 yn + 1),   yn' = (h3 xn + h4 yn + h5) / (h6 xn + h7 yn + 1)
// and the sample position in the moving frame is (cx + R xn', cy + R yn').
constexpr int ECC_NSUM_H = 66;   // 6 scalars, SJ[8], SJiw[8], SJir[8], H[36]

struct EccStateH {
    double h[8];
    double rho, last_rho;
    int iters, failed, active, pad_;
};

// symmetric N x N solve as ecc_solve4 (column-scaled Gaussian elimination with partial pivoting, double)
template <int N>
__device__ inline bool ecc_solve_n(const double* Hs, const double* r, double* d) {
    double A[N][N + 1], sc[N];
    {
        double Hm[N][N];
        int k = 0;
        for (int i = 0; i < N; ++i)
            for (int j = i; j < N; ++j) Hm[i][j] = Hm[j][i] = Hs[k++];
        for (int i = 0; i < N; ++i) {
            if (!(Hm[i][i] > 0)) return false;
            sc[i] = 1.0 / sqrt(Hm[i][i]);
        }
        for (int i = 0; i < N; ++i) {
            for (int j = 0; j < N; ++j) A[i][j] = Hm[i][j] * sc[i] * sc[j];
            A[i][N] = r[i] * sc[i];
        }
    }
    for (int c = 0; c < N; ++c) {
        int piv = c;
        for (int i = c + 1; i < N; ++i)
            if (fabs(A[i][c]) > fabs(A[piv][c])) piv = i;
        if (fabs(A[piv][c]) < 1e-12) return false;
        if (piv != c)
            for (int j = 0; j <= N; ++j) { const double t = A[c][j]; A[c][j] = A[piv][j]; A[piv][j] = t; }
        for (int i = c + 1; i < N; ++i) {
            const double f = A[i][c] / A[c][c];
            for (int j = c; j <= N; ++j) A[i][j] -= f * A[c][j];
        }
    }
    for (int i = N - 1; i >= 0; --i) {
        double v = A[i][N];
        for (int j = i + 1; j < N; ++j) v -= A[i][j] * d[j];
        d[i] = v / A[i][i];
    }
    for (int i = 0; i < N; ++i) d[i] *= sc[i];
    return true;
}

// the forward-additive ECC step of ecc_update for 8 parameters; `R`: the normalisation radius in pixels
__device__ inline void ecc_update_h(EccStateH& f, const double* S, double R, double eps) {
    ++f.iters;
    const double cnt = S[0];
    if (cnt < 64) { f.failed = 1; f.active = 0; return; }
    const double mw = S[1] / cnt, mr = S[2] / cnt;
    const double wn2 = S[3] - cnt * mw * mw, rn2 = S[4] - cnt * mr * mr, corr = S[5] - cnt * mw * mr;
    if (!(wn2 > 0) || !(rn2 > 0)) { f.failed = 1; f.active = 0; return; }
    f.rho = corr / sqrt(wn2 * rn2);
    double ip[8], tp[8], Hi_ip[8];
    for (int q = 0; q < 8; ++q) {
        ip[q] = S[14 + q] - mw * S[6 + q];
        tp[q] = S[22 + q] - mr * S[6 + q];
    }
    if (!ecc_solve_n<8>(&S[30], ip, Hi_ip)) { f.active = 0; return; }
    double ipH = 0, tpH = 0;
    for (int q = 0; q < 8; ++q) { ipH += ip[q] * Hi_ip[q]; tpH += tp[q] * Hi_ip[q]; }
    const double lam_n = wn2 - ipH, lam_d = corr - tpH;
    if (!(lam_d > 0)) { f.active = 0; return; }
    const double lam = lam_n / lam_d;
    double ep[8], dp[8];
    for (int q = 0; q < 8; ++q) ep[q] = lam * tp[q] - ip[q];
    if (!ecc_solve_n<8>(&S[30], ep, dp)) { f.active = 0; return; }
    double move = 0.0;
    for (int q = 0; q < 8; ++q) { f.h[q] += dp[q]; move += fabs(dp[q]); }
    // every parameter moves a pixel of the unit disc by at most its own change: `move * R` bounds the displacement
    if (move * R < 2e-3 || fabs(f.rho - f.last_rho) < eps) f.active = 0;
    f.last_rho = f.rho;
}

__global__ __launch_bounds__(256) void ecc_accumulate_h(const float* __restrict__ tmpl, const float* __restrict__ img,
                                                        size_t fstride, int h, int w, EccStateH* __restrict__ state, int step,
                                                        double* __restrict__ partial, unsigned int* __restrict__ ticket, double eps) {
    const int f = blockIdx.y;
    if (!state[f].active) return;
    float p[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) p[q] = (float)state[f].h[q];
    img += (size_t)f * fstride;
    partial += (size_t)f * ECC_MAX_BLOCKS * ECC_NSUM_H;
    ticket += f;
    double acc[ECC_NSUM_H];
#pragma unroll
    for (int i = 0; i < ECC_NSUM_H; ++i) acc[i] = 0.0;
    const float cx = 0.5f * (w - 1), cy = 0.5f * (h - 1);
    const float R = hypotf(cx, cy), iR = 1.0f / R;
    const int nx = (w + step - 1) / step, ny = (h + step - 1) / step;
    const unsigned total = (unsigned)nx * ny;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int y = (int)(i / nx) * step, x = (int)(i % nx) * step;
        const float xn = (x - cx) * iR, yn = (y - cy) * iR;
        const float den = p[6] * xn + p[7] * yn + 1.0f;
        if (!(den > 1e-3f)) continue;
        const float id = 1.0f / den;
        const float xp = (p[0] * xn + p[1] * yn + p[2]) * id, yp = (p[3] * xn + p[4] * yn + p[5]) * id;
        const float u = cx + R * xp, v = cy + R * yp;
        const int x0 = (int)floorf(u), y0 = (int)floorf(v);
        if (x0 < 1 || y0 < 1 || x0 >= w - 2 || y0 >= h - 2) continue;
        const float fx = u - x0, fy = v - y0;
        const float* r1 = img + (size_t)y0 * w + x0;
        const float* r0 = r1 - w;
        const float* r2 = r1 + w;
        const float* r3 = r2 + w;
        const float a_m = r1[-1], a_0 = r1[0], a_1 = r1[1], a_2 = r1[2];
        const float b_m = r2[-1], b_0 = r2[0], b_1 = r2[1], b_2 = r2[2];
        const float t_0 = r0[0], t_1 = r0[1], u_0 = r3[0], u_1 = r3[1];
        const float iw = (a_0 + fx * (a_1 - a_0)) + fy * ((b_0 + fx * (b_1 - b_0)) - (a_0 + fx * (a_1 - a_0)));
        const float gxa0 = 0.5f * (a_1 - a_m), gxa1 = 0.5f * (a_2 - a_0), gxb0 = 0.5f * (b_1 - b_m), gxb1 = 0.5f * (b_2 - b_0);
        const float gya0 = 0.5f * (b_0 - t_0), gya1 = 0.5f * (b_1 - t_1), gyb0 = 0.5f * (u_0 - a_0), gyb1 = 0.5f * (u_1 - a_1);
        const float gxt = gxa0 + fx * (gxa1 - gxa0), gxb = gxb0 + fx * (gxb1 - gxb0);
        const float gyt = gya0 + fx * (gya1 - gya0), gyb = gyb0 + fx * (gyb1 - gyb0);
        // gradients with respect to the NORMALISED warped coordinates (a step of 1 in xn' is R pixels)
        const float dgx = (gxt + fy * (gxb - gxt)) * R, dgy = (gyt + fy * (gyb - gyt)) * R;
        const float ir = tmpl[(size_t)y * w + x];
        const float gxd = dgx * id, gyd = dgy * id, pr = -(dgx * xp + dgy * yp) * id;
        const float J[8] = {gxd * xn, gxd * yn, gxd, gyd * xn, gyd * yn, gyd, pr * xn, pr * yn};
        acc[0] += 1.0;
        acc[1] += iw;
        acc[2] += ir;
        acc[3] += (double)iw * iw;
        acc[4] += (double)ir * ir;
        acc[5] += (double)iw * ir;
        int hk = 30;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            acc[6 + k] += J[k];
            acc[14 + k] += (double)J[k] * iw;
            acc[22 + k] += (double)J[k] * ir;
#pragma unroll
            for (int m = k; m < 8; ++m) acc[hk++] += (double)J[k] * J[m];
        }
    }
    __shared__ double red[4][ECC_NSUM_H];
    __shared__ bool last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int s = 0; s < ECC_NSUM_H; ++s) {
        const double v = wave_sum(acc[s]);
        if (lane == 0) red[wave][s] = v;
    }
    __syncthreads();
    if (threadIdx.x < ECC_NSUM_H)
        partial[(size_t)blockIdx.x * ECC_NSUM_H + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    // fixed summation order (3 interleaved slices of the block list, then the slices in order)
    constexpr int SLICES = 3;
    __shared__ double slice[SLICES][ECC_NSUM_H];
    if (threadIdx.x < ECC_NSUM_H * SLICES) {
        const int sidx = threadIdx.x % ECC_NSUM_H, part = threadIdx.x / ECC_NSUM_H;
        double t = 0.0;
        for (unsigned bk = part; bk < gridDim.x; bk += SLICES)
            t += __builtin_nontemporal_load(&partial[(size_t)bk * ECC_NSUM_H + sidx]);
        slice[part][sidx] = t;
    }
    __syncthreads();
    __shared__ double tot[ECC_NSUM_H];
    if (threadIdx.x < ECC_NSUM_H) tot[threadIdx.x] = (slice[0][threadIdx.x] + slice[1][threadIdx.x]) + slice[2][threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        EccStateH fs = state[f];
        ecc_update_h(fs, tot, (double)R, eps);
        state[f] = fs;
        *ticket = 0;
    }
}

// the converged similarity of the finest level (centred a, b, tx, ty in pixels) as the homography's starting point
__global__ void ecc_h_from_similarity(const EccState* __restrict__ sim, EccStateH* __restrict__ hs, int n, double R) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const EccState& s = sim[k];
    EccStateH o{};
    o.h[0] = s.a; o.h[1] = -s.b; o.h[2] = s.tx / R;
    o.h[3] = s.b; o.h[4] = s.a;  o.h[5] = s.ty / R;
    o.h[6] = 0.0; o.h[7] = 0.0;
    o.rho = s.rho; o.last_rho = -2.0;
    o.iters = 0; o.failed = s.failed; o.active = !s.failed;
    hs[k] = o;
}

}  // namespace mi
