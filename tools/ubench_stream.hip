// micro-benchmark: register-streaming 25-tap stencil (wave-private, barrier-free).
// Each wave owns a strip of 64 lanes x 4 pixels and marches down the rows; horizontal neighbours
// come from the adjacent lanes by DPP wave shifts (4 v_mov_dpp per input row), vertical reuse by
// PARTIAL ACCUMULATORS: an arriving input row adds tap-row r to the output row that sees it as its
// r-th row, in arrival order, so every output is still the row-major fma chain from 0 of the
// reference-order 5x5 filter (bit-identical to the LDS-tiled kernel), with 5 live accumulators per
// pixel instead of a 5-row input window.
// Question answered: how close to the 60 Tfma/s plain-v_fma_f32 rate does hipcc get this way?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#pragma clang fp contract(off)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

struct K25 { float k[5][5]; };

__device__ __forceinline__ float from_prev_lane(float v) {   // lane i <- lane i-1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float from_next_lane(float v) {   // lane i <- lane i+1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

// out(y, x) = sum_{r,c} k[r][c] * in(clamp(y+r-2), clamp(x+c-2)), row-major fma chain from 0.
// grid.x = strips, grid.y = row segments of SEG rows; block = 64 (one wave).
template <int SEG, int REP, int PF>
__global__ __launch_bounds__(64) void blur_stream(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                  K25 kk, int seg) {
    const int lane = threadIdx.x;
    const int x0 = blockIdx.x * 248 - 4 + lane * 4;   // lanes 1..62 are useful: 248 px per strip
    const int y0 = blockIdx.y * SEG;
    in += (size_t)blockIdx.z * 0;  // frames of a batch share the input here
    // clamped column of each of this lane's 4 pixels (replicate border)
    int xc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) xc[p] = min(max(x0 + p, 0), W - 1);
    const bool vec = x0 >= 0 && x0 + 3 < W;
    float acc[REP][5][4];
#pragma unroll
    for (int q = 0; q < REP; ++q)
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[q][r][p] = 0.f;
    const bool store = lane >= 1 && lane <= 62 && x0 < W;
    // input rows y0-2 .. y0+SEG+1; output row o = yin - 2 completes when input row yin arrives
    auto load_row = [&](int i) -> f4 {
        const int yr = min(max(y0 - 2 + i, 0), H - 1);
        f4 v;
        if (vec) v = __builtin_nontemporal_load((const f4*)(in + (size_t)yr * W + x0));
        else { v.x = in[(size_t)yr * W + xc[0]]; v.y = in[(size_t)yr * W + xc[1]]; v.z = in[(size_t)yr * W + xc[2]]; v.w = in[(size_t)yr * W + xc[3]]; }
        return v;
    };
    f4 pre[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) pre[i] = load_row(i);
#pragma unroll 5
    for (int i = 0; i < seg + 4; ++i) {
        const int yin = y0 - 2 + i;
        f4 v = pre[0];
#pragma unroll
        for (int k = 0; k + 1 < PF; ++k) pre[k] = pre[k + 1];
        pre[PF - 1] = load_row(min(i + PF, seg + 3));
        float a[8];
        a[0] = from_prev_lane(v.z); a[1] = from_prev_lane(v.w);
        a[2] = v.x; a[3] = v.y; a[4] = v.z; a[5] = v.w;
        a[6] = from_next_lane(v.x); a[7] = from_next_lane(v.y);
        // this input row is tap-row r of the accumulator in slot r
#pragma unroll
        for (int q = 0; q < REP; ++q)
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int c = 0; c < 5; ++c) acc[q][r][p] = __builtin_fmaf(kk.k[(r + q) % 5][(c + q / 5) % 5], a[p + c], acc[q][r][p]);
        // slot 4 is complete: output row yin - 2
        const int yo = yin - 2;
        if (i >= 4 && yo < H && store) {
            float o[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                o[p] = acc[0][4][p];
#pragma unroll
                for (int q = 1; q < REP; ++q) o[p] = fmaxf(o[p], acc[q][4][p] - 1e30f);
            }
            if (x0 + 3 < W) *(f4*)(out + (size_t)yo * W + x0) = f4{o[0], o[1], o[2], o[3]};
            else
                for (int p = 0; p < 4; ++p)
                    if (x0 + p < W) out[(size_t)yo * W + x0 + p] = o[p];
        }
#pragma unroll
        for (int q = 0; q < REP; ++q) {
#pragma unroll
        for (int r = 4; r > 0; --r)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[q][r][p] = acc[q][r - 1][p];
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[q][0][p] = 0.f;
        }
    }
}

// out(y, x) = sum_{r,c} k[r][c] * in(clamp(y+r-2), clamp(x+c-2)), row-major fma chain from 0.
// grid.x = strips, grid.y = row segments of SEG rows; block = 64 (one wave).
template <int SEG, int REP, int PF>
__global__ __launch_bounds__(64) void blur_stream_pk(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                  K25 kk, int seg) {
    const int lane = threadIdx.x;
    const int x0 = blockIdx.x * 248 - 4 + lane * 4;   // lanes 1..62 are useful: 248 px per strip
    const int y0 = blockIdx.y * SEG;
    in += (size_t)blockIdx.z * 0;  // frames of a batch share the input here
    // clamped column of each of this lane's 4 pixels (replicate border)
    int xc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) xc[p] = min(max(x0 + p, 0), W - 1);
    const bool vec = x0 >= 0 && x0 + 3 < W;
    v2f acc[REP][5][2];
#pragma unroll
    for (int q = 0; q < REP; ++q)
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int p = 0; p < 2; ++p) acc[q][r][p] = v2f{0.f, 0.f};
    const bool store = lane >= 1 && lane <= 62 && x0 < W;
    // input rows y0-2 .. y0+SEG+1; output row o = yin - 2 completes when input row yin arrives
    auto load_row = [&](int i) -> f4 {
        const int yr = min(max(y0 - 2 + i, 0), H - 1);
        f4 v;
        if (vec) v = __builtin_nontemporal_load((const f4*)(in + (size_t)yr * W + x0));
        else { v.x = in[(size_t)yr * W + xc[0]]; v.y = in[(size_t)yr * W + xc[1]]; v.z = in[(size_t)yr * W + xc[2]]; v.w = in[(size_t)yr * W + xc[3]]; }
        return v;
    };
    f4 pre[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) pre[i] = load_row(i);
#pragma unroll 5
    for (int i = 0; i < seg + 4; ++i) {
        const int yin = y0 - 2 + i;
        f4 v = pre[0];
#pragma unroll
        for (int k = 0; k + 1 < PF; ++k) pre[k] = pre[k + 1];
        pre[PF - 1] = load_row(min(i + PF, seg + 3));
        float a[8];
        a[0] = from_prev_lane(v.z); a[1] = from_prev_lane(v.w);
        a[2] = v.x; a[3] = v.y; a[4] = v.z; a[5] = v.w;
        a[6] = from_next_lane(v.x); a[7] = from_next_lane(v.y);
        // this input row is tap-row r of the accumulator in slot r
        v2f A[4], B[3];
#pragma unroll
        for (int k = 0; k < 4; ++k) A[k] = v2f{a[2 * k], a[2 * k + 1]};
#pragma unroll
        for (int k = 0; k < 3; ++k) B[k] = v2f{a[2 * k + 1], a[2 * k + 2]};
#pragma unroll
        for (int q = 0; q < REP; ++q)
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int c = 0; c < 5; ++c) {
                    const float kc = kk.k[(r + q) % 5][(c + q / 5) % 5];
                    const int o = 2 * p + c;
                    acc[q][r][p] = __builtin_elementwise_fma(v2f{kc, kc}, (o & 1) ? B[o / 2] : A[o / 2], acc[q][r][p]);
                }
        // slot 4 is complete: output row yin - 2
        const int yo = yin - 2;
        if (i >= 4 && yo < H && store) {
            float o[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                o[p] = acc[0][4][p / 2][p % 2];
#pragma unroll
                for (int q = 1; q < REP; ++q) o[p] = fmaxf(o[p], acc[q][4][p / 2][p % 2] - 1e30f);
            }
            if (x0 + 3 < W) *(f4*)(out + (size_t)yo * W + x0) = f4{o[0], o[1], o[2], o[3]};
            else
                for (int p = 0; p < 4; ++p)
                    if (x0 + p < W) out[(size_t)yo * W + x0 + p] = o[p];
        }
#pragma unroll
        for (int q = 0; q < REP; ++q) {
#pragma unroll
        for (int r = 4; r > 0; --r)
#pragma unroll
            for (int p = 0; p < 2; ++p) acc[q][r][p] = acc[q][r - 1][p];
#pragma unroll
        for (int p = 0; p < 2; ++p) acc[q][0][p] = v2f{0.f, 0.f};
        }
    }
}

int main(int argc, char** argv) {
    const int H = 4000, W = 6000;
    std::vector<float> hin((size_t)H * W), hout((size_t)H * W);
    srand(1);
    for (auto& v : hin) v = (float)(rand() % 2048) / 8.f;
    K25 kk;
    const double k1[5] = {0.05, 0.25, 0.4, 0.25, 0.05};
    for (int r = 0; r < 5; ++r) for (int c = 0; c < 5; ++c) kk.k[r][c] = (float)(k1[r] * k1[c]);
    float *din, *dout;
    hipMalloc(&din, hin.size() * 4); hipMalloc(&dout, hin.size() * 4);
    hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice);
    hipMemset(dout, 0xff, hin.size() * 4);
    constexpr int SEG = 64;
    dim3 grid((W + 247) / 248, (H + SEG - 1) / SEG), blk(64);
    blur_stream<SEG, 1, 2><<<grid, blk>>>(din, dout, H, W, kk, SEG);
    hipDeviceSynchronize();
    hipMemcpy(hout.data(), dout, hin.size() * 4, hipMemcpyDeviceToHost);
    // bit-exact check against the row-major chain on a sample of pixels (incl. the borders)
    long bad = 0, checked = 0;
    for (int y = 0; y < H; y += 37) for (int x = 0; x < W; x += (y % 2 ? 1 : 41)) {
        float acc = 0.f;
        for (int r = 0; r < 5; ++r) for (int c = 0; c < 5; ++c) {
            int yy = std::min(std::max(y + r - 2, 0), H - 1), xx = std::min(std::max(x + c - 2, 0), W - 1);
            acc = fmaf(kk.k[r][c], hin[(size_t)yy * W + xx], acc);
        }
        ++checked;
        if (memcmp(&acc, &hout[(size_t)y * W + x], 4)) { if (bad < 5) printf("mismatch y=%d x=%d: %g vs %g\n", y, x, acc, hout[(size_t)y * W + x]); ++bad; }
    }
    printf("checked %ld pixels, %ld mismatches\n", checked, bad);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int reps = 5, NF = 16;
    grid.z = NF;
    const double px = (double)H * W;
#define RUN(REP, PF)                                                                                          \
    {                                                                                                     \
        blur_stream<SEG, REP, PF><<<grid, blk>>>(din, dout, H, W, kk, SEG);                                        \
        hipEventRecord(a);                                                                                \
        for (int i = 0; i < reps; ++i) blur_stream<SEG, REP, PF><<<grid, blk>>>(din, dout, H, W, kk, SEG);         \
        hipEventRecord(b); hipEventSynchronize(b);                                                        \
        float ms; hipEventElapsedTime(&ms, a, b); ms /= reps * NF;                                        \
        printf("blur_stream<SEG=%d, REP=%d, PF=%d>: %.4f ms per 24 MP  useful %.1f Tfma/s  issued %.1f Tfma/s  %.2f TB/s (8 B/px)\n", \
               SEG, REP, PF, ms, REP * 25 * px / ms / 1e9, REP * 25.0 * grid.x * grid.y * 256.0 * (SEG + 4) / ms / 1e9, 8 * px / ms / 1e9); \
    }
    RUN(1, 4) RUN(4, 4) RUN(6, 4) RUN(8, 4)
#undef RUN
#define RUN(REP, PF)                                                                                      \
    {                                                                                                     \
        blur_stream_pk<SEG, REP, PF><<<grid, blk>>>(din, dout, H, W, kk, SEG);                                 \
        hipEventRecord(a);                                                                                \
        for (int i = 0; i < reps; ++i) blur_stream_pk<SEG, REP, PF><<<grid, blk>>>(din, dout, H, W, kk, SEG);  \
        hipEventRecord(b); hipEventSynchronize(b);                                                        \
        float ms; hipEventElapsedTime(&ms, a, b); ms /= reps * NF;                                        \
        printf("blur_stream_pk<SEG=%d, REP=%d, PF=%d>: %.4f ms per 24 MP  useful %.1f Tfma/s  issued %.1f Tfma/s  %.2f TB/s (8 B/px)\n", \
               SEG, REP, PF, ms, REP * 25 * px / ms / 1e9, REP * 25.0 * grid.x * grid.y * 256.0 * (SEG + 4) / ms / 1e9, 8 * px / ms / 1e9); \
    }
    RUN(1, 4) RUN(2, 4) RUN(4, 4) RUN(4, 6) RUN(6, 4) RUN(8, 4)
    grid.z = 1;
    hipMemset(dout, 0xff, hin.size() * 4);
    blur_stream_pk<SEG, 1, 2><<<grid, blk>>>(din, dout, H, W, kk, SEG);
    hipDeviceSynchronize();
    std::vector<float> h2(hin.size());
    hipMemcpy(h2.data(), dout, hin.size() * 4, hipMemcpyDeviceToHost);
    printf("packed variant identical to plain: %s\n", memcmp(h2.data(), hout.data(), hin.size() * 4) ? "NO" : "yes");
    return bad != 0;
}
