// micro-benchmark: plain v_fma_f32 vs v_pk_fma_f32 issue rate on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#pragma clang fp contract(off)
template <int MODE>
__global__ void k(float* out, float a, float b, int iters) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    v2f pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb);
                p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb);
                p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb);
                p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = MODE == 0 ? x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 : p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
int main() {
    float* out; hipMalloc(&out, 256 * 4096 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4000;
    for (int mode = 0; mode < 2; ++mode) for (int waves = 1; waves <= 8; waves *= 2) {
        dim3 grid(256 * 4 * waves / 4), blk(256);   // `waves` waves per SIMD resident
        if (mode == 0) k<0><<<grid, blk>>>(out, 1.0001f, 0.5f, 10); else k<1><<<grid, blk>>>(out, 1.0001f, 0.5f, 10);
        hipDeviceSynchronize();
        hipEventRecord(a);
        if (mode == 0) k<0><<<grid, blk>>>(out, 1.0001f, 0.5f, iters); else k<1><<<grid, blk>>>(out, 1.0001f, 0.5f, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double fma = (double)grid.x * 256 * iters * 64.0 * (mode == 0 ? 1 : 2);
        printf("%s waves/SIMD=%d: %.3f ms  %.1f TFMA/s (%.1f TFLOP/s)\n", mode == 0 ? "v_fma_f32   " : "v_pk_fma_f32", waves, ms, fma / ms / 1e9, 2 * fma / ms / 1e9);
    }
    return 0;
}
