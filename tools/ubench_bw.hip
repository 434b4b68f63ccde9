// sustained HBM bandwidth of plain streaming kernels on this MI355X: read-only, write-only, copy
// (4 GiB >> L2 + Infinity Cache), for several grid sizes and access widths.  Context for the
// roofline fractions in DESIGN.md: "8 TB/s" is the pin rate, this is what a kernel can get.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void rd(const v4f* p, size_t n, float* sink) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    v4f a = {0, 0, 0, 0};
    for (; i + 3 * st < n; i += 4 * st) {
        v4f v0 = __builtin_nontemporal_load(p + i), v1 = __builtin_nontemporal_load(p + i + st);
        v4f v2 = __builtin_nontemporal_load(p + i + 2 * st), v3 = __builtin_nontemporal_load(p + i + 3 * st);
        a += v0 + v1 + v2 + v3;
    }
    for (; i < n; i += st) a += p[i];
    if (a.x + a.y + a.z + a.w == 12345.678f) *sink = a.x;
}
__global__ void wr(v4f* p, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) __builtin_nontemporal_store(v4f{1.f, 2.f, 3.f, 4.f}, p + i);
}
__global__ void cp(const v4f* s, v4f* d, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i);
}
int main() {
    const size_t bytes = 4ull << 30;
    void *a, *b; float* sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t n = bytes / 16;
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        float ms[3];
        for (int k = 0; k < 3; ++k) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (k == 0) rd<<<grid, 256>>>((const v4f*)a, n, sink);
                else if (k == 1) wr<<<grid, 256>>>((v4f*)b, n);
                else cp<<<grid, 256>>>((const v4f*)a, (v4f*)b, n);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[k], e0, e1);
            }
        }
        printf("grid %5d x 256: read %.2f TB/s   write %.2f TB/s   copy %.2f TB/s (read+write)\n", grid,
               bytes / ms[0] / 1e9, bytes / ms[1] / 1e9, 2.0 * bytes / ms[2] / 1e9);
    }
    return 0;
}
