// 16-byte-per-lane streaming reads at 16-, 8- and 4-byte alignment (the level kernels stage 12-byte pixels with 16-byte
// loads whose alignment follows the tile origin): does the alignment cost HBM bandwidth?
#include <hip/hip_runtime.h>
#include <cstdio>
struct __attribute__((packed, aligned(4))) U4 { float v[4]; };
__global__ void rd(const char* base, size_t n, float* sink) {
    const U4* p = (const U4*)base;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    float a = 0;
    for (; i + 3 * st < n; i += 4 * st) {
        U4 v0 = p[i], v1 = p[i + st], v2 = p[i + 2 * st], v3 = p[i + 3 * st];
        a += v0.v[0] + v0.v[3] + v1.v[1] + v1.v[2] + v2.v[0] + v2.v[3] + v3.v[1] + v3.v[2];
    }
    if (a == 12345.678f) *sink = a;
}
int main() {
    const size_t bytes = 4ull << 30;
    char* a; float* sink;
    hipMalloc(&a, bytes + 64); hipMalloc(&sink, 4);
    hipMemset(a, 1, bytes + 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t n = bytes / 16;
    for (int off : {0, 8, 4, 12}) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            rd<<<8192, 256>>>(a + off, n, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("offset %2d B: read %.2f TB/s\n", off, bytes / ms / 1e9);
    }
    return 0;
}
