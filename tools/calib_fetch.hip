// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the
// fused level kernel uses (8-byte global loads, 16-byte and 4-byte global stores):
// stream a known byte count (far larger than L2 + Infinity Cache) and compare.
//   hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o /tmp/calib_fetch
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- /tmp/calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void read8(const v2f* p, size_t n, float* sink) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    float a = 0;
    for (; i < n; i += st) { v2f v = p[i]; a += v.x + v.y; }
    if (a == 12345.678f) *sink = a;
}
__global__ void read16(const v4f* p, size_t n, float* sink) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    float a = 0;
    for (; i < n; i += st) { v4f v = p[i]; a += v.x + v.y + v.z + v.w; }
    if (a == 12345.678f) *sink = a;
}
__global__ void write16(v4f* p, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) p[i] = v4f{1.f, 2.f, 3.f, 4.f};
}
__global__ void write4(float* p, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) p[i] = 1.f;
}
int main() {
    const size_t bytes = 4ull << 30;  // 4 GiB >> 256 MiB Infinity Cache
    void* buf; float* sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 4);
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    read8<<<4096, 256>>>((const v2f*)buf, bytes / 8, sink);
    read16<<<4096, 256>>>((const v4f*)buf, bytes / 16, sink);
    write16<<<4096, 256>>>((v4f*)buf, bytes / 16);
    write4<<<4096, 256>>>((float*)buf, bytes / 4);
    hipDeviceSynchronize();
    printf("streamed %zu bytes per kernel\n", bytes);
    return 0;
}
