// Sustained HBM bandwidth of plain streaming kernels on this MI355X (4 GiB per array >> L2 + Infinity Cache): context for
// the roofline fractions in DESIGN.md -- "8 TB/s" is the pin rate, this is what a kernel can get.
//
// Round 5: the round-4 `cp` kept ONE 16-byte load in flight per thread and read 15-25 % below the
// guide's float4 copy (6.29 TB/s).  Every kernel here keeps U independent 16-byte loads in flight per thread (U = 1, 4, 8)
// before the first dependent instruction, and two kernels have level_sep<float>'s own access shape:
//   mix41    4 x b128 loads + 1 x b128 non-temporal store per step (80 / 20 read / write)
//   mix_b96  16 x b128 loads (256 B) + 4 x b96 non-temporal stores at a 12-byte stride (48 B): what a lane of level_sep
//            does per frame in the steady state -- 12 B of fp32 input per pixel (x 1.2 halo), 3 B of G_1 per pixel
// Each variant runs with grid-strided and with block-contiguous addressing, several grid sizes; best of 3 repetitions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v3u __attribute__((ext_vector_type(3)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// element i of thread t in step s: grid-strided (CONTIG 0: neighbouring workgroups touch neighbouring KBs at any time)
// or block-contiguous (CONTIG 1: a workgroup streams its own span)
template <int U, bool CONTIG>
__device__ __forceinline__ size_t idx0(size_t n, size_t& step) {
    if (CONTIG) {
        const size_t per = (n + gridDim.x - 1) / gridDim.x;
        step = blockDim.x;
        return (size_t)blockIdx.x * per + threadIdx.x;
    }
    step = (size_t)gridDim.x * blockDim.x;
    return (size_t)blockIdx.x * blockDim.x + threadIdx.x;
}

template <int U, bool CONTIG>
__global__ __launch_bounds__(256) void rd(const v4f* __restrict__ p, size_t n, float* sink) {
    size_t st, i = idx0<U, CONTIG>(n, st);
    const size_t end = CONTIG ? min(n, ((size_t)blockIdx.x + 1) * ((n + gridDim.x - 1) / gridDim.x)) : n;
    v4f a = {0, 0, 0, 0};
    for (; i + (U - 1) * st < end; i += U * st) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + i + u * st);
#pragma unroll
        for (int u = 0; u < U; ++u) a += v[u];
    }
    if (a.x + a.y + a.z + a.w == 12345.678f) *sink = a.x;
}
template <int U, bool CONTIG>
__global__ __launch_bounds__(256) void wr(v4f* __restrict__ p, size_t n) {
    size_t st, i = idx0<U, CONTIG>(n, st);
    const size_t end = CONTIG ? min(n, ((size_t)blockIdx.x + 1) * ((n + gridDim.x - 1) / gridDim.x)) : n;
    for (; i + (U - 1) * st < end; i += U * st) {
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v4f{1.f, 2.f, 3.f, 4.f}, p + i + u * st);
    }
}
template <int U, bool CONTIG, bool NT>
__global__ __launch_bounds__(256) void cp(const v4f* __restrict__ s, v4f* __restrict__ d, size_t n) {
    size_t st, i = idx0<U, CONTIG>(n, st);
    const size_t end = CONTIG ? min(n, ((size_t)blockIdx.x + 1) * ((n + gridDim.x - 1) / gridDim.x)) : n;
    for (; i + (U - 1) * st < end; i += U * st) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(s + i + u * st) : s[i + u * st];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], d + i + u * st);
            else d[i + u * st] = v[u];
        }
    }
}
// 4 loads : 1 store, all 16 bytes.  n counts the READ array's 16-byte elements; the written array has n / 4.
template <int U, bool CONTIG>
__global__ __launch_bounds__(256) void mix41(const v4f* __restrict__ s, v4f* __restrict__ d, size_t n) {
    // unit = 4 consecutive-in-step loads of one thread -> one store
    const size_t nu = n / 4;
    size_t st, i = idx0<U, CONTIG>(nu, st);
    const size_t end = CONTIG ? min(nu, ((size_t)blockIdx.x + 1) * ((nu + gridDim.x - 1) / gridDim.x)) : nu;
    for (; i + (U - 1) * st < end; i += U * st) {
        v4f v[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[u][k] = __builtin_nontemporal_load(s + (size_t)k * nu + i + u * st);
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u][0] + v[u][1] + v[u][2] + v[u][3], d + i + u * st);
    }
}
// level_sep<float>'s shape: 16 buffer b128 loads (4 "patch rows" of 4 chunks) in flight, then 4 b96 non-temporal buffer stores
// at a 12-byte stride.  Read bytes : written bytes = 256 : 48.
template <bool CONTIG>
__global__ __launch_bounds__(256) void mix_b96(const v4f* __restrict__ s, uint32_t* __restrict__ d, size_t n) {
    const size_t nu = n / 16;                    // units of 16 loads
    size_t st, i = idx0<1, CONTIG>(nu, st);
    const size_t end = CONTIG ? min(nu, ((size_t)blockIdx.x + 1) * ((nu + gridDim.x - 1) / gridDim.x)) : nu;
    for (; i < end; i += st) {
        v4f v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = __builtin_nontemporal_load(s + (size_t)k * nu + i);
        // four 12-byte "pixels" at consecutive 12-byte slots of the lane's 48-byte span
        const size_t base = i * 12;              // dwords
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const v4f a = v[4 * q] + v[4 * q + 1] + v[4 * q + 2] + v[4 * q + 3];
            v3u pv = {__builtin_bit_cast(uint32_t, a.x), __builtin_bit_cast(uint32_t, a.y), __builtin_bit_cast(uint32_t, a.z)};
            // lanes of a wave write consecutive 12-byte pixels of "row" q: coalesced like the G_1 store of level_sep
            uint32_t* row = d + (size_t)q * nu * 3 + (base / 12) * 3;
            __builtin_nontemporal_store(pv, (v3u*)row);
        }
    }
}

struct Arm { const char* name; double bytes; void (*launch)(int grid, const void* a, void* b, size_t n, float* sink); };

template <typename F> static float time_best(F f, hipEvent_t e0, hipEvent_t e1) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        f();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;   // first repetition warms up
    }
    return best;
}

int main(int argc, char** argv) {
    const size_t bytes = 4ull << 30;
    void *a, *b;
    float* sink;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, bytes));
    CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t n = bytes / 16;
    printf("# MI355X streaming bandwidth, 4 GiB arrays, TB/s of (read + written) bytes; best of 3 after warm-up\n");
    printf("# U = independent 16-byte loads in flight per thread; gs = grid-strided, bc = block-contiguous addressing\n");
    printf("%-22s", "kernel");
    const int grids[] = {1024, 2048, 4096, 8192, 16384, 65536};
    for (int g : grids) printf(" %8d", g);
    printf("   (workgroups x 256 threads)\n");
#define ROW(label, total_bytes, call)                                                          \
    do {                                                                                       \
        printf("%-22s", label);                                                                \
        for (int grid : grids) {                                                               \
            const float ms = time_best([&] { call; }, e0, e1);                                \
            printf(" %8.2f", (double)(total_bytes) / ms / 1e9);                                \
        }                                                                                      \
        printf("\n");                                                                          \
        fflush(stdout);                                                                        \
    } while (0)
    ROW("read  U1 gs", bytes, (rd<1, false><<<grid, 256>>>((const v4f*)a, n, sink)));
    ROW("read  U4 gs", bytes, (rd<4, false><<<grid, 256>>>((const v4f*)a, n, sink)));
    ROW("read  U8 gs", bytes, (rd<8, false><<<grid, 256>>>((const v4f*)a, n, sink)));
    ROW("read  U4 bc", bytes, (rd<4, true><<<grid, 256>>>((const v4f*)a, n, sink)));
    ROW("write U1 gs", bytes, (wr<1, false><<<grid, 256>>>((v4f*)b, n)));
    ROW("write U4 gs", bytes, (wr<4, false><<<grid, 256>>>((v4f*)b, n)));
    ROW("copy  U1 gs nt (r4)", 2.0 * bytes, (cp<1, false, true><<<grid, 256>>>((const v4f*)a, (v4f*)b, n)));
    ROW("copy  U4 gs nt", 2.0 * bytes, (cp<4, false, true><<<grid, 256>>>((const v4f*)a, (v4f*)b, n)));
    ROW("copy  U8 gs nt", 2.0 * bytes, (cp<8, false, true><<<grid, 256>>>((const v4f*)a, (v4f*)b, n)));
    ROW("copy  U4 gs plain", 2.0 * bytes, (cp<4, false, false><<<grid, 256>>>((const v4f*)a, (v4f*)b, n)));
    ROW("copy  U4 bc nt", 2.0 * bytes, (cp<4, true, true><<<grid, 256>>>((const v4f*)a, (v4f*)b, n)));
    ROW("mix 4:1 U1 gs", 1.25 * bytes, (mix41<1, false><<<grid, 256>>>((const v4f*)a, (v4f*)b, n)));
    ROW("mix 4:1 U2 gs", 1.25 * bytes, (mix41<2, false><<<grid, 256>>>((const v4f*)a, (v4f*)b, n)));
    ROW("mix 4:1 U4 gs", 1.25 * bytes, (mix41<4, false><<<grid, 256>>>((const v4f*)a, (v4f*)b, n)));
    ROW("mix 4:1 U2 bc", 1.25 * bytes, (mix41<2, true><<<grid, 256>>>((const v4f*)a, (v4f*)b, n)));
    ROW("mix 256:48 b96 gs", (1.0 + 48.0 / 256.0) * bytes, (mix_b96<false><<<grid, 256>>>((const v4f*)a, (uint32_t*)b, n)));
    ROW("mix 256:48 b96 bc", (1.0 + 48.0 / 256.0) * bytes, (mix_b96<true><<<grid, 256>>>((const v4f*)a, (uint32_t*)b, n)));
    CK(hipDeviceSynchronize());
    return 0;
}
