// common.hpp -- shared device/host helpers for libmi355stack (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>

#include "../../include/mi355stack.h"

// Every multiply-add in this library is either an explicit fma or two
// separately rounded operations -- never the compiler's choice -- because the
// results are compared bit-for-bit with the CPU oracle.
#pragma clang fp contract(off)

namespace mi {

struct K25 {
    float k[25];
};

// ---- thread-local error text ------------------------------------------------
inline std::string& last_error() {
    static thread_local std::string e;
    return e;
}
inline int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}

#define MI_HIP(call)                                                                      \
    do {                                                                                  \
        hipError_t _e = (call);                                                           \
        if (_e != hipSuccess)                                                             \
            return mi::fail(_e == hipErrorOutOfMemory ? MI_ERR_NOMEM : MI_ERR_HIP,        \
                            "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e),        \
                            __FILE__, __LINE__);                                          \
    } while (0)

// ---- device helpers -----------------------------------------------------------
// BORDER_REFLECT101 for an overshoot smaller than n (all pyramid stencils: |o| <= 2 < 4 <= n)
__device__ __forceinline__ int r101(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}
// any overshoot (base-level window with a large kernel_size on a tiny base)
__device__ __forceinline__ int r101_loop(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * n - 2 - i;
    }
    return i;
}

template <bool FMA>
__device__ __forceinline__ float mac(float k, float x, float s) {
    if constexpr (FMA) {
        return __builtin_fmaf(k, x, s);
    } else {
        float p = k * x;  // contraction is off for this TU
        return s + p;
    }
}

// cv2.cvtColor(BGR2GRAY) on float32 [from memory: fma(R,cr, fma(G,cg, B*cb))]
template <bool FMA>
__device__ __forceinline__ float gray_of(float b, float g, float r) {
    const float cb = 0.114f, cg = 0.587f, cr = 0.299f;
    if constexpr (FMA) {
        return __builtin_fmaf(r, cr, __builtin_fmaf(g, cg, b * cb));
    } else {
        float pb = b * cb, pg = g * cg, pr = r * cr;
        float s = pb + pg;
        return s + pr;
    }
}

template <typename T>
__device__ __forceinline__ float to_f32(T v) {
    return (float)v;
}

__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// MI_ARITH_SEPARABLE, taps of the reduce (kernels_sep.hpp header, oracle/oracle.py::red_taps_f32 is the same rule): when 20 k is
// integral for the generating kernel k = [1/4 - a/2, 1/4, a, 1/4, 1/4 - a/2] (a = 0.4, the reference's default: 1 5 8 5 1) the
// taps are those integers and the result is scaled once by float32(1/400); else float32(k) and scale 1.  Returns whether
// the integer (MFMA) form of the level-0 reduce applies: non-negative integer taps whose products fit a signed byte.
inline bool red_taps(double a, float rk[4]) {
    const double k0 = 0.25 - a / 2.0, k1 = 0.25, k2 = a;
    const double w0 = 20.0 * k0, w1 = 20.0 * k1, w2 = 20.0 * k2;
    const double r0 = (double)(long long)(w0 + (w0 < 0 ? -0.5 : 0.5)), r2 = (double)(long long)(w2 + (w2 < 0 ? -0.5 : 0.5));
    // integral, and small enough for the partial sums w0 (a + e) + w2 c of 16-bit input to stay exact:
    // (2 |w0| + |w2|) * 65535 < 2^24 -- only the last fma (w1 = 5) may round then (kernels_sep.hpp header).  A negative outer
    // tap (a > 0.5: a = 0.7 gives -2 5 14) is fine; oracle.red_taps_f32 is the same rule (np.round against this
    // round-half-away: they differ on exact halves only, which are not integral).
    const double a0 = r0 < 0 ? -r0 : r0, a2 = r2 < 0 ? -r2 : r2;
    const bool integral = (w0 - r0 < 1e-9 && r0 - w0 < 1e-9) && (w2 - r2 < 1e-9 && r2 - w2 < 1e-9) && w1 == 5.0 &&
                          2.0 * a0 + a2 <= 255.0;
    if (!integral) {
        rk[0] = (float)k0; rk[1] = (float)k1; rk[2] = (float)k2; rk[3] = 1.0f;
        return false;
    }
    rk[0] = (float)r0; rk[1] = 5.0f; rk[2] = (float)r2; rk[3] = (float)(1.0 / 400.0);
    const double m = r2 > r0 ? r2 : r0;
    return r0 >= 0.0 && r2 >= 0.0 && m * (m > 5.0 ? m : 5.0) <= 127.0;
}

// Timing-study knobs (phase ablation, tile / batch variants) exist only in -DMI_STUDY builds (tools/study_build.sh
// -> libmi355stack_study.so); the release library reads no environment variable on its compute paths.
#ifdef MI_STUDY
inline int study_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#define MI_ABL(bits) ((a.ablate & (bits)) != 0)
#else
inline int study_env(const char*, int dflt) { return dflt; }
#define MI_ABL(bits) (false)
#endif

}  // namespace mi
