// capi.hip -- C ABI of libmi355stack.so (declared in include/mi355stack.h).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC
#include <stdarg.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "common.hpp"
#include "kernels_simple.hpp"
#include "kernels_tiled.hpp"
#include "kernels_sep.hpp"
#include "kernels_align.hpp"
#include "kernels_ecc.hpp"
#include "kernels_phase.hpp"
#include "kernels_balance.hpp"
#include "kernels_f64.hpp"
#include "kernels_steps.hpp"

using namespace mi;

struct mi_stack;
namespace mi {
// LDS-tiled production path + host-frame staging (tiled_host.hpp)
bool tiled_available();
int tiled_create(mi_stack* s);
void tiled_destroy(mi_stack* s);
int tiled_reset(mi_stack* s);
int tiled_pending(const mi_stack* s);
void tiled_side_streams(const mi_stack* s, hipStream_t out[2]);
int tiled_push(mi_stack* s, const void* dev_frames, int n, size_t stride);
int tiled_push_host(mi_stack* s, const void* host_bgr, size_t row_stride_bytes, bool pinned = false);
int tiled_wait_uploads(mi_stack* s, int max_outstanding);
int tiled_flush(mi_stack* s);
int tiled_sync_all(mi_stack* s);
int tiled_sync_level0(mi_stack* s);
const float* tiled_last_gauss(mi_stack* s, int level);
int dispatch_push(mi_stack* s, const void* dev_frames, int n, size_t stride);
}  // namespace mi

namespace {

size_t dtype_size(int dt) {
    switch (dt) {
        case MI_U8: return 1;
        case MI_U16: return 2;
        case MI_F32: return 4;
        case MI_F64: return 8;
    }
    return 0;
}

struct ProfRec {
    int kind;
    hipEvent_t a, b;
    double bytes;
    int launches = 1;   // kernel launches between the two events (consecutive launches of one level share a pair)
};

}  // namespace

struct mi_stack {
    mi_stack_params_t p{};
    int L = 0;                // number of Laplacian levels; base is level L
    std::vector<int> lh, lw;  // level shapes, 0..L
    K25 K{};
    float k1d[3] = {0.f, 0.f, 0.f};   // MI_ARITH_SEPARABLE: float32 of the 1-D generating kernel (k0, k1, k2)
    float rk[4] = {0.f, 0.f, 0.f, 1.f};   // MI_ARITH_SEPARABLE, reduce: taps (w0, w1, w2) and final scale (red_taps)
    bool mfma_ok = false;             // integer taps small enough for the MFMA form of the level-0 reduce
    bool sep = false;         // p.arith == MI_ARITH_SEPARABLE
    K25d Kd{};                // float-64 mode: the float64 generating kernel (np.outer, pyramid.py:21)
    bool f64 = false;         // float_type == MI_F64: float buffers below hold doubles (allocated twice as large)
    int pad = 2;
    int nlevels_hist = 256;
    float maxv = 255.f;
    hipStream_t stream = nullptr;

    void* frame_dev = nullptr;  // staging for host-pushed frames (in_dtype)
    std::vector<float*> G;      // G[l], l = 1..L, of the frame being processed
    float* lap_tmp = nullptr;   // simple impl scratch
    float* q_tmp = nullptr;
    std::vector<float*> bestE, bestLap;
    std::vector<int32_t*> bestIdx;
    // float-32 stacks: the selection state of all levels and of the two base features lives in three
    // contiguous slabs (segments padded to 64 pixels), so the cross-GPU combine sees one flat vector
    float *slabE = nullptr, *slabL = nullptr;
    int32_t* slabI = nullptr;
    size_t slab_px = 0;

    int32_t* lev = nullptr;
    uint32_t* cnt = nullptr;
    float* logp = nullptr;
    float *bEnt = nullptr, *bDev = nullptr, *baseE = nullptr, *baseD = nullptr;
    int32_t *idxE = nullptr, *idxD = nullptr;
    float* fusedBase = nullptr;

    float *colA = nullptr, *colB = nullptr, *clipped = nullptr;
    const float* collapse_src = nullptr;   // float-32 stacks: input of the finest collapse step (set by finish)
    bool have_clipped = true;              // `clipped` holds the collapsed image (else the tap rebuilds it)
    void* out_dev = nullptr;

    int n_pushed = 0;
    int first_index = 0;
    int index_stride = 1;          // global index of the k-th pushed frame = first_index + k * index_stride (interleaved shards)
    uint64_t idx_exported = 0;     // bit l: the indices of state level l (L, L + 1: the base twins) are in their global form
    hipStream_t aux = nullptr;     // stream of the index export (not ordered behind the side streams as s->stream is after a push)
    bool finished = false;

    bool prof = false;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> ev_pool;
    double prof_ms[MI_PROF_KINDS] = {};
    int64_t prof_n[MI_PROF_KINDS] = {};
    double prof_bytes[MI_PROF_KINDS] = {};

    std::vector<void*> allocs;
    void* tiled = nullptr;  // TiledState (tiled_host.hpp)
};

namespace {

int dev_alloc(mi_stack* s, void** p, size_t bytes) {
    MI_HIP(hipMalloc(p, bytes ? bytes : 1));
    s->allocs.push_back(*p);
    return MI_OK;
}
template <typename T>
int dev_alloc_t(mi_stack* s, T** p, size_t count) {
    return dev_alloc(s, (void**)p, count * sizeof(T));
}

hipEvent_t get_event(mi_stack* s) {
    if (!s->ev_pool.empty()) {
        hipEvent_t e = s->ev_pool.back();
        s->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

struct ProfScope {
    mi_stack* s;
    ProfRec r{};
    bool on;
    hipStream_t st;
    ProfScope(mi_stack* s_, int kind, double bytes, hipStream_t stream = nullptr)
        : s(s_), on(s_->prof), st(stream ? stream : s_->stream) {
        if (!on) return;
        r.kind = kind;
        r.bytes = bytes;
        r.a = get_event(s);
        r.b = get_event(s);
        if (!r.a || !r.b) { on = false; return; }
        (void)hipEventRecord(r.a, st);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(r.b, st);
        s->recs.push_back(r);
    }
};

int prof_drain(mi_stack* s) {
    if (s->recs.empty()) return MI_OK;
    int rc0 = tiled_sync_all(s);
    if (rc0) return rc0;
    MI_HIP(hipStreamSynchronize(s->stream));
    for (auto& r : s->recs) {
        float ms = 0.f;
        MI_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        s->prof_ms[r.kind] += ms;
        s->prof_n[r.kind] += r.launches;
        s->prof_bytes[r.kind] += r.bytes;
        s->ev_pool.push_back(r.a);
        s->ev_pool.push_back(r.b);
    }
    s->recs.clear();
    return MI_OK;
}

// algorithmic bytes of one frame's pyramid build + select:
// read level 0 once, write each G_l (l>=1) once and read it twice.
double algorithmic_bytes_per_frame(const mi_stack* s) {
    double b = (double)dtype_size(s->p.in_dtype) * 3.0 * s->lh[0] * s->lw[0];
    for (int l = 1; l <= s->L; ++l) b += 36.0 * s->lh[l] * s->lw[l];
    return b;
}

inline dim3 grid2d(int w, int h, dim3 blk) { return dim3(cdiv(w, blk.x), cdiv(h, blk.y)); }

// ------------------------------------------------------------------ simple implementation
template <typename TIn, bool FMA>
int process_frame_simple(mi_stack* s, const TIn* frame) {
    const dim3 blk(64, 4);
    const int idx = s->first_index + s->n_pushed;
    const int first = s->n_pushed == 0;
    if (s->sep) {   // MI_ARITH_SEPARABLE: the same sequence with the two-pass kernels (kernels_sep.hpp)
        ProfScope ps(s, MI_PROF_LEVEL, algorithmic_bytes_per_frame(s));
        const float k0 = s->k1d[0], k1 = s->k1d[1], k2 = s->k1d[2];
        hipLaunchKernelGGL((reduce_sep_simple<TIn>), grid2d(s->lw[1], s->lh[1], blk), blk, 0, s->stream, frame, s->lh[0],
                           s->lw[0], s->G[1], s->lh[1], s->lw[1], s->rk[0], s->rk[1], s->rk[2], s->rk[3]);
        for (int l = 1; l < s->L; ++l)
            hipLaunchKernelGGL((reduce_sep_simple<float>), grid2d(s->lw[l + 1], s->lh[l + 1], blk), blk, 0, s->stream,
                               (const float*)s->G[l], s->lh[l], s->lw[l], s->G[l + 1], s->lh[l + 1], s->lw[l + 1], s->rk[0],
                               s->rk[1], s->rk[2], s->rk[3]);
        for (int l = 0; l < s->L; ++l) {
            dim3 g = grid2d(s->lw[l], s->lh[l], blk);
            if (l == 0)
                hipLaunchKernelGGL((lapq_sep_simple<TIn>), g, blk, 0, s->stream, frame, s->lh[0], s->lw[0], (const float*)s->G[1],
                                   s->lh[1], s->lw[1], s->lap_tmp, s->q_tmp, k0, k1, k2);
            else
                hipLaunchKernelGGL((lapq_sep_simple<float>), g, blk, 0, s->stream, (const float*)s->G[l], s->lh[l], s->lw[l],
                                   (const float*)s->G[l + 1], s->lh[l + 1], s->lw[l + 1], s->lap_tmp, s->q_tmp, k0, k1, k2);
            hipLaunchKernelGGL(select_sep_simple, g, blk, 0, s->stream, (const float*)s->q_tmp, (const float*)s->lap_tmp,
                               s->lh[l], s->lw[l], idx, first, s->bestE[l], s->bestLap[l], s->bestIdx[l], k0, k1, k2);
        }
        MI_HIP(hipGetLastError());
        return MI_OK;
    }
    {
        ProfScope ps(s, MI_PROF_LEVEL, algorithmic_bytes_per_frame(s));
        // Gaussian pyramid
        hipLaunchKernelGGL((reduce_simple<TIn, FMA>), grid2d(s->lw[1], s->lh[1], blk), blk, 0,
                           s->stream, frame, s->lh[0], s->lw[0], s->G[1], s->lh[1], s->lw[1],
                           s->K);
        for (int l = 1; l < s->L; ++l)
            hipLaunchKernelGGL((reduce_simple<float, FMA>),
                               grid2d(s->lw[l + 1], s->lh[l + 1], blk), blk, 0, s->stream,
                               s->G[l], s->lh[l], s->lw[l], s->G[l + 1], s->lh[l + 1],
                               s->lw[l + 1], s->K);
        // Laplacian levels + selection
        for (int l = 0; l < s->L; ++l) {
            dim3 g = grid2d(s->lw[l], s->lh[l], blk);
            if (l == 0)
                hipLaunchKernelGGL((lapq_simple<TIn, FMA>), g, blk, 0, s->stream, frame,
                                   s->lh[0], s->lw[0], s->G[1], s->lh[1], s->lw[1], s->lap_tmp,
                                   s->q_tmp, s->K);
            else
                hipLaunchKernelGGL((lapq_simple<float, FMA>), g, blk, 0, s->stream, s->G[l],
                                   s->lh[l], s->lw[l], s->G[l + 1], s->lh[l + 1], s->lw[l + 1],
                                   s->lap_tmp, s->q_tmp, s->K);
            hipLaunchKernelGGL((select_simple<FMA>), g, blk, 0, s->stream, s->q_tmp, s->lap_tmp,
                               s->lh[l], s->lw[l], idx, first, s->bestE[l], s->bestLap[l],
                               s->bestIdx[l], s->K);
        }
    }
    MI_HIP(hipGetLastError());
    return MI_OK;
}

// base-level features of the frame whose G_L sits in s->G[L] (or `base` for L == 0)
template <bool FMA>
int process_base(mi_stack* s, const float* base) {
    ProfScope ps(s, MI_PROF_BASE, 0.0);
    const int hb = s->lh[s->L], wb = s->lw[s->L], npix = hb * wb;
    const int idx = s->first_index + s->n_pushed;
    const int first = s->n_pushed == 0;
    MI_HIP(hipMemsetAsync(s->cnt, 0, sizeof(uint32_t) * s->nlevels_hist, s->stream));
    hipLaunchKernelGGL((base_gray_hist<FMA>), dim3(cdiv(npix, 256)), dim3(256), 0, s->stream, base,
                       npix, s->nlevels_hist, s->lev, s->cnt);
    hipLaunchKernelGGL(base_logp, dim3(cdiv(s->nlevels_hist, 256)), dim3(256), 0, s->stream,
                       s->cnt, s->nlevels_hist, npix, s->logp);
    const dim3 blk(32, 8);
    hipLaunchKernelGGL(base_feat_select, grid2d(wb, hb, blk), blk, 0, s->stream, s->lev, s->logp,
                       base, hb, wb, s->pad, idx, first, s->bEnt, s->bDev, s->idxE, s->idxD,
                       s->baseE, s->baseD);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

// ------------------------------------------------------------------ float_type = float-64
template <typename TIn, bool FMA>
int process_frame_f64(mi_stack* s, const TIn* frame) {
    const dim3 blk(64, 4);
    const int idx = s->first_index + s->n_pushed;
    const int first = s->n_pushed == 0;
    auto G = [&](int l) { return reinterpret_cast<double*>(s->G[l]); };
    double* lap = reinterpret_cast<double*>(s->lap_tmp);
    {
        ProfScope ps(s, MI_PROF_LEVEL, algorithmic_bytes_per_frame(s));
        hipLaunchKernelGGL((reduce_f64<TIn, FMA>), grid2d(s->lw[1], s->lh[1], blk), blk, 0, s->stream, frame,
                           s->lh[0], s->lw[0], G(1), s->lh[1], s->lw[1], s->Kd);
        for (int l = 1; l < s->L; ++l)
            hipLaunchKernelGGL((reduce_f64<double, FMA>), grid2d(s->lw[l + 1], s->lh[l + 1], blk), blk, 0, s->stream,
                               (const double*)G(l), s->lh[l], s->lw[l], G(l + 1), s->lh[l + 1], s->lw[l + 1], s->Kd);
        for (int l = 0; l < s->L; ++l) {
            dim3 g = grid2d(s->lw[l], s->lh[l], blk);
            if (l == 0)
                hipLaunchKernelGGL((lapq_f64<TIn, FMA>), g, blk, 0, s->stream, frame, s->lh[0], s->lw[0],
                                   (const double*)G(1), s->lh[1], s->lw[1], lap, s->q_tmp, s->Kd);
            else
                hipLaunchKernelGGL((lapq_f64<double, FMA>), g, blk, 0, s->stream, (const double*)G(l), s->lh[l],
                                   s->lw[l], (const double*)G(l + 1), s->lh[l + 1], s->lw[l + 1], lap, s->q_tmp, s->Kd);
            hipLaunchKernelGGL((select_f64<FMA>), g, blk, 0, s->stream, s->q_tmp, (const double*)lap, s->lh[l],
                               s->lw[l], idx, first, s->bestE[l], reinterpret_cast<double*>(s->bestLap[l]),
                               s->bestIdx[l], s->K);
        }
    }
    MI_HIP(hipGetLastError());
    return MI_OK;
}

// base features in float64; the 256 / 65536-entry table log(count / npix) is filled on the host
// (logl rounded to double, see kernels_f64.hpp)
template <bool FMA>
int process_base_f64(mi_stack* s, const double* base) {
    ProfScope ps(s, MI_PROF_BASE, 0.0);
    const int hb = s->lh[s->L], wb = s->lw[s->L], npix = hb * wb;
    const int idx = s->first_index + s->n_pushed;
    const int first = s->n_pushed == 0;
    MI_HIP(hipMemsetAsync(s->cnt, 0, sizeof(uint32_t) * s->nlevels_hist, s->stream));
    hipLaunchKernelGGL((base_gray_hist_f64<FMA>), dim3(cdiv(npix, 256)), dim3(256), 0, s->stream, base, npix,
                       s->nlevels_hist, s->lev, s->cnt);
    std::vector<uint32_t> cnt(s->nlevels_hist);
    std::vector<double> lp(s->nlevels_hist, 0.0);
    MI_HIP(hipMemcpyAsync(cnt.data(), s->cnt, cnt.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    MI_HIP(hipStreamSynchronize(s->stream));
    for (int l = 0; l < s->nlevels_hist; ++l)
        if (cnt[l]) {
            const double pr = (double)cnt[l] / (double)npix;   // counts.astype(float64) / counts.sum()
            lp[l] = (double)logl((long double)pr);
        }
    MI_HIP(hipMemcpyAsync(s->logp, lp.data(), lp.size() * sizeof(double), hipMemcpyHostToDevice, s->stream));
    MI_HIP(hipStreamSynchronize(s->stream));   // lp goes out of scope
    const dim3 blk(32, 8);
    hipLaunchKernelGGL(base_feat_select_f64, grid2d(wb, hb, blk), blk, 0, s->stream, s->lev,
                       (const double*)s->logp, base, hb, wb, s->pad, idx, first, reinterpret_cast<double*>(s->bEnt),
                       reinterpret_cast<double*>(s->bDev), s->idxE, s->idxD, reinterpret_cast<double*>(s->baseE),
                       reinterpret_cast<double*>(s->baseD));
    MI_HIP(hipGetLastError());
    return MI_OK;
}

template <typename TIn, bool FMA>
int push_device_frames_f64(mi_stack* s, const void* dev_frames, int n, size_t stride) {
    for (int f = 0; f < n; ++f) {
        const TIn* fr = (const TIn*)((const char*)dev_frames + (size_t)f * stride);
        int rc;
        if (s->L == 0) {
            const size_t cnt = (size_t)s->lh[0] * s->lw[0] * 3;
            hipLaunchKernelGGL((frame_to_f64<TIn>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s->stream, fr,
                               cnt, reinterpret_cast<double*>(s->colA));
            rc = process_base_f64<FMA>(s, reinterpret_cast<const double*>(s->colA));
        } else {
            rc = process_frame_f64<TIn, FMA>(s, fr);
            if (rc) return rc;
            rc = process_base_f64<FMA>(s, reinterpret_cast<const double*>(s->G[s->L]));
        }
        if (rc) return rc;
        s->n_pushed++;
    }
    return MI_OK;
}

template <bool FMA>
int finish_f64(mi_stack* s) {
    ProfScope ps(s, MI_PROF_COLLAPSE, 0.0);
    const int L = s->L;
    const size_t nb = (size_t)s->lh[L] * s->lw[L] * 3;
    double* fused = reinterpret_cast<double*>(s->fusedBase);
    hipLaunchKernelGGL(base_fuse_f64, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s->stream,
                       (const double*)s->baseE, (const double*)s->baseD, nb, fused);
    const double* up = fused;
    double* bufs[2] = {reinterpret_cast<double*>(s->colA), reinterpret_cast<double*>(s->colB)};
    const dim3 blk(64, 4);
    for (int l = L - 1; l >= 0; --l) {
        double* out = bufs[l & 1];
        hipLaunchKernelGGL((collapse_f64<FMA>), grid2d(s->lw[l], s->lh[l], blk), blk, 0, s->stream, up, s->lh[l + 1],
                           s->lw[l + 1], (const double*)s->bestLap[l], s->lh[l], s->lw[l], out, s->Kd);
        up = out;
    }
    const size_t n = (size_t)s->lh[0] * s->lw[0] * 3;
    const dim3 g((unsigned)((n + 255) / 256));
    double* clipped = reinterpret_cast<double*>(s->clipped);
    if (s->p.out_dtype == MI_U8)
        hipLaunchKernelGGL((finalize_cast_f64<uint8_t>), g, dim3(256), 0, s->stream, up, n, (double)s->maxv, clipped,
                           (uint8_t*)s->out_dev);
    else
        hipLaunchKernelGGL((finalize_cast_f64<uint16_t>), g, dim3(256), 0, s->stream, up, n, (double)s->maxv, clipped,
                           (uint16_t*)s->out_dev);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

template <typename TIn, bool FMA>
int push_device_frames(mi_stack* s, const void* dev_frames, int n, size_t stride) {
    for (int f = 0; f < n; ++f) {
        const TIn* fr = (const TIn*)((const char*)dev_frames + (size_t)f * stride);
        int rc;
        if (s->L == 0) {
            // frames smaller than 2*min_size: no Laplacian levels, the frame itself is the base
            // (colA is free until the collapse)
            const size_t n = (size_t)s->lh[0] * s->lw[0] * 3;
            hipLaunchKernelGGL((frame_to_f32<TIn>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream,
                               fr, n, s->colA);
            rc = process_base<FMA>(s, s->colA);
            if (rc) return rc;
            s->n_pushed++;
            continue;
        }
        rc = process_frame_simple<TIn, FMA>(s, fr);
        if (rc) return rc;
        rc = process_base<FMA>(s, s->G[s->L]);
        if (rc) return rc;
        s->n_pushed++;
    }
    return MI_OK;
}

}  // namespace
int mi::dispatch_push(mi_stack* s, const void* dev_frames, int n, size_t stride) {
    const bool fma = s->p.use_fma != 0;
    if (s->p.impl == MI_IMPL_TILED) return tiled_push(s, dev_frames, n, stride);
    if (s->f64) {
        switch (s->p.in_dtype) {
            case MI_U8:
                return fma ? push_device_frames_f64<uint8_t, true>(s, dev_frames, n, stride)
                           : push_device_frames_f64<uint8_t, false>(s, dev_frames, n, stride);
            case MI_U16:
                return fma ? push_device_frames_f64<uint16_t, true>(s, dev_frames, n, stride)
                           : push_device_frames_f64<uint16_t, false>(s, dev_frames, n, stride);
            case MI_F32:
                return fma ? push_device_frames_f64<float, true>(s, dev_frames, n, stride)
                           : push_device_frames_f64<float, false>(s, dev_frames, n, stride);
        }
        return fail(MI_ERR_INVALID, "bad in_dtype %d", s->p.in_dtype);
    }
    switch (s->p.in_dtype) {
        case MI_U8:
            return fma ? push_device_frames<uint8_t, true>(s, dev_frames, n, stride)
                       : push_device_frames<uint8_t, false>(s, dev_frames, n, stride);
        case MI_U16:
            return fma ? push_device_frames<uint16_t, true>(s, dev_frames, n, stride)
                       : push_device_frames<uint16_t, false>(s, dev_frames, n, stride);
        case MI_F32:
            return fma ? push_device_frames<float, true>(s, dev_frames, n, stride)
                       : push_device_frames<float, false>(s, dev_frames, n, stride);
    }
    return fail(MI_ERR_INVALID, "bad in_dtype %d", s->p.in_dtype);
}
namespace {

template <bool FMA>
int finish_impl(mi_stack* s) {
    ProfScope ps(s, MI_PROF_COLLAPSE, 0.0);
    const int L = s->L;
    size_t nb = (size_t)s->lh[L] * s->lw[L] * 3;
    hipLaunchKernelGGL(base_fuse, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s->stream,
                       s->baseE, s->baseD, nb, s->fusedBase);
    const float* up = s->fusedBase;
    float* bufs[2] = {s->colA, s->colB};
    const dim3 blk(64, 4);
    if (s->sep && L >= 1) {   // MI_ARITH_SEPARABLE: two-pass expand (kernels_sep.hpp), same structure
        const float k0 = s->k1d[0], k1 = s->k1d[1], k2 = s->k1d[2];
        for (int l = L - 1; l >= 1; --l) {
            float* out = bufs[l & 1];
            hipLaunchKernelGGL((collapse_sep<float>), grid2d(cdiv(s->lw[l], 2), cdiv(s->lh[l], 2), blk), blk, 0, s->stream, up, s->lh[l + 1],
                               s->lw[l + 1], (const float*)s->bestLap[l], s->lh[l], s->lw[l], s->maxv, out, k0, k1, k2);
            up = out;
        }
        s->collapse_src = up;
        s->have_clipped = false;
        const dim3 g0 = grid2d(cdiv(s->lw[0], 2), cdiv(s->lh[0], 2), blk);
        if (s->p.out_dtype == MI_U8)
            hipLaunchKernelGGL((collapse_sep<uint8_t>), g0, blk, 0, s->stream, up, s->lh[1], s->lw[1],
                               (const float*)s->bestLap[0], s->lh[0], s->lw[0], s->maxv, (uint8_t*)s->out_dev, k0, k1, k2);
        else
            hipLaunchKernelGGL((collapse_sep<uint16_t>), g0, blk, 0, s->stream, up, s->lh[1], s->lw[1],
                               (const float*)s->bestLap[0], s->lh[0], s->lw[0], s->maxv, (uint16_t*)s->out_dev, k0, k1, k2);
        MI_HIP(hipGetLastError());
        return MI_OK;
    }
    for (int l = L - 1; l >= 1; --l) {
        float* out = bufs[l & 1];
        hipLaunchKernelGGL((collapse_exact_quad<FMA, float>), grid2d(cdiv(s->lw[l], 2), cdiv(s->lh[l], 2), blk), blk, 0,
                           s->stream, up, s->lh[l + 1], s->lw[l + 1], (const float*)s->bestLap[l], s->lh[l],
                           s->lw[l], s->maxv, out, s->K);
        up = out;
    }
    s->collapse_src = up;     // level-1 image (or the fused base): what the collapsed-image tap restarts from
    s->have_clipped = false;
    if (L >= 1) {             // finest step fused with clip(abs) and the cast
        const dim3 g0 = grid2d(cdiv(s->lw[0], 2), cdiv(s->lh[0], 2), blk);
        if (s->p.out_dtype == MI_U8)
            hipLaunchKernelGGL((collapse_exact_quad<FMA, uint8_t>), g0, blk, 0, s->stream, up, s->lh[1], s->lw[1],
                               (const float*)s->bestLap[0], s->lh[0], s->lw[0], s->maxv, (uint8_t*)s->out_dev, s->K);
        else
            hipLaunchKernelGGL((collapse_exact_quad<FMA, uint16_t>), g0, blk, 0, s->stream, up, s->lh[1], s->lw[1],
                               (const float*)s->bestLap[0], s->lh[0], s->lw[0], s->maxv, (uint16_t*)s->out_dev, s->K);
        MI_HIP(hipGetLastError());
        return MI_OK;
    }
    s->have_clipped = true;
    size_t n = (size_t)s->lh[0] * s->lw[0] * 3;
    dim3 g((unsigned)((n + 255) / 256));
    if (s->p.out_dtype == MI_U8)
        hipLaunchKernelGGL((finalize_cast<uint8_t>), g, dim3(256), 0, s->stream, up, n, s->maxv,
                           s->clipped, (uint8_t*)s->out_dev);
    else
        hipLaunchKernelGGL((finalize_cast<uint16_t>), g, dim3(256), 0, s->stream, up, n, s->maxv,
                           s->clipped, (uint16_t*)s->out_dev);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

int check_handle(const mi_stack* s) {
    if (!s) return fail(MI_ERR_INVALID, "null handle");
    return MI_OK;
}

}  // namespace

// tiled implementation hooks (kernels_tiled.hpp) need the handle layout
#include "tiled_host.hpp"

namespace {

void invert_affine_host(const double* M, double* iM) {
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0.0 ? 1.0 / D : 0.0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    iM[0] = A11;
    iM[1] = M[1] * (-D);
    iM[3] = M[3] * (-D);
    iM[4] = A22;
    iM[2] = -iM[0] * M[2] - iM[1] * M[5];
    iM[5] = -iM[3] * M[2] - iM[4] * M[5];
}

int round_sat(double v, int hi) {
    double r = std::nearbyint(v);
    return r < 0 ? 0 : (r > hi ? hi : (int)r);
}

// Scratch of the border blur's tile list (kernels_align.hpp): [count, pad to 64 B][bitmap: one bit per 32 x 64 tile]
// [list: one entry per tile].  One buffer per (device, stream) -- calls on one stream are ordered by the stream, calls on
// different streams must not share it -- grown on demand, kept for the life of the process.
struct WarpScratch {
    void* ptr = nullptr;
    size_t bytes = 0;
};
int warp_scratch(int device, hipStream_t st, size_t ntiles, uint32_t** cnt, uint32_t** bitmap, uint32_t** list, size_t* clear_bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, WarpScratch> cache;
    const size_t bm_bytes = (((ntiles + 31) / 32) * 4 + 63) & ~(size_t)63;
    const size_t need = 64 + bm_bytes + ntiles * 4;
    std::lock_guard<std::mutex> lk(mu);
    WarpScratch& w = cache[{device, st}];
    if (w.bytes < need) {
        if (w.ptr) {
            MI_HIP(hipStreamSynchronize(st));
            (void)hipFree(w.ptr);
            w.ptr = nullptr;
            w.bytes = 0;
        }
        MI_HIP(hipMalloc(&w.ptr, need));
        w.bytes = need;
    }
    *cnt = (uint32_t*)w.ptr;
    *bitmap = (uint32_t*)((char*)w.ptr + 64);
    *list = (uint32_t*)((char*)w.ptr + 64 + bm_bytes);
    *clear_bytes = 64 + bm_bytes;
    return MI_OK;
}

// the coordinate tables of one warp (warp_coord_tables): one buffer per (device, stream), as above
int warp_coord_scratch(int device, hipStream_t st, size_t n_ints, int** tab) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, WarpScratch> cache;
    const size_t need = n_ints * sizeof(int);
    std::lock_guard<std::mutex> lk(mu);
    WarpScratch& w = cache[{device, st}];
    if (w.bytes < need) {
        if (w.ptr) {
            MI_HIP(hipStreamSynchronize(st));
            (void)hipFree(w.ptr);
            w.ptr = nullptr;
            w.bytes = 0;
        }
        MI_HIP(hipMalloc(&w.ptr, need));
        w.bytes = need;
    }
    *tab = (int*)w.ptr;
    return MI_OK;
}

// the blurred-border composite behind a warp (align.py:245-251): passes over the tiles that hold a masked pixel
template <typename T>
int blur_launch(int device, hipStream_t st, void* side, void* out, uint8_t* valid, int h, int w, const GaussArgs& g,
                bool tiles_marked) {
    auto kb = border_blur_tiles<T>;
    const int r = g.ksize / 2;
    const size_t lds_blur = ((size_t)(BT_H + 2 * r) * (BT_W + 2 * r) * (sizeof(T) == 1 ? 1 : 2) + 3 * (size_t)(BT_H + 2 * r) * BT_W) * 4;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        // largest blur kernel (31 taps): 62 x 94 raw pixels + 3 x 62 x 64 floats
        MI_HIP(hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(((size_t)62 * 94 * 2 + 3 * 62 * 64) * 4)));
        attr_set = true;
    }
    const int tiles_x = cdiv(w, BT_W), tiles_y = cdiv(h, BT_H);
    uint32_t *cnt = nullptr, *bitmap = nullptr, *list = nullptr;
    size_t clear = 0;
    int rc = warp_scratch(device, st, (size_t)tiles_x * tiles_y, &cnt, &bitmap, &list, &clear);
    if (rc) return rc;
    const size_t npx = (size_t)h * w;
    if (!tiles_marked) {   // the warp kernel did not mark the tiles with masked pixels itself: scan the mask
        MI_HIP(hipMemsetAsync(cnt, 0, clear, st));
        hipLaunchKernelGGL(mask_scan_tiles, dim3((unsigned)std::min<size_t>(2048, (npx / 16 + 256) / 256)), dim3(256), 0, st,
                           (const uint8_t*)valid, h, w, tiles_x, bitmap);
    }
    if (!tiles_marked)   // (the affine warp kernel lists the tiles it marks)
        hipLaunchKernelGGL(tile_bitmap_to_list, dim3(1), dim3(1024), 0, st, (const uint32_t*)bitmap, (tiles_x * tiles_y + 31) / 32, cnt, list);
    hipLaunchKernelGGL(kb, dim3(1024), dim3(256), lds_blur, st, (const T*)out, (const uint8_t*)valid, (T*)side, h, w, tiles_x, g,
                       (const uint32_t*)cnt, (const uint32_t*)list);
    hipLaunchKernelGGL((border_blur_scatter<T>), dim3(1024), dim3(256), 0, st, (T*)out, (const uint8_t*)valid, (const T*)side,
                       h, w, tiles_x, (const uint32_t*)cnt, (const uint32_t*)list);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

template <typename T>
int warp_launch(int device, hipStream_t st, const void* src, void* side, void* out, uint8_t* valid, int h, int w,
                const AffineArgs& a, bool blur, const GaussArgs& g, const PerspArgs* persp) {
    bool tiles_marked = false;
    if (persp) {
        const dim3 blk(64, 4), grid(cdiv(w, 64), cdiv(h, 4));
        hipLaunchKernelGGL((warp_perspective_kernel<T>), grid, blk, 0, st, (const T*)src, (T*)out, valid, a, *persp);
    } else {
        uint32_t *bitmap = nullptr, *cnt = nullptr, *list = nullptr;
        size_t clear = 0;
        const int blur_tiles_x = cdiv(w, BT_W);
        if (blur) {   // the kernel marks and lists the blur tiles that hold masked pixels (the scratch is zeroed by warp_coord_tables)
            int rc = warp_scratch(device, st, (size_t)blur_tiles_x * cdiv(h, BT_H), &cnt, &bitmap, &list, &clear);
            if (rc) return rc;
            tiles_marked = true;
        }
        // LDS-staged tiles of 256 x 32 (16) destination pixels, four pixels x 8 (4) rows per thread; whole-dword stores
        // when every row starts 4-byte aligned
        int* tab = nullptr;
        {
            int rc = warp_coord_scratch(device, st, 2 * ((size_t)w + h), &tab);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(warp_coord_tables, dim3(cdiv(std::max(w, h), 256)), dim3(256), 0, st, a, tab, cnt, (int)(clear / 4));
        const bool vec = (w % 4) == 0 && ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(valid)) & 3) == 0;
        const int gx = cdiv(w, WT_W), gy = cdiv(h, WarpTile<T>::TH);   // tiles; the kernel's header has the block order
        const dim3 grid(gx >= 3 && gy >= 3 ? (2 * gx + 2 * (gy - 2)) * WT_SPLIT + gx * (gy - 2) : gx * gy);
        const size_t lds = (size_t)WT_LDS_DWORDS * 4;
        auto kv = warp_affine_tiled<T, true>;
        auto ks = warp_affine_tiled<T, false>;
        static thread_local bool attr_set = false;
        if (!attr_set) {
            MI_HIP(hipFuncSetAttribute((const void*)kv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            MI_HIP(hipFuncSetAttribute((const void*)ks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
        if (vec) hipLaunchKernelGGL(kv, grid, dim3(256), lds, st, (const T*)src, (T*)out, valid, a, bitmap, list, blur_tiles_x, (const int*)tab, gx, gy);
        else hipLaunchKernelGGL(ks, grid, dim3(256), lds, st, (const T*)src, (T*)out, valid, a, bitmap, list, blur_tiles_x, (const int*)tab, gx, gy);
    }
    MI_HIP(hipGetLastError());
    // the warped image went straight to `out`; the few pixels outside the source frame are blurred from it into `side`
    // and copied back
    return blur ? blur_launch<T>(device, st, side, out, valid, h, w, g, tiles_marked) : MI_OK;
}

}  // namespace

namespace {

// solve the symmetric 4x4 system H d = r (column-scaled Gaussian elimination, partial pivoting)
bool solve4(const double Hs[10], const double r[4], double d[4]) {
    double A[4][5];
    int k = 0;
    double Hm[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = i; j < 4; ++j) Hm[i][j] = Hm[j][i] = Hs[k++];
    double sc[4];
    for (int i = 0; i < 4; ++i) {
        if (!(Hm[i][i] > 0)) return false;
        sc[i] = 1.0 / std::sqrt(Hm[i][i]);
    }
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 4; ++j) A[i][j] = Hm[i][j] * sc[i] * sc[j];
        A[i][4] = r[i] * sc[i];
    }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int i = c + 1; i < 4; ++i)
            if (std::fabs(A[i][c]) > std::fabs(A[piv][c])) piv = i;
        if (std::fabs(A[piv][c]) < 1e-12) return false;
        if (piv != c)
            for (int j = 0; j < 5; ++j) std::swap(A[c][j], A[piv][j]);
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

struct EccLevel {
    int h, w;
    float *tmpl, *img;
};

}  // namespace

// Device-resident aligner: the pyramids of the reference (template) and of the moving frame, the
// scratch for the sums, all allocated once per (shape, dtype, subsample).
struct mi_aligner {
    int device = 0, height = 0, width = 0, dtype = 0, subsample = 1;
    int area = 0;      // sub-sample by area mean (cv2.resize INTER_AREA) instead of img[::s, ::s]
    int h = 0, w = 0;  // size of the sub-sampled images the estimate works on
    std::vector<EccLevel> lv;   // tmpl: one image; img: `cap` images (frame f at + f * h * w)
    float* gray = nullptr;
    int cap = 0;                     // moving frames the per-frame buffers hold
    double* partial = nullptr;       // [cap][ECC_MAX_BLOCKS][ECC_NSUM]
    unsigned int* ticket = nullptr;  // [cap]
    EccState* dstate = nullptr;      // [cap] the frames' Gauss-Newton state (device memory: the kernels iterate on it)
    EccState* hstate = nullptr;      // pinned host copy: initial values up, `active` flags and results down
    EccStateH* dstate_h = nullptr;   // [cap] state of the 8-DoF refinement (ALIGN_HOMOGRAPHY)
    EccStateH* hstate_h = nullptr;   // pinned host copy
    std::vector<void*> bufs;         // template pyramid + gray scratch
    std::vector<void*> fbufs;        // per-frame buffers (re-allocated when the capacity grows)
    hipStream_t own = nullptr;       // used when the caller passes no stream: handles on different host
                                     // threads then run side by side instead of serialising on stream 0
    bool have_ref = false;
    // mi_align_stack_device: side lanes of the warp stage.  The frames of a batch are warped round-robin on the stacker's
    // stream and on WARP_LANES - 1 streams of the handle, each with its own border-blur scratch, so that one frame's short,
    // latency-bound kernels (tables, tile list, border blur of a few hundred tiles, scatter) run beside the next frame's warp
    // instead of in front of it.
    hipStream_t wst[3] = {nullptr, nullptr, nullptr};   // (the stacker's side streams: borrowed per call, never destroyed here)
    hipEvent_t wev[3] = {nullptr, nullptr, nullptr}, wstart = nullptr;
    void *wtmp[3] = {nullptr, nullptr, nullptr}, *wmask[3] = {nullptr, nullptr, nullptr};
    // optional coarse initialiser (mi_aligner_set_phase_init): phase correlation on pyramid level `pc_level`
    bool phase_init = false;
    int pc_level = 0, pc_P = 0, pc_Q = 0;
    float2 *pc_ref = nullptr, *pc_mov = nullptr;   // P x Q spectra: the template's (kept per reference) and a frame's
    double* pc_out = nullptr;                      // [cap][3] (dx, dy, response) on the device
    bool pc_ref_valid = false;
};

namespace {

int pc_log2(int n) { int l = 0; while ((1 << l) < n) ++l; return l; }

// 2-D DFT of a P x Q complex plane in place (rows, then columns)
template <bool INVERSE>
void pc_fft2(hipStream_t st, float2* d, int P, int Q) {
    hipLaunchKernelGGL((pc_fft_lines<INVERSE>), dim3(P), dim3(std::max(Q / 2, 1)), 0, st, d, Q, pc_log2(Q), (size_t)1, (size_t)Q);
    hipLaunchKernelGGL((pc_fft_lines<INVERSE>), dim3(Q), dim3(std::max(P / 2, 1)), 0, st, d, P, pc_log2(P), (size_t)Q, (size_t)1);
}

// spectrum of one windowed, padded plane
void pc_spectrum(hipStream_t st, const float* plane, int h, int w, float2* out, int P, int Q) {
    hipLaunchKernelGGL(pc_prepare, dim3(cdiv(Q, 64), cdiv(P, 4)), dim3(64, 4), 0, st, plane, h, w, out, P, Q);
    pc_fft2<false>(st, out, P, Q);
}

// mov's spectrum (in `fm`, destroyed) against the reference's `fr`: (dx, dy, response) -> dev_out3
void pc_correlate(hipStream_t st, const float2* fr, float2* fm, int P, int Q, double* dev_out3) {
    const size_t n = (size_t)P * Q;
    hipLaunchKernelGGL(pc_cross_power, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, fm, fr, n);
    pc_fft2<true>(st, fm, P, Q);
    hipLaunchKernelGGL(pc_peak, dim3(1), dim3(1024), 0, st, (const float2*)fm, P, Q, dev_out3);
}

void aligner_free_frames(mi_aligner* al) {
    for (void* b : al->fbufs) (void)hipFree(b);
    al->fbufs.clear();
    if (al->hstate) (void)hipHostFree(al->hstate);
    al->hstate = nullptr;
    if (al->hstate_h) (void)hipHostFree(al->hstate_h);
    al->hstate_h = nullptr;
    al->cap = 0;
}

void aligner_free(mi_aligner* al) {
    aligner_free_frames(al);
    for (void* b : al->bufs) (void)hipFree(b);
    al->bufs.clear();
    if (al->own) (void)hipStreamDestroy(al->own);
    al->own = nullptr;
    for (int j = 0; j < 3; ++j) {
        if (al->wev[j]) (void)hipEventDestroy(al->wev[j]);
        (void)hipFree(al->wtmp[j]);
        (void)hipFree(al->wmask[j]);
        al->wst[j] = nullptr; al->wev[j] = nullptr; al->wtmp[j] = al->wmask[j] = nullptr;
    }
    if (al->wstart) (void)hipEventDestroy(al->wstart);
    al->wstart = nullptr;
}

// The side lanes of mi_align_stack_device's warp stage (scratch created on first use; one frame + one mask each).
// Round 5 built them on streams of their own (1 against 4 lanes: 0.0371 vs 0.0369 s) -- with the stacker's three streams and
// the estimator's that made seven streams on the runtime's four hardware queues, so the lanes queued behind each
// other anyway.  Round 6: the lanes ARE the stacker's side streams (border tiles / levels >= 1), which carry nothing while a
// batch buffer fills: no stream is added.
#ifndef MI_WARP_LANES
#define MI_WARP_LANES 2
#endif
constexpr int WARP_LANES = MI_WARP_LANES;
int aligner_warp_lanes(mi_aligner* al, size_t frame_bytes, size_t mask_bytes) {
    if (al->wstart) return MI_OK;
    static_assert(WARP_LANES >= 1 && WARP_LANES <= 3, "lanes: the stacker's own stream and its two side streams");
    for (int j = 0; j < WARP_LANES - 1; ++j) {
        MI_HIP(hipEventCreateWithFlags(&al->wev[j], hipEventDisableTiming));
        if (hipMalloc(&al->wtmp[j], frame_bytes) != hipSuccess || hipMalloc(&al->wmask[j], mask_bytes) != hipSuccess)
            return fail(MI_ERR_NOMEM, "out of device memory");
    }
    MI_HIP(hipEventCreateWithFlags(&al->wstart, hipEventDisableTiming));
    return MI_OK;
}

// per-frame buffers for `n` moving frames
int aligner_reserve(mi_aligner* al, int n) {
    if (n <= al->cap) return MI_OK;
    MI_HIP(hipDeviceSynchronize());
    aligner_free_frames(al);
    auto dalloc = [&](size_t bytes) -> void* {
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        al->fbufs.push_back(p);
        return p;
    };
    bool ok = true;
    for (auto& L : al->lv) {
        const size_t nb = (size_t)L.h * L.w * 4 * n;
        L.img = (float*)dalloc(nb);
        ok = ok && L.img;
    }
    al->partial = (double*)dalloc((size_t)n * ECC_MAX_BLOCKS * (ECC_NSUM_H > ECC_NSUM ? ECC_NSUM_H : ECC_NSUM) * sizeof(double));
    al->ticket = (unsigned int*)dalloc(sizeof(unsigned int) * n);
    ok = ok && al->partial && al->ticket;
    if (ok) ok = hipMemset(al->ticket, 0, sizeof(unsigned int) * n) == hipSuccess;
    al->dstate = (EccState*)dalloc(sizeof(EccState) * n);
    ok = ok && al->dstate;
    al->pc_out = (double*)dalloc(sizeof(double) * 3 * n);
    ok = ok && al->pc_out;
    if (ok) ok = hipHostMalloc((void**)&al->hstate, sizeof(EccState) * n, hipHostMallocDefault) == hipSuccess;
    al->dstate_h = (EccStateH*)dalloc(sizeof(EccStateH) * n);
    ok = ok && al->dstate_h;
    if (ok) ok = hipHostMalloc((void**)&al->hstate_h, sizeof(EccStateH) * n, hipHostMallocDefault) == hipSuccess;
    if (!ok) {
        aligner_free_frames(al);
        return fail(MI_ERR_NOMEM, "out of device memory");
    }
    al->cap = n;
    return MI_OK;
}

// gray + pyramid (+ gradients) of one image: the template, or moving frame `slot`
// `upto`: build levels 0 .. upto-1 only (the batch builds the small levels of all its frames with one launch each)
int aligner_build(mi_aligner* al, hipStream_t st, const void* dev_img, bool is_tmpl, int slot, int upto = 1 << 30) {
    const dim3 blk(64, 4), g0(cdiv(al->w, 64), cdiv(al->h, 4));
    auto& lv = al->lv;
    auto at = [&](float* base, size_t l) { return base + (size_t)slot * lv[l].h * lv[l].w; };
    size_t first = 0;   // first level the loop below still has to build
    if (al->subsample == 2 && lv.size() >= 2 && upto >= 2) {
        // the common factor: gray, level 0 and level 1 in one pass over the frame (ecc_pyramid2)
        float* d0 = is_tmpl ? lv[0].tmpl : at(lv[0].img, 0);
        float* d1 = is_tmpl ? lv[1].tmpl : at(lv[1].img, 1);
        const dim3 gp(cdiv(al->w, 64), cdiv(al->h, 32));
        if (al->dtype == MI_U8) {
            if (al->area) hipLaunchKernelGGL((ecc_pyramid2<uint8_t, true>), gp, dim3(256), 0, st, (const uint8_t*)dev_img, al->height, al->width, al->h, al->w, d0, lv[1].h, lv[1].w, d1);
            else hipLaunchKernelGGL((ecc_pyramid2<uint8_t, false>), gp, dim3(256), 0, st, (const uint8_t*)dev_img, al->height, al->width, al->h, al->w, d0, lv[1].h, lv[1].w, d1);
        } else {
            if (al->area) hipLaunchKernelGGL((ecc_pyramid2<uint16_t, true>), gp, dim3(256), 0, st, (const uint16_t*)dev_img, al->height, al->width, al->h, al->w, d0, lv[1].h, lv[1].w, d1);
            else hipLaunchKernelGGL((ecc_pyramid2<uint16_t, false>), gp, dim3(256), 0, st, (const uint16_t*)dev_img, al->height, al->width, al->h, al->w, d0, lv[1].h, lv[1].w, d1);
        }
        first = 2;
    } else if (al->subsample == 2) {   // the common factor: four outputs per thread from 8-byte loads
        const dim3 g2(cdiv(cdiv(al->w, 4), 64), cdiv(al->h, 4));
        if (al->dtype == MI_U8) {
            if (al->area) hipLaunchKernelGGL((ecc_gray_s2<uint8_t, true>), g2, blk, 0, st, (const uint8_t*)dev_img, al->height, al->width, al->h, al->w, al->gray);
            else hipLaunchKernelGGL((ecc_gray_s2<uint8_t, false>), g2, blk, 0, st, (const uint8_t*)dev_img, al->height, al->width, al->h, al->w, al->gray);
        } else {
            if (al->area) hipLaunchKernelGGL((ecc_gray_s2<uint16_t, true>), g2, blk, 0, st, (const uint16_t*)dev_img, al->height, al->width, al->h, al->w, al->gray);
            else hipLaunchKernelGGL((ecc_gray_s2<uint16_t, false>), g2, blk, 0, st, (const uint16_t*)dev_img, al->height, al->width, al->h, al->w, al->gray);
        }
    } else if (al->dtype == MI_U8)
        hipLaunchKernelGGL((ecc_gray<uint8_t>), g0, blk, 0, st, (const uint8_t*)dev_img, al->height, al->width, al->h, al->w,
                           al->subsample, al->area, al->gray);
    else
        hipLaunchKernelGGL((ecc_gray<uint16_t>), g0, blk, 0, st, (const uint16_t*)dev_img, al->height, al->width, al->h, al->w,
                           al->subsample, al->area, al->gray);
    for (size_t l = first; l < lv.size() && (int)l < upto; ++l) {
        float* dst = is_tmpl ? lv[l].tmpl : at(lv[l].img, l);
        const float* src = l == 0 ? al->gray : (is_tmpl ? lv[l - 1].tmpl : at(lv[l - 1].img, l - 1));
        const int sh = l == 0 ? al->h : lv[l - 1].h, sw = l == 0 ? al->w : lv[l - 1].w;
        const dim3 gt(cdiv(lv[l].w, 64), cdiv(lv[l].h, 16));
        if (l == 0) hipLaunchKernelGGL((ecc_blur_tile<0>), gt, dim3(256), 0, st, src, sh, sw, dst, lv[l].h, lv[l].w);
        else hipLaunchKernelGGL((ecc_blur_tile<1>), gt, dim3(256), 0, st, src, sh, sw, dst, lv[l].h, lv[l].w);
    }
    MI_HIP(hipGetLastError());
    return MI_OK;
}

// gray + pyramid of the n moving frames of a batch (slots 0 .. n-1): levels 0 and 1 of all frames in one launch where
// ecc_pyramid2 applies (sub-sampling 2, n > 1), frame by frame otherwise; the levels from `split` on, one launch per level
int aligner_build_batch(mi_aligner* al, hipStream_t st, const void* const* dev_movs, int n) {
    static_assert(ECC_MAXF <= 128, "EccFramePtrs holds 128 frames");
    auto& lv = al->lv;
    const int split = n > 1 ? 2 : 1 << 30;
    int rc;
    if (n > 1 && al->subsample == 2 && lv.size() >= 2) {
        EccFramePtrs fr{};
        for (int k = 0; k < n; ++k) fr.p[k] = dev_movs[k];
        const dim3 gp(cdiv(al->w, 64), cdiv(al->h, 32), n);
        float *d0 = lv[0].img, *d1 = lv[1].img;
        if (al->dtype == MI_U8) {
            if (al->area) hipLaunchKernelGGL((ecc_pyramid2_batch<uint8_t, true>), gp, dim3(256), 0, st, fr, al->height, al->width, al->h, al->w, d0, lv[1].h, lv[1].w, d1);
            else hipLaunchKernelGGL((ecc_pyramid2_batch<uint8_t, false>), gp, dim3(256), 0, st, fr, al->height, al->width, al->h, al->w, d0, lv[1].h, lv[1].w, d1);
        } else {
            if (al->area) hipLaunchKernelGGL((ecc_pyramid2_batch<uint16_t, true>), gp, dim3(256), 0, st, fr, al->height, al->width, al->h, al->w, d0, lv[1].h, lv[1].w, d1);
            else hipLaunchKernelGGL((ecc_pyramid2_batch<uint16_t, false>), gp, dim3(256), 0, st, fr, al->height, al->width, al->h, al->w, d0, lv[1].h, lv[1].w, d1);
        }
    } else {
        for (int k = 0; k < n; ++k)
            if ((rc = aligner_build(al, st, dev_movs[k], false, k, split))) return rc;
    }
    for (size_t l = (size_t)split; l < lv.size(); ++l) {
        const auto& a = lv[l - 1];
        const auto& b = lv[l];
        hipLaunchKernelGGL((ecc_blur_tile<1>), dim3(cdiv(b.w, 64), cdiv(b.h, 16), n), dim3(256), 0, st, (const float*)a.img, a.h, a.w,
                           b.img, b.h, b.w);
    }
    MI_HIP(hipGetLastError());
    return MI_OK;
}

// Coarse-to-fine forward-additive ECC of `n` <= ECC_MAXF moving frames (pyramids in slots 0..n-1)
// against the template: one launch and one host round trip per Gauss-Newton iteration for the whole
// batch; frames that converged on a level sit out the rest of it.  M_out[f] in FULL-resolution pixel
// coordinates (translation scaled by the sub-sampling factor, as align.py:224-231 does).  A frame
// the method fails on (no overlap, constant image, degenerate transform) gets cc = -2.
// iterations enqueued between two looks at the frames' `active` flags: the first look of a level comes late (a level
// rarely converges in fewer), the following ones sooner
// (the two coarsest levels start far from their optimum; the finer ones inherit it and stop after one to three steps)
#ifndef MI_ECC_CHUNK_FINE
#define MI_ECC_CHUNK_FINE 4
#endif
constexpr int ECC_CHUNK_FIRST = 8, ECC_CHUNK_FIRST_FINE = MI_ECC_CHUNK_FINE, ECC_CHUNK_NEXT = 4;
// samples per Gauss-Newton sum at least: levels with more pixels are sampled on every `step`-th row and column
#ifndef MI_ECC_MIN_SAMPLES
#define MI_ECC_MIN_SAMPLES 300000
#endif
inline int ecc_sample_step(size_t np) {
    int step = 1;
    while ((size_t)(step + 1) * (step + 1) * (size_t)MI_ECC_MIN_SAMPLES <= np) ++step;
    return step;
}

// `M9_out` (optional): ALIGN_HOMOGRAPHY -- the similarity is refined to 8 degrees of freedom on the finest level
// (ecc_accumulate_h) and row f of M9_out receives the 3 x 3 matrix (moving -> reference, full-resolution pixels, M[8] = 1);
// a frame whose refinement fails or does not raise the correlation keeps its similarity, as a 3 x 3.
// `M_init` (n x 6, moving -> reference in full-resolution pixels, what a previous estimate returned) + `start_level`: the
// iteration starts from that transform on pyramid level `start_level` instead of from the identity on the coarsest level
// (mi_aligner_refine_batch).
// `tslots` (n, optional): frame k is registered against the pyramid of the batch's frame tslots[k] instead of the handle's
// reference (mi_aligner_estimate_pairs; no phase-correlation start there).
int aligner_solve(mi_aligner* al, hipStream_t st, int n, int max_iters, double eps, double* M_out, double* cc_out,
                  int* iters_out, double* M9_out = nullptr, const double* M_init = nullptr, int start_level = -1,
                  const int* tslots = nullptr) {
    if (max_iters < 1) max_iters = 50;
    if (!(eps > 0)) eps = 1e-8;
    auto& lv = al->lv;
    EccState* const hs = al->hstate;
    const int l_first = M_init ? std::min(std::max(start_level, 0), (int)lv.size() - 1) : (int)lv.size() - 1;
    for (int k = 0; k < n; ++k) {
        hs[k] = EccState{};
        hs[k].a = 1.0;
        hs[k].rho = -1.0;
        hs[k].last_rho = -2.0;
        hs[k].tslot = tslots ? tslots[k] : -1;
        if (M_init) {
            // W = M^-1 (reference -> moving, what the iteration works on): A = [a -b; b a], T in origin coordinates of level
            // l_first (sub-sampled pixels / 2^l_first) -- the inverse of the read-out at the end of this function
            const double* M = M_init + 6 * k;
            const double ia = M[0], ib = M[3], det = ia * ia + ib * ib;
            if (!(det > 1e-12)) return fail(MI_ERR_INVALID, "degenerate starting transform for frame %d", k);
            const double a = ia / det, b = -ib / det, sc = (double)al->subsample * std::ldexp(1.0, l_first);
            hs[k].a = a;
            hs[k].b = b;
            hs[k].T0 = -(a * M[2] - b * M[5]) / sc;
            hs[k].T1 = -(b * M[2] + a * M[5]) / sc;
        }
    }
    if (al->phase_init && !M_init && !tslots) {
        // coarse initialiser: the translation phase correlation finds on level pc_level (<= 512 pixels per side) becomes
        // the starting translation of the Gauss-Newton iteration at the coarsest level
        const EccLevel& PL = lv[al->pc_level];
        const int P = al->pc_P, Q = al->pc_Q;
        if (!al->pc_ref) {
            void *a = nullptr, *b = nullptr;
            if (hipMalloc(&a, sizeof(float2) * P * Q) != hipSuccess || hipMalloc(&b, sizeof(float2) * P * Q) != hipSuccess)
                return fail(MI_ERR_NOMEM, "out of device memory");
            al->bufs.push_back(a);
            al->bufs.push_back(b);
            al->pc_ref = (float2*)a;
            al->pc_mov = (float2*)b;
        }
        if (!al->pc_ref_valid) {
            pc_spectrum(st, PL.tmpl, PL.h, PL.w, al->pc_ref, P, Q);
            al->pc_ref_valid = true;
        }
        for (int k = 0; k < n; ++k) {
            pc_spectrum(st, PL.img + (size_t)k * PL.h * PL.w, PL.h, PL.w, al->pc_mov, P, Q);
            pc_correlate(st, al->pc_ref, al->pc_mov, P, Q, al->pc_out + 3 * k);
        }
        MI_HIP(hipGetLastError());
        std::vector<double> o3((size_t)3 * n);
        MI_HIP(hipMemcpyAsync(o3.data(), al->pc_out, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, st));
        MI_HIP(hipStreamSynchronize(st));
        const double down = std::ldexp(1.0, al->pc_level - ((int)lv.size() - 1));   // level pc_level -> the coarsest level
        for (int k = 0; k < n; ++k) {
            // R = Mov conj(Ref): the peak sits at d with mov(x) = ref(x - d), i.e. ref(x) ~ mov(x + d) -- exactly the
            // translation of W (kernels_ecc.hpp); a weak peak (flat or unrelated content) is not used
            if (o3[3 * k + 2] > 0.02) {
                hs[k].T0 = o3[3 * k] * down;
                hs[k].T1 = o3[3 * k + 1] * down;
            }
        }
    }
    MI_HIP(hipMemcpyAsync(al->dstate, hs, sizeof(EccState) * n, hipMemcpyHostToDevice, st));
    for (int l = l_first; l >= 0; --l) {
        const EccLevel& L = lv[l];
        const double cx = 0.5 * (L.w - 1), cy = 0.5 * (L.h - 1);
        const size_t np = (size_t)L.h * L.w;
        // sample every `step`-th pixel in both directions: a third of a million samples and more are plenty for 4 parameters
        // (6 MP level: step 4 = 375 K samples, 1.5 MP level: step 2; rounds 3-4 kept half a million -- 667 K and all 1.5 M --:
        // the recovered transforms of config 4 stay ten times inside the tolerances either way, profiles/r05)
        static const int ecc_step = study_env("MI_ECC_STEP", 0);   // -DMI_STUDY: force a step
        int step = ecc_sample_step(np);
        if (ecc_step > 0) step = ecc_step;
        // ~48 samples per thread (the 28 double sums cost a thread ~500 instructions to reduce, as much as 5 samples),
        // at most ECC_MAX_BLOCKS blocks (4 per CU).  Measured (round 3, config 4 / the estimate of a 16-frame batch alone):
        // 3072 -> 0.054 s / 6.6 ms, 6144 -> 0.047 / 5.5, 12288 -> 0.042 / 5.5, 24576 -> 0.041 / 6.7: fewer, longer
        // workgroups leave more of the GPU to the warps and the fuse that run beside the estimate
        static const size_t per_blk = (size_t)study_env("MI_ECC_PER_BLOCK", 12288);   // study knob
        const size_t work = (np / ((size_t)step * step) + per_blk - 1) / per_blk;
        const unsigned nblk = (unsigned)(work < 1 ? 1 : (work > (size_t)ECC_MAX_BLOCKS ? ECC_MAX_BLOCKS : work));
        hipLaunchKernelGGL(ecc_level_begin, dim3(cdiv(n, 64)), dim3(64), 0, st, al->dstate, n, cx, cy);
        const double reach = std::hypot(cx, cy);
        for (int it = 0; it < max_iters;) {
            const int chunk = std::min(it == 0 ? (l + 2 >= (int)lv.size() ? ECC_CHUNK_FIRST : ECC_CHUNK_FIRST_FINE) : ECC_CHUNK_NEXT,
                                       max_iters - it);
            for (int j = 0; j < chunk; ++j)   // frames that have stopped leave at once
                hipLaunchKernelGGL(ecc_accumulate, dim3(nblk, n), dim3(256), 0, st, L.tmpl, L.img, np, L.h, L.w, al->dstate, step,
                                   al->partial, al->ticket, reach, eps);
            it += chunk;
            MI_HIP(hipMemcpyAsync(hs, al->dstate, sizeof(EccState) * n, hipMemcpyDeviceToHost, st));
            MI_HIP(hipStreamSynchronize(st));
            int nact = 0;
            for (int k = 0; k < n; ++k) nact += hs[k].active;
            if (!nact) break;
        }
        // back to origin coordinates, then up to the next finer level (u_f = 2 u_c, x_f = 2 x_c)
        hipLaunchKernelGGL(ecc_level_end, dim3(cdiv(n, 64)), dim3(64), 0, st, al->dstate, n, cx, cy, l > 0 ? 1 : 0);
    }
    EccStateH* const hh = al->hstate_h;
    double Rn = 1.0, cx0 = 0.0, cy0 = 0.0;
    if (M9_out) {
        // 8-DoF refinement on the finest level, from the converged similarity
        const EccLevel& L = lv[0];
        cx0 = 0.5 * (L.w - 1);
        cy0 = 0.5 * (L.h - 1);
        Rn = (double)hypotf((float)cx0, (float)cy0);   // the kernel's own (float) radius
        const size_t np = (size_t)L.h * L.w;
        const int step = ecc_sample_step(np);
        const size_t work = (np / ((size_t)step * step) + 12287) / 12288;
        const unsigned nblk = (unsigned)(work < 1 ? 1 : (work > (size_t)ECC_MAX_BLOCKS ? ECC_MAX_BLOCKS : work));
        hipLaunchKernelGGL(ecc_h_from_similarity, dim3(cdiv(n, 64)), dim3(64), 0, st, (const EccState*)al->dstate, al->dstate_h, n, Rn);
        for (int it = 0; it < max_iters;) {
            const int chunk = std::min(it == 0 ? ECC_CHUNK_FIRST : ECC_CHUNK_NEXT, max_iters - it);
            for (int j = 0; j < chunk; ++j)
                hipLaunchKernelGGL(ecc_accumulate_h, dim3(nblk, n), dim3(256), 0, st, L.tmpl, L.img, np, L.h, L.w, al->dstate_h, step,
                                   al->partial, al->ticket, eps);
            it += chunk;
            MI_HIP(hipMemcpyAsync(hh, al->dstate_h, sizeof(EccStateH) * n, hipMemcpyDeviceToHost, st));
            MI_HIP(hipStreamSynchronize(st));
            int nact = 0;
            for (int k = 0; k < n; ++k) nact += hh[k].active;
            if (!nact) break;
        }
        MI_HIP(hipGetLastError());
    }
    MI_HIP(hipMemcpyAsync(hs, al->dstate, sizeof(EccState) * n, hipMemcpyDeviceToHost, st));
    if (M9_out) MI_HIP(hipMemcpyAsync(hh, al->dstate_h, sizeof(EccStateH) * n, hipMemcpyDeviceToHost, st));
    MI_HIP(hipStreamSynchronize(st));
    const double s = (double)al->subsample;
    for (int k = 0; k < n; ++k) {
        const EccState& f = hs[k];
        double Msim[6];
        double* M = M_out ? M_out + 6 * k : Msim;
        // M (moving -> reference) = W^-1, translation back to full-resolution pixels
        const double det = f.a * f.a + f.b * f.b;
        const bool bad = f.failed || !(det > 1e-12);
        const double ia = bad ? 1.0 : f.a / det, ib = bad ? 0.0 : -f.b / det;  // A^-1 = [ia -ib; ib ia]
        M[0] = ia;  M[1] = -ib; M[2] = bad ? 0.0 : -(ia * f.T0 - ib * f.T1) * s;
        M[3] = ib;  M[4] = ia;  M[5] = bad ? 0.0 : -(ib * f.T0 + ia * f.T1) * s;
        if (cc_out) cc_out[k] = bad ? -2.0 : f.rho;
        if (iters_out) iters_out[k] = f.iters;
        if (M9_out) {
            double* M9 = M9_out + 9 * k;
            const EccStateH& g = hh[k];
            bool use_h = !bad && !g.failed && g.iters > 0 && g.rho >= f.rho - 1e-9;
            if (use_h) {
                // W (reference -> moving) = T(c) S(R) Hn S(1/R) T(-c) at the sub-sampled scale, S(s) W S(1/s) in full-resolution
                // pixels; M = W^-1, normalised
                const double Hn[9] = {g.h[0], g.h[1], g.h[2], g.h[3], g.h[4], g.h[5], g.h[6], g.h[7], 1.0};
                auto mul = [](const double* A, const double* B, double* C) {
                    for (int i = 0; i < 3; ++i)
                        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
                };
                const double L1[9] = {s * Rn, 0, s * cx0, 0, s * Rn, s * cy0, 0, 0, 1};                 // S(s) T(c) S(R)
                const double R1[9] = {1 / (s * Rn), 0, -cx0 / Rn, 0, 1 / (s * Rn), -cy0 / Rn, 0, 0, 1};   // S(1/R) T(-c) S(1/s)
                double t1[9], Wm[9];
                mul(L1, Hn, t1);
                mul(t1, R1, Wm);
                const double det = Wm[0] * (Wm[4] * Wm[8] - Wm[5] * Wm[7]) - Wm[1] * (Wm[3] * Wm[8] - Wm[5] * Wm[6]) +
                                   Wm[2] * (Wm[3] * Wm[7] - Wm[4] * Wm[6]);
                if (fabs(det) > 1e-12) {
                    const double id = 1.0 / det;
                    double I[9] = {(Wm[4] * Wm[8] - Wm[5] * Wm[7]) * id, (Wm[2] * Wm[7] - Wm[1] * Wm[8]) * id, (Wm[1] * Wm[5] - Wm[2] * Wm[4]) * id,
                                   (Wm[5] * Wm[6] - Wm[3] * Wm[8]) * id, (Wm[0] * Wm[8] - Wm[2] * Wm[6]) * id, (Wm[2] * Wm[3] - Wm[0] * Wm[5]) * id,
                                   (Wm[3] * Wm[7] - Wm[4] * Wm[6]) * id, (Wm[1] * Wm[6] - Wm[0] * Wm[7]) * id, (Wm[0] * Wm[4] - Wm[1] * Wm[3]) * id};
                    if (fabs(I[8]) > 1e-12) {
                        for (int q = 0; q < 9; ++q) M9[q] = I[q] / I[8];
                        if (cc_out) cc_out[k] = g.rho;
                        if (iters_out) iters_out[k] = f.iters + g.iters;
                    } else use_h = false;
                } else use_h = false;
            }
            if (!use_h) {
                for (int q = 0; q < 6; ++q) M9[q] = M[q];
                M9[6] = 0.0; M9[7] = 0.0; M9[8] = 1.0;
            }
        }
    }
    return MI_OK;
}

}  // namespace

extern "C" {

int mi_abi_version(void) { return MI_ABI_VERSION; }

const char* mi_last_error(void) { return last_error().c_str(); }

int mi_device_count(int* count) {
    if (!count) return fail(MI_ERR_INVALID, "null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(MI_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return MI_OK;
}

int mi_device_name(int device, char* buf, size_t buflen) {
    if (!buf || !buflen) return fail(MI_ERR_INVALID, "null buffer");
    hipDeviceProp_t prop;
    MI_HIP(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return MI_OK;
}

void mi_stack_default_params(mi_stack_params_t* p) {
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->in_dtype = MI_U8;
    p->out_dtype = MI_U8;
    p->min_size = 32;
    p->kernel_size = 5;
    p->gen_kernel = 0.4;
    p->float_type = MI_F32;
    p->use_fma = 1;
    p->device = 0;
    p->impl = MI_IMPL_AUTO;
    p->batch_frames = 0;
}

int mi_device_malloc(int device, size_t bytes, void** dev_ptr) {
    if (!dev_ptr) return fail(MI_ERR_INVALID, "null out pointer");
    MI_HIP(hipSetDevice(device));
    MI_HIP(hipMalloc(dev_ptr, bytes ? bytes : 1));
    return MI_OK;
}
int mi_device_free(int device, void* dev_ptr) {
    MI_HIP(hipSetDevice(device));
    MI_HIP(hipFree(dev_ptr));
    return MI_OK;
}
int mi_memcpy_h2d(int device, void* dev_dst, const void* host_src, size_t bytes) {
    MI_HIP(hipSetDevice(device));
    MI_HIP(hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
    return MI_OK;
}
int mi_memcpy_d2h(int device, void* host_dst, const void* dev_src, size_t bytes) {
    MI_HIP(hipSetDevice(device));
    MI_HIP(hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost));
    return MI_OK;
}

int mi_memcpy_d2d(int device, void* dev_dst, const void* dev_src, size_t bytes) {
    if (!dev_dst || !dev_src) return fail(MI_ERR_INVALID, "null pointer");
    MI_HIP(hipSetDevice(device));
    MI_HIP(hipMemcpy(dev_dst, dev_src, bytes, hipMemcpyDeviceToDevice));
    return MI_OK;
}

int mi_memcpy_d2d_async(int device, void* stream, void* dev_dst, const void* dev_src, size_t bytes) {
    if (!dev_dst || !dev_src) return fail(MI_ERR_INVALID, "null pointer");
    MI_HIP(hipSetDevice(device));
    MI_HIP(hipMemcpyAsync(dev_dst, dev_src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MI_OK;
}
int mi_device_synchronize(int device) {
    MI_HIP(hipSetDevice(device));
    MI_HIP(hipDeviceSynchronize());
    return MI_OK;
}

int mi_pyr_step(int device, int op, int use_fma, double gen_kernel, const void* host_in, const void* host_in2, int n, int h,
                int w, int c, int h2, int w2, double maxv, void* host_out) {
    if (!host_in || !host_out) return fail(MI_ERR_INVALID, "null argument");
    if (h < 1 || w < 1 || (c != 1 && c != 3) || n < 1) return fail(MI_ERR_INVALID, "bad geometry");
    if (op < MI_PYR_CONVOLVE || op > MI_PYR_CLIP_ABS) return fail(MI_ERR_INVALID, "bad op %d", op);
    MI_HIP(hipSetDevice(device));
    K25 K{};
    {
        const double a = gen_kernel, k[5] = {0.25 - a / 2.0, 0.25, a, 0.25, 0.25 - a / 2.0};
        for (int i = 0; i < 5; ++i)
            for (int j = 0; j < 5; ++j) K.k[i * 5 + j] = (float)(k[i] * k[j]);
    }
    const size_t npx = (size_t)h * w;
    size_t in_elems = npx * c, out_elems = npx * c, in2_elems = 0;
    if (op == MI_PYR_REDUCE) out_elems = (size_t)((h + 1) / 2) * ((w + 1) / 2) * c;
    else if (op == MI_PYR_EXPAND) out_elems = npx * 4 * c;
    else if (op == MI_PYR_FUSE_LAPLACIAN) {
        if (c != 3) return fail(MI_ERR_INVALID, "fuse_laplacian takes 3-channel images");
        in_elems = npx * 3 * n;
        out_elems = npx * 3;
    } else if (op == MI_PYR_COLLAPSE_STEP) {
        if (!host_in2 || h2 < 1 || w2 < 1 || 2 * h2 < h || 2 * w2 < w) return fail(MI_ERR_INVALID, "bad coarse level");
        in2_elems = (size_t)h2 * w2 * c;
    }
    struct Scratch {
        std::vector<void*> v;
        ~Scratch() {
            (void)hipDeviceSynchronize();
            for (void* q : v) (void)hipFree(q);
        }
        float* get(size_t elems) {
            void* q = nullptr;
            if (hipMalloc(&q, (elems ? elems : 1) * sizeof(float)) != hipSuccess) return nullptr;
            v.push_back(q);
            return (float*)q;
        }
    } tmp;
    float *din = tmp.get(in_elems), *dout = tmp.get(out_elems), *din2 = in2_elems ? tmp.get(in2_elems) : nullptr;
    if (!din || !dout || (in2_elems && !din2)) return fail(MI_ERR_NOMEM, "out of device memory");
    hipStream_t st = nullptr;
    MI_HIP(hipMemcpyAsync(din, host_in, in_elems * sizeof(float), hipMemcpyHostToDevice, st));
    const bool fma = use_fma != 0;
    if (op <= MI_PYR_EXPAND) {
        if (c == 3) fma ? pyr_step_launch<3, true>(op, st, din, h, w, dout, K) : pyr_step_launch<3, false>(op, st, din, h, w, dout, K);
        else fma ? pyr_step_launch<1, true>(op, st, din, h, w, dout, K) : pyr_step_launch<1, false>(op, st, din, h, w, dout, K);
    } else if (op == MI_PYR_FUSE_LAPLACIAN) {
        float *q = tmp.get(npx * n), *e = tmp.get(npx * n);
        if (!q || !e) return fail(MI_ERR_NOMEM, "out of device memory");
        const unsigned g1 = (unsigned)((npx * n + 255) / 256);
        if (fma) hipLaunchKernelGGL((step_gray_sq<true>), dim3(g1), dim3(256), 0, st, (const float*)din, npx * n, q);
        else hipLaunchKernelGGL((step_gray_sq<false>), dim3(g1), dim3(256), 0, st, (const float*)din, npx * n, q);
        for (int i = 0; i < n; ++i)
            fma ? pyr_step_launch<1, true>(MI_PYR_CONVOLVE, st, q + (size_t)i * npx, h, w, e + (size_t)i * npx, K)
                : pyr_step_launch<1, false>(MI_PYR_CONVOLVE, st, q + (size_t)i * npx, h, w, e + (size_t)i * npx, K);
        hipLaunchKernelGGL(step_fuse, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, st, (const float*)e, (const float*)din, n, npx, dout);
    } else if (op == MI_PYR_COLLAPSE_STEP) {
        MI_HIP(hipMemcpyAsync(din2, host_in2, in2_elems * sizeof(float), hipMemcpyHostToDevice, st));
        float* up = tmp.get((size_t)h2 * w2 * 4 * c);
        if (!up) return fail(MI_ERR_NOMEM, "out of device memory");
        if (c == 3) fma ? pyr_step_launch<3, true>(MI_PYR_EXPAND, st, din2, h2, w2, up, K) : pyr_step_launch<3, false>(MI_PYR_EXPAND, st, din2, h2, w2, up, K);
        else fma ? pyr_step_launch<1, true>(MI_PYR_EXPAND, st, din2, h2, w2, up, K) : pyr_step_launch<1, false>(MI_PYR_EXPAND, st, din2, h2, w2, up, K);
        hipLaunchKernelGGL(step_add_crop, dim3(cdiv(w, 64), cdiv(h, 4)), dim3(64, 4), 0, st, (const float*)up, 2 * w2, (const float*)din, h, w, c, dout);
    } else {   // MI_PYR_CLIP_ABS
        hipLaunchKernelGGL(step_clip_abs, dim3((unsigned)((out_elems + 255) / 256)), dim3(256), 0, st, (const float*)din, out_elems, (float)maxv, dout);
    }
    MI_HIP(hipGetLastError());
    MI_HIP(hipMemcpyAsync(host_out, dout, out_elems * sizeof(float), hipMemcpyDeviceToHost, st));
    MI_HIP(hipStreamSynchronize(st));
    return MI_OK;
}

int mi_device_mem_info(int device, size_t* free_bytes, size_t* total_bytes) {
    if (!free_bytes || !total_bytes) return fail(MI_ERR_INVALID, "null argument");
    MI_HIP(hipSetDevice(device));
    MI_HIP(hipMemGetInfo(free_bytes, total_bytes));
    return MI_OK;
}

int mi_stack_create(mi_stack_t** out, const mi_stack_params_t* params) {
    if (!out || !params) return fail(MI_ERR_INVALID, "null argument");
    *out = nullptr;
    const mi_stack_params_t& p = *params;
    if (p.height < 1 || p.width < 1) return fail(MI_ERR_INVALID, "bad frame size %dx%d", p.width, p.height);
    // the kernels address pixels inside a frame with 32-bit byte offsets (12 bytes per float pixel)
    if ((uint64_t)p.height * (uint64_t)p.width * 12ull >= (1ull << 32))
        return fail(MI_ERR_UNSUPPORTED, "frames of %dx%d exceed the 357-megapixel limit of the 32-bit in-frame addressing",
                    p.width, p.height);
    if (p.in_dtype != MI_U8 && p.in_dtype != MI_U16 && p.in_dtype != MI_F32)
        return fail(MI_ERR_INVALID, "in_dtype must be MI_U8, MI_U16 or MI_F32");
    if (p.out_dtype != MI_U8 && p.out_dtype != MI_U16)
        return fail(MI_ERR_INVALID, "out_dtype must be MI_U8 or MI_U16");
    if (p.float_type != MI_F32 && p.float_type != MI_F64) return fail(MI_ERR_INVALID, "bad float_type %d", p.float_type);
    if (p.min_size < 1) return fail(MI_ERR_INVALID, "min_size must be >= 1");
    if (p.kernel_size < 1 || p.kernel_size > 12)
        return fail(MI_ERR_INVALID, "kernel_size must be in [1, 12] (base window <= 11x11)");
    if (p.impl < MI_IMPL_AUTO || p.impl > MI_IMPL_TILED) return fail(MI_ERR_INVALID, "bad impl %d", p.impl);
    if (p.arith != MI_ARITH_EXACT && p.arith != MI_ARITH_SEPARABLE) return fail(MI_ERR_INVALID, "bad arith %d", p.arith);
    if (p.arith == MI_ARITH_SEPARABLE && p.float_type != MI_F32)
        return fail(MI_ERR_INVALID, "MI_ARITH_SEPARABLE needs float_type MI_F32");
    if (p.pair_levels < 0 || p.pair_levels > 3) return fail(MI_ERR_INVALID, "pair_levels must be 0 (automatic), 1 (pairs from level 0 on), 2 (none) or 3 (pairs from level 1 on)");
    int ndev = 0;
    int rc = mi_device_count(&ndev);
    if (rc) return rc;
    if (ndev == 0) return fail(MI_ERR_NO_DEVICE, "no HIP device visible");
    if (p.device < 0 || p.device >= ndev) return fail(MI_ERR_INVALID, "device %d out of range (have %d)", p.device, ndev);
    MI_HIP(hipSetDevice(p.device));

    mi_stack* s = new mi_stack();
    s->p = p;
    if (s->p.impl == MI_IMPL_AUTO) s->p.impl = tiled_available() ? MI_IMPL_TILED : MI_IMPL_SIMPLE;
    // levels = int(log2(min(h,w)/min_size)), pyramid.py:165; stop when a side < 4, :129-130
    {
        double r = (double)(p.height < p.width ? p.height : p.width) / (double)p.min_size;
        int req = r >= 1.0 ? (int)std::log2(r) : 0;
        // guard against log2 rounding at exact powers of two (np.log2 is exact there)
        while (req > 0 && (double)(1 << req) > r) --req;
        while ((double)(1 << (req + 1)) <= r) ++req;
        int h = p.height, w = p.width;
        s->lh.push_back(h);
        s->lw.push_back(w);
        for (int i = 0; i < req; ++i) {
            h = (h + 1) / 2;
            w = (w + 1) / 2;
            if ((h < w ? h : w) < 4) break;
            s->lh.push_back(h);
            s->lw.push_back(w);
        }
        s->L = (int)s->lh.size() - 1;
    }
    if (s->L == 0) s->p.impl = MI_IMPL_SIMPLE;  // base-only stacks (tiny frames): one frame at a time
    s->f64 = p.float_type == MI_F64;
    if (s->f64) s->p.impl = MI_IMPL_SIMPLE;     // the precision option runs the one-frame-at-a-time formulation
    {
        double a = p.gen_kernel;
        double k[5] = {0.25 - a / 2.0, 0.25, a, 0.25, 0.25 - a / 2.0};
        for (int i = 0; i < 5; ++i)
            for (int j = 0; j < 5; ++j) {
                s->K.k[i * 5 + j] = (float)(k[i] * k[j]);
                s->Kd.k[i * 5 + j] = k[i] * k[j];
            }
        for (int i = 0; i < 3; ++i) s->k1d[i] = (float)k[i];
        s->sep = p.arith == MI_ARITH_SEPARABLE;
        s->mfma_ok = red_taps(a, s->rk);
    }
    s->pad = (p.kernel_size - 1) / 2;
    s->nlevels_hist = p.out_dtype == MI_U8 ? 256 : 65536;
    s->maxv = p.out_dtype == MI_U8 ? 255.f : 65535.f;

#define TRY(x)                     \
    do {                           \
        rc = (x);                  \
        if (rc) {                  \
            mi_stack_destroy(s);   \
            return rc;             \
        }                          \
    } while (0)
    hipError_t he = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (he != hipSuccess) {
        delete s;
        return fail(MI_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(he));
    }
    const int L = s->L;
    const size_t P0 = (size_t)p.height * p.width;
    const size_t fm = s->f64 ? 2 : 1;   // float_type-sized buffers are allocated in units of float
    TRY(dev_alloc(s, &s->frame_dev, P0 * 3 * dtype_size(p.in_dtype)));
    s->G.assign(L + 1, nullptr);
    s->bestE.assign(L, nullptr);
    s->bestLap.assign(L, nullptr);
    s->bestIdx.assign(L, nullptr);
    for (int l = 1; l <= L; ++l) TRY(dev_alloc_t(s, &s->G[l], (size_t)s->lh[l] * s->lw[l] * 3 * fm));
    std::vector<size_t> seg_off(L + 2, 0);
    if (!s->f64) {
        size_t off = 0;
        for (int k = 0; k < L + 2; ++k) {
            seg_off[k] = off;
            const int lv = k < L ? k : L;
            off += (((size_t)s->lh[lv] * s->lw[lv]) + 63) / 64 * 64;
        }
        s->slab_px = off;
        TRY(dev_alloc_t(s, &s->slabE, off));
        TRY(dev_alloc_t(s, &s->slabI, off));
        TRY(dev_alloc_t(s, &s->slabL, off * 3));
        MI_HIP(hipMemset(s->slabE, 0, off * 4));
        MI_HIP(hipMemset(s->slabI, 0, off * 4));
        MI_HIP(hipMemset(s->slabL, 0, off * 12));
    }
    for (int l = 0; l < L; ++l) {
        size_t np = (size_t)s->lh[l] * s->lw[l];
        if (!s->f64) {
            s->bestE[l] = s->slabE + seg_off[l];
            s->bestLap[l] = s->slabL + seg_off[l] * 3;
            s->bestIdx[l] = s->slabI + seg_off[l];
            continue;
        }
        TRY(dev_alloc_t(s, &s->bestE[l], np));
        TRY(dev_alloc_t(s, &s->bestLap[l], np * 3 * fm));
        TRY(dev_alloc_t(s, &s->bestIdx[l], np));
    }
    if (s->p.impl == MI_IMPL_SIMPLE) {
        TRY(dev_alloc_t(s, &s->lap_tmp, P0 * 3 * fm));
        TRY(dev_alloc_t(s, &s->q_tmp, P0));
    }
    {
        size_t nb = (size_t)s->lh[L] * s->lw[L];
        TRY(dev_alloc_t(s, &s->lev, nb));
        TRY(dev_alloc_t(s, &s->cnt, (size_t)s->nlevels_hist));
        TRY(dev_alloc_t(s, &s->logp, (size_t)s->nlevels_hist * fm));
        if (!s->f64) {
            s->bEnt = s->slabE + seg_off[L];      s->bDev = s->slabE + seg_off[L + 1];
            s->idxE = s->slabI + seg_off[L];      s->idxD = s->slabI + seg_off[L + 1];
            s->baseE = s->slabL + seg_off[L] * 3; s->baseD = s->slabL + seg_off[L + 1] * 3;
        } else {
            TRY(dev_alloc_t(s, &s->bEnt, nb * fm));
            TRY(dev_alloc_t(s, &s->bDev, nb * fm));
            TRY(dev_alloc_t(s, &s->idxE, nb));
            TRY(dev_alloc_t(s, &s->idxD, nb));
            TRY(dev_alloc_t(s, &s->baseE, nb * 3 * fm));
            TRY(dev_alloc_t(s, &s->baseD, nb * 3 * fm));
        }
        TRY(dev_alloc_t(s, &s->fusedBase, nb * 3 * fm));
    }
    TRY(dev_alloc_t(s, &s->colA, P0 * 3 * fm));
    TRY(dev_alloc_t(s, &s->colB, (L >= 2 ? (size_t)s->lh[1] * s->lw[1] * 3 : 1) * fm));
    TRY(dev_alloc_t(s, &s->clipped, P0 * 3 * fm));
    TRY(dev_alloc(s, &s->out_dev, P0 * 3 * dtype_size(p.out_dtype)));
    TRY(tiled_create(s));
#undef TRY
    *out = s;
    return MI_OK;
}

void mi_stack_destroy(mi_stack_t* s) {
    if (!s) return;
    (void)hipSetDevice(s->p.device);
    (void)tiled_sync_all(s);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    tiled_destroy(s);
    for (auto& r : s->recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (auto e : s->ev_pool) (void)hipEventDestroy(e);
    for (void* p : s->allocs) (void)hipFree(p);
    if (s->aux) (void)hipStreamDestroy(s->aux);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

int mi_stack_reset(mi_stack_t* s) {
    int rc = check_handle(s);
    if (rc) return rc;
    MI_HIP(hipSetDevice(s->p.device));
    rc = tiled_sync_all(s);
    if (rc) return rc;
    MI_HIP(hipStreamSynchronize(s->stream));
    rc = prof_drain(s);
    if (rc) return rc;
    s->n_pushed = 0;
    s->finished = false;
    s->idx_exported = 0;
    return tiled_reset(s);
}

int mi_stack_levels(const mi_stack_t* s, int* levels) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (!levels) return fail(MI_ERR_INVALID, "null levels");
    *levels = s->L;
    return MI_OK;
}

int mi_stack_level_shape(const mi_stack_t* s, int level, int* h, int* w) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (level < 0 || level > s->L || !h || !w) return fail(MI_ERR_INVALID, "bad level %d", level);
    *h = s->lh[level];
    *w = s->lw[level];
    return MI_OK;
}

int mi_stack_frames_pushed(const mi_stack_t* s, int* n) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (!n) return fail(MI_ERR_INVALID, "null n");
    *n = s->n_pushed + tiled_pending(s);
    return MI_OK;
}

int mi_stack_set_first_index(mi_stack_t* s, int first_global_index) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (s->n_pushed + tiled_pending(s) != 0) return fail(MI_ERR_STATE, "set_first_index after frames were pushed");
    s->first_index = first_global_index;
    return MI_OK;
}

int mi_stack_set_index_stride(mi_stack_t* s, int stride) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (stride < 1) return fail(MI_ERR_INVALID, "index stride must be >= 1");
    if (s->n_pushed + tiled_pending(s) != 0) return fail(MI_ERR_STATE, "set_index_stride after frames were pushed");
    s->index_stride = stride;
    return MI_OK;
}

// The kernels number the frames of a handle consecutively from first_index; with an index stride the stored winner indices
// are rewritten once, in place, into the global numbering (first + k * stride), level by level: the level's bit in
// idx_exported is set, and the handle takes no more frames until it is reset (the payload passes look winners up by the
// consecutive numbers).
static int export_level_indices(mi_stack* s, int level) {
    if (s->index_stride == 1 || ((s->idx_exported >> level) & 1ull)) return MI_OK;
    int32_t* idx = level < s->L ? s->bestIdx[level] : level == s->L ? s->idxE : s->idxD;
    const size_t n = (size_t)s->lh[std::min(level, s->L)] * s->lw[std::min(level, s->L)];
    if (!s->aux) MI_HIP(hipStreamCreateWithFlags(&s->aux, hipStreamNonBlocking));
    hipLaunchKernelGGL(idx_export, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->aux, idx, n, s->first_index, s->index_stride);
    MI_HIP(hipGetLastError());
    s->idx_exported |= 1ull << level;
    return MI_OK;
}

int mi_stack_export_indices(mi_stack_t* s, int level) {
    int rc = check_handle(s);
    if (rc) return rc;
    MI_HIP(hipSetDevice(s->p.device));
    if (level < -1 || level > s->L + 1) return fail(MI_ERR_INVALID, "bad level %d", level);
    if ((rc = tiled_flush(s))) return rc;
    // the level's state must be final: level 0 is, once its last payload pass is through; everything else needs all of it
    if (level == 0) rc = tiled_sync_level0(s);
    else { rc = tiled_sync_all(s); if (!rc) MI_HIP(hipStreamSynchronize(s->stream)); }
    if (rc) return rc;
    for (int l = (level < 0 ? 0 : level); l <= (level < 0 ? s->L + 1 : level); ++l)
        if ((rc = export_level_indices(s, l))) return rc;
    if (s->aux) MI_HIP(hipStreamSynchronize(s->aux));
    return MI_OK;
}

int mi_stack_push_frames_device(mi_stack_t* s, const void* dev_frames, int n, size_t frame_stride_bytes) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (s->idx_exported) return fail(MI_ERR_STATE, "the winner indices were exported (index stride > 1): reset the handle before pushing more frames");
    if (!dev_frames || n < 0) return fail(MI_ERR_INVALID, "bad frames argument");
    if (s->finished) return fail(MI_ERR_STATE, "push after finish; call mi_stack_reset first");
    size_t fb = (size_t)s->p.height * s->p.width * 3 * dtype_size(s->p.in_dtype);
    if (frame_stride_bytes == 0) frame_stride_bytes = fb;
    if (frame_stride_bytes < fb) return fail(MI_ERR_INVALID, "frame stride smaller than a frame");
    MI_HIP(hipSetDevice(s->p.device));
    return dispatch_push(s, dev_frames, n, frame_stride_bytes);
}

int mi_stack_push_frame(mi_stack_t* s, const void* host_bgr, size_t row_stride_bytes) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (s->idx_exported) return fail(MI_ERR_STATE, "the winner indices were exported (index stride > 1): reset the handle before pushing more frames");
    if (!host_bgr) return fail(MI_ERR_INVALID, "null frame");
    if (s->finished) return fail(MI_ERR_STATE, "push after finish; call mi_stack_reset first");
    MI_HIP(hipSetDevice(s->p.device));
    return tiled_push_host(s, host_bgr, row_stride_bytes);
}

int mi_stack_push_frame_pinned(mi_stack_t* s, const void* host_bgr, size_t row_stride_bytes) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (!host_bgr) return fail(MI_ERR_INVALID, "null frame");
    if (s->finished) return fail(MI_ERR_STATE, "push after finish; call mi_stack_reset first");
    if (s->idx_exported) return fail(MI_ERR_STATE, "the winner indices were exported (index stride > 1): reset the handle before pushing more frames");
    if (s->p.impl != MI_IMPL_TILED || s->f64) return mi_stack_push_frame(s, host_bgr, row_stride_bytes);   // synchronous paths
    MI_HIP(hipSetDevice(s->p.device));
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, host_bgr) != hipSuccess || at.type != hipMemoryTypeHost) {
        (void)hipGetLastError();
        return fail(MI_ERR_INVALID, "mi_stack_push_frame_pinned: the frame is not in pinned host memory (mi_host_alloc / mi_host_register)");
    }
    return tiled_push_host(s, host_bgr, row_stride_bytes, true);
}

int mi_stack_wait_uploads(mi_stack_t* s, int max_outstanding) {
    int rc = check_handle(s);
    if (rc) return rc;
    MI_HIP(hipSetDevice(s->p.device));
    return tiled_wait_uploads(s, max_outstanding);
}

int mi_host_alloc(void** ptr, size_t bytes) {
    if (!ptr || !bytes) return fail(MI_ERR_INVALID, "bad argument");
    MI_HIP(hipHostMalloc(ptr, bytes, hipHostMallocPortable));
    return MI_OK;
}

int mi_host_free(void* ptr) {
    if (ptr) MI_HIP(hipHostFree(ptr));
    return MI_OK;
}

int mi_host_register(void* ptr, size_t bytes) {
    if (!ptr || !bytes) return fail(MI_ERR_INVALID, "bad argument");
    MI_HIP(hipHostRegister(ptr, bytes, hipHostRegisterPortable));
    return MI_OK;
}

int mi_host_unregister(void* ptr) {
    if (ptr) MI_HIP(hipHostUnregister(ptr));
    return MI_OK;
}

int mi_stack_sync(mi_stack_t* s) {
    int rc = check_handle(s);
    if (rc) return rc;
    MI_HIP(hipSetDevice(s->p.device));
    rc = tiled_sync_all(s);
    if (rc) return rc;
    MI_HIP(hipStreamSynchronize(s->stream));
    return MI_OK;
}

int mi_stack_sync_level(mi_stack_t* s, int level) {
    int rc = check_handle(s);
    if (rc) return rc;
    MI_HIP(hipSetDevice(s->p.device));
    rc = tiled_flush(s);
    if (rc) return rc;
    if (level != 0) return mi_stack_sync(s);
    return tiled_sync_level0(s);
}

int mi_stack_finish_device(mi_stack_t* s, void* dev_out) {
    int rc = check_handle(s);
    if (rc) return rc;
    MI_HIP(hipSetDevice(s->p.device));
    rc = tiled_flush(s);
    if (rc) return rc;
    if (s->n_pushed == 0) return fail(MI_ERR_STATE, "finish with no frames pushed");
    if (s->f64) rc = s->p.use_fma ? finish_f64<true>(s) : finish_f64<false>(s);
    else rc = s->p.use_fma ? finish_impl<true>(s) : finish_impl<false>(s);
    if (rc) return rc;
    s->finished = true;
    if (dev_out) {
        size_t nb = (size_t)s->p.height * s->p.width * 3 * dtype_size(s->p.out_dtype);
        MI_HIP(hipMemcpyAsync(dev_out, s->out_dev, nb, hipMemcpyDeviceToDevice, s->stream));
    }
    return MI_OK;
}

int mi_stack_finish(mi_stack_t* s, void* host_out, size_t row_stride_bytes) {
    int rc = mi_stack_finish_device(s, nullptr);
    if (rc) return rc;
    if (!host_out) return fail(MI_ERR_INVALID, "null output buffer");
    size_t rb = (size_t)s->p.width * 3 * dtype_size(s->p.out_dtype);
    if (row_stride_bytes == 0) row_stride_bytes = rb;
    if (row_stride_bytes < rb) return fail(MI_ERR_INVALID, "row stride smaller than a row");
    MI_HIP(hipMemcpy2DAsync(host_out, row_stride_bytes, s->out_dev, rb, rb, s->p.height,
                            hipMemcpyDeviceToHost, s->stream));
    MI_HIP(hipStreamSynchronize(s->stream));
    return MI_OK;
}

int mi_stack_get_level(mi_stack_t* s, int level, int what, void* host_out, size_t out_bytes) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (!host_out) return fail(MI_ERR_INVALID, "null output");
    MI_HIP(hipSetDevice(s->p.device));
    rc = tiled_flush(s);
    if (rc) return rc;
    const int L = s->L;
    const void* src = nullptr;
    size_t bytes = 0;
    auto np = [&](int l) { return (size_t)s->lh[l] * s->lw[l]; };
    const size_t fb = s->f64 ? 8 : 4;   // bytes of a float_type element
    switch (what) {
        case MI_TAP_GAUSS:
            if (level < 1 || level > L) return fail(MI_ERR_INVALID, "MI_TAP_GAUSS: level in [1, %d]", L);
            src = tiled_last_gauss(s, level);
            bytes = np(level) * 3 * fb;
            break;
        case MI_TAP_FUSED_LAP:
            if (level < 0 || level >= L) return fail(MI_ERR_INVALID, "bad level %d", level);
            src = s->bestLap[level]; bytes = np(level) * 3 * fb; break;
        case MI_TAP_ENERGY:
            if (level < 0 || level >= L) return fail(MI_ERR_INVALID, "bad level %d", level);
            src = s->bestE[level]; bytes = np(level) * 4; break;
        case MI_TAP_INDEX:
            if (level < 0 || level >= L) return fail(MI_ERR_INVALID, "bad level %d", level);
            if ((rc = mi_stack_export_indices(s, level))) return rc;   // (index stride > 1: the tap shows global frame numbers)
            src = s->bestIdx[level]; bytes = np(level) * 4; break;
        case MI_TAP_FUSED_BASE:
            if (!s->finished) return fail(MI_ERR_STATE, "fused base is available after finish");
            src = s->fusedBase; bytes = np(L) * 3 * fb; break;
        case MI_TAP_BASE_IDX_E:
            if ((rc = mi_stack_export_indices(s, L))) return rc;
            src = s->idxE; bytes = np(L) * 4; break;
        case MI_TAP_BASE_IDX_D:
            if ((rc = mi_stack_export_indices(s, L + 1))) return rc;
            src = s->idxD; bytes = np(L) * 4; break;
        case MI_TAP_BASE_ENT: src = s->bEnt; bytes = np(L) * fb; break;
        case MI_TAP_BASE_DEV: src = s->bDev; bytes = np(L) * fb; break;
        case MI_TAP_COLLAPSED:
            if (!s->finished) return fail(MI_ERR_STATE, "collapsed image is available after finish");
            if (!s->f64 && !s->have_clipped) {   // finish fused the finest collapse step with the cast: redo it unfused
                const dim3 blk(64, 4);
                if (s->sep)
                    hipLaunchKernelGGL((collapse_sep<float>), grid2d(cdiv(s->lw[0], 2), cdiv(s->lh[0], 2), blk), blk, 0, s->stream, s->collapse_src,
                                       s->lh[1], s->lw[1], (const float*)s->bestLap[0], s->lh[0], s->lw[0], s->maxv, s->colA,
                                       s->k1d[0], s->k1d[1], s->k1d[2]);
                else if (s->p.use_fma)
                    hipLaunchKernelGGL((collapse_simple<true>), grid2d(s->lw[0], s->lh[0], blk), blk, 0, s->stream,
                                       s->collapse_src, s->lh[1], s->lw[1], s->bestLap[0], s->lh[0], s->lw[0], s->colA, s->K);
                else
                    hipLaunchKernelGGL((collapse_simple<false>), grid2d(s->lw[0], s->lh[0], blk), blk, 0, s->stream,
                                       s->collapse_src, s->lh[1], s->lw[1], s->bestLap[0], s->lh[0], s->lw[0], s->colA, s->K);
                const size_t n = np(0) * 3;
                const dim3 g((unsigned)((n + 255) / 256));
                if (s->p.out_dtype == MI_U8)
                    hipLaunchKernelGGL((finalize_cast<uint8_t>), g, dim3(256), 0, s->stream, s->colA, n, s->maxv, s->clipped,
                                       (uint8_t*)s->out_dev);
                else
                    hipLaunchKernelGGL((finalize_cast<uint16_t>), g, dim3(256), 0, s->stream, s->colA, n, s->maxv, s->clipped,
                                       (uint16_t*)s->out_dev);
                MI_HIP(hipGetLastError());
                s->have_clipped = true;
            }
            src = s->clipped; bytes = np(0) * 3 * fb; break;
        default: return fail(MI_ERR_INVALID, "unknown tap %d", what);
    }
    if (out_bytes < bytes) return fail(MI_ERR_INVALID, "output buffer too small: need %zu bytes", bytes);
    MI_HIP(hipMemcpyAsync(host_out, src, bytes, hipMemcpyDeviceToHost, s->stream));
    MI_HIP(hipStreamSynchronize(s->stream));
    return MI_OK;
}

int mi_stack_state(mi_stack_t* s, int level, void** dev_energy, void** dev_lap, void** dev_index, size_t* npixels) {
    int rc = check_handle(s);
    if (rc) return rc;
    MI_HIP(hipSetDevice(s->p.device));
    rc = tiled_flush(s);
    if (rc) return rc;
    // hand-off point to foreign streams (RCCL, hipMemcpy on the null stream ...): everything this
    // handle enqueued on its own non-blocking streams must have finished
    rc = tiled_sync_all(s);
    if (rc) return rc;
    MI_HIP(hipStreamSynchronize(s->stream));
    if (s->f64) return fail(MI_ERR_UNSUPPORTED, "the frame-sharded combine works on float-32 state only");
    const int L = s->L;
    void *e = nullptr, *l = nullptr, *i = nullptr;
    size_t n = 0;
    if (level == -1) {   // everything at once: the contiguous slabs (padding pixels included)
        e = s->slabE; l = s->slabL; i = s->slabI; n = s->slab_px;
    } else if (level >= 0 && level < L) {
        e = s->bestE[level]; l = s->bestLap[level]; i = s->bestIdx[level];
        n = (size_t)s->lh[level] * s->lw[level];
    } else if (level == L) {
        e = s->bEnt; l = s->baseE; i = s->idxE; n = (size_t)s->lh[L] * s->lw[L];
    } else if (level == L + 1) {
        e = s->bDev; l = s->baseD; i = s->idxD; n = (size_t)s->lh[L] * s->lw[L];
    } else
        return fail(MI_ERR_INVALID, "bad level %d", level);
    if (dev_energy) *dev_energy = e;
    if (dev_lap) *dev_lap = l;
    if (dev_index) *dev_index = i;
    if (npixels) *npixels = n;
    return MI_OK;
}

int mi_stack_stream(mi_stack_t* s, void** stream) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (!stream) return fail(MI_ERR_INVALID, "null stream out");
    *stream = (void*)s->stream;
    return MI_OK;
}

int mi_stack_profile(mi_stack_t* s, int enable) {
    int rc = check_handle(s);
    if (rc) return rc;
    MI_HIP(hipSetDevice(s->p.device));
    rc = prof_drain(s);
    if (rc) return rc;
    if (enable) {  // (re)enabling starts a fresh accumulation
        for (int k = 0; k < MI_PROF_KINDS; ++k) {
            s->prof_ms[k] = 0;
            s->prof_n[k] = 0;
            s->prof_bytes[k] = 0;
        }
    }
    s->prof = enable != 0;
    return MI_OK;
}

int mi_stack_profile_get(mi_stack_t* s, int kind, double* total_ms, int64_t* launches, double* algorithmic_bytes) {
    int rc = check_handle(s);
    if (rc) return rc;
    if (kind < 0 || kind >= MI_PROF_KINDS) return fail(MI_ERR_INVALID, "bad kind %d", kind);
    MI_HIP(hipSetDevice(s->p.device));
    rc = prof_drain(s);
    if (rc) return rc;
    if (total_ms) *total_ms = s->prof_ms[kind];
    if (launches) *launches = s->prof_n[kind];
    if (algorithmic_bytes) *algorithmic_bytes = s->prof_bytes[kind];
    return MI_OK;
}

int mi_combine_select(int device, void* stream, int n, const void* cand_e, const void* cand_lap,
                      const void* cand_idx, size_t npix, void* out_e, void* out_lap, void* out_idx) {
    if (n < 1 || !cand_e || !cand_lap || !out_e || !out_lap) return fail(MI_ERR_INVALID, "bad argument");
    if (npix == 0) return MI_OK;
    MI_HIP(hipSetDevice(device));
    hipLaunchKernelGGL(combine_select, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, n, (const float*)cand_e, (const float*)cand_lap,
                       (const int32_t*)cand_idx, npix, (float*)out_e, (float*)out_lap,
                       (int32_t*)out_idx);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

int mi_combine_winner(int device, void* stream, int n, const void* cand_e, size_t npix, void* win) {
    if (npix == 0) return MI_OK;   // an empty chunk (its buffers may be null pointers) is a no-op
    if (n < 1 || n > CB_MAXR || !cand_e || !win) return fail(MI_ERR_INVALID, "bad argument (1 <= ranks <= %d)", CB_MAXR);
    MI_HIP(hipSetDevice(device));
    hipLaunchKernelGGL(combine_winner, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n,
                       (const float*)cand_e, npix, (uint8_t*)win);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

int mi_combine_winner_idx(int device, void* stream, int n, const void* cand_e, const void* cand_idx, size_t npix, void* win) {
    if (npix == 0) return MI_OK;
    if (n < 1 || n > CB_MAXR || !cand_e || !cand_idx || !win) return fail(MI_ERR_INVALID, "bad argument (1 <= ranks <= %d)", CB_MAXR);
    MI_HIP(hipSetDevice(device));
    hipLaunchKernelGGL(combine_winner_idx, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n,
                       (const float*)cand_e, (const int32_t*)cand_idx, npix, (uint8_t*)win);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

size_t mi_combine_plan_bytes(size_t npix, int n_ranks) {
    return ((npix + CB_PX - 1) / CB_PX) * (size_t)n_ranks * sizeof(uint32_t) + CB_MAXR * sizeof(unsigned long long);
}

int mi_combine_plan(int device, void* stream, const void* win, size_t npix, int n_ranks, void* plan, int64_t* totals) {
    if (n_ranks < 1 || n_ranks > CB_MAXR || !totals || (npix && (!win || !plan))) return fail(MI_ERR_INVALID, "bad argument");
    if (npix == 0) {   // nothing to plan: no launch, no synchronisation
        for (int r = 0; r < n_ranks; ++r) totals[r] = 0;
        return MI_OK;
    }
    MI_HIP(hipSetDevice(device));
    const size_t nblocks = (npix + CB_PX - 1) / CB_PX;
    uint32_t* counts = (uint32_t*)plan;
    unsigned long long* tot = (unsigned long long*)(counts + nblocks * n_ranks);
    hipStream_t st = (hipStream_t)stream;
    if (nblocks) {
        switch ((n_ranks + 3) / 4) {
            case 1: hipLaunchKernelGGL(combine_count<1>, dim3((unsigned)nblocks), dim3(256), 0, st, (const uint8_t*)win, npix, n_ranks, counts); break;
            case 2: hipLaunchKernelGGL(combine_count<2>, dim3((unsigned)nblocks), dim3(256), 0, st, (const uint8_t*)win, npix, n_ranks, counts); break;
            default: hipLaunchKernelGGL(combine_count<CB_WORDS>, dim3((unsigned)nblocks), dim3(256), 0, st, (const uint8_t*)win, npix, n_ranks, counts); break;
        }
        hipLaunchKernelGGL(combine_scan, dim3(n_ranks), dim3(1024), 0, st, counts, (int)nblocks, n_ranks, tot);
        MI_HIP(hipGetLastError());
        unsigned long long h[CB_MAXR];
        MI_HIP(hipMemcpyAsync(h, tot, sizeof(unsigned long long) * n_ranks, hipMemcpyDeviceToHost, st));
        MI_HIP(hipStreamSynchronize(st));
        for (int r = 0; r < n_ranks; ++r) totals[r] = (int64_t)h[r];
    } else {
        for (int r = 0; r < n_ranks; ++r) totals[r] = 0;
    }
    return MI_OK;
}

int mi_combine_pack(int device, void* stream, const void* win, size_t npix, int n_ranks, int rank, const void* plan,
                    const void* src, int width, void* out) {
    if (npix == 0) return MI_OK;
    if (n_ranks < 1 || n_ranks > CB_MAXR || rank < 0 || rank >= n_ranks || !win || !plan || !src || !out || width < 1)
        return fail(MI_ERR_INVALID, "bad argument");
    MI_HIP(hipSetDevice(device));
    const dim3 grd((unsigned)((npix + CB_PX - 1) / CB_PX));
#define MI_CB_PACK(NW) hipLaunchKernelGGL((combine_move<true, NW>), grd, dim3(256), 0, (hipStream_t)stream, (const uint8_t*)win, npix, \
                                          n_ranks, rank, (const uint32_t*)plan, (const float*)src, (const float* const*)nullptr, width, (float*)out)
    switch ((n_ranks + 3) / 4) {
        case 1: MI_CB_PACK(1); break;
        case 2: MI_CB_PACK(2); break;
        default: MI_CB_PACK(CB_WORDS); break;
    }
#undef MI_CB_PACK
    MI_HIP(hipGetLastError());
    return MI_OK;
}

int mi_combine_unpack(int device, void* stream, const void* win, size_t npix, int n_ranks, int rank, const void* plan,
                      const void* const* dev_bufs, int width, void* dst) {
    if (npix == 0) return MI_OK;
    if (n_ranks < 1 || n_ranks > CB_MAXR || rank < 0 || rank >= n_ranks || !win || !plan || !dev_bufs || !dst || width < 1)
        return fail(MI_ERR_INVALID, "bad argument");
    MI_HIP(hipSetDevice(device));
    const dim3 grd((unsigned)((npix + CB_PX - 1) / CB_PX));
#define MI_CB_UNPACK(NW) hipLaunchKernelGGL((combine_move<false, NW>), grd, dim3(256), 0, (hipStream_t)stream, (const uint8_t*)win, npix, \
                                            n_ranks, rank, (const uint32_t*)plan, (const float*)nullptr, (const float* const*)dev_bufs, width, (float*)dst)
    switch ((n_ranks + 3) / 4) {
        case 1: MI_CB_UNPACK(1); break;
        case 2: MI_CB_UNPACK(2); break;
        default: MI_CB_UNPACK(CB_WORDS); break;
    }
#undef MI_CB_UNPACK
    MI_HIP(hipGetLastError());
    return MI_OK;
}

namespace {
// shared body of mi_warp_affine_device (M: 2x3, persp == false) and mi_warp_perspective_device (M: 3x3)
int warp_device_impl(int device, void* stream, const void* dev_src, void* dev_dst, void* dev_tmp, void* dev_mask, int height,
                     int width, int dtype, const double* M, bool persp, int border_mode, const double* border_value,
                     int blur_ksize, double blur_sigma) {
    if (!dev_src || !dev_dst || !M || height < 1 || width < 1) return fail(MI_ERR_INVALID, "bad argument");
    if (dtype != MI_U8 && dtype != MI_U16) return fail(MI_ERR_INVALID, "dtype must be MI_U8 or MI_U16");
    if (border_mode < 0 || border_mode > 2) return fail(MI_ERR_INVALID, "bad border_mode %d", border_mode);
    const bool blur = border_mode == 2;
    if (blur && (!dev_tmp || !dev_mask)) return fail(MI_ERR_INVALID, "border blur needs dev_tmp and dev_mask");
    if (blur && (blur_ksize < 1 || blur_ksize > 31 || !(blur_ksize & 1) || !(blur_sigma > 0)))
        return fail(MI_ERR_INVALID, "blur kernel size must be odd and <= 31, sigma > 0");
    MI_HIP(hipSetDevice(device));
    AffineArgs a{};
    PerspArgs pa{};
    if (persp) {
        // cv::invert of a 3x3 double matrix: cofactors times 1 / det [from memory]; singular -> zeros
        const double* m = M;
        double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
        d = d != 0.0 ? 1.0 / d : 0.0;
        pa.iM[0] = (m[4] * m[8] - m[5] * m[7]) * d; pa.iM[1] = (m[2] * m[7] - m[1] * m[8]) * d; pa.iM[2] = (m[1] * m[5] - m[2] * m[4]) * d;
        pa.iM[3] = (m[5] * m[6] - m[3] * m[8]) * d; pa.iM[4] = (m[0] * m[8] - m[2] * m[6]) * d; pa.iM[5] = (m[2] * m[3] - m[0] * m[5]) * d;
        pa.iM[6] = (m[3] * m[7] - m[4] * m[6]) * d; pa.iM[7] = (m[1] * m[6] - m[0] * m[7]) * d; pa.iM[8] = (m[0] * m[4] - m[1] * m[3]) * d;
        const int bh0 = height < 16 ? height : 16;
        pa.bw0 = 1024 / bh0 < width ? 1024 / bh0 : width;
    } else {
        invert_affine_host(M, a.iM);
    }
    a.h = height;
    a.w = width;
    a.mode = border_mode == 0 ? 0 : 1;
    const int hi = dtype == MI_U8 ? 255 : 65535;
    for (int c = 0; c < 3; ++c) a.border[c] = border_value ? round_sat(border_value[c], hi) : 0;
    GaussArgs g{};
    if (blur) {
        // OpenCV's fixed-point taps for 8- / 16-bit images (getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED,
        // restated in oracle/align_oracle.c::orc_gauss_kernel_fixed [from memory]): exp(x^2 * (-0.125 / sigma^2)) at
        // x = 2i - (n - 1), normalised in double; outer taps rounded half to even with the error carried inwards, the
        // centre tap takes the rest so that the sum is exactly 1 << bits
        g.ksize = blur_ksize;
        const int n = blur_ksize, n2 = (n - 1) / 2, bits = dtype == MI_U8 ? 8 : 16;
        const double sg = blur_sigma > 0 ? blur_sigma : n * 0.15 + 0.35;
        const double scale2x = -0.125 / (sg * sg);
        double t[16], sum = 0.0;
        for (int i = 0, x = 1 - n; i < n2; ++i, x += 2) {
            t[i] = std::exp((double)(x * x) * scale2x);
            sum += t[i];
        }
        sum = sum * 2.0 + 1.0;
        const double mul1 = 1.0 / sum, fixed_1 = (double)(1u << bits);
        int64_t acc = 0;
        double carry = 0.0;
        for (int i = 0; i < n2; ++i) {
            const double adj = t[i] * mul1 * fixed_1 + carry;
            const int64_t v = (int64_t)std::nearbyint(adj);
            carry = adj - (double)v;
            g.k[i] = g.k[n - 1 - i] = (uint32_t)v;
            acc += 2 * v;
        }
        g.k[n2] = (uint32_t)(((int64_t)1 << bits) - acc);
    }
    // The tile scratch (counter / bitmap / list) is cached per (device, stream): two host threads that warp on the SAME
    // stream -- the two step_process chains of pipeline._align_chains_device on the default stream -- must not interleave
    // their enqueue sequences (memset, warp marks tiles, bitmap -> list, blur, scatter), or one warp's memset lands between
    // the other's marks and its list.  One lock across the whole sequence; the stream then runs the sequences in order.
    // The lock is per (device, stream) -- the key of the scratch cache: warps on other GPUs or streams of the same process
    // (the threads of a multi-GPU job) neither serialise here nor wait behind another device's scratch re-allocation.
    static std::mutex map_mu;
    static std::map<std::pair<int, void*>, std::unique_ptr<std::mutex>> enqueue_mus;
    std::mutex* enqueue_mu;
    {
        std::lock_guard<std::mutex> lk(map_mu);
        auto& slot = enqueue_mus[{device, stream}];
        if (!slot) slot.reset(new std::mutex);
        enqueue_mu = slot.get();
    }
    std::lock_guard<std::mutex> lk(*enqueue_mu);
    if (dtype == MI_U8)
        return warp_launch<uint8_t>(device, (hipStream_t)stream, dev_src, dev_tmp, dev_dst, (uint8_t*)dev_mask, height, width, a,
                                    blur, g, persp ? &pa : nullptr);
    return warp_launch<uint16_t>(device, (hipStream_t)stream, dev_src, dev_tmp, dev_dst, (uint8_t*)dev_mask, height, width, a, blur,
                                 g, persp ? &pa : nullptr);
}

// host-buffer form of both
int warp_host_impl(int device, const void* host_src, void* host_dst, void* host_mask, int height, int width, int dtype,
                   const double* M, bool persp, int border_mode, const double* border_value, int blur_ksize, double blur_sigma) {
    if (!host_src || !host_dst) return fail(MI_ERR_INVALID, "null image");
    if (dtype != MI_U8 && dtype != MI_U16) return fail(MI_ERR_INVALID, "dtype must be MI_U8 or MI_U16");
    if (height < 1 || width < 1) return fail(MI_ERR_INVALID, "bad image size");
    int ndev = 0;
    int rc = mi_device_count(&ndev);
    if (rc) return rc;
    if (ndev == 0) return fail(MI_ERR_NO_DEVICE, "no HIP device visible");
    MI_HIP(hipSetDevice(device));
    const size_t nb = (size_t)height * width * 3 * dtype_size(dtype), np = (size_t)height * width;
    void *src = nullptr, *dst = nullptr, *tmp = nullptr, *mask = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(src); (void)hipFree(dst); (void)hipFree(tmp); (void)hipFree(mask);
    };
#define TRYH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(e_ == hipErrorOutOfMemory ? MI_ERR_NOMEM : MI_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_)); } } while (0)
    TRYH(hipMalloc(&src, nb));
    TRYH(hipMalloc(&dst, nb));
    TRYH(hipMalloc(&tmp, nb));
    TRYH(hipMalloc(&mask, np));
    TRYH(hipMemcpy(src, host_src, nb, hipMemcpyHostToDevice));
    rc = warp_device_impl(device, nullptr, src, dst, tmp, mask, height, width, dtype, M, persp, border_mode, border_value,
                          blur_ksize, blur_sigma);
    if (rc) { cleanup(); return rc; }
    TRYH(hipDeviceSynchronize());
    TRYH(hipMemcpy(host_dst, dst, nb, hipMemcpyDeviceToHost));
    if (host_mask) TRYH(hipMemcpy(host_mask, mask, np, hipMemcpyDeviceToHost));
#undef TRYH
    cleanup();
    return MI_OK;
}
}  // namespace

int mi_warp_affine_device(int device, void* stream, const void* dev_src, void* dev_dst, void* dev_tmp,
                          void* dev_mask, int height, int width, int dtype, const double* M,
                          int border_mode, const double* border_value, int blur_ksize, double blur_sigma) {
    return warp_device_impl(device, stream, dev_src, dev_dst, dev_tmp, dev_mask, height, width, dtype, M, false, border_mode,
                            border_value, blur_ksize, blur_sigma);
}

int mi_warp_perspective_device(int device, void* stream, const void* dev_src, void* dev_dst, void* dev_tmp,
                               void* dev_mask, int height, int width, int dtype, const double* M,
                               int border_mode, const double* border_value, int blur_ksize, double blur_sigma) {
    return warp_device_impl(device, stream, dev_src, dev_dst, dev_tmp, dev_mask, height, width, dtype, M, true, border_mode,
                            border_value, blur_ksize, blur_sigma);
}

int mi_warp_perspective(int device, const void* host_src, void* host_dst, void* host_mask, int height, int width,
                        int dtype, const double* M, int border_mode, const double* border_value,
                        int blur_ksize, double blur_sigma) {
    return warp_host_impl(device, host_src, host_dst, host_mask, height, width, dtype, M, true, border_mode, border_value,
                          blur_ksize, blur_sigma);
}

int mi_warp_affine(int device, const void* host_src, void* host_dst, void* host_mask, int height, int width,
                   int dtype, const double* M, int border_mode, const double* border_value,
                   int blur_ksize, double blur_sigma) {
    return warp_host_impl(device, host_src, host_dst, host_mask, height, width, dtype, M, false, border_mode, border_value,
                          blur_ksize, blur_sigma);
}

int mi_aligner_create(mi_aligner_t* out, int device, int height, int width, int dtype, int subsample,
                      int max_levels) {
    if (!out) return fail(MI_ERR_INVALID, "null argument");
    *out = nullptr;
    if (dtype != MI_U8 && dtype != MI_U16) return fail(MI_ERR_INVALID, "dtype must be MI_U8 or MI_U16");
    if (subsample < 1) return fail(MI_ERR_INVALID, "subsample must be >= 1");
    const int h = (height + subsample - 1) / subsample, w = (width + subsample - 1) / subsample;
    if (h < 16 || w < 16) return fail(MI_ERR_INVALID, "image too small for ECC");
    int ndev = 0;
    int rc = mi_device_count(&ndev);
    if (rc) return rc;
    if (ndev == 0) return fail(MI_ERR_NO_DEVICE, "no HIP device visible");
    MI_HIP(hipSetDevice(device));
    mi_aligner* al = new (std::nothrow) mi_aligner();
    if (!al) return fail(MI_ERR_NOMEM, "out of host memory");
    al->device = device; al->height = height; al->width = width; al->dtype = dtype; al->subsample = subsample;
    al->h = h; al->w = w;
    auto dalloc = [&](size_t bytes) -> void* {
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        al->bufs.push_back(p);
        return p;
    };
    bool ok = true;
    al->gray = (float*)dalloc((size_t)h * w * 4);
    ok = al->gray != nullptr;
    // pyramid geometry: halve while the short side stays >= 48 pixels
    {
        int lh = h, lw = w;
        for (int l = 0; l < (max_levels > 0 ? max_levels : 8); ++l) {
            al->lv.push_back({lh, lw, nullptr, nullptr});
            if ((lh < lw ? lh : lw) / 2 < 48) break;
            lh = (lh + 1) / 2;
            lw = (lw + 1) / 2;
        }
    }
    for (auto& L : al->lv) {
        L.tmpl = (float*)dalloc((size_t)L.h * L.w * 4);
        ok = ok && L.tmpl;
    }
    if (ok) ok = hipStreamCreateWithFlags(&al->own, hipStreamNonBlocking) == hipSuccess;
    if (!ok || aligner_reserve(al, 1)) {
        aligner_free(al);
        delete al;
        return fail(MI_ERR_NOMEM, "out of device memory");
    }
    *out = al;
    return MI_OK;
}

int mi_aligner_set_area_subsampling(mi_aligner_t al, int enable) {
    if (!al) return fail(MI_ERR_INVALID, "null handle");
    al->area = enable != 0;
    al->have_ref = false;   // the reference pyramid was built with the other rule
    // cv2.resize(INTER_AREA) by 1/s keeps round-half-even(dim / s) pixels where img[::s, ::s] keeps ceil(dim / s): the
    // sub-sampled grid, its centre and the pyramid geometry follow the rule in force (the buffers were allocated for the
    // larger of the two sizes)
    const int s = al->subsample;
    const int h = al->area ? (int)std::nearbyint(al->height * (1.0 / s)) : (al->height + s - 1) / s;
    const int w = al->area ? (int)std::nearbyint(al->width * (1.0 / s)) : (al->width + s - 1) / s;
    if (h < 16 || w < 16) return fail(MI_ERR_INVALID, "image too small for ECC");
    al->h = h;
    al->w = w;
    int lh = h, lw = w;
    for (size_t l = 0; l < al->lv.size(); ++l) {
        al->lv[l].h = lh;
        al->lv[l].w = lw;
        lh = (lh + 1) / 2;
        lw = (lw + 1) / 2;
    }
    return MI_OK;
}

int mi_aligner_destroy(mi_aligner_t al) {
    if (!al) return MI_OK;
    (void)hipSetDevice(al->device);
    (void)hipDeviceSynchronize();
    aligner_free(al);
    delete al;
    return MI_OK;
}

int mi_aligner_set_reference(mi_aligner_t al, void* stream, const void* dev_ref) {
    if (!al || !dev_ref) return fail(MI_ERR_INVALID, "null argument");
    MI_HIP(hipSetDevice(al->device));
    int rc = aligner_build(al, stream ? (hipStream_t)stream : al->own, dev_ref, true, 0);
    if (rc) return rc;
    al->have_ref = true;
    al->pc_ref_valid = false;
    return MI_OK;
}

int mi_aligner_set_phase_init(mi_aligner_t al, int enable) {
    if (!al) return fail(MI_ERR_INVALID, "null argument");
    al->phase_init = enable != 0;
    if (al->phase_init && al->pc_P == 0) {
        // the finest pyramid level with at most 512 pixels per side (the coarsest one if every level is larger)
        int lp = (int)al->lv.size() - 1;
        for (int l = 0; l < (int)al->lv.size(); ++l)
            if (std::max(al->lv[l].h, al->lv[l].w) <= 512) { lp = l; break; }
        al->pc_level = lp;
        al->pc_P = 1 << pc_log2(al->lv[lp].h);
        al->pc_Q = 1 << pc_log2(al->lv[lp].w);
        if (al->pc_P > PC_MAX_N || al->pc_Q > PC_MAX_N || al->pc_P < 2 || al->pc_Q < 2) {
            al->phase_init = false;
            al->pc_P = al->pc_Q = 0;
            return fail(MI_ERR_UNSUPPORTED, "phase correlation: no pyramid level between 2 and %d pixels per side", PC_MAX_N);
        }
    }
    return MI_OK;
}

int mi_phase_correlate_device(int device, void* stream, const void* dev_ref, const void* dev_mov, int height, int width,
                              double* out3) {
    if (!dev_ref || !dev_mov || !out3) return fail(MI_ERR_INVALID, "null argument");
    if (height < 2 || width < 2) return fail(MI_ERR_INVALID, "plane too small");
    const int P = 1 << pc_log2(height), Q = 1 << pc_log2(width);
    if (P > PC_MAX_N || Q > PC_MAX_N) return fail(MI_ERR_UNSUPPORTED, "phase correlation takes planes of at most %d pixels per side", PC_MAX_N);
    MI_HIP(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    float2 *a = nullptr, *b = nullptr;
    double* o = nullptr;
    MI_HIP(hipMalloc((void**)&a, sizeof(float2) * P * Q));
    if (hipMalloc((void**)&b, sizeof(float2) * P * Q) != hipSuccess || hipMalloc((void**)&o, 3 * sizeof(double)) != hipSuccess) {
        (void)hipFree(a);
        if (b) (void)hipFree(b);
        return fail(MI_ERR_NOMEM, "out of device memory");
    }
    pc_spectrum(st, (const float*)dev_ref, height, width, a, P, Q);
    pc_spectrum(st, (const float*)dev_mov, height, width, b, P, Q);
    pc_correlate(st, a, b, P, Q, o);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out3, o, 3 * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(a);
    (void)hipFree(b);
    (void)hipFree(o);
    if (e != hipSuccess) return fail(MI_ERR_HIP, "phase correlation: %s", hipGetErrorString(e));
    return MI_OK;
}

int mi_aligner_estimate_batch(mi_aligner_t al, void* stream, const void* const* dev_movs, int n, int max_iters,
                              double eps, double* M_out, double* cc_out, int* iters_out) {
    if (!al || !dev_movs || !M_out) return fail(MI_ERR_INVALID, "null argument");
    if (n < 1 || n > ECC_MAXF) return fail(MI_ERR_INVALID, "batch of %d frames (1..%d)", n, ECC_MAXF);
    if (!al->have_ref) return fail(MI_ERR_STATE, "mi_aligner_set_reference has not been called");
    for (int k = 0; k < n; ++k)
        if (!dev_movs[k]) return fail(MI_ERR_INVALID, "null frame %d", k);
    MI_HIP(hipSetDevice(al->device));
    hipStream_t st = stream ? (hipStream_t)stream : al->own;
    int rc = aligner_reserve(al, n);
    if (rc) return rc;
    if ((rc = aligner_build_batch(al, st, dev_movs, n))) return rc;   // gray + pyramids of the batch's frames
    return aligner_solve(al, st, n, max_iters, eps, M_out, cc_out, iters_out);
}

int mi_aligner_estimate_pairs(mi_aligner_t al, void* stream, const void* const* dev_frames, int n, const int* ref_of,
                              int max_iters, double eps, double* M_out, double* cc_out, int* iters_out) {
    if (!al || !dev_frames || !ref_of || !M_out) return fail(MI_ERR_INVALID, "null argument");
    if (n < 1 || n > ECC_MAXF) return fail(MI_ERR_INVALID, "batch of %d frames (1..%d)", n, ECC_MAXF);
    for (int k = 0; k < n; ++k) {
        if (!dev_frames[k]) return fail(MI_ERR_INVALID, "null frame %d", k);
        if (ref_of[k] < 0 || ref_of[k] >= n) return fail(MI_ERR_INVALID, "frame %d: reference %d is not a frame of the batch", k, ref_of[k]);
    }
    MI_HIP(hipSetDevice(al->device));
    hipStream_t st = stream ? (hipStream_t)stream : al->own;
    int rc = aligner_reserve(al, n);
    if (rc) return rc;
    if ((rc = aligner_build_batch(al, st, dev_frames, n))) return rc;   // gray + pyramids of the batch's frames
    return aligner_solve(al, st, n, max_iters, eps, M_out, cc_out, iters_out, nullptr, nullptr, -1, ref_of);
}

int mi_aligner_refine_batch(mi_aligner_t al, void* stream, const void* const* dev_movs, int n, const double* M_init, int levels,
                            int max_iters, double eps, double* M_out, double* cc_out, int* iters_out) {
    if (!al || !dev_movs || !M_out || !M_init) return fail(MI_ERR_INVALID, "null argument");
    if (n < 1 || n > ECC_MAXF) return fail(MI_ERR_INVALID, "batch of %d frames (1..%d)", n, ECC_MAXF);
    if (levels < 1) return fail(MI_ERR_INVALID, "levels must be >= 1");
    if (!al->have_ref) return fail(MI_ERR_STATE, "mi_aligner_set_reference has not been called");
    for (int k = 0; k < n; ++k)
        if (!dev_movs[k]) return fail(MI_ERR_INVALID, "null frame %d", k);
    MI_HIP(hipSetDevice(al->device));
    hipStream_t st = stream ? (hipStream_t)stream : al->own;
    int rc = aligner_reserve(al, n);
    if (rc) return rc;
    if ((rc = aligner_build_batch(al, st, dev_movs, n))) return rc;   // gray + pyramids of the batch's frames
    return aligner_solve(al, st, n, max_iters, eps, M_out, cc_out, iters_out, nullptr, M_init, levels - 1);
}

int mi_aligner_estimate_homography_batch(mi_aligner_t al, void* stream, const void* const* dev_movs, int n, int max_iters,
                                         double eps, double* M9_out, double* cc_out, int* iters_out) {
    if (!al || !dev_movs || !M9_out) return fail(MI_ERR_INVALID, "null argument");
    if (n < 1 || n > ECC_MAXF) return fail(MI_ERR_INVALID, "batch of %d frames (1..%d)", n, ECC_MAXF);
    if (!al->have_ref) return fail(MI_ERR_STATE, "mi_aligner_set_reference has not been called");
    for (int k = 0; k < n; ++k)
        if (!dev_movs[k]) return fail(MI_ERR_INVALID, "null frame %d", k);
    MI_HIP(hipSetDevice(al->device));
    hipStream_t st = stream ? (hipStream_t)stream : al->own;
    int rc = aligner_reserve(al, n);
    if (rc) return rc;
    if ((rc = aligner_build_batch(al, st, dev_movs, n))) return rc;   // gray + pyramids of the batch's frames
    return aligner_solve(al, st, n, max_iters, eps, nullptr, cc_out, iters_out, M9_out);
}

int mi_aligner_estimate(mi_aligner_t al, void* stream, const void* dev_mov, int max_iters, double eps,
                        double* M_out, double* cc_out, int* iters_out) {
    if (!dev_mov) return fail(MI_ERR_INVALID, "null argument");
    double cc = -2.0;
    int rc = mi_aligner_estimate_batch(al, stream, &dev_mov, 1, max_iters, eps, M_out, &cc, iters_out);
    if (rc) return rc;
    if (cc_out) *cc_out = cc;
    if (cc == -2.0) return fail(MI_ERR_STATE, "ECC: no overlap, constant image or degenerate transform");
    return MI_OK;
}

int mi_align_stack_device(mi_stack_t* st, mi_aligner_t al, const void* dev_frames, int n_frames, size_t frame_stride,
                          int ref_idx, const mi_align_stack_opts_t* o, const mi_balance_linear_opts_t* bal, void* dev_batches,
                          void* dev_tmp, void* dev_mask, double* M_out, double* cc_out, int* failed_frame) {
    int rc = check_handle(st);
    if (rc) return rc;
    if (!al || !dev_frames || !o || !dev_batches || !M_out || !cc_out) return fail(MI_ERR_INVALID, "null argument");
    if (n_frames < 1 || ref_idx < 0 || ref_idx >= n_frames) return fail(MI_ERR_INVALID, "bad frame count / reference index");
    if (o->ecc_batch < 1 || o->ecc_batch > ECC_MAXF || o->batch_frames < 1) return fail(MI_ERR_INVALID, "bad batch sizes");
    if (al->height != st->p.height || al->width != st->p.width || al->dtype != st->p.in_dtype || al->device != st->p.device)
        return fail(MI_ERR_INVALID, "the estimator and the stack handle are for different frames / devices");
    if (failed_frame) *failed_frame = -1;
    const int H = st->p.height, W = st->p.width, B = o->batch_frames;
    const size_t fb = (size_t)H * W * 3 * dtype_size(st->p.in_dtype);
    if (frame_stride == 0) frame_stride = fb;
    const char* frames = (const char*)dev_frames;
    MI_HIP(hipSetDevice(st->p.device));
    if ((rc = mi_aligner_set_reference(al, nullptr, frames + (size_t)ref_idx * frame_stride))) return rc;
    const bool persp = o->transform == 1;
    std::vector<double> est((size_t)n_frames * 9, 0.0);
    std::vector<char> have((size_t)n_frames, 0);
    for (int i = 0; i < n_frames; ++i) {
        for (int k = 0; k < 9; ++k) M_out[(size_t)i * 9 + k] = 0.0;
        cc_out[i] = 1.0;
    }
    // warp lanes (see mi_aligner): not with the in-place balance, whose histogram / table scratch is one per call
    hipStream_t side[2] = {nullptr, nullptr};
    tiled_side_streams(st, side);
    const int lanes = bal || !side[0] || !side[1] ? 1 : WARP_LANES;
    if (lanes > 1 && (rc = aligner_warp_lanes(al, fb, (size_t)H * W))) return rc;
    for (int j = 0; j < lanes - 1; ++j) al->wst[j] = side[j];
    unsigned lanes_used = 0;
    int cur = 0, filled = 0;
    auto flush = [&]() -> int {
        if (!filled) return MI_OK;
        for (int j = 1; j < lanes; ++j)   // the batch is complete when every lane's warps are
            if (lanes_used & (1u << j)) {
                MI_HIP(hipEventRecord(al->wev[j - 1], al->wst[j - 1]));
                MI_HIP(hipStreamWaitEvent(st->stream, al->wev[j - 1], 0));
            }
        lanes_used = 0;
        // no host synchronisation: the warps ran on the stacker's stream, where the level-0 kernels that read this batch
        // are enqueued next, and the stacker joins its side streams into that stream after every push
        int r = mi_stack_push_frames_device(st, (char*)dev_batches + (size_t)cur * B * fb, filled, fb);
        cur ^= 1;
        filled = 0;
        return r;
    };
    for (int i = 0; i < n_frames; ++i) {
        char* dst = (char*)dev_batches + ((size_t)cur * B + filled) * fb;
        if (filled == 0 && lanes > 1) MI_HIP(hipEventRecord(al->wstart, st->stream));   // this batch buffer is free from here on
        if (i == ref_idx) {
            MI_HIP(hipMemcpyAsync(dst, frames + (size_t)i * frame_stride, fb, hipMemcpyDeviceToDevice, st->stream));
        } else {
            if (!have[i]) {   // the next ecc_batch moving frames in one batched Gauss-Newton (the estimator's own stream)
                const void* ptrs[ECC_MAXF];
                int idx[ECC_MAXF], nb = 0;
                for (int k = i; k < n_frames && nb < o->ecc_batch; ++k)
                    if (k != ref_idx) { idx[nb] = k; ptrs[nb++] = frames + (size_t)k * frame_stride; }
                double Ms[ECC_MAXF * 9], ccs[ECC_MAXF];
                int its[ECC_MAXF];
                if (persp) {   // ALIGN_HOMOGRAPHY: the similarity refined to 8 degrees of freedom
                    if ((rc = mi_aligner_estimate_homography_batch(al, nullptr, ptrs, nb, o->max_iters, o->eps, Ms, ccs, its))) return rc;
                } else if ((rc = mi_aligner_estimate_batch(al, nullptr, ptrs, nb, o->max_iters, o->eps, Ms, ccs, its))) return rc;
                const int per = persp ? 9 : 6;
                for (int k = 0; k < nb; ++k) {
                    for (int q = 0; q < per; ++q) est[(size_t)idx[k] * 9 + q] = Ms[k * per + q];
                    cc_out[idx[k]] = ccs[k];
                    have[idx[k]] = 1;
                }
            }
            if (!(cc_out[i] >= o->min_correlation)) {
                if (failed_frame) *failed_frame = i;
                return fail(MI_ERR_ALIGNMENT, "frame %d: correlation %.3f < %.3f", i, cc_out[i], o->min_correlation);
            }
            double* m = M_out + (size_t)i * 9;
            for (int q = 0; q < (persp ? 9 : 6); ++q) m[q] = est[(size_t)i * 9 + q];
            // lane of this frame: the stacker's stream itself (0) or a side stream, which first waits for everything the
            // stacker's stream has done when this batch buffer started to fill (the push that last read it, joined)
            const int lane = lanes > 1 ? filled % lanes : 0;
            hipStream_t ws = st->stream;
            if (lane) {
                ws = al->wst[lane - 1];
                if (!(lanes_used & (1u << lane))) MI_HIP(hipStreamWaitEvent(ws, al->wstart, 0));
                lanes_used |= 1u << lane;
            }
            if ((rc = warp_device_impl(st->p.device, ws, frames + (size_t)i * frame_stride, dst, lane ? al->wtmp[lane - 1] : dev_tmp,
                                       lane ? al->wmask[lane - 1] : dev_mask, H, W, st->p.in_dtype, m, persp, o->border_mode,
                                       o->border_value, o->blur_ksize, o->blur_sigma)))
                return rc;
            if (bal) {   // LINEAR balance of the aligned frame, in place, no host round trip
                const size_t npx = (size_t)H * W;
                if (bal->cvt_to >= 0 && (rc = mi_cvt_color_device(st->p.device, st->stream, dst, dst, npx, st->p.in_dtype, bal->cvt_to))) return rc;
                if ((rc = mi_balance_linear_device(st->p.device, st->stream, dst, bal->dev_hist_scratch, bal->dev_lut, H, W,
                                                   st->p.in_dtype, bal->mode, bal->subsample, bal->fast, bal->mask_size, bal->lo,
                                                   bal->hi, bal->first_channel, bal->ref_means,
                                                   bal->dev_corr_out ? bal->dev_corr_out + (size_t)i * bal->ncorr : nullptr)))
                    return rc;
                if (bal->cvt_from >= 0 && (rc = mi_cvt_color_device(st->p.device, st->stream, dst, dst, npx, st->p.in_dtype, bal->cvt_from))) return rc;
            }
        }
        if (++filled == B && (rc = flush())) return rc;
    }
    return flush();
}

int mi_ecc_similarity(int device, const void* host_ref, const void* host_mov, int height, int width,
                      int dtype, int max_levels, int max_iters, double eps, double* M_out, double* cc_out,
                      int* iters_out) {
    if (!host_ref || !host_mov || !M_out) return fail(MI_ERR_INVALID, "null argument");
    mi_aligner_t al = nullptr;
    int rc = mi_aligner_create(&al, device, height, width, dtype, 1, max_levels);
    if (rc) return rc;
    const size_t nb = (size_t)height * width * 3 * dtype_size(dtype);
    void* raw = nullptr;
    if (hipMalloc(&raw, nb) != hipSuccess) {
        mi_aligner_destroy(al);
        return fail(MI_ERR_NOMEM, "out of device memory");
    }
    auto run = [&]() -> int {
        MI_HIP(hipMemcpy(raw, host_ref, nb, hipMemcpyHostToDevice));
        int r = mi_aligner_set_reference(al, nullptr, raw);
        if (r) return r;
        MI_HIP(hipStreamSynchronize(al->own));   // the pyramid is built before `raw` is reused
        MI_HIP(hipMemcpy(raw, host_mov, nb, hipMemcpyHostToDevice));
        return mi_aligner_estimate(al, nullptr, raw, max_iters, eps, M_out, cc_out, iters_out);
    };
    rc = run();
    (void)hipFree(raw);
    mi_aligner_destroy(al);
    return rc;
}

namespace {
// histogram of a device frame into dev_scratch (uint32 counts, channel-major), enqueued on `st`, no synchronisation
int hist_enqueue(int device, hipStream_t st, const void* dev_img, void* dev_scratch, int height, int width, int dtype, int mode,
                 int subsample, int fast, double mask_size) {
    if (dtype != MI_U8 && dtype != MI_U16) return fail(MI_ERR_INVALID, "dtype must be MI_U8 or MI_U16");
    if (height < 1 || width < 1 || subsample < 1 || (mode != 0 && mode != 1)) return fail(MI_ERR_INVALID, "bad argument");
    MI_HIP(hipSetDevice(device));
    const int nbins = dtype == MI_U8 ? 256 : 65536, nch = mode == 0 ? 3 : 1;
    HistArgs a{};
    a.img = dev_img; a.h = height; a.w = width; a.s = subsample; a.fast = fast ? 1 : 0; a.gray = mode;
    if (subsample == 1) { a.hs = height; a.ws = width; }
    else if (fast) { a.hs = cdiv(height, subsample); a.ws = cdiv(width, subsample); }
    else {   // cv2.resize's output size: round half to even of dim / s
        a.hs = (int)std::nearbyint((double)height * (1.0 / subsample));
        a.ws = (int)std::nearbyint((double)width * (1.0 / subsample));
    }
    if (a.hs < 1 || a.ws < 1) return fail(MI_ERR_INVALID, "image smaller than the sub-sampling factor");
    a.masked = mask_size > 0.0;
    if (a.masked) {   // balance.py:165-175 on the sub-sampled grid
        const double r = (double)(a.ws < a.hs ? a.ws : a.hs) * mask_size / 2.0;
        a.cx = (double)a.ws / 2.0; a.cy = (double)a.hs / 2.0; a.r2 = r * r;
    }
    a.counts = (uint32_t*)dev_scratch;
    MI_HIP(hipMemsetAsync(dev_scratch, 0, sizeof(uint32_t) * nch * nbins, st));
    const size_t total = (size_t)a.hs * a.ws;
    const unsigned nblk = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 8);
    if (dtype == MI_U8) hipLaunchKernelGGL(hist_u8, dim3(nblk), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(hist_u16, dim3(nblk), dim3(256), 0, st, a);
    MI_HIP(hipGetLastError());
    return MI_OK;
}
}  // namespace

int mi_histogram_device(int device, void* stream, const void* dev_img, void* dev_scratch, int height,
                        int width, int dtype, int mode, int subsample, int fast, double mask_size,
                        int64_t* counts) {
    if (!dev_img || !dev_scratch || !counts) return fail(MI_ERR_INVALID, "null argument");
    hipStream_t st = (hipStream_t)stream;
    int rc = hist_enqueue(device, st, dev_img, dev_scratch, height, width, dtype, mode, subsample, fast, mask_size);
    if (rc) return rc;
    const int nbins = dtype == MI_U8 ? 256 : 65536, nch = mode == 0 ? 3 : 1;
    std::vector<uint32_t> tmp((size_t)nch * nbins);
    MI_HIP(hipMemcpyAsync(tmp.data(), dev_scratch, tmp.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    MI_HIP(hipStreamSynchronize(st));
    for (size_t i = 0; i < tmp.size(); ++i) counts[i] = (int64_t)tmp[i];
    return MI_OK;
}

int mi_histogram_device_batch(int device, void* stream, const void* const* dev_imgs, int n, void* dev_scratch, int height,
                              int width, int dtype, int mode, int subsample, int fast, double mask_size, int64_t* counts) {
    if (!dev_imgs || !dev_scratch || !counts || n < 1) return fail(MI_ERR_INVALID, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int nbins = dtype == MI_U8 ? 256 : 65536, nch = mode == 0 ? 3 : 1;
    const size_t per = (size_t)nch * nbins;
    for (int k = 0; k < n; ++k) {
        if (!dev_imgs[k]) return fail(MI_ERR_INVALID, "null frame %d", k);
        int rc = hist_enqueue(device, st, dev_imgs[k], (uint32_t*)dev_scratch + (size_t)k * per, height, width, dtype, mode,
                              subsample, fast, mask_size);
        if (rc) return rc;
    }
    std::vector<uint32_t> tmp(per * n);   // one copy and ONE synchronisation for the whole batch
    MI_HIP(hipMemcpyAsync(tmp.data(), dev_scratch, tmp.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    MI_HIP(hipStreamSynchronize(st));
    for (size_t i = 0; i < tmp.size(); ++i) counts[i] = (int64_t)tmp[i];
    return MI_OK;
}

int mi_histogram(int device, const void* host_img, int height, int width, int dtype, int mode,
                 int subsample, int fast, double mask_size, int64_t* counts) {
    if (!host_img || !counts) return fail(MI_ERR_INVALID, "null argument");
    if (dtype != MI_U8 && dtype != MI_U16) return fail(MI_ERR_INVALID, "dtype must be MI_U8 or MI_U16");
    if (height < 1 || width < 1) return fail(MI_ERR_INVALID, "bad image size");
    int ndev = 0;
    int rc = mi_device_count(&ndev);
    if (rc) return rc;
    if (ndev == 0) return fail(MI_ERR_NO_DEVICE, "no HIP device visible");
    MI_HIP(hipSetDevice(device));
    const size_t nb = (size_t)height * width * 3 * dtype_size(dtype);
    void *img = nullptr, *scr = nullptr;
    auto cleanup = [&]() { (void)hipFree(img); (void)hipFree(scr); };
    if (hipMalloc(&img, nb) != hipSuccess || hipMalloc(&scr, sizeof(uint32_t) * 3 * 65536) != hipSuccess) {
        cleanup();
        return fail(MI_ERR_NOMEM, "out of device memory");
    }
    if (hipMemcpy(img, host_img, nb, hipMemcpyHostToDevice) != hipSuccess) { cleanup(); return fail(MI_ERR_HIP, "upload failed"); }
    rc = mi_histogram_device(device, nullptr, img, scr, height, width, dtype, mode, subsample, fast, mask_size, counts);
    cleanup();
    return rc;
}

int mi_apply_lut_device(int device, void* stream, const void* dev_src, void* dev_dst, size_t npixels,
                        int dtype, const void* dev_lut, int nlut) {
    if (!dev_src || !dev_dst || !dev_lut) return fail(MI_ERR_INVALID, "null argument");
    if (dtype != MI_U8 && dtype != MI_U16) return fail(MI_ERR_INVALID, "dtype must be MI_U8 or MI_U16");
    if (nlut != 1 && nlut != 3) return fail(MI_ERR_INVALID, "nlut must be 1 or 3");
    MI_HIP(hipSetDevice(device));
    const size_t n = npixels * 3;
    if (n == 0) return MI_OK;
    const unsigned nblk = (unsigned)std::min<size_t>((n / 12 + 255) / 256 + 1, 256 * 16);
    if (dtype == MI_U8) {
        if (((uintptr_t)dev_src | (uintptr_t)dev_dst) & 3) return fail(MI_ERR_INVALID, "images must be 4-byte aligned");
        hipLaunchKernelGGL(lut_apply_u8, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)dev_src,
                           (uint8_t*)dev_dst, n, (const uint8_t*)dev_lut, nlut);
    } else {
        hipLaunchKernelGGL(lut_apply_u16, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dev_src,
                           (uint16_t*)dev_dst, n, (const uint16_t*)dev_lut, nlut);
    }
    MI_HIP(hipGetLastError());
    return MI_OK;
}

int mi_balance_linear_device(int device, void* stream, void* dev_img, void* dev_hist_scratch, void* dev_lut, int height,
                             int width, int dtype, int mode, int subsample, int fast, double mask_size, int lo, int hi,
                             int first_channel, const double* ref_means, double* dev_corr_out) {
    if (!dev_img || !dev_hist_scratch || !dev_lut || !ref_means) return fail(MI_ERR_INVALID, "null argument");
    const int nbins = dtype == MI_U8 ? 256 : 65536, ntab = mode == 0 ? 3 : 1;
    if (first_channel < 0 || first_channel >= ntab) return fail(MI_ERR_INVALID, "bad first_channel %d", first_channel);
    if (lo < 0 || hi > nbins || lo >= hi) return fail(MI_ERR_INVALID, "bad intensity interval [%d, %d)", lo, hi);
    hipStream_t st = (hipStream_t)stream;
    int rc = hist_enqueue(device, st, dev_img, dev_hist_scratch, height, width, dtype, mode, subsample, fast, mask_size);
    if (rc) return rc;
    LinRef ref{};
    for (int c = 0; c < ntab - first_channel; ++c) ref.mean[c] = ref_means[c];
    if (dtype == MI_U8)
        hipLaunchKernelGGL((lut_linear_build<uint8_t>), dim3(ntab), dim3(256), 0, st, (const uint32_t*)dev_hist_scratch, nbins, lo, hi,
                           first_channel, ref, (uint8_t*)dev_lut, dev_corr_out);
    else
        hipLaunchKernelGGL((lut_linear_build<uint16_t>), dim3(ntab), dim3(256), 0, st, (const uint32_t*)dev_hist_scratch, nbins, lo,
                           hi, first_channel, ref, (uint16_t*)dev_lut, dev_corr_out);
    MI_HIP(hipGetLastError());
    return mi_apply_lut_device(device, stream, dev_img, dev_img, (size_t)height * width, dtype, dev_lut, ntab);
}

int mi_cvt_color_device(int device, void* stream, const void* dev_src, void* dev_dst, size_t npixels, int dtype, int code) {
    if (!dev_src || !dev_dst) return fail(MI_ERR_INVALID, "null argument");
    if (dtype != MI_U8)
        return fail(MI_ERR_UNSUPPORTED, "BGR <-> HSV / HLS is defined for 8-bit images only (as cv2.cvtColor: CV_8U / CV_32F)");
    if (code < CVT_BGR2HSV || code > CVT_HLS2BGR) return fail(MI_ERR_INVALID, "bad conversion code %d", code);
    MI_HIP(hipSetDevice(device));
    if (npixels == 0) return MI_OK;
    const dim3 grid((unsigned)std::min<size_t>((npixels + 255) / 256, 256 * 16)), blk(256);
    hipStream_t st = (hipStream_t)stream;
    const uint8_t* s = (const uint8_t*)dev_src;
    uint8_t* d = (uint8_t*)dev_dst;
    switch (code) {
        case CVT_BGR2HSV: hipLaunchKernelGGL((cvt_color_u8<CVT_BGR2HSV>), grid, blk, 0, st, s, d, npixels); break;
        case CVT_HSV2BGR: hipLaunchKernelGGL((cvt_color_u8<CVT_HSV2BGR>), grid, blk, 0, st, s, d, npixels); break;
        case CVT_BGR2HLS: hipLaunchKernelGGL((cvt_color_u8<CVT_BGR2HLS>), grid, blk, 0, st, s, d, npixels); break;
        default: hipLaunchKernelGGL((cvt_color_u8<CVT_HLS2BGR>), grid, blk, 0, st, s, d, npixels); break;
    }
    MI_HIP(hipGetLastError());
    return MI_OK;
}

int mi_cvt_color(int device, const void* host_src, void* host_dst, int height, int width, int dtype, int code) {
    if (!host_src || !host_dst || height < 1 || width < 1) return fail(MI_ERR_INVALID, "bad argument");
    if (dtype != MI_U8)
        return fail(MI_ERR_UNSUPPORTED, "BGR <-> HSV / HLS is defined for 8-bit images only (as cv2.cvtColor: CV_8U / CV_32F)");
    int ndev = 0;
    int rc = mi_device_count(&ndev);
    if (rc) return rc;
    if (ndev == 0) return fail(MI_ERR_NO_DEVICE, "no HIP device visible");
    MI_HIP(hipSetDevice(device));
    const size_t np = (size_t)height * width, nb = np * 3;
    void* buf = nullptr;
    MI_HIP(hipMalloc(&buf, nb));
    hipError_t e = hipMemcpy(buf, host_src, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = mi_cvt_color_device(device, nullptr, buf, buf, np, dtype, code);
        if (rc) { (void)hipFree(buf); return rc; }
        e = hipDeviceSynchronize();
    }
    if (e == hipSuccess) e = hipMemcpy(host_dst, buf, nb, hipMemcpyDeviceToHost);
    (void)hipFree(buf);
    if (e != hipSuccess) return fail(MI_ERR_HIP, "mi_cvt_color: %s", hipGetErrorString(e));
    return MI_OK;
}

int mi_apply_lut(int device, const void* host_src, void* host_dst, int height, int width, int dtype,
                 const void* host_lut, int nlut) {
    if (!host_src || !host_dst || !host_lut) return fail(MI_ERR_INVALID, "null argument");
    if (dtype != MI_U8 && dtype != MI_U16) return fail(MI_ERR_INVALID, "dtype must be MI_U8 or MI_U16");
    if (height < 1 || width < 1 || (nlut != 1 && nlut != 3)) return fail(MI_ERR_INVALID, "bad argument");
    int ndev = 0;
    int rc = mi_device_count(&ndev);
    if (rc) return rc;
    if (ndev == 0) return fail(MI_ERR_NO_DEVICE, "no HIP device visible");
    MI_HIP(hipSetDevice(device));
    const size_t np = (size_t)height * width, nb = np * 3 * dtype_size(dtype);
    const size_t lb = (size_t)nlut * (dtype == MI_U8 ? 256 : 65536) * dtype_size(dtype);
    void *src = nullptr, *dst = nullptr, *lut = nullptr;
    auto cleanup = [&]() { (void)hipFree(src); (void)hipFree(dst); (void)hipFree(lut); };
    if (hipMalloc(&src, nb) != hipSuccess || hipMalloc(&dst, nb) != hipSuccess || hipMalloc(&lut, lb) != hipSuccess) {
        cleanup();
        return fail(MI_ERR_NOMEM, "out of device memory");
    }
    if (hipMemcpy(src, host_src, nb, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(lut, host_lut, lb, hipMemcpyHostToDevice) != hipSuccess) { cleanup(); return fail(MI_ERR_HIP, "upload failed"); }
    rc = mi_apply_lut_device(device, nullptr, src, dst, np, dtype, lut, nlut);
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(MI_ERR_HIP, "LUT kernel failed");
    if (!rc && hipMemcpy(host_dst, dst, nb, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(MI_ERR_HIP, "download failed");
    cleanup();
    return rc;
}

int mi_synth_frames_device(int device, void* dev_out, int dtype, int height, int width,
                           int first_frame, int n_frames, int stack_size, uint32_t seed) {
    if (!dev_out || height < 1 || width < 1 || n_frames < 0 || stack_size < 1)
        return fail(MI_ERR_INVALID, "bad argument");
    MI_HIP(hipSetDevice(device));
    size_t per = (size_t)height * width * 3;
    // one launch per frame keeps the grid within 32-bit block counts
    for (int f = 0; f < n_frames; ++f) {
        dim3 g((unsigned)((per + 255) / 256));
        char* dst = (char*)dev_out + (size_t)f * per * dtype_size(dtype);
        switch (dtype) {
            case MI_U8:
                hipLaunchKernelGGL((synth_frames<uint8_t>), g, dim3(256), 0, 0, (uint8_t*)dst,
                                   height, width, first_frame + f, 1, stack_size, seed, 1);
                break;
            case MI_U16:
                hipLaunchKernelGGL((synth_frames<uint16_t>), g, dim3(256), 0, 0, (uint16_t*)dst,
                                   height, width, first_frame + f, 1, stack_size, seed, 257);
                break;
            case MI_F32:
                hipLaunchKernelGGL((synth_frames<float>), g, dim3(256), 0, 0, (float*)dst, height,
                                   width, first_frame + f, 1, stack_size, seed, 1);
                break;
            default: return fail(MI_ERR_INVALID, "bad dtype %d", dtype);
        }
    }
    MI_HIP(hipGetLastError());
    MI_HIP(hipDeviceSynchronize());
    return MI_OK;
}

}  // extern "C"

// DepthMapStack handle (mi_dmap_*)
#include "depthmap_host.hpp"
