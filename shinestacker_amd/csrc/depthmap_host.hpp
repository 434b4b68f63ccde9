// depthmap_host.hpp -- the mi_dmap_* handle: DepthMapStack (reference algorithms/depth_map.py:10-123) with the
// frames and their energy planes resident in HBM.  Included by capi.hip (one translation unit).
//
// The reference keeps N gray planes, N energy planes and N weight planes in host memory and reads every file
// twice; here the frame (input dtype) and one float_type plane per frame stay on the device -- 24 MP x 256 frames
// of 8-bit input are 18 + 25 GB of the 288 -- and everything else is per-frame scratch.
#pragma once
#include "kernels_depthmap.hpp"

struct mi_dmap {
    mi_dmap_params_t p{};
    double temperature = 0.1;        // p.temperature, or the exact double handed to mi_dmap_set_temperature (float-64 stacks)
    size_t esz = 1;                  // bytes per input element
    hipStream_t stream = nullptr;
    std::vector<void*> frames;       // device copies of the pushed frames (kept across reset for reuse)
    // F = float_type (float / double), W = type of the weights (float when the bilateral filter ran, else F).
    // Planes are raw allocations of fsz bytes per pixel; the typed views are made where they are used.
    size_t fsz = 4;
    bool f64 = false;
    std::vector<void*> en;           // energy (F) -> smoothed energy (W) -> relative weight (W) of frame i
    float* spare = nullptr;          // float-32: the plane the bilateral filter writes into (swapped with en[i])
    void *tmpA = nullptr, *tmpB = nullptr, *tmpC = nullptr;   // gray / row-blurred / blurred plane (F)
    void *tot = nullptr, *mx = nullptr;                       // running sum / maximum of the (smoothed) energies (W)
    void* scalF = nullptr;           // F: [0] global max, [1..2] min / max written by dm_normalise<F>
    float* scalf = nullptr;          // float: [0..1] min / max of the bilateral input, [2..3] scale / flat flag, [4] = 0
    float* fmm = nullptr;            // [DM_FMM_FRAMES][2] per-frame raw energy min (preset +inf) / max (preset 0):
                                     // written by dm_energy_lap5 so that finish needs no dm_normalise pass (float-32, smoothing on)
    std::vector<char> have_fmm;      // per frame: fmm[i] is valid
    float* lut = nullptr;
    int2* disc = nullptr;            // bilateral disc: (LDS patch offset, space weight bits) per tap
    int radius = 0, ntaps = 0;
    double color_coeff = 0.0;
    std::vector<int> lh, lw;         // level shapes, 0 .. levels-1
    std::vector<void*> G, W, B;      // Gaussian levels of the current frame (F x 3, G[0] unused), weights (W), blend sums (F x 3)
    void* out_dev = nullptr;
    mi::DmTapsT<float> taps{};
    mi::DmTapsT<double> tapsd{};
    mi::DmK2 k2{};
    int n = 0;
    bool finished = false;
    std::vector<void*> allocs;
};

namespace {

using namespace mi;

void dmap_free(mi_dmap* d) {
    for (void* p : d->allocs) (void)hipFree(p);
    d->allocs.clear();
    if (d->stream) (void)hipStreamDestroy(d->stream);
    d->stream = nullptr;
}

template <typename T>
int dmap_alloc(mi_dmap* d, T** p, size_t count) {
    void* q = nullptr;
    MI_HIP(hipMalloc(&q, count * sizeof(T) ? count * sizeof(T) : 1));
    d->allocs.push_back(q);
    *p = (T*)q;
    return MI_OK;
}

int dmap_alloc_bytes(mi_dmap* d, void** p, size_t bytes) {
    MI_HIP(hipMalloc(p, bytes ? bytes : 1));
    d->allocs.push_back(*p);
    return MI_OK;
}

// cv2.getGaussianKernel(ksize, 0, CV_32F / CV_64F) (oracle/depth_map_oracle.py gaussian_kernel)
template <typename F>
void dmap_gauss_taps(int ksize, DmTapsT<F>& t) {
    static const double s1[] = {1.0}, s3[] = {0.25, 0.5, 0.25}, s5[] = {0.0625, 0.25, 0.375, 0.25, 0.0625},
                        s7[] = {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125};
    t.ksize = ksize;
    const double* tab = ksize == 1 ? s1 : ksize == 3 ? s3 : ksize == 5 ? s5 : ksize == 7 ? s7 : nullptr;
    if (tab) {
        for (int i = 0; i < ksize; ++i) t.k[i] = (F)tab[i];
        return;
    }
    const double sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8, scale = -0.5 / (sigma * sigma);
    double sum = 0.0;
    for (int i = 0; i < ksize; ++i) {
        const double x = i - (ksize - 1) * 0.5;
        t.k[i] = (F)std::exp(scale * x * x);
        sum += (double)t.k[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < ksize; ++i) t.k[i] = (F)((double)t.k[i] * sum);
}

// cv2.getDerivKernels (Sobel family): ksize - order - 1 steps [1 1], `order` steps [-1 1]
void dmap_sobel_kernel(int order, int ksize, double* out) {
    std::vector<double> k{1.0};
    auto conv = [&](double a, double b) {
        std::vector<double> r(k.size() + 1, 0.0);
        for (size_t i = 0; i < k.size(); ++i) {
            r[i] += k[i] * a;
            r[i + 1] += k[i] * b;
        }
        k = r;
    };
    for (int i = 0; i < ksize - order - 1; ++i) conv(1.0, 1.0);
    for (int i = 0; i < order; ++i) conv(1.0, -1.0);
    for (int i = 0; i < ksize; ++i) out[i] = (order & 1) ? k[ksize - 1 - i] : k[i];
}

void dmap_laplacian_kernel(int ksize, DmK2& K) {
    if (ksize == 1 || ksize == 3) {
        static const double a1[9] = {0, 1, 0, 1, -4, 1, 0, 1, 0}, a3[9] = {2, 0, 2, 0, -8, 0, 2, 0, 2};
        K.ksize = 3;
        for (int i = 0; i < 9; ++i) K.k[i] = ksize == 1 ? a1[i] : a3[i];
        return;
    }
    double kd[15], ks[15];
    dmap_sobel_kernel(2, ksize, kd);
    dmap_sobel_kernel(0, ksize, ks);
    K.ksize = ksize;
    for (int i = 0; i < ksize; ++i)
        for (int j = 0; j < ksize; ++j) K.k[i * ksize + j] = ks[i] * kd[j] + kd[i] * ks[j];
}

inline dim3 dm_grid2(int h, int w) { return dim3(cdiv(w, 64), cdiv(h, 4)); }
inline dim3 dm_grid1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

// cv2.pyrDown: the LDS-tiled kernel when its two LDS images fit, else one thread per output
template <typename TSrc, int C, typename F>
void dm_pyrdown_launch(hipStream_t st, const TSrc* src, int h, int w, F* dst, int ho, int wo) {
    constexpr int TH = dm_pyrdown_tile_rows<TSrc, C, F>();
    if constexpr (TH > 0)
        hipLaunchKernelGGL((dm_pyrdown_tile<TSrc, C, F, TH>), dim3(cdiv(wo, 64), cdiv(ho, TH)), dim3(256), 0, st, src, h, w, dst, ho, wo);
    else
        hipLaunchKernelGGL((dm_pyrdown<TSrc, C, F>), dm_grid2(ho, wo), dim3(256), 0, st, src, h, w, dst, ho, wo);
}

// six pixels per thread: three workgroups share a CU's 160 KB of LDS at every radius (53 KB at radius 15).  A/B at radius 7,
// 64 x 24 MP, ms per frame of the whole stacker: 4 pixels 1.726, 6 -> 1.69, 8 -> 1.94 (two workgroups per CU) / 2.33 (three)
constexpr int DM_BIL_NP = 6;
inline void dm_bilateral_launch(hipStream_t st, const DmBilateral& a) {
    const dim3 grid(cdiv(a.w, 64), cdiv(a.h, 4 * DM_BIL_NP));
    const size_t lds = dm_bilateral_lds(DM_BIL_NP, a.radius);
    if (a.radius == 7) hipLaunchKernelGGL((dm_bilateral<DM_BIL_NP, 7>), grid, dim3(256), lds, st, a);   // smooth_size 15: the default
    else hipLaunchKernelGGL((dm_bilateral<DM_BIL_NP, 0>), grid, dim3(256), lds, st, a);
}

// pass 1 for the frame just stored in d->frames[i]
constexpr int DM_FMM_FRAMES = 4096;   // frames with a per-frame min / max slot (more: the normalising pass stays)

template <typename T, typename F>
int dmap_energy(mi_dmap* d, int i) {
    const int h = d->p.height, w = d->p.width;
    const size_t np = (size_t)h * w;
    hipStream_t st = d->stream;
    F *tA = (F*)d->tmpA, *tB = (F*)d->tmpB, *tC = (F*)d->tmpC, *en = (F*)d->en[i], *gmax = (F*)d->scalF;
    const DmTapsT<F>& taps = [&]() -> const DmTapsT<F>& {
        if constexpr (sizeof(F) == 4) return d->taps; else return d->tapsd;
    }();
    if constexpr (sizeof(F) == 4) {
        // the defaults (5-tap blur, 5 x 5 aperture) in one pass over the frame; -DMI_DMAP_SEPARATE_ENERGY=1 keeps the four kernels
#ifndef MI_DMAP_SEPARATE_ENERGY
#define MI_DMAP_SEPARATE_ENERGY 0
#endif
        if (!MI_DMAP_SEPARATE_ENERGY && d->p.energy == MI_DM_ENERGY_LAPLACIAN && taps.ksize == 5 && d->k2.ksize == 5 && h >= 8 && w >= 8) {
            DmK25 K;
            for (int q = 0; q < 25; ++q) K.k[q] = d->k2.k[q];
            float* fmm = (d->fmm && i < DM_FMM_FRAMES && d->p.smooth_size > 0) ? d->fmm + 2 * i : nullptr;
            hipLaunchKernelGGL((dm_energy_lap5<T>), dim3(cdiv(w, 64), cdiv(h, 32)), dim3(256), 0, st, (const T*)d->frames[i], h, w, en,
                               gmax, taps, K, fmm);
            if ((int)d->have_fmm.size() <= i) d->have_fmm.resize(i + 1, 0);
            d->have_fmm[i] = fmm != nullptr;
            MI_HIP(hipGetLastError());
            return MI_OK;
        }
    }
    if ((int)d->have_fmm.size() <= i) d->have_fmm.resize(i + 1, 0);
    d->have_fmm[i] = 0;
    hipLaunchKernelGGL((dm_gray<T, F>), dm_grid1(np), dim3(256), 0, st, (const T*)d->frames[i], np, tA);
    if (d->p.energy == MI_DM_ENERGY_SOBEL) {
        hipLaunchKernelGGL((dm_sobel<F>), dm_grid2(h, w), dim3(256), 0, st, (const F*)tA, h, w, en, gmax);
    } else {
        hipLaunchKernelGGL((dm_blur<true, F>), dm_grid2(h, w), dim3(256), 0, st, (const F*)tA, h, w, tB, taps);
        hipLaunchKernelGGL((dm_blur<false, F>), dm_grid2(h, w), dim3(256), 0, st, (const F*)tB, h, w, tC, taps);
        if (d->k2.ksize == 5)
            hipLaunchKernelGGL((dm_laplacian_rows<5, F>), dim3(cdiv(w, 64), cdiv(h, 4 * DM_LAP_ROWS)), dim3(256), 0, st, (const F*)tC, h, w,
                               en, gmax, d->k2);
        else if (d->k2.ksize == 3)
            hipLaunchKernelGGL((dm_laplacian_rows<3, F>), dim3(cdiv(w, 64), cdiv(h, 4 * DM_LAP_ROWS)), dim3(256), 0, st, (const F*)tC, h, w,
                               en, gmax, d->k2);
        else
            hipLaunchKernelGGL((dm_laplacian<0, F>), dm_grid2(h, w), dim3(256), 0, st, (const F*)tC, h, w, en, gmax, d->k2);
    }
    MI_HIP(hipGetLastError());
    return MI_OK;
}

int dmap_new_slot(mi_dmap* d) {
    if ((size_t)d->n < d->frames.size()) return MI_OK;   // reuse after reset
    const size_t np = (size_t)d->p.height * d->p.width;
    void *f = nullptr, *e = nullptr;
    int rc = dmap_alloc_bytes(d, &f, np * 3 * d->esz);
    if (rc || (rc = dmap_alloc_bytes(d, &e, np * d->fsz))) return rc;
    d->frames.push_back(f);
    d->en.push_back(e);
    return MI_OK;
}

// pass 2 and the collapse, with the weight planes already in en[i] (type W)
template <typename T, typename F, typename W>
int dmap_blend(mi_dmap* d) {
    const int h = d->p.height, w = d->p.width, L = d->p.levels, N = d->n;
    const size_t np = (size_t)h * w;
    hipStream_t st = d->stream;
    const bool avg = d->p.map_type == MI_DM_MAP_AVERAGE;
    auto Gl = [&](int l) { return (F*)d->G[l]; };
    auto Wl = [&](int l) { return (W*)d->W[l]; };
    auto Bl = [&](int l) { return (F*)d->B[l]; };
    for (int i = 0; i < N; ++i) {
        const T* frame = (const T*)d->frames[i];
        const int first = i == 0;
        hipLaunchKernelGGL((dm_weight<W>), dm_grid1(np), dim3(256), 0, st, (const W*)d->en[i], (const W*)d->tot, np,
                           avg ? 1 : 0, Wl(0));
        for (int l = 1; l < L; ++l) {
            if (l == 1) dm_pyrdown_launch<T, 3, F>(st, frame, h, w, Gl(1), d->lh[1], d->lw[1]);
            else dm_pyrdown_launch<F, 3, F>(st, (const F*)Gl(l - 1), d->lh[l - 1], d->lw[l - 1], Gl(l), d->lh[l], d->lw[l]);
            dm_pyrdown_launch<W, 1, W>(st, (const W*)Wl(l - 1), d->lh[l - 1], d->lw[l - 1], Wl(l), d->lh[l], d->lw[l]);
        }
        const size_t ntop = (size_t)d->lh[L - 1] * d->lw[L - 1];
        if (L == 1)
            hipLaunchKernelGGL((dm_top_blend<T, F, W>), dm_grid1(ntop), dim3(256), 0, st, frame, ntop, (const W*)Wl(0), Bl(0),
                               first);
        else
            hipLaunchKernelGGL((dm_top_blend<F, F, W>), dm_grid1(ntop), dim3(256), 0, st, (const F*)Gl(L - 1), ntop,
                               (const W*)Wl(L - 1), Bl(L - 1), first);
        for (int l = L - 2; l >= 0; --l) {
            const dim3 g = dm_grid2(cdiv(d->lh[l], 2), cdiv(d->lw[l], 2));   // one lane per 2 x 2 quad
            if (l == 0)
                hipLaunchKernelGGL((dm_lap_blend_quad<T, F, W>), g, dim3(256), 0, st, frame, h, w, (const F*)Gl(1), d->lh[1],
                                   d->lw[1], (const W*)Wl(0), Bl(0), first);
            else
                hipLaunchKernelGGL((dm_lap_blend_quad<F, F, W>), g, dim3(256), 0, st, (const F*)Gl(l), d->lh[l], d->lw[l],
                                   (const F*)Gl(l + 1), d->lh[l + 1], d->lw[l + 1], (const W*)Wl(l), Bl(l), first);
        }
    }
    // collapse in place (level l takes pyrUp of the finished level l+1), clip, cast
    for (int l = L - 2; l >= 0; --l)
        hipLaunchKernelGGL((dm_collapse_quad<F>), dm_grid2(cdiv(d->lh[l], 2), cdiv(d->lw[l], 2)), dim3(256), 0, st,
                           (const F*)Bl(l + 1), d->lh[l + 1], d->lw[l + 1], (const F*)Bl(l), d->lh[l], d->lw[l], Bl(l));
    hipLaunchKernelGGL((dm_finalize<T, F>), dm_grid1(np * 3), dim3(256), 0, st, (const F*)Bl(0), np * 3,
                       (F)(sizeof(T) == 1 ? 255 : 65535), (T*)d->out_dev);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

// focus map: running sum / maximum of the energies of type W in en[i] (already accumulated by the caller for the
// AVERAGE map); the MAX map turns them into relatives and sums those
template <typename W>
int dmap_focus_map(mi_dmap* d) {
    const size_t np = (size_t)d->p.height * d->p.width;
    if (d->p.map_type == MI_DM_MAP_MAX)
        for (int i = 0; i < d->n; ++i)
            hipLaunchKernelGGL((dm_relative<W>), dm_grid1(np), dim3(256), 0, d->stream, (W*)d->en[i], (const W*)d->mx, np,
                               (W)d->temperature, (W*)d->tot, i == 0);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

template <typename T, typename F>
int dmap_finish_t(mi_dmap* d) {
    const int h = d->p.height, w = d->p.width, N = d->n;
    const size_t np = (size_t)h * w;
    hipStream_t st = d->stream;
    const bool avg = d->p.map_type == MI_DM_MAP_AVERAGE, smooth = d->p.smooth_size > 0;
    static const uint32_t mm_init[2] = {0x7f800000u, 0u};
    static const uint64_t mmd_init[2] = {0x7ff0000000000000ull, 0ull};
    const dim3 gnorm((unsigned)std::min<size_t>((np + 255) / 256, 4096));
    F* sF = (F*)d->scalF;
    // energies / max, smoothing, running sum (AVERAGE) or maximum (MAX) over the frames
    for (int i = 0; i < N; ++i) {
        const bool folded = sizeof(F) == 4 && smooth && i < (int)d->have_fmm.size() && d->have_fmm[i];
        if (folded) {
            // the energy kernel left the frame's raw min / max in fmm: no normalising pass, dm_bilateral divides as it stages
        } else if (sizeof(F) == 4 && smooth) {   // float-32: min / max of the normalised plane feed the bilateral filter
            MI_HIP(hipMemcpyAsync(d->scalf, mm_init, 8, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL((dm_normalise<float>), gnorm, dim3(256), 0, st, (float*)d->en[i], np, (const float*)d->scalF,
                               d->scalf);
        } else {
            if (sizeof(F) == 4) MI_HIP(hipMemcpyAsync(sF + 1, mm_init, 8, hipMemcpyHostToDevice, st));
            else MI_HIP(hipMemcpyAsync(sF + 1, mmd_init, 16, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL((dm_normalise<F>), gnorm, dim3(256), 0, st, (F*)d->en[i], np, (const F*)sF, sF + 1);
        }
        if (smooth) {
            const float* bsrc = (const float*)d->en[i];
            float* bdst = d->spare;
            if (sizeof(F) == 8) {   // float-64: smoothing works on a float32 copy and returns float32 (depth_map.py:46-51)
                hipLaunchKernelGGL(dm_to_f32, dm_grid1(np), dim3(256), 0, st, (const double*)d->en[i], np, (float*)d->tmpA);
                MI_HIP(hipMemcpyAsync(d->scalf, mm_init, 8, hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL((dm_normalise<float>), gnorm, dim3(256), 0, st, (float*)d->tmpA, np,
                                   (const float*)(d->scalf + 4), d->scalf);   // *gmax = 0: min / max only
                bsrc = (const float*)d->tmpA;
                bdst = (float*)d->en[i];
            }
            const float* nrm = folded ? (const float*)d->scalF : nullptr;
            hipLaunchKernelGGL(dm_bilateral_lut, dim3(1), dim3(1024), 0, st, folded ? (const float*)(d->fmm + 2 * i) : (const float*)d->scalf,
                               d->color_coeff, d->lut, d->scalf + 2, nrm);
            DmBilateral a{bsrc, bdst, h, w, d->radius, d->ntaps, d->disc, d->lut, d->scalf + 2,
                          (float*)(avg ? d->tot : d->mx), avg ? 0 : 1, i == 0, nrm};
            dm_bilateral_launch(st, a);
            if (sizeof(F) == 4) {
                void* t = d->en[i];
                d->en[i] = d->spare;
                d->spare = (float*)t;
            }
        } else {
            hipLaunchKernelGGL((dm_accumulate<F>), dm_grid1(np), dim3(256), 0, st, (const F*)d->en[i], np,
                               (F*)(avg ? d->tot : d->mx), avg ? 0 : 1, i == 0);
        }
    }
    MI_HIP(hipGetLastError());
    int rc;
    if (smooth || sizeof(F) == 4) {   // weights are float32
        if ((rc = dmap_focus_map<float>(d))) return rc;
        return dmap_blend<T, F, float>(d);
    }
    if ((rc = dmap_focus_map<F>(d))) return rc;
    return dmap_blend<T, F, F>(d);
}

// The stacker's steps one at a time on host planes (the reference's public methods, depth_map.py:28-62): same kernels as the
// fused path, scratch planes allocated per call.
template <typename F>
int dmap_planes_t(mi_dmap* d, int stage, const void* host_in, int n, void* host_out) {
    const int h = d->p.height, w = d->p.width;
    const size_t np = (size_t)h * w;
    hipStream_t st = d->stream;
    const bool smoothed_w = d->p.smooth_size > 0 || sizeof(F) == 4;   // type of the planes the focus map works on (W)
    struct Scratch {   // freed on every way out, the early returns of MI_HIP included
        hipStream_t st;
        std::vector<void*> v;
        ~Scratch() {
            (void)hipStreamSynchronize(st);
            for (void* q : v) (void)hipFree(q);
        }
    } tmp{st, {}};
    auto dalloc = [&](size_t bytes) -> void* {
        void* q = nullptr;
        if (hipMalloc(&q, bytes ? bytes : 1) != hipSuccess) return nullptr;
        tmp.v.push_back(q);
        return q;
    };
    auto cleanup = [&](int rc) { return rc; };
    const DmTapsT<F>& taps = [&]() -> const DmTapsT<F>& {
        if constexpr (sizeof(F) == 4) return d->taps; else return d->tapsd;
    }();
    if (stage == 0 || stage == 1) {            // get_sobel_map / get_laplacian_map: gray planes (F) -> energies (F)
        F* in = (F*)dalloc(np * sizeof(F));
        F *tB = (F*)dalloc(np * sizeof(F)), *tC = (F*)dalloc(np * sizeof(F)), *en = (F*)dalloc(np * sizeof(F)), *gmax = (F*)dalloc(sizeof(F));
        if (!in || !tB || !tC || !en || !gmax) return cleanup(fail(MI_ERR_NOMEM, "out of device memory"));
        for (int i = 0; i < n; ++i) {
            MI_HIP(hipMemcpyAsync(in, (const F*)host_in + (size_t)i * np, np * sizeof(F), hipMemcpyHostToDevice, st));
            MI_HIP(hipMemsetAsync(gmax, 0, sizeof(F), st));
            if (stage == 0) {
                hipLaunchKernelGGL((dm_sobel<F>), dm_grid2(h, w), dim3(256), 0, st, (const F*)in, h, w, en, gmax);
            } else {
                hipLaunchKernelGGL((dm_blur<true, F>), dm_grid2(h, w), dim3(256), 0, st, (const F*)in, h, w, tB, taps);
                hipLaunchKernelGGL((dm_blur<false, F>), dm_grid2(h, w), dim3(256), 0, st, (const F*)tB, h, w, tC, taps);
                if (d->k2.ksize == 5)
                    hipLaunchKernelGGL((dm_laplacian_rows<5, F>), dim3(cdiv(w, 64), cdiv(h, 4 * DM_LAP_ROWS)), dim3(256), 0, st, (const F*)tC, h, w, en, gmax, d->k2);
                else if (d->k2.ksize == 3)
                    hipLaunchKernelGGL((dm_laplacian_rows<3, F>), dim3(cdiv(w, 64), cdiv(h, 4 * DM_LAP_ROWS)), dim3(256), 0, st, (const F*)tC, h, w, en, gmax, d->k2);
                else
                    hipLaunchKernelGGL((dm_laplacian<0, F>), dm_grid2(h, w), dim3(256), 0, st, (const F*)tC, h, w, en, gmax, d->k2);
            }
            MI_HIP(hipGetLastError());
            MI_HIP(hipMemcpyAsync((F*)host_out + (size_t)i * np, en, np * sizeof(F), hipMemcpyDeviceToHost, st));
            MI_HIP(hipStreamSynchronize(st));
        }
        return cleanup(MI_OK);
    }
    if (stage == 2) {                          // smooth_energy: planes (F) -> float32 planes (cv2.bilateralFilter on float32)
        if (d->p.smooth_size <= 0) return cleanup(fail(MI_ERR_STATE, "smooth_size <= 0: nothing to smooth"));
        static const uint32_t mm_init[2] = {0x7f800000u, 0u};
        const dim3 gnorm((unsigned)std::min<size_t>((np + 255) / 256, 4096));
        F* in = (F*)dalloc(np * sizeof(F));
        float *f32 = (float*)dalloc(np * 4), *out = (float*)dalloc(np * 4), *acc = (float*)dalloc(np * 4), *sc = (float*)dalloc(8 * 4);
        if (!in || !f32 || !out || !acc || !sc) return cleanup(fail(MI_ERR_NOMEM, "out of device memory"));
        MI_HIP(hipMemsetAsync(sc, 0, 8 * 4, st));
        for (int i = 0; i < n; ++i) {
            MI_HIP(hipMemcpyAsync(in, (const F*)host_in + (size_t)i * np, np * sizeof(F), hipMemcpyHostToDevice, st));
            const float* src = (const float*)in;
            if constexpr (sizeof(F) == 8) {
                hipLaunchKernelGGL(dm_to_f32, dm_grid1(np), dim3(256), 0, st, (const double*)in, np, f32);
                src = f32;
            }
            MI_HIP(hipMemcpyAsync(sc, mm_init, 8, hipMemcpyHostToDevice, st));
            // *gmax = sc[4] = 0: the plane is left as it is, only its min / max are taken
            hipLaunchKernelGGL((dm_normalise<float>), gnorm, dim3(256), 0, st, (float*)src, np, (const float*)(sc + 4), sc);
            hipLaunchKernelGGL(dm_bilateral_lut, dim3(1), dim3(1024), 0, st, (const float*)sc, d->color_coeff, d->lut, sc + 2);
            DmBilateral a{src, out, h, w, d->radius, d->ntaps, d->disc, d->lut, sc + 2, acc, 0, 1, nullptr};
            dm_bilateral_launch(st, a);
            MI_HIP(hipGetLastError());
            MI_HIP(hipMemcpyAsync((float*)host_out + (size_t)i * np, out, np * 4, hipMemcpyDeviceToHost, st));
            MI_HIP(hipStreamSynchronize(st));
        }
        return cleanup(MI_OK);
    }
    if (stage == 3) {                          // get_focus_map: n energy planes -> n weight planes, same type (W)
        auto run = [&](auto zero) -> int {
            using W = decltype(zero);
            W* e = (W*)dalloc(np * sizeof(W) * (size_t)n);
            W *tot = (W*)dalloc(np * sizeof(W)), *mx = (W*)dalloc(np * sizeof(W)), *wgt = (W*)dalloc(np * sizeof(W));
            if (!e || !tot || !mx || !wgt) return fail(MI_ERR_NOMEM, "out of device memory");
            MI_HIP(hipMemcpyAsync(e, host_in, np * sizeof(W) * (size_t)n, hipMemcpyHostToDevice, st));
            const bool avg = d->p.map_type == MI_DM_MAP_AVERAGE;
            for (int i = 0; i < n; ++i)
                hipLaunchKernelGGL((dm_accumulate<W>), dm_grid1(np), dim3(256), 0, st, (const W*)(e + (size_t)i * np), np, avg ? tot : mx,
                                   avg ? 0 : 1, i == 0);
            if (!avg)
                for (int i = 0; i < n; ++i)
                    hipLaunchKernelGGL((dm_relative<W>), dm_grid1(np), dim3(256), 0, st, e + (size_t)i * np, (const W*)mx, np,
                                       (W)d->temperature, tot, i == 0);
            for (int i = 0; i < n; ++i) {
                hipLaunchKernelGGL((dm_weight<W>), dm_grid1(np), dim3(256), 0, st, (const W*)(e + (size_t)i * np), (const W*)tot, np,
                                   avg ? 1 : 0, wgt);
                MI_HIP(hipGetLastError());
                MI_HIP(hipMemcpyAsync((W*)host_out + (size_t)i * np, wgt, np * sizeof(W), hipMemcpyDeviceToHost, st));
                MI_HIP(hipStreamSynchronize(st));
            }
            return MI_OK;
        };
        return cleanup(smoothed_w ? run(0.0f) : run(F(0)));
    }
    return cleanup(fail(MI_ERR_INVALID, "stage must be 0 .. 3"));
}

}  // namespace

extern "C" {

int mi_dmap_set_temperature(mi_dmap_t* d, double temperature) {
    if (!d) return fail(MI_ERR_INVALID, "null handle");
    if (d->p.map_type == MI_DM_MAP_MAX && !(temperature != 0.0)) return fail(MI_ERR_INVALID, "temperature must not be 0");
    d->temperature = temperature;
    return MI_OK;
}

int mi_dmap_planes(mi_dmap_t* d, int stage, const void* host_in, int n, void* host_out) {
    if (!d || !host_in || !host_out || n < 1) return fail(MI_ERR_INVALID, "bad argument");
    MI_HIP(hipSetDevice(d->p.device));
    return d->f64 ? dmap_planes_t<double>(d, stage, host_in, n, host_out) : dmap_planes_t<float>(d, stage, host_in, n, host_out);
}

void mi_dmap_default_params(mi_dmap_params_t* p) {
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->dtype = MI_U8;
    p->map_type = MI_DM_MAP_AVERAGE;     // constants.py:151-157
    p->energy = MI_DM_ENERGY_LAPLACIAN;
    p->kernel_size = 5;
    p->blur_size = 5;
    p->smooth_size = 15;
    p->temperature = 0.1f;
    p->levels = 3;
    p->float_type = MI_F32;
}

int mi_dmap_create(mi_dmap_t** out, const mi_dmap_params_t* params) {
    if (!out || !params) return fail(MI_ERR_INVALID, "null argument");
    *out = nullptr;
    const mi_dmap_params_t p = *params;
    if (p.height < 1 || p.width < 1) return fail(MI_ERR_INVALID, "bad frame size %dx%d", p.width, p.height);
    if ((size_t)p.height * p.width > ((size_t)1 << 30)) return fail(MI_ERR_UNSUPPORTED, "frame too large");
    if (p.dtype != MI_U8 && p.dtype != MI_U16) return fail(MI_ERR_INVALID, "dtype must be MI_U8 or MI_U16");
    if (p.float_type != MI_F32 && p.float_type != MI_F64) return fail(MI_ERR_INVALID, "float_type must be MI_F32 or MI_F64");
    if (p.map_type != MI_DM_MAP_AVERAGE && p.map_type != MI_DM_MAP_MAX) return fail(MI_ERR_INVALID, "bad map_type %d", p.map_type);
    if (p.energy != MI_DM_ENERGY_LAPLACIAN && p.energy != MI_DM_ENERGY_SOBEL) return fail(MI_ERR_INVALID, "bad energy %d", p.energy);
    if (p.energy == MI_DM_ENERGY_LAPLACIAN) {
        if (p.kernel_size < 1 || p.kernel_size > 15 || p.kernel_size % 2 == 0)
            return fail(MI_ERR_INVALID, "kernel_size must be odd and in [1, 15]");
        if (p.blur_size < 1 || p.blur_size > 31 || p.blur_size % 2 == 0)
            return fail(MI_ERR_INVALID, "blur_size must be odd and in [1, 31]");
    }
    if (p.smooth_size > 31) return fail(MI_ERR_UNSUPPORTED, "smooth_size above 31 (bilateral radius above 15)");
    if (p.levels < 1 || p.levels > 16) return fail(MI_ERR_INVALID, "levels must be in [1, 16]");
    if (p.map_type == MI_DM_MAP_MAX && !(p.temperature != 0.f)) return fail(MI_ERR_INVALID, "temperature must not be 0");
    int ndev = 0;
    int rc = mi_device_count(&ndev);
    if (rc) return rc;
    if (ndev == 0) return fail(MI_ERR_NO_DEVICE, "no HIP device visible");
    if (p.device < 0 || p.device >= ndev) return fail(MI_ERR_INVALID, "bad device %d", p.device);
    MI_HIP(hipSetDevice(p.device));
    mi_dmap* d = new (std::nothrow) mi_dmap();
    if (!d) return fail(MI_ERR_NOMEM, "out of host memory");
    d->p = p;
    d->temperature = (double)p.temperature;
    d->esz = p.dtype == MI_U8 ? 1 : 2;
    d->f64 = p.float_type == MI_F64;
    d->fsz = d->f64 ? 8 : 4;
    if (p.energy == MI_DM_ENERGY_LAPLACIAN) {
        dmap_gauss_taps(p.blur_size, d->taps);
        dmap_gauss_taps(p.blur_size, d->tapsd);
        dmap_laplacian_kernel(p.kernel_size, d->k2);
    }
    const size_t np = (size_t)p.height * p.width;
    auto body = [&]() -> int {
        MI_HIP(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
        int r;
        const size_t fsz = d->fsz;
        if ((r = dmap_alloc_bytes(d, &d->tmpA, np * fsz)) || (r = dmap_alloc_bytes(d, &d->tmpB, np * fsz)) ||
            (r = dmap_alloc_bytes(d, &d->tmpC, np * fsz)) || (r = dmap_alloc_bytes(d, &d->tot, np * fsz)) ||
            (r = dmap_alloc_bytes(d, &d->scalF, 64)) || (r = dmap_alloc(d, &d->scalf, 16)) ||
            (r = dmap_alloc(d, &d->lut, DM_LUT_BINS + 2)))
            return r;
        if (!d->f64 && (r = dmap_alloc(d, &d->spare, np))) return r;
        if (!d->f64 && (r = dmap_alloc(d, &d->fmm, 2 * DM_FMM_FRAMES))) return r;
        if (p.map_type == MI_DM_MAP_MAX && (r = dmap_alloc_bytes(d, &d->mx, np * fsz))) return r;
        if ((r = dmap_alloc_bytes(d, &d->out_dev, np * 3 * d->esz))) return r;
        if (p.smooth_size > 0) {   // bilateral disc, cv2.bilateralFilter(e, smooth_size, 25, 25) (depth_map.py:50)
            const double sigma_color = 25.0, sigma_space = 25.0;
            d->radius = std::max(p.smooth_size / 2, 1);
            d->color_coeff = -0.5 / (sigma_color * sigma_color);
            const double cs = -0.5 / (sigma_space * sigma_space);
            std::vector<int2> taps;
            const int pw = 64 + 2 * d->radius;   // patch row length of dm_bilateral's 16 x 64 tile
            for (int i = -d->radius; i <= d->radius; ++i)
                for (int j = -d->radius; j <= d->radius; ++j) {
                    const double r = std::sqrt((double)i * i + (double)j * j);
                    if (r > d->radius) continue;
                    const float sw = (float)std::exp(r * r * cs);
                    int bits;
                    memcpy(&bits, &sw, 4);
                    taps.push_back(make_int2(i * pw + j, bits));
                }
            d->ntaps = (int)taps.size();
            if ((r = dmap_alloc(d, &d->disc, taps.size()))) return r;
            MI_HIP(hipMemcpy(d->disc, taps.data(), taps.size() * sizeof(int2), hipMemcpyHostToDevice));
        }
        int lh = p.height, lw = p.width;
        for (int l = 0; l < p.levels; ++l) {
            d->lh.push_back(lh);
            d->lw.push_back(lw);
            void *g = nullptr, *wgt = nullptr, *b = nullptr;
            const size_t n = (size_t)lh * lw;
            if (l > 0 && (r = dmap_alloc_bytes(d, &g, n * 3 * fsz))) return r;
            if ((r = dmap_alloc_bytes(d, &wgt, n * fsz)) || (r = dmap_alloc_bytes(d, &b, n * 3 * fsz))) return r;
            d->G.push_back(g);
            d->W.push_back(wgt);
            d->B.push_back(b);
            lh = (lh + 1) / 2;
            lw = (lw + 1) / 2;
        }
        MI_HIP(hipMemsetAsync(d->scalF, 0, 64, d->stream));
        MI_HIP(hipMemsetAsync(d->scalf, 0, 64, d->stream));
        if (d->fmm) hipLaunchKernelGGL(dm_fmm_reset, dim3(cdiv(DM_FMM_FRAMES, 256)), dim3(256), 0, d->stream, d->fmm, DM_FMM_FRAMES);
        return MI_OK;
    };
    rc = body();
    if (rc) {
        dmap_free(d);
        delete d;
        return rc;
    }
    *out = d;
    return MI_OK;
}

void mi_dmap_destroy(mi_dmap_t* d) {
    if (!d) return;
    (void)hipSetDevice(d->p.device);
    if (d->stream) (void)hipStreamSynchronize(d->stream);
    dmap_free(d);
    delete d;
}

int mi_dmap_reset(mi_dmap_t* d) {
    if (!d) return fail(MI_ERR_INVALID, "null handle");
    MI_HIP(hipSetDevice(d->p.device));
    d->n = 0;
    d->finished = false;
    MI_HIP(hipMemsetAsync(d->scalF, 0, 64, d->stream));
    if (d->fmm) hipLaunchKernelGGL(dm_fmm_reset, dim3(cdiv(DM_FMM_FRAMES, 256)), dim3(256), 0, d->stream, d->fmm, DM_FMM_FRAMES);
    return MI_OK;
}

int mi_dmap_frames_pushed(const mi_dmap_t* d, int* n) {
    if (!d || !n) return fail(MI_ERR_INVALID, "null argument");
    *n = d->n;
    return MI_OK;
}

static int dmap_push_common(mi_dmap_t* d, const void* src, size_t row_stride_bytes, bool on_device) {
    if (!d) return fail(MI_ERR_INVALID, "null handle");
    if (!src) return fail(MI_ERR_INVALID, "null frame");
    if (d->finished) return fail(MI_ERR_STATE, "push after finish; call mi_dmap_reset first");
    MI_HIP(hipSetDevice(d->p.device));
    int rc = dmap_new_slot(d);
    if (rc) return rc;
    const size_t rb = (size_t)d->p.width * 3 * d->esz;
    if (row_stride_bytes == 0) row_stride_bytes = rb;
    if (row_stride_bytes < rb) return fail(MI_ERR_INVALID, "row stride smaller than a row");
    MI_HIP(hipMemcpy2DAsync(d->frames[d->n], rb, src, row_stride_bytes, rb, d->p.height,
                            on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, d->stream));
    if (!on_device) MI_HIP(hipStreamSynchronize(d->stream));   // the caller may reuse its buffer
    if (d->f64) rc = d->p.dtype == MI_U8 ? dmap_energy<uint8_t, double>(d, d->n) : dmap_energy<uint16_t, double>(d, d->n);
    else rc = d->p.dtype == MI_U8 ? dmap_energy<uint8_t, float>(d, d->n) : dmap_energy<uint16_t, float>(d, d->n);
    if (rc) return rc;
    d->n++;
    return MI_OK;
}

int mi_dmap_push_frame(mi_dmap_t* d, const void* host_bgr, size_t row_stride_bytes) {
    return dmap_push_common(d, host_bgr, row_stride_bytes, false);
}

int mi_dmap_push_frame_device(mi_dmap_t* d, const void* dev_bgr) {
    return dmap_push_common(d, dev_bgr, 0, true);
}

int mi_dmap_finish_device(mi_dmap_t* d, void* dev_out) {
    if (!d) return fail(MI_ERR_INVALID, "null handle");
    if (d->finished) return fail(MI_ERR_STATE, "finish called twice; call mi_dmap_reset first");
    if (d->n == 0) return fail(MI_ERR_STATE, "finish with no frames pushed");
    MI_HIP(hipSetDevice(d->p.device));
    int rc;
    if (d->f64) rc = d->p.dtype == MI_U8 ? dmap_finish_t<uint8_t, double>(d) : dmap_finish_t<uint16_t, double>(d);
    else rc = d->p.dtype == MI_U8 ? dmap_finish_t<uint8_t, float>(d) : dmap_finish_t<uint16_t, float>(d);
    if (rc) return rc;
    d->finished = true;
    if (dev_out)
        MI_HIP(hipMemcpyAsync(dev_out, d->out_dev, (size_t)d->p.height * d->p.width * 3 * d->esz, hipMemcpyDeviceToDevice,
                              d->stream));
    MI_HIP(hipStreamSynchronize(d->stream));
    return MI_OK;
}

int mi_dmap_finish(mi_dmap_t* d, void* host_out, size_t row_stride_bytes) {
    if (!host_out) return fail(MI_ERR_INVALID, "null output buffer");
    int rc = mi_dmap_finish_device(d, nullptr);
    if (rc) return rc;
    const size_t rb = (size_t)d->p.width * 3 * d->esz;
    if (row_stride_bytes == 0) row_stride_bytes = rb;
    if (row_stride_bytes < rb) return fail(MI_ERR_INVALID, "row stride smaller than a row");
    MI_HIP(hipMemcpy2D(host_out, row_stride_bytes, d->out_dev, rb, rb, d->p.height, hipMemcpyDeviceToHost));
    return MI_OK;
}

}  // extern "C"
