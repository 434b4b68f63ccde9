// tiled_host.hpp -- host side of the tiled production path and of host-frame staging.
// Included by capi.hip after `struct mi_stack` is complete.
#pragma once

namespace mi {

constexpr int TILE_H = MI_TILE_H, TILE_W = MI_TILE_W;

struct TiledState {
    int bcap = 0;                // frames per fused launch
    std::vector<float*> Gb;      // Gb[l], l = 1..L : bcap images of level l
    std::vector<size_t> gstride; // floats between frames in Gb[l]
    void* ring = nullptr;        // staging ring for host-pushed frames (bcap frames, in_dtype)
    size_t frame_bytes = 0;
    int pending = 0;             // frames staged in the ring, not yet processed
    int last_nb = 0;             // size of the most recently processed batch
    int32_t* lev = nullptr;      // base batch scratch
    uint32_t* cnt = nullptr;
    float* logp = nullptr;
};

inline TiledState*& tstate(mi_stack* s) {
    return *reinterpret_cast<TiledState**>(&s->tiled);
}
inline TiledState* tstate(const mi_stack* s) { return reinterpret_cast<TiledState*>(s->tiled); }

bool tiled_available() { return true; }

int tiled_create(mi_stack* s) {
    auto* t = new TiledState();
    tstate(s) = t;
    const int L = s->L;
    t->bcap = s->p.batch_frames > 0 ? s->p.batch_frames : 32;
    t->frame_bytes = (size_t)s->p.height * s->p.width * 3 * dtype_size(s->p.in_dtype);
    if (s->p.impl != MI_IMPL_TILED) {
        t->bcap = 1;  // simple impl: the ring holds one frame (s->frame_dev)
        return MI_OK;
    }
    t->Gb.assign(L + 1, nullptr);
    t->gstride.assign(L + 1, 0);
    int rc;
    for (int l = 1; l <= L; ++l) {
        t->gstride[l] = (size_t)s->lh[l] * s->lw[l] * 3;
        if ((rc = dev_alloc_t(s, &t->Gb[l], t->gstride[l] * t->bcap))) return rc;
    }
    const size_t nb = (size_t)s->lh[L] * s->lw[L];
    if ((rc = dev_alloc_t(s, &t->lev, nb * t->bcap))) return rc;
    if ((rc = dev_alloc_t(s, &t->cnt, (size_t)s->nlevels_hist * t->bcap))) return rc;
    if ((rc = dev_alloc_t(s, &t->logp, (size_t)s->nlevels_hist * t->bcap))) return rc;
    return MI_OK;
}

void tiled_destroy(mi_stack* s) {
    delete tstate(s);
    tstate(s) = nullptr;
}

int tiled_reset(mi_stack* s) {
    tstate(s)->pending = 0;
    tstate(s)->last_nb = 0;
    return MI_OK;
}

int tiled_pending(const mi_stack* s) { return tstate(s) ? tstate(s)->pending : 0; }

const float* tiled_last_gauss(mi_stack* s, int level) {
    TiledState* t = tstate(s);
    if (s->p.impl != MI_IMPL_TILED) return s->G[level];
    int last = t->last_nb > 0 ? t->last_nb - 1 : 0;
    return t->Gb[level] + (size_t)last * t->gstride[level];
}

template <typename Kern>
int set_lds_once(Kern kern, size_t lds) {
    MI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return MI_OK;
}

template <typename TIn, bool FMA>
int launch_level(mi_stack* s, int l, const void* src, size_t src_stride, int nb) {
    using Gm = TileGeom<TILE_H, TILE_W>;
    TiledState* t = tstate(s);
    LevelArgs a{};
    a.src = src;
    a.src_stride = src_stride;
    a.gnext = t->Gb[l + 1];
    a.gnext_stride = t->gstride[l + 1];
    a.nframes = nb;
    a.h = s->lh[l];
    a.w = s->lw[l];
    a.hn = s->lh[l + 1];
    a.wn = s->lw[l + 1];
    a.tiles_x = cdiv(a.w, TILE_W);
    a.tiles_y = cdiv(a.h, TILE_H);
    // interior tiles: the whole 6-pixel-haloed patch lies inside the image
    a.ty_lo = cdiv(6, TILE_H);
    a.tx_lo = cdiv(6, TILE_W);
    a.ty_hi = (a.h - 6) / TILE_H;
    a.tx_hi = (a.w - 6) / TILE_W;
    int nyi = a.ty_hi - a.ty_lo, nxi = a.tx_hi - a.tx_lo;
    if (nyi <= 0 || nxi <= 0) {
        a.ty_lo = a.ty_hi = a.tx_lo = a.tx_hi = 0;
        nyi = nxi = 0;
    }
    a.sb_x = nxi > 0 ? cdiv(nxi, SB) : 1;
    a.best_e = s->bestE[l];
    a.best_lap = s->bestLap[l];
    a.best_idx = s->bestIdx[l];
    a.first = s->n_pushed == 0;
    a.frame_idx0 = s->first_index + s->n_pushed;
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 3; ++j) a.K.c[i == 0 ? j : (i == 1 ? 2 + j : 5)] = s->K.k[i * 5 + j];
    {
        const char* ab = getenv("MI_ABLATE");  // timing studies only; results are wrong when set
        a.ablate = ab ? atoi(ab) : 0;
    }
    const size_t lds = (size_t)Gm::LDS_FLOATS * sizeof(float);
    auto kin = level_fused<TIn, FMA, true, TILE_H, TILE_W>;
    auto kbd = level_fused<TIn, FMA, false, TILE_H, TILE_W>;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        int rc;
        if ((rc = set_lds_once(kin, lds)) || (rc = set_lds_once(kbd, lds))) return rc;
        attr_set = true;
    }
    // algorithmic bytes of this level pass, SURVEY.md 8(d) attribution: read G_l once,
    // write G_{l+1} once, read G_{l+1} once as the expand source.
    const double bytes = ((double)(l == 0 ? dtype_size(s->p.in_dtype) : 4) * 3.0 * a.h * a.w +
                          24.0 * a.hn * a.wn) * nb;
    const double frac_in = (double)nyi * TILE_H * nxi * TILE_W / ((double)a.h * a.w);
    if (nyi > 0) {
        const int nsb = a.sb_x * cdiv(nyi, SB);
        ProfScope ps(s, l == 0 ? MI_PROF_LEVEL0 : MI_PROF_LEVEL, bytes * frac_in);
        hipLaunchKernelGGL(kin, dim3(cdiv(nsb, 8) * 8 * SB * SB), dim3(MI_TILE_NT), lds, s->stream, a);
    }
    {
        const int nborder = a.tiles_x * a.tiles_y - nyi * nxi;
        ProfScope ps(s, MI_PROF_LEVEL, bytes * (1.0 - frac_in));
        if (nborder > 0) hipLaunchKernelGGL(kbd, dim3(nborder), dim3(MI_TILE_NT), lds, s->stream, a);
    }
    return MI_OK;
}

template <typename TIn, bool FMA>
int run_batch(mi_stack* s, const void* frames, size_t stride, int nb) {
    TiledState* t = tstate(s);
    const int L = s->L;
    int rc;
    if ((rc = launch_level<TIn, FMA>(s, 0, frames, stride, nb))) return rc;
    for (int l = 1; l < L; ++l)
        if ((rc = launch_level<float, FMA>(s, l, t->Gb[l], t->gstride[l] * sizeof(float), nb))) return rc;
    MI_HIP(hipGetLastError());
    {   // base level of the whole batch
        ProfScope ps(s, MI_PROF_BASE, 0.0);
        const int hb = s->lh[L], wb = s->lw[L], npix = hb * wb;
        MI_HIP(hipMemsetAsync(t->cnt, 0, sizeof(uint32_t) * s->nlevels_hist * nb, s->stream));
        hipLaunchKernelGGL((base_gray_hist_batch<FMA>), dim3(cdiv(npix, 256), nb), dim3(256), 0,
                           s->stream, t->Gb[L], t->gstride[L], npix, s->nlevels_hist, t->lev, t->cnt);
        hipLaunchKernelGGL(base_logp_batch, dim3(cdiv(s->nlevels_hist, 256), nb), dim3(256), 0,
                           s->stream, t->cnt, s->nlevels_hist, npix, t->logp);
        const dim3 blk(32, 8);
        for (int f = 0; f < nb; ++f)
            hipLaunchKernelGGL(base_feat_select, grid2d(wb, hb, blk), blk, 0, s->stream,
                               t->lev + (size_t)f * npix, t->logp + (size_t)f * s->nlevels_hist,
                               t->Gb[L] + (size_t)f * t->gstride[L], hb, wb, s->pad,
                               s->first_index + s->n_pushed + f, (s->n_pushed + f) == 0, s->bEnt,
                               s->bDev, s->idxE, s->idxD, s->baseE, s->baseD);
        MI_HIP(hipGetLastError());
    }
    s->n_pushed += nb;
    t->last_nb = nb;
    return MI_OK;
}

int tiled_push(mi_stack* s, const void* dev_frames, int n, size_t stride) {
    TiledState* t = tstate(s);
    if (s->L == 0) return fail(MI_ERR_UNSUPPORTED, "frames smaller than 2*min_size have no pyramid levels");
    int rc = tiled_flush(s);  // keep global frame order: staged host frames come first
    if (rc) return rc;
    const bool fma = s->p.use_fma != 0;
    for (int f0 = 0; f0 < n; f0 += t->bcap) {
        int nb = n - f0 < t->bcap ? n - f0 : t->bcap;
        const void* fr = (const char*)dev_frames + (size_t)f0 * stride;
        switch (s->p.in_dtype) {
            case MI_U8: rc = fma ? run_batch<uint8_t, true>(s, fr, stride, nb) : run_batch<uint8_t, false>(s, fr, stride, nb); break;
            case MI_U16: rc = fma ? run_batch<uint16_t, true>(s, fr, stride, nb) : run_batch<uint16_t, false>(s, fr, stride, nb); break;
            case MI_F32: rc = fma ? run_batch<float, true>(s, fr, stride, nb) : run_batch<float, false>(s, fr, stride, nb); break;
            default: rc = fail(MI_ERR_INVALID, "bad in_dtype");
        }
        if (rc) return rc;
    }
    return MI_OK;
}

int tiled_flush(mi_stack* s) {
    TiledState* t = tstate(s);
    if (!t || t->pending == 0) return MI_OK;
    int n = t->pending;
    t->pending = 0;
    return dispatch_push(s, t->ring, n, t->frame_bytes);
}

// One host frame: stage it in the device ring; a full ring triggers a fused batch.
int tiled_push_host(mi_stack* s, const void* host_bgr, size_t row_stride_bytes) {
    TiledState* t = tstate(s);
    const size_t rb = (size_t)s->p.width * 3 * dtype_size(s->p.in_dtype);
    if (row_stride_bytes == 0) row_stride_bytes = rb;
    if (row_stride_bytes < rb) return fail(MI_ERR_INVALID, "row stride smaller than a row");
    if (!t->ring) {
        if (t->bcap == 1) t->ring = s->frame_dev;
        else {
            int rc = dev_alloc(s, &t->ring, t->frame_bytes * t->bcap);
            if (rc) return rc;
        }
    }
    char* dst = (char*)t->ring + (size_t)t->pending * t->frame_bytes;
    MI_HIP(hipMemcpy2DAsync(dst, rb, host_bgr, row_stride_bytes, rb, s->p.height,
                            hipMemcpyHostToDevice, s->stream));
    // pageable source: the host buffer must be reusable when we return
    MI_HIP(hipStreamSynchronize(s->stream));
    t->pending++;
    if (t->pending == t->bcap) return tiled_flush(s);
    return MI_OK;
}

}  // namespace mi
