// tiled_host.hpp -- host side of the tiled production path and of host-frame staging.
// Included by capi.hip after `struct mi_stack` is complete.
//
// Batch pipeline over three HIP streams (all work of one handle):
//   st0 = s->stream : level-0 interior kernel of batch k      (the big one)
//   st1             : level-0 border kernel of batch k, then the border kernels of levels 1..L-1, and the payload passes
//                     (`st3` below is an alias of st1: a level's payload pass beside the next level's energy pass)
//   st2             : interior kernels of levels 1..L-1 (joined with st1 after every level) and
//                     the base features of batch k
// MI_ARITH_EXACT: batches of 32 frames; level 0 of batch k+1 runs while st2 still works on batch k, so the
// latency-bound small levels hide behind the level-0 kernel.  MI_ARITH_SEPARABLE: a resident push is ONE batch
// (up to 256 frames), processed level after level -- many-tile levels as consecutive launches of 16 frames,
// few-tile levels as parallel frame chunks + merge (launch_level_sep), the payload pass of a level behind it on st1.
// The per-batch Gaussian images Gb[set][l] are double-buffered (set = k & 1) and allocated on demand; events
// order producers and consumers.  Selection state is only ever touched by one stream per level (level 0: st0/st1
// on disjoint pixels; levels >= 1 and base: st2; payload passes: events per level), so stream order keeps the first-max semantics.
#pragma once

namespace mi {

#ifndef MI_NUP
#define MI_NUP 64
#endif
struct TiledState {
    int bcap = 0;                   // frames per fused launch of the host-frame ring (and of MI_ARITH_EXACT)
    int dev_cap = 0;                // MI_ARITH_SEPARABLE, frames resident in HBM: largest batch (0 = not decided yet)
    int gcap[2] = {0, 0};           // frames the per-batch buffers of a set hold at the moment
    std::vector<float*> Gb[2];      // Gb[set][l], l = 1..L : gcap[set] images of level l
    std::vector<float*> partE;      // [l] frame-chunk partial maxima / arg-maxima of level l (launch_level_sep)
    std::vector<int32_t*> partI;
    std::vector<size_t> part_cap;   // elements allocated in partE[l] / partI[l]
    std::vector<uint16_t*> sbOrder; // [l] super-block order of the level's interior launch (LevelArgs::sb_order), device
    std::vector<int> sbGroups;      // [l] entries of sbOrder[l] (a multiple of 8)
    std::vector<size_t> gstride;    // floats between frames in Gb[.][l]
    // MI_ARITH_SEPARABLE, level pairs (kernels_sep.hpp "PAIR", run_batch): a batch that runs levels l and l + 1 as a pair keeps
    // gray(G_{l+1}) -- one float per pixel -- in Gb[set][l+1] and the three-channel G_{l+1} of its LAST frame in
    // Gkeep[set][l+1] (the tap; allocated when the first such batch runs)
    std::vector<float*> Gkeep[2];
    unsigned last_kept = 0;         // bit l: the most recent batch kept level l's Gaussian images as gray + one frame
    std::vector<uint8_t*> tileFlag; // [l][level-l tile] the pair's tile-by-tile payload pass left this tile to the per-quad kernels
    void* ring = nullptr;           // staging ring for host-pushed frames (bcap frames, in_dtype)
    size_t frame_bytes = 0;
    int pending = 0;                // frames staged in the ring, not yet processed
    int last_nb = 0;                // size of the most recently processed batch
    int last_set = 0;
    long batch_no = 0;
    int32_t* lev[2] = {nullptr, nullptr};   // base batch scratch, per set
    uint32_t* cnt[2] = {nullptr, nullptr};
    float* logp[2] = {nullptr, nullptr};
    float* feat[2] = {nullptr, nullptr};    // (entropy, deviation) of every (frame, pixel) of the batch
    hipStream_t st1 = nullptr, st2 = nullptr;
    // st3 (round 6): the stream of the payload passes.  A level's payload pass only needs that level's arg-max; the next level's
    // energy pass does not need the payload -- gather-bound kernel beside streaming kernel instead of one after the other.
    // It IS st1, the border tiles' stream (idle behind level 0 but for a few short launches), not a stream of its own: the
    // runtime maps streams onto four hardware queues, and a fifth stream in the process shares one -- config 4 (stacker +
    // estimator) went from 34.6 to 41.6 ms with a separate payload stream because the estimator's stream then queued behind
    // the warps'; the 256-frame job measures the same either way (docs/studies.md, round 6).
    hipStream_t st3 = nullptr;
    std::vector<hipEvent_t> evPayIn;   // [set][level]: the level's energy pass (all streams) is through
    hipEvent_t evPay[2] = {nullptr, nullptr};   // the batch's payload passes are through
    std::vector<hipEvent_t> evGrp;     // pair, piped: level 0's launch of frame group g is through (level 1's energy launch of g waits)
    hipEvent_t evL0i[2] = {nullptr, nullptr}, evL0b[2] = {nullptr, nullptr}, evRest[2] = {nullptr, nullptr};
    hipEvent_t evL0done[2] = {nullptr, nullptr};   // level-0 state of the batch is final (separable: after its payload pass)
    std::vector<hipEvent_t> evLvl;  // [set][level][interior|border]: per-level joins of st2 and st1
    bool streams_dirty = false;     // work may be in flight on st1/st2
    // host-frame upload: pinned bounce buffers + a copy stream, so push_frame returns after a
    // host memcpy and the PCIe transfer overlaps decoding of the next frame and the kernels
    static constexpr int NPIN = 3;
    void* pin[NPIN] = {nullptr, nullptr, nullptr};
    hipEvent_t evPin[NPIN] = {nullptr, nullptr, nullptr};   // H2D out of pin[i] finished
    hipStream_t stc = nullptr;                              // copy stream
    hipEvent_t evCopied = nullptr;                          // all H2D of the staged batch done
    hipEvent_t evRingFree = nullptr;                        // level-0 kernels finished reading the ring
    hipEvent_t evInput = nullptr;                           // device pushes: the frames are complete in s->stream order
    long pin_no = 0;
    // zero-copy uploads (mi_stack_push_frame_pinned): one event per upload in a ring, for mi_stack_wait_uploads
    static constexpr int NUP = MI_NUP;
    hipEvent_t evUp[NUP] = {};
    long up_no = 0;
};

inline TiledState*& tstate(mi_stack* s) { return *reinterpret_cast<TiledState**>(&s->tiled); }
inline TiledState* tstate(const mi_stack* s) { return reinterpret_cast<TiledState*>(s->tiled); }

bool tiled_available() { return true; }

void dev_release(mi_stack* s, void* p) {
    if (!p) return;
    for (size_t i = 0; i < s->allocs.size(); ++i)
        if (s->allocs[i] == p) {
            s->allocs[i] = s->allocs.back();
            s->allocs.pop_back();
            break;
        }
    (void)hipFree(p);
}

int tiled_sync_all(mi_stack* s);

// bytes of per-batch buffers one frame needs in a set (its Gaussian levels 1..L and the base scratch)
size_t tiled_bytes_per_frame(const mi_stack* s) {
    const TiledState* t = tstate(s);
    size_t b = 0;
    for (int l = 1; l <= s->L; ++l) b += t->gstride[l] * sizeof(float);
    return b + (size_t)s->lh[s->L] * s->lw[s->L] * 12 + (size_t)s->nlevels_hist * 8;
}

// make the per-batch buffers of `set` hold `nb` frames (grows only; growing waits for the work in flight)
int tiled_reserve(mi_stack* s, int set, int nb) {
    TiledState* t = tstate(s);
    if (nb <= t->gcap[set]) return MI_OK;
    int rc;
    if (t->gcap[set] > 0) {
        if ((rc = tiled_sync_all(s))) return rc;
        MI_HIP(hipStreamSynchronize(s->stream));
        for (int l = 1; l <= s->L; ++l) { dev_release(s, t->Gb[set][l]); t->Gb[set][l] = nullptr; }
        dev_release(s, t->lev[set]); dev_release(s, t->cnt[set]); dev_release(s, t->logp[set]); dev_release(s, t->feat[set]);
        t->lev[set] = nullptr; t->cnt[set] = nullptr; t->logp[set] = nullptr; t->feat[set] = nullptr;
        t->gcap[set] = 0;
    }
    for (int l = 1; l <= s->L; ++l)
        if ((rc = dev_alloc_t(s, &t->Gb[set][l], t->gstride[l] * nb))) return rc;
    const size_t npb = (size_t)s->lh[s->L] * s->lw[s->L];
    if ((rc = dev_alloc_t(s, &t->lev[set], npb * nb))) return rc;
    if ((rc = dev_alloc_t(s, &t->cnt[set], (size_t)s->nlevels_hist * nb))) return rc;
    if ((rc = dev_alloc_t(s, &t->logp[set], (size_t)s->nlevels_hist * nb))) return rc;
    if ((rc = dev_alloc_t(s, &t->feat[set], 2 * npb * nb))) return rc;
    t->gcap[set] = nb;
    return MI_OK;
}

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
    // the short, latency-bound kernels (level-0 border frame, coarser levels, base) get the
    // highest priority so they are dispatched ahead of the bulk level-0 interior kernel
    // they run beside; otherwise they starve and become the critical path.
    int prio_lo = 0, prio_hi = 0;
    MI_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    if (study_env("MI_SERIAL", 0)) t->st1 = t->st2 = t->st3 = s->stream;   // -DMI_STUDY: every kernel alone on the GPU
    else {
        const int bd = study_env("MI_BD_PRIO", 0), co = study_env("MI_CO_PRIO", 0);   // 0 high, 1 normal, 2 low
        MI_HIP(hipStreamCreateWithPriority(&t->st1, hipStreamNonBlocking, bd == 0 ? prio_hi : bd == 1 ? 0 : prio_lo));
        MI_HIP(hipStreamCreateWithPriority(&t->st2, hipStreamNonBlocking, co == 0 ? prio_hi : co == 1 ? 0 : prio_lo));
        const int ps = study_env("MI_PAYLOAD_STREAM", 2);   // -DMI_STUDY: 0 = in line with the levels (rounds 2-5), 1 = a stream of their own
        if (ps == 1) MI_HIP(hipStreamCreateWithPriority(&t->st3, hipStreamNonBlocking, co == 0 ? prio_hi : co == 1 ? 0 : prio_lo));
        else t->st3 = ps == 2 ? t->st1 : t->st2;
        if (study_env("MI_BD_PRIO", 0)) fprintf(stderr, "priority range lo=%d hi=%d\n", prio_lo, prio_hi);
    }
    t->gstride.assign(L + 1, 0);
    MI_HIP(hipEventCreateWithFlags(&t->evInput, hipEventDisableTiming));
    for (int set = 0; set < 2; ++set) {
        MI_HIP(hipEventCreateWithFlags(&t->evL0i[set], hipEventDisableTiming));
        MI_HIP(hipEventCreateWithFlags(&t->evL0b[set], hipEventDisableTiming));
        MI_HIP(hipEventCreateWithFlags(&t->evRest[set], hipEventDisableTiming));
        MI_HIP(hipEventCreateWithFlags(&t->evL0done[set], hipEventDisableTiming));
        MI_HIP(hipEventCreateWithFlags(&t->evPay[set], hipEventDisableTiming));
        for (int i = 0; i <= L; ++i) {
            hipEvent_t e;
            MI_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            t->evPayIn.push_back(e);
        }
        for (int i = 0; i < 2 * (L + 1); ++i) {
            hipEvent_t e;
            MI_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            t->evLvl.push_back(e);
        }
        t->Gb[set].assign(L + 1, nullptr);
        t->Gkeep[set].assign(L + 1, nullptr);
        for (int l = 1; l <= L; ++l) t->gstride[l] = (size_t)s->lh[l] * s->lw[l] * 3;
        // The per-batch buffers are allocated by the first batch that needs them (tiled_reserve in run_batch): a stack
        // that arrives as one resident push never uses the second set, and growing a set later (free + allocate behind a
        // device synchronisation) cost 0.7 s for a 256-frame push.  A caller that names its batch size pushes batch after
        // batch: both sets up front, so that no allocation synchronises the device in the middle of its pipeline.
        if (s->p.batch_frames > 0) {
            int rc = tiled_reserve(s, set, t->bcap);
            if (rc) return rc;
        }
    }
    t->tileFlag.assign(L + 1, nullptr);
    t->partE.assign(L + 1, nullptr);
    t->partI.assign(L + 1, nullptr);
    t->part_cap.assign(L + 1, 0);
    return MI_OK;
}

int tiled_sync_all(mi_stack* s) {
    TiledState* t = tstate(s);
    if (!t) return MI_OK;
    if (t->stc) MI_HIP(hipStreamSynchronize(t->stc));
    if (t->st1) MI_HIP(hipStreamSynchronize(t->st1));
    if (t->st2) MI_HIP(hipStreamSynchronize(t->st2));
    if (t->st3 && t->st3 != t->st2 && t->st3 != t->st1) MI_HIP(hipStreamSynchronize(t->st3));
    t->streams_dirty = false;
    return MI_OK;
}

void tiled_destroy(mi_stack* s) {
    TiledState* t = tstate(s);
    if (!t) return;
    for (int set = 0; set < 2; ++set) {
        if (t->evL0i[set]) (void)hipEventDestroy(t->evL0i[set]);
        if (t->evL0b[set]) (void)hipEventDestroy(t->evL0b[set]);
        if (t->evRest[set]) (void)hipEventDestroy(t->evRest[set]);
        if (t->evL0done[set]) (void)hipEventDestroy(t->evL0done[set]);
        if (t->evPay[set]) (void)hipEventDestroy(t->evPay[set]);
    }
    for (auto e : t->evLvl) (void)hipEventDestroy(e);
    for (auto e : t->evPayIn) (void)hipEventDestroy(e);
    for (auto e : t->evGrp) (void)hipEventDestroy(e);
    for (int i = 0; i < TiledState::NUP; ++i)
        if (t->evUp[i]) (void)hipEventDestroy(t->evUp[i]);
    for (int i = 0; i < TiledState::NPIN; ++i) {
        if (t->pin[i]) (void)hipHostFree(t->pin[i]);
        if (t->evPin[i]) (void)hipEventDestroy(t->evPin[i]);
    }
    if (t->evCopied) (void)hipEventDestroy(t->evCopied);
    if (t->evRingFree) (void)hipEventDestroy(t->evRingFree);
    if (t->evInput) (void)hipEventDestroy(t->evInput);
    if (t->stc) (void)hipStreamDestroy(t->stc);
    if (t->st1 && t->st1 != s->stream) (void)hipStreamDestroy(t->st1);
    if (t->st3 && t->st3 != s->stream && t->st3 != t->st2 && t->st3 != t->st1) (void)hipStreamDestroy(t->st3);
    if (t->st2 && t->st2 != s->stream) (void)hipStreamDestroy(t->st2);
    delete t;
    tstate(s) = nullptr;
}

int tiled_reset(mi_stack* s) {
    TiledState* t = tstate(s);
    t->pending = 0;
    t->last_nb = 0;
    t->batch_no = 0;
    return MI_OK;
}

int tiled_pending(const mi_stack* s) { return tstate(s) ? tstate(s)->pending : 0; }
// the side streams of a tiled handle (null for the simple implementation): mi_align_stack_device borrows them as warp lanes
// while a batch buffer fills -- they are idle then, and the process stays inside its four hardware queues
void tiled_side_streams(const mi_stack* s, hipStream_t out[2]) {
    const TiledState* t = tstate(s);
    out[0] = t && t->st1 != s->stream ? t->st1 : nullptr;
    out[1] = t && t->st2 != s->stream ? t->st2 : nullptr;
}

const float* tiled_last_gauss(mi_stack* s, int level) {
    TiledState* t = tstate(s);
    if (s->p.impl != MI_IMPL_TILED) return s->G[level];
    int last = t->last_nb > 0 ? t->last_nb - 1 : 0;
    if ((t->last_kept >> level) & 1u) return t->Gkeep[t->last_set][level];   // (Gb[.][level] holds gray(G_level) then)
    return t->Gb[t->last_set][level] + (size_t)last * t->gstride[level];
}

template <typename Kern>
int set_lds_once(Kern kern, size_t lds) {
    MI_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return MI_OK;
}

constexpr int ilcm(int a, int b) {
    int x = a, y = b;
    while (y) { int t = x % y; x = y; y = t; }
    return a / x * b;
}

// MI_ARITH_SEPARABLE: interior and border tiles of level l on the separable kernel (kernels_sep.hpp); one tile
// grid (28 x 56, origin at the image corner) for both, the interior rectangle = the whole tiles whose 6-pixel
// halo stays inside the image.
// How a level with `tiles` workgroups walks the `nb` frames of a batch.  A workgroup visits its frames one after the
// other, so (a) a level with few tiles is latency-bound unless the batch is cut into chunks that run side by side
// (blockIdx.y; about four rounds of workgroups for the GPU's 768 slots, chunks of 16 to 32 frames, partial maxima merged
// afterwards), and (b) over a long walk neighbouring workgroups drift apart in time and stop sharing their halos in L2
// (measured on 24 MP frames: 256 frames in one launch cost 3.1 ms per 32 frames, launches of 32 frames 2.9 ms, of 16
// frames 2.7 ms, of 8 frames 2.75 ms), so a level with many tiles runs as consecutive launches of 16 frames.  Returns the frames per chunk / per launch.
constexpr int SEP_LAUNCH_FRAMES = 16;
inline int level_chunk_frames(int nb, int tiles, bool* parallel) {
    static const int on = study_env("MI_CHUNK", 1);           // -DMI_STUDY: 0 = never in parallel chunks
    static const int lf = study_env("MI_LAUNCH_FRAMES", SEP_LAUNCH_FRAMES);
    static const int par = study_env("MI_PAR_TILES", 3072);   // -DMI_STUDY: tile count below which chunks run side by side
    const int c = par / std::max(tiles, 1);
    *parallel = on && c > 1 && nb >= 32;
    if (!*parallel) return std::min(nb, lf);
    return std::min(lf, std::max(16, cdiv(cdiv(nb, c), 4) * 4));
}

// Super-block order of an interior launch: longest-processing-time-first assignment of the super-blocks (weight = tiles
// inside the interior rectangle) to the eight XCDs; group g = 8 * round + XCD of the launch takes order[g].
inline int build_sb_order(mi_stack* s, int l, int nty, int ntx, int sbw = SB, int sbh = SB) {
    TiledState* t = tstate(s);
    if ((int)t->sbOrder.size() <= l) { t->sbOrder.resize(l + 1, nullptr); t->sbGroups.resize(l + 1, 0); }
    if (t->sbOrder[l]) return MI_OK;
    const int sbx = cdiv(ntx, sbw), sby = cdiv(nty, sbh), nsb = sbx * sby;
    if (nsb >= 0xFFFF) return MI_OK;   // (never: 65 535 super-blocks are 4 M tiles) keep the plain order
    std::vector<std::pair<int, int>> sb(nsb);   // (tiles, index)
    for (int S = 0; S < nsb; ++S) {
        const int y = S / sbx, x = S - y * sbx;
        sb[S] = {std::min(sbh, nty - y * sbh) * std::min(sbw, ntx - x * sbw), S};
    }
    std::stable_sort(sb.begin(), sb.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first > b.first; });
    std::vector<int> lists[8];
    long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (const auto& e : sb) {
        int best = 0;
        for (int x = 1; x < 8; ++x)
            if (load[x] < load[best]) best = x;
        lists[best].push_back(e.second);
        load[best] += e.first;
    }
    size_t rounds = 0;
    for (auto& v : lists) {
        std::sort(v.begin(), v.end());   // an XCD walks its super-blocks in raster order
        rounds = std::max(rounds, v.size());
    }
    std::vector<uint16_t> order(rounds * 8, (uint16_t)0xFFFF);
    for (int x = 0; x < 8; ++x)
        for (size_t r = 0; r < lists[x].size(); ++r) order[r * 8 + x] = (uint16_t)lists[x][r];
    uint16_t* d = nullptr;
    int rc = dev_alloc_t(s, &d, order.size());
    if (rc) return rc;
    MI_HIP(hipMemcpy(d, order.data(), order.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    t->sbOrder[l] = d;
    t->sbGroups[l] = (int)order.size();
    return MI_OK;
}

// Launch the interior and border kernels of level l for `nb` frames.
//   interior kernel: tile config A (TH, TW, NT, padded LDS), stream st_in
//   border kernel  : tile config B (BH, BW, BNT, unpadded LDS: small enough to co-reside
//                    with two level-0 interior workgroups on one CU), stream st_bd
template <typename TIn, bool FMA, int TH, int TW, int NT, bool PADA, int BH, int BW, int BNT, bool COARSE_NAME = false>
int launch_level(mi_stack* s, int l, int set, const void* src, size_t src_stride, int nb,
                 hipStream_t st_in, hipStream_t st_bd, hipEvent_t ev_bd) {
    using GA = TileGeom<TH, TW, NT, PADA>;
    using GB = TileGeom<BH, BW, BNT, false>;
    TiledState* t = tstate(s);
    LevelArgs a{};
    a.src = src;
    a.src_stride = src_stride;
    a.gnext = t->Gb[set][l + 1];
    a.gnext_stride = t->gstride[l + 1];
    a.nframes = nb;
    a.h = s->lh[l];
    a.w = s->lw[l];
    a.hn = s->lh[l + 1];
    a.wn = s->lw[l + 1];
    // interior rectangle: whole 6-pixel-haloed patches inside the image, aligned to both tilings
    constexpr int AY = ilcm(TH, BH), AX = ilcm(TW, BW);
    if (TH % BH == 0 && TW % BW == 0) {
        // origin on the border tiling, whole interior tiles from there
        a.iy0 = cdiv(6, BH) * BH;
        a.ix0 = cdiv(6, BW) * BW;
        a.iy1 = a.iy0 + (a.h - 6 - a.iy0 > 0 ? (a.h - 6 - a.iy0) / TH * TH : 0);
        a.ix1 = a.ix0 + (a.w - 6 - a.ix0 > 0 ? (a.w - 6 - a.ix0) / TW * TW : 0);
    } else {
        a.iy0 = cdiv(6, AY) * AY;
        a.ix0 = cdiv(6, AX) * AX;
        a.iy1 = (a.h - 6) / AY * AY;
        a.ix1 = (a.w - 6) / AX * AX;
    }
    if (a.iy1 <= a.iy0 || a.ix1 <= a.ix0) a.iy0 = a.iy1 = a.ix0 = a.ix1 = 0;
    const int nyi = (a.iy1 - a.iy0) / TH, nxi = (a.ix1 - a.ix0) / TW;
    a.best_e = s->bestE[l];
    a.best_lap = s->bestLap[l];
    a.best_idx = s->bestIdx[l];
    a.first = s->n_pushed == 0;
    a.frame_idx0 = s->first_index + s->n_pushed;
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 3; ++j) a.K.c[i == 0 ? j : (i == 1 ? 2 + j : 5)] = s->K.k[i * 5 + j];
    a.ablate = study_env("MI_ABLATE", 0);   // -DMI_STUDY builds only (results are wrong when set)
#ifdef MI_PHASE_CLOCK
    static unsigned long long* dbg_dev = nullptr;
    if (l == 0) {
        if (!dbg_dev) {
            MI_HIP(hipMalloc(&dbg_dev, 16 * 16 * 8));
            MI_HIP(hipMemset(dbg_dev, 0, 16 * 16 * 8));
        } else {   // print the previous launch's numbers
            unsigned long long hbuf[16 * 16];
            MI_HIP(hipMemcpy(hbuf, dbg_dev, sizeof hbuf, hipMemcpyDeviceToHost));
            MI_HIP(hipMemset(dbg_dev, 0, 16 * 16 * 8));
            static const char* nm[9] = {"stage", "bar1", "prefetch", "reduce", "bar2", "gnstore", "lapq", "bar3", "energy"};
            for (int wv = 0; wv < NT / 64; ++wv) {
                if (!hbuf[wv * 16 + 15]) continue;
                fprintf(stderr, "wave %d:", wv);
                for (int i = 0; i < 9; ++i)
                    fprintf(stderr, " %s %.0f", nm[i], (double)hbuf[wv * 16 + i] / (double)hbuf[wv * 16 + 15] / nb);
                fprintf(stderr, "  (cycles per frame)\n");
            }
        }
        a.dbg = dbg_dev;
    } else a.dbg = nullptr;
#endif
    const size_t ldsA = (size_t)GA::LDS_FLOATS * sizeof(float), ldsB = (size_t)GB::LDS_FLOATS * sizeof(float);
    auto kin = COARSE_NAME ? level_fused_coarse<TIn, FMA, true, TH, TW, NT, PADA> : level_fused<TIn, FMA, true, TH, TW, NT, PADA>;
    auto kbd = level_fused<TIn, FMA, false, BH, BW, BNT, false>;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        int rc;
        if ((rc = set_lds_once(kin, ldsA)) || (rc = set_lds_once(kbd, ldsB))) return rc;
        attr_set = true;
    }
    // algorithmic bytes of this level pass, SURVEY.md 8(d) attribution: read G_l once,
    // write G_{l+1} once, read G_{l+1} once as the expand source.
    const double bytes = ((double)(l == 0 ? dtype_size(s->p.in_dtype) : 4) * 3.0 * a.h * a.w +
                          24.0 * a.hn * a.wn) * nb;
    const double frac_in = (double)(a.iy1 - a.iy0) * (a.ix1 - a.ix0) / ((double)a.h * a.w);
    // Frames per launch / per chunk exactly as for the separable kernel (level_chunk_frames): levels with many tiles as
    // consecutive launches of 16 frames, levels with few tiles as frame chunks side by side + merge_chunks.
    const int tbx = cdiv(a.w, BW), tby = cdiv(a.h, BH);
    const int nborder = tbx * tby - ((a.iy1 - a.iy0) / BH) * ((a.ix1 - a.ix0) / BW);
    const int ntiles = nyi * nxi + nborder;
    const size_t npx = (size_t)a.h * a.w;
    bool parallel = false;
    const int fc = level_chunk_frames(nb, ntiles, &parallel);
    const int nchunks = parallel ? cdiv(nb, fc) : 1;
    if (nchunks > 1) {
        const size_t need = npx * (size_t)(nchunks - 1);
        if (t->part_cap[l] < need) {   // grows only; the buffers may still be in use by the previous batch
            int rc;
            if ((rc = tiled_sync_all(s))) return rc;
            MI_HIP(hipStreamSynchronize(s->stream));
            dev_release(s, t->partE[l]);
            dev_release(s, t->partI[l]);
            t->partE[l] = nullptr;
            t->partI[l] = nullptr;
            t->part_cap[l] = 0;
            if ((rc = dev_alloc_t(s, &t->partE[l], need)) || (rc = dev_alloc_t(s, &t->partI[l], need))) return rc;
            t->part_cap[l] = need;
        }
        a.part_e = t->partE[l];
        a.part_idx = t->partI[l];
        a.part_stride = npx;
    }
    const int first = a.first, idx0 = a.frame_idx0;
    const int step = parallel ? nb : fc, nlaunch = cdiv(nb, step);
    auto frames_of = [&](int f0) {
        const int nf = parallel ? nb : std::min(fc, nb - f0);
        a.src = (const char*)src + (size_t)f0 * src_stride;
        a.gnext = t->Gb[set][l + 1] + (size_t)f0 * a.gnext_stride;
        a.nframes = nf;
        a.chunk_frames = parallel ? fc : nf;
        a.first = first && f0 == 0;
        a.frame_idx0 = idx0 + f0;
    };
    if (nborder > 0 && !MI_ABL(256)) {   // border first: its few, latency-bound workgroups should claim their slots early
        ProfScope ps(s, MI_PROF_LEVEL, bytes * (1.0 - frac_in), st_bd);
        ps.r.launches = nlaunch;
        for (int f0 = 0; f0 < nb; f0 += step) {
            frames_of(f0);
            hipLaunchKernelGGL(kbd, dim3(nborder, nchunks), dim3(BNT), ldsB, st_bd, a);
        }
    }
    if (!MI_ABL(512) && nyi > 0) {
        const int nsb = cdiv(nxi, SB) * cdiv(nyi, SB);
        ProfScope ps(s, l == 0 ? MI_PROF_LEVEL0 : MI_PROF_LEVEL, bytes * frac_in, st_in);
        ps.r.launches = nlaunch;
        for (int f0 = 0; f0 < nb; f0 += step) {
            frames_of(f0);
            hipLaunchKernelGGL(kin, dim3(cdiv(nsb, 8) * 8 * SB * SB, nchunks), dim3(NT), ldsA, st_in, a);
        }
    }
    if (nchunks > 1) {
        MI_HIP(hipEventRecord(ev_bd, st_bd));
        MI_HIP(hipStreamWaitEvent(st_in, ev_bd, 0));
        ProfScope ps(s, MI_PROF_LEVEL, 0.0, st_in);
        hipLaunchKernelGGL(merge_chunks, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, st_in, s->bestE[l], s->bestIdx[l],
                           t->partE[l], t->partI[l], npx, nchunks - 1, npx);
    }
    return MI_OK;
}

// `ev_bd`: recorded on st_bd behind the border kernel when the level ran in chunks (the merge on st_in waits for it).
// `f_begin`, `f_end`: launch the frames [f_begin, f_end) of the batch only (levels that run as consecutive launches; the
// interleaved level-0 / level-1 schedule of run_batch) -- the whole batch by default.
// `MF`: level 0 of 8 / 16-bit frames with the reduce on the matrix pipe (kernels_sep.hpp, MI_SEP_MFMA): its own tile height,
// for the interior and the border launch alike.
// `info` (optional): what run_batch needs to finish the level -- how many chunk partials sep_payload has to fold (it then takes
// merge_chunks' place) and whether border tiles were launched on st_bd (only then the streams have to join).  `ev_sync`
// (optional): recorded on st_in and waited for on st_bd in front of a border launch (everything st_in has done so far).
// `PM` (level pairs, kernels_sep.hpp "PAIR"): 0 = a level on its own; 1 = the first level of a pair (level_sep_pair: writes
// gray(G_{l+1}) into Gb[set][l+1] -- one float per pixel --, G_{l+2} into Gb[set][l+2] and the three-channel G_{l+1} of the
// batch's last frame into Gkeep[set][l+1]); 2 = the second level of a pair (level_sep_e: `src` = that gray, reads G_{l+1});
// 3 = the pair's payload pass tile by tile (level_sep_pl, one launch over the whole batch: fills bestLap[l] and bestLap[l+1]
// of the tiles with few distinct winners, flags the others in tileFlag).
struct SepLevelInfo { int nparts = 0; bool border = false; };
template <typename TIn, bool L0_NAME, bool MF = false, int PM = 0>
int launch_level_sep(mi_stack* s, int l, int set, const void* src, size_t src_stride, int nb, hipStream_t st_in,
                     hipStream_t st_bd, hipEvent_t ev_bd, int f_begin = 0, int f_end = -1, SepLevelInfo* info = nullptr,
                     hipEvent_t ev_sync = nullptr) {
    constexpr int TH = MF ? SEP_MF_TH : MI_SEP_TH, NT = MF ? SEP_MF_NT : sep_nt<TIn>();
    using SG = SepGeom<TH, NT>;
    constexpr int TW = SG::TW;
    TiledState* t = tstate(s);
    LevelArgs a{};
    a.src = src;
    a.src_stride = src_stride;
#ifdef MI_STUDY_SAME_FRAME   // study: every frame of a level-0 launch reads frame 0's addresses (L2 / Infinity Cache instead of HBM)
    if (l == 0) a.src_stride = 0;
#endif
    a.gnext = t->Gb[set][l + 1];
    a.gnext_stride = t->gstride[l + 1];
    a.nframes = nb;
    a.h = s->lh[l];
    a.w = s->lw[l];
    a.hn = s->lh[l + 1];
    a.wn = s->lw[l + 1];
    a.g1_keep = -1;
    const size_t gray_stride = (size_t)a.hn * a.wn;
    if constexpr (PM == 1 || PM == 3) {
        static_assert(!MF, "the matrix-pipe reduce has no pair form");
        if (l + 2 > s->L) return fail(MI_ERR_INVALID, "level pair at level %d of %d", l, s->L);
        if (!t->Gkeep[set][l + 1]) {   // (first pair batch of this level and set: one image)
            int rc = dev_alloc_t(s, &t->Gkeep[set][l + 1], t->gstride[l + 1]);
            if (rc) return rc;
        }
        a.gnext = t->Gkeep[set][l + 1];
        a.gnext_stride = 0;
        a.gray1_stride = gray_stride;
        a.g2_stride = t->gstride[l + 2];
        a.hn2 = s->lh[l + 2];
        a.wn2 = s->lw[l + 2];
    }
    if constexpr (PM == 3) {
        a.g2 = t->Gb[set][l + 2];
        a.idx1 = s->bestIdx[l + 1];
        a.lap1 = s->bestLap[l + 1];
        a.tile_flag = t->tileFlag[l];
    }
    // The "interior" launch covers every tile whose staged patch may be mirrored into place (kernels_sep.hpp, edge tiles):
    // all of the grid, except the tile rows / columns that reach an ODD far edge (those stay with the border kernel), and
    // nothing at all on levels too small for a single reflection per side.
    static const int edge_fold = study_env("MI_EDGE_FOLD", 1);   // -DMI_STUDY: 0 = the round-2 interior / border split
    if (edge_fold && a.h >= 16 && a.w >= 16) {
        a.iy0 = a.ix0 = 0;
        a.iy1 = (a.h & 1) ? std::max(0, (a.h - 6) / TH * TH) : cdiv(a.h, TH) * TH;
        a.ix1 = (a.w & 1) ? std::max(0, (a.w - 6) / TW * TW) : cdiv(a.w, TW) * TW;
    } else {
        a.iy0 = cdiv(6, TH) * TH;
        a.ix0 = cdiv(6, TW) * TW;
        a.iy1 = (a.h - 6) / TH * TH;
        a.ix1 = (a.w - 6) / TW * TW;
    }
    if (a.iy1 <= a.iy0 || a.ix1 <= a.ix0) a.iy0 = a.iy1 = a.ix0 = a.ix1 = 0;
    const int nyi = (a.iy1 - a.iy0) / TH, nxi = (a.ix1 - a.ix0) / TW;
    a.best_e = s->bestE[l];
    a.best_lap = s->bestLap[l];
    a.best_idx = s->bestIdx[l];
    a.first = s->n_pushed == 0;
    a.frame_idx0 = s->first_index + s->n_pushed;
    for (int i = 0; i < 3; ++i) a.k1d[i] = s->k1d[i];
    for (int i = 0; i < 4; ++i) a.rk[i] = s->rk[i];
    a.mfma_ok = s->mfma_ok;
    a.ablate = study_env("MI_ABLATE", 0);   // -DMI_STUDY builds only (results are wrong when set)
#ifdef MI_PHASE_CLOCK
    static unsigned long long* dbg_dev = nullptr;
    if (l == 0) {
        if (!dbg_dev) {
            MI_HIP(hipMalloc(&dbg_dev, 16 * 16 * 8));
            MI_HIP(hipMemset(dbg_dev, 0, 16 * 16 * 8));
        } else {   // print the previous level-0 pass's numbers (cycles per wave and frame)
            unsigned long long hbuf[16 * 16];
            MI_HIP(hipMemcpy(hbuf, dbg_dev, sizeof hbuf, hipMemcpyDeviceToHost));
            MI_HIP(hipMemset(dbg_dev, 0, 16 * 16 * 8));
            static const char* nm[10] = {"stage", "bar1", "pfissue", "P1", "bar2", "P2", "bar3", "P3", "bar4", "P4"};
            for (int wv = 0; wv < NT / 64; ++wv) {
                if (!hbuf[wv * 16 + 15]) continue;
                fprintf(stderr, "wave %d:", wv);
                double tot = 0;
                for (int i = 0; i < 10; ++i) {
                    const double v = (double)hbuf[wv * 16 + i] / (double)hbuf[wv * 16 + 15] / std::min(nb, SEP_LAUNCH_FRAMES);
                    tot += v;
                    fprintf(stderr, " %s %.0f", nm[i], v);
                }
                fprintf(stderr, "  total %.0f (clock ticks per frame)\n", tot);
            }
        }
        a.dbg = dbg_dev;
    } else a.dbg = nullptr;
#endif
    const size_t lds = (size_t)(PM == 3 ? sep_pl_lds_floats<TIn, TH, NT>(false) : PM == 2 ? sep_e_lds_floats<TH, NT>()
                                        : SG::LDS_FLOATS + (PM == 1 ? 3 * SG::NH * SG::NW : 0)) * sizeof(float);   // border tiles
    const size_t lds_in = (size_t)(PM == 3 ? sep_pl_lds_floats<TIn, TH, NT>(true) : PM == 2 ? sep_e_lds_floats<TH, NT>()
                                           : MF ? SG::lds_floats_mf((int)sizeof(TIn))
                                                : SG::lds_floats((int)sizeof(TIn), true) + (PM == 1 && sizeof(TIn) <= 2 ? 3 * SG::NH * SG::NW : 0)) * sizeof(float);   // interior tiles
    void (*kin)(LevelArgs);
    void (*kbd)(LevelArgs);
    if constexpr (PM == 1) { kin = level_sep_pair<TIn, true, TH, NT>; kbd = level_sep_pair<TIn, false, TH, NT>; }
    else if constexpr (PM == 2) { kin = level_sep_e<true, TH, NT>; kbd = level_sep_e<false, TH, NT>; }
    else if constexpr (PM == 3) { kin = level_sep_pl<TIn, true, TH, NT>; kbd = level_sep_pl<TIn, false, TH, NT>; }
    else {
        if constexpr (MF) kin = level_sep_mf<TIn, TH, NT>;
        else kin = L0_NAME ? level_sep<TIn, true, TH, NT> : level_sep_coarse<TIn, true, TH, NT>;
        kbd = level_sep<TIn, false, TH, NT>;
    }
    static thread_local bool attr_set = false;
    if (!attr_set) {
        int rc;
        if ((rc = set_lds_once(kin, lds_in)) || (rc = set_lds_once(kbd, lds))) return rc;
        attr_set = true;
    }
    const double bytes = PM == 3 ? 0.0 : ((double)(l == 0 ? dtype_size(s->p.in_dtype) : 4) * 3.0 * a.h * a.w + 24.0 * a.hn * a.wn) * nb;
    const double frac_in = (double)(std::min(a.iy1, a.h) - a.iy0) * (std::min(a.ix1, a.w) - a.ix0) / ((double)a.h * a.w);
    const int ntiles = cdiv(a.w, TW) * cdiv(a.h, TH);
    const size_t npx = (size_t)a.h * a.w;
    bool parallel = false;
    int fc = level_chunk_frames(nb, ntiles, &parallel);
    if (PM == 3) { parallel = false; fc = nb; }   // the payload pass: one launch, the workgroup picks its frames itself
    const int nchunks = parallel ? cdiv(nb, fc) : 1;
    if (nchunks > 1) {
        const size_t need = npx * (size_t)(nchunks - 1);
        if (t->part_cap[l] < need) {   // grows only; the buffers may still be in use by the previous batch
            int rc;
            if ((rc = tiled_sync_all(s))) return rc;
            MI_HIP(hipStreamSynchronize(s->stream));
            dev_release(s, t->partE[l]);
            dev_release(s, t->partI[l]);
            t->partE[l] = nullptr;
            t->partI[l] = nullptr;
            t->part_cap[l] = 0;
            if ((rc = dev_alloc_t(s, &t->partE[l], need)) || (rc = dev_alloc_t(s, &t->partI[l], need))) return rc;
            t->part_cap[l] = need;
        }
        a.part_e = t->partE[l];
        a.part_idx = t->partI[l];
        a.part_stride = npx;
    }
    // parallel chunks: one launch over all the frames; otherwise consecutive launches of `fc` frames (the interior and the
    // border launches of a level touch disjoint pixels, so each sequence only has to keep its own order)
    const int nborder = ntiles - nyi * nxi;
    const int nsb = cdiv(nxi, SEP_SBW) * cdiv(nyi, SEP_SBH);
    int ngroups = cdiv(nsb, 8) * 8;   // super-block-sized groups of workgroups of the interior launch
    if (nyi > 0 && !MI_ABL(4096)) {
        int rc = build_sb_order(s, l, nyi, nxi, SEP_SBW, SEP_SBH);
        if (rc) return rc;
        if (t->sbOrder[l]) { a.sb_order = t->sbOrder[l]; ngroups = t->sbGroups[l]; }
    }
    const int first = a.first, idx0 = a.frame_idx0;
    if (f_end < 0 || parallel) { f_begin = 0; f_end = nb; }
    const int step = parallel ? nb : fc, nlaunch = cdiv(f_end - f_begin, step);
    const double part = (double)(f_end - f_begin) / (double)nb;   // share of the batch's bytes this call launches
    auto frames_of = [&](int f0) {
        const int nf = parallel ? nb : std::min(fc, f_end - f0);
        a.src = (const char*)src + (size_t)f0 * src_stride;
#ifdef MI_STUDY_SAME_FRAME
        if (l == 0) a.src = src;
#endif
        if constexpr (PM == 3) {
        } else if constexpr (PM == 1) {
            a.gray1 = t->Gb[set][l + 1] + (size_t)f0 * gray_stride;
            a.g2 = t->Gb[set][l + 2] + (size_t)f0 * a.g2_stride;
            a.g1_keep = nb - 1 >= f0 && nb - 1 < f0 + nf ? nb - 1 - f0 : -1;   // the batch's last frame, if this launch holds it
        } else a.gnext = t->Gb[set][l + 1] + (size_t)f0 * a.gnext_stride;
        a.nframes = nf;
        a.chunk_frames = parallel ? fc : nf;
        a.first = first && f0 == 0;
        a.frame_idx0 = idx0 + f0;
    };
    // one timing-event pair around each stream's sequence of launches (an event record between two kernels of a stream
    // costs a few microseconds of idle GPU: 30 launches per level pass)
    if (info) { info->nparts = nchunks - 1; info->border = nborder > 0; }
    if (nborder > 0 && !MI_ABL(256)) {
        if (ev_sync) {
            MI_HIP(hipEventRecord(ev_sync, st_in));
            MI_HIP(hipStreamWaitEvent(st_bd, ev_sync, 0));
        }
        ProfScope ps(s, MI_PROF_LEVEL, bytes * (1.0 - frac_in) * part, st_bd);
        ps.r.launches = nlaunch;
        for (int f0 = f_begin; f0 < f_end; f0 += step) {
            frames_of(f0);
            hipLaunchKernelGGL(kbd, dim3(nborder, nchunks), dim3(NT), lds, st_bd, a);
        }
    }
    if (nyi > 0) {
        ProfScope ps(s, l == 0 && PM != 3 ? MI_PROF_LEVEL0 : MI_PROF_LEVEL, bytes * frac_in * part, st_in);
        ps.r.launches = nlaunch;
        for (int f0 = f_begin; f0 < f_end; f0 += step) {
            frames_of(f0);
            hipLaunchKernelGGL(kin, dim3(ngroups * SEP_SBW * SEP_SBH, nchunks), dim3(NT), lds_in, st_in, a);
        }
    }
    if (nchunks > 1 && !info) {   // (callers that pass `info` fold the partials in their payload pass)
        MI_HIP(hipEventRecord(ev_bd, st_bd));
        MI_HIP(hipStreamWaitEvent(st_in, ev_bd, 0));
        ProfScope ps(s, MI_PROF_LEVEL, 0.0, st_in);
        hipLaunchKernelGGL(merge_chunks, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, st_in, s->bestE[l], s->bestIdx[l],
                           t->partE[l], t->partI[l], npx, nchunks - 1, npx);
    }
    return MI_OK;
}

// MI_ARITH_SEPARABLE: the winners' Laplacians of level l for the frames of this batch (the level kernel keeps only
// the running maximum and its frame index); reads the batch's G_l (level 0: the frames) and G_{l+1}.
template <typename TIn>
int launch_payload_sep(mi_stack* s, int l, int set, const void* src, size_t src_stride, int nb, hipStream_t st, int nparts = 0) {
    TiledState* t = tstate(s);
    const dim3 blk(32, 8);
    const dim3 grd(cdiv(cdiv(s->lw[l], 2), blk.x), cdiv(cdiv(s->lh[l], 2), blk.y));
    ProfScope ps(s, MI_PROF_LEVEL, 0.0, st);
    hipLaunchKernelGGL((sep_payload<TIn>), grd, blk, 0, st, src, src_stride, (const float*)t->Gb[set][l + 1], t->gstride[l + 1],
                       nb, s->lh[l], s->lw[l], s->lh[l + 1], s->lw[l + 1], s->bestIdx[l],
                       s->first_index + s->n_pushed, s->bestLap[l], s->k1d[0], s->k1d[1], s->k1d[2], s->bestE[l],
                       (const float*)(nparts > 0 ? t->partE[l] : nullptr), (const int32_t*)(nparts > 0 ? t->partI[l] : nullptr),
                       (size_t)s->lh[l] * s->lw[l], nparts);
    return MI_OK;
}

// Level pair (l, l + 1): the payload passes that recompute the winners' G_{l+1} from level l's images `src` (l = 0: the frames,
// TIn; else Gb[set][l], float) -- sep_payload_pair0 / 1
template <typename TIn>
int launch_payload_pair0(mi_stack* s, int l, int set, const void* src, size_t src_stride, int nb, hipStream_t st, int nparts,
                         const uint8_t* tile_flag = nullptr) {
    TiledState* t = tstate(s);
    const dim3 blk(32, 8);
    const dim3 grd(cdiv(cdiv(s->lw[l], 2), blk.x), cdiv(cdiv(s->lh[l], 2), blk.y));
    ProfScope ps(s, MI_PROF_LEVEL, 0.0, st);
    hipLaunchKernelGGL((sep_payload_pair0<TIn>), grd, blk, 0, st, src, src_stride, nb, s->lh[l], s->lw[l], s->lh[l + 1], s->lw[l + 1],
                       s->bestIdx[l], s->first_index + s->n_pushed, s->bestLap[l], s->k1d[0], s->k1d[1], s->k1d[2], s->rk[0],
                       s->rk[1], s->rk[2], s->rk[3], s->bestE[l], (const float*)(nparts > 0 ? t->partE[l] : nullptr),
                       (const int32_t*)(nparts > 0 ? t->partI[l] : nullptr), (size_t)s->lh[l] * s->lw[l], nparts, tile_flag,
                       MI_SEP_TH, 56);
    return MI_OK;
}
template <typename TIn>
int launch_payload_pair1(mi_stack* s, int l, int set, const void* src, size_t src_stride, int nb, hipStream_t st, int nparts,
                         const uint8_t* tile_flag = nullptr) {
    TiledState* t = tstate(s);
    const dim3 blk(32, 8);
    const dim3 grd(cdiv(cdiv(s->lw[l + 1], 2), blk.x), cdiv(cdiv(s->lh[l + 1], 2), blk.y));
    ProfScope ps(s, MI_PROF_LEVEL, 0.0, st);
    hipLaunchKernelGGL((sep_payload_pair1<TIn>), grd, blk, 0, st, src, src_stride, (const float*)t->Gb[set][l + 2], t->gstride[l + 2], nb,
                       s->lh[l], s->lw[l], s->lh[l + 1], s->lw[l + 1], s->lh[l + 2], s->lw[l + 2], s->bestIdx[l + 1],
                       s->first_index + s->n_pushed, s->bestLap[l + 1], s->k1d[0], s->k1d[1], s->k1d[2], s->rk[0], s->rk[1], s->rk[2],
                       s->rk[3], s->bestE[l + 1], (const float*)(nparts > 0 ? t->partE[l + 1] : nullptr),
                       (const int32_t*)(nparts > 0 ? t->partI[l + 1] : nullptr), (size_t)s->lh[l + 1] * s->lw[l + 1], nparts, tile_flag,
                       MI_SEP_TH, 56);
    return MI_OK;
}
// The pair's payload, tile by tile (level_sep_pl) + the per-quad kernels on the tiles it flags: levels l and l + 1 at once,
// behind both levels' energy passes.  Needs unchunked levels (the per-quad kernels alone fold chunk partials) and frame
// numbers that fit the 256-bit winner map.
template <typename TIn>
int launch_payload_pair_tiles(mi_stack* s, int l, int set, const void* src, size_t src_stride, int nb, hipStream_t st) {
    TiledState* t = tstate(s);
    const size_t ntiles = (size_t)cdiv(s->lw[l], 56) * cdiv(s->lh[l], MI_SEP_TH);
    int rc;
    if (!t->tileFlag[l] && (rc = dev_alloc_t(s, &t->tileFlag[l], ntiles))) return rc;
    MI_HIP(hipMemsetAsync(t->tileFlag[l], 0, ntiles, st));
    if ((rc = launch_level_sep<TIn, false, false, 3>(s, l, set, src, src_stride, nb, st, st, nullptr))) return rc;
    if ((rc = launch_payload_pair0<TIn>(s, l, set, src, src_stride, nb, st, 0, t->tileFlag[l]))) return rc;
    return launch_payload_pair1<TIn>(s, l, set, src, src_stride, nb, st, 0, t->tileFlag[l]);
}
// The batch's pair plan: pm[l] = 1: level l is the first level of a pair, 2: the second, 0: on its own.  pair_levels 1 = pairs
// from level 0 on -- (0, 1), (2, 3), ... --, 3 = from level 1 on, 2 = none (both forced plans exist for the tests: every pair
// gives the same bits); 0 = automatic = where it measured faster: the pair (0, 1) for float-32 frames in batches of
// SEP_PAIR_MIN_FRAMES and more.
//  * float-32, level 0: the kernel is bound by memory AND issue, the times add: 30 MB less per frame against the G_2 reduce it
//    takes on -- +0.05 ms per launch of 16 frames; level 1's pass 0.33 -> 0.15 ms per launch; the payload recomputation
//    (1.56 ms per batch at 24 MP) hides behind levels 2+ (off the levels' stream).  Interleaved A/B on three boxes, 256 x 24 MP:
//    +2.5 to +4.5 % (profiles/r06/pair_ab_box*.txt).
//  * 8- / 16-bit frames, level 0: issue-bound kernels, +0.17 / +0.24 ms per launch -- more than level 1 saves: -3 % / -8 %.
//  * deeper pairs -- (2, 3) behind (0, 1) for float-32, (1, 2) and (3, 4) for 8- / 16-bit frames -- measured no gain or a
//    loss (27.5-27.8 -> 27.7-28.1 ms; 22.2 -> 22.7 ms; 24.1 -> 24.9 ms): those levels run in frame chunks, their pairs'
//    payload is the per-quad recomputation, and it ends up as the tail of the batch.
//  * short batches: the once-per-batch payload recomputation outweighs level 1's gain (a 64-frame shard: 11.0 against 8.1 ms).
constexpr int SEP_PAIR_MIN_FRAMES = 192;
#ifndef MI_L1E_PIPE_DEFAULT
#define MI_L1E_PIPE_DEFAULT 0
#endif
inline void sep_pair_plan(const mi_stack* s, int nb, std::vector<int>& pm) {
    pm.assign(std::max(s->L, 1), 0);
    if (!s->sep || s->p.pair_levels == 2) return;
    static const int plan = study_env("MI_PAIR_PLAN", -1);   // -DMI_STUDY: bit l = level l is the first level of a pair
    int l0;
    if (plan >= 0) {
        for (int l = 0; l + 2 <= s->L; ++l)
            if (((plan >> l) & 1) && pm[l] == 0) { pm[l] = 1; pm[l + 1] = 2; }
        return;
    }
    if (s->p.pair_levels == 1) l0 = 0;
    else if (s->p.pair_levels == 3) l0 = 1;
    else {
        if (nb >= SEP_PAIR_MIN_FRAMES && s->p.in_dtype == MI_F32 && s->L >= 2) { pm[0] = 1; pm[1] = 2; }
        return;
    }
    for (int l = l0; l + 2 <= s->L; l += 2) { pm[l] = 1; pm[l + 1] = 2; }
}

// MI_ARITH_EXACT: the same for the reference-order arithmetic (exact_payload, kernels_tiled.hpp)
template <typename TIn, bool FMA>
int launch_payload_exact(mi_stack* s, int l, int set, const void* src, size_t src_stride, int nb, hipStream_t st) {
    TiledState* t = tstate(s);
    const dim3 blk(32, 8);
    const dim3 grd(cdiv(cdiv(s->lw[l], 2), blk.x), cdiv(cdiv(s->lh[l], 2), blk.y));
    K6 K{};
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 3; ++j) K.c[i == 0 ? j : (i == 1 ? 2 + j : 5)] = s->K.k[i * 5 + j];
    ProfScope ps(s, MI_PROF_LEVEL, 0.0, st);
    hipLaunchKernelGGL((exact_payload<TIn, FMA>), grd, blk, 0, st, src, src_stride, (const float*)t->Gb[set][l + 1],
                       t->gstride[l + 1], nb, s->lh[l], s->lw[l], s->lh[l + 1], s->lw[l + 1], (const int32_t*)s->bestIdx[l],
                       s->first_index + s->n_pushed, s->bestLap[l], K);
    return MI_OK;
}

// level 0 of the separable arithmetic: the matrix-pipe form of the reduce for 8 / 16-bit frames when the taps allow it
template <typename TIn>
int launch_level0_sep(mi_stack* s, int set, const void* src, size_t src_stride, int nb, hipStream_t st_in, hipStream_t st_bd,
                      hipEvent_t ev_bd, int f_begin = 0, int f_end = -1, SepLevelInfo* info = nullptr) {
    if constexpr (MI_SEP_MFMA && sizeof(TIn) == 1) {   // (16-bit frames: two byte planes, ten matrix instructions per task -- measured slower)
        static const int mf_off = study_env("MI_NO_MFMA", 0);   // -DMI_STUDY: the VALU form, for A/B runs
        if (s->mfma_ok && !mf_off)
            return launch_level_sep<TIn, true, true>(s, 0, set, src, src_stride, nb, st_in, st_bd, ev_bd, f_begin, f_end, info);
    }
    return launch_level_sep<TIn, true, false>(s, 0, set, src, src_stride, nb, st_in, st_bd, ev_bd, f_begin, f_end, info);
}

template <typename TIn, bool FMA>
int run_batch(mi_stack* s, const void* frames, size_t stride, int nb) {
    TiledState* t = tstate(s);
    const int L = s->L;
    const int set = (int)(t->batch_no & 1);
    hipStream_t st0 = s->stream, st1 = t->st1, st2 = t->st2, st3 = t->st3;
    int rc;
    if ((rc = tiled_reserve(s, set, nb))) return rc;
    // the previous batch's payload passes (st3) fold chunk partials into the running state and read what this batch's level
    // kernels are about to overwrite
    for (hipStream_t st : {st0, st1, st2}) MI_HIP(hipStreamWaitEvent(st, t->evPay[set ^ 1], 0));
    // level `l`'s energy pass is through on st2 (its border tiles joined): its payload pass may start on st3
    auto payload_after = [&](int l) -> int {
        hipEvent_t e = t->evPayIn[set * (L + 1) + l];
        MI_HIP(hipEventRecord(e, st2));
        MI_HIP(hipStreamWaitEvent(st3, e, 0));
        return MI_OK;
    };
    // Gb[set] is free once st2 finished batch k-2 (no-op for the first two batches)
    MI_HIP(hipStreamWaitEvent(st0, t->evRest[set], 0));
    MI_HIP(hipStreamWaitEvent(st1, t->evRest[set], 0));
    if (s->sep && !t->partE.empty() && t->part_cap[0] > 0) {
        // level 0 in frame chunks (small frames): the previous batch's payload pass (st2) folds partE[0] / partI[0], which
        // this batch's level-0 kernels overwrite
        MI_HIP(hipStreamWaitEvent(st0, t->evL0done[set ^ 1], 0));
        MI_HIP(hipStreamWaitEvent(st1, t->evL0done[set ^ 1], 0));
    }
    // whatever produced the frames on s->stream (a warp, a table apply, a copy enqueued by the caller) is ordered
    // before the level-0 interior kernel by the stream itself; the border kernel runs on st1 and needs the event
    MI_HIP(hipEventRecord(t->evInput, st0));
    MI_HIP(hipStreamWaitEvent(st1, t->evInput, 0));
    // -DMI_STUDY, MI_INTERLEAVE01=1: level 1 of frame group g right behind level 0 of the same group, on the same stream (the
    // G_1 images level 0 just wrote are the freshest lines of the 256 MB Infinity Cache when level 1 reads them), instead of
    // all of level 0, then all of level 1.  Only when both levels run as consecutive launches without border tiles.
    static const int interleave01 = study_env("MI_INTERLEAVE01", 0);
    bool il = false;
    SepLevelInfo li0;
    std::vector<int> pm;   // the batch's pair plan (sep_pair_plan): 1 = first level of a pair, 2 = second
    sep_pair_plan(s, nb, pm);
    const bool pair = pm[0] == 1;
    // pair: pipe level 1's energy pass behind level 0's launches, group by group -- when both levels run as consecutive
    // launches and have no border tiles (even sizes)
    bool l1e_pipe = false;
    if (pair) {
        static const int pipe_on = study_env("MI_L1E_PIPE", MI_L1E_PIPE_DEFAULT);
        bool p0 = false, p1 = false;
        level_chunk_frames(nb, cdiv(s->lw[0], 56) * cdiv(s->lh[0], MI_SEP_TH), &p0);
        level_chunk_frames(nb, cdiv(s->lw[1], 56) * cdiv(s->lh[1], MI_SEP_TH), &p1);
        l1e_pipe = pipe_on && !p0 && !p1 && !(s->lh[0] & 1) && !(s->lw[0] & 1) && !(s->lh[1] & 1) && !(s->lw[1] & 1);
    }
    if (s->sep && !pair && interleave01 && L >= 2 && nb > SEP_LAUNCH_FRAMES) {
        bool p0 = false, p1 = false;
        const int nt0 = cdiv(s->lw[0], 56) * cdiv(s->lh[0], MI_SEP_TH), nt1 = cdiv(s->lw[1], 56) * cdiv(s->lh[1], MI_SEP_TH);
        level_chunk_frames(nb, nt0, &p0);
        level_chunk_frames(nb, nt1, &p1);
        il = !p0 && !p1 && !(s->lh[0] & 1) && !(s->lw[0] & 1) && !(s->lh[1] & 1) && !(s->lw[1] & 1);
    }
    if (il) {
        for (int f0 = 0; f0 < nb && !rc; f0 += SEP_LAUNCH_FRAMES) {
            const int f1 = std::min(nb, f0 + SEP_LAUNCH_FRAMES);
            rc = launch_level0_sep<TIn>(s, set, frames, stride, nb, st0, st1, t->evL0b[set], f0, f1);
            if (!rc) rc = launch_level_sep<float, false>(s, 1, set, t->Gb[set][1], t->gstride[1] * sizeof(float), nb, st0, st1,
                                                         t->evLvl[(set * (L + 1) + 1) * 2 + 1], f0, f1);
        }
    } else if (pair && l1e_pipe) {
        // level 1's energy launch of a frame group right behind level 0's launch of that group, on st2 (it needs the group's
        // gray(G_1) and G_2 only): the light kernel beside the heavy one
        const int ngrp = cdiv(nb, SEP_LAUNCH_FRAMES);
        while ((int)t->evGrp.size() < ngrp) {
            hipEvent_t e;
            MI_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            t->evGrp.push_back(e);
        }
        for (int g = 0; g < ngrp && !rc; ++g) {
            const int f0 = g * SEP_LAUNCH_FRAMES, f1 = std::min(nb, f0 + SEP_LAUNCH_FRAMES);
            rc = launch_level_sep<TIn, true, false, 1>(s, 0, set, frames, stride, nb, st0, st1, t->evL0b[set], f0, f1, &li0);
            if (rc) break;
            MI_HIP(hipEventRecord(t->evGrp[g], st0));
            MI_HIP(hipStreamWaitEvent(st2, t->evGrp[g], 0));
            SepLevelInfo li1;
            rc = launch_level_sep<float, false, false, 2>(s, 1, set, t->Gb[set][1], (size_t)s->lh[1] * s->lw[1] * sizeof(float), nb,
                                                          st2, st1, t->evLvl[(set * (L + 1) + 1) * 2 + 1], f0, f1, &li1);
        }
    } else if (pair) rc = launch_level_sep<TIn, true, false, 1>(s, 0, set, frames, stride, nb, st0, st1, t->evL0b[set], 0, -1, &li0);
    else if (s->sep) rc = launch_level0_sep<TIn>(s, set, frames, stride, nb, st0, st1, t->evL0b[set], 0, -1, &li0);
    else
        rc = launch_level<TIn, FMA, MI_TILE0_H, MI_TILE0_W, MI_TILE0_NT, MI_TILE_PAD != 0, MI_TILE_H, MI_TILE_W, MI_TILE_NT>(
            s, 0, set, frames, stride, nb, st0, st1, t->evL0b[set]);
    if (rc) return rc;
    MI_HIP(hipEventRecord(t->evL0i[set], st0));
    MI_HIP(hipEventRecord(t->evL0b[set], st1));
    MI_HIP(hipStreamWaitEvent(st2, t->evL0i[set], 0));
    MI_HIP(hipStreamWaitEvent(st2, t->evL0b[set], 0));
    MI_HIP(hipStreamWaitEvent(st1, t->evL0i[set], 0));
    // the first level of a pair gets its payload together with the second level's, behind that level's energy pass (below);
    // level 0's state is final only then
    MI_HIP(hipStreamWaitEvent(st3, t->evL0i[set], 0));
    MI_HIP(hipStreamWaitEvent(st3, t->evL0b[set], 0));
    if (!pair) {
        if ((rc = s->sep ? launch_payload_sep<TIn>(s, 0, set, frames, stride, nb, st3, li0.nparts)
                         : launch_payload_exact<TIn, FMA>(s, 0, set, frames, stride, nb, st3)))
            return rc;
        MI_HIP(hipEventRecord(t->evL0done[set], st3));   // st3 waited for both level-0 kernels above
    }
    int nparts_first = li0.nparts;   // chunk partials of the pending pair's first level
    // The payload of the pair (l - 1, l), on st3 behind level l's energy pass: tile by tile when both levels ran unchunked,
    // else the per-quad kernels (which fold the chunks' partial maxima).  `src` = the images of level l - 1.
    auto pair_payload = [&](int l, int nparts_second, auto tag, const void* src, size_t src_stride) -> int {
        using T = decltype(tag);
        int r;
        if (nparts_first == 0 && nparts_second == 0 && nb <= 256) r = launch_payload_pair_tiles<T>(s, l - 1, set, src, src_stride, nb, st3);
        else {
            r = launch_payload_pair0<T>(s, l - 1, set, src, src_stride, nb, st3, nparts_first);
            if (!r) r = launch_payload_pair1<T>(s, l - 1, set, src, src_stride, nb, st3, nparts_second);
        }
        if (!r && l == 1) MI_HIP(hipEventRecord(t->evL0done[set], st3));
        return r;
    };
    // coarser levels: interior tiles on st2, border tiles on st1 (disjoint tiles of one level run
    // side by side); both streams join after every level because level l+1 reads all of G_{l+1}
    static const int only_l0 = study_env("MI_ONLY_L0", 0);   // -DMI_STUDY: level 0 alone on the GPU (results are wrong)
    for (int l = 1; l < L && !only_l0; ++l) {
        // coarser levels that still have thousands of tiles (4 MP and more: level 1 of a 24 MP frame) run on
        // level 0's tile configuration -- less halo per tile: +2 % on the 256 x 24 MP job; MI_WIDE_LEVELS overrides
        static const int wide_levels = study_env("MI_WIDE_LEVELS", -1);
        const bool wide = wide_levels >= 0 ? l <= wide_levels : (size_t)s->lh[l] * s->lw[l] >= ((size_t)4 << 20);
        hipEvent_t ei = t->evLvl[(set * (L + 1) + l) * 2], eb = t->evLvl[(set * (L + 1) + l) * 2 + 1];
        if (pm[l] == 2) {
            // the second level of a pair: energy only, from gray(G_l) and G_{l+1} (both written by level l - 1's kernel);
            // the pair's payload pass recomputes the winners' G_l from level l - 1's images
            SepLevelInfo li;
            if (l == 1 && l1e_pipe) ;   // (launched group by group behind level 0, above: no chunks, no border tiles)
            else if ((rc = launch_level_sep<float, false, false, 2>(s, l, set, t->Gb[set][l], (size_t)s->lh[l] * s->lw[l] * sizeof(float), nb,
                                                                    st2, st1, eb, 0, -1, &li, ei)))
                return rc;
            if (li.border) {
                MI_HIP(hipEventRecord(eb, st1));
                MI_HIP(hipStreamWaitEvent(st2, eb, 0));
            }
            if ((rc = payload_after(l))) return rc;
            if ((rc = l == 1 ? pair_payload(l, li.nparts, TIn{}, frames, stride)
                             : pair_payload(l, li.nparts, float{}, t->Gb[set][l - 1], t->gstride[l - 1] * sizeof(float))))
                return rc;
            continue;
        }
        if (pm[l] == 1) {
            // the first level of a pair beyond level 0: level_sep_pair on the float images of level l (gray(G_{l+1}) and G_{l+2}
            // out); its payload waits for level l + 1
            SepLevelInfo li;
            if ((rc = launch_level_sep<float, false, false, 1>(s, l, set, t->Gb[set][l], t->gstride[l] * sizeof(float), nb, st2, st1, eb,
                                                               0, -1, &li, ei)))
                return rc;
            if (li.border) {
                MI_HIP(hipEventRecord(eb, st1));
                MI_HIP(hipStreamWaitEvent(st2, eb, 0));
            }
            nparts_first = li.nparts;
            continue;
        }
        if (s->sep && !(il && l == 1)) {
            // interior tiles on st2; border tiles, if the level has any, on st1 behind everything st2 has done so far (`ei`);
            // the streams join again (`eb`) in front of the payload pass, which also folds the frame chunks' partial maxima
            SepLevelInfo li;
            if ((rc = launch_level_sep<float, false>(s, l, set, t->Gb[set][l], t->gstride[l] * sizeof(float), nb, st2, st1, eb, 0, -1,
                                                     &li, ei)))
                return rc;
            if (li.border) {
                MI_HIP(hipEventRecord(eb, st1));
                MI_HIP(hipStreamWaitEvent(st2, eb, 0));
            }
            if ((rc = payload_after(l))) return rc;
            if ((rc = launch_payload_sep<float>(s, l, set, t->Gb[set][l], t->gstride[l] * sizeof(float), nb, st3, li.nparts))) return rc;
            continue;
        }
        if (s->sep && il && l == 1)
            rc = MI_OK;   // level 1 ran interleaved with level 0 on st0 (st2 waited for evL0i, recorded behind both)
        else if (wide)
            rc = launch_level<float, FMA, MI_TILE0_H, MI_TILE0_W, MI_TILE0_NT, MI_TILE_PAD != 0, MI_TILE_H, MI_TILE_W, MI_TILE_NT, true>(
                s, l, set, t->Gb[set][l], t->gstride[l] * sizeof(float), nb, st2, st1, t->evLvl[(set * (L + 1) + l) * 2 + 1]);
        else
            rc = launch_level<float, FMA, MI_TILE_H, MI_TILE_W, MI_TILE_NT, false, MI_TILE_H, MI_TILE_W, MI_TILE_NT>(
                s, l, set, t->Gb[set][l], t->gstride[l] * sizeof(float), nb, st2, st1, t->evLvl[(set * (L + 1) + l) * 2 + 1]);
        if (rc)
            return rc;
        MI_HIP(hipEventRecord(ei, st2));
        MI_HIP(hipEventRecord(eb, st1));
        MI_HIP(hipStreamWaitEvent(st2, eb, 0));
        MI_HIP(hipStreamWaitEvent(st1, ei, 0));
        if ((rc = payload_after(l))) return rc;
        if ((rc = s->sep ? launch_payload_sep<float>(s, l, set, t->Gb[set][l], t->gstride[l] * sizeof(float), nb, st3)
                         : launch_payload_exact<float, FMA>(s, l, set, t->Gb[set][l], t->gstride[l] * sizeof(float), nb, st3)))
            return rc;
    }
    MI_HIP(hipGetLastError());
    if (!only_l0) {   // base level of the whole batch
        ProfScope ps(s, MI_PROF_BASE, 0.0, st2);
        const int hb = s->lh[L], wb = s->lw[L], npix = hb * wb;
        MI_HIP(hipMemsetAsync(t->cnt[set], 0, sizeof(uint32_t) * s->nlevels_hist * nb, st2));
        hipLaunchKernelGGL((base_gray_hist_batch<FMA>), dim3(cdiv(npix, 256), nb), dim3(256), 0, st2,
                           t->Gb[set][L], t->gstride[L], npix, s->nlevels_hist, t->lev[set], t->cnt[set]);
        hipLaunchKernelGGL(base_logp_batch, dim3(cdiv(s->nlevels_hist, 256), nb), dim3(256), 0, st2,
                           t->cnt[set], s->nlevels_hist, npix, t->logp[set]);
        const dim3 blk(16, 4);
        dim3 grd = grid2d(wb, hb, blk);
        grd.z = nb;
        if (s->pad == 2)
            hipLaunchKernelGGL(base_feat_batch_c<2>, grd, blk, 0, st2, t->lev[set], t->logp[set], s->nlevels_hist, hb, wb, t->feat[set]);
        else
            hipLaunchKernelGGL(base_feat_batch, grd, blk, 0, st2, t->lev[set], t->logp[set], s->nlevels_hist, hb, wb, s->pad,
                               t->feat[set]);
        if (nb >= 32)   // many frames: a pixel's scan over eight lanes
            hipLaunchKernelGGL(base_select_batch_seg<8>, dim3(cdiv(npix * 8, 256)), dim3(256), 0, st2, t->feat[set], t->Gb[set][L],
                               t->gstride[L], nb, npix, s->first_index + s->n_pushed, s->n_pushed == 0, s->bEnt, s->bDev,
                               s->idxE, s->idxD, s->baseE, s->baseD);
        else
            hipLaunchKernelGGL(base_select_batch, dim3(cdiv(npix, 64)), dim3(64), 0, st2, t->feat[set], t->Gb[set][L],
                               t->gstride[L], nb, npix, s->first_index + s->n_pushed, s->n_pushed == 0, s->bEnt, s->bDev,
                               s->idxE, s->idxD, s->baseE, s->baseD);
        MI_HIP(hipGetLastError());
    }
    MI_HIP(hipEventRecord(t->evPay[set], st3));
    MI_HIP(hipStreamWaitEvent(st2, t->evPay[set], 0));   // evRest covers the payload passes too (they read Gb[set])
    MI_HIP(hipEventRecord(t->evRest[set], st2));
    t->streams_dirty = true;
    s->n_pushed += nb;
    t->last_nb = nb;
    t->last_set = set;
    t->last_kept = 0;
    for (int l = 0; l + 1 < L; ++l)
        if (pm[l] == 1) t->last_kept |= 1u << (l + 1);
    t->batch_no++;
    return MI_OK;
}

// host waits until the level-0 selection state of every pushed frame is final (the coarser levels of the last batch
// may still be running): the cross-GPU exchange of level 0 -- 3/4 of the state -- can start here
int tiled_sync_level0(mi_stack* s) {
    TiledState* t = tstate(s);
    if (!t || s->p.impl != MI_IMPL_TILED || t->batch_no == 0) return tiled_sync_all(s);
    // batches run in order on every stream, so the last batch's event covers the earlier ones
    MI_HIP(hipEventSynchronize(t->evL0done[t->last_set]));
    return MI_OK;
}

// make s->stream wait for everything enqueued on the side streams
int tiled_join(mi_stack* s) {
    TiledState* t = tstate(s);
    if (!t || !t->streams_dirty) return MI_OK;
    for (int set = 0; set < 2; ++set) {
        MI_HIP(hipStreamWaitEvent(s->stream, t->evL0b[set], 0));
        MI_HIP(hipStreamWaitEvent(s->stream, t->evRest[set], 0));
    }
    return MI_OK;
}

int tiled_push(mi_stack* s, const void* dev_frames, int n, size_t stride) {
    TiledState* t = tstate(s);
    if (s->L == 0) return fail(MI_ERR_UNSUPPORTED, "frames smaller than 2*min_size have no pyramid levels");
    int rc = tiled_flush(s);  // keep global frame order: staged host frames come first
    if (rc) return rc;
    const bool fma = s->p.use_fma != 0;
    // MI_ARITH_SEPARABLE: the whole push is one batch when its per-batch buffers fit (level after level over all the
    // frames: every kernel has the GPU to itself, a pixel's winning Laplacian is filled in once, and the small levels run
    // in frame chunks); longer pushes are cut into equal batches.  MI_ARITH_EXACT runs on the same schedule since round 3.
    int sep_nb = 0;
    if (n > 0) {
        if (t->dev_cap == 0) {
            if (s->p.batch_frames > 0) t->dev_cap = s->p.batch_frames;
            else {
                size_t free_b = 0, total_b = 0;
                MI_HIP(hipMemGetInfo(&free_b, &total_b));
                // a quarter of what is free now per set of buffers (two sets when a push needs more than one batch)
                const size_t fit = free_b / 4 / std::max<size_t>(tiled_bytes_per_frame(s), 1);
                t->dev_cap = (int)std::min<size_t>(256, std::max<size_t>(t->bcap, fit & ~(size_t)7));
            }
        }
        const int nbat = cdiv(n, t->dev_cap);
        sep_nb = std::min(t->dev_cap, cdiv(cdiv(n, nbat), 4) * 4);
    }
    for (int f0 = 0; f0 < n;) {
        // Full batches while more than one batch is left; the last `bcap` frames are tapered
        // (1/2, 1/4, 1/4): what cannot overlap anything is the final batch's chain of coarser
        // levels, which is latency-bound and scales with the batch length.
        const int left = n - f0;
        int nb = left > t->bcap ? t->bcap : left;
        static const int taper = study_env("MI_TAPER", 1);   // 0 / 2: timing studies (-DMI_STUDY)
        if (taper == 1 && left <= t->bcap && left >= 16 && n > t->bcap) nb = (left / 2 + 3) & ~3;
        if (taper == 2 && left == t->bcap && n > t->bcap) nb = t->bcap / 2;
        if (sep_nb) nb = std::min(left, sep_nb);
        const void* fr = (const char*)dev_frames + (size_t)f0 * stride;
        switch (s->p.in_dtype) {
            case MI_U8: rc = fma ? run_batch<uint8_t, true>(s, fr, stride, nb) : run_batch<uint8_t, false>(s, fr, stride, nb); break;
            case MI_U16: rc = fma ? run_batch<uint16_t, true>(s, fr, stride, nb) : run_batch<uint16_t, false>(s, fr, stride, nb); break;
            case MI_F32: rc = fma ? run_batch<float, true>(s, fr, stride, nb) : run_batch<float, false>(s, fr, stride, nb); break;
            default: rc = fail(MI_ERR_INVALID, "bad in_dtype");
        }
        if (rc) return rc;
        f0 += nb;
    }
    // later work on s->stream (host-frame copies, collapse, taps) is ordered behind all of it
    return tiled_join(s);
}

int tiled_flush(mi_stack* s) {
    TiledState* t = tstate(s);
    if (!t || t->pending == 0) return MI_OK;
    int n = t->pending;
    t->pending = 0;
    if (t->stc) {
        // the kernels must see every staged frame: order the compute streams behind the copies
        MI_HIP(hipEventRecord(t->evCopied, t->stc));
        MI_HIP(hipStreamWaitEvent(s->stream, t->evCopied, 0));
        if (t->st1) MI_HIP(hipStreamWaitEvent(t->st1, t->evCopied, 0));
    }
    int rc = dispatch_push(s, t->ring, n, t->frame_bytes);
    if (rc) return rc;
    if (t->stc) {
        // ... and the next batch's copies must not overwrite the ring before level 0 has read it
        // (tiled_push joined st1 into s->stream, so one event on s->stream covers both kernels)
        MI_HIP(hipEventRecord(t->evRingFree, s->stream));
        MI_HIP(hipStreamWaitEvent(t->stc, t->evRingFree, 0));
    }
    return MI_OK;
}

// Persistent helper threads for the bounce copy of big frames (round 3 spawned three std::threads per frame): workers
// sleep on a condition variable and take the row bands of ONE copy at a time.
class CopyPool {
public:
    static CopyPool& get() {
        static CopyPool p;
        return p;
    }
    // runs fn(k) for k = 0 .. parts-1: k = 0 on the caller, the others on the workers; returns when all are done
    void run(int parts, const std::function<void(int)>& fn) {
        std::lock_guard<std::mutex> serial(call_mu_);   // one copy at a time (several stacks may push from several threads)
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn;
            next_ = 1;
            parts_ = parts;
            pending_ = parts - 1;
        }
        cv_.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    CopyPool() {
        for (int i = 0; i < NW; ++i) th_[i] = std::thread([this] { loop(); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    void loop() {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_.wait(lk, [&] { return stop_ || (fn_ && next_ < parts_); });
            if (stop_) return;
            const int k = next_++;
            const auto* fn = fn_;
            lk.unlock();
            (*fn)(k);
            lk.lock();
            if (--pending_ == 0) done_.notify_all();
        }
    }
    static constexpr int NW = 3;
    std::thread th_[NW];
    std::mutex mu_, call_mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* fn_ = nullptr;
    int next_ = 0, parts_ = 0, pending_ = 0;
    bool stop_ = false;
};

// One host frame: copy it into a pinned bounce buffer, start the asynchronous upload into the
// device ring and return (the caller's buffer is free again); a full ring triggers a fused batch.
// `pinned`: the caller's buffer IS pinned memory (mi_host_alloc / mi_host_register) with contiguous rows: the upload
// reads it directly -- no bounce copy -- and the caller must leave it alone until mi_stack_wait_uploads says so.
int tiled_push_host(mi_stack* s, const void* host_bgr, size_t row_stride_bytes, bool pinned) {
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
    if (s->p.impl != MI_IMPL_TILED) {
        // simple implementation: synchronous staging of the single frame
        MI_HIP(hipMemcpy2DAsync(dst, rb, host_bgr, row_stride_bytes, rb, s->p.height,
                                hipMemcpyHostToDevice, s->stream));
        MI_HIP(hipStreamSynchronize(s->stream));
    } else {
        if (!t->stc) {
            // The copy stream gets a priority class of its own -- the LOW one: the main stream is normal, the side streams
            // high, and no kernel ever runs at low priority.  HIP maps a process's streams onto a handful of hardware queues
            // per priority class, and a copy that shares a queue with ANOTHER handle's kernels waits behind them: two handles
            // that alternate (pipeline.bunches_then_stack) then upload and compute one after the other instead of side by
            // side (measured: 60 ms per 10-frame bunch = 52 upload + 8 compute).  MI_COPY_PRIO: 0 plain, 1 high, 2 low.
#ifndef MI_COPY_PRIO
#define MI_COPY_PRIO 2
#endif
            if (MI_COPY_PRIO) {
                int prio_lo = 0, prio_hi = 0;
                MI_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
                MI_HIP(hipStreamCreateWithPriority(&t->stc, hipStreamNonBlocking, MI_COPY_PRIO == 1 ? prio_hi : prio_lo));
            } else MI_HIP(hipStreamCreateWithFlags(&t->stc, hipStreamNonBlocking));
            MI_HIP(hipEventCreateWithFlags(&t->evCopied, hipEventDisableTiming));
            MI_HIP(hipEventCreateWithFlags(&t->evRingFree, hipEventDisableTiming));
        }
        if (!pinned && !t->pin[0])
            for (int i = 0; i < TiledState::NPIN; ++i) {
                MI_HIP(hipHostMalloc(&t->pin[i], t->frame_bytes, hipHostMallocDefault));
                MI_HIP(hipEventCreateWithFlags(&t->evPin[i], hipEventDisableTiming));
            }
        if (pinned) {
            if (row_stride_bytes != rb) return fail(MI_ERR_INVALID, "a pinned frame must have contiguous rows");
            hipEvent_t& ev = t->evUp[t->up_no % TiledState::NUP];
            if (!ev) MI_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            else MI_HIP(hipEventSynchronize(ev));   // (only when more than NUP uploads are in flight)
            MI_HIP(hipMemcpyAsync(dst, host_bgr, t->frame_bytes, hipMemcpyHostToDevice, t->stc));
            MI_HIP(hipEventRecord(ev, t->stc));
            t->up_no++;
        } else {
            const int slot = (int)(t->pin_no % TiledState::NPIN);
            MI_HIP(hipEventSynchronize(t->evPin[slot]));  // bounce buffer free again? (no-op when unused)
            char* pb = (char*)t->pin[slot];
            // One thread copies ~25 GB/s into pinned memory -- half of what the PCIe link then moves; big
            // frames are copied by four threads (the caller + the pool's three), in row bands.
            const int H = s->p.height;
            const size_t total = rb * (size_t)H;
            const int nthr = total >= ((size_t)32 << 20) ? 4 : 1;
            const std::function<void(int)> band = [&](int k) {
                const int y0 = (int)((long)H * k / nthr), y1 = (int)((long)H * (k + 1) / nthr);
                if (row_stride_bytes == rb) memcpy(pb + (size_t)y0 * rb, (const char*)host_bgr + (size_t)y0 * rb, rb * (size_t)(y1 - y0));
                else
                    for (int y = y0; y < y1; ++y)
                        memcpy(pb + (size_t)y * rb, (const char*)host_bgr + (size_t)y * row_stride_bytes, rb);
            };
            if (nthr == 1) band(0);
            else CopyPool::get().run(nthr, band);
            MI_HIP(hipMemcpyAsync(dst, pb, t->frame_bytes, hipMemcpyHostToDevice, t->stc));
            MI_HIP(hipEventRecord(t->evPin[slot], t->stc));
            t->pin_no++;
        }
    }
    t->pending++;
    if (t->pending == t->bcap) return tiled_flush(s);
    return MI_OK;
}

// host blocks until at most `max_outstanding` of the zero-copy uploads pushed so far are still in flight
int tiled_wait_uploads(mi_stack* s, int max_outstanding) {
    TiledState* t = tstate(s);
    if (!t || t->up_no == 0) return MI_OK;
    if (max_outstanding < 0) max_outstanding = 0;
    const long upto = t->up_no - max_outstanding;          // uploads 0 .. upto-1 must be complete
    if (upto <= 0) return MI_OK;
    if (max_outstanding >= TiledState::NUP) return MI_OK;   // (older ones were waited for when their event slot was reused)
    hipEvent_t ev = t->evUp[(upto - 1) % TiledState::NUP];
    if (ev) MI_HIP(hipEventSynchronize(ev));
    return MI_OK;
}

}  // namespace mi
