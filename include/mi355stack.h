/*
 * mi355stack.h -- C ABI of libmi355stack.so: the MI355X (gfx950) focus-stacking
 * hot path behind shinestacker's PyramidStack / align_images.
 *
 * Plain pointers and sizes only; no exceptions cross this boundary: every entry
 * point returns an MI_* status and mi_last_error() gives the thread-local text
 * (the Python shim maps codes to shinestacker's exception types,
 * reference core/exceptions.py:2-52).
 *
 * Reference interfaces replaced (paths under /root/reference/src/shinestacker):
 *   mi_stack_create / _reset / _destroy   PyramidStack.__init__ + per-stack state,
 *                                         algorithms/pyramid.py:114-123, :150-165
 *   mi_stack_push_frame[_device]          process_single_image per frame,
 *                                         pyramid.py:125-139 (called at :173), fused with
 *                                         the per-level selection of fuse_laplacian :48-55
 *                                         and the base features of get_fused_base :95-102
 *                                         as a running first-max (frames in index order)
 *   mi_stack_finish[_device]              fuse_pyramids tail + collapse + cast,
 *                                         pyramid.py:103-111, :57-64, :178-179
 *   mi_stack_get_level                    debug/parity taps (no reference counterpart)
 *   mi_stack_state / _set_first_index     hooks for the frame-sharded multi-GPU combine
 *   mi_warp_affine[_device]               cv2.warpAffine + warped mask + border blur composite,
 *                                         algorithms/align.py:238-251
 *
 * Ownership: the caller owns every host buffer it passes and every device
 * buffer obtained from mi_device_malloc; the library owns all device memory
 * inside a handle; nothing returned by pointer outlives mi_stack_destroy.
 * Threading: one thread at a time per handle; any thread may call (the device
 * is selected per call).  The library never calls back into the host language.
 */
#ifndef MI355STACK_H
#define MI355STACK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_ABI_VERSION 1
#define MI_API __attribute__((visibility("default")))

/* status codes */
enum {
    MI_OK = 0,
    MI_ERR_INVALID = 1,   /* bad argument / option            -> InvalidOptionError / ValueError */
    MI_ERR_NO_DEVICE = 2, /* no HIP device visible            -> RuntimeError                    */
    MI_ERR_HIP = 3,       /* a HIP runtime call failed        -> RuntimeError                    */
    MI_ERR_STATE = 4,     /* call order (finish before push)  -> RuntimeError                    */
    MI_ERR_NOMEM = 5,     /* device allocation failed         -> MemoryError                     */
    MI_ERR_UNSUPPORTED = 6,
    MI_ERR_ALIGNMENT = 7  /* a frame could not be registered    -> AlignmentError                  */
};

/* pixel / arithmetic types */
enum { MI_U8 = 0, MI_U16 = 1, MI_F32 = 2, MI_F64 = 3 };

/* mi_stack_get_level selectors */
enum {
    MI_TAP_GAUSS = 0,      /* G_l of the most recently pushed frame, h x w x 3 f32 (l >= 1)   */
    MI_TAP_FUSED_LAP = 1,  /* running fused Laplacian of level l, h x w x 3 f32               */
    MI_TAP_ENERGY = 2,     /* running max energy of level l, h x w f32                        */
    MI_TAP_INDEX = 3,      /* running arg-max (global frame index), h x w i32                 */
    MI_TAP_FUSED_BASE = 4, /* fused base (level == levels), hb x wb x 3 f32; valid after finish */
    MI_TAP_BASE_IDX_E = 5, /* hb x wb i32 */
    MI_TAP_BASE_IDX_D = 6, /* hb x wb i32 */
    MI_TAP_COLLAPSED = 7,  /* clip(abs(collapse)), H x W x 3 f32; valid after finish            */
    MI_TAP_BASE_ENT = 8,   /* running max entropy, hb x wb f32 */
    MI_TAP_BASE_DEV = 9    /* running max deviation, hb x wb f32 */
};

/* implementation selector (both give bit-identical results) */
enum {
    MI_IMPL_AUTO = 0,
    MI_IMPL_SIMPLE = 1, /* one thread per output, global-memory taps: the on-GPU cross-check */
    MI_IMPL_TILED = 2   /* LDS-tiled fused level kernels                                     */
};

/* arithmetic of the pyramid stencils (mi_stack_params.arith)
 *   MI_ARITH_EXACT      every 5x5 stencil as the reference's own 25-tap row-major chain (cv2.filter2D order as
 *                       restated in oracle/): every intermediate bit-identical to the oracle.  The value of a
 *                       zero-initialised struct and of mi_stack_default_params: at the C boundary the caller NAMES the
 *                       arithmetic, and what it gets without naming one is the bit-pinned order (the audit mode).  The
 *                       Python host mirror's high-level entry points (PyramidStack, align_and_stack[_device],
 *                       bunches_then_stack) all pass ONE default explicitly -- defaults.resolve_arith, "separable" -- so no
 *                       two of them differ for the same input (round 5).
 *   MI_ARITH_SEPARABLE  the same stencils as two 5-tap passes with the float32 generating kernel (polyphase for
 *                       the expand): ~half the arithmetic, coefficients within the float32 forward-error bound
 *                       of a float64 evaluation, per-pixel arg-max may flip at near ties (tolerance in
 *                       tests/test_sep_tolerance.py).  float_type must be MI_F32. */
enum { MI_ARITH_EXACT = 0, MI_ARITH_SEPARABLE = 1 };

typedef struct mi_stack mi_stack_t;

typedef struct mi_stack_params {
    int32_t height, width; /* frame geometry; channels fixed at 3, BGR interleaved            */
    int32_t in_dtype;      /* MI_U8 / MI_U16 / MI_F32 (f32 frames hold integer pixel values)  */
    int32_t out_dtype;     /* MI_U8 / MI_U16: dtype of the fused image (pyramid.py:159-164)   */
    int32_t min_size;      /* pyramid.py:115,165  default 32                                  */
    int32_t kernel_size;   /* pyramid.py:116,18   base-level window only; default 5           */
    double gen_kernel;     /* pyramid.py:117,19   default 0.4                                 */
    int32_t float_type;    /* MI_F32 / MI_F64 (base_stack_algo.py:14-22; MI_F64 runs one frame at a time) */
    int32_t use_fma;       /* 1: fma chain (OpenCV AVX2 path) 0: mul+add (SSE baseline path)  */
    int32_t device;        /* HIP device ordinal                                              */
    int32_t impl;          /* MI_IMPL_*                                                       */
    int32_t batch_frames;  /* frames per batch (tiled impl).  0 = automatic: host frames are staged in a ring of 32;
                            * MI_ARITH_SEPARABLE takes a resident push as ONE batch of up to 256 frames while its
                            * per-batch buffers fit a quarter of the free device memory (allocated on demand).
                            * > 0: batches of exactly this many frames, buffers allocated by mi_stack_create (for
                            * callers that push batch after batch and must not stall on an allocation)           */
    int32_t arith;         /* MI_ARITH_*; 0 = exact (the C default, see the enum's comment)     */
    int32_t pair_levels;   /* MI_ARITH_SEPARABLE: pyramid levels as pass PAIRS (l, l + 1) -- level l's kernel hands level l + 1
                            * gray(G_{l+1}) and G_{l+2}, the three-channel G_{l+1} of the batch never reaches HBM (pyramid.py:27-46,
                            * :125-139: the reduce -> expand dependency), the winners' pixels of it are recomputed once per batch.
                            * Results are bit-identical either way.  0 = automatic = where it measured faster: the pair (0, 1)
                            * for float-32 frames in batches of 192 and more; 1 = pairs (0, 1), (2, 3), ...; 2 = none;
                            * 3 = pairs (1, 2), (3, 4), ...                                                */
    int32_t reserved[3];
} mi_stack_params_t;

/* ---- library ---- */
MI_API int mi_abi_version(void);
MI_API const char* mi_last_error(void);
MI_API int mi_device_count(int* count);
MI_API int mi_device_name(int device, char* buf, size_t buflen);
MI_API void mi_stack_default_params(mi_stack_params_t* p);

/* ---- device memory helpers (so a host language needs no other GPU binding) ---- */
MI_API int mi_device_malloc(int device, size_t bytes, void** dev_ptr);
MI_API int mi_device_free(int device, void* dev_ptr);
MI_API int mi_memcpy_h2d(int device, void* dev_dst, const void* host_src, size_t bytes);
MI_API int mi_memcpy_d2h(int device, void* host_dst, const void* dev_src, size_t bytes);
MI_API int mi_memcpy_d2d(int device, void* dev_dst, const void* dev_src, size_t bytes);
MI_API int mi_memcpy_d2d_async(int device, void* stream, void* dev_dst, const void* dev_src, size_t bytes);
MI_API int mi_device_synchronize(int device);
/* ---- PyramidStack's step methods, one at a time, on host float32 images of c = 1 or 3 interleaved channels -- the
 * reference's public methods of the same names (pyramid.py:24-63), always in ITS evaluation order (row-major 25-tap chain):
 *   MI_PYR_CONVOLVE        in (h, w, c)                  -> out (h, w, c)             cv2.filter2D, REFLECT101   (:24-25)
 *   MI_PYR_REDUCE          in (h, w, c)                  -> out (ceil(h/2), ceil(w/2), c)                      (:27-32)
 *   MI_PYR_EXPAND          in (h, w, c)                  -> out (2h, 2w, c)                                    (:34-46)
 *   MI_PYR_FUSE_LAPLACIAN  in (n, h, w, 3)               -> out (h, w, 3)   energy, first arg-max, where-sum  (:48-55)
 *   MI_PYR_COLLAPSE_STEP   in = layer (h, w, c), in2 = coarser image (h2, w2, c) -> expand(in2)[:h, :w] + in   (:59-63)
 *   MI_PYR_CLIP_ABS        in (h, w, c)                  -> clip(abs(in), 0, maxv)                             (:64)    */
enum { MI_PYR_CONVOLVE = 0, MI_PYR_REDUCE = 1, MI_PYR_EXPAND = 2, MI_PYR_FUSE_LAPLACIAN = 3, MI_PYR_COLLAPSE_STEP = 4,
       MI_PYR_CLIP_ABS = 5 };
MI_API int mi_pyr_step(int device, int op, int use_fma, double gen_kernel, const void* host_in, const void* host_in2, int n,
                int h, int w, int c, int h2, int w2, double maxv, void* host_out);
/* free / total device memory in bytes (hipMemGetInfo): callers size resident stacks and batches with it */
MI_API int mi_device_mem_info(int device, size_t* free_bytes, size_t* total_bytes);

/* ---- stacker handle ---- */
MI_API int mi_stack_create(mi_stack_t** out, const mi_stack_params_t* params);
MI_API void mi_stack_destroy(mi_stack_t* s);
MI_API int mi_stack_reset(mi_stack_t* s);
MI_API int mi_stack_levels(const mi_stack_t* s, int* levels);
MI_API int mi_stack_level_shape(const mi_stack_t* s, int level, int* h, int* w);
MI_API int mi_stack_frames_pushed(const mi_stack_t* s, int* n);
/* global index of this handle's first frame (multi-GPU frame sharding); default 0 */
MI_API int mi_stack_set_first_index(mi_stack_t* s, int first_global_index);
/* interleaved frame shards (rank r of W holds frames r, r + W, ...: every rank sees the whole focus range, so its winners are
 * as coherent as the whole stack's): the global index of the handle's k-th frame = first_global_index + k * stride.  The
 * kernels number the frames consecutively; mi_stack_export_indices rewrites the winner indices of one level (level == levels,
 * levels + 1: the base twins; -1: all) into the global numbering, in place -- before they leave the handle (the cross-GPU
 * combine breaks ties by them: np.argmax's first maximum, pyramid.py:51; the index taps call it themselves).  After an export
 * the handle takes no more frames until mi_stack_reset.  Default stride 1: nothing to export. */
MI_API int mi_stack_set_index_stride(mi_stack_t* s, int stride);
MI_API int mi_stack_export_indices(mi_stack_t* s, int level);

/* One frame from host memory (H rows of row_stride_bytes; 0 = tightly packed).
 * Returns after the work is enqueued; the host buffer may be reused on return. */
MI_API int mi_stack_push_frame(mi_stack_t* s, const void* host_bgr, size_t row_stride_bytes);
/* The same for a frame that already lies in PINNED host memory with contiguous rows (mi_host_alloc, or any buffer pinned
 * with mi_host_register -- e.g. the array an image decoder writes into): the upload reads the caller's buffer directly, no
 * bounce copy (the reference's frames come from cv2.imread, stack.py:61-97 -> utils.py:10-19; config 5 is upload-bound).
 * The buffer must stay untouched until mi_stack_wait_uploads reports the upload done.  MI_ERR_INVALID when the memory is
 * not pinned. */
MI_API int mi_stack_push_frame_pinned(mi_stack_t* st, const void* host_bgr, size_t row_stride_bytes);
/* blocks until at most `max_outstanding` of the frames pushed with mi_stack_push_frame_pinned are still being uploaded
 * (0: all of them are on the device; a decoder that cycles k pinned buffers calls it with k - 1 before it reuses one) */
MI_API int mi_stack_wait_uploads(mi_stack_t* st, int max_outstanding);
/* pinned host memory: allocate / free, or pin / unpin memory the caller owns */
MI_API int mi_host_alloc(void** ptr, size_t bytes);
MI_API int mi_host_free(void* ptr);
MI_API int mi_host_register(void* ptr, size_t bytes);
MI_API int mi_host_unregister(void* ptr);
/* n frames already resident in device memory (tightly packed rows, frames
 * frame_stride_bytes apart), dtype = params.in_dtype.  The frames must be complete either on the host's
 * timeline (a finished copy / kernel) or in the order of the handle's stream (mi_stack_stream): work the
 * caller enqueued there -- a warp, a table apply -- is waited for on the device, not on the host. */
MI_API int mi_stack_push_frames_device(mi_stack_t* s, const void* dev_frames, int n,
                                size_t frame_stride_bytes);
/* wait for everything enqueued so far */
MI_API int mi_stack_sync(mi_stack_t* s);

/* wait until the selection state of `level` covers every pushed frame.  level 0 (3/4 of the state) is final while the
 * coarser levels of the last batch are still running: the cross-GPU exchange of level 0 can start behind this call
 * (multigpu.Combiner does).  Any other level: same as mi_stack_sync. */
MI_API int mi_stack_sync_level(mi_stack_t* s, int level);

/* base fusion + collapse + abs/clip/truncating cast.  The handle stays valid
 * (taps readable) until reset/destroy.  host_out: H x W x 3 of out_dtype. */
MI_API int mi_stack_finish(mi_stack_t* s, void* host_out, size_t row_stride_bytes);
MI_API int mi_stack_finish_device(mi_stack_t* s, void* dev_out);

MI_API int mi_stack_get_level(mi_stack_t* s, int level, int what, void* host_out, size_t out_bytes);

/* Device pointers of the running selection state of one level (level ==
 * levels addresses the base: energy -> entropy max, lap -> base_e; the
 * deviation twin comes from level == levels + 1).  level == -1: the state of ALL levels and of both
 * base twins at once -- float-32 stacks keep it in three contiguous slabs (per-level segments padded
 * to 64 pixels; npixels counts the padding) so that the combine can move it as one flat vector.
 * For the cross-GPU combine.
 * Synchronises the handle: on return all enqueued work has finished, so the pointers may be
 * used from any other stream. */
MI_API int mi_stack_state(mi_stack_t* s, int level, void** dev_energy, void** dev_lap, void** dev_index,
                   size_t* npixels);
/* the HIP stream the handle launches on (hipStream_t as void*) */
MI_API int mi_stack_stream(mi_stack_t* s, void** stream);

/* ---- per-kernel timing (hipEvent pairs on the handle's stream) ---- */
enum {
    MI_PROF_LEVEL = 0,    /* fused level launches of levels >= 1 (simple impl: whole frames) */
    MI_PROF_BASE = 1,     /* base-level feature kernels                                      */
    MI_PROF_COLLAPSE = 2, /* base fusion + collapse + finalise                               */
    MI_PROF_LEVEL0 = 3,   /* fused level-0 launches: the dominant kernel                     */
    MI_PROF_KINDS = 4
};
MI_API int mi_stack_profile(mi_stack_t* s, int enable);
/* sums since the last reset; algorithmic_bytes follows SURVEY.md 8(d).  The consecutive launches of one level pass on
 * one stream share an event pair (first start .. last end, idle gaps between them included); `launches` counts kernels. */
MI_API int mi_stack_profile_get(mi_stack_t* s, int kind, double* total_ms, int64_t* launches,
                         double* algorithmic_bytes);

/* ---- cross-GPU combine kernels (local parts of the frame-sharded reduce) ---- */
/* cand_e: [n][npix] f32, cand_lap: [n][npix*3] f32, cand_idx: [n][npix] i32 (may be
 * NULL), candidates in ascending global-frame order; writes the first-max winner's
 * energy / lap / index to out_*.  `stream` is a hipStream_t (NULL = default stream). */
MI_API int mi_combine_select(int device, void* stream, int n, const void* cand_e, const void* cand_lap,
                      const void* cand_idx, size_t npix, void* out_e, void* out_lap, void* out_idx);

/* "Winners only" form of the same reduce (what multigpu.Combiner uses; <= 16 ranks): the ranks exchange energies only,
 * mi_combine_winner names the winning rank of every pixel of a chunk (first maximum in rank order), every rank learns the
 * whole winner map (1 byte per pixel), and each rank sends just the payload rows it WON -- packed in pixel order by
 * mi_combine_pack -- straight to the collapsing rank, which puts them in place with mi_combine_unpack.
 * mi_combine_plan: per-block start positions of every rank's rows in its packed buffer (`plan`: device scratch of
 * mi_combine_plan_bytes) and the row totals per rank on the host (synchronises the stream).
 * dev_bufs: DEVICE array of n_ranks device pointers, entry r = rank r's packed rows (the entry of `rank` itself unused). */
MI_API int mi_combine_winner(int device, void* stream, int n_ranks, const void* cand_e, size_t npix, void* win_u8);
/* the same for interleaved shards (mi_stack_set_index_stride): the candidates' global frame indices travel with their
 * energies and break ties (the lower index wins) -- the rank order is not the frame order there */
MI_API int mi_combine_winner_idx(int device, void* stream, int n_ranks, const void* cand_e, const void* cand_idx, size_t npix,
                                 void* win_u8);
MI_API size_t mi_combine_plan_bytes(size_t npix, int n_ranks);
MI_API int mi_combine_plan(int device, void* stream, const void* win_u8, size_t npix, int n_ranks, void* plan,
                           int64_t* totals);
MI_API int mi_combine_pack(int device, void* stream, const void* win_u8, size_t npix, int n_ranks, int rank,
                           const void* plan, const void* src, int width, void* out);
MI_API int mi_combine_unpack(int device, void* stream, const void* win_u8, size_t npix, int n_ranks, int rank,
                             const void* plan, const void* const* dev_bufs, int width, void* dst);

/* ---- alignment apply step: cv2.warpAffine(img, M, (w,h), borderMode, borderValue) for the
 * ALIGN_RIGID transform of align_images (reference algorithms/align.py:238-251), H x W x 3
 * uint8 / uint16.  M is the 2x3 src->dst matrix as OpenCV takes it (row-major, 6 doubles).
 * border_mode: 0 = BORDER_CONSTANT(border_value), 1 = BORDER_REPLICATE,
 *              2 = BORDER_REPLICATE_BLUR: replicate, then pixels whose warped all-ones mask is 0
 *                  are replaced by GaussianBlur(warp, (blur_ksize, blur_ksize), blur_sigma).
 * mask (optional, H x W bytes): the warped all-ones uint8 mask (align.py:245-247).
 * The _device form works on device buffers (dev_tmp: one image of scratch, needed for mode 2). */
MI_API int mi_warp_affine(int device, const void* host_src, void* host_dst, void* host_mask, int height,
                   int width, int dtype, const double* M, int border_mode, const double* border_value,
                   int blur_ksize, double blur_sigma);
MI_API int mi_warp_affine_device(int device, void* stream, const void* dev_src, void* dev_dst, void* dev_tmp,
                          void* dev_mask, int height, int width, int dtype, const double* M,
                          int border_mode, const double* border_value, int blur_ksize, double blur_sigma);

/* The same for the ALIGN_HOMOGRAPHY transform: cv2.warpPerspective(img, M, (w, h), borderMode, borderValue) of image and
 * all-ones mask (reference algorithms/align.py:231-237) + the same blurred-border composite.  M: the 3x3 src->dst matrix
 * as OpenCV takes it (row-major, 9 doubles). */
MI_API int mi_warp_perspective(int device, const void* host_src, void* host_dst, void* host_mask, int height, int width,
                        int dtype, const double* M, int border_mode, const double* border_value, int blur_ksize,
                        double blur_sigma);
MI_API int mi_warp_perspective_device(int device, void* stream, const void* dev_src, void* dev_dst, void* dev_tmp,
                               void* dev_mask, int height, int width, int dtype, const double* M, int border_mode,
                               const double* border_value, int blur_ksize, double blur_sigma);

/* ---- GPU transform estimator (new capability; north_star's "ECC warp-affine alignment loop"):
 * Enhanced-Correlation-Coefficient maximisation of a 4-DoF similarity -- the motion model of the
 * reference's default ALIGN_RIGID estimate (cv2.estimateAffinePartial2D, algorithms/align.py:141-148)
 * -- coarse-to-fine on a Gaussian pyramid of the two H x W x 3 uint8/uint16 images.
 * M_out: 2x3 row-major matrix mapping the MOVING image onto the REFERENCE, i.e. what
 * cv2.warpAffine / mi_warp_affine take.  cc_out: final correlation coefficient, iters_out: total
 * Gauss-Newton iterations.  max_levels <= 0: automatic pyramid depth. */
MI_API int mi_ecc_similarity(int device, const void* host_ref, const void* host_mov, int height, int width,
                      int dtype, int max_levels, int max_iters, double eps, double* M_out, double* cc_out,
                      int* iters_out);

/* Device-resident form of the same estimator for the AlignFrames loop (align.py:255-330): the
 * pyramids are allocated once, the reference frame's pyramid is built once per reference, every
 * moving frame costs one pyramid build plus the Gauss-Newton iterations.  `subsample` folds the
 * reference's fast sub-sampling img[::s, ::s] (utils.py img_subsample) into the first kernel; the
 * returned M is in full-resolution pixels (translation scaled back as align.py:224-231 does).
 * dev_ref / dev_mov: H x W x 3 device images of the handle's dtype.  The kernels run on `stream`, or on
 * a stream the handle owns when `stream` is NULL (handles driven from different host threads then
 * run side by side);
 * mi_aligner_estimate returns after the last iteration (the Gauss-Newton steps run on the device; the host reads the frames'
 * `active` flags back every few iterations). */
typedef struct mi_aligner* mi_aligner_t;
MI_API int mi_aligner_create(mi_aligner_t* out, int device, int height, int width, int dtype, int subsample,
                      int max_levels);
/* sub-sample like the reference's default (fast_subsampling = False: cv2.resize(INTER_AREA), the mean of every s x s
 * block, utils.py:83) instead of img[::s, ::s]; call before mi_aligner_set_reference */
MI_API int mi_aligner_set_area_subsampling(mi_aligner_t al, int enable);
/* Coarse initialiser (north_star: "ECC/phase-correlation"): before the Gauss-Newton iteration every estimate takes the
 * TRANSLATION that phase correlation finds on the finest pyramid level of at most 512 pixels per side (Hann window, DFT,
 * normalised cross-power spectrum, inverse DFT, 5 x 5 centroid around the peak -- cv2.phaseCorrelate's recipe) as the
 * starting point; a peak with a response below 0.02 is ignored.  Extends the capture range from a few pixels of the
 * coarsest level to half the frame.  Off by default. */
MI_API int mi_aligner_set_phase_init(mi_aligner_t al, int enable);
/* The correlation alone, for two float32 planes (height x width, at most 1024 pixels per side) on the device:
 * out3 = (dx, dy, response) with mov(x + dx, y + dy) ~ ref(x, y). */
MI_API int mi_phase_correlate_device(int device, void* stream, const void* dev_ref, const void* dev_mov, int height, int width,
                              double* out3);
MI_API int mi_aligner_destroy(mi_aligner_t al);
MI_API int mi_aligner_set_reference(mi_aligner_t al, void* stream, const void* dev_ref);
MI_API int mi_aligner_estimate(mi_aligner_t al, void* stream, const void* dev_mov, int max_iters, double eps,
                        double* M_out, double* cc_out, int* iters_out);
/* n <= 128 moving frames against the same reference in one batched Gauss-Newton: one launch per iteration for the
 * whole batch, the step itself on the device.  M_out: n x 6, cc_out / iters_out: n entries; a frame
 * the method fails on (no overlap, constant image) gets cc = -2 and an identity matrix. */
MI_API int mi_aligner_estimate_batch(mi_aligner_t al, void* stream, const void* const* dev_movs, int n, int max_iters,
                              double eps, double* M_out, double* cc_out, int* iters_out);

/* Every frame of a batch against ANOTHER FRAME OF THE BATCH (ref_of[k] = index of frame k's reference; a frame may name itself:
 * identity, correlation 1): the pyramids of the n frames are built once, frame k's template is frame ref_of[k]'s pyramid.  The
 * chained order of the reference's jobs (stack_framework.py:214-232: frame f against frame f - 1) as ONE batched Gauss-Newton
 * -- the steps are then composed by the caller.  M_out[k]: frame k -> frame ref_of[k], full-resolution pixels. */
MI_API int mi_aligner_estimate_pairs(mi_aligner_t al, void* stream, const void* const* dev_frames, int n_frames, const int* ref_of,
                                     int max_iters, double eps, double* M_out, double* cc_out, int* iters_out);
/* The same iteration started from given transforms instead of from the identity: `M_init` (n x 6, moving -> reference,
 * full-resolution pixels: what an earlier estimate returned) is refined on the `levels` finest pyramid levels (1 = the
 * finest alone).  The chained order of the reference's jobs (step_process, stack_framework.py:214-232) uses it to pull
 * every frame's chain estimate back onto the GLOBAL reference frame, so that the steps' errors do not add up
 * (shinestacker_amd/pipeline.py::_align_chains_device).  Outputs as mi_aligner_estimate_batch. */
MI_API int mi_aligner_refine_batch(mi_aligner_t al, void* stream, const void* const* dev_movs, int n, const double* M_init,
                            int levels, int max_iters, double eps, double* M_out, double* cc_out, int* iters_out);

/* ALIGN_HOMOGRAPHY (align.py:138-140, cv2.findHomography): the same estimate refined to 8 degrees of freedom -- the
 * forward-additive ECC iteration in cv2.findTransformECC's MOTION_HOMOGRAPHY form, on the finest level, from the
 * converged similarity.  M9_out: n x 9 doubles, row-major 3 x 3 (moving -> reference, full-resolution pixels, M[8] = 1),
 * ready for mi_warp_perspective; a frame whose refinement fails or does not raise the correlation keeps its similarity. */
MI_API int mi_aligner_estimate_homography_batch(mi_aligner_t al, void* stream, const void* const* dev_movs, int n, int max_iters,
                                         double eps, double* M9_out, double* cc_out, int* iters_out);

/* ---- the resident align -> stack loop in ONE call (reference: CombinedActions.run_frame over AlignFrames with a fixed
 * reference frame, stack_framework.py:191-232, :269-297, followed by FocusStack, stack.py:101-113; BASELINE config 4).
 * Every frame of `dev_frames` (n_frames x H x W x 3, `frame_stride` bytes apart, the stack handle's in_dtype) except
 * frame `ref_idx` is registered against frame `ref_idx` with the device estimator (`ecc_batch` <= 128 frames per batched
 * Gauss-Newton), warped with align.py:230-251's border handling straight into the stacker's input batch and pushed
 * (`batch_frames` warped frames per push; two batches alternate, so the next one fills while the last is fused); the
 * reference frame passes through untouched (align.py:279-280).  The host-side sequencing that
 * shinestacker_amd/pipeline.py does call by call (~25 library calls per frame) runs inside the library here -- same kernels
 * in the same order on the same streams, same results.
 *   dev_batches : 2 * batch_frames frames of scratch; dev_tmp: one frame, dev_mask: H x W bytes (border blur).
 *   M_out       : n_frames x 9 doubles; row i holds the 2x3 (transform 0) or 3x3 (transform 1: the similarity applied through
 *                 warpPerspective) matrix of frame i, zeros for the reference frame;  cc_out: n_frames correlation coefficients.
 * A frame whose correlation stays below min_correlation stops the loop: MI_ERR_ALIGNMENT, *failed_frame = its index
 * (-> AlignmentError).  Only COMPLETE batches of `batch_frames` warped frames before it have been pushed: the frames
 * already warped into the partially filled batch are not flushed, so the stack holds a prefix of the frames and must be
 * reset (mi_stack_reset) before it is used again. */
typedef struct mi_align_stack_opts {
    int transform;            /* 0: ALIGN_RIGID (warpAffine), 1: ALIGN_HOMOGRAPHY (warpPerspective) */
    int border_mode;          /* as mi_warp_affine_device */
    double border_value[4];
    int blur_ksize;
    double blur_sigma;
    double min_correlation;
    int max_iters;
    double eps;
    int ecc_batch;            /* 1..128 */
    int batch_frames;         /* >= 1 */
} mi_align_stack_opts_t;
/* optional: every aligned frame (not the reference frame) is balanced in place with the LINEAR map before it is pushed
 * (the example projects' order: align, balance, stack -- CombinedActions([AlignFrames, BalanceFrames])); the fields are
 * mi_balance_linear_device's, plus the 8-bit colour conversions around it for the HSV / HLS channel modes (-1 = none) */
typedef struct mi_balance_linear_opts {
    int mode, subsample, fast;
    double mask_size;
    int lo, hi, first_channel;
    int cvt_to, cvt_from;      /* MI_CVT_* codes applied before / after; -1 = stay in BGR */
    double ref_means[3];
    void* dev_hist_scratch;    /* 3 * nbins uint32 */
    void* dev_lut;             /* 3 * nbins entries of the image dtype */
    double* dev_corr_out;      /* NULL or device array of n_frames * ncorr doubles (row i = frame i) */
    int ncorr;
} mi_balance_linear_opts_t;
MI_API int mi_align_stack_device(mi_stack_t* st, mi_aligner_t al, const void* dev_frames, int n_frames, size_t frame_stride,
                          int ref_idx, const mi_align_stack_opts_t* opts, const mi_balance_linear_opts_t* balance,
                          void* dev_batches, void* dev_tmp, void* dev_mask, double* M_out, double* cc_out, int* failed_frame);

/* ---- BalanceFrames device steps (reference algorithms/balance.py).
 * mi_histogram: histogram of an H x W x 3 uint8/uint16 BGR image as balance.py:158-180
 * (calc_hist_1ch) takes it -- after sub-sampling by `subsample` (fast: img[::s, ::s]; otherwise the
 * integer-factor area mean of cv2.resize(INTER_AREA), utils.py:79-86) and, when mask_size > 0,
 * inside the centred circle of radius min(w, h) * mask_size / 2 of the sub-sampled image.
 * mode 0: one histogram per channel B, G, R (RGBCorrection, balance.py:264-266) -> counts[3][nbins];
 * mode 1: histogram of cv2.cvtColor(BGR2GRAY) (LumiCorrection, balance.py:235-236) -> counts[1][nbins];
 * nbins = 256 (uint8) or 65536 (uint16).  counts: int64, as np.histogram returns.
 * The _device form takes a device image and a device scratch of 3 * nbins uint32. */
MI_API int mi_histogram(int device, const void* host_img, int height, int width, int dtype, int mode,
                 int subsample, int fast, double mask_size, int64_t* counts);
MI_API int mi_histogram_device(int device, void* stream, const void* dev_img, void* dev_scratch, int height,
                        int width, int dtype, int mode, int subsample, int fast, double mask_size,
                        int64_t* counts);
/* the same for n frames with ONE host round trip: frame k's histogram lands in counts + k * nch * nbins; dev_scratch holds
 * n * 3 * nbins uint32 (the resident pipeline balances a whole batch of warped frames behind one synchronisation instead
 * of one per frame: balance.py:181-201 for the GAMMA / MATCH_HIST maps, whose tables SciPy builds on the host) */
MI_API int mi_histogram_device_batch(int device, void* stream, const void* const* dev_imgs, int n, void* dev_scratch,
                              int height, int width, int dtype, int mode, int subsample, int fast, double mask_size,
                              int64_t* counts);
/* mi_apply_lut: dst[p][c] = lut[nlut == 1 ? 0 : c][src[p][c]] -- cv2.LUT (uint8) / np.take (uint16)
 * as balance.py:30-50 applies them: one table for all channels (LUMI) or one per channel (RGB).
 * lut: nlut tables of nbins entries of the image dtype. */
MI_API int mi_apply_lut(int device, const void* host_src, void* host_dst, int height, int width, int dtype,
                 const void* host_lut, int nlut);
MI_API int mi_apply_lut_device(int device, void* stream, const void* dev_src, void* dev_dst, size_t npixels,
                        int dtype, const void* dev_lut, int nlut);
/* The whole LINEAR correction of one device frame, in place, enqueued on `stream` without any host round trip: histogram
 * (as mi_histogram_device) -> LinearMap's table (balance.py:87-105: ratio = reference mean / histogram mean over the bins
 * [lo, hi), table[i] = trunc(clip(i * ratio, 0, max)), float64) -> table apply.  mode 1 (luminance): one table for the
 * three channels; mode 0: one per channel, the channels below `first_channel` unchanged (1 = HSV / HLS: hue passes).
 * ref_means: host array of the reference frame's means, one per corrected channel.  dev_hist_scratch: 3 * nbins uint32;
 * dev_lut: 3 * nbins entries of the image dtype; dev_corr_out: NULL or device array that receives the ratios.
 * A histogram that is empty inside [lo, hi) -- np.average raises ZeroDivisionError in the reference there, which an
 * asynchronous call cannot do -- gives ratio 1 (the channel is left unchanged). */
MI_API int mi_balance_linear_device(int device, void* stream, void* dev_img, void* dev_hist_scratch, void* dev_lut,
                             int height, int width, int dtype, int mode, int subsample, int fast, double mask_size,
                             int lo, int hi, int first_channel, const double* ref_means, double* dev_corr_out);

/* 8-bit BGR <-> HSV / HLS (cv2.cvtColor COLOR_BGR2HSV, _HSV2BGR, _BGR2HLS, _HLS2BGR): the pre- and post-processing of the
 * HSV / HLS channel modes of BalanceFrames (balance.py:340-363: SVCorrection / LSCorrection balance S and V, or L and S,
 * and leave the hue alone).  uint8 only, as in OpenCV (MI_ERR_UNSUPPORTED for 16-bit; the reference raises there too).
 * In place is allowed (dev_dst == dev_src). */
enum { MI_CVT_BGR2HSV = 0, MI_CVT_HSV2BGR = 1, MI_CVT_BGR2HLS = 2, MI_CVT_HLS2BGR = 3 };
MI_API int mi_cvt_color(int device, const void* host_src, void* host_dst, int height, int width, int dtype, int code);
MI_API int mi_cvt_color_device(int device, void* stream, const void* dev_src, void* dev_dst, size_t npixels, int dtype,
                        int code);

/* ---- DepthMapStack: the second stacker behind the same plug-in boundary ----
 * Replaces the arithmetic of DepthMapStack.focus_stack (reference algorithms/depth_map.py:64-123) for
 * both float types: push = the first file loop (:67-75: read, img_bw, then per frame get_sobel_map :28-34
 * or get_laplacian_map :36-41 and the running np.max :88); finish = energies / max (:90), smooth_energy
 * (:43-52), get_focus_map (:54-62), the per-frame pyrDown / pyrUp weighted Laplacian pyramids (:94-112),
 * the collapse and np.clip(np.absolute()).astype (:117-123).  The handle keeps every pushed frame and one
 * float32 plane per frame in device memory (the reference re-reads every file in its second loop).
 * Parity: oracle/depth_map_oracle.py (OpenCV primitives restated, unpinned -- see DESIGN.md). */
enum { MI_DM_MAP_AVERAGE = 0, MI_DM_MAP_MAX = 1 };          /* constants.py:147-148 */
enum { MI_DM_ENERGY_LAPLACIAN = 0, MI_DM_ENERGY_SOBEL = 1 };  /* constants.py:145-146 */
typedef struct mi_dmap mi_dmap_t;
typedef struct mi_dmap_params {
    int32_t height, width;
    int32_t dtype;         /* MI_U8 / MI_U16: frames in, fused frame out                       */
    int32_t device;
    int32_t map_type;      /* MI_DM_MAP_*       (depth_map.py:11)                              */
    int32_t energy;        /* MI_DM_ENERGY_*    (:12)                                          */
    int32_t kernel_size;   /* cv2.Laplacian aperture, odd, <= 15 (:13)                         */
    int32_t blur_size;     /* cv2.GaussianBlur size, odd, <= 31 (:14)                          */
    int32_t smooth_size;   /* cv2.bilateralFilter diameter, <= 0: no smoothing, <= 31 (:15)    */
    int32_t levels;        /* blend pyramid levels, >= 1 (:17)                                 */
    float temperature;     /* softmax temperature of the MAX map (:16)                         */
    int32_t float_type;    /* MI_F32 / MI_F64 (:18, base_stack_algo.py:14-17)                  */
} mi_dmap_params_t;
MI_API void mi_dmap_default_params(mi_dmap_params_t* p);   /* constants.py:151-157 */
MI_API int mi_dmap_create(mi_dmap_t** out, const mi_dmap_params_t* params);
MI_API void mi_dmap_destroy(mi_dmap_t* d);
MI_API int mi_dmap_reset(mi_dmap_t* d);                    /* forget the frames, keep the buffers */
MI_API int mi_dmap_frames_pushed(const mi_dmap_t* d, int* n);
/* the MAX map's temperature as a double: mi_dmap_params_t carries it as a float, which is exact for float-32 stacks (NumPy
 * divides a float32 array by the float32 of a Python float) but not for float-64 ones (depth_map.py:60) */
MI_API int mi_dmap_set_temperature(mi_dmap_t* d, double temperature);
/* The stacker's steps one at a time, on n host planes of height x width -- the reference's public methods of the same
 * names (depth_map.py:28-62; its tests call them), run by the kernels of the fused path with the handle's options:
 *   stage 0  get_sobel_map      gray planes (float_type) -> |Sobel x| + |Sobel y|                (:28-34)
 *   stage 1  get_laplacian_map  gray planes (float_type) -> |Laplacian(GaussianBlur)|            (:36-41)
 *   stage 2  smooth_energy      energy planes (float_type) -> float32 planes, cv2.bilateralFilter (:43-52)
 *   stage 3  get_focus_map      n energy planes -> n weight planes; both float32 when smooth_size > 0 or float_type is
 *                               float-32, else float64; a zero total gives weight 0          (:54-62) */
MI_API int mi_dmap_planes(mi_dmap_t* d, int stage, const void* host_in, int n, void* host_out);
/* H x W x 3 BGR frame of `dtype`; row_stride_bytes 0 = packed.  The host form returns once the caller's
 * buffer may be reused; the device form copies on the handle's stream. */
MI_API int mi_dmap_push_frame(mi_dmap_t* d, const void* host_bgr, size_t row_stride_bytes);
MI_API int mi_dmap_push_frame_device(mi_dmap_t* d, const void* dev_bgr);
/* fused frame (dtype, H x W x 3) to host / device memory; both return after the work has completed */
MI_API int mi_dmap_finish(mi_dmap_t* d, void* host_out, size_t row_stride_bytes);
MI_API int mi_dmap_finish_device(mi_dmap_t* d, void* dev_out);

/* ---- synthetic stack generator, device side ---- */
MI_API int mi_synth_frames_device(int device, void* dev_out, int dtype, int height, int width,
                           int first_frame, int n_frames, int stack_size, uint32_t seed);

#ifdef __cplusplus
}
#endif
#endif /* MI355STACK_H */
