"""In-memory align -> stack pipeline (BASELINE config 4).

The reference round-trips every frame through image files between `AlignFrames` and
`FocusStack` (stack_framework.py:269-297, pyramid.py:158,172).  Here a frame is uploaded once,
warped on the device (`mi_warp_affine_device`) straight into the stacker's input batch and fused
(`mi_stack_push_frames_device`); only the transform estimate runs on the host.

Result: identical to `align_images` on every frame (reference frame passed through untouched)
followed by `PyramidStack` on the aligned frames -- without the intermediate quantised files the
two-action reference job writes, which for lossless formats changes nothing.
"""
import ctypes as C
import time

import numpy as np

from . import _lib
from .align import (_BORDER_CODE, _DEFAULT_ALIGNMENT_CONFIG, _DEFAULT_FEATURE_CONFIG,
                    _DEFAULT_MATCHING_CONFIG, img_subsample, rescale_transform, resolve_estimator)
from .defaults import constants, resolve_arith
from .errors import AlignmentError, InvalidOptionError
from .imageio import validate_image


def align_and_stack(frames, ref_idx=-1, estimator=None, alignment_config=None, feature_config=None,
                    matching_config=None, device=0, batch_frames=16, check_running=None,
                    **stack_kwargs):
    """Align every frame to frames[ref_idx] (fixed reference, `step_process=False` order,
    stack_framework.py:191-232) and fuse them.  `frames`: sequence of H x W x 3 uint8/uint16 BGR
    arrays.  Returns (fused image, list of n_good_matches)."""
    _lib.require_device()
    n = len(frames)
    if n == 0:
        raise ValueError("no frames")
    feature_config = {**_DEFAULT_FEATURE_CONFIG, **(feature_config or {})}
    matching_config = {**_DEFAULT_MATCHING_CONFIG, **(matching_config or {})}
    cfg = {**_DEFAULT_ALIGNMENT_CONFIG, **(alignment_config or {})}
    if cfg['border_mode'] not in _BORDER_CODE:
        raise InvalidOptionError("border_mode", cfg['border_mode'])
    if cfg['transform'] not in (constants.ALIGN_RIGID, constants.ALIGN_HOMOGRAPHY):
        raise InvalidOptionError("transform", cfg['transform'])
    homography = cfg['transform'] == constants.ALIGN_HOMOGRAPHY
    min_matches = 4 if homography else 3
    estimator = resolve_estimator(estimator, device)
    if ref_idx == -1:
        ref_idx = n // 2
    ref = np.ascontiguousarray(frames[ref_idx])
    h, w = ref.shape[:2]
    dt = ref.dtype
    fb = h * w * 3 * dt.itemsize
    lib = _lib.load()
    stack_kwargs["arith"] = resolve_arith(stack_kwargs.get("arith"), stack_kwargs.get("float_type"))   # one default for every entry point
    stack = _lib.Stack(h, w, in_dtype=dt, out_dtype=dt, device=device, batch_frames=batch_frames,
                       **stack_kwargs)
    src = _lib.DeviceBuffer(fb, device)            # uploaded moving frame
    tmp = _lib.DeviceBuffer(fb, device)            # warp scratch (border blur)
    mask = _lib.DeviceBuffer(h * w, device)
    batch = _lib.DeviceBuffer(fb * batch_frames, device)
    mode = _BORDER_CODE[cfg['border_mode']]
    bv = (C.c_double * 4)(*(list(cfg['border_value']) + [0, 0, 0, 0])[:4])
    matches, filled = [], 0

    def flush():
        nonlocal filled
        if filled:
            lib.mi_device_synchronize(device)  # warps ran on the default stream
            stack.push_frames_device(batch.ptr, filled, fb)
            stack.sync()                       # the batch buffer is reused next
            filled = 0

    for i, fr in enumerate(frames):
        fr = np.ascontiguousarray(fr)
        validate_image(fr, (h, w), dt)
        dst = batch.ptr + filled * fb
        if i == ref_idx:
            batch.upload(fr, filled * fb)      # reference frame: untouched (align.py:279-280)
            matches.append(0)
        else:
            sub = cfg['subsample']
            while True:
                a, b = (img_subsample(fr, sub, cfg['fast_subsampling']),
                        img_subsample(ref, sub, cfg['fast_subsampling'])) if sub > 1 else (fr, ref)
                ng, m = estimator(a, b, feature_config, matching_config, cfg)
                if ng > cfg['min_good_matches'] or sub == 1:
                    break
                sub = 1
            matches.append(ng)
            if ng < min_matches or m is None:
                raise AlignmentError(i, f"too few matches found: {ng} < {min_matches}")
            m = np.asarray(m)
            if homography and m.shape == (2, 3):
                m = np.vstack([m, [0.0, 0.0, 1.0]])
            if sub > 1:
                m = rescale_transform(m, cfg['transform'], sub, fr.shape, a.shape)
            mm = (C.c_double * m.size)(*np.asarray(m, dtype=np.float64).reshape(-1))
            src.upload(fr)
            warp = lib.mi_warp_perspective_device if homography else lib.mi_warp_affine_device
            _lib.check(warp(device, None, src.ptr, dst, tmp.ptr, mask.ptr, h, w, _lib.DTYPE_CODE[dt], mm, mode, bv, 21,
                            float(cfg['border_blur'])))
        filled += 1
        if filled == batch_frames:
            flush()
        if check_running is not None and check_running() is False:
            from .errors import RunStopException
            raise RunStopException("align_and_stack")
    flush()
    out = stack.finish()
    stack.close()
    return out, matches


# chain_refine: how far (pixels, at the frame corners) the refinement against the global reference frame may move a chain
# estimate before it is distrusted and the chain estimate kept
CHAIN_REFINE_MAX_SHIFT = 2.0


def _corner_shift(m0, m1, height, width):
    """largest displacement between two 2x3 transforms at the frame corners"""
    pts = np.array([[0.0, 0.0, 1.0], [width - 1.0, 0.0, 1.0], [0.0, height - 1.0, 1.0], [width - 1.0, height - 1.0, 1.0]]).T
    return float(np.abs(np.asarray(m0) @ pts - np.asarray(m1) @ pts).max())


CHAIN_PAIR_WORKERS = 4   # estimator handles (one host thread each) that share the chain's neighbour pairs


def _to33(m):
    m = np.asarray(m, np.float64)
    return m if m.shape == (3, 3) else np.vstack([m, [0.0, 0.0, 1.0]])


def _align_chains_pairs_device(lib, dev_frames, aligned, n_frames, height, width, dt, ref_idx, cfg, min_correlation, max_iters,
                               device, corr=None, chain_refine=True, reuse=None):
    """`step_process=True` without its serial dependency (round 6).  The reference registers frame f against the ALIGNED frame
    f - 1 (stack_framework.py:214-232): neighbours look alike, so the estimate is easy, and the aligned neighbour carries the
    transform found so far.  The same chain, factored: every frame is registered against its UNWARPED neighbour -- independent
    estimates: ALIGN_RIGID as batched Gauss-Newton over up to 127 pairs at once (mi_aligner_estimate_pairs), ALIGN_HOMOGRAPHY
    pair by pair on CHAIN_PAIR_WORKERS estimator handles --, the steps are composed along each chain in
    float64 (frame -> neighbour -> ... -> reference), and, as in the serial form, every composed estimate is refined against
    the GLOBAL reference frame (one batched call for all frames: the datum that keeps the steps' errors from adding up,
    `chain_refine`); then all frames are warped.  No step waits for a warp, nothing synchronises the device per frame.
    Differences from the serial form: a step's reference has not been resampled (its estimate sees the sharper image) and,
    with `corr`, has not been balanced yet (the ECC criterion is invariant to gain and offset); the balanced, aligned frames
    the stack sees are produced exactly as before.  `reuse` (ALIGN_RIGID): (pair estimator, global-reference estimator, frame
    scratch, mask scratch) of an earlier call -- nothing is allocated then.  Returns (transforms, correlation coefficients
    -- the refinement's where it was accepted)."""
    import threading
    fb = height * width * 3 * dt.itemsize
    mode = _BORDER_CODE[cfg['border_mode']]
    bv = (C.c_double * 4)(*(list(cfg['border_value']) + [0, 0, 0, 0])[:4])
    homography = cfg['transform'] == constants.ALIGN_HOMOGRAPHY
    sub = max(1, int(cfg['subsample']))
    transforms, ccs = [None] * n_frames, [1.0] * n_frames
    _lib.check(lib.mi_memcpy_d2d(device, aligned + ref_idx * fb, dev_frames + ref_idx * fb, fb))   # align.py:279-280
    chains = [list(range(ref_idx + 1, n_frames)), list(range(ref_idx - 1, -1, -1))]
    pairs = []                      # (frame, its neighbour towards the reference frame)
    for ch in chains:
        prev = ref_idx
        for i in ch:
            pairs.append((i, prev))
            prev = i
    if not pairs:
        return transforms, ccs
    step_m, step_cc, errors = {}, {}, []
    import os, time
    _t = [time.perf_counter()]
    def _lap(label):
        if os.environ.get("MI_CHAIN_TIMING"):
            lib.mi_device_synchronize(device)
            _t.append(time.perf_counter())
            print(f"[chain] {label}: {(_t[-1] - _t[-2]) * 1e3:.1f} ms", flush=True)

    if not homography:
        # ALIGN_RIGID: the pairs of a chain as batches of ONE Gauss-Newton each (mi_aligner_estimate_pairs: the pyramids of the
        # batch's frames are built once, a frame's template is its neighbour's pyramid).  A batch = a run of up to MAX_BATCH
        # frames along a chain, led by the frame the run's first step refers to (the reference frame, or the last frame of
        # the run before), which is registered against itself.
        al = None
        try:
            al = reuse[0] if reuse else _lib.Aligner(height, width, dt, subsample=sub, device=device, fast=bool(cfg['fast_subsampling']))
            for ch in chains:
                lead = ref_idx
                for b0 in range(0, len(ch), _lib.Aligner.MAX_BATCH - 1):
                    run = ch[b0:b0 + _lib.Aligner.MAX_BATCH - 1]
                    frames_b = [lead] + run
                    ms, cs, _ = al.estimate_pairs([dev_frames + i * fb for i in frames_b], [0] + list(range(len(run))),
                                                  max_iters=max_iters)
                    for k, i in enumerate(run, start=1):
                        if not cs[k] >= min_correlation:
                            raise AlignmentError(i, f"correlation {cs[k]:.3f} < {min_correlation}")
                        step_m[i], step_cc[i] = _to33(ms[k]), float(cs[k])
                    lead = run[-1]
        finally:
            if al is not None and not reuse:
                al.close()
    else:
        nw = max(1, min(CHAIN_PAIR_WORKERS, len(pairs)))

        def worker(k):
            al = None
            try:
                al = _lib.Aligner(height, width, dt, subsample=sub, device=device, fast=bool(cfg['fast_subsampling']))
                lo, hi = k * len(pairs) // nw, (k + 1) * len(pairs) // nw
                for i, prev in pairs[lo:hi]:
                    al.set_reference(dev_frames + prev * fb)
                    ms, cs, _ = al.estimate_homography_batch([dev_frames + i * fb], max_iters=max_iters)
                    if not cs[0] >= min_correlation:
                        raise AlignmentError(i, f"correlation {cs[0]:.3f} < {min_correlation}")
                    step_m[i], step_cc[i] = _to33(ms[0]), float(cs[0])
            except Exception as e:  # noqa: BLE001  re-raised on the calling thread
                errors.append(e)
            finally:
                if al is not None:
                    al.close()

        threads = [threading.Thread(target=worker, args=(k,)) for k in range(nw)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
    _lap("pair estimates")
    # frame -> reference: the steps composed along the chain (x_ref = M_prev_total M_step x_frame)
    for ch in chains:
        total = np.eye(3)
        for i in ch:
            total = total @ step_m[i]
            transforms[i] = total.copy() if homography else total[:2].copy()
            ccs[i] = step_cc[i]
    if chain_refine and not homography:
        gref = None
        try:
            gref = reuse[1] if reuse else _lib.Aligner(height, width, dt, subsample=sub, device=device, fast=bool(cfg['fast_subsampling']))
            gref.set_reference(dev_frames + ref_idx * fb)
            todo = [i for ch in chains for i in ch[1:]]     # (a chain's first step IS an estimate against the global reference)
            for b0 in range(0, len(todo), _lib.Aligner.MAX_BATCH):
                idx = todo[b0:b0 + _lib.Aligner.MAX_BATCH]
                try:
                    m2, c2, _ = gref.refine_batch([dev_frames + i * fb for i in idx], np.stack([transforms[i] for i in idx]),
                                                  levels=2, max_iters=max_iters)
                except (_lib.DeviceError, ValueError):
                    continue
                for j, i in enumerate(idx):
                    if c2[j] >= min_correlation and _corner_shift(transforms[i], m2[j], height, width) <= CHAIN_REFINE_MAX_SHIFT:
                        transforms[i], ccs[i] = m2[j], float(c2[j])
        finally:
            if gref is not None and not reuse:
                gref.close()
    _lap("refine")
    tmp = mask = None
    try:
        tmp = reuse[2] if reuse else _lib.DeviceBuffer(fb, device)
        mask = reuse[3] if reuse else _lib.DeviceBuffer(height * width, device)
        warp = lib.mi_warp_perspective_device if homography else lib.mi_warp_affine_device
        for ch in chains:
            for i in ch:
                m = np.asarray(transforms[i], np.float64)
                arr = (C.c_double * m.size)(*m.reshape(-1))
                _lib.check(warp(device, None, dev_frames + i * fb, aligned + i * fb, tmp.ptr, mask.ptr, height, width,
                                _lib.DTYPE_CODE[dt], arr, mode, bv, 21, float(cfg['border_blur'])))
                if corr is not None:
                    corr.apply_correction_device(i, aligned + i * fb, None)
        _lib.check(lib.mi_device_synchronize(device))
        _lap("warps")
    finally:
        for b in (tmp, mask):
            if b is not None and not reuse:
                b.free()
    return transforms, ccs


def _align_chains_device(lib, dev_frames, aligned, n_frames, height, width, dt, ref_idx, cfg, min_correlation, max_iters,
                         device, corr=None, chain_refine=True):
    """`step_process=True` (stack_framework.py:214-232, the documented default of the reference's jobs): frame ref+1 is
    aligned to the reference frame, ref+2 to the ALIGNED ref+1, ... and ref-1, ref-2, ... the same way downwards -- two
    serial chains.  Every step needs the previous step's warped frame, so the batched estimator does not apply; the two
    chains are independent and run side by side (one host thread and one estimator handle each).  Aligned frames are
    written to `aligned` at their own index.  `corr` (a BalanceFrames correction, already begun on the reference frame):
    every aligned frame is balanced BEFORE it becomes the next step's reference, as the reference's CombinedActions does
    (the step reference is read back from the output directory, i.e. after align AND balance,
    stack_framework.py:259-262, :282-289).  Returns (transforms, correlation coefficients).

    `chain_refine` (ALIGN_RIGID): the errors of the chain's steps add up like a random walk (0.5 px at the ends of a
    128-frame stack, against the 0.2 px a single pair is held to, tests/test_0031_align_precision.py:62-65), because each
    step only sees the previous step's output.  Every step's estimate is therefore refined against the GLOBAL reference
    frame -- the same iteration on the two finest pyramid levels, started from the chain estimate -- before the frame is
    warped: the chain supplies the capture range (neighbouring frames look alike), the global frame the datum.  A
    refinement that fails, correlates worse than `min_correlation` or moves a corner by more than CHAIN_REFINE_MAX_SHIFT
    pixels is distrusted and the chain estimate is kept."""
    import threading
    corr_lock = threading.Lock()   # one correction object (its histogram / table scratch) serves both chains
    fb = height * width * 3 * dt.itemsize
    mode = _BORDER_CODE[cfg['border_mode']]
    bv = (C.c_double * 4)(*(list(cfg['border_value']) + [0, 0, 0, 0])[:4])
    homography = cfg['transform'] == constants.ALIGN_HOMOGRAPHY
    transforms, ccs, errors = [None] * n_frames, [1.0] * n_frames, []
    _lib.check(lib.mi_memcpy_d2d(device, aligned + ref_idx * fb, dev_frames + ref_idx * fb, fb))   # align.py:279-280

    def chain(indices):
        aligner = gref = tmp = mask = None
        try:
            aligner = _lib.Aligner(height, width, dt, subsample=max(1, int(cfg['subsample'])), device=device,
                                   fast=bool(cfg['fast_subsampling']))
            if chain_refine and not homography and indices:
                gref = _lib.Aligner(height, width, dt, subsample=max(1, int(cfg['subsample'])), device=device,
                                    fast=bool(cfg['fast_subsampling']))
                gref.set_reference(dev_frames + ref_idx * fb)
            tmp = _lib.DeviceBuffer(fb, device)
            mask = _lib.DeviceBuffer(height * width, device)
            prev = ref_idx
            for i in indices:
                aligner.set_reference(aligned + prev * fb)
                if homography:   # the similarity refined to 8 degrees of freedom
                    ms, cs, _ = aligner.estimate_homography_batch([dev_frames + i * fb], max_iters=max_iters)
                    m, cc = ms[0], cs[0]
                else:
                    m, cc, _ = aligner.estimate(dev_frames + i * fb, max_iters=max_iters)
                if not cc >= min_correlation:
                    raise AlignmentError(i, f"correlation {cc:.3f} < {min_correlation}")
                if gref is not None and prev != ref_idx:   # (the first step IS an estimate against the global reference)
                    try:
                        m2, c2, _ = gref.refine_batch([dev_frames + i * fb], m[None], levels=2, max_iters=max_iters)
                        if c2[0] >= min_correlation and _corner_shift(m, m2[0], height, width) <= CHAIN_REFINE_MAX_SHIFT:
                            m, cc = m2[0], c2[0]     # (the coefficient reported is that of the transform that is used)
                    except (_lib.DeviceError, ValueError):
                        pass
                transforms[i], ccs[i] = m, float(cc)
                arr = (C.c_double * m.size)(*m.reshape(-1))
                warp = lib.mi_warp_perspective_device if homography else lib.mi_warp_affine_device
                _lib.check(warp(device, None, dev_frames + i * fb, aligned + i * fb, tmp.ptr, mask.ptr, height, width,
                                _lib.DTYPE_CODE[dt], arr, mode, bv, 21, float(cfg['border_blur'])))
                if corr is not None:
                    with corr_lock:
                        corr.apply_correction_device(i, aligned + i * fb, None)
                _lib.check(lib.mi_device_synchronize(device))    # the next step's reference is this step's output
                prev = i
        except Exception as e:  # noqa: BLE001  re-raised on the calling thread
            errors.append(e)
        finally:   # a chain that raises must not leak its estimator handle and buffers
            for al in (aligner, gref):
                if al is not None:
                    al.close()
            for b in (tmp, mask):
                if b is not None:
                    b.free()

    threads = [threading.Thread(target=chain, args=(list(range(ref_idx + 1, n_frames)),)),
               threading.Thread(target=chain, args=(list(range(ref_idx - 1, -1, -1)),))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return transforms, ccs


class StackHandles:
    """What `align_and_stack_device(keep_handles=True)` returns and `handles=` takes back: the stacker, the estimator and
    the scratch buffers of ONE geometry.  The library never sees the sizes of `batches` / `tmp` / `mask`, so the geometry
    they were made for travels with them and a later call is checked against it (`matches`)."""

    def __init__(self, stack, aligner, batches, tmp, mask, gref=None, **geometry):
        self.stack, self.aligner, self.batches, self.tmp, self.mask = stack, aligner, batches, tmp, mask
        self.gref = gref    # step_process: the estimator that holds the GLOBAL reference frame (chain_refine)
        self.geometry = geometry

    def __iter__(self):    # (stack, aligner, batches, tmp, mask), as rounds 2-3 returned them
        return iter((self.stack, self.aligner, self.batches, self.tmp, self.mask))

    def mismatch(self, **geometry):
        """names of the geometry items that differ from what the handles were created for"""
        return [k for k, v in geometry.items() if self.geometry.get(k) != v]

    def close(self):
        self.aligner.close()
        if self.gref is not None:
            self.gref.close()
        self.stack.close()
        for b in (self.batches, self.tmp, self.mask):
            b.free()


def close_handles(handles):
    """release what `align_and_stack_device(keep_handles=True)` returned"""
    handles.close()


def _make_correction(balance, device):
    """The correction object of BalanceFrames for a dict of its options (balance.py:366-388: channel -> class; the
    sub-sampling default depends on the map)."""
    from .balance import LSCorrection, LumiCorrection, RGBCorrection, SVCorrection
    classes = {constants.BALANCE_LUMI: LumiCorrection, constants.BALANCE_RGB: RGBCorrection,
               constants.BALANCE_HSV: SVCorrection, constants.BALANCE_HLS: LSCorrection}
    opts = dict(balance)
    channel = opts.pop('channel', constants.DEFAULT_CHANNEL)
    if channel not in classes:
        raise InvalidOptionError("channel", channel)
    if opts.get('subsample', -1) == -1:
        opts['subsample'] = 1 if opts.get('corr_map') == constants.BALANCE_MATCH_HIST \
            else constants.DEFAULT_BALANCE_SUBSAMPLE
    return classes[channel](device=device, **opts)


def auto_batch_frames(n_frames, height, width, dtype, device=0, share=0.25):
    """Warped frames per push into the stacker for the resident align -> stack flow: as many as the job has (a multiple of 16,
    at most 128) while two input batches plus the stacker's two sets of coarser Gaussian levels stay inside `share` of the
    free device memory.  Bigger pushes fuse more efficiently (the coarse levels of a 16-frame push are launch-bound): config 4,
    128 x 24 MP u8, 0.0417 s at 16 frames per push, 0.0400 at 32, 0.0387 at 64, 0.0367 at 128."""
    dt = np.dtype(dtype)
    px = int(height) * int(width)
    per_frame = 2 * px * 3 * dt.itemsize + 2 * (px // 3 + 1) * 12   # two batch buffers + two sets of G_1.. (fp32 RGB, P_0 / 3 pixels)
    free, _total = _lib.mem_info(device)
    fit = int(share * free // max(per_frame, 1))
    want = min(128, -(-int(n_frames) // 16) * 16)
    return max(16, min(want, fit // 16 * 16))


def align_and_stack_device(dev_frames, n_frames, height, width, dtype, ref_idx=-1, alignment_config=None,
                           min_correlation=0.5, max_iters=60, device=0, batch_frames=None, out_dev=None,
                           balance=None, ecc_batch=16, step_process=False, native_loop=True, handles=None,
                           keep_handles=False, info=None, chain_refine=True, chain_serial=False, **stack_kwargs):
    """BASELINE config 4 with every frame resident in HBM: `dev_frames` is the device address of
    `n_frames` contiguous H x W x 3 frames.  Each frame is registered against frames[ref_idx] by
    the device ECC estimator (mi_aligner_*), warped with the blurred replicate border of
    align.py:238-251 straight into the stacker's input batch, and fused.  No frame crosses PCIe;
    per frame the host sees 28 doubles per Gauss-Newton iteration.

    The transforms are estimated `ecc_batch` (<= 16) frames at a time in one batched Gauss-Newton
    (mi_aligner_estimate_batch: one launch and one host round trip per iteration for the whole batch --
    a single frame's chain of small kernels leaves the GPU mostly idle), then the frames are warped and
    fused in order.  (Estimating on helper threads, one handle each or one batch ahead, measured no
    faster: the round trips serialise in the runtime.)

    `batch_frames`: warped frames per push into the stacker; None (default) = `auto_batch_frames`: the whole job in one
    push when the device memory allows it (the coarse pyramid levels of small pushes are launch-bound: config 4 takes 0.042 s
    at 16 frames per push, 0.037 s at 128); handles of an earlier call keep the value they were created with.

    `balance`: optional dict of BalanceFrames options (channel, corr_map, subsample, fast_subsampling,
    mask_size, intensity_interval): every aligned frame is then balanced against the reference frame
    (balance.py; the order of the reference's example projects: align, balance, stack) in place on the
    device -- histogram on the GPU, the 256/65536-entry table on the host, table apply on the GPU.

    `keep_handles` / `handles`: a job that fuses many stacks of one geometry keeps the stacker, the estimator and the
    scratch buffers between calls (`keep_handles=True` returns them as a fourth value, `handles=` takes them back; close them
    with `close_handles`): creating them costs 10-35 ms -- pinned host buffers, several GB of device buffers -- beside a
    128-frame job of 45-80 ms.

    `native_loop` (default): without balancing the frame loop runs inside the library (`mi_align_stack_device`); False
    keeps the call-by-call Python loop below (the two are tested equal).

    `info`: an optional dict; with `balance` it receives `info["corrections"]` = the per-frame correction factors the
    reference's sub-action records (balance.py: `self.corrections`), for the frames that were processed.

    `step_process=True`: the reference's chained order (see `_align_chains_device`): every frame is registered against
    its already-aligned neighbour; the aligned frames are kept in one extra device buffer (n_frames frames) and fused in
    file order afterwards, so that the stack sees them exactly as the reference's FocusStack reads the aligned files.
    `chain_refine` (default True; ALIGN_RIGID): every chain estimate is refined against the global reference frame before the
    frame is warped, so that the steps' errors do not add up (`_align_chains_device`); False = the plain chain.
    `chain_serial` (default False): the chain as rounds 3-5 ran it -- every step registered against the WARPED neighbour, one
    device synchronisation per frame (`_align_chains_device`); the default factors the chain into independent
    neighbour estimates + composition (`_align_chains_pairs_device`).

    Returns (fused image as ndarray, or None when `out_dev` -- a device address for the result --
    is given; list of 2x3 transforms, None at ref_idx; list of correlation coefficients)."""
    stack_kwargs["arith"] = resolve_arith(stack_kwargs.get("arith"), stack_kwargs.get("float_type"))   # one default for every entry point
    _lib.require_device()
    if n_frames < 1:
        raise ValueError("no frames")
    cfg = {**_DEFAULT_ALIGNMENT_CONFIG, **(alignment_config or {})}
    if cfg['border_mode'] not in _BORDER_CODE:
        raise InvalidOptionError("border_mode", cfg['border_mode'])
    if cfg['transform'] not in (constants.ALIGN_RIGID, constants.ALIGN_HOMOGRAPHY):
        raise InvalidOptionError("transform", cfg['transform'])
    # the device estimator finds a similarity; with ALIGN_HOMOGRAPHY it is applied through the projective warp
    homography = cfg['transform'] == constants.ALIGN_HOMOGRAPHY
    if ref_idx == -1:
        ref_idx = n_frames // 2
    dt = np.dtype(dtype)
    fb = height * width * 3 * dt.itemsize
    lib = _lib.load()
    if step_process:
        reusing = handles is not None or keep_handles
        if reusing and (chain_serial or homography):
            raise InvalidOptionError("handles", "reuse", ": with step_process, handle reuse is implemented for the factored ALIGN_RIGID "
                                     "chain (not chain_serial, not ALIGN_HOMOGRAPHY)")
        sub = max(1, int(cfg['subsample']))
        geometry = dict(height=int(height), width=int(width), dtype=dt.name, step_process=True, frames=int(n_frames), subsample=sub,
                        fast=bool(cfg['fast_subsampling']), device=int(device),
                        stack_kwargs=tuple(sorted((k, repr(v)) for k, v in stack_kwargs.items())))
        created = []
        if handles is not None:
            bad = handles.mismatch(**geometry)
            if bad:
                raise InvalidOptionError("handles", {k: geometry[k] for k in bad},
                                         f": these handles were created for {({k: handles.geometry.get(k) for k in bad})}")
            stack, aligner, aligned, tmp, mask = handles
            gref = handles.gref
            stack.reset()
        else:
            aligned = _lib.DeviceBuffer(fb * n_frames, device)
            created.append(aligned.free)
            aligner = gref = tmp = mask = None
            if reusing:
                aligner = _lib.Aligner(height, width, dt, subsample=sub, device=device, fast=bool(cfg['fast_subsampling']))
                gref = _lib.Aligner(height, width, dt, subsample=sub, device=device, fast=bool(cfg['fast_subsampling']))
                tmp, mask = _lib.DeviceBuffer(fb, device), _lib.DeviceBuffer(height * width, device)
                created += [aligner.close, gref.close, tmp.free, mask.free]
            stack = None
        done = False
        try:
            corr = None
            if balance is not None:
                corr = _make_correction(balance, device)
                corr.begin_device(dev_frames + ref_idx * fb, height, width, dt, n_frames)
            if chain_serial:
                transforms, ccs = _align_chains_device(lib, dev_frames, aligned.ptr, n_frames, height, width, dt, ref_idx, cfg,
                                                       min_correlation, max_iters, device, corr, chain_refine=chain_refine)
            else:
                transforms, ccs = _align_chains_pairs_device(lib, dev_frames, aligned.ptr, n_frames, height, width, dt, ref_idx, cfg,
                                                             min_correlation, max_iters, device, corr, chain_refine=chain_refine,
                                                             reuse=(aligner, gref, tmp, mask) if reusing else None)
            if stack is None:
                stack = _lib.Stack(height, width, in_dtype=dt, out_dtype=dt, device=device, **stack_kwargs)
                created.append(stack.close)
            stack.push_frames_device(aligned.ptr, n_frames, fb)
            if out_dev is not None:
                stack.finish_device(out_dev)
                stack.sync()
                out = None
            else:
                out = stack.finish()
            done = True
        finally:
            if not (keep_handles and done):
                for release in reversed(created):
                    release()
        if keep_handles:
            return out, transforms, ccs, (handles if handles is not None else
                                          StackHandles(stack, aligner, aligned, tmp, mask, gref=gref, **geometry))
        return out, transforms, ccs
    ecc_batch = max(1, min(int(ecc_batch), _lib.Aligner.MAX_BATCH))
    if batch_frames is None:   # handles of an earlier call fix it; else as many frames per push as memory allows
        batch_frames = handles.geometry["batch_frames"] if handles is not None else auto_batch_frames(n_frames, height, width, dt, device)
    geometry = dict(height=int(height), width=int(width), dtype=dt.name, batch_frames=int(batch_frames),
                    subsample=max(1, int(cfg['subsample'])), fast=bool(cfg['fast_subsampling']), device=int(device),
                    stack_kwargs=tuple(sorted((k, repr(v)) for k, v in stack_kwargs.items())))   # the stacker's own options
    # everything that can be refused is refused BEFORE anything is allocated
    corr = bal_opts = None
    if balance is not None:
        corr = _make_correction(balance, device)
    native = native_loop and (balance is None or corr.supports_native_linear())
    if (handles is not None or keep_handles) and not native:
        raise InvalidOptionError("handles", "reuse", ": handle reuse is implemented for the native loop (no balancing, or the "
                                 "LINEAR map)")
    if handles is not None:   # handles of an earlier call (a job of many stacks): nothing is allocated here
        bad = handles.mismatch(**geometry)
        if bad:
            raise InvalidOptionError("handles", {k: geometry[k] for k in bad},
                                     f": these handles were created for {({k: handles.geometry.get(k) for k in bad})}; the "
                                     "scratch buffers are sized for that geometry (close them and let the call allocate)")
        stack, aligner, batches, tmp, mask = handles
        stack.reset()
        created = None
    else:
        stack = _lib.Stack(height, width, in_dtype=dt, out_dtype=dt, device=device, batch_frames=batch_frames,
                           **stack_kwargs)
        aligner = _lib.Aligner(height, width, dt, subsample=max(1, int(cfg['subsample'])), device=device,
                               fast=bool(cfg['fast_subsampling']))
        created = [stack, aligner]     # released here whenever an exception leaves this call
    if native:
        # the whole loop below in ONE library call (mi_align_stack_device): same kernels in the same order on the same
        # streams; the ~25 ctypes calls per frame of the Python loop made the pipeline's pace depend on how busy the host is
        # (0.08 s on an idle box, 0.3 s on a shared one, for 128 x 24 MP)
        done = False
        try:
            if corr is not None:
                corr.begin_device(dev_frames + ref_idx * fb, height, width, dt, n_frames)
                bal_opts = corr.native_linear_opts()
            if handles is None:
                batches = _lib.DeviceBuffer(2 * fb * batch_frames, device)
                created.append(batches)
                tmp = _lib.DeviceBuffer(fb, device)
                created.append(tmp)
                mask = _lib.DeviceBuffer(height * width, device)
                created.append(mask)
            opts = _lib.AlignStackOpts(transform=int(homography), border_mode=_BORDER_CODE[cfg['border_mode']],
                                       border_value=(C.c_double * 4)(*(list(cfg['border_value']) + [0, 0, 0, 0])[:4]),
                                       blur_ksize=21, blur_sigma=float(cfg['border_blur']),
                                       min_correlation=float(min_correlation), max_iters=int(max_iters), eps=1e-9,
                                       ecc_batch=ecc_batch, batch_frames=int(batch_frames))
            M = (C.c_double * (9 * n_frames))()
            cc = (C.c_double * n_frames)()
            failed = C.c_int(-1)
            rc = lib.mi_align_stack_device(stack._h, aligner._h, dev_frames, n_frames, fb, ref_idx, C.byref(opts),
                                           C.byref(bal_opts) if bal_opts is not None else None, batches.ptr,
                                           tmp.ptr, mask.ptr, M, cc, C.byref(failed))
            if bal_opts is not None:
                # only the frames the library got to have a correction row on the device (it stops at the first
                # frame whose correlation is too low: those from `failed` on were never balanced)
                stop = failed.value if rc == _lib.MI_ERR_ALIGNMENT and failed.value >= 0 else n_frames
                corr._corr_pending.extend(i for i in range(stop) if i != ref_idx)
                if info is not None:
                    info["corrections"] = corr.fetch_corrections()
            if rc == _lib.MI_ERR_ALIGNMENT:
                raise AlignmentError(failed.value, f"correlation {cc[failed.value]:.3f} < {min_correlation}")
            _lib.check(rc)
            ms = np.array(M, dtype=np.float64).reshape(n_frames, 9)
            transforms = [None if i == ref_idx else (ms[i].reshape(3, 3).copy() if homography else ms[i, :6].reshape(2, 3).copy())
                          for i in range(n_frames)]
            ccs = [float(c) for c in cc]
            if out_dev is not None:
                stack.finish_device(out_dev)
                stack.sync()
                out = None
            else:
                out = stack.finish()
            done = True
        finally:
            if created is not None and not (keep_handles and done):
                for obj in created:
                    obj.close() if hasattr(obj, "close") else obj.free()
        if keep_handles:
            return out, transforms, ccs, (handles if handles is not None else
                                          StackHandles(stack, aligner, batches, tmp, mask, **geometry))
        return out, transforms, ccs
    try:
        if corr is not None:
            corr.begin_device(dev_frames + ref_idx * fb, height, width, dt, n_frames)
    except BaseException:
        for obj in created or ():
            obj.close()
        raise
    tmp = _lib.DeviceBuffer(fb, device)
    mask = _lib.DeviceBuffer(height * width, device)
    # two batches of warped frames: one is being fused while the next is being filled
    batches = [_lib.DeviceBuffer(fb * batch_frames, device) for _ in range(2)]
    mode = _BORDER_CODE[cfg['border_mode']]
    bv = (C.c_double * 4)(*(list(cfg['border_value']) + [0, 0, 0, 0])[:4])
    transforms, ccs = [], []
    cur, filled = 0, 0
    st = stack.stream   # warps, balance and the reference-frame copy run on the stacker's own stream
    aligner.set_reference(dev_frames + ref_idx * fb)
    estimates = {}

    def estimate_from(i):
        """transforms of the next `ecc_batch` moving frames starting at frame i"""
        idx = [k for k in range(i, n_frames) if k != ref_idx][:ecc_batch]
        fit = aligner.estimate_homography_batch if homography else aligner.estimate_batch
        ms, cs, _ = fit([dev_frames + k * fb for k in idx], max_iters=max_iters)
        for k, m, c in zip(idx, ms, cs):
            estimates[k] = (m, float(c))

    to_balance = []   # (frame index, device address) of the warped frames of the batch being filled

    def flush():
        nonlocal cur, filled
        if filled and corr is not None and to_balance:
            # the whole batch behind ONE host round trip (GAMMA / MATCH_HIST: the histograms come back together, SciPy
            # builds the tables, the table applies are enqueued; LINEAR: no round trip at all)
            corr.apply_correction_device_batch([i for i, _ in to_balance], [p for _, p in to_balance], st)
            to_balance.clear()
        if filled:
            # No host synchronisation here: the warps run on the stacker's stream, where the level-0 kernels that read
            # this batch are enqueued next, and the stacker joins its side streams into that stream after every push --
            # so the batch buffer written two flushes later is free by stream order.  The host goes straight on to the
            # next batch's estimate (its own stream), which then overlaps these warps and the fuse.
            stack.push_frames_device(batches[cur].ptr, filled, fb)
            cur ^= 1
            filled = 0

    try:
        for i in range(n_frames):
            src = dev_frames + i * fb
            dst = batches[cur].ptr + filled * fb
            if i == ref_idx:
                _lib.check(lib.mi_memcpy_d2d_async(device, st, dst, src, fb))   # align.py:279-280
                transforms.append(None)
                ccs.append(1.0)
            else:
                if i not in estimates:
                    estimate_from(i)
                m, cc = estimates.pop(i)
                if not cc >= min_correlation:
                    raise AlignmentError(i, f"correlation {cc:.3f} < {min_correlation}")
                mm = (C.c_double * m.size)(*m.reshape(-1))
                warp = lib.mi_warp_perspective_device if homography else lib.mi_warp_affine_device
                _lib.check(warp(device, st, src, dst, tmp.ptr, mask.ptr, height, width, _lib.DTYPE_CODE[dt], mm, mode, bv, 21,
                                float(cfg['border_blur'])))
                if corr is not None:
                    to_balance.append((i, dst))
                transforms.append(m)
                ccs.append(cc)
            filled += 1
            if filled == batch_frames:
                flush()
        flush()
        if out_dev is not None:
            stack.finish_device(out_dev)
            stack.sync()
            out = None
        else:
            out = stack.finish()
        if corr is not None and info is not None:
            info["corrections"] = corr.fetch_corrections()
    finally:
        aligner.close()
        stack.close()
        for b in [tmp, mask] + batches:
            b.free()
    return out, transforms, ccs


def bunches_then_stack(get_frame, n_frames, height, width, dtype, frames=constants.DEFAULT_FRAMES,
                       overlap=constants.DEFAULT_OVERLAP, device=0, out_dev=None, on_bunch=None, on_final=None,
                       check_running=None, stacks=None, results_buf=None, info=None, zero_copy=False, **stack_kwargs):
    """BASELINE config 5's two-stage flow in memory (the reference's `FocusStackBunch` followed by `FocusStack`,
    stack.py:61-113, examples/stack-from-frames): the frames are fused in bunches of `frames` with `overlap` shared
    (`get_bunches`), every bunch result is the stacker's OUTPUT type -- truncated to the input dtype exactly as the file
    the reference writes between the two actions (stack.py:33-36, pyramid.py:179) -- and the bunch results are fused once
    more.  Stage 1 takes the frames from HOST memory through the pinned asynchronous upload of `mi_stack_push_frame`
    (`get_frame(i) -> H x W x 3 array`; decode / generate, PCIe and kernels overlap); the bunch results never leave the
    device: they are written side by side into one buffer and stage 2 consumes them in place.

    The reference fuses the bunches in the order `chunks[count - 1]` (stack.py:97: the last one first); every bunch is an
    independent stack and the second action reads the results sorted by the name of each bunch's first frame, i.e. in
    bunch order -- which is the order used here.

    `on_bunch(k, stack)` / `on_final(stack, results_buffer)`: called after a bunch's / the final stack's frames were pushed and before it is
    finished (tests tap the selection state there).  `stacks`: optional `_lib.Stack` handles to reuse, the LAST one for stage 2
    and the others (one, or two that alternate so that uploads and kernels of neighbouring bunches overlap) for stage 1 -- a handle owns pinned upload buffers and gigabytes of device buffers whose allocation costs more than a
    short job; they are reset, not closed.  `results_buf`: an optional `_lib.DeviceBuffer` of at least n_bunches frames for the
    bunch results (`hipMalloc` / `hipFree` of the 38 GB a 1024-frame job needs cost ~2 s: a caller that runs job after job
    keeps it); it is not freed here.  `info`: an optional dict that receives `stage1_s` (first push to the last bunch result
    on the device) and `stage2_s` (the stack over the bunch results, to its result).  Returns the fused image (or None when `out_dev` is given) and the list of
    bunches (frame indices)."""
    from .actions import get_bunches
    stack_kwargs["arith"] = resolve_arith(stack_kwargs.get("arith"), stack_kwargs.get("float_type"))   # one default for every entry point
    _lib.require_device()
    if overlap >= frames:
        raise InvalidOptionError("overlap", overlap, "overlap must be smaller than batch size")
    dt = np.dtype(dtype)
    fb = height * width * 3 * dt.itemsize
    bunches = get_bunches(list(range(n_frames)), frames, overlap)
    if not bunches:
        raise ValueError("no frames")
    if stacks is not None and len(stacks) < 2:
        raise InvalidOptionError("stacks", len(stacks), ": at least two handles are needed (the last one is stage 2, the others stage 1)")
    own_results = results_buf is None
    if not own_results and results_buf.nbytes < fb * len(bunches):
        raise InvalidOptionError("results_buf", results_buf.nbytes, f": {len(bunches)} bunch results need {fb * len(bunches)} bytes")
    results = _lib.DeviceBuffer(fb * len(bunches), device) if own_results else results_buf
    t_start = time.perf_counter()
    # Stage 1 alternates between TWO handles: resetting a handle waits for ITS previous bunch only, so the uploads of bunch
    # k + 1 (handle B's copy stream) run while bunch k is still being fused (handle A) -- with one handle the PCIe link
    # idled through every bunch's kernels and collapse (measured: 38 GB/s of a 57 GB/s link; config 5 is upload-bound).
    if stacks:
        stage1 = list(stacks[:-1])
    else:
        stage1 = [_lib.Stack(height, width, in_dtype=dt, out_dtype=dt, device=device, **stack_kwargs)
                  for _ in range(2 if len(bunches) > 1 else 1)]
    try:
        for k, bunch in enumerate(bunches):
            st = stage1[k % len(stage1)]
            st.reset()      # a handle serves every other bunch, as one stacker object serves FocusStackBunch (stack.py:94-97)
            for i in bunch:
                st.push_frame(get_frame(i), zero_copy=zero_copy)
                if check_running is not None and check_running() is False:
                    from .errors import RunStopException
                    raise RunStopException("bunches_then_stack")
            if on_bunch is not None:
                on_bunch(k, st)
            st.finish_device(results.ptr + k * fb)
        for st in stage1:
            st.sync()
    except BaseException:
        for st in stage1:
            try:                # the cleanup must not mask the exception that brought us here
                if stacks:
                    st.sync()   # a reused handle may still be writing into `results`
                else:
                    st.close()
            except Exception:   # noqa: BLE001
                pass
        if own_results:
            results.free()
        raise
    if not stacks:
        for st in stage1:
            st.close()
    st2 = stacks[-1] if stacks else _lib.Stack(height, width, in_dtype=dt, out_dtype=dt, device=device, **stack_kwargs)
    t_stage1 = time.perf_counter()
    try:
        st2.reset()
        st2.push_frames_device(results.ptr, len(bunches), fb)
        if on_final is not None:
            on_final(st2, results)
        if out_dev is not None:
            st2.finish_device(out_dev)
            st2.sync()
            out = None
        else:
            out = st2.finish()
    finally:
        if not stacks:
            st2.close()
        else:
            st2.sync()      # `results` is freed below
        if own_results:
            results.free()
    if info is not None:
        info.update(stage1_s=t_stage1 - t_start, stage2_s=time.perf_counter() - t_stage1)
    return out, bunches
