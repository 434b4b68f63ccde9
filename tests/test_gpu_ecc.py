"""GPU: the ECC similarity estimator (mi_ecc_similarity).  No reference output exists for it (the
reference's estimator is OpenCV's SIFT+RANSAC); it is validated against ground-truth transforms
with the acceptance tolerances of the reference's own precision test
(tests/test_0031_align_precision.py:62-65): angle < 0.005 deg, shift < 0.2 px, scale < 1e-4."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L(hiplib):
    hiplib.require_device()
    return hiplib


def texture(h, w, seed):
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w), np.float64)
    for sigma, amp in ((1.5, 60), (4, 50), (12, 40)):
        img += amp * ndimage.gaussian_filter(rng.standard_normal((h, w)), sigma) * sigma
    img = 128 + img * (60 / img.std())
    # a few hard-edged shapes, as in the reference's synthetic test image
    yy, xx = np.mgrid[0:h, 0:w]
    img[(yy - h * 0.3) ** 2 + (xx - w * 0.6) ** 2 < (0.08 * h) ** 2] += 70
    img[int(h * 0.55):int(h * 0.8), int(w * 0.15):int(w * 0.4)] -= 60
    return np.clip(img, 0, 255)


def similarity(theta_deg, s, tx, ty, cx, cy):
    t = np.deg2rad(theta_deg)
    a, b = s * np.cos(t), s * np.sin(t)
    return np.array([[a, -b, cx - a * cx + b * cy + tx], [b, a, cy - b * cx - a * cy + ty]])


def invert(M):
    A = M[:, :2]
    Ai = np.linalg.inv(A)
    return np.hstack([Ai, -Ai @ M[:, 2:3]])


def decompose(M):
    s = np.hypot(M[0, 0], M[1, 0])
    return np.rad2deg(np.arctan2(M[1, 0], M[0, 0])), s, M[0, 2], M[1, 2]


def make_pair(oracle, T, h=512, w=512, noise=5.0, seed=0, dtype=np.uint8):
    rng = np.random.default_rng(seed + 100)
    base = texture(h, w, seed)
    scale = 1 if dtype == np.uint8 else 257
    ref3 = np.repeat(base[:, :, None], 3, 2)
    mov3 = oracle.warp_affine(np.clip(ref3, 0, 255).astype(np.uint8), T, border_mode=oracle.BORDER_REPLICATE)
    ref = np.clip(ref3 + rng.normal(0, noise, ref3.shape), 0, 255)
    mov = np.clip(mov3.astype(np.float64) + rng.normal(0, noise, ref3.shape), 0, 255)
    return (ref * scale).astype(dtype), (mov * scale).astype(dtype)


@pytest.mark.parametrize("theta,s,tx,ty,dtype", [
    (0.5, 1.003, 7.3, -4.6, np.uint8),
    (-0.8, 0.994, -12.4, 9.7, np.uint8),
    (0.02, 1.0001, 0.37, -0.21, np.uint16),      # one step of the config-4 sequence
    (1.28, 1.0064, 23.7, -13.4, np.uint8),       # the far end of the config-4 sequence
])
def test_recovers_known_similarity(L, oracle, theta, s, tx, ty, dtype):
    T = similarity(theta, s, tx, ty, 255.5, 255.5)       # moving = warp(ref, T)
    ref, mov = make_pair(oracle, T, dtype=dtype)
    M, cc, iters = L.ecc_similarity(ref, mov)
    want = invert(T)                                      # moving -> reference
    a, sc, mx, my = decompose(M)
    a0, s0, x0, y0 = decompose(want)
    assert cc > 0.9
    assert abs(a - a0) < 0.005, (a, a0)
    assert abs(sc - s0) < 1e-4, (sc, s0)
    # compare the mapping of the image centre (a translation-like quantity independent of the
    # rotation pivot convention)
    c = np.array([255.5, 255.5, 1.0])
    assert np.abs(M @ c - want @ c).max() < 0.2


def test_large_motion_of_the_reference_precision_test(L, oracle):
    """15 deg rotation + (30, 20) px shift (tests/test_0031_align_precision.py:44-47)."""
    T = similarity(15.0, 1.0, 30.0, 20.0, 255.5, 255.5)
    ref, mov = make_pair(oracle, T, noise=10.0)
    M, cc, iters = L.ecc_similarity(ref, mov, max_iters=150)
    want = invert(T)
    a, sc, _, _ = decompose(M)
    a0, s0, _, _ = decompose(want)
    c = np.array([255.5, 255.5, 1.0])
    assert cc > 0.8
    assert abs(a - a0) < 0.005 and abs(sc - s0) < 1e-4
    assert np.abs(M @ c - want @ c).max() < 0.2


def test_align_images_with_the_gpu_estimator(L, oracle):
    """End to end: align_images(estimator=ecc_estimator()) brings the moving frame back onto the
    reference (config 4's building block: estimate + warp + border blur, all on the GPU)."""
    from shinestacker_amd import align_images
    from shinestacker_amd.align import ecc_estimator
    T = similarity(0.6, 1.002, 9.1, -6.3, 255.5, 255.5)
    ref, mov = make_pair(oracle, T, noise=0.0)
    n, M, warp = align_images(ref, mov, estimator=ecc_estimator(), alignment_config={'fast_subsampling': True})
    assert n == 1000 and warp.shape == ref.shape
    inner = (slice(40, -40), slice(40, -40))
    err = np.abs(warp[inner].astype(np.int32) - ref[inner].astype(np.int32))
    before = np.abs(mov[inner].astype(np.int32) - ref[inner].astype(np.int32)).mean()
    # what remains is the double bilinear interpolation of a sharp texture, not misalignment
    assert err.mean() < 4.0 and before > 5 * err.mean(), (err.mean(), before)


def test_non_overlapping_or_flat_images_fail_cleanly(L):
    flat = np.full((128, 128, 3), 7, np.uint8)
    with pytest.raises(L.DeviceError):
        L.ecc_similarity(flat, flat)


def test_device_resident_aligner_matches_host_entry(L, oracle):
    """mi_aligner_* (frames in HBM, sub-sampling folded into the first kernel) returns what
    mi_ecc_similarity returns on img[::2, ::2] with the translation scaled back (align.py:224-231)."""
    T = similarity(0.3, 1.002, 9.0, -6.0, 383.5, 255.5)
    ref, mov = make_pair(oracle, T, h=512, w=768, seed=3)
    m_host, cc_host, it_host = L.ecc_similarity(ref[::2, ::2], mov[::2, ::2])
    m_host[:, 2] *= 2
    buf = L.DeviceBuffer(2 * ref.nbytes)
    buf.upload(ref)
    buf.upload(mov, ref.nbytes)
    al = L.Aligner(512, 768, np.uint8, subsample=2)
    al.set_reference(buf.ptr)
    m_dev, cc_dev, it_dev = al.estimate(buf.ptr + ref.nbytes)
    # a second estimate on the same handle must reproduce it (no state carried over)
    m_dev2, _, _ = al.estimate(buf.ptr + ref.nbytes)
    al.close()
    assert it_dev == it_host
    np.testing.assert_allclose(m_dev, m_host, rtol=0, atol=1e-7)
    np.testing.assert_allclose(m_dev2, m_dev, rtol=0, atol=1e-7)
    assert abs(cc_dev - cc_host) < 1e-9


def test_align_and_stack_device_equals_host_pipeline(L, oracle):
    """Frames resident in HBM -> same fused picture as the host-array pipeline with the GPU estimator."""
    from shinestacker_amd.align import ecc_estimator
    from shinestacker_amd.pipeline import align_and_stack, align_and_stack_device
    h, w, n = 384, 512, 5
    frames = []
    for f in range(n):
        d = f - n // 2
        T = similarity(0.1 * d, 1 + 5e-4 * d, 1.7 * d, -1.1 * d, (w - 1) / 2, (h - 1) / 2)
        ref, mov = make_pair(oracle, T, h=h, w=w, seed=11, noise=3.0)
        frames.append(ref if d == 0 else mov)
    cfg = {'fast_subsampling': True, 'subsample': 2}
    fused_host, _ = align_and_stack(frames, estimator=ecc_estimator(), alignment_config=cfg, batch_frames=2)
    buf = L.DeviceBuffer(n * frames[0].nbytes)
    for f, fr in enumerate(frames):
        buf.upload(fr, f * fr.nbytes)
    fused_dev, transforms, ccs = align_and_stack_device(buf.ptr, n, h, w, np.uint8, alignment_config=cfg,
                                                        batch_frames=2)
    assert transforms[n // 2] is None and all(c > 0.9 for c in ccs)
    # the host pipeline rounds M to float32 as the reference does (align.py:224-231); the device one keeps
    # doubles: sub-1e-6 differences in M can flip single fixed-point roundings of the warp
    diff = np.abs(fused_host.astype(np.int32) - fused_dev.astype(np.int32))
    assert (diff > 0).mean() < 0.02 and diff.max() <= 8, ((diff > 0).mean(), diff.max())
    # a device-side result buffer gives the same bytes as the host copy
    out = L.DeviceBuffer(frames[0].nbytes)
    none, _, _ = align_and_stack_device(buf.ptr, n, h, w, np.uint8, alignment_config=cfg, batch_frames=2,
                                        out_dev=out.ptr)
    assert none is None
    np.testing.assert_array_equal(out.download((h, w, 3), np.uint8), fused_dev)


@pytest.mark.parametrize("transform", ["ALIGN_RIGID", "ALIGN_HOMOGRAPHY"])
def test_native_align_stack_loop_equals_the_python_loop(L, oracle, transform):
    """mi_align_stack_device (the frame loop inside the library) == the call-by-call loop of pipeline.py: same transforms,
    same correlation coefficients, same fused image bit for bit; batches that do not divide the frame count, the reference
    frame in the middle of a batch; a frame that cannot be registered raises AlignmentError with its index."""
    from shinestacker_amd.errors import AlignmentError
    from shinestacker_amd.pipeline import align_and_stack_device
    h, w, n = 384, 512, 9
    frames = []
    for f in range(n):
        d = f - n // 2
        T = similarity(0.1 * d, 1 + 5e-4 * d, 1.7 * d, -1.1 * d, (w - 1) / 2, (h - 1) / 2)
        ref, mov = make_pair(oracle, T, h=h, w=w, seed=19, noise=3.0)
        frames.append(ref if d == 0 else mov)
    fb = frames[0].nbytes
    buf = L.DeviceBuffer(n * fb)
    for f, fr in enumerate(frames):
        buf.upload(fr, f * fb)
    cfg = {'subsample': 2, 'fast_subsampling': True, 'transform': transform}
    res = {}
    for native in (True, False):
        res[native] = align_and_stack_device(buf.ptr, n, h, w, np.uint8, alignment_config=cfg, batch_frames=4, ecc_batch=3,
                                             native_loop=native)
    (fa, ta, ca), (fb_, tb, cb) = res[True], res[False]
    assert ta[n // 2] is None and tb[n // 2] is None
    for i in range(n):
        if i != n // 2:
            assert ta[i].shape == ((3, 3) if transform == "ALIGN_HOMOGRAPHY" else (2, 3))
            assert np.array_equal(ta[i], tb[i]), i
    assert ca == cb and np.array_equal(fa, fb_)
    # a flat frame cannot be registered: the loop stops there and names it
    buf.upload(np.full_like(frames[0], 90), 6 * fb)
    with pytest.raises(AlignmentError) as e:
        align_and_stack_device(buf.ptr, n, h, w, np.uint8, alignment_config=cfg, batch_frames=4)
    assert "6" in str(e.value)
    buf.free()


def test_batched_estimate_equals_single_estimates(L, oracle):
    """mi_aligner_estimate_batch: every frame gets exactly what a single estimate gives (the per-frame
    sums are reduced in the same order), including a frame the method fails on."""
    h, w = 384, 512
    frames, truth = [], []
    for k, (th, s, tx, ty) in enumerate([(0.2, 1.001, 3.0, -2.0), (-0.4, 0.998, -5.5, 4.25), (0.0, 1.0, 0.0, 0.0),
                                          (1.0, 1.004, 8.0, 6.0)]):
        T = similarity(th, s, tx, ty, (w - 1) / 2, (h - 1) / 2)
        ref, mov = make_pair(oracle, T, h=h, w=w, seed=9)
        frames.append(mov)
    flat = np.full_like(frames[0], 77)
    frames.insert(2, flat)
    buf = L.DeviceBuffer((len(frames) + 1) * ref.nbytes)
    buf.upload(ref)
    for k, fr in enumerate(frames):
        buf.upload(fr, (k + 1) * ref.nbytes)
    al = L.Aligner(h, w, np.uint8, subsample=1)
    al.set_reference(buf.ptr)
    ptrs = [buf.ptr + (k + 1) * ref.nbytes for k in range(len(frames))]
    ms, ccs, its = al.estimate_batch(ptrs)
    assert ccs[2] == -2.0 and np.array_equal(ms[2], [[1, 0, 0], [0, 1, 0]])
    for k in (0, 1, 3, 4):
        m1, cc1, it1 = al.estimate(ptrs[k])
        assert np.array_equal(ms[k], m1) and ccs[k] == cc1 and its[k] == it1, k
    with pytest.raises(Exception):
        al.estimate(ptrs[2])
    al.close()


@pytest.mark.parametrize("refine,n", [(False, 7), (True, 7), (True, 15)])
def test_align_and_stack_device_step_process_chains(L, oracle, refine, n):
    """step_process=True (the chained order of stack_framework.py:214-232; the class default is False, :192): every frame is registered
    against its already ALIGNED neighbour, in two chains away from the reference frame.  The resident pipeline must give
    what the same procedure gives step by step through the single-frame entry points, fused in file order.  With
    `chain_refine` (the default) every step's estimate is refined against the global reference frame, and every frame --
    whatever its place in the chain -- lands within the 0.2 px a single pair is held to
    (tests/test_0031_align_precision.py:62-65); the plain chain's errors add up and are only held to that per step."""
    from shinestacker_amd.pipeline import align_and_stack_device, CHAIN_REFINE_MAX_SHIFT, _corner_shift
    h, w = 384, 512
    ref_idx = n // 2
    frames, truth = [], []
    for f in range(n):
        d = f - ref_idx
        T = similarity(0.12 * d, 1 + 4e-4 * d, 1.4 * d, -0.9 * d, (w - 1) / 2, (h - 1) / 2)
        ref, mov = make_pair(oracle, T, h=h, w=w, seed=13, noise=2.0)
        frames.append(ref if d == 0 else mov)
        truth.append(T)
    fb = frames[0].nbytes
    buf = L.DeviceBuffer(n * fb)
    for f, fr in enumerate(frames):
        buf.upload(fr, f * fb)
    cfg = {'subsample': 1}
    fused, tr, ccs = align_and_stack_device(buf.ptr, n, h, w, np.uint8, alignment_config=cfg, step_process=True,
                                            chain_refine=refine, chain_serial=True)
    assert tr[ref_idx] is None and all(c > 0.9 for c in ccs)
    # the same chains, one step at a time: estimate against the previous ALIGNED frame, (refine against the global
    # reference frame,) warp, remember
    aligned = {ref_idx: frames[ref_idx]}
    gref = L.Aligner(h, w, np.uint8, subsample=1)
    gref.set_reference(buf.ptr + ref_idx * fb)
    for chain in (range(ref_idx + 1, n), range(ref_idx - 1, -1, -1)):
        prev = ref_idx
        for i in chain:
            m, cc, _ = L.ecc_similarity(aligned[prev], frames[i])
            if refine and prev != ref_idx:
                m2, c2, _ = gref.refine_batch([buf.ptr + i * fb], m[None], levels=2)
                assert c2[0] > 0.9 and _corner_shift(m, m2[0], h, w) <= CHAIN_REFINE_MAX_SHIFT, i
                m = m2[0]
            assert np.allclose(m, tr[i], rtol=0, atol=1e-9), i
            aligned[i] = L.warp_affine(frames[i], tr[i])
            prev = i
    gref.close()
    so = oracle.StreamingOracle(h, w, np.uint8, keep_gauss=False, arith="separable")   # the entry points' default
    for i in range(n):
        so.push_frame(aligned[i])
    assert np.array_equal(fused, so.finish())
    # accuracy against the known transforms (tests/test_0031_align_precision.py:62-65: 0.2 px); the plain chain is held to
    # that per step -- its errors add up --, the refined chain as a whole
    cx, cy = (w - 1) / 2, (h - 1) / 2
    for i in range(n):
        if i == ref_idx:
            continue
        A = np.array(truth[i])[:, :2]
        Ai = np.linalg.inv(A)
        want = np.hstack([Ai, -Ai @ np.array(truth[i])[:, 2:3]])
        ctr = np.array([cx, cy, 1.0])
        assert np.abs(tr[i] @ ctr - want @ ctr).max() < (0.2 if refine else 0.2 * abs(i - ref_idx)), i


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
@pytest.mark.parametrize("size", [(384, 512), (387, 509), (130, 1031)])
def test_one_pass_pyramid_equals_the_separate_kernels(L, oracle, dtype, size):
    """Sub-sample 2 builds gray, level 0 and level 1 in one pass over the frame (ecc_pyramid2); the estimate on the strided
    image itself (sub-sample 1: ecc_gray + ecc_blur_tile<0> + <1>) must see the same pyramids: same transform to 1e-9."""
    h, w = size
    T = similarity(0.25, 1.001, 5.0, -3.0, (w - 1) / 2, (h - 1) / 2)
    ref, mov = make_pair(oracle, T, h=h, w=w, seed=33, noise=2.0)
    if dtype == np.uint16:
        ref, mov = ref.astype(np.uint16) * 257, mov.astype(np.uint16) * 257
    buf = L.DeviceBuffer(2 * ref.nbytes)
    buf.upload(ref)
    buf.upload(mov, ref.nbytes)
    al = L.Aligner(h, w, dtype, subsample=2, fast=True)
    al.set_reference(buf.ptr)
    m_dev, cc_dev, _ = al.estimate(buf.ptr + ref.nbytes)
    al.close()
    m_host, cc_host, _ = L.ecc_similarity(np.ascontiguousarray(ref[::2, ::2]), np.ascontiguousarray(mov[::2, ::2]))
    m_host = m_host.copy()
    m_host[:, 2] *= 2
    assert np.allclose(m_dev, m_host, rtol=0, atol=1e-9) and abs(cc_dev - cc_host) < 1e-12


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
@pytest.mark.parametrize("s", [2, 4, 8])
@pytest.mark.parametrize("size", ["divisible", "ragged"])
def test_device_area_subsampling_equals_host_resize(L, oracle, dtype, s, size):
    """fast_subsampling=False (the reference's default and the setting of its example projects, utils.py:79-86:
    cv2.resize(INTER_AREA) by an integer factor) folded into the device estimator's first kernel == the host-side
    img_subsample followed by the estimate on the small images (translation rescaled as align.py:223 does)."""
    from shinestacker_amd.align import img_subsample
    h, w = (384, 512) if size == "divisible" else (387, 509)   # 387 / 4 = 96.75 -> 97 rows = ceil; 509 / 4 = 127.25 -> 127 < ceil
    T = similarity(0.3, 1.002, 6.0, -4.0, (w - 1) / 2, (h - 1) / 2)
    ref, mov = make_pair(oracle, T, h=h, w=w, seed=21, noise=2.0)
    if dtype == np.uint16:
        ref, mov = ref.astype(np.uint16) * 257, mov.astype(np.uint16) * 257
    buf = L.DeviceBuffer(2 * ref.nbytes)
    buf.upload(ref)
    buf.upload(mov, ref.nbytes)
    al = L.Aligner(h, w, dtype, subsample=s, fast=False)
    al.set_reference(buf.ptr)
    m_dev, cc_dev, _ = al.estimate(buf.ptr + ref.nbytes)
    al.close()
    m_host, cc_host, _ = L.ecc_similarity(img_subsample(ref, s, fast=False), img_subsample(mov, s, fast=False))
    m_host = m_host.copy()
    m_host[:, 2] *= s
    assert np.allclose(m_dev, m_host, rtol=0, atol=1e-9) and abs(cc_dev - cc_host) < 1e-12
    # and it is not the strided sub-sampling
    al = L.Aligner(h, w, dtype, subsample=s, fast=True)
    al.set_reference(buf.ptr)
    m_fast, _, _ = al.estimate(buf.ptr + ref.nbytes)
    al.close()
    assert not np.array_equal(m_fast, m_dev)


def test_step_process_chains_balance_before_the_next_reference(L, oracle):
    """step_process with BalanceFrames: the reference's CombinedActions reads the step reference back from the output
    directory, i.e. AFTER align and balance (stack_framework.py:259-262, :282-289).  The resident pipeline must register
    frame i against the aligned AND balanced frame i-1 -- checked against the same steps through the single-frame entry
    points and the host form of the correction classes."""
    from shinestacker_amd import constants
    from shinestacker_amd.pipeline import _make_correction, align_and_stack_device
    h, w, n = 256, 384, 5
    ref_idx = n // 2
    frames = []
    for f in range(n):
        d = f - ref_idx
        T = similarity(0.1 * d, 1 + 3e-4 * d, 1.2 * d, -0.7 * d, (w - 1) / 2, (h - 1) / 2)
        ref, mov = make_pair(oracle, T, h=h, w=w, seed=21, noise=2.0)
        img = ref if d == 0 else mov
        # a per-frame exposure change for the balance to undo
        frames.append(np.clip(img.astype(np.float32) * (1.0 + 0.04 * d), 0, 255).astype(np.uint8))
    fb = frames[0].nbytes
    buf = L.DeviceBuffer(n * fb)
    for f, fr in enumerate(frames):
        buf.upload(fr, f * fb)
    bal = dict(channel=constants.BALANCE_LUMI, corr_map=constants.BALANCE_LINEAR, subsample=1)
    fused, tr, _ = align_and_stack_device(buf.ptr, n, h, w, np.uint8, alignment_config={'subsample': 1}, step_process=True,
                                          balance=bal, chain_refine=False, chain_serial=True)   # (the plain serial chain: this test is about the balance step)
    corr = _make_correction(bal, 0)
    corr.begin(frames[ref_idx], n, ref_idx)
    out = {ref_idx: frames[ref_idx]}
    for chain in (range(ref_idx + 1, n), range(ref_idx - 1, -1, -1)):
        prev = ref_idx
        for i in chain:
            m, _, _ = L.ecc_similarity(out[prev], frames[i])
            assert np.allclose(m, tr[i], rtol=0, atol=1e-9), i
            out[i] = corr.apply_correction(i, L.warp_affine(frames[i], tr[i]))
            prev = i
    so = oracle.StreamingOracle(h, w, np.uint8, keep_gauss=False, arith="separable")   # the entry points' default
    for i in range(n):
        so.push_frame(out[i])
    assert np.array_equal(fused, so.finish())
    buf.free()


@pytest.mark.parametrize("refine", [False, True])
@pytest.mark.parametrize("balance", [False, True])
def test_step_process_as_neighbour_pairs(L, oracle, refine, balance):
    """step_process=True as the pipeline runs it by default (round 6, `_align_chains_pairs_device`): every frame registered
    against its UNWARPED neighbour -- independent estimates --, the steps composed along the two chains in float64, the
    composed estimates refined against the global reference frame (`chain_refine`), then all frames warped (and balanced).
    Checked: (1) the transforms are the composition of the single-pair estimates (through the single-frame entry point);
    (2) every frame lands within the 0.2 px of tests/test_0031_align_precision.py:62-65 when refined -- the plain chain's
    errors may add up, 0.2 px per step; (3) the fused image is the stack of the frames warped by the returned transforms
    (and balanced) -- file order, as the reference's FocusStack reads the aligned files."""
    from shinestacker_amd import constants
    from shinestacker_amd.pipeline import _make_correction, _to33, align_and_stack_device, CHAIN_REFINE_MAX_SHIFT, _corner_shift
    h, w, n = 384, 512, 9
    ref_idx = n // 2
    frames, truth = [], []
    for f in range(n):
        d = f - ref_idx
        T = similarity(0.12 * d, 1 + 4e-4 * d, 1.4 * d, -0.9 * d, (w - 1) / 2, (h - 1) / 2)
        ref, mov = make_pair(oracle, T, h=h, w=w, seed=29, noise=2.0)
        img = ref if d == 0 else mov
        if balance:
            img = np.clip(img.astype(np.float32) * (1.0 + 0.03 * d), 0, 255).astype(np.uint8)
        frames.append(img)
        truth.append(T)
    fb = frames[0].nbytes
    buf = L.DeviceBuffer(n * fb)
    for f, fr in enumerate(frames):
        buf.upload(fr, f * fb)
    bal = dict(channel=constants.BALANCE_LUMI, corr_map=constants.BALANCE_LINEAR, subsample=1) if balance else None
    fused, tr, ccs = align_and_stack_device(buf.ptr, n, h, w, np.uint8, alignment_config={'subsample': 1}, step_process=True,
                                            chain_refine=refine, balance=bal)
    assert tr[ref_idx] is None and all(c > 0.9 for c in ccs)
    gref = L.Aligner(h, w, np.uint8, subsample=1)
    gref.set_reference(buf.ptr + ref_idx * fb)
    for chain in (range(ref_idx + 1, n), range(ref_idx - 1, -1, -1)):
        prev, total = ref_idx, np.eye(3)
        for i in chain:
            m, cc, _ = L.ecc_similarity(frames[prev], frames[i])       # the UNWARPED neighbour is the step's reference
            total = total @ _to33(m)
            want = total[:2]
            if refine and prev != ref_idx:
                m2, c2, _ = gref.refine_batch([buf.ptr + i * fb], want[None], levels=2)
                assert c2[0] > 0.9 and _corner_shift(want, m2[0], h, w) <= CHAIN_REFINE_MAX_SHIFT, i
                want = m2[0]
            assert np.allclose(want, tr[i], rtol=0, atol=1e-9), i
            prev = i
    gref.close()
    cx, cy = (w - 1) / 2, (h - 1) / 2
    for i in range(n):
        if i == ref_idx:
            continue
        A = np.array(truth[i])[:, :2]
        Ai = np.linalg.inv(A)
        want = np.hstack([Ai, -Ai @ np.array(truth[i])[:, 2:3]])
        ctr = np.array([cx, cy, 1.0])
        assert np.abs(tr[i] @ ctr - want @ ctr).max() < (0.2 if refine else 0.2 * abs(i - ref_idx)), i
    corr = None
    if balance:
        corr = _make_correction(bal, 0)
        corr.begin(frames[ref_idx], n, ref_idx)
    so = oracle.StreamingOracle(h, w, np.uint8, keep_gauss=False, arith="separable")
    for i in range(n):
        img = frames[i] if i == ref_idx else L.warp_affine(frames[i], tr[i])
        so.push_frame(corr.apply_correction(i, img) if corr is not None and i != ref_idx else img)
    assert np.array_equal(fused, so.finish())
    buf.free()


def test_handles_of_one_stack_serve_the_next_and_are_checked_against_its_geometry(L, oracle):
    """A job of many stacks keeps the stacker, the estimator and the scratch buffers (`keep_handles` / `handles=`): a second,
    DIFFERENT stack through the same handles equals a fresh call; a call whose geometry the scratch buffers were not made
    for (more frames per batch, another frame size, another sub-sampling) is refused before the library could write past
    them; an alignment failure releases what the failing call itself created; the LINEAR balance factors are returned."""
    from shinestacker_amd.errors import AlignmentError, InvalidOptionError
    from shinestacker_amd.pipeline import StackHandles, align_and_stack_device, close_handles
    h, w, n = 256, 384, 6

    def stack_frames(seed):
        frames = []
        for f in range(n):
            d = f - n // 2
            T = similarity(0.1 * d, 1 + 3e-4 * d, 1.1 * d, -0.7 * d, (w - 1) / 2, (h - 1) / 2)
            ref, mov = make_pair(oracle, T, h=h, w=w, seed=seed, noise=2.0)
            frames.append(np.clip((ref if d == 0 else mov) * (1.0 + 0.05 * d), 0, 255).astype(np.uint8))
        buf = L.DeviceBuffer(n * frames[0].nbytes)
        for f, fr in enumerate(frames):
            buf.upload(fr, f * fr.nbytes)
        return buf
    a, b = stack_frames(41), stack_frames(42)
    cfg = {'subsample': 1}
    kw = dict(alignment_config=cfg, batch_frames=4)
    fresh_a, tr_a, _ = align_and_stack_device(a.ptr, n, h, w, np.uint8, **kw)
    fresh_b, tr_b, _ = align_and_stack_device(b.ptr, n, h, w, np.uint8, **kw)
    out, tr, _, hd = align_and_stack_device(a.ptr, n, h, w, np.uint8, keep_handles=True, **kw)
    assert isinstance(hd, StackHandles) and np.array_equal(out, fresh_a)
    out, tr, _ = align_and_stack_device(b.ptr, n, h, w, np.uint8, handles=hd, **kw)          # another stack, same handles
    assert np.array_equal(out, fresh_b) and all(np.array_equal(x, y) for x, y in zip(tr[:n // 2], tr_b[:n // 2]))
    out, _, _, hd2 = align_and_stack_device(a.ptr, n, h, w, np.uint8, handles=hd, keep_handles=True, **kw)
    assert hd2 is hd and np.array_equal(out, fresh_a)
    for bad_kw, bad_shape in ((dict(alignment_config=cfg, batch_frames=8), (h, w)),
                              (dict(alignment_config={'subsample': 2}, batch_frames=4), (h, w)),
                              (dict(kw, arith="exact"), (h, w)),          # the stacker's own options are part of the geometry
                              (kw, (h // 2, w))):
        with pytest.raises(InvalidOptionError, match="handles"):
            align_and_stack_device(a.ptr, n, bad_shape[0], bad_shape[1], np.uint8, handles=hd, **bad_kw)
    with pytest.raises(InvalidOptionError, match="handles"):                                  # refused before any allocation
        align_and_stack_device(a.ptr, n, h, w, np.uint8, keep_handles=True, native_loop=False, **kw)
    # the handles survive the refusals
    out, _, _ = align_and_stack_device(b.ptr, n, h, w, np.uint8, handles=hd, **kw)
    assert np.array_equal(out, fresh_b)
    # an alignment failure with keep_handles: AlignmentError, nothing returned, and what the failing call created is released
    # (free device memory is the same after a second failing call as after the first: nothing accumulates)
    free = []
    for _ in range(3):
        with pytest.raises(AlignmentError):
            align_and_stack_device(a.ptr, n, h, w, np.uint8, keep_handles=True, min_correlation=2.0, **kw)
        free.append(L.mem_info()[0])
    assert free[2] >= free[1] - (1 << 20), free
    # LINEAR balance inside the native loop: the correction factors come back, one row per processed frame
    info = {}
    align_and_stack_device(a.ptr, n, h, w, np.uint8, balance={'channel': 'LUMI', 'corr_map': 'LINEAR', 'subsample': 2},
                           info=info, **kw)
    c = info["corrections"]
    assert c.shape == (n, 1) and c[n // 2, 0] == 1.0 and np.all(c > 0.5) and np.all(c < 2.0) and np.ptp(c) > 0.05
    close_handles(hd)
    a.free()
    b.free()


def _phase_correlate_numpy(ref, mov):
    """cv2.phaseCorrelate's recipe [from memory] in float64: Hann window, zero-padding to powers of two, R = Mov conj(Ref)
    normalised, inverse DFT, first arg-max, 5 x 5 weighted centroid (wrapping), response = window sum / (P Q)."""
    h, w = ref.shape
    P, Q = 1 << int(np.ceil(np.log2(h))), 1 << int(np.ceil(np.log2(w)))
    wy = 0.5 * (1 - np.cos(2 * np.pi * np.arange(h) / (h - 1)))
    wx = 0.5 * (1 - np.cos(2 * np.pi * np.arange(w) / (w - 1)))
    win = np.outer(wy, wx)
    a, b = np.zeros((P, Q)), np.zeros((P, Q))
    a[:h, :w], b[:h, :w] = ref * win, mov * win
    R = np.fft.fft2(b) * np.conj(np.fft.fft2(a))
    mag = np.abs(R)
    R = np.where(mag > 1e-20, R / np.maximum(mag, 1e-300), 0)
    c = np.fft.ifft2(R).real * (P * Q)
    py, px = np.unravel_index(np.argmax(c), c.shape)
    m = mx = my = 0.0
    for dy in range(-2, 3):
        for dx in range(-2, 3):
            v = c[(py + dy) % P, (px + dx) % Q]
            m, mx, my = m + v, mx + v * (px + dx), my + v * (py + dy)
    cx, cy = mx / m, my / m
    return (cx - Q if cx > Q / 2 else cx), (cy - P if cy > P / 2 else cy), m / (P * Q)


@pytest.mark.parametrize("shape,shift", [((250, 375), (17.0, -9.0)), ((256, 512), (-40.0, 31.0)), ((125, 188), (3.4, 7.7)),
                                         ((300, 300), (0.0, 0.0))])
def test_phase_correlation_equals_its_float64_statement_and_finds_the_shift(L, shape, shift):
    """mi_phase_correlate_device (hand-written LDS radix-2 DFT, cross-power spectrum, peak + centroid) against the NumPy
    float64 statement of the same recipe, and against the shift the second plane was made with:
    mov(x + dx, y + dy) ~ ref(x, y)."""
    from scipy import ndimage
    h, w = shape
    big = texture(h + 200, w + 200, seed=7).astype(np.float64)
    big = ndimage.gaussian_filter(big, 1.0)
    dx, dy = shift
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    ref = ndimage.map_coordinates(big, [yy + 100, xx + 100], order=3)
    mov = ndimage.map_coordinates(big, [yy + 100 - dy, xx + 100 - dx], order=3)      # mov(x) = ref(x - d)
    gx, gy, gr = L.phase_correlate(ref.astype(np.float32), mov.astype(np.float32))
    wx, wy, wr = _phase_correlate_numpy(ref.astype(np.float32).astype(np.float64), mov.astype(np.float32).astype(np.float64))
    assert abs(gx - wx) < 2e-3 and abs(gy - wy) < 2e-3 and abs(gr - wr) < 2e-3 * max(1.0, wr), (gx, gy, gr, wx, wy, wr)
    assert abs(gx - dx) < 0.35 and abs(gy - dy) < 0.35 and gr > 0.1, (gx, gy, gr)


def test_phase_correlation_extends_the_capture_range_of_the_ecc_estimator(L, oracle):
    """A frame displaced by a fifth of its width (beyond what the coarsest ECC level can pull in) plus a small rotation and
    scale: the plain estimator fails or lands elsewhere, the one that starts from the phase-correlation translation recovers
    the transform to the reference's precision-test tolerances (tests/test_0031_align_precision.py:62-65)."""
    h, w = 768, 1024
    T = similarity(0.3, 1.002, 205.0, -118.0, (w - 1) / 2, (h - 1) / 2)
    ref, mov = make_pair(oracle, T, h=h, w=w, seed=21, noise=2.0)
    buf = L.DeviceBuffer(2 * ref.nbytes)
    buf.upload(ref)
    buf.upload(mov, ref.nbytes)
    want = invert(T)                                   # moving -> reference, what the estimator returns
    out = {}
    for phase in (False, True):
        al = L.Aligner(h, w, np.uint8, subsample=1, phase_init=phase)
        al.set_reference(buf.ptr)
        try:
            m, cc, _ = al.estimate(buf.ptr + ref.nbytes)
            ang, sc = decompose(m)[:2]
            wang, wsc = decompose(want)[:2]
            ctr = np.array([(w - 1) / 2, (h - 1) / 2, 1.0])
            out[phase] = (abs(ang - wang), abs(sc - wsc), np.abs(m @ ctr - want @ ctr).max(), cc)
        except Exception:   # noqa: BLE001
            out[phase] = None
        al.close()
    buf.free()
    good = out[True]
    assert good is not None and good[0] < 0.005 and good[1] < 1e-4 and good[2] < 0.2 and good[3] > 0.9, out
    assert out[False] is None or out[False][2] > 5.0 or out[False][3] < 0.5, out    # the plain estimator does not get there


def _corner_error(M_est, M_true, h, w):
    pts = np.array([[0, 0, 1], [w - 1, 0, 1], [0, h - 1, 1], [w - 1, h - 1, 1], [(w - 1) / 2, (h - 1) / 2, 1]], float).T
    a, b = M_est @ pts, M_true @ pts
    return np.abs(a[:2] / a[2] - b[:2] / b[2]).max()


@pytest.mark.parametrize("dtype,subsample", [(np.uint8, 1), (np.uint16, 1), (np.uint8, 2)])
def test_homography_refinement_recovers_a_projective_transform(L, oracle, dtype, subsample):
    """ALIGN_HOMOGRAPHY without OpenCV (cv2.findHomography's role, align.py:138-140): the ECC similarity refined to 8
    degrees of freedom on the finest level (mi_aligner_estimate_homography_batch).  A frame seen through a mild
    perspective -- 2.5 % change of scale across the frame on top of a rotation, scale and shift -- is registered with all
    four corners within 0.2 px (the shift tolerance of tests/test_0031_align_precision.py:62-65); the plain similarity
    leaves more than a pixel; a frame that IS a similarity gives the same transform either way."""
    h, w = 768, 1024
    cx, cy = (w - 1) / 2, (h - 1) / 2
    S = np.vstack([similarity(0.4, 1.003, 6.0, -4.0, cx, cy), [0, 0, 1]])
    P = np.array([[1, 0, 0], [0, 1, 0], [2.4e-5, -1.2e-5, 1]], float)
    C, Ci = np.array([[1, 0, cx], [0, 1, cy], [0, 0, 1.0]]), np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1.0]])
    T = S @ C @ P @ Ci                                   # reference -> moving: perspective about the centre, then the similarity
    T /= T[2, 2]
    base = np.clip(np.repeat(texture(h, w, 17)[:, :, None], 3, 2), 0, 255).astype(np.uint8)
    rng = np.random.default_rng(9)
    mov8 = oracle.warp_perspective(base, T, border_mode=oracle.BORDER_REPLICATE)
    scale = 1 if dtype == np.uint8 else 257
    ref = (np.clip(base + rng.normal(0, 2, base.shape), 0, 255) * scale).astype(dtype)
    mov = (np.clip(mov8 + rng.normal(0, 2, base.shape), 0, 255) * scale).astype(dtype)
    want = np.linalg.inv(T)
    want /= want[2, 2]
    buf = L.DeviceBuffer(2 * ref.nbytes)
    buf.upload(ref)
    buf.upload(mov, ref.nbytes)
    al = L.Aligner(h, w, dtype, subsample=subsample)
    al.set_reference(buf.ptr)
    ms, ccs, its = al.estimate_homography_batch([buf.ptr + ref.nbytes])
    m_sim, cc_sim, _ = al.estimate(buf.ptr + ref.nbytes)
    e_h = _corner_error(ms[0], want, h, w)
    e_s = _corner_error(np.vstack([m_sim, [0, 0, 1]]), want, h, w)
    assert ms[0].shape == (3, 3) and ms[0][2, 2] == 1.0 and ccs[0] >= cc_sim - 1e-9
    assert e_h < 0.2 and e_s > 1.0, (e_h, e_s, ccs[0], cc_sim, its)
    # the apply step takes the matrix as it is: the warped frame matches the reference better than the similarity's
    got_h = L.warp_perspective(mov, ms[0], border_mode=L.BORDER_REPLICATE)
    got_s = L.warp_affine(mov, m_sim, border_mode=L.BORDER_REPLICATE)
    inner = (slice(60, h - 60), slice(60, w - 60))
    err = lambda a: np.abs(a[inner].astype(np.float64) - ref[inner]).mean() / scale   # noqa: E731
    assert err(got_h) < err(got_s)
    # a frame that is a similarity: the refinement has nothing to add
    ref2, mov2 = make_pair(oracle, similarity(0.3, 1.002, 5.0, -3.0, cx, cy), h=h, w=w, seed=23, noise=2.0, dtype=dtype)
    buf.upload(ref2)
    buf.upload(mov2, ref.nbytes)
    al.set_reference(buf.ptr)
    ms2, _, _ = al.estimate_homography_batch([buf.ptr + ref.nbytes])
    m2, _, _ = al.estimate(buf.ptr + ref.nbytes)
    assert _corner_error(ms2[0], np.vstack([m2, [0, 0, 1]]), h, w) < 0.1
    al.close()
    buf.free()


def test_align_images_homography_with_the_gpu_estimator(L, oracle):
    """`align_images(..., transform=ALIGN_HOMOGRAPHY)` with `ecc_estimator()`: a 3 x 3 matrix that is NOT affine comes back
    and the warp is applied through mi_warp_perspective."""
    from shinestacker_amd.align import align_images, ecc_estimator
    from shinestacker_amd.defaults import constants as c
    h, w = 512, 640
    cx, cy = (w - 1) / 2, (h - 1) / 2
    T = np.array([[1, 0, 3.0], [0, 1, -2.0], [0, 0, 1.0]]) @ np.array([[1, 0, cx], [0, 1, cy], [0, 0, 1.0]]) @ \
        np.array([[1, 0, 0], [0, 1, 0], [4e-5, 3e-5, 1.0]]) @ np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1.0]])
    base = np.clip(np.repeat(texture(h, w, 5)[:, :, None], 3, 2), 0, 255).astype(np.uint8)
    mov = oracle.warp_perspective(base, T / T[2, 2], border_mode=oracle.BORDER_REPLICATE)
    n, m, warp = align_images(base, mov, alignment_config={'transform': c.ALIGN_HOMOGRAPHY, 'subsample': 1},
                              estimator=ecc_estimator())
    want = np.linalg.inv(T)
    assert n == 1000 and m.shape == (3, 3) and abs(m[2, 0]) > 1e-5 and _corner_error(m, want / want[2, 2], h, w) < 0.2
    assert np.array_equal(warp, oracle.warp_perspective(mov, m))


def test_auto_batch_frames_is_bounded_by_the_job_and_by_memory(L):
    """pipeline.auto_batch_frames: a multiple of 16 between 16 and 128, never more than the job needs, and small when the
    share of device memory it may use is small."""
    from shinestacker_amd.pipeline import auto_batch_frames
    assert auto_batch_frames(5, 4000, 6000, np.uint8) == 16
    assert auto_batch_frames(40, 4000, 6000, np.uint8) == 48
    assert auto_batch_frames(1000, 400, 600, np.uint8) == 128
    free, _ = L.mem_info()
    per_frame = 2 * 4000 * 6000 * 3 + 2 * (4000 * 6000 // 3 + 1) * 12
    tight = 40 * per_frame / free          # room for 40 frames -> 32
    assert auto_batch_frames(1000, 4000, 6000, np.uint8, share=tight) == 32
    assert auto_batch_frames(1000, 4000, 6000, np.uint8, share=1e-9) == 16



def test_chain_refinement_on_a_simulated_focus_stack(L):
    """The refinement's datum is the GLOBAL reference frame, which in a focus stack looks different from a far frame
    (tools/parity_report.py::defocus_frames: every frame blurred by its distance from the focal plane).  With a known
    similarity per frame (rotation, focus-breathing scale, shift) the refined chain must stay inside the 0.2 px of
    tests/test_0031_align_precision.py:62-65 at every corner, and must not be worse than the plain chain."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import parity_report as pr
    from shinestacker_amd.pipeline import align_and_stack_device
    n, h, w = 12, 1000, 1500
    frames = pr.defocus_frames(h, w, n, np.uint8)
    ref = n // 2
    cx, cy = (w - 1) / 2, (h - 1) / 2
    truth, moved = [], []
    for f in range(n):
        d = f - ref
        T = similarity(0.03 * d, 1 + 4e-4 * d, 0.8 * d, -0.5 * d, cx, cy)
        truth.append(np.array(T))
        moved.append(frames[f] if d == 0 else L.warp_affine(frames[f], np.array(T), border_mode=L.BORDER_REPLICATE))
    fb = moved[0].nbytes
    buf = L.DeviceBuffer(n * fb)
    for f, fr in enumerate(moved):
        buf.upload(fr, f * fb)
    corners = np.array([[0, 0, 1.0], [w - 1, 0, 1], [0, h - 1, 1], [w - 1, h - 1, 1]]).T

    def worst(tr):
        out = 0.0
        for f in range(n):
            if f == ref:
                continue
            Ai = np.linalg.inv(truth[f][:, :2])
            want = np.hstack([Ai, -Ai @ truth[f][:, 2:3]])
            out = max(out, float(np.abs(np.asarray(tr[f])[:2] @ corners - want @ corners).max()))
        return out
    for serial in (True, False):      # the serial chain (rounds 3-5) and its factored form (neighbour pairs + composition)
        err = {}
        for refine in (False, True):
            _, tr, ccs = align_and_stack_device(buf.ptr, n, h, w, np.uint8, ref_idx=ref, alignment_config={'subsample': 1},
                                                step_process=True, chain_refine=refine, chain_serial=serial)
            assert min(ccs) > 0.9
            err[refine] = worst(tr)
        print(f"\n[chain on a simulated focus stack, {'serial' if serial else 'pairs'}] worst corner error, plain / refined:",
              err[False], err[True])
        # (the serial chain's plain errors add up and the refinement must not make them worse; the factored chain registers
        # against unresampled neighbours -- its plain form is already the better one on this stack, 0.09 against 0.10 px)
        assert err[True] < 0.2 and (err[True] <= err[False] + 0.02 or not serial), (serial, err)
    buf.free()
