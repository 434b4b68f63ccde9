#!/usr/bin/env python3
"""BASELINE config 4 at reduced scale: alignment + fusion end to end on one GPU, no OpenCV.

N frames of the SURVEY 8(d) generator at H x W (uint8), each warped by a known similarity
(theta_f = 0.02 deg*(f-ref), s_f = 1 + 1e-4*(f-ref), t_f = (0.37, -0.21)*(f-ref) px); every frame
is aligned to the middle frame with the GPU ECC estimator, warped + border-blurred on the GPU and
fused in memory (shinestacker_amd.pipeline.align_and_stack).  Prints accuracy and throughput."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def report(N, H, W, dt, recovered, truth, ref, cx, cy, where, shape):
    # accuracy: recovered (moving -> reference) vs inverse of the applied transform, at sub-sampled scale
    worst = {"angle_deg": 0.0, "scale": 0.0, "shift_px": 0.0}
    k = 0
    for f in range(N):
        if f == ref:
            continue
        m = np.asarray(recovered[k])[:2]   # ALIGN_HOMOGRAPHY: the similarity sits in the first two rows of the 3 x 3 matrix
        k += 1
        A = truth[f][:, :2]
        Ai = np.linalg.inv(A)
        want = np.hstack([Ai, -Ai @ truth[f][:, 2:3]])
        want_sub = want.copy()
        want_sub[:, 2] /= 2  # the estimator saw 2x sub-sampled images
        ang = np.rad2deg(np.arctan2(m[1, 0], m[0, 0])) - np.rad2deg(np.arctan2(want[1, 0], want[0, 0]))
        sc = np.hypot(m[0, 0], m[1, 0]) - np.hypot(want[0, 0], want[1, 0])
        c = np.array([cx / 2, cy / 2, 1.0])
        sh = np.abs(m @ c - want_sub @ c).max() * 2
        worst = {"angle_deg": max(worst["angle_deg"], abs(ang)), "scale": max(worst["scale"], abs(sc)),
                 "shift_px": max(worst["shift_px"], sh)}
    print(json.dumps({"config": f"{N}x{W}x{H} u8, ECC estimate + warp + border blur + pyramid stack ({where})",
                      "seconds": dt, "Mpixels_per_s": N * H * W / dt / 1e6, "worst_error": worst,
                      "tolerance": {"angle_deg": 0.005, "scale": 1e-4, "shift_px": 0.2},
                      "fused_shape": shape}))




def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--height", type=int, default=4000)
    ap.add_argument("--width", type=int, default=6000)
    ap.add_argument("--balance", action="store_true", help="also balance every frame (LUMI / LINEAR, sub-sample 8)")
    ap.add_argument("--resident", action="store_true", help="frames resident in HBM (mi_aligner_* + device warp)")
    ap.add_argument("--homography", action="store_true", help="ALIGN_HOMOGRAPHY: the estimate applied with warpPerspective (resident mode)")
    ap.add_argument("--batch", type=int, default=0, help="warped frames per push into the stacker (resident mode; 0 = the pipeline's choice: as many as memory allows, up to 128)")
    ap.add_argument("--step-process", action="store_true", help="the reference's chained order (resident mode)")
    ap.add_argument("--no-chain-refine", action="store_true", help="with --step-process: the plain chain (no refinement against the global reference frame)")
    ap.add_argument("--chain-serial", action="store_true", help="with --step-process: every step against the WARPED neighbour, one device synchronisation per frame (rounds 3-5)")
    ap.add_argument("--arith", default="exact", choices=["exact", "separable"], help="stacker arithmetic (resident mode)")
    ap.add_argument("--ecc-batch", type=int, default=0, help="frames per batched Gauss-Newton (0 = the pipeline's default)")
    ap.add_argument("--reuse-handles", action="store_true", help="time a second stack on the handles of a first one (no allocation in the timed region)")
    ap.add_argument("--python-loop", action="store_true", help="the call-by-call Python loop instead of mi_align_stack_device")
    args = ap.parse_args()
    from shinestacker_amd import _lib as L
    from shinestacker_amd.align import ecc_estimator
    from shinestacker_amd.pipeline import align_and_stack
    N, H, W = args.frames, args.height, args.width
    ref = N // 2
    cx, cy = (W - 1) / 2, (H - 1) / 2
    frames, truth = [], []
    # every frame shares the same scene (broadband: octaves of smooth random fields plus pixel noise;
    # the SURVEY 8(d) generator's periodic ramp has an aperture problem, so it is not used here), seen
    # under a slightly different similarity, as focus breathing produces
    from scipy import ndimage
    rng = np.random.default_rng(4)
    field = np.zeros((H, W), np.float32)
    for k in range(2, 9):
        g = rng.standard_normal((H // 2 ** k + 2, W // 2 ** k + 2)).astype(np.float32)
        up = ndimage.zoom(g, 2 ** k, order=1)[:H, :W]
        field += up * (2.0 ** (k - 5))
    field = (field - field.min()) / (field.max() - field.min())
    scene = np.empty((H, W, 3), np.uint8)
    for c in range(3):
        scene[..., c] = np.clip(30 + 190 * field + rng.integers(-6, 7, (H, W)) + 5 * c, 0, 255).astype(np.uint8)
    for f in range(N):
        d = f - ref
        t = np.deg2rad(0.02 * d)
        s = 1 + 1e-4 * d
        a, b = s * np.cos(t), s * np.sin(t)
        T = np.array([[a, -b, cx - a * cx + b * cy + 0.37 * d], [b, a, cy - b * cx - a * cy - 0.21 * d]])
        frames.append(scene if d == 0 else L.warp_affine(scene, T, border_mode=L.BORDER_REPLICATE))
        truth.append(T)
    if args.resident:
        from shinestacker_amd.pipeline import align_and_stack_device
        fb = H * W * 3
        buf = L.DeviceBuffer(N * fb)
        for f, fr in enumerate(frames):
            buf.upload(fr, f * fb)
        out = L.DeviceBuffer(fb)
        bal = {'channel': 'LUMI', 'corr_map': 'LINEAR', 'subsample': 8} if args.balance else None
        acfg = {'transform': 'ALIGN_HOMOGRAPHY'} if args.homography else None
        align_and_stack_device(buf.ptr, min(N, 4), H, W, np.uint8, out_dev=out.ptr, balance=bal, arith=args.arith, step_process=args.step_process, chain_refine=not args.no_chain_refine, chain_serial=args.chain_serial, batch_frames=(args.batch or None), alignment_config=acfg, native_loop=not args.python_loop, **({'ecc_batch': args.ecc_batch} if args.ecc_batch else {}))   # warm-up
        if args.reuse_handles:   # what a job of many stacks pays per stack: the handles exist already
            if args.python_loop or (args.step_process and (args.chain_serial or args.homography)):
                # (round 3 dropped these flags silently here and wrote a non-chained run into config4_resident_step.json)
                raise SystemExit("--reuse-handles times the native non-chained loop or the factored step_process chain; it cannot be combined with --python-loop / --chain-serial")
            kw = dict(balance=bal, arith=args.arith, batch_frames=(args.batch or None), alignment_config=acfg, **({'ecc_batch': args.ecc_batch} if args.ecc_batch else {}))
            if args.step_process:
                kw = dict(balance=bal, arith=args.arith, alignment_config=acfg, step_process=True, chain_refine=not args.no_chain_refine)
            from shinestacker_amd.pipeline import close_handles
            *_, hd = align_and_stack_device(buf.ptr, N, H, W, np.uint8, ref_idx=ref, out_dev=out.ptr, keep_handles=True, **kw)
            t0 = time.perf_counter()
            _, tr, ccs = align_and_stack_device(buf.ptr, N, H, W, np.uint8, ref_idx=ref, out_dev=out.ptr, handles=hd, **kw)
            dt = time.perf_counter() - t0
            close_handles(hd)
            recovered = {k: m.copy() for k, m in enumerate(t for t in tr if t is not None)}
            for m in recovered.values():
                m[:, 2] /= 2
            report(N, H, W, dt, recovered, truth, ref, cx, cy, ("resident, step_process (chained, neighbour pairs + composition" + (", plain" if args.no_chain_refine else ", refined against the global reference") + "), handles reused") if args.step_process else "resident, handles reused", list(out.download((H, W, 3), np.uint8).shape))
            return
        t0 = time.perf_counter()
        _, tr, ccs = align_and_stack_device(buf.ptr, N, H, W, np.uint8, ref_idx=ref, out_dev=out.ptr, balance=bal, arith=args.arith, step_process=args.step_process, chain_refine=not args.no_chain_refine, chain_serial=args.chain_serial, batch_frames=(args.batch or None), alignment_config=acfg, native_loop=not args.python_loop, **({'ecc_batch': args.ecc_batch} if args.ecc_batch else {}))
        dt = time.perf_counter() - t0
        recovered = {k: m.copy() for k, m in enumerate(t for t in tr if t is not None)}
        for m in recovered.values():
            m[:, 2] /= 2    # compare at the sub-sampled scale like the host path below
        report(N, H, W, dt, recovered, truth, ref, cx, cy, ("resident, step_process (chained" + (", serial" if args.chain_serial else ", neighbour pairs + composition") + (", plain" if args.no_chain_refine else ", refined against the global reference") + ")" if args.step_process else "resident") + (" + balance" if args.balance else ""), list(out.download((H, W, 3), np.uint8).shape))
        return
    est = ecc_estimator()
    recovered = {}

    def recording(i0, i1, fc, mc, ac):
        n, m = est(i0, i1, fc, mc, ac)
        recovered[len(recovered)] = m
        return n, m
    t0 = time.perf_counter()
    fused, matches = align_and_stack(frames, ref_idx=ref, estimator=recording,
                                     alignment_config={'fast_subsampling': True})
    dt = time.perf_counter() - t0
    report(N, H, W, dt, recovered, truth, ref, cx, cy, "host arrays", list(fused.shape))


if __name__ == "__main__":
    main()
