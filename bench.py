#!/usr/bin/env python3
"""bench.py -- Mpixels/s fused (pyramid build + select + collapse) on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N>1 is launched by torch.distributed.run, one rank per GPU (RCCL).
A "step" is one whole focus-stack job over a device-resident synthetic stack:
push all frames (pyramid build + per-level selection), base fusion, cross-GPU
combine (N>1), collapse, abs/clip/truncating cast -- result left in HBM.
Workload at N=1: BASELINE.json configs[1] -- 256 x 24 MP (4000x6000x3) fp32
frames, 6 Laplacian levels + 63x94 base.
  --scaling weak   (default) every rank holds its own 256-frame shard of a 256*N-frame stack;
  --scaling strong BASELINE.json configs[2]: the SAME 256 frames, 256/N per rank.
  --shards interleaved (default): rank r of N holds frames r, r + N, ...; contiguous: the block [r F, (r + 1) F).
  --arith separable (default) north_star's LDS-staged 5-tap separable / polyphase form (MI_ARITH_SEPARABLE,
                    tolerance-tested against float64, bit-exact against oracle/separable_oracle.c);
  --arith exact     the reference-order 25-tap form, bit-identical to the oracle of the reference's own run.
At N=1 the other mode is measured too (same K and W) and reported under "other_mode".
After the timed region the last step's result is verified (outside the timing): level-0 arg-max against the
generator's known band structure, and the level-0 state of a 128x128 corner against the CPU oracle fed the same
frames cropped -> "verified".

One JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
CROP, CROP_LEVELS, CROP_MARGIN = 512, 3, 40   # verify(): crop side, pyramid levels compared, level-0 pixels a crop border can reach


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=256,
                    help="frames per GPU (weak scaling) / frames of the whole stack (strong scaling)")
    ap.add_argument("--height", type=int, default=4000)
    ap.add_argument("--width", type=int, default=6000)
    ap.add_argument("--dtype", default="f32", choices=["u8", "u16", "f32"])
    ap.add_argument("--impl", default="auto", choices=["auto", "simple", "tiled"])
    ap.add_argument("--arith", default="separable", choices=["separable", "exact"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--shards", default="interleaved", choices=["interleaved", "contiguous"],
                    help="how the frames of the stack are dealt to the ranks: rank r of W holds frames r, r + W, ... (default: every "
                         "rank sees the whole focus range, its winners are as coherent as the whole stack's and its payload pass "
                         "as cheap) or the contiguous block [r F, (r + 1) F) (rounds 1-5)")
    ap.add_argument("--batch", type=int, default=0, help="frames per fused launch (0 = library default)")
    ap.add_argument("--source", default="device", choices=["device", "host", "host-pinned"],
                    help="host: frames are pushed from host memory one by one (PCIe-inclusive rate; "
                         "never the headline value)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-other-mode", action="store_true", help="do not also measure the other arithmetic mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-dtypes", action="store_true",
                    help="do not also time the same stack held as uint8 / uint16 frames (the reference's real input types)")
    ap.add_argument("--no-projection", action="store_true",
                    help="do not add the PROJECTED 2 / 4 / 8-GPU lines (measured single-GPU compute at 256/N frames + measured "
                         "local combine kernels + the xGMI link rate; labelled 'projected', never a measurement)")
    ap.add_argument("--cpu-frames", type=int, default=64,
                    help="frames of the CPU-baseline sample (about 10 s of CPU work at 24 MP on 16 cores)")
    ap.add_argument("--cpu-refshaped", type=int, default=4,
                    help="also time the reference-SHAPED restatement (all pyramids resident, full-size 25-tap "
                         "filters, argmax over the frame axis: SURVEY 8(d) leg (i)) on this many frames (~4 s per 24 MP "
                         "frame; 0 = skip)")
    ap.add_argument("--force-combine", action="store_true",
                    help="world 1 only: run the cross-GPU exchange's per-rank kernel work (winner map, plan, pack, unpack) "
                         "inside every step although nothing has to move -> combine_ms (timing of the local part of the "
                         "exchange on a box with one GPU; the fabric part needs the driver's multi-GPU run)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: every rank runs the CPU oracle on its block of a small stack and the ranks combine over "
                         "gloo with the same protocol (multigpu.combine_winners) -- exercises the launch contract, the "
                         "weak / strong split, the timing reduction and the JSON fields on a box without GPUs")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic.json"),
                    help="per-launch HBM bytes from the PMC passes (tools/pmc_traffic.py)")
    return ap.parse_args()


def kernel_source_sha():
    """provenance of profiles/traffic.json: sha256 over the sources of the level kernels and their launch schedule"""
    import hashlib
    h = hashlib.sha256()
    for f in ("kernels_sep.hpp", "kernels_tiled.hpp", "tiled_host.hpp"):
        with open(os.path.join(ROOT, "shinestacker_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, total_frames):
    """The oracle's streaming C port, timed on this host's cores on a bounded sample of
    the same workload (same generator, same geometry, fewer frames)."""
    from oracle import oracle as orc
    orc.build()
    n = min(args.cpu_frames, total_frames)
    H, W = args.height, args.width

    def frame(f):   # generated one at a time, outside the timed sections
        fr = orc.synth_frame_u8(H, W, f, total_frames)
        if args.dtype == "f32":   # the workload's own type: float-32 frames holding the generator's 8-bit values
            return fr.astype(np.float32)
        return fr.astype(np.uint16) * 257 if args.dtype == "u16" else fr
    first = frame(0)
    so = orc.StreamingOracle(H, W, np.uint8 if first.dtype == np.float32 else first.dtype, keep_gauss=False, arith=args.arith)   # (the result's type)
    dt = 0.0
    for f in range(n):
        fr = first if f == 0 else frame(f)
        t0 = time.perf_counter()
        so.push_frame(fr)
        dt += time.perf_counter() - t0
    t0 = time.perf_counter()
    so.finish()
    dt += time.perf_counter() - t0
    out = {"value": n * H * W / dt / 1e6, "unit": "Mpixels/s", "cores": orc.lib().orc_num_threads(),
           "kind": "port", "cpu_model": cpu_model(), "os_cpu_count": os.cpu_count(),
           "sample": f"{n} of {total_frames} frames of {W}x{H} (same generator, the same values as {first.dtype} frames -- "
                     f"the GPU leg's type is {args.dtype}), oracle.StreamingOracle(arith={args.arith}) push+finish, "
                     f"OpenMP on {orc.lib().orc_num_threads()} threads, {dt:.1f} s"}
    if args.cpu_refshaped > 0:
        # SURVEY 8(d) leg (i): the reference's own structure (per-channel full-size 25-tap filters, zero-stuffed
        # expand, every frame's pyramid resident, argmax over the frame axis) on a reduced stack
        m = args.cpu_refshaped
        # (8- / 16-bit frames, as the reference reads them from files: its base rule takes the histogram of the integer image)
        frames = [orc.synth_frame_u8(H, W, f, total_frames) for f in range(m)]
        if args.dtype == "u16":
            frames = [fr.astype(np.uint16) * 257 for fr in frames]
        t0 = time.perf_counter()
        orc.RefShaped().stack(frames)
        dr = time.perf_counter() - t0
        out["reference_shaped"] = {"value": m * H * W / dr / 1e6, "unit": "Mpixels/s", "frames": m,
                                   "seconds": dr, "note": "oracle.RefShaped (NumPy control flow of pyramid.py, "
                                   "filter primitives in C/OpenMP); linear in the frame count"}
    # The cv2 primitives the oracle restates are third-party code this repository cannot pin (no OpenCV in the build image,
    # unpinned in the reference's pyproject.toml:26).  Whenever a box that runs this bench HAS OpenCV, the real primitives are
    # compared with the restatements here (oracle/probe_cv2.py; outside every timed region) and the verdict travels in the line.
    try:
        from oracle import probe_cv2
        out["cv2"] = probe_cv2.probe(write=False)
    except Exception as e:  # noqa: BLE001  (a probe failure must not cost the bench line)
        out["cv2"] = {"available": None, "error": f"probe failed: {type(e).__name__}: {e}"}
    return out


def verify(L, st, args, total_frames, world):
    """Checks on the state the LAST timed step left behind (rank 0; outside the timed region)."""
    from oracle import oracle as orc
    orc.build()
    H, W = args.height, args.width
    res = {}
    ok = True
    if world == 1:
        idx = st.tap(L.TAP_INDEX, 0)
        band = (np.arange(H, dtype=np.int64) * total_frames // H)[:, None]
        res["band_match"] = float((idx == band).mean())
        ok &= res["band_match"] > 0.9
    # SURVEY 8(d) at full size: crops at the four image corners, the centre and a super-block seam of the level-0 tile
    # grid; the oracle is fed the generator frames cropped to each window and must reproduce the running state of
    # levels 0..CROP_LEVELS-1 wherever the crop's own (artificial) borders cannot reach.  A crop side that coincides
    # with an image edge is exact up to that edge (same REFLECT101, same parities: origins are multiples of 2^levels).
    taps = {}

    def tap(kind, lv):
        if (kind, lv) not in taps:
            taps[(kind, lv)] = st.tap(kind, lv)
        return taps[(kind, lv)]
    crops = []
    all_eq = True
    c = min(CROP, H // 8 * 8, W // 8 * 8)
    nlev = min(CROP_LEVELS, st.levels)
    while nlev > 1 and (c >> nlev) < 8:
        nlev -= 1
    al = 1 << nlev

    def origin(pos, n):                      # (origin, touches the near edge, touches the far edge)
        o = max(0, min(pos, n - c)) // al * al
        return o, o == 0, o + c == n
    seam_y, seam_x = (H // 2) // (8 * 28) * (8 * 28), (W // 2) // (8 * 56) * (8 * 56)   # corner of four 8x8-tile super-blocks
    wanted = [("top-left", 0, 0), ("top-right", 0, W), ("bottom-left", H, 0), ("bottom-right", H, W),
              ("centre", H // 2 - c // 2, W // 2 - c // 2), ("super-block seam", seam_y - c // 2 + 120, seam_x - c // 2 - 88)]
    seen = set()
    for name, py, px in wanted:
        (y0, top, bot), (x0, left, right) = origin(py, H), origin(px, W)
        if (y0, x0) in seen:
            continue
        seen.add((y0, x0))
        so = orc.StreamingOracle(c, c, np.uint16 if args.dtype == "u16" else np.uint8, levels=nlev, arith=args.arith)
        for f in range(total_frames):
            crop = orc.synth_crop_u8(H, W, f, total_frames, y0, x0, c, c)
            so.push_frame(crop.astype(np.uint16) * 257 if args.dtype == "u16" else crop)
        entry = {"name": name, "origin": [int(y0), int(x0)], "size": int(c), "levels": {}}
        eq_all = True
        for lv in range(nlev):
            m = -(-CROP_MARGIN // (1 << lv)) + 2          # level-lv pixels a crop border can reach
            cl = c >> lv
            ys = slice(0 if top else m, cl if bot else cl - m)
            xs = slice(0 if left else m, cl if right else cl - m)
            gy = slice((y0 >> lv) + ys.start, (y0 >> lv) + ys.stop)
            gx = slice((x0 >> lv) + xs.start, (x0 >> lv) + xs.stop)
            eq = bool(np.array_equal(tap(L.TAP_FUSED_LAP, lv)[gy, gx], so.best_lap[lv][ys, xs]))
            if world == 1:
                eq &= bool(np.array_equal(tap(L.TAP_ENERGY, lv)[gy, gx], so.best_e[lv][ys, xs]))
                eq &= bool(np.array_equal(tap(L.TAP_INDEX, lv)[gy, gx], so.best_idx[lv][ys, xs]))
            entry["levels"][str(lv)] = {"equal": eq, "pixels": int((ys.stop - ys.start) * (xs.stop - xs.start))}
            eq_all &= eq
        entry["equal"] = eq_all
        all_eq &= eq_all
        crops.append(entry)
    res["crops"] = crops
    res["crops_equal"] = bool(all_eq)
    res["corner_equal"] = bool(crops[0]["equal"])     # (the round 1-3 name of the first crop's verdict)
    res["corner"] = (f"fused Laplacian{' / energy / arg-max' if world == 1 else ''} of levels 0..{nlev - 1} inside "
                     f"{len(crops)} crops of {c}x{c} (image corners, centre, a super-block seam; {CROP_MARGIN} px in from "
                     f"every artificial crop border) == oracle.StreamingOracle(arith={args.arith}, levels={nlev}) fed the "
                     f"{total_frames} generator frames cropped to each window")
    ok &= all_eq
    res["ok"] = bool(ok)
    return res


XGMI_LINK_GBS = 153.0      # per link and direction; 7 links per GPU, point to point (the task's hardware notes)
XGMI_EFFICIENCY = 0.8      # assumed achievable fraction of the link rate for multi-megabyte RCCL transfers


def project_scaling(L, measure, args, H, W, total_frames, t1_s, device, job_bytes_per_frame):
    """PROJECTED strong scaling of this job over N = 2, 4, 8 GPUs of one node -- NOT a measurement (no multi-GPU node was
    available to any round).  Built from two things measured here and one taken from the hardware notes:
      compute   the single-GPU step on total/N frames (the rank's block), timed like the headline;
      local     the per-rank kernels of the winners-only exchange (winner map over the rank's pixel chunk, plan over the
                whole map, pack of the rows the rank won, unpack of the other ranks' rows at rank 0), timed on this GPU on a
                synthetic winner map in which rank r wins the r-th horizontal band of every level;
      link      bytes that cross rank 0's busiest xGMI link once: 4 + 4 (energies and, for the interleaved shards' ties, the
                winners' frame indices, all-to-all) + 1 (winner map, all-gather) + 12 (winners' Laplacians to rank 0) bytes per
                state pixel, divided by N (one peer's share), at XGMI_LINK_GBS x XGMI_EFFICIENCY.
    step(N) = compute + local + link with no overlap credited (the implementation exchanges level 0 while the rank's
    coarser levels still run; that saving is listed as `hidden_behind_coarse_levels_ms` and NOT subtracted)."""
    import ctypes as C
    out = {"label": "projected", "basis": "measured compute at total/N frames + measured local combine kernels + "
           f"{XGMI_LINK_GBS:.0f} GB/s x {XGMI_EFFICIENCY} per xGMI link; strong scaling of the {total_frames}-frame job",
           "measured_n1": {"ms_per_step": t1_s * 1e3}, "lines": [],
           # MEASURED on this GPU: the single-rank step on a shard of total/N frames (what one rank of an N-GPU job computes
           # before the exchange); vs_ideal = (the full job's step / N) / this
           "shard_step": []}
    lib = L.load()
    for n in (2, 4, 8):
        if total_frames % n:
            continue
        fn = total_frames // n
        # a shard step is a few milliseconds: enough steps that the timed region is ~0.2 s of steady state, three warm-ups
        # (the first steps after the host-side set-up run at a lower clock)
        k = max(args.steps, int(0.2 / max(t1_s / n, 1e-4)) + 1)
        # rank 0's shard of the N-GPU job: frames 0, n, 2n, ... (interleaved: every rank sees the whole focus range)
        dt_np = {"u8": np.uint8, "u16": np.uint16, "f32": np.float32}[args.dtype]
        shard = L.DeviceBuffer(H * W * 3 * np.dtype(dt_np).itemsize * fn, device)
        L.synth_frames_device(shard.ptr, dt_np, H, W, 0, fn, total_frames, device=device, frame_step=n)
        st, dt_s, prof, _ = measure(args.arith, k, 3, buf=shard, F=fn, index_step=n)
        compute_ms = dt_s / k * 1e3
        coarse_ms = prof["levels"][0] / k
        e_ptr, l_ptr, _i_ptr, npx = st.state_ptrs(-1)
        # synthetic winner map: the generator's bands, dealt to the ranks like its frames (band b is frame b's: rank b mod n)
        win_h = np.concatenate([np.repeat(((np.arange(h, dtype=np.int64) * total_frames // max(h, 1)) % n).astype(np.uint8), w)
                                for (h, w) in list(st.shapes[:-1]) + [st.shapes[-1], st.shapes[-1]]])
        win_h = np.concatenate([win_h, np.zeros(max(0, npx - win_h.size), np.uint8)])[:npx]
        win = L.DeviceBuffer(npx, device)
        win.upload(win_h)
        chunk = -(-npx // n)
        cand = L.DeviceBuffer(4 * chunk * n, device)
        wchunk = L.DeviceBuffer(chunk, device)
        plan = L.DeviceBuffer(lib.mi_combine_plan_bytes(npx, n), device)
        rows = L.DeviceBuffer(12 * npx, device)
        totals = (C.c_int64 * n)()
        ptrs_h = np.array([0] + [rows.ptr] * (n - 1), np.int64)   # (offsets are per rank: one shared buffer serves the timing)
        ptrs = L.DeviceBuffer(8 * n, device)
        ptrs.upload(ptrs_h)

        def local(rank):
            L.check(lib.mi_combine_winner(device, None, n, cand.ptr, chunk, wchunk.ptr))
            L.check(lib.mi_combine_plan(device, None, win.ptr, npx, n, plan.ptr, totals))
            if rank:
                L.check(lib.mi_combine_pack(device, None, win.ptr, npx, n, rank, plan.ptr, l_ptr, 3, rows.ptr))
            else:
                L.check(lib.mi_combine_unpack(device, None, win.ptr, npx, n, 0, plan.ptr, ptrs.ptr, 3, l_ptr))
            L.check(lib.mi_device_synchronize(device))
        t_local = {}
        for rank in (0, 1):
            local(rank)
            t0 = time.perf_counter()
            for _ in range(5):
                local(rank)
            t_local[rank] = (time.perf_counter() - t0) / 5 * 1e3
        for b in (win, cand, wchunk, plan, rows, ptrs, shard):
            b.free()
        st.close()
        local_ms = max(t_local.values())      # rank 0 unpacks, the others pack: the slower of the two bounds the step
        link_ms = 21.0 * npx / n / (XGMI_LINK_GBS * XGMI_EFFICIENCY * 1e9) * 1e3
        step_ms = compute_ms + local_ms + link_ms
        out["shard_step"].append({"frames": fn, "ms": compute_ms, "steps": k, "measured": True,
                                  "job_roofline_frac": job_bytes_per_frame * fn / (compute_ms * 1e-3) / (HBM_PEAK_GBS * 1e9),
                                  "vs_ideal": (t1_s * 1e3 / n) / compute_ms})
        out["lines"].append({"n_gpus": n, "label": "projected", "frames_per_gpu": fn,
                             "value": total_frames * H * W / step_ms / 1e3, "unit": "Mpixels/s", "ms_per_step": step_ms,
                             "compute_ms": compute_ms, "local_combine_ms": local_ms,
                             "local_combine_ms_by_role": {"root_unpack": t_local[0], "sender_pack": t_local[1]},
                             "link_ms": link_ms, "hidden_behind_coarse_levels_ms": min(coarse_ms, link_ms * 0.75),
                             "scaling_efficiency_vs_measured_n1": (t1_s * 1e3 / step_ms) / n})
    return out


def dry_run(args, rank, world):
    """The distributed skeleton of main() on CPU (see --dry-run)."""
    import torch
    import torch.distributed as dist
    from oracle import oracle as orc
    from shinestacker_amd import multigpu
    orc.build()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    H, W = min(args.height, 96), min(args.width, 128)
    total = args.frames if args.scaling == "strong" else args.frames * world
    if args.scaling == "strong" and total % world:
        raise SystemExit(f"--scaling strong: {total} frames do not split over {world} ranks")
    F = total // world
    inter = args.shards == "interleaved" and world > 1
    mine = range(rank, total, world) if inter else range(rank * F, (rank + 1) * F)
    frames = [orc.synth_frame_numpy(H, W, f, total) for f in mine]
    gidx = (lambda a: a * world + rank) if inter else (lambda a: a + rank * F)    # local frame number -> global frame index
    ops = multigpu.TorchWinnerOps()
    phase = {"compute": 0.0, "combine": 0.0, "collapse": 0.0}
    out = None

    def step():
        nonlocal out
        t0 = time.perf_counter()
        so = orc.StreamingOracle(H, W, np.uint8, min_size=16, arith=args.arith)
        for f in frames:
            so.push_frame(f)
        hb, wb = so.shapes[so.levels]
        yy, xx = np.mgrid[0:hb, 0:wb]
        bases = np.stack(so.bases)
        state = [(so.best_e[lv], so.best_lap[lv], gidx(so.best_idx[lv])) for lv in range(so.levels)]
        state += [(so.b_ent, bases[so.idx_e, yy, xx], gidx(so.idx_e)), (so.b_dev, bases[so.idx_d, yy, xx], gidx(so.idx_d))]
        e = torch.cat([torch.from_numpy(np.ascontiguousarray(a, np.float32).ravel()) for a, _, _ in state])
        lp = torch.cat([torch.from_numpy(np.ascontiguousarray(b, np.float32).ravel()) for _, b, _ in state])
        ix = torch.cat([torch.from_numpy(np.ascontiguousarray(c, np.int32).ravel()) for _, _, c in state])
        t1 = time.perf_counter()
        if world > 1:
            n0 = state[0][0].size
            kw = dict(with_index=False, root_energy=False, tiebreak_index=inter)
            multigpu.combine_winners(e[:n0], lp[:3 * n0], ix[:n0], dist.group.WORLD, ops, **kw)
            multigpu.combine_winners(e[n0:], lp[3 * n0:], ix[n0:], dist.group.WORLD, ops, **kw)
        t2 = time.perf_counter()
        if rank == 0:   # collapse from the combined payloads
            off = 0
            for lv in range(so.levels):
                n = so.best_e[lv].size
                so.best_lap[lv][...] = lp[3 * off:3 * (off + n)].numpy().reshape(so.best_lap[lv].shape)
                off += n
            nb = hb * wb
            be, bd = lp[3 * off:3 * (off + nb)].numpy(), lp[3 * (off + nb):3 * (off + 2 * nb)].numpy()
            so.fused_base = lambda: (((0.0 + be) + bd) / 2.0).astype(np.float32).reshape(hb, wb, 3)
            out = so.finish()
        phase["compute"] += t1 - t0
        phase["combine"] += t2 - t1
        phase["collapse"] += time.perf_counter() - t2

    for _ in range(args.warmup):
        step()
    phase = {k: 0.0 for k in phase}
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    dt_s = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt_s], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_s = float(t.item())
    if rank == 0:
        whole = orc.StreamingOracle(H, W, np.uint8, min_size=16, arith=args.arith)
        for f in range(total):
            whole.push_frame(orc.synth_frame_numpy(H, W, f, total))
        line = {"metric": "Mpixels/s fused (pyramid build+select+collapse)", "value": total * H * W * args.steps / dt_s / 1e6,
                "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": dt_s / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
                "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                "config": {"workload": f"DRY RUN on CPU (oracle + gloo): {total}x{W}x{H}x3 u8 frames", "frames_per_gpu": F,
                           "arith": args.arith, "shards": args.shards if world > 1 else None,
                           "parallelism": f"{total} frames, {F} per rank ({args.shards if world > 1 else 'one rank'}) over {world} rank(s)"},
                "roofline": None,
                "breakdown_ms_per_step": {f"{k}_ms_host": v / args.steps * 1e3 for k, v in phase.items()},
                "verified": bool(np.array_equal(out, whole.finish())), "dry_run": True}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run")
        args.gpus = world
    if args.dry_run:
        return dry_run(args, rank, world)

    dist = None
    # the host driver only supports dmabuf IPC: without this RCCL's buffer sharing across processes fails
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    force_dist = os.environ.get("MI_BENCH_FORCE_COMBINE") == "1" or args.force_combine  # exercise the combine at world 1
    backend = os.environ.get("MI_BENCH_BACKEND", "nccl")
    # MI_BENCH_ONE_GPU=1: every rank on device 0, collectives staged through the host over gloo (multigpu.HostStagedComm):
    # the whole multi-rank flow -- sharding, device combine kernels, rank 0's collapse, verification -- on a box with ONE GPU
    # (RCCL refuses two ranks on one device).  A functional run, not a scaling measurement.
    one_gpu = os.environ.get("MI_BENCH_ONE_GPU") == "1"
    if one_gpu:
        backend, local_rank = "gloo", 0
    if world > 1 or force_dist:
        # torch ships its own HIP runtime: it must be loaded BEFORE libmi355stack.so pulls in
        # /opt/rocm's, otherwise the process holds two runtimes and torch sees no GPU
        import torch
        import torch.distributed as dist
        torch.cuda.init()

    from shinestacker_amd import _lib as L
    from shinestacker_amd import build
    build.build_extension()
    L.require_device()

    if world > 1 or force_dist:
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    device = local_rank

    dt = {"u8": np.uint8, "u16": np.uint16, "f32": np.float32}[args.dtype]
    H, W = args.height, args.width
    if args.scaling == "strong":
        if args.frames % world:
            raise SystemExit(f"--scaling strong: {args.frames} frames do not split over {world} ranks")
        F, total_frames = args.frames // world, args.frames
    else:
        F, total_frames = args.frames, args.frames * world
    per = H * W * 3 * np.dtype(dt).itemsize
    buf = L.DeviceBuffer(per * F, device)
    inter = args.shards == "interleaved" and world > 1
    if inter:   # rank r: frames r, r + world, ... of the stack
        L.synth_frames_device(buf.ptr, dt, H, W, rank, F, total_frames, device=device, frame_step=world)
    else:
        L.synth_frames_device(buf.ptr, dt, H, W, rank * F, F, total_frames, device=device)

    def index_first(st, F=F, step=None):   # the global frame index of the handle's k-th frame
        step = (world if inter else 1) if step is None else step
        st.set_first_index(rank if step > 1 else rank * F, step)
    impl = {"auto": L.IMPL_AUTO, "simple": L.IMPL_SIMPLE, "tiled": L.IMPL_TILED}[args.impl]
    out_dt = np.uint16 if args.dtype == "u16" else np.uint8

    host_frames = None
    if args.source != "device":
        nh = min(F, 8)  # a few distinct host frames, cycled
        host_frames = [buf.download((H, W, 3), dt, offset=i * per) for i in range(nh)]
        if args.source == "host-pinned":   # the caller's frames lie in pinned memory: uploaded without the bounce copy
            pinned = [L.host_alloc((H, W, 3), dt) for _ in host_frames]
            for a, b in zip(pinned, host_frames):
                a[...] = b
            host_frames = pinned

    def barrier(st):
        if world > 1 or force_dist:
            import torch
            dist.barrier()
            torch.cuda.synchronize()
        st.sync()

    def measure(arith, steps, warmup, buf=buf, dt=dt, out_dt=out_dt, F=F, index_step=None):
        st = L.Stack(H, W, in_dtype=dt, out_dtype=out_dt, device=device, impl=impl, batch_frames=args.batch, arith=arith)
        index_first(st, F, index_step)
        combiner = None
        if world > 1 or force_dist:
            from shinestacker_amd import multigpu
            combiner = multigpu.Combiner(st, dist.group.WORLD, force=world == 1,
                                         comm=multigpu.HostStagedComm(dist.group.WORLD) if backend != "nccl" and world > 1 else None)
        phase = {"compute": 0.0, "combine": 0.0, "collapse": 0.0}
        cphase = {}

        def step(timed=False):
            st.reset()
            index_first(st, F, index_step)
            t0 = time.perf_counter()
            if host_frames is not None:
                for i in range(F):
                    st.push_frame(host_frames[i % len(host_frames)], zero_copy=args.source == "host-pinned")
            else:
                st.push_frames_device(buf.ptr, F)
            if combiner is not None:
                # the exchange needs this rank's level-0 state complete: that wait is the compute time of the step
                # (the coarser levels of the last batch overlap the level-0 exchange)
                st.sync_level(0)
                t1 = time.perf_counter()
                combiner.combine_winners(with_index=False, root_energy=False)
                t2 = time.perf_counter()
                if rank == 0:
                    st.finish_device()
                    st.sync()
                t3 = time.perf_counter()
                if timed:
                    phase["compute"] += t1 - t0
                    phase["combine"] += t2 - t1
                    phase["collapse"] += t3 - t2
                    for k, v in combiner.timings.items():
                        cphase[k] = cphase.get(k, 0.0) + v
            else:
                st.finish_device()

        for _ in range(warmup):
            step()
        if warmup == 0:
            # --warmup 0: the handle's per-batch buffers (tens of GB of hipMalloc) and the kernels' code objects are still
            # set-up, not a step -- one pass outside the timed region creates them
            step()
        barrier(st)
        st.profile(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            step(timed=True)
        barrier(st)
        dt_s = time.perf_counter() - t0
        if world > 1 or force_dist:
            import torch
            t = torch.tensor([dt_s], device=f"cuda:{local_rank}" if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_s = float(t.item())
        prof = {k: st.profile_get(v) for k, v in (("level0", L.PROF_LEVEL0), ("levels", L.PROF_LEVEL),
                                                  ("base", L.PROF_BASE), ("collapse", L.PROF_COLLAPSE))}
        phase.update({"combine_" + k[:-3]: v * 1e-3 for k, v in cphase.items()})   # "..._ms" -> seconds, like the others
        return st, dt_s, prof, phase

    st, dt_s, prof, phase = measure(args.arith, args.steps, args.warmup)
    tiled = prof["level0"][1] > 0
    ms_level, n_level, bytes_level = prof["level0"] if tiled else prof["levels"]
    # read level 0 once + 36 B per pixel of every coarser Gaussian level
    job_bytes_per_frame = float(np.dtype(dt).itemsize * 3 * H * W +
                                36 * sum(h * w for (h, w) in st.shapes[1:]))

    def roofline(ms_level, n_level, bytes_level):
        return (bytes_level / n_level) / (ms_level / n_level * 1e-3) / 1e9 if n_level else 0.0

    if rank == 0:
        ms_per_step = dt_s / args.steps * 1e3
        value = total_frames * H * W * args.steps / dt_s / 1e6
        achieved = roofline(ms_level, n_level, bytes_level)
        # roofline.traffic: PMC bytes per launch of the dominant kernel from the separate --pmc passes of tools/profile.sh
        # (profiles/traffic.json).  The entry names the kernel sources it was measured on (sha256) and the workload; a
        # number measured on other sources or another input type is refused (null + the reason), never silently reused.
        traffic, traffic_note = None, None
        try:
            with open(args.traffic_json) as fh:
                tj = json.load(fh)
            ent = tj.get(args.arith, {})
            if ent.get("source_sha") != kernel_source_sha():
                traffic_note = (f"profiles/traffic.json[{args.arith}] was measured on kernel sources {ent.get('source_sha')}, "
                                f"these are {kernel_source_sha()}: re-run tools/profile_r06.sh (its last step writes profiles/traffic.json)")
            elif ent.get("dtype", "f32") != args.dtype or ent.get("frames_per_launch") is None:
                traffic_note = f"profiles/traffic.json[{args.arith}] is for dtype {ent.get('dtype')}, this run is {args.dtype}"
            else:
                traffic = ent.get("hbm_bytes_per_launch")
                traffic_note = f"{ent.get('kernel')}; {ent.get('dispatches')} dispatches; {ent.get('note', '')}"
        except (OSError, AttributeError, ValueError) as e:
            traffic_note = f"profiles/traffic.json unreadable: {e}"
        pair_on = (args.arith == "separable" and args.dtype == "f32" and F >= 192 and st.levels >= 2
                   and os.environ.get("SHINESTACKER_AMD_PAIR_LEVELS", "0") in ("0", "1")) or \
                  (args.arith == "separable" and st.levels >= 2 and os.environ.get("SHINESTACKER_AMD_PAIR_LEVELS") == "1")
        kernel = {"separable": ("level_sep_pair<level 0> (levels 0 / 1 as a pair: stage + separable reduce + gray Laplacian + separable "
                                "energy + select of level 0, AND gray(G_1) + the tile's pixels of G_2 for level 1; one launch = 16 "
                                "frames of the resident push; `achieved` counts level 0's algorithmic bytes only)" if pair_on else
                                "level_sep<level 0> (stage + separable reduce + gray Laplacian + separable energy + "
                                "select; one launch = 16 frames of the resident push)"),
                  "exact": "level_fused<level 0> (stage+reduce+laplacian+energy+select, one launch per frame batch)"}
        breakdown = {k: v[0] / args.steps for k, v in prof.items()}
        if world > 1 or force_dist:
            breakdown.update({f"{k}_ms_host": v / args.steps * 1e3 for k, v in phase.items()})
            # combine_ms: host time of the exchange proper per step (both phases; the waits for this rank's own levels are
            # listed beside it: wait_rest ~ 0 means the coarser levels were hidden behind the level-0 exchange)
            line_combine_ms = (phase.get("combine_exchange_level0", 0.0) + phase.get("combine_exchange_rest", 0.0)) / args.steps * 1e3
        line = {
            "metric": "Mpixels/s fused (pyramid build+select+collapse)",
            "value": value, "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "setup_pass_outside_timing": args.warmup == 0, "ms_per_step": ms_per_step,
            # (one GPU: nothing is scaled -- the key stays for the driver's parser, the value says so)
            "higher_is_better": True, "scaling": args.scaling if world > 1 else None, "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"{total_frames}x{W}x{H}x3 {args.dtype} frames "
                                   f"{'resident in HBM' if args.source == 'device' else 'pushed from ' + ('pinned ' if args.source == 'host-pinned' else '') + 'host memory (PCIe inside the timed region)'}, "
                                   f"{st.levels}-level Laplacian pyramid fusion "
                                   f"(BASELINE.json configs[{1 if world == 1 or args.scaling == 'weak' else 2}])",
                       "frames_per_gpu": F, "source": args.source, "arith": args.arith,
                       "pair_levels": ("levels 0 / 1 run as a pair (mi_stack_params.pair_levels, automatic: float-32 batches of 192 "
                                       "frames and more): G_1 never through HBM, same bits" if pair_on else
                                       "off for this run (level-by-level kernels)") if args.arith == "separable" else None,
                       "arith_parity": ("separable: bit-identical to oracle/separable_oracle.c (its specification); against the "
                                        "reference's evaluation order the stated bound is 32 u maxv per convolution (u = 2^-24; "
                                        "SURVEY 7), not north_star's 1 ULP: measured Gaussian / energy differences <= 8 % / 1.5 % of "
                                        "it, arg-max flips only at proven near ties (other_mode.parity)"
                                        if args.arith == "separable" else
                                        "exact: the reference's row-major 25-tap order, bit-identical to the restatement the "
                                        "reference's own recordings pin (cv2.filter2D's real order: unpinned third party)"),
                       "impl": args.impl if args.impl != "auto" else ["auto", "simple", "tiled"][st.params.impl],
                       "device": L.device_name(device),
                       "shards": args.shards if world > 1 else None,
                       "parallelism": f"{total_frames} frames, {F} per rank" + (f" ({args.shards}: rank r holds " + ("frames r, r + " + str(world) + ", ..." if inter else "the block [r F, (r + 1) F)") + ")" if world > 1 else "") + f" over {world} " +
                                      ("rank(s) SHARING ONE GPU (functional run, collectives over gloo through the host)"
                                       if one_gpu else "GPU(s)")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                         "kernel": kernel[args.arith] if tiled else "simple impl: all level kernels of one frame",
                         "algorithmic_bytes_per_launch": bytes_level / max(n_level, 1),
                         "avg_launch_ms": ms_level / max(n_level, 1), "launches": n_level},
            "breakdown_ms_per_step": breakdown,
            "job_roofline_frac": (job_bytes_per_frame * total_frames * args.steps / dt_s)
                                 / (HBM_PEAK_GBS * 1e9 * world),
        }
        if world > 1 or force_dist:
            line["combine_ms"] = line_combine_ms
            line["combine_note"] = ("world 1, --force-combine: the per-rank kernel work of the exchange on this rank's own "
                                    "rows, nothing crosses a link" if world == 1 else
                                    "winners-only exchange over RCCL: level 0 while the coarser levels still run, then the rest")
        if not args.no_verify:
            v = verify(L, st, args, total_frames, world)
            line["verified"] = v.pop("ok")
            line["verify"] = v
    do_other = world == 1 and not force_dist and not args.no_other_mode and args.source == "device"
    fused_main = st.finish() if do_other and not args.no_verify else None   # outside the timed region
    st.close()
    if do_other:
        other = "exact" if args.arith == "separable" else "separable"
        k2 = max(1, args.steps)   # the same K and W as the headline mode: a step is 30-40 ms
        st2, dt2, prof2, _ = measure(other, k2, max(1, args.warmup))
        ms2, n2, b2 = prof2["level0"] if prof2["level0"][1] > 0 else prof2["levels"]
        line["other_mode"] = {"arith": other, "value": total_frames * H * W * k2 / dt2 / 1e6, "unit": "Mpixels/s",
                              "steps": k2, "ms_per_step": dt2 / k2 * 1e3,
                              "roofline_frac": roofline(ms2, n2, b2) / HBM_PEAK_GBS,
                              "job_roofline_frac": (job_bytes_per_frame * total_frames * k2 / dt2) / (HBM_PEAK_GBS * 1e9)}
        if fused_main is not None:
            # SURVEY 8(d) "parity reporting" between the two arithmetics on the SAME full-size stack (outside the timed
            # region): per-level Gaussian / energy differences against the stated bounds, selection-mismatch rates, a
            # near-tie proof for every flipped arg-max, the final-image histogram with every value off by >= 2 counts
            # traced to a flip (tools/parity_report.py)
            st2.close()
            st2 = None
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import parity_report
            pr = parity_report.report(L, buf.ptr, F, H, W, dt, device=device)
            line["other_mode"]["parity"] = pr
            hist = pr["final_abs_diff_counts_0_1_2_3plus"]
            line["other_mode"]["fused_image_abs_diff_counts_0_1_2_3plus"] = hist
            line["other_mode"]["fused_image_values_differing"] = float(sum(hist[1:])) / float(sum(hist))
            line["other_mode"]["fused_image_max_abs_diff"] = pr["final_max_abs_diff"]
        if st2 is not None:
            st2.close()
    extras = world == 1 and not force_dist and args.source == "device" and args.impl != "simple"
    if extras and not args.no_projection and total_frames >= 16:
        line["projected_scaling"] = project_scaling(L, measure, args, H, W, total_frames, dt_s / args.steps, device,
                                                    job_bytes_per_frame)
        line["shard_step"] = line["projected_scaling"].pop("shard_step")
    if extras and not args.no_other_dtypes:
        # the reference's real input types (pyramid.py:159-164: uint8 / uint16 files): the same stack, same values, held as
        # integer frames -- 3 / 6 bytes per pixel cross HBM instead of 12, and the level-0 kernel converts on the fly
        line["other_dtypes"] = {}
        buf.free()
        buf = None
        for name, d2 in (("u8", np.uint8), ("u16", np.uint16), ("f32", np.float32)):
            if name == args.dtype:
                continue
            per2 = H * W * 3 * np.dtype(d2).itemsize
            b2 = L.DeviceBuffer(per2 * F, device)
            L.synth_frames_device(b2.ptr, d2, H, W, 0, F, total_frames, device=device)
            st3, dt3, prof3, _ = measure(args.arith, args.steps, max(1, args.warmup), buf=b2, dt=d2,
                                         out_dt=np.uint16 if name == "u16" else np.uint8)
            ms3, n3, by3 = prof3["level0"] if prof3["level0"][1] > 0 else prof3["levels"]
            jb = float(np.dtype(d2).itemsize * 3 * H * W + 36 * sum(h * w for (h, w) in st3.shapes[1:]))
            line["other_dtypes"][name] = {
                "value": total_frames * H * W * args.steps / dt3 / 1e6, "unit": "Mpixels/s", "ms_per_step": dt3 / args.steps * 1e3,
                "job_roofline_frac": (jb * total_frames * args.steps / dt3) / (HBM_PEAK_GBS * 1e9),
                "roofline_frac": roofline(ms3, n3, by3) / HBM_PEAK_GBS, "avg_launch_ms": ms3 / max(n3, 1),
                "algorithmic_bytes_per_launch": by3 / max(n3, 1), "arith": args.arith}
            st3.close()
            b2.free()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, total_frames)
            line["cv2"] = line["cpu_baseline"].pop("cv2")   # is OpenCV importable on this box, and if so: do the restatements hold?
        if world > 1 or force_dist:
            # RCCL prints its version banner through C stdio when NCCL_DEBUG is set (it is, on the GPU boxes): push
            # that out first, so that the JSON line is the last line of rank 0's stdout
            import ctypes
            ctypes.CDLL(None).fflush(None)
        print(json.dumps(line), flush=True)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
