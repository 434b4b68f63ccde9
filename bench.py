#!/usr/bin/env python3
"""bench.py -- Mpixels/s fused (pyramid build + select + collapse) on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N>1 is launched by torch.distributed.run, one rank per GPU (RCCL).
A "step" is one whole focus-stack job over a device-resident synthetic stack:
push all frames (pyramid build + per-level selection), base fusion, cross-GPU
combine (N>1), collapse, abs/clip/truncating cast -- result left in HBM.
Workload at N=1: BASELINE.json configs[1] -- 256 x 24 MP (4000x6000x3) fp32
frames, 6 Laplacian levels + 63x94 base.  N>1: weak scaling, every rank holds
its own 256-frame shard of a 256*N-frame stack (contiguous global indices).

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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=256, help="frames per GPU")
    ap.add_argument("--height", type=int, default=4000)
    ap.add_argument("--width", type=int, default=6000)
    ap.add_argument("--dtype", default="f32", choices=["u8", "u16", "f32"])
    ap.add_argument("--impl", default="auto", choices=["auto", "simple", "tiled"])
    ap.add_argument("--batch", type=int, default=0, help="frames per fused launch (0 = library default)")
    ap.add_argument("--source", default="device", choices=["device", "host"],
                    help="host: frames are pushed from host memory one by one (PCIe-inclusive rate; "
                         "never the headline value)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=64,
                    help="frames of the CPU-baseline sample (about 10 s of CPU work at 24 MP on 16 cores)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic.json"),
                    help="per-launch HBM bytes from the PMC passes (tools/pmc_traffic.py)")
    return ap.parse_args()


def cpu_baseline(args):
    """The oracle's streaming C port, timed on this host's cores on a bounded sample of
    the same workload (same generator, same geometry, fewer frames)."""
    from oracle import oracle as orc
    orc.build()
    n = min(args.cpu_frames, args.frames)
    H, W = args.height, args.width

    def frame(f):   # generated one at a time, outside the timed sections
        fr = orc.synth_frame_u8(H, W, f, args.frames)
        return fr.astype(np.uint16) * 257 if args.dtype == "u16" else fr
    first = frame(0)
    so = orc.StreamingOracle(H, W, first.dtype, keep_gauss=False)
    dt = 0.0
    for f in range(n):
        fr = first if f == 0 else frame(f)
        t0 = time.perf_counter()
        so.push_frame(fr)
        dt += time.perf_counter() - t0
    t0 = time.perf_counter()
    so.finish()
    dt += time.perf_counter() - t0
    frames = [first]
    return {"value": n * H * W / dt / 1e6, "unit": "Mpixels/s", "cores": orc.lib().orc_num_threads(),
            "kind": "port",
            "sample": f"{n} of {args.frames} frames of {W}x{H} (same generator, values as "
                      f"{frames[0].dtype}), oracle.StreamingOracle push+finish, {dt:.1f} s"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run")
        args.gpus = world

    dist = None
    # the host driver only supports dmabuf IPC: without this RCCL's buffer sharing across processes fails
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    force_dist = os.environ.get("MI_BENCH_FORCE_COMBINE") == "1"  # exercise the combine at world 1
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
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    device = local_rank

    dt = {"u8": np.uint8, "u16": np.uint16, "f32": np.float32}[args.dtype]
    H, W, F = args.height, args.width, args.frames
    per = H * W * 3 * np.dtype(dt).itemsize
    total_frames = F * world
    buf = L.DeviceBuffer(per * F, device)
    L.synth_frames_device(buf.ptr, dt, H, W, rank * F, F, total_frames, device=device)
    impl = {"auto": L.IMPL_AUTO, "simple": L.IMPL_SIMPLE, "tiled": L.IMPL_TILED}[args.impl]
    st = L.Stack(H, W, in_dtype=dt, out_dtype=np.uint16 if args.dtype == "u16" else np.uint8,
                 device=device, impl=impl, batch_frames=args.batch)
    st.set_first_index(rank * F)
    combiner = None
    if world > 1 or force_dist:
        from shinestacker_amd import multigpu
        combiner = multigpu.Combiner(st, dist.group.WORLD)

    def barrier():
        if world > 1 or force_dist:
            import torch
            dist.barrier()
            torch.cuda.synchronize()
        st.sync()

    host_frames = None
    if args.source == "host":
        nh = min(F, 8)  # a few distinct host frames, cycled
        host_frames = [buf.download((H, W, 3), dt, offset=i * per) for i in range(nh)]

    def step():
        st.reset()
        st.set_first_index(rank * F)
        if host_frames is not None:
            for i in range(F):
                st.push_frame(host_frames[i % len(host_frames)])
        else:
            st.push_frames_device(buf.ptr, F)
        if combiner is not None:
            # arg-max-with-payload exchange over xGMI; the fused image needs the winners' payloads only, so the
            # winner indices are not exchanged and the winners' energies stay on the chunk owners
            combiner.combine(with_index=False, root_energy=False)
            if rank == 0:
                st.finish_device()
        else:
            st.finish_device()

    for _ in range(args.warmup):
        step()
    barrier()
    st.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt_s = time.perf_counter() - t0
    if world > 1 or force_dist:
        import torch
        t = torch.tensor([dt_s], device=f"cuda:{local_rank}", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_s = float(t.item())

    prof = {k: st.profile_get(v) for k, v in (("level0", L.PROF_LEVEL0), ("levels", L.PROF_LEVEL),
                                              ("base", L.PROF_BASE), ("collapse", L.PROF_COLLAPSE))}
    tiled = prof["level0"][1] > 0
    ms_level, n_level, bytes_level = prof["level0"] if tiled else prof["levels"]
    # read level 0 once + 36 B per pixel of every coarser Gaussian level
    job_bytes_per_frame = float(np.dtype(dt).itemsize * 3 * H * W +
                                36 * sum(h * w for (h, w) in st.shapes[1:]))
    if rank == 0:
        ms_per_step = dt_s / args.steps * 1e3
        value = total_frames * H * W * args.steps / dt_s / 1e6
        achieved = (bytes_level / n_level) / (ms_level / n_level * 1e-3) / 1e9 if n_level else 0.0
        traffic = None
        try:
            with open(args.traffic_json) as fh:
                traffic = json.load(fh).get("hbm_bytes_per_launch")
        except OSError:
            pass
        line = {
            "metric": "Mpixels/s fused (pyramid build+select+collapse)",
            "value": value, "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{total_frames}x{W}x{H}x3 {args.dtype} frames resident in "
                                   f"HBM, {st.levels}-level Laplacian pyramid fusion "
                                   f"(BASELINE.json configs[1])",
                       "frames_per_gpu": F, "source": args.source, "impl": args.impl if args.impl != "auto" else ["auto", "simple", "tiled"][st.params.impl],
                       "device": L.device_name(device),
                       "parallelism": f"frames sharded over {world} GPU(s)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": ("level_fused<level 0> (stage+reduce+laplacian+energy+select, "
                                    "one launch per frame batch)") if tiled else
                                   "simple impl: all level kernels of one frame",
                         "algorithmic_bytes_per_launch": bytes_level / max(n_level, 1),
                         "avg_launch_ms": ms_level / max(n_level, 1), "launches": n_level},
            "breakdown_ms_per_step": {k: v[0] / args.steps for k, v in prof.items()},
            "job_roofline_frac": (job_bytes_per_frame * total_frames * args.steps / dt_s)
                                 / (HBM_PEAK_GBS * 1e9 * world),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
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
