#!/usr/bin/env python3
"""BASELINE config 5 on one GPU at reduced length: 16-bit 50 MP frames in HOST memory, stacked in
bunches (FocusStackBunch geometry: frames=10, overlap=2, stack.py:61-64) through the pinned asynchronous
upload path of mi_stack_push_frame -- PCIe, the bounce copy and the kernels overlap.  One stacker handle
serves every bunch (reset between bunches), as FocusStackBunch does.

Several GPUs (SURVEY 8(e) bunch mode, no collective in the data path): start it with
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/config5.py
Every rank fuses a contiguous block of the bunches on GPU LOCAL_RANK (actions.shard_steps, what
FocusStackBunch(shard='env') does); a gloo group is used for the two barriers and the max-over-ranks time only."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def two_stage(args):
    """config 5 is H2D-bound -- report OVERLAP EFFICIENCY = max(upload, compute) / wall:
      upload_s   the pushed bytes at this box's own pinned hipMemcpy rate (measured here: one 50 MP frame, 5 copies);
      compute_s  stage 1 with the same bunches taken from frames RESIDENT in HBM (no PCIe at all);
      wall_s     stage 1 as it runs: frames in host memory, upload + kernels overlapped.
    `--pinned` (default): the host frames lie in pinned memory (mi_host_alloc -- what a decoder writing into
    `_lib.host_alloc` arrays gives) and are uploaded without the bounce copy; `--pageable`: ordinary NumPy arrays through the
    three pinned bounce buffers and the copy-thread pool."""
    from shinestacker_amd import _lib as L
    from shinestacker_amd.pipeline import bunches_then_stack
    N, H, W = args.frames, args.height, args.width
    per = H * W * 3 * 2
    ndist = 8   # distinct host frames, cycled (a real job decodes files here; 130 distinct 50 MP frames are 39 GB)
    buf = L.DeviceBuffer(per * ndist)
    L.synth_frames_device(buf.ptr, np.uint16, H, W, 0, ndist, ndist)
    host = []
    for i in range(ndist):
        fr = buf.download((H, W, 3), np.uint16, offset=i * per)
        if not args.pageable:
            pin = L.host_alloc((H, W, 3), np.uint16)
            pin[...] = fr
            fr = pin
        host.append(fr)
    # this box's pinned host-to-device rate: the ceiling of any upload path
    pin0 = host[0] if not args.pageable else L.host_alloc((H, W, 3), np.uint16)
    # ... measured through the library's own copy stream (asynchronous copies out of pinned memory, ten frames back to back,
    # the best of three rounds; a synchronous hipMemcpy read 29 or 57 GB/s from run to run on the same box)
    cal = L.Stack(H, W, in_dtype=np.uint16, out_dtype=np.uint16)
    rates = []
    for _ in range(3):
        cal.reset()
        t0 = time.perf_counter()
        for _k in range(10):
            cal.push_frame(pin0, zero_copy=True)
        cal.wait_uploads(0)
        rates.append(10 * per / (time.perf_counter() - t0))
    cal.close()
    pinned_rate = max(rates)
    out = L.DeviceBuffer(per)
    nst = 2 if args.one_handle else 3     # stage 1 alternates between two handles unless --one-handle (the round-3 flow)
    stacks = tuple(L.Stack(H, W, in_dtype=np.uint16, out_dtype=np.uint16) for _ in range(nst))

    # the buffer of the bunch results is made once, like the handles (a job of many stacks keeps it): allocating and freeing
    # it costs ~50 ms per GB, 2 s for the 38 GB of a 1024-frame job, and is reported separately
    from shinestacker_amd.actions import get_bunches
    nbunch = len(get_bunches(list(range(N)), 10, 2))
    t0 = time.perf_counter()
    results = L.DeviceBuffer(per * nbunch)
    L.check(L.load().mi_device_synchronize(0))
    results_alloc_s = time.perf_counter() - t0

    def run():
        info = {}
        _, bunches = bunches_then_stack(lambda i: host[i % ndist], N, H, W, np.uint16, out_dev=out.ptr,
                                        stacks=stacks, results_buf=results, info=info, zero_copy=not args.pageable)
        return bunches, info["stage1_s"], info["stage2_s"]
    run()
    bunches, s1, s2 = run()
    pushed = sum(len(b) for b in bunches)
    # stage 1 with the frames resident: the compute side alone
    st = stacks[0]

    def resident():
        for b in bunches:
            st.reset()
            for i in b:
                st.push_frames_device(buf.ptr + (i % ndist) * per, 1, per)
            st.finish_device(out.ptr)
        st.sync()
    resident()
    t0 = time.perf_counter()
    resident()
    compute_s = time.perf_counter() - t0
    upload_s = pushed * per / pinned_rate
    print(json.dumps({"config": f"{N} x {W}x{H} u16 frames from {'pageable' if args.pageable else 'PINNED'} host memory -> "
                                f"{len(bunches)} bunches of <= 10 (overlap 2) -> one stack over the {len(bunches)} bunch results "
                                f"(resident, uint16), one GPU",
                      "frames_pushed_stage1": pushed, "stage1_seconds": s1, "stage2_seconds": s2, "seconds": s1 + s2,
                      "stage1_Mpixels_per_s": pushed * H * W / s1 / 1e6, "stage2_Mpixels_per_s": len(bunches) * H * W / s2 / 1e6,
                      "host_to_device_GB_per_s": pushed * per / s1 / 1e9,
                      "pinned_memcpy_GB_per_s": pinned_rate / 1e9, "upload_s": upload_s, "compute_s": compute_s, "wall_s": s1,
                      "overlap_efficiency": max(upload_s, compute_s) / s1,
                      "results_buffer_GB": per * nbunch / 1e9, "results_alloc_s": results_alloc_s,
                      "stage1_handles": nst - 1, "upload_path": "bounce copy (3 pinned buffers, copy-thread pool)" if args.pageable else
                                     "zero-copy from pinned frames (mi_stack_push_frame_pinned)"}))
    for s_ in stacks:
        s_.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--height", type=int, default=5792)
    ap.add_argument("--width", type=int, default=8640)
    ap.add_argument("--two-stage", action="store_true",
                    help="the whole config-5 flow on this GPU: bunches from host memory, then the bunch results -- kept on "
                         "the device, truncated to uint16 as the reference's intermediate files are -- fused once more "
                         "(pipeline.bunches_then_stack)")
    ap.add_argument("--one-handle", action="store_true", help="stage 1 on a single handle (no overlap across bunches)")
    ap.add_argument("--pageable", action="store_true", help="host frames in ordinary (pageable) memory: the bounce-copy path")
    args = ap.parse_args()
    if args.two_stage:
        return two_stage(args)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = int(os.environ.get("MI_TOOL_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from shinestacker_amd import _lib as L
    from shinestacker_amd.actions import get_bunches, shard_steps
    N, H, W = args.frames, args.height, args.width
    per = H * W * 3 * 2
    ndist = 8   # distinct host frames, cycled
    buf = L.DeviceBuffer(per * ndist, dev)
    L.synth_frames_device(buf.ptr, np.uint16, H, W, 0, ndist, ndist, device=dev)
    host = [buf.download((H, W, 3), np.uint16, offset=i * per) for i in range(ndist)]
    all_bunches = get_bunches(list(range(N)), 10, 2)
    bunches = [all_bunches[i] for i in shard_steps(len(all_bunches), rank, world)]
    st = L.Stack(H, W, in_dtype=np.uint16, out_dtype=np.uint16, device=dev)
    out = L.DeviceBuffer(per, dev)

    def run():
        pushed = 0
        for b in bunches:
            st.reset()
            for f in b:
                st.push_frame(host[f % ndist], zero_copy=not args.pageable)
                pushed += 1
            st.finish_device(out.ptr)
        st.sync()
        return pushed
    run()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    pushed = run()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt, float(pushed)], dtype=torch.float64)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, pushed = float(tmax[0]), int(t[1])
        dist.barrier()
    if rank == 0:
        print(json.dumps({"config": f"{N} x {W}x{H} u16 frames from host memory, {len(all_bunches)} bunches of <= 10 "
                                    f"(overlap 2) over {world} process(es)",
                          "frames_pushed": pushed, "seconds": dt, "Mpixels_per_s": pushed * H * W / dt / 1e6,
                          "host_to_device_GB_per_s": pushed * per / dt / 1e9}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
