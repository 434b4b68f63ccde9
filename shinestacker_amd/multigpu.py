"""Frame-sharded multi-GPU stacking: the cross-GPU combine of the per-level selection state.

Frames of one stack are split into contiguous blocks, one block per rank (one process per
GPU, `torch.distributed`, backend "nccl" = RCCL over xGMI).  Every rank runs the ordinary
single-GPU path on its block (`Stack.set_first_index(first global frame)`); what remains is an
arg-max-with-payload reduction of the running state `(E, idx, lap)` of every level and of the
two base-level features -- the multi-GPU form of `np.argmax(energies, axis=0)` with first-max
tie-breaking (reference algorithms/pyramid.py:48-55, :103-110).

No stock collective has that operator, and a ring all-reduce of the ~640 MB of state would be
per-link bound on xGMI.  The exchange is therefore a pixel-domain reduce-scatter over the full
point-to-point mesh:

  1. all_to_all_single  -- rank r receives, from every rank, the r-th pixel chunk of (E, idx, lap)
                           (7 of 8 chunks travel, each over its own xGMI link, in parallel);
  2. local first-max    -- candidates are ordered by rank = by global frame index, strict '>'
                           keeps the first maximum (HIP kernel `mi_combine_select`);
  3. send/recv          -- the winners' chunks go to rank 0, which owns the collapse.

The state of all levels (and of the two base features) is exchanged as ONE flat pixel vector per
array -- 3 all-to-all + 3 gather-to-root collectives per stack in total (`combine_all`), a handful of
large transfers instead of some 190 small ones (per level: 3 all-to-all + 21 point-to-point at 8 GPUs).

Everything here is host-side plumbing on torch tensors; the same function runs on CPU tensors
under the "gloo" backend in tests/ (with a torch implementation of step 2 injected there).
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib


def chunk_bounds(n, world):
    """Pixel range owned by every rank: contiguous, sizes differ by at most one chunk tail."""
    per = -(-n // world)
    return [(min(r * per, n), min((r + 1) * per, n)) for r in range(world)]


def combine_state(energy, lap, index, group, select_fn, width=3):
    """In-place combine of one level's state across `group`; rank 0 ends up with the result.

    energy: (n,) f32, lap: (n*width,) f32, index: (n,) i32 -- this rank's running state.
    select_fn(cand_e [W,m], cand_lap [W,m*width], cand_idx [W,m]) -> (e [m], lap [m*width], idx [m])
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = energy.numel()
    bounds = chunk_bounds(n, world)
    sizes = [b - a for a, b in bounds]
    mine = sizes[rank]

    def exchange(t, w):
        out = torch.empty(world * mine * w, dtype=t.dtype, device=t.device)
        dist.all_to_all_single(out, t, output_split_sizes=[mine * w] * world,
                               input_split_sizes=[s * w for s in sizes], group=group)
        return out.view(world, mine * w)

    cand_e = exchange(energy, 1)
    cand_l = exchange(lap, width)
    cand_i = exchange(index, 1)
    win_e, win_l, win_i = select_fn(cand_e, cand_l, cand_i)
    # winners to rank 0
    if rank == 0:
        a, b = bounds[0]
        energy[a:b] = win_e
        lap[a * width:b * width] = win_l
        index[a:b] = win_i
        for r in range(1, world):
            a, b = bounds[r]
            if b > a:
                dist.recv(energy[a:b], src=dist.get_global_rank(group, r), group=group)
                dist.recv(lap[a * width:b * width], src=dist.get_global_rank(group, r), group=group)
                dist.recv(index[a:b], src=dist.get_global_rank(group, r), group=group)
    elif mine > 0:
        root = dist.get_global_rank(group, 0)
        dist.send(win_e.contiguous(), dst=root, group=group)
        dist.send(win_l.contiguous(), dst=root, group=group)
        dist.send(win_i.contiguous(), dst=root, group=group)


def combine_all(states, group, select_fn, width=3, with_index=True, root_energy=True):
    """Combine the state of all levels at once.  `states`: list of (energy (n_l,), lap (n_l*width,),
    index (n_l,)) tensors of this rank; on return rank 0's tensors hold the combined state.
    Same arithmetic as `combine_state` level by level (the pixel chunks just run across level
    boundaries), with 6 collectives in total -- 4 with `with_index=False`, which leaves the winner
    indices of rank 0 stale (they only feed the debug taps; the fused image needs E and lap alone)
    and moves 16 instead of 20 bytes per pixel.  `root_energy=False` also keeps the winners' energies on the
    chunk owners (3 collectives, 12 instead of 20 bytes per pixel to rank 0): the collapse reads the fused Laplacians
    and base pixels only, so the fused image is the same; rank 0's energy taps are then stale too."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    single = len(states) == 1   # the library's contiguous slabs: exchanged in place, no packing copies
    if single:
        e_all, l_all = states[0][0].reshape(-1), states[0][1].reshape(-1)
        i_all = states[0][2].reshape(-1) if with_index else None
    else:
        e_all = torch.cat([e.reshape(-1) for e, _, _ in states])
        l_all = torch.cat([lp.reshape(-1) for _, lp, _ in states])
        i_all = torch.cat([ix.reshape(-1) for _, _, ix in states]) if with_index else None
    n = e_all.numel()
    bounds = chunk_bounds(n, world)
    sizes = [b - a for a, b in bounds]
    mine = sizes[rank]

    def exchange(t, w):
        out = torch.empty(world * mine * w, dtype=t.dtype, device=t.device)
        dist.all_to_all_single(out, t, output_split_sizes=[mine * w] * world,
                               input_split_sizes=[s_ * w for s_ in sizes], group=group)
        return out.view(world, mine * w)

    win_e, win_l, win_i = select_fn(exchange(e_all, 1), exchange(l_all, width),
                                    exchange(i_all, 1) if with_index else None)

    def to_root(win, full, w):
        # every rank's winners to rank 0: an all-to-all in which only rank 0 receives
        out = full if rank == 0 else torch.empty(0, dtype=win.dtype, device=win.device)
        dist.all_to_all_single(out, win.contiguous(),
                               output_split_sizes=[s_ * w for s_ in sizes] if rank == 0 else [0] * world,
                               input_split_sizes=[mine * w] + [0] * (world - 1), group=group)

    if root_energy:
        to_root(win_e, e_all, 1)
    to_root(win_l, l_all, width)
    if with_index:
        to_root(win_i, i_all, 1)
    if rank == 0 and not single:
        off = 0
        for e, lp, ix in states:
            m = e.numel()
            if root_energy:
                e.reshape(-1).copy_(e_all[off:off + m])
            lp.reshape(-1).copy_(l_all[off * width:(off + m) * width])
            if with_index:
                ix.reshape(-1).copy_(i_all[off:off + m])
            off += m


def first_max_rank(cand_e, cand_i=None):
    """torch form of mi_combine_winner[_idx] (CPU tests / reference semantics): winning rank per pixel, strict '>' in rank
    order; with the candidates' global frame indices `cand_i` (interleaved shards: the rank order is not the frame order) a
    tie goes to the lower index -- np.argmax's first maximum"""
    world, m = cand_e.shape
    best = torch.zeros(m, dtype=torch.uint8, device=cand_e.device)
    be = cand_e[0].clone()
    bf = cand_i[0].clone() if cand_i is not None else None
    for r in range(1, world):
        win = cand_e[r] > be
        if cand_i is not None:
            win = win | ((cand_e[r] == be) & (cand_i[r] < bf))
            bf = torch.where(win, cand_i[r], bf)
        be = torch.where(win, cand_e[r], be)
        best = torch.where(win, torch.full_like(best, r), best)
    return best


class TorchWinnerOps:
    """The four local steps of `combine_winners` in plain torch (CPU tensors under gloo in tests/; on the GPU the
    Combiner uses the library's kernels: boolean-mask indexing of 32 Mpixel states costs milliseconds per call)."""

    def winner(self, cand_e, cand_i=None):
        return first_max_rank(cand_e, cand_i)

    def plan(self, win, world):
        return None, [int((win == r).sum()) for r in range(world)]

    def pack(self, win, plan, world, rank, arr, width, count):
        return arr.view(-1, width)[win == rank].reshape(-1).contiguous()

    def unpack(self, win, plan, world, rank, bufs, arr, width):
        rows = arr.view(-1, width)
        for r, b in enumerate(bufs):
            if r != rank and b is not None and b.numel():
                rows[win == r] = b.view(-1, width)


class DirectComm:
    """The collectives of `combine_winners` straight on the tensors handed in: RCCL for device tensors under the "nccl"
    backend (xGMI, one process per GPU), gloo for CPU tensors."""

    def __init__(self, group):
        self.group = group

    def all_to_all(self, out, inp, out_splits, in_splits):
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group)

    def all_gather(self, out, inp):
        dist.all_gather_into_tensor(out, inp, group=self.group)

    def send_to_root(self, tensors):
        root = dist.get_global_rank(self.group, 0)
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, b, root, self.group) for b in tensors]):
            req.wait()

    def recv_at_root(self, bufs_by_rank):
        ops_ = [dist.P2POp(dist.irecv, b, dist.get_global_rank(self.group, r), self.group) for r, b in bufs_by_rank]
        if ops_:
            for req in dist.batch_isend_irecv(ops_):
                req.wait()


class HostStagedComm(DirectComm):
    """The same collectives for DEVICE tensors over a backend that only moves host memory (gloo): every buffer is staged
    through the host.  This is the seam that lets `Combiner.combine_winners` -- the HIP kernels mi_combine_winner / plan /
    pack / unpack on device-resident state -- run with world > 1 on a box with ONE GPU (two processes sharing it,
    tests/test_gpu_combine.py); it is also a working, if slow, fallback for nodes without peer access."""

    def all_to_all(self, out, inp, out_splits, in_splits):
        o = torch.empty(out.shape, dtype=out.dtype)
        super().all_to_all(o, inp.cpu(), out_splits, in_splits)
        out.copy_(o)

    def all_gather(self, out, inp):
        o = torch.empty(out.shape, dtype=out.dtype)
        super().all_gather(o, inp.cpu())
        out.copy_(o)

    def send_to_root(self, tensors):
        super().send_to_root([t.cpu() for t in tensors])

    def recv_at_root(self, bufs_by_rank):
        host = [(r, torch.empty(b.shape, dtype=b.dtype)) for r, b in bufs_by_rank]
        super().recv_at_root(host)
        for (_, b), (_, h) in zip(bufs_by_rank, host):
            b.copy_(h)


def combine_winners(e_all, l_all, i_all, group, ops, width=3, with_index=True, root_energy=True, comm=None, force=False,
                    tiebreak_index=False):
    """Cross-rank first-max of a flat state (e_all (n,), l_all (n*width,), i_all (n,) or None), result on rank 0.
    Same outcome as `combine_all`, less traffic -- the payload crosses the fabric once, not twice:

      1. all-to-all of the ENERGIES by pixel chunk (4 B/pixel; 7 of 8 chunks travel, one per xGMI link);
      2. chunk owners name the winning rank of every pixel (first maximum in rank = frame order);
      3. all-gather of that map (1 B/pixel): every rank now knows which of its own pixels won;
      4. every rank packs the payload rows it won (pixel order) and sends them straight to rank 0, which unpacks
         them into place -- 12 B/pixel in total INTO rank 0, spread over its 7 links, instead of 12 B/pixel all-to-all
         plus 12 B/pixel to rank 0.  Optionally the winners' energies / indices travel the same way.

    `tiebreak_index` (interleaved shards -- rank r holds frames r, r + W, ...): the candidates' global frame indices travel with
    their energies in step 1 (8 instead of 4 B/pixel there) and break ties in step 2, because the rank order is not the
    frame order then.

    `comm`: how the collectives move the tensors (default `DirectComm(group)`; `HostStagedComm` stages device tensors
    through the host).  `force`: run the local steps at world 1 as well -- rank 0 then packs the rows it won and unpacks
    them from its own buffer, which moves nothing anywhere and only serves to TIME the per-rank kernel work
    (bench.py `combine_ms`).
    """
    world = dist.get_world_size(group) if group is not None else 1
    rank = dist.get_rank(group) if group is not None else 0
    n = e_all.numel()
    if (world == 1 and not force) or n == 0:
        return
    comm = comm or DirectComm(group)
    bounds = chunk_bounds(n, world)
    sizes = [b - a for a, b in bounds]
    per = -(-n // world)
    mine = sizes[rank]
    cand_i = None
    if world > 1:
        cand = torch.empty(world * mine, dtype=e_all.dtype, device=e_all.device)
        comm.all_to_all(cand, e_all, [mine] * world, sizes)
        if tiebreak_index:
            cand_i = torch.empty(world * mine, dtype=i_all.dtype, device=i_all.device)
            comm.all_to_all(cand_i, i_all, [mine] * world, sizes)
    else:
        cand = e_all
        cand_i = i_all if tiebreak_index else None
    win_chunk = (ops.winner(cand.view(world, mine), cand_i.view(world, mine)) if cand_i is not None
                 else ops.winner(cand.view(world, mine)))
    if world > 1:
        padded = torch.zeros(per, dtype=torch.uint8, device=e_all.device)
        padded[:mine] = win_chunk
        gathered = torch.empty(world * per, dtype=torch.uint8, device=e_all.device)
        comm.all_gather(gathered, padded)
        win = gathered[:n]          # chunk r starts at r * per: the padded layout IS the pixel order
    else:
        win = win_chunk
    plan, totals = ops.plan(win, max(world, 2))
    arrays = [(l_all, width)]
    if root_energy:
        arrays.append((e_all, 1))
    if with_index and i_all is not None:
        arrays.append((i_all.view(torch.float32), 1))   # moved bit for bit
    if world == 1:      # force: a sender's and the root's kernel work on this rank's own rows (rank 0 "won" them all;
        for arr, w in arrays:                       # unpacked as if by a second rank: every row lands where it was)
            own = ops.pack(win, plan, 2, 0, arr, w, totals[0])
            ops.unpack(win, plan, 2, 1, [own, None], arr, w)
        return
    if rank != 0:
        packed = [ops.pack(win, plan, world, rank, arr, w, totals[rank]) for arr, w in arrays]
        if totals[rank]:
            comm.send_to_root(packed)
    else:
        for arr, w in arrays:
            bufs = [None] + [torch.empty(totals[r] * w, dtype=arr.dtype, device=arr.device) for r in range(1, world)]
            comm.recv_at_root([(r, bufs[r]) for r in range(1, world) if totals[r]])
            ops.unpack(win, plan, world, 0, bufs, arr, w)


class _DevArray:
    """`__cuda_array_interface__` view of library-owned device memory (no copy)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False),
                                         "version": 2}


def wrap_device(ptr, n, dtype, device):
    typestr = {torch.float32: "<f4", torch.int32: "<i4"}[dtype]
    return torch.as_tensor(_DevArray(ptr, n, typestr), device=torch.device("cuda", device))


class Combiner:
    """Cross-GPU combine for a `_lib.Stack` on every rank of `group`."""

    def __init__(self, stack, group=None, comm=None, force=False):
        self.stack = stack
        # force (timing at world 1) works without any process group
        self.group = group if group is not None else (dist.group.WORLD if dist.is_initialized() else None)
        self.device = stack.device
        self.comm = comm        # None: DirectComm (RCCL on device tensors); HostStagedComm: device tensors over gloo
        self.force = force      # run the local kernels at world 1 too (timing only)
        self._slabs = None
        self._keep = []         # receive buffers / pointer tables the enqueued unpack kernels still read
        self.timings = {}       # host milliseconds of the last combine_winners(): see there

    def _select_hip(self, cand_e, cand_l, cand_i):
        world, m = cand_e.shape
        out_e = torch.empty(m, dtype=torch.float32, device=cand_e.device)
        out_l = torch.empty(cand_l.shape[1], dtype=torch.float32, device=cand_e.device)
        out_i = torch.empty(m, dtype=torch.int32, device=cand_e.device) if cand_i is not None else None
        stream = torch.cuda.current_stream(cand_e.device).cuda_stream
        _lib.check(_lib.load().mi_combine_select(
            self.device, C.c_void_p(stream), world, cand_e.data_ptr(), cand_l.data_ptr(),
            cand_i.data_ptr() if cand_i is not None else None, m, out_e.data_ptr(), out_l.data_ptr(),
            out_i.data_ptr() if out_i is not None else None))
        return out_e, out_l, out_i

    # ---- the library's kernels behind the TorchWinnerOps protocol
    def winner(self, cand_e, cand_i=None):
        world, m = cand_e.shape
        out = torch.empty(m, dtype=torch.uint8, device=cand_e.device)
        stream = torch.cuda.current_stream(cand_e.device).cuda_stream
        if cand_i is not None:
            _lib.check(_lib.load().mi_combine_winner_idx(self.device, C.c_void_p(stream), world, cand_e.data_ptr(),
                                                         cand_i.data_ptr(), m, out.data_ptr()))
        else:
            _lib.check(_lib.load().mi_combine_winner(self.device, C.c_void_p(stream), world, cand_e.data_ptr(), m, out.data_ptr()))
        return out

    def plan(self, win, world):
        lib = _lib.load()
        n = win.numel()
        plan = torch.empty(lib.mi_combine_plan_bytes(n, world), dtype=torch.uint8, device=win.device)
        totals = (C.c_int64 * world)()
        stream = torch.cuda.current_stream(win.device).cuda_stream
        _lib.check(lib.mi_combine_plan(self.device, C.c_void_p(stream), win.data_ptr(), n, world, plan.data_ptr(), totals))
        return plan, [int(t) for t in totals]

    def pack(self, win, plan, world, rank, arr, width, count):
        out = torch.empty(count * width, dtype=arr.dtype, device=arr.device)
        if count == 0:
            return out
        stream = torch.cuda.current_stream(arr.device).cuda_stream
        _lib.check(_lib.load().mi_combine_pack(self.device, C.c_void_p(stream), win.data_ptr(), win.numel(), world, rank,
                                               plan.data_ptr(), arr.data_ptr(), width, out.data_ptr()))
        return out

    def unpack(self, win, plan, world, rank, bufs, arr, width):
        ptrs = torch.tensor([b.data_ptr() if b is not None and b.numel() else 0 for b in bufs], dtype=torch.int64,
                            device=arr.device)
        stream = torch.cuda.current_stream(arr.device).cuda_stream
        _lib.check(_lib.load().mi_combine_unpack(self.device, C.c_void_p(stream), win.data_ptr(), win.numel(), world, rank,
                                                 plan.data_ptr(), ptrs.data_ptr(), width, arr.data_ptr()))
        # no synchronisation per array: the pointer table and the receive buffers stay alive until combine_winners()
        # synchronises once at its end
        self._keep.append((ptrs, bufs, win, plan))

    def combine_winners(self, with_index=False, root_energy=False):
        """The winners-only protocol (`combine_winners`) in two phases: level 0 -- 3/4 of the state, final as soon as
        the last batch's level-0 kernels are through -- is exchanged while that batch's coarser levels still run on
        the stacker's side streams; the rest follows after the full synchronisation."""
        st = self.stack
        if self._slabs is None:
            e_ptr, l_ptr, i_ptr, n = st.state_ptrs(-1)     # (synchronises; once per handle: the slabs do not move)
            n0 = -(-(st.shapes[0][0] * st.shapes[0][1]) // 64) * 64 if st.levels > 0 else 0
            self._slabs = (wrap_device(e_ptr, n, torch.float32, self.device), wrap_device(l_ptr, n * 3, torch.float32, self.device),
                           wrap_device(i_ptr, n, torch.int32, self.device), n0)
        e, l, i, n0 = self._slabs
        import time
        ts = torch.cuda.current_stream(torch.device("cuda", self.device))
        interleaved = getattr(st, "index_stride", 1) > 1    # the indices break ties, in their global form (export_indices)
        kw = dict(with_index=with_index, root_energy=root_energy, comm=self.comm, force=self.force, tiebreak_index=interleaved)
        t0 = time.perf_counter()
        st.sync_level(0)
        if interleaved and n0:
            st.export_indices(0)
        t1 = time.perf_counter()
        if n0:
            combine_winners(e[:n0], l[:3 * n0], i[:n0], self.group, self, **kw)
        t2 = time.perf_counter()
        st.sync()       # the coarser levels + base of this rank: they ran on the stacker's streams beside the exchange above
        if interleaved:
            st.export_indices(-1)
        t3 = time.perf_counter()
        combine_winners(e[n0:], l[3 * n0:], i[n0:], self.group, self, **kw)
        ts.synchronize()
        t4 = time.perf_counter()
        self._keep.clear()
        # wait_level0: host wait for this rank's level-0 state; exchange_level0: the level-0 exchange (enqueue + collectives,
        # its unpack kernels may still be running); wait_rest: what was LEFT of the coarser levels after that exchange
        # (0 = they were hidden behind it); exchange_rest: the coarse levels' exchange + the final synchronisation
        self.timings = {"wait_level0_ms": (t1 - t0) * 1e3, "exchange_level0_ms": (t2 - t1) * 1e3,
                        "wait_rest_ms": (t3 - t2) * 1e3, "exchange_rest_ms": (t4 - t3) * 1e3}

    def combine(self, with_index=True, root_energy=True):
        """Call on every rank after its frames were pushed; rank 0 may then finish().
        `with_index=False`: do not exchange the winner indices (debug taps only); `root_energy=False`: do not send the
        winners' energies to rank 0 either (the fused image needs the winners' Laplacians / base pixels alone)."""
        st = self.stack
        st.sync()  # the library's streams are not torch's
        # all levels, then base entropy twin, base deviation twin: one contiguous slab per array
        e_ptr, l_ptr, i_ptr, n = st.state_ptrs(-1)
        states = [(wrap_device(e_ptr, n, torch.float32, self.device),
                   wrap_device(l_ptr, n * 3, torch.float32, self.device),
                   wrap_device(i_ptr, n, torch.int32, self.device))]
        combine_all(states, self.group, self._select_hip, with_index=with_index, root_energy=root_energy)
        torch.cuda.current_stream(torch.device("cuda", self.device)).synchronize()
