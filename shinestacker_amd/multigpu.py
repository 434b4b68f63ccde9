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

    def __init__(self, stack, group=None):
        self.stack = stack
        self.group = group if group is not None else dist.group.WORLD
        self.device = stack.device

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
