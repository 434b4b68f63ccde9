"""CPU, world_size 2, gloo: the frame-sharded combine (shinestacker_amd/multigpu.py).

Each rank builds the running state of ITS frame block with the oracle (standing in for the
per-GPU HIP path), the ranks exchange and combine with the product's combine_state(), and rank
0 must hold exactly the state of the whole stack processed in one go -- including first-max
tie-breaking across ranks (the stack contains duplicate frames on different ranks)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _frames():
    rng = np.random.default_rng(77)
    fr = [rng.integers(0, 256, (72, 104, 3), dtype=np.uint8) for _ in range(5)]
    fr.insert(3, fr[1].copy())   # duplicates: index 3 == index 1 (other rank), 6 == 0
    fr.append(fr[0].copy())
    return fr  # 7 frames: rank 0 gets [0,4), rank 1 gets [4,7)


def _torch_select(cand_e, cand_l, cand_i):
    """Reference first-max for CPU tensors (test double of mi_combine_select)."""
    world, m = cand_e.shape
    best = torch.zeros(m, dtype=torch.long)
    be = cand_e[0].clone()
    for r in range(1, world):
        win = cand_e[r] > be
        be = torch.where(win, cand_e[r], be)
        best = torch.where(win, torch.full_like(best, r), best)
    ar = torch.arange(m)
    lap = cand_l.view(world, m, -1)[best, ar].reshape(-1)
    return be, lap, (cand_i[best, ar] if cand_i is not None else None)


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        from shinestacker_amd import multigpu
        frames = _frames()
        cut = [0, 4, 7]
        so = orc.StreamingOracle(72, 104, np.uint8, min_size=8)
        so.n = 0
        first = cut[rank]
        for k, f in enumerate(frames[cut[rank]:cut[rank + 1]]):
            so.push_frame(f)
        # global frame indices (the HIP path gets them from set_first_index)
        for lv in range(so.levels):
            so.best_idx[lv] += first
        so.idx_e += first
        so.idx_d += first
        levels = [(so.best_e[lv], so.best_lap[lv], so.best_idx[lv]) for lv in range(so.levels)]
        # base twins: (entropy, winner's base pixel), (deviation, winner's base pixel)
        bases = np.stack(so.bases)
        hb, wb = so.shapes[so.levels]
        yy, xx = np.mgrid[0:hb, 0:wb]
        base_e = bases[so.idx_e - first, yy, xx]
        base_d = bases[so.idx_d - first, yy, xx]
        levels.append((so.b_ent, base_e, so.idx_e))
        levels.append((so.b_dev, base_d, so.idx_d))
        def tensors():
            return [(torch.from_numpy(np.ascontiguousarray(e).ravel().copy()),
                     torch.from_numpy(np.ascontiguousarray(l).ravel().copy()),
                     torch.from_numpy(np.ascontiguousarray(i).ravel().copy())) for e, l, i in levels]
        # level by level ...
        per_level = tensors()
        for te, tl, ti in per_level:
            multigpu.combine_state(te, tl, ti, dist.group.WORLD, _torch_select)
        # ... and all levels in one flat exchange (what Combiner.combine uses): identical result
        flat = tensors()
        multigpu.combine_all(flat, dist.group.WORLD, _torch_select)
        # one contiguous slab per array (what Combiner.combine hands over): exchanged in place
        t = tensors()
        slab = [(torch.cat([a for a, _, _ in t]), torch.cat([b for _, b, _ in t]), torch.cat([c for _, _, c in t]))]
        multigpu.combine_all(slab, dist.group.WORLD, _torch_select)
        if rank == 0:
            assert torch.equal(slab[0][0], torch.cat([a for a, _, _ in flat]))
            assert torch.equal(slab[0][1], torch.cat([b for _, b, _ in flat]))
            assert torch.equal(slab[0][2], torch.cat([c for _, _, c in flat]))
        # without the indices: energies and laps identical, indices untouched
        noidx = tensors()
        multigpu.combine_all(noidx, dist.group.WORLD, _torch_select, with_index=False)
        if rank == 0:
            for a, b, c, orig in zip(per_level, flat, noidx, tensors()):
                for x, y in zip(a, b):
                    assert torch.equal(x, y)
                assert torch.equal(c[0], a[0]) and torch.equal(c[1], a[1]) and torch.equal(c[2], orig[2])
        # payloads only (what bench.py uses): the winners' Laplacians reach rank 0, energies and indices stay local
        pay = tensors()
        multigpu.combine_all(pay, dist.group.WORLD, _torch_select, with_index=False, root_energy=False)
        if rank == 0:
            for a, c, orig in zip(per_level, pay, tensors()):
                assert torch.equal(c[1], a[1]) and torch.equal(c[0], orig[0]) and torch.equal(c[2], orig[2])
            ret["state"] = [(te.numpy(), tl.numpy(), ti.numpy()) for te, tl, ti in flat]
        # the winners-only protocol (energies -> winner map -> packed rows straight to rank 0): same state on rank 0,
        # with everything / with the payload alone; in two phases like Combiner.combine_winners (level 0 first)
        ops = multigpu.TorchWinnerOps()
        for kw in ({}, {"with_index": False, "root_energy": False}):
            t = tensors()
            e_all, l_all, i_all = (torch.cat([a for a, _, _ in t]), torch.cat([b for _, b, _ in t]), torch.cat([c for _, _, c in t]))
            n0 = t[0][0].numel()
            multigpu.combine_winners(e_all[:n0], l_all[:3 * n0], i_all[:n0], dist.group.WORLD, ops, **kw)
            multigpu.combine_winners(e_all[n0:], l_all[3 * n0:], i_all[n0:], dist.group.WORLD, ops, **kw)
            if rank == 0:
                assert torch.equal(l_all, torch.cat([b for _, b, _ in flat]))
                if not kw:
                    assert torch.equal(e_all, torch.cat([a for a, _, _ in flat]))
                    assert torch.equal(i_all, torch.cat([c for _, _, c in flat]))
                else:
                    o = tensors()
                    assert torch.equal(e_all, torch.cat([a for a, _, _ in o])) and torch.equal(i_all, torch.cat([c for _, _, c in o]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_combine_equals_single_stack(oracle):
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
        state = ret["state"]
    frames = _frames()
    so = oracle.StreamingOracle(72, 104, np.uint8, min_size=8)
    for f in frames:
        so.push_frame(f)
    for lv in range(so.levels):
        e, l, i = state[lv]
        assert np.array_equal(e, so.best_e[lv].ravel())
        assert np.array_equal(i, so.best_idx[lv].ravel())
        assert np.array_equal(l, so.best_lap[lv].ravel())
        # duplicates live on the other rank: the earlier copy must have won
        assert not np.isin(i, [3, 6]).any()
    e, l, i = state[so.levels]
    assert np.array_equal(e, so.b_ent.ravel()) and np.array_equal(i, so.idx_e.ravel())
    e, l, i = state[so.levels + 1]
    assert np.array_equal(e, so.b_dev.ravel()) and np.array_equal(i, so.idx_d.ravel())
    # fused base from the combined twins == oracle
    be = state[so.levels][1].reshape(-1)
    bd = state[so.levels + 1][1].reshape(-1)
    fused = ((0.0 + be) + bd) / 2.0
    assert np.array_equal(fused.astype(np.float32), so.fused_base().ravel())


def _worker_interleaved(rank, world, port, ret):
    """rank r holds frames r, r + world, ...: the winners-only protocol with the frame indices as tie-breakers"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        from shinestacker_amd import multigpu
        frames = _frames()            # 3 == 1 (both on rank 1), 6 == 0 (both on rank 0) ...
        frames[4] = frames[1].copy()  # ... and 4 (rank 0) == 1 (rank 1): the LOWER index lives on the HIGHER rank
        so = orc.StreamingOracle(72, 104, np.uint8, min_size=8)
        for f in frames[rank::world]:
            so.push_frame(f)
        g = lambda a: a * world + rank                      # local frame number -> global frame index
        levels = [(so.best_e[lv], so.best_lap[lv], g(so.best_idx[lv])) for lv in range(so.levels)]
        bases = np.stack(so.bases)
        hb, wb = so.shapes[so.levels]
        yy, xx = np.mgrid[0:hb, 0:wb]
        levels.append((so.b_ent, bases[so.idx_e, yy, xx], g(so.idx_e)))
        levels.append((so.b_dev, bases[so.idx_d, yy, xx], g(so.idx_d)))
        t = [(torch.from_numpy(np.ascontiguousarray(e).ravel().copy()), torch.from_numpy(np.ascontiguousarray(l).ravel().copy()),
              torch.from_numpy(np.ascontiguousarray(i, np.int32).ravel().copy())) for e, l, i in levels]
        e_all, l_all, i_all = (torch.cat([a for a, _, _ in t]), torch.cat([b for _, b, _ in t]), torch.cat([c for _, _, c in t]))
        n0 = t[0][0].numel()
        ops = multigpu.TorchWinnerOps()
        multigpu.combine_winners(e_all[:n0], l_all[:3 * n0], i_all[:n0], dist.group.WORLD, ops, tiebreak_index=True)
        multigpu.combine_winners(e_all[n0:], l_all[3 * n0:], i_all[n0:], dist.group.WORLD, ops, tiebreak_index=True)
        if rank == 0:
            ret["state"] = (e_all.numpy(), l_all.numpy(), i_all.numpy(), [a.numel() for a, _, _ in t])
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_combine_of_interleaved_shards_equals_single_stack(oracle):
    """SURVEY 8(e) with the frames dealt round-robin (rank r: frames r, r + W, ...): np.argmax's first maximum
    (pyramid.py:51) across ranks needs the candidates' global frame indices -- a tie goes to the lower index, whichever rank
    holds it."""
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker_interleaved, args=(2, port, ret), nprocs=2, join=True)
        e_all, l_all, i_all, sizes = ret["state"]
    frames = _frames()
    frames[4] = frames[1].copy()
    so = oracle.StreamingOracle(72, 104, np.uint8, min_size=8)
    for f in frames:
        so.push_frame(f)
    hb, wb = so.shapes[so.levels]
    yy, xx = np.mgrid[0:hb, 0:wb]
    bases = np.stack(so.bases)
    want = [(so.best_e[lv], so.best_lap[lv], so.best_idx[lv]) for lv in range(so.levels)]
    want += [(so.b_ent, bases[so.idx_e, yy, xx], so.idx_e), (so.b_dev, bases[so.idx_d, yy, xx], so.idx_d)]
    off = 0
    for (e, l, i), n in zip(want, sizes):
        assert np.array_equal(e_all[off:off + n], np.asarray(e, np.float32).ravel())
        assert np.array_equal(i_all[off:off + n], np.asarray(i, np.int32).ravel())
        assert np.array_equal(l_all[3 * off:3 * (off + n)], np.asarray(l, np.float32).ravel())
        off += n
    assert not np.isin(i_all, [3, 4, 6]).any()       # every duplicate lost to its earlier copy


def test_chunk_bounds_cover_everything():
    from shinestacker_amd.multigpu import chunk_bounds
    for n in (0, 1, 7, 8, 9, 1000003):
        for w in (1, 2, 3, 8):
            b = chunk_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("scaling,nproc,shards", [("strong", 2, "interleaved"), ("weak", 2, "interleaved"), ("strong", 8, "interleaved"),
                                                  ("strong", 2, "contiguous")])
def test_bench_dry_run_under_torch_distributed_run(scaling, nproc, shards):
    """bench.py's distributed skeleton the way the driver starts it (python -m torch.distributed.run, one process per
    rank), on CPU: --dry-run replaces the HIP path by the oracle and RCCL by gloo; the strong split (BASELINE configs[2]:
    the SAME stack, frames / N per rank), the winners-only combine and the JSON contract are the real ones."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(nproc), "--steps", "2", "--warmup", "1",
           "--frames", "6" if nproc == 2 else "16", "--scaling", scaling, "--shards", shards, "--dry-run"]   # 8 ranks: the driver's configs[2] launch
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=root)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and lines, r.stdout[-1500:] + r.stderr[-3000:]
    d = json.loads(lines[-1])
    assert d["n_gpus"] == nproc and d["scaling"] == scaling and d["verified"] is True and d["steps"] == 2
    assert d["config"]["shards"] == shards
    assert d["config"]["frames_per_gpu"] == ((3 if scaling == "strong" else 6) if nproc == 2 else 2)
    assert {"compute_ms_host", "combine_ms_host", "collapse_ms_host"} <= set(d["breakdown_ms_per_step"])
    for k in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data"):
        assert k in d
