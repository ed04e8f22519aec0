"""Data-parallel host logic on CPU with the gloo backend, world_size 2 (the N>1 path of bench.py / Trainer):
parameter broadcast, per-layer bucket plan, overlapped bucket all-reduce == plain sum over ranks, SyncBN-style
small reductions and batch sharding.  The HIP kernels themselves need a GPU; what is covered here is everything
``realpdebench_amd.dp`` adds around them."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from realpdebench_amd.dp import DataParallel, layer_buckets
from realpdebench_amd.model.fno import FNO3d


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_model(seed):
    torch.manual_seed(seed)
    return FNO3d(2, 3, 3, 3, 32, (4, 8, 8, 2), (4, 8, 8, 2))


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _make_model(seed=100 + rank)             # different init per rank on purpose
        model.bn_running_mean.fill_(float(rank + 1))
        os.environ["RPB_DP_CHUNK_MB"] = "0.05"           # 13107-element chunks: every spectral bucket of this model travels in several pieces
        dp = DataParallel(model)
        res = {}
        # 1. rank 0's parameters and buffers were broadcast
        ref = _make_model(seed=100)
        res["bcast"] = bool(torch.equal(model.flat.data, ref.flat.data)) and float(model.bn_running_mean[0, 0]) == 1.0
        # 2. bucketed gradient all-reduce: announce buckets in backward order, rest flushed by finish_step
        g = torch.full_like(model.flat.data, float(rank + 1))
        g[::7] = float(10 * (rank + 1))
        dp.begin_step(g)
        dp.bucket_ready(g)            # tail (fc1, fc2)
        dp.bucket_ready(g)            # layer L-1
        dp.finish_step(g)             # remaining layers + head
        expect = torch.full_like(g, 3.0)
        expect[::7] = 30.0
        res["allreduce"] = bool(torch.equal(g, expect))
        big = max(e - s0 for s0, e in dp.buckets)
        res["chunked"] = big > dp.chunk_elems and all(b - a <= dp.chunk_elems for s0, e in dp.buckets for a, b in dp.chunks(s0, e)) \
            and [c for s0, e in dp.buckets for c in dp.chunks(s0, e)][0][0] == dp.buckets[0][0] \
            and sum(b - a for s0, e in dp.buckets for a, b in dp.chunks(s0, e)) == g.numel()
        # 3. small synchronous reduction used for the BatchNorm statistics (fp64)
        s = torch.tensor([1.0 + rank, 2.0], dtype=torch.float64)
        dp.all_reduce_sum(s)
        res["bn"] = s.tolist() == [3.0, 4.0]
        res["shard"] = list(dp.shard(8))
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_bucket_plan_covers_arena_in_backward_order():
    m = _make_model(0)
    total = m.flat.numel()
    b = layer_buckets(m._seg, m.n_layers, total)
    assert len(b) == m.n_layers + 2
    # disjoint cover
    cover = sorted(b)
    assert cover[0][0] == 0 and cover[-1][1] == total
    for (s0, e0), (s1, e1) in zip(cover, cover[1:]):
        assert e0 == s1
    # completion order of the backward pass: tail first, then layers L-1..0, then the small fc0 head
    assert b[0][0] == m._seg["fc1.weight"][0]
    assert b[1][0] == m._seg[f"spec.{m.n_layers - 1}"][0]
    assert b[-2] == (m._seg["spec.0"][0], m._seg["spec.1"][0])
    assert b[-1] == (0, m._seg["spec.0"][0])
    # the spectral weights dominate each per-layer bucket
    assert m._seg["spec.1"][1] > 0.9 * (b[2][1] - b[2][0])


def test_dataparallel_world2_gloo():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    assert set(res) == {0, 1}
    for r in (0, 1):
        assert res[r]["bcast"], "parameters / buffers were not broadcast from rank 0"
        assert res[r]["allreduce"], "bucketed all-reduce != sum over ranks"
        assert res[r]["chunked"], "buckets were not cut into <= chunk_elems pieces covering the arena"
        assert res[r]["bn"]
    assert res[0]["shard"] == [0, 1, 2, 3] and res[1]["shard"] == [4, 5, 6, 7]


def _shard_worker(rank, world, port, out):
    """The sharded optimizer step's data movement on CPU tensors: reduce-scatter (gloo: all-reduce) of the spectral chunks, an update of
    the OWNED pieces only, all-gather of the parameter pieces == all-reduce + update of the whole arena, bit for bit."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        os.environ["RPB_DP_CHUNK_MB"] = "0.05"
        res = {}
        model = _make_model(seed=7)
        dp = DataParallel(model, shard_optimizer=True)
        torch.manual_seed(1000 + rank)
        g = torch.randn_like(model.flat.data)
        # reference: plain all-reduce of the whole arena, every rank updates everything
        g_ref = g.clone()
        dist.all_reduce(g_ref)
        p_ref = model.flat.data.clone() - 0.1 * g_ref
        # sharded: buckets announced in backward order with the held tails, as the backward pass does
        dp.begin_step(g, sharded=True)
        res["step_sharded"] = dp._step_sharded
        dp.bucket_ready(g)
        for l in range(model.n_layers - 1, -1, -1):
            dp.bucket_ready(g, hold_small_of=l if l > 0 else None)
            dp.small_ready(g, l)
        dp.finish_step(g)
        pieces, chunks = dp.shard_plan()
        # gloo all-reduces where RCCL reduce-scatters: poison every sharded piece this rank does NOT own, as a real reduce-scatter leaves
        # them (partial sums) -- nothing below may read them
        for row in chunks:
            for a, b, sh in row:
                if sh:
                    n = (b - a) // world
                    for q in range(world):
                        if q != rank:
                            g[a + q * n:a + (q + 1) * n] = float("nan")
        res["norm"] = dp.sharded_grad_norm(g)                 # clip_grad_norm with the sharded step (fp64 partials, one all-reduce)
        res["norm_ref"] = float(g_ref.double().norm())
        n_sharded = sum(1 for row in chunks for _, _, sh in row if sh)
        res["some_sharded"] = n_sharded >= model.n_layers
        res["owned_fraction"] = sum(n for _, n in pieces) / g.numel()
        p = model.flat.data
        for a, n in pieces:                                   # the stand-in for rpb_adam_step_ranges: only the owned ranges move
            p[a:a + n] -= 0.1 * g[a:a + n]
        tab, nr, total = dp.owned_table("cpu")
        res["table"] = nr == len(pieces) and total == sum(n for _, n in pieces) and [int(v) for v in tab[:, 0]] == [a for a, _ in pieces]
        stale = float((p - p_ref).abs().max())                # before the gather the foreign pieces are still the old parameters
        dp.gather_params(p)
        res["stale_before_gather"] = stale > 0
        res["equal"] = bool(torch.equal(p, p_ref))
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_sharded_optimizer_step_equals_allreduce_world2_gloo():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_shard_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    for r in (0, 1):
        assert res[r]["some_sharded"] and res[r]["table"]
        assert 0.5 < res[r]["owned_fraction"] < 0.6, res[r]["owned_fraction"]       # half of the spectral ranges + all the small pieces
        assert res[r]["stale_before_gather"]
        assert res[r]["equal"], "reduce-scatter + owned update + all-gather != all-reduce + full update"
        assert res[r]["step_sharded"]
        assert abs(res[r]["norm"] - res[r]["norm_ref"]) < 1e-6 * res[r]["norm_ref"], (res[r]["norm"], res[r]["norm_ref"])


def _unsharded_caller_worker(rank, world, port, out):
    """A caller that did not opt in (the autograd path, a torch optimizer) gets plain all-reduces even with shard_optimizer=True."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _make_model(seed=7)
        dp = DataParallel(model, shard_optimizer=True)
        dp.begin_step(model.flat.data)
        flag = dp._step_sharded
        try:
            DataParallel(_make_model(seed=8), shard_optimizer=True, shard_world=8)
            refused = False
        except ValueError:
            refused = True
        out[rank] = {"step_sharded": flag, "refused": refused}
    finally:
        dist.destroy_process_group()


def test_reduce_scatter_is_opt_in_per_step_and_shard_world_is_checked():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_unsharded_caller_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    for r in (0, 1):
        assert res[r]["step_sharded"] is False and res[r]["refused"]
