"""The C-ABI collective (include/rpb.h rpb_dp_*: RCCL all-reduce on a side HIP stream) called through ctypes, as a maintainer's
binding would: a one-rank communicator on the 1-GPU box (plumbing, stream / event ordering, both dtypes) and -- when the node
has two GPUs -- a two-rank sum over xGMI with one process per GPU."""
import ctypes
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from realpdebench_amd import _lib
    _lib.load()
    return _lib


def _comm(id_bytes, rank, world):
    L = _lib()
    buf = (ctypes.c_char * 128).from_buffer_copy(id_bytes)
    h = ctypes.c_void_p()
    L.call("rpb_dp_allreduce_init", ctypes.addressof(buf), rank, world, ctypes.addressof(h))
    return h.value


def _unique_id():
    L = _lib()
    buf = (ctypes.c_char * 128)()
    L.call("rpb_dp_unique_id", ctypes.addressof(buf))
    return bytes(buf)


def test_single_rank_communicator_orders_against_the_compute_stream():
    L = _lib()
    assert L.load().rpb_dp_available() == 1
    torch.cuda.set_device(0)
    h = _comm(_unique_id(), 0, 1)
    try:
        st = torch.cuda.current_stream().cuda_stream
        a = torch.randn(4096, 4096, device="cuda")
        for _ in range(4):
            a = torch.tanh(a @ a * 1e-3)                       # milliseconds of queued work the side stream must wait for
        want = a.clone()
        g = a.view(-1)
        L.call("rpb_dp_allreduce_enqueue", h, g.data_ptr(), g.numel(), 0, st)          # sum over one rank = identity
        L.call("rpb_dp_allreduce_enqueue", h, g[:1000].data_ptr(), 1000, 0, st)        # a second bucket behind it
        L.call("rpb_dp_allreduce_wait", h, st)
        b = a * 2.0                                            # consumer on the compute stream
        d = torch.arange(7, device="cuda", dtype=torch.float64)
        L.call("rpb_dp_allreduce_inline", h, d.data_ptr(), d.numel(), 1, st)           # fp64, on the compute stream itself
        torch.cuda.synchronize()
        assert torch.equal(a, want) and torch.equal(b, want * 2.0)
        assert torch.equal(d.cpu(), torch.arange(7, dtype=torch.float64))
        with pytest.raises(L.RpbError):
            L.call("rpb_dp_allreduce_enqueue", h, g.data_ptr(), g.numel(), 5, st)      # unknown dtype code
    finally:
        L.call("rpb_dp_allreduce_destroy", h)


def _two_rank_worker(rank, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("gloo", rank=rank, world_size=2)              # rendezvous only: carries the RCCL id
    try:
        box = [_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        L = _lib()
        h = _comm(box[0], rank, 2)
        st = torch.cuda.current_stream().cuda_stream
        g = torch.full((1 << 20,), float(rank + 1), device="cuda") * torch.arange(1 << 20, device="cuda").remainder(7).float()
        L.call("rpb_dp_allreduce_enqueue", h, g.data_ptr(), g.numel(), 0, st)
        L.call("rpb_dp_allreduce_wait", h, st)
        torch.cuda.synchronize()
        out[rank] = g.cpu()
        L.call("rpb_dp_allreduce_destroy", h)
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="two-rank RCCL needs two GPUs (RCCL refuses two ranks on one device)")
def test_two_ranks_sum_over_xgmi():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_two_rank_worker, args=(port, out), nprocs=2, join=True)
        res = {k: v for k, v in out.items()}
    want = 3.0 * torch.arange(1 << 20).remainder(7).float()
    assert torch.equal(res[0], want) and torch.equal(res[1], want)


def _rcclcomm_worker(rank, world, port, two_comms, out):
    """dp.RcclComm as the trainers build it, under the nccl backend: buckets on the side stream, SyncBN-sized inline reductions on the
    compute stream, the instrumentation of the N > 1 bench line, and teardown -- with one communicator (the default) or two."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if two_comms:
        os.environ["RPB_DP_TWO_COMMS"] = "1"
    else:
        os.environ.pop("RPB_DP_TWO_COMMS", None)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        from realpdebench_amd.dp import RcclComm
        c = RcclComm()
        assert (c.small != c.handle) == bool(two_comms)
        g = torch.full((25 << 20,), float(rank + 1), device="cuda")              # a 100 MB bucket
        s = torch.full((128,), float(rank + 1), device="cuda", dtype=torch.float64)
        for it in range(3):                                                      # bucket in flight while the statistics reduce
            if it == 2:
                c.set_timing(True)                                               # instrument the last step only (what bench.py does)
            g.fill_(float(rank + 1))
            s.fill_(float(rank + 1))
            c.enqueue(g)
            c.inline(s)
            c.inline(s)
            c.wait()
        torch.cuda.synchronize()
        tot = world * (world + 1) / 2
        out[rank] = (float(g[0]), float(g[-1]), float(s[0]), tot)
        t = c.step_times()
        assert len(t["buckets"]) == 1 and len(t["inline_ms"]) == 2 and t["buckets"][0]["MB"] > 100
        c.close()
        with pytest.raises(RuntimeError):
            c.inline(s)                                                          # closed communicators refuse, they do not detach
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("two_comms", [False, True])
def test_rcclcomm_buckets_and_statistics_one_rank(two_comms):
    """One rank through the whole dp.RcclComm path on the 1-GPU box (the two-rank variant below needs two GPUs)."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_rcclcomm_worker, args=(1, port, two_comms, out), nprocs=1, join=True)
        res = dict(out)
    assert res[0][:2] == (1.0, 1.0) and res[0][2] == 1.0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
@pytest.mark.parametrize("two_comms", [False, True])
def test_rcclcomm_buckets_and_statistics_two_ranks(two_comms):
    """The multi-GPU gate of the advisor's round-3 finding: both communicator modes with a 100 MB bucket in flight while SyncBN-sized
    reductions run on the compute stream, then step_times() and close()."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_rcclcomm_worker, args=(2, port, two_comms, out), nprocs=2, join=True)
        res = dict(out)
    for r in (0, 1):
        assert res[r][0] == 3.0 and res[r][1] == 3.0       # 1 + 2
        assert res[r][2] == 6.0                            # two successive in-place sums: (1, 2) -> 3 on both ranks -> 6


def test_reduce_scatter_allgather_marks_and_modelled_transfer_one_rank():
    """rpb_dp_reduce_scatter_enqueue / rpb_dp_allgather_enqueue on a one-rank communicator are the identity, ordered against the compute
    stream; rpb_dp_mark / rpb_dp_wait_mark let a stream wait for a point of the side stream; rpb_dp_set_model idles the side stream for
    the modelled transfer (the one-GPU strong-scaling proxy) -- visible as time, not in the data."""
    import time
    L = _lib()
    torch.cuda.set_device(0)
    h = _comm(_unique_id(), 0, 1)
    try:
        st = torch.cuda.current_stream().cuda_stream
        g = torch.randn(1 << 22, device="cuda")
        want = g.clone()
        L.call("rpb_dp_reduce_scatter_enqueue", h, g.data_ptr(), g.numel(), 0, st)
        L.call("rpb_dp_allgather_enqueue", h, g.data_ptr(), g.numel(), 0, st)
        L.call("rpb_dp_mark", h, 3)
        L.call("rpb_dp_wait_mark", h, 3, st)
        b = g * 2.0
        torch.cuda.synchronize()
        assert torch.equal(g, want) and torch.equal(b, want * 2.0)
        with pytest.raises(L.RpbError):
            L.call("rpb_dp_mark", h, 99)
        # modelled transfer: 64 MB over 8 ranks at 100 GB/s per link = 2 phases x 0.56 ms for the all-reduce
        L.call("rpb_dp_set_model", h, 8, 100.0, 0.0)
        big = torch.zeros(16 << 20, device="cuda")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        L.call("rpb_dp_allreduce_enqueue", h, big.data_ptr(), big.numel(), 0, st)
        L.call("rpb_dp_allreduce_wait", h, st)
        torch.cuda.synchronize()
        dt = 1e3 * (time.perf_counter() - t0)
        assert 0.9 < dt < 3.0, dt
        L.call("rpb_dp_set_model", h, 0, 0.0, 0.0)
    finally:
        L.call("rpb_dp_allreduce_destroy", h)


def _sharded_step_worker(rank, port, out):
    import torch.distributed as dist
    from realpdebench_amd.dp import DataParallel
    from realpdebench_amd.model.fno import FNO3d
    from realpdebench_amd.trainer import Trainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", RPB_LINE_CLAIM="0",
                      RPB_DP_CHUNK_MB="0.25")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        shape = (6, 16, 16, 2)
        finals = []
        for shard, clip in ((False, 0.0), (True, 0.0), (False, 0.05), (True, 0.05)):
            torch.manual_seed(3)
            m = FNO3d(2, 4, 4, 2, 64, shape, shape).to("cuda:0")
            DataParallel(m, shard_optimizer=shard)
            tr = Trainer(m, lr=1e-3, num_update=10, clip_grad_norm=clip)
            torch.manual_seed(4)
            x, y = torch.randn(2, *shape, device="cuda"), torch.randn(2, *shape, device="cuda")
            losses = [float(tr.step(x, y)) for _ in range(3)]
            ck = tr.checkpoint()                         # (waits for the parameter gathers of the sharded step)
            finals.append((m.flat.data.clone().cpu(), losses))
            if shard and clip == 0.0:
                pieces, chunks = m.dp.shard_plan()
                out["sharded_chunks"] = sum(1 for row in chunks for _, _, sh in row if sh)
                out["ranges"] = len(pieces)
                # validation straight after a sharded step (train.py): the eval forward -- eager, then replayed from its hipGraph
                # (captured on the third call), which never enters _forward_impl -- and state_dict() must wait for the gathers
                pend = []
                for _ in range(5):
                    tr.step(x, y)
                    pend.append(m.dp._pending)
                    m.eval()
                    with torch.no_grad():
                        m(x)
                    pend.append(m.dp._pending)
                    m.train()
                tr.step(x, y)
                m.state_dict()
                pend.append(m.dp._pending)
                out["pending"] = pend
            tr.close()
        out["equal"] = bool(torch.equal(finals[0][0], finals[1][0])) and finals[0][1] == finals[1][1]
        # (the clip coefficient comes from an fp64 sum of per-range norms in the sharded step, from one fp32 norm in the plain one: the
        # same step up to the rounding of that one scalar)
        out["equal_clip"] = (float((finals[2][0] - finals[3][0]).abs().max()) <= 2e-6 * float(finals[2][0].abs().max())
                             and all(abs(a - b) <= 1e-6 * abs(a) for a, b in zip(finals[2][1], finals[3][1])))
        out["clip_acts"] = not torch.equal(finals[0][0], finals[2][0])
        out["diag"] = (float((finals[0][0] - finals[1][0]).abs().max()), float((finals[2][0] - finals[3][0]).abs().max()),
                       finals[0][1], finals[1][1])
    finally:
        dist.destroy_process_group()


def test_sharded_optimizer_step_one_rank_equals_the_plain_step():
    """Trainer.step with DataParallel(shard_optimizer=True) on a one-rank RCCL group: reduce-scatter per chunk, rpb_adam_step_ranges over
    the owned ranges (here: the whole arena, in several ranges), all-gather + per-bucket waits in the next forward == the plain step,
    bit for bit over three steps (RPB_LINE_CLAIM=0: bit-reproducible training launches)."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_sharded_step_worker, args=(port, out), nprocs=1, join=True)
        res = dict(out)
    assert res["sharded_chunks"] >= 2 and res["ranges"] > res["sharded_chunks"]
    assert res["equal"], f"sharded optimizer step != plain step on one rank: {res['diag']}"
    assert res["clip_acts"] and res["equal_clip"], f"sharded step with clip_grad_norm != plain step with it: {res['diag']}"
    # after every sharded step gathers are pending; after an eval forward (eager or graph replay) / state_dict() they are not
    assert res["pending"] == [True, False] * 5 + [False], res["pending"]
