"""The peer-pointer optimizer step (DataParallel(p2p=True): C ABI rpb_dp_p2p_*, csrc/rpb_p2p.hip) on ONE MI355X:
two processes share cuda:0, map each other's gradient / parameter arenas and flag blocks through CUDA IPC handles and run the protocol
the N-GPU run would (announce gradients -> wait -> sum the W arenas' slice, Adam, store into all W parameter arenas -> announce -> the
next forward waits).  RCCL refuses two ranks on one device; peer pointers do not.  Checked: the two ranks end bit-identical, the result
equals the gloo all-reduce path bit for bit (two ranks: a + b is commutative), eval / state_dict after a step wait for the peers' slices,
and a rank whose peer never shows up gets a timeout status instead of a hung GPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

SHAPE, MODES, WIDTH, L = (5, 12, 10, 2), (2, 3, 4), 32, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data():
    g = torch.Generator().manual_seed(5)
    return torch.randn(4, *SHAPE, generator=g), torch.randn(4, *SHAPE, generator=g)


def _model():
    from realpdebench_amd.model.fno import FNO3d
    torch.manual_seed(11)
    return FNO3d(*MODES, L, WIDTH, SHAPE, SHAPE)


def _worker(rank, world, port, p2p, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", RPB_LINE_CLAIM="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from realpdebench_amd.dp import DataParallel
        from realpdebench_amd.trainer import Trainer
        torch.cuda.set_device(0)
        model = _model().cuda()
        dp = DataParallel(model, p2p=p2p)
        tr = Trainer(model, lr=1e-3, num_update=10)
        x, y = _data()
        idx = list(dp.shard(4))
        xs, ys = x[idx].cuda(), y[idx].cuda()
        losses, pend = [], []
        for it in range(3):
            losses.append(float(tr.step(xs, ys)))
            pend.append(bool(dp._pending))
            if it == 1:                      # validation straight after a step: the eval forward must wait for the peers' slices
                model.eval()
                with torch.no_grad():
                    ev = model(xs).clone()
                model.train()
                pend.append(bool(dp._pending))
        sd = {k: v.cpu() for k, v in model.state_dict().items()}
        torch.cuda.synchronize()
        if p2p:
            dp.peer.check()
            a, n = dp.peer.owned()
        else:
            a, n = 0, 0
        dist.barrier()
        out[rank] = {"flat": model.flat.data.cpu(), "loss": losses, "pend": pend, "eval": ev.cpu(), "owned": (a, n),
                     "fc0": sd["fc0.weight"], "total": model.flat.numel()}
        tr.close()
    finally:
        dist.destroy_process_group()


def _run(p2p):
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, p2p, out), nprocs=world, join=True)
        return {k: v for k, v in out.items()}


def test_peer_pointer_step_equals_the_allreduce_step_two_ranks_one_gpu():
    ref = _run(False)
    got = _run(True)
    assert torch.equal(got[0]["flat"], got[1]["flat"]), "the two ranks hold different parameters after the peer-pointer steps"
    assert torch.equal(got[0]["flat"], ref[0]["flat"]), float((got[0]["flat"] - ref[0]["flat"]).abs().max())
    assert got[0]["loss"] == ref[0]["loss"] and got[1]["loss"] == ref[1]["loss"]
    assert torch.equal(got[0]["eval"], ref[0]["eval"])
    assert torch.equal(got[1]["fc0"], ref[1]["fc0"])
    # after a step the peers' stores are pending; the eval forward (and only it) clears that; the slices tile the arena
    assert got[0]["pend"] == [True, True, False, True], got[0]["pend"]
    total = got[0]["total"]
    (a0, n0), (a1, n1) = got[0]["owned"], got[1]["owned"]
    assert a0 == 0 and a1 == n0 and n0 + n1 == total and n0 % 4 == 0


def _timeout_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from realpdebench_amd.dp import PeerExchange
        torch.cuda.set_device(0)
        flat, grad = torch.ones(1024, device="cuda"), torch.ones(1024, device="cuda")
        m, v = torch.zeros_like(flat), torch.zeros_like(flat)
        peer = PeerExchange(flat, grad, timeout_ms=300)
        if rank == 0:                        # rank 1 never announces its gradients: rank 0 must come back with a status, not hang
            peer.adam(m, v, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0)
            torch.cuda.synchronize()
            try:
                peer.check()
                out["raised"] = False
            except RuntimeError as e:
                out["raised"] = "rank 1" in str(e)
            out["untouched"] = bool(torch.equal(flat.cpu(), torch.ones(1024)))
        dist.barrier()
        peer.close()
    finally:
        dist.destroy_process_group()


def test_silent_peer_times_out_instead_of_hanging():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_timeout_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    assert res["raised"] is True and res["untouched"] is True


# ------------------------------------------------------------------------------------------ the arena models (ArenaTrainer)
def _arena_worker(rank, world, port, kind, p2p, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", RPB_DP_P2P="1" if p2p else "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from test_gpu_dp import _small_model
        from realpdebench_amd.trainer import ArenaTrainer, make_trainer
        torch.cuda.set_device(0)
        model, shape = _small_model(kind)
        model = model.cuda()
        tr = make_trainer(model, lr=1e-3, num_update=10)
        assert isinstance(tr, ArenaTrainer) and tr.world == 2 and (tr.peer is not None) == p2p
        g = torch.Generator().manual_seed(8)
        x, y = torch.randn(4, *shape, generator=g), torch.randn(4, *shape, generator=g)
        idx = list(range(rank * 2, rank * 2 + 2))
        xs, ys = x[idx].cuda(), y[idx].cuda()
        losses = [float(tr.step(xs, ys)) for _ in range(2)]
        model.eval()
        with torch.no_grad():
            ev = model(xs).cpu()                          # the forward pre-hook waits for the peers' slices of the second update
        ck = tr.checkpoint()
        torch.cuda.synchronize()
        if p2p:
            tr.peer.check()
        dist.barrier()
        out[rank] = {"flat": tr.flat.cpu(), "loss": losses, "eval": ev}
        tr.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["transolver", "unet"])
def test_arena_trainer_peer_pointer_step_equals_allreduce_step(kind):
    """RPB_DP_P2P=1 with the nn.Parameter models (ArenaTrainer: one flat arena per rank, no buckets, rpb_dp_p2p_adam) == the gloo
    all-reduce + rpb_adam_step path, bit for bit over two steps with two ranks on one GPU."""
    res = {}
    for p2p in (False, True):
        world, port = 2, _free_port()
        with mp.Manager() as mgr:
            out = mgr.dict()
            mp.spawn(_arena_worker, args=(world, port, kind, p2p, out), nprocs=world, join=True)
            res[p2p] = {k: v for k, v in out.items()}
    assert torch.equal(res[True][0]["flat"], res[True][1]["flat"])
    assert torch.equal(res[True][0]["flat"], res[False][0]["flat"]), float((res[True][0]["flat"] - res[False][0]["flat"]).abs().max())
    assert res[True][0]["loss"] == res[False][0]["loss"] and res[True][1]["loss"] == res[False][1]["loss"]
    assert torch.equal(res[True][1]["eval"], res[False][1]["eval"])
