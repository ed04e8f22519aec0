"""Fused training step for the MI355X backend -- the reference's hot loop (realpdebench/train.py:321-334):

    optimizer.zero_grad(); loss = model.train_loss(input, target).mean(); loss.backward();
    optimizer.step(); scheduler.step()

run as one stream-ordered chain of HIP kernels with no host synchronisation: forward, fused MSE +
dLoss/dpred, backward into a persistent flat gradient arena, (RCCL all-reduce when data-parallel), one
Adam launch over the flat parameter arena, closed-form LR schedule on the host.  ``loss`` is returned as a
device scalar; callers decide when to ``.item()`` it (the reference syncs 3-4 times per step, train.py:335-342).
"""
import math
import os

import torch

from . import ops


class Trainer:
    def __init__(self, model, lr, num_update, scheduler="cosine", step_size=1000, betas=(0.9, 0.999), eps=1e-8,
                 clip_grad_norm=0.0):
        self.clip = float(clip_grad_norm or 0.0)       # train.py:330-331 (0 in every shipped FNO YAML)
        if scheduler not in ("cosine", "step"):
            raise ValueError(f"Scheduler {scheduler} not supported")          # train.py:296
        self.model = model
        self.lr0, self.num_update, self.scheduler, self.step_size = float(lr), int(num_update), scheduler, int(step_size)
        self.betas, self.eps = betas, eps
        self.iteration = 0           # number of optimizer steps taken
        flat = model.flat
        self.grad = torch.zeros_like(flat.data)
        self.exp_avg = torch.zeros_like(flat.data)
        self.exp_avg_sq = torch.zeros_like(flat.data)
        self._owned = None           # sharded optimizer step: the device table of the ranges this rank updates

    # ---- learning-rate schedule (closed forms of CosineAnnealingLR(T_max=num_update) / StepLR(gamma=0.5))
    def current_lr(self):
        k = self.iteration
        if self.scheduler == "cosine":
            return self.lr0 * (1.0 + math.cos(math.pi * k / self.num_update)) / 2.0
        return self.lr0 * (0.5 ** (k // self.step_size))

    def step(self, input, target):
        """One training iteration on pre-processed (normalised) device tensors.  Returns the loss (device scalar)."""
        model = self.model
        model.train()
        x = model._check_input(input)
        B = x.shape[0]
        ws = model._workspace(B, True, x.device)
        dp = model.dp
        world = dp.world_size if dp is not None else 1
        # sharded optimizer step: THIS trainer consumes reduce-scattered gradients, so it is the one that asks for them (dp.begin_step)
        sharded = dp is not None and bool(getattr(dp, "shard_opt", False))
        # peer-pointer step (opt-in): no bucket travels during backward; the reduction happens inside the update kernel below
        p2p = dp is not None and bool(getattr(dp, "p2p_opt", False))
        if p2p:
            if self.clip > 0:
                raise NotImplementedError("DataParallel(p2p=True) has no gradient-norm clipping (the norm needs the reduced gradient "
                                          "before the update); use the all-reduce or the sharded path with clip_grad_norm")
            sharded = False
            dp.peer_setup(model.flat.data, self.grad)
        tgt = model._unshape_grad(target.contiguous().float(), B)
        n = tgt.numel()
        if getattr(ws, "head_fused", False) and ws.head_loss_fused:
            # one-launch head (csrc/rpb_pjf.hip): its forward, the squared error and dLoss/dout = 2 (pred - target) / N_global are formed
            # inside the head's backward kernel -- no rpb_proj_fwd, no rpb_mse, no pred tensor in the training step
            model._forward_impl(x, ws, training=True, skip_head=True)
            if dp is not None:
                dp.begin_step(self.grad, sharded=sharded, p2p=p2p)
            model._backward_impl(x, None, ws, self.grad, target=tgt, gscale=2.0 / (n * world))
            ops.reduce_partials(ws.hb_loss_part, ws.hb_slots, 1, out_f32=ws.loss, scale=1.0 / n)
        else:
            out = model._forward_impl(x, ws, training=True)                   # [ncrop][DO]
            ops.mse(out, tgt, None, ws.gout, ws.mse_part, n, 2.0 / 2.0 / (n * world))   # gout = 2*(p-t)/N_global
            ops.reduce_partials(ws.mse_part, ws.mse_part.numel(), 1, out_f32=ws.loss, scale=1.0 / n)
            if dp is not None:
                dp.begin_step(self.grad, sharded=sharded, p2p=p2p)
            model._backward_impl(x, ws.gout, ws, self.grad)
        if dp is not None:
            dp.finish_step(self.grad)
        gscale = 1.0
        if self.clip > 0:
            # torch.nn.utils.clip_grad_norm_ (train.py:330-331): total 2-norm over all parameters (complex weights count both
            # parts: the arena holds them as 2 x fp32), coefficient min(1, max_norm / (norm + 1e-6)) folded into the Adam kernel's
            # gradient scale; one host sync, only when clipping is on.  The arena's alignment gaps are zero.
            # Sharded step: the arena is only summed in the pieces this rank owns -> norm^2 partial per owned range + one inline
            # fp64 all-reduce (dp.sharded_grad_norm).
            norm = dp.sharded_grad_norm(self.grad) if sharded else float(torch.linalg.vector_norm(self.grad))
            gscale = min(1.0, self.clip / (norm + 1e-6))
        lr = self.current_lr()
        self.iteration += 1
        if p2p:
            # (finish_step above enqueued nothing: rank r's slice of the W gradient arenas is summed inside the update kernel, the new
            #  parameters are stored into all W parameter arenas; the next reader of the parameters waits in dp.params_ready*)
            dp.peer.adam(self.exp_avg, self.exp_avg_sq, lr, self.betas[0], self.betas[1], self.eps, self.iteration, gscale)
            dp._pending = True
        elif sharded:
            # sharded optimizer step (dp.DataParallel): the gradient chunks were reduce-scattered, this rank updates the pieces it owns
            # (1 / world of the arena in ONE launch over the range table) and the parameter pieces travel back on the side stream while
            # the next forward pass starts (it waits per layer: model._forward_impl -> dp.params_ready)
            if self._owned is None:
                self._owned = dp.owned_table(model.flat.device)
            tab, nr, total = self._owned
            ops.adam_step_ranges(model.flat.data, self.grad, self.exp_avg, self.exp_avg_sq, tab, nr, total, lr,
                                 self.betas[0], self.betas[1], self.eps, self.iteration, gscale)
            dp.gather_params(model.flat.data)
        else:
            ops.adam_step(model.flat.data, self.grad, self.exp_avg, self.exp_avg_sq, model.flat.numel(), lr,
                          self.betas[0], self.betas[1], self.eps, self.iteration, gscale)
        return ws.loss

    def close(self):
        """Teardown: destroy the RCCL communicators of the data-parallel wrapper (also done at interpreter exit)."""
        if self.model.dp is not None and hasattr(self.model.dp, "close"):
            self.model.dp.close()

    # ---- checkpoint in the reference's format (train.py:410-418)
    def checkpoint(self, extra=None):
        if self.model.dp is not None and hasattr(self.model.dp, "params_ready_all"):
            self.model.dp.params_ready_all()                 # a sharded step may still be gathering parameter pieces on the side stream
        ck = {"model_state_dict": self.model.state_dict(), "iteration": self.iteration}
        if extra:
            ck.update(extra)
        return ck


class ArenaTrainer:
    """Training step for the models whose parameters are ordinary ``nn.Parameter`` tensors (Transolver, Galerkin Transformer,
    U-Net) -- the reference's loop body (train.py:323-334) with the optimizer and the gradient exchange on the arena design
    of the FNO trainer:

    * every parameter is re-homed into ONE flat fp32 arena (``p.data`` becomes a view; ``state_dict`` / checkpoints are
      unchanged), gradients and the two Adam moments are arenas of the same layout, so Adam is ONE ``rpb_adam_step`` launch
      (torch.optim.Adam semantics: defaults, complex weights as 2 x fp32, parameters without a gradient are left alone
      because a zero gradient moves nothing) and the LR schedule is the closed form of CosineAnnealingLR / StepLR;
    * ``train_loss(...).mean().backward()`` runs the model's HIP forward / backward through its autograd Function;
    * data parallel: gradients are summed over ranks in a few large contiguous buckets by RCCL on a side HIP stream
      (``rpb_dp_allreduce_*``).  A model may announce a finished gradient from INSIDE its backward pass through
      ``model._dp_early(param, grad)`` -- the Galerkin Transformer does so for its 670 MB spectral regressor, whose
      all-reduce then overlaps the rest of the backward; the remaining buckets go out right after backward and Adam waits.
      The sum is turned into the mean inside the Adam kernel (``gscale = 1 / world``).
    """

    BUCKET_ELEMS = 16 * 1024 * 1024          # 64 MB buckets: large enough for xGMI ring bandwidth, several in flight

    def __init__(self, model, lr, num_update, scheduler="cosine", step_size=1000, betas=(0.9, 0.999), eps=1e-8,
                 clip_grad_norm=0.0, dp_group=None, micro_batch=None, p2p=None):
        if scheduler not in ("cosine", "step"):
            raise ValueError(f"Scheduler {scheduler} not supported")
        # micro_batch: a step over B samples runs as ceil(B / micro_batch) forward / backward passes whose gradients accumulate
        # before ONE all-reduce and ONE Adam update -- the same step, exactly, for models whose samples do not interact
        # (GroupNorm / LayerNorm: U-Net, Transolver, DPOT).  This is how BASELINE.json configs[2] (U-Net on the 256^2 fsi mesh at
        # 16 samples per GPU) fits: fp32 activations of ONE such sample take 116 GiB.
        self.micro_batch = int(micro_batch) if micro_batch else None
        if self.micro_batch and not getattr(model, "batch_independent", False):
            raise ValueError("micro_batch needs a model without batch statistics (BatchNorm couples the samples of a step): "
                             f"{type(model).__name__} does not declare batch_independent")
        self.model = model
        self.lr0, self.num_update, self.scheduler, self.step_size = float(lr), int(num_update), scheduler, int(step_size)
        self.betas, self.eps, self.clip = betas, eps, float(clip_grad_norm or 0.0)
        self.iteration = 0
        self.params = [p for p in model.parameters() if p.requires_grad]
        if any(p.dtype != torch.float32 for p in self.params):
            raise TypeError("ArenaTrainer: fp32 parameters only")
        dev = self.params[0].device
        offs, off = [], 0
        for p in self.params:
            offs.append(off)
            off += (p.numel() + 63) // 64 * 64                     # 256 B-aligned segments
        self.offsets, self.total = offs, off
        self.flat = torch.zeros(off, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                self.flat[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = self.flat[o:o + p.numel()].view(p.shape)   # the parameter now LIVES in the arena
        self.grad = torch.zeros_like(self.flat)
        self.gviews = [self.grad[o:o + p.numel()].view(p.shape) for p, o in zip(self.params, offs)]
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        # ---- data parallel
        self.world, self.comm, self.group = 1, None, dp_group
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world = torch.distributed.get_world_size(dp_group)
        if self.world > 1:
            from .dp import RcclComm, StatsSync, use_rccl_abi
            torch.distributed.broadcast(self.flat, src=0, group=dp_group)      # one init for everyone, like the reference
            for b in model.buffers():
                torch.distributed.broadcast(b.data, src=0, group=dp_group)
            if use_rccl_abi(self.flat, dp_group):
                self.comm = RcclComm(dp_group)
            core = getattr(model, "regressor", None)
            if core is not None and hasattr(core, "dp"):
                core.dp = StatsSync(dp_group, comm=self.comm)       # BatchNorm3d over the global batch (SyncBN)
        # p2p (opt-in, RPB_DP_P2P=1): gradients and parameters travel over peer pointers inside the update kernel (dp.PeerExchange,
        # rpb_dp_p2p_*) instead of all-reduce buckets + a full-arena Adam on every rank; nothing is enqueued during backward
        self.peer = None
        want_p2p = bool(p2p) if p2p is not None else os.environ.get("RPB_DP_P2P") == "1"
        if want_p2p and self.world > 1:
            if self.clip > 0:
                raise NotImplementedError("the peer-pointer optimizer step has no gradient-norm clipping; use the all-reduce path")
            from .dp import PeerExchange
            self.peer = PeerExchange(self.flat, self.grad, dp_group)
            # evaluation through model(x) straight after a step must see every rank's slice (train_loss -> forward inside step() is
            # covered by the explicit wait at the top of step())
            model.register_forward_pre_hook(lambda _m, _a: self.settle())
        self._early = {}            # param -> its gradient tensor, all-reduce already in flight
        self._works = []
        self._had_grad = [False] * len(self.params)

    def settle(self):
        """Peer-pointer step: the current stream waits until every rank has stored its slice into this rank's parameter arena
        (no-op otherwise / when nothing is pending).  Called before every read of the parameters outside ``step``."""
        if self.peer is not None:
            self.peer.params_wait()

    def close(self):
        """Teardown: destroy the RCCL communicators (also done at interpreter exit)."""
        if self.comm is not None:
            self.comm.close()
            self.comm = None
        if self.peer is not None:
            self.peer.close()
            self.peer = None

    def current_lr(self):
        k = self.iteration
        if self.scheduler == "cosine":
            return self.lr0 * (1.0 + math.cos(math.pi * k / self.num_update)) / 2.0
        return self.lr0 * (0.5 ** (k // self.step_size))

    # ---- gradient exchange
    def _reduce(self, t):
        if self.comm is not None:
            self.comm.enqueue(t)
        else:
            self._works.append(torch.distributed.all_reduce(t, group=self.group, async_op=True))

    def _dp_early(self, param, grad):
        """Called by a model from inside its backward pass: ``grad`` (the final gradient of ``param``) starts its all-reduce now."""
        if self.world > 1 and grad.is_contiguous():
            self._reduce(grad.view(-1))
            self._early[param] = grad

    def step(self, input, target):
        model = self.model
        model.train()
        self.settle()                                    # (peer-pointer step: the previous update's slices from the other ranks)
        for p in self.params:
            p.grad = None
        self._early, self._works = {}, []
        B = input.shape[0]
        if self.micro_batch and B > self.micro_batch:
            model._dp_early = None                       # gradients are final only after the last micro-batch
            loss = None
            for i in range(0, B, self.micro_batch):
                xi, yi = input[i:i + self.micro_batch], target[i:i + self.micro_batch]
                li = model.train_loss(xi, yi).mean() * (xi.shape[0] / B)      # mean over the step = weighted mean of the parts
                li.backward()                            # autograd accumulates into p.grad
                loss = li.detach() if loss is None else loss + li.detach()
        else:
            model._dp_early = self._dp_early if (self.world > 1 and self.peer is None) else None
            loss = model.train_loss(input, target).mean()
            loss.backward()
            model._dp_early = None
        # ---- gradients -> arena (storage plumbing; parameters the loss does not reach keep a zero gradient)
        late_dst, late_src, early = [], [], []
        for i, (p, gv) in enumerate(zip(self.params, self.gviews)):
            if p.grad is None:
                # torch.optim.Adam SKIPS a parameter without a gradient (no moment decay, no update, its own step count); the
                # one-launch arena update treats it as a zero gradient, which is the same thing only while the parameter has
                # never had one (moments stay 0: e.g. DPOT's cls_head, the Galerkin layer's unused `fc`).  A parameter that
                # loses its gradient mid-run would drift on stale momentum here -- refuse instead of diverging silently.
                if self._had_grad[i]:
                    raise RuntimeError(f"ArenaTrainer: parameter #{i} {tuple(p.shape)} received gradients in earlier steps but none "
                                       "in this one; the fused Adam launch cannot skip it the way torch.optim.Adam would "
                                       "(conditionally used parameters are not supported)")
                gv.zero_()
            elif p in self._early:
                # the tensor the model handed to _dp_early is the one being reduced in place; p.grad may be a COPY autograd made
                # when it accumulated the gradient (it cannot steal a tensor we hold a reference to), taken while the all-reduce
                # was still in flight on another stream
                early.append((gv, self._early[p]))
            else:
                late_dst.append(gv)
                late_src.append(p.grad)
            if p.grad is not None:
                self._had_grad[i] = True
        if late_dst:
            torch._foreach_copy_(late_dst, late_src)
        if self.world > 1 and self.peer is None:
            # everything not announced early: contiguous runs of the arena, cut into large buckets
            runs, start = [], None
            for p, o in zip(self.params, self.offsets):
                if p in self._early:
                    if start is not None:
                        runs.append((start, o))
                        start = None
                elif start is None:
                    start = o
            if start is not None:
                runs.append((start, self.total))
            for s0, e0 in runs:
                for b0 in range(s0, e0, self.BUCKET_ELEMS):
                    self._reduce(self.grad[b0:min(e0, b0 + self.BUCKET_ELEMS)])
            if self.comm is not None:
                self.comm.wait()
            for w in self._works:
                w.wait()
            for gv, g in early:            # reduced in place where the backward pass left them; now into the arena
                gv.copy_(g)
        elif early:
            for gv, g in early:
                gv.copy_(g)
        for p in self.params:
            p.grad = None
        gscale = 1.0 / self.world
        if self.clip > 0:                                            # train.py:330-331 (unused by the shipped YAMLs: one sync)
            norm = float(torch.linalg.vector_norm(self.grad)) * gscale
            gscale *= min(1.0, self.clip / (norm + 1e-6))
        lr = self.current_lr()
        self.iteration += 1
        if self.peer is not None:
            # rank r sums slice r of the W gradient arenas, updates it and stores it into all W parameter arenas (rpb_dp_p2p_adam)
            self.peer.adam(self.exp_avg, self.exp_avg_sq, lr, self.betas[0], self.betas[1], self.eps, self.iteration, gscale)
        else:
            ops.adam_step(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.total, lr, self.betas[0], self.betas[1],
                          self.eps, self.iteration, gscale)
        # the kernel wrote the arena through raw pointers: tell autograd's version counters, which the models' caches of
        # re-laid-out / split weights are keyed on (an in-place torch op would have done this)
        for p in self.params:
            torch.autograd.graph.increment_version(p)
        return loss.detach().reshape(1)

    def mean_grads(self):
        """{parameter: gradient of the last step averaged over ranks} -- views of the gradient arena (tests, diagnostics)."""
        return {p: gv / self.world for p, gv in zip(self.params, self.gviews)}

    def checkpoint(self, extra=None, optimizer=False):
        """The reference's checkpoint (train.py:410-418: model weights + bookkeeping, NO optimizer state -- a resumed run starts Adam
        from zero moments there, and here).  ``optimizer=True`` adds the two moment arenas for callers that want an exact resume."""
        self.settle()
        ck = {"model_state_dict": self.model.state_dict(), "iteration": self.iteration}
        if optimizer:
            ck["arena_adam_state"] = {"exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone()}
        if extra:
            ck.update(extra)
        return ck

    def load_optimizer_state(self, state, iteration=None):
        """Restore the moment arenas written by ``checkpoint(optimizer=True)``.  The "had a gradient before" flags that guard the
        one-launch Adam update against conditionally used parameters are re-derived from the second moments (a parameter has
        received a gradient iff its ``exp_avg_sq`` is non-zero), so the guard survives a resume (round-3 advisor finding)."""
        self.exp_avg.copy_(state["exp_avg"])
        self.exp_avg_sq.copy_(state["exp_avg_sq"])
        if iteration is not None:
            self.iteration = int(iteration)
        flags = torch.stack([self.exp_avg_sq[o:o + p.numel()].any() for p, o in zip(self.params, self.offsets)])
        self._had_grad = [bool(f) for f in flags.cpu()]


def make_trainer(model, lr, num_update, scheduler="cosine", step_size=1000, clip_grad_norm=0.0, micro_batch=None):
    """Fused trainer for FNO3d (one flat arena built into the model), ArenaTrainer for the nn.Parameter models."""
    if hasattr(model, "flat"):
        if torch.distributed.is_available() and torch.distributed.is_initialized() and model.dp is None:
            from .dp import DataParallel
            DataParallel(model)
        if micro_batch:
            raise ValueError("micro_batch: FNO3d's BatchNorm3d couples the samples of a step")
        return Trainer(model, lr=lr, num_update=num_update, scheduler=scheduler, step_size=step_size,
                       clip_grad_norm=clip_grad_norm)
    return ArenaTrainer(model, lr=lr, num_update=num_update, scheduler=scheduler, step_size=step_size,
                        clip_grad_norm=clip_grad_norm, micro_batch=micro_batch)
