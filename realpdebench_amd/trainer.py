"""Fused training step for the MI355X backend -- the reference's hot loop (realpdebench/train.py:321-334):

    optimizer.zero_grad(); loss = model.train_loss(input, target).mean(); loss.backward();
    optimizer.step(); scheduler.step()

run as one stream-ordered chain of HIP kernels with no host synchronisation: forward, fused MSE +
dLoss/dpred, backward into a persistent flat gradient arena, (RCCL all-reduce when data-parallel), one
Adam launch over the flat parameter arena, closed-form LR schedule on the host.  ``loss`` is returned as a
device scalar; callers decide when to ``.item()`` it (the reference syncs 3-4 times per step, train.py:335-342).
"""
import math

import torch

from . import ops


class Trainer:
    def __init__(self, model, lr, num_update, scheduler="cosine", step_size=1000, betas=(0.9, 0.999), eps=1e-8,
                 clip_grad_norm=0.0):
        if clip_grad_norm and clip_grad_norm > 0:
            raise NotImplementedError("clip_grad_norm > 0 is not used by any FNO config of the reference "
                                      "(configs/*/fno.yaml: clip_grad_norm: 0.) and is not implemented")
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
        out = model._forward_impl(x, ws, training=True)                       # [ncrop][DO]
        tgt = model._unshape_grad(target.contiguous().float(), B)
        n = out.numel()
        ops.mse(out, tgt, None, ws.gout, ws.mse_part, n, 2.0 / 2.0 / (n * world))   # gout = 2*(p-t)/N_global
        ops.reduce_partials(ws.mse_part, ws.mse_part.numel(), 1, out_f32=ws.loss, scale=1.0 / n)
        if dp is not None:
            dp.begin_step(self.grad)
        model._backward_impl(x, ws.gout, ws, self.grad)
        if dp is not None:
            dp.finish_step(self.grad)
        lr = self.current_lr()
        self.iteration += 1
        ops.adam_step(model.flat.data, self.grad, self.exp_avg, self.exp_avg_sq, model.flat.numel(), lr,
                      self.betas[0], self.betas[1], self.eps, self.iteration)
        return ws.loss

    # ---- checkpoint in the reference's format (train.py:410-418)
    def checkpoint(self, extra=None):
        ck = {"model_state_dict": self.model.state_dict(), "iteration": self.iteration}
        if extra:
            ck.update(extra)
        return ck


class ProtocolTrainer:
    """The reference's loop body verbatim in structure (train.py:323-334) for models without a flat arena
    (Transolver, Galerkin Transformer): ``zero_grad``; ``train_loss(...).mean().backward()`` runs the HIP forward/backward through the model's
    autograd Function; Adam + LR schedule are ``torch.optim`` (4 M parameters: off the critical path).  Under data
    parallelism the gradients are averaged with one RCCL all-reduce per parameter after backward."""

    def __init__(self, model, lr, num_update, scheduler="cosine", step_size=1000, clip_grad_norm=0.0, dp_group=None):
        self.model = model
        self.opt = torch.optim.Adam(model.parameters(), lr=lr)
        if scheduler == "step":
            self.sched = torch.optim.lr_scheduler.StepLR(self.opt, step_size=step_size, gamma=0.5)
        elif scheduler == "cosine":
            self.sched = torch.optim.lr_scheduler.CosineAnnealingLR(self.opt, T_max=num_update)
        else:
            raise ValueError(f"Scheduler {scheduler} not supported")
        self.clip = float(clip_grad_norm or 0.0)
        self.iteration = 0
        self.world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world = torch.distributed.get_world_size(dp_group)
            for p in list(model.parameters()) + list(model.buffers()):   # one init for everyone, like the single-process reference
                torch.distributed.broadcast(p.data, src=0, group=dp_group)
            core = getattr(model, "regressor", None)
            if core is not None and hasattr(core, "dp") and self.world > 1:
                from .dp import StatsSync
                core.dp = StatsSync(dp_group)                    # BatchNorm3d over the global batch (SyncBN)
        self.group = dp_group

    def current_lr(self):
        return self.opt.param_groups[0]["lr"]

    def step(self, input, target):
        self.model.train()
        self.opt.zero_grad()
        loss = self.model.train_loss(input, target).mean()
        loss.backward()
        if self.world > 1:
            works = [torch.distributed.all_reduce(p.grad, group=self.group, async_op=True)
                     for p in self.model.parameters() if p.grad is not None]
            for w in works:
                w.wait()
            for p in self.model.parameters():
                if p.grad is not None:
                    p.grad.div_(self.world)
        if self.clip > 0:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip)
        self.opt.step()
        self.sched.step()
        self.iteration += 1
        return loss.detach().reshape(1)


def make_trainer(model, lr, num_update, scheduler="cosine", step_size=1000, clip_grad_norm=0.0):
    """Fused arena trainer for FNO3d, protocol trainer for everything else."""
    if hasattr(model, "flat"):
        if torch.distributed.is_available() and torch.distributed.is_initialized() and model.dp is None:
            from .dp import DataParallel
            DataParallel(model)
        return Trainer(model, lr=lr, num_update=num_update, scheduler=scheduler, step_size=step_size,
                       clip_grad_norm=clip_grad_norm)
    return ProtocolTrainer(model, lr=lr, num_update=num_update, scheduler=scheduler, step_size=step_size,
                           clip_grad_norm=clip_grad_norm)
