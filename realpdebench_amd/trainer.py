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
