#!/usr/bin/env python
"""Transolver forward timing at the reference's cylinder config (N = 20*64*128 = 163840 tokens, hidden 256)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realpdebench_amd import _lib  # noqa: E402
from realpdebench_amd.model.transolver import Transolver  # noqa: E402

B = int(os.environ.get("TB_B", 4))
m = Transolver(space_dim=3, n_layers=1, n_hidden=256, n_head=8, fun_dim=0, out_dim=3, slice_num=16, mlp_ratio=4,
               H=128, W=64, D=20, dropout=0.1).cuda().eval()
x = torch.randn(B, 20, 64, 128, 3, device="cuda")
with torch.no_grad():
    m(x)
    _lib.PROFILE = {}
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        m(x)
    e.record()
    torch.cuda.synchronize()
ms = s.elapsed_time(e) / 3
print(f"Transolver forward B={B}: {ms:.2f} ms -> {B * 20 / ms * 1e3:.0f} fields/s, {B / ms * 1e3:.2f} samples/s")
tot = sum(v["total_ms"] for v in _lib.profile_summary().values())
for k, v in sorted(_lib.profile_summary().items(), key=lambda kv: -kv[1]["total_ms"]):
    print(f"{k:40s} calls {v['calls'] / 3:5.1f} avg {v['avg_ms']:8.3f} ms {100 * v['total_ms'] / tot:5.1f}%  "
          f"{v['bytes'] / v['avg_ms'] / 1e6:8.1f} GB/s {v['flops'] / v['avg_ms'] / 1e9:7.2f} TF/s")

# ---- training step through the drop-in protocol (train_loss(...).mean().backward() + torch.optim.Adam)
_lib.PROFILE = None
mt = Transolver(space_dim=3, n_layers=1, n_hidden=256, n_head=8, fun_dim=0, out_dim=3, slice_num=16, mlp_ratio=4,
                H=128, W=64, D=20, dropout=0.0).cuda().train()
opt = torch.optim.Adam(mt.parameters(), lr=7e-4)
y = torch.randn(B, 20, 64, 128, 3, device="cuda")


def step():
    opt.zero_grad()
    loss = mt.train_loss(x, y).mean()
    loss.backward()
    opt.step()
    return loss


step()
_lib.PROFILE = {}
torch.cuda.synchronize()
s.record()
for _ in range(3):
    step()
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / 3
print(f"Transolver train step B={B}: {ms:.2f} ms -> {B / ms * 1e3:.2f} samples/s  (peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB)")
tot = sum(v["total_ms"] for v in _lib.profile_summary().values())
for k, v in sorted(_lib.profile_summary().items(), key=lambda kv: -kv[1]["total_ms"]):
    print(f"{k:40s} calls {v['calls'] / 3:5.1f} avg {v['avg_ms']:8.3f} ms {100 * v['total_ms'] / tot:5.1f}%  "
          f"{v['bytes'] / v['avg_ms'] / 1e6:8.1f} GB/s {v['flops'] / v['avg_ms'] / 1e9:7.2f} TF/s")
