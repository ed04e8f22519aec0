#!/usr/bin/env python
"""U-Net timing at the reference's cylinder config (configs/cylinder/unet.yaml: [12,20,64,128,3], dim = H = 64 ->
64/128/256 channels): eval forward and the training step through the drop-in protocol, with the per-kernel HIP-event
table.   UB_B=<batch> overrides B."""
import os
import sys

import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realpdebench_amd import _lib  # noqa: E402
from realpdebench_amd.model.unet import Unet3d  # noqa: E402

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(root, "realpdebench_amd", "configs", "cylinder", "unet.yaml")) as fh:
    cfg = yaml.safe_load(fh)
B = int(os.environ.get("UB_B", cfg["train_batch_size"]))
ITERS = int(os.environ.get("UB_ITERS", 2))
T, H, W, C = cfg["shape_in"]
m = Unet3d(dim=H, out_channels=cfg["shape_out"][-1], dim_mults=cfg["dim_mults"], channels=C, in_time=T,
           out_time=cfg["shape_out"][0]).cuda()
x = torch.randn(B, T, H, W, C, device="cuda")
y = torch.randn(B, *cfg["shape_out"], device="cuda")
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def table(n):
    ps = _lib.profile_summary()
    tot = sum(v["total_ms"] for v in ps.values())
    for k, v in sorted(ps.items(), key=lambda kv: -kv[1]["total_ms"])[:n]:
        print(f"{k:40s} calls {v['calls'] / ITERS:6.1f} avg {v['avg_ms']:8.3f} ms {100 * v['total_ms'] / tot:5.1f}%  "
              f"{v['bytes'] / v['avg_ms'] / 1e6:8.1f} GB/s {v['flops'] / v['avg_ms'] / 1e9:7.2f} TF/s")
    print(f"sum of HIP-event kernel times: {tot / ITERS:.1f} ms per iteration")


m.eval()
with torch.no_grad():
    m(x)
    _lib.PROFILE = {}
    torch.cuda.synchronize()
    s.record()
    for _ in range(ITERS):
        m(x)
    e.record()
    torch.cuda.synchronize()
ms = s.elapsed_time(e) / ITERS
print(f"U-Net forward B={B}: {ms:.2f} ms -> {B * cfg['shape_out'][0] / ms * 1e3:.0f} fields/s, {B / ms * 1e3:.2f} samples/s")
table(16)

_lib.PROFILE = None
m.train()
opt = torch.optim.Adam(m.parameters(), lr=cfg["lr"])


def step():
    opt.zero_grad()
    loss = m.train_loss(x, y).mean()
    loss.backward()
    opt.step()
    return loss


step()
_lib.PROFILE = {}
torch.cuda.synchronize()
s.record()
for _ in range(ITERS):
    step()
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / ITERS
print(f"U-Net train step B={B}: {ms:.2f} ms -> {B / ms * 1e3:.2f} samples/s  "
      f"(peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB)")
table(28)
