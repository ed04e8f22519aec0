#!/usr/bin/env python
"""U-Net on the fsi-shaped C3 mesh (256 x 256, dim = H = 256, bottleneck 64 x 64 = 4096 tokens per frame): one training step
and one forward at B = 1, to show that the configuration runs and what it costs (C3_T frames, default 20)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realpdebench_amd.model.unet import Unet3d  # noqa: E402

T = int(os.environ.get("C3_T", 20))
B = int(os.environ.get("C3_B", 1))
m = Unet3d(dim=256, out_channels=3, dim_mults=[1, 2, 4], channels=3, in_time=T, out_time=T).cuda().train()
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
x, y = torch.randn(B, T, 256, 256, 3, device="cuda"), torch.randn(B, T, 256, 256, 3, device="cuda")
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss = m.train_loss(x, y).mean()
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    print(f"step {it}: {1e3 * (time.perf_counter() - t0):.1f} ms, loss {float(loss):.4f}, "
          f"peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
