#!/usr/bin/env python
"""Per-kernel HIP-event table of one FNO3d train step at the reference's configs/fsi/fno.yaml (width 128, modes (4,16,16),
[32,20,64,64,3] -> padded 26 x 70 x 70) -- the width-128 instance of the Fourier layer (also the Galerkin regressor's)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realpdebench_amd import _lib  # noqa: E402
from realpdebench_amd.model.fno import FNO3d  # noqa: E402
from realpdebench_amd.trainer import Trainer  # noqa: E402

B = int(os.environ.get("KB_B", 32))
shape, modes, width, L = (20, 64, 64, 3), (4, 16, 16), 128, 4
torch.manual_seed(0)
m = FNO3d(*modes, L, width, shape, shape).cuda()
tr = Trainer(m, lr=1e-4, num_update=4000)
x, y = torch.randn(B, *shape, device="cuda"), torch.randn(B, *shape, device="cuda")
for _ in range(2):
    tr.step(x, y)
torch.cuda.synchronize()
_lib.PROFILE, _lib.PROFILE_ONLY = {}, None
tr.step(x, y)
torch.cuda.synchronize()
prof = _lib.profile_summary()
_lib.PROFILE = None
tot = sum(v["total_ms"] for v in prof.values())
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
    print(f"{k:48s} calls {v['calls']:3d}  avg {v['avg_ms']:8.3f} ms  {100 * v['total_ms'] / tot:5.1f}%  "
          f"{v['bytes'] / v['avg_ms'] / 1e6:8.1f} GB/s  {v['flops'] / v['avg_ms'] / 1e9:7.2f} TF/s")
print(f"kernel time per step {tot:.2f} ms")
