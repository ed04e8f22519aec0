#!/usr/bin/env python
"""Per-kernel HIP-event table of one FNO3d eval forward at the headline shape:  python tools/fwd_probe.py [B]   (RPB_ARITH=f16x2: opt-in arithmetic)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realpdebench_amd import _lib  # noqa: E402
from realpdebench_amd.model.fno import FNO3d  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
if len(sys.argv) > 2 and sys.argv[2].startswith("comb"):       # BASELINE.json configs[4]: combustion volume, optional bf16 storage
    m = FNO3d(4, 16, 16, 4, 64, (64, 64, 64, 16), (64, 64, 64, 16)).cuda().eval()
    x = torch.randn(B, 64, 64, 64, 16, device="cuda")
    if sys.argv[2].endswith("bf16"):
        m.set_storage("bf16")
else:
    m = FNO3d(4, 12, 16, 4, 64, (20, 128, 128, 2), (20, 128, 128, 2)).cuda().eval()
    x = torch.randn(B, 20, 128, 128, 2, device="cuda")
if os.environ.get("RPB_ARITH"):                                 # e.g. RPB_ARITH=f16x2: the opt-in eval arithmetic
    m.set_arith(os.environ["RPB_ARITH"])
with torch.no_grad():
    for _ in range(2):
        m(x)
    torch.cuda.synchronize()
    _lib.PROFILE, _lib.PROFILE_ONLY = {}, None
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
prof = _lib.profile_summary()
tot = sum(v["total_ms"] for v in prof.values()) / 3
for label, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
    print(f"{label:44s} calls/fwd {v['calls'] / 3:5.1f}  avg {v['total_ms'] / v['calls']:7.3f} ms  {100 * v['total_ms'] / 3 / tot:5.1f}%  "
          f"{v['bytes'] * v['calls'] / v['total_ms'] / 1e6:8.1f} GB/s  {v['flops'] * v['calls'] / v['total_ms'] / 1e9:7.2f} TF/s")
print(f"kernel time per forward {tot:.3f} ms")
