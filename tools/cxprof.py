#!/usr/bin/env python
"""One rpb_conv3x launch at the Transolver shape (for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realpdebench_amd import ops  # noqa: E402

B, mesh, Ci, N = 4, (20, 64, 128), 256, 512
if os.environ.get("CX_SHAPE") == "unet":
    B, mesh, Ci, N = 12, (20, 64, 128), 64, 64
M = B * mesh[0] * mesh[1] * mesh[2]
x = torch.randn(M, Ci, device="cuda")
w = torch.randn(N, 27 * Ci, device="cuda") / (27 * Ci) ** 0.5
planes = torch.empty(3 * M * Ci, dtype=torch.int16, device="cuda")
wz = torch.empty(3 * N * 27 * Ci, dtype=torch.int16, device="cuda")
y = torch.empty(M, N, device="cuda")
ops.split3(x, planes, M, Ci)
ops.conv3x_wprep(w, wz, N, Ci)
for _ in range(2):
    ops.conv3x(planes, wz, y, M, N, Ci, mesh)
torch.cuda.synchronize()
if os.environ.get("CX_MODE") == "wgrad":
    Co = N
    g = torch.randn(M, Co, device="cuda")
    for _ in range(2):
        ops.conv3_wgrad_parts(g, x, M, Co, Ci, mesh)
    torch.cuda.synchronize()
