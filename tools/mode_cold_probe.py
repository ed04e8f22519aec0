#!/usr/bin/env python
"""Mode contraction with COLD weights (the training step reads each layer's 100 MB once per pass): six weight buffers are cycled so that a
buffer has left the 256 MB Infinity Cache when its turn comes again; HIP events around every launch.   python tools/mode_cold_probe.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realpdebench_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
C, M = 64, 8 * 24 * 16
f = dict(device="cuda", dtype=torch.float32)
X, Y = torch.randn(B * 2 * M * C, **f), torch.empty(B * 2 * M * C, **f)
Ws = [torch.randn(M * C * C * 2, **f) for _ in range(6)]
G = torch.empty(M * C * C * 2, **f)
for name, fn in (("fwd", lambda W: ops.mode_contract_fwd(X, W, Y, B, M, C)), ("dgrad", lambda W: ops.mode_contract_dgrad(X, W, Y, B, M, C)),
                 ("wgrad", lambda W: ops.mode_contract_wgrad(X, Y, G, B, M, C))):
    for W in Ws:
        fn(W)
    torch.cuda.synchronize()
    evs = []
    for rep in range(4):
        for W in Ws:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fn(W)
            e.record()
            evs.append((s, e))
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in evs)
    print(f"mode_contract_{name:6s} B={B:3d} cold weights: median {ms[len(ms) // 2]:.4f} ms  min {ms[0]:.4f}  max {ms[-1]:.4f}   ({100.66 / ms[len(ms) // 2] / 1e3:.2f} TB/s of the weight tile alone)")
