#!/usr/bin/env python
"""Wall time per eval forward with and without hipGraph replay:  python tools/graph_probe.py [headline|comb|comb_bf16] [B]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import realpdebench_amd.model.fno as F  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "headline"
if which.startswith("comb"):
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    m = F.FNO3d(4, 16, 16, 4, 64, (64, 64, 64, 16), (64, 64, 64, 16)).cuda().eval()
    x = torch.randn(B, 64, 64, 64, 16, device="cuda")
    if which.endswith("bf16"):
        m.set_storage("bf16")
else:
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    m = F.FNO3d(4, 12, 16, 4, 64, (20, 128, 128, 2), (20, 128, 128, 2)).cuda().eval()
    x = torch.randn(B, 20, 128, 128, 2, device="cuda")
res = {}
with torch.no_grad():
    for graph in (False, True):
        F._EVAL_GRAPH = graph
        m._ws = {k: v for k, v in m._ws.items() if False}
        for _ in range(4):
            y = m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            y = m(x)
        torch.cuda.synchronize()
        res[graph] = (1e3 * (time.perf_counter() - t0) / 20, y.clone())
print(f"{which} B={B}: eager {res[False][0]:.3f} ms/forward, graph {res[True][0]:.3f} ms/forward, "
      f"outputs equal: {torch.equal(res[False][1], res[True][1])}")
