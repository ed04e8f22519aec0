#!/usr/bin/env python
"""The strong-scaling proxy's first phase alone, in a fresh process: FNO3d train step at B = 32 (no DP), then B = 16 / 8 / 4 with the
whole DP path on a one-rank RCCL group (buckets on the side stream, SyncBN reductions inline).  A/B switch for the side stream's
priority: RPB_DP_SIDE_PRIORITY=0.   python tools/dp_strong_probe.py [steps]"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realpdebench_amd.dp import DataParallel  # noqa: E402
from realpdebench_amd.model.fno import FNO3d  # noqa: E402
from realpdebench_amd.trainer import Trainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29521")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
shape, modes, width, L = (20, 128, 128, 2), (4, 12, 16), 64, 4


def run(B, dp):
    torch.manual_seed(0)
    model = FNO3d(*modes, L, width, shape, shape).to(dev)
    if dp:
        DataParallel(model)
        model.dp.sync_stats_always = True
    tr = Trainer(model, lr=1e-4, num_update=4000)
    x, y = torch.randn(B, *shape, device=dev), torch.randn(B, *shape, device=dev)
    for _ in range(2):
        tr.step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(x, y)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    tr.close()
    del tr, model, x, y
    torch.cuda.empty_cache()
    return ms


for B, dp in ((32, False), (4, False), (16, True), (8, True), (4, True), (4, False)):
    print(f"priority={os.environ.get('RPB_DP_SIDE_PRIORITY', 'high')}  B={B:2d} dp={int(dp)}  {run(B, dp):7.2f} ms/step", flush=True)
dist.destroy_process_group()
