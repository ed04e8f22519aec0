#!/usr/bin/env python
"""A/B of the weak-scaling step (B = 32 per rank) under MODELLED collectives on one GPU: how much of the slow-down bench.py's
`weak_scaling_B32_8rank_shape` reports is the side stream's kernels taking a CU slot away from compute kernels that launch exactly one
persistent workgroup per CU -- and how much of it chip-wide line claiming (RPB_LINE_CLAIM=2) gives back.

    python tools/dp_weak_probe.py [steps]            # prints one line per (claim mode, variant)

Variants: no DP wrapper | DP, one-rank group, no model | DP + ring model at RPB_PROXY_GBPS (200) | the same at half the rate.
"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realpdebench_amd import _lib  # noqa: E402
from realpdebench_amd.dp import DataParallel  # noqa: E402
from realpdebench_amd.model.fno import FNO3d  # noqa: E402
from realpdebench_amd.trainer import Trainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29519")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
shape, modes, width, L, B = (20, 128, 128, 2), (4, 12, 16), 64, 4, 32
gbps, lat = float(os.environ.get("RPB_PROXY_GBPS", "200")), float(os.environ.get("RPB_PROXY_LAT_US", "15"))
x, y = torch.randn(B, *shape, device=dev), torch.randn(B, *shape, device=dev)


def run(dp, rate, shard=False):
    torch.manual_seed(0)
    model = FNO3d(*modes, L, width, shape, shape).to(dev)
    comm = None
    if dp:
        DataParallel(model, shard_optimizer=shard, shard_world=8 if shard else None)
        model.dp.sync_stats_always = True
        comm = model.dp.comm
        if rate:
            comm.set_model(8, rate, lat)
    tr = Trainer(model, lr=1e-4, num_update=4000)
    for _ in range(2):
        tr.step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(x, y)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    extra = ""
    if comm is not None:
        comm.set_timing(True)
        tr.step(x, y)
        tr.step(x, y)
        torch.cuda.synchronize()
        t = comm.step_times()
        extra = f"exposed {t['exposed_ms']:.2f} ms, modelled {sum(b['ms'] for b in t['buckets']):.2f} ms, gather-exposed {t.get('gather_exposed_ms')}"
        comm.set_timing(False)
    tr.close()
    del tr, model
    torch.cuda.empty_cache()
    return ms, extra


for mode in (1, 2):
    _lib.call("rpb_line_claim_set", mode)
    for name, dp, rate, shard in (("no DP wrapper", False, 0, False), ("DP, no model", True, 0, False), (f"DP + ring model {gbps:.0f} GB/s", True, gbps, False),
                                  (f"DP + ring model {gbps / 2:.0f} GB/s", True, gbps / 2, False), (f"sharded Adam + ring model {gbps:.0f} GB/s", True, gbps, True)):
        ms, extra = run(dp, rate, shard)
        print(f"claim mode {mode}  {name:38s} {ms:7.2f} ms/step  {extra}", flush=True)
dist.destroy_process_group()
