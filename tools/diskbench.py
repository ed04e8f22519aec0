#!/usr/bin/env python
"""Throughput of the on-disk reader (disk.DiskBatchLoader): a synthetic cylinder-shaped dataset in the reference's V2 Arrow layout
(numerical data at 128 x 256, sub-sampled by 2 to the 64 x 128 mesh) is written to a scratch directory and read back as
device batches; prints samples/s and the host bytes moved per second.  DB_SIMS / DB_FRAMES / DB_BATCH / DB_STEPS."""
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realpdebench_amd import disk  # noqa: E402

SIMS, FRAMES = int(os.environ.get("DB_SIMS", 4)), int(os.environ.get("DB_FRAMES", 200))
B, STEPS = int(os.environ.get("DB_BATCH", 32)), int(os.environ.get("DB_STEPS", 20))
H, W = 128, 256
root = tempfile.mkdtemp(prefix="rpb_disk_")
try:
    from datasets import Dataset
    rng = np.random.default_rng(0)
    sims = [f"{100 * (i + 1)}.h5" for i in range(SIMS)]
    base = os.path.join(root, "cylinder", "hf_dataset")
    rows = {k: [] for k in ("sim_id", "u", "v", "p", "shape_t", "shape_h", "shape_w")}
    for s in sims:
        rows["sim_id"].append(s)
        for k in ("u", "v", "p"):
            rows[k].append(rng.standard_normal((FRAMES, H, W), dtype=np.float32).tobytes())
        rows["shape_t"].append(FRAMES)
        rows["shape_h"].append(H)
        rows["shape_w"].append(W)
    Dataset.from_dict(rows).save_to_disk(os.path.join(base, "numerical"))
    idx = [{"sim_id": s, "time_id": t} for s in sims for t in range(0, FRAMES - 40, 4)]
    with open(os.path.join(base, "train_index_numerical.json"), "w") as fh:
        json.dump(idx, fh)
    w = disk.FluidWindows("cylinder", root, "numerical", "train", mask_prob=0.5)
    print(f"{len(w)} samples, window {w.horizon} x {H} x {W} x 3 at full resolution = "
          f"{w.horizon * H * W * 3 * 4 / 2**20:.1f} MiB per sample on the host side")
    stats = (torch.zeros(3), torch.zeros(3), torch.ones(3), torch.ones(3))
    for threads in (1, 4, 8, 16):
        depth = 3
        loader = disk.DiskBatchLoader(w, B, "cuda", stats=stats, shuffle=True, seed=0, depth=depth, copy_threads=threads)
        for _ in range(3):
            next(loader)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(STEPS):
            x, y = next(loader)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        loader.close()
        per = w.horizon * (H // w.sub_s) * W * 3 * 4                    # every sub_s-th row of u, v, p (p only when not masked)
        print(f"copy threads {threads:2d}: {STEPS * B / dt:8.1f} samples/s   {STEPS * B * per / dt / 1e9:6.2f} GB/s (upper bound: p is masked half the time) host rows -> pinned   "
              f"batch {tuple(x.shape)} / {tuple(y.shape)}")
    if os.environ.get("DB_TRAIN", "1") != "0":
        # the fused FNO train step fed (a) by one resident batch, (b) by the reader: what the disk path costs end to end
        from realpdebench_amd.model.fno import FNO3d
        from realpdebench_amd.trainer import Trainer
        shape = (w.in_step, H // w.sub_s, W // w.sub_s, 3)
        model = FNO3d(4, 12, 16, 4, 64, shape, shape).cuda()
        tr = Trainer(model, lr=1e-4, num_update=1000)
        loader = disk.DiskBatchLoader(w, B, "cuda", stats=stats, shuffle=True, seed=0, depth=3, copy_threads=4)
        x, y = next(loader)
        def run(feed, n):
            for _ in range(3):
                tr.step(*feed())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                tr.step(*feed())
            torch.cuda.synchronize()
            return n * B / (time.perf_counter() - t0)
        a = run(lambda: (x, y), STEPS)
        b = run(lambda: next(loader), STEPS)
        loader.close()
        print(f"FNO3d train step, batch {B} x {shape}: {a:8.1f} samples/s from a resident batch, {b:8.1f} samples/s from the on-disk reader "
              f"({100 * b / a:.1f} %)")
finally:
    shutil.rmtree(root, ignore_errors=True)
