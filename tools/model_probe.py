#!/usr/bin/env python
"""One secondary-model measurement of bench.py with its per-kernel table:  python tools/model_probe.py dpot|galerkin|transolver|unet"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RPB_BENCH_TABLE"] = "1"
import bench  # noqa: E402

fn = {"dpot": bench.bench_dpot, "galerkin": bench.bench_galerkin, "transolver": bench.bench_transolver, "unet": bench.bench_unet}[sys.argv[1]]
print(json.dumps(fn(torch.device("cuda:0")), indent=1))
