#!/usr/bin/env python
"""Forward token GEMM at DPOT's small token counts: split-bf16 (64-row tiles) against the fp32 MFMA kernel, incl. the weight preparation."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realpdebench_amd import ops  # noqa: E402

f = dict(device="cuda", dtype=torch.float32)


def timeit(name, fn, flops, iters=20):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print(f"{name:56s} {ms:8.4f} ms  {flops / ms / 1e9:7.2f} TF/s", flush=True)


for M, N, K in ((4096, 1024, 1024), (4096, 2048, 1024), (4096, 1024, 2048), (4096, 1280, 1024), (4096, 1024, 1280), (8192, 1024, 1024),
                (16384, 256, 256), (32768, 512, 512)):
    A, W, out = torch.randn(M, K, **f), torch.randn(N, K, **f) * 0.03, torch.empty(M, N, **f)
    b, res = torch.randn(N, **f), torch.randn(M, N, **f)
    for rows in (0, 1 << 40):
        ops.GEMM_SPLIT_MIN_ROWS = rows
        tag = "split-bf16" if rows == 0 else "fp32 MFMA "
        timeit(f"M={M} N={N} K={K} {tag} bias+gelu", lambda: ops.gemm_nt(A, W, out, M, N, K, bias=b, act=1), 2 * M * N * K)
        timeit(f"M={M} N={N} K={K} {tag} bias+residual", lambda: ops.gemm_nt(A, W, out, M, N, K, bias=b, residual=res), 2 * M * N * K)
