#!/usr/bin/env python
"""Token-GEMM micro-benchmark: rpb_gemm_nt / rpb_gemm_tn at skinny shapes (M tokens, small K or N)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realpdebench_amd import ops  # noqa: E402

M = int(os.environ.get("GB_M", 1966080))
f = dict(device="cuda", dtype=torch.float32)


def timeit(name, fn, nbytes, flops, iters=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print(f"{name:34s} {ms:8.3f} ms  {nbytes / ms / 1e6:8.1f} GB/s  {flops / ms / 1e9:7.2f} TF/s", flush=True)


for N, K in ((384, 64), (128, 64), (128, 256), (64, 128), (64, 384), (32, 64), (256, 256), (768, 256)):
    A, W, out = torch.randn(M, K, **f), torch.randn(N, K, **f), torch.empty(M, N, **f)
    timeit(f"gemm_nt M={M} N={N} K={K}", lambda: ops.gemm_nt(A, W, out, M, N, K), 4 * M * (N + K), 2 * M * N * K)
    del A, W, out

for N, K in ((768, 256), (256, 256), (384, 64), (64, 128), (1024, 256), (256, 1024)):
    G, A = torch.randn(M, N, **f), torch.randn(M, K, **f)
    splits = ops.gemm_tn_splits(M, N, K)
    part = torch.empty(splits, N * K + N, **f)
    timeit(f"gemm_tn M={M} N={N} K={K} sp={splits}", lambda: ops.gemm_tn(G, A, part, M, N, K), 4 * M * (N + K), 2 * M * N * K)
    del G, A, part
