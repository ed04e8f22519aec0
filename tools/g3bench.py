#!/usr/bin/env python
"""gemm3x at the Transolver / Galerkin MLP shapes with their training epilogues:  [RPB_G3_STAGGER=n] python tools/g3bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realpdebench_amd import ops  # noqa: E402

M = int(os.environ.get("GB_M", 655360))
f = dict(device="cuda", dtype=torch.float32)


def timeit(name, fn, flops, iters=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print(f"{name:48s} {ms:8.3f} ms  {flops / ms / 1e9:7.2f} TF/s", flush=True)


for N, K in ((1024, 256), (256, 1024), (512, 256), (256, 256), (256, 512)):
    A, W, out = torch.randn(M, K, **f), torch.randn(N, K, **f) * 0.05, torch.empty(M, N, **f)
    b = torch.randn(N, **f)
    pre, aux, res = torch.empty(M, N, **f), torch.randn(M, N, **f), torch.randn(M, N, **f)
    fl = 2 * M * N * K
    timeit(f"N={N} K={K} plain", lambda: ops.gemm_nt(A, W, out, M, N, K), fl)
    timeit(f"N={N} K={K} bias+gelu+pre_out", lambda: ops.gemm_nt(A, W, out, M, N, K, bias=b, act=1, pre_out=pre), fl)
    timeit(f"N={N} K={K} *gelu'(aux)", lambda: ops.gemm_nt(A, W, out, M, N, K, act=2, aux=aux), fl)
    timeit(f"N={N} K={K} bias+residual", lambda: ops.gemm_nt(A, W, out, M, N, K, bias=b, residual=res), fl)
    del A, W, out, pre, aux, res
