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

print("# weight gradients dW = G^T A (ops.gemm_tn): split-bf16 kernel, then RPB_GEMM_TN_F32-equivalent fp32 kernel through the C ABI")
from realpdebench_amd import _lib  # noqa: E402
for M2, N, K in ((M, 1024, 256), (M, 256, 1024), (M, 256, 512), (M, 256, 256), (4 * M, 768, 256), (4 * M, 256, 256)):
    G, A = torch.randn(M2, N, **f), torch.randn(M2, K, **f)
    sp = ops.gemm_tn_splits(M2, N, K)
    part = torch.empty(sp, N * K + N, **f)
    timeit(f"gemm_tn M={M2} N={N} K={K} split-bf16 sp={sp}", lambda: ops.gemm_tn(G, A, part, M2, N, K), 2 * M2 * N * K)
    sp32 = _lib.query("rpb_gemm_tn_splits", M2, N, K, 0)
    part32 = torch.empty(sp32, N * K + N, **f)
    timeit(f"gemm_tn M={M2} N={N} K={K} fp32 MFMA sp={sp32}",
           lambda: _lib.call("rpb_gemm_tn", G.data_ptr(), A.data_ptr(), part32.data_ptr(), M2, N, K, N, K, 0, 0, 0, 0,
                             torch.cuda.current_stream().cuda_stream), 2 * M2 * N * K)
    d = (part.double().sum(0) - part32.double().sum(0)).norm() / part32.double().sum(0).norm()
    print(f"    rel. difference of the two results {float(d):.2e}")
    del G, A, part, part32
