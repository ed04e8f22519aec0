#!/usr/bin/env python
"""Split-bf16 3x3x3 convolution (rpb_conv3x) vs the exact-fp32 implicit GEMM (rpb_gemm_nt conv mode 1): accuracy against
fp64 on a small mesh, and time at the Transolver (Ci 256 -> 512) and U-Net (64 -> 64, 256 -> 256) cylinder shapes."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realpdebench_amd import ops  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def run(B, mesh, Ci, N, check, iters=3):
    T, H, W = mesh
    M = B * T * H * W
    torch.manual_seed(0)
    x = torch.randn(M, Ci, device="cuda")
    w = torch.randn(N, 27 * Ci, device="cuda") / (27 * Ci) ** 0.5
    bias = torch.randn(N, device="cuda")
    planes = torch.empty(3 * M * Ci, dtype=torch.int16, device="cuda")
    wz = torch.empty(3 * N * 27 * Ci, dtype=torch.int16, device="cuda")
    y = torch.empty(M, N, device="cuda")
    y32 = torch.empty(M, N, device="cuda")
    ops.split3(x, planes, M, Ci)
    ops.conv3x_wprep(w, wz, N, Ci)
    ops.conv3x(planes, wz, y, M, N, Ci, mesh, bias=bias)
    ops.gemm_nt(x, w, y32, M, N, 27 * Ci, bias=bias, conv=mesh)
    torch.cuda.synchronize()
    print(f"B={B} mesh={mesh} Ci={Ci} N={N}: conv3x vs fp32 path rel {rel(y, y32):.2e}", flush=True)
    if check:
        xr = x.view(B, T, H, W, Ci).permute(0, 4, 1, 2, 3).double().cpu()
        wr = w.view(N, 3, 3, 3, Ci).permute(0, 4, 1, 2, 3).double().cpu()
        ref = F.conv3d(xr, wr, bias.double().cpu(), padding=1).permute(0, 2, 3, 4, 1).reshape(M, N)
        print(f"   vs fp64: conv3x {rel(y.cpu(), ref):.2e}   exact-fp32 path {rel(y32.cpu(), ref):.2e}", flush=True)
    for name, fn in (("split3", lambda: ops.split3(x, planes, M, Ci)),
                     ("conv3x", lambda: ops.conv3x(planes, wz, y, M, N, Ci, mesh, bias=bias)),
                     ("gemm_nt fp32", lambda: ops.gemm_nt(x, w, y32, M, N, 27 * Ci, bias=bias, conv=mesh))):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / iters
        print(f"   {name:14s} {ms:8.3f} ms   {2 * M * N * 27 * Ci / ms / 1e9:7.1f} TF/s (fp32-equivalent)", flush=True)


def run_wgrad(B, mesh, Ci, Co, check, iters=3):
    T, H, W = mesh
    M = B * T * H * W
    torch.manual_seed(1)
    x = torch.randn(M, Ci, device="cuda")
    g = torch.randn(M, Co, device="cuda")
    K = 27 * Ci

    def finish(part):
        dW = torch.empty(Co, K, device="cuda")
        db = torch.empty(Co, device="cuda")
        ops.reduce_partials(part, part.shape[0], Co * K, out_f32=dW.view(-1), row_stride=Co * K + Co)
        ops.reduce_partials(part, part.shape[0], Co, out_f32=db, row_stride=Co * K + Co, col0=Co * K)
        return dW, db

    ops.CONV3_SPLIT = True
    part, rev = ops.conv3_wgrad_parts(g, x, M, Co, Ci, mesh)
    dW, db = finish(part)
    if rev:
        dW = ops.conv3_taps_restore(dW, Co, Ci)
    ops.CONV3_SPLIT = False
    dW32, db32 = finish(ops.conv3_wgrad_parts(g, x, M, Co, Ci, mesh)[0])
    torch.cuda.synchronize()
    print(f"wgrad B={B} mesh={mesh} Ci={Ci} Co={Co}: split vs fp32 path dW {rel(dW, dW32):.2e} db {rel(db, db32):.2e}", flush=True)
    if check:
        xr = x.view(B, T, H, W, Ci).permute(0, 4, 1, 2, 3).double().cpu()
        gr = g.view(B, T, H, W, Co).permute(0, 4, 1, 2, 3).double().cpu()
        wr = torch.zeros(Co, Ci, 3, 3, 3, dtype=torch.float64, requires_grad=True)
        br = torch.zeros(Co, dtype=torch.float64, requires_grad=True)
        F.conv3d(xr, wr, br, padding=1).backward(gr)
        ref = wr.grad.permute(0, 2, 3, 4, 1).reshape(Co, K)
        print(f"   vs fp64: split dW {rel(dW.cpu(), ref):.2e} db {rel(db.cpu(), br.grad):.2e}   exact-fp32 path dW {rel(dW32.cpu(), ref):.2e}", flush=True)
    for name, flag in (("wgrad split", True), ("wgrad fp32", False)):
        ops.CONV3_SPLIT = flag
        ops.conv3_wgrad_parts(g, x, M, Co, Ci, mesh)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            ops.conv3_wgrad_parts(g, x, M, Co, Ci, mesh)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / iters
        print(f"   {name:14s} {ms:8.3f} ms   {2 * M * Co * K / ms / 1e9:7.1f} TF/s (fp32-equivalent, incl. splits)", flush=True)
    ops.CONV3_SPLIT = True


def run_gemm(M, N, K, iters=3):
    torch.manual_seed(2)
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") / K ** 0.5
    bias = torch.randn(N, device="cuda")
    o1, o2 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    for name, flag, o in (("gemm3x", True, o1), ("gemm_nt fp32", False, o2)):
        ops.GEMM_SPLIT = flag
        ops.gemm_nt(A, W, o, M, N, K, bias=bias, act=1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            ops.gemm_nt(A, W, o, M, N, K, bias=bias, act=1)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / iters
        print(f"   {name:14s} M={M} N={N} K={K}: {ms:8.3f} ms   {2 * M * N * K / ms / 1e9:7.1f} TF/s-eq   "
              f"{4 * M * (N + K) / ms / 1e6:7.0f} GB/s", flush=True)
    ops.GEMM_SPLIT = True
    print(f"   gemm3x vs fp32 kernel rel {rel(o1, o2):.2e}", flush=True)


if os.environ.get("CX_GEMM", "0") == "1":
    ops.GEMM_SPLIT_MIN_K = ops.GEMM_SPLIT_MIN_N = 64
    for shp in ((2621440, 768, 256), (2621440, 256, 256), (2621440, 256, 768), (2621440, 128, 256), (655360, 1024, 256),
                (655360, 256, 1024), (655360, 512, 256)):
        run_gemm(*shp)
    sys.exit(0)
if os.environ.get("CX_WGRAD", "1") == "1":
    run_wgrad(2, (3, 5, 16), 64, 64, True)
    run_wgrad(1, (4, 6, 40), 128, 64, True)
    run_wgrad(1, (2, 9, 24), 64, 128, True)
    run_wgrad(2, (16, 6, 5), 64, 64, True)
    run_wgrad(4, (128, 64, 20), 256, 512, False)
    run_wgrad(4, (20, 64, 128), 256, 512, False)
    run_wgrad(12, (20, 64, 128), 64, 64, False)
    run_wgrad(12, (20, 16, 32), 256, 256, False)
if os.environ.get("CX_FWD", "1") != "1":
    sys.exit(0)
run(2, (3, 5, 7), 64, 64, True)
run(1, (4, 6, 40), 128, 128, True)
run(1, (2, 9, 33), 64, 256, True)
run(4, (20, 64, 128), 256, 512, False)
run(12, (20, 64, 128), 64, 64, False)
run(12, (20, 16, 32), 256, 256, False)
run(12, (20, 32, 64), 128, 128, False)
