#!/usr/bin/env python
"""Micro-benchmarks of individual C-ABI kernels at the BASELINE configs[1] sizes (B=32) with HIP events.

  python tools/kbench.py [cell_mix] [wgrad] [axis] [bn] [proj] [lift] [mode]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realpdebench_amd import ops  # noqa: E402
from realpdebench_amd.dft import SpectralPlan  # noqa: E402

B, T, H, W, Cin, C = int(os.environ.get("KB_B", 32)), 20, 128, 128, 2, 64
d = ops.Dims(B, T, H, W, Cin, C, 6)
plan = SpectralPlan(d.Tp, d.Hp, d.Wp, (4, 12, 16), device="cuda")
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
    print(f"{name:40s} {ms:8.3f} ms  {nbytes / ms / 1e6:8.1f} GB/s  {flops / ms / 1e9:7.2f} TF/s", flush=True)


which = set(sys.argv[1:]) or {"cell_mix", "wgrad", "axis", "bn", "proj", "lift", "mode", "bwd_row"}
x = torch.randn(d.ncell, C, **f)
y = torch.empty(d.ncell, C, **f)

if "cell_mix" in which:
    K2 = 2 * plan.KW
    z2 = torch.randn(B * d.Tp * d.Hp * K2 * C, **f)
    Wc, bias = torch.randn(C, C, **f), torch.randn(C, **f)
    rows = ops.cell_mix_stat_rows(d.ncell, C, C, K2, d.Wp, True)
    part = torch.empty(rows * 2 * C, **f)
    nb = 4 * (2 * d.ncell * C + d.ncell // d.Wp * K2 * C)
    fl = 2 * d.ncell * C * (K2 + C)
    timeit("cell_mix fwd (spec+stats)",
           lambda: ops.cell_mix(x, Wc, bias, z2, plan.GWt, y, part, d.ncell, C, C, K2, d.Wp), nb, fl)
    timeit("cell_mix bwd (spec)",
           lambda: ops.cell_mix(x, Wc, None, z2, plan.FW, y, None, d.ncell, C, C, K2, d.Wp, transpose_w=True), nb, fl)
    mean, invstd, gamma, beta = torch.zeros(C, **f), torch.ones(C, **f), torch.ones(C, **f), torch.zeros(C, **f)
    xfg = (mean, invstd, gamma, beta, True)
    timeit("cell_mix fwd (lazy gelu in, spec+stats)",
           lambda: ops.cell_mix(x, Wc, bias, z2, plan.GWt, y, part, d.ncell, C, C, K2, d.Wp, xf=xfg), nb, fl)
    rows2 = ops.cell_mix_stat_rows(d.ncell, C, C, K2, d.Wp, True, True)
    part2 = torch.empty(rows2 * 2 * C, **f)
    s_prev = torch.randn(d.ncell, C, **f)
    timeit("cell_mix bwd (spec + BN-bwd sums, gelu)",
           lambda: ops.cell_mix(x, Wc, None, z2, plan.FW, y, part2, d.ncell, C, C, K2, d.Wp, transpose_w=True,
                                bnb=(s_prev,) + xfg), nb + 4 * d.ncell * C, fl)
    if ops.cell_mix_wgrad_supported(d.ncell, K2, d.Wp):
        cs = ops.cell_mix_wgrad_slots(d.ncell, d.Wp)
        sp3, wp3 = torch.empty(cs * 2 * C, **f), torch.empty(cs * C * C, **f)
        timeit("cell_mix bwd + BN-bwd sums + conv wgrad (gelu, gz)  [round 4]",
               lambda: ops.cell_mix_wgrad(x, Wc, z2, plan.FW, y, sp3, wp3, d.ncell, K2, d.Wp, (s_prev,) + xfg, write_gz=True),
               nb + 4 * d.ncell * C, fl + 2 * d.ncell * C * C)
    timeit("cell_mix eval (out = gelu(bn(.)))",
           lambda: ops.cell_mix(x, Wc, bias, z2, plan.GWt, y, None, d.ncell, C, C, K2, d.Wp, oxf=xfg), nb, fl)
    phic = torch.randn(d.ncell, 8, **f)
    wcomp = torch.randn(C, 8, **f)
    timeit("cell_mix layer 0 (feature fields, spec+stats)",
           lambda: ops.cell_mix_feat(phic, wcomp, bias, z2, plan.GWt, y, part, d.ncell, 8, K2, d.Wp),
           4 * (d.ncell * (8 + C) + d.ncell // d.Wp * K2 * C), 2 * d.ncell * C * (K2 + 8))
    gu = torch.randn(d.ncrop, 128, **f)
    w1 = torch.randn(128, C, **f)
    timeit("cell_mix gather (fc1 dgrad)",
           lambda: ops.cell_mix(gu, w1, None, None, None, y, None, d.ncell, 128, C, 0, 1, transpose_w=True,
                                gather=True, crop6=d.crop6),
           4 * (d.ncrop * 128 + d.ncell * C), 2 * d.ncrop * 128 * C)

if "bwd_row" in which:
    K2 = 2 * plan.KW
    G = B * d.Tp * d.Hp
    mean, invstd, gamma, beta = torch.zeros(C, **f), torch.ones(C, **f), torch.ones(C, **f), torch.zeros(C, **f)
    sums = torch.zeros(2 * C, **f)
    gy = torch.randn(d.ncell, C, **f)
    Y1 = torch.empty(G * K2 * C, **f)
    part = torch.empty(ops.bn_bwd_row_slots(G) * (C * C + C), **f)
    xf = (mean, invstd, gamma, beta, True)
    timeit("bn_bwd_row (gelu, lazy x)",
           lambda: ops.bn_bwd_row(x, gy, y, gy, mean, invstd, gamma, beta, sums, d.ncell, True, xf, plan.GW, Y1, part, G,
                                  d.Wp, C, K2), 4 * (4 * d.ncell * C + G * K2 * C), 2 * d.ncell * C * (C + K2))

    timeit("bn_bwd_row (gz in, lazy x)  [the step's variant]",
           lambda: ops.bn_bwd_row(x, gy, y, gy, mean, invstd, gamma, beta, sums, d.ncell, False, xf, plan.GW, Y1, part, G,
                                  d.Wp, C, K2), 4 * (4 * d.ncell * C + G * K2 * C), 2 * d.ncell * C * (C + K2))
    timeit("bn_bwd_row (gz in, plain x)",
           lambda: ops.bn_bwd_row(x, gy, y, gy, mean, invstd, gamma, beta, sums, d.ncell, False, None, plan.GW, Y1, part, G,
                                  d.Wp, C, K2), 4 * (4 * d.ncell * C + G * K2 * C), 2 * d.ncell * C * (C + K2))
    timeit("bn_bwd_row (gz in, NO weight gradient: x not read)  [round 4]",
           lambda: ops.bn_bwd_row(x, gy, None, gy, mean, invstd, gamma, beta, sums, d.ncell, False, None, plan.GW, Y1, part, G,
                                  d.Wp, C, K2), 4 * (3 * d.ncell * C + G * K2 * C), 2 * d.ncell * C * K2)
    phic = torch.randn(d.ncell, 8, **f)
    timeit("bn_bwd_row layer 0 (feature fields)",
           lambda: ops.bn_bwd_row_feat(x, gy, phic, gy, mean, invstd, gamma, beta, sums, d.ncell, False, plan.GW, Y1, part, G,
                                       d.Wp, C, K2, 8), 4 * (3 * d.ncell * C + d.ncell * 8 + G * K2 * C), 2 * d.ncell * C * (8 + K2))

if "wgrad" in which:
    slots = ops.cell_wgrad_slots(d.ncell, C, C)
    part = torch.empty(slots * (C * C + C), **f)
    timeit("cell_wgrad 64x64", lambda: ops.cell_wgrad(x, y, part, d.ncell, C, C), 8 * d.ncell * C, 2 * d.ncell * C * C)
    gu = torch.randn(d.ncrop, 128, **f)
    slots = ops.cell_wgrad_slots(d.ncrop, 128, C)
    part = torch.empty(slots * (128 * C + 128), **f)
    timeit("cell_wgrad 128x64 crop",
           lambda: ops.cell_wgrad(gu, x, part, d.ncrop, 128, C, crop=True, crop6=d.crop6),
           4 * d.ncrop * (128 + C), 2 * d.ncrop * 128 * C)

if "axis" in which:
    m3, KH, KT = plan.KW, plan.KH, plan.KT
    N2, N3 = m3 * C, KH * m3 * C
    Y1 = torch.randn(B * d.Tp * d.Hp * 2 * m3 * C, **f)
    Y2 = torch.randn(B * d.Tp * 2 * KH * N2, **f)
    Xh = torch.randn(B * 2 * KT * N3, **f)
    G = B * d.Tp * d.Hp
    timeit("axis W fwd  K134xO32",
           lambda: ops.axis_gemm(x, Y1, plan.FWt, G, d.Wp, 2 * m3, C, d.Wp * C, C, 2 * m3 * C, C),
           4 * G * C * (d.Wp + 2 * m3), 2 * G * C * d.Wp * 2 * m3)
    _z, _o = torch.zeros(C, **f), torch.ones(C, **f)
    timeit("axis W fwd  K134xO32, lazy BN+GELU on load  [the step's forward variant]",
           lambda: ops.axis_gemm(x, Y1, plan.FWt, G, d.Wp, 2 * m3, C, d.Wp * C, C, 2 * m3 * C, C, xf=(_z, _o, _o, _z, True)),
           4 * G * C * (d.Wp + 2 * m3), 2 * G * C * d.Wp * 2 * m3)
    timeit("axis W fwd layer0 (k_valid=128)",
           lambda: ops.axis_gemm(x, Y1, plan.FWt, G, d.Wp, 2 * m3, C, d.Wp * C, C, 2 * m3 * C, C, k_valid=W),
           4 * G * C * (W + 2 * m3), 2 * G * C * W * 2 * m3)
    timeit("axis H fwd  K268xO48",
           lambda: ops.axis_gemm(Y1, Y2, plan.FHt, B * d.Tp, 2 * d.Hp, 2 * KH, N2, 2 * d.Hp * N2, N2, 2 * KH * N2, N2),
           4 * B * d.Tp * N2 * (2 * d.Hp + 2 * KH), 2 * B * d.Tp * N2 * 2 * d.Hp * 2 * KH)
    timeit("axis T fwd  K52xO16",
           lambda: ops.axis_gemm(Y2, Xh, plan.FTt, B, 2 * d.Tp, 2 * KT, N3, 2 * d.Tp * N3, N3, 2 * KT * N3, N3),
           4 * B * N3 * (2 * d.Tp + 2 * KT), 2 * B * N3 * 2 * d.Tp * 2 * KT)
    timeit("axis T inv  K16xO52",
           lambda: ops.axis_gemm(Xh, Y2, plan.GTt, B, 2 * KT, 2 * d.Tp, N3, 2 * KT * N3, N3, 2 * d.Tp * N3, N3),
           4 * B * N3 * (2 * d.Tp + 2 * KT), 2 * B * N3 * 2 * d.Tp * 2 * KT)
    timeit("axis H inv  K48xO268",
           lambda: ops.axis_gemm(Y2, Y1, plan.GHt, B * d.Tp, 2 * KH, 2 * d.Hp, N2, 2 * KH * N2, N2, 2 * d.Hp * N2, N2),
           4 * B * d.Tp * N2 * (2 * d.Hp + 2 * KH), 2 * B * d.Tp * N2 * 2 * d.Hp * 2 * KH)

if "bn" in which:
    mean, invstd, gamma, beta = torch.zeros(C, **f), torch.ones(C, **f), torch.ones(C, **f), torch.zeros(C, **f)
    timeit("bn_act_fwd gelu", lambda: ops.bn_act_fwd(x, mean, invstd, gamma, beta, y, d.ncell, C, True), 8 * d.ncell * C, 0)
    timeit("bn_act_fwd id", lambda: ops.bn_act_fwd(x, mean, invstd, gamma, beta, y, d.ncell, C, False), 8 * d.ncell * C, 0)
    part = torch.empty(ops.bn_bwd_rows() * 2 * C, **f)
    sums = torch.zeros(2 * C, **f)
    timeit("bn_bwd_reduce gelu",
           lambda: ops.bn_bwd_reduce(x, y, mean, invstd, gamma, beta, part, d.ncell, C, True), 8 * d.ncell * C, 0)
    timeit("bn_bwd_apply gelu",
           lambda: ops.bn_bwd_apply(x, y, mean, invstd, gamma, beta, sums, d.ncell, y, d.ncell, C, True),
           12 * d.ncell * C, 0)

if "proj" in which:
    w1, b1, w2, b2 = torch.randn(128, C, **f), torch.randn(128, **f), torch.randn(2, 128, **f), torch.randn(2, **f)
    out = torch.empty(d.ncrop, 2, **f)
    timeit("proj_fwd", lambda: ops.proj_fwd(x, w1, b1, w2, b2, out, d, 2), 4 * d.ncrop * (C + 2),
           2 * d.ncrop * 128 * (C + 2))
    gu = torch.empty(d.ncrop, 128, **f)
    part = torch.empty(ops.proj_slots(d.ncrop, C, 2) * (2 * 128 + 128 + 2), **f)
    timeit("proj_bwd", lambda: ops.proj_bwd(x, w1, b1, w2, b2, out, gu, part, d, 2), 4 * d.ncrop * (C + 2 + 128),
           2 * d.ncrop * 128 * (C + 4))

if "proj" in which and ops.proj_bwd_fused_supported(C, 2, W, d.Wp):
    mean, invstd, gamma, beta = torch.zeros(C, **f), torch.ones(C, **f), torch.ones(C, **f), torch.zeros(C, **f)
    xfl = (mean, invstd, gamma, beta, False)
    gout = torch.randn(d.ncrop, 2, **f)
    g = torch.empty(d.ncell, C, **f)
    sp = torch.empty(ops.proj_dgrad_slots(d) * 2 * C, **f)
    timeit("proj_dgrad (bf16 pipe, no gu)", lambda: ops.proj_dgrad(x, w1, b1, w2, gout, g, sp, d, 2, xfl),
           4 * (d.ncrop * (C + 2) + d.ncell * C), 2 * d.ncrop * 128 * 2 * C)
    gu2 = torch.randn(d.ncrop, 128, **f)
    timeit("proj_dgrad (bf16 pipe, reads gu)", lambda: ops.proj_dgrad(x, w1, b1, w2, None, g, sp, d, 2, xfl, gu=gu2),
           4 * (d.ncrop * (C + 128) + d.ncell * C), 2 * d.ncrop * 128 * C)
    wp = torch.empty(ops.proj_wgrad_slots(d) * ops.proj_wgrad_row(2), **f)
    timeit("proj_wgrad (bf16 pipe, no gu)", lambda: ops.proj_wgrad(x, w1, b1, w2, gout, wp, d, 2, xfl),
           4 * d.ncrop * (C + 2), 2 * d.ncrop * 128 * 2 * C)

if "proj" in which and ops.head_bwd_supported(C, 2, W, d.Wp, False, 0):
    mean, invstd, gamma, beta = torch.zeros(C, **f), torch.ones(C, **f), torch.ones(C, **f), torch.zeros(C, **f)
    xfl = (mean, invstd, gamma, beta, False)
    gout = torch.randn(d.ncrop, 2, **f)
    g = torch.empty(d.ncell, C, **f)
    hp = torch.empty(ops.head_bwd_slots(d) * ops.head_bwd_row(2), **f)
    timeit("head_bwd (one pass, gh never in HBM)", lambda: ops.head_bwd(x, w1, b1, w2, gout, g, hp, d, 2, xfl),
           4 * (d.ncrop * (C + 2) + d.ncell * C), 2 * d.ncrop * 128 * 3 * C)

    lp = torch.empty(ops.head_bwd_slots(d), **f)
    timeit("head_fwd_bwd (forward + loss + backward, one pass)",
           lambda: ops.head_fwd_bwd(x, w1, b1, w2, b2, gout, 1e-7, g, hp, lp, d, 2, xfl),
           4 * (d.ncrop * (C + 2) + d.ncell * C), 2 * d.ncrop * 128 * (3 * C + 2))

if "lift" in which:
    xin = torch.randn(B, T, H, W, Cin, **f)
    grids = [torch.linspace(0, 1, n, **f) for n in (T, H, W)]
    w0, b0 = torch.randn(C, Cin + 3, **f), torch.randn(C, **f)
    timeit("lift_pad_fwd", lambda: ops.lift_pad_fwd(xin, grids, w0, b0, y, d), 4 * (d.ncrop * Cin + d.ncell * C), 0)
    part = torch.empty(ops._lib.query("rpb_lift_bwd_rows") * (C * (Cin + 3) + C), **f)
    timeit("lift_bwd", lambda: ops.lift_bwd(x, xin, grids, part, d), 4 * d.ncrop * (Cin + C), 0)
    phic = torch.empty(d.ncell, 8, **f)
    timeit("lift_feat (feature fields, FW = 8)", lambda: ops.lift_feat(xin, grids, phic, d, 8), 4 * (d.ncrop * Cin + d.ncell * 8), 0)

if "mode" in which:
    M = plan.M
    X, Y = torch.randn(B * 2 * M * C, **f), torch.empty(B * 2 * M * C, **f)
    Wt = torch.randn(M * C * C * 2, **f)
    nb, fl = 8 * M * C * (C + 2 * B), 8 * B * M * C * C
    timeit("mode_contract_fwd", lambda: ops.mode_contract_fwd(X, Wt, Y, B, M, C), nb, fl)
    timeit("mode_contract_dgrad", lambda: ops.mode_contract_dgrad(X, Wt, Y, B, M, C), nb, fl)
    timeit("mode_contract_wgrad", lambda: ops.mode_contract_wgrad(X, Y, Wt, B, M, C), nb, fl)
