"""Per-kernel parity on a real MI355X: every C-ABI entry point against an fp64 PyTorch-CPU statement of the
same operator (tolerance: fp32 round-off, Rel-L2 < 2e-6 unless stated)."""
import math

import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
TOL = 2e-6


@pytest.fixture(scope="module")
def ops():
    from realpdebench_amd import ops as o
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return o


def dev(t):
    return t.float().cuda().contiguous()


@pytest.mark.parametrize("G,K,O,N,kv", [
    (5, 134, 32, 64, None), (3, 268, 48, 128, None), (2, 52, 16, 256, None), (2, 16, 52, 96, None),
    (3, 48, 268, 64, None), (7, 32, 134, 32, None), (4, 7, 5, 32, None), (3, 40, 12, 64, 29), (2, 134, 32, 64, 128),
])
def test_axis_gemm(ops, G, K, O, N, kv):
    torch.manual_seed(G * 1000 + K)
    M = torch.randn(O, K, dtype=torch.float64)
    x = torch.randn(G, K, N, dtype=torch.float64)
    xx = x.clone()
    if kv is not None:
        xx[:, kv:] = 0
    ref = torch.einsum("ok,gkn->gon", M, xx)
    out = torch.full((G, O, N), float("nan"), device="cuda")
    ops.axis_gemm(dev(x), out, dev(M.t()), G, K, O, N, K * N, N, O * N, N, k_valid=kv)
    assert rel_l2(out.cpu(), ref) < TOL
    # accumulate
    ops.axis_gemm(dev(x), out, dev(M.t()), G, K, O, N, K * N, N, O * N, N, k_valid=kv, accumulate=True)
    assert rel_l2(out.cpu(), 2 * ref) < TOL


@pytest.mark.parametrize("B,M,C", [(3, 10, 32), (32, 6, 64), (5, 7, 64), (70, 3, 64), (5, 4, 128), (32, 3, 128), (70, 2, 128)])
def test_mode_contract(ops, B, M, C):
    torch.manual_seed(B + M + C)
    X = torch.randn(B, 2, M, C, dtype=torch.float64)
    W = torch.randn(M, C, C, 2, dtype=torch.float64)
    G = torch.randn(B, 2, M, C, dtype=torch.float64)
    xc, gc = torch.complex(X[:, 0], X[:, 1]), torch.complex(G[:, 0], G[:, 1])
    wc = torch.view_as_complex(W)
    y = torch.einsum("bmi,mio->bmo", xc, wc)
    gx = torch.einsum("bmo,mio->bmi", gc, wc.conj())
    gw = torch.einsum("bmi,bmo->mio", xc.conj(), gc)
    planar = lambda z: torch.stack([z.real, z.imag], dim=1)
    Y = torch.empty(B, 2, M, C, device="cuda")
    ops.mode_contract_fwd(dev(X), dev(W), Y, B, M, C)
    assert rel_l2(Y.cpu(), planar(y)) < TOL
    GX = torch.empty(B, 2, M, C, device="cuda")
    ops.mode_contract_dgrad(dev(G), dev(W), GX, B, M, C)          # width 128, B=32 goes in two batch slices
    assert rel_l2(GX.cpu(), planar(gx)) < TOL
    GW = torch.empty(M, C, C, 2, device="cuda")
    ops.mode_contract_wgrad(dev(X), dev(G), GW, B, M, C)
    assert rel_l2(GW.cpu(), torch.view_as_real(gw)) < TOL


@pytest.mark.parametrize("C,Wp,rows,K2", [(32, 18, 11, 8), (64, 134, 5, 32), (128, 22, 7, 6), (64, 13, 9, 7)])
def test_cell_mix_spectral_conv_stats(ops, C, Wp, rows, K2):
    torch.manual_seed(C + Wp)
    ncell = rows * Wp
    x = torch.randn(ncell, C, dtype=torch.float64)
    Wc = torch.randn(C, C, dtype=torch.float64) / math.sqrt(C)
    bias = torch.randn(C, dtype=torch.float64)
    z2 = torch.randn(rows, K2, C, dtype=torch.float64)
    GW = torch.randn(Wp, K2, dtype=torch.float64)
    ref = torch.einsum("wk,gkc->gwc", GW, z2).reshape(ncell, C) + x @ Wc.t() + bias
    out = torch.empty(ncell, C, device="cuda")
    nrows = ops.cell_mix_stat_rows(ncell, C, C, K2, Wp, True)
    part = torch.zeros(nrows, 2, C, device="cuda")
    ops.cell_mix(dev(x), dev(Wc), dev(bias), dev(z2), dev(GW.t()), out, part, ncell, C, C, K2, Wp)
    assert rel_l2(out.cpu(), ref) < TOL
    s = part.double().sum(0).cpu()
    assert rel_l2(s[0], ref.sum(0)) < 1e-5
    assert rel_l2(s[1], (ref ** 2).sum(0)) < 1e-5
    # dgrad flavour: transposed weight, no bias, no stats
    ref2 = torch.einsum("wk,gkc->gwc", GW, z2).reshape(ncell, C) + x @ Wc
    ops.cell_mix(dev(x), dev(Wc), None, dev(z2), dev(GW.t()), out, None, ncell, C, C, K2, Wp, transpose_w=True)
    assert rel_l2(out.cpu(), ref2) < TOL


def test_cell_mix_gather(ops):
    torch.manual_seed(3)
    B, T, H, W, pad, KC, CO = 2, 3, 5, 7, 2, 128, 64
    Tp, Hp, Wp = T + pad, H + pad, W + pad
    gu = torch.randn(B, T, H, W, KC, dtype=torch.float64)
    Wm = torch.randn(KC, CO, dtype=torch.float64)
    ref = torch.zeros(B, Tp, Hp, Wp, CO, dtype=torch.float64)
    ref[:, :T, :H, :W] = gu @ Wm
    out = torch.full((B * Tp * Hp * Wp, CO), float("nan"), device="cuda")
    ops.cell_mix(dev(gu).view(-1, KC), dev(Wm), None, None, None, out, None, B * Tp * Hp * Wp, KC, CO, 0, 1,
                 transpose_w=True, gather=True, crop6=(T, H, W, Tp, Hp, Wp))
    assert rel_l2(out.cpu().view_as(ref), ref) < TOL


@pytest.mark.parametrize("CO,CI,ncell", [(64, 64, 1000), (32, 32, 77), (128, 64, 650), (128, 128, 300), (64, 32, 33)])
def test_cell_wgrad(ops, CO, CI, ncell):
    torch.manual_seed(CO + CI)
    gs = torch.randn(ncell, CO, dtype=torch.float64)
    x = torch.randn(ncell, CI, dtype=torch.float64)
    slots = ops.cell_wgrad_slots(ncell, CO, CI)
    part = torch.full((slots, CO * CI + CO), float("nan"), device="cuda")
    ops.cell_wgrad(dev(gs), dev(x), part, ncell, CO, CI)
    got = part.double().sum(0).cpu()
    assert rel_l2(got[:CO * CI].view(CO, CI), gs.t() @ x) < TOL
    assert rel_l2(got[CO * CI:], gs.sum(0)) < TOL


@pytest.mark.parametrize("ncell,gelu", [(70 * 77 + 7, True), (300, False), (33, True), (64 * 1024 + 1, None)])
def test_cell_wgrad_c128_bf16_pipe(ops, ncell, gelu):
    """(CO, CI) = (128, 128) without the crop runs on the bf16 matrix pipe (csrc/rpb_cwx.hip: quadrant waves, split operands): fp32-grade
    against fp64 with the lazy BatchNorm (+ GELU) of the layer input, a cell count that ends inside a tile, fewer tiles than streams."""
    torch.manual_seed(ncell)
    C = 128
    f8 = dict(dtype=torch.float64)
    gs = torch.randn(ncell, C, **f8) * (torch.arange(C, **f8) % 7 + 1)          # a transposed result would not match
    x = torch.randn(ncell, C, **f8) * 1.3 + 0.2
    mean, invstd = torch.randn(C, **f8) * 0.2, torch.rand(C, **f8) + 0.5
    gamma, beta = torch.rand(C, **f8) + 0.5, torch.randn(C, **f8) * 0.3
    a = x if gelu is None else _xf_ref(x, mean, invstd, gamma, beta, gelu)
    xf = None if gelu is None else (dev(mean), dev(invstd), dev(gamma), dev(beta), gelu)
    slots = ops.cell_wgrad_slots(ncell, C, C)
    part = torch.full((slots, C * C + C), float("nan"), device="cuda")
    ops.cell_wgrad(dev(gs), dev(x), part, ncell, C, C, xf=xf)
    got = part.double().sum(0).cpu()
    assert rel_l2(got[:C * C].view(C, C), gs.t() @ a) < 3e-6
    assert rel_l2(got[C * C:], gs.sum(0)) < 3e-6


def test_cell_wgrad_crop(ops):
    torch.manual_seed(5)
    B, T, H, W, pad, CO, CI = 2, 3, 4, 9, 3, 128, 64
    Tp, Hp, Wp = T + pad, H + pad, W + pad
    gs = torch.randn(B, T, H, W, CO, dtype=torch.float64)
    xp = torch.randn(B, Tp, Hp, Wp, CI, dtype=torch.float64)
    ref = torch.einsum("bthwo,bthwi->oi", gs, xp[:, :T, :H, :W])
    ncrop = B * T * H * W
    slots = ops.cell_wgrad_slots(ncrop, CO, CI)
    part = torch.zeros(slots, CO * CI + CO, device="cuda")
    ops.cell_wgrad(dev(gs).view(-1, CO), dev(xp).view(-1, CI), part, ncrop, CO, CI, crop=True, crop6=(T, H, W, Tp, Hp, Wp))
    assert rel_l2(part.double().sum(0).cpu()[:CO * CI].view(CO, CI), ref) < TOL


@pytest.mark.parametrize("W,gelu", [(32, True), (64, False), (40, True)])
def test_cell_wgrad_crop_c128(ops, W, gelu):
    """fc1 weight gradient at width 128 (configs/fsi/fno.yaml): gs over the crop, x = act(bn(.)) of the padded tensor.  W % 32 == 0 runs
    on the shared-plane bf16-pipe kernel (csrc/rpb_cwx.hip, tiles inside w-rows), anything else on the fp32 MFMA kernel."""
    torch.manual_seed(W)
    B, T, H, pad, C = 2, 3, 5, 6, 128
    Tp, Hp, Wp = T + pad, H + pad, W + pad
    gs = torch.randn(B, T, H, W, C, dtype=torch.float64)
    xp = torch.randn(B, Tp, Hp, Wp, C, dtype=torch.float64)
    mean, var = torch.randn(C, dtype=torch.float64) * 0.1, torch.rand(C, dtype=torch.float64) + 0.5
    gamma, beta = torch.rand(C, dtype=torch.float64) + 0.5, torch.randn(C, dtype=torch.float64) * 0.1
    invstd = (var + 1e-5).rsqrt()
    a = (xp - mean) * invstd * gamma + beta
    if gelu:
        a = torch.nn.functional.gelu(a)
    ref = torch.einsum("bthwo,bthwi->oi", gs, a[:, :T, :H, :W])
    ncrop = B * T * H * W
    slots = ops.cell_wgrad_slots(ncrop, C, C)
    part = torch.zeros(slots, C * C + C, device="cuda")
    xf = (dev(mean), dev(invstd), dev(gamma), dev(beta), gelu)
    ops.cell_wgrad(dev(gs).view(-1, C), dev(xp).view(-1, C), part, ncrop, C, C, crop=True, crop6=(T, H, W, Tp, Hp, Wp), xf=xf)
    got = part.double().sum(0).cpu()
    assert rel_l2(got[:C * C].view(C, C), ref) < 3e-6
    assert rel_l2(got[C * C:], gs.sum((0, 1, 2, 3))) < 3e-6


@pytest.mark.parametrize("C,DO,W", [(64, 2, 7), (64, 2, 40), (64, 1, 16), (64, 4, 21), (32, 3, 7), (128, 5, 7), (64, 16, 7), (64, 16, 70),
                                    (64, 12, 40), (64, 5, 21)])
def test_proj_fwd_bwd(ops, C, DO, W):
    """C = 64 with DO <= 16 (forward; fc2 on the matrix pipe too above 4 outputs) / <= 2 (backward) runs on the bf16 matrix pipe
    (csrc/rpb_pjx.hip: line-walking waves, partial last tile), everything else on the fp32 MFMA kernels."""
    torch.manual_seed(C + DO)
    B, T, H, pad = 2, 3, 6, 2
    d = ops.Dims(B, T, H, W, 2, C, pad)
    a = torch.randn(B, d.Tp, d.Hp, d.Wp, C, dtype=torch.float64)
    w1 = (torch.randn(128, C, dtype=torch.float64) / math.sqrt(C)).requires_grad_(True)
    b1 = torch.randn(128, dtype=torch.float64, requires_grad=True)
    w2 = (torch.randn(DO, 128, dtype=torch.float64) / 11).requires_grad_(True)
    b2 = torch.randn(DO, dtype=torch.float64, requires_grad=True)
    ac = a[:, :T, :H, :W].reshape(-1, C)
    u = ac @ w1.t() + b1
    u.retain_grad()
    v = torch.nn.functional.gelu(u)
    out_ref = v @ w2.t() + b2
    gout = torch.randn_like(out_ref)
    out_ref.backward(gout)
    out = torch.empty(d.ncrop, DO, device="cuda")
    A, W1, B1, W2, B2 = dev(a).view(-1, C), dev(w1.detach()), dev(b1.detach()), dev(w2.detach()), dev(b2.detach())
    ops.proj_fwd(A, W1, B1, W2, B2, out, d, DO)
    assert rel_l2(out.cpu(), out_ref.detach()) < TOL
    slots = ops.proj_slots(d.ncrop, C, DO)
    part = torch.zeros(slots, DO * 128 + 128 + DO, device="cuda")
    gu = torch.empty(d.ncrop, 128, device="cuda")
    ops.proj_bwd(A, W1, B1, W2, B2, dev(gout), gu, part, d, DO)
    assert rel_l2(gu.cpu(), u.grad) < 5e-6
    s = part.double().sum(0).cpu()
    assert rel_l2(s[:DO * 128].view(DO, 128), w2.grad) < 5e-6
    assert rel_l2(s[DO * 128:DO * 128 + 128], b1.grad) < 5e-6
    assert rel_l2(s[DO * 128 + 128:], b2.grad) < 5e-6


@pytest.mark.parametrize("C", [64, 128])
@pytest.mark.parametrize("B,T,H,W,pad,bn,DO", [(2, 3, 6, 40, 2, True, 2), (1, 2, 5, 70, 6, True, 2), (3, 2, 4, 32, 6, False, 2), (2, 2, 3, 5, 2, True, 2),
                                                (2, 3, 5, 40, 2, True, 3), (1, 2, 5, 70, 6, True, 4), (2, 2, 4, 33, 2, False, 1), (2, 2, 3, 5, 6, True, 3)])
def test_eval_head_up_to_four_outputs(ops, B, T, H, W, pad, bn, DO, C):
    """The evaluation forward of the head at <= 4 fc2 outputs (csrc/rpb_pjh.hip: 32-cell tiles, BatchNorm folded into the staged fc1
    planes, bias folded into GELU) against fp64, with and without the lazy BatchNorm of the last layer; partial last tiles, lines
    shorter than a tile, more lines than waves; RPB_HEAD_PJH=0 (the 16x16x32 kernel) must agree with it.  C = 128 (configs/fsi/fno.yaml):
    the same kernel's width-128 instance (one workgroup per CU, eight K-steps)."""
    torch.manual_seed(B * 100 + W)
    d = ops.Dims(B, T, H, W, 2, C, pad)
    a = torch.randn(B, d.Tp, d.Hp, d.Wp, C, dtype=torch.float64) * 1.5 + 0.3
    w1 = torch.randn(128, C, dtype=torch.float64) / math.sqrt(C)
    b1 = torch.randn(128, dtype=torch.float64)
    w2 = torch.randn(DO, 128, dtype=torch.float64) / 11
    b2 = torch.randn(DO, dtype=torch.float64)
    mean, var = torch.randn(C, dtype=torch.float64) * 0.2 + 0.3, torch.rand(C, dtype=torch.float64) + 0.5
    gamma, beta = torch.rand(C, dtype=torch.float64) + 0.5, torch.randn(C, dtype=torch.float64) * 0.1
    gamma[5] = 0.0                                       # a dead channel: the folded scale is an exact zero
    invstd = (var + 1e-5).rsqrt()
    ac = a[:, :T, :H, :W].reshape(-1, C)
    if bn:
        ac = (ac - mean) * invstd * gamma + beta
    ref = torch.nn.functional.gelu(ac @ w1.t() + b1) @ w2.t() + b2
    out = torch.full((d.ncrop, DO), float("nan"), device="cuda")
    xf = (dev(mean), dev(invstd), dev(gamma), dev(beta), 0) if bn else None
    ops.proj_fwd(dev(a).view(-1, C), dev(w1), dev(b1), dev(w2), dev(b2), out, d, DO, xf=xf)
    assert rel_l2(out.cpu(), ref) < TOL
    assert (out.cpu().double() - ref).abs().max() < 2e-6 * ref.abs().max()


def test_lift_fwd_bwd(ops):
    torch.manual_seed(9)
    B, T, H, W, Cin, C, pad = 2, 4, 5, 6, 3, 64, 6
    d = ops.Dims(B, T, H, W, Cin, C, pad)
    x = torch.randn(B, T, H, W, Cin, dtype=torch.float64)
    grids = [torch.tensor(np.linspace(0, 1, n), dtype=torch.float) for n in (T, H, W)]
    w0 = torch.randn(C, Cin + 3, dtype=torch.float64, requires_grad=True)
    b0 = torch.randn(C, dtype=torch.float64, requires_grad=True)
    grid = torch.stack([grids[0].double().view(1, T, 1, 1).expand(B, T, H, W),
                        grids[1].double().view(1, 1, H, 1).expand(B, T, H, W),
                        grids[2].double().view(1, 1, 1, W).expand(B, T, H, W)], dim=-1)
    inner = torch.cat([x, grid], -1) @ w0.t() + b0
    ref = torch.zeros(B, d.Tp, d.Hp, d.Wp, C, dtype=torch.float64)
    ref[:, :T, :H, :W] = inner.detach()
    out = torch.full((d.ncell, C), float("nan"), device="cuda")
    dg = [g.cuda() for g in grids]
    ops.lift_pad_fwd(dev(x), dg, dev(w0.detach()), dev(b0.detach()), out, d)
    assert rel_l2(out.cpu().view_as(ref), ref) < TOL
    g = torch.randn(B, d.Tp, d.Hp, d.Wp, C, dtype=torch.float64)
    inner.backward(g[:, :T, :H, :W])
    rows = ops._lib.query("rpb_lift_bwd_rows")
    F = Cin + 3
    part = torch.zeros(rows, C * F + C, device="cuda")
    ops.lift_bwd(dev(g).view(-1, C), dev(x), dg, part, d)
    s = part.double().sum(0).cpu()
    assert rel_l2(s[:C * F].view(C, F), w0.grad) < TOL
    assert rel_l2(s[C * F:], b0.grad) < TOL


@pytest.mark.parametrize("gelu", [True, False])
def test_batchnorm_fwd_bwd(ops, gelu):
    torch.manual_seed(11)
    ncell, C = 5000, 64
    s = (torch.randn(ncell, C, dtype=torch.float64) * 1.7 + 0.4).requires_grad_(True)
    gamma = (torch.rand(C, dtype=torch.float64) + 0.5).requires_grad_(True)
    beta = torch.randn(C, dtype=torch.float64, requires_grad=True)
    mean = s.mean(0)
    var = ((s - mean) ** 2).mean(0)
    z = (s - mean) / torch.sqrt(var + 1e-5) * gamma + beta
    y = torch.nn.functional.gelu(z) if gelu else z
    gy = torch.randn_like(y)
    y.backward(gy)
    S = dev(s.detach())
    sums = torch.stack([s.detach().sum(0), (s.detach() ** 2).sum(0)]).reshape(-1).cuda()
    mean_d, invstd_d = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    ops.bn_finalize(sums, ncell, 1e-5, 0.1, mean_d, invstd_d, rm, rv, C)
    assert rel_l2(mean_d.cpu(), mean.detach()) < TOL
    assert rel_l2(invstd_d.cpu(), 1 / torch.sqrt(var.detach() + 1e-5)) < TOL
    assert rel_l2(rm.cpu(), 0.1 * mean.detach()) < TOL
    assert rel_l2(rv.cpu(), 0.9 + 0.1 * var.detach() * ncell / (ncell - 1)) < TOL
    Y = torch.empty(ncell, C, device="cuda")
    ops.bn_act_fwd(S, mean_d, invstd_d, dev(gamma.detach()), dev(beta.detach()), Y, ncell, C, gelu)
    assert rel_l2(Y.cpu(), y.detach()) < TOL
    rows = ops.bn_bwd_rows()
    part = torch.zeros(rows, 2 * C, device="cuda")
    GY = dev(gy)
    ops.bn_bwd_reduce(S, GY, mean_d, invstd_d, dev(gamma.detach()), dev(beta.detach()), part, ncell, C, gelu)
    sums32 = torch.empty(2 * C, device="cuda")
    ops.reduce_partials(part, rows, 2 * C, out_f32=sums32)
    assert rel_l2(sums32[:C].cpu(), beta.grad) < 5e-6
    assert rel_l2(sums32[C:].cpu(), gamma.grad) < 5e-6
    GS = torch.empty(ncell, C, device="cuda")
    ops.bn_bwd_apply(S, GY, mean_d, invstd_d, dev(gamma.detach()), dev(beta.detach()), sums32, ncell, GS, ncell, C, gelu)
    assert rel_l2(GS.cpu(), s.grad) < 5e-6
    # eval-mode prep
    inv = torch.empty(C, device="cuda")
    ops.bn_eval_prep(rv, 1e-5, inv, C)
    assert rel_l2(inv.cpu(), 1 / torch.sqrt(rv.cpu().double() + 1e-5)) < TOL


def test_mse_adam_affine_reduce(ops):
    torch.manual_seed(13)
    n = 100003
    p, t = torch.randn(n, dtype=torch.float64), torch.randn(n, dtype=torch.float64)
    elem, gout = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
    part = torch.zeros(ops.mse_rows(), device="cuda")
    ops.mse(dev(p), dev(t), elem, gout, part, n, 1.0 / n)
    loss = torch.empty(1, device="cuda")
    ops.reduce_partials(part, part.numel(), 1, out_f32=loss, scale=1.0 / n)
    assert rel_l2(elem.cpu(), (p - t) ** 2) < TOL
    assert rel_l2(gout.cpu(), 2 * (p - t) / n) < TOL
    assert abs(float(loss) - float(((p - t) ** 2).mean())) < 1e-6
    # Adam vs torch.optim.Adam over 3 steps with changing lr
    w = torch.randn(1003, dtype=torch.float32)
    wt = w.clone().requires_grad_(True)
    opt = torch.optim.Adam([wt], lr=1e-2)
    P, M_, V = w.clone().cuda(), torch.zeros(1003, device="cuda"), torch.zeros(1003, device="cuda")
    for step in range(1, 4):
        g = torch.randn(1003)
        lr = 1e-2 * (0.7 ** step)
        for gr in opt.param_groups:
            gr["lr"] = lr
        wt.grad = g.clone()
        opt.step()
        ops.adam_step(P, g.cuda(), M_, V, 1003, lr, 0.9, 0.999, 1e-8, step)
    assert rel_l2(P.cpu(), wt.detach()) < 1e-6
    # rollout affine with control channels
    ncell, Cp, Cx = 999, 3, 2
    pred, para = torch.randn(ncell, Cp), torch.randn(ncell, Cx)
    mt, st = torch.randn(Cp), torch.rand(Cp) + 0.5
    mi, si = torch.randn(Cp + Cx), torch.rand(Cp + Cx) + 0.5
    ref = (torch.cat([pred * st + mt, para], -1) - mi) / si
    out = torch.empty(ncell, Cp + Cx, device="cuda")
    ops.rollout_affine(dev(pred), dev(para), out, ncell, Cp, Cx, dev(mt), dev(st), dev(mi), dev(si))
    assert rel_l2(out.cpu(), ref) < 1e-6
    out2 = torch.empty(ncell, Cp, device="cuda")
    ops.rollout_affine(dev(pred), None, out2, ncell, Cp, 0, None, None, None, None)
    assert torch.equal(out2.cpu(), pred)
    x = torch.randn(50, 7, 3)
    o = torch.empty(50, 7, 3, device="cuda")
    ops.channel_affine(dev(x), o, x.numel(), 3, dev(mt), dev(st), False)
    assert rel_l2(o.cpu(), (x - mt) / st) < 1e-6
    ops.channel_affine(dev(x), o, x.numel(), 3, dev(mt), dev(st), True)
    assert rel_l2(o.cpu(), x * st + mt) < 1e-6
    # strided partial reduction
    part2 = torch.randn(37, 50)
    o2 = torch.empty(20, device="cuda")
    ops.reduce_partials(dev(part2), 37, 20, out_f32=o2, row_stride=50, col0=11)
    assert rel_l2(o2.cpu(), part2[:, 11:31].double().sum(0)) < 1e-6


def test_errors_are_loud(ops):
    from realpdebench_amd._lib import RpbError
    x = torch.zeros(4, 33, device="cuda")
    with pytest.raises(RpbError):
        ops.axis_gemm(x, x, torch.zeros(2, 2, device="cuda"), 1, 2, 2, 33, 66, 33, 66, 33)     # N % 32 != 0
    with pytest.raises(RpbError):
        ops.bn_eval_prep(torch.zeros(4), 1e-5, torch.zeros(4), 4)                                 # CPU tensor


def _xf_ref(x, mean, invstd, gamma, beta, gelu):
    z = (x - mean) * invstd * gamma + beta
    return torch.nn.functional.gelu(z) if gelu else z


@pytest.mark.parametrize("gelu", [True, False])
def test_lazy_activation_in_consumers(ops, gelu):
    """Every consumer of a layer output applies BatchNorm(+GELU) on load (fno.py:117-119 fused): W-stage, cell_mix,
    cell_wgrad and the projection must equal `op(act(bn(s)))` computed in fp64."""
    torch.manual_seed(21)
    C, Wp, rows, K2 = 64, 22, 9, 32
    ncell = rows * Wp
    s = torch.randn(ncell, C, dtype=torch.float64) * 1.5 + 0.3
    mean, invstd = torch.randn(C, dtype=torch.float64) * 0.2, torch.rand(C, dtype=torch.float64) + 0.5
    gamma, beta = torch.rand(C, dtype=torch.float64) + 0.5, torch.randn(C, dtype=torch.float64) * 0.3
    a = _xf_ref(s, mean, invstd, gamma, beta, gelu)
    xf = (dev(mean), dev(invstd), dev(gamma), dev(beta), gelu)
    S = dev(s)
    # W stage
    M = torch.randn(32, Wp, dtype=torch.float64)
    out = torch.empty(rows, 32, C, device="cuda")
    ops.axis_gemm(S, out, dev(M.t()), rows, Wp, 32, C, Wp * C, C, 32 * C, C, xf=xf)
    assert rel_l2(out.cpu(), torch.einsum("ok,gkn->gon", M, a.view(rows, Wp, C))) < 3e-6
    # cell_mix
    Wc, bias = torch.randn(C, C, dtype=torch.float64) / 8, torch.randn(C, dtype=torch.float64)
    z2, GW = torch.randn(rows, K2, C, dtype=torch.float64), torch.randn(Wp, K2, dtype=torch.float64)
    ref = torch.einsum("wk,gkc->gwc", GW, z2).reshape(ncell, C) + a @ Wc.t() + bias
    o2 = torch.empty(ncell, C, device="cuda")
    ops.cell_mix(S, dev(Wc), dev(bias), dev(z2), dev(GW.t()), o2, None, ncell, C, C, K2, Wp, xf=xf)
    assert rel_l2(o2.cpu(), ref) < 3e-6
    # cell_wgrad
    gs = torch.randn(ncell, C, dtype=torch.float64)
    slots = ops.cell_wgrad_slots(ncell, C, C)
    part = torch.zeros(slots, C * C + C, device="cuda")
    ops.cell_wgrad(dev(gs), S, part, ncell, C, C, xf=xf)
    assert rel_l2(part.double().sum(0).cpu()[:C * C].view(C, C), gs.t() @ a) < 3e-6
    # projection (forward + gu of backward)
    B, T, H, W, pad, DO = 1, 2, 4, 32, 2, 2
    d = ops.Dims(B, T, H, W, 2, C, pad)
    sp = torch.randn(d.ncell, C, dtype=torch.float64)
    ap = _xf_ref(sp, mean, invstd, gamma, beta, gelu).view(B, d.Tp, d.Hp, d.Wp, C)[:, :T, :H, :W].reshape(-1, C)
    w1 = (torch.randn(128, C, dtype=torch.float64) / 8).requires_grad_(True)
    b1 = torch.randn(128, dtype=torch.float64)
    w2, b2 = torch.randn(DO, 128, dtype=torch.float64) / 11, torch.randn(DO, dtype=torch.float64)
    u = ap @ w1.t() + b1
    u.retain_grad()
    oref = torch.nn.functional.gelu(u) @ w2.t() + b2
    gout = torch.randn_like(oref)
    oref.backward(gout)
    o3 = torch.empty(d.ncrop, DO, device="cuda")
    args = (dev(sp), dev(w1.detach()), dev(b1), dev(w2), dev(b2))
    ops.proj_fwd(*args, o3, d, DO, xf=xf)
    assert rel_l2(o3.cpu(), oref.detach()) < 3e-6
    gu = torch.empty(d.ncrop, 128, device="cuda")
    pp = torch.zeros(ops.proj_slots(d.ncrop, C, DO), DO * 128 + 128 + DO, device="cuda")
    ops.proj_bwd(*args, dev(gout), gu, pp, d, DO, xf=xf)
    assert rel_l2(gu.cpu(), u.grad) < 5e-6


@pytest.mark.parametrize("C,gelu,lazy_x", [(64, True, True), (64, False, False), (32, True, False)])
def test_fused_bn_bwd_row(ops, C, gelu, lazy_x):
    """rpb_bn_bwd_row == bn_bwd_apply + adjoint W stage + conv weight gradient, each stated in fp64."""
    torch.manual_seed(31 + C)
    G, Wp, K2 = 13, 22, 32 if C == 64 else 8
    ncell = G * Wp
    s = torch.randn(ncell, C, dtype=torch.float64) * 1.3 + 0.2
    gy = torch.randn(ncell, C, dtype=torch.float64)
    xs = torch.randn(ncell, C, dtype=torch.float64)
    mean, invstd = s.mean(0), 1 / torch.sqrt(s.var(0, unbiased=False) + 1e-5)
    gamma, beta = torch.rand(C, dtype=torch.float64) + 0.5, torch.randn(C, dtype=torch.float64) * 0.3
    sh = (s - mean) * invstd
    z = sh * gamma + beta
    if gelu:
        zz = z.clone().requires_grad_(True)
        torch.nn.functional.gelu(zz).backward(gy)
        gz = zz.grad
    else:
        gz = gy
    sums = torch.cat([gz.sum(0), (gz * sh).sum(0)])
    gs_ref = gamma * invstd * (gz - sums[:C] / ncell - sh * sums[C:] / ncell)
    pm, pi = torch.randn(C, dtype=torch.float64) * 0.1, torch.rand(C, dtype=torch.float64) + 0.5
    pg, pb = torch.rand(C, dtype=torch.float64) + 0.5, torch.randn(C, dtype=torch.float64) * 0.2
    x_ref = _xf_ref(xs, pm, pi, pg, pb, True) if lazy_x else xs
    GWt = torch.randn(K2, Wp, dtype=torch.float64)
    y1_ref = torch.einsum("ok,gkc->goc", GWt, gs_ref.view(G, Wp, C))
    xf = (dev(pm), dev(pi), dev(pg), dev(pb), True) if lazy_x else None
    g = dev(gy)
    Y1 = torch.full((G, K2, C), float("nan"), device="cuda")
    slots = ops.bn_bwd_row_slots(G)
    part = torch.full((slots, C * C + C), float("nan"), device="cuda")
    ops.bn_bwd_row(dev(s), g, dev(xs), g, dev(mean), dev(invstd), dev(gamma), dev(beta), dev(sums), ncell, gelu, xf,
                   dev(GWt.t()), Y1, part, G, Wp, C, K2)
    assert rel_l2(g.cpu(), gs_ref) < 5e-6                       # in place over gy
    assert rel_l2(Y1.cpu(), y1_ref) < 5e-6
    tot = part.double().sum(0).cpu()
    assert rel_l2(tot[:C * C].view(C, C), gs_ref.t() @ x_ref) < 5e-6
    assert float((tot[C * C:] - gs_ref.sum(0)).abs().max()) < 1e-4 * float(gs_ref.abs().sum(0).max())


@pytest.mark.parametrize("Wp,rows,K2,C", [(134, 7, 32, 64), (70, 11, 32, 64), (38, 5, 24, 64), (33, 3, 7, 64), (134, 1, 32, 64),
                                          (70, 11, 32, 128), (134, 7, 32, 128), (33, 3, 7, 128), (70, 300, 32, 128)])
def test_cell_mix_bf16_pipe_all_modes(ops, Wp, rows, K2, C):
    """The C = 64 and C = 128 (one 64-channel output half per workgroup; configs/fsi/fno.yaml, the Galerkin regressor) spectral
    cell_mix runs on the bf16 matrix pipe from operands split into three bf16 planes
    (csrc/rpb_cmx.hip): fp32-grade against fp64 in every mode -- plain / transposed weight, lazy input transform,
    BatchNorm forward sums, BatchNorm-backward sums (with and without GELU) and the eval output transform -- on row
    lengths that make tiles straddle (b,t,h) lines and on cell counts that are not a multiple of the 32-cell tile."""
    torch.manual_seed(Wp * 7 + K2)
    ncell = rows * Wp
    assert ops.cell_mix_writes_gz(ncell, C, C, K2, Wp, True)                   # = "this shape runs on the bf16-pipe kernel"
    if C == 64:
        assert ops.cell_mix_stat_rows(ncell, C, C, K2, Wp, True) % 8 == 0      # 8 waves per workgroup
    f8 = dict(dtype=torch.float64)
    s = torch.randn(ncell, C, **f8) * 1.5 + 0.3
    Wc, bias = torch.randn(C, C, **f8) / 8, torch.randn(C, **f8)
    z2, GW = torch.randn(rows, K2, C, **f8), torch.randn(Wp, K2, **f8)
    mean, invstd = torch.randn(C, **f8) * 0.2, torch.rand(C, **f8) + 0.5
    gamma, beta = torch.rand(C, **f8) + 0.5, torch.randn(C, **f8) * 0.3
    spec = torch.einsum("wk,gkc->gwc", GW, z2).reshape(ncell, C)
    for gelu in (True, False):
        a = _xf_ref(s, mean, invstd, gamma, beta, gelu)
        xf = (dev(mean), dev(invstd), dev(gamma), dev(beta), gelu)
        ref = spec + a @ Wc.t() + bias
        out = torch.full((ncell, C), float("nan"), device="cuda")
        nrows = ops.cell_mix_stat_rows(ncell, C, C, K2, Wp, True)
        part = torch.zeros(nrows, 2, C, device="cuda")
        ops.cell_mix(dev(s), dev(Wc), dev(bias), dev(z2), dev(GW.t()), out, part, ncell, C, C, K2, Wp, xf=xf)
        assert rel_l2(out.cpu(), ref) < 2e-6
        tot = part.double().sum(0).cpu()
        assert rel_l2(tot[0], ref.sum(0)) < 1e-5 and rel_l2(tot[1], (ref ** 2).sum(0)) < 1e-5
        # eval: output transform act(BN(out)) with another layer's statistics
        om, oi = torch.randn(C, **f8) * 0.1, torch.rand(C, **f8) + 0.5
        og, ob = torch.rand(C, **f8) + 0.5, torch.randn(C, **f8) * 0.2
        ops.cell_mix(dev(s), dev(Wc), dev(bias), dev(z2), dev(GW.t()), out, None, ncell, C, C, K2, Wp, xf=xf,
                     oxf=(dev(om), dev(oi), dev(og), dev(ob), gelu))
        assert rel_l2(out.cpu(), _xf_ref(ref, om, oi, og, ob, gelu)) < 3e-6
        # backward flavour: transposed weight, no bias, BatchNorm-backward sums of the layer that produced the input
        g = torch.randn(ncell, C, **f8)
        ref2 = spec + g @ Wc
        sh = (s - mean) * invstd
        if gelu:
            zz = (sh * gamma + beta).clone().requires_grad_(True)
            torch.nn.functional.gelu(zz).backward(ref2)
            gz = zz.grad
        else:
            gz = ref2
        nb = ops.cell_mix_stat_rows(ncell, C, C, K2, Wp, True, True)
        part2 = torch.zeros(nb, 2, C, device="cuda")
        ops.cell_mix(dev(g), dev(Wc), None, dev(z2), dev(GW.t()), out, part2, ncell, C, C, K2, Wp, transpose_w=True,
                     bnb=(dev(s),) + xf)
        assert rel_l2(out.cpu(), ref2) < 2e-6
        tot = part2.double().sum(0).cpu()
        assert rel_l2(tot[0], gz.sum(0)) < 2e-5 and rel_l2(tot[1], (gz * sh).sum(0)) < 2e-5
        if gelu:            # "store gz": the tensor handed to the BatchNorm-backward apply already carries gelu'(z)
            assert ops.cell_mix_writes_gz(ncell, C, C, K2, Wp, True)
            part2.zero_()
            ops.cell_mix(dev(g), dev(Wc), None, dev(z2), dev(GW.t()), out, part2, ncell, C, C, K2, Wp, transpose_w=True,
                         bnb=(dev(s),) + xf, write_gz=True)
            assert rel_l2(out.cpu(), gz) < 3e-6
            tot = part2.double().sum(0).cpu()
            assert rel_l2(tot[0], gz.sum(0)) < 2e-5 and rel_l2(tot[1], (gz * sh).sum(0)) < 2e-5
    ops.cell_mix(dev(g), dev(Wc), None, dev(z2), dev(GW.t()), out, None, ncell, C, C, K2, Wp, transpose_w=True)
    assert rel_l2(out.cpu(), spec + g @ Wc) < 2e-6


@pytest.mark.parametrize("C,Wp,rows", [(64, 134, 700), (64, 38, 3000), (128, 70, 900), (64, 134, 5)])
def test_cell_mix_line_claim_modes_agree(ops, C, Wp, rows):
    """rpb_line_claim_set: the lines of a cell_mix launch dealt round-robin (0), claimed from the workgroup's LDS counter (1) or from one
    counter in HBM (2).  The OUTPUT tensor is bit-identical in every mode (a line's arithmetic does not depend on which wave walks it),
    the statistics agree to fp32 summation order, every line is produced exactly once (NaN-prefilled output), and the chip-wide counter
    is back at zero after each launch -- 300 launches wrap its ring of counters."""
    from realpdebench_amd import _lib
    torch.manual_seed(C + Wp)
    K2, ncell = 32, rows * Wp
    f = dict(device="cuda", dtype=torch.float32)
    x, z2 = torch.randn(ncell, C, **f), torch.randn(rows, K2, C, **f)
    Wc, bias, GWt = torch.randn(C, C, **f) / 8, torch.randn(C, **f), torch.randn(K2, Wp, **f)
    xf = (torch.randn(C, **f) * 0.2, torch.rand(C, **f) + 0.5, torch.rand(C, **f) + 0.5, torch.randn(C, **f) * 0.3, True)
    nrows = ops.cell_mix_stat_rows(ncell, C, C, K2, Wp, True)
    outs, sums = {}, {}
    try:
        for mode in (0, 1, 2):
            _lib.call("rpb_line_claim_set", mode)
            out = torch.full((ncell, C), float("nan"), **f)
            part = torch.zeros(nrows, 2, C, **f)
            ops.cell_mix(x, Wc, bias, z2, GWt, out, part, ncell, C, C, K2, Wp, xf=xf)
            outs[mode], sums[mode] = out, part.double().sum(0)
            assert not torch.isnan(out).any()
        assert torch.equal(outs[1], outs[0]) and torch.equal(outs[2], outs[0])
        for mode in (1, 2):
            assert rel_l2(sums[mode].cpu(), sums[0].cpu()) < 1e-6
        _lib.call("rpb_line_claim_set", 2)
        out = torch.empty(ncell, C, **f)
        for _ in range(300 if rows < 100 else 20):      # > the ring of 256 counters on the small case
            out.fill_(float("nan"))
            ops.cell_mix(x, Wc, bias, z2, GWt, out, None, ncell, C, C, K2, Wp, xf=xf)
        assert torch.equal(out, outs[0])
    finally:
        _lib.call("rpb_line_claim_set", -1)


@pytest.mark.parametrize("Wp,rows,K2", [(134, 9, 32), (70, 11, 32), (38, 5, 24), (33, 3, 7), (134, 1, 32)])
@pytest.mark.parametrize("gelu,write_gz", [(True, True), (True, False), (False, False)])
def test_cell_mix_with_the_conv_weight_gradient(ops, Wp, rows, K2, gelu, write_gz):
    """rpb_cell_mix_wgrad (csrc/rpb_cmw.hip) == the STATS = 2 backward cell_mix + d convs.weight = gs^T act(BN(s_prev)), each stated in
    fp64: rows that end inside a tile (Wp % 32 != 0), fewer lines than waves, weights that are far from symmetric (transpose-detecting)."""
    torch.manual_seed(Wp * 11 + K2 + int(gelu))
    C = 64
    ncell = rows * Wp
    assert ops.cell_mix_wgrad_supported(ncell, K2, Wp)
    f8 = dict(dtype=torch.float64)
    s = torch.randn(ncell, C, **f8) * 1.5 + 0.3
    gs = torch.randn(ncell, C, **f8) * torch.linspace(0.2, 3.0, C, **f8)      # per-channel scales: a transposed dWc would not match
    Wc = torch.randn(C, C, **f8) / 8
    z2, FW = torch.randn(rows, K2, C, **f8), torch.randn(Wp, K2, **f8)
    mean, invstd = torch.randn(C, **f8) * 0.2, torch.rand(C, **f8) + 0.5
    gamma, beta = torch.rand(C, **f8) + 0.5, torch.randn(C, **f8) * 0.3
    sh = (s - mean) * invstd
    act = _xf_ref(s, mean, invstd, gamma, beta, gelu)
    gx = torch.einsum("wk,gkc->gwc", FW, z2).reshape(ncell, C) + gs @ Wc
    if gelu:
        zz = (sh * gamma + beta).clone().requires_grad_(True)
        torch.nn.functional.gelu(zz).backward(gx)
        gz = zz.grad
    else:
        gz = gx
    slots = ops.cell_mix_wgrad_slots(ncell, Wp)
    out = torch.full((ncell, C), float("nan"), device="cuda")
    sp = torch.full((slots, 2, C), float("nan"), device="cuda")
    wp = torch.full((slots, C, C), float("nan"), device="cuda")
    ops.cell_mix_wgrad(dev(gs), dev(Wc), dev(z2), dev(FW.t()), out, sp, wp, ncell, K2, Wp,
                       (dev(s), dev(mean), dev(invstd), dev(gamma), dev(beta), gelu), write_gz=write_gz)
    assert rel_l2(out.cpu(), gz if write_gz else gx) < 3e-6
    tot = sp.double().sum(0).cpu()
    assert rel_l2(tot[0], gz.sum(0)) < 2e-5 and rel_l2(tot[1], (gz * sh).sum(0)) < 2e-5
    dW = wp.double().sum(0).cpu()
    assert rel_l2(dW, gs.t() @ act) < 3e-6
    # ... and the row kernel without its weight-gradient operand: gs, Y1 and the bias sums as before, the C x C block untouched
    G = rows
    gy = torch.randn(ncell, C, **f8)
    sums = torch.cat([gy.sum(0), (gy * sh).sum(0)])
    gs_ref = gamma * invstd * (gy - sums[:C] / ncell - sh * sums[C:] / ncell)
    GWt = torch.randn(K2, Wp, **f8)
    g = dev(gy)
    Y1 = torch.full((G, K2, C), float("nan"), device="cuda")
    rs = ops.bn_bwd_row_slots(G)
    part = torch.full((rs, C * C + C), 7.0, device="cuda")
    ops.bn_bwd_row(dev(s), g, None, g, dev(mean), dev(invstd), dev(gamma), dev(beta), dev(sums), ncell, False, None,
                   dev(GWt.t()), Y1, part, G, Wp, C, K2)
    assert rel_l2(g.cpu(), gs_ref) < 5e-6
    assert rel_l2(Y1.cpu(), torch.einsum("ok,gkc->goc", GWt, gs_ref.view(G, Wp, C))) < 5e-6
    tot = part.double().sum(0).cpu()
    assert float((tot[C * C:] - gs_ref.sum(0)).abs().max()) < 1e-4 * float(gs_ref.abs().sum(0).max())


@pytest.mark.parametrize("store_gs", [True, False])
def test_bn_bwd_row_on_the_feature_fields(ops, store_gs):
    """rpb_bn_bwd_row_feat (layer 0: the weight-gradient operand is the 8-float feature tensor) in fp64 terms -- gs, the adjoint W stage
    and the field moments sum_cells gs (x) phi -- and its gs == NULL form: same Y1 and moments, the gradient tensor left untouched."""
    torch.manual_seed(77)
    C, G, Wp, K2, FW = 64, 11, 38, 24, 8
    ncell = G * Wp
    f8 = dict(dtype=torch.float64)
    s = torch.randn(ncell, C, **f8) * 1.3 + 0.2
    gy = torch.randn(ncell, C, **f8)
    phi = torch.randn(ncell, FW, **f8)
    mean, invstd = s.mean(0), 1 / torch.sqrt(s.var(0, unbiased=False) + 1e-5)
    gamma, beta = torch.rand(C, **f8) + 0.5, torch.randn(C, **f8) * 0.3
    sh = (s - mean) * invstd
    sums = torch.cat([gy.sum(0), (gy * sh).sum(0)])
    gs_ref = gamma * invstd * (gy - sums[:C] / ncell - sh * sums[C:] / ncell)
    GWt = torch.randn(K2, Wp, **f8)
    g = dev(gy)
    g0 = g.clone()
    Y1 = torch.full((G, K2, C), float("nan"), device="cuda")
    slots = ops.bn_bwd_row_slots(G)
    part = torch.zeros(slots, C * C + C, device="cuda")
    ops.bn_bwd_row_feat(dev(s), g, dev(phi), g if store_gs else None, dev(mean), dev(invstd), dev(gamma), dev(beta), dev(sums), ncell,
                        False, dev(GWt.t()), Y1, part, G, Wp, C, K2, FW)
    if store_gs:
        assert rel_l2(g.cpu(), gs_ref) < 5e-6
    else:
        assert torch.equal(g, g0)
    assert rel_l2(Y1.cpu(), torch.einsum("ok,gkc->goc", GWt, gs_ref.view(G, Wp, C))) < 5e-6
    tot = part.double().sum(0).cpu()
    assert rel_l2(tot[:C * C].view(C, C)[:, :FW], gs_ref.t() @ phi) < 5e-6
    assert float((tot[C * C:] - gs_ref.sum(0)).abs().max()) < 1e-4 * float(gs_ref.abs().sum(0).max())


@pytest.mark.parametrize("gelu,Wp,K2", [(True, 70, 32), (False, 38, 32), (False, 134, 40), (True, 45, 48)])
def test_bn_bwd_row_width_128(ops, gelu, Wp, K2):
    """rpb_bn_bwd_row_c128 (width 128: two 64-channel half launches of the C = 64 row kernel over 512-byte rows) in fp64 terms: the
    BatchNorm(+GELU) backward apply, in place, and the adjoint W stage Y1 = GW^T gs."""
    torch.manual_seed(128 + Wp)          # K2 > 32 (the Galerkin regressor's modes (4, 16, 20)): three 16-mode row tiles
    C, G = 128, 13
    ncell = G * Wp
    f8 = dict(dtype=torch.float64)
    s = torch.randn(ncell, C, **f8) * 1.3 + 0.2
    gy = torch.randn(ncell, C, **f8)
    mean, invstd = s.mean(0), 1 / torch.sqrt(s.var(0, unbiased=False) + 1e-5)
    gamma, beta = torch.rand(C, **f8) + 0.5, torch.randn(C, **f8) * 0.3
    sh = (s - mean) * invstd
    gz = gy
    if gelu:
        z = (sh * gamma + beta).requires_grad_(True)
        torch.nn.functional.gelu(z).backward(gy)
        gz = z.grad
    sums = torch.cat([gz.sum(0), (gz * sh).sum(0)])
    gs_ref = gamma * invstd * (gz - sums[:C] / ncell - sh * sums[C:] / ncell)
    GWt = torch.randn(K2, Wp, **f8)
    assert ops.bn_bwd_row_c128_supported(Wp, K2)
    g = dev(gy)
    Y1 = torch.full((G, K2, C), float("nan"), device="cuda")
    part = torch.empty(2 * ops.bn_bwd_row_slots(G) * (64 * 64 + 64), device="cuda")
    ops.bn_bwd_row_c128(dev(s), g, g, dev(mean), dev(invstd), dev(gamma), dev(beta), dev(sums), ncell, gelu, dev(GWt.t()), Y1, part,
                        G, Wp, K2)
    assert rel_l2(g.cpu(), gs_ref) < 5e-6
    assert rel_l2(Y1.cpu(), torch.einsum("ok,gkc->goc", GWt, gs_ref.view(G, Wp, C))) < 5e-6


def _bf16_ulps(a, b):
    """|a - b| in units of the bf16 spacing at |b| (both bf16 tensors)."""
    a, b = a.float(), b.float()
    ulp = torch.clamp(b.abs(), min=1e-30).log2().floor().exp2() * 2.0 ** -7
    return (a - b).abs() / ulp


@pytest.mark.parametrize("Wp,rows,K2", [(70, 9, 32), (134, 3, 24)])
def test_bf16_storage_kernels(ops, Wp, rows, K2):
    """BASELINE.json configs[4] activation storage: lift / W stage / cell_mix / projection with bf16 activations are the computation on the
    (exactly representable) bf16 inputs with the fp32 constants (weights, stage matrices) taken to 2^-16 -- two bf16 planes, round 5: the
    third plane is 1 / 128 of the rounding a stored operand already carries (csrc/rpb_common.h, RPB_BF16_CONST_PLANES) -- rounded once to
    nearest even on store: against fp64 on the same bf16 inputs the stored bf16 values differ by at most one unit in the last place, and
    fp32 outputs hold 3e-5 (2e-6 with -DRPB_BF16_CONST_PLANES=3)."""
    TOLB = 3e-5
    torch.manual_seed(Wp + K2)
    C = 64
    ncell = rows * Wp
    f8 = dict(dtype=torch.float64)
    xb = (torch.randn(ncell, C) * 1.3).to(torch.bfloat16)
    x8 = xb.double()
    # ---- W stage, bf16 in
    M = torch.randn(32, Wp, **f8)
    out = torch.full((rows, 32, C), float("nan"), device="cuda")
    ops.axis_gemm_bf16in(xb.cuda(), out, dev(M.t()), rows, Wp, 32, C, Wp * C, C, 32 * C, C)
    assert rel_l2(out.cpu(), torch.einsum("ok,gkn->gon", M, x8.view(rows, Wp, C))) < TOLB
    ops.axis_gemm_bf16in(xb.cuda(), out, dev(M.t()), rows, Wp, 32, C, Wp * C, C, 32 * C, C, k_valid=Wp - 6)
    xx = x8.view(rows, Wp, C).clone()
    xx[:, Wp - 6:] = 0
    assert rel_l2(out.cpu(), torch.einsum("ok,gkn->gon", M, xx)) < TOLB
    # ---- cell_mix, bf16 in / out, eval output transform
    Wc, bias = torch.randn(C, C, **f8) / 8, torch.randn(C, **f8)
    z2, GW = torch.randn(rows, K2, C, **f8), torch.randn(Wp, K2, **f8)
    om, oi = torch.randn(C, **f8) * 0.1, torch.rand(C, **f8) + 0.5
    og, ob = torch.rand(C, **f8) + 0.5, torch.randn(C, **f8) * 0.2
    for gelu in (True, False):
        ref = _xf_ref(torch.einsum("wk,gkc->gwc", GW, z2).reshape(ncell, C) + x8 @ Wc.t() + bias, om, oi, og, ob, gelu)
        o = torch.zeros(ncell, C, device="cuda", dtype=torch.bfloat16)
        ops.cell_mix_bf16(xb.cuda(), dev(Wc), dev(bias), dev(z2), dev(GW.t()), o, ncell, C, K2, Wp,
                          oxf=(dev(om), dev(oi), dev(og), dev(ob), gelu))
        # one bf16 unit in the last place, plus the 2^-16-level error of the O(10) sums for results that cancel to ~0
        refb = ref.to(torch.bfloat16)
        excess = (o.cpu().float() - refb.float()).abs() - 1e-3
        u = excess / (torch.clamp(refb.float().abs(), min=1e-30).log2().floor().exp2() * 2.0 ** -7)
        assert float(u.max()) <= 1.0 and float(((o.cpu() != refb) & (excess > 0)).float().mean()) < 0.02   # boundary cases only
        assert rel_l2(o.cpu().double(), ref) < 4e-3
    # ---- round 4: the spectra next to the activations stored as bf16 too -- z2 rows read as one exact plane, the fused W stage's rows and
    #      the inverse H stage's rows rounded once on store
    z2b = z2.to(torch.bfloat16)
    ref = _xf_ref(torch.einsum("wk,gkc->gwc", GW, z2b.double()).reshape(ncell, C) + x8 @ Wc.t() + bias, om, oi, og, ob, True)
    o = torch.zeros(ncell, C, device="cuda", dtype=torch.bfloat16)
    oxf = (dev(om), dev(oi), dev(og), dev(ob), True)
    ops.cell_mix_bf16(xb.cuda(), dev(Wc), dev(bias), z2b.cuda(), dev(GW.t()), o, ncell, C, K2, Wp, oxf=oxf)
    refb = ref.to(torch.bfloat16)
    big = refb.float().abs() > 0.5                       # (results that cancel to ~0 carry the absolute error of the O(10) sums)
    assert rel_l2(o.cpu().double(), ref) < 4e-3 and float(_bf16_ulps(o.cpu(), refb)[big].max()) <= 1.0
    assert float((o.cpu().float() - refb.float()).abs()[~big].max()) < 2e-3
    if ops.cell_mix_eval_dft_supported(ncell, K2, Wp, 32):
        FWt = torch.randn(Wp, 32, **f8)
        o2 = torch.zeros(ncell, C, device="cuda", dtype=torch.bfloat16)
        y1 = torch.full((rows, 32, C), float("nan"), device="cuda", dtype=torch.bfloat16)
        ops.cell_mix_eval_dft(xb.cuda(), dev(Wc), dev(bias), z2b.cuda(), dev(GW.t()), o2, ncell, K2, Wp, oxf, dev(FWt), 32, y1)
        assert torch.equal(o2, o)                                      # the same activations ...
        y1_ref = torch.einsum("wk,gwc->gkc", FWt, o2.cpu().double().view(rows, Wp, C))       # ... and the stage applied to the ROUNDED ones
        assert rel_l2(y1.cpu().double(), y1_ref) < 3e-3
        assert float((y1.cpu().float() - y1_ref.to(torch.bfloat16).float()).abs().max()) <= 2.0 ** -7 * float(y1_ref.abs().max()) * 1.01
    Mi = torch.randn(140, 48, **f8)                                     # inverse H stage shape: K = 48 -> O = 140, N = 2 strips
    zin = torch.randn(rows, 48, 2 * C, **f8)
    zo32 = torch.empty(rows, 140, 2 * C, device="cuda")
    zo16 = torch.full((rows, 140, 2 * C), float("nan"), device="cuda", dtype=torch.bfloat16)
    ops.axis_gemm(dev(zin), zo32, dev(Mi.t()), rows, 48, 140, 2 * C, 48 * 2 * C, 2 * C, 140 * 2 * C, 2 * C)
    ops.axis_gemm_bf16out(dev(zin), zo16, dev(Mi.t()), rows, 48, 140, 2 * C, 48 * 2 * C, 2 * C, 140 * 2 * C, 2 * C)
    assert torch.equal(zo16, zo32.to(torch.bfloat16))
    # ---- lift, bf16 out == round(fp32 lift)
    B, T, H, W, pad, Cin = 2, 3, 5, 34, 2, 3
    d = ops.Dims(B, T, H, W, Cin, C, pad)
    xin = torch.randn(B, T, H, W, Cin, device="cuda")
    grids = [torch.linspace(0, 1, n, device="cuda") for n in (T, H, W)]
    w0, b0 = torch.randn(C, Cin + 3, device="cuda"), torch.randn(C, device="cuda")
    a32 = torch.empty(d.ncell, C, device="cuda")
    a16 = torch.empty(d.ncell, C, device="cuda", dtype=torch.bfloat16)
    ops.lift_pad_fwd(xin, grids, w0, b0, a32, d)
    ops.lift_pad_fwd_bf16(xin, grids, w0, b0, a16, d)
    assert torch.equal(a16, a32.to(torch.bfloat16))
    # ---- projection, bf16 in == fp32 projection of the widened input
    w1, b1 = torch.randn(128, C, device="cuda") / 8, torch.randn(128, device="cuda")
    w2, b2 = torch.randn(2, 128, device="cuda") / 11, torch.randn(2, device="cuda")
    o32, o16 = torch.empty(d.ncrop, 2, device="cuda"), torch.empty(d.ncrop, 2, device="cuda")
    ops.proj_fwd(a16.float(), w1, b1, w2, b2, o32, d, 2)           # fp32 input: bf16 matrix pipe, split operands
    ops.proj_fwd_bf16(a16, w1, b1, w2, b2, o16, d, 2)              # bf16 input: one stored plane x two planes of fc1.weight
    assert rel_l2(o16, o32) < TOLB


@pytest.mark.parametrize("T,H,W,pad", [(3, 5, 64, 6), (2, 3, 40, 6), (2, 4, 33, 2)])
def test_lift_bf16_wide_input_on_the_matrix_pipe(ops, T, H, W, pad, monkeypatch):
    """rpb_lift_pad_fwd_bf16 at C_in = 16 (the combustion volume) runs csrc/rpb_lift_mx.hip: one MFMA K-step over 16 inputs + 3 coordinates +
    the bias, fp32-grade split products, bf16 out.  Against round(fp32 lift): equal except where the fp32 value sits on a rounding
    boundary (at most one bf16 unit, < 0.1 % of the elements); the pad -- whole lines and the cells w >= W -- is exact zeros."""
    torch.manual_seed(W)
    B, Cin, C = 2, 16, 64
    d = ops.Dims(B, T, H, W, Cin, C, pad)
    xin = torch.randn(B, T, H, W, Cin, device="cuda")
    grids = [torch.linspace(0, 1, n, device="cuda") for n in (T, H, W)]
    w0, b0 = torch.randn(C, Cin + 3, device="cuda"), torch.randn(C, device="cuda")
    a32 = torch.empty(d.ncell, C, device="cuda")
    a16 = torch.full((d.ncell, C), 7.0, device="cuda", dtype=torch.bfloat16)
    ops.lift_pad_fwd(xin, grids, w0, b0, a32, d)
    ops.lift_pad_fwd_bf16(xin, grids, w0, b0, a16, d)
    ref = a32.to(torch.bfloat16)
    neq = a16 != ref
    assert float(neq.float().mean()) < 1e-3
    # a correctly rounded bf16 of a value that agrees with the fp32 lift to fp32 round-off (sums of 20 O(1) terms: ~4e-6 absolute)
    assert bool(((a16.float() - a32).abs() <= 2.0 ** -8 * a32.abs() * (1 + 1e-3) + 4e-6).all())
    v = a16.view(B, d.Tp, d.Hp, d.Wp, C)
    assert float(v[:, T:].abs().max() if pad else 0) == 0 and float(v[:, :, H:].abs().max()) == 0 and float(v[:, :, :, W:].abs().max()) == 0
    assert float(v[:, :T, :H, :W].abs().min()) > 0
    # the vector kernel stays reachable (and is the bit-exact rounding of the fp32 lift)
    # (RPB_LIFT_MX is read once per process: checked in a child process by tools, not here)


@pytest.mark.parametrize("B,T,H,W,pad,DO,gelu,act", [(2, 3, 5, 32, 2, 2, False, 0), (1, 2, 4, 48, 3, 1, False, 0),
                                                      (1, 2, 3, 40, 6, 3, True, 1)])
def test_projection_backward_without_gu(ops, B, T, H, W, pad, DO, gelu, act):
    """rpb_proj_dgrad / rpb_proj_wgrad (bf16 matrix pipe, gh recomputed, never stored) against fp64 autograd of
    fno.py:121-125 on the cropped cells: gradient w.r.t. the padded layer output (zeros in the margin), the BatchNorm-backward
    sums, and the four parameter gradients; row lengths that are not a multiple of the 32-cell wave tile included."""
    torch.manual_seed(B * 100 + W)
    C = 64
    d = ops.Dims(B, T, H, W, 2, C, pad)
    assert ops.proj_bwd_fused_supported(C, DO, W, d.Wp)
    f8 = dict(dtype=torch.float64)
    s = (torch.randn(d.ncell, C, **f8) * 1.2 + 0.2).requires_grad_(True)
    mean, invstd = torch.randn(C, **f8) * 0.2, torch.rand(C, **f8) + 0.5
    gamma, beta = torch.rand(C, **f8) + 0.5, torch.randn(C, **f8) * 0.3
    w1 = (torch.randn(128, C, **f8) / 8).requires_grad_(True)
    b1 = torch.randn(128, **f8).requires_grad_(True)
    w2 = (torch.randn(DO, 128, **f8) / 11).requires_grad_(True)
    b2 = torch.randn(DO, **f8).requires_grad_(True)
    sh = (s - mean) * invstd
    a_full = _xf_ref(s, mean, invstd, gamma, beta, gelu)
    a = a_full.view(B, d.Tp, d.Hp, d.Wp, C)[:, :T, :H, :W].reshape(-1, C)
    u = a @ w1.t() + b1
    v = torch.nn.functional.silu(u) if act == 1 else torch.nn.functional.gelu(u)
    out = v @ w2.t() + b2
    gout = torch.randn_like(out)
    a_full.retain_grad()
    out.backward(gout)
    g_ref = a_full.grad                                   # gradient w.r.t. act(BN(s)): zero in the pad margin
    xf = (dev(mean), dev(invstd), dev(gamma), dev(beta), gelu)
    S = dev(s.detach())
    g = torch.full((d.ncell, C), float("nan"), device="cuda")
    rows = ops.proj_dgrad_slots(d)
    part = torch.zeros(rows, 2, C, device="cuda")
    args = (S, dev(w1.detach()), dev(b1.detach()), dev(w2.detach()), dev(gout))
    ops.proj_dgrad(*args, g, part, d, DO, xf, act=act)
    assert rel_l2(g.cpu(), g_ref) < 3e-6
    tot = part.double().sum(0).cpu()
    assert rel_l2(tot[0], g_ref.sum(0)) < 2e-5 and rel_l2(tot[1], (g_ref * sh.detach()).sum(0)) < 2e-5
    # the same dgrad reading gh from HBM (the default path: rpb_proj_bwd writes it, cell_wgrad reads it too)
    u2 = (a.detach() @ w1.detach().t() + b1.detach()).requires_grad_(True)
    v2 = torch.nn.functional.silu(u2) if act == 1 else torch.nn.functional.gelu(u2)
    (v2 @ w2.detach().t()).backward(gout)
    g.fill_(float("nan"))
    part.zero_()
    ops.proj_dgrad(S, dev(w1.detach()), dev(b1.detach()), dev(w2.detach()), None, g, part, d, DO, xf, act=act, gu=dev(u2.grad))
    assert rel_l2(g.cpu(), g_ref) < 3e-6
    tot = part.double().sum(0).cpu()
    assert rel_l2(tot[0], g_ref.sum(0)) < 2e-5 and rel_l2(tot[1], (g_ref * sh.detach()).sum(0)) < 2e-5
    slots, row, roles = ops.proj_wgrad_slots(d), ops.proj_wgrad_row(DO), ops.proj_wgrad_roles()
    HB = 128 // roles
    wp = torch.full((slots, row), float("nan"), device="cuda")
    ops.proj_wgrad(*args, wp, d, DO, xf, act=act)
    tot = wp.double().view(slots // roles, roles, row).sum(0).cpu()           # [role][row]
    dw1 = tot[:, :HB * 64].reshape(128, 64)
    dw2 = tot[:, HB * 64:HB * 64 + DO * HB].reshape(roles, DO, HB).permute(1, 0, 2).reshape(DO, 128)
    db1 = tot[:, HB * 64 + DO * HB:HB * 64 + DO * HB + HB].reshape(128)
    db2 = tot[:, HB * 64 + DO * HB + HB:].sum(0)
    assert rel_l2(dw1, w1.grad) < 5e-6 and rel_l2(dw2, w2.grad) < 5e-6
    assert rel_l2(db1, b1.grad) < 5e-6 and rel_l2(db2, b2.grad) < 5e-6


@pytest.mark.parametrize("B,T,H,W,pad,DO", [(2, 3, 5, 32, 2, 2), (1, 2, 4, 48, 3, 1), (1, 2, 3, 40, 6, 3), (2, 2, 3, 128, 6, 2),
                                             (1, 3, 2, 7, 6, 4), (5, 4, 7, 64, 6, 2)])
def test_head_backward_one_pass(ops, B, T, H, W, pad, DO):
    """rpb_head_bwd + rpb_head_bwd_finalize (csrc/rpb_pjf.hip: gh never in HBM, cells as MFMA rows, one LDS transposition for the data
    gradient, BatchNorm-backward sums derived from M = gh^T shat) against fp64 autograd of fno.py:121-125 on the cropped cells:
    gradient w.r.t. the padded layer output (zeros in the margin), the four parameter gradients and the BatchNorm-backward sums;
    row lengths that are not a multiple of the 32-cell tile, more lines than waves, and fewer."""
    torch.manual_seed(B * 100 + W + DO)
    C = 64
    d = ops.Dims(B, T, H, W, 2, C, pad)
    assert ops.head_bwd_supported(C, DO, W, d.Wp, False, 0)
    f8 = dict(dtype=torch.float64)
    s = (torch.randn(d.ncell, C, **f8) * 1.2 + 0.2).requires_grad_(True)
    mean, invstd = torch.randn(C, **f8) * 0.2, torch.rand(C, **f8) + 0.5
    gamma, beta = torch.rand(C, **f8) + 0.5, torch.randn(C, **f8) * 0.3
    gamma[3], beta[5] = 0.0, 0.0                                    # the algebra behind d fc1 / the sums must not divide by gamma
    w1 = (torch.randn(128, C, **f8) / 8).requires_grad_(True)
    b1 = torch.randn(128, **f8).requires_grad_(True)
    w2 = (torch.randn(DO, 128, **f8) / 11).requires_grad_(True)
    b2 = torch.randn(DO, **f8).requires_grad_(True)
    sh = (s - mean) * invstd
    a_full = _xf_ref(s, mean, invstd, gamma, beta, False)
    a = a_full.view(B, d.Tp, d.Hp, d.Wp, C)[:, :T, :H, :W].reshape(-1, C)
    out = torch.nn.functional.gelu(a @ w1.t() + b1) @ w2.t() + b2
    gout = torch.randn_like(out)
    a_full.retain_grad()
    out.backward(gout)
    g_ref = a_full.grad                                   # gradient w.r.t. BN(s): zero in the pad margin
    xf = (dev(mean), dev(invstd), dev(gamma), dev(beta), False)
    g = torch.full((d.ncell, C), float("nan"), device="cuda")
    slots, row = ops.head_bwd_slots(d), ops.head_bwd_row(DO)
    part = torch.full((slots, row), float("nan"), device="cuda")
    W1 = dev(w1.detach())
    ops.head_bwd(dev(s.detach()), W1, dev(b1.detach()), dev(w2.detach()), dev(gout), g, part, d, DO, xf)
    assert rel_l2(g.cpu(), g_ref) < 3e-6
    assert torch.equal(g.view(B, d.Tp, d.Hp, d.Wp, C)[:, :, :, W:], torch.zeros(B, d.Tp, d.Hp, pad, C, device="cuda"))
    tot = torch.empty(row, device="cuda")
    ops.reduce_partials(part, slots, row, out_f32=tot)
    dw1, dw2 = torch.empty(128, C, device="cuda"), torch.empty(DO, 128, device="cuda")
    db1, db2, sums = torch.empty(128, device="cuda"), torch.empty(DO, device="cuda"), torch.empty(2, C, device="cuda")
    ops.head_bwd_finalize(tot, W1, xf[2], xf[3], DO, dw1, dw2, db1, db2, sums)
    assert rel_l2(dw1.cpu(), w1.grad) < 5e-6 and rel_l2(dw2.cpu(), w2.grad) < 5e-6
    assert rel_l2(db1.cpu(), b1.grad) < 5e-6 and rel_l2(db2.cpu(), b2.grad) < 5e-6
    assert rel_l2(sums[0].cpu(), g_ref.sum(0)) < 2e-5 and rel_l2(sums[1].cpu(), (g_ref * sh.detach()).sum(0)) < 2e-5
    # ---- rpb_head_fwd_bwd: the same pass with the forward, the squared-error loss and dLoss/dout = gscale (out - y) formed inside
    y = out.detach() - gout / 0.37                          # chosen so that 0.37 (out - y) == gout: every gradient above must come out again
    g2 = torch.full((d.ncell, C), float("nan"), device="cuda")
    part2 = torch.full((slots, row), float("nan"), device="cuda")
    lpart = torch.full((slots,), float("nan"), device="cuda")
    ops.head_fwd_bwd(dev(s.detach()), W1, dev(b1.detach()), dev(w2.detach()), dev(b2.detach()), dev(y), 0.37, g2, part2, lpart, d, DO, xf)
    assert rel_l2(g2.cpu(), g_ref) < 2e-5                   # gout itself carries the forward's rounding now (out - y cancels ~3 digits)
    assert abs(float(lpart.double().sum()) - float(((out.detach() - y) ** 2).sum())) < 2e-5 * float(((out.detach() - y) ** 2).sum())
    tot2 = torch.empty(row, device="cuda")
    ops.reduce_partials(part2, slots, row, out_f32=tot2)
    ops.head_bwd_finalize(tot2, W1, xf[2], xf[3], DO, dw1, dw2, db1, db2, sums)
    assert rel_l2(dw1.cpu(), w1.grad) < 2e-5 and rel_l2(dw2.cpu(), w2.grad) < 2e-5
    assert rel_l2(db1.cpu(), b1.grad) < 2e-5 and rel_l2(db2.cpu(), b2.grad) < 2e-5


@pytest.mark.parametrize("feat_w,Wp,K2f", [(0, 70, 32), (0, 134, 32), (8, 134, 32), (0, 45, 24), (32, 40, 16)])
def test_eval_cell_mix_with_fused_forward_w_stage(ops, feat_w, Wp, K2f):
    """rpb_cell_mix_eval_dft == rpb_cell_mix (oxf) followed by rpb_axis_gemm along w of what it wrote (the rollout's fused stage):
    same `out` bit for bit, y1 to fp32-grade accuracy vs fp64; lines with a partial last tile and a half tile."""
    torch.manual_seed(Wp + K2f)
    G, C, K2 = 6, 64, 32
    ncell = G * Wp
    KC = feat_w or C
    x = torch.randn(ncell, KC, device="cuda")
    Wm = torch.randn(C, KC, device="cuda") / KC ** 0.5
    bias, z2, GW = torch.randn(C, device="cuda"), torch.randn(G * K2 * C, device="cuda"), torch.randn(K2, Wp, device="cuda") / 5
    oxf = (torch.randn(C, device="cuda"), torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda"), torch.randn(C, device="cuda"), True)
    FWt = torch.randn(Wp, K2f, device="cuda") / Wp ** 0.5
    assert ops.cell_mix_eval_dft_supported(ncell, K2, Wp, K2f)
    out, y1 = torch.empty(ncell, C, device="cuda"), torch.full((G, K2f, C), float("nan"), device="cuda")
    ops.cell_mix_eval_dft(x, Wm, bias, z2, GW, out, ncell, K2, Wp, oxf, FWt, K2f, y1, feat_w=feat_w)
    ref = torch.empty(ncell, C, device="cuda")
    if feat_w:
        ops.cell_mix_feat(x, Wm, bias, z2, GW, ref, None, ncell, feat_w, K2, Wp, oxf=oxf)
    else:
        ops.cell_mix(x, Wm, bias, z2, GW, ref, None, ncell, C, C, K2, Wp, oxf=oxf)
    assert torch.equal(out, ref)
    y_ref = torch.einsum("wk,gwc->gkc", FWt.double().cpu(), ref.double().cpu().view(G, Wp, C))
    assert rel_l2(y1.cpu(), y_ref) < 5e-6


@pytest.mark.parametrize("rows,L,stride", [(2048, 128, 128), (300, 4160, 4200), (255, 64, 64), (4096, 3, 8), (5, 100000, 100000)])
def test_reduce_partials_shapes(ops, rows, L, stride):
    """rpb_reduce_partials in both block shapes (tall: >= 256 rows of <= 16 K columns; wide otherwise): fp64 sums, scale, accumulate."""
    torch.manual_seed(rows + L)
    part = torch.randn(rows, stride, dtype=torch.float64)
    ref = part[:, :L].sum(0)
    o32 = torch.empty(L, device="cuda")
    o64 = torch.empty(L, device="cuda", dtype=torch.float64)
    p32 = dev(part)
    ops.reduce_partials(p32, rows, L, out_f32=o32, out_f64=o64, row_stride=stride)
    ref32 = p32.double().cpu()[:, :L].sum(0)
    assert rel_l2(o64.cpu(), ref32) < 1e-14 and rel_l2(o32.cpu(), ref32) < 2e-7
    ops.reduce_partials(p32, rows, L, out_f32=o32, row_stride=stride, scale=0.5, accumulate=True)
    assert rel_l2(o32.cpu(), 1.5 * ref32) < 3e-7


def test_grouped_partial_reductions_equal_the_single_launches(ops):
    """ops.deferred_reductions: reductions of different shapes / strides / column offsets marked ``deferrable`` inside the block run as ONE
    rpb_reduce_partials_grouped launch and give bit-identical results to the per-item launches (same fp64 accumulation order); a
    reduction that does not opt in runs at once inside the block (its output may be read there)."""
    torch.manual_seed(5)
    shapes = [(7, 100, 100, 0), (300, 64, 64, 0), (12, 5000, 5064, 0), (33, 64, 5064, 5000), (1, 3, 3, 0), (64, 1024 * 33, 1024 * 33, 0)]
    parts = [torch.randn(rows, stride, device="cuda") for rows, L, stride, c0 in shapes]
    ref = []
    for p, (rows, L, stride, c0) in zip(parts, shapes):
        o = torch.empty(L, device="cuda")
        ops.reduce_partials(p, rows, L, out_f32=o, row_stride=stride, col0=c0)
        ref.append(o)
    outs = [torch.full((L,), float("nan"), device="cuda") for rows, L, stride, c0 in shapes]
    with ops.deferred_reductions():
        for p, o, (rows, L, stride, c0) in zip(parts, outs, shapes):
            ops.reduce_partials(p, rows, L, out_f32=o, row_stride=stride, col0=c0, deferrable=True)
        assert all(bool(torch.isnan(o).all()) for o in outs)          # nothing ran yet
        now = torch.full((100,), float("nan"), device="cuda")
        ops.reduce_partials(parts[0], 7, 100, out_f32=now)            # not deferrable: runs inside the block
        assert torch.equal(now, ref[0])
    torch.cuda.synchronize()
    for o, r in zip(outs, ref):
        assert torch.equal(o, r)
    assert ops.deferred_reductions.current() is None
    with ops.deferred_reductions():                                   # a second block reuses the pinned / device tables
        ops.reduce_partials(parts[1], 300, 64, out_f32=outs[1].fill_(float("nan")), deferrable=True)
    assert torch.equal(outs[1], ref[1])
    # an exception inside the block drops the queue and restores immediate mode
    with pytest.raises(RuntimeError):
        with ops.deferred_reductions():
            ops.reduce_partials(parts[0], 7, 100, out_f32=outs[0], deferrable=True)
            raise RuntimeError("boom")
    assert ops.deferred_reductions.current() is None


@pytest.mark.parametrize("B,T,H,W,Cin,pad", [(2, 3, 5, 7, 2, 6), (1, 2, 4, 40, 2, 3), (3, 4, 6, 33, 2, 6), (2, 3, 5, 9, 3, 6)])
def test_lift_feat_feature_fields(ops, B, T, H, W, Cin, pad):
    """rpb_lift_feat: Phi_c[cell] = (x_0 .. x_{Cin-1}, grid_t, grid_h, grid_w, 1, 0 ..) on the data cells, zeros in the pad margin
    (the input of layer 0's channel mixing, fno.py:106-111 with the lift folded into the layer): the row-walking Cin = 2 instance and the
    generic kernel (Cin = 3), bit for bit against the construction in torch."""
    torch.manual_seed(B + W)
    d = ops.Dims(B, T, H, W, Cin, 64, pad)
    x = torch.randn(B, T, H, W, Cin, device="cuda")
    grids = [torch.linspace(0, 1, n, device="cuda") for n in (T, H, W)]
    FW = 8
    out = torch.full((d.ncell, FW), float("nan"), device="cuda")
    ops.lift_feat(x, grids, out, d, FW)
    ref = torch.zeros(B, d.Tp, d.Hp, d.Wp, FW, device="cuda")
    ref[:, :T, :H, :W, :Cin] = x
    ref[:, :T, :H, :W, Cin] = grids[0].view(1, T, 1, 1)
    ref[:, :T, :H, :W, Cin + 1] = grids[1].view(1, 1, H, 1)
    ref[:, :T, :H, :W, Cin + 2] = grids[2].view(1, 1, 1, W)
    ref[:, :T, :H, :W, Cin + 3] = 1.0
    assert torch.equal(out.view_as(ref), ref)


@pytest.mark.parametrize("B,T,H,W,pad,bn,DO", [(2, 3, 6, 40, 2, True, 2), (1, 2, 5, 70, 6, True, 3), (3, 2, 4, 32, 6, False, 4), (2, 2, 3, 5, 2, True, 1)])
def test_proj_bwd_width_128_on_the_matrix_pipe(ops, B, T, H, W, pad, bn, DO, monkeypatch):
    """rpb_proj_bwd at C = 128 (configs/fsi/fno.yaml) runs csrc/rpb_pjh.hip's backward instance: gu = (fc2^T gout) gelu'(fc1 a + b1) and the
    partial rows of d fc2.weight / d fc1.bias / d fc2.bias, against fp64 and against the fp32-pipe kernel it replaces (RPB_HEAD_PJH_128_BWD=0)."""
    torch.manual_seed(B * 100 + W + DO)
    C = 128
    d = ops.Dims(B, T, H, W, 2, C, pad)
    a = torch.randn(B, d.Tp, d.Hp, d.Wp, C, dtype=torch.float64) * 1.5 + 0.3
    w1 = torch.randn(128, C, dtype=torch.float64) / math.sqrt(C)
    b1 = torch.randn(128, dtype=torch.float64)
    w2 = torch.randn(DO, 128, dtype=torch.float64) / 11
    b2 = torch.randn(DO, dtype=torch.float64)
    mean, var = torch.randn(C, dtype=torch.float64) * 0.2 + 0.3, torch.rand(C, dtype=torch.float64) + 0.5
    gamma, beta = torch.rand(C, dtype=torch.float64) + 0.5, torch.randn(C, dtype=torch.float64) * 0.1
    invstd = (var + 1e-5).rsqrt()
    ac = a[:, :T, :H, :W].reshape(-1, C)
    if bn:
        ac = (ac - mean) * invstd * gamma + beta
    gout = torch.randn(d.ncrop, DO, dtype=torch.float64)
    u = (ac @ w1.t() + b1).requires_grad_(True)
    act = torch.nn.functional.gelu(u)
    (act @ w2.t() * gout).sum().backward()
    gu_ref = u.grad
    dw2_ref, db1_ref, db2_ref = gout.t() @ act.detach(), gu_ref.sum(0), gout.sum(0)
    xf = (dev(mean), dev(invstd), dev(gamma), dev(beta), 0) if bn else None
    slots = ops.proj_slots(d.ncrop, C, DO)
    row = DO * 128 + 128 + DO
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("RPB_HEAD_PJH_128_BWD", flag)
        gu = torch.full((d.ncrop, 128), float("nan"), device="cuda")
        part = torch.full((slots, row), float("nan"), device="cuda")
        ops.proj_bwd(dev(a).view(-1, C), dev(w1), dev(b1), dev(w2), dev(b2), dev(gout), gu, part, d, DO, xf=xf)
        tot = part.double().sum(0).cpu()
        res[flag] = (gu.cpu(), tot)
        assert rel_l2(gu.cpu(), gu_ref) < 2e-5
        assert rel_l2(tot[:DO * 128].view(DO, 128), dw2_ref) < 2e-5
        assert rel_l2(tot[DO * 128:DO * 128 + 128], db1_ref) < 2e-5 and rel_l2(tot[DO * 128 + 128:], db2_ref) < 2e-5
    assert rel_l2(res["1"][0], res["0"][0]) < 2e-5


@pytest.mark.parametrize("B,T,H,W,pad", [(2, 3, 6, 40, 2), (1, 2, 5, 64, 6), (3, 2, 4, 32, 6), (2, 2, 3, 5, 2), (1, 2, 3, 70, 6)])
def test_fc1_data_gradient_width_128_on_the_matrix_pipe(ops, B, T, H, W, pad, monkeypatch):
    """rpb_cell_mix(gather) at 128 -> 128 channels without statistics (the fc1 data gradient of configs/fsi/fno.yaml) runs csrc/rpb_pjh.hip's
    MODE 2: g[padded cell] = gu[cropped cell] fc1.weight, exact zeros in the whole pad; against fp64 and the fp32-pipe kernel it replaces."""
    torch.manual_seed(B * 100 + W)
    C = 128
    d = ops.Dims(B, T, H, W, 2, C, pad)
    gu = torch.randn(d.ncrop, 128, dtype=torch.float64)
    w1 = torch.randn(128, C, dtype=torch.float64) / math.sqrt(C)
    ref = torch.zeros(B, d.Tp, d.Hp, d.Wp, C, dtype=torch.float64)
    ref[:, :T, :H, :W] = (gu @ w1).view(B, T, H, W, C)
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("RPB_GATHER_128_PJH", flag)
        g = torch.full((d.ncell, C), float("nan"), device="cuda")
        ops.cell_mix(dev(gu), dev(w1), None, None, None, g, None, d.ncell, 128, C, 0, 1, transpose_w=True, gather=True, crop6=d.crop6)
        outs[flag] = g.cpu().view(B, d.Tp, d.Hp, d.Wp, C)
        assert rel_l2(outs[flag], ref) < TOL
        assert float(outs[flag][:, T:].abs().max() if pad else 0) == 0 and float(outs[flag][:, :, H:].abs().max()) == 0
        assert float(outs[flag][:, :, :, W:].abs().max()) == 0
    assert rel_l2(outs["1"], outs["0"]) < 2e-6
    # ... and with the BatchNorm-backward sums of the last layer in the same launch (MODE 3): sum g, sum g * shat per channel
    monkeypatch.setenv("RPB_GATHER_128_PJH", "1")
    sl = torch.randn(d.ncell, C, dtype=torch.float64) * 1.3 + 0.2
    mean, invstd = torch.randn(C, dtype=torch.float64) * 0.2, torch.rand(C, dtype=torch.float64) + 0.5
    rows = ops.cell_mix_stat_rows(d.ncell, 128, C, 0, 1, False, True)
    part = torch.full((rows, 2 * C), float("nan"), device="cuda")
    g = torch.full((d.ncell, C), float("nan"), device="cuda")
    ops.cell_mix(dev(gu), dev(w1), None, None, None, g, part, d.ncell, 128, C, 0, 1, transpose_w=True, gather=True, crop6=d.crop6,
                 bnb=(dev(sl), dev(mean), dev(invstd), dev(torch.ones(C)), dev(torch.zeros(C)), False))
    assert torch.equal(g.cpu().view(B, d.Tp, d.Hp, d.Wp, C), outs["1"])
    tot = part.double().sum(0).cpu()
    gref = ref.view(-1, C)
    assert rel_l2(tot[:C], gref.sum(0)) < 2e-5
    assert rel_l2(tot[C:], (gref * ((sl - mean) * invstd)).sum(0)) < 2e-5


@pytest.mark.parametrize("B,T,H,W,pad,DO", [(2, 3, 6, 40, 2, 2), (1, 2, 5, 70, 6, 3), (2, 2, 3, 5, 2, 4)])
def test_width_128_head_with_silu(ops, B, T, H, W, pad, DO):
    """The Galerkin SpectralRegressor's head (SiLU, 128 channels; galerkin_transformer_libs/model.py:631-632) on csrc/rpb_pjh.hip's width-128
    instances: forward and rpb_proj_bwd's outputs against fp64."""
    torch.manual_seed(B * 10 + W + DO)
    C = 128
    d = ops.Dims(B, T, H, W, 2, C, pad)
    a = torch.randn(B, d.Tp, d.Hp, d.Wp, C, dtype=torch.float64) * 1.5 + 0.3
    w1 = torch.randn(128, C, dtype=torch.float64) / math.sqrt(C)
    b1 = torch.randn(128, dtype=torch.float64)
    w2 = torch.randn(DO, 128, dtype=torch.float64) / 11
    b2 = torch.randn(DO, dtype=torch.float64)
    ac = a[:, :T, :H, :W].reshape(-1, C)
    u = (ac @ w1.t() + b1).requires_grad_(True)
    act = torch.nn.functional.silu(u)
    ref = act @ w2.t() + b2
    gout = torch.randn(d.ncrop, DO, dtype=torch.float64)
    (ref * gout).sum().backward()
    out = torch.full((d.ncrop, DO), float("nan"), device="cuda")
    ops.proj_fwd(dev(a).view(-1, C), dev(w1), dev(b1), dev(w2), dev(b2), out, d, DO, act=1)
    assert rel_l2(out.cpu(), ref.detach()) < TOL
    slots, row = ops.proj_slots(d.ncrop, C, DO), DO * 128 + 128 + DO
    gu = torch.full((d.ncrop, 128), float("nan"), device="cuda")
    part = torch.full((slots, row), float("nan"), device="cuda")
    ops.proj_bwd(dev(a).view(-1, C), dev(w1), dev(b1), dev(w2), dev(b2), dev(gout), gu, part, d, DO, act=1)
    tot = part.double().sum(0).cpu()
    assert rel_l2(gu.cpu(), u.grad) < 2e-5
    assert rel_l2(tot[:DO * 128].view(DO, 128), gout.t() @ act.detach()) < 2e-5
    assert rel_l2(tot[DO * 128:DO * 128 + 128], u.grad.sum(0)) < 2e-5 and rel_l2(tot[DO * 128 + 128:], gout.sum(0)) < 2e-5
