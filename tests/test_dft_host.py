"""Host logic: truncated-DFT stage matrices + weight-layout conversion reproduce SpectralConv3d (CPU, fp32)."""
import torch

from conftest import rel_l2
from oracle import fno3d_oracle as O
from realpdebench_amd.dft import SpectralPlan, mode_major_to_ref_weights, ref_weights_to_mode_major


def staged_spectral_conv(x_cl, plan, wm):
    """Pure-torch emulation of the kernel pipeline on channels-last data; mirrors the HIP data layouts."""
    B, Tp, Hp, Wp, C = x_cl.shape
    m3, KH, KT = plan.KW, plan.KH, plan.KT
    y1 = torch.einsum("ow,gwc->goc", plan.FW, x_cl.reshape(B * Tp * Hp, Wp, C))            # [G,(ri,kw),C]
    y2 = torch.einsum("ok,gkn->gon", plan.FH, y1.reshape(B * Tp, Hp * 2, m3 * C))           # [(b,t),(ri,kh),(kw,c)]
    xh = torch.einsum("ok,gkn->gon", plan.FT, y2.reshape(B, Tp * 2, KH * m3 * C))           # [b,(ri,kt),(kh,kw,c)]
    xh = xh.reshape(B, 2, plan.M, C)
    xc = torch.complex(xh[:, 0], xh[:, 1])                                                  # [B,M,C]
    wc = torch.view_as_complex(wm.contiguous())                                             # [M,Ci,Co]
    yc = torch.einsum("bmi,mio->bmo", xc, wc)
    yh = torch.stack([yc.real, yc.imag], dim=1).reshape(B, 2 * KT, KH * m3 * C)
    z1 = torch.einsum("ok,gkn->gon", plan.GT, yh)                                           # [b,(t,ri),...]
    z2 = torch.einsum("ok,gkn->gon", plan.GH, z1.reshape(B * Tp, 2 * KH, m3 * C))           # [(b,t),(h,ri),(kw,c)]
    out = torch.einsum("wk,gkc->gwc", plan.GW, z2.reshape(B * Tp * Hp, 2 * m3, C))
    return out.reshape(B, Tp, Hp, Wp, C)


def test_staged_dft_matches_oracle_spectral_conv():
    torch.manual_seed(0)
    B, C, Tp, Hp, Wp = 2, 4, 10, 14, 18
    modes = (2, 3, 4)
    x = torch.randn(B, C, Tp, Hp, Wp)
    ws = [torch.randn(C, C, *modes, dtype=torch.cfloat) for _ in range(4)]
    ref = O.spectral_conv3d(x, ws, modes)
    plan = SpectralPlan(Tp, Hp, Wp, modes)
    wm = ref_weights_to_mode_major(*ws)
    got = staged_spectral_conv(x.permute(0, 2, 3, 4, 1).contiguous(), plan, wm).permute(0, 4, 1, 2, 3)
    assert rel_l2(got, ref) < 2e-6


def test_adjoint_is_transpose():
    """The backward pass uses the transposed matrices: check against autograd through the oracle."""
    torch.manual_seed(1)
    B, C, Tp, Hp, Wp = 1, 2, 8, 10, 12
    modes = (2, 2, 3)
    x = torch.randn(B, C, Tp, Hp, Wp, requires_grad=True)
    ws = [torch.randn(C, C, *modes, dtype=torch.cfloat, requires_grad=True) for _ in range(4)]
    y = O.spectral_conv3d(x, ws, modes)
    g = torch.randn_like(y)
    y.backward(g)
    plan = SpectralPlan(Tp, Hp, Wp, modes)
    wm = ref_weights_to_mode_major(*[w.detach() for w in ws])
    xcl = x.detach().permute(0, 2, 3, 4, 1).contiguous().requires_grad_(True)
    wm = wm.requires_grad_(True)
    out = staged_spectral_conv(xcl, plan, wm)
    out.backward(g.permute(0, 2, 3, 4, 1))
    assert rel_l2(xcl.grad.permute(0, 4, 1, 2, 3), x.grad) < 2e-6
    gws = mode_major_to_ref_weights(wm.grad, modes)
    for a, b in zip(gws, ws):
        # torch stores the conjugate-free gradient for complex leaves: d/d(re) + i d/d(im)
        assert rel_l2(a, b.grad) < 2e-6


def test_weight_layout_roundtrip():
    modes = (2, 3, 4)
    ws = [torch.randn(5, 6, *modes, dtype=torch.cfloat) for _ in range(4)]
    back = mode_major_to_ref_weights(ref_weights_to_mode_major(*ws), modes)
    for a, b in zip(ws, back):
        assert torch.equal(a, b)


def test_f16x2_spec_exp_puts_the_inverse_stage_matrix_in_fp16_range():
    """ops.spec_exp (the power of two the opt-in f16x2 eval arithmetic moves from the last inverse-stage matrix onto the z2 rows): for the
    shapes of the reference's FNO YAMLs and a tiny grid, max |GW| 2^e lies in (0.5, 1] and no entry that matters turns subnormal in fp16;
    set_arith validates its argument without a GPU."""
    import torch
    from realpdebench_amd import ops
    from realpdebench_amd.dft import SpectralPlan
    for (T, H, W), modes in (((20, 128, 128), (4, 12, 16)), ((20, 64, 128), (4, 12, 16)), ((20, 64, 64), (4, 16, 16)), ((64, 64, 64), (4, 16, 16)),
                              ((3, 9, 40), (2, 4, 8))):
        d = ops.Dims(2, T, H, W, 2, 64, 6)
        plan = SpectralPlan(d.Tp, d.Hp, d.Wp, modes)
        e = ops.spec_exp(d)
        m = float(plan.GWt.abs().max()) * 2.0 ** e
        assert 0.5 < m <= 1.0 + 1e-6, (T, H, W, e, m)
        big = plan.GWt.abs() * 2.0 ** e
        assert float(big[big > 2.0 ** -10].min()) > 6.2e-5          # fp16's smallest normal: entries above 1e-3 of the maximum stay normal
    from realpdebench_amd.model.fno import FNO3d
    mdl = FNO3d(2, 4, 8, 2, 64, (3, 9, 40, 2), (3, 9, 40, 2))
    assert mdl.set_arith("f16x2").arith == "f16x2" and mdl.set_arith("f32").arith == "f32"
    import pytest
    with pytest.raises(ValueError):
        mdl.set_arith("fp8")
    with pytest.raises(NotImplementedError):
        FNO3d(2, 4, 8, 2, 128, (3, 9, 40, 2), (3, 9, 40, 2)).set_arith("f16x2")
