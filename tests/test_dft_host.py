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
