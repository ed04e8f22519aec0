"""Split-bf16 3x3x3 convolution kernels (csrc/rpb_conv3x.hip) through the C ABI vs fp64 PyTorch: the bf16 MFMA path must be
fp32-grade (every operand is hi + mid + lo = its full 24-bit significand, six of the nine cross products are kept), so the
tolerances are the same 3e-6 the exact-fp32 implicit GEMM is held to (nn.Conv3d(Ci, Co, 3, padding=1) of
Physics_Attention.py:154-157 / unet.py:196,201 and its autograd)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from realpdebench_amd import ops as o
    return o


def _bf16_to_f64(t):
    return (t.to(torch.int32) << 16).view(torch.float32).double()


def test_split3_reconstructs_every_significand_bit(ops):
    torch.manual_seed(0)
    M, C = 1000, 64
    x = (torch.randn(M, C, device="cuda") * torch.logspace(-6, 6, C, device="cuda")).contiguous()
    planes = torch.empty(3 * M * C, dtype=torch.int16, device="cuda")
    ops.split3(x, planes, M, C)
    p = planes.view(3, M, C)
    rec = _bf16_to_f64(p[0]) + _bf16_to_f64(p[1]) + _bf16_to_f64(p[2])
    err = ((rec - x.double()).abs() / x.double().abs().clamp_min(1e-30)).max()
    assert float(err) < 2.0 ** -23                       # the three bf16 terms carry all 24 significand bits
    # token-run planes: same numbers in [M/8][C][8] order, also for the reversed mesh order
    pt = torch.empty_like(planes)
    ops.split3t(x, pt, M, C)
    assert torch.equal(pt.view(3, M // 8, C, 8).permute(0, 1, 3, 2).reshape(3, M, C), p)
    d0, d1, d2 = 10, 5, 20
    ops.split3t(x, pt, M, C, rev_mesh=(d0, d1, d2))
    xr = x.view(1, d0, d1, d2, C).permute(0, 3, 2, 1, 4).reshape(M, C).contiguous()
    ops.split3(xr, planes, M, C)
    assert torch.equal(pt.view(3, M // 8, C, 8).permute(0, 1, 3, 2).reshape(3, M, C), planes.view(3, M, C))


@pytest.mark.parametrize("B,mesh,Ci,Co", [(2, (3, 5, 7), 64, 64), (1, (4, 6, 40), 128, 128), (1, (2, 9, 33), 64, 256),
                                          (1, (5, 4, 13), 192, 512), (3, (2, 2, 2), 64, 64)])
def test_conv3x_forward_vs_fp64(ops, B, mesh, Ci, Co):
    T, H, W = mesh
    M = B * T * H * W
    torch.manual_seed(Ci + Co)
    x = torch.randn(M, Ci, device="cuda")
    w = torch.randn(Co, 27 * Ci, device="cuda") / (27 * Ci) ** 0.5
    bias = torch.randn(Co, device="cuda")
    y = torch.full((M, Co), float("nan"), device="cuda")
    assert ops.conv3_split_ok(Co, Ci)
    ops.conv3(x, w, y, M, Co, Ci, mesh, bias=bias)
    xr = x.view(B, T, H, W, Ci).permute(0, 4, 1, 2, 3).double().cpu()
    wr = w.view(Co, 3, 3, 3, Ci).permute(0, 4, 1, 2, 3).double().cpu()
    ref = F.conv3d(xr, wr, bias.double().cpu(), padding=1).permute(0, 2, 3, 4, 1).reshape(M, Co)
    assert rel_l2(y.cpu(), ref) < 1e-6


@pytest.mark.parametrize("B,mesh,Ci,Co", [(2, (3, 5, 16), 64, 64), (1, (4, 6, 40), 128, 64), (1, (2, 9, 24), 64, 128),
                                          (2, (16, 6, 5), 64, 64), (1, (3, 3, 32), 64, 192), (2, (2, 3, 64), 64, 64),
                                          (1, (2, 2, 128), 128, 64)])
def test_conv3x_weight_gradient_vs_fp64(ops, B, mesh, Ci, Co):
    """Also the mesh whose innermost dimension is not a multiple of 8 (reversed token order, Transolver's 128 x 64 x 20 case) and
    innermost sizes 32 / 64 / 128, which take the kernel that keeps the X rows in an LDS ring."""
    T, H, W = mesh
    M = B * T * H * W
    K = 27 * Ci
    torch.manual_seed(Ci * 3 + Co)
    x, g = torch.randn(M, Ci, device="cuda"), torch.randn(M, Co, device="cuda")
    assert ops.conv3_wgrad_split_mode(Co, Ci, mesh, M) == (1 if W % 8 == 0 else 2)
    part, rev = ops.conv3_wgrad_parts(g, x, M, Co, Ci, mesh)
    dW, db = torch.empty(Co, K, device="cuda"), torch.empty(Co, device="cuda")
    ops.reduce_partials(part, part.shape[0], Co * K, out_f32=dW.view(-1), row_stride=Co * K + Co)
    ops.reduce_partials(part, part.shape[0], Co, out_f32=db, row_stride=Co * K + Co, col0=Co * K)
    if rev:
        dW = ops.conv3_taps_restore(dW, Co, Ci)
    xr = x.view(B, T, H, W, Ci).permute(0, 4, 1, 2, 3).double().cpu()
    gr = g.view(B, T, H, W, Co).permute(0, 4, 1, 2, 3).double().cpu()
    wr = torch.zeros(Co, Ci, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    br = torch.zeros(Co, dtype=torch.float64, requires_grad=True)
    F.conv3d(xr, wr, br, padding=1).backward(gr)
    assert rel_l2(dW.cpu(), wr.grad.permute(0, 2, 3, 4, 1).reshape(Co, K)) < 1e-6
    assert rel_l2(db.cpu(), br.grad) < 1e-6


def test_exact_fp32_switch_takes_the_fp32_mfma_path(ops, monkeypatch):
    """RPB_CONV3_EXACT=1 (ops.CONV3_SPLIT False): the same entry points run the exact-fp32 implicit GEMM."""
    monkeypatch.setattr(ops, "CONV3_SPLIT", False)
    assert not ops.conv3_split_ok(64, 64) and ops.conv3_wgrad_split_mode(64, 64, (4, 8, 16), 512) == 0
    B, (T, H, W), C = 1, (2, 4, 8), 64
    M = B * T * H * W
    x, w = torch.randn(M, C, device="cuda"), torch.randn(C, 27 * C, device="cuda") / 40
    y = torch.empty(M, C, device="cuda")
    ops.conv3(x, w, y, M, C, C, (T, H, W))
    ref = F.conv3d(x.view(B, T, H, W, C).permute(0, 4, 1, 2, 3).double().cpu(),
                   w.view(C, 3, 3, 3, C).permute(0, 4, 1, 2, 3).double().cpu(), padding=1).permute(0, 2, 3, 4, 1).reshape(M, C)
    assert rel_l2(y.cpu(), ref) < 3e-6


@pytest.mark.parametrize("M,N,K", [(300, 64, 64), (1000, 128, 192), (777, 256, 256), (130, 768, 64), (4100, 256, 1024), (515, 384, 128)])
def test_gemm3x_matches_fp64_and_the_fp32_kernel_epilogues(ops, monkeypatch, M, N, K):
    """The split-bf16 dense GEMM (csrc/rpb_gemm3x.hip) behind ops.gemm_nt: every epilogue of rpb_gemm_nt, K-split variants
    (N = 64, 128), ragged M; in-kernel dropout must drop the same elements as the exact-fp32 kernel (same Philox counters)."""
    monkeypatch.setattr(ops, "GEMM_SPLIT_MIN_ROWS", 0)
    monkeypatch.setattr(ops, "GEMM_SPLIT_MIN_K", 64)
    monkeypatch.setattr(ops, "GEMM_SPLIT_MIN_N", 64)
    torch.manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") / K ** 0.5
    bias, addv = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
    res, aux = torch.randn(M, N, device="cuda"), torch.randn(M, N, device="cuda")
    ref = A.double().cpu() @ W.double().cpu().t()
    assert ops.gemm_split_ok(M, N, K, K, N, None, v2=N % 128 == 0)
    out, pre = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    ops.gemm_nt(A, W, out, M, N, K)
    assert rel_l2(out.cpu(), ref) < 1e-6
    ops.gemm_nt(A, W, out, M, N, K, bias=bias, act=1, pre_out=pre)                      # bias + GELU, pre-activation saved
    z = ref + bias.double().cpu()
    assert rel_l2(pre.cpu(), z) < 1e-6 and rel_l2(out.cpu(), F.gelu(z)) < 2e-6
    ops.gemm_nt(A, W, out, M, N, K, act=2, aux=aux, residual=res)                       # * gelu'(aux) + residual
    a64 = aux.double().cpu().requires_grad_(True)
    F.gelu(a64).sum().backward()
    assert rel_l2(out.cpu(), ref * a64.grad + res.double().cpu()) < 2e-6
    ops.gemm_nt(A, W, out, M, N, K, bias=bias, act=3, addvec=addv)                      # ReLU + broadcast vector
    assert rel_l2(out.cpu(), torch.relu(z) + addv.double().cpu()) < 1e-6
    ops.gemm_nt(A, W, out, M, N, K, act=4, aux=aux)                                     # ReLU'
    assert rel_l2(out.cpu(), torch.where(aux.double().cpu() > 0, ref, torch.zeros_like(ref))) < 1e-6
    # in-kernel dropout: identical keep pattern and scale as the exact-fp32 kernel for the same (seed, keep)
    ops.gemm_nt(A, W, out, M, N, K, bias=bias, drop=(1234, 0.7), residual=res)
    monkeypatch.setattr(ops, "GEMM_SPLIT", False)
    out32 = torch.empty_like(out)
    ops.gemm_nt(A, W, out32, M, N, K, bias=bias, drop=(1234, 0.7), residual=res)
    assert torch.equal((out - res) == 0, (out32 - res) == 0) or rel_l2(out.cpu(), out32.cpu()) < 2e-6
    assert rel_l2(out.cpu(), out32.cpu()) < 2e-6


@pytest.mark.parametrize("M,N,K,ldg,lda", [(4096, 1024, 1280, None, None), (4099, 256, 256, None, None), (65536, 256, 256, None, None), (70001, 512, 256, None, None), (65536 + 77, 256, 768, 800, 1024),
                                           (131072, 768, 256, 768, 300)])
def test_gemm3x_tn_weight_gradient_vs_fp64(ops, monkeypatch, M, N, K, ldg, lda):
    """The split-bf16 weight-gradient GEMM (csrc/rpb_gemm3x_tn.hip) behind ops.gemm_tn: dW = G^T A and db = colsum G from row-major token
    tensors (ragged M, leading dimensions larger than the widths, sub-block operands), fp32-grade against fp64; the exact-fp32 kernel
    (RPB_GEMM_TN_F32=1 at library load, or unsupported shapes) agrees to the same tolerance."""
    torch.manual_seed(M + N + K)
    ldg_, lda_ = ldg or N, lda or K
    Gf, Af = torch.randn(M, ldg_, device="cuda"), torch.randn(M, lda_, device="cuda")
    assert ops.gemm_tn_split_bf16(M, N, K, ldg, lda)
    splits = ops.gemm_tn_splits(M, N, K, ldg=ldg, lda=lda)
    assert splits % 8 == 0
    part = torch.full((splits, N * K + N), float("nan"), device="cuda")
    g0, a0 = (ldg_ - N) // 2, lda_ - K                      # operand blocks that do not start at column 0
    ops.gemm_tn(ops.Sub(Gf, g0) if g0 else Gf, ops.Sub(Af, a0) if a0 else Af, part, M, N, K, ldg=ldg, lda=lda)
    tot = part.double().sum(0).cpu()
    G64, A64 = Gf[:, g0:g0 + N].double().cpu(), Af[:, a0:a0 + K].double().cpu()
    assert rel_l2(tot[:N * K].view(N, K), G64.t() @ A64) < 3e-6
    assert rel_l2(tot[N * K:], G64.sum(0)) < 3e-6
    # a shape the split kernel does not take (N % 256) goes to the fp32 kernel through the same entry point
    assert not ops.gemm_tn_split_bf16(M, 128, K, ldg, lda)


def test_gemm3x_tn_one_workgroup_per_cu_variant():
    """RPB_GEMM3X_TN_TILE=256 (256 x 256 outputs per workgroup, one workgroup per CU; the default is 128 x 256, two per CU) is read when the
    library loads: checked in a child process against fp64."""
    import os
    import subprocess
    import sys
    code = """
import torch
from realpdebench_amd import ops
torch.manual_seed(0)
M, N, K = 70001, 512, 256
G, A = torch.randn(M, N, device="cuda"), torch.randn(M, K, device="cuda")
sp = ops.gemm_tn_splits(M, N, K)
part = torch.full((sp, N * K + N), float("nan"), device="cuda")
ops.gemm_tn(G, A, part, M, N, K)
tot = part.double().sum(0).cpu()
ref = G.double().cpu().t() @ A.double().cpu()
e1 = float((tot[:N * K].view(N, K) - ref).norm() / ref.norm())
e2 = float((tot[N * K:] - G.double().cpu().sum(0)).norm() / G.double().cpu().sum(0).norm())
assert e1 < 3e-6 and e2 < 3e-6, (e1, e2)
print("ok", sp)
"""
    env = dict(os.environ, RPB_GEMM3X_TN_TILE="256")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok 128" in r.stdout, r.stdout + r.stderr
