"""Transolver forward on MI355X through the C ABI: GEMM / implicit-GEMM convolution / LayerNorm / slice kernels vs
fp64 PyTorch, and the whole model vs (a) the golden vector generated from the reference and (b) the CPU oracle at the
reference's width (n_hidden 256, 8 heads, 16 slices)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from realpdebench_amd import ops as o
    return o


def dev(t):
    return t.float().cuda().contiguous()


@pytest.mark.parametrize("M,N,K,act,res", [(300, 256, 512, 0, False), (1000, 1024, 256, 1, False), (257, 256, 1024, 0, True),
                                           (130, 3, 256, 0, False), (64, 64, 64, 1, True), (500, 40, 96, 0, False)])
def test_gemm_nt(ops, M, N, K, act, res):
    torch.manual_seed(M + N)
    A, W = torch.randn(M, K, dtype=torch.float64), torch.randn(N, K, dtype=torch.float64) / K ** 0.5
    b, av = torch.randn(N, dtype=torch.float64), torch.randn(N, dtype=torch.float64)
    R = torch.randn(M, N, dtype=torch.float64)
    ref = A @ W.t() + b
    if act:
        ref = torch.nn.functional.gelu(ref)
    ref = ref + av + (R if res else 0)
    out = dev(R) if res else torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_nt(dev(A), dev(W), out, M, N, K, bias=dev(b), addvec=dev(av), residual=out if res else None, act=act)
    assert rel_l2(out.cpu(), ref) < 3e-6


def test_conv3d_dual_as_implicit_gemm(ops):
    torch.manual_seed(4)
    B, H, W, D, C = 2, 5, 4, 6, 64
    x = torch.randn(B, H, W, D, C, dtype=torch.float64)
    w1, w2 = torch.randn(C, C, 3, 3, 3, dtype=torch.float64) / 40, torch.randn(C, C, 3, 3, 3, dtype=torch.float64) / 40
    b1, b2 = torch.randn(C, dtype=torch.float64), torch.randn(C, dtype=torch.float64)
    xc = x.permute(0, 4, 1, 2, 3)
    y1 = torch.nn.functional.conv3d(xc, w1, b1, padding=1).permute(0, 2, 3, 4, 1)
    y2 = torch.nn.functional.conv3d(xc, w2, b2, padding=1).permute(0, 2, 3, 4, 1)
    ref = torch.cat([y1, y2], dim=-1).reshape(-1, 2 * C)
    wcat = torch.cat([w1, w2], 0).permute(0, 2, 3, 4, 1).reshape(2 * C, 27 * C)
    M = B * H * W * D
    out = torch.empty(M, 2 * C, device="cuda")
    ops.gemm_nt(dev(x).view(M, C), dev(wcat), out, M, 2 * C, 27 * C, bias=dev(torch.cat([b1, b2])), conv=(H, W, D))
    assert rel_l2(out.cpu(), ref) < 3e-6


@pytest.mark.parametrize("C", [64, 256])
def test_layernorm_and_lift(ops, C):
    torch.manual_seed(C)
    M = 777
    x = torch.randn(M, C, dtype=torch.float64) * 2 + 0.5
    g, b = torch.rand(C, dtype=torch.float64) + 0.5, torch.randn(C, dtype=torch.float64)
    out = torch.empty(M, C, device="cuda")
    ops.layernorm_fwd(dev(x), dev(g), dev(b), out, M, C)
    assert rel_l2(out.cpu(), torch.nn.functional.layer_norm(x, (C,), g, b, 1e-5)) < 2e-6
    xi, W, bb = torch.randn(M, 3, dtype=torch.float64), torch.randn(2 * C, 3, dtype=torch.float64), torch.randn(2 * C, dtype=torch.float64)
    o2 = torch.empty(M, 2 * C, device="cuda")
    ops.tokens_lift(dev(xi), dev(W), dev(bb), o2, M, 3, 2 * C, True)
    assert rel_l2(o2.cpu(), torch.nn.functional.gelu(xi @ W.t() + bb)) < 2e-6


@pytest.mark.parametrize("heads,G,ntok,B", [(8, 16, 300, 2), (2, 16, 70, 3), (4, 8, 129, 1), (8, 32, 200, 1)])
def test_slice_attention_deslice(ops, heads, G, ntok, B):
    torch.manual_seed(heads + G)
    C = heads * 32
    M = B * ntok
    xf = torch.randn(M, 2 * C, dtype=torch.float64)
    Ws, bs = torch.randn(G, 32, dtype=torch.float64) / 4, torch.randn(G, dtype=torch.float64)
    temp = torch.tensor([0.05, 0.5, 1.3, 7.0, 0.9, 0.4, 2.0, 0.2][:heads], dtype=torch.float64)
    fx = xf[:, :C].reshape(B, ntok, heads, 32).permute(0, 2, 1, 3)
    xm = xf[:, C:].reshape(B, ntok, heads, 32).permute(0, 2, 1, 3)
    sw = torch.softmax((xm @ Ws.t() + bs) / temp.clamp(0.1, 5).view(1, heads, 1, 1), dim=-1)        # B h N G
    norm = sw.sum(2)
    tokS = torch.einsum("bhnc,bhng->bhgc", fx, sw)
    bps = ops.slice_blocks_per_sample(B)
    w = torch.full((M, heads * G), float("nan"), device="cuda")
    tp, npart = torch.zeros(B * bps, heads * G * 32, device="cuda"), torch.zeros(B * bps, heads * G, device="cuda")
    ops.slice_fwd(dev(xf), dev(Ws), dev(bs), dev(temp), w, tp, npart, B, ntok, heads, G, 2 * C)
    assert rel_l2(w.cpu().view(B, ntok, heads, G).permute(0, 2, 1, 3), sw) < 3e-6
    assert rel_l2(tp.double().view(B, bps, -1).sum(1).cpu().view(B, heads, G, 32), tokS) < 3e-6
    assert rel_l2(npart.double().view(B, bps, -1).sum(1).cpu().view(B, heads, G), norm) < 3e-6
    Wq, Wk, Wv = (torch.randn(32, 32, dtype=torch.float64) / 5 for _ in range(3))
    tok = tokS / (norm + 1e-5)[..., None]
    q, k, v = tok @ Wq.t(), tok @ Wk.t(), tok @ Wv.t()
    ot = torch.softmax(q @ k.transpose(-1, -2) * 32 ** -0.5, -1) @ v
    o = torch.empty(B, heads * G * 32, device="cuda")
    ops.slice_attn(dev(tokS).view(B, -1), dev(norm).view(B, -1), dev(Wq), dev(Wk), dev(Wv), o, B * heads, G)
    assert rel_l2(o.cpu().view(B, heads, G, 32), ot) < 3e-6
    ox = torch.empty(M, C, device="cuda")
    ops.deslice_fwd(dev(sw.permute(0, 2, 1, 3)).view(M, heads * G), dev(ot).view(B, -1), ox, B, ntok, heads, G)
    assert rel_l2(ox.cpu(), torch.einsum("bhgc,bhng->bhnc", ot, sw).permute(0, 2, 1, 3).reshape(M, C)) < 3e-6


def test_model_matches_reference_golden():
    from realpdebench_amd.model.transolver import Transolver
    z = np.load(os.path.join(GOLDEN_DIR, "transolver_small.npz"))
    sd = {k[3:]: torch.from_numpy(np.array(z[k])) for k in z.files if k.startswith("sd/")}
    cfg = {k[4:]: z[k] for k in z.files if k.startswith("cfg/")}
    m = Transolver(space_dim=3, n_layers=int(cfg["n_layers"]), n_hidden=int(cfg["n_hidden"]), n_head=int(cfg["n_head"]),
                   fun_dim=0, out_dim=3, slice_num=int(cfg["slice_num"]), mlp_ratio=int(cfg["mlp_ratio"]),
                   H=int(cfg["H"]), W=int(cfg["W"]), D=int(cfg["D"]), dropout=float(cfg["dropout"]))
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd)
    m = m.cuda().eval()
    with torch.no_grad():
        y = m(torch.from_numpy(np.array(z["x"])).cuda())
    assert y.shape == z["y"].shape
    assert rel_l2(y.cpu(), torch.from_numpy(np.array(z["y"]))) < 1e-5


def test_model_reference_width_vs_oracle():
    from oracle import transolver_oracle as TO
    from realpdebench_amd.model.transolver import Transolver
    torch.manual_seed(12)
    H, W, D = 10, 6, 5
    m = Transolver(space_dim=3, n_layers=1, n_hidden=256, n_head=8, fun_dim=0, out_dim=3, slice_num=16, mlp_ratio=4,
                   H=H, W=W, D=D, dropout=0.1).eval()
    x = torch.randn(2, D, W, H, 3)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    ref = TO.transolver_forward(sd, x, 1, 8, H, W, D)
    with torch.no_grad():
        y = m.cuda()(x.cuda())
    assert rel_l2(y.cpu(), ref) < 1e-5


@pytest.mark.parametrize("n_hidden,heads,layers,drop", [(64, 2, 2, 0.0), (256, 8, 1, 0.0), (64, 2, 2, 0.1)])
def test_train_loss_gradients_vs_oracle_autograd(n_hidden, heads, layers, drop):
    """Every parameter gradient of `train_loss(...).mean().backward()` (drop-in protocol, HIP backward) against
    PyTorch autograd through the CPU oracle (itself pinned to the reference)."""
    from oracle import transolver_oracle as TO
    from realpdebench_amd.model.transolver import Transolver
    torch.manual_seed(5 + n_hidden)
    H, W, D = 7, 6, 5
    m = Transolver(space_dim=3, n_layers=layers, n_hidden=n_hidden, n_head=heads, fun_dim=0, out_dim=3, slice_num=16,
                   mlp_ratio=4, H=H, W=W, D=D, dropout=drop)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("bias") or "ln_" in n:
                p.add_(0.1 * torch.randn_like(p))
        m.blocks[0].Attn.temperature.view(-1)[0] = 0.05          # below the clamp: zero gradient expected
    x, y = torch.randn(2, D, W, H, 3), torch.randn(2, D, W, H, 3)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    masks = None
    if drop > 0:          # same (pre-scaled) inverted-dropout masks on both sides: the RNG streams cannot be matched
        ntk = H * W * D
        masks = [((torch.rand(2, heads, 16, 16) < 1 - drop).float() / (1 - drop),
                  (torch.rand(2 * ntk, n_hidden) < 1 - drop).float() / (1 - drop)) for _ in range(layers)]
        m._mask_override = [(a.cuda(), b.cuda()) for a, b in masks]
    loss_ref = ((TO.transolver_forward(sd, x, layers, heads, H, W, D, masks) - y) ** 2).mean()
    loss_ref.backward()
    m = m.cuda().train()
    loss = m.train_loss(x.cuda(), y.cuda()).mean()
    loss.backward()
    assert abs(float(loss) - float(loss_ref)) < 1e-5 * float(loss_ref)
    worst = {}
    for n, p in m.named_parameters():
        assert p.grad is not None, n
        ref = sd[n].grad
        if float(ref.norm()) < 1e-12:
            assert float(p.grad.norm()) < 1e-9, n
            continue
        worst[n] = rel_l2(p.grad.cpu(), ref)
    bad = {k: v for k, v in worst.items() if v > 1e-4}
    assert not bad, bad
