import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realpdebench_amd import ops
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
torch.manual_seed(0)
B, T, H, W, pad, DO, C = int(os.environ.get("DB_B", 1)), 1, 1, int(os.environ.get("DB_W", 32)), 2, 2, 64
d = ops.Dims(B, T, H, W, 2, C, pad)
f8 = dict(dtype=torch.float64)
s = (torch.randn(d.ncell, C, **f8) * 1.2 + 0.2).requires_grad_(True)
mean, invstd = torch.randn(C, **f8) * 0.2, torch.rand(C, **f8) + 0.5
gamma, beta = torch.rand(C, **f8) + 0.5, torch.randn(C, **f8) * 0.3
w1 = (torch.randn(128, C, **f8) / 8).requires_grad_(True)
b1 = torch.randn(128, **f8).requires_grad_(True)
w2 = (torch.randn(DO, 128, **f8) / 11).requires_grad_(True)
b2 = torch.randn(DO, **f8).requires_grad_(True)
sh = (s - mean) * invstd
a_full = sh * gamma + beta
a = a_full.view(B, d.Tp, d.Hp, d.Wp, C)[:, :T, :H, :W].reshape(-1, C)
u = a @ w1.t() + b1
u.retain_grad()
out = torch.nn.functional.gelu(u) @ w2.t() + b2
gout = torch.randn_like(out)
a_full.retain_grad()
out.backward(gout)
g_ref = a_full.grad
dev = lambda t: t.detach().float().cuda().contiguous()
xf = (dev(mean), dev(invstd), dev(gamma), dev(beta), False)
g = torch.full((d.ncell, C), float("nan"), device="cuda")
slots, row = ops.head_bwd_slots(d), ops.head_bwd_row(DO)
part = torch.zeros((slots, row), device="cuda")
ops.head_bwd(dev(s), dev(w1), dev(b1), dev(w2), dev(gout), g, part, d, DO, xf)
torch.cuda.synchronize()
print("g", rel(g.cpu(), g_ref))
gc = g.cpu().view(B, d.Tp, d.Hp, d.Wp, C)[:, :T, :H, :W].reshape(-1, C)
gr = g_ref.view(B, d.Tp, d.Hp, d.Wp, C)[:, :T, :H, :W].reshape(-1, C)
err = (gc.double() - gr).abs()
print("per-cell err", err.amax(1)[:40])
print("per-chan err", err.amax(0))
tot = part.double().sum(0).cpu()
M = tot[:128 * 64].view(128, 64)
gh = u.grad
Mref = gh.t() @ sh.detach().view(B, d.Tp, d.Hp, d.Wp, C)[:, :T, :H, :W].reshape(-1, C)
print("M", rel(M, Mref))
print("dw2", rel(tot[128 * 64:128 * 64 + DO * 128].view(DO, 128), w2.grad))
print("db1", rel(tot[128 * 64 + DO * 128:128 * 64 + DO * 128 + 128], b1.grad))
print("db2", rel(tot[128 * 64 + DO * 128 + 128:], b2.grad))
