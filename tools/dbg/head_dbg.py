import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from realpdebench_amd import ops
B, T, H, W, pad, DO = 2, 3, 5, 32, 2, 2
torch.manual_seed(1)
C = 64
d = ops.Dims(B, T, H, W, 2, C, pad)
f8 = dict(dtype=torch.float64)
dev = lambda t: t.to(device="cuda", dtype=torch.float32).contiguous()
s = torch.randn(d.ncell, C, **f8) * 1.2 + 0.2
mean, invstd = torch.randn(C, **f8) * 0.2, torch.rand(C, **f8) + 0.5
gamma, beta = torch.rand(C, **f8) + 0.5, torch.randn(C, **f8) * 0.3
w1 = torch.randn(128, C, **f8) / 8
b1 = torch.randn(128, **f8)
w2 = torch.randn(DO, 128, **f8) / 11
sh = (s - mean) * invstd
a_full = sh * gamma + beta
crop = lambda t: t.view(B, d.Tp, d.Hp, d.Wp, C)[:, :T, :H, :W].reshape(-1, C)
a = crop(a_full)
u = a @ w1.t() + b1
u.requires_grad_(True)
v = torch.nn.functional.gelu(u)
gout = torch.randn(a.shape[0], DO, **f8)
(v @ w2.t()).backward(gout)
gh = u.grad
M = gh.t() @ crop(sh)
M_nomean = gh.t() @ (crop(s) * invstd)
xf = (dev(mean), dev(invstd), dev(gamma), dev(beta), False)
g = torch.full((d.ncell, C), float("nan"), device="cuda")
slots, row = ops.head_bwd_slots(d), ops.head_bwd_row(DO)
part = torch.full((slots, row), float("nan"), device="cuda")
ops.head_bwd(dev(s), dev(w1), dev(b1), dev(w2), dev(gout), g, part, d, DO, xf)
tot = part.double().sum(0).cpu()
Mk = tot[:128 * 64].view(128, 64)
rl = lambda x, y: float((x - y).norm() / y.norm())
print("M even", rl(Mk[:, 0::2], M[:, 0::2]), "odd", rl(Mk[:, 1::2], M[:, 1::2]))
print("odd vs no-mean", rl(Mk[:, 1::2], M_nomean[:, 1::2]), "odd vs even cols of ref", rl(Mk[:, 1::2], M[:, 0::2]))
print("odd vs ref scaled: ratio sample", (Mk[:4, 1] / M[:4, 1]).tolist(), (Mk[:4, 3] / M[:4, 3]).tolist())
# per hidden tile
for mt in range(4):
    print("mt", mt, "odd err", rl(Mk[32 * mt:32 * mt + 32, 1::2], M[32 * mt:32 * mt + 32, 1::2]))
print("db1", rl(tot[128 * 64 + 2 * 128:128 * 64 + 3 * 128], gh.sum(0)), "dw2", rl(tot[128 * 64:128 * 64 + 256].view(2, 128), gout.t() @ v.detach()), "db2", rl(tot[-2:], gout.sum(0)))
# which reference column does each odd kernel column resemble?
Mraw = gh.t() @ crop(s)          # no mean, no invstd
db = gh.sum(0)
for j in (1, 3, 5, 63):
    col = Mk[:, j]
    best = sorted(((float(torch.dot(col, Mraw[:, c]) / (col.norm() * Mraw[:, c].norm())), c) for c in range(64)), reverse=True)[:3]
    # fit col = alpha * Mraw[:, c] + beta * db for c = j
    A = torch.stack([Mraw[:, j], db], 1)
    sol = torch.linalg.lstsq(A, col.unsqueeze(1)).solution.squeeze()
    res = float((A @ sol - col).norm() / col.norm())
    print("col", j, "best cos", best, "fit alpha,beta", sol.tolist(), "res", res, "expected alpha", float(invstd[j]), "beta", float(-mean[j] * invstd[j]))
print("---- LOSS variant")
b2 = torch.randn(DO, **f8)
out = v.detach() @ w2.t() + b2
y = out - gout / 0.37
g_ref_c = gh @ w1          # [ncrop][64] gradient wrt a on the cropped cells
g2 = torch.full((d.ncell, C), float("nan"), device="cuda")
part2 = torch.full((slots, row), float("nan"), device="cuda")
lpart = torch.full((slots,), float("nan"), device="cuda")
ops.head_fwd_bwd(dev(s), dev(w1), dev(b1), dev(w2), dev(b2), dev(y), 0.37, g2, part2, lpart, d, DO, xf)
g2c = crop(g2.cpu().double())
print("loss", float(lpart.double().sum()), "ref", float(((out - y) ** 2).sum()))
err = (g2c - g_ref_c).norm(dim=1) / g_ref_c.norm(dim=1)
print("per-cell rel err of the first line (32 cells):", [round(float(e), 3) for e in err[:32]])
tot2 = part2.double().sum(0).cpu()
print("db2", tot2[-2:].tolist(), "ref", gout.sum(0).tolist())
