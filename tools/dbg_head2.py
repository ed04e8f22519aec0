import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realpdebench_amd import ops
torch.manual_seed(0)
B, T, H, W, pad, DO, C = 1, 1, 1, 32, 2, 2, 64
d = ops.Dims(B, T, H, W, 2, C, pad)
f = dict(device="cuda", dtype=torch.float32)
s = torch.zeros(d.ncell, C, **f)
mean, invstd, gamma, beta = torch.zeros(C, **f), torch.ones(C, **f), torch.ones(C, **f), torch.zeros(C, **f)
w1 = torch.zeros(128, C, **f); b1 = torch.zeros(128, **f)
w2 = torch.ones(DO, 128, **f)
gout = torch.arange(d.ncrop * DO, **f).view(d.ncrop, DO)
g = torch.full((d.ncell, C), float("nan"), **f)
slots, row = ops.head_bwd_slots(d), ops.head_bwd_row(DO)
part = torch.zeros((slots, row), **f)
ops.head_bwd(s, w1, b1, w2, gout, g, part, d, DO, (mean, invstd, gamma, beta, False))
torch.cuda.synchronize()
print("slots", slots, "db2 rows", part[:, -DO:].cpu(), "expect", gout.sum(0).cpu())
# u = 0 -> gelu'(0) = 0.5, gh = 0.5 * (go0 + go1); db1[h] = sum_cells gh
print("db1 rows", part[:, 128*64 + DO*128: 128*64 + DO*128 + 4].cpu(), "expect", 0.5 * gout.sum().item())
print("dw2 (v=gelu(0)=0)", part[:, 128*64:128*64+4].cpu())
