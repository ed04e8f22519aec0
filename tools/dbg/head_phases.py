"""Phase cycle stamps of the pjg head kernel (instrumented build: tools/dbg/build_timing.sh; run with RPB_LIB_PATH=tools/dbg/librpb_timing.so)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from realpdebench_amd import ops
B, T, H, W, C = 32, 20, 128, 128, 64
d = ops.Dims(B, T, H, W, 2, C, 6)
f = dict(device="cuda", dtype=torch.float32)
x = torch.randn(d.ncell, C, **f)
w1, b1, w2, b2 = torch.randn(128, C, **f) / 8, torch.randn(128, **f), torch.randn(2, 128, **f) / 11, torch.randn(2, **f)
xf = (torch.zeros(C, **f), torch.ones(C, **f), torch.ones(C, **f), torch.zeros(C, **f), False)
g = torch.empty(d.ncell, C, **f)
slots, row = ops.head_bwd_slots(d), ops.head_bwd_row(2)
part, lp = torch.empty(slots, row, **f), torch.empty(slots, **f)
y = torch.randn(d.ncrop, 2, **f)
names = ["tile start (loads)", "contraction 1 + split", "activation + out partials", "reduce + gather", "d fc2", "second view planes",
         "gh + split (4x)", "LDS write + weight grad MFMAs (4x)", "tr reads + data grad MFMAs (4x)", "-", "stores / line end", "-"]
for loss in (True, False):
    for _ in range(2):
        if loss:
            ops.head_fwd_bwd(x, w1, b1, w2, b2, y, 0.37, g, part, lp, d, 2, xf)
        else:
            ops.head_bwd(x, w1, b1, w2, y, g, part, d, 2, xf)
    torch.cuda.synchronize()
    t = part[:, :12].double().cpu()
    tiles = d.B * d.T * d.H * ((W + 31) // 32) / slots
    print(f"## {'head_fwd_bwd' if loss else 'head_bwd'}: cycles per tile (mean over {slots} waves, {tiles:.0f} tiles per wave)")
    tot = 0.0
    for i, nm in enumerate(names):
        c = float(t[:, i].mean()) / tiles
        tot += c
        if nm != "-":
            print(f"  {nm:40s} {c:9.0f}")
    print(f"  {'total':40s} {tot:9.0f}")
