"""Shader clock under the eval head: RPB_LIB_PATH=tools/dbg/librpb_pjh_timing.so python tools/dbg/pjh_clock.py
(a -DPH_TIMING build of csrc/rpb_pjh.hip writes per wave its shader cycles and 100 MHz ticks over the tile loop into the output tensor)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from realpdebench_amd import ops

B, T, H, W, C = 32, 20, 128, 128, 64
d = ops.Dims(B, T, H, W, 2, C, 6)
f = dict(device="cuda", dtype=torch.float32)
x = torch.randn(d.ncell, C, **f)
w1, b1, w2, b2 = torch.randn(128, C, **f), torch.randn(128, **f), torch.randn(2, 128, **f), torch.randn(2, **f)
out = torch.zeros(d.ncrop, 2, **f)
for _ in range(5):
    ops.proj_fwd(x, w1, b1, w2, b2, out, d, 2)
torch.cuda.synchronize()
nw = 2 * torch.cuda.get_device_properties(0).multi_processor_count * 4
t = out.view(-1)[:2 * nw].view(nw, 2).double().cpu()
cyc, ticks = t[:, 0], t[:, 1]
tiles = (B * T * H // nw) * ((W + 31) // 32)
print(f"waves {nw}: shader cycles per wave {cyc.mean():.0f} (min {cyc.min():.0f}, max {cyc.max():.0f}), 100 MHz ticks {ticks.mean():.0f} "
      f"-> {ticks.mean() / 100:.1f} us, clock {cyc.mean() / ticks.mean() * 0.1:.3f} GHz; {tiles} tiles per wave -> {cyc.mean() / tiles:.0f} cycles per tile and wave")
wg = cyc.view(-1, 4)
print(f"per workgroup: mean of means {wg.mean(1).mean():.0f}, min / max of workgroup means {wg.mean(1).min():.0f} / {wg.mean(1).max():.0f}, "
      f"mean spread inside a workgroup (max - min) {(wg.max(1).values - wg.min(1).values).mean():.0f}")
xcd = wg.mean(1).view(-1, 8)
print("per XCD (workgroup index % 8) mean cycles:", [round(float(v)) for v in xcd.mean(0)])
tk = ticks.view(-1, 4).mean(1).view(-1, 8)
print("per XCD mean ticks:", [round(float(v)) for v in tk.mean(0)], " max ticks", float(ticks.max()))
