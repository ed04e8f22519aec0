#!/usr/bin/env python
"""Residency profile of the C = 64 cell_mix launches: per-wave start / end ticks (rpb_cmx_debug_wave_times) at the headline shape.

  python tools/wave_times.py            # forward + stats, eval, backward + sums, backward + weight gradient

Prints for each launch: the launch span (first start .. last end), the mean / min / max wave lifetime, the distribution of the
end times, and the per-XCD mean end time (workgroup b runs on XCD b % 8).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realpdebench_amd import ops, _lib  # noqa: E402
from realpdebench_amd.dft import SpectralPlan  # noqa: E402

B, T, H, W, Cin, C = int(os.environ.get("KB_B", 32)), 20, 128, 128, 2, 64
d = ops.Dims(B, T, H, W, Cin, C, 6)
plan = SpectralPlan(d.Tp, d.Hp, d.Wp, (4, 12, 16), device="cuda")
f = dict(device="cuda", dtype=torch.float32)
x = torch.randn(d.ncell, C, **f)
y = torch.empty(d.ncell, C, **f)
K2 = 2 * plan.KW
z2 = torch.randn(B * d.Tp * d.Hp * K2 * C, **f)
Wc, bias = torch.randn(C, C, **f), torch.randn(C, **f)
rows = ops.cell_mix_stat_rows(d.ncell, C, C, K2, d.Wp, True)
part = torch.empty(rows * 2 * C, **f)
mean, invstd, gamma, beta = torch.zeros(C, **f), torch.ones(C, **f), torch.ones(C, **f), torch.zeros(C, **f)
xfg = (mean, invstd, gamma, beta, True)
rows2 = ops.cell_mix_stat_rows(d.ncell, C, C, K2, d.Wp, True, True)
part2 = torch.empty(rows2 * 2 * C, **f)
s_prev = torch.randn(d.ncell, C, **f)
buf = torch.zeros(256 * 16 * 2, device="cuda", dtype=torch.int64)


def profile(name, fn, waves):
    fn()
    fn()
    torch.cuda.synchronize()
    buf.zero_()
    _lib.call("rpb_cmx_debug_wave_times", buf.data_ptr())
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    fn()
    e.record()
    torch.cuda.synchronize()
    _lib.call("rpb_cmx_debug_wave_times", 0)
    t = buf.cpu().view(-1, 2)[: 256 * waves].double()
    t = t[t[:, 1] > 0]
    t0 = t[:, 0].min()
    st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0         # microseconds
    life = en - st
    span = en.max()
    q = torch.quantile(en, torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], dtype=torch.float64))
    print(f"{name}: event {s.elapsed_time(e) * 1e3:.0f} us, span {span:.0f} us, {len(t)} waves; start max {st.max():.0f} us; "
          f"lifetime mean {life.mean():.0f} min {life.min():.0f} max {life.max():.0f} us "
          f"(mean / span = {life.mean() / span:.3f})")
    print("   end-time quantiles 0/10/50/90/100 %: " + " ".join(f"{v:.0f}" for v in q.tolist()))
    blk = torch.arange(len(t)) // waves
    per_xcd = [en[(blk % 8) == k].mean().item() for k in range(8)]
    print("   mean end per XCD: " + " ".join(f"{v:.0f}" for v in per_xcd))
    per_blk = torch.stack([en[blk == b].max() for b in range(int(blk.max()) + 1)])
    o = torch.argsort(per_blk)
    print("   earliest blocks: " + " ".join(f"{int(b)}:{per_blk[b]:.0f}" for b in o[:6].tolist())
          + " | latest: " + " ".join(f"{int(b)}:{per_blk[b]:.0f}" for b in o[-6:].tolist()))


profile("forward + stats (lazy BN+GELU in)",
        lambda: ops.cell_mix(x, Wc, bias, z2, plan.GWt, y, part, d.ncell, C, C, K2, d.Wp, xf=xfg), 8)
profile("forward + stats (plain in)",
        lambda: ops.cell_mix(x, Wc, bias, z2, plan.GWt, y, part, d.ncell, C, C, K2, d.Wp), 8)
profile("backward (spec)",
        lambda: ops.cell_mix(x, Wc, None, z2, plan.FW, y, None, d.ncell, C, C, K2, d.Wp, transpose_w=True), 8)
profile("backward + BN sums",
        lambda: ops.cell_mix(x, Wc, None, z2, plan.FW, y, part2, d.ncell, C, C, K2, d.Wp, transpose_w=True, bnb=(s_prev,) + xfg), 8)
if ops.cell_mix_wgrad_supported(d.ncell, K2, d.Wp):
    slots = ops.cell_mix_wgrad_slots(d.ncell, d.Wp)
    sp, wp = torch.empty(slots * 2 * C, **f), torch.empty(slots * C * C, **f)
    profile("backward + BN sums + weight gradient (wave pairs)",
            lambda: ops.cell_mix_wgrad(x, Wc, z2, plan.FW, y, sp, wp, d.ncell, K2, d.Wp, (s_prev,) + xfg), 4)
