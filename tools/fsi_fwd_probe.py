import os, sys, torch
sys.path.insert(0, os.getcwd())
from realpdebench_amd import _lib
from realpdebench_amd.model.fno import FNO3d
m = FNO3d(4, 16, 16, 4, 128, (20, 64, 64, 3), (20, 64, 64, 3)).cuda().eval()
x = torch.randn(32, 20, 64, 64, 3, device="cuda")
with torch.no_grad():
    for _ in range(2): m(x)
    torch.cuda.synchronize()
    _lib.PROFILE, _lib.PROFILE_ONLY = {}, None
    for _ in range(3): m(x)
    torch.cuda.synchronize()
prof = _lib.profile_summary()
tot = sum(v["total_ms"] for v in prof.values()) / 3
for label, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
    print(f"{label:44s} calls/fwd {v['calls'] / 3:5.1f}  avg {v['total_ms'] / v['calls']:7.3f} ms  {100 * v['total_ms'] / 3 / tot:5.1f}%  {v['bytes'] * v['calls'] / v['total_ms'] / 1e6:8.1f} GB/s")
print(f"kernel time per forward {tot:.3f} ms")
