"""Only the strong-scaling proxy of bench.py (tools/dbg)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
ms32 = float(sys.argv[1]) if len(sys.argv) > 1 else 36.8
print(json.dumps(bench.strong_scaling_proxy(dev, ms32), indent=1))
