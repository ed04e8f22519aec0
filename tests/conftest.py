import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
GOLDEN_CASES = ["tiny_w8", "small_w32", "ctrl_w32"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _flat64(a):
    a = torch.as_tensor(a)
    if a.is_complex():
        a = torch.view_as_real(a.resolve_conj())
    return a.double().flatten()


def rel_l2(a, b):
    a, b = _flat64(a), _flat64(b)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


class Golden:
    """One ``tests/golden/fno3d_<case>.npz`` fixture (made by tests/golden/make_golden.py)."""

    def __init__(self, case):
        self.z = np.load(os.path.join(GOLDEN_DIR, f"fno3d_{case}.npz"))
        self.B = int(self.z["cfg/B"])
        self.width = int(self.z["cfg/width"])
        self.n_layers = int(self.z["cfg/n_layers"])
        self.shape_in = tuple(int(v) for v in self.z["cfg/shape_in"])
        self.shape_out = tuple(int(v) for v in self.z["cfg/shape_out"])
        self.modes = tuple(int(v) for v in self.z["cfg/modes"])
        self.lr0 = float(self.z["lr0"])
        self.t_max = int(self.z["t_max"])

    def t(self, key):
        return torch.from_numpy(np.array(self.z[key]))

    def sd(self, prefix):
        pre = prefix + "/"
        return {k[len(pre):]: torch.from_numpy(np.array(self.z[k])) for k in self.z.files if k.startswith(pre)}

    def norm(self):
        return tuple(self.t(k) for k in ("mean_in", "std_in", "mean_tg", "std_tg"))


@pytest.fixture(params=GOLDEN_CASES)
def golden(request):
    return Golden(request.param)


def galerkin_golden():
    """tests/golden/galerkin_small.npz (made by tests/golden/make_golden_galerkin.py) as torch tensors."""
    z = np.load(os.path.join(GOLDEN_DIR, "galerkin_small.npz"))

    def group(pre):
        out = {}
        for k in z.files:
            if k.startswith(pre):
                v = torch.from_numpy(np.array(z[k]))
                if "spectral_conv" in k:
                    v = torch.view_as_complex(v.contiguous())
                out[k[len(pre):]] = v
        return out

    g = {"sd": group("sd/"), "grad": group("grad/"), "buf1": group("buf1/")}
    for k in ("x", "target", "y_eval", "loss"):
        g[k] = torch.from_numpy(np.array(z[k]))
    g["heads"] = int(z["cfg/heads"])
    g["modes"] = tuple(int(v) for v in z["cfg/modes"])
    g["shape_out"] = tuple(int(v) for v in z["cfg/shape_out"])
    return g


def dpot_golden(name="dpot_small"):
    """tests/golden/dpot_small.npz / dpot_resize_small.npz (make_golden_dpot.py: imported reference DPOT, forward + loss + every
    gradient; the second one at a data resolution that differs from img_size)."""
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    g = dict(sd={}, grad={}, cfg={})
    for k in z.files:
        if k.startswith("sd/"):
            g["sd"][k[3:]] = torch.from_numpy(z[k])
        elif k.startswith("grad/"):
            g["grad"][k[5:]] = torch.from_numpy(z[k])
        elif k.startswith("cfg/"):
            g["cfg"][k[4:]] = int(z[k])
    g["x"], g["y"], g["pred"], g["loss"] = torch.from_numpy(z["x"]), torch.from_numpy(z["y"]), torch.from_numpy(z["pred"]), float(z["loss"])
    g["cfg"].update(time_agg="exp_mlp", data_out_channels=g["y"].shape[-1])
    return g
