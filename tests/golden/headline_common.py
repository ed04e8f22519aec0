"""Shared by tests/golden/make_golden_headline.py (reference side) and tests/test_gpu_headline.py (HIP side): the seeded
weights / inputs of the headline workload and the fixed sub-sampling of what the fixture stores."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from realpdebench_amd.synthetic import bench_batch, checksum, normal_batch      # noqa: E402,F401

SHAPE, MODES, WIDTH, N_LAYERS = (20, 128, 128, 2), (4, 12, 16), 64, 4           # BASELINE.json configs[0]/[1]


def headline_state_dict(seed=11):
    """Reference-shaped weights from torch.rand only (no transcendental functions -> identical on every host); BatchNorm
    affine parameters and running statistics made non-trivial so that eval-mode parity means something."""
    from oracle import fno3d_oracle as O
    sd = O.init_state_dict(MODES, N_LAYERS, WIDTH, SHAPE, SHAPE, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    for l in range(N_LAYERS):
        sd[f"bns.{l}.weight"] = torch.rand(WIDTH, generator=g) + 0.5
        sd[f"bns.{l}.bias"] = torch.rand(WIDTH, generator=g) * 0.6 - 0.3
        sd[f"bns.{l}.running_mean"] = torch.rand(WIDTH, generator=g) * 0.4 - 0.2
        sd[f"bns.{l}.running_var"] = torch.rand(WIDTH, generator=g) + 0.5
    return sd


def headline_batch(B, seed=77):
    return normal_batch(seed, B, *SHAPE), normal_batch(seed + 1, B, *SHAPE)


def sample_index(numel, name, n=256):
    """Fixed pseudo-random positions of a flattened (real-viewed) gradient, keyed by the parameter name."""
    rng = np.random.default_rng(abs(hash_name(name)) % (2 ** 31))
    return torch.from_numpy(rng.integers(0, numel, size=min(n, numel)))


def hash_name(name):
    h = 0
    for ch in name:
        h = (h * 131 + ord(ch)) % 1000000007
    return h


def strided(t):
    """[B, T', H, W, C] -> every 5th frame, every 16th row / column: a 1/1280 sub-sample that touches all (b, c)."""
    return t[:, ::5, ::16, ::16, :].contiguous()
