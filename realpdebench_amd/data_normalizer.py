"""Normalisers with the reference's interface (realpdebench/data/data_normalizer.py): ``preprocess(x, y)`` /
``postprocess(x, y)`` move to the device and apply the per-channel affine maps -- here as HIP kernels.

Statistics are passed in (the reference computes / caches them from its HDF5 datasets, :64-95, which is
dataset I/O and outside the hot path)."""
import torch

from . import ops


class IdentityNormalizer:
    def __init__(self, device):
        self.device = device

    def preprocess(self, x, y):
        return x.to(self.device), y.to(self.device)

    def postprocess(self, x, y):
        return x.to(self.device), y.to(self.device)


class GaussianNormalizer:
    def __init__(self, mean_inputs, mean_targets, std_inputs, std_targets, device):
        self.device = device
        f = lambda t: torch.as_tensor(t, dtype=torch.float32).flatten().to(device).contiguous()
        self.mean_inputs, self.mean_targets = f(mean_inputs), f(mean_targets)
        self.std_inputs, self.std_targets = f(std_inputs), f(std_targets)
        # data_normalizer.py:47-48: zero std -> 1
        self.std_inputs = torch.where(self.std_inputs == 0, torch.ones_like(self.std_inputs), self.std_inputs)
        self.std_targets = torch.where(self.std_targets == 0, torch.ones_like(self.std_targets), self.std_targets)

    def _apply(self, t, mean, std, inverse):
        t = t.to(self.device, non_blocking=True).contiguous().float()
        c = t.shape[-1]
        out = torch.empty_like(t)
        ops.channel_affine(t, out, t.numel(), c, mean[:c].contiguous(), std[:c].contiguous(), inverse)
        return out

    def preprocess(self, x, y):
        return (self._apply(x, self.mean_inputs, self.std_inputs, False),
                self._apply(y, self.mean_targets, self.std_targets, False))

    def postprocess(self, x, y):
        return (self._apply(x, self.mean_inputs, self.std_inputs, True),
                self._apply(y, self.mean_targets, self.std_targets, True))


class RangeNormalizer(GaussianNormalizer):
    """data_normalizer.py:98-159: ``x / max|x|`` per channel (zero max -> 1) -- the affine kernel with shift 0, scale max."""

    def __init__(self, max_inputs, max_targets, device):
        z = lambda t: torch.zeros_like(torch.as_tensor(t, dtype=torch.float32).flatten())
        super().__init__(z(max_inputs), z(max_targets), max_inputs, max_targets, device)
        self.max_inputs, self.max_targets = self.std_inputs, self.std_targets
