"""Reproducible synthetic batches for the benchmark and the headline parity fixtures.

``numpy.random.default_rng`` (PCG64 + ziggurat, plain C, no SIMD dispatch) gives the same N(0,1) stream on every
host, unlike ``torch.randn`` whose vectorised CPU path depends on the CPU capability and whose device generator differs
from the host one; ``tests/golden/fno3d_headline.npz`` stores checksums of these batches so a differing stream fails loudly.
"""
import numpy as np
import torch


def normal_batch(seed, *shape):
    return torch.from_numpy(np.random.default_rng(seed).standard_normal(shape, dtype=np.float32))


def bench_batch(B, rank=0, shape=(20, 128, 128, 2)):
    """(input, target) of ``bench.py`` on ``rank``: N(0,1), seeds 1000 + 2*rank and 1001 + 2*rank."""
    return normal_batch(1000 + 2 * rank, B, *shape), normal_batch(1001 + 2 * rank, B, *shape)


def checksum(t):
    """Order-independent-enough float64 fingerprint of a tensor: (sum, sum of squares, sum of i-weighted values)."""
    t = torch.as_tensor(t)
    if t.is_complex():
        t = torch.view_as_real(t)
    v = t.detach().double().flatten().cpu()
    w = torch.arange(v.numel(), dtype=torch.float64).remainder_(997.0).add_(1.0)
    return np.array([float(v.sum()), float((v * v).sum()), float((v * w).sum())])
