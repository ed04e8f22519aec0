"""Dataset hook.  The reference's HDF5 / Arrow readers (realpdebench/data/*.py) are I/O outside the hot path
(SURVEY.md section 2, rows 12-13): any ``torch.utils.data.Dataset`` yielding ``(input[T,H,W,C_in], target[T,H,W,C_out])``
fp32 pairs plugs in through ``--dataset_factory module:function``; ``synthetic`` is built in for benchmarks."""
import importlib

import torch
from torch.utils.data import Dataset


class SyntheticDataset(Dataset):
    """N(0,1) trajectories, deterministic per index (the shape contract of fluid_dataset.py:346-398)."""

    def __init__(self, shape_in, shape_out, n, seed=0):
        self.shape_in, self.shape_out, self.n, self.seed = tuple(shape_in), tuple(shape_out), int(n), seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        return torch.randn(*self.shape_in, generator=g), torch.randn(*self.shape_out, generator=g)


def make_datasets(args):
    """Returns ``(train_dataset, val_dataset, normalizer_stats or None)``."""
    factory = getattr(args, "dataset_factory", None)
    if factory:
        mod, fn = factory.split(":")
        return getattr(importlib.import_module(mod), fn)(args)
    if getattr(args, "dataset_name", "synthetic") != "synthetic":
        raise ValueError(f"dataset_name={args.dataset_name!r}: dataset readers are outside this backend; pass "
                         "--dataset_factory module:function returning (train, val, stats) or use dataset_name: synthetic")
    return (SyntheticDataset(args.shape_in, args.shape_out, args.n_train, seed=args.seed),
            SyntheticDataset(args.shape_in, args.shape_out, args.n_val, seed=args.seed + 1), None)
