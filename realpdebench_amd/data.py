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


def make_datasets(args, for_eval=False):
    """Returns ``(train_dataset, val_dataset, normalizer_stats or None)``."""
    factory = getattr(args, "dataset_factory", None)
    if factory:
        mod, fn = factory.split(":")
        return getattr(importlib.import_module(mod), fn)(args)
    name = getattr(args, "dataset_name", "synthetic")
    if name != "synthetic":
        from . import disk
        if name not in disk.SCENARIOS and name != "combustion":
            raise ValueError(f"dataset_name={name!r}: the built-in on-disk reader covers {sorted(disk.SCENARIOS)} + combustion "
                             "(the reference's V2 Arrow layout); pass --dataset_factory module:function returning "
                             "(train, val, stats) otherwise")
        return fluid_datasets(args, for_eval)
    return (SyntheticDataset(args.shape_in, args.shape_out, args.n_train, seed=args.seed),
            SyntheticDataset(args.shape_in, args.shape_out, args.n_val, seed=args.seed + 1), None)


def fluid_datasets(args, for_eval=False):
    """The three datasets of realpdebench/train.py:118-266 for a fluid scenario -- train (mode 'train', ``--train_data_type``,
    mask_prob / noise_scale from the YAML), val (mode 'val', real data; ``for_eval``: the 'test' split with the rollout horizon
    of realpdebench/eval.py:91-98) and the normaliser's (mode 'train', numerical) -- as
    ``disk.FluidWindows`` sample lists over the memory-mapped Arrow files, + the GaussianNormalizer statistics (read from /
    written to ``{dataset_dir}/mean_std.pt`` like data_normalizer.py:22-34)."""
    import logging
    import os

    from . import disk
    kw = dict(dataset_name=args.dataset_name, dataset_root=args.dataset_root)
    extra = {"mask_prob": args.mask_prob} if hasattr(args, "mask_prob") else {}       # else the scenario's default (0.5 / 0.8)
    train = disk.open_windows(mode="train", dataset_type=getattr(args, "train_data_type", "numerical"),
                              noise_scale=getattr(args, "noise_scale", 0.0), **extra, **kw)
    if for_eval:        # realpdebench/eval.py:91-98: the test split of the real data, horizon = in_step + out_step * N_autoregressive
        val = disk.open_windows(mode="test", dataset_type="real", N_autoregressive=getattr(args, "N_autoregressive", 1),
                                test_mode=getattr(args, "test_mode", "all"), **kw)
    else:
        val = disk.open_windows(mode="val", dataset_type="real", **kw)
    stats = None
    if getattr(args, "normalizer", "none") == "gaussian":
        cache = os.path.join(train.dataset_dir, "mean_std.pt")
        try:
            stats = torch.load(cache, map_location="cpu", weights_only=True)
        except Exception:
            norm_set = disk.open_windows(mode="train", dataset_type="numerical", **kw)
            stats = disk.compute_mean_std(norm_set, 512)
            try:
                torch.save(tuple(stats), cache)
            except OSError as exc:
                logging.info(f"normaliser statistics not cached ({exc})")
    return train, val, stats


class DevicePrefetcher:
    """Double-buffered asynchronous input pipeline (replaces the blocking ``.to(device)`` of
    realpdebench/data/data_normalizer.py:50-55 on the critical path of train.py:323-327).

    While step k runs on the compute stream, batch k+1 is copied from pinned host memory to HBM and normalised (the
    per-channel affine HIP kernel) on a side HIP stream; ``next()`` only makes the compute stream wait on the event of the
    batch it hands out.  ``loader`` is any iterator of CPU ``(input, target)`` batches (``pin_memory=True`` loaders copy
    asynchronously; unpinned batches are pinned here first)."""

    def __init__(self, loader, normalizer, device):
        self.it = iter(loader)
        self.normalizer = normalizer
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(self.device)
        self._next = None

    def _load(self):
        try:
            inp, tgt = next(self.it)
        except StopIteration:
            return None
        pin = lambda t: t if (t.is_cuda or t.is_pinned()) else t.pin_memory()
        with torch.cuda.stream(self.stream):
            inp = pin(inp).to(self.device, non_blocking=True)
            tgt = pin(tgt).to(self.device, non_blocking=True)
            inp, tgt = self.normalizer.preprocess(inp, tgt)          # HIP kernels, enqueued on the side stream
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return inp, tgt, ev

    def __iter__(self):
        return self

    def __next__(self):
        if self._next is None:
            self._next = self._load()
        if self._next is None:
            raise StopIteration
        inp, tgt, ev = self._next
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        inp.record_stream(cur)          # the caching allocator must not hand these blocks back to the side stream early
        tgt.record_stream(cur)
        self._next = self._load()       # batch k+1 starts moving before step k is even launched
        return inp, tgt
