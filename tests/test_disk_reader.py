"""On-disk reader against windows produced by the IMPORTED reference datasets
(tests/golden/make_golden_disk.py -> tests/golden/disk_small/ + disk_small.npz): sample order, filters, slicing,
sub-sampling, pressure masking (same ``random`` stream), parameter channels and normaliser statistics on the host side
(CPU tests); the device batch path -- pinned slabs -> rpb_window_pack -- bit-exactly on the GPU."""
import json
import os
import random

import numpy as np
import pytest
import torch

from realpdebench_amd.disk import ArrowTrajectories, FluidWindows, batch_plan, compute_mean_std, open_windows

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "golden", "disk_small")
CASES = ["cyl_num_train", "cyl_num_train_masked", "cyl_real_val", "cyl_num_test_ar", "cyl_real_test_unseen", "ctl_num_train",
         "comb_num_train", "comb_real_test_ar", "cyl_num_train_noise", "cyl_real_val_noise_ignored", "comb_num_train_noise",
         "ctl_num_train_noise"]


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "disk_small.npz"))


def _windows(gold, key):
    return open_windows(dataset_root=ROOT, **json.loads(str(gold[key + "/kw"])))


def test_arrow_store_is_a_zero_copy_view():
    st = ArrowTrajectories(os.path.join(ROOT, "cylinder", "hf_dataset", "numerical"))
    assert sorted(st.sim_ids()) == ["100.h5", "200.h5", "300.h5"] and st.has("p")
    u = st.array("200.h5", "u")
    assert u.shape == st.shape("200.h5") == (14, 6, 8) and u.dtype == np.float32
    assert not u.flags.writeable and not u.flags.owndata          # backed by the memory map, not a copy
    assert u[3:9].flags.c_contiguous                               # a time window is one contiguous slab


@pytest.mark.parametrize("key", CASES)
def test_samples_match_the_reference_dataset(gold, key):
    w = _windows(gold, key)
    assert len(w) == int(gold[key + "/n"])
    random.seed(1234)                                              # the reference draws one random.random() per numerical sample
    torch.manual_seed(99)                                          # ... and, with noise_scale > 0, randn_like(input) then randn_like(output)
    items = [w[i] for i in range(len(w))]
    assert torch.equal(torch.stack([a for a, _ in items]), torch.from_numpy(gold[key + "/inp"]))
    assert torch.equal(torch.stack([b for _, b in items]), torch.from_numpy(gold[key + "/tgt"]))


def test_normaliser_statistics_match_the_reference(gold):
    w = _windows(gold, "cyl_num_train")
    mi, mt, si, st = compute_mean_std(w, batch_size=4)
    for got, key in ((mi, "mean_in"), (mt, "mean_tgt"), (si, "std_in"), (st, "std_tgt")):
        # same formulas and batching; torch's CPU reductions may order the sums differently on another host: last-bit tolerance
        assert torch.allclose(got, torch.from_numpy(gold["stats/" + key]), rtol=2e-6, atol=1e-7), key


def test_missing_files_fail_loudly(gold):
    with pytest.raises(FileNotFoundError):
        FluidWindows("cylinder", ROOT, "numerical", "nosuchsplit")
    with pytest.raises(ValueError):
        FluidWindows("combustion", ROOT, "numerical", "train")
    with pytest.raises(ValueError):
        FluidWindows("cylinder", ROOT, "real", "test", test_mode="bogus")


def test_rank_shards_partition_every_global_batch():
    n, B, world = 37, 4, 3
    for epoch in (0, 1):
        per_rank = [list(batch_plan(n, B, world, r, True, 5, epoch)) for r in range(world)]
        assert len({len(p) for p in per_rank}) == 1                  # every rank takes the same number of steps
        seen = [i for p in per_rank for b in p for i in b]
        assert len(seen) == len(set(seen)) == (n // (B * world)) * B * world
        for step in range(len(per_rank[0])):                         # ranks' batches of one step are disjoint slices of one global batch
            assert all(len(per_rank[r][step]) == B for r in range(world))
    assert list(batch_plan(n, B, world, 0, True, 5, 0)) != list(batch_plan(n, B, world, 0, True, 5, 1))   # reshuffled per epoch
    assert list(batch_plan(6, 4, 1, 0, False, 0, 0, drop_last=False)) == [[0, 1, 2, 3], [4, 5]]


@pytest.mark.gpu
@pytest.mark.parametrize("key", [k for k in CASES if not k.endswith("_noise")])    # device noise is another random stream: below
def test_device_batches_equal_reference_windows(gold, key):
    from realpdebench_amd.disk import DiskBatchLoader
    w = _windows(gold, key)
    B = 2
    random.seed(1234)                                              # single producer thread, samples in index order: same mask stream
    loader = DiskBatchLoader(w, B, "cuda", stats=None, shuffle=False, drop_last=False, epochs=1)
    inps, tgts = zip(*[(a.cpu(), b.cpu()) for a, b in loader])
    assert torch.equal(torch.cat(inps), torch.from_numpy(gold[key + "/inp"]))
    assert torch.equal(torch.cat(tgts), torch.from_numpy(gold[key + "/tgt"]))


@pytest.mark.gpu
def test_device_batches_are_normalised_like_the_reference(gold):
    from realpdebench_amd.disk import DiskBatchLoader
    w = _windows(gold, "ctl_num_train")
    torch.manual_seed(0)
    stats = (torch.randn(5), torch.randn(3), torch.rand(5) + 0.5, torch.tensor([0.7, 0.0, 1.3]))   # a zero std -> 1
    assert (w.Cp, w.Cl, w.n_para) == (3, 0, 2)
    random.seed(1234)
    loader = DiskBatchLoader(w, 3, "cuda", stats=stats, shuffle=False, epochs=1)
    x = torch.cat([a.cpu() for a, _ in loader])
    random.seed(1234)
    loader = DiskBatchLoader(w, 3, "cuda", stats=stats, shuffle=False, epochs=1)
    y = torch.cat([b.cpu() for _, b in loader])
    st = torch.where(stats[3] == 0, torch.ones(3), stats[3])
    assert torch.equal(x, (torch.from_numpy(gold["ctl_num_train/inp"]) - stats[0]) / stats[2])     # data_normalizer.py:50-55
    assert torch.equal(y, (torch.from_numpy(gold["ctl_num_train/tgt"]) - stats[1]) / st)


@pytest.mark.gpu
def test_two_ranks_read_disjoint_halves_of_one_permutation(gold):
    from realpdebench_amd.disk import DiskBatchLoader
    w = _windows(gold, "cyl_real_val")                              # 6 samples
    ref = torch.from_numpy(gold["cyl_real_val/inp"])
    got = []
    for rank in (0, 1):
        loader = DiskBatchLoader(w, 1, "cuda", shuffle=True, seed=3, rank=rank, world=2, epochs=1)
        got.append([a.cpu() for a, _ in loader])
    plan = [list(batch_plan(6, 1, 2, r, True, 3, 0)) for r in (0, 1)]
    assert len(got[0]) == len(got[1]) == 3
    for r in (0, 1):
        for step, idxs in enumerate(plan[r]):
            assert torch.equal(got[r][step], ref[idxs])
    assert not (set(i for b in plan[0] for i in b) & set(i for b in plan[1] for i in b))


@pytest.mark.gpu
@pytest.mark.parametrize("key,clean", [("cyl_num_train_noise", None), ("comb_num_train_noise", "comb_num_train"),
                                       ("ctl_num_train_noise", "ctl_num_train")])
def test_device_gaussian_noise_has_the_reference_distribution(gold, key, clean):
    """x + x * N(0, 1) * scale per element (fluid_hf_dataset.py:309-311): the device draws its own stream, so the check is the
    distribution -- (noisy - clean) / clean is N(0, scale^2), masked channels stay exactly zero, parameter channels are not
    perturbed (the reference appends them after the noise)."""
    from realpdebench_amd.disk import DiskBatchLoader
    kw = json.loads(str(gold[key + "/kw"]))
    scale = kw["noise_scale"]
    def batches(noise):
        random.seed(1234)
        torch.manual_seed(5)
        w = open_windows(dataset_root=ROOT, **{**kw, "noise_scale": noise})
        out = list(DiskBatchLoader(w, 3, "cuda", stats=None, shuffle=False, epochs=1))
        return torch.cat([a.cpu() for a, _ in out]), torch.cat([b.cpu() for _, b in out]), w
    xn, yn, w = batches(scale)
    x0, y0, _ = batches(0.0)
    nf = w.Cp + w.Cl                                               # field channels; the rest are parameter channels
    assert torch.equal(xn[..., nf:], x0[..., nf:])
    for a, b in ((xn[..., :nf], x0[..., :nf]), (yn, y0)):
        assert torch.equal(a[b == 0], b[b == 0])                   # masked pressure / numerical block
        r = ((a - b) / b)[b.abs() > 1e-3]
        assert r.numel() > 300 and abs(float(r.mean())) < 4 * scale / r.numel() ** 0.5 + 1e-3
        assert abs(float(r.std()) / scale - 1) < 0.12, (float(r.std()), scale)


@pytest.mark.gpu
def test_device_poisson_noise(tmp_path):
    """x + Poisson(x) * scale (fluid_hf_dataset.py:312-314) on non-negative data written here in the reference's layout."""
    from datasets import Dataset
    from realpdebench_amd.disk import DiskBatchLoader
    base = tmp_path / "cylinder" / "hf_dataset"
    T, H, W, lam, scale = 12, 8, 16, 4.0, 0.5
    rows = {"sim_id": ["100.h5"], "shape_t": [T], "shape_h": [H], "shape_w": [W]}
    for k in ("u", "v", "p"):
        rows[k] = [np.full((T, H, W), lam, dtype=np.float32).tobytes()]
    Dataset.from_dict(rows).save_to_disk(str(base / "numerical"))
    (base / "train_index_numerical.json").write_text(json.dumps([{"sim_id": "100.h5", "time_id": t} for t in range(6)]))
    w = FluidWindows("cylinder", str(tmp_path), "numerical", "train", mask_prob=0.0, in_step=3, out_step=3, sub_s_numerical=1,
                     noise_scale=scale, noise_type="poisson")
    torch.manual_seed(2)
    x = torch.cat([torch.cat([a, b], dim=1).cpu() for a, b in DiskBatchLoader(w, 3, "cuda", shuffle=False, epochs=1)])
    k = (x - lam) / scale                                          # Poisson(lam) counts
    assert torch.equal(k, k.round()) and float(k.min()) >= 0
    assert abs(float(k.mean()) - lam) < 0.1 and abs(float(k.var()) - lam) < 0.3


def test_scenario_defaults_are_the_reference_subclasses():
    """The entrypoints pass none of in_step / out_step / n_sim_frame / sub_s_* (like realpdebench/train.py:131-154), so the
    defaults of the sample-list classes must be those of the reference's per-scenario dataset classes
    (tests/golden/scenario_defaults.json, read from the imported classes by make_golden_defaults.py)."""
    from realpdebench_amd.disk import SCENARIOS, CombustionWindows
    with open(os.path.join(HERE, "golden", "scenario_defaults.json")) as fh:
        ref = json.load(fh)
    assert set(ref) == set(SCENARIOS) | {"combustion"}
    for scen, want in ref.items():
        have = (CombustionWindows.SPEC if scen == "combustion" else SCENARIOS[scen])["defaults"]
        assert have == want, scen


def test_default_constructed_controlled_cylinder_windows_are_10_to_10():
    """ControlledCylinderHFDataset defaults to in_step = out_step = 10 (fluid_hf_dataset.py:498-499): the default-constructed
    window list must take them from the scenario, not from the cylinder class."""
    w = open_windows("controlled_cylinder", dataset_root=ROOT, dataset_type="numerical", mode="train")
    assert (w.in_step, w.out_step, w.horizon, w.sub_s, w.n_sim_frame) == (10, 10, 20, 2, 3990)
    w = open_windows("cylinder", dataset_root=ROOT, dataset_type="real", mode="val")
    assert (w.in_step, w.out_step, w.sub_s) == (20, 20, 1)
    w = open_windows("combustion", dataset_root=ROOT, dataset_type="real", mode="test")
    assert (w.in_step, w.sub_s, w.n_sim_frame, w.mask_prob) == (20, 2, 2001, 0.8)


@pytest.mark.gpu
def test_batches_survive_a_consumer_that_never_synchronises(gold):
    """A sync-free trainer runs many steps ahead of the GPU: batch k must still be intact when a kernel enqueued long after
    ``next(loader)`` returned finally reads it, although the loader has meanwhile produced batches k+1, k+2, ... on its side
    stream (the batch tensors are allocated on the side stream so record_stream defers their reuse)."""
    from realpdebench_amd.disk import DiskBatchLoader
    w = _windows(gold, "cyl_num_train")
    random.seed(1234)
    want = [a.cpu() for a, _ in DiskBatchLoader(w, 1, "cuda", stats=None, shuffle=False, drop_last=False, epochs=3)]
    random.seed(1234)
    loader = DiskBatchLoader(w, 1, "cuda", stats=None, shuffle=False, drop_last=False, epochs=3)
    big = torch.randn(4096, 4096, device="cuda")
    late = []
    for inp, _ in loader:                                          # no host synchronisation anywhere in this loop
        for _ in range(6):
            big = torch.tanh(big @ big * 1e-4)                     # ~ms of queued work before the batch is finally read
        late.append(inp.clone())
        del inp
    torch.cuda.synchronize()
    assert len(late) == len(want) and len(want) >= 6
    for a, b in zip(late, want):
        assert torch.equal(a.cpu(), b)
