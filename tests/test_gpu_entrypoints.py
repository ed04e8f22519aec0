"""`python -m realpdebench_amd.train` / `.eval` end to end on a tiny synthetic config (checkpoint format included)."""
import glob
import os

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu


def test_train_then_eval(tmp_path):
    from realpdebench_amd import eval as ev
    from realpdebench_amd import train as tr
    cfg = dict(exp_name="t", gpu=0, seed=0, results_path=str(tmp_path), dataset_name="synthetic", dataset_root="",
               num_workers=0, normalizer="none", shape_in=[4, 12, 10, 2], shape_out=[4, 12, 10, 2], n_train=8, n_val=4,
               model_name="fno", checkpoint_path="", modes1=2, modes2=3, modes3=3, n_layers=2, width=32, is_use_tb=None,
               scheduler="cosine", step_size=10, num_update=100, train_batch_size=4, test_batch_size=4, lr=1e-3,
               clip_grad_norm=0.0, N_autoregressive=1)
    path = tmp_path / "fno.yaml"
    path.write_text(yaml.safe_dump(cfg))
    exp = tr.main(["--config", str(path), "--max_updates", "4"])
    ckpts = sorted(glob.glob(os.path.join(exp, "model_*.pth")))
    assert ckpts, "no checkpoint written"
    ck = torch.load(ckpts[-1], map_location="cpu")
    assert set(ck) == {"model_state_dict", "train_losses", "val_losses", "iteration", "best_iteration", "best_val_loss"}
    assert ck["iteration"] == 4 and len(ck["train_losses"]) == 4
    assert set(ck["val_losses"]) == {"normalized_mse", "rmse", "mae", "rel_l2_error", "r2", "ke_error", "f_error", "low_f_error",
                                     "mid_f_error", "high_f_error", "rel_low_f_error", "rel_mid_f_error", "rel_high_f_error",
                                     "freq_error"}                   # train.py:303-319
    assert all(len(v) == len(ck["val_losses"]["rmse"]) >= 1 for v in ck["val_losses"].values())
    assert all(x == x for k in ("normalized_mse", "rmse", "mae", "rel_l2_error", "r2") for x in ck["val_losses"][k])
    # (the Fourier band means are over empty bin ranges on a 4 x 12 x 10 grid -- NaN in the reference as well)
    assert ck["model_state_dict"]["spectral_convs.0.weights1"].dtype == torch.complex64
    assert ck["train_losses"][-1] < ck["train_losses"][0] * 1.5          # finite and sane
    ev.main(["--config", str(path), "--checkpoint_path", ckpts[-1]])
    assert os.path.exists(os.path.join(exp, "eval.log"))


def test_transolver_train_then_eval(tmp_path):
    """Same entrypoints with model_name: transolver (ProtocolTrainer + HIP autograd path, dropout 0.1, 2-step rollout)."""
    from realpdebench_amd import eval as ev
    from realpdebench_amd import train as tr
    cfg = dict(exp_name="t", gpu=0, seed=0, results_path=str(tmp_path), dataset_name="synthetic", dataset_root="",
               num_workers=0, normalizer="none", shape_in=[4, 6, 8, 3], shape_out=[4, 6, 8, 3], n_train=8, n_val=4,
               model_name="transolver", space_dim=3, n_layers=2, n_hidden=64, n_head=2, H=8, W=6, D=4, fun_dim=0,
               out_dim=3, ref=4, dropout=0.1, act="gelu", mlp_ratio=2, slice_num=16, checkpoint_path="", is_use_tb=None,
               scheduler="cosine", step_size=10, num_update=100, train_batch_size=4, test_batch_size=4, lr=1e-3,
               clip_grad_norm=0.0, N_autoregressive=2)
    path = tmp_path / "ts.yaml"
    path.write_text(yaml.safe_dump(cfg))
    exp = tr.main(["--config", str(path), "--max_updates", "6"])
    ckpts = sorted(glob.glob(os.path.join(exp, "model_*.pth")))
    ck = torch.load(ckpts[-1], map_location="cpu")
    assert ck["iteration"] == 6 and "blocks.0.Attn.in_project_x.weight" in ck["model_state_dict"]
    assert all(l == l and l < 1e3 for l in ck["train_losses"])                  # finite
    assert ck["train_losses"][-1] < 1.5 * ck["train_losses"][0]                 # sane after six updates (dropout on)
    ev.main(["--config", str(path), "--checkpoint_path", ckpts[-1]])


def test_galerkin_train_then_eval(tmp_path):
    """Same entrypoints with model_name: galerkin_transformer (reference YAML key surface, reduced mesh / freq_dim)."""
    from realpdebench_amd import eval as ev
    from realpdebench_amd import train as tr
    with open(os.path.join(os.path.dirname(tr.__file__), "configs", "cylinder", "galerkin_transformer.yaml")) as fh:
        cfg = yaml.safe_load(fh)
    cfg.update(exp_name="g", results_path=str(tmp_path), shape_in=[4, 6, 8, 3], shape_out=[4, 6, 8, 3], n_train=8, n_val=4,
               freq_dim=32, fourier_modes_t=2, fourier_modes_x=3, fourier_modes_y=4, num_update=100, train_batch_size=4,
               test_batch_size=4, lr=1e-3, N_autoregressive=2)
    path = tmp_path / "gk.yaml"
    path.write_text(yaml.safe_dump(cfg))
    exp = tr.main(["--config", str(path), "--max_updates", "6"])
    ckpts = sorted(glob.glob(os.path.join(exp, "model_*.pth")))
    ck = torch.load(ckpts[-1], map_location="cpu")
    assert ck["iteration"] == 6 and "regressor.spectral_conv.0.weights1" in ck["model_state_dict"]
    assert ck["model_state_dict"]["regressor.spectral_conv.0.weights1"].dtype == torch.complex64
    assert all(l == l and l < 1e3 for l in ck["train_losses"])
    ev.main(["--config", str(path), "--checkpoint_path", ckpts[-1]])
    assert os.path.exists(os.path.join(exp, "eval.log"))


def test_device_prefetcher_matches_blocking_path():
    """Row f1: batches delivered through the side-stream prefetcher equal the blocking normaliser path, in order."""
    from realpdebench_amd.data import DevicePrefetcher
    from realpdebench_amd.data_normalizer import GaussianNormalizer
    torch.manual_seed(0)
    batches = [(torch.randn(3, 4, 6, 8, 3), torch.randn(3, 4, 6, 8, 3)) for _ in range(5)]
    mean_i, mean_t = torch.randn(3), torch.randn(3)
    std_i, std_t = torch.rand(3) + 0.5, torch.rand(3) + 0.5
    norm = GaussianNormalizer(mean_i, mean_t, std_i, std_t, device="cuda")
    got = []
    for inp, tgt in DevicePrefetcher(iter(batches), norm, "cuda"):
        got.append((inp * 1.0, tgt * 1.0))            # consume on the compute stream
    torch.cuda.synchronize()
    assert len(got) == len(batches)
    for (gi, gt), (bi, bt) in zip(got, batches):
        ri, rt = norm.preprocess(bi, bt)
        assert torch.equal(gi, ri) and torch.equal(gt, rt)


def test_unet_train_then_eval(tmp_path):
    """Same entrypoints with model_name: unet (reference YAML key surface, reduced mesh; dim = H = 64)."""
    from realpdebench_amd import eval as ev
    from realpdebench_amd import train as tr
    with open(os.path.join(os.path.dirname(tr.__file__), "configs", "cylinder", "unet.yaml")) as fh:
        cfg = yaml.safe_load(fh)
    cfg.update(exp_name="u", results_path=str(tmp_path), shape_in=[2, 64, 16, 3], shape_out=[2, 64, 16, 3], n_train=4, n_val=2,
               num_update=100, train_batch_size=2, test_batch_size=2, lr=1e-4, N_autoregressive=2)
    path = tmp_path / "unet.yaml"
    path.write_text(yaml.safe_dump(cfg))
    exp = tr.main(["--config", str(path), "--max_updates", "4"])
    ckpts = sorted(glob.glob(os.path.join(exp, "model_*.pth")))
    ck = torch.load(ckpts[-1], map_location="cpu")
    assert ck["iteration"] == 4 and "downs.1.4.weight" in ck["model_state_dict"]
    assert all(l == l and l < 1e3 for l in ck["train_losses"])
    ev.main(["--config", str(path), "--checkpoint_path", ckpts[-1]])
    assert os.path.exists(os.path.join(exp, "eval.log"))


def _write_v2_dataset(root, scenario="cylinder", n_sim=2, t_full=46, hw_real=(8, 12)):
    """A small dataset in the reference's V2 Arrow layout (utils/convert_hdf5_to_hf.py:20-51), written with the same library
    call the reference's converter uses; numerical data at twice the resolution of real data (sub_s_numerical = 2)."""
    import json

    import numpy as np
    from datasets import Dataset
    rng = np.random.default_rng(3)
    base = os.path.join(root, scenario, "hf_dataset")
    sims = [f"{100 * (i + 1)}.h5" for i in range(n_sim)]
    for dtype, (h, w) in (("real", hw_real), ("numerical", (2 * hw_real[0], 2 * hw_real[1]))):
        rows = {k: [] for k in ("sim_id", "u", "v", "shape_t", "shape_h", "shape_w")}
        if dtype == "numerical":
            rows["p"] = []
        for sid in sims:
            rows["sim_id"].append(sid)
            for k in ("u", "v") + (("p",) if dtype == "numerical" else ()):
                rows[k].append((rng.standard_normal((t_full, h, w)) + 0.5).astype(np.float32).tobytes())
            rows["shape_t"].append(t_full)
            rows["shape_h"].append(h)
            rows["shape_w"].append(w)
        Dataset.from_dict(rows).save_to_disk(os.path.join(base, dtype))
        for split, times in (("train", range(0, 6)), ("val", (0, 3)), ("test", (1, 2))):
            with open(os.path.join(base, f"{split}_index_{dtype}.json"), "w") as fh:
                json.dump([{"sim_id": s, "time_id": t} for s in sims for t in times], fh)


def test_train_from_the_on_disk_arrow_layout(tmp_path):
    """dataset_name: cylinder + dataset_root: the built-in reader feeds the trainer -- memory-mapped Arrow
    slabs, rpb_window_pack with the Gaussian normaliser fused, statistics cached as mean_std.pt like the reference."""
    from realpdebench_amd import eval as ev
    from realpdebench_amd import train as tr
    root = tmp_path / "data"
    _write_v2_dataset(str(root))
    cfg = dict(exp_name="t", gpu=0, seed=0, results_path=str(tmp_path), dataset_name="cylinder", dataset_root=str(root),
               num_workers=0, normalizer="gaussian", mask_prob=0.5, noise_scale=0.1, model_name="fno", checkpoint_path="",
               modes1=2, modes2=3, modes3=3, n_layers=2, width=32, is_use_tb=None, scheduler="cosine", step_size=10,
               num_update=100, train_batch_size=4, test_batch_size=2, lr=1e-3, clip_grad_norm=0.0, N_autoregressive=1)
    path = tmp_path / "fno.yaml"
    path.write_text(yaml.safe_dump(cfg))
    exp = tr.main(["--config", str(path), "--max_updates", "6", "--train_data_type", "numerical"])
    ck = torch.load(sorted(glob.glob(os.path.join(exp, "model_*.pth")))[-1], map_location="cpu")
    assert ck["iteration"] == 6 and all(l == l and l < 1e3 for l in ck["train_losses"])     # finite, normalised-scale losses
    ckpt = sorted(glob.glob(os.path.join(exp, "model_*.pth")))[-1]
    res = ev.main(["--config", str(path), "--checkpoint_path", ckpt, "--use_hf_dataset"])   # test split of the real data
    assert os.path.exists(os.path.join(exp, "eval.log"))
    assert res["evaluated_channels"] == 2                           # real data has no pressure: eval.py:297-302 drops the all-zero channel
    assert all(v == v for v in res.values()) and res["rmse"] > 0
    _check_evaluate_against_a_plain_restatement(str(root), path, ckpt, res)
    stats = torch.load(os.path.join(str(root), "cylinder", "mean_std.pt"), weights_only=True)
    assert len(stats) == 4 and stats[0].shape == (3,) and abs(float(stats[0][0]) - 0.5) < 0.05      # fields ~ N(0.5, 1)


def _check_evaluate_against_a_plain_restatement(root, cfg_path, ckpt, res):
    """eval.evaluate vs the loop of realpdebench/eval.py:294-352 written out with plain torch ops on the same model and split."""
    import argparse

    from torch.utils.data import DataLoader

    from realpdebench_amd import eval as ev
    from realpdebench_amd.data import make_datasets
    from realpdebench_amd.data_normalizer import GaussianNormalizer
    from realpdebench_amd.model import load_model
    from realpdebench_amd.utils import add_args_from_config
    args = add_args_from_config(argparse.Namespace(config=str(cfg_path), train_data_type="numerical"))
    train_ds, test_ds, stats = make_datasets(args, for_eval=True)
    model = load_model(train_ds, device="cuda", **vars(args))
    model.load_checkpoint(ckpt, "cuda")
    model.eval()
    mi, mt, si, st = (t.cuda() for t in stats)
    loss, preds, tgts = 0.0, [], []
    loader = DataLoader(test_ds, batch_size=args.test_batch_size, shuffle=False)
    with torch.no_grad():
        for inp, tgt in loader:
            x, t = (inp.cuda() - mi) / si, (tgt.cuda() - mt) / st
            p = model(x)                                            # N_autoregressive = 1
            p = (p * st + mt - mi) / si                             # eval.py:315-318: postprocess, then preprocess as an INPUT --
                                                                    # the reference's rollout does this after the last step too
            loss += float(((p[..., :2] - t[..., :2]) ** 2).reshape(inp.size(0), -1).mean())
            preds.append(p * st + mt)
            tgts.append(t * st + mt)
    pred, target = torch.cat(preds)[..., :2], torch.cat(tgts)[..., :2]
    b = pred.shape[0]
    rmse = float(((pred - target) ** 2).mean().sqrt())
    mae = float((pred - target).abs().mean())
    rel = float((torch.norm((pred - target).reshape(b, -1), dim=1) / torch.norm(target.reshape(b, -1), dim=1)).mean())
    r2 = float(1 - ((pred - target) ** 2).sum() / ((target - target.mean(0, keepdim=True)) ** 2).sum())
    for k, v in (("normalized_mse", loss / len(loader)), ("rmse", rmse), ("mae", mae), ("rel l2 error", rel), ("r2", r2)):
        assert abs(res[k] - v) <= 2e-5 * max(1.0, abs(v)), (k, res[k], v)
    res2, p2, t2 = ev.evaluate(model, loader, GaussianNormalizer(*stats, device="cuda"), 1, args.test_batch_size)
    assert torch.allclose(p2[..., :2], pred, rtol=1e-5, atol=1e-5) and abs(res2["rmse"] - res["rmse"]) < 1e-7


def test_eval_metrics_hip_path_matches_reference():
    """Row f3 on the device: eval_metrics through rpb_axis_gemm (three truncated DFT stages) + rpb_spectrum_bin equals the
    reference's fftn + Python-triple-loop values (tests/golden/metrics_small.npz, generated from the imported reference)."""
    import os
    import numpy as np
    from conftest import GOLDEN_DIR
    from realpdebench_amd import _lib
    from realpdebench_amd.metrics import eval_metrics
    z = np.load(os.path.join(GOLDEN_DIR, "metrics_small.npz"))
    _lib.PROFILE, _lib.PROFILE_ONLY = {}, None
    try:
        for name in ("a", "b", "c"):
            pred, tgt = torch.from_numpy(z[f"{name}/pred"]).cuda(), torch.from_numpy(z[f"{name}/target"]).cuda()
            bs = int(z[f"{name}/bs"]) or None
            vals = eval_metrics(pred, tgt, int(z[f"{name}/c"]), batch_size=bs)
            ref = z[f"{name}/vals"]
            for i, (v, r) in enumerate(zip(vals, ref)):
                if not np.isfinite(r):
                    assert float(v) == r, (name, i, float(v), r)
                    continue
                assert abs(float(v) - r) <= 2e-5 * max(abs(r), 1e-3), (name, i, float(v), r)
        torch.cuda.synchronize()
        labels = set(_lib.profile_summary())
    finally:
        _lib.PROFILE = None
    assert "spectrum_bin" in labels and any(k.startswith("axis_gemm[metricsW") for k in labels)      # the HIP path ran


def test_dpot_train_then_eval(tmp_path):
    """Same entrypoints with model_name: dpot (the shipped configs/cylinder/dpot_s.yaml key surface at a reduced size; clip_grad_norm 1
    as in the reference YAML; N_autoregressive 2 feeds the 4 predicted frames back as the next window)."""
    from realpdebench_amd import eval as ev
    from realpdebench_amd import train as tr
    with open(os.path.join(os.path.dirname(tr.__file__), "configs", "cylinder", "dpot_s.yaml")) as fh:
        cfg = yaml.safe_load(fh)
    cfg.update(exp_name="d", results_path=str(tmp_path), shape_in=[4, 32, 32, 2], shape_out=[4, 32, 32, 2], n_train=8, n_val=4,
               img_size=32, embed_dim=128, depth=2, in_timesteps=4, out_timesteps=4, num_update=100, train_batch_size=4,
               test_batch_size=4, lr=1e-3, N_autoregressive=2)
    path = tmp_path / "dp.yaml"
    path.write_text(yaml.safe_dump(cfg))
    exp = tr.main(["--config", str(path), "--max_updates", "6"])
    ckpts = sorted(glob.glob(os.path.join(exp, "model_*.pth")))
    ck = torch.load(ckpts[-1], map_location="cpu")
    assert ck["iteration"] == 6 and "dpot_model.blocks.0.filter.w1" in ck["model_state_dict"]
    assert all(l == l and l < 1e3 for l in ck["train_losses"])
    assert ck["train_losses"][-1] < 1.5 * ck["train_losses"][0]
    ev.main(["--config", str(path), "--checkpoint_path", ckpts[-1]])
