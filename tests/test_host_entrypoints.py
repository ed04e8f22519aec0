"""Host logic that runs without a GPU: YAML/args merge, config fallback, registry errors, LR schedule, datasets."""
import argparse
import math
import os

import pytest
import torch

from conftest import ROOT


def test_yaml_merge_matches_reference_semantics(tmp_path):
    from realpdebench_amd.utils import add_args_from_config, resolve_config
    cfg = resolve_config("configs/cylinder/fno.yaml")
    assert os.path.exists(cfg)
    args = argparse.Namespace(config=cfg, gpu=3)                 # CLI value wins over the YAML's gpu: 0
    args = add_args_from_config(args)
    assert args.gpu == 3 and args.model_name == "fno" and (args.modes1, args.modes2, args.modes3) == (4, 12, 16)
    assert args.is_use_tb is None and args.width == 64 and args.N_autoregressive == 10


def test_load_model_registry_and_shapes():
    from realpdebench_amd.data import SyntheticDataset
    from realpdebench_amd.model import load_model
    ds = SyntheticDataset((4, 8, 8, 5), (4, 8, 8, 3), 2)
    m = load_model(ds, device="cpu", model_name="fno", modes1=2, modes2=2, modes3=3, n_layers=1, width=32, unused_key=1)
    assert m.shape_in == (4, 8, 8, 5) and m.shape_out == (4, 8, 8, 3) and m.dim_out == 3
    with pytest.raises(ValueError):
        load_model(ds, model_name="dmd")
    x, y = ds[1]
    x2, _ = ds[1]
    assert torch.equal(x, x2) and x.shape == (4, 8, 8, 5) and y.shape == (4, 8, 8, 3)


def test_lr_schedules_match_torch():
    from realpdebench_amd.model.fno import FNO3d
    from realpdebench_amd.trainer import Trainer
    m = FNO3d(2, 2, 3, 1, 32, (4, 8, 8, 2), (4, 8, 8, 2))
    for sched in ("cosine", "step"):
        tr = Trainer(m, lr=1e-2, num_update=20, scheduler=sched, step_size=5)
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([p], lr=1e-2)
        ts = (torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=20) if sched == "cosine"
              else torch.optim.lr_scheduler.StepLR(opt, step_size=5, gamma=0.5))
        for k in range(20):
            assert math.isclose(tr.current_lr(), opt.param_groups[0]["lr"], rel_tol=1e-6, abs_tol=1e-12), (sched, k)
            tr.iteration += 1
            opt.step()
            ts.step()
    with pytest.raises(ValueError):
        Trainer(m, lr=1e-2, num_update=10, scheduler="linear")
    with pytest.raises(NotImplementedError):
        Trainer(m, lr=1e-2, num_update=10, clip_grad_norm=1.0)
