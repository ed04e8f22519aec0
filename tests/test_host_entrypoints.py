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
    assert Trainer(m, lr=1e-2, num_update=10, clip_grad_norm=1.0).clip == 1.0      # train.py:330-331 (tests/test_gpu_fno.py runs it)


def test_galerkin_state_dict_contract_cpu():
    """Reference checkpoint names / shapes / dtypes without a GPU: build on CPU, load the golden
    state dict, read it back; a CPU forward must refuse loudly (no fallback)."""
    import pytest
    from conftest import galerkin_golden, rel_l2
    from realpdebench_amd.model.load_model import load_model
    g = galerkin_golden()
    T, H, W, Cin = g["x"].shape[1:]

    class DS:
        def __getitem__(self, i):
            return torch.zeros(T, H, W, Cin), torch.zeros(*g["shape_out"])

    m = load_model(DS(), device="cpu", model_name="galerkin_transformer", n_hidden=256, n_head=4, dim_feedforward=256,
                   freq_dim=32, fourier_modes_t=2, fourier_modes_x=3, fourier_modes_y=4, norm_eps=1e-7, pos_dim=1,
                   attention_type="galerkin", num_encoder_layers=1, decoder_type="ifft2", num_regressor_layers=1,
                   spacial_fc=True, spacial_dim=3, layer_norm=False, attn_norm=True, regressor_activation="silu")
    assert set(m.state_dict()) == set(g["sd"])
    m.load_state_dict(g["sd"])
    sd = m.state_dict()
    for k, v in g["sd"].items():
        assert sd[k].shape == v.shape and sd[k].dtype == v.dtype, k
        assert float(v.abs().max()) == 0 or rel_l2(sd[k], v) < 1e-7, k
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(g["x"])
    with pytest.raises(NotImplementedError):
        load_model(DS(), device="cpu", model_name="galerkin_transformer", attention_type="fourier")


def test_eval_metrics_matches_reference():
    """Row f3: the truncated-DFT / GEMM-binned eval_metrics equals the reference's fftn + triple-loop implementation on
    vectors generated from it (tests/golden/make_golden_metrics.py), incl. chunked batches and the c < 2 branch."""
    import os
    import numpy as np
    from conftest import GOLDEN_DIR
    from realpdebench_amd.metrics import eval_metrics
    z = np.load(os.path.join(GOLDEN_DIR, "metrics_small.npz"))
    for name in ("a", "b", "c"):
        pred, tgt = torch.from_numpy(z[f"{name}/pred"]), torch.from_numpy(z[f"{name}/target"])
        bs = int(z[f"{name}/bs"]) or None
        vals = eval_metrics(pred, tgt, int(z[f"{name}/c"]), batch_size=bs)
        ref = z[f"{name}/vals"]
        assert len(vals) == 13
        for i, (v, r) in enumerate(zip(vals, ref)):
            if not np.isfinite(r):          # a 1-sample chunk makes R^2 = -inf in the reference too (0 batch variance)
                assert float(v) == r, (name, i, float(v), r)
                continue
            assert abs(float(v) - r) <= 2e-5 * max(abs(r), 1e-3), (name, i, float(v), r)


def test_dpot_registry_state_dict_and_loud_limits():
    """load_model('dpot') with the reference's keyword surface (load_model.py:108-131): the state_dict carries the reference's key set
    (fixture taken from the imported reference), unsupported members of the family are refused by name, and there is no CPU path."""
    import yaml
    from conftest import dpot_golden
    from realpdebench_amd.model.load_model import load_model
    g = dpot_golden()
    cfg = yaml.safe_load(open(os.path.join(ROOT, "realpdebench_amd", "configs", "cylinder", "dpot_s.yaml")))
    cfg.update({k: v for k, v in g["cfg"].items() if k not in ("data_out_channels",)})
    m = load_model([(g["x"][0], g["y"][0])], device="cpu", **cfg)
    assert list(m.state_dict().keys()) == list(g["sd"].keys())
    m.load_state_dict(g["sd"], strict=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(g["x"])
    with pytest.raises(NotImplementedError, match="normalize=True"):
        load_model([(g["x"][0], g["y"][0])], device="cpu", **dict(cfg, normalize=True))
    assert load_model([(torch.zeros(4, 64, 64, 2), torch.zeros(4, 64, 64, 2))], device="cpu", **cfg).needs_resize     # FFT resize path
    with pytest.raises(NotImplementedError, match="data resolution"):                    # 20 x 2 = 40 columns: not a GEMM depth
        load_model([(torch.zeros(4, 20, 20, 2), torch.zeros(4, 20, 20, 2))], device="cpu", **cfg)
    with pytest.raises(NotImplementedError, match="DPOTNet3D"):
        load_model([(g["x"][0], g["y"][0])], device="cpu", **dict(cfg, model_type="dpot3d"))


def test_bench_pmc_csv_to_family_traffic(tmp_path):
    """bench.py's reading of rocprofv3 counter CSVs (roofline.traffic): bytes = request size x request count summed per dispatch, mean
    over a kernel's dispatches, read + write passes combined, family average over one step's launch mix (3 x <1>, 1 x <1,feat>, 3 x <2,WG>:
    round 4's backward launch is the wave-pair variant that also forms the Conv3d weight gradient)."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    names = {"f": "void cmx_kernel<1, false, false, false, false, false, 0, false>(CmxArgs)", "l0": "void cmx_kernel<1, false, true, false, false, false, 0, false>(CmxArgs)",
             "b": "void cmx_kernel<2, false, false, false, true, false, 0, false>(CmxArgs)", "x": "void other_kernel(Args)"}
    def write(path, rows):
        with open(path, "w") as fh:
            fh.write("Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\n")
            for r in rows:
                fh.write(",".join(str(v) if "," not in str(v) else f'"{v}"' for v in r) + "\n")
    rd, wr = tmp_path / "rd.csv", tmp_path / "wr.csv"
    write(rd, [(1, names["f"], "TCC_EA0_RDREQ_128B", 10), (1, names["f"], "TCC_EA0_RDREQ_64B", 2), (1, names["f"], "TCC_EA0_RDREQ", 12),
               (2, names["f"], "TCC_EA0_RDREQ_128B", 14), (3, names["l0"], "TCC_EA0_RDREQ_32B", 8), (4, names["b"], "TCC_EA0_RDREQ_128B", 20),
               (5, names["x"], "TCC_EA0_RDREQ_128B", 999)])
    write(wr, [(1, names["f"], "TCC_EA0_WRREQ_64B", 4), (2, names["f"], "TCC_EA0_WRREQ_64B", 4), (3, names["l0"], "TCC_EA0_WRREQ_64B", 4),
               (4, names["b"], "TCC_EA0_WRREQ_64B", 6), (4, names["b"], "TCC_EA0_WRREQ", 6)])
    per = {}
    for tag, path in (("rd", rd), ("wr", wr)):
        for kn, v in bench.pmc_bytes_per_dispatch(str(path), "cmx_kernel").items():
            per.setdefault(kn, {})[tag] = v
    assert names["x"] not in per
    f_bytes = ((10 * 128 + 2 * 64) + 14 * 128) / 2 + 4 * 64          # mean over the two dispatches + writes
    assert per[names["f"]] == {"rd": ((10 * 128 + 2 * 64) + 14 * 128) / 2, "wr": 256.0}
    fam = bench.cell_mix_family_bytes(per)
    assert fam == pytest.approx((3 * f_bytes + (8 * 32 + 4 * 64) + 3 * (20 * 128 + 6 * 64)) / 7)


def test_shipped_yamls_carry_the_references_values():
    """Every YAML under realpdebench_amd/configs/<scenario>/ for the four north-star models has the reference's key set and values
    (tests/golden/reference_configs.json: yaml.safe_load of realpdebench/configs/**, written by make_golden_configs.py), except the
    documented dataset defaults (synthetic generator, no normaliser, no checkpoint) and the headline file's 128 x 128 shape / batch."""
    import json

    import yaml
    doc = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_configs.json")))
    local = {"dataset_name", "dataset_root", "num_workers", "normalizer", "checkpoint_path", "shape_in", "shape_out", "n_train", "n_val",
             "test_batch_size", "is_use_tb"}
    n = 0
    for scen, per in doc["configs"].items():
        for stem in ("fno", "unet", "trainsolver", "galerkin_transformer"):
            path = os.path.join(ROOT, "realpdebench_amd", "configs", scen, stem + ".yaml")
            assert os.path.exists(path), path
            mine = yaml.safe_load(open(path))
            ref = per[stem]
            missing = [k for k in ref if k not in mine and k not in local]
            assert not missing, (scen, stem, missing)
            for k, v in ref.items():
                if k in local or k not in mine:
                    continue
                assert mine[k] == v or (isinstance(v, float) and abs(mine[k] - v) < 1e-12), (scen, stem, k, mine[k], v)
            assert mine["model_name"] == ref["model_name"]
            n += 1
    assert n == 20
