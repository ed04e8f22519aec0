"""Every in-scope reference YAML through the HIP path.

`tests/golden/reference_configs.json` holds the hyper-parameters of the reference's 5 scenarios x {fno, unet, trainsolver (sic),
galerkin_transformer} YAMLs (written by tests/golden/make_golden_configs.py from /root/reference/realpdebench/configs/**; data
only).  For each of the 20:

* `test_native_shape_*`: the WHOLE YAML dict goes through `realpdebench_amd.model.load_model` exactly as train.py:286-290 passes
  it (`**vars(args)`), on a one-sample dataset of the scenario's native shape; one training step of the trainer train.py uses
  (`make_trainer` with the YAML's lr / scheduler / clip) at B = 1, then an eval forward: shapes, finiteness, every parameter that
  should receive a gradient moved, eval determinism.
* `test_oracle_*`: the same constructor path at a size the CPU oracle finishes in seconds (FNO: the native shape itself, all
  layers; Transolver / Galerkin: the native mesh and channel counts with 4 frames; U-Net: dim = H = 64 and the real channel
  counts on a 64 x 16 mesh with 2 frames), loss + eval prediction + parameter gradients against the oracle.
"""
import json
import os

import pytest
import torch

from conftest import GOLDEN_DIR, rel_l2

pytestmark = pytest.mark.gpu

with open(os.path.join(GOLDEN_DIR, "reference_configs.json")) as _fh:
    DOC = json.load(_fh)
SCENARIOS = ("cylinder", "controlled_cylinder", "fsi", "foil", "combustion")
STEMS = ("fno", "unet", "trainsolver", "galerkin_transformer")
MATRIX = [(s, m) for s in SCENARIOS for m in STEMS]


class OneSample:
    """What `load_model` reads: `train_dataset[0]` -> (input [T,H,W,C_in], target [T_out,H,W,C_out])  (load_model.py:7-9)."""

    def __init__(self, shape_in, shape_out):
        self.shape_in, self.shape_out = tuple(shape_in), tuple(shape_out)

    def __len__(self):
        return 1

    def __getitem__(self, i):
        return torch.zeros(self.shape_in), torch.zeros(self.shape_out)


def shapes(scen, T=None, H=None, W=None):
    n = DOC["native_shapes"][scen]
    si, so = list(n["shape_in"]), list(n["shape_out"])
    for i, v in enumerate((T, H, W)):
        if v is not None:
            si[i] = so[i] = v
    return tuple(si), tuple(so)


def build(scen, stem, si, so, **override):
    from realpdebench_amd.model.load_model import load_model
    kw = dict(DOC["configs"][scen][stem])
    kw.update(override)
    if stem == "trainsolver":              # the YAML's D / H / W restate the dataset's mesh; a reduced mesh restates them the same way
        T, H, W = si[:3]
        if (kw["H"] * kw["W"] * kw["D"]) != T * H * W:
            kw["D"] = T
            assert kw["H"] * kw["W"] * kw["D"] == T * H * W
    return load_model(OneSample(si, so), device="cuda", **kw), kw


def test_matrix_is_the_references():
    assert set(DOC["configs"]) == set(SCENARIOS)
    for s in SCENARIOS:
        assert set(STEMS) <= set(DOC["configs"][s])
    assert DOC["configs"]["combustion"]["trainsolver"]["space_dim"] == 16
    assert DOC["native_shapes"]["controlled_cylinder"] == {"shape_in": [10, 64, 128, 5], "shape_out": [10, 64, 128, 3]}


@pytest.mark.parametrize("scen,stem", MATRIX, ids=[f"{s}-{m}" for s, m in MATRIX])
def test_native_shape_train_step_and_eval(scen, stem):
    from realpdebench_amd.trainer import make_trainer
    si, so = shapes(scen)
    torch.manual_seed(MATRIX.index((scen, stem)))
    model, kw = build(scen, stem, si, so)
    before = {n: p.detach().clone() for n, p in model.named_parameters()} if not hasattr(model, "flat") else None
    flat0 = model.flat.detach().clone() if hasattr(model, "flat") else None
    tr = make_trainer(model, lr=kw["lr"], num_update=kw["num_update"], scheduler=kw["scheduler"], step_size=kw["step_size"],
                      clip_grad_norm=kw["clip_grad_norm"])
    x, y = torch.randn(1, *si, device="cuda"), torch.randn(1, *so, device="cuda")
    loss = tr.step(x, y)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss).all()) and 0.1 < float(loss) < 100.0, float(loss)
    if flat0 is not None:
        assert bool(torch.isfinite(model.flat).all())
        moved = (model.flat != flat0).float().mean()
        assert float(moved) > 0.9, float(moved)
    else:
        still = []
        for n, p in model.named_parameters():
            assert bool(torch.isfinite(p).all()), n
            if stem == "galerkin_transformer" and ".attn.fc." in n:      # allocated, never used with pos=None
                assert torch.equal(p, before[n])
            elif torch.equal(p, before[n]):
                still.append(n)
        assert len(still) <= 0.1 * len(before), still                   # Adam moved (nearly) every tensor the loss reaches
    model.eval()
    with torch.no_grad():
        p1 = model(x)
        p2 = model(x)
    assert p1.shape == (1, *so) and bool(torch.isfinite(p1).all())
    assert torch.equal(p1, p2)
    if hasattr(tr, "close"):
        tr.close()


# ---------------------------------------------------------------------------------------------- parity at oracle-sized shapes
@pytest.mark.parametrize("scen", SCENARIOS)
def test_oracle_fno(scen):
    from oracle import fno3d_oracle as O
    from realpdebench_amd.trainer import Trainer
    si, so = shapes(scen)
    cfg = DOC["configs"][scen]["fno"]
    modes, L, width = (cfg["modes1"], cfg["modes2"], cfg["modes3"]), cfg["n_layers"], cfg["width"]
    torch.manual_seed(3)
    sd = O.init_state_dict(modes, L, width, si, so, seed=5)
    x, y = torch.randn(1, *si), torch.randn(1, *so)
    loss_ref, pred_ref, grads, _ = O.loss_and_grads(sd, x, y, modes, L, si, so)
    m, _ = build(scen, "fno", si, so)
    m.load_state_dict(sd)
    tr = Trainer(m, lr=1e-3, num_update=10)
    loss = tr.step(x.cuda(), y.cuda())
    assert abs(float(loss) - float(loss_ref)) < 2e-5 * float(loss_ref), scen
    got = m.grads_as_state_dict(tr.grad)
    for k in ("fc0.weight", "fc0.bias", "spectral_convs.0.weights1", f"spectral_convs.{L - 1}.weights4", "convs.1.weight",
              "fc1.weight", "fc2.weight", "fc2.bias", "bns.0.weight", f"bns.{L - 1}.bias"):
        assert rel_l2(got[k].cpu(), grads[k]) < 2e-4, (scen, k, rel_l2(got[k].cpu(), grads[k]))


@pytest.mark.parametrize("scen", SCENARIOS)
def test_oracle_transolver(scen):
    """space_dim / out_dim 3, 5 -> 3 and 16 -> 16; the YAML's (H, W, D) token reinterpretation at the native H x W with 4 frames."""
    from oracle import transolver_oracle as TO
    si, so = shapes(scen, T=4)
    torch.manual_seed(21)
    m, kw = build(scen, "trainsolver", si, so, dropout=0.0)      # the oracle comparison needs the deterministic step
    x, y = torch.randn(1, *si), torch.randn(1, *so)
    sd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    pred_ref = TO.transolver_forward(sd, x, kw["n_layers"], kw["n_head"], kw["H"], kw["W"], kw["D"], None)
    loss_ref = ((pred_ref - y) ** 2).mean()
    loss_ref.backward()
    m.train()
    loss = m.train_loss(x.cuda(), y.cuda()).mean()
    loss.backward()
    assert abs(float(loss) - float(loss_ref)) < 1e-5 * float(loss_ref)
    bad = {}
    for n, p in m.named_parameters():
        ref = sd[n].grad
        if ref is None or float(ref.norm()) < 1e-12:
            continue
        e = rel_l2(p.grad.cpu(), ref)
        if e > 1e-4:
            bad[n] = e
    assert not bad, bad
    m.eval()
    with torch.no_grad():
        assert rel_l2(m(x.cuda()).cpu(), pred_ref.detach()) < 1e-5


@pytest.mark.parametrize("scen", SCENARIOS)
def test_oracle_galerkin(scen):
    """node_feats 3, 5 and 16 (combustion: the configuration round 5 refused); the YAML's Fourier mode counts at freq_dim 128."""
    from oracle import galerkin_oracle as GO
    si, so = shapes(scen, T=4)
    cfg = DOC["configs"][scen]["galerkin_transformer"]
    modes = (cfg["fourier_modes_t"], cfg["fourier_modes_x"], cfg["fourier_modes_y"])
    torch.manual_seed(11)
    m, kw = build(scen, "galerkin_transformer", si, so)
    with torch.no_grad():
        for l in m.encoder_layers[0].attn.linears:          # the diagonal-dominated init would hide errors in the attention
            l.weight.add_(0.05 * torch.randn_like(l.weight))
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    x, y = torch.randn(1, *si), torch.randn(1, *so)
    m.eval()
    with torch.no_grad():
        pred = m(x.cuda())
    ref, _ = GO.galerkin_forward(sd, x, kw["n_head"], modes, so, eps=kw["norm_eps"])
    assert rel_l2(pred.cpu(), ref) < 2e-5
    m.train()
    m._mask_override = {}                                    # every dropout site off == the oracle without masks
    loss = m.train_loss(x.cuda(), y.cuda()).mean()
    loss.backward()
    loss_ref, _, grads_ref, _ = GO.loss_and_grads(sd, x, y, kw["n_head"], modes, so, eps=kw["norm_eps"])
    assert abs(float(loss) - float(loss_ref)) < 2e-5 * abs(float(loss_ref))
    full = m.grads_as_state_dict({p: p.grad for p in m.parameters()})
    assert set(full) == set(grads_ref)
    for k, g in grads_ref.items():
        if k == "regressor.convs.0.bias":                    # cancelled exactly by the BatchNorm that follows
            assert float(full[k].abs().max()) < 1e-5
            continue
        assert rel_l2(full[k].cpu(), g) < 5e-4, (scen, k, rel_l2(full[k].cpu(), g))


@pytest.mark.parametrize("scen", SCENARIOS)
def test_oracle_unet(scen):
    """dim = H = 64 (load_model.py:52) with channels / out_channels 3 -> 3, 5 -> 3 and 16 -> 16 on a 64 x 16 mesh, 2 frames."""
    from oracle import unet_oracle as UO
    si, so = shapes(scen, T=2, W=16)
    torch.manual_seed(13)
    m, _ = build(scen, "unet", si, so)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("bias") or n.endswith("gamma") or "norm" in n:
                p.add_(0.1 * torch.randn_like(p))
    x, y = torch.randn(1, *si), torch.randn(1, *so)
    m.train()
    loss = m.train_loss(x.cuda(), y.cuda()).mean()
    loss.backward()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    loss_ref, pred_ref, grads_ref = UO.loss_and_grads(sd, x, y)
    assert abs(float(loss.detach()) - float(loss_ref)) < 2e-5 * abs(float(loss_ref))
    named = dict(m.named_parameters())
    worst = max((rel_l2(named[k].grad.cpu(), g) if float(g.abs().max()) > 1e-7 else float(named[k].grad.abs().max()), k)
                for k, g in grads_ref.items())
    assert worst[0] < 1e-3, worst
    m.eval()
    with torch.no_grad():
        assert rel_l2(m(x.cuda()).cpu(), pred_ref) < 2e-5


# ---------------------------------------------------------------------------------------------- the ten DPOT YAMLs
DPOT_MATRIX = [(s, m) for s in SCENARIOS for m in ("dpot_s", "dpot_l")]


@pytest.mark.parametrize("scen,stem", DPOT_MATRIX, ids=[f"{s}-{m}" for s, m in DPOT_MATRIX])
def test_dpot_yaml_native_shape_train_step_and_eval(scen, stem):
    """configs/<scenario>/dpot_{s,l}.yaml through load_model at the scenario's native sample shape (the wrapper resizes to img_size 128
    spectrally, pads the data channels with ones and slices the output channels, model/dpot.py:199-227): one ArenaTrainer step with the
    YAML's optimiser settings at B = 1 and an eval forward.  `checkpoint_path` is the one key not taken from the YAML: it names the
    pretrained weights the reference downloads (no network here) -- random init instead."""
    from realpdebench_amd.trainer import make_trainer
    si, so = shapes(scen)
    torch.manual_seed(7)
    model, kw = build(scen, stem, si, so, checkpoint_path=None)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    tr = make_trainer(model, lr=kw["lr"], num_update=kw["num_update"], scheduler=kw["scheduler"], step_size=kw.get("step_size", 1000),
                      clip_grad_norm=kw.get("clip_grad_norm", 0.0))
    x, y = torch.randn(1, *si, device="cuda"), torch.randn(1, *so, device="cuda")
    loss = tr.step(x, y)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss).all()) and 0.05 < float(loss) < 1e3, float(loss)
    moved = sum(1 for n, p in model.named_parameters() if not torch.equal(p, before[n]))
    assert moved >= 0.8 * len(before), (moved, len(before))          # (cls_head and friends never receive a gradient)
    assert all(bool(torch.isfinite(p).all()) for p in model.parameters())
    model.eval()
    with torch.no_grad():
        p1 = model(x)
        p2 = model(x)
    assert p1.shape == (1, *so) and bool(torch.isfinite(p1).all()) and torch.equal(p1, p2)
    if hasattr(tr, "close"):
        tr.close()
    del tr, model, before
    torch.cuda.empty_cache()
