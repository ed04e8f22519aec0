"""End-to-end parity of the MI355X FNO3d path (through the C ABI) against
  * golden vectors generated from the reference itself (tests/golden/*.npz), and
  * the CPU oracle on freshly seeded inputs at width 64,
with the tolerance BASELINE.json states: fp32 Rel-L2 < 1e-5 on outputs (gradients: 5e-5, see per-assert notes).
"""
import pytest
import torch

from conftest import Golden, rel_l2

pytestmark = pytest.mark.gpu
OUT_TOL = 1e-5
GRAD_TOL = 5e-5


def build(g, sd=None):
    from realpdebench_amd.model.fno import FNO3d
    m = FNO3d(*g.modes, g.n_layers, g.width, g.shape_in, g.shape_out)
    m.load_state_dict(g.sd("sd0") if sd is None else sd)
    return m.cuda()


@pytest.fixture(params=["small_w32", "ctrl_w32"])
def gold(request):
    return Golden(request.param)


def test_width_8_is_rejected_loudly():
    from realpdebench_amd.model.fno import FNO3d
    g = Golden("tiny_w8")
    with pytest.raises(ValueError):
        FNO3d(*g.modes, g.n_layers, g.width, g.shape_in, g.shape_out)


def test_eval_forward_matches_reference(gold):
    m = build(gold).eval()
    with torch.no_grad():
        out = m(gold.t("x0").cuda())
    assert out.shape == gold.t("fwd_eval").shape
    assert rel_l2(out.cpu(), gold.t("fwd_eval")) < OUT_TOL


def test_train_forward_loss_grads_match_reference(gold):
    m = build(gold).train()
    x, y = gold.t("x0").cuda(), gold.t("y0").cuda()
    elem = m.train_loss(x, y)                      # drop-in protocol: elementwise loss, caller takes .mean()
    loss = elem.mean()
    loss.backward()
    assert abs(float(loss) - float(gold.z["loss0"])) < 1e-5 * abs(float(gold.z["loss0"]))
    grads = m.grads_as_state_dict(m.flat.grad)
    ref = gold.sd("grad0")
    assert set(grads) == set(ref)
    for k, gr in ref.items():
        got = grads[k].cpu()
        if k.startswith("convs.") and k.endswith(".bias"):
            assert float((got - gr).abs().max()) < 1e-5, k      # true gradient is 0 (BatchNorm), both sides are noise
        else:
            assert rel_l2(got, gr) < GRAD_TOL, k
    # running statistics were updated like nn.BatchNorm3d does (momentum 0.1, unbiased var)
    with torch.no_grad():
        m.eval()
        sd = m.state_dict()
    assert int(sd["bns.0.num_batches_tracked"]) == int(gold.sd("sd0")["bns.0.num_batches_tracked"]) + 1


def test_two_fused_train_steps_match_reference(gold):
    from realpdebench_amd.trainer import Trainer
    m = build(gold)
    tr = Trainer(m, lr=gold.lr0, num_update=gold.t_max, scheduler="cosine")
    l0 = tr.step(gold.t("x0").cuda(), gold.t("y0").cuda()).clone()
    l1 = tr.step(gold.t("x1").cuda(), gold.t("y1").cuda()).clone()
    assert abs(float(l0) - float(gold.z["loss0"])) < 1e-5 * abs(float(gold.z["loss0"]))
    assert abs(float(l1) - float(gold.z["loss1"])) < 5e-5 * abs(float(gold.z["loss1"]))
    sd, ref = m.state_dict(), gold.sd("sd2")
    for k, v in ref.items():
        got = sd[k].cpu()
        if v.dtype == torch.int64:
            assert int(got) == int(v), k
        elif k.startswith("convs.") and k.endswith(".bias"):
            assert float((got - v).abs().max()) <= 2 * 2 * gold.lr0 * 1.01, k      # Adam on zero-gradient noise
        elif "running_mean" in k:
            assert rel_l2(got, v) < 5e-3, k
        else:
            assert rel_l2(got, v) < 1e-3, k                                         # see tests/test_oracle_golden.py


def test_rollout_matches_reference(gold):
    from realpdebench_amd.data_normalizer import GaussianNormalizer
    from realpdebench_amd.rollout import autoregressive_rollout
    m = build(gold)
    mi, si, mt, st = gold.norm()
    norm = GaussianNormalizer(mi, mt, si, st, "cuda")
    raw = gold.t("x1")
    cin, cout = gold.shape_in[-1], gold.shape_out[-1]
    para = raw[..., cout:].contiguous() if cin != cout else None
    x, _ = norm.preprocess(raw, gold.t("y1"))
    out = autoregressive_rollout(m, x, 3, normalizer=norm, para_input=para)
    assert rel_l2(out.cpu(), gold.t("rollout3")) < 2e-5       # three chained forwards


@pytest.mark.parametrize("width,B", [(64, 2), (128, 1)])
def test_against_oracle_wider(width, B):
    """Width 64 / 128 (the reference's configs) vs the CPU oracle on seeded inputs: forward, loss and grads."""
    from oracle import fno3d_oracle as O
    from realpdebench_amd.model.fno import FNO3d
    torch.manual_seed(7)
    shape = (5, 14, 12, 2)
    modes, L = (2, 4, 4), 2
    sd = O.init_state_dict(modes, L, width, shape, shape, seed=3)
    for l in range(L):
        sd[f"bns.{l}.weight"] = torch.rand(width) + 0.5
        sd[f"bns.{l}.bias"] = torch.randn(width) * 0.2
    x, y = torch.randn(B, *shape), torch.randn(B, *shape)
    loss, pred, grads, new_buf = O.loss_and_grads(sd, x, y, modes, L, shape, shape)
    m = FNO3d(*modes, L, width, shape, shape)
    m.load_state_dict(sd)
    m = m.cuda().train()
    elem = m.train_loss(x.cuda(), y.cuda())
    elem.mean().backward()
    assert abs(float(elem.mean()) - float(loss)) < 1e-5 * float(loss)
    got = m.grads_as_state_dict(m.flat.grad)
    for k, gr in grads.items():
        if k.startswith("convs.") and k.endswith(".bias"):
            continue
        assert rel_l2(got[k].cpu(), gr) < GRAD_TOL, k
    sd.update(new_buf)          # the train-mode forward moved the running statistics on both sides
    msd = m.state_dict()
    for k in new_buf:
        assert rel_l2(msd[k].cpu(), new_buf[k]) < 1e-5, k
    m.eval()
    with torch.no_grad():
        out = m(x.cuda())
    ref, _ = O.fno3d_forward(sd, x, modes, L, shape, shape, training=False)
    assert rel_l2(out.cpu(), ref) < OUT_TOL


def test_state_dict_roundtrip_and_checkpoint(tmp_path, gold):
    m = build(gold)
    sd = m.state_dict()
    ref = gold.sd("sd0")
    assert list(sd.keys()) == list(ref.keys())             # reference key order and names
    for k in ref:
        assert sd[k].dtype == ref[k].dtype and tuple(sd[k].shape) == tuple(ref[k].shape), k
        assert torch.equal(sd[k].cpu(), ref[k]), k
    path = tmp_path / "model_0001.pth"
    torch.save({"model_state_dict": {k: v.cpu() for k, v in sd.items()}, "train_losses": [1.0], "val_losses": {},
                "iteration": 1, "best_iteration": 1, "best_val_loss": 0.5}, path)
    m2 = build(gold, sd=gold.sd("sd2"))
    meta = m2.load_checkpoint(str(path), "cuda")
    assert meta["iteration"] == 1 and meta["best_val_loss"] == 0.5
    assert torch.equal(m2.flat.data, m.flat.data)


def test_fused_trainer_clips_like_clip_grad_norm_(gold):
    """train.py:330-331 with clip_grad_norm > 0: the fused step scales the gradient by min(1, c / (||g|| + 1e-6)) before Adam --
    checked against torch.nn.utils.clip_grad_norm_ + torch.optim.Adam on the reference-layout gradients of the same step."""
    from realpdebench_amd.trainer import Trainer
    m = build(gold)
    tr0 = Trainer(m, lr=gold.lr0, num_update=gold.t_max)
    x, y = gold.t("x0").cuda(), gold.t("y0").cuda()
    tr0.step(x, y)
    gnorm = float(torch.linalg.vector_norm(tr0.grad))
    clip = 0.25 * gnorm                                         # make the clip bite
    m2 = build(gold)
    tr = Trainer(m2, lr=gold.lr0, num_update=gold.t_max, clip_grad_norm=clip)
    before = m2.flat.data.clone()
    tr.step(x, y)
    # reference update: first Adam step with the clipped gradient = lr * sign-like step m_hat / (sqrt(v_hat) + eps)
    g = tr.grad * min(1.0, clip / (gnorm + 1e-6))
    upd = gold.lr0 * g / (g.abs() + 1e-8)
    sel = g.abs() > 1e-6 * float(g.abs().max())                 # skip round-off-level entries (sign is noise)
    assert rel_l2((before - m2.flat.data)[sel].cpu(), upd[sel].cpu()) < 1e-4
    # and the unclipped trainer moved differently where eps matters, identically in direction
    assert abs(float(torch.linalg.vector_norm(tr.grad)) - gnorm) < 1e-5 * gnorm


@pytest.mark.parametrize("W", [32, 40])
def test_eval_last_layer_crop_only_equals_full_lines(W, monkeypatch):
    """Eval forward at width 64 on lines long enough for the matrix-pipe cell_mix: the last layer produces only the crop
    (rpb_cell_mix_eval_crop); the result is the oracle's and bit-equal to the path that computes whole padded lines, in both storages."""
    from oracle import fno3d_oracle as O
    from realpdebench_amd.model.fno import FNO3d
    torch.manual_seed(11)
    shape, modes, L, width, B = (3, 9, W, 2), (2, 4, 8), 3, 64, 2
    sd = O.init_state_dict(modes, L, width, shape, shape, seed=5)
    for l in range(L):
        sd[f"bns.{l}.weight"] = torch.rand(width) + 0.5
        sd[f"bns.{l}.bias"] = torch.randn(width) * 0.2
        sd[f"bns.{l}.running_mean"] = torch.randn(width) * 0.1
        sd[f"bns.{l}.running_var"] = torch.rand(width) + 0.5
    x = torch.randn(B, *shape)
    ref, _ = O.fno3d_forward(sd, x, modes, L, shape, shape, training=False)
    m = FNO3d(*modes, L, width, shape, shape)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    outs = {}
    for storage in ("f32", "bf16"):
        for crop in ("1", "0"):
            monkeypatch.setenv("RPB_EVAL_CROP_LAST", crop)
            m._ws = {}
            m.set_storage(storage)
            with torch.no_grad():
                outs[storage, crop] = m(x.cuda()).float().cpu().clone()
            ws = next(iter(m._ws.values()))
            assert ws.crop_last == (crop == "1")
        assert torch.equal(outs[storage, "1"], outs[storage, "0"]), storage
    # the next layer's forward W stage fused into cell_mix (both storages) against the separate stage
    monkeypatch.setenv("RPB_EVAL_FUSE_W", "0")
    for storage, tol in (("f32", 1e-6), ("bf16", 1e-3)):
        m._ws = {}
        m.set_storage(storage)
        with torch.no_grad():
            sep = m(x.cuda()).float().cpu()
        assert not next(iter(m._ws.values())).fuse_w
        assert rel_l2(outs[storage, "1"], sep) < tol, storage
    m.set_storage("f32")
    assert rel_l2(outs["f32", "1"], ref) < OUT_TOL
    assert rel_l2(outs["bf16", "1"], ref) < 5e-3


def test_eval_forward_from_hipgraph_equals_eager(monkeypatch):
    """RPB_EVAL_GRAPH=1: the eval forward is captured into a hipGraph on the third call with a workspace and replayed afterwards; every
    replay (fresh inputs, parameters updated in place) is bit-equal to the eager launches.  (Off by default: replay is not faster)"""
    import realpdebench_amd.model.fno as F
    torch.manual_seed(3)
    shape, modes, L, width, B = (3, 9, 40, 2), (2, 4, 8), 3, 64, 2
    m = F.FNO3d(*modes, L, width, shape, shape).cuda().eval()
    xs = [torch.randn(B, *shape, device="cuda") for _ in range(5)]
    with torch.no_grad():
        eager = [m(x).clone() for x in xs]
        monkeypatch.setattr(F, "_EVAL_GRAPH", True)
        m._ws = {}
        got = [m(x).clone() for x in xs]
        ws = next(iter(m._ws.values()))
        assert ws.graph is not None
        for a, b in zip(eager, got):
            assert torch.equal(a, b)
        m.flat.data.mul_(1.01)                           # in-place update: the graph reads the same arena
        y_graph = m(xs[0]).clone()
        monkeypatch.setattr(F, "_EVAL_GRAPH", False)
        assert torch.equal(m(xs[0]), y_graph)


def test_eval_last_layer_crop_only_width_128(monkeypatch):
    """Width 128 (configs/fsi/fno.yaml): the last layer's eval cell_mix over the crop only (rpb_cell_mix_eval_crop_c128) is bit-equal to the
    launch over whole padded lines, and the forward is the oracle's."""
    from oracle import fno3d_oracle as O
    from realpdebench_amd.model.fno import FNO3d
    torch.manual_seed(12)
    shape, modes, L, width, B = (3, 9, 40, 2), (2, 4, 8), 3, 128, 2
    sd = O.init_state_dict(modes, L, width, shape, shape, seed=6)
    for l in range(L):
        sd[f"bns.{l}.weight"] = torch.rand(width) + 0.5
        sd[f"bns.{l}.bias"] = torch.randn(width) * 0.2
        sd[f"bns.{l}.running_mean"] = torch.randn(width) * 0.1
        sd[f"bns.{l}.running_var"] = torch.rand(width) + 0.5
    x = torch.randn(B, *shape)
    ref, _ = O.fno3d_forward(sd, x, modes, L, shape, shape, training=False)
    m = FNO3d(*modes, L, width, shape, shape)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    outs = {}
    for crop in ("1", "0"):
        monkeypatch.setenv("RPB_EVAL_CROP_LAST", crop)
        m._ws = {}
        with torch.no_grad():
            outs[crop] = m(x.cuda()).cpu().clone()
        assert next(iter(m._ws.values())).crop_last == (crop == "1")
    assert torch.equal(outs["1"], outs["0"])
    assert rel_l2(outs["1"], ref) < OUT_TOL
