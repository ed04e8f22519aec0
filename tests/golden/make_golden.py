"""Generate golden vectors by IMPORTING the reference (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
Needs /root/reference on disk; the GPU box never runs this -- it only reads the ``.npz`` files
written next to this script.  A fixture holds data only: weights, inputs and the outputs the
reference produced for them (forward train/eval, loss, every parameter gradient, weights after
two Adam + CosineAnnealingLR steps, BN running statistics, a 3-step autoregressive rollout under
the Gaussian normaliser).  No reference source text is stored.
"""
import copy
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
from realpdebench.model.fno import FNO3d            # noqa: E402
from realpdebench.utils.metrics import mse_loss     # noqa: E402,F401  (used via train_loss)

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: B, (T,H,W,Cin), (T_out,H,W,Cout), modes, width, layers
    "tiny_w8": dict(B=2, shape_in=(6, 10, 12, 2), shape_out=(6, 10, 12, 2), modes=(2, 3, 4), width=8, n_layers=2),
    "small_w32": dict(B=2, shape_in=(6, 10, 12, 2), shape_out=(6, 10, 12, 2), modes=(2, 3, 3), width=32, n_layers=2),
    "ctrl_w32": dict(B=3, shape_in=(4, 9, 7, 5), shape_out=(4, 9, 7, 3), modes=(2, 2, 3), width=32, n_layers=3),
}
LR0, T_MAX = 1e-3, 10


def sd_to_np(sd, prefix):
    return {f"{prefix}/{k}": v.detach().cpu().numpy().copy() for k, v in sd.items()}


def make(name, cfg):
    torch.manual_seed(1234)
    np.random.seed(1234)
    model = FNO3d(*cfg["modes"], cfg["n_layers"], cfg["width"], cfg["shape_in"], cfg["shape_out"])
    # make BN affine / running stats non-trivial so that eval-mode parity is meaningful
    with torch.no_grad():
        for bn in model.bns:
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
            bn.running_mean.uniform_(-0.2, 0.2)
            bn.running_var.uniform_(0.5, 1.5)
    out = {}
    out.update(sd_to_np(model.state_dict(), "sd0"))
    B = cfg["B"]
    xs = [torch.randn(B, *cfg["shape_in"]) for _ in range(2)]
    ys = [torch.randn(B, *cfg["shape_out"]) for _ in range(2)]
    for i in range(2):
        out[f"x{i}"] = xs[i].numpy()
        out[f"y{i}"] = ys[i].numpy()

    # eval-mode forward (running statistics) -- eval.py:315 / train.py:361
    model.eval()
    with torch.no_grad():
        out["fwd_eval"] = model(xs[0]).numpy()

    # rollout under a Gaussian normaliser -- eval.py:311-321 + data_normalizer.py:50-62
    cin, cout = cfg["shape_in"][-1], cfg["shape_out"][-1]
    mean_in = torch.linspace(-0.2, 0.3, cin)
    std_in = torch.linspace(0.8, 1.4, cin)
    mean_tg = torch.linspace(0.1, -0.1, cout)
    std_tg = torch.linspace(1.2, 0.7, cout)
    out.update(mean_in=mean_in.numpy(), std_in=std_in.numpy(), mean_tg=mean_tg.numpy(), std_tg=std_tg.numpy())
    raw = xs[1]
    para = raw[..., cout:] if cin != cout else None
    with torch.no_grad():
        inp = (raw - mean_in) / std_in
        preds = [inp]
        for _ in range(3):
            p = model(preds[-1])
            p = p * std_tg + mean_tg
            if para is not None:
                p = torch.cat([p, para], dim=-1)
            p = (p - mean_in[..., :p.shape[-1]]) / std_in[..., :p.shape[-1]]
            preds.append(p)
        out["rollout3"] = torch.cat(preds[1:], dim=1).numpy()

    # two training steps: train.py:321-334
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=LR0)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=T_MAX)
    for i in range(2):
        opt.zero_grad()
        elem = model.train_loss(xs[i], ys[i])
        loss = elem.mean()
        loss.backward()
        if i == 0:
            with torch.no_grad():
                # train-mode prediction from a side copy, so the model's running stats stay untouched
                out["fwd_train"] = copy.deepcopy(model)(xs[0]).numpy()
            for k, p in model.named_parameters():
                out[f"grad0/{k}"] = p.grad.detach().numpy().copy()
        out[f"loss{i}"] = np.float64(loss.item())
        opt.step()
        sched.step()
        if i == 1:
            out.update(sd_to_np(model.state_dict(), "sd2"))
    out["lr0"] = np.float64(LR0)
    out["t_max"] = np.int64(T_MAX)
    for k in ("B", "width", "n_layers"):
        out[f"cfg/{k}"] = np.int64(cfg[k])
    for k in ("shape_in", "shape_out", "modes"):
        out[f"cfg/{k}"] = np.asarray(cfg[k], dtype=np.int64)
    path = os.path.join(HERE, f"fno3d_{name}.npz")
    np.savez_compressed(path, **out)
    print(name, "->", path, f"{os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    torch.set_num_threads(4)
    for n, c in CASES.items():
        make(n, c)
