"""TEST INFRASTRUCTURE ONLY -- CPU (PyTorch fp32) restatement of the reference FNO3d hot path.

This file is the *checker*, never the product: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  It is a functional restatement
(weights live in a plain ``dict`` keyed by the reference's ``state_dict`` names) of

* ``SpectralConv3d.forward``      -- /root/reference/realpdebench/model/fno.py:41-64
* ``FNO3d.forward`` / ``get_grid`` -- fno.py:105-129, 135-143
* ``FNO3d.train_loss`` + mse_loss  -- fno.py:131-133, realpdebench/utils/metrics.py:11-13
* the training step               -- realpdebench/train.py:321-334 (Adam + CosineAnnealingLR)
* the autoregressive rollout      -- realpdebench/eval.py:311-321
* GaussianNormalizer pre/post      -- realpdebench/data/data_normalizer.py:50-62

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function here against
golden vectors produced by importing the reference itself in the build container
(``tests/golden/make_golden.py``; the reference ships no tests of its own).
"""
import math

import numpy as np
import torch

BN_EPS = 1e-5          # nn.BatchNorm3d default, fno.py:101
BN_MOMENTUM = 0.1      # nn.BatchNorm3d default
PADDING = 6            # fno.py:87


# ----------------------------------------------------------------------------- helpers
def grid_vectors(T, H, W):
    """Per-axis coordinate vectors of ``FNO3d.get_grid`` (fno.py:135-143).

    The reference builds ``np.linspace(0, 1, n)`` in float64 and casts to float32.
    """
    return tuple(torch.tensor(np.linspace(0, 1, n), dtype=torch.float) for n in (T, H, W))


def gelu(x):
    """Exact (erf) GELU == ``F.gelu`` default used at fno.py:119,124."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def spectral_conv3d(x, weights, modes):
    """fno.py:45-64.  x ``[B,Ci,Tp,Hp,Wp]`` real, weights = 4 complex ``[Ci,Co,m1,m2,m3]``."""
    m1, m2, m3 = modes
    B, _, Tp, Hp, Wp = x.shape
    Co = weights[0].shape[1]
    x_ft = torch.fft.rfftn(x, dim=(-3, -2, -1))
    out_ft = torch.zeros(B, Co, Tp, Hp, Wp // 2 + 1, dtype=torch.cfloat)
    corners = [
        (slice(None, m1), slice(None, m2)),      # weights1, fno.py:53
        (slice(-m1, None), slice(None, m2)),     # weights2, fno.py:55
        (slice(None, m1), slice(-m2, None)),     # weights3, fno.py:57
        (slice(-m1, None), slice(-m2, None)),    # weights4, fno.py:59
    ]
    for (st, sh), w in zip(corners, weights):
        blk = x_ft[:, :, st, sh, :m3]
        out_ft[:, :, st, sh, :m3] = torch.einsum("bithw,iothw->bothw", blk, w)
    return torch.fft.irfftn(out_ft, s=(Tp, Hp, Wp))


def batch_norm3d(x, weight, bias, running_mean, running_var, training):
    """nn.BatchNorm3d semantics (fno.py:117): stats over (B,T,H,W) incl. the padded cells.

    Returns ``(y, new_running_mean, new_running_var)``.
    """
    dims = (0, 2, 3, 4)
    shp = (1, -1, 1, 1, 1)
    if training:
        n = x.numel() // x.shape[1]
        mean = x.mean(dim=dims)
        var = ((x - mean.view(shp)) ** 2).mean(dim=dims)            # biased, used to normalise
        new_rm = (1 - BN_MOMENTUM) * running_mean + BN_MOMENTUM * mean.detach()
        new_rv = (1 - BN_MOMENTUM) * running_var + BN_MOMENTUM * var.detach() * (n / (n - 1))
    else:
        mean, var = running_mean, running_var
        new_rm, new_rv = running_mean, running_var
    y = (x - mean.view(shp)) / torch.sqrt(var.view(shp) + BN_EPS) * weight.view(shp) + bias.view(shp)
    return y, new_rm, new_rv


# ----------------------------------------------------------------------------- model
def fno3d_forward(sd, x, modes, n_layers, shape_in, shape_out, training=False):
    """FNO3d.forward, fno.py:105-129.  ``sd`` = reference-named state dict (plain tensors).

    Returns ``(out [B,T_out,H,W,C_out], new_buffers)``; ``new_buffers`` holds the updated BN
    running statistics when ``training`` (the reference mutates its modules in place).
    """
    B, T, H, W, _ = x.shape
    gt, gh, gw = grid_vectors(T, H, W)
    grid = torch.stack([
        gt.view(1, T, 1, 1).expand(B, T, H, W),
        gh.view(1, 1, H, 1).expand(B, T, H, W),
        gw.view(1, 1, 1, W).expand(B, T, H, W)], dim=-1)
    h = torch.cat([x, grid], dim=-1) @ sd["fc0.weight"].t() + sd["fc0.bias"]       # fno.py:107-108
    h = h.permute(0, 4, 1, 2, 3)                                                    # fno.py:109
    h = torch.nn.functional.pad(h, [0, PADDING, 0, PADDING, 0, PADDING])            # fno.py:111
    new_buf = {}
    for l in range(n_layers):
        ws = [sd[f"spectral_convs.{l}.weights{k}"] for k in (1, 2, 3, 4)]
        x1 = spectral_conv3d(h, ws, modes)
        wc = sd[f"convs.{l}.weight"].reshape(sd[f"convs.{l}.weight"].shape[0], -1)  # [Co,Ci]
        x2 = torch.einsum("oi,bithw->bothw", wc, h) + sd[f"convs.{l}.bias"].view(1, -1, 1, 1, 1)
        h, rm, rv = batch_norm3d(x1 + x2, sd[f"bns.{l}.weight"], sd[f"bns.{l}.bias"],
                                 sd[f"bns.{l}.running_mean"], sd[f"bns.{l}.running_var"], training)
        if training:
            new_buf[f"bns.{l}.running_mean"] = rm
            new_buf[f"bns.{l}.running_var"] = rv
            new_buf[f"bns.{l}.num_batches_tracked"] = sd[f"bns.{l}.num_batches_tracked"] + 1
        if l < n_layers - 1:
            h = gelu(h)                                                             # fno.py:118-119
    h = h[..., :-PADDING, :-PADDING, :-PADDING].permute(0, 2, 3, 4, 1)              # fno.py:121-122
    h = gelu(h @ sd["fc1.weight"].t() + sd["fc1.bias"])                             # fno.py:123-124
    h = h @ sd["fc2.weight"].t() + sd["fc2.bias"]                                   # fno.py:125
    r = shape_out[0] // shape_in[0]
    h = h.reshape(*h.shape[:-1], shape_out[-1], r)                                  # fno.py:127
    out = h.permute(0, 1, 5, 2, 3, 4).reshape(B, *shape_out)                        # fno.py:128
    return out, new_buf


PARAM_SUFFIXES = ("weight", "bias", "weights1", "weights2", "weights3", "weights4")


def is_param(name):
    return name.rsplit(".", 1)[-1] in PARAM_SUFFIXES


def cosine_lr(lr0, k, t_max, eta_min=0.0):
    """Closed form of CosineAnnealingLR after ``k`` scheduler steps (train.py:294)."""
    return eta_min + (lr0 - eta_min) * (1 + math.cos(math.pi * k / t_max)) / 2


def loss_and_grads(sd, x, y, modes, n_layers, shape_in, shape_out):
    """``model.train_loss(input,target).mean(); loss.backward()`` -- train.py:328-329."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if is_param(k)}
    full = dict(sd)
    full.update(leaves)
    pred, new_buf = fno3d_forward(full, x, modes, n_layers, shape_in, shape_out, training=True)
    elem = (pred - y) ** 2                                # mse_loss(reduction='none'), metrics.py:11-13
    loss = elem.mean()
    loss.backward()
    grads = {k: v.grad.detach() for k, v in leaves.items()}
    return loss.detach(), pred.detach(), grads, new_buf


def adam_update(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (train.py:290); complex tensors are treated as 2x real."""
    if p.is_complex():
        p, g, m, v = (torch.view_as_real(t) for t in (p, g, m, v))
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def train_steps(sd, batches, modes, n_layers, shape_in, shape_out, lr0, t_max, start_iter=1, stamps=None):
    """Reference hot loop, train.py:321-334: zero_grad, fwd, mean loss, bwd, Adam, cosine step.

    ``sd`` is updated in place.  Returns the list of losses.  ``stamps``: optional list that receives ``time.time()``
    after every completed step (bench.py's CPU baseline times steady-state steps with it).
    """
    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in sd.items() if is_param(k)}
    losses = []
    for it, (x, y) in enumerate(batches, start=start_iter):
        loss, _, grads, new_buf = loss_and_grads(sd, x, y, modes, n_layers, shape_in, shape_out)
        lr = cosine_lr(lr0, it - 1, t_max)
        for k, g in grads.items():
            adam_update(sd[k], g, state[k][0], state[k][1], it - start_iter + 1, lr)
        sd.update(new_buf)
        losses.append(float(loss))
        if stamps is not None:
            import time
            stamps.append(time.time())
    return losses


def gaussian_preprocess(x, mean, std):
    """data_normalizer.py:50-55 (input half)."""
    c = x.shape[-1]
    return (x - mean[..., :c]) / std[..., :c]


def gaussian_postprocess(y, mean, std):
    """data_normalizer.py:57-62 (target half)."""
    c = y.shape[-1]
    return y * std[..., :c] + mean[..., :c]


def rollout(sd, x, n_ar, modes, n_layers, shape_in, shape_out, norm=None, para_input=None):
    """eval.py:311-321.  ``x`` is the already pre-processed input.  ``norm`` =
    ``(mean_in, std_in, mean_tgt, std_tgt)`` or ``None`` for the identity normaliser.
    Returns the normalised predictions concatenated over time ``[B, n_ar*T_out, H, W, C]``.
    """
    preds = [x]
    for _ in range(n_ar):
        p, _ = fno3d_forward(sd, preds[-1], modes, n_layers, shape_in, shape_out, training=False)
        if norm is not None:
            p = gaussian_postprocess(p, norm[2], norm[3])
        if para_input is not None:
            p = torch.cat([p, para_input], dim=-1)
        if norm is not None:
            p = gaussian_preprocess(p, norm[0], norm[1])
        preds.append(p)
    return torch.cat(preds[1:], dim=1)


def init_state_dict(modes, n_layers, width, shape_in, shape_out, seed=0):
    """Random weights with the reference's shapes/dtypes/init family (fno.py:30-38, nn.Linear /
    nn.Conv3d default init) -- used for synthetic benchmarks where no checkpoint exists."""
    g = torch.Generator().manual_seed(seed)
    cin = shape_in[-1]
    dim_out = shape_out[-1] * shape_out[0] // shape_in[0]
    sd = {}

    def lin(name, fo, fi, extra=()):
        bound = 1.0 / math.sqrt(fi)
        sd[f"{name}.weight"] = (torch.rand(fo, fi, *extra, generator=g) * 2 - 1) * bound
        sd[f"{name}.bias"] = (torch.rand(fo, generator=g) * 2 - 1) * bound

    lin("fc0", width, cin + 3)
    scale = 1.0 / (width * width)
    for l in range(n_layers):
        for k in (1, 2, 3, 4):
            re = torch.rand(width, width, *modes, generator=g)
            im = torch.rand(width, width, *modes, generator=g)
            sd[f"spectral_convs.{l}.weights{k}"] = scale * torch.complex(re, im)
        lin(f"convs.{l}", width, width, (1, 1, 1))
        sd[f"bns.{l}.weight"] = torch.ones(width)
        sd[f"bns.{l}.bias"] = torch.zeros(width)
        sd[f"bns.{l}.running_mean"] = torch.zeros(width)
        sd[f"bns.{l}.running_var"] = torch.ones(width)
        sd[f"bns.{l}.num_batches_tracked"] = torch.tensor(0)
    lin("fc1", 128, width)
    lin("fc2", dim_out, 128)
    return sd
