"""Autoregressive rollout of the reference's eval loop (realpdebench/eval.py:311-321) on MI355X.

    preds = [input]
    for i in range(N_autoregressive):
        p = model(preds[-1]); _, p = postprocess(preds[-1], p)
        if in_control: p = cat([p, para_input]); p, _ = preprocess(p, target); preds.append(p)
    pred = cat(preds[1:], dim=1)

The post-process / concat / pre-process chain between two steps is one kernel (``rpb_rollout_affine``) that
writes straight into the next step's input buffer; predictions land in one preallocated output tensor.
"""
import torch

from . import ops
from .data_normalizer import GaussianNormalizer


@torch.no_grad()
def autoregressive_rollout(model, input, n_autoregressive, normalizer=None, para_input=None):
    """``input``: pre-processed ``[B,T,H,W,C_in]`` on the device.  Returns normalised predictions
    ``[B, n*T_out, H, W, C_in]`` (control channels included, as in eval.py:321 before the ``[..., :-para_c]`` cut)."""
    model.eval()
    x = model._check_input(input)
    B = x.shape[0]
    T_out, H, W, Cp = model.shape_out
    Cin = model.dim_in
    Cx = Cin - Cp
    if (Cx > 0) != (para_input is not None):
        raise ValueError("control channels and para_input must come together (eval.py:305-309)")
    if model.shape_out[:3] != model.shape_in[:3]:
        raise ValueError("autoregression needs T_out == T_in")
    g = normalizer if isinstance(normalizer, GaussianNormalizer) else None
    if para_input is not None:
        para_input = para_input.to(x.device).contiguous().float()
    out = torch.empty(B, n_autoregressive * T_out, H, W, Cin, device=x.device, dtype=torch.float32)
    steps = out.view(B, n_autoregressive, T_out, H, W, Cin)
    cur = x
    nxt = torch.empty(B, T_out, H, W, Cin, device=x.device, dtype=torch.float32)
    ncell = B * T_out * H * W
    for i in range(n_autoregressive):
        p = model(cur)                                                   # [B,T,H,W,Cp]
        ops.rollout_affine(p.reshape(ncell, Cp), para_input, nxt, ncell, Cp, Cx,
                           g.mean_targets[:Cp].contiguous() if g else None,
                           g.std_targets[:Cp].contiguous() if g else None,
                           g.mean_inputs[:Cin].contiguous() if g else None,
                           g.std_inputs[:Cin].contiguous() if g else None)
        steps[:, i].copy_(nxt)
        cur, nxt = nxt, (torch.empty_like(nxt) if i == 0 else cur)
    return out
