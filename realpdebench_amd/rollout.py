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
    if not input.is_cuda:
        raise RuntimeError("autoregressive_rollout runs on MI355X only: move the model and inputs to 'cuda'")
    x = input.contiguous().float()
    B, T, H, W, Cin = x.shape
    g = normalizer if isinstance(normalizer, GaussianNormalizer) else None
    if para_input is not None:
        para_input = para_input.to(x.device).contiguous().float()
    out = steps = nxt = None
    cur = x
    ncell = B * T * H * W
    for i in range(n_autoregressive):
        p = model(cur)                                                   # [B,T_out,H,W,Cp]
        if i == 0:
            Cp = p.shape[-1]
            Cx = Cin - Cp
            if tuple(p.shape[:4]) != (B, T, H, W):
                raise ValueError("autoregression needs T_out == T_in (eval.py:314-319 feeds predictions back)")
            if (Cx > 0) != (para_input is not None):
                raise ValueError("control channels and para_input must come together (eval.py:305-309)")
            out = torch.empty(B, n_autoregressive * T, H, W, Cin, device=x.device, dtype=torch.float32)
            steps = out.view(B, n_autoregressive, T, H, W, Cin)
            nxt = torch.empty(B, T, H, W, Cin, device=x.device, dtype=torch.float32)
        ops.rollout_affine(p.reshape(ncell, Cp).contiguous(), para_input, nxt, ncell, Cp, Cx,
                           g.mean_targets[:Cp].contiguous() if g else None,
                           g.std_targets[:Cp].contiguous() if g else None,
                           g.mean_inputs[:Cin].contiguous() if g else None,
                           g.std_inputs[:Cin].contiguous() if g else None)
        steps[:, i].copy_(nxt)
        cur, nxt = nxt, (torch.empty_like(nxt) if i == 0 else cur)
    return out
