"""TEST INFRASTRUCTURE ONLY -- CPU (PyTorch fp32) restatement of the reference U-Net:
``Unet3d.forward`` as ``load_model`` builds it (realpdebench/model/load_model.py:47-58: ``dim = H``, ``dim_mults`` from the
YAML, ``channels = C_in``, ``out_channels = C_out``, no conditioning, ``use_sparse_linear_attn=True``).

Follows realpdebench/model/unet.py:
* ``RelativePositionBias`` (T5 buckets, 32 buckets, max_distance 32)   -- :78-116
* channel ``LayerNorm`` (gamma only), ``PreNorm``, ``Residual``            -- :128-135, 169-187
* ``Block`` / ``ResnetBlock`` (3^3 conv, GroupNorm(8), scale/shift, SiLU)   -- :193-234
* ``SpatialLinearAttention``                                              -- :236-261
* ``Attention`` (temporal with rotary + bias; spatial at the bottleneck)  -- :280-356
* ``Unet3d.forward`` (time = 0, prob_focus_present = 0)                    -- :497-567

Weights come as a plain dict keyed by the reference's ``state_dict`` names; the architecture is read off the keys.

Parity status: PINNED by ``tests/golden/unet_small.npz`` (forward, loss and every parameter gradient of the imported
reference, tests/golden/make_golden_unet.py) EXCEPT the rotary embedding: ``rotary_embedding_torch`` is a third-party,
version-unpinned dependency that is absent from /root/reference and from this image (pyproject.toml:47); ``rotary`` below
restates its published algorithm (theta = 10000, freqs = theta^(-2j/dim), positions 0..n-1, interleaved-pair rotation)
and the fixture was generated with that same restatement stubbed in -- **parity unpinned** for that one function.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
import math

import torch
import torch.nn.functional as F

HEADS, DIM_HEAD = 4, 32          # unet.py:369-370 defaults, never overridden by load_model


def rotary(t, freqs):
    """rotary_embedding_torch.RotaryEmbedding.rotate_queries_or_keys (restated, see header).  t: [..., n, d]."""
    n = t.shape[-2]
    pos = torch.arange(n, dtype=freqs.dtype)
    ang = torch.repeat_interleave(pos[:, None] * freqs[None, :], 2, dim=-1)            # [n, d]: (f0,f0,f1,f1,...)
    x = t.reshape(*t.shape[:-1], -1, 2)
    rot = torch.stack((-x[..., 1], x[..., 0]), dim=-1).reshape(t.shape)
    return t * ang.cos() + rot * ang.sin()


def rel_pos_bias(weight, n, num_buckets=32, max_distance=32):
    """unet.py:78-116 -> [heads, n, n]."""
    q = torch.arange(n)
    rel = q[None, :] - q[:, None]
    nb = num_buckets // 2
    m = -rel
    ret = (m < 0).long() * nb
    m = m.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(m.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    ret = ret + torch.where(m < max_exact, m, large)
    return weight[ret].permute(2, 0, 1)


def chan_ln(x, gamma, eps=1e-5):
    var = x.var(dim=1, unbiased=False, keepdim=True)
    return (x - x.mean(dim=1, keepdim=True)) / (var + eps).sqrt() * gamma


def attention(sd, pre, x, pos_bias=None, use_rotary=False):
    """unet.py:296-356 on x [..., n, C] (heads split, scaled q, optional rotary / bias, softmax, to_out without bias)."""
    qkv = x @ sd[pre + "to_qkv.weight"].t()
    q, k, v = (t.reshape(*t.shape[:-1], HEADS, DIM_HEAD).transpose(-2, -3) for t in qkv.chunk(3, dim=-1))
    q = q * DIM_HEAD ** -0.5
    if use_rotary:
        q, k = rotary(q, sd[pre + "rotary_emb.freqs"]), rotary(k, sd[pre + "rotary_emb.freqs"])
    sim = q @ k.transpose(-1, -2)
    if pos_bias is not None:
        sim = sim + pos_bias
    attn = (sim - sim.amax(dim=-1, keepdim=True).detach()).softmax(dim=-1)
    out = (attn @ v).transpose(-2, -3).reshape(*x.shape[:-1], HEADS * DIM_HEAD)
    return out @ sd[pre + "to_out.weight"].t()


def temporal_attn(sd, pre, x, pos_bias):
    """Residual(PreNorm(EinopsToAndFrom('b c f h w', 'b (h w) f c', Attention))) -- unet.py:388-390."""
    y = chan_ln(x, sd[pre + "norm.gamma"]).permute(0, 3, 4, 2, 1)                      # b h w f c
    y = attention(sd, pre + "fn.fn.", y, pos_bias, use_rotary=True)
    return x + y.permute(0, 4, 3, 1, 2)


def mid_spatial_attn(sd, pre, x):
    """Residual(PreNorm(EinopsToAndFrom('b c f h w', 'b f (h w) c', Attention))) -- unet.py:455-457."""
    b, c, f, h, w = x.shape
    y = chan_ln(x, sd[pre + "norm.gamma"]).permute(0, 2, 3, 4, 1).reshape(b, f, h * w, c)
    y = attention(sd, pre + "fn.fn.", y)
    return x + y.reshape(b, f, h, w, c).permute(0, 4, 1, 2, 3)


def spatial_linear_attn(sd, pre, x):
    """Residual(PreNorm(SpatialLinearAttention)) -- unet.py:236-261."""
    b, c, f, h, w = x.shape
    y = chan_ln(x, sd[pre + "norm.gamma"]).permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    qkv = F.conv2d(y, sd[pre + "fn.to_qkv.weight"]).chunk(3, dim=1)
    q, k, v = (t.reshape(b * f, HEADS, DIM_HEAD, h * w) for t in qkv)
    q = q.softmax(dim=-2) * DIM_HEAD ** -0.5
    k = k.softmax(dim=-1)
    context = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", context, q).reshape(b * f, HEADS * DIM_HEAD, h, w)
    out = F.conv2d(out, sd[pre + "fn.to_out.weight"], sd[pre + "fn.to_out.bias"])
    return x + out.reshape(b, f, c, h, w).permute(0, 2, 1, 3, 4)


def block(sd, pre, x, scale_shift=None):
    x = F.conv3d(x, sd[pre + "proj.weight"], sd[pre + "proj.bias"], padding=1)
    x = F.group_norm(x, 8, sd[pre + "norm.weight"], sd[pre + "norm.bias"])
    if scale_shift is not None:
        x = x * (scale_shift[0] + 1) + scale_shift[1]
    return F.silu(x)


def resnet_block(sd, pre, x, t_emb=None):
    ss = None
    if t_emb is not None and (pre + "mlp.1.weight") in sd:
        e = F.silu(t_emb) @ sd[pre + "mlp.1.weight"].t() + sd[pre + "mlp.1.bias"]
        ss = e[:, :, None, None, None].chunk(2, dim=1)
    h = block(sd, pre + "block2.", block(sd, pre + "block1.", x, ss))
    res = F.conv3d(x, sd[pre + "res_conv.weight"], sd[pre + "res_conv.bias"]) if (pre + "res_conv.weight") in sd else x
    return h + res


def unet_forward(sd, x, out_time=None):
    """x: [B,T,H,W,C_in] -> [B,T_out,H,W,C_out]  (unet.py:497-567 with time = 0, no conditioning)."""
    B, T = x.shape[0], x.shape[1]
    out_time = out_time or T
    x = x.permute(0, 4, 1, 2, 3)
    bias = rel_pos_bias(sd["time_rel_pos_bias.relative_attention_bias.weight"], out_time)
    if out_time > T:
        x = x.repeat(1, 1, out_time // T, 1, 1)
    x = F.conv3d(x, sd["init_conv.weight"], sd["init_conv.bias"], padding=sd["init_conv.weight"].shape[-1] // 2)
    x = temporal_attn(sd, "init_temporal_attn.fn.", x, bias)
    r = x
    dim = sd["time_mlp.1.weight"].shape[1]
    emb = torch.cat((torch.zeros(B, dim // 2), torch.ones(B, dim // 2)), dim=-1)       # SinusoidalPosEmb(time = 0)
    t = F.gelu(emb @ sd["time_mlp.1.weight"].t() + sd["time_mlp.1.bias"]) @ sd["time_mlp.3.weight"].t() + sd["time_mlp.3.bias"]
    n_res = len({k.split(".")[1] for k in sd if k.startswith("downs.")})
    skips = []
    for i in range(n_res):
        p = f"downs.{i}."
        x = resnet_block(sd, p + "0.", x, t)
        x = resnet_block(sd, p + "1.", x, t)
        x = spatial_linear_attn(sd, p + "2.fn.", x)
        x = temporal_attn(sd, p + "3.fn.", x, bias)
        skips.append(x)
        if (p + "4.weight") in sd:
            x = F.conv3d(x, sd[p + "4.weight"], sd[p + "4.bias"], stride=(1, 2, 2), padding=(0, 1, 1))
    x = resnet_block(sd, "mid_block1.", x, t)
    x = mid_spatial_attn(sd, "mid_spatial_attn.fn.", x)
    x = temporal_attn(sd, "mid_temporal_attn.fn.", x, bias)
    x = resnet_block(sd, "mid_block2.", x, t)
    for i in range(n_res):
        p = f"ups.{i}."
        x = torch.cat((x, skips.pop()), dim=1)
        x = resnet_block(sd, p + "0.", x, t)
        x = resnet_block(sd, p + "1.", x, t)
        x = spatial_linear_attn(sd, p + "2.fn.", x)
        x = temporal_attn(sd, p + "3.fn.", x, bias)
        if (p + "4.weight") in sd:
            x = F.conv_transpose3d(x, sd[p + "4.weight"], sd[p + "4.bias"], stride=(1, 2, 2), padding=(0, 1, 1))
    x = torch.cat((x, r), dim=1)
    x = resnet_block(sd, "final_conv.0.", x)
    x = F.conv3d(x, sd["final_conv.1.weight"], sd["final_conv.1.bias"])
    return x.permute(0, 2, 3, 4, 1)


def loss_and_grads(sd, x, y):
    """mean((pred - y)^2) (train.py:328) and its gradient w.r.t. every trainable parameter."""
    names = [k for k in sd if not k.endswith("rotary_emb.freqs")]
    leaf = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    pred = unet_forward(leaf, x, y.shape[1])
    loss = ((pred - y) ** 2).mean()
    grads = dict(zip(names, torch.autograd.grad(loss, [leaf[k] for k in names])))
    return loss.detach(), pred.detach(), grads
