"""Transolver (structured 3-D mesh) on MI355X -- drop-in for the reference's
``realpdebench.model.TRANSOLVER_libs.Transolver_Structured_Mesh_3D.Model`` (``Transolver_Structured_Mesh_3D.py:80-214``),
built by ``load_model`` exactly like ``realpdebench/model/load_model.py:145-152``.

State: **forward / rollout path** (eval mode) in HIP; the backward pass is the next row and
``train_loss`` under autograd raises instead of silently falling back to PyTorch.  Parameter names, shapes and dtypes
equal the reference's ``state_dict`` so its checkpoints load.

Pipeline per block (Transolver_Structured_Mesh_3D.py:71-77, Physics_Attention.py:148-176), tokens channels-last:
LayerNorm -> [both 3x3x3 convolutions as ONE implicit GEMM, N = 2C] -> slice softmax + slice-token sums ->
16-token attention -> deslice -> to_out GEMM (+residual) -> LayerNorm -> MLP GEMM (+GELU) -> GEMM (+residual)
[-> LayerNorm -> head GEMM].
"""
import torch
import torch.nn as nn

from .. import ops
from .model import Model as _ModelBase


class _Lin(nn.Module):
    def __init__(self, fin, fout, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin))
        nn.init.trunc_normal_(self.weight, std=0.02)          # Transolver_Structured_Mesh_3D.py:137-140
        self.bias = nn.Parameter(torch.zeros(fout)) if bias else None


class _LN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class _Conv(nn.Module):
    def __init__(self, c):
        super().__init__()
        k = 1.0 / (c * 27) ** 0.5                               # nn.Conv3d default init family
        self.weight = nn.Parameter(torch.empty(c, c, 3, 3, 3).uniform_(-k, k))
        self.bias = nn.Parameter(torch.empty(c).uniform_(-k, k))


class _MLP(nn.Module):
    def __init__(self, n_in, n_hid, n_out):
        super().__init__()
        self.linear_pre = nn.Sequential(_Lin(n_in, n_hid))
        self.linear_post = _Lin(n_hid, n_out)
        self.linears = nn.ModuleList()


class _Attn(nn.Module):
    def __init__(self, dim, heads, slice_num):
        super().__init__()
        dh = dim // heads
        self.temperature = nn.Parameter(torch.ones(1, heads, 1, 1) * 0.5)
        self.in_project_x = _Conv(dim)
        self.in_project_fx = _Conv(dim)
        self.in_project_slice = _Lin(dh, slice_num)
        nn.init.orthogonal_(self.in_project_slice.weight)      # Physics_Attention.py:141-142
        self.to_q, self.to_k, self.to_v = _Lin(dh, dh, False), _Lin(dh, dh, False), _Lin(dh, dh, False)
        self.to_out = nn.Sequential(_Lin(dim, dim))


class _Block(nn.Module):
    def __init__(self, dim, heads, mlp_ratio, slice_num, last, out_dim):
        super().__init__()
        self.ln_1, self.Attn, self.ln_2 = _LN(dim), _Attn(dim, heads, slice_num), _LN(dim)
        self.mlp = _MLP(dim, dim * mlp_ratio, dim)
        self.last_layer = last
        if last:
            self.ln_3, self.mlp2 = _LN(dim), _Lin(dim, out_dim)


class Transolver(_ModelBase):
    def __init__(self, space_dim=1, n_layers=5, n_hidden=256, dropout=0.0, n_head=8, Time_Input=False, act="gelu",
                 mlp_ratio=1, fun_dim=1, out_dim=1, slice_num=32, ref=8, unified_pos=False, H=32, W=32, D=32):
        super().__init__()
        if Time_Input or unified_pos:
            raise NotImplementedError("load_model hard-codes Time_Input=False, unified_pos=False (load_model.py:148)")
        if act != "gelu":
            raise NotImplementedError(f"act={act!r}: every reference config uses gelu")
        if n_hidden % n_head or n_hidden // n_head != 32 or n_hidden % 64:
            raise ValueError("MI355X Transolver kernels need dim_head == 32 and n_hidden % 64 == 0 "
                             f"(got n_hidden={n_hidden}, n_head={n_head})")
        if slice_num > 32:
            raise ValueError("slice_num <= 32")
        self.H, self.W, self.D = H, W, D
        self.n_hidden, self.n_head, self.slice_num, self.out_dim = n_hidden, n_head, slice_num, out_dim
        self.mlp_ratio, self.dropout_p, self.in_dim = mlp_ratio, dropout, fun_dim + space_dim
        self.preprocess = _MLP(self.in_dim, n_hidden * 2, n_hidden)
        self.blocks = nn.ModuleList([_Block(n_hidden, n_head, mlp_ratio, slice_num, i == n_layers - 1, out_dim)
                                     for i in range(n_layers)])
        self.placeholder = nn.Parameter((1.0 / n_hidden) * torch.rand(n_hidden))
        self._wcat = {}

    # fused, re-laid-out weight of the two convolutions: rows 0..C-1 = in_project_fx, C..2C-1 = in_project_x;
    # column = ((kh*3+kw)*3+kd)*Ci + ci  (what rpb_gemm_nt's implicit-GEMM loader walks)
    def _conv_cat(self, i):
        a = self.blocks[i].Attn
        key = (i, a.in_project_fx.weight._version, a.in_project_x.weight._version, a.in_project_fx.bias._version,
               a.in_project_x.bias._version, str(a.in_project_x.weight.device))
        if self._wcat.get(i, (None,))[0] != key:
            C = self.n_hidden
            w = torch.cat([a.in_project_fx.weight.detach(), a.in_project_x.weight.detach()], dim=0)     # [2C,Ci,3,3,3]
            w = w.permute(0, 2, 3, 4, 1).reshape(2 * C, 27 * C).contiguous()
            b = torch.cat([a.in_project_fx.bias.detach(), a.in_project_x.bias.detach()]).contiguous()
            self._wcat[i] = (key, w, b)
        return self._wcat[i][1], self._wcat[i][2]

    @torch.no_grad()
    def _forward_hip(self, x):
        B = x.shape[0]
        Cin, C, heads, G = x.shape[-1], self.n_hidden, self.n_head, self.slice_num
        ntok = x[0].numel() // Cin
        if ntok != self.H * self.W * self.D:
            raise ValueError(f"{ntok} tokens per sample but H*W*D = {self.H * self.W * self.D}")
        if Cin != self.in_dim:
            raise ValueError(f"expected {self.in_dim} input channels, got {Cin}")
        M = B * ntok
        dev = x.device
        f = dict(device=dev, dtype=torch.float32)
        x2 = x.reshape(M, Cin)
        pre = self.preprocess
        h1 = torch.empty(M, 2 * C, **f)
        ops.tokens_lift(x2, pre.linear_pre[0].weight.data, pre.linear_pre[0].bias.data, h1, M, Cin, 2 * C, True)
        fx = torch.empty(M, C, **f)
        ops.gemm_nt(h1, pre.linear_post.weight.data, fx, M, C, 2 * C, bias=pre.linear_post.bias.data,
                    addvec=self.placeholder.data)                     # Transolver_Structured_Mesh_3D.py:182-183
        del h1
        a = torch.empty(M, C, **f)
        xf = torch.empty(M, 2 * C, **f)
        w = torch.empty(M, heads * G, **f)
        ox = torch.empty(M, C, **f)
        hid = torch.empty(M, C * self.mlp_ratio, **f)
        bps = ops.slice_blocks_per_sample(B)
        tok_part = torch.empty(B * bps, heads * G * 32, **f)
        norm_part = torch.empty(B * bps, heads * G, **f)
        tokS, norm = torch.empty(B, heads * G * 32, **f), torch.empty(B, heads * G, **f)
        tok2 = torch.empty(B, heads * G * 32, **f)
        out = None
        for i, blk in enumerate(self.blocks):
            at = blk.Attn
            ops.layernorm_fwd(fx, blk.ln_1.weight.data, blk.ln_1.bias.data, a, M, C)
            wcat, bcat = self._conv_cat(i)
            ops.gemm_nt(a, wcat, xf, M, 2 * C, 27 * C, bias=bcat, conv=(self.H, self.W, self.D))
            ops.slice_fwd(xf, at.in_project_slice.weight.data, at.in_project_slice.bias.data,
                          at.temperature.data.reshape(-1).contiguous(), w, tok_part, norm_part, B, ntok, heads, G, 2 * C)
            for b in range(B):            # per-sample finish of the block partials (deterministic fp64 sums)
                ops.reduce_partials(tok_part[b * bps:(b + 1) * bps], bps, heads * G * 32, out_f32=tokS[b])
                ops.reduce_partials(norm_part[b * bps:(b + 1) * bps], bps, heads * G, out_f32=norm[b])
            ops.slice_attn(tokS, norm, at.to_q.weight.data, at.to_k.weight.data, at.to_v.weight.data, tok2, B * heads, G)
            ops.deslice_fwd(w, tok2, ox, B, ntok, heads, G)
            ops.gemm_nt(ox, at.to_out[0].weight.data, fx, M, C, C, bias=at.to_out[0].bias.data, residual=fx)
            ops.layernorm_fwd(fx, blk.ln_2.weight.data, blk.ln_2.bias.data, a, M, C)
            ops.gemm_nt(a, blk.mlp.linear_pre[0].weight.data, hid, M, C * self.mlp_ratio, C,
                        bias=blk.mlp.linear_pre[0].bias.data, act=1)
            ops.gemm_nt(hid, blk.mlp.linear_post.weight.data, fx, M, C, C * self.mlp_ratio,
                        bias=blk.mlp.linear_post.bias.data, residual=fx)
            if blk.last_layer:
                ops.layernorm_fwd(fx, blk.ln_3.weight.data, blk.ln_3.bias.data, a, M, C)
                out = torch.empty(M, self.out_dim, **f)
                ops.gemm_nt(a, blk.mlp2.weight.data, out, M, self.out_dim, C, bias=blk.mlp2.bias.data)
        return out.reshape(*x.shape[:-1], self.out_dim)

    def forward(self, x, fx=None, T=None):
        if fx is not None or T is not None:
            raise NotImplementedError("fx / T inputs are never used by the reference's train/eval loops")
        if not x.is_cuda:
            raise RuntimeError("realpdebench_amd.Transolver runs on MI355X only: there is no CPU fallback")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("Transolver backward is not built yet (DESIGN.md section 9): call under "
                                      "torch.no_grad() / model.eval() for inference and rollout")
        if self.training and self.dropout_p > 0:
            raise NotImplementedError("training-mode dropout is part of the (unbuilt) training path; use model.eval()")
        return self._forward_hip(x.contiguous().float())

    def train_loss(self, input, target):
        pred = self.forward(input)            # raises under autograd, see forward()
        return (pred - target) ** 2
