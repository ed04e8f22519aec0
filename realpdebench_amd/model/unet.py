"""U-Net (3-D, video-diffusion style) on MI355X -- drop-in for ``realpdebench.model.unet.Unet3d`` as ``load_model``
builds it (reference realpdebench/model/unet.py:360-571, realpdebench/model/load_model.py:47-58: ``dim = H``, no
conditioning, sparse linear attention on).  Same constructor, ``forward(x[B,T,H,W,C_in]) -> [B,T_out,H,W,C_out]``,
``train_loss`` and the reference's ``state_dict`` key set.

Activations are channels-last token tensors ``[B*T*H*W][C]`` at three resolutions; the reference's
``permute / rearrange`` chains are index arithmetic.  Kernel map (all through the C ABI, ``include/rpb.h``):

* 3x3x3 convolutions, the (1,4,4)/stride-2 down-sampling convolution, the transposed up-sampling convolution (four
  output-parity classes), every 1x1 convolution / ``nn.Linear``: implicit-GEMM modes of ``rpb_gemm_nt`` on fp32 MFMA;
  their weight gradients ``rpb_gemm_tn``; the 7x7x7 ``init_conv`` (C_in = 3) as ``rpb_im2col`` + plain GEMM;
* GroupNorm(8) + time scale/shift + SiLU: ``rpb_chan_stats`` / ``rpb_affine_silu_*`` (per-(sample, channel) passes) around
  ``rpb_gn_affine_{fwd,bwd}`` (group statistics -> per-(sample, channel) affine and back, ``B x C`` numbers); the time-embedding MLPs
  are small ``rpb_gemm_nt`` / ``rpb_gemm_tn`` calls + ``rpb_silu_*``, the relative-position bias ``rpb_relpos_bias_*`` -- no torch
  autograd anywhere inside the model since round 3;
* channel LayerNorm of ``PreNorm``: ``rpb_layernorm_{fwd,bwd}``;
* temporal attention (rotary + T5 relative-position bias): ``rpb_tattn_{fwd,bwd}``; bottleneck softmax attention:
  ``rpb_sattn_{fwd,bwd}``; spatial linear attention: ``rpb_linattn_prep_*`` + ``rpb_head_scores`` / ``rpb_head_apply``.

The backward pass is a reverse walk over a tape of the forward's operations, each with a hand-written HIP backward.
Every multiply-accumulate, normalisation, activation and attention on token-sized tensors is a HIP kernel; torch touches
token-sized data only to copy (skip-connection ``cat`` and its split) and to add two gradients that meet at a fork.  Rotary embedding: the reference imports the third-party,
version-unpinned ``rotary_embedding_torch``; its published algorithm is restated here (``_rotary_tables``) -- parity for
that one function is unpinned, everything around it is pinned through ``oracle/unet_oracle.py``.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .model import Model as _ModelBase

HEADS, DH = 4, 32
HID = HEADS * DH           # 128
GROUPS, GN_EPS, LN_EPS = 8, 1e-5, 1e-5


def _new(*shape, like):
    return torch.empty(*shape, device=like.device, dtype=torch.float32)


def _reduce(part, rows, L, scale=1.0, f64=False):
    out = torch.empty(L, device=part.device, dtype=torch.float64 if f64 else torch.float32)
    ops.reduce_partials(part, rows, L, out_f64=out if f64 else None, out_f32=None if f64 else out, scale=scale)
    return out


def _wgrad(G, A, M, N, K, conv=None, conv_mode=1, ldg=None, lda=None):
    """(dW [N,K], db [N]) = (G^T A(im2col), colsum G) through the TN GEMM + fp64 partial reduction."""
    taps_rev = False
    if conv is not None and conv_mode == 1:
        part, taps_rev = ops.conv3_wgrad_parts(G, A, M, N, K // 27, conv, ldg=ldg, ldx=lda)
        splits = part.shape[0]
    else:
        splits = ops.gemm_tn_splits(M, N, K, conv is not None, conv_mode, ldg=ldg, lda=lda)
        part = _new(splits, N * K + N, like=G)
        ops.gemm_tn(G, A, part, M, N, K, conv=conv, conv_mode=conv_mode, ldg=ldg, lda=lda)
    dW, db = _new(N, K, like=G), _new(N, like=G)
    ops.reduce_partials(part, splits, N * K, out_f32=dW.view(-1), row_stride=N * K + N)
    ops.reduce_partials(part, splits, N, out_f32=db, row_stride=N * K + N, col0=N * K)
    if taps_rev:
        dW = ops.conv3_taps_restore(dW, N, K // 27)
    return dW, db


def _colsum(x, M, N):
    rows = ops.colsum_rows()
    part = _new(rows, N, like=x)
    ops.colsum(x, part, M, N)
    return _reduce(part, rows, N)


class _Tape:
    """Reverse-mode bookkeeping over token tensors: gradients keyed by tensor identity, parameter gradients by name."""

    def __init__(self, record):
        self.record, self.ops, self.g, self.pg = record, [], {}, {}

    def add(self, fn):
        if self.record:
            self.ops.append(fn)

    def acc(self, t, g):
        k = id(t)
        self.g[k] = g if k not in self.g else ops.add(self.g[k], g)

    def grad(self, t):
        return self.g.pop(id(t))

    def pacc(self, name, g):
        self.pg[name] = g if name not in self.pg else self.pg[name] + g


def _rotary_tables(freqs, T):
    """cos / sin [T][32] of rotary_embedding_torch (restated): angle[t][d] = t * freqs[d // 2]."""
    ang = torch.repeat_interleave(torch.arange(T, device=freqs.device, dtype=torch.float32)[:, None] * freqs[None, :], 2, dim=-1)
    return ang.cos().contiguous(), ang.sin().contiguous()


def _rel_pos_index(n, device, num_buckets=32, max_distance=32):
    """unet.py:78-116: the T5 bucket of every (i, j) pair, int32 [n * n] -- integer bookkeeping, computed with the reference's own
    expression; ``rpb_relpos_bias_{fwd,bwd}`` gather / scatter the table through it."""
    q = torch.arange(n, device=device)
    rel = q[None, :] - q[:, None]
    nb = num_buckets // 2
    m = -rel
    ret = (m < 0).long() * nb
    m = m.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(m.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    ret = ret + torch.where(m < max_exact, m, large)
    return ret.reshape(-1).to(torch.int32).contiguous()


class Unet3d(_ModelBase):
    batch_independent = True      # no batch statistics (GroupNorm / LayerNorm): a step may run in micro-batches (trainer.ArenaTrainer)
    def __init__(self, dim, cond_dim=None, out_channels=None, dim_mults=(1, 2, 4, 8), channels=6, attn_heads=4,
                 attn_dim_head=32, use_bert_text_cond=False, init_dim=None, init_kernel_size=7,
                 use_sparse_linear_attn=True, block_type="resnet", resnet_groups=8, out_channel=-1, in_time=10,
                 out_time=10):
        super().__init__()
        if cond_dim is not None or use_bert_text_cond or not use_sparse_linear_attn or init_dim not in (None, dim):
            raise NotImplementedError("load_model builds the unconditioned U-Net with sparse linear attention only "
                                      "(load_model.py:51-58)")
        if attn_heads != HEADS or attn_dim_head != DH or resnet_groups != GROUPS or init_kernel_size % 2 == 0:
            raise NotImplementedError("MI355X U-Net kernels: 4 heads x 32, GroupNorm(8), odd init kernel (the reference defaults)")
        if dim % 64:
            raise ValueError(f"MI355X U-Net kernels need dim (= H of the dataset, load_model.py:52) to be a multiple of 64, got {dim}")
        self.dim, self.channels, self.out_dim = dim, channels, (out_channels if out_channels is not None else channels)
        self.dim_mults, self.in_time, self.out_time, self.ks = tuple(dim_mults), in_time, out_time, init_kernel_size
        dims = [dim] + [dim * m for m in dim_mults]
        self.in_out = list(zip(dims[:-1], dims[1:]))
        self._names = []
        g = torch.Generator().manual_seed(torch.initial_seed() % (2 ** 31))

        def P(name, *shape, fan_in=None, ones=False, zeros=False, normal=False):
            t = torch.empty(*shape)
            if ones:
                t.fill_(1.0)
            elif zeros:
                t.zero_()
            elif normal:
                t.normal_(generator=g)
            else:
                b = 1.0 / math.sqrt(fan_in)
                t.uniform_(-b, b, generator=g)
            self._register(name, nn.Parameter(t))

        def lin(name, fout, fin, bias=True):
            P(name + ".weight", fout, fin, fan_in=fin)
            if bias:
                P(name + ".bias", fout, fan_in=fin)

        def conv(name, co, ci, *k):
            P(name + ".weight", co, ci, *k, fan_in=ci * int(torch.tensor(k).prod()))
            P(name + ".bias", co, fan_in=ci * int(torch.tensor(k).prod()))

        def tattn(name, c):
            self._register(name + ".fn.fn.rotary_emb.freqs",
                           1.0 / (10000 ** (torch.arange(0, DH, 2)[:DH // 2].float() / DH)), buffer=True)
            lin(name + ".fn.fn.to_qkv", 3 * HID, c, bias=False)
            lin(name + ".fn.fn.to_out", c, HID, bias=False)
            P(name + ".norm.gamma", 1, c, 1, 1, 1, ones=True)

        def lattn(name, c):
            P(name + ".fn.to_qkv.weight", 3 * HID, c, 1, 1, fan_in=c)
            conv(name + ".fn.to_out", c, HID, 1, 1)
            P(name + ".norm.gamma", 1, c, 1, 1, 1, ones=True)

        def resnet(name, ci, co, temb=True):
            if temb:
                lin(name + ".mlp.1", 2 * co, 4 * dim)
            for blk, cin in (("block1", ci), ("block2", co)):
                conv(f"{name}.{blk}.proj", co, cin, 3, 3, 3)
                P(f"{name}.{blk}.norm.weight", co, ones=True)
                P(f"{name}.{blk}.norm.bias", co, zeros=True)
            if ci != co:
                conv(name + ".res_conv", co, ci, 1, 1, 1)

        P("time_rel_pos_bias.relative_attention_bias.weight", 32, HEADS, normal=True)
        conv("init_conv", dim, channels, self.ks, self.ks, self.ks)
        tattn("init_temporal_attn.fn", dim)
        lin("time_mlp.1", 4 * dim, dim)
        lin("time_mlp.3", 4 * dim, 4 * dim)
        nres = len(self.in_out)
        for i, (ci, co) in enumerate(self.in_out):
            resnet(f"downs.{i}.0", ci, co)
            resnet(f"downs.{i}.1", co, co)
            lattn(f"downs.{i}.2.fn", co)
            tattn(f"downs.{i}.3.fn", co)
            if i < nres - 1:
                conv(f"downs.{i}.4", co, co, 1, 4, 4)
        mid = dims[-1]
        resnet("mid_block1", mid, mid)
        lin("mid_spatial_attn.fn.fn.fn.to_qkv", 3 * HID, mid, bias=False)
        lin("mid_spatial_attn.fn.fn.fn.to_out", mid, HID, bias=False)
        P("mid_spatial_attn.fn.norm.gamma", 1, mid, 1, 1, 1, ones=True)
        tattn("mid_temporal_attn.fn", mid)
        resnet("mid_block2", mid, mid)
        for i, (ci, co) in enumerate(reversed(self.in_out)):
            resnet(f"ups.{i}.0", co * 2, ci)
            resnet(f"ups.{i}.1", ci, ci)
            lattn(f"ups.{i}.2.fn", ci)
            tattn(f"ups.{i}.3.fn", ci)
            if i < nres - 1:
                P(f"ups.{i}.4.weight", ci, ci, 1, 4, 4, fan_in=ci * 16)          # ConvTranspose3d: [in, out, 1, 4, 4]
                P(f"ups.{i}.4.bias", ci, fan_in=ci * 16)
        resnet("final_conv.0", dim * 2, dim, temb=False)
        conv("final_conv.1", self.out_dim, dim, 1, 1, 1)
        self._wcache = {}

    # ------------------------------------------------------------------ parameter tree with the reference's names
    def _register(self, name, value, buffer=False):
        parts = name.split(".")
        mod = self
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        if buffer:
            mod.register_buffer(parts[-1], value)
        else:
            mod.register_parameter(parts[-1], value)
            self._names.append(name)

    def p(self, name):
        mod = self
        for part in name.split(".")[:-1]:
            mod = mod._modules[part]
        leaf = name.split(".")[-1]
        return mod._parameters[leaf] if leaf in mod._parameters else mod._buffers[leaf]

    def load_state_dict(self, state_dict, strict=True, assign=False):
        # the rotary ``freqs`` entries depend on the installed rotary_embedding_torch version: never strict on them
        own = self.state_dict()
        filt = {k: v for k, v in state_dict.items() if k in own or not k.endswith("rotary_emb.freqs")}
        missing = [k for k in own if k not in filt and not k.endswith("rotary_emb.freqs")]
        unexpected = [k for k in filt if k not in own]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for Unet3d: missing {missing}, unexpected {unexpected}")
        with torch.no_grad():
            for k, v in filt.items():
                if k in own:
                    self.p(k).copy_(torch.as_tensor(v).reshape(self.p(k).shape))
        self._wcache = {}
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    # ------------------------------------------------------------------ weight re-layouts for the implicit GEMMs (cached)
    def _w(self, kind, name):
        w = self.p(name)
        key = (kind, name)
        hit = self._wcache.get(key)
        if hit is not None and hit[0] == w._version and hit[1].device == w.device:
            return hit[1]
        d = w.detach()
        if kind == "conv3":            # [Co,Ci,3,3,3] -> [Co][27*Ci], column = ((kt*3+kh)*3+kw)*Ci + ci
            v = d.permute(0, 2, 3, 4, 1).reshape(d.shape[0], -1)
        elif kind == "conv3_dgrad":    # flipped taps, [Ci][27*Co]
            co, ci = d.shape[:2]
            v = d.permute(0, 2, 3, 4, 1).reshape(co, 27, ci).flip(1).permute(2, 1, 0).reshape(ci, 27 * co)
        elif kind == "down":           # Conv3d [Co,Ci,1,4,4] -> [Co][16*Ci], column = (kh*4+kw)*Ci + ci
            v = d[:, :, 0].permute(0, 2, 3, 1).reshape(d.shape[0], -1)
        elif kind == "down_from_up":   # data gradient of the transposed conv [Ci,Co,1,4,4] = strided conv with W[o=Ci][i=Co]
            v = d[:, :, 0].permute(0, 2, 3, 1).reshape(d.shape[0], -1)
        elif kind in ("up", "up_from_down"):
            # transposed conv with weight [In,Out,1,4,4] (ConvTranspose3d layout; the data gradient of the strided conv
            # [Co,Ci,1,4,4] reads the same way with In = Co): 4 parity classes [Out][4*In], column = (jh*2+jw)*In + i
            cls = []
            for ph in (0, 1):
                for pw in (0, 1):
                    kh = (0, 2) if ph else (1, 3)
                    kw = (0, 2) if pw else (1, 3)
                    sub = d[:, :, 0][:, :, list(kh)][:, :, :, list(kw)]           # [In, Out, jh, jw]
                    cls.append(sub.permute(1, 2, 3, 0).reshape(d.shape[1], -1))
            v = torch.stack(cls)
        elif kind == "T":
            v = d.reshape(d.shape[0], -1).t()
        elif kind == "flat":
            v = d.reshape(d.shape[0], -1)
        elif kind == "init":           # [dim,Cin,k,k,k] -> [dim][ldc] with column = tap*Cin + ci, zero-padded
            k = d.shape[2]
            cols = k * k * k * d.shape[1]
            ldc = (cols + 127) // 128 * 128        # 128-column k blocks of the LDS-tiled weight-gradient kernel
            v = torch.zeros(d.shape[0], ldc, device=d.device)
            v[:, :cols] = d.permute(0, 2, 3, 4, 1).reshape(d.shape[0], cols)
        else:
            raise KeyError(kind)
        v = v.contiguous()
        self._wcache[key] = (w._version, v)
        return v

    # ------------------------------------------------------------------ primitives (forward + taped backward)
    def _linear(self, tp, x, wname, bname, M, N, K, residual=None):
        """y = x W^T + b (+ residual) for nn.Linear / 1x1 convolutions."""
        W = self._w("flat", wname)
        b = self.p(bname).detach() if bname else None
        y = _new(M, N, like=x)
        ops.gemm_nt(x, W, y, M, N, K, bias=b, residual=residual)

        def bwd():
            gy = tp.grad(y)
            if residual is not None:
                tp.acc(residual, gy)
            if N % 4:                                            # e.g. the 3-channel output head
                gp = torch.zeros(M, (N + 3) // 4 * 4, device=gy.device)
                gp[:, :N] = gy
                dW, db = _wgrad(gp, x, M, N, K, ldg=gp.shape[1])
            else:
                dW, db = _wgrad(gy, x, M, N, K)
            gx = _new(M, K, like=x)
            if N % 32 and N <= 32:                               # skinny contraction (3 or 16 output channels: configs/combustion/unet.yaml)
                ops.tokens_lift(gy, self._w("T", wname), torch.zeros(K, device=gy.device), gx, M, N, K, False)
            else:
                ops.gemm_nt(gy, self._w("T", wname), gx, M, K, N)
            tp.pacc(wname, dW.view(self.p(wname).shape))
            if bname:
                tp.pacc(bname, db)
            tp.acc(x, gx)

        tp.add(bwd)
        return y

    def _conv3(self, tp, x, name, B, mesh, Ci, Co):
        """nn.Conv3d(Ci, Co, 3, padding=1) on the (T,H,W) mesh."""
        M = B * mesh[0] * mesh[1] * mesh[2]
        y = _new(M, Co, like=x)
        ops.conv3(x, self._w("conv3", name + ".weight"), y, M, Co, Ci, mesh, bias=self.p(name + ".bias").detach())

        def bwd():
            gy = tp.grad(y)
            dW, db = _wgrad(gy, x, M, Co, 27 * Ci, conv=mesh)
            tp.pacc(name + ".weight", dW.view(Co, 3, 3, 3, Ci).permute(0, 4, 1, 2, 3).contiguous())
            tp.pacc(name + ".bias", db)
            gx = _new(M, Ci, like=x)
            ops.conv3(gy, self._w("conv3_dgrad", name + ".weight"), gx, M, Ci, Co, mesh)
            tp.acc(x, gx)

        tp.add(bwd)
        return y

    def _strided(self, x, Wd, bias, B, mesh_in, Ci, Co):
        """Conv3d((1,4,4), stride (1,2,2), padding (0,1,1)) with gather-layout weight Wd [Co][16*Ci]."""
        T, H, W = mesh_in
        Mo = B * T * (H // 2) * (W // 2)
        y = _new(Mo, Co, like=x)
        ops.gemm_nt(x, Wd, y, Mo, Co, 16 * Ci, bias=bias, conv=mesh_in, conv_mode=2)
        return y

    def _transposed(self, x, Wc, bias, B, mesh_in, Ci, Co):
        """ConvTranspose3d((1,4,4), (1,2,2), (0,1,1)) as 4 parity-class GEMMs; Wc [4][Co][4*Ci]."""
        T, H, W = mesh_in
        M = B * T * H * W
        y = _new(4 * M, Co, like=x)
        for cls in range(4):
            ops.gemm_nt(x, Wc[cls], y, M, Co, 4 * Ci, bias=bias, conv=mesh_in, conv_mode=3, cls=cls)
        return y

    def _down(self, tp, x, name, B, mesh, C):
        T, H, W = mesh
        y = self._strided(x, self._w("down", name + ".weight"), self.p(name + ".bias").detach(), B, mesh, C, C)
        Mo = y.shape[0]

        def bwd():
            gy = tp.grad(y)
            dW, db = _wgrad(gy, x, Mo, C, 16 * C, conv=mesh, conv_mode=2)
            tp.pacc(name + ".weight", dW.view(C, 4, 4, C).permute(0, 3, 1, 2).unsqueeze(2).contiguous())
            tp.pacc(name + ".bias", db)
            tp.acc(x, self._transposed(gy, self._w("up_from_down", name + ".weight"), None, B, (T, H // 2, W // 2), C, C))

        tp.add(bwd)
        return y

    def _up(self, tp, x, name, B, mesh, C):
        T, H, W = mesh
        M = B * T * H * W
        y = self._transposed(x, self._w("up", name + ".weight"), self.p(name + ".bias").detach(), B, mesh, C, C)

        def bwd():
            gy = tp.grad(y)
            # d W[i][o][kh][kw] = sum_in x[in][i] * gy[2*in - 1 + k][o]: the strided gather with the roles swapped
            dW, _ = _wgrad(x, gy, M, C, 16 * C, conv=(T, 2 * H, 2 * W), conv_mode=2)
            tp.pacc(name + ".weight", dW.view(C, 4, 4, C).permute(0, 3, 1, 2).unsqueeze(2).contiguous())
            tp.pacc(name + ".bias", _colsum(gy, 4 * M, C))
            tp.acc(x, self._strided(gy, self._w("down_from_up", name + ".weight"), None, B, (T, 2 * H, 2 * W), C, C))

        tp.add(bwd)
        return y

    def _gn_silu(self, tp, x, name, B, n, C, ss=None, res=None):
        """GroupNorm(8) -> x*(scale+1)+shift -> SiLU (unet.py:200-208).  ``ss``: [B, 2C] scale|shift or None."""
        nblk = ops.chan_blocks(B, n)
        part = _new(nblk, B * 2 * C, like=x)
        ops.chan_stats(x, part, B, n, C)
        sums = _reduce(part, nblk, B * 2 * C, f64=True)                      # [B][2][C] fp64: per-channel sum x, sum x^2
        cnt = float(n * (C // GROUPS))
        gam, bet = self.p(name + ".weight").detach(), self.p(name + ".bias").detach()
        Ad, Bd, stat = _new(B, C, like=x), _new(B, C, like=x), _new(B, GROUPS, 2, like=x)
        ops.gn_affine_fwd(sums, gam, bet, ss, cnt, GN_EPS, Ad, Bd, stat, B, C, GROUPS)   # statistics -> y = A x + Bc (scale|shift folded in)
        y = _new(B * n, C, like=x)
        ops.affine_silu_fwd(x, Ad, Bd, y, B, n, C, res=res)        # (+ res: the identity shortcut of ResnetBlock)

        def bwd():
            gy = tp.grad(y)
            if res is not None:
                tp.acc(res, gy)
            part2 = _new(nblk, B * 2 * C, like=x)
            ops.affine_silu_bwd_reduce(x, gy, Ad, Bd, part2, B, n, C)
            d = _reduce(part2, nblk, B * 2 * C)                               # [B][2][C] = (dL/dA, dL/dBc)
            dgam, dbet, Pc, Qc = (_new(B, C, like=x) for _ in range(4))
            dss = _new(B, 2 * C, like=x) if ss is not None else None
            ops.gn_affine_bwd(d, stat, gam, bet, ss, cnt, dgam, dbet, dss, Pc, Qc, B, C, GROUPS)
            tp.pacc(name + ".weight", _reduce(dgam, B, C))
            tp.pacc(name + ".bias", _reduce(dbet, B, C))
            if ss is not None:
                tp.acc(ss, dss)
            gx = _new(B * n, C, like=x)
            ops.affine_silu_bwd_apply(x, gy, Ad, Bd, Pc, Qc, gx, B, n, C)
            tp.acc(x, gx)

        tp.add(bwd)
        return y

    def _chan_ln(self, tp, x, name, M, C):
        """PreNorm's channel LayerNorm (gamma only, unet.py:169-178)."""
        gam = self.p(name).detach().reshape(C).contiguous()
        y = _new(M, C, like=x)
        ops.layernorm_fwd(x, gam, torch.zeros(C, device=x.device), y, M, C, LN_EPS)

        def bwd():
            gy = tp.grad(y)
            rows = ops.layernorm_bwd_rows(M)
            part = _new(rows, 2 * C, like=x)
            gx = _new(M, C, like=x)
            ops.layernorm_bwd(x, gam, gy, None, gx, part, M, C, LN_EPS)
            tp.pacc(name, _reduce(part, rows, 2 * C)[:C].view(self.p(name).shape))
            tp.acc(x, gx)

        tp.add(bwd)
        return y

    def _temporal_attn(self, tp, x, pre, B, mesh, C, bias):
        """Residual(PreNorm(temporal Attention)) -- unet.py:388-390.  ``bias``: [4,T,T] relative-position bias (shared by every temporal attention)."""
        T, H, W = mesh
        M = B * T * H * W
        y = self._chan_ln(tp, x, pre + "norm.gamma", M, C)
        qkv = self._linear(tp, y, pre + "fn.fn.to_qkv.weight", None, M, 3 * HID, C)
        rc, rs = _rotary_tables(self.p(pre + "fn.fn.rotary_emb.freqs"), T)
        bd = bias.detach().contiguous()
        o = _new(M, HID, like=x)
        ops.tattn_fwd(qkv, rc, rs, bd, o, B, T, H * W)

        def bwd():
            go = tp.grad(o)
            gqkv = _new(M, 3 * HID, like=x)
            rows = ops.tattn_blocks(B * H * W) * 4
            part = _new(rows, T * T, like=x)
            ops.tattn_bwd(qkv, rc, rs, bd, go, gqkv, part, B, T, H * W)
            tp.acc(qkv, gqkv)
            tp.acc(bias, _reduce(part.view(rows // 4, 4 * T * T), rows // 4, 4 * T * T).view(HEADS, T, T))

        tp.add(bwd)
        return self._linear(tp, o, pre + "fn.fn.to_out.weight", None, M, C, HID, residual=x)

    def _mid_spatial_attn(self, tp, x, pre, B, mesh, C):
        """Residual(PreNorm(softmax Attention over h*w)) at the bottleneck -- unet.py:455-457."""
        T, H, W = mesh
        M, Fr, n = B * T * H * W, B * T, H * W
        y = self._chan_ln(tp, x, pre + "norm.gamma", M, C)
        qkv = self._linear(tp, y, pre + "fn.fn.to_qkv.weight", None, M, 3 * HID, C)
        o, lse = _new(M, HID, like=x), _new(Fr * HEADS * n, like=x)
        ops.sattn_fwd(qkv, o, lse, Fr, n)

        def bwd():
            go = tp.grad(o)
            gqkv = _new(M, 3 * HID, like=x)
            ops.sattn_bwd(qkv, o, go, lse, gqkv, Fr, n)
            tp.acc(qkv, gqkv)

        tp.add(bwd)
        return self._linear(tp, o, pre + "fn.fn.to_out.weight", None, M, C, HID, residual=x)

    @staticmethod
    def _diag32(S):
        """[F,2,64,64] products of the 2 x 64 channel pairs -> the four 32 x 32 per-head blocks [F,4,32,32]."""
        Fr = S.shape[0]
        S = S.view(Fr, 2, 2, 32, 2, 32)
        return torch.stack([S[:, 0, 0, :, 0], S[:, 0, 1, :, 1], S[:, 1, 0, :, 0], S[:, 1, 1, :, 1]], dim=1)

    @staticmethod
    def _embed32(Wh):
        """[F,4,32,32] per-head matrices -> block-diagonal [F,2,64,64] operands of rpb_head_apply (nheads = 2)."""
        Fr = Wh.shape[0]
        out = torch.zeros(Fr, 2, 2, 32, 2, 32, device=Wh.device)
        out[:, 0, 0, :, 0], out[:, 0, 1, :, 1], out[:, 1, 0, :, 0], out[:, 1, 1, :, 1] = Wh[:, 0], Wh[:, 1], Wh[:, 2], Wh[:, 3]
        return out.view(Fr, 2, 64, 64)

    def _scores(self, G, ldg, A, lda, Fr, n):
        chunks = ops.head_scores_chunks(Fr, n)
        part = torch.empty(chunks, Fr * 2 * 4096, device=self.p("init_conv.bias").device)
        ops.head_scores(G, ldg, A, lda, part, Fr, n, nheads=2)
        return self._diag32(_reduce(part, chunks, Fr * 2 * 4096).view(Fr, 2, 64, 64))

    def _linear_attn(self, tp, x, pre, B, mesh, C):
        """Residual(PreNorm(SpatialLinearAttention)) -- unet.py:236-261."""
        T, H, W = mesh
        M, Fr, n = B * T * H * W, B * T, H * W
        y = self._chan_ln(tp, x, pre + "norm.gamma", M, C)
        qkv = self._linear(tp, y, pre + "fn.to_qkv.weight", None, M, 3 * HID, C)
        nblk = ops.chan_blocks(Fr, n)
        pm = _new(nblk, Fr * HID, like=x)
        ops.col_reduce(ops.Sub(qkv, HID), 3 * HID, pm, Fr, n, HID, 0)
        kmax = pm.view(nblk, Fr * HID).amax(dim=0).contiguous()                      # partial maxima -> [F,128]
        qe = _new(M, 2 * HID, like=x)
        ops.linattn_prep_fwd(qkv, kmax, qe, Fr, n)
        ops.col_reduce(ops.Sub(qe, HID), 2 * HID, pm, Fr, n, HID, 1)
        Z = _reduce(pm, nblk, Fr * HID).view(Fr, HEADS, DH)                          # sum_n exp(k - kmax)
        S = self._scores(ops.Sub(qe, HID), 2 * HID, ops.Sub(qkv, 2 * HID), 3 * HID, Fr, n)          # E^T v  [F,4,32,32]
        ctx = S / Z[..., None]                                                       # context[d][e]
        o = _new(M, HID, like=x)
        ops.head_apply(qe, 2 * HID, self._embed32(ctx), o, HID, Fr, n, nheads=2)     # out[n][e] = sum_d q'[n][d] ctx[d][e]

        def bwd():
            go = tp.grad(o)
            dqe = _new(M, 2 * HID, like=x)
            ops.head_apply(go, HID, self._embed32(ctx.transpose(-1, -2)), dqe, 2 * HID, Fr, n, nheads=2)     # d q'
            dctx = self._scores(qe, 2 * HID, go, HID, Fr, n)                          # q'^T g  [F,4,32,32]
            dS = dctx / Z[..., None]
            dZ = -(dctx * S).sum(-1) / (Z * Z)                                         # [F,4,32]
            gqkv = _new(M, 3 * HID, like=x)
            ops.head_apply(ops.Sub(qkv, 2 * HID), 3 * HID, self._embed32(dS.transpose(-1, -2)), ops.Sub(dqe, HID), 2 * HID,
                           Fr, n, nheads=2)                                            # d E' = v dS^T
            ops.head_apply(ops.Sub(qe, HID), 2 * HID, self._embed32(dS), ops.Sub(gqkv, 2 * HID), 3 * HID, Fr, n, nheads=2)  # d v
            ops.linattn_prep_bwd(qe, dqe, dZ.reshape(Fr, HID).contiguous(), gqkv, Fr, n)
            tp.acc(qkv, gqkv)

        tp.add(bwd)
        return self._linear(tp, o, pre + "fn.to_out.weight", pre + "fn.to_out.bias", M, C, HID, residual=x)

    def _resnet(self, tp, x, name, B, mesh, Ci, Co, temb):
        n = mesh[0] * mesh[1] * mesh[2]
        ss = None
        if temb is not None and (name + ".mlp.1.weight") in self._names:
            Wl, bl = self.p(name + ".mlp.1.weight").detach(), self.p(name + ".mlp.1.bias").detach()
            TD = temb.shape[1]
            ss = _new(B, 2 * Co, like=x)                                         # scale | shift = Linear(SiLU(t)) (unet.py:223-227); temb
            ops.gemm_nt(temb, Wl, ss, B, 2 * Co, TD, bias=bl)                    # arrives as SiLU(time embedding), computed once

            def bwd_ss():
                g = tp.grad(ss)
                dW, db = _wgrad(g, temb, B, 2 * Co, TD)
                tp.pacc(name + ".mlp.1.weight", dW)
                tp.pacc(name + ".mlp.1.bias", db)
                gt = _new(B, TD, like=x)
                ops.gemm_nt(g, Wl.t().contiguous(), gt, B, TD, 2 * Co)
                tp.acc(temb, gt)

            tp.add(bwd_ss)
        h = self._conv3(tp, x, name + ".block1.proj", B, mesh, Ci, Co)
        h = self._gn_silu(tp, h, name + ".block1.norm", B, n, Co, ss)
        h = self._conv3(tp, h, name + ".block2.proj", B, mesh, Co, Co)
        if Ci != Co:
            h = self._gn_silu(tp, h, name + ".block2.norm", B, n, Co)
            return self._linear(tp, x, name + ".res_conv.weight", name + ".res_conv.bias", B * n, Co, Ci, residual=h)
        return self._gn_silu(tp, h, name + ".block2.norm", B, n, Co, res=x)

    def _cat(self, tp, a, b, Ca, Cb):
        M = a.shape[0]
        y = _new(M, Ca + Cb, like=a)                         # torch.cat((x, skip), dim=1) of unet.py:463,479 as two row-block copies
        ops.copy_cols(a, y, M, Ca, Ca, Ca + Cb, 0, 0)
        ops.copy_cols(b, y, M, Cb, Cb, Ca + Cb, 0, Ca)

        def bwd():
            g = tp.grad(y)
            ga, gb = _new(M, Ca, like=g), _new(M, Cb, like=g)
            ops.copy_cols(g, ga, M, Ca, Ca + Cb, Ca, 0, 0)
            ops.copy_cols(g, gb, M, Cb, Ca + Cb, Cb, Ca, 0)
            tp.acc(a, ga)
            tp.acc(b, gb)

        tp.add(bwd)
        return y

    # ------------------------------------------------------------------ forward / backward drivers
    @torch.no_grad()
    def _forward_hip(self, x, tp):
        B, T, H, W, Cin = x.shape
        dim = self.dim
        if self.out_time > T:
            x = x.repeat(1, self.out_time // T, 1, 1, 1)        # unet.py:520 (input replication along time; no gradient)
            T = x.shape[1]
        f = dict(device=x.device, dtype=torch.float32)
        # ---- relative-position bias of every temporal attention (unet.py:78-116): table gathered through the bucket index map
        tname = "time_rel_pos_bias.relative_attention_bias.weight"
        table = self.p(tname).detach().contiguous()
        ridx = _rel_pos_index(T, x.device)
        bias = torch.empty(HEADS, T, T, **f)
        ops.relpos_bias_fwd(table, ridx, bias, T * T, HEADS)
        # ---- time embedding (unet.py:419-424 with time = 0: sin 0 | cos 0): Linear -> GELU -> Linear, then the SiLU every ResnetBlock's
        #      scale/shift projection starts with (unet.py:223), computed once
        TD = 4 * dim
        emb = torch.cat((torch.zeros(B, dim // 2, **f), torch.ones(B, dim // 2, **f)), -1)
        W1, b1 = self.p("time_mlp.1.weight").detach(), self.p("time_mlp.1.bias").detach()
        W3, b3 = self.p("time_mlp.3.weight").detach(), self.p("time_mlp.3.bias").detach()
        h1, h1pre, tlin, temb = (torch.empty(B, TD, **f) for _ in range(4))
        ops.gemm_nt(emb, W1, h1, B, TD, dim, bias=b1, act=1, pre_out=h1pre)
        ops.gemm_nt(h1, W3, tlin, B, TD, TD, bias=b3)
        ops.silu_fwd(tlin, temb)

        def bwd_glue():                 # first on the tape = last in the reverse walk: every user has accumulated by then
            if id(bias) in tp.g:
                gt_ = torch.empty(32, HEADS, **f)
                ops.relpos_bias_bwd(tp.grad(bias).contiguous(), ridx, gt_, T * T, HEADS, 32)
                tp.pacc(tname, gt_)
            if id(temb) in tp.g:
                gl = torch.empty(B, TD, **f)
                ops.silu_bwd(tlin, tp.grad(temb), gl)
                dW3, db3 = _wgrad(gl, h1, B, TD, TD)
                gh = torch.empty(B, TD, **f)
                ops.gemm_nt(gl, W3.t().contiguous(), gh, B, TD, TD, act=2, aux=h1pre)
                dW1, db1 = _wgrad(gh, emb, B, TD, dim)
                tp.pacc("time_mlp.3.weight", dW3)
                tp.pacc("time_mlp.3.bias", db3)
                tp.pacc("time_mlp.1.weight", dW1)
                tp.pacc("time_mlp.1.bias", db1)

        tp.add(bwd_glue)
        mesh = (T, H, W)
        M = B * T * H * W
        # ---- init conv (7^3, C_in channels) as im2col + GEMM
        Wi = self._w("init", "init_conv.weight")
        ldc = Wi.shape[1]
        col = _new(M, ldc, like=x)
        ops.im2col(x.reshape(M, Cin).contiguous(), col, B, T, H, W, Cin, self.ks, ldc)
        h = _new(M, dim, like=x)
        ops.gemm_nt(col, Wi, h, M, dim, ldc, bias=self.p("init_conv.bias").detach())
        if tp.record:
            h0 = h

            def bwd_init():
                gy = tp.grad(h0)
                dW, db = _wgrad(gy, col, M, dim, ldc)
                k, cols = self.ks, self.ks ** 3 * Cin
                tp.pacc("init_conv.weight", dW[:, :cols].reshape(dim, k, k, k, Cin).permute(0, 4, 1, 2, 3).contiguous())
                tp.pacc("init_conv.bias", db)

            tp.add(bwd_init)
        else:
            del col
        h = self._temporal_attn(tp, h, "init_temporal_attn.fn.", B, mesh, dim, bias)
        r = h
        skips = []
        nres = len(self.in_out)
        for i, (ci, co) in enumerate(self.in_out):
            p = f"downs.{i}."
            h = self._resnet(tp, h, p + "0", B, mesh, ci, co, temb)
            h = self._resnet(tp, h, p + "1", B, mesh, co, co, temb)
            h = self._linear_attn(tp, h, p + "2.fn.", B, mesh, co)
            h = self._temporal_attn(tp, h, p + "3.fn.", B, mesh, co, bias)
            skips.append((h, co))
            if i < nres - 1:
                h = self._down(tp, h, p + "4", B, mesh, co)
                mesh = (mesh[0], mesh[1] // 2, mesh[2] // 2)
        mid = self.in_out[-1][1]
        h = self._resnet(tp, h, "mid_block1", B, mesh, mid, mid, temb)
        h = self._mid_spatial_attn(tp, h, "mid_spatial_attn.fn.", B, mesh, mid)
        h = self._temporal_attn(tp, h, "mid_temporal_attn.fn.", B, mesh, mid, bias)
        h = self._resnet(tp, h, "mid_block2", B, mesh, mid, mid, temb)
        for i, (ci, co) in enumerate(reversed(self.in_out)):
            p = f"ups.{i}."
            sk, cs = skips.pop()
            h = self._cat(tp, h, sk, co, cs)
            h = self._resnet(tp, h, p + "0", B, mesh, co * 2, ci, temb)
            h = self._resnet(tp, h, p + "1", B, mesh, ci, ci, temb)
            h = self._linear_attn(tp, h, p + "2.fn.", B, mesh, ci)
            h = self._temporal_attn(tp, h, p + "3.fn.", B, mesh, ci, bias)
            if i < nres - 1:
                h = self._up(tp, h, p + "4", B, mesh, ci)
                mesh = (mesh[0], mesh[1] * 2, mesh[2] * 2)
        h = self._cat(tp, h, r, dim, dim)
        h = self._resnet(tp, h, "final_conv.0", B, mesh, 2 * dim, dim, None)
        out = self._linear(tp, h, "final_conv.1.weight", "final_conv.1.bias", M, self.out_dim, dim)
        tp.out = out
        return out.view(B, T, H, W, self.out_dim)

    @torch.no_grad()
    def _backward_hip(self, tp, g_out):
        """Gradients of every parameter given dLoss/d(out): reverse walk over the tape."""
        tp.acc(tp.out, g_out.reshape(tp.out.shape).contiguous())
        for fn in reversed(tp.ops):
            fn()
        # the closures on the tape reference the tape (cycles): release the saved activations now, not at the next GC run
        pg = tp.pg
        tp.ops.clear()
        tp.g.clear()
        tp.out = None
        return pg

    # ------------------------------------------------------------------ Model protocol
    def forward(self, x, cond=None, null_cond_prob=0.0, focus_present_mask=None, prob_focus_present=0.0):
        if cond is not None or focus_present_mask is not None or prob_focus_present:
            raise NotImplementedError("conditioning / focus masks are never used by the reference's train/eval loops")
        if not x.is_cuda:
            raise RuntimeError("realpdebench_amd.Unet3d runs on MI355X only: there is no CPU fallback")
        x = x.contiguous().float()
        params = [self.p(n) for n in self._names]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _UnetFunction.apply(x, self, *params)
        return self._forward_hip(x, _Tape(False))

    def train_loss(self, input, target):
        """unet.py:569-571: elementwise mse_loss(pred, target) (callers take .mean())."""
        pred = self.forward(input)
        return (pred - target) ** 2


class _UnetFunction(torch.autograd.Function):
    """Autograd glue: one forward / backward call into the HIP pipelines above."""

    @staticmethod
    def forward(ctx, x, model, *params):
        tp = _Tape(True)
        out = model._forward_hip(x, tp)
        ctx.model, ctx.tp = model, tp
        return out

    @staticmethod
    def backward(ctx, g_out):
        model = ctx.model
        pg = model._backward_hip(ctx.tp, g_out.contiguous().float())
        ctx.tp = None
        return (None, None) + tuple(pg.get(n) for n in model._names)
