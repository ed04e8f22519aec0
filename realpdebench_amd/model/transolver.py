"""Transolver (structured 3-D mesh) on MI355X -- drop-in for the reference's
``realpdebench.model.TRANSOLVER_libs.Transolver_Structured_Mesh_3D.Model`` (``Transolver_Structured_Mesh_3D.py:80-214``),
built by ``load_model`` exactly like ``realpdebench/model/load_model.py:145-152``.

Forward, rollout and the whole backward pass (``train_loss(...).mean().backward()`` through one autograd Function) run in
the HIP kernels of ``csrc/`` -- including the training-mode slice-token attention with its dropout mask and its backward
(``rpb_slice_attn_train``); there is no PyTorch fallback and no torch autograd inside the backward.  Parameter names, shapes and dtypes equal the reference's
``state_dict`` so its checkpoints load.

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
    batch_independent = True      # no batch statistics (GroupNorm / LayerNorm): a step may run in micro-batches (trainer.ArenaTrainer)
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
        self._mask_override = None      # tests: list of (attn_mask [B,h,G,G], out_mask [M,C]) per block, already scaled

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

    # ------------------------------------------------------------------ forward (optionally saving for backward)
    @torch.no_grad()
    def _forward_hip(self, x, save=None):
        """``save``: dict that receives every tensor the backward pass needs (training), or None (inference)."""
        B = x.shape[0]
        Cin, C, heads, G = x.shape[-1], self.n_hidden, self.n_head, self.slice_num
        ntok = x[0].numel() // Cin
        if ntok != self.H * self.W * self.D:
            raise ValueError(f"{ntok} tokens per sample but H*W*D = {self.H * self.W * self.D}")
        if Cin != self.in_dim:
            raise ValueError(f"expected {self.in_dim} input channels, got {Cin}")
        M = B * ntok
        f = dict(device=x.device, dtype=torch.float32)
        keep = save is not None
        new = lambda *shape: torch.empty(*shape, **f)
        x2 = x.reshape(M, Cin)
        pre = self.preprocess
        h1 = new(M, 2 * C)
        ops.tokens_lift(x2, pre.linear_pre[0].weight.data, pre.linear_pre[0].bias.data, h1, M, Cin, 2 * C, True)
        fx = new(M, C)
        ops.gemm_nt(h1, pre.linear_post.weight.data, fx, M, C, 2 * C, bias=pre.linear_post.bias.data,
                    addvec=self.placeholder.data)                     # Transolver_Structured_Mesh_3D.py:182-183
        if keep:
            save.update(x2=x2, h1=h1, B=B, ntok=ntok, M=M, blocks=[])
        else:
            del h1
        a = new(M, C)
        xf, w, ox = new(M, 2 * C), new(M, heads * G), new(M, C)
        hid = new(M, C * self.mlp_ratio)
        bps = ops.slice_blocks_per_sample(B)
        tok_part, norm_part = new(B * bps, heads * G * 32), new(B * bps, heads * G)
        tokS, norm, tok2 = new(B, heads * G * 32), new(B, heads * G), new(B, heads * G * 32)
        out = None
        for i, blk in enumerate(self.blocks):
            at = blk.Attn
            st = {}
            if keep:      # fresh buffers per block: everything below is read again by the backward pass
                a, xf, w, ox, hid = new(M, C), new(M, 2 * C), new(M, heads * G), new(M, C), new(M, C * self.mlp_ratio)
                tokS, norm, tok2 = new(B, heads * G * 32), new(B, heads * G), new(B, heads * G * 32)
                st.update(fx0=fx, a1=a, xf=xf, w=w, tokS=tokS, norm=norm, tok2=tok2, ox=ox)
            ops.layernorm_fwd(fx, blk.ln_1.weight.data, blk.ln_1.bias.data, a, M, C)
            wcat, bcat = self._conv_cat(i)
            ops.conv3(a, wcat, xf, M, 2 * C, C, (self.H, self.W, self.D), bias=bcat)
            ops.slice_fwd(xf, at.in_project_slice.weight.data, at.in_project_slice.bias.data,
                          at.temperature.data.reshape(-1).contiguous(), w, tok_part, norm_part, B, ntok, heads, G, 2 * C)
            # per-sample finish of the block partials (deterministic fp64 sums), all samples in one launch each
            ops.reduce_partials_batched(tok_part, B, bps, heads * G * 32, tokS)
            ops.reduce_partials_batched(norm_part, B, bps, heads * G, norm)
            amask = omask = None
            if keep and self.training and self.dropout_p > 0:
                # nn.Dropout(p) on the slice attention map and after to_out (Physics_Attention.py:169,144): inverted-
                # dropout masks drawn with torch's device RNG (the stream necessarily differs from the reference's)
                if self._mask_override is not None:
                    amask, omask = self._mask_override[i]
                else:
                    keep_p = 1.0 - self.dropout_p
                    amask = (torch.rand(B, heads, G, G, **f) < keep_p).float() / keep_p
                    # the token-sized mask after to_out never exists: (seed, keep) expanded by Philox in the GEMM epilogue
                    omask = (int(torch.randint(0, 2 ** 62, (1,))), keep_p)
                # slice-token attention with the dropout mask on the attention map (rpb_slice_attn_train)
                ops.slice_attn_train(tokS, norm, at.to_q.weight.data, at.to_k.weight.data, at.to_v.weight.data,
                                     amask.contiguous(), B * heads, G, out=tok2)
            else:
                ops.slice_attn(tokS, norm, at.to_q.weight.data, at.to_k.weight.data, at.to_v.weight.data, tok2,
                               B * heads, G)
            ops.deslice_fwd(w, tok2, ox, B, ntok, heads, G)
            fx1 = new(M, C) if keep else fx
            ops.gemm_nt(ox, at.to_out[0].weight.data, fx1, M, C, C, bias=at.to_out[0].bias.data, residual=fx,
                        mask=None if isinstance(omask, tuple) else omask, drop=omask if isinstance(omask, tuple) else None)
            if keep:
                st.update(amask=amask, omask=omask)
            a2 = new(M, C) if keep else a
            ops.layernorm_fwd(fx1, blk.ln_2.weight.data, blk.ln_2.bias.data, a2, M, C)
            hpre = new(M, C * self.mlp_ratio) if keep else None
            ops.gemm_nt(a2, blk.mlp.linear_pre[0].weight.data, hid, M, C * self.mlp_ratio, C,
                        bias=blk.mlp.linear_pre[0].bias.data, act=1, pre_out=hpre)
            fx2 = new(M, C) if keep else fx1
            ops.gemm_nt(hid, blk.mlp.linear_post.weight.data, fx2, M, C, C * self.mlp_ratio,
                        bias=blk.mlp.linear_post.bias.data, residual=fx1)
            if keep:
                st.update(fx1=fx1, a2=a2, hpre=hpre, hid=hid, fx2=fx2)
            fx = fx2
            if blk.last_layer:
                a3 = new(M, C) if keep else a
                ops.layernorm_fwd(fx, blk.ln_3.weight.data, blk.ln_3.bias.data, a3, M, C)
                out = new(M, self.out_dim)
                ops.gemm_nt(a3, blk.mlp2.weight.data, out, M, self.out_dim, C, bias=blk.mlp2.bias.data)
                if keep:
                    st.update(a3=a3)
            if keep:
                save["blocks"].append(st)
        return out.reshape(*x.shape[:-1], self.out_dim)

    # ------------------------------------------------------------------ backward
    def _wgrad(self, G, A, M, N, K, ldg=None, lda=None, conv=None):
        """(dW [N,K], db [N]) = (G^T A, colsum G) through the TN GEMM + fp64 partial reduction."""
        taps_rev = False
        if conv is not None:
            part, taps_rev = ops.conv3_wgrad_parts(G, A, M, N, K // 27, conv, ldg=ldg, ldx=lda)
            splits = part.shape[0]
        else:
            splits = ops.gemm_tn_splits(M, N, K, False, ldg=ldg, lda=lda)
            part = torch.empty(splits, N * K + N, device=G.device, dtype=torch.float32)
            ops.gemm_tn(G, A, part, M, N, K, ldg=ldg, lda=lda)
        dW = torch.empty(N, K, device=G.device, dtype=torch.float32)
        db = torch.empty(N, device=G.device, dtype=torch.float32)
        ops.reduce_partials(part, splits, N * K, out_f32=dW.view(-1), row_stride=N * K + N)
        ops.reduce_partials(part, splits, N, out_f32=db, row_stride=N * K + N, col0=N * K)
        if taps_rev:
            dW = ops.conv3_taps_restore(dW, N, K // 27)
        return dW, db

    def _ln_bwd(self, x, ln, gy, gadd, M, C):
        rows = ops.layernorm_bwd_rows(M)
        part = torch.empty(rows, 2 * C, device=x.device, dtype=torch.float32)
        gx = torch.empty(M, C, device=x.device, dtype=torch.float32)
        ops.layernorm_bwd(x, ln.weight.data, gy, gadd, gx, part, M, C)
        dgb = torch.empty(2 * C, device=x.device, dtype=torch.float32)
        ops.reduce_partials(part, rows, 2 * C, out_f32=dgb)
        return gx, dgb[:C].clone(), dgb[C:].clone()

    @torch.no_grad()
    def _backward_hip(self, sv, g_out):
        """Gradients of every parameter given dLoss/d(out) -- autograd of Transolver_Structured_Mesh_3D.py:170-196.
        Returns {parameter: gradient}."""
        C, heads, G, M, B, ntok = self.n_hidden, self.n_head, self.slice_num, sv["M"], sv["B"], sv["ntok"]
        Hm = C * self.mlp_ratio
        f = dict(device=g_out.device, dtype=torch.float32)
        new = lambda *shape: torch.empty(*shape, **f)
        T = lambda wt: wt.data.t().contiguous()
        grads = {}
        g = None
        bps = ops.slice_blocks_per_sample(B)
        for i in range(len(self.blocks) - 1, -1, -1):
            blk, st, at = self.blocks[i], sv["blocks"][i], self.blocks[i].Attn
            if blk.last_layer:
                od = self.out_dim
                go = g_out.reshape(M, od).contiguous()
                gpad = torch.zeros(M, (od + 3) // 4 * 4, **f)               # even leading dimension for float2 loads
                gpad[:, :od] = go
                grads[blk.mlp2.weight], grads[blk.mlp2.bias] = self._wgrad(gpad, st["a3"], M, od, C, ldg=gpad.shape[1])
                ga3 = new(M, C)
                ops.tokens_lift(go, T(blk.mlp2.weight), torch.zeros(C, **f), ga3, M, od, C, False)   # g_out @ W2
                g, grads[blk.ln_3.weight], grads[blk.ln_3.bias] = self._ln_bwd(st["fx2"], blk.ln_3, ga3, None, M, C)
            # ---- MLP (fx2 = fx1 + post(gelu(pre(LN2(fx1)))))
            grads[blk.mlp.linear_post.weight], grads[blk.mlp.linear_post.bias] = self._wgrad(g, st["hid"], M, C, Hm)
            ghp = new(M, Hm)
            ops.gemm_nt(g, T(blk.mlp.linear_post.weight), ghp, M, Hm, C, act=2, aux=st["hpre"])
            grads[blk.mlp.linear_pre[0].weight], grads[blk.mlp.linear_pre[0].bias] = self._wgrad(ghp, st["a2"], M, Hm, C)
            ga2 = new(M, C)
            ops.gemm_nt(ghp, T(blk.mlp.linear_pre[0].weight), ga2, M, C, Hm)
            del ghp
            g1, grads[blk.ln_2.weight], grads[blk.ln_2.bias] = self._ln_bwd(st["fx1"], blk.ln_2, ga2, g, M, C)
            # ---- attention output projection (fx1 = fx0 + to_out(ox))
            g1m = g1
            if st["omask"] is not None:                      # dropout after to_out: gradient flows through the same mask
                g1m = new(M, C)
                if isinstance(st["omask"], tuple):
                    ops.dropout_mul(g1, g1m, M * C, *st["omask"])
                else:
                    ops.mul(g1, st["omask"], g1m, M * C)
            grads[at.to_out[0].weight], grads[at.to_out[0].bias] = self._wgrad(g1m, st["ox"], M, C, C)
            gox = new(M, C)
            ops.gemm_nt(g1m, T(at.to_out[0].weight), gox, M, C, C)
            # ---- deslice backward w.r.t. the attended slice tokens: g_tok2 = sum_n w * g_ox
            tp = new(B * bps, heads * G * 32)
            ops.slice_fwd(gox, None, None, None, None, tp, None, B, ntok, heads, G, C, w_in=st["w"])
            gtok2 = new(B, heads * G * 32)
            ops.reduce_partials_batched(tp, B, bps, heads * G * 32, gtok2)
            # ---- slice-token attention backward (rpb_slice_attn_train with go): gradients w.r.t. the slice-token sums, their
            #      masses and the shared to_q / to_k / to_v weights (per-(b,h) partials summed in fp64)
            gT, gN, gWp = new(B, heads * G * 32), new(B, heads * G), new(B * heads, 3 * 1024)
            am = st["amask"]
            ops.slice_attn_train(st["tokS"], st["norm"], at.to_q.weight.data, at.to_k.weight.data, at.to_v.weight.data,
                                 None if am is None else am.contiguous(), B * heads, G, go=gtok2, gT=gT, gN=gN, gW=gWp)
            gW3 = new(3 * 1024)
            ops.reduce_partials(gWp, B * heads, 3 * 1024, out_f32=gW3)
            grads[at.to_q.weight], grads[at.to_k.weight], grads[at.to_v.weight] = (gW3[i * 1024:(i + 1) * 1024].view(32, 32).clone()
                                                                                   for i in range(3))
            # ---- slice + deslice backward w.r.t. the dual-convolution output
            gxf = new(M, 2 * C)
            sp = new(B * bps, G * 32 + G + heads)
            temp = at.temperature.data.reshape(-1).contiguous()
            ops.slice_bwd(st["xf"], st["w"], gox, st["tok2"], gT, gN,
                          at.in_project_slice.weight.data, temp, gxf, sp, B, ntok, heads, G)
            tot = new(G * 32 + G + heads)
            ops.reduce_partials(sp, B * bps, G * 32 + G + heads, out_f32=tot)
            grads[at.in_project_slice.weight] = tot[:G * 32].view(G, 32).clone()
            grads[at.in_project_slice.bias] = tot[G * 32:G * 32 + G].clone()
            inside = ((temp >= 0.1) & (temp <= 5.0)).float()                    # torch.clamp passes the gradient inside
            grads[at.temperature] = (tot[G * 32 + G:] * inside).view(1, heads, 1, 1)
            # ---- the two convolutions: weight gradient (TN implicit GEMM) and data gradient (flipped-tap implicit GEMM)
            dWc, dbc = self._wgrad(gxf, st["a1"], M, 2 * C, 27 * C, conv=(self.H, self.W, self.D))
            dWc = dWc.view(2 * C, 3, 3, 3, C).permute(0, 4, 1, 2, 3).contiguous()
            grads[at.in_project_fx.weight], grads[at.in_project_x.weight] = dWc[:C].clone(), dWc[C:].clone()
            grads[at.in_project_fx.bias], grads[at.in_project_x.bias] = dbc[:C].clone(), dbc[C:].clone()
            wcat, _ = self._conv_cat(i)
            wflip = wcat.view(2 * C, 27, C).flip(1).permute(2, 1, 0).reshape(C, 27 * 2 * C).contiguous()
            ga1 = new(M, C)
            ops.conv3(gxf, wflip, ga1, M, C, 2 * C, (self.H, self.W, self.D))
            del gxf
            g, grads[blk.ln_1.weight], grads[blk.ln_1.bias] = self._ln_bwd(st["fx0"], blk.ln_1, ga1, g1, M, C)
        # ---- preprocess MLP (+ placeholder): fx0 = post(gelu(pre(x))) + placeholder
        pre = self.preprocess
        dW, db = self._wgrad(g, sv["h1"], M, C, 2 * C)
        grads[pre.linear_post.weight], grads[pre.linear_post.bias], grads[self.placeholder] = dW, db, db.clone()
        Cin = self.in_dim
        h1pre = new(M, 2 * C)
        ops.tokens_lift(sv["x2"], pre.linear_pre[0].weight.data, pre.linear_pre[0].bias.data, h1pre, M, Cin, 2 * C, False)
        gh1 = new(M, 2 * C)
        ops.gemm_nt(g, T(pre.linear_post.weight), gh1, M, 2 * C, C, act=2, aux=h1pre)
        xpad = torch.zeros(M, (Cin + 3) // 4 * 4, **f)
        xpad[:, :Cin] = sv["x2"]
        grads[pre.linear_pre[0].weight], grads[pre.linear_pre[0].bias] = self._wgrad(gh1, xpad, M, 2 * C, Cin,
                                                                                     lda=xpad.shape[1])
        return grads

    def forward(self, x, fx=None, T=None):
        if fx is not None or T is not None:
            raise NotImplementedError("fx / T inputs are never used by the reference's train/eval loops")
        if not x.is_cuda:
            raise RuntimeError("realpdebench_amd.Transolver runs on MI355X only: there is no CPU fallback")
        x = x.contiguous().float()
        params = [p for p in self.parameters()]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _TransolverFunction.apply(x, self, *params)
        return self._forward_hip(x)

    def train_loss(self, input, target):
        """Transolver_Structured_Mesh_3D.py:198-201: elementwise (pred - target)**2 (callers take .mean())."""
        pred = self.forward(input)
        return (pred - target) ** 2


class _TransolverFunction(torch.autograd.Function):
    """Autograd glue: one forward / backward call into the HIP pipelines above."""

    @staticmethod
    def forward(ctx, x, model, *params):
        sv = {}
        out = model._forward_hip(x, save=sv)
        ctx.model, ctx.sv, ctx.params = model, sv, params
        return out

    @staticmethod
    def backward(ctx, g_out):
        grads = ctx.model._backward_hip(ctx.sv, g_out.contiguous().float())
        ctx.sv = None
        return (None, None) + tuple(grads.get(p) for p in ctx.params)
