"""FNO3d on MI355X -- drop-in for ``realpdebench.model.fno.FNO3d`` (reference realpdebench/model/fno.py:66-143).

Same constructor, ``forward(x[B,T,H,W,C_in]) -> [B,T_out,H,W,C_out]``, ``train_loss`` and reference-named
``state_dict``; everything between input and output runs in the HIP kernels of ``csrc/`` on channels-last
activations.  Parameters live in ONE flat fp32 arena (``self.flat``) so Adam is a single launch and the
data-parallel gradient all-reduce is a handful of large contiguous buckets; the spectral weights are kept
mode-major ``[M][Ci][Co][2]`` (see ``dft.py``) and converted from / to the reference's four complex corner
tensors only in ``state_dict`` / ``load_state_dict``.
"""
import math
import os
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from .. import _lib, ops
from ..dft import SpectralPlan, mode_major_to_ref_weights, ref_weights_to_mode_major
from .model import Model

HID = 128              # fc1 width, fno.py:103
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
_EVAL_GRAPH = os.environ.get("RPB_EVAL_GRAPH", "0") == "1"      # eval forward replayed from a hipGraph (FNO3d._forward_graphed)
_ALIGN = 64            # floats: every parameter segment starts on a 256 B boundary


class _Workspace:
    """Device buffers for one (batch, mode) configuration; allocated once and reused every step."""

    def __init__(self, model, B, training, device):
        d = ops.Dims(B, *model.shape_in[:3], model.dim_in, model.width, model.padding)
        self.d, self.training = d, training
        self.bf16 = (not training) and model.storage == "bf16"       # eval / rollout only: activations stored as bf16
        # eval / rollout on fp32 storage only: the opt-in two-fp16-plane arithmetic of the fused eval launches (FNO3d.set_arith)
        self.arith = model.arith if (not training and not self.bf16) else "f32"
        self.graph = self.graph_out = self.graph_x = self.graph_key = None      # eval: hipGraph of one forward (FNO3d._forward_graphed)
        self.graph_calls = 0
        self.generation = 0          # bumped by every training-mode forward: a backward whose graph saw an older value must not run
        C, L, plan = model.width, model.n_layers, model.plan
        f = dict(device=device, dtype=torch.float32)
        # Lazy activations: a_{l+1} = act(BN(s_l)) is never materialised (its consumers transform s_l on load), so the
        # step keeps the lifted input A0 and the pre-BatchNorm tensors S[l] only.  Eval ping-pongs two S buffers.
        fa = dict(device=device, dtype=torch.bfloat16) if self.bf16 else f
        # layer 0 entirely on the feature fields (csrc/rpb_feat.hip): the 64-channel lifted tensor A0 is never materialised
        self.FW = 8 if model.dim_in + 4 <= 8 else 32
        self.featfull = (type(model)._lift_fwd is FNO3d._lift_fwd and C == 64 and not self.bf16 and model.dim_in + 4 <= 32
                         and os.environ.get("RPB_LAYER0_GENERIC") != "1" and os.environ.get("RPB_LAYER0_A0") != "1"
                         and ops.cell_mix_writes_gz(d.ncell, C, C, 2 * plan.KW, d.Wp, True))
        self.A0 = None if self.featfull else torch.empty(d.ncell, C, **fa)
        if self.featfull:
            self.phic = torch.empty(d.ncell, self.FW, **f)
            self.w0ext = torch.zeros(C, self.FW, **f)
            self.wcomp = torch.empty(C, self.FW, **f)
        self.S = [torch.empty(d.ncell, C, **fa) for _ in range(L if training else 2)]
        G1, N1 = B * d.Tp * d.Hp, 2 * plan.KW * C
        self.Y1 = torch.empty(G1 * N1, **f)                                  # after W stage / before last stage
        self.Y2 = torch.empty(B * d.Tp * 2 * plan.KH * plan.KW * C, **f)     # after H stage
        # eval: cell_mix also applies the next layer's forward W stage (csrc/rpb_cmx.hip, DFT variant) into its own buffer
        self.fuse_w = (not training and C == 64 and os.environ.get("RPB_EVAL_FUSE_W", "1") != "0"
                       and ops.cell_mix_eval_dft_supported(d.ncell, 2 * plan.KW, d.Wp, 2 * plan.KW))
        # bf16 storage (BASELINE.json configs[4]), round 4: the spectral intermediates that are as large as the bf16 activations -- the
        # fused W stage's rows Y1f and the inverse H stage's rows z2 -- are stored as bf16 as well (RPB_BF16_SPECTRA=0: fp32 as in round 3)
        self.spec_bf16 = (self.bf16 and self.fuse_w and os.environ.get("RPB_BF16_SPECTRA", "1") != "0"
                          and 2 * plan.KH <= 64 and 2 * d.Hp > 64 and (plan.KW * C) % 64 == 0)
        self.Y1f = (torch.empty(G1 * N1, device=device, dtype=torch.bfloat16 if self.spec_bf16 else torch.float32)
                    if self.fuse_w else None)
        self.Z2 = torch.empty(G1 * N1, device=device, dtype=torch.bfloat16) if self.spec_bf16 else self.Y1
        # eval: the last layer's cell_mix produces the crop only (the head reads nothing else): 0.70 of the cells at the headline shape
        self.crop_last = (not training and C == 64 and model.n_layers > 1 and type(model)._lift_fwd is FNO3d._lift_fwd
                          and os.environ.get("RPB_EVAL_CROP_LAST", "1") != "0"
                          and ops.cell_mix_eval_dft_supported(d.ncell, 2 * plan.KW, d.Wp, 2 * plan.KW))
        if (not training and C == 128 and not self.bf16 and model.n_layers > 1 and type(model)._lift_fwd is FNO3d._lift_fwd
                and os.environ.get("RPB_EVAL_CROP_LAST", "1") != "0" and ops.cell_mix_eval_crop_c128_supported(d.ncell, 2 * plan.KW, d.Wp)):
            self.crop_last = True            # width 128 (configs/fsi/fno.yaml): rpb_cell_mix_eval_crop_c128
        self.Xh = [torch.empty(B * 2 * plan.M * C, **f) for _ in range(L if training else 1)]
        self.Yh = torch.empty(B * 2 * plan.M * C, **f)
        self.mean = torch.empty(L, C, **f)
        self.invstd = torch.empty(L, C, **f)
        self.stat_rows = ops.cell_mix_stat_rows(d.ncell, C, C, 2 * plan.KW, d.Wp, True)
        self.stat_part = torch.empty(self.stat_rows * 2 * C, **f) if training else None
        self.sums64 = torch.empty(2 * C, device=device, dtype=torch.float64)
        self.out = torch.empty(d.ncrop, model.dim_out, **f)
        # layer-0 algebra (csrc/rpb_feat.hip): the first spectral layer transforms the Cin + 4 feature FIELDS instead of the 64
        # lifted channels; in training its data gradient is never formed (only fc0's 64 x (Cin + 4) gradient is needed)
        self.feat0 = (type(model)._lift_fwd is FNO3d._lift_fwd and C == 64 and os.environ.get("RPB_LAYER0_GENERIC") != "1")
        if self.feat0:
            T_, H_, W_, Cin = d.T, d.H, d.W, model.dim_in
            self.NB = (B * Cin + 4 + 63) // 64 * 64
            grids, _ = model._consts(device)
            phi = torch.zeros(T_, H_, W_, self.NB, **f)
            c0 = B * Cin
            phi[..., c0] = grids[0].view(T_, 1, 1)
            phi[..., c0 + 1] = grids[1].view(1, H_, 1)
            phi[..., c0 + 2] = grids[2].view(1, 1, W_)
            phi[..., c0 + 3] = 1.0
            self.phi = phi
            n1 = 2 * plan.KW * self.NB
            self.fy1 = torch.empty(T_ * H_ * n1, **f)
            self.fy2 = torch.empty(T_ * 2 * plan.KH * plan.KW * self.NB, **f)
            self.PhiH = torch.empty(2 * plan.M * self.NB, **f)
            if training:
                self.fm_rows = ops.feat_mix_wgrad_rows()
                self.fm_part = torch.empty(self.fm_rows * C * (Cin + 4), **f)
                self.fm_sum = torch.empty(C * (Cin + 4), **f)
                self.mgf = torch.empty(C * (Cin + 3) + C, **f)
                self.mgf2 = torch.empty(C * C, **f)
                self.dconv = torch.empty(C * (Cin + 4), **f)
        if training:
            self.G = [torch.empty(d.ncell, C, **f) for _ in range(2)]
            # projection-head backward: bf16-pipe kernels that recompute gh (no gu tensor) when the shape allows, else the
            # round-1 chain through gu
            # projection-head backward (csrc/rpb_pjx.hip): "gu" (default where supported) = round-1 proj_bwd + cell_wgrad with the
            # fc1 dgrad on the bf16 pipe reading gu; "recompute" (RPB_PROJ_RECOMPUTE=1) = no gu tensor at all, both kernels rebuild
            # gh -- measured slower (act' on 128 hidden units per cell twice); None = the round-1 chain
            ok = (C == 64 and type(model)._lift_fwd is FNO3d._lift_fwd and ops.proj_bwd_fused_supported(C, model.dim_out, d.W, d.Wp))
            self.proj_mode = None if not ok else ("recompute" if os.environ.get("RPB_PROJ_RECOMPUTE") == "1" else "gu")
            self.proj_fused = self.proj_mode == "recompute"
            # round 3: the whole head backward in one pass, gh never in HBM (csrc/rpb_pjf.hip); RPB_HEAD_BWD_FUSED=0 restores the chain
            self.head_fused = (C == 64 and type(model)._lift_fwd is FNO3d._lift_fwd and
                               ops.head_bwd_supported(C, model.dim_out, d.W, d.Wp, False, model.proj_act))
            if self.head_fused:
                self.proj_fused = False
            self.gu = None if (self.proj_fused or self.head_fused) else torch.empty(d.ncrop, HID, **f)
            if ok:
                self.pd_slots = ops.proj_dgrad_slots(d)
            if self.head_fused:
                self.hb_slots, self.hb_row = ops.head_bwd_slots(d), ops.head_bwd_row(model.dim_out)
                self.hb_part = torch.empty(self.hb_slots * self.hb_row, **f)
                self.hb_tot = torch.empty(self.hb_row, **f)
                self.hb_loss_part = torch.empty(self.hb_slots, **f)
                # fused trainer: the head's forward + loss ride in the backward launch (RPB_HEAD_LOSS_FUSED=0: proj_fwd + mse + head_bwd)
                # (fc2 widths 3-4: the fused variant keeps 32 more accumulators and spills -- measured slower at the cylinder's native C = 3)
                self.head_loss_fused = os.environ.get("RPB_HEAD_LOSS_FUSED", "1") != "0" and model.dim_out <= 2
            if self.proj_fused:
                self.pw_slots, self.pw_row, self.pw_roles = ops.proj_wgrad_slots(d), ops.proj_wgrad_row(model.dim_out), ops.proj_wgrad_roles()
                self.pw_part = torch.empty(self.pw_slots * self.pw_row, **f)
                self.pw_sum = torch.empty(self.pw_roles * self.pw_row, **f)
            # BatchNorm-backward sums are produced by the kernels that write the gradient (cell_mix STATS=2)
            self.bnb_rows_gather = ops.cell_mix_stat_rows(d.ncell, HID, C, 0, 1, False, True)
            self.bnb_rows_conv = ops.cell_mix_stat_rows(d.ncell, C, C, 2 * plan.KW, d.Wp, True, True)
            self.bn_part = torch.empty(max(self.bnb_rows_gather, self.bnb_rows_conv, getattr(self, "pd_slots", 0)) * 2 * C, **f)
            self.bn_sums = torch.empty(2 * C, **f)
            # width 128 (configs/fsi/fno.yaml, the Galerkin regressor): the fp32-MFMA cell_mix with the BatchNorm-backward sums in its
            # epilogue spills (8.6 ms at the fsi shape against 2.5 ms without the sums: tools/fsi_probe.py); the plain launch plus the
            # streaming reduction over (s, g) is 3.3 ms.  RPB_BNB_FUSED_128=1 restores the fused launch.
            # The bf16-pipe instance at C = 128 (csrc/rpb_cmx.hip, output halves) carries the sums again.
            self.bnb_unfused = (C == 128 and os.environ.get("RPB_BNB_FUSED_128") != "1"
                                and not ops.cell_mix_writes_gz(d.ncell, C, C, 2 * plan.KW, d.Wp, True))
            if self.bnb_unfused:
                self.bnr_rows = ops.bn_bwd_rows()
                self.bnr_part = torch.empty(self.bnr_rows * 2 * C, **f)
            self.proj_rows = ops.proj_slots(d.ncrop, C, model.dim_out)
            self.proj_part = torch.empty(self.proj_rows * (model.dim_out * HID + HID + model.dim_out), **f)
            self.fused_bwd = C <= 64             # rpb_bn_bwd_row: BN-backward apply + adjoint W stage + conv wgrad in one pass
            # width 128: BN-backward apply + adjoint W stage in one pass as two 64-channel half launches of the C = 64 row kernel
            # (rpb_bn_bwd_row_c128; the weight gradient stays with rpb_cell_wgrad).  RPB_BWD_ROW_128=0: bn_bwd_apply + axis_gemm.
            self.row128 = (C == 128 and os.environ.get("RPB_BWD_ROW_128", "1") != "0"
                           and ops.bn_bwd_row_c128_supported(d.Wp, 2 * plan.KW))
            if self.row128:
                self.row128_part = torch.empty(2 * ops.bn_bwd_row_slots(B * d.Tp * d.Hp) * (64 * 64 + 64), **f)
            self.wg_rows_c = (ops.bn_bwd_row_slots(B * d.Tp * d.Hp) if self.fused_bwd
                              else ops.cell_wgrad_slots(d.ncell, C, C))
            self.wg_rows_p = ops.cell_wgrad_slots(d.ncrop, HID, C)
            self.wg_part = torch.empty(max(self.wg_rows_c * (C * C + C), self.wg_rows_p * (HID * C + HID)), **f)
            # round 4: d convs.l.weight of layers l >= 1 rides in the backward cell_mix of the same layer (wave pairs inside
            # csrc/rpb_cmx.hip: the launch streams gs_l and s_{l-1} anyway), and the row kernel no longer reads the layer input: 24.8
            # instead of 28.6 GB per layer.  The fused launch is bound by instruction issue (3.17 ms against 2.45 ms without the
            # product) while bn_bwd_row drops 3.04 -> 2.25 ms: step 39.3 -> 38.9 ms.  RPB_CELL_MIX_WGRAD=0: the round-3 split.
            self.wg_in_cmx = (os.environ.get("RPB_CELL_MIX_WGRAD", "1") != "0" and self.fused_bwd and C == 64
                              and ops.cell_mix_writes_gz(d.ncell, C, C, 2 * plan.KW, d.Wp, True)
                              and ops.cell_mix_wgrad_supported(d.ncell, 2 * plan.KW, d.Wp))
            if self.wg_in_cmx:
                self.cmw_slots = ops.cell_mix_wgrad_slots(d.ncell, d.Wp)
                self.cmw_wpart = torch.empty(self.cmw_slots * C * C, **f)
                if self.bn_part.numel() < self.cmw_slots * 2 * C:
                    self.bn_part = torch.empty(self.cmw_slots * 2 * C, **f)
            model._alloc_lift_ws(self, f)
            self.tmp_b = torch.empty(HID, **f)
            self.gout = torch.empty(d.ncrop, model.dim_out, **f)
            self.mse_part = torch.empty(ops.mse_rows(), **f)
            self.loss = torch.zeros(1, **f)


class _FNO3dFunction(torch.autograd.Function):
    """Autograd glue: forward / backward are single calls into the HIP pipeline (tensors are storage only)."""

    @staticmethod
    def forward(ctx, x, flat, model):
        ws = model._workspace(x.shape[0], True, x.device)
        out = model._forward_impl(x, ws, training=True)
        ctx.model, ctx.ws, ctx.x, ctx.generation = model, ws, x, ws.generation
        return model._shape_output(out.clone(), x.shape[0])

    @staticmethod
    def backward(ctx, gout):
        model, ws = ctx.model, ctx.ws
        if ws.generation != ctx.generation:
            raise RuntimeError("FNO3d backward: the activations of this graph were overwritten by a later training-mode forward "
                               "with the same batch size (they live in a per-batch-size workspace, not in the autograd graph). "
                               "Call backward() before the next training-mode forward; gradient accumulation over micro-batches "
                               "is forward -> backward -> forward -> backward")
        g = model._unshape_grad(gout.contiguous(), ctx.x.shape[0])
        gflat = torch.zeros_like(model.flat)
        if model.dp is not None:            # caller took a local .mean(): average over ranks like DDP would
            model.dp.begin_step(gflat)
        model._backward_impl(ctx.x, g, ws, gflat)
        if model.dp is not None:
            model.dp.finish_step(gflat)
            gflat.div_(model.dp.world_size)
        return None, gflat, None


class FNO3d(Model):
    def __init__(self, modes1, modes2, modes3, n_layers, width, shape_in, shape_out):
        super().__init__()
        self.modes1, self.modes2, self.modes3 = modes1, modes2, modes3
        self.modes = (modes1, modes2, modes3)
        self.width = width
        self.n_layers = n_layers
        self.shape_in = tuple(int(v) for v in shape_in)
        self.shape_out = tuple(int(v) for v in shape_out)
        self.dim_in = self.shape_in[-1]
        self.r = self.shape_out[0] // self.shape_in[0]
        self.dim_out = self.shape_out[-1] * self.r            # C_out * T_out / T_in, fno.py:86
        self.padding = 6                                       # fno.py:87
        if width % 32 != 0 or width > 128:
            raise ValueError(f"MI355X FNO3d kernels need width in (32, 64, 96?, 128) multiples of 32, got {width}")
        T, H, W = self.shape_in[:3]
        self.plan = SpectralPlan(T + 6, H + 6, W + 6, self.modes)
        M, C, F = self.plan.M, width, self.dim_in + 3

        # ---- flat parameter arena
        self._seg = OrderedDict()
        off = 0

        def seg(name, *shape):
            nonlocal off
            n = int(np.prod(shape))
            self._seg[name] = (off, n, tuple(shape))
            off += (n + _ALIGN - 1) // _ALIGN * _ALIGN

        seg("fc0.weight", C, F)
        seg("fc0.bias", C)
        for l in range(n_layers):
            seg(f"spec.{l}", M, C, C, 2)
            seg(f"convs.{l}.weight", C, C)
            seg(f"convs.{l}.bias", C)
            seg(f"bns.{l}.weight", C)
            seg(f"bns.{l}.bias", C)
        seg("fc1.weight", HID, C)
        seg("fc1.bias", HID)
        seg("fc2.weight", self.dim_out, HID)
        seg("fc2.bias", self.dim_out)
        self.flat = nn.Parameter(torch.zeros(off))
        self.register_buffer("bn_running_mean", torch.zeros(n_layers, C))
        self.register_buffer("bn_running_var", torch.ones(n_layers, C))
        self.register_buffer("bn_num_batches_tracked", torch.zeros(n_layers, dtype=torch.long))
        self.reset_parameters()
        self._ws = {}
        self._dev_cache = None
        self.dp = None            # set by realpdebench_amd.dp.DataParallel (RCCL)
        self.storage = "f32"      # activation storage of the eval / rollout forward, see set_storage
        self.arith = "f32"        # arithmetic of the eval / rollout forward's fused launches, see set_arith

    # ------------------------------------------------------------------ parameters
    def reset_parameters(self):
        """Same init family as the reference: nn.Linear / nn.Conv3d defaults, ``scale * U[0,1)`` spectral
        weights (fno.py:30-38), BatchNorm weight 1 / bias 0."""
        C = self.width
        with torch.no_grad():
            def uni(name, fan_in):
                b = 1.0 / math.sqrt(fan_in)
                self.pview(name).uniform_(-b, b)

            uni("fc0.weight", self.dim_in + 3)
            uni("fc0.bias", self.dim_in + 3)
            scale = 1.0 / (C * C)
            for l in range(self.n_layers):
                self.pview(f"spec.{l}").uniform_(0, 1).mul_(scale)
                uni(f"convs.{l}.weight", C)
                uni(f"convs.{l}.bias", C)
                self.pview(f"bns.{l}.weight").fill_(1.0)
                self.pview(f"bns.{l}.bias").zero_()
            uni("fc1.weight", C)
            uni("fc1.bias", C)
            uni("fc2.weight", HID)
            uni("fc2.bias", HID)

    def pview(self, name, flat=None):
        off, n, shape = self._seg[name]
        base = self.flat.data if flat is None else flat
        return base[off:off + n].view(shape)

    # ------------------------------------------------------------------ reference-compatible state dict
    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        self._params_settled()
        sd = OrderedDict() if destination is None else destination
        C = self.width
        v = lambda n: self.pview(n).detach().clone()
        sd[prefix + "fc0.weight"] = v("fc0.weight")
        sd[prefix + "fc0.bias"] = v("fc0.bias")
        for l in range(self.n_layers):
            ws = mode_major_to_ref_weights(self.pview(f"spec.{l}").detach(), self.modes)
            for k, w in enumerate(ws, start=1):
                sd[prefix + f"spectral_convs.{l}.weights{k}"] = w
        for l in range(self.n_layers):
            sd[prefix + f"convs.{l}.weight"] = v(f"convs.{l}.weight").view(C, C, 1, 1, 1)
            sd[prefix + f"convs.{l}.bias"] = v(f"convs.{l}.bias")
        for l in range(self.n_layers):
            sd[prefix + f"bns.{l}.weight"] = v(f"bns.{l}.weight")
            sd[prefix + f"bns.{l}.bias"] = v(f"bns.{l}.bias")
            sd[prefix + f"bns.{l}.running_mean"] = self.bn_running_mean[l].detach().clone()
            sd[prefix + f"bns.{l}.running_var"] = self.bn_running_var[l].detach().clone()
            sd[prefix + f"bns.{l}.num_batches_tracked"] = self.bn_num_batches_tracked[l].detach().clone()
        for n in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"):
            sd[prefix + n] = v(n)
        return sd

    def load_state_dict(self, state_dict, strict=True, assign=False):
        self._params_settled()          # (in-flight gathers would otherwise land on top of the loaded weights)
        expected = set(self.state_dict().keys())
        missing = expected - set(state_dict.keys())
        unexpected = set(state_dict.keys()) - expected
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for FNO3d: missing {sorted(missing)}, "
                               f"unexpected {sorted(unexpected)}")
        dev = self.flat.device
        with torch.no_grad():
            def put(name, src):
                dst = self.pview(name)
                src = torch.as_tensor(src).to(dev)
                if src.numel() != dst.numel():
                    raise RuntimeError(f"size mismatch for {name}: {tuple(src.shape)} vs {tuple(dst.shape)}")
                dst.copy_(src.reshape(dst.shape))

            for n in ("fc0.weight", "fc0.bias", "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"):
                if n in state_dict:
                    put(n, state_dict[n])
            for l in range(self.n_layers):
                keys = [f"spectral_convs.{l}.weights{k}" for k in (1, 2, 3, 4)]
                if all(k in state_dict for k in keys):
                    ws = [torch.as_tensor(state_dict[k]).to(torch.complex64) for k in keys]
                    put(f"spec.{l}", ref_weights_to_mode_major(*ws))
                for n in (f"convs.{l}.weight", f"convs.{l}.bias", f"bns.{l}.weight", f"bns.{l}.bias"):
                    if n in state_dict:
                        put(n, state_dict[n])
                if f"bns.{l}.running_mean" in state_dict:
                    self.bn_running_mean[l].copy_(torch.as_tensor(state_dict[f"bns.{l}.running_mean"]))
                if f"bns.{l}.running_var" in state_dict:
                    self.bn_running_var[l].copy_(torch.as_tensor(state_dict[f"bns.{l}.running_var"]))
                if f"bns.{l}.num_batches_tracked" in state_dict:
                    self.bn_num_batches_tracked[l] = int(state_dict[f"bns.{l}.num_batches_tracked"])
        return torch.nn.modules.module._IncompatibleKeys(sorted(missing), sorted(unexpected))

    def grads_as_state_dict(self, gflat):
        """Flat gradient arena -> ``{reference parameter name: gradient}`` (complex for the spectral weights,
        matching what autograd leaves in ``p.grad`` of the reference model)."""
        C = self.width
        out = OrderedDict()
        g = lambda n: self.pview(n, gflat).detach().clone()
        for n in ("fc0.weight", "fc0.bias", "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"):
            out[n] = g(n)
        for l in range(self.n_layers):
            for k, w in enumerate(mode_major_to_ref_weights(self.pview(f"spec.{l}", gflat).detach(), self.modes), 1):
                out[f"spectral_convs.{l}.weights{k}"] = w
            out[f"convs.{l}.weight"] = g(f"convs.{l}.weight").view(C, C, 1, 1, 1)
            for n in (f"convs.{l}.bias", f"bns.{l}.weight", f"bns.{l}.bias"):
                out[n] = g(n)
        return out

    def set_storage(self, storage):
        """Activation storage of the eval / rollout forward: ``"f32"`` (the parity path) or ``"bf16"`` (BASELINE.json configs[4]:
        the [cells][C] tensors between kernels are stored as bf16, round to nearest even; weights, spectra, accumulation and
        BatchNorm stay fp32).  Training always stores fp32.  bf16 needs width 64 and 2 * modes3 <= 32 (the bf16-pipe kernels)."""
        if storage not in ("f32", "bf16"):
            raise ValueError(f"storage must be 'f32' or 'bf16', got {storage!r}")
        if storage == "bf16" and (self.width != 64 or 2 * self.modes3 > 32 or type(self)._lift_fwd is not FNO3d._lift_fwd):
            raise NotImplementedError("bf16 activation storage is built for FNO3d at width 64 with modes3 <= 16")
        if storage != self.storage:
            self.storage = storage
            self._ws = {k: v for k, v in self._ws.items() if k[1]}       # drop the eval workspaces (dtype changed)
        return self

    def set_arith(self, arith):
        """Arithmetic of the eval / rollout forward on fp32 storage: ``"f32"`` (default, the parity path: operands as three bf16 planes, six
        products per fp32 product, dropped terms <= 2^-24) or ``"f16x2"`` (opt-in: operands as two fp16 planes rounded to nearest -- one fp32
        unit in the last place --, three products, dropped term <= 2^-22: the grade of 3xTF32, half the matrix-pipe time).  ``"f16x2"``
        covers the eval ``cell_mix`` launches and the projection head at width 64; activations must stay inside fp16's range (they are
        BatchNorm outputs; the spectra and the inverse-stage matrix are rescaled by exact powers of two).  Training is not affected."""
        if arith not in ("f32", "f16x2"):
            raise ValueError(f"arith must be 'f32' or 'f16x2', got {arith!r}")
        if arith == "f16x2" and (self.width != 64 or 2 * self.modes3 > 32 or type(self)._lift_fwd is not FNO3d._lift_fwd):
            raise NotImplementedError("the f16x2 eval arithmetic is built for FNO3d at width 64 with modes3 <= 16")
        if arith != self.arith:
            self.arith = arith
            self._ws = {k: v for k, v in self._ws.items() if k[1]}       # drop the eval workspaces (their graphs hold the old launches)
        return self

    # ------------------------------------------------------------------ device-side constants
    def _consts(self, device):
        if self._dev_cache is None or self._dev_cache[0] != device:
            T, H, W = self.shape_in[:3]
            grids = tuple(torch.tensor(np.linspace(0, 1, n), dtype=torch.float).to(device) for n in (T, H, W))
            plan = SpectralPlan(T + 6, H + 6, W + 6, self.modes, device=device)     # fno.py:135-143 grid in f64->f32
            self._dev_cache = (device, grids, plan)
            self._ws = {}
        return self._dev_cache[1], self._dev_cache[2]

    def _workspace(self, B, training, device):
        key = (B, training, str(device), self.storage if not training else "f32", self.arith if not training else "f32")
        if key not in self._ws:
            self._consts(device)
            self._ws[key] = _Workspace(self, B, training, device)
        return self._ws[key]

    # ------------------------------------------------------------------ spectral stages
    def _spectral_forward_stages(self, x, ws, xh, mats, first_layer, xf=None, y1=None):
        """x [cells][C] -> truncated spectrum xh [B][2][M][C] with stage matrices ``mats`` = (W, H, T).
        ``xf``: lazy BatchNorm(+GELU) of the producing layer, applied while the W stage loads ``x``.
        ``y1``: the W stage's result when a fused producer already wrote it (then ``mats[0]`` is None)."""
        d, p, C = ws.d, self.plan, self.width
        m3, KH, KT = p.KW, p.KH, p.KT
        MW, MH, MT = mats
        N2, N3 = m3 * C, KH * m3 * C
        y1 = ws.Y1 if y1 is None else y1
        if MW is not None and x.dtype == torch.bfloat16:     # eval with bf16 activation storage
            ops.axis_gemm_bf16in(x, ws.Y1, MW, d.B * d.Tp * d.Hp, d.Wp, 2 * m3, C, d.Wp * C, C, 2 * m3 * C, C,
                                 k_valid=d.W if first_layer else None)
        elif MW is not None:                    # None: y1 was already produced (fused backward row kernel / eval cell_mix)
            ops.axis_gemm(x, ws.Y1, MW, d.B * d.Tp * d.Hp, d.Wp, 2 * m3, C, d.Wp * C, C, 2 * m3 * C, C,
                          k_valid=d.W if first_layer else None, xf=xf)
        if y1.dtype == torch.bfloat16:          # the fused W stage stored its rows as bf16 (ws.spec_bf16)
            ops.axis_gemm_bf16in(y1, ws.Y2, MH, d.B * d.Tp, 2 * d.Hp, 2 * KH, N2, 2 * d.Hp * N2, N2, 2 * KH * N2, N2,
                                 k_valid=2 * d.H if first_layer else None)
        else:
            ops.axis_gemm(y1, ws.Y2, MH, d.B * d.Tp, 2 * d.Hp, 2 * KH, N2, 2 * d.Hp * N2, N2, 2 * KH * N2, N2,
                          k_valid=2 * d.H if first_layer else None)
        ops.axis_gemm(ws.Y2, xh, MT, d.B, 2 * d.Tp, 2 * KT, N3, 2 * d.Tp * N3, N3, 2 * KT * N3, N3,
                      k_valid=2 * d.T if first_layer else None)

    def _spectral_inverse_stages(self, yh, ws, mats):
        """yh [B][2][M][C] -> ws.Y1 = rows [B*Tp*Hp][2*m3][C] ready for the fused last stage (cell_mix)."""
        d, p, C = ws.d, self.plan, self.width
        m3, KH, KT = p.KW, p.KH, p.KT
        MT, MH = mats
        N2, N3 = m3 * C, KH * m3 * C
        ops.axis_gemm(yh, ws.Y2, MT, d.B, 2 * KT, 2 * d.Tp, N3, 2 * KT * N3, N3, 2 * d.Tp * N3, N3)
        if getattr(ws, "spec_bf16", False):     # eval on bf16 storage: the rows cell_mix reads are written as bf16
            ops.axis_gemm_bf16out(ws.Y2, ws.Z2, MH, d.B * d.Tp, 2 * KH, 2 * d.Hp, N2, 2 * KH * N2, N2, 2 * d.Hp * N2, N2)
        else:
            ops.axis_gemm(ws.Y2, ws.Y1, MH, d.B * d.Tp, 2 * KH, 2 * d.Hp, N2, 2 * KH * N2, N2, 2 * d.Hp * N2, N2)

    def _feature_spectrum(self, x, ws, plan):
        """ws.PhiH [2][M][NB] = truncated DFT of the Cin + 4 feature fields (x_j per sample, grid_t, grid_h, grid_w, 1; zero in
        the pad margin, which the stages skip through k_valid), batch-innermost so that the contiguous index is wide."""
        d = ws.d
        Cin, NB, m3, KH = self.dim_in, ws.NB, plan.KW, plan.KH
        ws.phi[..., :d.B * Cin].copy_(x.permute(1, 2, 3, 0, 4).reshape(d.T, d.H, d.W, d.B * Cin))      # storage plumbing
        n1, n2, n3 = NB, m3 * NB, KH * m3 * NB
        ops.axis_gemm(ws.phi, ws.fy1, plan.FWt, d.T * d.H, d.Wp, 2 * m3, n1, d.W * n1, n1, 2 * m3 * n1, n1, k_valid=d.W,
                      tag="featW")
        ops.axis_gemm(ws.fy1, ws.fy2, plan.FHt, d.T, 2 * d.Hp, 2 * KH, n2, 2 * d.H * n2, n2, 2 * KH * n2, n2, k_valid=2 * d.H,
                      tag="featH")
        ops.axis_gemm(ws.fy2, ws.PhiH, plan.FTt, 1, 2 * d.Tp, 2 * plan.KT, n3, 2 * d.T * n3, n3, 2 * plan.KT * n3, n3,
                      k_valid=2 * d.T, tag="featT")

    # ------------------------------------------------------------------ forward / backward pipelines
    def _layer_xf(self, ws, l, training):
        """Lazy-activation descriptor of layer ``l``'s output: (mean, invstd, gamma, beta, gelu)."""
        mean = ws.mean[l] if training else self.bn_running_mean[l]
        return (mean, ws.invstd[l], self.pview(f"bns.{l}.weight"), self.pview(f"bns.{l}.bias"), l < self.n_layers - 1)

    def _forward_impl(self, x, ws, training, skip_head=False):
        """``skip_head`` (fused trainer with the one-launch head): stop after the last Fourier layer; ``_backward_impl(target=...)`` runs the
        head's forward inside its backward launch."""
        d, C, L = ws.d, self.width, self.n_layers
        grids, plan = self._consts(x.device)
        P = self.pview
        if training:
            ws.generation += 1
        # a sharded optimizer step (dp.DataParallel.gather_params) leaves the parameter all-gathers in flight on the side stream: the
        # training forward waits per bucket, right before the first kernel that reads the bucket's weights; everything else waits for all
        dpw = self.dp if (self.dp is not None and getattr(self.dp, "_pending", False)) else None
        if dpw is not None and not training:
            dpw.params_ready_all()
            dpw = None
        if dpw is not None:
            dpw.params_ready(L + 1)                  # fc0
            dpw.params_ready(dpw.layer_bucket(0))    # (the layer-0 composite weight of the lift is formed from convs.0 as well)
        self._lift_fwd(x, ws)
        world = self.dp.world_size if (self.dp is not None and training) else 1
        a_in, xf = ws.A0, None                   # layer input tensor and its lazy transform
        y1_ready = False                         # eval: the previous cell_mix already applied this layer's forward W stage
        for l in range(L):
            s = ws.S[l] if training else ws.S[l % 2]
            xh = ws.Xh[l] if training else ws.Xh[0]
            if dpw is not None:
                dpw.params_ready(dpw.layer_bucket(l))
            if l == 0 and ws.feat0:
                self._feature_spectrum(x, ws, plan)
                ops.feat_mix(ws.PhiH, P("fc0.weight"), P("fc0.bias"), xh, d.B, 2 * plan.M, ws.NB, self.dim_in, C)
            elif y1_ready:
                self._spectral_forward_stages(a_in, ws, xh, (None, plan.FHt, plan.FTt), first_layer=False, y1=ws.Y1f)
            else:
                self._spectral_forward_stages(a_in, ws, xh, (plan.FWt, plan.FHt, plan.FTt), first_layer=(l == 0), xf=xf)
            ops.mode_contract_fwd(xh, P(f"spec.{l}"), ws.Yh, d.B, plan.M, C)
            self._spectral_inverse_stages(ws.Yh, ws, (plan.GTt, plan.GHt))
            if training:
                if l == 0 and ws.featfull:
                    ops.cell_mix_feat(ws.phic, ws.wcomp, P("convs.0.bias"), ws.Y1, plan.GWt, s, ws.stat_part, d.ncell, ws.FW,
                                      2 * plan.KW, d.Wp)
                else:
                    ops.cell_mix(a_in, P(f"convs.{l}.weight"), P(f"convs.{l}.bias"), ws.Y1, plan.GWt, s, ws.stat_part,
                                 d.ncell, C, C, 2 * plan.KW, d.Wp, xf=xf)
                ops.reduce_partials(ws.stat_part, ws.stat_rows, 2 * C, out_f64=ws.sums64)
                if world > 1 or (self.dp is not None and training and getattr(self.dp, "sync_stats_always", False)):
                    self.dp.all_reduce_sum(ws.sums64)
                ops.bn_finalize(ws.sums64, float(d.ncell) * world, BN_EPS, BN_MOMENTUM, ws.mean[l], ws.invstd[l],
                                self.bn_running_mean[l], self.bn_running_var[l], C)
                self.bn_num_batches_tracked[l] += 1
                a_in, xf = s, self._layer_xf(ws, l, True)          # fno.py:117-119, applied lazily by the next consumers
            else:
                # eval: the running statistics are known before the launch, so BatchNorm (+GELU) is applied to the tile in
                # cell_mix's epilogue and the next W stage / cell_mix / projection read plain activations (one erf per
                # element instead of two)
                ops.bn_eval_prep(self.bn_running_var[l], BN_EPS, ws.invstd[l], C)
                y1_ready = False
                if ws.fuse_w and l < L - 1 and (l > 0 or ws.featfull or ws.bf16):
                    # ... and the NEXT layer's forward W stage rides in the same launch: the activated line is never read for it
                    feat = l == 0 and ws.featfull
                    ops.cell_mix_eval_dft(ws.phic if feat else a_in, ws.wcomp if feat else P(f"convs.{l}.weight"), P(f"convs.{l}.bias"),
                                          ws.Z2, plan.GWt, s, d.ncell, 2 * plan.KW, d.Wp, self._layer_xf(ws, l, False), plan.FWt,
                                          2 * plan.KW, ws.Y1f, feat_w=ws.FW if feat else 0, arith=ws.arith, spec_e=ops.spec_exp(d))
                    y1_ready = True
                elif l == 0 and ws.featfull:
                    ops.cell_mix_feat(ws.phic, ws.wcomp, P("convs.0.bias"), ws.Z2, plan.GWt, s, None, d.ncell, ws.FW,
                                      2 * plan.KW, d.Wp, oxf=self._layer_xf(ws, l, False))
                elif ws.crop_last and l == L - 1:
                    ops.cell_mix_eval_crop(a_in, P(f"convs.{l}.weight"), P(f"convs.{l}.bias"), ws.Z2, plan.GWt, s, d, 2 * plan.KW,
                                           self._layer_xf(ws, l, False), arith=ws.arith)
                elif ws.bf16:
                    ops.cell_mix_bf16(a_in, P(f"convs.{l}.weight"), P(f"convs.{l}.bias"), ws.Z2, plan.GWt, s, d.ncell, C,
                                      2 * plan.KW, d.Wp, oxf=self._layer_xf(ws, l, False))
                else:
                    ops.cell_mix(a_in, P(f"convs.{l}.weight"), P(f"convs.{l}.bias"), ws.Z2, plan.GWt, s, None, d.ncell, C, C,
                                 2 * plan.KW, d.Wp, oxf=self._layer_xf(ws, l, False))
                a_in, xf = s, None
        if dpw is not None:
            dpw.params_ready(0)                      # fc1 / fc2 (read by the head, in this pass or fused into the backward launch)
            self.dp._pending = False                 # every bucket has been waited for on this stream
        if skip_head:
            return None
        if not training and ws.bf16:
            ops.proj_fwd_bf16(a_in, P("fc1.weight"), P("fc1.bias"), P("fc2.weight"), P("fc2.bias"), ws.out, d, self.dim_out,
                              act=self.proj_act)
        elif not training and ws.arith == "f16x2" and xf is None and self.proj_act == 0 and C == 64 and self.dim_out <= 4:
            ops.proj_fwd_f16x2(a_in, P("fc1.weight"), P("fc1.bias"), P("fc2.weight"), P("fc2.bias"), ws.out, d, self.dim_out)
        else:
            ops.proj_fwd(a_in, P("fc1.weight"), P("fc1.bias"), P("fc2.weight"), P("fc2.bias"), ws.out, d, self.dim_out,
                         xf=xf, act=self.proj_act)
        return ws.out

    def _backward_impl(self, x, gout, ws, gflat, target=None, gscale=None):
        """gout [ncrop][DO] = dLoss/d(fc2 output); writes every parameter gradient into ``gflat``.  With ``target`` (and ``gout`` None; the
        forward ran with ``skip_head``): the head's forward, the squared-error partial sums (``ws.hb_loss_part``) and
        gout = gscale * (out - target) are formed inside the head's backward launch."""
        d, C, L, DO = ws.d, self.width, self.n_layers, self.dim_out
        grids, plan = self._consts(x.device)
        P = self.pview
        GP = lambda n: self.pview(n, gflat)
        world = self.dp.world_size if self.dp is not None else 1
        # ---- projection
        a_last, xf_last = ws.S[L - 1], self._layer_xf(ws, L - 1, True)
        g, g2 = ws.G
        if ws.head_fused:
            # one pass over (s, gout): g, and the partial rows of M = gh^T shat, d fc2, d fc1.bias, d fc2.bias; the finalize kernel
            # derives d fc1.weight and the BatchNorm-backward sums of the last layer from them (csrc/rpb_pjf.hip)
            w1 = P("fc1.weight")
            if target is not None:
                ops.head_fwd_bwd(a_last, w1, P("fc1.bias"), P("fc2.weight"), P("fc2.bias"), target, gscale, g, ws.hb_part, ws.hb_loss_part,
                                 d, DO, xf_last)
            else:
                ops.head_bwd(a_last, w1, P("fc1.bias"), P("fc2.weight"), gout, g, ws.hb_part, d, DO, xf_last)
            ops.reduce_partials(ws.hb_part, ws.hb_slots, ws.hb_row, out_f32=ws.hb_tot)
            ops.head_bwd_finalize(ws.hb_tot, w1, xf_last[2], xf_last[3], DO, GP("fc1.weight"), GP("fc2.weight"), GP("fc1.bias"),
                                  GP("fc2.bias"), ws.bn_sums)
            if self.dp is not None:
                self.dp.bucket_ready(gflat)                      # fc1 / fc2 gradients are final: start their all-reduce
        elif ws.proj_fused:
            # bf16 matrix pipe, no gu tensor: each kernel recomputes gh = (fc2^T gout) * act'(fc1 a + b1) in the operand
            # orientation it needs (csrc/rpb_pjx.hip).  wgrad first: its partials feed the first all-reduce bucket
            w1, b1, w2 = P("fc1.weight"), P("fc1.bias"), P("fc2.weight")
            ops.proj_wgrad(a_last, w1, b1, w2, gout, ws.pw_part, d, DO, xf_last, act=self.proj_act)
            roles, row = ws.pw_roles, ws.pw_row
            HB = HID // roles
            # slot = k * roles + role: one reduction over k leaves [role][row]; the four blocks then move into the arena
            ops.reduce_partials(ws.pw_part, ws.pw_slots // roles, roles * row, out_f32=ws.pw_sum)
            tot = ws.pw_sum.view(roles, row)
            GP("fc1.weight").view(roles, HB * C).copy_(tot[:, :HB * C])
            GP("fc2.weight").view(DO, roles, HB).copy_(tot[:, HB * C:HB * C + DO * HB].view(roles, DO, HB).permute(1, 0, 2))
            GP("fc1.bias").view(roles, HB).copy_(tot[:, HB * C + DO * HB:HB * C + DO * HB + HB])
            GP("fc2.bias").copy_(tot[0, HB * C + DO * HB + HB:])
            if self.dp is not None:
                self.dp.bucket_ready(gflat)                      # fc1 / fc2 gradients are final: start their all-reduce
            ops.proj_dgrad(a_last, w1, b1, w2, gout, g, ws.bn_part, d, DO, xf_last, act=self.proj_act)
            ops.reduce_partials(ws.bn_part, ws.pd_slots, 2 * C, out_f32=ws.bn_sums)
        else:
            ops.proj_bwd(a_last, P("fc1.weight"), P("fc1.bias"), P("fc2.weight"), P("fc2.bias"), gout, ws.gu,
                         ws.proj_part, d, DO, xf=xf_last, act=self.proj_act)
            row = DO * HID + HID + DO
            part = ws.proj_part.view(ws.proj_rows, row)
            # partial row layout: [d fc2.weight | d fc1.bias | d fc2.bias]; the three segments are reduced separately
            self._reduce_cols(part, 0, DO * HID, GP("fc2.weight"))
            self._reduce_cols(part, DO * HID, HID, GP("fc1.bias"))
            self._reduce_cols(part, DO * HID + HID, DO, GP("fc2.bias"))
            ops.cell_wgrad(ws.gu, a_last, ws.wg_part, d.ncrop, HID, C, crop=True, crop6=d.crop6, xf=xf_last)
            partp = ws.wg_part[:ws.wg_rows_p * (HID * C + HID)].view(ws.wg_rows_p, HID * C + HID)
            self._reduce_cols(partp, 0, HID * C, GP("fc1.weight"))
            if self.dp is not None:
                self.dp.bucket_ready(gflat)                      # fc1 / fc2 gradients are final: start their all-reduce
            # fc1 dgrad scattered into the padded layout; its epilogue also accumulates the BatchNorm-backward sums
            # (sum gz, sum gz*shat) of the last Fourier layer, so no separate reduction pass reads g again
            if ws.proj_mode == "gu":           # bf16 matrix pipe, 16 B stores, line-persistent waves (csrc/rpb_pjx.hip)
                ops.proj_dgrad(a_last, P("fc1.weight"), P("fc1.bias"), P("fc2.weight"), None, g, ws.bn_part, d, DO, xf_last,
                               act=self.proj_act, gu=ws.gu)
                ops.reduce_partials(ws.bn_part, ws.pd_slots, 2 * C, out_f32=ws.bn_sums)
            elif C == 128 and os.environ.get("RPB_GATHER_BNB_FUSED_128", "1") == "0":
                # width 128, rounds 5-6a: the fp32 gather instance with the BatchNorm-backward sums in its epilogue spilled (3.2 ms at the
                # fsi shape), so the plain gather + a streaming reduction over (s, g) ran instead.  Round 6b: the gather is csrc/rpb_pjh.hip's
                # matrix-pipe kernel and carries the sums (MODE 3) -- the branch below; RPB_GATHER_BNB_FUSED_128=0 keeps the two launches
                ops.cell_mix(ws.gu, P("fc1.weight"), None, None, None, g, None, d.ncell, HID, C, 0, 1, transpose_w=True,
                             gather=True, crop6=d.crop6)
                xfb = self._layer_xf(ws, L - 1, True)
                if not hasattr(ws, "bnr_part"):
                    ws.bnr_rows = ops.bn_bwd_rows()
                    ws.bnr_part = torch.empty(ws.bnr_rows * 2 * C, device=g.device, dtype=torch.float32)
                ops.bn_bwd_reduce(ws.S[L - 1], g, xfb[0], xfb[1], xfb[2], xfb[3], ws.bnr_part, d.ncell, C, xfb[4])
                ops.reduce_partials(ws.bnr_part, ws.bnr_rows, 2 * C, out_f32=ws.bn_sums)
            else:
                ops.cell_mix(ws.gu, P("fc1.weight"), None, None, None, g, ws.bn_part, d.ncell, HID, C, 0, 1, transpose_w=True,
                             gather=True, crop6=d.crop6, bnb=(ws.S[L - 1],) + self._layer_xf(ws, L - 1, True))
                ops.reduce_partials(ws.bn_part, ws.bnb_rows_gather, 2 * C, out_f32=ws.bn_sums)
        # the C = 64 bf16-pipe cell_mix stores gz = g * gelu'(z) when asked (it evaluates gelu' for the BatchNorm-backward sums
        # anyway), so the BatchNorm-backward apply of the layer below skips gelu'
        gz_ok = ws.fused_bwd and ops.cell_mix_writes_gz(d.ncell, C, C, 2 * plan.KW, d.Wp, True)
        g_is_gz = False                                  # the last layer has no GELU
        # ---- Fourier layers, last to first
        for l in range(L - 1, -1, -1):
            gelu = l < L - 1 and not g_is_gz
            gam, bet = P(f"bns.{l}.weight"), P(f"bns.{l}.bias")
            # ws.bn_sums = (sum gz, sum gz*shat) of layer l, left by the kernel that produced g
            GP(f"bns.{l}.bias").copy_(ws.bn_sums[:C])          # local sums are this rank's d beta / d gamma
            GP(f"bns.{l}.weight").copy_(ws.bn_sums[C:])
            if world > 1 or (self.dp is not None and getattr(self.dp, "sync_stats_always", False)):
                self.dp.all_reduce_sum(ws.bn_sums)
            a_in = ws.A0 if l == 0 else ws.S[l - 1]          # layer input = lazily activated output of layer l-1
            xf_in = None if l == 0 else self._layer_xf(ws, l - 1, True)
            feat_l0 = l == 0 and ws.featfull
            if feat_l0:
                # layer 0: the weight-gradient operand is the feature tensor; the partial's C x C block holds the field moments
                # (gs_0 itself is not stored: with the layer-0 algebra nothing reads it -- Y1 and the field moments are all that leaves)
                ops.bn_bwd_row_feat(ws.S[l], g, ws.phic, None if ws.feat0 else g, ws.mean[l], ws.invstd[l], gam, bet, ws.bn_sums,
                                    float(d.ncell) * world, gelu, plan.GW, ws.Y1, ws.wg_part, d.B * d.Tp * d.Hp, d.Wp, C,
                                    2 * plan.KW, ws.FW)
            elif ws.fused_bwd:
                # one pass: gs = BN/GELU backward (in place), Y1 = GW^T gs (adjoint W stage), conv wgrad partials -- or, when the
                # backward cell_mix below forms the weight gradient, no read of the layer input at all
                wg_later = ws.wg_in_cmx and l > 0
                ops.bn_bwd_row(ws.S[l], g, None if wg_later else a_in, g, ws.mean[l], ws.invstd[l], gam, bet, ws.bn_sums,
                               float(d.ncell) * world, gelu, xf_in, plan.GW, ws.Y1, ws.wg_part, d.B * d.Tp * d.Hp,
                               d.Wp, C, 2 * plan.KW)
            elif ws.row128:
                ops.bn_bwd_row_c128(ws.S[l], g, g, ws.mean[l], ws.invstd[l], gam, bet, ws.bn_sums, float(d.ncell) * world, gelu,
                                    plan.GW, ws.Y1, ws.row128_part, d.B * d.Tp * d.Hp, d.Wp, 2 * plan.KW)
                ops.cell_wgrad(g, a_in, ws.wg_part, d.ncell, C, C, xf=xf_in)
            else:
                ops.bn_bwd_apply(ws.S[l], g, ws.mean[l], ws.invstd[l], gam, bet, ws.bn_sums, float(d.ncell) * world,
                                 g, d.ncell, C, gelu)
                ops.cell_wgrad(g, a_in, ws.wg_part, d.ncell, C, C, xf=xf_in)
            partc = ws.wg_part[:ws.wg_rows_c * (C * C + C)].view(ws.wg_rows_c, C * C + C)
            if feat_l0:
                # Mgf[o][f] = sum_cells gs0[o] phi_f (columns f < FW of the block);  d convs.0.weight = Mgf W0ext^T
                self._reduce_cols(partc, 0, C * C, ws.mgf2)
                ops.small_gemm(ws.mgf2, ws.w0ext, GP("convs.0.weight"), C, C, ws.FW, C, 1, 1, ws.FW, C)
            elif not (ws.fused_bwd and ws.wg_in_cmx and l > 0):
                self._reduce_cols(partc, 0, C * C, GP(f"convs.{l}.weight"))
            self._reduce_cols(partc, C * C, C, GP(f"convs.{l}.bias"))
            # spectral branch: G^ = adjoint of the inverse stages applied to gs
            self._spectral_forward_stages(g, ws, ws.Yh, (None if (ws.fused_bwd or ws.row128) else plan.GW, plan.GH, plan.GT),
                                          first_layer=False)
            ops.mode_contract_wgrad(ws.Xh[l], ws.Yh, GP(f"spec.{l}"), d.B, plan.M, C)
            if self.dp is not None:                          # layer l's 100 MB bucket overlaps the rest of backward
                self.dp.bucket_ready(gflat, hold_small_of=l if (ws.fused_bwd and ws.wg_in_cmx and l > 0) else None)
            gxh = ws.Xh[l]                                   # X^ of this layer is dead after wgrad: reuse for gX^
            ops.mode_contract_dgrad(ws.Yh, P(f"spec.{l}"), gxh, d.B, plan.M, C)
            if l == 0 and ws.feat0:
                # d fc0 = sum_cells g_A0 (x) phi,  g_A0 = Wc0^T gs0 + D^T gX^: the spectral path contracts gX^ with the field
                # spectra, the conv path is Wc0^T times the field moments of gs0 (rpb_lift_bwd applied to gs0 = g); the data
                # gradient g_A0 itself (inverse stages + cell_mix + 3.8 GB) is never formed
                Cin, F = self.dim_in, self.dim_in + 3
                ops.feat_mix_wgrad(gxh, ws.PhiH, ws.fm_part, d.B, 2 * plan.M, ws.NB, Cin, C)
                ops.reduce_partials(ws.fm_part, ws.fm_rows, C * (F + 1), out_f32=ws.fm_sum)
                tot = ws.fm_sum.view(C, F + 1)
                GP("fc0.weight").copy_(tot[:, :F])
                GP("fc0.bias").copy_(tot[:, F])
                wc = P("convs.0.weight")
                if ws.featfull:            # field moments already left by the layer-0 row kernel: conv path = Wc0^T Mgf[:, :F+1]
                    ops.small_gemm(wc, ws.mgf2, ws.dconv, C, F + 1, C, 1, C, C, 1, F + 1)
                    dc = ws.dconv.view(C, F + 1)
                    GP("fc0.weight").add_(dc[:, :F])
                    GP("fc0.bias").add_(dc[:, F])
                    break
                ops.lift_bwd(g, x, grids, ws.lift_part, ws.d)
                partl = ws.lift_part.view(ws.lift_rows, C * F + C)
                self._reduce_cols(partl, 0, C * F + C, ws.mgf)
                # out[i][j] += sum_o Wc0[o][i] * Mgf[o][j]
                ops.small_gemm(wc, ws.mgf[:C * F], GP("fc0.weight"), C, F, C, 1, C, F, 1, F, accumulate=True)
                ops.small_gemm(wc, ws.mgf[C * F:], GP("fc0.bias"), C, 1, C, 1, C, 1, 1, 1, accumulate=True)
                break
            self._spectral_inverse_stages(gxh, ws, (plan.FT, plan.FH))
            if l > 0:      # g_x of layer l = gradient w.r.t. act(BN(s_{l-1})): also leave layer l-1's BN-backward sums
                g_is_gz = gz_ok and l - 1 < L - 1
                if ws.fused_bwd and ws.wg_in_cmx:
                    # ... and d convs.l.weight = gs_l^T act(BN(s_{l-1})): both factors stream through this launch anyway
                    ops.cell_mix_wgrad(g, P(f"convs.{l}.weight"), ws.Y1, plan.FW, g2, ws.bn_part, ws.cmw_wpart, d.ncell, 2 * plan.KW,
                                       d.Wp, (ws.S[l - 1],) + self._layer_xf(ws, l - 1, True), write_gz=g_is_gz)
                    ops.reduce_partials(ws.bn_part, ws.cmw_slots, 2 * C, out_f32=ws.bn_sums)
                    ops.reduce_partials(ws.cmw_wpart, ws.cmw_slots, C * C, out_f32=GP(f"convs.{l}.weight").view(-1))
                    if self.dp is not None:
                        self.dp.small_ready(gflat, l)    # convs.l / bns.l (17 KB) are final only now: their own tiny all-reduce
                elif ws.bnb_unfused:
                    ops.cell_mix(g, P(f"convs.{l}.weight"), None, ws.Y1, plan.FW, g2, None, d.ncell, C, C, 2 * plan.KW, d.Wp,
                                 transpose_w=True)
                    xfb = self._layer_xf(ws, l - 1, True)
                    ops.bn_bwd_reduce(ws.S[l - 1], g2, xfb[0], xfb[1], xfb[2], xfb[3], ws.bnr_part, d.ncell, C, xfb[4])
                    ops.reduce_partials(ws.bnr_part, ws.bnr_rows, 2 * C, out_f32=ws.bn_sums)
                else:
                    ops.cell_mix(g, P(f"convs.{l}.weight"), None, ws.Y1, plan.FW, g2, ws.bn_part, d.ncell, C, C,
                                 2 * plan.KW, d.Wp, transpose_w=True, bnb=(ws.S[l - 1],) + self._layer_xf(ws, l - 1, True),
                                 write_gz=g_is_gz)
                    ops.reduce_partials(ws.bn_part, ws.bnb_rows_conv, 2 * C, out_f32=ws.bn_sums)
            else:
                ops.cell_mix(g, P(f"convs.{l}.weight"), None, ws.Y1, plan.FW, g2, None, d.ncell, C, C, 2 * plan.KW,
                             d.Wp, transpose_w=True)
            g, g2 = g2, g
        else:
            self._lift_bwd(g, x, ws, gflat)

    # ------------------------------------------------------------------ lift stage (overridden by the Galerkin regressor)
    proj_act = 0           # activation between fc1 and fc2: 0 exact GELU (fno.py:124)

    def _alloc_lift_ws(self, ws, f):
        ws.lift_rows = ops._lib.query("rpb_lift_bwd_rows")
        ws.lift_part = torch.empty(ws.lift_rows * (self.width * (self.dim_in + 3) + self.width), **f)

    def _lift_fwd(self, x, ws):
        """fno.py:106-111: ws.A0 = pad(fc0(cat(x, grid))), channels-last."""
        grids, _ = self._consts(x.device)
        if ws.featfull:
            F = self.dim_in + 3
            ops.lift_feat(x, grids, ws.phic, ws.d, ws.FW)
            ws.w0ext[:, :F].copy_(self.pview("fc0.weight"))          # W0ext = [fc0.weight | fc0.bias | 0]: storage plumbing
            ws.w0ext[:, F].copy_(self.pview("fc0.bias"))
            # composite weight of layer 0's channel mixing: Wcomp = convs.0.weight W0ext  [64][FW]
            ops.small_gemm(self.pview("convs.0.weight"), ws.w0ext, ws.wcomp, self.width, ws.FW, self.width, self.width, 1,
                           ws.FW, 1, ws.FW)
        elif ws.A0.dtype == torch.bfloat16:
            ops.lift_pad_fwd_bf16(x, grids, self.pview("fc0.weight"), self.pview("fc0.bias"), ws.A0, ws.d)
        else:
            ops.lift_pad_fwd(x, grids, self.pview("fc0.weight"), self.pview("fc0.bias"), ws.A0, ws.d)

    def _lift_bwd(self, g, x, ws, gflat):
        """g = dLoss/d(ws.A0) -> d fc0.weight, d fc0.bias."""
        grids, _ = self._consts(x.device)
        C, F = self.width, self.dim_in + 3
        ops.lift_bwd(g, x, grids, ws.lift_part, ws.d)
        partl = ws.lift_part.view(ws.lift_rows, C * F + C)
        self._reduce_cols(partl, 0, C * F, self.pview("fc0.weight", gflat))
        self._reduce_cols(partl, C * F, C, self.pview("fc0.bias", gflat))

    @staticmethod
    def _reduce_cols(part2d, col0, ncols, out):
        """out[:] = sum over rows of part2d[:, col0:col0+ncols]  (fp64 accumulate, HIP kernel)."""
        rows, L = part2d.shape
        ops.reduce_partials(part2d, rows, ncols, out_f32=out.view(-1), row_stride=L, col0=col0)

    # ------------------------------------------------------------------ output reshape (fno.py:127-128)
    def _shape_output(self, out, B):
        T, H, W = self.shape_in[:3]
        Co = self.shape_out[-1]
        if self.r == 1:
            return out.view(B, T, H, W, Co)
        return out.view(B, T, H, W, Co, self.r).permute(0, 1, 5, 2, 3, 4).reshape(B, *self.shape_out)

    def _unshape_grad(self, g, B):
        T, H, W = self.shape_in[:3]
        Co = self.shape_out[-1]
        if self.r == 1:
            return g.reshape(B * T * H * W, Co)
        return g.view(B, T, self.r, H, W, Co).permute(0, 1, 3, 4, 5, 2).reshape(B * T * H * W, Co * self.r).contiguous()

    # ------------------------------------------------------------------ Model protocol
    def _check_input(self, x):
        if not x.is_cuda:
            raise RuntimeError("realpdebench_amd.FNO3d runs on MI355X only: move the model and inputs to 'cuda' "
                               "(there is no CPU fallback)")
        if tuple(x.shape[1:]) != self.shape_in:
            raise ValueError(f"expected input [B,{','.join(map(str, self.shape_in))}], got {tuple(x.shape)}")
        return x.contiguous().float()

    def _params_settled(self):
        """A sharded optimizer step (dp.DataParallel.gather_params) leaves the parameter all-gathers in flight on the side stream.  The
        training forward waits per bucket; EVERY other reader of the parameter arena -- eval forward (eager or a replayed hipGraph),
        state_dict / load_state_dict, checkpoints -- makes the current stream wait for all of them here (round-5 advisor finding:
        train.py validates and saves straight after trainer.step)."""
        dp = self.dp
        if dp is not None and getattr(dp, "_pending", False):
            dp.params_ready_all()

    def forward(self, x):
        x = self._check_input(x)
        if torch.is_grad_enabled() and self.flat.requires_grad:
            if not self.training:
                raise NotImplementedError("gradients through eval-mode BatchNorm are not implemented; wrap "
                                          "evaluation in torch.no_grad() as the reference does (train.py:345-361)")
            return _FNO3dFunction.apply(x, self.flat, self)
        ws = self._workspace(x.shape[0], self.training, x.device)
        if not self.training:
            self._params_settled()      # a replayed eval graph never enters _forward_impl, where the per-bucket waits live
        if not self.training and _EVAL_GRAPH and _lib.PROFILE is None and type(self)._forward_impl is FNO3d._forward_impl:
            out = self._forward_graphed(x, ws)
        else:
            out = self._forward_impl(x, ws, training=self.training)
        return self._shape_output(out.clone(), x.shape[0])

    def _forward_graphed(self, x, ws):
        """Eval forward replayed from a hipGraph (the ~40 launches of one forward, 20-200 us each on the small spectral stages, leave
        the host out of the loop: eval.py:314-319 calls this forward back to back).  The graph is captured on the third call with a
        given workspace (the first two run eagerly: lazy allocations, function attributes) and reads the input through a fixed
        staging buffer; it is dropped whenever a pointer it baked in (parameter arena, BatchNorm buffers) changes."""
        key = (self.flat.data_ptr(), self.bn_running_mean.data_ptr(), self.bn_running_var.data_ptr(), torch.cuda.current_device())
        if ws.graph is not None and ws.graph_key != key:
            ws.graph = ws.graph_out = None
            ws.graph_calls = 0
        if ws.graph is None:
            ws.graph_calls += 1
            if ws.graph_calls < 3:
                return self._forward_impl(x, ws, training=False)
            if ws.graph_x is None:
                ws.graph_x = torch.empty_like(x)
            ws.graph_x.copy_(x)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                ws.graph_out = self._forward_impl(ws.graph_x, ws, training=False)
            ws.graph, ws.graph_key = g, key
        else:
            ws.graph_x.copy_(x)
        ws.graph.replay()
        return ws.graph_out

    def train_loss(self, input, target):
        """fno.py:131-133: elementwise ``mse_loss(pred, target)`` (callers take ``.mean()``)."""
        pred = self.forward(input)
        return (pred - target) ** 2
