"""Thin tensor-level wrappers over the C ABI (``include/rpb.h``).

PyTorch-ROCm tensors are storage only: every wrapper checks device / dtype / contiguity, passes raw
``data_ptr()``s plus the current HIP stream, and raises on any error.  No wrapper has a non-HIP path.
"""
import os
import threading

import torch

from . import _lib

F32 = torch.float32


def _stream():
    return torch.cuda.current_stream().cuda_stream


class Sub:
    """A row-major sub-block of a contiguous tensor: base tensor + element offset (the leading dimension travels in the
    call's ``ld*`` argument).  Lets the token GEMMs read / write column ranges of e.g. the fused Q|K|V tensor in place."""

    def __init__(self, t, offset):
        self.t, self.offset = t, int(offset)


def _p(t, dtype=F32):
    if t is None:
        return None
    if isinstance(t, Sub):
        return _p(t.t, dtype) + 4 * t.offset
    if not t.is_cuda:
        raise _lib.RpbError("realpdebench_amd ops need tensors on a HIP device (no CPU fallback exists)")
    if t.dtype != dtype or not t.is_contiguous():
        raise _lib.RpbError(f"expected contiguous {dtype} tensor, got {t.dtype} contiguous={t.is_contiguous()}")
    return t.data_ptr()


def _xf(xf, gelu_code=None):
    """Lazy-activation argument group: ``None`` or ``(mean, invstd, gamma, beta, gelu)`` -> 5 C arguments.
    ``gelu_code``: the integer passed instead of 1 when the GELU flag is set (cell_mix: 2 = "and store gz")."""
    if xf is None:
        return (None, None, None, None, 0)
    mean, invstd, gamma, beta, gelu = xf
    return (_p(mean), _p(invstd), _p(gamma), _p(beta), (gelu_code or 1) if gelu else 0)


class Dims:
    """Shape bundle of one FNO3d problem: unpadded (T,H,W), padded (Tp,Hp,Wp), channels."""

    def __init__(self, B, T, H, W, Cin, C, pad):
        self.B, self.T, self.H, self.W, self.Cin, self.C = B, T, H, W, Cin, C
        self.Tp, self.Hp, self.Wp = T + pad, H + pad, W + pad
        self.ncell = B * self.Tp * self.Hp * self.Wp
        self.ncrop = B * T * H * W

    @property
    def crop6(self):
        return (self.T, self.H, self.W, self.Tp, self.Hp, self.Wp)


def lift_pad_fwd(x, grids, w0, b0, out, d):
    _lib.call("rpb_lift_pad_fwd", _p(x), _p(grids[0]), _p(grids[1]), _p(grids[2]), _p(w0), _p(b0), _p(out),
              d.B, d.T, d.H, d.W, d.Cin, d.C, d.Tp, d.Hp, d.Wp, _stream(),
              label="lift_pad", nbytes=4 * (d.ncrop * d.Cin + d.ncell * d.C), flops=2 * d.ncrop * d.C * (d.Cin + 3))


def lift_pad_fwd_bf16(x, grids, w0, b0, out, d):
    assert out.dtype == torch.bfloat16
    _lib.call("rpb_lift_pad_fwd_bf16", _p(x), _p(grids[0]), _p(grids[1]), _p(grids[2]), _p(w0), _p(b0), _p(out, torch.bfloat16),
              d.B, d.T, d.H, d.W, d.Cin, d.C, d.Tp, d.Hp, d.Wp, _stream(),
              label="lift_pad_bf16", nbytes=4 * d.ncrop * d.Cin + 2 * d.ncell * d.C, flops=2 * d.ncrop * d.C * (d.Cin + 3))


def lift_bwd(g, x, grids, part, d):
    _lib.call("rpb_lift_bwd", _p(g), _p(x), _p(grids[0]), _p(grids[1]), _p(grids[2]), _p(part),
              d.B, d.T, d.H, d.W, d.Cin, d.C, d.Tp, d.Hp, d.Wp, _stream(),
              label="lift_bwd", nbytes=4 * d.ncrop * (d.Cin + d.C), flops=2 * d.ncrop * d.C * (d.Cin + 3))


def axis_gemm(inp, out, Mt, G, K, O, N, in_g, in_k, out_g, out_o, k_valid=None, accumulate=False, tag="", xf=None):
    """``Mt`` is the stage matrix transposed, ``[K, O]`` (out[g,o,n] = sum_k Mt[k,o] * in[g,k,n])."""
    assert tuple(Mt.shape) == (K, O), (Mt.shape, K, O)
    M = Mt
    kv = K if k_valid is None else k_valid
    _lib.call("rpb_axis_gemm", _p(inp), _p(out), _p(M), G, K, O, N, in_g, in_k, out_g, out_o, kv, int(accumulate),
              *_xf(xf), _stream(), label=f"axis_gemm[{tag}K{K}xO{O}]", nbytes=4 * G * N * (kv + O), flops=2 * G * N * kv * O)


def axis_gemm_bf16in(inp, out, Mt, G, K, O, N, in_g, in_k, out_g, out_o, k_valid=None):
    """The forward W stage reading bf16-stored activations (strides in bf16 elements); spectra stay fp32."""
    assert tuple(Mt.shape) == (K, O) and inp.dtype == torch.bfloat16
    kv = K if k_valid is None else k_valid
    _lib.call("rpb_axis_gemm_bf16in", _p(inp, torch.bfloat16), _p(out), _p(Mt), G, K, O, N, in_g, in_k, out_g, out_o, kv, _stream(),
              label=f"axis_gemm_bf16in[K{K}xO{O}]", nbytes=G * N * (2 * kv + 4 * O), flops=2 * G * N * kv * O)


def axis_gemm_bf16out(inp, out, Mt, G, K, O, N, in_g, in_k, out_g, out_o, k_valid=None):
    """The inverse H stage writing its rows as bf16 (``out`` bf16, out strides in bf16 elements): K <= 64, O > 64."""
    assert out.dtype == torch.bfloat16
    kv = K if k_valid is None else k_valid
    _lib.call("rpb_axis_gemm_bf16out", _p(inp), _p(out, torch.bfloat16), _p(Mt), G, K, O, N, in_g, in_k, out_g, out_o, kv, _stream(),
              label=f"axis_gemm_bf16out[K{K}xO{O}]", nbytes=G * N * (4 * kv + 2 * O), flops=2 * G * N * kv * O)


def _zs(z2):
    """(pointer, spectra_bf16 flag, bytes per element) of a z2 / y1 tensor that is fp32 or bf16."""
    return _p(z2, z2.dtype), int(z2.dtype == torch.bfloat16), (2 if z2.dtype == torch.bfloat16 else 4)


def cell_mix_bf16(x, Wm, bias, z2, GW, out, ncell, C, K2, Wp, oxf=None):
    """Eval cell_mix on bf16-stored activations: x, out bf16 ``[ncell][C]``; ``oxf`` = (mean, invstd, gamma, beta, gelu); ``z2`` fp32 or
    bf16 (spectra stored as bf16 too)."""
    assert x.dtype == torch.bfloat16 and out.dtype == torch.bfloat16
    zp, sb, zb = _zs(z2)
    _lib.call("rpb_cell_mix_bf16", _p(x, torch.bfloat16), _p(Wm), _p(bias), zp, _p(GW), _p(out, torch.bfloat16), ncell, C, K2, Wp,
              *_xf(oxf), sb, _stream(), label="cell_mix_bf16", nbytes=4 * ncell * C + zb * (ncell // Wp) * K2 * C,
              flops=2 * ncell * C * (K2 + C))


def spec_exp(d):
    """The power of two the opt-in f16x2 eval arithmetic moves from the last inverse-stage matrix (entries <= 2 / (Tp Hp Wp): subnormal in
    fp16) to the z2 rows: floor(log2(Tp Hp Wp)) - 1, so that max |GW| 2^e lies in (0.5, 1]."""
    return int(d.Tp * d.Hp * d.Wp).bit_length() - 2


def cell_mix_eval_crop(x, Wm, bias, z2, GW, out, d, K2, oxf, arith="f32"):
    """Eval cell_mix of the last Fourier layer over the crop only (``d`` = the padded / cropped sizes); x / out f32 or bf16 ``[ncell][64]``.
    ``arith="f16x2"``: the opt-in two-fp16-plane arithmetic (fp32 storage only; csrc/rpb_cmx.hip, H2)."""
    bf = x.dtype == torch.bfloat16
    assert out.dtype == x.dtype
    nl, tq = d.B * d.T * d.H, (d.W + 31) // 32
    ncell = nl * min(32 * tq, d.Wp)
    zp, sb, zb = _zs(z2)
    assert bf or not sb
    if x.shape[-1] == 128:          # width 128 (fsi FNO, the Galerkin regressor): the C = 128 instance, fp32 storage
        assert not bf and arith == "f32"
        _lib.call("rpb_cell_mix_eval_crop_c128", _p(x), _p(Wm), _p(bias), zp, _p(GW), _p(out), d.B, d.T, d.H, d.W, d.Tp, d.Hp, d.Wp, K2,
                  *_xf(oxf), _stream(), label="cell_mix[KC128->CO128,spec=1,stats=oxf,crop]",
                  nbytes=8 * ncell * 128 + zb * nl * K2 * 128, flops=2 * ncell * 128 * (K2 + 128))
        return
    if arith == "f16x2":
        assert not bf and not sb
        _lib.call("rpb_cell_mix_eval_crop_f16x2", _p(x), _p(Wm), _p(bias), zp, _p(GW), _p(out), d.B, d.T, d.H, d.W, d.Tp, d.Hp, d.Wp, K2,
                  *_xf(oxf), spec_exp(d), _stream(), label="cell_mix[KC64->CO64,spec=1,stats=oxf,crop,f16x2]",
                  nbytes=8 * ncell * 64 + zb * nl * K2 * 64, flops=2 * ncell * 64 * (K2 + 64))
        return
    assert arith == "f32"
    _lib.call("rpb_cell_mix_eval_crop", _p(x, x.dtype), _p(Wm), _p(bias), zp, _p(GW), _p(out, x.dtype), d.B, d.T, d.H, d.W, d.Tp, d.Hp,
              d.Wp, K2, *_xf(oxf), int(bf), sb, _stream(), label="cell_mix_bf16[crop]" if bf else "cell_mix[KC64->CO64,spec=1,stats=oxf,crop]",
              nbytes=(4 if bf else 8) * ncell * 64 + zb * nl * K2 * 64, flops=2 * ncell * 64 * (K2 + 64))


_DFT_SCRATCH = {}


def cell_mix_eval_crop_c128_supported(ncell, K2, Wp):
    return bool(_lib.query("rpb_cell_mix_eval_crop_c128_supported", ncell, K2, Wp))


def cell_mix_eval_dft_supported(ncell, K2, Wp, K2f):
    return bool(_lib.query("rpb_cell_mix_eval_dft_supported", ncell, K2, Wp, K2f))


def cell_mix_eval_dft(x, Wm, bias, z2, GW, out, ncell, K2, Wp, oxf, FWt, K2f, y1, feat_w=0, arith="f32", spec_e=None):
    """Eval cell_mix (C = 64, output transform ``oxf``) + the next layer's forward W stage: ``y1 [ncell/Wp][K2f][64]``.
    ``arith="f16x2"`` (with ``spec_e = spec_exp(d)``): the opt-in two-fp16-plane arithmetic, fp32 storage only."""
    assert tuple(FWt.shape) == (Wp, K2f)
    assert arith in ("f32", "f16x2")
    KC = feat_w or 64
    key = (str(out.device), Wp)
    if key not in _DFT_SCRATCH:
        _DFT_SCRATCH[key] = torch.empty(3 * Wp * 16, device=out.device, dtype=torch.float32)     # GW planes, rewritten by every launch
    if x.dtype == torch.bfloat16:
        assert out.dtype == torch.bfloat16 and not feat_w
        zp, sb, zb = _zs(z2)
        assert y1.dtype == z2.dtype, "spectra are fp32 or bf16 on both sides of the launch"
        _lib.call("rpb_cell_mix_eval_dft_bf16", _p(x, torch.bfloat16), _p(Wm), _p(bias), zp, _p(GW), _p(out, torch.bfloat16), ncell, K2, Wp,
                  *_xf(oxf), _p(FWt), K2f, _p(y1, y1.dtype), _p(_DFT_SCRATCH[key]), sb, _stream(), label="cell_mix_bf16[+W]",
                  nbytes=4 * ncell * 64 + zb * (ncell // Wp) * (K2 + K2f) * 64, flops=2 * ncell * 64 * (K2 + 64 + K2f))
        return
    if arith == "f16x2":
        assert spec_e is not None
        _lib.call("rpb_cell_mix_eval_dft_f16x2", _p(x), _p(Wm), _p(bias), _p(z2), _p(GW), _p(out), ncell, K2, Wp, int(feat_w), *_xf(oxf),
                  _p(FWt), K2f, _p(y1), _p(_DFT_SCRATCH[key]), int(spec_e), _stream(),
                  label=f"cell_mix[{'feat%d' % feat_w if feat_w else 'KC64'}->CO64,spec=1,stats=oxf+W,f16x2]",
                  nbytes=4 * ncell * (KC + 64) + 4 * (ncell // Wp) * (K2 + K2f) * 64, flops=2 * ncell * 64 * (K2 + KC + K2f))
        return
    _lib.call("rpb_cell_mix_eval_dft", _p(x), _p(Wm), _p(bias), _p(z2), _p(GW), _p(out), ncell, K2, Wp, int(feat_w), *_xf(oxf),
              _p(FWt), K2f, _p(y1), _p(_DFT_SCRATCH[key]), _stream(), label=f"cell_mix[{'feat%d' % feat_w if feat_w else 'KC64'}->CO64,spec=1,stats=oxf+W]",
              nbytes=4 * ncell * (KC + 64) + 4 * (ncell // Wp) * (K2 + K2f) * 64, flops=2 * ncell * 64 * (K2 + KC + K2f))


def mode_contract_fwd(X, W, Y, B, M, C):
    _lib.call("rpb_mode_contract_fwd", _p(X), _p(W), _p(Y), B, M, C, _stream(), label="mode_contract_fwd",
              nbytes=8 * M * C * (C + 2 * B), flops=8 * B * M * C * C)


def mode_contract_dgrad(GY, W, GX, B, M, C):
    # the kernel keeps the [B][2][C] gradient tile and the padded [C][C+1] complex weight tile of a mode in LDS:
    # at width 128 that caps the batch per launch, so larger batches go in batch-major slices (contiguous in X)
    # (widths that are multiples of 32 take the 32-row-chunk kernel: a [32][C+1] weight tile)
    lds_cap = 160 * 1024 - (32 if C % 32 == 0 else C) * (C + 1) * 8
    bmax = max(1, lds_cap // (2 * C * 4))
    b0 = 0
    while b0 < B:
        nb = min(bmax, B - b0)
        off = b0 * 2 * M * C
        _lib.call("rpb_mode_contract_dgrad", _p(GY) + 4 * off, _p(W), _p(GX) + 4 * off, nb, M, C, _stream(),
                  label="mode_contract_dgrad", nbytes=8 * M * C * (C + 2 * nb), flops=8 * nb * M * C * C)
        b0 += nb


def mode_contract_wgrad(X, GY, GW, B, M, C, accumulate=False):
    _lib.call("rpb_mode_contract_wgrad", _p(X), _p(GY), _p(GW), B, M, C, int(accumulate), _stream(),
              label="mode_contract_wgrad", nbytes=8 * M * C * (C + 2 * B), flops=8 * B * M * C * C)


def cell_mix_stat_rows(ncell, KC, CO, K2, Wp, has_spec, bn_bwd_stats=False):
    return _lib.query("rpb_cell_mix_stat_rows", ncell, KC, CO, K2, Wp, int(has_spec), int(bn_bwd_stats))


def cell_mix_writes_gz(ncell, KC, CO, K2, Wp, has_spec, gather=False):
    return bool(_lib.query("rpb_cell_mix_writes_gz", ncell, KC, CO, K2, Wp, int(has_spec), int(gather)))


def cell_mix(x, Wm, bias, z2, GW, out, stats_part, ncell, KC, CO, K2, Wp, transpose_w=False, gather=False,
             crop6=(0, 0, 0, 1, 1, 1), xf=None, bnb=None, oxf=None, write_gz=False):
    """``bnb`` = (s, mean, invstd, gamma, beta, gelu) of the layer whose output gradient this launch produces: the
    stats partials then hold that layer's BatchNorm-backward sums (sum gz, sum gz*shat).
    ``oxf`` = (mean, invstd, gamma, beta, gelu): store act(BN(out)) instead of out (eval mode, no statistics).
    ``write_gz`` (with ``bnb`` and its GELU): store gz = out * gelu'(z); the consumer's BatchNorm backward then runs with gelu off."""
    tag = None
    if oxf is not None:
        assert bnb is None and stats_part is None
        bnb, tag = (None,) + tuple(oxf), "oxf"

    spec, stats = z2 is not None, stats_part is not None
    rows_in = (crop6[0] * crop6[1] * crop6[2] * (ncell // (crop6[3] * crop6[4] * crop6[5]))) if gather else ncell
    _lib.call("rpb_cell_mix", _p(x), _p(Wm), _p(bias), _p(z2), _p(GW), _p(out), _p(stats_part), ncell, KC, CO, K2, Wp,
              int(transpose_w), int(gather), *crop6, *_xf(xf),
              *((None,) + _xf(None) if bnb is None else (_p(bnb[0]),) + _xf(bnb[1:], 2 if write_gz else None)), _stream(),
              label=f"cell_mix[KC{KC}->CO{CO},spec={int(spec)},stats={tag or int(stats) + int(bnb is not None)}]",
              nbytes=4 * (rows_in * KC + ncell * CO + (ncell // Wp * K2 * CO if spec else 0)
                          + (ncell * CO if bnb is not None and oxf is None else 0)),     # BN-backward sums: the pre-BN tensor is read too
              flops=2 * ncell * CO * ((K2 if spec else 0)) + 2 * rows_in * CO * KC)


def cell_mix_wgrad_supported(ncell, K2, Wp):
    return bool(_lib.query("rpb_cell_mix_wgrad_supported", ncell, K2, Wp))


def cell_mix_wgrad_slots(ncell, Wp):
    return _lib.query("rpb_cell_mix_wgrad_slots", ncell, Wp)


def cell_mix_wgrad(gs, Wc, z2, FW, out, stats_part, wg_part, ncell, K2, Wp, bnb, write_gz=True):
    """Backward cell_mix of a Fourier layer at C = 64 with d convs.l.weight riding along (the wave pairs of csrc/rpb_cmx.hip): ``bnb`` = (s_prev, mean,
    invstd, gamma, beta, gelu) of the layer below, whose activation is this layer's input.  ``stats_part`` / ``wg_part``:
    ``cell_mix_wgrad_slots`` partial rows of [2][64] / [64][64]."""
    C = 64
    _lib.call("rpb_cell_mix_wgrad", _p(gs), _p(Wc), _p(z2), _p(FW), _p(out), _p(stats_part), _p(wg_part), ncell, K2, Wp, _p(bnb[0]),
              _p(bnb[1]), _p(bnb[2]), _p(bnb[3]), _p(bnb[4]), (2 if write_gz else 1) if bnb[5] else 0, _stream(),
              label="cell_mix[KC64->CO64,spec=1,stats=2,wgrad]", nbytes=4 * (3 * ncell * C + ncell // Wp * K2 * C),
              flops=2 * ncell * C * (K2 + 2 * C))


def cell_wgrad_slots(ncell, CO, CI):
    return _lib.query("rpb_cell_wgrad_slots", ncell, CO, CI)


def cell_wgrad(gs, x, part, ncell, CO, CI, crop=False, crop6=(0, 0, 0, 1, 1, 1), xf=None):
    _lib.call("rpb_cell_wgrad", _p(gs), _p(x), _p(part), ncell, CO, CI, int(crop), *crop6, *_xf(xf), _stream(),
              label=f"cell_wgrad[CO{CO},CI{CI}]", nbytes=4 * ncell * (CO + CI), flops=2 * ncell * CO * CI)


class deferred_reductions:
    """Context manager: every ``reduce_partials(..., deferrable=True)`` issued inside (fp32 output, no scale / accumulate) is queued and all
    of them run as ONE ``rpb_reduce_partials_grouped`` launch on exit.  For backward passes whose blocks each end in ~10 partial reductions
    of a few microseconds whose results nobody reads before the pass is over (parameter gradients).  Deferral is OPT-IN per call site: a
    reduction whose output is read inside the block simply does not pass ``deferrable`` and runs at once; the caller of a deferrable one
    guarantees that its partial buffer is not overwritten and its output not read inside the block.  The queue is per host thread."""
    _tls = threading.local()
    _staging = {}                    # (device, thread, stream) -> (pinned host table, device table, event): the item table travels without blocking the host

    def __init__(self, enabled=True):
        self.enabled, self.items, self.keep = enabled, [], []

    @classmethod
    def current(cls):
        return getattr(cls._tls, "active", None)

    def __enter__(self):
        # no deferral while the stream is being captured into a hipGraph: the H2D copy node of the item table would re-read the shared
        # pinned buffer at REPLAY time (after later blocks have overwritten it) and an event recorded during capture cannot be
        # synchronised on -- inside a capture every reduction runs at once, as if deferral were off (round-5 advisor finding)
        if self.enabled and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            self.enabled = False
        if self.enabled:
            self.prev, deferred_reductions._tls.active = deferred_reductions.current(), self
        return self

    def add(self, part_ptr, out_ptr, rows, L, stride, tensors):
        self.items.append((part_ptr, out_ptr, rows, L, stride))
        self.keep.extend(tensors)

    def __exit__(self, *exc):
        if not self.enabled:
            return False
        deferred_reductions._tls.active = self.prev
        if self.items and exc[0] is None:
            dev, n = self.keep[0].device, len(self.items)
            key = (dev, threading.get_ident(), _stream())     # per STREAM: the device table is reused in stream order only
            st = deferred_reductions._staging.get(key)
            if st is None or st[0].shape[0] < n:              # pinned host table + device table, grown to the largest block seen
                cap = max(256, 2 * n)
                st = (torch.empty(cap, 6, dtype=torch.int64).pin_memory(), torch.empty(cap, 6, dtype=torch.int64, device=dev),
                      torch.cuda.Event())
                deferred_reductions._staging[key] = st
            host, d, ev = st
            ev.synchronize()                                  # the previous block's copy has left the pinned table (no-op the first time)
            chunk0, cols = 0, _lib.query("rpb_reduce_partials_grouped_cols")
            for i, (pp, op, rows, L, stride) in enumerate(self.items):
                host[i] = torch.tensor((pp, op, rows, L, stride, chunk0), dtype=torch.int64)
                chunk0 += (L + cols - 1) // cols
            d[:n].copy_(host[:n], non_blocking=True)          # pinned -> device on the current stream: the host does not wait for the stream
            ev.record()
            _lib.call("rpb_reduce_partials_grouped", d.data_ptr(), n, chunk0, _stream(), label="reduce_partials_grouped")
        self.items, self.keep = [], []
        return False


def reduce_partials(part, rows, L, out_f32=None, out_f64=None, scale=1.0, accumulate=False, row_stride=None,
                    col0=0, deferrable=False):
    """out[j] (+)= scale * sum_r part[r*row_stride + col0 + j] for j < L.  ``deferrable``: inside a ``deferred_reductions`` block the
    reduction may run at the block's end (nobody reads ``out`` before that)."""
    q = deferred_reductions.current() if deferrable else None
    if q is not None and out_f64 is None and out_f32 is not None and scale == 1.0 and not accumulate:
        q.add(_p(part) + 4 * col0, _p(out_f32), rows, L, L if row_stride is None else row_stride,
              [part.t if isinstance(part, Sub) else part, out_f32.t if isinstance(out_f32, Sub) else out_f32])
        return
    _lib.call("rpb_reduce_partials", _p(part) + 4 * col0, rows, L, L if row_stride is None else row_stride,
              _p(out_f32), _p(out_f64, torch.float64), float(scale), int(accumulate), _stream())


def reduce_partials_batched(part, nbatch, rows, L, out):
    """out[b][j] = sum_r part[b][r][j]: ``nbatch`` independent fp64 reductions of contiguous [rows][L] blocks, one launch."""
    _lib.call("rpb_reduce_partials_batched", _p(part), nbatch, rows, L, L, rows * L, _p(out), _stream())


def bn_finalize(sums, count, eps, momentum, mean, invstd, rmean, rvar, C):
    _lib.call("rpb_bn_finalize", _p(sums, torch.float64), float(count), eps, momentum, _p(mean), _p(invstd), _p(rmean),
              _p(rvar), C, _stream())


def bn_eval_prep(rvar, eps, invstd, C):
    _lib.call("rpb_bn_eval_prep", _p(rvar), eps, _p(invstd), C, _stream())


def bn_act_fwd(s, mean, invstd, gamma, beta, y, ncell, C, gelu):
    _lib.call("rpb_bn_act_fwd", _p(s), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(y), ncell, C, int(gelu), _stream(),
              label=f"bn_act_fwd[gelu={int(gelu)}]", nbytes=8 * ncell * C, flops=12 * ncell * C)


def bn_bwd_rows():
    return _lib.query("rpb_bn_bwd_rows")


def bn_bwd_reduce(s, gy, mean, invstd, gamma, beta, part, ncell, C, gelu):
    _lib.call("rpb_bn_bwd_reduce", _p(s), _p(gy), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(part), ncell, C,
              int(gelu), _stream(), label=f"bn_bwd_reduce[gelu={int(gelu)}]", nbytes=8 * ncell * C, flops=16 * ncell * C)


def bn_bwd_apply(s, gy, mean, invstd, gamma, beta, sums, count, gs, ncell, C, gelu):
    _lib.call("rpb_bn_bwd_apply", _p(s), _p(gy), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(sums), float(count),
              _p(gs), ncell, C, int(gelu), _stream(), label=f"bn_bwd_apply[gelu={int(gelu)}]", nbytes=12 * ncell * C,
              flops=20 * ncell * C)


def bn_bwd_row_slots(G):
    return _lib.query("rpb_bn_bwd_row_slots", G)


def bn_bwd_row(s, gy, x, gs, mean, invstd, gamma, beta, sums, count, gelu, xf, GWt, Y1, part, G, Wp, C, K2):
    """``x`` None (C = 64): no weight gradient in this launch -- ``cell_mix_wgrad`` of the same layer forms it and the layer input is
    not read here (3 instead of 4 tensor passes)."""
    ncell = G * Wp
    nox = x is None
    _lib.call("rpb_bn_bwd_row", _p(s), _p(gy), _p(x), _p(gs), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(sums),
              float(count), int(gelu), *_xf(None if nox else xf), _p(GWt), _p(Y1), _p(part), G, Wp, C, K2, _stream(),
              label=f"bn_bwd_row[C{C},gelu={int(gelu)}{',nox' if nox else ''}]", nbytes=4 * ((3 if nox else 4) * ncell * C + G * K2 * C),
              flops=2 * ncell * C * ((0 if nox else C) + K2))


def bn_bwd_row_c128_supported(Wp, K2):
    return bool(_lib.query("rpb_bn_bwd_row_c128_supported", Wp, K2))


def bn_bwd_row_c128(s, gy, gs, mean, invstd, gamma, beta, sums, count, gelu, GWt, Y1, part, G, Wp, K2):
    """Width 128: BatchNorm(+GELU) backward apply + adjoint W stage in one pass (two 64-channel half launches of the C = 64 row kernel)."""
    ncell, C = G * Wp, 128
    _lib.call("rpb_bn_bwd_row_c128", _p(s), _p(gy), _p(gs), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(sums), float(count), int(gelu),
              _p(GWt), _p(Y1), _p(part), G, Wp, K2, _stream(),
              label=f"bn_bwd_row[C128,gelu={int(gelu)},nox]", nbytes=4 * (3 * ncell * C + G * K2 * C), flops=2 * ncell * C * K2)


def proj_slots(ncrop, C, DO):
    return _lib.query("rpb_proj_slots", ncrop, C, DO)


def proj_fwd(a, w1, b1, w2, b2, out, d, DO, xf=None, act=0):
    """act: 0 exact GELU (FNO head), 1 SiLU (Galerkin SpectralRegressor head)."""
    _lib.call("rpb_proj_fwd", _p(a), _p(w1), _p(b1), _p(w2), _p(b2), _p(out), d.ncrop, d.C, DO, *d.crop6, *_xf(xf),
              int(act), _stream(),
              label="proj_fwd", nbytes=4 * d.ncrop * (d.C + DO), flops=2 * d.ncrop * 128 * (d.C + DO))


def proj_fwd_f16x2(a, w1, b1, w2, b2, out, d, DO):
    """The evaluation head on the opt-in two-fp16-plane arithmetic (C = 64, DO <= 4, exact GELU, plain activations)."""
    assert d.C == 64 and 1 <= DO <= 4
    _lib.call("rpb_proj_fwd_f16x2", _p(a), _p(w1), _p(b1), _p(w2), _p(b2), _p(out), d.ncrop, DO, *d.crop6, _stream(),
              label="proj_fwd[f16x2]", nbytes=4 * d.ncrop * (d.C + DO), flops=2 * d.ncrop * 128 * (d.C + DO))


def proj_fwd_bf16(a, w1, b1, w2, b2, out, d, DO, act=0):
    assert a.dtype == torch.bfloat16
    _lib.call("rpb_proj_fwd_bf16", _p(a, torch.bfloat16), _p(w1), _p(b1), _p(w2), _p(b2), _p(out), d.ncrop, d.C, DO, *d.crop6,
              int(act), _stream(), label="proj_fwd_bf16", nbytes=d.ncrop * (2 * d.C + 4 * DO), flops=2 * d.ncrop * 128 * (d.C + DO))


def proj_bwd(a, w1, b1, w2, b2, gout, gu, part, d, DO, xf=None, act=0):
    _lib.call("rpb_proj_bwd", _p(a), _p(w1), _p(b1), _p(w2), _p(b2), _p(gout), _p(gu), _p(part), d.ncrop, d.C, DO,
              *d.crop6, *_xf(xf), int(act), _stream(), label="proj_bwd", nbytes=4 * d.ncrop * (d.C + DO + 128),
              flops=2 * d.ncrop * 128 * (d.C + 2 * DO))


def mse_rows():
    return _lib.query("rpb_mse_rows")


def mse(pred, target, elem, gout, part, n, gscale):
    _lib.call("rpb_mse", _p(pred), _p(target), _p(elem), _p(gout), _p(part), n, float(gscale), _stream())


def adam_step(p, g, m, v, n, lr, beta1, beta2, eps, step, gscale=1.0):
    _lib.call("rpb_adam_step", _p(p), _p(g), _p(m), _p(v), n, lr, beta1, beta2, eps, step, gscale, _stream(),
              label="adam_step", nbytes=28 * n, flops=12 * n)


def adam_step_ranges(p, g, m, v, tab, nr, total, lr, beta1, beta2, eps, step, gscale=1.0):
    """Adam on the ranges listed in ``tab`` (device int64 [nr][2]: first element, float4 groups before the range): the sharded optimizer step."""
    _lib.call("rpb_adam_step_ranges", _p(p), _p(g), _p(m), _p(v), _p(tab, torch.int64), nr, total, lr, beta1, beta2, eps, step, gscale, _stream(),
              label="adam_step", nbytes=28 * total, flops=12 * total)


def rollout_affine(pred, para, out, ncell, Cp, Cx, mean_t, std_t, mean_i, std_i):
    _lib.call("rpb_rollout_affine", _p(pred), _p(para), _p(out), ncell, Cp, Cx, _p(mean_t), _p(std_t), _p(mean_i),
              _p(std_i), _stream())


def channel_affine(inp, out, n, C, mean, std, inverse):
    _lib.call("rpb_channel_affine", _p(inp), _p(out), n, C, _p(mean), _p(std), int(inverse), _stream())


# ----------------------------------------------------------------------------- Transolver forward kernels
def mul(a, b, out, n):
    _lib.call("rpb_mul", _p(a), _p(b), _p(out), n, _stream(), label="mul", nbytes=12 * n)


GEMM_SPLIT = os.environ.get("RPB_GEMM_EXACT", "0") != "1"
GEMM_SPLIT_WIDE_N = 8192               # ... unless the output is wide enough to fill the chip from few rows (DPOT's TimeAggregator data
                                       # gradient, M = 4096, N = 20480: 167 vs 124 TF/s; its N = 1024 GEMMs: 83-108 vs 93-118, stay fp32)
# K <= 256, N <= 256 with an epilogue that reads or writes a second [M][N] tensor: the exact-fp32 kernel is faster (2.6 M rows, GELU +
# residual: 3.43 vs 4.12 ms; plain: 3.14 vs 2.49 -- DESIGN.md section 9)
GEMM_F32_SHORT_HEAVY = os.environ.get("RPB_GEMM_F32_SHORT_HEAVY", "1") != "0"
GEMM3X_V2 = os.environ.get("RPB_GEMM3X_V2", "1") != "0"
GEMM3X_V2_MIN_ROWS = 4096
GEMM_SPLIT_MIN_ROWS = 65536           # below this the GEMM is launch-bound and the weight preparation does not pay


GEMM_SPLIT_MIN_K = int(os.environ.get("RPB_GEMM_SPLIT_MIN_K", 256))                # K = 256 (4 LDS stages per 128-row tile): 110-117 vs 104-108 TF/s, larger K 145-160 vs 120-125;
GEMM_SPLIT_MIN_N = 256                # the K-split tiles of N = 64 / 128 lose to the fp32 kernel (64 vs 96 TF/s)


def gemm_split_ok(M, N, K, lda, ldo, conv, v2=False):
    # the 64-row-tile variant fills the chip from 4096 rows (DPOT's token GEMMs: 0.063 vs 0.090 ms at N = K = 1024, tools/nt_small.py)
    big = M >= GEMM_SPLIT_MIN_ROWS or (M >= 2048 and N >= GEMM_SPLIT_WIDE_N) or (v2 and M >= GEMM3X_V2_MIN_ROWS)
    return (GEMM_SPLIT and not conv and big and K % 64 == 0 and K >= (64 if v2 else GEMM_SPLIT_MIN_K)
            and (N in (64, 128) or N % 256 == 0 or (v2 and N % 128 == 0)) and N >= (128 if v2 else GEMM_SPLIT_MIN_N)
            and lda % 4 == 0 and ldo % 4 == 0)


def gemm_nt(A, W, out, M, N, K, bias=None, addvec=None, residual=None, act=0, lda=None, ldo=None, conv=None, aux=None,
            pre_out=None, mask=None, conv_mode=1, cls=0, drop=None):
    """out[M,N] = epilogue(A[M,K] @ W[N,K]^T); ``conv=(Hc,Wc,Dc)`` makes A an implicit im2col of a token tensor:
    ``conv_mode`` 1 = 3x3x3 padding 1, 2 = (1,4,4) stride (1,2,2) padding (0,1,1) (M = output tokens), 3 = parity class
    ``cls`` of the matching transposed convolution (M = input tokens).
    act=1: GELU (``pre_out`` optionally receives the pre-activation); act=2: multiply by gelu'(``aux``); 3: ReLU; 4: ReLU'.
    ``mask``: inverted-dropout multiplier tensor, or ``drop=(seed, keep)``: the same dropout generated in the epilogue.
    Plain GEMMs with K % 64 == 0 run on the bf16 MFMA from split fp32 operands (csrc/rpb_gemm3x.hip, fp32-grade accuracy): the
    64-row-tile organisation (csrc/rpb_gemm3x2.hip) from 4096 rows and K >= 64 for N % 256 == 0 without a mask tensor and for
    N % 128 == 0 without dropout; the 128-row kernel for N = 64 / 128 / 256 k from K >= 256.  The weight's three bf16 planes are
    prepared per call (rpb_gemm3x_wprep, N*K elements: 0.01 ms at 256 x 256); RPB_GEMM_EXACT=1 keeps everything on the exact-fp32 kernel."""
    hc, wc, dc = conv if conv else (0, 0, 0)
    mode = int(conv_mode) if conv else 0
    taps = {0: 1, 1: 27, 2: 16, 3: 4}[mode]
    lda = (K // taps) if lda is None else lda
    ldo = N if ldo is None else ldo
    heavy_epilogue = residual is not None or aux is not None or mask is not None or pre_out is not None
    # rpb_gemm3x's 64-row-tile variant (csrc/rpb_gemm3x2.hip: N % 256 == 0 without a mask tensor, N % 128 == 0 without dropout) hides
    # the epilogue; only the 128-row kernel loses to the fp32 one on short products with a heavy epilogue
    v2 = GEMM3X_V2 and mask is None and (N % 256 == 0 or (N % 128 == 0 and not drop))
    if gemm_split_ok(M, N, K, lda, ldo, conv, v2) and not (GEMM_F32_SHORT_HEAVY and K <= 256 and N <= 256 and heavy_epilogue and not v2):
        wsrc = W.t if isinstance(W, Sub) else W
        wz = torch.empty(3 * N * K, dtype=torch.int16, device=wsrc.device)
        _lib.call("rpb_gemm3x_wprep", _p(W), _p(wz, torch.int16), N, K, _stream(), label="gemm3x_wprep", nbytes=10 * N * K)
        _lib.call("rpb_gemm3x", _p(A), _p(wz, torch.int16), _p(bias), _p(addvec), _p(residual), _p(out), M, N, K, lda, ldo,
                  int(act), _p(aux), _p(pre_out), _p(mask), int(drop[0]) if drop else 0, float(drop[1]) if drop else 0.0,
                  _stream(), label=f"gemm3x[N{N},K{K}]", nbytes=4 * (M * lda + M * N + N * K), flops=2 * M * N * K)
        return
    _lib.call("rpb_gemm_nt", _p(A), _p(W), _p(bias), _p(addvec), _p(residual), _p(out), M, N, K, lda,
              ldo, int(act), _p(aux), _p(pre_out), _p(mask), mode, hc, wc, dc, int(cls),
              int(drop[0]) if drop else 0, float(drop[1]) if drop else 0.0, _stream(),
              label=f"gemm_nt[N{N},K{K},conv={mode}]", nbytes=4 * (M * lda + M * N + N * K),
              flops=2 * M * N * K)


def gemm_tn_split_bf16(M, N, K, ldg=None, lda=None):
    """True when the plain weight-gradient GEMM runs on the bf16 matrix pipe from split operands (csrc/rpb_gemm3x_tn.hip)."""
    return bool(_lib.query("rpb_gemm3x_tn_supported", M, N, K, N if ldg is None else ldg, K if lda is None else lda))


def gemm_tn_splits(M, N, K, conv=False, conv_mode=1, ldg=None, lda=None):
    if not conv and gemm_tn_split_bf16(M, N, K, ldg, lda):
        return _lib.query("rpb_gemm3x_tn_splits", M, N, K)
    return _lib.query("rpb_gemm_tn_splits", M, N, K, int(conv_mode) if conv else 0)


def gemm_tn(G, A, part, M, N, K, ldg=None, lda=None, conv=None, conv_mode=1):
    """part[splits][N*K + N]: partials of dW = G^T A(im2col) and db = colsum(G)."""
    if conv is None and gemm_tn_split_bf16(M, N, K, ldg, lda):
        if part.shape[0] != _lib.query("rpb_gemm3x_tn_splits", M, N, K):
            raise _lib.RpbError("gemm_tn: size the partials with gemm_tn_splits(M, N, K, ldg=, lda=) of the same leading dimensions")
        _lib.call("rpb_gemm3x_tn", _p(G), _p(A), _p(part), M, N, K, N if ldg is None else ldg, K if lda is None else lda, _stream(),
                  label=f"gemm3x_tn[N{N},K{K}]", nbytes=4 * M * (N + K), flops=2 * M * N * K)
        return
    hc, wc, dc = conv if conv else (0, 0, 0)
    mode = int(conv_mode) if conv else 0
    taps = {0: 1, 1: 27, 2: 16}[mode]
    _lib.call("rpb_gemm_tn", _p(G), _p(A), _p(part), M, N, K, N if ldg is None else ldg,
              (K // taps) if lda is None else lda, mode, hc, wc, dc, _stream(),
              label=f"gemm_tn[N{N},K{K},conv={mode}]", nbytes=4 * M * (N + K // taps),
              flops=2 * M * N * K)


def layernorm_bwd_rows(M):
    return _lib.query("rpb_layernorm_bwd_rows", M)


def layernorm_bwd(x, gamma, gy, gadd, gx, part, M, C, eps=1e-5):
    _lib.call("rpb_layernorm_bwd", _p(x), _p(gamma), _p(gy), _p(gadd), _p(gx), _p(part), M, C, eps, _stream(),
              label="layernorm_bwd", nbytes=(12 + (4 if gadd is not None else 0)) * M * C, flops=16 * M * C)


def slice_bwd(xf, w, gox, tok2, gT, gN, Ws, temp, gxf, part, B, ntok, heads, G):
    _lib.call("rpb_slice_bwd", _p(xf), _p(w), _p(gox), _p(tok2), _p(gT), _p(gN), _p(Ws), _p(temp), _p(gxf), _p(part), B,
              ntok, heads, G, _stream(), label="slice_bwd", nbytes=4 * B * ntok * heads * (32 * 5 + G),
              flops=2 * B * ntok * heads * G * 32 * 5)


def colsum_rows():
    return _lib.query("rpb_colsum_rows")


def colsum(x, part, M, N, ld=None):
    _lib.call("rpb_colsum", _p(x), _p(part), M, N, N if ld is None else ld, _stream(), label="colsum", nbytes=4 * M * N)


def tokens_lift(x, W, b, out, M, K, N, act):
    _lib.call("rpb_tokens_lift", _p(x), _p(W), _p(b), _p(out), M, K, N, int(act), _stream(), label="tokens_lift",
              nbytes=4 * M * (K + N), flops=2 * M * K * N)


def layernorm_fwd(x, gamma, beta, out, M, C, eps=1e-5):
    _lib.call("rpb_layernorm_fwd", _p(x), _p(gamma), _p(beta), _p(out), M, C, eps, _stream(), label="layernorm_fwd",
              nbytes=8 * M * C, flops=8 * M * C)


def slice_blocks_per_sample(B):
    return _lib.query("rpb_slice_blocks_per_sample", B)


def slice_fwd(xf, Ws, bs, temp, w_out, tok_part, norm_part, B, ntok, heads, G, ldx, w_in=None):
    _lib.call("rpb_slice_fwd", _p(xf), _p(Ws), _p(bs), _p(temp), _p(w_out), _p(tok_part), _p(norm_part), B, ntok, heads,
              G, ldx, _p(w_in), _stream(), label="slice_fwd", nbytes=4 * B * ntok * (ldx + 2 * heads * G),
              flops=2 * B * ntok * heads * G * 64)


def slice_attn(tokS, norm, Wq, Wk, Wv, out, BH, G):
    _lib.call("rpb_slice_attn", _p(tokS), _p(norm), _p(Wq), _p(Wk), _p(Wv), _p(out), BH, G, _stream(), label="slice_attn")


def slice_attn_train(tokS, norm, Wq, Wk, Wv, amask, BH, G, out=None, go=None, gT=None, gN=None, gW=None):
    """Training-mode slice-token attention (attention-map dropout mask ``amask`` or None) and, with ``go``, its backward."""
    _lib.call("rpb_slice_attn_train", _p(tokS), _p(norm), _p(Wq), _p(Wk), _p(Wv), _p(amask), _p(go), _p(out), _p(gT), _p(gN),
              _p(gW), BH, G, _stream(), label="slice_attn_train")


def deslice_fwd(w, tok2, out, B, ntok, heads, G):
    _lib.call("rpb_deslice_fwd", _p(w), _p(tok2), _p(out), B, ntok, heads, G, _stream(), label="deslice_fwd",
              nbytes=4 * B * ntok * heads * (G + 32), flops=2 * B * ntok * heads * G * 32)


# ----------------------------------------------------------------------------- Galerkin Transformer kernels
def headnorm_fwd(x, ldx, gamma, beta, out, ldo, M, C, eps, col0=0, ocol0=0):
    """Per-head LayerNorm(64) of columns [col0, col0+C) of the row-major token tensor ``x`` (leading dim ``ldx``)."""
    _lib.call("rpb_headnorm_fwd", _p(x) + 4 * col0, ldx, _p(gamma), _p(beta), _p(out) + 4 * ocol0, ldo, M, C, eps,
              _stream(), label="headnorm_fwd", nbytes=8 * M * C, flops=10 * M * C)


def headnorm_bwd_rows(M):
    return _lib.query("rpb_headnorm_bwd_rows", M)


def headnorm_bwd(x, ldx, gamma, gy, ldg, gx, ldgx, part, M, C, eps, col0=0, gcol0=0, xcol0=0):
    _lib.call("rpb_headnorm_bwd", _p(x) + 4 * col0, ldx, _p(gamma), _p(gy) + 4 * gcol0, ldg, _p(gx) + 4 * xcol0, ldgx,
              _p(part), M, C, eps, _stream(), label="headnorm_bwd", nbytes=12 * M * C, flops=20 * M * C)


def pad_grid_fwd(U, grids, Wg, bias, out, d):
    _lib.call("rpb_pad_grid_fwd", _p(U), _p(grids[0]), _p(grids[1]), _p(grids[2]), _p(Wg), _p(bias), _p(out), d.B, d.T,
              d.H, d.W, d.C, d.Tp, d.Hp, d.Wp, _stream(), label="pad_grid_fwd", nbytes=4 * d.C * (d.ncrop + d.ncell))


def crop_gather(g, out, d):
    _lib.call("rpb_crop_gather", _p(g), _p(out), d.B, d.T, d.H, d.W, d.C, d.Tp, d.Hp, d.Wp, _stream(),
              label="crop_gather", nbytes=8 * d.C * d.ncrop)


def head_scores_chunks(B, n):
    return _lib.query("rpb_head_scores_chunks", B, n)


def head_scores(G, ldg, A, lda, part, B, n, nheads=4):
    """part[chunks][B][nheads][64][64]: per-chunk partials of G_h^T A_h for every sample and 64-channel head."""
    _lib.call("rpb_head_scores", _p(G), ldg, _p(A), lda, _p(part), B, n, nheads, _stream(), label="head_scores",
              nbytes=4 * B * n * 128 * nheads, flops=2 * B * n * nheads * 64 * 64)


def head_apply(X, ldx, Wm, out, ldo, B, n, residual=None, ldr=0, mask=None, ldm=0, nheads=4, drop=None):
    """out[b,m][64h+j] = (sum_i X[b,m][64h+i] Wm[b][h][i][j]) * mask + residual (``drop=(seed, keep)``: in-kernel mask)."""
    _lib.call("rpb_head_apply", _p(X), ldx, _p(Wm), _p(out), ldo, _p(residual), ldr, _p(mask), ldm, B, n, nheads,
              int(drop[0]) if drop else 0, float(drop[1]) if drop else 0.0, _stream(), label="head_apply",
              nbytes=4 * B * n * 64 * nheads * (2 + (residual is not None) + (mask is not None)),
              flops=2 * B * n * nheads * 64 * 64)


# ----------------------------------------------------------------------------- U-Net kernels
def chan_blocks(B, n):
    return _lib.query("rpb_chan_blocks", B, n)


def chan_stats(x, part, B, n, C):
    _lib.call("rpb_chan_stats", _p(x), _p(part), B, n, C, _stream(), label="chan_stats", nbytes=4 * B * n * C)


def gn_affine_fwd(sums64, gamma, beta, ss, count, eps, A, Bc, stat, B, C, G):
    _lib.call("rpb_gn_affine_fwd", _p(sums64, torch.float64), _p(gamma), _p(beta), _p(ss), float(count), float(eps), _p(A), _p(Bc),
              _p(stat), B, C, G, _stream())


def gn_affine_bwd(d, stat, gamma, beta, ss, count, dgam, dbet, dss, P, Q, B, C, G):
    _lib.call("rpb_gn_affine_bwd", _p(d), _p(stat), _p(gamma), _p(beta), _p(ss), float(count), _p(dgam), _p(dbet), _p(dss), _p(P), _p(Q),
              B, C, G, _stream())


def silu_fwd(x, y):
    _lib.call("rpb_silu_fwd", _p(x), _p(y), x.numel(), _stream())


def silu_bwd(x, gy, gx):
    _lib.call("rpb_silu_bwd", _p(x), _p(gy), _p(gx), x.numel(), _stream())


def relpos_bias_fwd(table, idx, bias, n2, heads):
    _lib.call("rpb_relpos_bias_fwd", _p(table), _p(idx, torch.int32), _p(bias), n2, heads, _stream())


def relpos_bias_bwd(gbias, idx, gtable, n2, heads, nbuckets):
    _lib.call("rpb_relpos_bias_bwd", _p(gbias), _p(idx, torch.int32), _p(gtable), n2, heads, nbuckets, _stream())


def affine_silu_fwd(x, A, Bc, y, B, n, C, res=None):
    _lib.call("rpb_affine_silu_fwd", _p(x), _p(A), _p(Bc), _p(res), _p(y), B, n, C, _stream(), label="affine_silu_fwd",
              nbytes=(8 + 4 * (res is not None)) * B * n * C)


def affine_silu_bwd_reduce(x, gy, A, Bc, part, B, n, C):
    _lib.call("rpb_affine_silu_bwd_reduce", _p(x), _p(gy), _p(A), _p(Bc), _p(part), B, n, C, _stream(),
              label="affine_silu_bwd_reduce", nbytes=8 * B * n * C)


def affine_silu_bwd_apply(x, gy, A, Bc, P, Q, gx, B, n, C):
    _lib.call("rpb_affine_silu_bwd_apply", _p(x), _p(gy), _p(A), _p(Bc), _p(P), _p(Q), _p(gx), B, n, C, _stream(),
              label="affine_silu_bwd_apply", nbytes=12 * B * n * C)


def split3(x, planes, M, C, ldx=None):
    """planes[3][M][C] (bf16 bit patterns in an int16 tensor) = hi / mid / lo terms of x[M][ldx]."""
    _lib.call("rpb_split3", _p(x), _p(planes, torch.int16), M, C, C if ldx is None else ldx, _stream(), label="split3", nbytes=10 * M * C)


def conv3x_wprep(W, Wz, N, Ci):
    _lib.call("rpb_conv3x_wprep", _p(W), _p(Wz, torch.int16), N, Ci, _stream(), label="conv3x_wprep", nbytes=10 * N * 27 * Ci)


def conv3x(planes, Wz, out, M, N, Ci, mesh, bias=None, ldo=None):
    """out[M][ldo] = Conv3d(Ci, N, 3, padding=1)(tokens) + bias on the bf16 MFMA from split operands (fp32-grade accuracy)."""
    hc, wc, dc = mesh
    _lib.call("rpb_conv3x", _p(planes, torch.int16), _p(Wz, torch.int16), _p(bias), _p(out), M, N, Ci, N if ldo is None else ldo, hc, wc, dc, _stream(),
              label=f"conv3x[N{N},Ci{Ci}]", nbytes=6 * M * Ci * 9 + 4 * M * N, flops=2 * M * N * 27 * Ci)


def split3t(x, planes_t, M, C, ldx=None, rev_mesh=None):
    """planes_t[3][M/8][C][8] (bf16 bit patterns): hi / mid / lo terms of x[M][ldx] in runs of 8 tokens per channel;
    ``rev_mesh=(d0, d1, d2)``: in the token order of the reversed mesh (d2, d1, d0)."""
    d0, d1, d2 = rev_mesh if rev_mesh else (0, 0, 0)
    _lib.call("rpb_split3t", _p(x), _p(planes_t, torch.int16), M, C, C if ldx is None else ldx, 1 if rev_mesh else 0, d0, d1, d2,
              _stream(), label="split3t", nbytes=10 * M * C)


def conv3x_wgrad_splits(M, Co, Ci):
    return _lib.query("rpb_conv3x_wgrad_splits", M, Co, Ci)


def conv3x_wgrad(Gt, Xt, part, M, Co, Ci, mesh):
    hc, wc, dc = mesh
    _lib.call("rpb_conv3x_wgrad", _p(Gt, torch.int16), _p(Xt, torch.int16), _p(part), M, Co, Ci, hc, wc, dc, _stream(),
              label=f"conv3x_wgrad[Co{Co},Ci{Ci}]", nbytes=6 * M * (Co + 3 * Ci), flops=2 * M * Co * 27 * Ci)


CONV3_SPLIT = os.environ.get("RPB_CONV3_EXACT", "0") != "1"


def conv3_wgrad_split_mode(Co, Ci, mesh, M):
    """0: exact-fp32 kernel; 1: split-bf16 kernel; 2: split-bf16 kernel on the reversed mesh (innermost dimension not % 8)."""
    if not (CONV3_SPLIT and Co % 64 == 0 and Ci % 64 == 0 and M % 8 == 0):
        return 0
    if mesh[2] % 8 == 0 and mesh[2] >= 16:
        return 1
    if mesh[0] % 8 == 0 and mesh[0] >= 16:
        return 2
    return 0


def conv3_wgrad_parts(G, X, M, Co, Ci, mesh, ldg=None, ldx=None):
    """(part, taps_reversed): per-split partials [splits][Co*27*Ci + Co] of (dW, db) of Conv3d(Ci, Co, 3, padding=1): split-bf16
    kernel where the shape allows, else the exact-fp32 LDS-tiled kernel (rpb_gemm_tn conv mode 1).  ``taps_reversed``: the
    three tap axes of dW are in (kw, kh, kt) order (see ``conv3_taps_restore``)."""
    K = 27 * Ci
    mode = conv3_wgrad_split_mode(Co, Ci, mesh, M)
    if mode:
        gt = torch.empty(3 * Co * M, dtype=torch.int16, device=G.device)
        xt = torch.empty(3 * Ci * M, dtype=torch.int16, device=G.device)
        rev = tuple(mesh) if mode == 2 else None
        split3t(G, gt, M, Co, ldg, rev_mesh=rev)
        split3t(X, xt, M, Ci, ldx, rev_mesh=rev)
        part = torch.empty(conv3x_wgrad_splits(M, Co, Ci), Co * K + Co, device=G.device)
        conv3x_wgrad(gt, xt, part, M, Co, Ci, tuple(mesh)[::-1] if mode == 2 else mesh)
        return part, mode == 2
    part = torch.empty(gemm_tn_splits(M, Co, K, conv=True), Co * K + Co, device=G.device)
    gemm_tn(G, X, part, M, Co, K, ldg=ldg, lda=ldx, conv=mesh)
    return part, False


def conv3_taps_restore(dW, Co, Ci):
    """dW[Co][27*Ci] computed on the reversed mesh -> the (kt, kh, kw) tap order of the original one."""
    return dW.view(Co, 3, 3, 3, Ci).permute(0, 3, 2, 1, 4).reshape(Co, 27 * Ci).contiguous()



def conv3_split_ok(N, Ci):
    return CONV3_SPLIT and Ci % 64 == 0 and (N in (64, 128) or N % 256 == 0)


def conv3(x, W, out, M, N, Ci, mesh, bias=None, ldx=None):
    """out[M][N] = Conv3d(Ci, N, 3, padding=1)(x tokens [M][ldx]) + bias with W[N][27*Ci] (tap-major rows) -- forward, or the
    data gradient when W holds the flipped / transposed taps.  Shapes the split-bf16 kernel covers (Ci % 64 == 0, N = 64, 128
    or 256 k) run on the bf16 MFMA from hi + mid + lo operands (csrc/rpb_conv3x.hip: fp32-grade accuracy, ~2x the fp32 MFMA
    rate); everything else, or RPB_CONV3_EXACT=1, takes the exact-fp32 implicit GEMM (rpb_gemm_nt conv mode 1)."""
    if not conv3_split_ok(N, Ci):
        return gemm_nt(x, W, out, M, N, 27 * Ci, bias=bias, conv=mesh, lda=ldx)
    planes = torch.empty(3 * M * Ci, dtype=torch.int16, device=out.device)
    wz = torch.empty(3 * N * 27 * Ci, dtype=torch.int16, device=out.device)
    split3(x, planes, M, Ci, ldx)
    conv3x_wprep(W, wz, N, Ci)
    conv3x(planes, wz, out, M, N, Ci, mesh, bias=bias)


def window_pack(planar, cl, flags, inp, tgt, B, horizon, in_step, Hf, Wf, sub_s, n_para, Cp, Cl, mean_in, mean_tgt, std_in,
                std_tgt, rows_subsampled=False):
    """Arrow time slabs (planar cells + optional channels-last cell) -> normalised channels-last (input, target) batch in one
    HBM pass (disk.DiskBatchLoader)."""
    _lib.call("rpb_window_pack", _p(planar), _p(cl), _p(flags), _p(inp), _p(tgt), B, horizon, in_step, Hf, Wf, sub_s,
              int(rows_subsampled), n_para, Cp, Cl, _p(mean_in), _p(mean_tgt), _p(std_in), _p(std_tgt), _stream(), label="window_pack",
              nbytes=4 * (planar.numel() + (cl.numel() if cl is not None else 0) + inp.numel() + tgt.numel()))


def pair_pack(num, real, para, inp, tgt, B, ntok, Cl, n_para, mean_in, mean_tgt, std_in, std_tgt):
    """Combustion surrogate batch: channels-last `numerical` windows + sim_id parameter channels -> normalised input, `real`
    windows -> normalised single-channel target, one HBM pass (disk.SurrogateBatchLoader)."""
    _lib.call("rpb_pair_pack", _p(num), _p(real), _p(para), _p(inp), _p(tgt), B, ntok, Cl, n_para, _p(mean_in), _p(mean_tgt),
              _p(std_in), _p(std_tgt), _stream(), label="pair_pack", nbytes=4 * (num.numel() + real.numel() + inp.numel() + tgt.numel()))


def im2col(x, col, B, T, H, W, Cin, KS, ldc):
    _lib.call("rpb_im2col", _p(x), _p(col), B, T, H, W, Cin, KS, ldc, _stream(), label="im2col",
              nbytes=4 * B * T * H * W * (Cin + ldc))


def tattn_blocks(nloc):
    return _lib.query("rpb_tattn_blocks", nloc)


def tattn_fwd(qkv, rcos, rsin, bias, out, B, T, HW):
    _lib.call("rpb_tattn_fwd", _p(qkv), _p(rcos), _p(rsin), _p(bias), _p(out), B, T, HW, _stream(), label="tattn_fwd",
              nbytes=4 * B * T * HW * 512, flops=4 * B * HW * 4 * T * T * 32)


def tattn_bwd(qkv, rcos, rsin, bias, go, gqkv, part, B, T, HW):
    _lib.call("rpb_tattn_bwd", _p(qkv), _p(rcos), _p(rsin), _p(bias), _p(go), _p(gqkv), _p(part), B, T, HW, _stream(),
              label="tattn_bwd", nbytes=4 * B * T * HW * 896, flops=10 * B * HW * 4 * T * T * 32)


def sattn_fwd(qkv, out, lse, F, n):
    _lib.call("rpb_sattn_fwd", _p(qkv), _p(out), _p(lse), F, n, _stream(), label="sattn_fwd",
              nbytes=4 * F * n * 512, flops=4 * F * 4 * n * n * 32)


def sattn_bwd(qkv, o, go, lse, gqkv, F, n):
    _lib.call("rpb_sattn_bwd", _p(qkv), _p(o), _p(go), _p(lse), _p(gqkv), F, n, _stream(), label="sattn_bwd",
              nbytes=4 * F * n * 1024, flops=14 * F * 4 * n * n * 32)


def linattn_prep_fwd(qkv, kmax, qe, F, n):
    _lib.call("rpb_linattn_prep_fwd", _p(qkv), _p(kmax), _p(qe), F, n, _stream(), label="linattn_prep_fwd",
              nbytes=4 * F * n * 512)


def linattn_prep_bwd(qe, dqe, dz, gqkv, F, n):
    _lib.call("rpb_linattn_prep_bwd", _p(qe), _p(dqe), _p(dz), _p(gqkv), F, n, _stream(), label="linattn_prep_bwd",
              nbytes=4 * F * n * 768)


def col_reduce(x, ldx, part, F, n, C, mode):
    """mode 0: per-frame column max partials, 1: column sum partials -- part[chan_blocks(F, n)][F][C]."""
    _lib.call("rpb_col_reduce", _p(x), ldx, _p(part), F, n, C, int(mode), _stream(), label="col_reduce",
              nbytes=4 * F * n * C)


def dropout_mul(g, out, n, seed, keep):
    """out = g * inverted-dropout mask regenerated from (seed, element index): backward of the in-kernel dropout."""
    _lib.call("rpb_dropout_mul", _p(g), _p(out), n, int(seed), float(keep), _stream(), label="dropout_mul", nbytes=8 * n)


def spectrum_bin(Y, out, R, NB):
    _lib.call("rpb_spectrum_bin", _p(Y), _p(out), R, NB, _stream(), label="spectrum_bin", nbytes=8 * R * R * R * NB)


def proj_bwd_fused_supported(C, DO, W, Wp):
    return bool(_lib.query("rpb_proj_bwd_fused_supported", C, DO, W, Wp))


def proj_dgrad_slots(d):
    return _lib.query("rpb_proj_dgrad_slots", d.B, d.Tp, d.Hp)


def proj_dgrad(s, w1, b1, w2, gout, g, stats_part, d, DO, xf, act=0, gu=None):
    """g [ncell][64] (padded layout) = fc1^T gh, gh = (fc2^T gout) * act'(fc1 a + b1) recomputed (a = xf(s) cropped) or read from
    ``gu`` [ncrop][128]; stats_part = BN-backward sums."""
    _lib.call("rpb_proj_dgrad", _p(s), _p(w1), _p(b1), _p(w2), _p(gout), _p(gu), _p(g), _p(stats_part), d.B, DO, *d.crop6,
              *_xf(xf), int(act), _stream(), label="proj_dgrad[gu]" if gu is not None else "proj_dgrad[recompute]",
              nbytes=4 * (d.ncrop * (d.C + (128 if gu is not None else DO)) + d.ncell * d.C),
              flops=2 * d.ncrop * 128 * (d.C if gu is not None else 2 * d.C))


def proj_wgrad_slots(d):
    return _lib.query("rpb_proj_wgrad_slots", d.B, d.T, d.H)


def proj_wgrad_row(DO):
    return _lib.query("rpb_proj_wgrad_row", DO)


def proj_wgrad_roles():
    return _lib.query("rpb_proj_wgrad_roles")


def proj_wgrad(s, w1, b1, w2, gout, part, d, DO, xf, act=0):
    """Per-wave partial rows of d fc1.weight / d fc2.weight / d fc1.bias / d fc2.bias (layout: include/rpb.h)."""
    _lib.call("rpb_proj_wgrad", _p(s), _p(w1), _p(b1), _p(w2), _p(gout), _p(part), d.B, DO, *d.crop6, *_xf(xf), int(act),
              _stream(), label="proj_wgrad", nbytes=4 * d.ncrop * (d.C + DO), flops=2 * d.ncrop * 128 * 2 * d.C)


def stream_probe(a, b, c, out, nread, threads=256):
    """Measurement aid (bench.py): out = a (* b (+ c)), ``nread`` tensors read + one written, 16 B per lane."""
    _lib.call("rpb_stream_probe", _p(a), _p(b), _p(c), _p(out), a.numel(), nread, threads, _stream(), label=f"stream_probe[R{nread}W1]",
              nbytes=4 * a.numel() * (nread + 1))


def mfma_probe(seed, out, iters, waves_per_simd=1):
    """Measurement aid (bench.py): sustained bf16 MFMA rate on operands built from ``seed`` (4096 floats); returns the launch's flops."""
    import ctypes
    fl = ctypes.c_double(0.0)
    _lib.call("rpb_mfma_probe", _p(seed), _p(out), int(iters), int(waves_per_simd), ctypes.addressof(fl), _stream())
    return fl.value


def head_bwd_supported(C, DO, W, Wp, xf_gelu, act):
    return bool(_lib.query("rpb_head_bwd_supported", C, DO, W, Wp, int(bool(xf_gelu)), int(act)))


def head_bwd_slots(d):
    return _lib.query("rpb_head_bwd_slots", d.B, d.T, d.H)


def head_bwd_row(DO):
    return _lib.query("rpb_head_bwd_row", DO)


def head_bwd(s, w1, b1, w2, gout, g, part, d, DO, xf):
    """The projection head's backward in one pass (csrc/rpb_pjf.hip): g [ncell][64] + per-wave partial rows
    [M = gh^T shat | d fc2.weight | d fc1.bias | d fc2.bias]; gh never reaches HBM."""
    mean, invstd, gamma, beta, gelu = xf
    if gelu:
        raise _lib.RpbError("head_bwd: the layer in front of the head must not end in GELU")
    _lib.call("rpb_head_bwd", _p(s), _p(w1), _p(b1), _p(w2), _p(gout), _p(g), _p(part), d.B, DO, *d.crop6, _p(mean), _p(invstd),
              _p(gamma), _p(beta), _stream(), label="head_bwd", nbytes=4 * (d.ncrop * (d.C + DO) + d.ncell * d.C),
              flops=2 * d.ncrop * 128 * 3 * d.C)


def head_fwd_bwd(s, w1, b1, w2, b2, target, gscale, g, part, loss_part, d, DO, xf):
    """The training step's head in one launch: forward + squared-error loss + dLoss/dout + the whole backward (csrc/rpb_pjf.hip)."""
    mean, invstd, gamma, beta, gelu = xf
    if gelu:
        raise _lib.RpbError("head_fwd_bwd: the layer in front of the head must not end in GELU")
    _lib.call("rpb_head_fwd_bwd", _p(s), _p(w1), _p(b1), _p(w2), _p(b2), _p(target), float(gscale), _p(g), _p(part), _p(loss_part), d.B, DO,
              *d.crop6, _p(mean), _p(invstd), _p(gamma), _p(beta), _stream(), label="head_fwd_bwd",
              nbytes=4 * (d.ncrop * (d.C + DO) + d.ncell * d.C), flops=2 * d.ncrop * 128 * (3 * d.C + DO))


def head_bwd_finalize(tot, w1, gamma, beta, DO, dw1, dw2, db1, db2, bn_sums):
    _lib.call("rpb_head_bwd_finalize", _p(tot), _p(w1), _p(gamma), _p(beta), DO, _p(dw1), _p(dw2), _p(db1), _p(db2), _p(bn_sums),
              _stream(), label="head_bwd_finalize")


def add(a, b):
    """a + b as a new tensor (HIP kernel; shapes equal, numel % 4 == 0 else torch adds the handful of numbers)."""
    if a.numel() % 4 or not a.is_cuda or a.dtype != torch.float32 or not (a.is_contiguous() and b.is_contiguous()):
        return a + b
    out = torch.empty_like(a)
    _lib.call("rpb_add", _p(a), _p(b), _p(out), a.numel(), _stream(), label="add", nbytes=12 * a.numel())
    return out


def copy_cols(src, dst, M, C, lds, ldd, soff, doff):
    _lib.call("rpb_copy_cols", _p(src), _p(dst), M, C, lds, ldd, soff, doff, _stream(), label="copy_cols", nbytes=8 * M * C)


def feat_mix(Phi, w0, b0, Xh, B, M2, NB, Cin, C):
    _lib.call("rpb_feat_mix", _p(Phi), _p(w0), _p(b0), _p(Xh), B, M2, NB, Cin, C, _stream(), label="feat_mix",
              nbytes=4 * B * M2 * C)


def feat_mix_wgrad_rows():
    return _lib.query("rpb_feat_mix_wgrad_rows")


def feat_mix_wgrad(G, Phi, part, B, M2, NB, Cin, C):
    _lib.call("rpb_feat_mix_wgrad", _p(G), _p(Phi), _p(part), B, M2, NB, Cin, C, _stream(), label="feat_mix_wgrad",
              nbytes=4 * B * M2 * C)


def small_gemm(A, Bm, out, M, N, K, a_rs, a_cs, b_rs, b_cs, ldo, accumulate=False):
    """out[m][n] (+)= sum_k A[m*a_rs + k*a_cs] * Bm[k*b_rs + n*b_cs] (tiny matrices, one workgroup)."""
    _lib.call("rpb_small_gemm", _p(A), _p(Bm), _p(out), M, N, K, a_rs, a_cs, b_rs, b_cs, ldo, int(accumulate), _stream(),
              label="small_gemm")


def lift_feat(x, grids, out, d, FW):
    _lib.call("rpb_lift_feat", _p(x), _p(grids[0]), _p(grids[1]), _p(grids[2]), _p(out), d.B, d.T, d.H, d.W, d.Cin, d.Tp, d.Hp,
              d.Wp, FW, _stream(), label="lift_feat", nbytes=4 * (d.ncrop * d.Cin + d.ncell * FW))


def cell_mix_feat(phi, Wcomp, bias, z2, GW, out, stats_part, ncell, FW, K2, Wp, oxf=None):
    """Layer 0 on the feature tensor: out = GW z2 + Wcomp phi + bias (+ BatchNorm forward sums, or eval output transform)."""
    _lib.call("rpb_cell_mix_feat", _p(phi), _p(Wcomp), _p(bias), _p(z2), _p(GW), _p(out), _p(stats_part), ncell, FW, K2, Wp,
              *_xf(oxf), _stream(), label=f"cell_mix[feat{FW}->CO64,spec=1,stats={'oxf' if oxf is not None else int(stats_part is not None)}]",
              nbytes=4 * (ncell * (FW + 64) + (ncell // Wp) * K2 * 64), flops=2 * ncell * 64 * (K2 + FW))


def bn_bwd_row_feat(s, gy, phi, gs, mean, invstd, gamma, beta, sums, count, gelu, GWt, Y1, part, G, Wp, C, K2, FW):
    """``gs`` None: the BatchNorm-backward tensor is not stored (the fused trainer's layer 0: nothing reads it)."""
    _lib.call("rpb_bn_bwd_row_feat", _p(s), _p(gy), _p(phi), _p(gs), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(sums),
              float(count), int(gelu), _p(GWt), _p(Y1), _p(part), G, Wp, C, K2, FW, _stream(),
              label=f"bn_bwd_row[C{C},feat{FW}{'' if gs is not None else ',nogs'}]",
              nbytes=4 * ((3 if gs is not None else 2) * G * Wp * C + G * Wp * FW + G * K2 * C),
              flops=2 * G * Wp * C * (FW + K2))


# ---------------------------------------------------------------------------------------------- DPOT (csrc/rpb_dpot.hip)
def dpot_patch_tokens(u, gx, gy, gt, P, B, T, H, W, Cd, Cm, ps):
    _lib.call("rpb_dpot_patch_tokens", _p(u), _p(gx), _p(gy), _p(gt), _p(P), B, T, H, W, Cd, Cm, ps, _stream(),
              label="dpot_patch_tokens", nbytes=4 * (u.numel() + P.numel()))


def dpot_patch_tokens_bwd(gP, gu, B, T, H, W, Cd, Cm, ps):
    _lib.call("rpb_dpot_patch_tokens_bwd", _p(gP), _p(gu), B, T, H, W, Cd, Cm, ps, _stream(), label="dpot_patch_tokens_bwd",
              nbytes=4 * (gu.numel() + gP.numel()))


def rowtable_add(x, table, M, C, rows_per_entry, nent):
    _lib.call("rpb_rowtable_add", _p(x), _p(table), M, C, rows_per_entry, nent, _stream(), label="rowtable_add", nbytes=8 * M * C)


def rowtable_grad(g, dtable, B, C, rows_per_entry, nent):
    _lib.call("rpb_rowtable_grad", _p(g), _p(dtable), B, C, rows_per_entry, nent, _stream(), label="rowtable_grad",
              nbytes=4 * B * nent * rows_per_entry * C)


def dpot_tagg_prep(w, gamma, tt, Wf, Wb, e_out, T, C):
    _lib.call("rpb_dpot_tagg_prep", _p(w), _p(gamma), _p(tt), _p(Wf), _p(Wb), _p(e_out), T, C, _stream(), label="dpot_tagg_prep",
              nbytes=12 * T * C * C)


def dpot_tagg_finish(dWb, w, gamma, tt, dw, dgamma, T, C, dWsum=None):
    _lib.call("rpb_dpot_tagg_finish", _p(dWb), _p(dWsum), _p(w), _p(gamma), _p(tt), _p(dw), _p(dgamma), T, C, _stream(),
              label="dpot_tagg_finish", nbytes=12 * T * C * C)


def gn_tokens_fwd(x, x2, gamma, beta, y, stat, B, P, C, G, eps=1e-5):
    _lib.call("rpb_gn_tokens_fwd", _p(x), _p(x2), _p(gamma), _p(beta), _p(y), _p(stat), B, P, C, G, float(eps), _stream(),
              label="gn_tokens_fwd", nbytes=4 * B * P * C * (2 if x2 is None else 3))


def gn_tokens_bwd(x, x2, gamma, stat, gy, gadd, gx, pg, pb, B, P, C, G):
    _lib.call("rpb_gn_tokens_bwd", _p(x), _p(x2), _p(gamma), _p(stat), _p(gy), _p(gadd), _p(gx), _p(pg), _p(pb), B, P, C, G, _stream(),
              label="gn_tokens_bwd", nbytes=4 * B * P * C * 4)


def afno_wprep(w, Wc, nb, bs, transpose):
    _lib.call("rpb_afno_wprep", _p(w), _p(Wc), nb, bs, bs, int(transpose), _stream(), label="afno_wprep", nbytes=24 * nb * bs * bs)


def afno_mlp(X, Wa, ba, Wb, bb, aux, mid, out, ntok, nb, bs, mode):
    _lib.call("rpb_afno_mlp", _p(X), _p(Wa), _p(ba), _p(Wb), _p(bb), _p(aux), _p(mid), _p(out), ntok, nb, bs, int(mode), _stream(),
              label=f"afno_mlp[mode={int(mode)}]", nbytes=4 * ntok * 2 * nb * bs * 3, flops=2 * 2 * ntok * nb * (2 * bs) ** 2)


def afno_wgrad_splits(ntok):
    return _lib.query("rpb_afno_wgrad_splits", ntok)


def afno_wgrad(A, G, part, dw, ntok, nb, bs, a_gelu):
    _lib.call("rpb_afno_wgrad", _p(A), _p(G), _p(part), _p(dw), ntok, nb, bs, int(a_gelu), _stream(), label="afno_wgrad",
              nbytes=4 * ntok * 4 * nb * bs, flops=2 * ntok * nb * (2 * bs) ** 2)


def dpot_unpatch(O, pred, B, T, H, W, Cd, Co, ps, ldo):
    _lib.call("rpb_dpot_unpatch", _p(O), _p(pred), B, T, H, W, Cd, Co, ps, ldo, _stream(), label="dpot_unpatch",
              nbytes=8 * pred.numel())


def dpot_unpatch_bwd(gpred, gO, B, T, H, W, Cd, Co, ps, ldo):
    _lib.call("rpb_dpot_unpatch_bwd", _p(gpred), _p(gO), B, T, H, W, Cd, Co, ps, ldo, _stream(), label="dpot_unpatch_bwd",
              nbytes=4 * (gpred.numel() + gO.numel()))
