"""Truncated-DFT stage matrices for the FNO3d spectral layer (host side, built once per shape).

``SpectralConv3d.forward`` (reference realpdebench/model/fno.py:45-64) is
``irfftn(corner_blocks(rfftn(x) * W))``.  Only ``2*m1 x 2*m2 x m3`` bins are ever used, so the MI355X
path applies three small dense *real* matrices (one per axis) instead of full FFTs of the awkward sizes
26 x 134 x 134.  Complex data is planar; the (re, im) index sits next to the transformed axis so each
stage is ``out[o] = sum_k M[o,k] in[k]`` over one strided axis (kernel ``rpb_axis_gemm``):

    forward stage   rows o = (ri_out, mode)  ri outer      cols k = (position, ri_in)  ri inner
    inverse stage   rows o = (position, ri_out) ri inner   cols k = (ri_in, mode)      ri outer

The two conventions are transposes of one another, so every adjoint needed by the backward pass is a plain
``.t()`` of a matrix built here -- including the c2r rule of ``irfftn`` (imaginary part of the k_w = 0 bin
ignored, bins 1..m3-1 doubled) which is baked into ``GW``.
"""
import math

import numpy as np
import torch


def kept_modes(n, m, two_sided):
    """Frequency indices retained along an axis of length ``n``: ``[:m]`` (+ ``[-m:]``), fno.py:53-60."""
    ks = list(range(m))
    if two_sided:
        if 2 * m > n:
            raise ValueError(f"modes {m} overlap on an axis of length {n} (2*modes must be <= padded size)")
        ks += list(range(n - m, n))
    elif m > n // 2 + 1:
        raise ValueError(f"modes {m} exceed the rfft length {n // 2 + 1}")
    return ks


def _fwd_complex(n, ks):
    """[2*len(ks), 2*n]: e^{-i theta}; rows (ri, mode), cols (pos, ri)."""
    k = np.asarray(ks, dtype=np.float64)[:, None]
    p = np.arange(n, dtype=np.float64)[None, :]
    th = 2.0 * math.pi * ((k * p) % n) / n
    c, s = np.cos(th), np.sin(th)
    M = np.zeros((2, len(ks), n, 2))
    M[0, :, :, 0] = c
    M[0, :, :, 1] = s
    M[1, :, :, 0] = -s
    M[1, :, :, 1] = c
    return M.reshape(2 * len(ks), 2 * n)


def _inv_complex(n, ks):
    """[2*n, 2*len(ks)]: e^{+i theta}; rows (pos, ri), cols (ri, mode)."""
    k = np.asarray(ks, dtype=np.float64)[None, :]
    p = np.arange(n, dtype=np.float64)[:, None]
    th = 2.0 * math.pi * ((k * p) % n) / n
    c, s = np.cos(th), np.sin(th)
    M = np.zeros((n, 2, 2, len(ks)))
    M[:, 0, 0, :] = c
    M[:, 0, 1, :] = -s
    M[:, 1, 0, :] = s
    M[:, 1, 1, :] = c
    return M.reshape(2 * n, 2 * len(ks))


def _fwd_real(n, ks):
    """[2*len(ks), n]: real input -> (ri, mode)."""
    k = np.asarray(ks, dtype=np.float64)[:, None]
    p = np.arange(n, dtype=np.float64)[None, :]
    th = 2.0 * math.pi * ((k * p) % n) / n
    return np.concatenate([np.cos(th), -np.sin(th)], axis=0)


def _inv_real(n, ks, norm):
    """[n, 2*len(ks)]: c2r of irfftn along the last axis incl. the 1/(Tp*Hp*Wp) factor."""
    k = np.asarray(ks, dtype=np.float64)[None, :]
    p = np.arange(n, dtype=np.float64)[:, None]
    th = 2.0 * math.pi * ((k * p) % n) / n
    w = np.where((k == 0) | ((n % 2 == 0) & (k == n // 2)), 1.0, 2.0)
    return np.concatenate([w * np.cos(th), -w * np.sin(th)], axis=1) / norm


class SpectralPlan:
    """All stage matrices (fp32, on ``device``) for padded sizes ``(Tp, Hp, Wp)`` and ``modes``."""

    def __init__(self, Tp, Hp, Wp, modes, device="cpu"):
        m1, m2, m3 = modes
        self.Tp, self.Hp, self.Wp = Tp, Hp, Wp
        self.modes = tuple(modes)
        self.kt = kept_modes(Tp, m1, True)
        self.kh = kept_modes(Hp, m2, True)
        self.kw = kept_modes(Wp, m3, False)
        self.KT, self.KH, self.KW = len(self.kt), len(self.kh), len(self.kw)
        self.M = self.KT * self.KH * self.KW
        mats = {
            "FW": _fwd_real(Wp, self.kw),          # [2*m3, Wp]
            "FH": _fwd_complex(Hp, self.kh),       # [2*KH, 2*Hp]
            "FT": _fwd_complex(Tp, self.kt),       # [2*KT, 2*Tp]
            "GT": _inv_complex(Tp, self.kt),       # [2*Tp, 2*KT]
            "GH": _inv_complex(Hp, self.kh),       # [2*Hp, 2*KH]
            "GW": _inv_real(Wp, self.kw, float(Tp) * Hp * Wp),   # [Wp, 2*m3]
        }
        for name, m in mats.items():
            t = torch.from_numpy(np.ascontiguousarray(m)).to(torch.float32)
            setattr(self, name, t.to(device).contiguous())
            setattr(self, name + "t", t.t().contiguous().to(device))   # adjoint used by the backward pass


# ----------------------------------------------------------------------------- weight layout conversion
def ref_weights_to_mode_major(w1, w2, w3, w4):
    """4 x complex ``[Ci,Co,m1,m2,m3]`` (fno.py:31-38) -> real ``[M, Ci, Co, 2]`` mode-major, M=(kt,kh,kw).

    Corner placement follows fno.py:53-60: weights1 (t<m1,h<m2), weights2 (t>=Tp-m1,h<m2),
    weights3 (t<m1,h>=Hp-m2), weights4 (t>=Tp-m1,h>=Hp-m2).
    """
    top = torch.cat([w1, w3], dim=3)          # kt low : kh = [low | high]
    bot = torch.cat([w2, w4], dim=3)          # kt high
    full = torch.cat([top, bot], dim=2)       # [Ci,Co,2m1,2m2,m3]
    full = full.permute(2, 3, 4, 0, 1).contiguous()          # [KT,KH,KW,Ci,Co]
    return torch.view_as_real(full).reshape(-1, full.shape[3], full.shape[4], 2).contiguous()


def mode_major_to_ref_weights(wm, modes):
    """Inverse of :func:`ref_weights_to_mode_major` -> (w1, w2, w3, w4) complex64."""
    m1, m2, m3 = modes
    Ci, Co = wm.shape[1], wm.shape[2]
    full = torch.view_as_complex(wm.reshape(2 * m1, 2 * m2, m3, Ci, Co, 2).contiguous())
    full = full.permute(3, 4, 0, 1, 2)        # [Ci,Co,KT,KH,KW]
    w1 = full[:, :, :m1, :m2].contiguous()
    w2 = full[:, :, m1:, :m2].contiguous()
    w3 = full[:, :, :m1, m2:].contiguous()
    w4 = full[:, :, m1:, m2:].contiguous()
    return w1, w2, w3, w4
