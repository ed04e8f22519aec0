#!/usr/bin/env python
"""Per-product and per-dot-product error of the operand splits this repo multiplies on the 16-bit matrix pipe (numpy, no GPU):

  bf16x3 x 6   the default: x = hi + mid + lo exactly (three TRUNCATED bf16 planes), six products, the three smallest dropped
  bf16x3r x 6  the same with every plane ROUNDED to nearest even (build switch -DRPB_SPLIT_RNE=1, csrc/rpb_common.h)
  f16x2 x 3    the opt-in eval arithmetic (FNO3d.set_arith("f16x2")): two fp16 planes, both rounded to nearest even, lo*lo dropped
  f16x2 x 4    the same with the fourth product kept
  fp32         one rounded fp32 multiply / a sequential fp32 dot product (what a CPU fp32 reference does)

against exact products in fp64.  Products are summed in fp64 here: the matrix pipe's own fp32 accumulation comes on top for every
variant alike.  `python tools/split_error.py > profiles/r06b_split_error.txt`
"""
import numpy as np

rng = np.random.default_rng(0)
N = 2_000_000


def trunc_bf16(x):
    return (x.astype(np.float32).view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)


def bf3(x):
    h = trunc_bf16(x)
    r = (x - h).astype(np.float32)
    m = trunc_bf16(r)
    l = trunc_bf16((r - m).astype(np.float32))
    return h.astype(np.float64), m.astype(np.float64), l.astype(np.float64)


def rne_bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32).view(np.float32)


def bf3r(x):
    h = rne_bf16(x)
    r = (x - h).astype(np.float32)
    m = rne_bf16(r)
    l = rne_bf16((r - m).astype(np.float32))
    assert np.all(h.astype(np.float64) + m + l == x.astype(np.float64))      # exact
    return h.astype(np.float64), m.astype(np.float64), l.astype(np.float64)


def h2(x):
    h = x.astype(np.float16)
    l = (x - h.astype(np.float32)).astype(np.float32).astype(np.float16)
    return h.astype(np.float64), l.astype(np.float64)


def stats(p, ex):
    e = np.abs(p - ex) / np.abs(ex)
    return f"max 2^{np.log2(e.max()):6.2f}  rms 2^{np.log2(np.sqrt((e ** 2).mean())):6.2f}"


print("# one product a*b, relative error (operands inside fp16's full-precision range: uniform in [1, 2))")
a, b = rng.uniform(1, 2, N).astype(np.float32), rng.uniform(1, 2, N).astype(np.float32)
ex = a.astype(np.float64) * b.astype(np.float64)
ah, am, al = bf3(a)
bh, bm, bl = bf3(b)
a1, a2 = h2(a)
b1, b2 = h2(b)
p3 = a1 * b2 + a2 * b1 + a1 * b1
print("bf16x3 x 6 :", stats(ah * bl + al * bh + am * bm + ah * bm + am * bh + ah * bh, ex))
rh, rm, rl_ = bf3r(a)
sh, sm, sl = bf3r(b)
print("bf16x3r x 6:", stats(rh * sl + rl_ * sh + rm * sm + rh * sm + rm * sh + rh * sh, ex))
print("f16x2  x 3 :", stats(p3, ex))
print("f16x2  x 4 :", stats(p3 + a2 * b2, ex))
print("fp32 mul   :", stats((a * b).astype(np.float64), ex))

print("# dot products, K = 64 (a ~ N(0,1): activations; b ~ N(0,1)/8: weights), Rel-L2 of the results over 200 000 dots")
K, M = 64, 200_000
a = rng.standard_normal((M, K)).astype(np.float32)
b = (rng.standard_normal((M, K)) / 8).astype(np.float32)
ex = (a.astype(np.float64) * b.astype(np.float64)).sum(1)
ah, am, al = bf3(a)
bh, bm, bl = bf3(b)
a1, a2 = h2(a)
b1, b2 = h2(b)
p3 = (a1 * b2 + a2 * b1 + a1 * b1).sum(1)
seq = np.zeros(M, np.float32)
for k in range(K):
    seq = seq + a[:, k] * b[:, k]
rl = lambda p: np.linalg.norm(p - ex) / np.linalg.norm(ex)
print(f"bf16x3 x 6 : {rl((ah * bl + al * bh + am * bm + ah * bm + am * bh + ah * bh).sum(1)):.3e}")
print(f"f16x2  x 3 : {rl(p3):.3e}      (weights unscaled: their lo planes are partly subnormal)")
print(f"f16x2  x 4 : {rl(p3 + (a2 * b2).sum(1)):.3e}")
b1, b2 = h2(b * 16)
print(f"f16x2  x 3 : {rl((a1 * b2 + a2 * b1 + a1 * b1).sum(1) / 16):.3e}      (weights x 2^4 as the kernels do)")
print(f"fp32 sequential dot product: {rl(seq.astype(np.float64)):.3e}")
