// Shared device/host helpers for the RealPDEBench MI355X (gfx950 / CDNA4) kernels.
// Wave = 64 lanes everywhere.  fp32 matrix work uses v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

// ---------------------------------------------------------------------------------- errors
#define RPB_OK 0
#define RPB_ERR_ARG -1
#define RPB_ERR_LAUNCH -2
#define RPB_ERR_UNSUPPORTED -3

extern thread_local char rpb_err_buf[512];

#define RPB_FAIL(code, ...)                                      \
    do {                                                         \
        snprintf(rpb_err_buf, sizeof(rpb_err_buf), __VA_ARGS__); \
        return (code);                                           \
    } while (0)

#define RPB_REQUIRE(cond, ...)                        \
    do {                                              \
        if (!(cond)) RPB_FAIL(RPB_ERR_ARG, __VA_ARGS__); \
    } while (0)

#define RPB_CHECK_LAUNCH(name)                                                                   \
    do {                                                                                         \
        hipError_t e_ = hipGetLastError();                                                       \
        if (e_ != hipSuccess) RPB_FAIL(RPB_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e_));   \
        return RPB_OK;                                                                           \
    } while (0)

int rpb_num_cus();   // cached hipDeviceAttributeMultiprocessorCount of the current device
// How the (b,t,h) lines of the C = 64 cell_mix launches reach their waves (RPB_LINE_CLAIM, default 1):
//   0  dealt round-robin: wave `slot` walks slot, slot + nslots, ...  -- every partial sum has a fixed order (bit-reproducible TRAINING
//      runs; the eval launches have no per-wave sums and are bit-reproducible in every mode)
//   1  the 8 waves of a workgroup CLAIM the workgroup's lines (the same set) from a counter in LDS: under the static deal the waves of
//      one CU -- two per SIMD -- progress up to 15 % apart (tools/wave_times.py) and the launch waits for the slowest
//   2  claimed chip-wide from a counter in HBM (perfectly level, but no faster than 1: see DESIGN.md)
// The one-wave-per-SIMD kernels (bn_bwd_row, head, projection) were measured with mode 1 as well: no change.  The axis GEMMs gain 1-2 % in
// their plain instances and LOSE 12-17 % in the forward W stage with the lazy BatchNorm + GELU (1.40 -> 1.57-1.64 ms, same box: the claimed
// item index changes the compiler's schedule of that instance), so they keep the static deal.
int rpb_line_claim_mode();

// ---------------------------------------------------------------------------------- MFMA
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// D = A(32x2) * B(2x32) + C, one wave.  Lane l supplies A[i = l&31][k = l>>5], B[k = l>>5][j = l&31].
// Lane l, register r of D holds D[row = 8*(r>>2) + 4*(l>>5) + (r&3)][col = l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma_row(int lane, int r) { return 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3); }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// ---------------------------------------------------------------------------------- buffer addressing
// SRSRC ("buffer") accesses: 128-bit descriptor in SGPRs + ONE 32-bit per-lane offset + scalar/immediate offsets,
// with hardware bounds checking (out-of-range lanes: loads return 0, stores are dropped).  Built per tile from
// wave-uniform values, this replaces per-lane 64-bit address arithmetic and every tail predicate.
// IMPORTANT: only voffset + the instruction's immediate are range-checked; soffset is NOT.  Anything that must be
// clipped by the descriptor therefore goes through `voff` (constants added to it are folded into the immediate).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void buf_store_f32(float v, rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}
__device__ __forceinline__ float buf_load_f32(rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
// bytes of `rows_left` rows of `row_bytes` each, clipped to one tile of `tile_rows` rows
__device__ __forceinline__ unsigned tile_bytes(long rows_left, int tile_rows, int row_bytes) {
    return (unsigned)((rows_left < tile_rows ? (rows_left < 0 ? 0 : rows_left) : tile_rows) * row_bytes);
}

// Cache policy of the STREAMING loads / stores of the line-walking kernels (the aux operand of raw_buffer_load / _store: 2 = nt,
// nontemporal; 0 = default).  Round 5: a plain copy gains 5 % from nt on these boxes (tools/ubench/stream_pat: 5.66 -> 5.94-5.98 TB/s for
// two / three reads + one write, 5.88 -> 6.26 for the contiguous copy), and inside the training step nt in the row kernels, cell_mix, the
// DFT stages and Adam is worth 0.44 ms of 37.1 (A/B alternating on one box; per-kernel micro-benchmarks had shown nothing: what nt buys is
// cache left to the operands that ARE reused).  -DRPB_STREAM_AUX=0 restores rounds 1-4.
#ifndef RPB_STREAM_AUX
#define RPB_STREAM_AUX 2
#endif

// the same policy for kernels that stream through plain pointers (16 B per lane): x = sld4(p), sst4(p, v)
#define RPB_SLD4(P) (RPB_STREAM_AUX == 2 ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P)) : *reinterpret_cast<const f32x4*>(P))
#define RPB_SST4(P, V)                                                                  \
    do {                                                                                \
        if (RPB_STREAM_AUX == 2) __builtin_nontemporal_store((V), reinterpret_cast<f32x4*>(P)); \
        else *reinterpret_cast<f32x4*>(P) = (V);                                        \
    } while (0)

// bf16 STORAGE (BASELINE.json configs[4]): an operand that was stored as bf16 carries a rounding of 2^-9 of its value, and it is multiplied
// with the planes of an fp32 constant (conv / fc1 weights, DFT stage matrices).  The constant's third plane contributes 2^-16 of the product:
// 1 / 128 of the error the stored operand already has.  RPB_BF16_CONST_PLANES = 2 (default) drops that product -- two MFMAs per stored plane
// instead of three in every kernel of the bf16-storage forward; 3 keeps it (round 4).  fp32 storage is not touched.
#ifndef RPB_BF16_CONST_PLANES
#define RPB_BF16_CONST_PLANES 2
#endif

// ---------------------------------------------------------------------------------- the three-plane operand split
// x = hi + mid + lo EXACTLY with three bf16 numbers, two ways:
//   RPB_SPLIT_RNE = 0 (default)  truncation at every level (v_perm packs): mid < 2^-7 |x|, lo < 2^-15 |x|, so the three products the
//       six-product scheme drops (mid*lo, lo*mid, lo*lo) are 2^-24 |a b| typically (rms 2^-24.1) and reach 2^-21.3 in the worst case;
//   RPB_SPLIT_RNE = 1  round to nearest even at every level (gfx950: v_cvt_pk_bf16_f32 converts a pair): |mid| <= 2^-8 |x|, |lo| <= 2^-16 |x|,
//       still exact (the last residual has at most 8 significant bits), dropped terms <= 2^-24.2 |a b| worst case, 2^-27.5 rms
//       (tools/split_error.py).  Fewer instructions (4.5 against 5.5 per value) but NOT faster: the conversions and packed subtracts
//       issue at the packed rate -- measured (profiles/r06b_ab4_split_rne.txt, A/B twice on one box): train step 35.63 / 35.67 ms with the
//       truncating split, 36.04 / 36.09 with the rounding one (the wave-pair backward cell_mix 2.95 -> 3.05 ms), eval forward equal
//       (11.35 ms).  Every parity test passes either way; the build switch is for a caller that wants the tighter worst case for 1.1 %.
// (The split3 passes of csrc/rpb_conv3x.hip -- memory-bound -- have rounded since round 2.)
#ifndef RPB_SPLIT_RNE
#define RPB_SPLIT_RNE 0
#endif
typedef __bf16 rpb_bf16x2 __attribute__((ext_vector_type(2)));
typedef float rpb_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void rpb_split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
#if RPB_SPLIT_RNE
    const rpb_f32x2 ab = {a, b};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(ab, rpb_bf16x2));
    const rpb_f32x2 r = ab - rpb_f32x2{__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xffff0000u)};        // exact
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, rpb_bf16x2));
    const rpb_f32x2 s = r - rpb_f32x2{__builtin_bit_cast(float, m << 16), __builtin_bit_cast(float, m & 0xffff0000u)};         // exact
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(s, rpb_bf16x2));                                                  // exact: <= 8 significant bits
#else
    const unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
    h = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
    const float ra = a - __builtin_bit_cast(float, ua & 0xffff0000u), rb = b - __builtin_bit_cast(float, ub & 0xffff0000u);
    const unsigned uc = __builtin_bit_cast(unsigned, ra), ud = __builtin_bit_cast(unsigned, rb);
    m = __builtin_amdgcn_perm(ud, uc, 0x07060302u);
    const float sa = ra - __builtin_bit_cast(float, uc & 0xffff0000u), sb = rb - __builtin_bit_cast(float, ud & 0xffff0000u);
    l = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, sb), __builtin_bit_cast(unsigned, sa), 0x07060302u);
#endif
}

// ---------------------------------------------------------------------------------- math
// Branch-free erf:  erf(x) = sign(x) * (1 - 2^(t*S(t))),  t = min(|x|, 4),  S = degree-8 weighted-minimax fit of
// log2(erfc(t))/t (fitted offline against scipy in fp64; max |error| 9.5e-8 in fp32 arithmetic = the rounding of
// "1 - e" itself, the same cancellation 0.5*x*(1+erf(x/sqrt2)) has in the reference's fp32 F.gelu).  ~15 VALU ops
// and no divergence, versus ~100 divergent ops for ocml erff: the BatchNorm+GELU passes become HBM-bound.
__device__ __forceinline__ float fast_erf(float x) {
    const float t = fminf(fabsf(x), 4.0f);
    float p = 1.160457393e-05f;
    p = fmaf(p, t, -1.529619341e-04f);
    p = fmaf(p, t, 8.482242992e-04f);
    p = fmaf(p, t, -2.274763673e-03f);
    p = fmaf(p, t, 8.477856228e-05f);
    p = fmaf(p, t, 2.772449465e-02f);
    p = fmaf(p, t, -1.483079179e-01f);
    p = fmaf(p, t, -9.184428993e-01f);
    p = fmaf(p, t, -1.627907267e+00f);
    const float e = __builtin_amdgcn_exp2f(p * t);
    return copysignf(1.0f - e, x);
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f)); }
// d/dx gelu(x) = Phi(x) + x*phi(x)
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + fast_erf(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);
    return cdf + x * pdf;
}

// The same functions on channel PAIRS: gfx950 issues v_pk_{fma,mul,add}_f32 (two fp32 lanes per VGPR pair) at the rate of the
// scalar forms, so the polynomial, the affine transforms and the statistics cost half the VALU slots; only |x|, min, exp2 and
// the sign transfer stay per element.  Element-wise the operations (and therefore the results) are those of the scalar forms.
__device__ __forceinline__ f32x2 pk2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 fast_erf2(f32x2 x) {
    const f32x2 t = __builtin_elementwise_min(__builtin_elementwise_abs(x), pk2(4.0f));
    f32x2 p = pk2(1.160457393e-05f);
    p = pk_fma(p, t, pk2(-1.529619341e-04f));
    p = pk_fma(p, t, pk2(8.482242992e-04f));
    p = pk_fma(p, t, pk2(-2.274763673e-03f));
    p = pk_fma(p, t, pk2(8.477856228e-05f));
    p = pk_fma(p, t, pk2(2.772449465e-02f));
    p = pk_fma(p, t, pk2(-1.483079179e-01f));
    p = pk_fma(p, t, pk2(-9.184428993e-01f));
    p = pk_fma(p, t, pk2(-1.627907267e+00f));
    const f32x2 q = p * t;
    const f32x2 e = f32x2{__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
    const f32x2 r = pk2(1.0f) - e;
    return f32x2{copysignf(r[0], x[0]), copysignf(r[1], x[1])};
}
__device__ __forceinline__ f32x2 gelu2(f32x2 x) {
    return (pk2(0.5f) * x) * (pk2(1.0f) + fast_erf2(x * pk2(0.70710678118654752440f)));
}
__device__ __forceinline__ f32x2 gelu_grad2(f32x2 x) {
    const f32x2 cdf = pk2(0.5f) * (pk2(1.0f) + fast_erf2(x * pk2(0.70710678118654752440f)));
    const f32x2 q = (pk2(-0.72134752044448170368f) * x) * x;
    const f32x2 e = f32x2{__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
    return cdf + x * (pk2(0.39894228040143267794f) * e);
}

// gelu(x) and gelu'(x) from ONE erf (the backward cell_mix needs the layer input act(z) for the weight gradient next to act'(z))
__device__ __forceinline__ void gelu_both2(f32x2 x, f32x2& g, f32x2& gp) {
    const f32x2 h = pk2(0.5f) * (pk2(1.0f) + fast_erf2(x * pk2(0.70710678118654752440f)));
    const f32x2 q = (pk2(-0.72134752044448170368f) * x) * x;
    const f32x2 e = f32x2{__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
    g = (pk2(0.5f) * x) * (pk2(1.0f) + fast_erf2(x * pk2(0.70710678118654752440f)));      // same expression as gelu2 (CSE'd erf)
    gp = h + x * (pk2(0.39894228040143267794f) * e);
}

// TWO packed pairs in lock-step.  A dependent v_pk_fma_f32 needs a wait state after its producer (the compiler fills it with s_nop 0)
// and cannot issue back to back; left alone the compiler evaluates one nine-term chain after the other (shortest live ranges), so a wave
// spends the polynomial waiting on itself.  Written as two interleaved chains every instruction has an independent neighbour.
// Element-wise the operations -- and therefore the results -- are those of fast_erf2 / gelu2 / gelu_both2.
#ifndef RPB_ERF_ILP_FENCE
#define RPB_ERF_ILP_FENCE 1     /* keep the interleaving: the scheduler may not move instructions across the step boundaries */
#endif
__device__ __forceinline__ void fast_erf2x2(f32x2 xa, f32x2 xb, f32x2& ra, f32x2& rb) {
    const f32x2 ta = __builtin_elementwise_min(__builtin_elementwise_abs(xa), pk2(4.0f));
    const f32x2 tb = __builtin_elementwise_min(__builtin_elementwise_abs(xb), pk2(4.0f));
    f32x2 pa = pk2(1.160457393e-05f), pb = pk2(1.160457393e-05f);
#if RPB_ERF_ILP_FENCE
#define RPB_ERF_STEP(c) pa = pk_fma(pa, ta, pk2(c)); pb = pk_fma(pb, tb, pk2(c)); __builtin_amdgcn_sched_barrier(0);
#else
#define RPB_ERF_STEP(c) pa = pk_fma(pa, ta, pk2(c)); pb = pk_fma(pb, tb, pk2(c));
#endif
    RPB_ERF_STEP(-1.529619341e-04f)
    RPB_ERF_STEP(8.482242992e-04f)
    RPB_ERF_STEP(-2.274763673e-03f)
    RPB_ERF_STEP(8.477856228e-05f)
    RPB_ERF_STEP(2.772449465e-02f)
    RPB_ERF_STEP(-1.483079179e-01f)
    RPB_ERF_STEP(-9.184428993e-01f)
    RPB_ERF_STEP(-1.627907267e+00f)
#undef RPB_ERF_STEP
    const f32x2 qa = pa * ta, qb = pb * tb;
    const f32x2 ea = f32x2{__builtin_amdgcn_exp2f(qa[0]), __builtin_amdgcn_exp2f(qa[1])};
    const f32x2 eb = f32x2{__builtin_amdgcn_exp2f(qb[0]), __builtin_amdgcn_exp2f(qb[1])};
    const f32x2 sa = pk2(1.0f) - ea, sb = pk2(1.0f) - eb;
    ra = f32x2{copysignf(sa[0], xa[0]), copysignf(sa[1], xa[1])};
    rb = f32x2{copysignf(sb[0], xb[0]), copysignf(sb[1], xb[1])};
}
// h + |h| m as ONE v_fma_f32 with the |.| source modifier.  Left to itself the vectoriser pairs two of them into 2 x v_and + v_pk_fma_f32
// (what the sign transfer it replaces cost); an EMPTY asm on the first result keeps them apart.  (Not the instruction itself as asm: the hazard
// recogniser does not see an asm statement as a vector instruction, so a DPP / permlane read of its result could lose its wait states.)
__device__ __forceinline__ float fma_abs(float h, float m) { return __builtin_fmaf(__builtin_fabsf(h), m, h); }
__device__ __forceinline__ float fma_abs_b(float h, float m) {      // the pair's first element: its RESULT passes the empty asm
    float d = __builtin_fmaf(__builtin_fabsf(h), m, h);
    asm("" : "+v"(d));
    return d;
}
// gelu(x) = x/2 + (x/2) erf(x / sqrt 2) = h + |h| (1 - e), h = x / 2 (h has the sign of the erf argument): the sign transfer of erf is the
// |.| source modifier of the last FMA -- two packed instructions per pair fewer than (0.5 x) (1 + copysign(1 - e, x))
#ifndef RPB_GELU_ABS_FMA
#define RPB_GELU_ABS_FMA 1
#endif
__device__ __forceinline__ void gelu2x2(f32x2& xa, f32x2& xb) {
#if RPB_GELU_ABS_FMA
    const f32x2 sa_ = xa * pk2(0.70710678118654752440f), sb_ = xb * pk2(0.70710678118654752440f);
    const f32x2 ta = __builtin_elementwise_min(__builtin_elementwise_abs(sa_), pk2(4.0f));
    const f32x2 tb = __builtin_elementwise_min(__builtin_elementwise_abs(sb_), pk2(4.0f));
    f32x2 pa = pk2(1.160457393e-05f), pb = pk2(1.160457393e-05f);
#if RPB_ERF_ILP_FENCE
#define RPB_ERF_STEP(c) pa = pk_fma(pa, ta, pk2(c)); pb = pk_fma(pb, tb, pk2(c)); __builtin_amdgcn_sched_barrier(0);
#else
#define RPB_ERF_STEP(c) pa = pk_fma(pa, ta, pk2(c)); pb = pk_fma(pb, tb, pk2(c));
#endif
    RPB_ERF_STEP(-1.529619341e-04f)
    RPB_ERF_STEP(8.482242992e-04f)
    RPB_ERF_STEP(-2.274763673e-03f)
    RPB_ERF_STEP(8.477856228e-05f)
    RPB_ERF_STEP(2.772449465e-02f)
    RPB_ERF_STEP(-1.483079179e-01f)
    RPB_ERF_STEP(-9.184428993e-01f)
    RPB_ERF_STEP(-1.627907267e+00f)
#undef RPB_ERF_STEP
    const f32x2 qa = pa * ta, qb = pb * tb;
    const f32x2 ea = f32x2{__builtin_amdgcn_exp2f(qa[0]), __builtin_amdgcn_exp2f(qa[1])};
    const f32x2 eb = f32x2{__builtin_amdgcn_exp2f(qb[0]), __builtin_amdgcn_exp2f(qb[1])};
    const f32x2 ma = pk2(1.0f) - ea, mb = pk2(1.0f) - eb;
    const f32x2 ha = pk2(0.5f) * xa, hb = pk2(0.5f) * xb;
    xa = f32x2{fma_abs_b(ha[0], ma[0]), fma_abs(ha[1], ma[1])};
    xb = f32x2{fma_abs_b(hb[0], mb[0]), fma_abs(hb[1], mb[1])};
#else
    f32x2 ea, eb;
    fast_erf2x2(xa * pk2(0.70710678118654752440f), xb * pk2(0.70710678118654752440f), ea, eb);
    xa = (pk2(0.5f) * xa) * (pk2(1.0f) + ea);
    xb = (pk2(0.5f) * xb) * (pk2(1.0f) + eb);
#endif
}
__device__ __forceinline__ void gelu_both2x2(f32x2 xa, f32x2 xb, f32x2& ga, f32x2& gb, f32x2& gpa, f32x2& gpb) {
    f32x2 ea, eb;
    fast_erf2x2(xa * pk2(0.70710678118654752440f), xb * pk2(0.70710678118654752440f), ea, eb);
    const f32x2 ha = pk2(0.5f) * (pk2(1.0f) + ea), hb = pk2(0.5f) * (pk2(1.0f) + eb);
    const f32x2 qa = (pk2(-0.72134752044448170368f) * xa) * xa, qb = (pk2(-0.72134752044448170368f) * xb) * xb;
    const f32x2 pa = f32x2{__builtin_amdgcn_exp2f(qa[0]), __builtin_amdgcn_exp2f(qa[1])};
    const f32x2 pb = f32x2{__builtin_amdgcn_exp2f(qb[0]), __builtin_amdgcn_exp2f(qb[1])};
    ga = (pk2(0.5f) * xa) * (pk2(1.0f) + ea);
    gb = (pk2(0.5f) * xb) * (pk2(1.0f) + eb);
    gpa = ha + xa * (pk2(0.39894228040143267794f) * pa);
    gpb = hb + xb * (pk2(0.39894228040143267794f) * pb);
}

// The same pair of results from ONE exponential and one reciprocal (Abramowitz-Stegun 26.2.17: Phi(-|u|) = phi(u) t P4(t), t = 1 / (1 + p |u|),
// |error| < 7.5e-8 -- the grade of fast_erf; the density phi(u) is what gelu' needs anyway): 9.5 packed / scalar instructions + 2
// transcendentals per value against 12 + 2 above.  Two pairs in lock-step, as gelu2x2.  (First used by the head, csrc/rpb_pjg.hip.)
__device__ __forceinline__ void gelu_both_as2x2(f32x2 ua, f32x2 ub, f32x2& va, f32x2& vb, f32x2& da, f32x2& db) {
    const f32x2 aa = __builtin_elementwise_abs(ua), ab = __builtin_elementwise_abs(ub);
    const f32x2 dna = pk_fma(aa, pk2(0.23164189f), pk2(1.0f)), dnb = pk_fma(ab, pk2(0.23164189f), pk2(1.0f));
    const f32x2 qa = (ua * pk2(-0.72134752044448170368f)) * ua, qb = (ub * pk2(-0.72134752044448170368f)) * ub;
    const f32x2 ta = {__builtin_amdgcn_rcpf(dna[0]), __builtin_amdgcn_rcpf(dna[1])}, tb = {__builtin_amdgcn_rcpf(dnb[0]), __builtin_amdgcn_rcpf(dnb[1])};
    const f32x2 ea = {__builtin_amdgcn_exp2f(qa[0]), __builtin_amdgcn_exp2f(qa[1])}, eb = {__builtin_amdgcn_exp2f(qb[0]), __builtin_amdgcn_exp2f(qb[1])};
#if RPB_ERF_ILP_FENCE
#define RPB_AS_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define RPB_AS_FENCE()
#endif
    RPB_AS_FENCE();
    f32x2 pa = pk_fma(pk2(0.5307027145f), ta, pk2(-0.7265760135f)), pb = pk_fma(pk2(0.5307027145f), tb, pk2(-0.7265760135f));
    const f32x2 hua = pk2(0.5f) * ua, hub = pk2(0.5f) * ub;
    RPB_AS_FENCE();
    pa = pk_fma(pa, ta, pk2(0.7107068705f)), pb = pk_fma(pb, tb, pk2(0.7107068705f));
    const f32x2 xa_ = aa * ea, xb_ = ab * eb;
    RPB_AS_FENCE();
    pa = pk_fma(pa, ta, pk2(-0.142248368f)), pb = pk_fma(pb, tb, pk2(-0.142248368f));
    const f32x2 tea = ta * ea, teb = tb * eb;
    RPB_AS_FENCE();
    pa = pk_fma(pa, ta, pk2(0.127414796f)), pb = pk_fma(pb, tb, pk2(0.127414796f));
    RPB_AS_FENCE();
    const f32x2 ha = pk_fma(-pa, tea, pk2(0.5f)), hb = pk_fma(-pb, teb, pk2(0.5f));           // 1/2 - Phi(-|u|)
    RPB_AS_FENCE();
    va = pk_fma(aa, ha, hua), vb = pk_fma(ab, hb, hub);
    const f32x2 wa = pk_fma(xa_, pk2(0.39894228040143267794f), ha), wb = pk_fma(xb_, pk2(0.39894228040143267794f), hb);
    RPB_AS_FENCE();
    da = pk2(0.5f) + f32x2{__builtin_copysignf(wa[0], ua[0]), __builtin_copysignf(wa[1], ua[1])};
    db = pk2(0.5f) + f32x2{__builtin_copysignf(wb[0], ub[0]), __builtin_copysignf(wb[1], ub[1])};
#undef RPB_AS_FENCE
}

// four channels at once (two packed pairs)
__device__ __forceinline__ f32x4 join4(f32x2 a, f32x2 b) { return f32x4{a[0], a[1], b[0], b[1]}; }
__device__ __forceinline__ f32x4 gelu4(f32x4 x) {
    f32x2 a = x.lo, b = x.hi;
    gelu2x2(a, b);
    return join4(a, b);
}
__device__ __forceinline__ f32x4 gelu_grad4(f32x4 x) {
    f32x2 ea, eb;
    fast_erf2x2(x.lo * pk2(0.70710678118654752440f), x.hi * pk2(0.70710678118654752440f), ea, eb);
    const f32x2 ca = pk2(0.5f) * (pk2(1.0f) + ea), cb = pk2(0.5f) * (pk2(1.0f) + eb);
    const f32x2 qa = (pk2(-0.72134752044448170368f) * x.lo) * x.lo, qb = (pk2(-0.72134752044448170368f) * x.hi) * x.hi;
    const f32x2 pa = f32x2{__builtin_amdgcn_exp2f(qa[0]), __builtin_amdgcn_exp2f(qa[1])};
    const f32x2 pb = f32x2{__builtin_amdgcn_exp2f(qb[0]), __builtin_amdgcn_exp2f(qb[1])};
    return join4(ca + x.lo * (pk2(0.39894228040143267794f) * pa), cb + x.hi * (pk2(0.39894228040143267794f) * pb));
}
// shat = (x - mu) * is;  z = shat * ga + be
__device__ __forceinline__ f32x4 bn4(f32x4 x, f32x4 mu, f32x4 is, f32x4 ga, f32x4 be, f32x4* shat = nullptr) {
    const f32x2 s0 = (x.lo - mu.lo) * is.lo, s1 = (x.hi - mu.hi) * is.hi;
    if (shat) *shat = join4(s0, s1);
    return join4(pk_fma(s0, ga.lo, be.lo), pk_fma(s1, ga.hi, be.hi));
}

// Lazy activation: a layer output a = act(gamma * (s - mean) * invstd + beta) is never written to HBM; its consumers
// read the pre-BatchNorm tensor s and apply this per-channel transform on load (fno.py:117-119 fused into :48,:115,:123).
struct XForm {
    const float* mean;      // nullptr => identity (plain tensor)
    const float* invstd;
    const float* gamma;
    const float* beta;
    int gelu;
};
struct XParam {
    float mu, is, ga, be;
};
__device__ __forceinline__ XParam xf_load(const XForm& x, int c) { return XParam{x.mean[c], x.invstd[c], x.gamma[c], x.beta[c]}; }
__device__ __forceinline__ float xf_apply(float v, const XParam& p, bool gelu) {
    const float z = (v - p.mu) * p.is * p.ga + p.be;
    return gelu ? gelu_f(z) : z;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// padded-cell index -> cropped-cell index, or -1 when the cell lies in the zero-pad margin
struct CropMap {
    int T, H, W, Tp, Hp, Wp;
};
__device__ __forceinline__ long pad_to_crop(const CropMap& m, long p) {
    int w = (int)(p % m.Wp);
    long r = p / m.Wp;
    int h = (int)(r % m.Hp);
    r /= m.Hp;
    int t = (int)(r % m.Tp);
    long b = r / m.Tp;
    if (w >= m.W || h >= m.H || t >= m.T) return -1;
    return ((b * m.T + t) * m.H + h) * (long)m.W + w;
}
__device__ __forceinline__ long crop_to_pad(const CropMap& m, long q) {
    int w = (int)(q % m.W);
    long r = q / m.W;
    int h = (int)(r % m.H);
    r /= m.H;
    int t = (int)(r % m.T);
    long b = r / m.T;
    return ((b * m.Tp + t) * m.Hp + h) * (long)m.Wp + w;
}

// ---------------------------------------------------------------------------------- in-kernel dropout
// Philox4x32-10 (Salmon et al., SC'11): counter-based, so the inverted-dropout multiplier of an element is a pure function
// of (seed, element index) -- the forward epilogue and the backward pass regenerate the same mask instead of writing and
// re-reading a mask tensor (nn.Dropout of the Galerkin / Transolver layers).  One call yields the 4 multipliers of the
// float4 whose first element has linear index 4*q in the dense [rows][C] tensor the dropout applies to.
struct DropSpec {
    unsigned long long seed;
    unsigned thr;        // keep iff random u32 < thr (= keep probability * 2^32); 0 = dropout disabled
    float inv_keep;
};
__device__ __forceinline__ f32x4 dropout4(const DropSpec& d, unsigned long long q) {
    unsigned c0 = (unsigned)q, c1 = (unsigned)(q >> 32), c2 = 0x243F6A88u, c3 = 0x85A308D3u;
    unsigned k0 = (unsigned)d.seed, k1 = (unsigned)(d.seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    f32x4 m;
    m[0] = c0 < d.thr ? d.inv_keep : 0.f;
    m[1] = c1 < d.thr ? d.inv_keep : 0.f;
    m[2] = c2 < d.thr ? d.inv_keep : 0.f;
    m[3] = c3 < d.thr ? d.inv_keep : 0.f;
    return m;
}
static inline DropSpec make_drop(long seed, float keep) {
    DropSpec d;
    d.seed = (unsigned long long)seed;
    d.thr = (keep > 0.f && keep < 1.f) ? (unsigned)((double)keep * 4294967296.0) : 0u;
    d.inv_keep = d.thr ? 1.0f / keep : 1.0f;
    return d;
}
