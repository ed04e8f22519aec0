// The projection head's backward (and, fused, its forward + loss) in one pass -- second organisation ("pjg"), the default for fc2 width 2.
// Same mathematics, arguments and partial-row format as rpb_pjf.hip (fno.py:121-125 + autograd, utils/metrics.py:11-13); what changed
// is how the work meets the SIMD, following the measurements of tools/ubench/issue_probe.hip:
//
//   * one wave per SIMD issues  ~4.4 cycles per vector instruction + ~11.5 per MFMA  whatever the MFMA's shape, so the three contractions run
//     on v_mfma_f32_32x32x16_bf16 (288 per 32-cell tile instead of 576 of the 16x16x32 shape: same matrix-pipe time, half the issue cost);
//   * v_pk_*_f32 next to MFMAs costs ~20 cycles per transition (the packed fp32 path waits for the matrix pipe): every vector
//     instruction here is the scalar form (the file is compiled with -fno-slp-vectorize), GELU included;
//   * gh is split into its three bf16 planes ONCE: the planes are the A operand of the weight gradient in registers, and reach the data
//     gradient's operand layout (lane = cell) through LDS and ds_read_b64_tr_b16 -- no second split, no fp32 transposition tile;
//   * BatchNorm costs one subtraction per element: u = W1 (gamma shat + beta) + b1 = W1' (s - mean) + b1' with W1' = W1 diag(gamma invstd),
//     b1' = b1 + W1 beta prepared per workgroup; the weight gradient accumulates gh^T (s - mean) and is scaled by invstd once, at the end;
//   * GELU and GELU' from ONE exponential: Phi(-|u|) = t P4(t) exp(-u^2/2), t = 1/(1 + p|u|) (Abramowitz-Stegun 7.1.26: |error| < 7.5e-8,
//     the same grade as fast_erf), and exp(-u^2/2) is the density GELU' needs anyway: 19 instructions for both against 27.
//
// Layouts (32x32x16: A lane (m = l & 31, kg = l >> 5) holds k = 8 kg + e; B lane (n, kg) likewise; D register r of lane (n, hg) is row
// 8 (r >> 2) + 4 hg + (r & 3), column n):
//   u  [cell][hidden]   = (s - mean) W1'^T     A = planes of the tile (cell m; channels 16 ks + 8 kg + e), B = W1B planes from LDS
//   M  [hidden][chan]  += gh^T (s - mean)      A = planes of gh straight from u's accumulators (lane = hidden, 8 cells per K-step),
//                                              B = planes of a second, B-layout view of the tile (lane = channel pair 2 n, 2 n + 1)
//   g^T[chan][cell]     = W1^T gh^T            A = W1D planes from LDS, B = gh planes read back transposed (lane = cell, 8 hidden units)
#include "rpb_pjf.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x16v __attribute__((ext_vector_type(16)));

#define PG_HID 128
#define PG_WAVES 4
#define PG_GH_PLANE 2048                   // bytes of one gh plane of 32 hidden units x 32 cells
#define PG_GH_BUF (3 * PG_GH_PLANE)
#define PG_GH_WAVE (2 * PG_GH_BUF)
// PG_PIPE = n > 0: ask the scheduler for n vector instructions behind every MFMA of the hidden-tile loop (0: its own order)
#ifndef PG_PIPE
#define PG_PIPE 0
#endif
// PG_PIPE1 = n > 0: the same for contraction 1 (n vector instructions + one LDS read behind every MFMA of a K-step)
#ifndef PG_PIPE1
#define PG_PIPE1 2     /* measured: 3.79-3.85 -> 3.64 ms (the compiler lumps 7-9 instructions into some MFMA gaps and none into others) */
#endif

// -DPG_TIMING: s_memtime stamps at the phase boundaries of the tile body; the wave writes its per-phase cycle sums over the first floats of
// its partial row (the results of such a build are garbage: tools/dbg/head_phases.py reads the stamps only)
#ifdef PG_TIMING
#define PG_T(i)                                    \
    {                                              \
        __builtin_amdgcn_sched_barrier(0);         \
        const long long now_ = clock64();          \
        __builtin_amdgcn_sched_barrier(0);         \
        tacc[i] += (float)(now_ - tlast);          \
        tlast = now_;                              \
    }
#else
#define PG_T(i)
#endif

#ifndef RPB_HEAD_AUX
#define RPB_HEAD_AUX 0   /* cache policy of the tile loads / stores (2 = nt): experiment switch */
#endif
namespace {
__device__ __forceinline__ u32x4 ld16(rsrc_t r, int voff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, RPB_HEAD_AUX));
}
__device__ __forceinline__ f32x2 ld8(rsrc_t r, int voff) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0));
}
// by VALUE: __builtin_bit_cast applied to a vector ELEMENT (v[1]) reads element 0 with hipcc 7.2 (it cost the b64 loads their second
// dword and the row swap its second result); a scalar copy first is safe
__device__ __forceinline__ float asf(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ void st16(f32x4v v, rsrc_t r, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, RPB_HEAD_AUX);
}
__device__ __forceinline__ float trunc_bf16(float v) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_hi(float a, float b) {
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
// three-plane truncation split of 8 values (rpb_cmx.hip): v = h + m + l to 2^-24, each plane exact in bf16
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
    u32x4 uh, um, ul;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = v[2 * q], b = v[2 * q + 1];
        {
            unsigned ph_, pm_, pl_;
            rpb_split_pair(a, b, ph_, pm_, pl_);
            uh[q] = ph_;
            um[q] = pm_;
            ul[q] = pl_;
        }
    }
    h = __builtin_bit_cast(bf16x8, uh);
    m = __builtin_bit_cast(bf16x8, um);
    l = __builtin_bit_cast(bf16x8, ul);
}
__device__ __forceinline__ f32x16v mfma32(bf16x8 a, bf16x8 b, f32x16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// v = u Phi(u), d = Phi(u) + u phi(u) from one reciprocal and one exponential (see the header); scalar instructions only
__device__ __forceinline__ void gelu_both_s(float u, float& v, float& d) {
    const float au = __builtin_fabsf(u);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(au, 0.23164189f, 1.0f));
    float pl = 0.5307027145f;
    pl = __builtin_fmaf(pl, t, -0.7265760135f);
    pl = __builtin_fmaf(pl, t, 0.7107068705f);
    pl = __builtin_fmaf(pl, t, -0.142248368f);
    pl = __builtin_fmaf(pl, t, 0.127414796f);
    const float e = __builtin_amdgcn_exp2f((u * -0.72134752044448170368f) * u);       // exp(-u^2 / 2)
    const float Q = (pl * t) * e;                                                     // Phi(-|u|)
    const float h = 0.5f - Q;                                                         // Phi(|u|) - 1/2
    v = __builtin_fmaf(au, h, 0.5f * u);
    const float w = __builtin_fmaf(au * e, 0.39894228040143267794f, h);               // Phi(|u|) - 1/2 + |u| phi(u): odd part of gelu'
    d = 0.5f + __builtin_copysignf(w, u);
}
// the same on a PAIR of values with packed fp32 instructions: the activation phase of a tile has no MFMA in flight, and one wave
// per SIMD issues a v_pk_* at the price of a scalar instruction (issue_probe: 5.3 cycles either way) -- half the issue slots
__device__ __forceinline__ void gelu_both_p(f32x2 u, f32x2& v, f32x2& d) {
    const f32x2 au = __builtin_elementwise_abs(u);
    const f32x2 den = pk_fma(au, pk2(0.23164189f), pk2(1.0f));
    const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    f32x2 pl = pk_fma(pk2(0.5307027145f), t, pk2(-0.7265760135f));
    pl = pk_fma(pl, t, pk2(0.7107068705f));
    pl = pk_fma(pl, t, pk2(-0.142248368f));
    pl = pk_fma(pl, t, pk2(0.127414796f));
    const f32x2 q = (u * pk2(-0.72134752044448170368f)) * u;
    const f32x2 e = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
    const f32x2 Q = (pl * t) * e;
    const f32x2 h = pk2(0.5f) - Q;
    v = pk_fma(au, h, pk2(0.5f) * u);
    const f32x2 w = pk_fma(au * e, pk2(0.39894228040143267794f), h);
    d = pk2(0.5f) + f32x2{__builtin_copysignf(w[0], u[0]), __builtin_copysignf(w[1], u[1])};
}
// TWO pairs in lock-step (scheduling fences between the steps): a dependent v_pk_fma_f32 needs a wait state after its producer, and left
// alone the compiler evaluates one chain after the other (shortest live ranges) with s_nop between the links
#define PG_FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void gelu_both_p2(f32x2 ua, f32x2 ub, f32x2& va, f32x2& da, f32x2& vb, f32x2& db) {
    const f32x2 aa = __builtin_elementwise_abs(ua), ab = __builtin_elementwise_abs(ub);
    const f32x2 dna = pk_fma(aa, pk2(0.23164189f), pk2(1.0f)), dnb = pk_fma(ab, pk2(0.23164189f), pk2(1.0f));
    const f32x2 qa = (ua * pk2(-0.72134752044448170368f)) * ua, qb = (ub * pk2(-0.72134752044448170368f)) * ub;
    const f32x2 ta = {__builtin_amdgcn_rcpf(dna[0]), __builtin_amdgcn_rcpf(dna[1])}, tb = {__builtin_amdgcn_rcpf(dnb[0]), __builtin_amdgcn_rcpf(dnb[1])};
    const f32x2 ea = {__builtin_amdgcn_exp2f(qa[0]), __builtin_amdgcn_exp2f(qa[1])}, eb = {__builtin_amdgcn_exp2f(qb[0]), __builtin_amdgcn_exp2f(qb[1])};
    PG_FENCE();
    f32x2 pa = pk_fma(pk2(0.5307027145f), ta, pk2(-0.7265760135f)), pb = pk_fma(pk2(0.5307027145f), tb, pk2(-0.7265760135f));
    const f32x2 hua = pk2(0.5f) * ua, hub = pk2(0.5f) * ub;
    PG_FENCE();
    pa = pk_fma(pa, ta, pk2(0.7107068705f)), pb = pk_fma(pb, tb, pk2(0.7107068705f));
    const f32x2 xa_ = aa * ea, xb_ = ab * eb;
    PG_FENCE();
    pa = pk_fma(pa, ta, pk2(-0.142248368f)), pb = pk_fma(pb, tb, pk2(-0.142248368f));
    const f32x2 tea = ta * ea, teb = tb * eb;
    PG_FENCE();
    pa = pk_fma(pa, ta, pk2(0.127414796f)), pb = pk_fma(pb, tb, pk2(0.127414796f));
    PG_FENCE();
    const f32x2 ha = pk_fma(-pa, tea, pk2(0.5f)), hb = pk_fma(-pb, teb, pk2(0.5f));           // 1/2 - Phi(-|u|)
    PG_FENCE();
    va = pk_fma(aa, ha, hua), vb = pk_fma(ab, hb, hub);
    const f32x2 wa = pk_fma(xa_, pk2(0.39894228040143267794f), ha), wb = pk_fma(xb_, pk2(0.39894228040143267794f), hb);
    PG_FENCE();
    da = pk2(0.5f) + f32x2{__builtin_copysignf(wa[0], ua[0]), __builtin_copysignf(wa[1], ua[1])};
    db = pk2(0.5f) + f32x2{__builtin_copysignf(wb[0], ub[0]), __builtin_copysignf(wb[1], ub[1])};
}
__device__ __forceinline__ float dpp_add(float x, float y, const int ctrl) {           // x + y from the lane the DPP control names
    switch (ctrl) {
    case 0: return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), 0x140, 0xF, 0xF, true));   // row_mirror
    case 1: return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), 0x141, 0xF, 0xF, true));   // row_half_mirror
    case 2: return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    default: return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    }
}
}  // namespace

// six products of the three-plane split, small terms first, NC independent accumulation chains advancing together
// (FIRST(c): the chain's first product of the tile takes a literal zero as its C operand -- an inline constant of the instruction -- instead
// of a zeroed accumulator: the compiler otherwise builds a 16-register zero block and copies it into every accumulator, 128 moves per tile)
#define PG_MAC6(NC, ACC, AH, AM, AL, BH, BM, BL) PG_MAC6F(NC, ACC, AH, AM, AL, BH, BM, BL, PG_NOTFIRST)
#define PG_NOTFIRST(c) false
#define PG_MAC6F(NC, ACC, AH, AM, AL, BH, BM, BL, FIRST)                               \
    _Pragma("unroll") for (int c_ = 0; c_ < NC; ++c_) ACC(c_) = mfma32(AH(c_), BL(c_), FIRST(c_) ? f32x16v{} : ACC(c_)); \
    _Pragma("unroll") for (int c_ = 0; c_ < NC; ++c_) ACC(c_) = mfma32(AL(c_), BH(c_), ACC(c_)); \
    _Pragma("unroll") for (int c_ = 0; c_ < NC; ++c_) ACC(c_) = mfma32(AM(c_), BM(c_), ACC(c_)); \
    _Pragma("unroll") for (int c_ = 0; c_ < NC; ++c_) ACC(c_) = mfma32(AH(c_), BM(c_), ACC(c_)); \
    _Pragma("unroll") for (int c_ = 0; c_ < NC; ++c_) ACC(c_) = mfma32(AM(c_), BH(c_), ACC(c_)); \
    _Pragma("unroll") for (int c_ = 0; c_ < NC; ++c_) ACC(c_) = mfma32(AH(c_), BH(c_), ACC(c_));

// LOSS: the head's FORWARD rides along (fused trainer): p.gout is the TARGET, the kernel forms out = fc2 gelu(u) + b2,
// gout = gscale (out - y) and the squared-error partial sums itself
template <bool LOSS>
__global__ __launch_bounds__(PG_WAVES * 64, 1) void pjg_kernel(PjfArgs p) {
    extern __shared__ u32x4 lds4[];
    u32x4* W1B = lds4;                                   // [ks 4][nt 4][plane 3][lane]   B of u:   W1'[32 nt + n][16 ks + 8 kg + e]
    u32x4* W1D = W1B + 4 * 4 * 3 * 64;                   // [ks3 8][mt 2][plane 3][lane]  A of g^T: W1[16 ks3 + 8 kg + e][32 mt + m]
    float* b1l = reinterpret_cast<float*>(W1D + 8 * 2 * 3 * 64);          // [128]  b1' = b1 + W1 beta
    float* meanl = b1l + PG_HID;                         // [64]   BatchNorm mean
    float* gball = meanl + 64;                           // [waves][64]  gout of the tile, gathered for every lane
    char* GHall = reinterpret_cast<char*>(gball + PG_WAVES * 64);         // [waves][buf 2][plane 3][hidden row 32][64 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hg = lane >> 5;
    for (int idx = tid; idx < 4 * 4 * 64; idx += blockDim.x) {
        const int l = idx & 63, nt = (idx >> 6) & 3, ks = idx >> 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 16 * ks + 8 * (l >> 5) + e;
            v[e] = p.w1[(32 * nt + (l & 31)) * 64 + c] * (p.xf.gamma[c] * p.xf.invstd[c]);
        }
        bf16x8 h, m, lo;
        split8(v, h, m, lo);
        W1B[((ks * 4 + nt) * 3 + 0) * 64 + l] = __builtin_bit_cast(u32x4, h);
        W1B[((ks * 4 + nt) * 3 + 1) * 64 + l] = __builtin_bit_cast(u32x4, m);
        W1B[((ks * 4 + nt) * 3 + 2) * 64 + l] = __builtin_bit_cast(u32x4, lo);
    }
    for (int idx = tid; idx < 8 * 2 * 64; idx += blockDim.x) {
        const int l = idx & 63, mt = (idx >> 6) & 1, ks3 = idx >> 7;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p.w1[(16 * ks3 + 8 * (l >> 5) + e) * 64 + 32 * mt + (l & 31)];
        bf16x8 h, m, lo;
        split8(v, h, m, lo);
        W1D[((ks3 * 2 + mt) * 3 + 0) * 64 + l] = __builtin_bit_cast(u32x4, h);
        W1D[((ks3 * 2 + mt) * 3 + 1) * 64 + l] = __builtin_bit_cast(u32x4, m);
        W1D[((ks3 * 2 + mt) * 3 + 2) * 64 + l] = __builtin_bit_cast(u32x4, lo);
    }
    for (int h = tid; h < PG_HID; h += blockDim.x) {
        float a = p.b1[h];
        for (int c = 0; c < 64; ++c) a = __builtin_fmaf(p.w1[h * 64 + c], p.xf.beta[c], a);
        b1l[h] = a;
    }
    if (tid < 64) meanl[tid] = p.xf.mean[tid];
    __syncthreads();

    const CropMap cm = p.cm;
    const long nslots = (long)gridDim.x * PG_WAVES;
    const long slot = (long)blockIdx.x * PG_WAVES + wave;
    const unsigned line_bytes = (unsigned)cm.Wp * 256u;
    const f32x4v z4 = {0.f, 0.f, 0.f, 0.f};

    // ---- pass 1: lines in the zero-pad margin (t >= T or h >= H) of the padded gradient tensor
    {
        const long G = (long)p.B * cm.Tp * cm.Hp;
        for (long g = slot; g < G; g += nslots) {
            const int h = (int)(g % cm.Hp);
            const int t = (int)((g / cm.Hp) % cm.Tp);
            if (h < cm.H && t < cm.T) continue;                      // uniform
            const rsrc_t ro = make_rsrc(p.g + g * cm.Wp * 64, line_bytes);
            for (int off = lane * 16; off < (int)line_bytes; off += 1024) st16(z4, ro, off);
        }
    }

    // ---- pass 2: cropped lines, 32-cell tiles
    const long GL = (long)p.B * cm.T * cm.H;
    const int TQ = (cm.W + 31) >> 5;
    char* GHw = GHall + wave * PG_GH_WAVE;
    float* gb = gball + wave * 64;
    // per-lane constants
    const float meanB0 = p.xf.mean[2 * n], meanB1 = p.xf.mean[2 * n + 1];          // B layout: the lane's channel pair
    float w2r[2][4], b1r[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        b1r[nt] = b1l[32 * nt + n];
        w2r[0][nt] = p.w2[32 * nt + n];
        w2r[1][nt] = p.w2[PG_HID + 32 * nt + n];
    }
    const float b2y = LOSS ? p.b2[n & 1] : 0.f;
    // the lane's own output element after the cross-lane reduction: (register row ry, feature n & 1) -> cell 8 (ry >> 2) + 4 hg + (ry & 3)
    const int celly = 8 * (n >> 3) + 4 * hg + ((n >> 1) & 3);
    // cell of column slot n in the data-gradient tile (order in which the transposing reads deliver the cells)
    const int cells = 8 * ((n >> 2) & 3) + 4 * (n >> 4) + (n & 3);
    // LDS addresses: write of the lane's 16 B (hidden row n, chunk 2 hg + kstep, swizzled) and the two transposing reads (rows 8 hg + 4 half + a)
    const int ghw0 = n * 64 + (((2 * hg + 0) ^ ((n >> 1) & 3)) << 4), ghw1 = n * 64 + (((2 * hg + 1) ^ ((n >> 1) & 3)) << 4);
    int tro[2];
    {
        const int s16 = lane & 15, a = s16 >> 2, b = s16 & 3, gsel = (lane >> 4) & 1;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int row = 8 * hg + 4 * hf + a;
            tro[hf] = row * 64 + (((2 * gsel + (b >> 1)) ^ ((row >> 1) & 3)) << 4) + (b & 1) * 8;
        }
    }
    const bool bit3 = (n >> 3) & 1, bit2 = (n >> 2) & 1, bit1 = (n >> 1) & 1, bit0 = n & 1;

    float dw2[2][4], db1[4], gacc = 0.f, lacc = 0.f;
    f32x16v accM[4][2];                                  // M: [hidden tile][channel parity]: row 32 mt + D row, column = channel 2 n + ct
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        db1[t] = 0.f;
        dw2[0][t] = dw2[1][t] = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) accM[t][c][r] = 0.f;
    }

    auto line_of = [&](int gl) {                         // cropped line -> padded line (32-bit: B * Tp * Hp lines)
        const unsigned h = (unsigned)gl % (unsigned)cm.H, r2 = (unsigned)gl / (unsigned)cm.H;
        return (int)(((r2 / (unsigned)cm.T) * cm.Tp + r2 % (unsigned)cm.T) * cm.Hp + h);
    };
    u32x4 xa[8];                                         // A layout: cell 32 q + n, channels 16 ks + 8 hg + 4 half ..   [2 ks + half]
    auto issue_xa = [&](int pl, int q) {                 // (past the wave's last tile the descriptor is empty: loads return 0, no branches)
        const bool ok = pl >= 0;
        const rsrc_t rx = make_rsrc(p.s + (long)(ok ? pl : 0) * cm.Wp * 64, ok ? (unsigned)cm.W * 256u : 0u);   // cells >= W read as 0
#pragma unroll
        for (int i = 0; i < 8; ++i) xa[i] = ld16(rx, (32 * q + n) * 256 + (i >> 1) * 64 + hg * 32 + (i & 1) * 16);
    };
    auto prefetch = [&](int pl, int q) -> unsigned {     // one dword per lane, 128 B apart: the next tile's 8 KB reach L2 a tile ahead
        const bool ok = pl >= 0;
        const rsrc_t rx = make_rsrc(p.s + (long)(ok ? pl : 0) * cm.Wp * 64, ok ? (unsigned)cm.W * 256u : 0u);
        return __builtin_amdgcn_raw_buffer_load_b32(rx, 32 * q * 256 + lane * 128, 0, 0);
    };
    issue_xa(slot < GL ? line_of((int)slot) : -1, 0);
#ifdef PG_TIMING
    float tacc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) tacc[i] = 0.f;
    long long tlast = clock64();
#endif
    for (long gl = slot; gl < GL; gl += nslots) {
        const int pl = line_of((int)gl);
        const int gln = gl + nslots < GL ? (int)(gl + nslots) : -1;
        const int pln = gln >= 0 ? line_of(gln) : -1;
        const rsrc_t ro = make_rsrc(p.g + (long)pl * cm.Wp * 64, line_bytes);
        const rsrc_t rxl = make_rsrc(p.s + (long)pl * cm.Wp * 64, (unsigned)cm.W * 256u);
        const rsrc_t rg = make_rsrc(p.gout + gl * cm.W * 2, (unsigned)cm.W * 8u);                        // target (LOSS) or gout: [W][2]
        for (int q = 0; q < TQ; ++q) {
            asm volatile("" ::: "memory");
            const bool last = q + 1 == TQ;
            const int pn = last ? pln : pl, qn = last ? 0 : q + 1;
            const unsigned pf = prefetch(pn, qn);
            // ---- the fc2-side inputs of the tile
            const float yv = buf_load_f32(rg, ((32 * q + celly) * 2 + (n & 1)) * 4, 0);                    // the lane's own (cell, feature) element
            f32x4v G4[8];                                // gout [register row r][feature]: G4[k] = (r = 2k: 0, 1 | r = 2k + 1: 0, 1)
            if (!LOSS) {
#pragma unroll
                for (int k = 0; k < 8; ++k) G4[k] = __builtin_bit_cast(f32x4v, ld16(rg, (32 * q + 8 * (k >> 1) + 4 * hg + 2 * (k & 1)) * 8));
            }
            PG_T(0)
            // ---- contraction 1: u = (s - mean) W1'^T + b1'
            f32x16v acc[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;               // (b1' is added at the activation: a resident splat costs 64 registers)
            // software pipeline over the K-steps: the planes of step ks + 1 (the split of the tile's next 16 channels, the 12 weight-plane
            // reads) are written BEFORE the 24 MFMAs of step ks and do not depend on them -- they issue in the shadow of the matrix pipe
            bf16x8 Ah[2], Am[2], Al[2], bh[2][4], bm[2][4], bl[2][4];
            auto prep = [&](int ks) {
                float v[8];
                const f32x4v x0 = __builtin_bit_cast(f32x4v, xa[2 * ks]), x1 = __builtin_bit_cast(f32x4v, xa[2 * ks + 1]);
                const f32x4v m0 = *reinterpret_cast<const f32x4v*>(meanl + 16 * ks + 8 * hg), m1 = *reinterpret_cast<const f32x4v*>(meanl + 16 * ks + 8 * hg + 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    v[c] = x0[c] - m0[c];
                    v[4 + c] = x1[c] - m1[c];
                }
                split8(v, Ah[ks & 1], Am[ks & 1], Al[ks & 1]);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    bh[ks & 1][nt] = __builtin_bit_cast(bf16x8, W1B[((ks * 4 + nt) * 3 + 0) * 64 + lane]);
                    bm[ks & 1][nt] = __builtin_bit_cast(bf16x8, W1B[((ks * 4 + nt) * 3 + 1) * 64 + lane]);
                    bl[ks & 1][nt] = __builtin_bit_cast(bf16x8, W1B[((ks * 4 + nt) * 3 + 2) * 64 + lane]);
                }
            };
            prep(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < 3) prep(ks + 1);
#define PG_ACC(c) acc[c]
#define PG_AH(c) Ah[ks & 1]
#define PG_AM(c) Am[ks & 1]
#define PG_AL(c) Al[ks & 1]
#define PG_BH(c) bh[ks & 1][c]
#define PG_BM(c) bm[ks & 1][c]
#define PG_BL(c) bl[ks & 1][c]
#define PG_FIRST(c) (ks == 0)
                PG_MAC6F(4, PG_ACC, PG_AH, PG_AM, PG_AL, PG_BH, PG_BM, PG_BL, PG_FIRST)
#undef PG_FIRST
#undef PG_ACC
#undef PG_AH
#undef PG_AM
#undef PG_AL
#undef PG_BH
#undef PG_BM
#undef PG_BL
#if PG_PIPE1
                if (ks < 3) {                            // spread the next step's preparation evenly behind this step's MFMAs
#pragma unroll
                    for (int i = 0; i < 24; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, PG_PIPE1, 0);
                        if (i < 14) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
            asm volatile("" ::"v"(pf));
            __builtin_amdgcn_sched_barrier(0);
            PG_T(1)
            // ---- the tile's second view (lane = channel pair, 16 cells): L1 / L2 hits, requested here so that only the hidden-tile loop
            //      holds them (the activation phase needs its registers for v and gelu')
            f32x2 xr[16];                                // [8 kstep + e]: cell 16 kstep + 8 (e >> 2) + 4 hg + (e & 3), channels 2 n, 2 n + 1
            auto issue_xr = [&]() {
#pragma unroll
                for (int i = 0; i < 16; ++i) xr[i] = ld8(rxl, (32 * q + 16 * (i >> 3) + 8 * ((i >> 2) & 1) + 4 * hg + (i & 3)) * 256 + n * 8);
            };
            // ---- activation: v = gelu(u) (kept for d fc2), gelu'(u) replaces u in the accumulators
            float VV[4][16];
            if (LOSS) {
                f32x2 pp[16];                            // [r] = (feature 0, feature 1): the lane's partial of out over its 4 hidden units
#pragma unroll
                for (int i = 0; i < 16; ++i) pp[i] = pk2(0.f);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const f32x2 wp = {w2r[0][nt], w2r[1][nt]};
#pragma unroll
                    for (int r = 0; r < 16; r += 4) {
                        f32x2 v, d, v2, d2;
                        gelu_both_p2(f32x2{acc[nt][r], acc[nt][r + 1]} + pk2(b1r[nt]), f32x2{acc[nt][r + 2], acc[nt][r + 3]} + pk2(b1r[nt]), v, d, v2, d2);
                        VV[nt][r] = v[0], VV[nt][r + 1] = v[1], VV[nt][r + 2] = v2[0], VV[nt][r + 3] = v2[1];
                        acc[nt][r] = d[0], acc[nt][r + 1] = d[1], acc[nt][r + 2] = d2[0], acc[nt][r + 3] = d2[1];
                        pp[r] = pk_fma(pk2(v[0]), wp, pp[r]);
                        pp[r + 1] = pk_fma(pk2(v[1]), wp, pp[r + 1]);
                        pp[r + 2] = pk_fma(pk2(v2[0]), wp, pp[r + 2]);
                        pp[r + 3] = pk_fma(pk2(v2[1]), wp, pp[r + 3]);
                    }
                }
                float po[32];
#pragma unroll
                for (int i = 0; i < 16; ++i) po[2 * i] = pp[i][0], po[2 * i + 1] = pp[i][1];
                PG_T(2)
                __builtin_amdgcn_sched_barrier(0);
                issue_xr();
                // sum over the 32 lanes of the half, halving the value set at every step: lane n ends with element n = 2 r + feature
                float q1[16], q2[8], q3[4], q4[2];
#pragma unroll
                for (int i = 0; i < 16; ++i) {           // lanes 16 apart: rows swap, then add (row 0 keeps i, row 1 keeps i + 16)
                    const u32x2 sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, po[i]), __builtin_bit_cast(unsigned, po[i + 16]), false, false);
                    q1[i] = asf(sw[0]) + asf(sw[1]);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) q2[i] = dpp_add(bit3 ? q1[i + 8] : q1[i], bit3 ? q1[i] : q1[i + 8], 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) q3[i] = dpp_add(bit2 ? q2[i + 4] : q2[i], bit2 ? q2[i] : q2[i + 4], 1);
#pragma unroll
                for (int i = 0; i < 2; ++i) q4[i] = dpp_add(bit1 ? q3[i + 2] : q3[i], bit1 ? q3[i] : q3[i + 2], 2);
                const float outv = dpp_add(bit0 ? q4[1] : q4[0], bit0 ? q4[0] : q4[1], 3) + b2y;
                const bool valid = 32 * q + celly < cm.W;                              // cells past the line end: no output element there
                const float diff = outv - yv;                                          // yv holds the target here
                lacc += valid ? diff * diff : 0.f;
                const float gv = valid ? p.gscale * diff : 0.f;
                gacc += gv;
                gb[lane] = gv;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int k = 0; k < 8; ++k) G4[k] = *reinterpret_cast<const f32x4v*>(gb + 32 * hg + 4 * k);
            } else {
                gacc += yv;                              // yv holds the lane's gout element here (cells >= W read as 0)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; r += 4) {
                        f32x2 v, d, v2, d2;
                        gelu_both_p2(f32x2{acc[nt][r], acc[nt][r + 1]} + pk2(b1r[nt]), f32x2{acc[nt][r + 2], acc[nt][r + 3]} + pk2(b1r[nt]), v, d, v2, d2);
                        VV[nt][r] = v[0], VV[nt][r + 1] = v[1], VV[nt][r + 2] = v2[0], VV[nt][r + 3] = v2[1];
                        acc[nt][r] = d[0], acc[nt][r + 1] = d[1], acc[nt][r + 2] = d2[0], acc[nt][r + 3] = d2[1];
                    }
                __builtin_amdgcn_sched_barrier(0);
                issue_xr();
            }
            PG_T(3)
            // ---- d fc2 += gout^T v  (here, so that v is dead before the planes and operands of the hidden-tile loop come alive; packed:
            //      no MFMA is in flight yet)
            {
                f32x2 dwp[4][2];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    dwp[nt][0] = f32x2{dw2[0][nt], dw2[1][nt]};
                    dwp[nt][1] = pk2(0.f);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r)                         // 8 independent chains advance together
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        dwp[nt][r & 1] = pk_fma(f32x2{G4[r >> 1][2 * (r & 1)], G4[r >> 1][2 * (r & 1) + 1]}, pk2(VV[nt][r]), dwp[nt][r & 1]);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const f32x2 t2 = dwp[nt][0] + dwp[nt][1];
                    dw2[0][nt] = t2[0], dw2[1][nt] = t2[1];
                }
            }
            PG_T(4)
            // ---- planes of the second view: (s - mean) with lane = channel, the B operand of the weight gradient
            bf16x8 Xh[2][2], Xm[2][2], Xl[2][2];         // [kstep][channel parity]
#pragma unroll
            for (int kstep = 0; kstep < 2; ++kstep) {
                float v0[8], v1[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v0[e] = xr[8 * kstep + e][0] - meanB0;
                    v1[e] = xr[8 * kstep + e][1] - meanB1;
                }
                split8(v0, Xh[kstep][0], Xm[kstep][0], Xl[kstep][0]);
                split8(v1, Xh[kstep][1], Xm[kstep][1], Xl[kstep][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            PG_T(5)
            // ---- per hidden tile: gh, its planes (once), the weight gradient from registers, the data gradient through LDS
            f32x16v acc3[2];                             // g^T: [channel tile mt]: row = channel 32 mt + D row, column = cell slot n
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc3[mt][r] = 0.f;
            // software pipeline over the hidden tiles: the vector work of tile nt + 1 (gh, its split, the plane stores) is written
            // AFTER the 48 MFMAs of tile nt and has no dependence on them, so it issues in their shadow (<= 5 scalar instructions per MFMA)
            bf16x8 Gh[2][2], Gm[2][2], Gl[2][2];         // [nt & 1][kstep]: register rows 8 kstep + e = cells 8 (2 kstep + (e >> 2)) + 4 hg + (e & 3)
            auto gh_stage = [&](int nt) {                // gh of hidden tile nt, its planes (once), the plane stores
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float g0 = G4[r >> 1][2 * (r & 1)], g1 = G4[r >> 1][2 * (r & 1) + 1];
                    const float gp = __builtin_fmaf(g1, w2r[1][nt], g0 * w2r[0][nt]);
                    const float gh = gp * acc[nt][r];                                   // cells >= W: gout == 0 -> gh == 0
                    acc[nt][r] = gh;
                    db1[nt] += gh;
                }
#pragma unroll
                for (int kstep = 0; kstep < 2; ++kstep) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = acc[nt][8 * kstep + e];
                    split8(v, Gh[nt & 1][kstep], Gm[nt & 1][kstep], Gl[nt & 1][kstep]);
                }
                char* ghb = GHw + (nt & 1) * PG_GH_BUF;
                *reinterpret_cast<u32x4*>(ghb + 0 * PG_GH_PLANE + ghw0) = __builtin_bit_cast(u32x4, Gh[nt & 1][0]);
                *reinterpret_cast<u32x4*>(ghb + 0 * PG_GH_PLANE + ghw1) = __builtin_bit_cast(u32x4, Gh[nt & 1][1]);
                *reinterpret_cast<u32x4*>(ghb + 1 * PG_GH_PLANE + ghw0) = __builtin_bit_cast(u32x4, Gm[nt & 1][0]);
                *reinterpret_cast<u32x4*>(ghb + 1 * PG_GH_PLANE + ghw1) = __builtin_bit_cast(u32x4, Gm[nt & 1][1]);
                *reinterpret_cast<u32x4*>(ghb + 2 * PG_GH_PLANE + ghw0) = __builtin_bit_cast(u32x4, Gl[nt & 1][0]);
                *reinterpret_cast<u32x4*>(ghb + 2 * PG_GH_PLANE + ghw1) = __builtin_bit_cast(u32x4, Gl[nt & 1][1]);
            };
            gh_stage(0);
            PG_T(6)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                asm volatile("" ::: "memory");           // the plane stores of this tile stay ahead of its transposing reads (LDS runs a wave's accesses in order)
                if (nt == 2) issue_xa(pn, qn);           // the next tile's A-layout loads (its lines were pulled into L2 a tile ago)
                char* ghb = GHw + (nt & 1) * PG_GH_BUF;
                bf16x8 Bp[2][3], ah[2][2], am[2][2], al[2][2];          // [kk]: transposed gh planes (lane = cell), W1^T planes [kk][mt]
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                    for (int pln_ = 0; pln_ < 3; ++pln_) {
                        typedef bf16x4 __attribute__((address_space(3))) * lds_b4;
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(ghb + pln_ * PG_GH_PLANE + kk * 1024 + tro[0]));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(ghb + pln_ * PG_GH_PLANE + kk * 1024 + tro[1]));
                        Bp[kk][pln_] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        ah[kk][mt] = __builtin_bit_cast(bf16x8, W1D[(((2 * nt + kk) * 2 + mt) * 3 + 0) * 64 + lane]);
                        am[kk][mt] = __builtin_bit_cast(bf16x8, W1D[(((2 * nt + kk) * 2 + mt) * 3 + 1) * 64 + lane]);
                        al[kk][mt] = __builtin_bit_cast(bf16x8, W1D[(((2 * nt + kk) * 2 + mt) * 3 + 2) * 64 + lane]);
                    }
                }
                asm volatile("" ::: "memory");           // ... and the next tile's plane stores behind these reads
                // M[nt] += gh^T (s - mean) over the 32 cells (two K-steps, both channel parities), g^T += W1^T gh^T over the tile's 32 hidden
                // units (K-steps ks3 = 2 nt, 2 nt + 1): four independent accumulators advance together
#pragma unroll
                for (int kstep = 0; kstep < 2; ++kstep) {
#define PG_ACC(c) (*((c) < 2 ? &accM[nt][(c)] : &acc3[(c) - 2]))
#define PG_AH(c) ((c) < 2 ? Gh[nt & 1][kstep] : ah[kstep][(c) - 2])
#define PG_AM(c) ((c) < 2 ? Gm[nt & 1][kstep] : am[kstep][(c) - 2])
#define PG_AL(c) ((c) < 2 ? Gl[nt & 1][kstep] : al[kstep][(c) - 2])
#define PG_BH(c) ((c) < 2 ? Xh[kstep][(c)] : Bp[kstep][0])
#define PG_BM(c) ((c) < 2 ? Xm[kstep][(c)] : Bp[kstep][1])
#define PG_BL(c) ((c) < 2 ? Xl[kstep][(c)] : Bp[kstep][2])
#define PG_FIRST(c) ((c) >= 2 && nt == 0 && kstep == 0)
                    PG_MAC6F(4, PG_ACC, PG_AH, PG_AM, PG_AL, PG_BH, PG_BM, PG_BL, PG_FIRST)
#undef PG_FIRST
#undef PG_ACC
#undef PG_AH
#undef PG_AM
#undef PG_AL
#undef PG_BH
#undef PG_BM
#undef PG_BL
                }
                if (nt < 3) gh_stage(nt + 1);
#if PG_PIPE
                if (nt < 3) {
#pragma unroll
                    for (int i = 0; i < 48; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, PG_PIPE, 0);     // PG_PIPE vector instructions
                    }
                }
#endif
                PG_T(8)
            }
            PG_T(9)
            // ---- g rows: channels 32 mt + 8 a + 4 hg .. + 3 of cell slot n (16 B per lane)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const f32x4v o = {acc3[mt][4 * a], acc3[mt][4 * a + 1], acc3[mt][4 * a + 2], acc3[mt][4 * a + 3]};
                    st16(o, ro, (32 * q + cells) * 256 + (32 * mt + 8 * a + 4 * hg) * 4);     // cells >= Wp: dropped; W .. Wp-1: zeros (gh == 0)
                }
        }
        PG_T(10)
        // margin cells 32 TQ .. Wp - 1 of the line (cells W .. 32 TQ - 1 were written as zeros by the last tile)
        for (int off = TQ * 32 * 256 + lane * 16; off < (int)line_bytes; off += 1024) st16(z4, ro, off);
    }

    if (LOSS) {
        const float ls = wave_sum(lacc);
        if (lane == 0) p.loss_part[slot] = ls;
    }
    // ---- the wave's partial row  (M = gh^T shat | d fc2 | d b1 | d b2)
    float* part = p.part + slot * ((long)PG_HID * 64 + 2L * PG_HID + PG_HID + 2);
    {
        const float is0 = p.xf.invstd[2 * n], is1 = p.xf.invstd[2 * n + 1];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int h = 32 * mt + 8 * (r >> 2) + 4 * hg + (r & 3);
                float2 o;
                o.x = accM[mt][0][r] * is0;
                o.y = accM[mt][1][r] * is1;
                *reinterpret_cast<float2*>(part + h * 64 + 2 * n) = o;
            }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {                     // the two lane halves hold different cells of the same hidden units
        float s1 = db1[nt];
        s1 += __shfl_xor(s1, 32, 64);
        if (hg == 0) part[PG_HID * 64 + 2 * PG_HID + 32 * nt + n] = s1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float s2 = dw2[j][nt];
            s2 += __shfl_xor(s2, 32, 64);
            if (hg == 0) part[PG_HID * 64 + j * PG_HID + 32 * nt + n] = s2;
        }
    }
#ifdef PG_TIMING
    if (lane < 12) {
        float tv = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) tv = lane == i ? tacc[i] : tv;
        part[lane] = tv;
    }
#endif
    {                                                    // d b2: lane n accumulated gout of feature n & 1
        float s3 = gacc;
#pragma unroll
        for (int off = 2; off < 64; off <<= 1) s3 += __shfl_xor(s3, off, 64);
        if (lane < 2) part[PG_HID * 64 + 2 * PG_HID + PG_HID + lane] = s3;
    }
}

size_t pjg_lds() { return (size_t)(4 * 4 * 3 * 64 + 8 * 2 * 3 * 64) * 16 + (PG_HID + 64 + PG_WAVES * 64) * 4 + (size_t)PG_WAVES * PG_GH_WAVE; }

int pjg_supported(int DO) {
    static const bool off = getenv("RPB_HEAD_PJG") && atoi(getenv("RPB_HEAD_PJG")) == 0;
    return !off && DO == 2;
}

int pjg_launch(PjfArgs& p, bool loss, int grid, hipStream_t st) {
    const size_t lds = pjg_lds();
    if (loss) {
        (void)hipFuncSetAttribute((const void*)pjg_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((pjg_kernel<true>), dim3(grid), dim3(PG_WAVES * 64), lds, st, p);
    } else {
        (void)hipFuncSetAttribute((const void*)pjg_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((pjg_kernel<false>), dim3(grid), dim3(PG_WAVES * 64), lds, st, p);
    }
    RPB_CHECK_LAUNCH("head_bwd");
}
