// The projection head's FORWARD for evaluation / rollout (fno.py:121-125: out = fc2 gelu(fc1 a + b1) + b2 on the cropped cells,
// a = BN(s_{L-1}) applied while loading) -- third organisation ("pjh"), the default for fc2 widths <= 4 with fp32 storage.
//
// rpb_pjx.hip's forward (16x16x32 tiles, lane = cell) spends per 32 cells 192 MFMA issues, 96 LDS reads of weight planes and ~1 900
// vector instructions, 17 of them per GELU: 1.37-1.42 ms at the headline shape for 2.77 GB (0.24 of the HBM peak, the worst kernel of the
// rollout).  This kernel applies the rules DESIGN.md section 4.0000 read off tools/ubench/issue_probe.hip, and the forward half of rpb_pjg.hip:
//
//   * u = (s - mean) W1'^T on v_mfma_f32_32x32x16_bf16 (cells = rows: 96 MFMAs and 48 LDS reads per 32-cell tile), W1' = W1 diag(gamma invstd)
//     and b1' = b1 + W1 beta formed per workgroup while the planes are staged: BatchNorm costs one subtraction per element;
//   * two waves per SIMD (two workgroups of four waves per CU, 49 KB of LDS each): a wave alone issues one vector instruction per ~5 cycles,
//     two issue one per 2.6; the weight planes of a K-step live in 48 registers (one buffer per plane kind, refilled behind its last use)
//     instead of pjg's 96, so that the wave fits 256 registers;
//   * every vector instruction in its scalar form (the file is compiled without packed fp32: a v_pk_* waits for the partner wave's MFMAs);
//   * the bias rides in the two affine uses of u inside GELU (x / sqrt 2 and x / 2 become FMAs with per-lane constants): 14 instructions
//     + one exponential per value (the sign transfer of erf is the |.| source modifier of the last FMA), same polynomial as fast_erf (rpb_common.h);
//   * fc2 (DOT = 2 or 4 outputs, padded): DOT FMAs per value into 16 DOT per-lane partials, then the halving butterfly of rpb_pjg.hip over the 32 lanes of a half
//     (v_permlane16_swap, row_mirror, row_half_mirror, two quad_perm adds): lane n ends with elements (DOT / 2) n .. = (cell, output) of the tile -- one
//     or two dword stores per lane, contiguous over the wave.
//
// Layouts as in rpb_pjg.hip (32x32x16: A lane (m = l & 31, kg = l >> 5) holds k = 8 kg + e; D register r of lane (n, hg) is row
// 8 (r >> 2) + 4 hg + (r & 3), column n).
#include "rpb_pjx.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x16v __attribute__((ext_vector_type(16)));

#define PH_HID 128
#define PH_WAVES 4
// PH_PIPE = n > 0: ask the scheduler for n vector instructions behind every MFMA of contraction 1 (0: its own order)
#ifndef PH_PIPE
#define PH_PIPE 2
#endif

#ifndef RPB_HEAD_AUX
#define RPB_HEAD_AUX 0   /* cache policy of the tile loads / stores (2 = nt): experiment switch */
#endif
namespace {
__device__ __forceinline__ u32x4 ld16(rsrc_t r, int voff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, RPB_HEAD_AUX));
}
__device__ __forceinline__ float asf(unsigned u) { return __builtin_bit_cast(float, u); }
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
// H2 -- the opt-in "f16x2" eval arithmetic (rpb_cmx.hip, same contract): two fp16 planes per operand, both rounded to nearest even, three
// products hi*lo + lo*hi + hi*hi on v_mfma_f32_32x32x16_f16 (dropped term <= 2^-22 |a b|); the planes travel in the bf16x8 containers,
// slot 0 = hi, slot 1 = lo.  W1' carries 2^PH_H2W (fp16's range), undone in the two affine uses of u inside GELU.
#define PH_H2W 4
#ifndef RPB_H2_FMAMIX
#define RPB_H2_FMAMIX 0   /* measured (profiles/r06b_ab.txt): the head +1 % with v_fma_mix_f32, the eval cell_mix launches -1.5 .. -7 % (rpb_cmx.hip keeps 1) */
#endif
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2w __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8h(const float (&v)[8], bf16x8& h, bf16x8& l) {
    u32x4 uh, ul;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x2w ab = {v[2 * q], v[2 * q + 1]};
        const f16x2v hh = __builtin_convertvector(ab, f16x2v);                       // v_cvt_pk_f16_f32 (RNE)
        uh[q] = __builtin_bit_cast(unsigned, hh);
#if RPB_H2_FMAMIX
        // residual a - float(hi) as ONE v_fma_mix_f32 per value (f16 half * -1 + f32; exact): 4 instead of 5 instructions per value pair
        // (left alone the compiler converts both halves and subtracts packed: 2 x v_cvt_f32_f16 + v_pk_add_f32)
        float r0, r1;
        const unsigned hu = uh[q];
        const float a0 = ab[0], a1 = ab[1];
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hu), "v"(a0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hu), "v"(a1));
        const f32x2w r = {r0, r1};
#else
        const f32x2w r = ab - __builtin_convertvector(hh, f32x2w);                   // exact in fp32
#endif
        const f16x2v ll = __builtin_convertvector(r, f16x2v);
        ul[q] = __builtin_bit_cast(unsigned, ll);
    }
    h = __builtin_bit_cast(bf16x8, uh);
    l = __builtin_bit_cast(bf16x8, ul);
}
__device__ __forceinline__ f32x16v mfma32h(bf16x8 a, bf16x8 b, f32x16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// gelu(a + b) with the bias folded into the two affine uses of u: bs = b / sqrt 2, hb = b / 2.  The erf is fast_erf's (rpb_common.h):
// erf(x) = sign(x) (1 - 2^(t S(t))), t = min(|x|, 4).  SC: the scale the accumulator carries (f16x2: 2^-PH_H2W), folded into both FMAs
template <bool H2 = false>
__device__ __forceinline__ float gelu_bias(float a, float bs, float hb) {
    constexpr float SC = H2 ? 1.0f / (float)(1 << PH_H2W) : 1.0f;
    const float xs = __builtin_fmaf(a, 0.70710678118654752440f * SC, bs);
    const float t = fminf(fabsf(xs), 4.0f);
    float p = 1.160457393e-05f;
    p = __builtin_fmaf(p, t, -1.529619341e-04f);
    p = __builtin_fmaf(p, t, 8.482242992e-04f);
    p = __builtin_fmaf(p, t, -2.274763673e-03f);
    p = __builtin_fmaf(p, t, 8.477856228e-05f);
    p = __builtin_fmaf(p, t, 2.772449465e-02f);
    p = __builtin_fmaf(p, t, -1.483079179e-01f);
    p = __builtin_fmaf(p, t, -9.184428993e-01f);
    p = __builtin_fmaf(p, t, -1.627907267e+00f);
    const float e = __builtin_amdgcn_exp2f(p * t);
    const float hx = __builtin_fmaf(a, 0.5f * SC, hb);     // x / 2 has the sign of x / sqrt 2:  hx erf(xs) = |hx| (1 - e)
    return __builtin_fmaf(fabsf(hx), 1.0f - e, hx);
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

struct PjhFwdArgs {
    const float* s;       // padded tensor of the last Fourier layer [B*Tp*Hp*Wp][64]
    const float* w1;      // fc1.weight [128][64]
    const float* b1;      // [128]
    const float* w2;      // fc2.weight [DO][128]
    const float* b2;      // [DO]
    float* out;           // [ncrop][DO]
    const float* gout;    // BWD: dLoss/dout [ncrop][DO]
    float* gu;            // BWD: (fc2^T gout) * gelu'(u)  [ncrop][128]
    float* part;          // BWD: one row per wave [DO*128 (d fc2.weight) | 128 (d fc1.bias) | DO (d fc2.bias)] -- rpb_proj_bwd's format
    int B, DO;
    CropMap cm;
    XForm xf;             // BatchNorm of the last layer (mean == nullptr: plain tensor); gelu must be 0
};

// DOT: fc2 outputs padded to 2 or 4 (DO = 1 .. DOT of them are real; the others carry zero weights and are not stored)
// BFIN: the activations are STORED as bf16 (BASELINE.json configs[4]; `s` is then a bf16 [cells][64] tensor): a lane's 16 B load is its
//       whole A operand of a K-step (one exact plane: no mean subtraction -- the caller passes plain activations --, no split), and it meets
//       RPB_BF16_CONST_PLANES planes of fc1.weight: 8 (12) MFMAs per K-step instead of 24.
// CW:   input channels, 64 or 128 (configs/fsi/fno.yaml; round 6b).  At 128 the fc1 planes are 96 KB of LDS: ONE workgroup of four waves per CU
//       (one wave per SIMD, the whole register file: 8 K-steps of inputs in flight), fp32 storage and the default arithmetic only.
// BWD:  rpb_proj_bwd's job at C = 128 (round 6b): u is recomputed on the matrix pipe as in the forward, then per lane (hidden unit 32 nt + n, its 16
//       cells) gu = (fc2^T gout) * gelu'(u) is stored for the data / weight gradient kernels and d fc2.weight, d fc1.bias, d fc2.bias
//       accumulate in registers (one partial row per wave, the format of csrc/rpb_proj.hip).  gelu / gelu' from one erf + one exponential,
//       the arithmetic of that kernel.
// MODE 2 ("DG"): the fc1 DATA gradient at C = 128 -- g[padded cell][ch] = sum_hid gu[cropped cell][hid] fc1.weight[hid][ch], zeros in the pad
//       (rpb_cell_mix(gather) on the fp32 pipe before): `s` is gu [ncrop][128], `w1` fc1.weight [128 hid][128 ch] read transposed, `out`
//       the padded gradient tensor [cells][128]; no bias, no activation: the accumulators are stored as they are.
// SILU: the activation of the Galerkin SpectralRegressor's head (galerkin_transformer_libs/model.py:631-632) instead of the exact GELU; width 128
//       forward / backward instances only (that head is 128 wide).
template <int DOT, bool H2 = false, bool BFIN = false, int CW = 64, int MODE = 0, bool SILU = false>
__global__ __launch_bounds__(PH_WAVES * 64, CW == 64 ? 2 : 1) void pjh_fwd_kernel(PjhFwdArgs p) {
    static_assert(!SILU || (CW == 128 && MODE < 2), "SiLU: the width-128 head");
    constexpr bool BWD = MODE == 1, DG = MODE == 2 || MODE == 3, DGS = MODE == 3;
    // MODE 3: MODE 2 + the BatchNorm-backward sums of the last Fourier layer in the epilogue -- per channel sum g and sum g * shat with
    //         shat = (s - mean) invstd read from `gout` (= the pre-BN tensor s, padded layout) and `xf` (that layer's statistics); one
    //         partial row [2][128] per wave in `part` (rpb_cell_mix's STATS = 2 format): bn_bwd_reduce's pass over (s, g) disappears
    static_assert(MODE == 0 || (CW == 128 && !H2 && !BFIN), "the backward instances serve width 128 (width 64 has the one-launch head)");
    static_assert(!(H2 && BFIN), "f16x2 is an fp32-storage arithmetic");
    static_assert(CW == 64 || (CW == 128 && !H2 && !BFIN), "C = 128: fp32 storage, default arithmetic");
    constexpr int KS = CW / 16;                          // K-steps of contraction 1
    constexpr int NV = 16 * DOT;                         // fc2 partials per lane: [register row r][output j] = element DOT r + j
    constexpr int EPL = NV / 32;                         // elements a lane owns after the butterfly: element EPL n + k
    extern __shared__ u32x4 lds4[];
    u32x4* W1B = lds4;                                   // [ks 4][nt 4][plane 3][lane]   B of u:   W1'[32 nt + n][16 ks + 8 kg + e]
    float* b1l = reinterpret_cast<float*>(W1B + KS * 4 * 3 * 64);         // [128]  b1' = b1 + W1 beta
    float* meanl = b1l + PH_HID;                         // [CW]   BatchNorm mean
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hg = lane >> 5;
    const bool has_xf = !DG && p.xf.mean != nullptr;     // (DG: `xf` carries the statistics of the sums, not an input transform)
    for (int idx = tid; idx < KS * 4 * 64; idx += blockDim.x) {
        const int l = idx & 63, nt = (idx >> 6) & 3, ks = idx >> 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 16 * ks + 8 * (l >> 5) + e;
            const float w = DG ? p.w1[c * CW + 32 * nt + (l & 31)] : p.w1[(32 * nt + (l & 31)) * CW + c];
            v[e] = has_xf ? w * (p.xf.gamma[c] * p.xf.invstd[c]) : w;
            if (H2) v[e] *= (float)(1 << PH_H2W);
        }
        bf16x8 h, m, lo;
        if (H2) {
            split8h(v, h, m);
            lo = m;
        } else {
            split8(v, h, m, lo);
        }
        W1B[((ks * 4 + nt) * 3 + 0) * 64 + l] = __builtin_bit_cast(u32x4, h);
        W1B[((ks * 4 + nt) * 3 + 1) * 64 + l] = __builtin_bit_cast(u32x4, m);
        W1B[((ks * 4 + nt) * 3 + 2) * 64 + l] = __builtin_bit_cast(u32x4, lo);
    }
    for (int h = tid; h < PH_HID; h += blockDim.x) {
        float a = DG ? 0.f : p.b1[h];
        if (has_xf)
            for (int c = 0; c < CW; ++c) a = __builtin_fmaf(p.w1[h * CW + c], p.xf.beta[c], a);
        b1l[h] = a;
    }
    if (tid < CW) meanl[tid] = has_xf ? p.xf.mean[tid] : 0.f;
    __syncthreads();

    const CropMap cm = p.cm;
    const long nslots = (long)gridDim.x * PH_WAVES;
    const long slot = (long)blockIdx.x * PH_WAVES + wave;
    const long GL = (long)p.B * cm.T * cm.H;
    const int TQ = (cm.W + 31) >> 5;
    // per-lane constants: hidden unit 32 nt + n
    const int DO = p.DO;
    float bs[4], hb[4], w2r[DOT][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const float b = b1l[32 * nt + n];
        bs[nt] = b * 0.70710678118654752440f;
        hb[nt] = 0.5f * b;
#pragma unroll
        for (int j = 0; j < DOT; ++j) w2r[j][nt] = j < DO ? p.w2[j * PH_HID + 32 * nt + n] : 0.f;
    }
    float meanr[8 * KS];                                 // BatchNorm mean of the lane's channels 16 ks + 8 hg + e  [8 ks + e]
#pragma unroll
    for (int i = 0; i < 8 * KS; ++i) meanr[i] = meanl[16 * (i >> 3) + 8 * hg + (i & 7)];
    // the lane's own output elements after the cross-lane reduction: EPL n + k = DOT ry + j -> register row ry = cell 8 (ry >> 2) + 4 hg + (ry & 3)
    const int ry = (EPL * n) / DOT, jy = (EPL * n) % DOT;
    const int celly = 8 * (ry >> 2) + 4 * hg + (ry & 3);
    float b2y[EPL];
#pragma unroll
    for (int k = 0; k < EPL; ++k) b2y[k] = jy + k < DO ? p.b2[jy + k] : 0.f;
    const bool bit3 = (n >> 3) & 1, bit2 = (n >> 2) & 1, bit1 = (n >> 1) & 1, bit0 = n & 1;
    float dw2[BWD ? DOT : 1][4], db1a[4], db2a[BWD ? DOT : 1], b1r[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        db1a[nt] = 0.f;
        b1r[nt] = b1l[32 * nt + n];
#pragma unroll
        for (int j = 0; j < (BWD ? DOT : 1); ++j) dw2[j][nt] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < (BWD ? DOT : 1); ++j) db2a[j] = 0.f;
    float smu[4], sis[4], sg1[4], sg2[4];                // DGS: mean / invstd of channel 32 nt + n, the two sums
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        smu[nt] = DGS ? p.xf.mean[32 * nt + n] : 0.f;
        sis[nt] = DGS ? p.xf.invstd[32 * nt + n] : 0.f;
        sg1[nt] = sg2[nt] = 0.f;
    }

    auto line_of = [&](int gl) {                         // cropped line -> padded line (32-bit: B * Tp * Hp lines)
        const unsigned h = (unsigned)gl % (unsigned)cm.H, r2 = (unsigned)gl / (unsigned)cm.H;
        return (int)(((r2 / (unsigned)cm.T) * cm.Tp + r2 % (unsigned)cm.T) * cm.Hp + h);
    };
    u32x4 xa[2 * KS];                                    // A layout: cell 32 q + n, channels 16 ks + 8 hg + 4 half ..   [2 ks + half]
    auto issue_pair = [&](int pl, int q, int ks) {       // (past the wave's last tile the descriptor is empty: loads return 0, no branches)
        const bool ok = pl >= 0;
        if (BFIN) {                                      // 128 B cell rows; channels 16 ks + 8 hg .. + 7 = 16 B
            const rsrc_t rb = make_rsrc(p.s + (long)(ok ? pl : 0) * cm.Wp * 32, ok ? (unsigned)cm.W * 128u : 0u);
            xa[2 * ks] = ld16(rb, (32 * q + n) * 128 + ks * 32 + hg * 16);
            return;
        }
        // (DG: `pl` is the CROPPED line index, the rows are gu's)
        const rsrc_t rx = make_rsrc(p.s + (long)(ok ? pl : 0) * (DG ? cm.W : cm.Wp) * CW, ok ? (unsigned)cm.W * (CW * 4u) : 0u);   // cells >= W read as 0
        xa[2 * ks] = ld16(rx, (32 * q + n) * (CW * 4) + ks * 64 + hg * 32);
        xa[2 * ks + 1] = ld16(rx, (32 * q + n) * (CW * 4) + ks * 64 + hg * 32 + 16);
    };
    {
        const int pl0 = slot < GL ? (DG ? (int)slot : line_of((int)slot)) : -1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) issue_pair(pl0, 0, ks);
    }
    // weight planes of a K-step: one buffer per plane kind, refilled behind its last use (those of K-step 0 are loaded for the NEXT tile
    // behind the last products of this one)
    bf16x8 BL[4], BM[4], BH[4];
#define PH_LOADB(DST, KS, PLANE) _Pragma("unroll") for (int nt_ = 0; nt_ < 4; ++nt_) DST[nt_] = __builtin_bit_cast(bf16x8, W1B[(((KS) * 4 + nt_) * 3 + (PLANE)) * 64 + lane]);
    if (!H2 && !(BFIN && RPB_BF16_CONST_PLANES <= 2)) { PH_LOADB(BL, 0, 2) }
    PH_LOADB(BM, 0, 1)
    PH_LOADB(BH, 0, 0)
#ifdef PH_TIMING   /* timing-only build: the wave's shader cycles and 100 MHz ticks over its tile loop land in out[2 slot ..] (tools/dbg/pjh_clock.py) */
    const long long tc0 = clock64(), tw0 = wall_clock64();
#endif
    // (lines are dealt statically.  Claimed chip-wide from a counter -- built and measured with -DPH_TIMING: every wave then ends within 4 %
    // of the mean instead of 0.8 .. 1.2 of it, and the launch takes the same 1.57 ms: a workgroup that finishes early leaves its CU to its
    // partner, which speeds up; the bound is per CU, not per wave)
    if constexpr (DG) {                                  // the pad lines of the gradient tensor: zeros, 1 KB per instruction
        const long nl = (long)p.B * cm.Tp * cm.Hp;
        for (long g = slot; g < nl; g += nslots) {
            const int h = (int)(g % cm.Hp), t = (int)((g / cm.Hp) % cm.Tp);
            if (h < cm.H && t < cm.T) continue;
            const rsrc_t rz = make_rsrc(p.out + g * cm.Wp * CW, (unsigned)cm.Wp * (CW * 4u));
            for (int off = lane * 16; off < cm.Wp * CW * 4; off += 64 * 16) __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, rz, off, 0, 0);
        }
    }
    for (long gl = slot; gl < GL;) {
        const int pl = line_of((int)gl);
        const long gnext = gl + nslots;
        const int gln = gnext < GL ? (int)gnext : -1;
        const int pln = gln >= 0 ? line_of(gln) : -1;
        const rsrc_t rgo = make_rsrc(p.out + gl * cm.W * DO, (unsigned)(cm.W * DO) * 4u);
        for (int q = 0; q < TQ; ++q) {
            asm volatile("" ::: "memory");
            const bool last = q + 1 == TQ;
            const int pn = DG ? (last ? gln : (int)gl) : (last ? pln : pl), qn = last ? 0 : q + 1;
            float sp[DGS ? 4 : 1][DGS ? 16 : 1];         // DGS: s at the lane's output positions, requested before the products
            if constexpr (DGS) {
                const rsrc_t rs = make_rsrc(p.gout + (long)pl * cm.Wp * CW, (unsigned)cm.Wp * (CW * 4u));
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        sp[nt][r] = buf_load_f32(rs, ((32 * q + 8 * (r >> 2) + 4 * hg + (r & 3)) * CW + 32 * nt + n) * 4, 0);
            }
            // ---- contraction 1: u = (s - mean) W1'^T, software-pipelined over the K-steps: the split of step ks + 1 is written before the
            //      24 MFMAs of step ks and does not depend on them
            f32x16v acc[4];
#ifdef PH_NOMFMA
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#endif
            bf16x8 Ah[2], Am[2], Al[2];
            auto prep = [&](int ks) {
                if (BFIN) {                              // the load IS the operand
                    Ah[ks & 1] = __builtin_bit_cast(bf16x8, xa[2 * ks]);
                    issue_pair(pn, qn, ks);
                    return;
                }
                float v[8];
                const f32x4v x0 = __builtin_bit_cast(f32x4v, xa[2 * ks]), x1 = __builtin_bit_cast(f32x4v, xa[2 * ks + 1]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    v[c] = x0[c] - meanr[8 * ks + c];
                    v[4 + c] = x1[c] - meanr[8 * ks + 4 + c];
                }
                if (H2) split8h(v, Ah[ks & 1], Am[ks & 1]);
                else split8(v, Ah[ks & 1], Am[ks & 1], Al[ks & 1]);
                issue_pair(pn, qn, ks);                  // the registers are free: the next tile's loads fly through the rest of this one
            };
            prep(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks < KS - 1) prep(ks + 1);
                const int kn = ks < KS - 1 ? ks + 1 : 0;
                // small terms first within a plane kind; BL, then BM, then BH
#ifdef PH_NOMFMA   /* timing-only build: the products replaced by one vector instruction per group */
#define PH_G(AP, BP, FIRST) _Pragma("unroll") for (int nt_ = 0; nt_ < 4; ++nt_) acc[nt_][0] = ((FIRST) ? 0.f : acc[nt_][0]) + __builtin_bit_cast(float, __builtin_bit_cast(u32x4, AP[ks & 1])[0] ^ __builtin_bit_cast(u32x4, BP[nt_])[1]);
#else
#define PH_G(AP, BP, FIRST) _Pragma("unroll") for (int nt_ = 0; nt_ < 4; ++nt_) acc[nt_] = mfma32(AP[ks & 1], BP[nt_], (FIRST) ? f32x16v{} : acc[nt_]);
#endif
#define PH_GH(AP, BP, FIRST) _Pragma("unroll") for (int nt_ = 0; nt_ < 4; ++nt_) acc[nt_] = mfma32h(AP[ks & 1], BP[nt_], (FIRST) ? f32x16v{} : acc[nt_]);
                if (BFIN) {                     // one stored plane x the planes of the constants, small terms first
                    if (RPB_BF16_CONST_PLANES > 2) {
                        PH_G(Ah, BL, ks == 0)
                        PH_LOADB(BL, kn, 2)
                    }
                    PH_G(Ah, BM, ks == 0 && RPB_BF16_CONST_PLANES <= 2)
                    PH_LOADB(BM, kn, 1)
                    PH_G(Ah, BH, false)
                    PH_LOADB(BH, kn, 0)
                    continue;
                }
                if (H2) {                       // slot 1 (the "m" registers) holds the lo plane
                    PH_GH(Ah, BM, ks == 0)
                    PH_LOADB(BM, kn, 1)
                    PH_GH(Am, BH, false)
                    PH_GH(Ah, BH, false)
                    PH_LOADB(BH, kn, 0)
#if PH_PIPE
#pragma unroll
                    for (int i = 0; i < 12; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (ks < KS - 1) __builtin_amdgcn_sched_group_barrier(0x002, PH_PIPE, 0);
                        if (i >= 4 && i < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        if (ks < KS - 1 && (i == 4 || i == 8)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    __builtin_amdgcn_sched_barrier(0);
#endif
                    continue;
                }
                PH_G(Ah, BL, ks == 0)
                PH_LOADB(BL, kn, 2)
                PH_G(Am, BM, false)
                PH_G(Ah, BM, false)
                PH_LOADB(BM, kn, 1)
                PH_G(Al, BH, false)
                PH_G(Am, BH, false)
                PH_G(Ah, BH, false)
                PH_LOADB(BH, kn, 0)
#undef PH_G
#undef PH_GH
#if PH_PIPE
                // the order asked of the scheduler: every MFMA followed by PH_PIPE vector instructions of the next step's split; the
                // refill of a plane buffer one read per MFMA behind the buffer's last product
#pragma unroll
                for (int i = 0; i < 24; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (ks < KS - 1) __builtin_amdgcn_sched_group_barrier(0x002, PH_PIPE, 0);
                    if ((i >= 4 && i < 8) || (i >= 12 && i < 16)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (ks < KS - 1 && (i == 8 || i == 16)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DG) {
                // ---- g[padded cell 32 q + row][channel 32 nt + n] = the accumulator (cells >= W: gu read 0 -> exact zeros; cells >= Wp dropped)
                const rsrc_t rg = make_rsrc(p.out + (long)pl * cm.Wp * CW, (unsigned)cm.Wp * (CW * 4u));
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                    {
                        buf_store_f32(acc[nt][r], rg, ((32 * q + 8 * (r >> 2) + 4 * hg + (r & 3)) * CW + 32 * nt + n) * 4, 0);
                        if constexpr (DGS) {             // (cells >= W: g == 0 exactly, whatever s holds there)
                            sg1[nt] += acc[nt][r];
                            sg2[nt] = __builtin_fmaf(acc[nt][r], (sp[nt][r] - smu[nt]) * sis[nt], sg2[nt]);
                        }
                    }
                if (last)                                // the rest of the w pad
                    for (int off = 32 * TQ * CW * 4 + lane * 16; off < cm.Wp * CW * 4; off += 64 * 16)
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, rg, off, 0, 0);
                continue;
            }
            if constexpr (BWD) {
                // ---- gu[cell][hid] = (sum_j gout[cell][j] w2[j][hid]) gelu'(u);  d w2, d b1, d b2 (cells >= W: gout reads 0, the store is dropped)
                const long row0 = gl * cm.W + 32 * q;
                const rsrc_t gr = make_rsrc(p.gout + row0 * DO, tile_bytes((long)cm.W - 32 * q, 32, DO * 4));
                const rsrc_t ur = make_rsrc(p.gu + row0 * PH_HID, tile_bytes((long)cm.W - 32 * q, 32, PH_HID * 4));
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cell = 8 * (r >> 2) + 4 * hg + (r & 3);
                    float g[DOT];
#pragma unroll
                    for (int j = 0; j < DOT; ++j) {
                        g[j] = j < DO ? buf_load_f32(gr, (cell * DO + j) * 4, 0) : 0.f;
                        db2a[j] += g[j];
                    }
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const float u = acc[nt][r] + b1r[nt];
                        float v, dd;
                        if (SILU) {                      // silu(u) = u sig(u), silu'(u) = sig (1 + u (1 - sig))   (csrc/rpb_proj.hip silu_pair)
                            const float sig = 1.0f / (1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * u));
                            v = u * sig;
                            dd = sig * (1.0f + u * (1.0f - sig));
                        } else {
                            const float cdf = 0.5f * (1.0f + fast_erf(u * 0.70710678118654752440f));
                            const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * u * u);
                            v = u * cdf;
                            dd = __builtin_fmaf(u, pdf, cdf);
                        }
                        float gvs = 0.f;
#pragma unroll
                        for (int j = 0; j < DOT; ++j) {
                            gvs = __builtin_fmaf(g[j], w2r[j][nt], gvs);
                            dw2[j][nt] = __builtin_fmaf(g[j], v, dw2[j][nt]);
                        }
                        const float guv = gvs * dd;
                        buf_store_f32(guv, ur, (cell * PH_HID + 32 * nt + n) * 4, 0);
                        db1a[nt] += guv;
                    }
                }
                continue;
            }
            // ---- activation + fc2 partials: the lane's hidden units 32 nt + n, cells = its 16 register rows
            float po[NV];                                // [DOT r + j]
#pragma unroll
            for (int i = 0; i < NV; ++i) po[i] = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
#ifdef PH_NOACT   /* timing-only build */
                    const float v = acc[nt][r] + bs[nt];
#else
                    float v;
                    if (SILU) {
                        const float u = acc[nt][r] + b1r[nt];
                        v = u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * u));
                    } else {
                        v = gelu_bias<H2>(acc[nt][r], bs[nt], hb[nt]);
                    }
#endif
#pragma unroll
                    for (int j = 0; j < DOT; ++j) po[DOT * r + j] = __builtin_fmaf(v, w2r[j][nt], po[DOT * r + j]);
                }
            // ---- sum over the 32 lanes of the half, halving the value set at every step: lane n ends with elements EPL n + k
            float q1[NV / 2], q2[NV / 4], q3[NV / 8], q4[NV / 16], q5[EPL];
#pragma unroll
            for (int i = 0; i < NV / 2; ++i) {           // lanes 16 apart: rows swap, then add (row 0 keeps i, row 1 keeps i + NV / 2)
                const u32x2 sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, po[i]), __builtin_bit_cast(unsigned, po[i + NV / 2]), false, false);
                q1[i] = asf(sw[0]) + asf(sw[1]);
            }
#pragma unroll
            for (int i = 0; i < NV / 4; ++i) q2[i] = dpp_add(bit3 ? q1[i + NV / 4] : q1[i], bit3 ? q1[i] : q1[i + NV / 4], 0);
#pragma unroll
            for (int i = 0; i < NV / 8; ++i) q3[i] = dpp_add(bit2 ? q2[i + NV / 8] : q2[i], bit2 ? q2[i] : q2[i + NV / 8], 1);
#pragma unroll
            for (int i = 0; i < NV / 16; ++i) q4[i] = dpp_add(bit1 ? q3[i + NV / 16] : q3[i], bit1 ? q3[i] : q3[i + NV / 16], 2);
#pragma unroll
            for (int i = 0; i < EPL; ++i) q5[i] = dpp_add(bit0 ? q4[i + EPL] : q4[i], bit0 ? q4[i] : q4[i + EPL], 3) + b2y[i];
#pragma unroll
            for (int k = 0; k < EPL; ++k)                // cells >= W: dropped by the descriptor; padded outputs: not stored
                if (jy + k < DO) buf_store_f32(q5[k], rgo, ((32 * q + celly) * DO + jy + k) * 4, 0);
        }
        gl = gnext;
    }
    if constexpr (DGS) {
        float* part = p.part + slot * (2 * CW);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const float a1 = sg1[nt] + __shfl_xor(sg1[nt], 32, 64), a2 = sg2[nt] + __shfl_xor(sg2[nt], 32, 64);
            if (hg == 0) {
                part[32 * nt + n] = a1;
                part[CW + 32 * nt + n] = a2;
            }
        }
    }
    if constexpr (BWD) {
        float* part = p.part + slot * ((long)DO * PH_HID + PH_HID + DO);
#pragma unroll
        for (int j = 0; j < DOT; ++j) {
            if (j < DO) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const float v = dw2[j][nt] + __shfl_xor(dw2[j][nt], 32, 64);
                    if (hg == 0) part[j * PH_HID + 32 * nt + n] = v;
                }
                const float b = db2a[j] + __shfl_xor(db2a[j], 32, 64);     // every lane of a half holds the same row sums
                if (lane == 0) part[DO * PH_HID + PH_HID + j] = b;
            }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const float v = db1a[nt] + __shfl_xor(db1a[nt], 32, 64);
            if (hg == 0) part[DO * PH_HID + 32 * nt + n] = v;
        }
    }
#ifdef PH_TIMING
    __builtin_amdgcn_s_waitcnt(0);
    const long long tc1 = clock64(), tw1 = wall_clock64();
    if (lane == 0) {
        p.out[2 * slot] = (float)(tc1 - tc0);
        p.out[2 * slot + 1] = (float)(tw1 - tw0);
    }
#endif
#undef PH_LOADB
}

int rpb_pjh_bwd128_launch(const float* s, const float* w1, const float* b1, const float* w2, const float* b2, const float* gout, float* gu,
                          float* part, long part_rows, int B, int DO, int T, int H, int W, int Tp, int Hp, int Wp, const XForm& xf, hipStream_t st,
                          bool silu);
static size_t pjh_lds(int CW = 64) { return (size_t)((CW / 16) * 4 * 3 * 64) * 16 + (PH_HID + CW) * 4; }

// 1 when this kernel takes the shape: C = 64, at most four fc2 outputs, exact-erf GELU, fp32 storage, no GELU inside the input transform
bool rpb_pjh_supported(int C, int DO, int act, const XForm& xf, bool a_bf16) {
    static const bool off = getenv("RPB_HEAD_PJH") && atoi(getenv("RPB_HEAD_PJH")) == 0;
    static const bool bf_off = getenv("RPB_HEAD_PJH_BF16") && atoi(getenv("RPB_HEAD_PJH_BF16")) == 0;
    if (a_bf16 && (bf_off || xf.mean)) return false;     // bf16 storage: plain activations only (the eval cell_mix applied the BatchNorm)
    static const bool c128_off = getenv("RPB_HEAD_PJH_128") && atoi(getenv("RPB_HEAD_PJH_128")) == 0;
    if (C == 128) return !off && !c128_off && !a_bf16 && DO >= 1 && DO <= 4 && (act == 0 || act == 1) && !(xf.mean && xf.gelu);
    return !off && C == 64 && DO >= 1 && DO <= 4 && act == 0 && !(xf.mean && xf.gelu);
}

int rpb_pjh_launch(const float* s, const float* w1, const float* b1, const float* w2, const float* b2, float* out, int B, int DO, int T, int H,
                   int W, int Tp, int Hp, int Wp, const XForm& xf, hipStream_t st, bool f16x2, bool a_bf16, int C, bool silu) {
    PjhFwdArgs p{};
    p.s = s; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.out = out; p.B = B; p.DO = DO;
    p.cm = CropMap{T, H, W, Tp, Hp, Wp};
    p.xf = xf;
    const long GL = (long)B * T * H;
#ifndef PH_WG_PER_CU
#define PH_WG_PER_CU 2L
#endif
    long grid = PH_WG_PER_CU * rpb_num_cus();
    const long need = (GL + PH_WAVES - 1) / PH_WAVES;
    if (grid > need) grid = need;
    if (C == 128) {                                      // one workgroup of four waves per CU (96 KB of fc1 planes)
        if (a_bf16 || f16x2) RPB_FAIL(RPB_ERR_UNSUPPORTED, "proj_fwd (pjh, C = 128): fp32 storage, default arithmetic");
        long g128 = rpb_num_cus();
        if (g128 > need) g128 = need;
        const size_t lds128 = pjh_lds(128);
#define PH_L128(D_, M_, S_)                                                                                                          \
    {                                                                                                                                  \
        (void)hipFuncSetAttribute((const void*)pjh_fwd_kernel<D_, false, false, 128, M_, S_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds128); \
        hipLaunchKernelGGL((pjh_fwd_kernel<D_, false, false, 128, M_, S_>), dim3((int)g128), dim3(PH_WAVES * 64), lds128, st, p);       \
    }
        if (DO <= 2 && !silu) PH_L128(2, 0, false)
        else if (DO <= 2) PH_L128(2, 0, true)
        else if (!silu) PH_L128(4, 0, false)
        else PH_L128(4, 0, true)
        RPB_CHECK_LAUNCH("proj_fwd (pjh, C = 128)");
    }
    const size_t lds = pjh_lds();
    if (a_bf16 && f16x2) RPB_FAIL(RPB_ERR_UNSUPPORTED, "proj_fwd (pjh): f16x2 is an fp32-storage arithmetic");
    if (a_bf16 && DO <= 2) {
        (void)hipFuncSetAttribute((const void*)pjh_fwd_kernel<2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((pjh_fwd_kernel<2, false, true>), dim3((int)grid), dim3(PH_WAVES * 64), lds, st, p);
    } else if (a_bf16) {
        (void)hipFuncSetAttribute((const void*)pjh_fwd_kernel<4, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((pjh_fwd_kernel<4, false, true>), dim3((int)grid), dim3(PH_WAVES * 64), lds, st, p);
    } else if (f16x2 && DO <= 2) {
        (void)hipFuncSetAttribute((const void*)pjh_fwd_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((pjh_fwd_kernel<2, true>), dim3((int)grid), dim3(PH_WAVES * 64), lds, st, p);
    } else if (f16x2) {
        (void)hipFuncSetAttribute((const void*)pjh_fwd_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((pjh_fwd_kernel<4, true>), dim3((int)grid), dim3(PH_WAVES * 64), lds, st, p);
    } else if (DO <= 2) {
        (void)hipFuncSetAttribute((const void*)pjh_fwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(pjh_fwd_kernel<2>, dim3((int)grid), dim3(PH_WAVES * 64), lds, st, p);
    } else {
        (void)hipFuncSetAttribute((const void*)pjh_fwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(pjh_fwd_kernel<4>, dim3((int)grid), dim3(PH_WAVES * 64), lds, st, p);
    }
    RPB_CHECK_LAUNCH("proj_fwd (pjh)");
}

// rpb_proj_bwd at C = 128 (configs/fsi/fno.yaml): gu + the fc2 / bias partial rows on this file's matrix-pipe organisation.  part_rows rows of
// [DO*128 + 128 + DO] floats were allocated by the caller (rpb_proj_slots); rows this launch does not write are zeroed.
int rpb_pjh_bwd128_launch(const float* s, const float* w1, const float* b1, const float* w2, const float* b2, const float* gout, float* gu,
                          float* part, long part_rows, int B, int DO, int T, int H, int W, int Tp, int Hp, int Wp, const XForm& xf, hipStream_t st,
                          bool silu) {
    PjhFwdArgs p{};
    p.s = s; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.out = nullptr; p.gout = gout; p.gu = gu; p.part = part; p.B = B; p.DO = DO;
    p.cm = CropMap{T, H, W, Tp, Hp, Wp};
    p.xf = xf;
    const long GL = (long)B * T * H;
    long grid = rpb_num_cus();
    const long need = (GL + PH_WAVES - 1) / PH_WAVES;
    if (grid > need) grid = need;
    if (grid * PH_WAVES > part_rows) grid = part_rows / PH_WAVES;
    RPB_REQUIRE(grid >= 1, "proj_bwd (pjh, C = 128): no partial rows");
    const long row = (long)DO * PH_HID + PH_HID + DO;
    if (part_rows > grid * PH_WAVES)
        (void)hipMemsetAsync(part + grid * PH_WAVES * row, 0, (size_t)(part_rows - grid * PH_WAVES) * row * 4, st);
    const size_t lds128 = pjh_lds(128);
    const long g128 = grid;
    if (DO <= 2 && !silu) PH_L128(2, 1, false)
    else if (DO <= 2) PH_L128(2, 1, true)
    else if (!silu) PH_L128(4, 1, false)
    else PH_L128(4, 1, true)
    RPB_CHECK_LAUNCH("proj_bwd (pjh, C = 128)");
}

// The fc1 data gradient at C = 128 gathered into the padded layout (rpb_cell_mix(..., gather = 1) without statistics): MODE 2 of the kernel above.
int rpb_pjh_dgrad128_launch(const float* gu, const float* w1, float* g, int B, int T, int H, int W, int Tp, int Hp, int Wp, hipStream_t st,
                            const float* bnb_s, const float* mean, const float* invstd, float* stats_part, long stats_rows) {
    PjhFwdArgs p{};
    p.s = gu; p.w1 = w1; p.out = g; p.B = B; p.DO = 0;
    p.cm = CropMap{T, H, W, Tp, Hp, Wp};
    p.xf = XForm{mean, invstd, nullptr, nullptr, 0};
    p.gout = bnb_s; p.part = stats_part;
    RPB_REQUIRE((long)Wp * 512 < (1l << 31), "proj dgrad (pjh, C = 128): line too long");
    const long GL = (long)B * T * H;
    long grid = rpb_num_cus();
    const long need = (GL + PH_WAVES - 1) / PH_WAVES;
    if (grid > need) grid = need;
    const size_t lds128 = pjh_lds(128);
    if (stats_part) {                                    // MODE 3: + the BatchNorm-backward sums; rows this launch does not write are zeroed
        RPB_REQUIRE(bnb_s && mean && invstd, "fc1 data gradient (pjh, C = 128): the sums need s, mean, invstd");
        if (grid * PH_WAVES > stats_rows) grid = stats_rows / PH_WAVES;
        RPB_REQUIRE(grid >= 1, "fc1 data gradient (pjh, C = 128): no partial rows");
        if (stats_rows > grid * PH_WAVES)
            (void)hipMemsetAsync(stats_part + grid * PH_WAVES * 256, 0, (size_t)(stats_rows - grid * PH_WAVES) * 256 * 4, st);
        (void)hipFuncSetAttribute((const void*)pjh_fwd_kernel<2, false, false, 128, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds128);
        hipLaunchKernelGGL((pjh_fwd_kernel<2, false, false, 128, 3>), dim3((int)grid), dim3(PH_WAVES * 64), lds128, st, p);
        RPB_CHECK_LAUNCH("fc1 data gradient + BN-backward sums (pjh, C = 128)");
    }
    (void)hipFuncSetAttribute((const void*)pjh_fwd_kernel<2, false, false, 128, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds128);
    hipLaunchKernelGGL((pjh_fwd_kernel<2, false, false, 128, 2>), dim3((int)grid), dim3(PH_WAVES * 64), lds128, st, p);
    RPB_CHECK_LAUNCH("fc1 data gradient (pjh, C = 128)");
}
