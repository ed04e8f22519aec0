// cell_mix on the bf16 matrix pipe ("cmx"): the C = 64 instance of rpb_cell_mix (last inverse-DFT stage + 1x1x1 Conv3d + bias +
// add, with the BatchNorm forward / backward sums in the epilogue -- fno.py:63,115-117 and their autograd) with both
// contractions on v_mfma_f32_16x16x32_bf16 from operands split into three bf16 planes (x = hi + mid + lo exactly, truncation
// splits: 8 + 8 + 8 significand bits; a product is accumulated in fp32 from hi*lo + lo*hi + mid*mid + hi*mid + mid*hi + hi*hi,
// each bf16 x bf16 product exact in fp32, the three dropped terms <= 2^-24 |a b| -- the same fp32-grade arithmetic as
// csrc/rpb_conv3x.hip, Rel-L2 vs fp64 ~2e-7).
//
// Why: the fp32 kernel (rpb_cell.hip) keeps the fp32 matrix pipe 66-76 % busy at a clock the chip throttles to ~1.7 GHz and still
// needs 6144 pipe cycles per 32 cells -- above the 5600 cycles the same 32 cells cost at 6.3 TB/s of HBM.  Six bf16 MFMAs replace
// eight fp32 MFMAs of twice the issue time (0.375x the pipe time), which leaves HBM as the only bound.
//
// No activation goes through LDS:
//  * a wave walks whole (b,t,h) lines; a wave tile = 32 consecutive cells of the line = two 16-row MFMA tiles.  Lane (m = lane & 15, kg = lane >> 4) loads 4 x 16 B of cell m:
//    bytes [64 i + 16 kg, +16) of the cell's 256 B channel row for i = 0..3, so every load instruction reads 16 x 64 contiguous
//    bytes and the four together whole 128 B lines; the 8 values of loads (2 ks, 2 ks + 1) ARE the lane's A operand of K-step ks
//    (the contraction index is permuted consistently on the weight side: k = (ks, kg, e) <-> channel 16 (2 ks + e / 4) + 4 kg + e % 4);
//  * the MFMA column index n of output tile t stands for channel 4 n + t, so a lane's accumulators are 4 consecutive channels of
//    4 cells: the store is 16 B per lane, 4 whole 256 B cell rows per instruction (the fp32 kernel: 32 dword stores per tile);
//  * the z2 row of a line (spectral branch, [K2][C] fp32) is loaded in B-operand layout with 16 B loads and split in registers
//    once per line (a tile-granular version fetched every row ~3x from HBM: 10.3 GB of traffic for 8.56 GB algorithmic);
//    GW and the conv weights are split once per workgroup into LDS in operand order.
#include "rpb_cmx.h"
#include <atomic>
#include <mutex>
// (cache policy of the streaming loads / stores: RPB_STREAM_AUX, rpb_common.h -- nt by default since round 5)
#include <stdlib.h>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 ld16(rsrc_t r, int voff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, RPB_STREAM_AUX));
}
// CMX_WG_X_DEFAULT: the weight-gradient pairs' mix wave loads gs with the default policy (its wgrad wave fetches the same lines a tile later)
#ifndef CMX_WG_X_DEFAULT
#define CMX_WG_X_DEFAULT 1
#endif
template <int AUX>
__device__ __forceinline__ u32x4 ld16a(rsrc_t r, int voff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, AUX));
}
__device__ __forceinline__ void st16(f32x4v v, rsrc_t r, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, RPB_STREAM_AUX);
}
// per-tensor policies of the SMALL streams (round 6 sweep, tools/eval_policy_sweep.sh): the z2 rows this launch reads were written by the
// inverse H stage right before it, the Y1 rows the fused W stage writes are read by the next H stage right after it (0.9 GB each at the
// headline shape; the 256 MB MALL can hold the tail of the producer / the head of the consumer)
#ifndef CMX_Z_AUX
#define CMX_Z_AUX RPB_STREAM_AUX
#endif
#ifndef CMX_Y_AUX
#define CMX_Y_AUX RPB_STREAM_AUX
#endif
template <int AUX>
__device__ __forceinline__ void st16a(f32x4v v, rsrc_t r, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, AUX);
}
__device__ __forceinline__ float trunc_bf16(float v) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & 0xffff0000u); }
// (a, b) fp32 -> one dword of two truncated bf16 (a low half, b high half)
__device__ __forceinline__ unsigned pack_hi(float a, float b) {
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
// 8 fp32 -> three bf16x8 planes (exact: hi + mid + lo == v)
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
__device__ __forceinline__ f32x4v mfma16(bf16x8 a, bf16x8 b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// H2 (the opt-in "f16x2" eval arithmetic, CmxArgs::h2): operands as TWO fp16 planes, both rounded to nearest even (x = hi + lo to one fp32
// unit in the last place: 11 + 11 significand bits and the sign of lo), three products hi*lo + lo*hi + hi*hi on v_mfma_f32_16x16x32_f16;
// the dropped lo*lo term is <= 2^-22 |a b| -- the grade of "3xTF32", NOT the 2^-24 grade of the default path.  Half the matrix-pipe time
// and 2.5 instead of 5.5 vector instructions per split value.  The planes travel in the bf16x8 containers of the default path (bit
// patterns only): plane slot 0 = hi, slot 1 = lo.  fp16's range is handled by exact power-of-two scalings, see CmxArgs::spec_exp.
#ifndef RPB_H2_FMAMIX
#define RPB_H2_FMAMIX 1
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
__device__ __forceinline__ f32x4v mfma16h(bf16x8 a, bf16x8 b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// waves per workgroup (= per CU): the z2 planes of a line take 12 KB of LDS per wave: 8 waves + operands = 147 KB
#ifndef CMX_WAVES_A
#define CMX_WAVES_A 8
#endif
#ifndef CMX_WAVES_B
#define CMX_WAVES_B 8
#endif
#define CMX_WAVES_OF(STATS) ((STATS) == 2 ? CMX_WAVES_B : CMX_WAVES_A)
#ifndef CMX_PF2_BF
#define CMX_PF2_BF 0    /* the same for the bf16-storage instances (half the bytes per tile in flight).  Measured (round 5, tools/fwd_probe.py 16 comb_bf16):
                           eval + fused W stage 0.577 -> 0.60 ms, crop-only 0.270 -> 0.278: slower (the static line walk it needs loses more) */
#endif
#ifndef CMX_PF2
#define CMX_PF2 0    /* 1: x tiles requested TWO wave tiles ahead where registers allow.  Measured (round 4, tools/kbench.py, B = 32): no change --
                        forward + stats 1.84-1.87 ms (1.80-1.86 one tile ahead), with the lazy GELU 2.18-2.24 (2.16-2.19), eval 1.83 (1.80-1.83):
                        these launches are not short of bytes in flight; what separates them from the 5.16 TB/s of the plain backward
                        variant (1.66 ms) is the ~70-400 extra vector instructions per tile (statistics, BatchNorm + GELU on load) */
#endif

// STATS: 0 none (bnb.mean != null: output transform) | 1 sum / sum of squares of the output | 2 BatchNorm-backward sums
// BF: activations are STORED as bf16 (x and out: [ncell][64] bf16, 128 B per cell; BASELINE.json configs[4]).  The x operand is
//     then exactly one bf16 plane (no split: 3 products per weight column instead of 6) and a lane's 16 B load is its whole
//     A operand of a K-step (k = (ks, kg, e) <-> channel 32 ks + 8 kg + e); arithmetic and accumulation stay fp32-grade.
// FEAT: layer 0 -- x is the feature tensor Phi_c [ncell][FW] (FW = 8 or 32 floats per cell, rpb_lift_feat) and Wm the COMPOSITE
//       weight Wc0 W0ext [64][FW]: A0 = W0ext Phi_c is never materialised, the channel mixing is one K-step (k = (kg, e) <->
//       field 8 kg + e; lanes with 8 kg >= FW load nothing).
// DFT:  eval only -- the activated line this wave writes is also the input of the next layer's forward W stage (fno.py:48): the
//       tile's outputs sit in the accumulators in exactly that stage's B-operand form (lane group kg holds cells {4 kg + r, 16 + 4 kg + r}
//       of channels 4 n + t), so Y1[line] = FW a accumulates on the matrix pipe from the split outputs and the stage never reads the
//       activations from HBM (one of the three activation passes per layer of the rollout).  The stage matrix per wave tile lives in
//       LDS in A-operand order; the inverse-stage matrix GW moves to a prepared global buffer to keep 8 waves per workgroup.
// WG:   STATS == 2 only -- the layer's Conv3d weight gradient dWc[co][ci] = sum_cells x[cell][co] act(BN(bnb_s))[cell][ci] rides along
//       (x = gs of the layer, bnb_s = the pre-BN tensor whose activation is the layer input; autograd of fno.py:115).  The product
//       contracts over CELLS, so it wants both factors as "8 cells of one channel per lane": act(z) has that form in the epilogue
//       (accumulator layout), gs does not (it is this kernel's A operand, lane = cell).  A wave that did both would need 64 more
//       accumulator registers than the 256 of a two-waves-per-SIMD kernel (the one-wave-per-SIMD version, tools/archive/rpb_cmw.hip, was
//       bound by its own instruction stream: 3.4 ms against 2.4 ms for this launch without the product).  So the workgroup is
//       FOUR PAIRS of waves: "mix" wave p (waves 0-3) is the STATS == 2 kernel and additionally leaves act(z) of its tile -- one erf
//       serves act and act' -- in an 8 KB LDS mailbox; "wgrad" wave p + 4 fetches the same tile's gs in accumulator layout (L2 hits:
//       the mix wave requested those lines a tile earlier), splits both factors and runs the 96 MFMAs into its own 64-register
//       accumulator.  Hand-off per tile through two monotonic LDS counters per pair (produced / consumed); the two waves of a pair
//       share a SIMD's issue slots, so the light wave runs in the stalls of the heavy one.
#define CMX_WAVES_DFT 8
// bf16 activations AND bf16 spectra with the fused W stage (configs[4]): the kernel holds 168 registers and moves half the bytes per tile,
// so one tile of loads in flight per wave leaves the memory system short (3.8 TB/s at 8 waves); its z2 slice is one plane (4 KB per wave
// instead of 12), which makes room for THREE waves per SIMD
#ifndef CMX_WAVES_DFT_SB
#define CMX_WAVES_DFT_SB 12
#endif
#define CMX_WAVES_DFTX(SB_) ((SB_) ? CMX_WAVES_DFT_SB : CMX_WAVES_DFT)
#ifndef CMX_WAVES_SB
#define CMX_WAVES_SB 8           /* the plain / crop-only launch on bf16 activations + spectra (128 registers): 12 waves measured SLOWER (0.285 -> 0.315 ms) */
#endif
#define CMX_WAVES_OFX(STATS_, SB_) ((SB_) ? CMX_WAVES_SB : CMX_WAVES_OF(STATS_))
#define CMX_WG_PAIRS 4
// SB:   with BF -- the SPECTRA are stored as bf16 too: the z2 rows this launch reads (written by rpb_axis_gemm_bf16out) and the Y1 rows
//       the fused W stage writes (read by rpb_axis_gemm_bf16in).  A z2 row is then exactly one bf16 plane: no split, three products.
// C2:   6 or 8 (= waves per workgroup) -- the C = 128 instance (configs/fsi/fno.yaml, the Galerkin regressor): a cell row is 512 B, the
//       channel mixing four K-steps, and a workgroup produces ONE 64-channel half of the output (workgroup b: half b & 1, line slots of
//       b >> 1): the conv-weight planes of one half are 48 KB of LDS, both would not fit next to the z2 planes.  x is read by both
//       halves (the second read mostly out of L2 / MALL: the pair walks the same lines).  The register image of the x tile stays 32
//       registers: K-steps 2, 3 of THIS tile are requested into the slots of K-steps 0, 1 as soon as those are split, the next tile's
//       K-steps 0, 1 into the slots of 2, 3 (half a tile of prefetch distance instead of a whole one).
// CMX_WG_PRIO: s_setprio 1 for one role of the weight-gradient wave pairs (1: the wgrad wave, the second-dispatched half of the
// workgroup; 2: the mix wave; 0: none).  Measured (profiles/r05_kbench_valu_variants.txt) -- see DESIGN.md section 4.0000
#ifndef CMX_WG_PRIO
#define CMX_WG_PRIO 0
#endif
// CMX_WG_GELU_AS: act / act' of the weight-gradient pairs' mix wave from one exponential + one reciprocal (gelu_both_as2x2, rpb_common.h)
// instead of the erf polynomial + two exponentials
#ifndef CMX_WG_ONE_EPILOGUE
#define CMX_WG_ONE_EPILOGUE 0    /* experiment: the wave pairs' mix wave always runs the masked epilogue (a smaller loop body for a few selects per tile) */
#endif
#ifndef CMX_ONE_EPILOGUE
#define CMX_ONE_EPILOGUE 1
#endif
#ifndef CMX_SPLIT_LAST
#define CMX_SPLIT_LAST 1
#endif
#ifndef CMX_SPLIT_LAST_EVAL
#define CMX_SPLIT_LAST_EVAL 0    /* experiment: the same for the fp32-storage eval launches now that they hold one epilogue copy */
#endif
#define CMX_SPLITL(STATS_, BF_) (CMX_SPLIT_LAST && ((STATS_) == 1 || (CMX_SPLIT_LAST_EVAL && (STATS_) == 0 && !(BF_))))
#ifndef CMX_WG_GELU_AS
#define CMX_WG_GELU_AS 1
#endif
// H2:   eval only (STATS == 0 with the output transform), fp32 storage -- the f16x2 arithmetic above for the channel mixing (not with FEAT:
//       the raw feature fields keep the range-safe bf16 planes, their mixing is one K-step), the last inverse stage and the fused W stage.
//       Scalings (exact: powers of two): conv weights and bias x 2^H2W, GW x 2^(spec_exp + H2W), z2 x 2^-spec_exp, and 2^-H2W rides in the
//       output transform's scale.
template <int STATS, bool BF = false, bool FEAT = false, bool DFT = false, bool WG = false, bool SB = false, int C2 = 0, bool H2 = false>
__global__ __launch_bounds__((C2 ? C2 : (WG ? 2 * CMX_WG_PAIRS : (DFT ? CMX_WAVES_DFTX(SB) : CMX_WAVES_OFX(STATS, SB)))) * 64) void cmx_kernel(CmxArgs a) {
    static_assert(!H2 || (STATS == 0 && !BF && !WG && !SB && !C2), "f16x2: the fp32-storage eval launches at C = 64");
    constexpr bool H2X = H2 && !FEAT;                    // channel mixing on fp16 planes
    constexpr int H2W = H2X ? 4 : 0;                     // log2 of the scale the accumulators carry
    static_assert(!C2 || (!BF && !FEAT && !DFT && !WG && !SB), "C = 128: the plain fp32-storage launches");
    static_assert(!SB || BF, "bf16 spectra come with bf16 activation storage");
    static_assert(!WG || (STATS == 2 && !BF && !FEAT && !DFT), "weight-gradient pairs: the fp32 backward launch");
    static_assert(!DFT || STATS == 0, "fused forward W stage: eval path");
    static_assert(!BF || STATS == 0, "bf16 storage: eval / rollout path only");
    static_assert(!(BF && FEAT), "the feature tensor is fp32");
    constexpr int KSN = FEAT ? 1 : (C2 ? 4 : 2);         // K-steps of the channel mixing
    constexpr int CC = C2 ? 128 : 64;                    // channels per cell row of x / out / z2 / bnb_s
    constexpr int CB = CC * 4;                           // bytes per cell row
    constexpr int BWN = (C2 ? 48 : 24) * 64;             // conv-weight planes in LDS (u32x4)
    const int hsel = C2 ? (int)(blockIdx.x & 1) : 0;     // C = 128: which 64-channel half of the output this workgroup produces
    const int bx = C2 ? (int)(blockIdx.x >> 1) : (int)blockIdx.x, gx = C2 ? (int)(gridDim.x >> 1) : (int)gridDim.x;
    const int FW = a.feat_w;
    constexpr int CMX_WAVES = C2 ? C2 : (WG ? CMX_WG_PAIRS : (DFT ? CMX_WAVES_DFTX(SB) : CMX_WAVES_OFX(STATS, SB)));     // line-walking ("mix") waves
    constexpr int ZST = SB ? 4 : 12;                     // u32x4 rows of a wave's z2 slice (bf16 spectra: one plane)
    extern __shared__ u32x4 lds4[];
    const int Wp = a.Wp, K2 = a.K2;
    u32x4* Bw = lds4;                        // [ks 2][plane 3][t 4][lane 64]   conv weights, B-operand order
    // [plane 3][w Wp][kg 4]  last-stage DFT matrix, A-operand rows.  DFT variant: read from a prepared global buffer (26 KB, L1-resident)
    // instead, which is what lets 8 waves fit next to the forward-stage matrix
    u32x4* GWs = DFT ? const_cast<u32x4*>(reinterpret_cast<const u32x4*>(a.gw_planes)) : Bw + BWN;
    u32x4* Zs = Bw + BWN + (DFT ? 0 : 3 * Wp * 4);   // [wave][plane 3][t 4][lane 64]   the current line's z2 row, B-operand order
    float* xfp = reinterpret_cast<float*>(Zs + CMX_WAVES * ZST * 64);   // [2][CC] (+ one unused row)  input transform: invstd*gamma, beta - mean*invstd*gamma
    u32x4* FWs = reinterpret_cast<u32x4*>(xfp + 3 * CC);               // DFT: [tile q][plane 3][mt2 2][lane 64]  forward W-stage matrix, A-operand rows
    u32x4* MBs = FWs;                                                  // WG: [pair][row (j, r) 8][lane 64]  act(z) of the pair's current tile (fp32)
    int* flags = reinterpret_cast<int*>(MBs + CMX_WG_PAIRS * 8 * 64);  // WG: [pair][2]  tiles produced / tiles consumed
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kg = lane >> 4;         // A role: cell row m, k group kg;  B / D role: column n = m, row group mg = kg
    const bool has_xf = a.xf.mean != nullptr;
    const unsigned long long t_wave0 = a.wave_times ? wall_clock64() : 0ull;

    // ---- per-workgroup operand preparation
    for (int idx = tid; idx < KSN * 4 * 64; idx += blockDim.x) {
        const int l = idx & 63, t = (idx >> 6) & 3, ks = idx >> 8;
        const int n = l & 15, kgb = l >> 4;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = (BF || FEAT) ? 32 * ks + 8 * kgb + e : 16 * (2 * ks + (e >> 2)) + 4 * kgb + (e & 3);
            const int co = 4 * n + t;
            if (FEAT) v[e] = ci < FW ? a.Wm[co * FW + ci] : 0.f;
            else v[e] = a.transpose_w ? a.Wm[ci * CC + 64 * hsel + co] : a.Wm[(64 * hsel + co) * CC + ci];
            if (H2X) v[e] *= (float)(1 << H2W);
        }
        bf16x8 h, md, lo;
        if (H2X) {
            split8h(v, h, md);
            lo = md;
        } else {
            split8(v, h, md, lo);
        }
        Bw[((ks * 3 + 0) * 4 + t) * 64 + l] = __builtin_bit_cast(u32x4, h);
        Bw[((ks * 3 + 1) * 4 + t) * 64 + l] = __builtin_bit_cast(u32x4, md);
        Bw[((ks * 3 + 2) * 4 + t) * 64 + l] = __builtin_bit_cast(u32x4, lo);
    }
    for (int idx = tid; idx < (DFT ? 0 : Wp * 4); idx += blockDim.x) {
        const int w = idx >> 2, kgw = idx & 3;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 8 * kgw + e;
            v[e] = k < K2 ? a.GW[k * Wp + w] : 0.f;
            if (H2) v[e] = __builtin_ldexpf(v[e], a.spec_exp + H2W);
        }
        bf16x8 h, md, lo;
        if (H2) {
            split8h(v, h, md);
            lo = md;
        } else {
            split8(v, h, md, lo);
        }
        GWs[(0 * Wp + w) * 4 + kgw] = __builtin_bit_cast(u32x4, h);
        GWs[(1 * Wp + w) * 4 + kgw] = __builtin_bit_cast(u32x4, md);
        GWs[(2 * Wp + w) * 4 + kgw] = __builtin_bit_cast(u32x4, lo);
    }
    if (DFT) {
        const int TQ0 = (Wp + 31) >> 5;
        for (int idx = tid; idx < TQ0 * 2 * 64; idx += blockDim.x) {
            const int l = idx & 63, mt2 = (idx >> 6) & 1, q = idx >> 7;
            const int k = 16 * mt2 + (l & 15);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int w = 32 * q + 16 * (e >> 2) + 4 * (l >> 4) + (e & 3);
                v[e] = (k < a.K2f && w < Wp) ? a.FWt[w * a.K2f + k] : 0.f;        // cells past the line end contribute nothing
            }
            bf16x8 h, md, lo;
            if (H2) {
                split8h(v, h, md);
                lo = md;
            } else {
                split8(v, h, md, lo);
            }
            FWs[((q * 3 + 0) * 2 + mt2) * 64 + l] = __builtin_bit_cast(u32x4, h);
            FWs[((q * 3 + 1) * 2 + mt2) * 64 + l] = __builtin_bit_cast(u32x4, md);
            FWs[((q * 3 + 2) * 2 + mt2) * 64 + l] = __builtin_bit_cast(u32x4, lo);
        }
    }
    if (WG) {           // stale rows (the skipped half of a line's last tile) must be finite: they meet gs = 0
        for (int idx = tid; idx < CMX_WG_PAIRS * 8 * 64; idx += blockDim.x) MBs[idx] = u32x4{0u, 0u, 0u, 0u};
        if (tid < 2 * CMX_WG_PAIRS) flags[tid] = 0;
    }
    if (has_xf && tid < CC) {
        // BatchNorm as ONE fused multiply-add per element: z = x * sc + sh,  sc = invstd * gamma,  sh = beta - mean * sc
        const float sc_ = a.xf.invstd[tid] * a.xf.gamma[tid];
        xfp[tid] = sc_;
        xfp[CC + tid] = a.xf.beta[tid] - a.xf.mean[tid] * sc_;
    }
    // DYN: the workgroup's lines (b * waves + i % waves + (i / waves) * nslots, i = 0, 1, ...: the same set as the static walk) are CLAIMED
    // by its waves from a counter in LDS instead of dealt round-robin.  Measured with rpb_cmx_debug_wave_times at the headline shape
    // (tools/wave_times.py): under the static deal the waves of one workgroup finish up to 15 % apart (they share a SIMD's issue slots
    // and the CU's memory path unevenly) and the mean wave lifetime is 0.87-0.92 of the launch.
    //      Mode 2 claims from ONE counter in HBM (device-scope atomic, a line ahead of its use): that also evens out the workgroups
    //      (those on odd XCDs run 3-8 % slower than those on even ones).
    __shared__ int claim_s;
    const int DYN = (WG || (CMX_PF2 && !DFT && !BF && STATS != 2) || (CMX_PF2_BF && BF)) ? 0 : ((C2 && a.claim_mode == 2) ? 1 : a.claim_mode);   // C2: both halves walk the workgroup pair's own lines
    if (DYN == 1 && tid == 0) claim_s = CMX_WAVES;
    __syncthreads();

    if constexpr (WG) {
        if (wave >= CMX_WG_PAIRS) {
            // ================= "wgrad" wave of pair p = wave - 4: walks the same tiles as mix wave p
#if CMX_WG_PRIO == 1
            __builtin_amdgcn_s_setprio(1);               // static priority for the second-dispatched half of the workgroup (A/B: see CMX_WG_PRIO)
#endif
            const int pair = wave - CMX_WG_PAIRS;
            const int G = __builtin_amdgcn_readfirstlane((int)((unsigned)a.ncell / (unsigned)Wp));
            const int TQ = (Wp + 31) >> 5;
            const int nslots = (int)gridDim.x * CMX_WG_PAIRS;
            const int slot = (int)blockIdx.x * CMX_WG_PAIRS + pair;
            const unsigned line_bytes = (unsigned)Wp * 256u;
            const int ooff = (4 * kg) * 256 + m * 16;        // accumulator layout: cell 4 mg + r of MFMA tile j, channels 4 n .. 4 n + 3
            const f32x4v z4 = {0.f, 0.f, 0.f, 0.f};
            f32x4v accW[4][4];                               // tile (uo, ui): row 4 mg + r <-> out channel 4 (4 mg + r) + uo, column n <-> in channel 4 n + ui
#pragma unroll
            for (int uo = 0; uo < 4; ++uo)
#pragma unroll
                for (int ui = 0; ui < 4; ++ui) accW[uo][ui] = z4;
            u32x4 gb[2][4];
            auto issue_g = [&](int g_, int q, bool ok) {     // !ok: empty descriptor, zeros without traffic
                const unsigned long long pa = (unsigned long long)(a.x + (long)g_ * Wp * 64);
                const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pa), hi = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));
                const rsrc_t rx = make_rsrc(reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo),
                                            __builtin_amdgcn_readfirstlane(ok ? line_bytes : 0u));
                const int qo = __builtin_amdgcn_readfirstlane(q) * 8192;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) gb[j][r] = ld16(rx, qo + ooff + j * 4096 + r * 256);
            };
            int* fl = flags + 2 * pair;
            const u32x4* mb = MBs + pair * 8 * 64 + lane;
            int g = slot, q = 0, k = 0;
            if (g < G) issue_g(g, 0, true);
            while (g < G) {
                int gn = g, qn = q + 1;
                if (qn == TQ) {
                    qn = 0;
                    gn = g + nslots;
                }
                // gs of this tile: K = (mg, e = 4 j + r) <-> cell 16 j + 4 mg + r; row n of tile uo <-> channel 4 n + uo
                bf16x8 Gh[4], Gm[4], Gl[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[4 * j + r] = __builtin_bit_cast(f32x4v, gb[j][r])[u];
                    split8(v, Gh[u], Gm[u], Gl[u]);
                }
                issue_g(gn, qn, gn < G);                     // next tile's gs: in flight while this wave waits for / multiplies act(z)
                // (readfirstlane: a branch on a loaded value is divergent to the compiler, and everything downstream of a divergent
                //  loop exit -- line indices, buffer descriptors -- would then be treated as per-lane values)
                while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(fl, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < k + 1)
                    __builtin_amdgcn_s_sleep(4);
                u32x4 av[2][4];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) av[j][r] = mb[(4 * j + r) * 64];
                bf16x8 Xh[4], Xm[4], Xl[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[4 * j + r] = __builtin_bit_cast(f32x4v, av[j][r])[u];
                    split8(v, Xh[u], Xm[u], Xl[u]);
                }
                ++k;
                if (lane == 0) __hip_atomic_store(fl + 1, k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);     // mailbox read: free again
#pragma unroll
                for (int uo = 0; uo < 4; ++uo) {
#define CMX_W(AP, BP) _Pragma("unroll") for (int ui = 0; ui < 4; ++ui) accW[uo][ui] = mfma16(AP[uo], BP[ui], accW[uo][ui]);
                    CMX_W(Gh, Xl) CMX_W(Gl, Xh) CMX_W(Gm, Xm) CMX_W(Gh, Xm) CMX_W(Gm, Xh) CMX_W(Gh, Xh)
#undef CMX_W
                }
                g = gn;
                q = qn;
            }
            float* wp = a.wg_part + (long)slot * (64 * 64);
#pragma unroll
            for (int uo = 0; uo < 4; ++uo)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = 4 * (4 * kg + r) + uo;
                    *reinterpret_cast<f32x4v*>(wp + o * 64 + 4 * m) = f32x4v{accW[uo][0][r], accW[uo][1][r], accW[uo][2][r], accW[uo][3][r]};
                }
            return;
        }
    }
#if CMX_WG_PRIO == 2
    if (WG) __builtin_amdgcn_s_setprio(1);
#endif
    int wg_tiles = 0;                                // WG, mix wave: tiles handed to the pair's wgrad wave so far
    u32x4* MBw = MBs + (WG ? wave : 0) * 8 * 64 + lane;
    int* wg_fl = flags + 2 * (WG ? wave : 0);

    // ---- per-lane output-channel constants (channels 4 n .. 4 n + 3)
    const bool oxf = STATS == 0 && a.bnb.mean != nullptr;
    float bv[4];
    XParam bp[4];
    f32x2 ssum[2], ssq[2];                           // channel pairs (4 n, 4 n + 1), (4 n + 2, 4 n + 3)
    ssum[0] = ssum[1] = ssq[0] = ssq[1] = pk2(0.f);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        bv[t] = a.bias ? a.bias[64 * hsel + 4 * m + t] : 0.f;
        if (STATS == 2 || oxf) bp[t] = xf_load(a.bnb, 64 * hsel + 4 * m + t);
        if (STATS == 0 && oxf) {            // eval output transform as one fused multiply-add: .is <- invstd * gamma, .be <- beta - mean * that
            bp[t].is *= bp[t].ga;
            bp[t].be -= bp[t].mu * bp[t].is;
        }
        if (H2W) {                          // the accumulators carry 2^H2W (scaled weights): the bias joins them, the transform's scale undoes it
            bv[t] *= (float)(1 << H2W);
            bp[t].is *= 1.0f / (float)(1 << H2W);
        }
    }
    const bool xgelu = a.xf.gelu != 0;
    const bool bgelu = a.bnb.gelu != 0;

    // ---- work item = one (b,t,h) line of Wp cells, walked in TQ = ceil(Wp / 32) wave tiles by ONE wave: the line's z2 row is
    //      fetched from HBM exactly once chip-wide and split once; tiles never straddle lines (the last tile of a line is partial:
    //      its out-of-range loads return 0 and its stores are dropped by the line's buffer descriptor)
    // crop_T > 0 (eval, last layer: only the un-padded cells are read by the projection head): the work list is the B * crop_T * crop_H
    // lines of the crop and a line ends at the tile holding cell crop_W - 1
    const bool crop = a.crop_T > 0;
    const long G = crop ? (a.ncell / ((long)Wp * a.Hp * a.Tp)) * a.crop_T * a.crop_H : a.ncell / Wp;
    const int TQ = ((crop ? a.crop_W : Wp) + 31) >> 5;
    auto line_of = [&](long gi) -> long {
        if (!crop) return gi;
        const unsigned u = (unsigned)gi, th = (unsigned)(a.crop_T * a.crop_H);
        const unsigned b = u / th, r = u - b * th, t = r / (unsigned)a.crop_H, h = r - t * (unsigned)a.crop_H;
        return ((long)b * a.Tp + t) * a.Hp + h;
    };
    const long nslots = (long)gx * CMX_WAVES;
    const long slot = (long)bx * CMX_WAVES + wave;
    const unsigned line_bytes = (unsigned)Wp * (BF ? 128u : (unsigned)CB);
    const long line_floats = (long)Wp * (BF ? 32 : CC);          // bf16 storage: two channels per float slot
    const unsigned xline_bytes = FEAT ? (unsigned)Wp * FW * 4u : line_bytes;
    const long xline_floats = FEAT ? (long)Wp * FW : line_floats;
    const int xoff = m * CB + kg * 16;                   // byte offset of the lane's first 16 B inside a 16-cell block
    const int ooff = (4 * kg) * CB + m * 16;             // output: cell 4 mg + r, channels 4 n ..

    // PF2 (compile-time experiment, off: see CMX_PF2): the x tiles requested TWO wave tiles ahead (two register images, the tile loop
    // unrolled by two) -- 128 instead of 64 KB of loads in flight per CU.  Not for STATS == 2 / the fused W stage (no registers left).
    constexpr bool PF2 = (CMX_PF2 && !DFT && !BF && STATS != 2 && !C2) || (CMX_PF2_BF && BF);
    u32x4 xaA[2][4], xaB[2][4];
    // loads i = 2 ks, 2 ks + 1 of MFMA tile j (its A operand of K-step ks) of wave tile q of line g
    auto issue_x = [&](u32x4 (&xa)[2][4], long g, int q, int j, int ks) {
        const rsrc_t rx = make_rsrc(a.x + g * xline_floats, xline_bytes);
        if (FEAT) {                 // 32 B of fields 8 kg .. 8 kg + 7 (lane groups past FW: an offset outside the descriptor -> 0)
            if (ks == 0) {
                const int off = 8 * kg < FW ? (32 * q + 16 * j + m) * FW * 4 + kg * 32 : 0x7ffffff0;
                xa[j][0] = ld16(rx, off);
                xa[j][1] = ld16(rx, off + 16);
            }
        } else if (BF) {
            xa[j][ks] = ld16(rx, q * 4096 + j * 2048 + m * 128 + ks * 64 + kg * 16);
        } else {
#pragma unroll
            // (C = 128: the two workgroups of a pair read the same x lines, the second out of L2 / MALL -- default policy there; measured:
            //  fsi step 43.0 -> 44.5 ms with nontemporal x loads)
            for (int hf = 0; hf < 2; ++hf)
                xa[j][2 * (C2 ? (ks & 1) : ks) + hf] = ld16a<(C2 || (WG && CMX_WG_X_DEFAULT)) ? 0 : RPB_STREAM_AUX>(rx, q * (32 * CB) + xoff + j * (16 * CB) + (2 * ks + hf) * 64);
        }
    };
    u32x4 zr[8];
    auto issue_z = [&](long g) {        // z2 row in B-operand layout: lane (n, kg) holds k = 8 kg + e, channels 4 n .. 4 n + 3
        if (SB) {                       // bf16 rows of 128 B: 8 B = channels 4 n .. 4 n + 3 per lane and k
            typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
            const rsrc_t rz = make_rsrc(a.z2 + g * K2 * 32, (unsigned)K2 * 128u);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const u32x2v w = __builtin_amdgcn_raw_buffer_load_b64(rz, (8 * kg + e) * 128 + m * 8, 0, 0);
                zr[e][0] = w[0];
                zr[e][1] = w[1];
            }
            return;
        }
        const rsrc_t rz = make_rsrc(a.z2 + g * K2 * CC + 64 * hsel, (unsigned)K2 * (unsigned)CB - 256u * (unsigned)hsel);
#pragma unroll
        for (int e = 0; e < 8; ++e) zr[e] = ld16a<CMX_Z_AUX>(rz, (8 * kg + e) * CB + m * 16);
    };

    u32x4* Zw = Zs + wave * ZST * 64 + lane;
    f32x4v Yacc[DFT ? 2 : 1][DFT ? 4 : 1];
    // one wave tile: (gi, q) = the tile computed from the register image `xa`; (ngi, nq) = the tile whose loads take the image's place
    // (the next tile, or with PF2 the one after it; ngi >= G: none)
    // WG: pin the (wave-uniform) line indices to SGPRs -- with the second wave role in the kernel the compiler keeps them in VGPRs and
    // wraps every buffer access in a waterfall loop
    auto U = [&](long v) -> long { return WG ? (long)__builtin_amdgcn_readfirstlane((int)v) : v; };
    // CMX_SPLIT_LAST: the tile body exists twice -- for a line's last tile (the only one that can be a half tile or carry masked cells, and the
    // one that requests the next line's z2 row) and for all the others, where `last` / `half_tile` are compile-time false: without the
    // specialisation every group of four MFMAs ends in a branch on `half_tile` (36 basic blocks per tile that the scheduler cannot cross).
    // Measured per variant (B = 32, two boxes' worth of A/B in profiles/r06b_ab2_split_last.txt): the training forward with the lazy
    // BatchNorm + GELU 2.06-2.13 -> 1.96 ms, layer 0 1.013 -> 0.99, the plain forward / backward -0.5 .. -2 %; but the larger bodies LOSE --
    // eval with the fused W stage 2.31 -> 2.56 ms, the wave-pair backward 2.88 -> 4.93 ms (the doubled body no longer sits in the
    // instruction cache).  So: the STATS == 1 instances only.
    auto do_tile = [&](auto last_tag, u32x4 (&xa)[2][4], long gi, int q, long ngi, int nq) {
        constexpr bool LASTC = decltype(last_tag)::value;
        constexpr bool SPLITL = CMX_SPLITL(STATS, BF);              // measured per variant (profiles/r06b_ab2_split_last.txt), see CMX_SPLIT_LAST
        {
            const long g = U(line_of(gi));
            const rsrc_t ro = make_rsrc(a.out + g * line_floats + 64 * hsel, line_bytes - 256u * (unsigned)hsel);
            const rsrc_t rs = make_rsrc(STATS == 2 ? a.bnb_s + g * Wp * CC + 64 * hsel : a.out, line_bytes - 256u * (unsigned)hsel);
            const bool last = SPLITL ? LASTC : (q + 1 == TQ);
            const bool more = ngi < G;
            const long gn = U(more ? line_of(ngi) : 0);                      // the tile to request: (gn, qn)
            const int qn = nq;
            const long gnl = DYN ? ngi : gi + nslots;                        // the wave's next line (DYN: meaningful on the line's last tile)
            const bool more_lines = gnl < G;
            const bool half_tile = (SPLITL && !LASTC) ? false : (32 * q + 16 >= Wp);   // uniform: the second MFMA tile lies past the line end
            asm volatile("" ::: "memory");   // keep the (tile-invariant) LDS operand reads inside the loop: hoisted, they cost 150 VGPRs

            f32x4v acc[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[j][t] = f32x4v{bv[t], bv[t], bv[t], bv[t]};     // the bias rides in the accumulator
            u32x4 spre[2][4];
            // ---- channel mixing: K = 64 = 2 steps of 32.  Per step: x registers -> A planes (lazy BN+GELU of the producer
            //      applied here), then the freed registers take the next wave tile's loads, in flight during everything below
#pragma unroll
            for (int ks = 0; ks < KSN; ++ks) {
                bf16x8 Ah[2], Am[2], Al[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (j == 1 && half_tile) continue;
                    if (BF) {                       // the 16 B load IS the operand
                        Ah[j] = __builtin_bit_cast(bf16x8, xa[j][ks]);
                        continue;
                    }
                    if (FEAT) {
                        float v[8];
#pragma unroll
                        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                            for (int c = 0; c < 4; ++c) v[4 * hf + c] = __builtin_bit_cast(f32x4v, xa[j][hf])[c];
                        split8(v, Ah[j], Am[j], Al[j]);
                        continue;
                    }
                    float v[8];
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int i = 2 * ks + hf;
                        const f32x4v xv = __builtin_bit_cast(f32x4v, xa[j][C2 ? 2 * (ks & 1) + hf : i]);
                        if (has_xf) {
                            const f32x4v sc = *reinterpret_cast<const f32x4v*>(xfp + 16 * i + 4 * kg);
                            const f32x4v sh = *reinterpret_cast<const f32x4v*>(xfp + CC + 16 * i + 4 * kg);
                            // channel pairs: packed fp32 math, the two pairs' erf polynomials in lock-step (rpb_common.h, gelu2x2)
                            f32x2 z0 = pk_fma(f32x2{xv[0], xv[1]}, f32x2{sc[0], sc[1]}, f32x2{sh[0], sh[1]});
                            f32x2 z1 = pk_fma(f32x2{xv[2], xv[3]}, f32x2{sc[2], sc[3]}, f32x2{sh[2], sh[3]});
                            if (xgelu) gelu2x2(z0, z1);
                            v[4 * hf] = z0[0];
                            v[4 * hf + 1] = z0[1];
                            v[4 * hf + 2] = z1[0];
                            v[4 * hf + 3] = z1[1];
                        } else {
#pragma unroll
                            for (int c = 0; c < 4; ++c) v[4 * hf + c] = xv[c];
                        }
                    }
                    if (H2X) split8h(v, Ah[j], Am[j]);
                    else split8(v, Ah[j], Am[j], Al[j]);
                }
                if (C2 && ks < 2) {                 // C = 128: this tile's K-steps 2, 3 take the slots of 0, 1
                    issue_x(xa, g, q, 0, ks + 2);
                    issue_x(xa, g, q, 1, ks + 2);
                } else if (more) {
                    issue_x(xa, gn, qn, 0, C2 ? ks - 2 : ks);
                    issue_x(xa, gn, qn, 1, C2 ? ks - 2 : ks);
                }
                if (ks == 0) {
                    if (q == 0) {                  // new line: its z2 row (requested one tile ago) -> three bf16 planes per channel,
#pragma unroll                                     // parked in the wave's own LDS slice (48 registers otherwise)
                        for (int t = 0; t < 4; ++t) {
                            if (SB) {           // channel 4 n + t of rows (2 q, 2 q + 1): halves of word t >> 1 -> one exact plane
                                u32x4 u;
#pragma unroll
                                for (int qq = 0; qq < 4; ++qq)
                                    u[qq] = __builtin_amdgcn_perm(zr[2 * qq + 1][t >> 1], zr[2 * qq][t >> 1], (t & 1) ? 0x07060302u : 0x05040100u);
                                Zw[(0 * 4 + t) * 64] = u;
                                continue;
                            }
                            float v[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = __builtin_bit_cast(f32x4v, zr[e])[t];
                            bf16x8 zh, zm, zl;
                            if (H2) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] = __builtin_ldexpf(v[e], -a.spec_exp);
                                split8h(v, zh, zm);
                                Zw[(0 * 4 + t) * 64] = __builtin_bit_cast(u32x4, zh);
                                Zw[(1 * 4 + t) * 64] = __builtin_bit_cast(u32x4, zm);
                                continue;
                            }
                            split8(v, zh, zm, zl);
                            Zw[(0 * 4 + t) * 64] = __builtin_bit_cast(u32x4, zh);
                            Zw[(1 * 4 + t) * 64] = __builtin_bit_cast(u32x4, zm);
                            Zw[(2 * 4 + t) * 64] = __builtin_bit_cast(u32x4, zl);
                        }
                    }
                    if (last && more_lines) issue_z(U(line_of(gnl)));   // next line's row: in flight for a whole tile
                    if (STATS == 2) {              // pre-BN values at the output positions (needed by the epilogue)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) spre[j][r] = ld16(rs, q * (32 * CB) + ooff + j * (16 * CB) + r * CB);
                    }
                }
#pragma unroll
                for (int tp = 0; tp < 2; ++tp) {
                    bf16x8 Bh[2], Bm[2], Bl[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int t = 2 * tp + u;
                        Bh[u] = __builtin_bit_cast(bf16x8, Bw[((ks * 3 + 0) * 4 + t) * 64 + lane]);
                        Bm[u] = __builtin_bit_cast(bf16x8, Bw[((ks * 3 + 1) * 4 + t) * 64 + lane]);
                        if (!H2X) Bl[u] = __builtin_bit_cast(bf16x8, Bw[((ks * 3 + 2) * 4 + t) * 64 + lane]);
                    }
#define CMX_PRODH(AP, BP)                                                    \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) if (j == 0 || !half_tile)  \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) acc[j][2 * tp + u] = mfma16h(AP[j], BP[u], acc[j][2 * tp + u]);
#define CMX_PROD(AP, BP)                                                     \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) if (j == 0 || !half_tile)  \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) acc[j][2 * tp + u] = mfma16(AP[j], BP[u], acc[j][2 * tp + u]);
                    if (BF) {
                        if (RPB_BF16_CONST_PLANES > 2) { CMX_PROD(Ah, Bl) }
                        CMX_PROD(Ah, Bm) CMX_PROD(Ah, Bh)
                    } else if (H2X) {           // slot 1 (the "m" registers) holds the lo plane
                        CMX_PRODH(Ah, Bm) CMX_PRODH(Am, Bh) CMX_PRODH(Ah, Bh)
                    } else {
                        CMX_PROD(Ah, Bl) CMX_PROD(Al, Bh) CMX_PROD(Am, Bm) CMX_PROD(Ah, Bm) CMX_PROD(Am, Bh) CMX_PROD(Ah, Bh)
                    }
#undef CMX_PROD
#undef CMX_PRODH
                }
            }
            // ---- last inverse-DFT stage: A = GW row of the cell's w (LDS; rows past the line end are clamped: their results are
            //      never stored), B = the line's z2 planes (the wave's LDS slice)
            {
                bf16x8 ah[2], am[2], al[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    int wl = 32 * q + 16 * j + m;
                    wl = wl < Wp ? wl : Wp - 1;
                    ah[j] = __builtin_bit_cast(bf16x8, GWs[(0 * Wp + wl) * 4 + kg]);
                    am[j] = __builtin_bit_cast(bf16x8, GWs[(1 * Wp + wl) * 4 + kg]);
                    if (!H2) al[j] = __builtin_bit_cast(bf16x8, GWs[(2 * Wp + wl) * 4 + kg]);
                }
#pragma unroll
                for (int tp = 0; tp < 2; ++tp) {
                    bf16x8 Zh[2], Zm[2], Zl[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        Zh[u] = __builtin_bit_cast(bf16x8, Zw[(0 * 4 + 2 * tp + u) * 64]);
                        if (!SB) {
                            Zm[u] = __builtin_bit_cast(bf16x8, Zw[(1 * 4 + 2 * tp + u) * 64]);
                            if (!H2) Zl[u] = __builtin_bit_cast(bf16x8, Zw[(2 * 4 + 2 * tp + u) * 64]);
                        }
                    }
#define CMX_SPECH(AP, ZP)                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) if (j == 0 || !half_tile)                                 \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) acc[j][2 * tp + u] = mfma16h(AP[j], ZP[u], acc[j][2 * tp + u]);
#define CMX_SPEC(AP, ZP)                                                                                    \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) if (j == 0 || !half_tile)                                 \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) acc[j][2 * tp + u] = mfma16(AP[j], ZP[u], acc[j][2 * tp + u]);
                    if (SB) {
                        if (RPB_BF16_CONST_PLANES > 2) { CMX_SPEC(al, Zh) }
                        CMX_SPEC(am, Zh) CMX_SPEC(ah, Zh)
                    } else if (H2) {
                        CMX_SPECH(ah, Zm) CMX_SPECH(am, Zh) CMX_SPECH(ah, Zh)
                    } else {
                        CMX_SPEC(ah, Zl) CMX_SPEC(al, Zh) CMX_SPEC(am, Zm) CMX_SPEC(ah, Zm) CMX_SPEC(am, Zh) CMX_SPEC(ah, Zh)
                    }
#undef CMX_SPEC
#undef CMX_SPECH
                }
            }
            // ---- epilogue: cell 32 q + 16 j + 4 mg + r of the line, channels 4 n + t: one 16 B store per (j, r).  Only the last
            //      tile of a line has cells past the line end (their stores are dropped by the descriptor; the sums skip them)
            auto epilogue = [&](auto masked_tag) {
                constexpr bool MASKED = decltype(masked_tag)::value;
                if (WG) {       // the pair's wgrad wave must have read the previous tile's act(z) out of the mailbox
                    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(wg_fl + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < wg_tiles)
                        __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (j == 1 && half_tile) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool valid = !MASKED || (32 * q + 16 * j + 4 * kg + r < Wp);
                        f32x4v o, avr;
                        // channel pairs (packed fp32 math); the erf polynomials of the two pairs run in lock-step (gelu2x2 / gelu_both2x2)
                        f32x2 vv[2], shv[2], gpv[2], acv[2];
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) {
                            const int t = 2 * tt;
                            vv[tt] = f32x2{acc[j][t][r], acc[j][t + 1][r]};
                            if (STATS == 0 && oxf) vv[tt] = pk_fma(vv[tt], f32x2{bp[t].is, bp[t + 1].is}, f32x2{bp[t].be, bp[t + 1].be});
                            if (STATS == 2) {
                                const f32x4v sp = __builtin_bit_cast(f32x4v, spre[j][r]);
                                shv[tt] = (f32x2{sp[t], sp[t + 1]} - f32x2{bp[t].mu, bp[t + 1].mu}) * f32x2{bp[t].is, bp[t + 1].is};
                                acv[tt] = pk_fma(shv[tt], f32x2{bp[t].ga, bp[t + 1].ga}, f32x2{bp[t].be, bp[t + 1].be});      // z
                                gpv[tt] = pk2(1.f);
                            }
                        }
                        if (STATS == 0 && oxf && bgelu) gelu2x2(vv[0], vv[1]);
                        if (STATS == 2 && bgelu) {
                            if (WG) {           // act(z) for the weight gradient: the same erf serves act and act'
#if CMX_WG_GELU_AS
                                gelu_both_as2x2(acv[0], acv[1], acv[0], acv[1], gpv[0], gpv[1]);
#else
                                gelu_both2x2(acv[0], acv[1], acv[0], acv[1], gpv[0], gpv[1]);
#endif
                            } else {
                                gpv[0] = gelu_grad2(acv[0]);
                                gpv[1] = gelu_grad2(acv[1]);
                            }
                        }
#pragma unroll
                        for (int t = 0; t < 4; t += 2) {
                            f32x2 v = vv[t >> 1];
                            if (STATS == 1) {
                                const f32x2 vm = valid ? v : pk2(0.f);
                                ssum[t >> 1] += vm;
                                ssq[t >> 1] = pk_fma(vm, vm, ssq[t >> 1]);
                            } else if (STATS == 2) {
                                const f32x2 sh = shv[t >> 1];
                                f32x2 gz = bgelu ? v * gpv[t >> 1] : v;
                                if (WG) {
                                    avr[t] = acv[t >> 1][0];
                                    avr[t + 1] = acv[t >> 1][1];
                                }
                                if (a.write_gz) v = gz;
                                gz = valid ? gz : pk2(0.f);
                                ssum[t >> 1] += gz;
                                ssq[t >> 1] = pk_fma(gz, sh, ssq[t >> 1]);
                            }
                            o[t] = v[0];
                            o[t + 1] = v[1];
                            if (DFT && !BF) {                        // keep the stored value: operand of the fused W stage below
                                acc[j][t][r] = v[0];
                                acc[j][t + 1][r] = v[1];
                            }
                        }
                        if (BF) {               // round to nearest even, 4 channels = 8 B per lane, 128 B per cell row
                            typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
                            typedef float f32x2v __attribute__((ext_vector_type(2)));
                            typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
                            u32x2v pk;
                            pk[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{o[0], o[1]}, bf16x2v));
                            pk[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{o[2], o[3]}, bf16x2v));
                            __builtin_amdgcn_raw_buffer_store_b64(pk, ro, q * 4096 + j * 2048 + (4 * kg + r) * 128 + m * 8, 0, 0);
                            if (DFT) {          // the fused W stage sees what the next layer would read back: the ROUNDED values (one exact bf16 plane)
#pragma unroll
                                for (int t = 0; t < 4; ++t)
                                    acc[j][t][r] = __builtin_bit_cast(float, (pk[t >> 1] >> (16 * (t & 1))) << 16);
                            }
                        } else {
                            st16(o, ro, q * (32 * CB) + ooff + j * (16 * CB) + r * CB);
                        }
                        if (WG) MBw[(4 * j + r) * 64] = __builtin_bit_cast(u32x4, avr);
                    }
                }
                if (WG) {
                    ++wg_tiles;
                    if (lane == 0) __hip_atomic_store(wg_fl, wg_tiles, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            };
            if ((STATS == 0 && CMX_ONE_EPILOGUE) || (WG && CMX_WG_ONE_EPILOGUE)) epilogue(std::true_type{});   // STATS == 0: no sums to mask, one copy of the code instead of two identical ones
            else if (last) epilogue(std::true_type{});
            else epilogue(std::false_type{});
            if (DFT) {
                if (q == 0) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int t = 0; t < 4; ++t) Yacc[i][t] = f32x4v{0.f, 0.f, 0.f, 0.f};
                }
                bf16x8 fh[2], fm[2], fl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fh[i] = __builtin_bit_cast(bf16x8, FWs[((q * 3 + 0) * 2 + i) * 64 + lane]);
                    fm[i] = __builtin_bit_cast(bf16x8, FWs[((q * 3 + 1) * 2 + i) * 64 + lane]);
                    if (!H2) fl[i] = __builtin_bit_cast(bf16x8, FWs[((q * 3 + 2) * 2 + i) * 64 + lane]);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = acc[0][t][r];
                        v[4 + r] = half_tile ? 0.f : acc[1][t][r];            // the second MFMA tile was not computed
                    }
                    if (BF) {                   // exactly one plane: three products per stage-matrix plane, no split
                        const bf16x8 yb = __builtin_bit_cast(bf16x8, u32x4{pack_hi(v[0], v[1]), pack_hi(v[2], v[3]), pack_hi(v[4], v[5]), pack_hi(v[6], v[7])});
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            if (RPB_BF16_CONST_PLANES > 2) Yacc[i][t] = mfma16(fl[i], yb, Yacc[i][t]);
                            Yacc[i][t] = mfma16(fm[i], yb, Yacc[i][t]);
                            Yacc[i][t] = mfma16(fh[i], yb, Yacc[i][t]);
                        }
                        continue;
                    }
                    bf16x8 yh, ym, yl;
                    if (H2) {
                        split8h(v, yh, ym);
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            Yacc[i][t] = mfma16h(fh[i], ym, Yacc[i][t]);
                            Yacc[i][t] = mfma16h(fm[i], yh, Yacc[i][t]);
                            Yacc[i][t] = mfma16h(fh[i], yh, Yacc[i][t]);
                        }
                        continue;
                    }
                    split8(v, yh, ym, yl);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        Yacc[i][t] = mfma16(fh[i], yl, Yacc[i][t]);
                        Yacc[i][t] = mfma16(fl[i], yh, Yacc[i][t]);
                        Yacc[i][t] = mfma16(fm[i], ym, Yacc[i][t]);
                        Yacc[i][t] = mfma16(fh[i], ym, Yacc[i][t]);
                        Yacc[i][t] = mfma16(fm[i], yh, Yacc[i][t]);
                        Yacc[i][t] = mfma16(fh[i], yh, Yacc[i][t]);
                    }
                }
                if (last) {                     // Y1[g][k = 16 i + 4 mg + r][channels 4 n ..]: 16 B per lane and row
                    if (SB) {                   // the same rows as bf16 (round to nearest even): 8 B per lane, 128 B per row
                        typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
                        typedef float f32x2v __attribute__((ext_vector_type(2)));
                        typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
                        const rsrc_t ry = make_rsrc(a.y1out + g * a.K2f * 32, (unsigned)a.K2f * 128u);
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                u32x2v pk;
                                pk[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{Yacc[i][0][r], Yacc[i][1][r]}, bf16x2v));
                                pk[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{Yacc[i][2][r], Yacc[i][3][r]}, bf16x2v));
                                __builtin_amdgcn_raw_buffer_store_b64(pk, ry, (16 * i + 4 * kg + r) * 128 + m * 8, 0, 0);
                            }
                    } else {
                    const rsrc_t ry = make_rsrc(a.y1out + g * a.K2f * 64, (unsigned)a.K2f * 256u);   // rows >= K2f: dropped
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            st16a<CMX_Y_AUX>(f32x4v{Yacc[i][0][r], Yacc[i][1][r], Yacc[i][2][r], Yacc[i][3][r]}, ry, (16 * i + 4 * kg + r) * 256 + m * 16);
                    }
                }
            }
        }
    };
#define CMX_DO_TILE(XA, G_, Q_, NG_, NQ_)                                                                      \
    do {                                                                                                       \
        if (CMX_SPLITL(STATS, BF) && (Q_) + 1 == TQ) do_tile(std::true_type{}, XA, G_, Q_, NG_, NQ_);              \
        else do_tile(std::false_type{}, XA, G_, Q_, NG_, NQ_);                                                 \
    } while (0)
    int pend = 0;                   // claim mode 2: the wave's outstanding claim (lane 0)
    if (DYN == 2 && slot < G && lane == 0) pend = __hip_atomic_fetch_add(a.claim_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    auto advance = [&](long& gi, int& q) {
        if (++q == TQ) {
            q = 0;
            if (DYN == 1) {
                int i = 0;
                if (lane == 0) i = __hip_atomic_fetch_add(&claim_s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                i = __builtin_amdgcn_readfirstlane(i);
                gi = (long)bx * CMX_WAVES + (i % CMX_WAVES) + (long)(i / CMX_WAVES) * nslots;
            } else if (DYN == 2) {
                // every wave with a first line claims until its first miss: G claims per launch in all, and the one that draws G - 1 made
                // the last access to the counter -- it puts the zero back for the next launch.  The claim consumed here was issued a
                // whole line earlier (the device-scope atomic takes microseconds under load: waited for in place it costs the layer-0
                // launch 20 %)
                if (gi < G) {
                    const int i = __builtin_amdgcn_readfirstlane(pend);
                    if (lane == 0 && i == (int)G - 1) __hip_atomic_store(a.claim_ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    gi = nslots + i;
                    if (gi < G && lane == 0) pend = __hip_atomic_fetch_add(a.claim_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                gi += nslots;
            }
        }
    };
    {
        long g0 = slot, g1 = slot;
        int q0 = 0, q1 = 0;
        advance(g1, q1);
        if (g0 < G) {
            const long l0 = U(line_of(g0));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                issue_x(xaA, l0, 0, j, 0);
                issue_x(xaA, l0, 0, j, 1);
            }
            issue_z(l0);
            if (PF2 && g1 < G) {
                const long l1 = line_of(g1);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    issue_x(xaB, l1, q1, j, 0);
                    issue_x(xaB, l1, q1, j, 1);
                }
            }
        }
        if (PF2) {
            while (g0 < G) {
                long g2 = g1;
                int q2 = q1;
                advance(g2, q2);
                CMX_DO_TILE(xaA, g0, q0, g2, q2);
                if (g1 >= G) break;
                long g3 = g2;
                int q3 = q2;
                advance(g3, q3);
                CMX_DO_TILE(xaB, g1, q1, g3, q3);
                g0 = g2; q0 = q2; g1 = g3; q1 = q3;
            }
        } else {
            while (g0 < G) {
                CMX_DO_TILE(xaA, g0, q0, g1, q1);
                g0 = g1; q0 = q1;
                advance(g1, q1);
            }
        }
    }
    if (STATS != 0) {
        float* part = a.stats_part + ((long)bx * CMX_WAVES + wave) * (2 * CC) + 64 * hsel;     // row [2][CC]: sums | second sums
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float s1 = ssum[t >> 1][t & 1], s2 = ssq[t >> 1][t & 1];
            s1 += __shfl_xor(s1, 16, 64);
            s2 += __shfl_xor(s2, 16, 64);
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (kg == 0) {
                part[4 * m + t] = s1;
                part[CC + 4 * m + t] = s2;
            }
        }
    }
    if (a.wave_times && lane == 0) {
        unsigned long long* wt = a.wave_times + ((long)blockIdx.x * CMX_WAVES + wave) * 2;
        wt[0] = t_wave0;
        wt[1] = wall_clock64();
    }
}

// diagnostics: when set, every cmx launch records per mix wave its start / end tick (100 MHz constant clock); tools/wave_times.py
// claim counters of mode 2: a ring, one per launch in flight (a launch leaves its counter at zero)
#define CMX_CLAIM_RING 256
// chip-wide line claiming (mode 2, not the default): a ring of counters PER DEVICE (a process may drive several GPUs), handed out with an
// atomic index (several host threads), and the slot a launch uses is zeroed on the launch stream right before it -- a kernel that
// aborted cannot leave a counter behind.  The ring is allocated by rpb_line_claim_set(2) (an explicit call, like a communicator's init);
// with RPB_LINE_CLAIM=2 from the environment the first launch on a device allocates it (never inside a stream capture).
static int* g_cmx_claim[64] = {nullptr};
static std::atomic<unsigned> g_cmx_claim_next{0};
static std::mutex g_cmx_claim_mu;                     // several host threads may launch on one device: allocate the ring once
static int* cmx_claim_ring(hipStream_t st, bool may_alloc) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(g_cmx_claim_mu);
    if (!g_cmx_claim[dev] && may_alloc) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (st) (void)hipStreamIsCapturing(st, &cs);
        int* ring = nullptr;
        if (cs == hipStreamCaptureStatusNone && hipMalloc(&ring, CMX_CLAIM_RING * 64) == hipSuccess) g_cmx_claim[dev] = ring;
    }
    return g_cmx_claim[dev];
}
void rpb_cmx_claim_prealloc() { (void)cmx_claim_ring(nullptr, true); }
static void cmx_claim_setup(CmxArgs& a, hipStream_t st) {
    a.claim_mode = rpb_line_claim_mode();
    a.claim_ctr = nullptr;
    if (a.claim_mode == 2) {
        int* ring = cmx_claim_ring(st, true);
        if (ring) {
            a.claim_ctr = ring + 16 * (g_cmx_claim_next.fetch_add(1) % CMX_CLAIM_RING);
            (void)hipMemsetAsync(a.claim_ctr, 0, 64, st);
        } else {
            a.claim_mode = 1;
        }
    }
}
static unsigned long long* g_cmx_wave_times = nullptr;
extern "C" int rpb_cmx_debug_wave_times(void* buf) {
    g_cmx_wave_times = static_cast<unsigned long long*>(buf);
    return 0;
}

static size_t cmx_lds(int Wp, int waves, bool dft = false, bool wg = false, bool dft_sb = false, bool sb = false) {
    if (sb) return (size_t)(24 * 64 + 3 * Wp * 4 + waves * 4 * 64) * 16 + 3 * 64 * 4 + 16;      // bf16 spectra without the fused stage: one-plane z2 slices
    if (dft_sb) return (size_t)(24 * 64 + waves * 4 * 64) * 16 + 3 * 64 * 4 + (size_t)((Wp + 31) / 32) * 3 * 2 * 64 * 16 + 16;
    if (wg) return (size_t)(24 * 64 + 3 * Wp * 4 + waves * 12 * 64) * 16 + 3 * 64 * 4 + (size_t)CMX_WG_PAIRS * 8 * 64 * 16 + 2 * CMX_WG_PAIRS * 4;
    return (size_t)(24 * 64 + (dft ? 0 : 3 * Wp * 4) + waves * 12 * 64) * 16 + 3 * 64 * 4 + (dft ? (size_t)((Wp + 31) / 32) * 3 * 2 * 64 * 16 : 0) + 16;
}

// GW [K2][Wp] -> three bf16 planes in A-operand row order [plane][w][kg] (the DFT variant's inverse-stage operand, read through L1)
// h2_exp >= 0 (f16x2): two fp16 planes of GW * 2^h2_exp in slots 0, 1 (slot 2 repeats slot 1)
__global__ void cmx_gw_prep_kernel(const float* __restrict__ GW, u32x4* __restrict__ out, int K2, int Wp, int h2_exp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Wp * 4) return;
    const int w = idx >> 2, kgw = idx & 3;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * kgw + e;
        v[e] = k < K2 ? GW[k * Wp + w] : 0.f;
        if (h2_exp >= 0) v[e] = __builtin_ldexpf(v[e], h2_exp);
    }
    bf16x8 h, md, lo;
    if (h2_exp >= 0) {
        split8h(v, h, md);
        lo = md;
    } else {
        split8(v, h, md, lo);
    }
    out[(0 * Wp + w) * 4 + kgw] = __builtin_bit_cast(u32x4, h);
    out[(1 * Wp + w) * 4 + kgw] = __builtin_bit_cast(u32x4, md);
    out[(2 * Wp + w) * 4 + kgw] = __builtin_bit_cast(u32x4, lo);
}
bool rpb_cmx_dft_supported(int Wp, int K2f) { return K2f > 0 && K2f <= 32 && cmx_lds(Wp, CMX_WAVES_DFT, true) <= 160 * 1024; }

// the bf16-pipe kernel covers the C = 64 spectral instances; everything else stays on rpb_cell.hip
bool rpb_cmx_supported(long ncell, int KC, int CO, int K2, int Wp, bool spec, bool gather) {
    static const bool off = getenv("RPB_CELL_MIX_F32") && atoi(getenv("RPB_CELL_MIX_F32")) == 1;   // exact-fp32 MFMA kernel
    return !off && spec && !gather && KC == 64 && CO == 64 && K2 > 0 && K2 <= 32 && Wp >= 32 && ncell % Wp == 0 &&
           cmx_lds(Wp, CMX_WAVES_A) <= 160 * 1024;
}

// ---- C = 128 (template parameter C2): workgroup pairs, one output half each
static size_t cmx_lds128(int Wp, int waves) { return (size_t)(48 * 64 + 3 * Wp * 4 + waves * 12 * 64) * 16 + 3 * 128 * 4 + 16; }
static int cmx_waves128(int Wp) { return cmx_lds128(Wp, 8) <= 160 * 1024 ? 8 : (cmx_lds128(Wp, 6) <= 160 * 1024 ? 6 : 0); }
bool rpb_cmx128_supported(long ncell, int KC, int CO, int K2, int Wp, bool spec, bool gather) {
    static const bool off = (getenv("RPB_CELL_MIX_F32") && atoi(getenv("RPB_CELL_MIX_F32")) == 1) ||
                            (getenv("RPB_CELL_MIX_128_F32") && atoi(getenv("RPB_CELL_MIX_128_F32")) == 1);   // exact-fp32 MFMA kernel
    return !off && spec && !gather && KC == 128 && CO == 128 && K2 > 0 && K2 <= 32 && Wp >= 32 && ncell % Wp == 0 && cmx_waves128(Wp) > 0;
}
long rpb_cmx128_stat_rows(long ncell, int Wp) {
    const int waves = cmx_waves128(Wp);
    const long G = ncell / Wp;
    long pairs = rpb_num_cus() / 2;
    const long need = (G + waves - 1) / waves;
    if (pairs > need) pairs = need;
    return pairs * waves;
}
static int cmx128_launch(const CmxArgs& a, int stats, hipStream_t st) {
    if (a.bf16_io || a.feat_w || a.y1out)
        RPB_FAIL(RPB_ERR_UNSUPPORTED, "cmx (C = 128): plain fp32-storage launches only (no bf16 storage, feature input or fused stage)");
    if (a.crop_T > 0 && (stats != 0 || !a.bnb.mean || a.crop_H <= 0 || a.crop_W <= 0 || a.ncell % ((long)a.Wp * a.Hp * a.Tp) != 0))
        RPB_FAIL(RPB_ERR_UNSUPPORTED, "cmx (C = 128): the crop-only mode is an eval path (output transform, no statistics)");
    const int waves = cmx_waves128(a.Wp);
    const int grid = 2 * (int)(rpb_cmx128_stat_rows(a.ncell, a.Wp) / waves);
    const size_t lds = cmx_lds128(a.Wp, waves);
#define RPB_CMX128(ST_, W_)                                                                                                                        \
    if (stats == ST_ && waves == W_) {                                                                                                             \
        (void)hipFuncSetAttribute((const void*)cmx_kernel<ST_, false, false, false, false, false, W_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((cmx_kernel<ST_, false, false, false, false, false, W_>), dim3(grid), dim3(W_ * 64), lds, st, a);                      \
        RPB_CHECK_LAUNCH("cell_mix(bf16x3, C = 128)");                                                                                             \
    }
    RPB_CMX128(0, 8) RPB_CMX128(1, 8) RPB_CMX128(2, 8) RPB_CMX128(0, 6) RPB_CMX128(1, 6) RPB_CMX128(2, 6)
#undef RPB_CMX128
    RPB_FAIL(RPB_ERR_UNSUPPORTED, "cmx (C = 128): bad stats mode %d", stats);
}

// ---- the STATS == 2 launch with the Conv3d weight gradient in wave pairs (WG)
long rpb_cmx_wg_slots(long ncell, int Wp) {
    const long G = ncell / Wp;
    long grid = rpb_num_cus();
    const long need = (G + CMX_WG_PAIRS - 1) / CMX_WG_PAIRS;
    if (grid > need) grid = need;
    return grid * CMX_WG_PAIRS;
}
int rpb_cmx_wg_launch(const CmxArgs& a_in, hipStream_t st) {
    RPB_REQUIRE(a_in.x && a_in.Wm && a_in.z2 && a_in.GW && a_in.out && a_in.stats_part && a_in.wg_part && a_in.bnb_s && a_in.bnb.mean,
                "cell_mix_wgrad: null pointer");
    RPB_REQUIRE(!a_in.bias && !a_in.xf.mean && !a_in.bf16_io && !a_in.feat_w && !a_in.y1out && a_in.crop_T == 0,
                "cell_mix_wgrad: plain fp32 backward launch only");
    CmxArgs a = a_in;
    a.wave_times = g_cmx_wave_times;
    cmx_claim_setup(a, st);
    const size_t lds = cmx_lds(a.Wp, CMX_WG_PAIRS, false, true);
    RPB_REQUIRE(lds <= 160 * 1024, "cell_mix_wgrad: Wp=%d does not fit LDS", a.Wp);
    const int grid = (int)(rpb_cmx_wg_slots(a.ncell, a.Wp) / CMX_WG_PAIRS);
    (void)hipFuncSetAttribute((const void*)cmx_kernel<2, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((cmx_kernel<2, false, false, false, true>), dim3(grid), dim3(2 * CMX_WG_PAIRS * 64), lds, st, a);
    RPB_CHECK_LAUNCH("cell_mix_wgrad(bf16x3, wave pairs)");
}

long rpb_cmx_stat_rows(long ncell, int Wp, int stats) {
    const int waves = CMX_WAVES_OF(stats);
    const long G = ncell / Wp;
    long grid = rpb_num_cus();
    const long need = (G + waves - 1) / waves;
    if (grid > need) grid = need;
    return grid * waves;
}

int rpb_cmx_launch(const CmxArgs& a_in, int stats, hipStream_t st) {
    CmxArgs a = a_in;
    a.wave_times = g_cmx_wave_times;
    cmx_claim_setup(a, st);
    if (a.c128) return cmx128_launch(a, stats, st);
    if (a.y1out) {                  // eval with the next layer's forward W stage fused in
        if (a.crop_T > 0 || stats != 0 || !a.bnb.mean || (a.bf16_io && a.feat_w) || !a.FWt || !a.gw_planes || !rpb_cmx_dft_supported(a.Wp, a.K2f))
            RPB_FAIL(RPB_ERR_UNSUPPORTED, "cmx: the fused W stage needs the eval path (output transform), a scratch buffer and K2f <= 32");
        if (a.h2 && a.bf16_io) RPB_FAIL(RPB_ERR_UNSUPPORTED, "cmx: the f16x2 arithmetic is an fp32-storage path");
        // (f16x2: the feature-field launch keeps unscaled accumulators, the C = 64 launch carries 2^4 -- H2W in the kernel)
        hipLaunchKernelGGL(cmx_gw_prep_kernel, dim3((a.Wp * 4 + 255) / 256), dim3(256), 0, st, a.GW, (u32x4*)a.gw_planes, a.K2, a.Wp,
                           a.h2 ? a.spec_exp + (a.feat_w ? 0 : 4) : -1);
        const bool dsb = a.bf16_io && a.spec_bf16;
        const int waves = dsb ? CMX_WAVES_DFT_SB : CMX_WAVES_DFT;
        const long G = a.ncell / a.Wp;
        long grid = rpb_num_cus();
        if (grid > (G + waves - 1) / waves) grid = (G + waves - 1) / waves;
        const size_t lds = cmx_lds(a.Wp, waves, true, false, dsb);
        if (a.bf16_io && a.spec_bf16) {
            (void)hipFuncSetAttribute((const void*)cmx_kernel<0, true, false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((cmx_kernel<0, true, false, true, false, true>), dim3((unsigned)grid), dim3(waves * 64), lds, st, a);
        } else if (a.bf16_io) {
            (void)hipFuncSetAttribute((const void*)cmx_kernel<0, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((cmx_kernel<0, true, false, true>), dim3((unsigned)grid), dim3(waves * 64), lds, st, a);
        } else if (a.feat_w && a.h2) {
            (void)hipFuncSetAttribute((const void*)cmx_kernel<0, false, true, true, false, false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((cmx_kernel<0, false, true, true, false, false, 0, true>), dim3((unsigned)grid), dim3(waves * 64), lds, st, a);
        } else if (a.h2) {
            (void)hipFuncSetAttribute((const void*)cmx_kernel<0, false, false, true, false, false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((cmx_kernel<0, false, false, true, false, false, 0, true>), dim3((unsigned)grid), dim3(waves * 64), lds, st, a);
        } else if (a.feat_w) {
            (void)hipFuncSetAttribute((const void*)cmx_kernel<0, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((cmx_kernel<0, false, true, true>), dim3((unsigned)grid), dim3(waves * 64), lds, st, a);
        } else {
            (void)hipFuncSetAttribute((const void*)cmx_kernel<0, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((cmx_kernel<0, false, false, true>), dim3((unsigned)grid), dim3(waves * 64), lds, st, a);
        }
        RPB_CHECK_LAUNCH("cell_mix(bf16x3, + next W stage)");
    }
    if (a.crop_T > 0 && (stats != 0 || a.feat_w || a.crop_H <= 0 || a.crop_W <= 0 || a.ncell % ((long)a.Wp * a.Hp * a.Tp) != 0))
        RPB_FAIL(RPB_ERR_UNSUPPORTED, "cmx: the crop-only mode is an eval path (no statistics, no fused stage, no feature input)");
    const int waves = CMX_WAVES_OF(stats);
    const int grid = (int)(rpb_cmx_stat_rows(a.ncell, a.Wp, stats) / waves);
    const size_t lds = cmx_lds(a.Wp, waves);
    if (a.h2) {                     // f16x2: the eval launches only (output transform; here: the crop-only last layer or a plain eval layer)
        if (stats != 0 || !a.bnb.mean || a.bf16_io || a.feat_w) RPB_FAIL(RPB_ERR_UNSUPPORTED, "cmx: the f16x2 arithmetic covers the fp32-storage eval launches");
        (void)hipFuncSetAttribute((const void*)cmx_kernel<0, false, false, false, false, false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((cmx_kernel<0, false, false, false, false, false, 0, true>), dim3(grid), dim3(waves * 64), lds, st, a);
        RPB_CHECK_LAUNCH("cell_mix(f16x2)");
    }
#define RPB_CMX(ST_)                                                                                                  \
    if (stats == ST_) {                                                                                               \
        (void)hipFuncSetAttribute((const void*)cmx_kernel<ST_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((cmx_kernel<ST_>), dim3(grid), dim3(waves * 64), lds, st, a);                          \
        RPB_CHECK_LAUNCH("cell_mix(bf16x3)");                                                                         \
    }
    if (a.feat_w) {
        if (stats == 2 || a.bf16_io) RPB_FAIL(RPB_ERR_UNSUPPORTED, "cmx: the feature-field input is a forward (layer 0) path");
#define RPB_CMXF(ST_)                                                                                                          \
    if (stats == ST_) {                                                                                                        \
        (void)hipFuncSetAttribute((const void*)cmx_kernel<ST_, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((cmx_kernel<ST_, false, true>), dim3(grid), dim3(waves * 64), lds, st, a);                          \
        RPB_CHECK_LAUNCH("cell_mix(bf16x3, feature fields)");                                                                  \
    }
        RPB_CMXF(0) RPB_CMXF(1)
#undef RPB_CMXF
    }
    if (a.bf16_io) {
        if (stats != 0) RPB_FAIL(RPB_ERR_UNSUPPORTED, "cmx: bf16 activation storage is an eval / rollout path (no statistics)");
        if (a.spec_bf16) {
            const int wsb = CMX_WAVES_SB;
            const long Gl = a.crop_T > 0 ? (a.ncell / ((long)a.Wp * a.Hp * a.Tp)) * a.crop_T * a.crop_H : a.ncell / a.Wp;
            long gsb = rpb_num_cus();
            if (gsb > (Gl + wsb - 1) / wsb) gsb = (Gl + wsb - 1) / wsb;
            const size_t ldsb = cmx_lds(a.Wp, wsb, false, false, false, true);
            (void)hipFuncSetAttribute((const void*)cmx_kernel<0, true, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
            hipLaunchKernelGGL((cmx_kernel<0, true, false, false, false, true>), dim3((unsigned)gsb), dim3(wsb * 64), ldsb, st, a);
            RPB_CHECK_LAUNCH("cell_mix(bf16x3, bf16 storage and spectra)");
        }
        (void)hipFuncSetAttribute((const void*)cmx_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((cmx_kernel<0, true>), dim3(grid), dim3(waves * 64), lds, st, a);
        RPB_CHECK_LAUNCH("cell_mix(bf16x3, bf16 storage)");
    }
    RPB_CMX(0) RPB_CMX(1) RPB_CMX(2)
#undef RPB_CMX
    RPB_FAIL(RPB_ERR_UNSUPPORTED, "cmx: bad stats mode %d", stats);
}
