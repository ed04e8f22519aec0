// Truncated-DFT stage on the bf16 matrix pipe ("axg"): the same contraction as rpb_axis_gemm.hip,
//
//   out[g][o][n] = sum_k M[o][k] * in[g][k][n]        n contiguous, k / o strided, g = batch
//
// (every stage of rfftn / irfftn of fno.py:48,63 and their adjoints) with both operands split into three bf16 planes and the
// products on v_mfma_f32_16x16x32_bf16 -- the fp32-grade arithmetic of rpb_cmx.hip / rpb_conv3x.hip (hi*lo + lo*hi + mid*mid +
// hi*mid + mid*hi + hi*hi, fp32 accumulate; Rel-L2 vs fp64 ~2e-7).
//
// Why: as fp32 MFMA the H stages are matrix-pipe-bound (832 x 32 x 2 x 134 MFMAs of 64 cycles = 0.30 ms of pipe time for a
// 1.08 GB problem whose HBM time is 0.18 ms; O = 48 pads to 64 rows).  The contraction index k is the SLOW memory index here, but
// an MFMA operand only needs "8 consecutive k per lane": lane (c = lane & 15, kg = lane >> 4) issues 8 loads of 16 B, one per
// k = 32 s + 8 kg + e, each fetching columns n0 + 4 c .. 4 c + 3 -- every load instruction reads 4 rows x 256 contiguous bytes,
// and the 4 columns are the lane's B operands of 4 MFMA column tiles (column j of tile t <-> n = n0 + 4 j + t), so the output
// store is 16 B per lane / 256 B per row as well.  The matrix is split once per workgroup into LDS in A-operand order
// (16-row tiles: O = 48 is exact).
#include "rpb_axg.h"
// (cache policy of the streaming loads / stores: RPB_STREAM_AUX, rpb_common.h -- nt by default since round 5)
#include <stdlib.h>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

namespace {
__device__ __forceinline__ u32x4 ld16(rsrc_t r, int voff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, RPB_STREAM_AUX));
}
__device__ __forceinline__ void st16(f32x4v v, rsrc_t r, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, RPB_STREAM_AUX);
}
__device__ __forceinline__ float trunc_bf16(float v) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_hi(float a, float b) {
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
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
// the same with only the first `npairs` element pairs live (wave-uniform): the rest are zero planes at no vector cost
__device__ __forceinline__ void split8n(const float (&v)[8], int npairs, bf16x8& h, bf16x8& m, bf16x8& l) {
    u32x4 uh = {0u, 0u, 0u, 0u}, um = uh, ul = uh;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q >= npairs) break;
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
}  // namespace

#define AXG_WAVES 8

// MT = 16-row output tiles per pass (the accumulators of a pass: MT x 4 column tiles x 4 registers)
// BFIN: the input is STORED as bf16 (in_g / in_k in bf16 elements; BASELINE.json configs[4]): the B operand is one exact bf16
//       plane, so a (row tile, column tile) costs 3 products instead of 6; a lane's 8 loads are 8 B each (4 columns).
// NW: waves per workgroup (= per CU).  The bf16-input instance (156 registers) at three waves per SIMD measured SLOWER (0.144 -> 0.162 ms,
//     combustion forward; the same change gained 11 % in rpb_cmx.hip's fused launch, profiles/r06b_ab5_waves_dft_sb.txt): 8 stays
#ifndef AXG_WAVES_XF
#define AXG_WAVES_XF 8      /* 12 (three waves per SIMD at 165 registers): the micro-benchmark gains 3-6 % (1.19-1.23 -> 1.16 ms), the train step loses
                               0.1 ms (34.24 / 34.26 -> 34.34 / 34.40 ms, A/B twice, tools/r6b_ab7.sh): 8 stays */
#endif
#ifndef AXG_WAVES_BFIN
#define AXG_WAVES_BFIN 8
#endif
template <int MT, bool XF, bool BFIN = false, int NW = AXG_WAVES>
__global__ __launch_bounds__(NW * 64) void axg_kernel(AxgArgs a) {
    static_assert(!(BFIN && XF), "bf16 storage holds materialised activations: no lazy transform");
    extern __shared__ u32x4 Ml[];            // [ks][plane 3][mt][lane 64]   M in A-operand order
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, kg = lane >> 4;
    const int KS = (a.k_valid + 31) >> 5;                    // K-steps that hold non-zero input
    const int mtiles = (a.O + 15) >> 4;
    // The LAST K-step usually holds few rows (Wp = 134: 6 of 32).  With the regular assignment k = 32 ks + 8 kg + e lane group 0 would
    // carry all of them and every lane would still run the lazy BatchNorm + GELU and the split on 8 elements: a fifth of the forward W
    // stage's vector work on zeros.  The tail step instead deals its rows `epl` per lane group, k = 32 ks + epl kg + e for e < epl, and
    // elements e >= epl are compile-time-uniformly skipped (zero planes): 6 rows cost 2 / 8 of a step.  The matrix is laid out to match.
    const int rem = a.k_valid - 32 * (KS - 1);
    const int epl = BFIN ? 8 : (rem + 3) >> 2;               // rows per lane group in the last step (8 = the regular assignment)
    for (int idx = tid; idx < KS * mtiles * 64; idx += blockDim.x) {
        const int l = idx & 63, mt = (idx >> 6) % mtiles, ks = (idx >> 6) / mtiles;
        const int o = 16 * mt + (l & 15);
        const int per = ks == KS - 1 ? epl : 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 32 * ks + per * (l >> 4) + e;
            v[e] = (o < a.O && e < per && k < a.k_valid) ? a.Mt[(long)k * a.O + o] : 0.f;
        }
        bf16x8 h, md, lo;
        split8(v, h, md, lo);
        Ml[((ks * 3 + 0) * mtiles + mt) * 64 + l] = __builtin_bit_cast(u32x4, h);
        Ml[((ks * 3 + 1) * mtiles + mt) * 64 + l] = __builtin_bit_cast(u32x4, md);
        Ml[((ks * 3 + 2) * mtiles + mt) * 64 + l] = __builtin_bit_cast(u32x4, lo);
    }
    __syncthreads();

    XParam xp[4];
    if (XF) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            xp[t] = xf_load(a.xf, 4 * c16 + t);       // N == channels == 64 * strips, strip 0 .. N/64
            xp[t].is *= xp[t].ga;                     // BatchNorm as one fused multiply-add: x * (invstd gamma) + (beta - mean invstd gamma)
            xp[t].be -= xp[t].mu * xp[t].is;
        }
    }
    const bool xgelu = a.xf.gelu != 0;
    const int strips = a.N >> 6;
    const long items = (long)a.G * strips;
    const long nslots = (long)gridDim.x * NW;
    const int passes = (mtiles + MT - 1) / MT;
    constexpr int EB = BFIN ? 2 : 4;                                         // bytes per input element
    const unsigned in_bytes = (unsigned)(((long)(a.k_valid - 1) * a.in_k + 64) * EB);
    const unsigned out_bytes = (unsigned)(((long)(a.O - 1) * a.out_o + 64) * 4);
    const int ioff = (8 * kg) * (int)a.in_k * EB + c16 * 4 * EB;             // row 8 kg, columns 4 c .. 4 c + 3
    const int ioff_tail = (epl * kg) * (int)a.in_k * EB + c16 * 4 * EB;      // last step: row epl kg
    const int ooff = (4 * kg) * (int)a.out_o * 4 + c16 * 16;                 // row 4 mg, columns 4 c ..
    const int istep = (int)a.in_k * EB;                                      // bytes between consecutive k rows

    for (long it = (long)blockIdx.x * NW + wave; it < items; it += nslots) {
        const long g = it / strips;
        const int n0 = (int)(it - g * strips) << 6;
        const rsrc_t ri = make_rsrc(reinterpret_cast<const char*>(a.in) + (g * a.in_g + n0) * EB, in_bytes);
        const rsrc_t ro = make_rsrc(a.out + g * a.out_g + n0, out_bytes);
        if (XF && strips > 1) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                xp[t] = xf_load(a.xf, n0 + 4 * c16 + t);
                xp[t].is *= xp[t].ga;
                xp[t].be -= xp[t].mu * xp[t].is;
            }
        }
        for (int p = 0; p < passes; ++p) {
            asm volatile("" ::: "memory");
            f32x4v acc[MT][4];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[i][t] = f32x4v{0.f, 0.f, 0.f, 0.f};
            u32x4 zr[8];
            auto issue = [&](int ks) {
                const bool tail = !BFIN && ks == KS - 1;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (BFIN) {
                        typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
                        const u32x2v w = __builtin_amdgcn_raw_buffer_load_b64(ri, ioff + (32 * ks + e) * istep, 0, 0);
                        zr[e][0] = w[0];
                        zr[e][1] = w[1];
                    } else if (!tail) {
                        zr[e] = ld16(ri, ioff + (32 * ks + e) * istep);
                    } else if (e < epl) {                                     // (uniform) rows past epl are not even requested
                        zr[e] = ld16(ri, ioff_tail + (32 * ks + e) * istep);
                    }
                }
            };
            issue(0);
            for (int ks = 0; ks < KS; ++ks) {
                const bool tail = !BFIN && ks == KS - 1;
                const int per = tail ? epl : 8;
                // ---- this step's 8 x 4 values -> B planes of the 4 column tiles (lazy BN+GELU of the producer applied here)
                bf16x8 Bh[4], Bm[4], Bl[4];
                if (BFIN) {                 // word 0 = columns (0, 1), word 1 = columns (2, 3): gather each column's 8 rows
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        u32x4 u;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            u[q] = __builtin_amdgcn_perm(zr[2 * q + 1][t >> 1], zr[2 * q][t >> 1], (t & 1) ? 0x07060302u : 0x05040100u);
                        Bh[t] = __builtin_bit_cast(bf16x8, u);
                    }
                }
                if (!BFIN) {
                    // TAIL_ is a compile-time copy of the wave-uniform `tail`: the masking of rows past k_valid (a compare and four selects per
                    // row) and the partial split exist in the last K-step's instance only -- every row of an earlier step is < 32 (KS - 1) < k_valid
                    auto planes = [&](auto tail_tag) {
                        constexpr bool TAIL_ = decltype(tail_tag)::value;
                        float v[4][8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            if (TAIL_ && e >= per) {                              // uniform: only in the last step
#pragma unroll
                                for (int t = 0; t < 4; ++t) v[t][e] = 0.f;
                                continue;
                            }
                            const f32x4v x4 = __builtin_bit_cast(f32x4v, zr[e]);
                            // channel pairs: packed fp32 math; the two pairs' erf polynomials in lock-step (rpb_common.h, gelu2x2)
                            f32x2 xa = f32x2{x4[0], x4[1]}, xb = f32x2{x4[2], x4[3]};
                            if (XF) {
                                xa = pk_fma(xa, f32x2{xp[0].is, xp[1].is}, f32x2{xp[0].be, xp[1].be});
                                xb = pk_fma(xb, f32x2{xp[2].is, xp[3].is}, f32x2{xp[2].be, xp[3].be});
                                if (xgelu) gelu2x2(xa, xb);
                                if (TAIL_) {                                      // rows past k_valid must stay zero: the transform of a masked load is not
                                    const bool live = 32 * ks + per * kg + e < a.k_valid;
                                    xa = live ? xa : pk2(0.f);
                                    xb = live ? xb : pk2(0.f);
                                }
                            }
                            v[0][e] = xa[0];
                            v[1][e] = xa[1];
                            v[2][e] = xb[0];
                            v[3][e] = xb[1];
                        }
                        if (TAIL_) {
#pragma unroll
                            for (int t = 0; t < 4; ++t) split8n(v[t], (per + 1) >> 1, Bh[t], Bm[t], Bl[t]);
                        } else {
#pragma unroll
                            for (int t = 0; t < 4; ++t) split8(v[t], Bh[t], Bm[t], Bl[t]);
                        }
                    };
                    if (tail) planes(std::true_type{});
                    else planes(std::false_type{});
                }
                // ---- next step's loads go out now and are in flight during the MFMAs below
                if (ks + 1 < KS) issue(ks + 1);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int mt = p * MT + i;
                    if (mt >= mtiles) break;                                   // uniform
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, Ml[((ks * 3 + 0) * mtiles + mt) * 64 + lane]);
                    const bf16x8 am = __builtin_bit_cast(bf16x8, Ml[((ks * 3 + 1) * mtiles + mt) * 64 + lane]);
                    const bf16x8 al = __builtin_bit_cast(bf16x8, Ml[((ks * 3 + 2) * mtiles + mt) * 64 + lane]);
                    if (BFIN) {
                        if (RPB_BF16_CONST_PLANES > 2) {
#pragma unroll
                            for (int t = 0; t < 4; ++t) acc[i][t] = mfma16(al, Bh[t], acc[i][t]);
                        }
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[i][t] = mfma16(am, Bh[t], acc[i][t]);
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[i][t] = mfma16(ah, Bh[t], acc[i][t]);
                        continue;
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[i][t] = mfma16(ah, Bl[t], acc[i][t]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[i][t] = mfma16(al, Bh[t], acc[i][t]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[i][t] = mfma16(am, Bm[t], acc[i][t]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[i][t] = mfma16(ah, Bm[t], acc[i][t]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[i][t] = mfma16(am, Bh[t], acc[i][t]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[i][t] = mfma16(ah, Bh[t], acc[i][t]);
                }
            }
            // ---- out[g][16 mt + 4 mg + r][n0 + 4 c + t]: 16 B per lane, 256 B per row; rows >= O fall outside the descriptor
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int mt = p * MT + i;
                if (mt >= mtiles) break;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    f32x4v o;
#pragma unroll
                    for (int t = 0; t < 4; ++t) o[t] = acc[i][t][r];
                    st16(o, ro, ooff + (16 * mt + r) * (int)a.out_o * 4);
                }
            }
        }
    }
}

// Short contractions (K <= 64, e.g. the inverse H stage K = 48 -> O = 268): the B planes of the whole K range stay in registers
// and the 16-row output tiles are walked in passes of MT against them -- the input is loaded and split once per item instead of
// once per pass.
template <int MT>
__global__ __launch_bounds__(AXG_WAVES * 64) void axg_resident_kernel(AxgArgs a) {
    extern __shared__ u32x4 Ml[];            // [ks][plane 3][mt][lane 64]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, kg = lane >> 4;
    const int KS = (a.k_valid + 31) >> 5;                    // 1 or 2
    const int mtiles = (a.O + 15) >> 4;
    for (int idx = tid; idx < KS * mtiles * 64; idx += blockDim.x) {
        const int l = idx & 63, mt = (idx >> 6) % mtiles, ks = (idx >> 6) / mtiles;
        const int o = 16 * mt + (l & 15);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 32 * ks + 8 * (l >> 4) + e;
            v[e] = (o < a.O && k < a.k_valid) ? a.Mt[(long)k * a.O + o] : 0.f;
        }
        bf16x8 h, md, lo;
        split8(v, h, md, lo);
        Ml[((ks * 3 + 0) * mtiles + mt) * 64 + l] = __builtin_bit_cast(u32x4, h);
        Ml[((ks * 3 + 1) * mtiles + mt) * 64 + l] = __builtin_bit_cast(u32x4, md);
        Ml[((ks * 3 + 2) * mtiles + mt) * 64 + l] = __builtin_bit_cast(u32x4, lo);
    }
    __syncthreads();
    const int strips = a.N >> 6;
    const long items = (long)a.G * strips;
    const long nslots = (long)gridDim.x * AXG_WAVES;
    const int passes = (mtiles + MT - 1) / MT;
    const unsigned in_bytes = (unsigned)(((long)(a.k_valid - 1) * a.in_k + 64) * 4);
    const int OB = a.out_bf16 ? 2 : 4;                                       // bytes per output element
    const unsigned out_bytes = (unsigned)(((long)(a.O - 1) * a.out_o + 64) * OB);
    const int ioff = (8 * kg) * (int)a.in_k * 4 + c16 * 16;
    const int ooff = (4 * kg) * (int)a.out_o * OB + c16 * 4 * OB;

    u32x4 zr[2][8];
    auto issue = [&](long it) {
        const long g = it / strips;
        const int n0 = (int)(it - g * strips) << 6;
        const rsrc_t ri = make_rsrc(a.in + g * a.in_g + n0, in_bytes);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) zr[ks][e] = ld16(ri, ioff + (32 * ks + e) * (int)a.in_k * 4);   // ks >= KS: out of range -> 0
    };
    long it = (long)blockIdx.x * AXG_WAVES + wave;
    if (it < items) issue(it);
    for (; it < items; it += nslots) {
        const long g = it / strips;
        const int n0 = (int)(it - g * strips) << 6;
        const rsrc_t ro = make_rsrc(reinterpret_cast<char*>(a.out) + (g * a.out_g + n0) * OB, out_bytes);
        bf16x8 Bh[2][4], Bm[2][4], Bl[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = __builtin_bit_cast(f32x4v, zr[ks][e])[t];
                split8(v, Bh[ks][t], Bm[ks][t], Bl[ks][t]);
            }
        if (it + nslots < items) issue(it + nslots);            // the next item's input is in flight during all passes below
        for (int p = 0; p < passes; ++p) {
            asm volatile("" ::: "memory");
            f32x4v acc[MT][4];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[i][t] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks >= KS) break;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int mt = p * MT + i;
                    if (mt >= mtiles) break;
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, Ml[((ks * 3 + 0) * mtiles + mt) * 64 + lane]);
                    const bf16x8 am = __builtin_bit_cast(bf16x8, Ml[((ks * 3 + 1) * mtiles + mt) * 64 + lane]);
                    const bf16x8 al = __builtin_bit_cast(bf16x8, Ml[((ks * 3 + 2) * mtiles + mt) * 64 + lane]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[i][t] = mfma16(ah, Bl[ks][t], acc[i][t]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[i][t] = mfma16(al, Bh[ks][t], acc[i][t]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[i][t] = mfma16(am, Bm[ks][t], acc[i][t]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[i][t] = mfma16(ah, Bm[ks][t], acc[i][t]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[i][t] = mfma16(am, Bh[ks][t], acc[i][t]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[i][t] = mfma16(ah, Bh[ks][t], acc[i][t]);
                }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int mt = p * MT + i;
                if (mt >= mtiles) break;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    f32x4v o;
#pragma unroll
                    for (int t = 0; t < 4; ++t) o[t] = acc[i][t][r];
                    if (a.out_bf16) {       // spectra stored as bf16 (round to nearest even): 8 B per lane, 128 B per row and strip
                        typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
                        typedef float f32x2v __attribute__((ext_vector_type(2)));
                        typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
                        u32x2v pk;
                        pk[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{o[0], o[1]}, bf16x2v));
                        pk[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{o[2], o[3]}, bf16x2v));
                        __builtin_amdgcn_raw_buffer_store_b64(pk, ro, ooff + (16 * mt + r) * (int)a.out_o * 2, 0, 0);
                    } else {
                        st16(o, ro, ooff + (16 * mt + r) * (int)a.out_o * 4);
                    }
                }
            }
        }
    }
}

static size_t axg_lds(int k_valid, int O) { return (size_t)((k_valid + 31) / 32) * 3 * ((O + 15) / 16) * 64 * 16; }

// N in 64-column strips, 16 B-aligned strides, 32-bit row offsets, the split matrix fits LDS; RPB_AXIS_GEMM_F32=1 forces the exact-fp32 kernel
bool rpb_axg_supported(int G, int K, int O, int N, long in_g, long in_k, long out_g, long out_o, int k_valid, int accumulate,
                       bool has_xf) {
    static const bool off = getenv("RPB_AXIS_GEMM_F32") && atoi(getenv("RPB_AXIS_GEMM_F32")) == 1;
    if (off || accumulate || N % 64 != 0 || k_valid < 1) return false;
    if ((in_g | in_k | out_g | out_o) & 3) return false;
    if (has_xf && N > 128) return false;
    if ((long)(k_valid + 32) * in_k * 4 + 256 >= (1L << 31) || (long)(O + 16) * out_o * 4 + 256 >= (1L << 31)) return false;
    return axg_lds(k_valid, O) <= 160 * 1024;
}

int rpb_axg_launch(const AxgArgs& a, hipStream_t st) {
    const int mtiles = (a.O + 15) / 16;
    const long items = (long)a.G * (a.N / 64);
    long grid = rpb_num_cus();
    const long need = (items + AXG_WAVES - 1) / AXG_WAVES;
    if (grid > need) grid = need;
    const size_t lds = axg_lds(a.k_valid, a.O);
    const bool xf = a.xf.mean != nullptr;
    if (a.out_bf16 && (a.in_bf16 || xf || a.k_valid > 64 || mtiles <= 4))
        RPB_FAIL(RPB_ERR_UNSUPPORTED, "axg: bf16 output is built for the short-K stages with many output rows (the inverse H stage)");
    if (a.in_bf16) {
        if (xf || mtiles > 4) RPB_FAIL(RPB_ERR_UNSUPPORTED, "axg: bf16 input supports plain stages with O <= 64 (the forward W stage)");
        long gridb = rpb_num_cus();
        const long needb = (items + AXG_WAVES_BFIN - 1) / AXG_WAVES_BFIN;
        if (gridb > needb) gridb = needb;
        (void)hipFuncSetAttribute((const void*)axg_kernel<4, false, true, AXG_WAVES_BFIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((axg_kernel<4, false, true, AXG_WAVES_BFIN>), dim3((unsigned)gridb), dim3(AXG_WAVES_BFIN * 64), lds, st, a);
        RPB_CHECK_LAUNCH("axis_gemm(bf16x3, bf16 input)");
    }
#define RPB_AXG(MT_, XF_)                                                                                                  \
    {                                                                                                                      \
        (void)hipFuncSetAttribute((const void*)axg_kernel<MT_, XF_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((axg_kernel<MT_, XF_>), dim3((unsigned)grid), dim3(AXG_WAVES * 64), lds, st, a);                \
        RPB_CHECK_LAUNCH("axis_gemm(bf16x3)");                                                                             \
    }
    if (!xf && a.k_valid <= 64 && mtiles > 4) {        // short K, many output rows: B planes resident across the passes
        (void)hipFuncSetAttribute((const void*)axg_resident_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((axg_resident_kernel<2>), dim3((unsigned)grid), dim3(AXG_WAVES * 64), lds, st, a);
        RPB_CHECK_LAUNCH("axis_gemm(bf16x3, resident)");
    }
    if (mtiles <= 1) {
        if (xf) RPB_AXG(1, true) else RPB_AXG(1, false)
    } else if (mtiles == 2) {
        if (xf) {       // the training forward's W stage with the lazy BatchNorm + GELU (165 registers): AXG_WAVES_XF waves per CU
            long gridx = rpb_num_cus();
            const long needx = (items + AXG_WAVES_XF - 1) / AXG_WAVES_XF;
            if (gridx > needx) gridx = needx;
            (void)hipFuncSetAttribute((const void*)axg_kernel<2, true, false, AXG_WAVES_XF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((axg_kernel<2, true, false, AXG_WAVES_XF>), dim3((unsigned)gridx), dim3(AXG_WAVES_XF * 64), lds, st, a);
            RPB_CHECK_LAUNCH("axis_gemm(bf16x3, lazy transform)");
        } else RPB_AXG(2, false)
    } else if (mtiles == 3) {
        if (xf) RPB_AXG(3, true) else RPB_AXG(3, false)
    } else {
        if (xf) RPB_AXG(4, true) else RPB_AXG(4, false)
    }
#undef RPB_AXG
}
