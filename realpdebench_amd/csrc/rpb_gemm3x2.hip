// Token GEMM on the bf16 MFMA from split fp32 operands, N % 256 == 0:  out[m][n] = epilogue( sum_k A[m][k] * W[n][k] )
// Second organisation of rpb_gemm3x (same arithmetic, same prepared weights, same epilogue semantics), built around what the first one
// loses at K = 256 .. 768: there a 128-row tile is 4-12 LDS stages of matrix work followed by an epilogue that parks the accumulators in
// LDS and walks them in a rolled loop of "load aux / residual, activate, store" -- with ONE workgroup per CU the matrix pipe idles for
// the whole epilogue, and every epilogue read exposes a full memory latency (M = 655 k, N = 1024, K = 256: 2.5 ms plain, 3.9 ms with
// a residual or gelu'(aux) read; tools/g3bench.py).  Here:
//   * tiles are 64 rows x 256 columns, 4 waves x (64 x 64) = 64 accumulator registers per lane, so TWO workgroups fit a CU (2 waves per
//     SIMD, 48 KB of LDS each): while one workgroup sits in its epilogue (memory waits, stores, activation code) the other issues MFMAs;
//   * the epilogue works in ACCUMULATOR layout (lane = column, register = row): no LDS parking, no rolled loop; every access is a dword
//     per lane = two 128 B row segments per instruction through a per-tile buffer descriptor (rows past M are clipped by it);
//   * the epilogue's reads do not depend on the product, so the first read stream of a tile (aux, else residual: 64 dwords per lane) is
//     issued BEFORE the tile's matrix work and has arrived when the epilogue starts.
//   * in-kernel dropout: Philox yields the keep bits of 4 consecutive columns per counter, and in this layout those are 4 LANES (a quad).
//     Lane j of a quad evaluates the counters of rows r = j (mod 4) of a block (4 instead of 16 per lane), packs their 4 x 4 keep bits,
//     and one DPP quad broadcast per owner hands every lane the word that holds its (row, column) bit.
// Not covered (rpb_gemm3x keeps them): mask tensors, N = 64 / 128.
#include "rpb_gemm3x2.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 g2_bf16x8;
typedef __bf16 g2_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned g2_u32x4 __attribute__((ext_vector_type(4)));

#define G2_BM 64

__device__ __forceinline__ f32x16 g2_mfma(g2_u32x4 a, g2_u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(g2_bf16x8, a), __builtin_bit_cast(g2_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void g2_split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const f32x2 v = {x0, x1};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, g2_bf16x2));                 // v_cvt_pk_bf16_f32 (RNE)
    const f32x2 hf = {__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xFFFF0000u)};
    const f32x2 r1 = v - hf;                                                                  // exact
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, g2_bf16x2));
    const f32x2 mf = {__builtin_bit_cast(float, m << 16), __builtin_bit_cast(float, m & 0xFFFF0000u)};
    const f32x2 r2 = r1 - mf;                                                                 // exact
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, g2_bf16x2));
}
__device__ __forceinline__ void g2_split8(f32x4 v0, f32x4 v1, g2_u32x4& h, g2_u32x4& m, g2_u32x4& l) {
    unsigned hh[4], mm[4], ll[4];
    g2_split_pair(v0[0], v0[1], hh[0], mm[0], ll[0]);
    g2_split_pair(v0[2], v0[3], hh[1], mm[1], ll[1]);
    g2_split_pair(v1[0], v1[1], hh[2], mm[2], ll[2]);
    g2_split_pair(v1[2], v1[3], hh[3], mm[3], ll[3]);
    h = g2_u32x4{hh[0], hh[1], hh[2], hh[3]};
    m = g2_u32x4{mm[0], mm[1], mm[2], mm[3]};
    l = g2_u32x4{ll[0], ll[1], ll[2], ll[3]};
}

// ACT: G2Args::act; RES: a residual is added; the epilogue is straight-line code per instantiation (with run-time switches its 64
// unrolled elements spill 70-90 registers)
// NB: 32-column blocks per wave -- 2: the workgroup covers 256 columns; 1: 128 columns (N = 128, 384, ..: short products, HBM-bound)
template <int ACT, bool RES, bool DROP, int NB = 2>
__global__ __launch_bounds__(256, 2) void gemm3x2_kernel(G2Args a) {
    extern __shared__ g2_u32x4 lds4[];                                  // two stage buffers [3 planes][8 pieces of 8 k][64 rows] x 16 B
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long ntiles = (a.M + G2_BM - 1) / G2_BM;
    long tile = blockIdx.x;
    if (tile >= ntiles) return;
    const int n0 = blockIdx.y * (128 * NB) + wave * (32 * NB);          // the wave's 32 * NB columns
    const int NT = a.N >> 5, nc64 = a.K >> 6;
    const uint16_t* wbase = a.Wz + ((long)(n0 >> 5) * 64 + lane) * 8;
    const long wplane = (long)NT * 512, wchunk = 3 * wplane;
    const f32x4 zf = {0.f, 0.f, 0.f, 0.f};

    float biasv[NB], addv[NB];
#pragma unroll
    for (int tn = 0; tn < NB; ++tn) {
        biasv[tn] = a.bias ? a.bias[n0 + tn * 32 + col] : 0.f;
        addv[tn] = a.addvec ? a.addvec[n0 + tn * 32 + col] : 0.f;
    }
    constexpr bool has_aux = ACT == 2 || ACT == 4;
    const float* pre_src = has_aux ? a.aux : (RES ? a.residual : nullptr);   // the epilogue's first read stream
    const int ldo4 = a.ldo * 4;
    const int lane_off = (4 * half * a.ldo + col) * 4;                  // byte offset of (row 4 * half, column col) inside a 32 x 32 block

    auto bload = [&](const uint16_t* src, g2_u32x4 (&b)[NB][3]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int tn = 0; tn < NB; ++tn) b[tn][p] = *reinterpret_cast<const g2_u32x4*>(src + p * wplane + tn * 512);
    };
    // ---- A staging: 64 rows x 8 pieces of 8 columns per stage = 2 pieces (2 float4 loads each) per thread
    f32x4 sa0 = zf, sb0 = zf, sa1 = zf, sb1 = zf;
    auto stage_load = [&](long mt, int c) __attribute__((always_inline)) {
        {
            const int row = tid >> 3, sh = tid & 7;
            const bool ok = mt + row < a.M;
            const float* src = a.A + (ok ? (mt + row) * a.lda + c * 64 + sh * 8 : 0);
            sa0 = *reinterpret_cast<const f32x4*>(src);
            sb0 = *reinterpret_cast<const f32x4*>(src + 4);
            if (!ok) sa0 = sb0 = zf;
        }
        {
            const int row = 32 + (tid >> 3), sh = tid & 7;
            const bool ok = mt + row < a.M;
            const float* src = a.A + (ok ? (mt + row) * a.lda + c * 64 + sh * 8 : 0);
            sa1 = *reinterpret_cast<const f32x4*>(src);
            sb1 = *reinterpret_cast<const f32x4*>(src + 4);
            if (!ok) sa1 = sb1 = zf;
        }
    };
    auto stage_store = [&](g2_u32x4* dst) __attribute__((always_inline)) {
        const int row = tid >> 3, sh = tid & 7;
        g2_u32x4 h, m, l;
        g2_split8(sa0, sb0, h, m, l);
        dst[(0 * 8 + sh) * G2_BM + row] = h;
        dst[(1 * 8 + sh) * G2_BM + row] = m;
        dst[(2 * 8 + sh) * G2_BM + row] = l;
        g2_split8(sa1, sb1, h, m, l);
        dst[(0 * 8 + sh) * G2_BM + 32 + row] = h;
        dst[(1 * 8 + sh) * G2_BM + 32 + row] = m;
        dst[(2 * 8 + sh) * G2_BM + 32 + row] = l;
    };

    g2_u32x4 bc[NB][3], bn[NB][3];                                          // W operands of the current / next 16-k chunk (chunks wrap: tile after tile)
    bload(wbase, bc);
    stage_load(tile * G2_BM, 0);
    stage_store(lds4);
    __syncthreads();
    int buf = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        const long m0 = tile * G2_BM;
        const long m0n = (tile + gridDim.x) * G2_BM;
        const bool next_tile = tile + gridDim.x < ntiles;
        long rows = a.M - m0;
        if (rows > G2_BM) rows = G2_BM;
        const unsigned rec = (unsigned)(((rows - 1) * a.ldo + 32 * NB) * 4);   // the wave's columns of the tile's valid rows
        const long tbase = m0 * a.ldo + n0;

        f32x16 acc[2][NB];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < NB; ++tn) acc[tm][tn] = zero16();
        for (int c = 0; c < nc64; ++c) {
            const bool last = c + 1 == nc64;
            const bool more = !last || next_tile;                       // a next stage exists (of this tile or of the next one)
            if (more) stage_load(last ? m0n : m0, last ? 0 : c + 1);
            const g2_u32x4* As = lds4 + buf * 24 * G2_BM;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bload(wbase + (long)((s == 3 && last) ? 0 : c * 4 + s + 1) * wchunk, bn);
                g2_u32x4 av[2][3];
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int p = 0; p < 3; ++p) av[tm][p] = As[((p * 4 + s) * 2 + half) * G2_BM + tm * 32 + col];
                __builtin_amdgcn_sched_barrier(0);                      // loads of chunk s + 1 stay here: hoisted over the unrolled chunks they spill
                // (A plane, W plane): lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi -- small terms first; four accumulators alternate
#define G2_MF(PA, PB)                                                               \
    acc[0][0] = g2_mfma(av[0][PA], bc[0][PB], acc[0][0]);                           \
    if (NB == 2) acc[0][NB - 1] = g2_mfma(av[0][PA], bc[NB - 1][PB], acc[0][NB - 1]); \
    acc[1][0] = g2_mfma(av[1][PA], bc[0][PB], acc[1][0]);                           \
    if (NB == 2) acc[1][NB - 1] = g2_mfma(av[1][PA], bc[NB - 1][PB], acc[1][NB - 1]);
                G2_MF(2, 0) G2_MF(0, 2) G2_MF(1, 1) G2_MF(1, 0) G2_MF(0, 1) G2_MF(0, 0)
#undef G2_MF
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tn = 0; tn < NB; ++tn)
#pragma unroll
                    for (int p = 0; p < 3; ++p) bc[tn][p] = bn[tn][p];
            }
            if (more) stage_store(lds4 + (buf ^ 1) * 24 * G2_BM);
            __syncthreads();
            buf ^= 1;
        }

        // ---- epilogue in accumulator layout: register r of block (tm, tn) = row tm*32 + 8*(r>>2) + 4*half + (r&3), column tn*32 + col
        const rsrc_t ro = make_rsrc(a.out + tbase, rec);
        const rsrc_t rpo = make_rsrc(a.pre_out ? a.pre_out + tbase : a.out + tbase, a.pre_out ? rec : 0u);
        const rsrc_t rr = make_rsrc((RES && has_aux) ? a.residual + tbase : a.out + tbase, (RES && has_aux) ? rec : 0u);
        // the epilogue's first read stream (aux, else residual), 32 rows at a time: 32 loads in flight per lane, their latency covered by
        // the CU's other workgroup (more live registers spill at two waves per SIMD: the matrix loop alone takes 251)
        f32x16 pv[NB];
        const rsrc_t rp = make_rsrc(pre_src ? pre_src + tbase : a.out + tbase, pre_src ? rec : 0u);
        auto pre_issue = [&](int tm, f32x16 (&pf)[NB]) __attribute__((always_inline)) {
#pragma unroll
            for (int tn = 0; tn < NB; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    pf[tn][r] = buf_load_f32(rp, lane_off + (tm * 32 + 8 * (r >> 2) + (r & 3)) * ldo4 + tn * 128, 0);
        };
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            if (has_aux || RES) pre_issue(tm, pv);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tn = 0; tn < NB; ++tn) {
                unsigned keepw[4];                                      // [owner j]: bit 4 k + c = keep (row r = 4 k + j, column 4 * (col / 4) + c)
                if (DROP) {
                    unsigned mine = 0;
                    const long e0 = tbase + (long)(tm * 32 + 4 * half + (lane & 3)) * a.ldo + tn * 32 + (col & ~3);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const f32x4 m4 = dropout4(a.drop, (unsigned long long)(e0 + (long)(8 * k) * a.ldo) >> 2);
                        mine |= ((m4[0] != 0.f ? 1u : 0u) | (m4[1] != 0.f ? 2u : 0u) | (m4[2] != 0.f ? 4u : 0u) | (m4[3] != 0.f ? 8u : 0u)) << (4 * k);
                    }
                    keepw[0] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0x00, 0xF, 0xF, false);
                    keepw[1] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0x55, 0xF, 0xF, false);
                    keepw[2] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0xAA, 0xF, 0xF, false);
                    keepw[3] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0xFF, 0xF, 0xF, false);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int off = lane_off + (tm * 32 + 8 * (r >> 2) + (r & 3)) * ldo4 + tn * 128;
                    float v = acc[tm][tn][r] + biasv[tn];
                    if (ACT == 1) {
                        buf_store_f32(v, rpo, off, 0);                  // dropped by the empty descriptor when there is no pre_out
                        v = gelu_f(v);
                    } else if (ACT == 2) {
                        v *= gelu_grad_f(pv[tn][r]);
                    } else if (ACT == 3) {
                        v = fmaxf(v, 0.f);
                    } else if (ACT == 4) {
                        v = pv[tn][r] > 0.f ? v : 0.f;
                    }
                    if (DROP) v = ((keepw[r & 3] >> (4 * (r >> 2) + (col & 3))) & 1u) ? v * a.drop.inv_keep : 0.f;
                    v += addv[tn];
                    if (RES) v += has_aux ? buf_load_f32(rr, off, 0) : pv[tn][r];
                    buf_store_f32(v, ro, off, 0);
                }
            }
        }
    }
}

bool rpb_gemm3x2_supported(long M, int N, int K, bool has_mask, bool has_drop) {
    static const int mode = getenv("RPB_GEMM3X_V2") ? atoi(getenv("RPB_GEMM3X_V2")) : 1;     // 0: off
    if (mode == 0 || has_mask || K % 64 != 0 || M <= 0) return false;
    return N % 256 == 0 || (N % 128 == 0 && !has_drop);                  // 128-column workgroups are built without dropout
}

int rpb_gemm3x2_launch(const G2Args& a, hipStream_t st) {
    const size_t lds = (size_t)2 * 24 * G2_BM * 16;
    const int nb = a.N % 256 == 0 ? 2 : 1, gy = a.N / (128 * nb);
    long gx = (a.M + G2_BM - 1) / G2_BM;
    const long cap = (long)rpb_num_cus() * 2 / gy;                       // two workgroups per CU in total
    if (gx > cap) gx = cap < 1 ? 1 : cap;
    const dim3 grid((unsigned)gx, gy);
#define G2_LAUNCH(ACT_, RES_, DROP_, NB_)                                                                                                   \
    if (a.act == ACT_ && (a.residual != nullptr) == RES_ && (a.drop.thr != 0) == DROP_ && nb == NB_) {                                      \
        (void)hipFuncSetAttribute((const void*)gemm3x2_kernel<ACT_, RES_, DROP_, NB_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((gemm3x2_kernel<ACT_, RES_, DROP_, NB_>), grid, dim3(256), lds, st, a);                                          \
    }
#define G2_LAUNCH2(ACT_)                                                                                            \
    G2_LAUNCH(ACT_, false, false, 2) G2_LAUNCH(ACT_, true, false, 2) G2_LAUNCH(ACT_, false, true, 2) G2_LAUNCH(ACT_, true, true, 2) \
    G2_LAUNCH(ACT_, false, false, 1) G2_LAUNCH(ACT_, true, false, 1)
    G2_LAUNCH2(0) G2_LAUNCH2(1) G2_LAUNCH2(2) G2_LAUNCH2(3) G2_LAUNCH2(4)
#undef G2_LAUNCH2
#undef G2_LAUNCH
    RPB_CHECK_LAUNCH("gemm3x (64-row tiles)");
}
