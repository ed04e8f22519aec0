// The whole backward of the projection head in ONE pass over the last layer's pre-BatchNorm tensor ("pjf"):
//
//   fno.py:121-125    u = fc1 a + b1 (64 -> 128),  v = gelu(u),  out = fc2 v + b2        a = crop(BN(s_{L-1}))   (no GELU after the
//   autograd          gh = (fc2^T gout) * gelu'(u)                                                               last layer, fno.py:118)
//                     g_a = gh fc1            scattered into the padded layout (zeros in the margin)
//                     d fc1 = gh^T a,  d b1 = sum gh,  d fc2 = gout^T v,  d b2 = sum gout
//                     BatchNorm-backward sums of the last Fourier layer:  sum g_a,  sum g_a * shat,   shat = (s - mean) * invstd
//
// Rounds 1-2 ran this as three launches chained through gu = gh in HBM ([ncrop][128] fp32 = 5.4 GB at B = 32; 30.8 GB of traffic for
// a head whose algorithmic bytes are 6.6 GB), because gh is needed in two orientations: lane = cell for the data gradient and
// lane = hidden unit for the weight gradient, and evaluating gelu' twice costs more than the round trip.
// Here a wave computes u = a W1^T for a 32-cell tile with CELLS as the matrix rows, so that
//
//   * the accumulators (lane = hidden unit, registers = 8 cells) ARE the A operand of the weight gradient  M = gh^T shat  (K = cells),
//     whose B operand -- shat with lane = channel -- is a plain channels-last load of the same tile (L1 / L2 hit);
//   * d fc2 / d b1 / d b2 are per-lane sums over the lane's cells (1 register per hidden tile instead of 4);
//   * only the data gradient needs gh transposed: a 16-cell x 128 row tile goes through the wave's own 8 KB LDS slice once
//     (32 ds_write_b32 + 8 ds_read_b128) and comes back as the A operand of  g = gh W1  (K = hidden units);
//   * the BatchNorm-backward sums need no pass at all:  sum_cells g[c] = sum_h W1[h][c] db1[h]  and
//     sum_cells g[c] shat[c] = sum_h W1[h][c] M[h][c]  with  M = gh^T shat, the matrix the kernel accumulates anyway
//     (d fc1 = gamma_c M + beta_c db1, exact algebra: a = gamma shat + beta) -- rpb_head_bwd_finalize does both on 128 x 64 numbers.
//
// All three contractions run on v_mfma_f32_16x16x32_bf16 from three-plane truncation splits (six products, fp32 accumulation:
// the fp32-grade arithmetic of rpb_cmx.hip / rpb_pjx.hip).  One wave owns a full 128 x 64 weight-gradient accumulator (128
// registers), so the kernel runs one wave per SIMD with the 512-register budget; HBM traffic is s (crop) + gout in, g out.
#include "rpb_common.h"
#include "rpb_pjf.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

#define PF_HID 128
#define PF_WAVES 4
#define PF_TS 132          // row stride (floats) of the wave's transposition tile: 4 * 132 = 16 (mod 32) -> conflict-free ds_write_b32
#define PF_DOMAX 4
// PF_PIPE=1 asks the scheduler (sched_group_barrier) to alternate each region's MFMAs with the neighbouring stage's VALU work.  Measured
// at B = 32: 4.27-4.31 ms with the forced interleaving, 4.00 ms without (the compiler's own order), 4.04 ms for the phase-by-phase
// version -- one wave per SIMD is bound by its issue slots (~4 cycles per instruction + 12 more per MFMA), not by missing overlap.
#ifndef PF_PIPE
#define PF_PIPE 0
#endif
#if PF_PIPE
#define PF_SGB(m, n, id) __builtin_amdgcn_sched_group_barrier(m, n, id)
#else
#define PF_SGB(m, n, id)
#endif

namespace {
__device__ __forceinline__ u32x4 ld16(rsrc_t r, int voff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
__device__ __forceinline__ void st16(f32x4v v, rsrc_t r, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, 0);
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
__device__ __forceinline__ f32x4v mfma16(bf16x8 a, bf16x8 b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// channel of contraction index (ks, kg, e) under the 4 x 16 B-per-cell load pattern (see rpb_cmx.hip)
__device__ __forceinline__ int chan_of(int ks, int kg, int e) { return 16 * (2 * ks + (e >> 2)) + 4 * kg + (e & 3); }
}  // namespace


// six products of the three-plane split, small terms first, NC independent accumulation chains advancing together
#define PF_MAC6(NC, ACC, AH, AM, AL, BH, BM, BL)                                       \
    _Pragma("unroll") for (int c_ = 0; c_ < NC; ++c_) ACC(c_) = mfma16(AH(c_), BL(c_), ACC(c_)); \
    _Pragma("unroll") for (int c_ = 0; c_ < NC; ++c_) ACC(c_) = mfma16(AL(c_), BH(c_), ACC(c_)); \
    _Pragma("unroll") for (int c_ = 0; c_ < NC; ++c_) ACC(c_) = mfma16(AM(c_), BM(c_), ACC(c_)); \
    _Pragma("unroll") for (int c_ = 0; c_ < NC; ++c_) ACC(c_) = mfma16(AH(c_), BM(c_), ACC(c_)); \
    _Pragma("unroll") for (int c_ = 0; c_ < NC; ++c_) ACC(c_) = mfma16(AM(c_), BH(c_), ACC(c_)); \
    _Pragma("unroll") for (int c_ = 0; c_ < NC; ++c_) ACC(c_) = mfma16(AH(c_), BH(c_), ACC(c_));

// DOT = register bound on the fc2 output features; EXACT: DO == DOT (vector loads of gout)
// LOSS: the head's FORWARD rides along (fused trainer): gout is not an input but scale * (fc2 gelu(u) + b2 - y), the squared error is
// summed per wave -- rpb_proj_fwd and rpb_mse disappear from the training step (u = fc1 a is recomputed here anyway)
template <int DOT, bool EXACT, bool LOSS>
__global__ __launch_bounds__(PF_WAVES * 64, 1) void pjf_kernel(PjfArgs p) {
    extern __shared__ u32x4 lds4[];
    u32x4* W1B = lds4;                                   // [ks 2][plane 3][t 8][lane]    B of u = a W1^T    (columns = hidden 16 t + n16)
    u32x4* W1D = W1B + 2 * 3 * 8 * 64;                   // [s 4][plane 3][u 4][lane]     B of g = gh W1     (column n16 of tile u = channel 4 n16 + u)
    float* xfl = reinterpret_cast<float*>(W1D + 4 * 3 * 4 * 64);          // [4][64] mean, invstd, gamma, beta
    float* b1l = xfl + 256;                              // [128]
    float* w2l = b1l + PF_HID;                           // [PF_DOMAX][128] (rows >= DO: zeros)
    float* Tall = w2l + PF_DOMAX * PF_HID;               // [waves][16][PF_TS] transposition tiles
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kg = lane >> 4;
    const int DO = p.DO;
    for (int idx = tid; idx < 2 * 8 * 64; idx += blockDim.x) {
        const int l = idx & 63, t = (idx >> 6) & 7, ks = idx >> 9;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p.w1[(16 * t + (l & 15)) * 64 + chan_of(ks, l >> 4, e)];
        bf16x8 h, m, lo;
        split8(v, h, m, lo);
        W1B[((ks * 3 + 0) * 8 + t) * 64 + l] = __builtin_bit_cast(u32x4, h);
        W1B[((ks * 3 + 1) * 8 + t) * 64 + l] = __builtin_bit_cast(u32x4, m);
        W1B[((ks * 3 + 2) * 8 + t) * 64 + l] = __builtin_bit_cast(u32x4, lo);
    }
    for (int idx = tid; idx < 4 * 4 * 64; idx += blockDim.x) {
        const int l = idx & 63, u = (idx >> 6) & 3, s = idx >> 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p.w1[(32 * s + 8 * (l >> 4) + e) * 64 + 4 * (l & 15) + u];
        bf16x8 h, m, lo;
        split8(v, h, m, lo);
        W1D[((s * 3 + 0) * 4 + u) * 64 + l] = __builtin_bit_cast(u32x4, h);
        W1D[((s * 3 + 1) * 4 + u) * 64 + l] = __builtin_bit_cast(u32x4, m);
        W1D[((s * 3 + 2) * 4 + u) * 64 + l] = __builtin_bit_cast(u32x4, lo);
    }
    for (int idx = tid; idx < PF_HID; idx += blockDim.x) b1l[idx] = p.b1[idx];
    for (int idx = tid; idx < PF_DOMAX * PF_HID; idx += blockDim.x) w2l[idx] = idx < DO * PF_HID ? p.w2[idx] : 0.f;
    if (tid < 64) {
        xfl[tid] = p.xf.mean[tid];
        xfl[64 + tid] = p.xf.invstd[tid];
        xfl[128 + tid] = p.xf.gamma[tid];
        xfl[192 + tid] = p.xf.beta[tid];
    }
    __syncthreads();

    const CropMap cm = p.cm;
    const long nslots = (long)gridDim.x * PF_WAVES;
    const long slot = (long)blockIdx.x * PF_WAVES + wave;
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
    float* Tw = Tall + wave * 16 * PF_TS;
    float dw2[DOT][8], db1[8], db2[DOT];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        db1[t] = 0.f;
#pragma unroll
        for (int j = 0; j < DOT; ++j) dw2[j][t] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < DOT; ++j) db2[j] = 0.f;
    float b2v[DOT], lacc = 0.f;
#pragma unroll
    for (int j = 0; j < DOT; ++j) b2v[j] = (LOSS && j < DO) ? p.b2[j] : 0.f;
    f32x4v acc3[8][4];                                   // M: [hidden tile t][channel tile u]; row 16 t + 4 mg + r, column = channel 4 n16 + u
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc3[t][u] = z4;

    auto line_of = [&](int gl) {                         // cropped line -> padded line (32-bit: B * Tp * Hp lines)
        const unsigned h = (unsigned)gl % (unsigned)cm.H, r2 = (unsigned)gl / (unsigned)cm.H;
        return (int)(((r2 / (unsigned)cm.T) * cm.Tp + r2 % (unsigned)cm.T) * cm.Hp + h);
    };
    u32x4 xa[2][4];                                      // A layout: cell 32 q + 16 j + n16, channels 16 i + 4 kg ..
    u32x4 xr[8];                                         // B layout: cell 32 q + 16 (e >> 2) + 4 kg + (e & 3), channels 4 n16 ..
    float go[8][DOT];                                    // gout of the lane's 8 cells (the B-layout cells = the accumulator rows)
    // (past the wave's last tile the descriptor is empty: the loads return 0 without touching memory -- no branches in the tile body)
    auto issue_xa = [&](int pl, int q) {
        const bool ok = pl >= 0;
        const rsrc_t rx = make_rsrc(p.s + (long)(ok ? pl : 0) * cm.Wp * 64, ok ? (unsigned)cm.W * 256u : 0u);   // cells >= W read as 0
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) xa[j][i] = ld16(rx, (32 * q + 16 * j + n16) * 256 + i * 64 + kg * 16);
    };
    auto issue_xr = [&](int pl, int q) {
        const rsrc_t rx = make_rsrc(p.s + (long)pl * cm.Wp * 64, (unsigned)cm.W * 256u);
#pragma unroll
        for (int e = 0; e < 8; ++e) xr[e] = ld16(rx, (32 * q + 16 * (e >> 2) + 4 * kg + (e & 3)) * 256 + n16 * 16);
    };
    auto issue_go = [&](int gl, int q) {                 // gl: CROPPED line, < 0 past the end
        const bool ok = gl >= 0;
        const rsrc_t rg = make_rsrc(p.gout + (long)(ok ? gl : 0) * cm.W * DO, ok ? (unsigned)(cm.W * DO) * 4u : 0u);       // cells >= W read as 0
        if (DOT == 2 && EXACT) {                         // the lane's cells 16 j + 4 kg .. + 3 are 32 contiguous bytes of gout
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4v w0 = __builtin_bit_cast(f32x4v, ld16(rg, (32 * q + 16 * j + 4 * kg) * 8));
                const f32x4v w1 = __builtin_bit_cast(f32x4v, ld16(rg, (32 * q + 16 * j + 4 * kg) * 8 + 16));
                go[4 * j][0] = w0[0], go[4 * j][1] = w0[1], go[4 * j + 1][0] = w0[2], go[4 * j + 1][1] = w0[3];
                go[4 * j + 2][0] = w1[0], go[4 * j + 2][1] = w1[1], go[4 * j + 3][0] = w1[2], go[4 * j + 3][1] = w1[3];
            }
            return;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int cell = 32 * q + 16 * (e >> 2) + 4 * kg + (e & 3);
            if (DOT == 4 && EXACT) {
                const f32x4v w = __builtin_bit_cast(f32x4v, ld16(rg, cell * 16));
#pragma unroll
                for (int j = 0; j < 4; ++j) go[e][j] = w[j];
            } else {
#pragma unroll
                for (int j = 0; j < DOT; ++j) go[e][j] = j < DO ? buf_load_f32(rg, (cell * DO + j) * 4, 0) : 0.f;
            }
        }
    };
    // prefetch = ONE dword per lane, 128 B apart: the 64 lines of the next tile's 8 KB are pulled into L2 a whole tile ahead at the price
    // of one register; the real operand-layout loads are issued late (registers are the scarce resource here) and hit L2
    auto prefetch = [&](int pl, int q) -> unsigned {
        const bool ok = pl >= 0;
        const rsrc_t rx = make_rsrc(p.s + (long)(ok ? pl : 0) * cm.Wp * 64, ok ? (unsigned)cm.W * 256u : 0u);
        return __builtin_amdgcn_raw_buffer_load_b32(rx, 32 * q * 256 + lane * 128, 0, 0);
    };
    {
        const bool ok0 = slot < GL;
        issue_xa(ok0 ? line_of((int)slot) : -1, 0);
        issue_go(ok0 ? (int)slot : -1, 0);
    }
    for (long gl = slot; gl < GL; gl += nslots) {
        const int pl = line_of((int)gl);                                              // this line and the wave's next one, decoded once per line
        const int gln = gl + nslots < GL ? (int)(gl + nslots) : -1;
        const int pln = gln >= 0 ? line_of(gln) : -1;
        const rsrc_t ro = make_rsrc(p.g + (long)pl * cm.Wp * 64, line_bytes);
        for (int q = 0; q < TQ; ++q) {
            asm volatile("" ::: "memory");
            const bool last = q + 1 == TQ;
            const int gn = last ? gln : (int)gl, pn = last ? pln : pl;                // next tile: cropped / padded line, tile index
            const int qn = last ? 0 : q + 1;
            const unsigned pf = prefetch(pn, qn);
            // The tile body is written as scheduling REGIONS (sched_barrier(0) between them).  One wave per SIMD issues in order, so
            // matrix and vector work only overlap when they alternate in the instruction stream: each region pairs the MFMAs of one
            // stage with independent VALU work of the neighbouring stage and asks the scheduler for that interleaving
            // (sched_group_barrier).  Measured before: 35 % matrix pipe busy, VALU and MFMA co-executing 3 % of the time.
            f32x4v acc[2][8];
            bf16x8 Ah[2][2], Am[2][2], Al[2][2];                     // [ks][j]: a in A-operand layout, split
            auto split_A = [&](int ks) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float v[8];
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int i = 2 * ks + hf;
                        const f32x4v xv = __builtin_bit_cast(f32x4v, xa[j][i]);
                        const f32x4v mu = *reinterpret_cast<const f32x4v*>(xfl + 16 * i + 4 * kg);
                        const f32x4v is = *reinterpret_cast<const f32x4v*>(xfl + 64 + 16 * i + 4 * kg);
                        const f32x4v ga = *reinterpret_cast<const f32x4v*>(xfl + 128 + 16 * i + 4 * kg);
                        const f32x4v be = *reinterpret_cast<const f32x4v*>(xfl + 192 + 16 * i + 4 * kg);
                        const f32x4v z = bn4(xv, mu, is, ga, be);
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[4 * hf + c] = z[c];
                    }
                    split8(v, Ah[ks][j], Am[ks][j], Al[ks][j]);
                }
            };
            // products of K-step ks for the hidden-tile pair (t0, t0 + 1), both row tiles: 24 MFMAs on 4 independent accumulators
            auto mac_pair = [&](int ks, int t0) {
                bf16x8 bh[2], bm[2], bl[2];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    bh[tt] = __builtin_bit_cast(bf16x8, W1B[((ks * 3 + 0) * 8 + t0 + tt) * 64 + lane]);
                    bm[tt] = __builtin_bit_cast(bf16x8, W1B[((ks * 3 + 1) * 8 + t0 + tt) * 64 + lane]);
                    bl[tt] = __builtin_bit_cast(bf16x8, W1B[((ks * 3 + 2) * 8 + t0 + tt) * 64 + lane]);
                }
#define PF_ACC(c) acc[(c) & 1][t0 + ((c) >> 1)]
#define PF_AH(c) Ah[ks][(c) & 1]
#define PF_AM(c) Am[ks][(c) & 1]
#define PF_AL(c) Al[ks][(c) & 1]
#define PF_BH(c) bh[(c) >> 1]
#define PF_BM(c) bm[(c) >> 1]
#define PF_BL(c) bl[(c) >> 1]
                PF_MAC6(4, PF_ACC, PF_AH, PF_AM, PF_AL, PF_BH, PF_BM, PF_BL)
#undef PF_ACC
#undef PF_AH
#undef PF_AM
#undef PF_AL
#undef PF_BH
#undef PF_BM
#undef PF_BL
            };
            // gh = (fc2^T gout) * gelu'(u) in place for hidden tile t of row tile j; d fc2, d b1: per-lane sums over the lane's cells
            auto act = [&](int j, int t) {
                const f32x4v u = acc[j][t];                          // rows = cells 16 j + 4 kg + r  <->  go[4 j + r]
                f32x2 e0, e1;
                        fast_erf2x2(u.lo * pk2(0.70710678118654752440f), u.hi * pk2(0.70710678118654752440f), e0, e1);
                const f32x4v cdf = join4(pk2(0.5f) * (pk2(1.0f) + e0), pk2(0.5f) * (pk2(1.0f) + e1));
                const f32x4v q2 = (f32x4v{-0.72134752044448170368f, -0.72134752044448170368f, -0.72134752044448170368f, -0.72134752044448170368f} * u) * u;
                f32x4v ex;
#pragma unroll
                for (int r = 0; r < 4; ++r) ex[r] = __builtin_amdgcn_exp2f(q2[r]);
                const f32x4v vv = u * cdf;
                const f32x4v dd = cdf + u * (ex * 0.39894228040143267794f);
                f32x4v gp = z4;
#pragma unroll
                for (int jj = 0; jj < DOT; ++jj) {
                    const f32x4v gv = {go[4 * j][jj], go[4 * j + 1][jj], go[4 * j + 2][jj], go[4 * j + 3][jj]};
                    gp += gv * w2l[jj * PF_HID + 16 * t + n16];
                    const f32x4v pr = gv * vv;
                    dw2[jj][t] += (pr[0] + pr[1]) + (pr[2] + pr[3]);
                }
                const f32x4v gh = gp * dd;                           // cells >= W: gout == 0 -> gh == 0
                acc[j][t] = gh;
                db1[t] += (gh[0] + gh[1]) + (gh[2] + gh[3]);
            };
            // ---- region 1: split of the first K-step (needs this tile's A-layout loads); the B-layout view of THIS tile starts its way
            if (!LOSS) issue_xr(pl, q);                              // L1 / L2 hits: the lines were fetched by the xa loads
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const float bv = b1l[16 * t + n16];
                    acc[j][t] = f32x4v{bv, bv, bv, bv};
                }
            split_A(0);
            __builtin_amdgcn_sched_barrier(0);
            // ---- region 2: u = a W1^T, first K-step (96 MFMAs)  ||  split of the second K-step
#pragma unroll
            for (int t0 = 0; t0 < 8; t0 += 2) mac_pair(0, t0);
            split_A(1);
#pragma unroll
            for (int i = 0; i < 96; ++i) {
                PF_SGB(0x008, 1, 0);
                PF_SGB(0x002, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (LOSS) {
                // ---- region 3 (fused forward): all second-K-step products, then one 16-cell row tile at a time:
                //      v = gelu(u), out = fc2 v + b2 (per-lane partial over the lane's hidden units, summed over the 16 lanes of the
                //      group with DPP), gout = gscale (out - y), then gh / d fc2 / d b1 as in the plain kernel
#pragma unroll
                for (int t0 = 0; t0 < 8; t0 += 2) mac_pair(1, t0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (j == 1) issue_xr(pl, q);                                      // (held back: registers) in flight during the second row tile
                    f32x4v VV[8];
                    float po[4][DOT];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int jj = 0; jj < DOT; ++jj) po[r][jj] = 0.f;
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const f32x4v u = acc[j][t];
                        f32x2 e0, e1;
                        fast_erf2x2(u.lo * pk2(0.70710678118654752440f), u.hi * pk2(0.70710678118654752440f), e0, e1);
                        const f32x4v cdf = join4(pk2(0.5f) * (pk2(1.0f) + e0), pk2(0.5f) * (pk2(1.0f) + e1));
                        const f32x4v q2 = (f32x4v{-0.72134752044448170368f, -0.72134752044448170368f, -0.72134752044448170368f, -0.72134752044448170368f} * u) * u;
                        f32x4v ex;
#pragma unroll
                        for (int r = 0; r < 4; ++r) ex[r] = __builtin_amdgcn_exp2f(q2[r]);
                        VV[t] = u * cdf;
                        acc[j][t] = cdf + u * (ex * 0.39894228040143267794f);          // gelu'(u), until gh replaces it below
#pragma unroll
                        for (int jj = 0; jj < DOT; ++jj) {
                            const float w = w2l[jj * PF_HID + 16 * t + n16];
#pragma unroll
                            for (int r = 0; r < 4; ++r) po[r][jj] = fmaf(VV[t][r], w, po[r][jj]);
                        }
                        if (t & 1) __builtin_amdgcn_sched_barrier(0);                  // two hidden tiles in flight: bounds the temporaries
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool valid = 32 * q + 16 * j + 4 * kg + r < cm.W;       // cells past the line end: no output element there
#pragma unroll
                        for (int jj = 0; jj < DOT; ++jj) {
                            float v = po[r][jj];                                      // sum over the 16 lanes n16 of the lane group
                            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
                            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
                            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
                            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
                            const float diff = (v + b2v[jj]) - go[4 * j + r][jj];     // go holds the target y here
                            lacc += (valid && n16 == 0 && jj < DO) ? diff * diff : 0.f;
                            go[4 * j + r][jj] = (valid && jj < DO) ? p.gscale * diff : 0.f;
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        f32x4v gp = z4;
#pragma unroll
                        for (int jj = 0; jj < DOT; ++jj) {
                            const f32x4v gv = {go[4 * j][jj], go[4 * j + 1][jj], go[4 * j + 2][jj], go[4 * j + 3][jj]};
                            gp += gv * w2l[jj * PF_HID + 16 * t + n16];
                            const f32x4v pr = gv * VV[t];
                            dw2[jj][t] += (pr[0] + pr[1]) + (pr[2] + pr[3]);
                        }
                        const f32x4v gh = gp * acc[j][t];
                        acc[j][t] = gh;
                        db1[t] += (gh[0] + gh[1]) + (gh[2] + gh[3]);
                        if (t & 1) __builtin_amdgcn_sched_barrier(0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // ---- region 3: second K-step of pair p (24 MFMAs)  ||  activation of pair p - 1 (whose accumulators are final)
    #pragma unroll
                for (int pp = 0; pp < 5; ++pp) {
                    if (pp < 4) mac_pair(1, 2 * pp);
                    if (pp > 0) {
    #pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            act(j, 2 * pp - 2);
                            act(j, 2 * pp - 1);
                        }
                    }
                    if (pp > 0 && pp < 4) {
    #pragma unroll
                        for (int i = 0; i < 24; ++i) {
                            PF_SGB(0x008, 1, 0);
                            PF_SGB(0x002, 11, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int jj = 0; jj < DOT; ++jj)
#pragma unroll
                for (int e = 0; e < 8; ++e) db2[jj] += go[e][jj];
            issue_go(gn, qn);
            // ---- region 4: M += gh^T shat: contraction over the 32 cells; lane group kg holds cells {4 kg + r, 16 + 4 kg + r}
            {
                bf16x8 Xh[4], Xm[4], Xl[4];                          // shat in B-operand layout: column n16 of tile u = channel 4 n16 + u
                const f32x4v bmu = *reinterpret_cast<const f32x4v*>(xfl + 4 * n16), bis = *reinterpret_cast<const f32x4v*>(xfl + 64 + 4 * n16);
                f32x4v sh[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) sh[e] = (__builtin_bit_cast(f32x4v, xr[e]) - bmu) * bis;     // rows of cells >= W: gh == 0 there
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = sh[e][u];
                    split8(v, Xh[u], Xm[u], Xl[u]);
                }
                bf16x8 Gh[2], Gm[2], Gl[2];                          // gh^T of hidden tile t in A-operand layout, double-buffered
                auto split_G = [&](int t) {
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = acc[0][t][r];
                        v[4 + r] = acc[1][t][r];
                    }
                    split8(v, Gh[t & 1], Gm[t & 1], Gl[t & 1]);
                };
                split_G(0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 8; ++t) {
#define PF_ACC(c) acc3[t][c]
#define PF_G1(c) Gh[t & 1]
#define PF_G2(c) Gm[t & 1]
#define PF_G3(c) Gl[t & 1]
#define PF_X1(c) Xh[c]
#define PF_X2(c) Xm[c]
#define PF_X3(c) Xl[c]
                    PF_MAC6(4, PF_ACC, PF_G1, PF_G2, PF_G3, PF_X1, PF_X2, PF_X3)
#undef PF_ACC
#undef PF_G1
#undef PF_G2
#undef PF_G3
#undef PF_X1
#undef PF_X2
#undef PF_X3
                    if (t < 7) {                                     // the next hidden tile's split rides between this tile's MFMAs
                        split_G(t + 1);
#pragma unroll
                        for (int i = 0; i < 22; ++i) {
                            PF_SGB(0x008, 1, 0);
                            PF_SGB(0x002, 2, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            asm volatile("" ::"v"(pf));
            issue_xa(pn, qn);                                        // the next tile's A-layout loads (prefetched into L2 a tile ago)
            // ---- region 5: g = gh W1, one 16-cell row tile at a time through the wave's transposition tile
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Tw[(4 * kg + r) * PF_TS + 16 * t + n16] = acc[j][t][r];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                f32x4v acc2[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) acc2[u] = z4;
                bf16x8 Ah2[2], Am2[2], Al2[2];
                auto split_T = [&](int s) {
                    const f32x4v t0 = *reinterpret_cast<const f32x4v*>(Tw + n16 * PF_TS + 32 * s + 8 * kg);
                    const f32x4v t1 = *reinterpret_cast<const f32x4v*>(Tw + n16 * PF_TS + 32 * s + 8 * kg + 4);
                    float v[8];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        v[c] = t0[c];
                        v[4 + c] = t1[c];
                    }
                    split8(v, Ah2[s & 1], Am2[s & 1], Al2[s & 1]);
                };
                split_T(0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    bf16x8 bh[4], bm[4], bl[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        bh[u] = __builtin_bit_cast(bf16x8, W1D[((s * 3 + 0) * 4 + u) * 64 + lane]);
                        bm[u] = __builtin_bit_cast(bf16x8, W1D[((s * 3 + 1) * 4 + u) * 64 + lane]);
                        bl[u] = __builtin_bit_cast(bf16x8, W1D[((s * 3 + 2) * 4 + u) * 64 + lane]);
                    }
#define PF_ACC(c) acc2[c]
#define PF_A1(c) Ah2[s & 1]
#define PF_A2(c) Am2[s & 1]
#define PF_A3(c) Al2[s & 1]
#define PF_B1(c) bh[c]
#define PF_B2(c) bm[c]
#define PF_B3(c) bl[c]
                    PF_MAC6(4, PF_ACC, PF_A1, PF_A2, PF_A3, PF_B1, PF_B2, PF_B3)
#undef PF_ACC
#undef PF_A1
#undef PF_A2
#undef PF_A3
#undef PF_B1
#undef PF_B2
#undef PF_B3
                    if (s < 3) {
                        split_T(s + 1);
#pragma unroll
                        for (int i = 0; i < 22; ++i) {
                            PF_SGB(0x008, 1, 0);
                            PF_SGB(0x002, 2, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();                     // the tile is rewritten by the next row tile
                // row 4 mg + r of the row tile = cell 32 q + 16 j + 4 kg + r; the lane's 4 column tiles are channels 4 n16 .. 4 n16 + 3
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const f32x4v o = {acc2[0][r], acc2[1][r], acc2[2][r], acc2[3][r]};
                    st16(o, ro, (32 * q + 16 * j + 4 * kg + r) * 256 + n16 * 16);      // cells >= Wp: dropped; W .. Wp-1: zeros (gh == 0)
                }
            }
        }
        // margin cells 32 TQ .. Wp - 1 of the line (cells W .. 32 TQ - 1 were written as zeros by the last tile)
        for (int off = TQ * 32 * 256 + lane * 16; off < (int)line_bytes; off += 1024) st16(z4, ro, off);
    }

    if (LOSS) {
        const float ls = wave_sum(lacc);
        if (lane == 0) p.loss_part[slot] = ls;
    }
    // ---- the wave's partial row
    float* part = p.part + slot * ((long)PF_HID * 64 + (long)DO * PF_HID + PF_HID + DO);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x4v o = {acc3[t][0][r], acc3[t][1][r], acc3[t][2][r], acc3[t][3][r]};
            *reinterpret_cast<f32x4v*>(part + (16 * t + 4 * kg + r) * 64 + 4 * n16) = o;
        }
    // per-lane sums over cells: the four lane groups hold different cells of the same hidden units -> add over kg
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        float s1 = db1[t];
        s1 += __shfl_xor(s1, 16, 64);
        s1 += __shfl_xor(s1, 32, 64);
        if (kg == 0) part[PF_HID * 64 + DO * PF_HID + 16 * t + n16] = s1;
#pragma unroll
        for (int j = 0; j < DOT; ++j) {
            if (j < DO) {
                float s2 = dw2[j][t];
                s2 += __shfl_xor(s2, 16, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (kg == 0) part[PF_HID * 64 + j * PF_HID + 16 * t + n16] = s2;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < DOT; ++j) {
        if (j < DO) {
            float s3 = db2[j];                       // identical on the 16 lanes of a group; groups hold different cells
            s3 += __shfl_xor(s3, 16, 64);
            s3 += __shfl_xor(s3, 32, 64);
            if (lane == 0) part[PF_HID * 64 + DO * PF_HID + PF_HID + j] = s3;
        }
    }
}

static size_t pjf_lds() { return (size_t)(2 * 3 * 8 * 64 + 4 * 3 * 4 * 64) * 16 + (256 + PF_HID + PF_DOMAX * PF_HID) * 4 + (size_t)PF_WAVES * 16 * PF_TS * 4; }

static bool pjf_off() {
    static const bool off = (getenv("RPB_PROJ_F32") && atoi(getenv("RPB_PROJ_F32")) == 1) ||
                            (getenv("RPB_HEAD_BWD_FUSED") && atoi(getenv("RPB_HEAD_BWD_FUSED")) == 0);
    return off;
}

// gelu head (act = 0) after a BatchNorm without GELU, width 64, fc2 out features <= 4
extern "C" int rpb_head_bwd_supported(int C, int DO, int W, int Wp, int xf_gelu, int act) {
    return !pjf_off() && C == 64 && DO >= 1 && DO <= PF_DOMAX && W >= 1 && Wp >= W && xf_gelu == 0 && act == 0;
}

extern "C" long rpb_head_bwd_slots(int B, int T, int H) {
    const long GL = (long)B * T * H;
    long grid = rpb_num_cus();                       // 512-register kernel, 133 KB of LDS: one workgroup of four waves per CU
    const long need = (GL + PF_WAVES - 1) / PF_WAVES;
    if (grid > need) grid = need;
    if (grid < 1) grid = 1;
    return grid * PF_WAVES;
}

extern "C" int rpb_head_bwd_row(int DO) { return PF_HID * 64 + DO * PF_HID + PF_HID + DO; }

// g [ncell][64] = crop-scatter(gh fc1) with gh = (fc2^T gout) * gelu'(fc1 a + b1), a = gamma * shat + beta, shat = (s - mean) * invstd on
// the cropped cells;  part [rpb_head_bwd_slots][rpb_head_bwd_row(DO)] = per-wave partial sums (M = gh^T shat | d fc2.weight | d fc1.bias |
// d fc2.bias): reduce over rows, then rpb_head_bwd_finalize
static int pjf_launch(PjfArgs& p, bool loss, hipStream_t st) {
    const int grid = (int)(rpb_head_bwd_slots(p.B, p.cm.T, p.cm.H) / PF_WAVES);
    if (pjg_supported(p.DO)) return pjg_launch(p, loss, grid, st);       // fc2 width 2: the 32x32x16 organisation (rpb_pjg.hip)
    const size_t lds = pjf_lds();
    const int DO = p.DO;
#define RPB_PJF(D_, E_, L_)                                                                                                   \
    if (loss == L_) {                                                                                                        \
        (void)hipFuncSetAttribute((const void*)pjf_kernel<D_, E_, L_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((pjf_kernel<D_, E_, L_>), dim3(grid), dim3(PF_WAVES * 64), lds, st, p);                            \
    }
#define RPB_PJF2(D_, E_) RPB_PJF(D_, E_, false) RPB_PJF(D_, E_, true)
    if (DO == 2) { RPB_PJF2(2, true) }
    else if (DO == 1) { RPB_PJF2(2, false) }
    else if (DO == 4) { RPB_PJF2(4, true) }
    else { RPB_PJF2(4, false) }
#undef RPB_PJF2
#undef RPB_PJF
    RPB_CHECK_LAUNCH("head_bwd");
}

extern "C" int rpb_head_bwd(const float* s, const float* w1, const float* b1, const float* w2, const float* gout, float* g,
                            float* part, int B, int DO, int T, int H, int W, int Tp, int Hp, int Wp, const float* xf_mean,
                            const float* xf_invstd, const float* xf_gamma, const float* xf_beta, void* stream) {
    RPB_REQUIRE(s && w1 && b1 && w2 && gout && g && part && xf_mean && xf_invstd && xf_gamma && xf_beta, "head_bwd: null pointer");
    RPB_REQUIRE(rpb_head_bwd_supported(64, DO, W, Wp, 0, 0), "head_bwd: unsupported shape (DO=%d W=%d Wp=%d)", DO, W, Wp);
    RPB_REQUIRE((long)Wp * 256 < (1L << 31), "head_bwd: line too long");
    PjfArgs p{};
    p.s = s; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.gout = gout; p.g = g; p.part = part; p.B = B; p.DO = DO;
    p.cm = CropMap{T, H, W, Tp, Hp, Wp};
    p.xf = XForm{xf_mean, xf_invstd, xf_gamma, xf_beta, 0};
    return pjf_launch(p, false, (hipStream_t)stream);
}

// The training step's head in ONE launch (fused trainer): forward out = fc2 gelu(fc1 a + b1) + b2 on the cropped cells, the squared-error
// loss against `target` [ncrop][DO] (loss_part [rpb_head_bwd_slots] = per-wave sums of (out - target)^2), dLoss/dout = gscale (out - target)
// and the whole backward of rpb_head_bwd from it.  Replaces rpb_proj_fwd + rpb_mse + rpb_head_bwd (fno.py:121-125, utils/metrics.py:11-13,
// train.py:328-329): fc1 is recomputed by the backward anyway.
extern "C" int rpb_head_fwd_bwd(const float* s, const float* w1, const float* b1, const float* w2, const float* b2, const float* target,
                                float gscale, float* g, float* part, float* loss_part, int B, int DO, int T, int H, int W, int Tp, int Hp,
                                int Wp, const float* xf_mean, const float* xf_invstd, const float* xf_gamma, const float* xf_beta,
                                void* stream) {
    RPB_REQUIRE(s && w1 && b1 && w2 && b2 && target && g && part && loss_part && xf_mean && xf_invstd && xf_gamma && xf_beta, "head_fwd_bwd: null pointer");
    RPB_REQUIRE(rpb_head_bwd_supported(64, DO, W, Wp, 0, 0), "head_fwd_bwd: unsupported shape (DO=%d W=%d Wp=%d)", DO, W, Wp);
    RPB_REQUIRE((long)Wp * 256 < (1L << 31), "head_fwd_bwd: line too long");
    PjfArgs p{};
    p.s = s; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.gout = target; p.b2 = b2; p.gscale = gscale; p.loss_part = loss_part; p.g = g; p.part = part;
    p.B = B; p.DO = DO;
    p.cm = CropMap{T, H, W, Tp, Hp, Wp};
    p.xf = XForm{xf_mean, xf_invstd, xf_gamma, xf_beta, 0};
    return pjf_launch(p, true, (hipStream_t)stream);
}

// tot [rpb_head_bwd_row(DO)] = the reduced partial row.  Writes d fc1.weight [128][64] = gamma_c M + beta_c db1, d fc2.weight [DO][128],
// d fc1.bias [128], d fc2.bias [DO] and the BatchNorm-backward sums of the last layer bn_sums [2][64] =
// (sum_cells g[c], sum_cells g[c] shat[c]) = (sum_h W1[h][c] db1[h], sum_h W1[h][c] M[h][c]).
__global__ __launch_bounds__(256) void pjf_finalize_kernel(const float* __restrict__ tot, const float* __restrict__ w1,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, int DO,
                                                            float* __restrict__ dw1, float* __restrict__ dw2, float* __restrict__ db1,
                                                            float* __restrict__ db2, float* __restrict__ bn_sums) {
    __shared__ double red[2][4][64];
    const int tid = threadIdx.x;
    const float* M = tot;
    const float* tw2 = tot + PF_HID * 64;
    const float* tb1 = tw2 + DO * PF_HID;
    const float* tb2 = tb1 + PF_HID;
    for (int idx = tid; idx < PF_HID * 64; idx += 256) {
        const int h = idx >> 6, c = idx & 63;
        dw1[idx] = fmaf(gamma[c], M[idx], beta[c] * tb1[h]);
    }
    for (int idx = tid; idx < DO * PF_HID; idx += 256) dw2[idx] = tw2[idx];
    for (int idx = tid; idx < PF_HID; idx += 256) db1[idx] = tb1[idx];
    for (int idx = tid; idx < DO; idx += 256) db2[idx] = tb2[idx];
    const int c = tid & 63, part = tid >> 6;
    double a = 0.0, q = 0.0;
    for (int h = part; h < PF_HID; h += 4) {
        const double w = (double)w1[h * 64 + c];
        a += w * (double)tb1[h];
        q += w * (double)M[h * 64 + c];
    }
    red[0][part][c] = a;
    red[1][part][c] = q;
    __syncthreads();
    if (tid < 128) {
        const int k = tid >> 6, cc = tid & 63;
        bn_sums[k * 64 + cc] = (float)((red[k][0][cc] + red[k][1][cc]) + (red[k][2][cc] + red[k][3][cc]));
    }
}

extern "C" int rpb_head_bwd_finalize(const float* tot, const float* w1, const float* gamma, const float* beta, int DO, float* dw1,
                                     float* dw2, float* db1, float* db2, float* bn_sums, void* stream) {
    RPB_REQUIRE(tot && w1 && gamma && beta && dw1 && dw2 && db1 && db2 && bn_sums && DO >= 1 && DO <= PF_DOMAX, "head_bwd_finalize: bad arguments");
    hipLaunchKernelGGL(pjf_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, tot, w1, gamma, beta, DO, dw1, dw2, db1, db2, bn_sums);
    RPB_CHECK_LAUNCH("head_bwd_finalize");
}
