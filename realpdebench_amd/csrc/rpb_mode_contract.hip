// K3: per-mode complex channel contraction of SpectralConv3d (reference fno.py:41-43, 53-60):
//
//   fwd    Y[b,o,m]  = sum_i X[b,i,m] * W[i,o,m]
//   dgrad  gX[b,i,m] = sum_o gY[b,o,m] * conj(W[i,o,m])
//   wgrad  gW[i,o,m] = sum_b conj(X[b,i,m]) * gY[b,o,m]
//
// MI355X layout (not the reference's): spectra are planar and mode-major-within-batch
//   X, Y : [B][2 (re,im)][M][C]          (what the truncated-DFT stages produce / consume)
// and the weights are stored MODE-MAJOR as interleaved complex
//   W    : [M][Ci][Co][2]                (one contiguous Ci*Co*8 B tile per retained mode)
// so one workgroup owns one mode, streams its 32 KB (C=64) weight tile exactly once with 512 B
// coalesced rows, and keeps the [B][2][C] coefficient tile of that mode in LDS (broadcast reads).
// At B=32 this kernel is a pure weight stream: 100.7 MB per layer against ~50 MB of coefficients.
#include "rpb_common.h"
#include <stdlib.h>

#define MC_THREADS 256
#define MC_BT 8   // batch entries accumulated per thread per pass

// ---- forward and dgrad share the structure "out[b][n] = sum_k in[b][k] (*) W(k,n)"
template <bool DGRAD>
__global__ __launch_bounds__(MC_THREADS) void mode_contract_kernel(const float* __restrict__ X,
                                                                   const float* __restrict__ Wt,
                                                                   float* __restrict__ Y, int B, int M, int C) {
    extern __shared__ float lds[];
    const int m = blockIdx.x;
    float* Xl = lds;                       // [B][2][C]
    float* Wl = lds + (long)B * 2 * C;     // dgrad only: [Ci][Co+1] complex (padded: conflict-free b64 column reads)
    const long plane = (long)M * C;
    for (int idx = threadIdx.x; idx < B * 2 * C; idx += blockDim.x) {
        const int c = idx % C, r = idx / C;            // r = b*2 + ri
        Xl[idx] = X[(long)r * plane + (long)m * C + c];
    }
    const float* Wm = Wt + (long)m * C * C * 2;
    if (DGRAD) {
        for (int idx = threadIdx.x; idx < C * C; idx += blockDim.x) {
            const int i = idx / C, o = idx - i * C;
            const f32x2 w = *reinterpret_cast<const f32x2*>(Wm + (long)idx * 2);
            *reinterpret_cast<f32x2*>(Wl + ((long)i * (C + 1) + o) * 2) = w;
        }
    }
    __syncthreads();

    const int n = threadIdx.x % C;          // output channel owned by this thread
    const int bg = threadIdx.x / C;
    const int nbg = blockDim.x / C;
    for (int b0 = bg * MC_BT; b0 < B; b0 += nbg * MC_BT) {
        float ar[MC_BT], ai[MC_BT];
#pragma unroll
        for (int j = 0; j < MC_BT; ++j) ar[j] = ai[j] = 0.f;
        for (int k = 0; k < C; ++k) {
            float wr, wi;
            if (DGRAD) {                    // n = i, k = o : conj(W[i][o])
                const f32x2 w = *reinterpret_cast<const f32x2*>(Wl + ((long)n * (C + 1) + k) * 2);
                wr = w[0];
                wi = -w[1];
            } else {                        // n = o, k = i : W[i][o], 512 B coalesced row
                const f32x2 w = *reinterpret_cast<const f32x2*>(Wm + ((long)k * C + n) * 2);
                wr = w[0];
                wi = w[1];
            }
#pragma unroll
            for (int j = 0; j < MC_BT; ++j) {
                const int b = b0 + j;
                if (b < B) {
                    const float xr = Xl[(b * 2 + 0) * C + k];
                    const float xi = Xl[(b * 2 + 1) * C + k];
                    ar[j] += xr * wr - xi * wi;
                    ai[j] += xr * wi + xi * wr;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < MC_BT; ++j) {
            const int b = b0 + j;
            if (b < B) {
                Y[(long)(b * 2 + 0) * plane + (long)m * C + n] = ar[j];
                Y[(long)(b * 2 + 1) * plane + (long)m * C + n] = ai[j];
            }
        }
    }
}

// ---- dgrad, one workgroup per (mode, 32 input channels): gX[b,i] = sum_o gY[b,o] conj(W[i,o]) walks ROWS of the mode's weight
// tile, so the 32 rows a workgroup needs are one contiguous 32*C*8 B block (coalesced), transposed into a padded LDS tile of
// 33 KB (C = 128) instead of the whole 132 KB tile: three to four workgroups per CU overlap their weight streams with each
// other's arithmetic, where the one-workgroup-per-mode version above ran one per CU with load and compute back to back
// (C = 128: 2.26 ms = 0.37 TB/s for a 671 MB weight stream).
#define MD_ROWS 32
__global__ __launch_bounds__(MC_THREADS) void mode_dgrad_kernel(const float* __restrict__ GY, const float* __restrict__ Wt,
                                                                float* __restrict__ GX, int B, int M, int C) {
    extern __shared__ float lds[];
    const int nchunk = C / MD_ROWS;
    const int m = blockIdx.x / nchunk, ch = blockIdx.x - m * nchunk;
    float* Xl = lds;                        // [B][2][C]   gY of this mode
    float* Wl = lds + (long)B * 2 * C;      // [32][C + 1] complex
    const long plane = (long)M * C;
    for (int idx = threadIdx.x; idx < B * 2 * (C / 4); idx += MC_THREADS) {
        const int c4 = idx % (C / 4), r = idx / (C / 4);
        *reinterpret_cast<f32x4*>(Xl + r * C + 4 * c4) = *reinterpret_cast<const f32x4*>(GY + (long)r * plane + (long)m * C + 4 * c4);
    }
    const float* Wm = Wt + ((long)m * C + ch * MD_ROWS) * C * 2;
    for (int idx = threadIdx.x; idx < MD_ROWS * C / 2; idx += MC_THREADS) {      // two complex numbers per load
        const int i = (2 * idx) / C, o = 2 * idx - i * C;
        const f32x4 w = *reinterpret_cast<const f32x4*>(Wm + (long)idx * 4);
        float* d = Wl + ((long)i * (C + 1) + o) * 2;
        const f32x2 w0 = {w[0], w[1]}, w1 = {w[2], w[3]};
        *reinterpret_cast<f32x2*>(d) = w0;
        *reinterpret_cast<f32x2*>(d + 2) = w1;
    }
    __syncthreads();
    const int il = threadIdx.x & 31, bg = threadIdx.x >> 5;                     // 8 batch groups
    const float* wrow = Wl + (long)il * (C + 1) * 2;
    for (int b0 = bg; b0 < B; b0 += 32) {
        float ar[4], ai[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) ar[j] = ai[j] = 0.f;
        for (int k = 0; k < C; k += 4) {
            f32x2 w[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) w[t] = *reinterpret_cast<const f32x2*>(wrow + (k + t) * 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int b = b0 + 8 * j;
                if (b < B) {
                    const f32x4 xr = *reinterpret_cast<const f32x4*>(Xl + (b * 2 + 0) * C + k);
                    const f32x4 xi = *reinterpret_cast<const f32x4*>(Xl + (b * 2 + 1) * C + k);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {                                   // conj(W): (wr, -wi)
                        ar[j] += xr[t] * w[t][0] + xi[t] * w[t][1];
                        ai[j] += xi[t] * w[t][0] - xr[t] * w[t][1];
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int b = b0 + 8 * j;
            if (b < B) {
                GX[(long)(b * 2 + 0) * plane + (long)m * C + ch * MD_ROWS + il] = ar[j];
                GX[(long)(b * 2 + 1) * plane + (long)m * C + ch * MD_ROWS + il] = ai[j];
            }
        }
    }
}

// ---- wgrad: gW[m][i][o] = sum_b conj(X[b][i]) * gY[b][o]
__global__ __launch_bounds__(MC_THREADS) void mode_wgrad_kernel(const float* __restrict__ X,
                                                                const float* __restrict__ GY,
                                                                float* __restrict__ GW, int B, int M, int C,
                                                                int accumulate) {
    extern __shared__ float lds[];
    const int m = blockIdx.x;
    float* Xl = lds;
    float* Gl = lds + (long)B * 2 * C;
    const long plane = (long)M * C;
    for (int idx = threadIdx.x; idx < B * 2 * C; idx += blockDim.x) {
        const int c = idx % C, r = idx / C;
        const long off = (long)r * plane + (long)m * C + c;
        Xl[idx] = X[off];
        Gl[idx] = GY[off];
    }
    __syncthreads();
    const int o = threadIdx.x % C;
    const int ig = threadIdx.x / C;
    const int nig = blockDim.x / C;
    float* Gm = GW + (long)m * C * C * 2;
    for (int i = ig; i < C; i += nig) {
        float ar = 0.f, ai = 0.f;
        for (int b = 0; b < B; ++b) {
            const float xr = Xl[(b * 2 + 0) * C + i], xi = Xl[(b * 2 + 1) * C + i];
            const float gr = Gl[(b * 2 + 0) * C + o], gi = Gl[(b * 2 + 1) * C + o];
            ar += xr * gr + xi * gi;        // conj(x) * g
            ai += xr * gi - xi * gr;
        }
        f32x2* dst = reinterpret_cast<f32x2*>(Gm + ((long)i * C + o) * 2);
        if (accumulate) {
            const f32x2 old = *dst;
            ar += old[0];
            ai += old[1];
        }
        f32x2 v = {ar, ai};
        *dst = v;
    }
}


// ---- C = 64 on the fp32 matrix pipe: per mode the complex contraction is a real GEMM with the composite [[wr, wi], [-wi, wr]]
// (forward: [B x 128] x [128 x 128]; dgrad: the conjugate transpose; wgrad: [128 x 2B]^T products with K = batch).  One workgroup per
// mode, 4 waves x 64 MFMAs of 32x32x2; the mode's coefficient tile(s) and its weight tile are staged in LDS planar and padded, so the
// operand reads of every variant (batch-major, channel-major, weight rows or weight columns) are conflict-free.  The VALU kernels
// above ran at ~20 TF/s (16 LDS reads per 32 FMAs); they remain the path for other widths.
//   MODE 0 fwd, 1 dgrad, 2 wgrad.  Batches larger than 32 are walked in passes of 32 rows.
#define MM_LD 65
template <int MODE>
__global__ __launch_bounds__(256) void mode_mfma_kernel(const float* __restrict__ X, const float* __restrict__ Wt, const float* __restrict__ GY,
                                                         float* __restrict__ OUT, int B, int M, int accumulate) {
    constexpr int C = 64;
    __shared__ float Xs[2 * 32 * MM_LD];          // [ri][b][c]      fwd: X, dgrad: gY, wgrad: X
    __shared__ float Ws[2 * 64 * MM_LD];          // fwd / dgrad: [ri][i][o] weights;  wgrad: [ri][b][o] = gY (first 2*32 rows)
    const int m = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    const long plane = (long)M * C;
    const float* src = MODE == 1 ? GY : X;
    if (MODE != 2) {
        const float* Wm = Wt + (long)m * C * C * 2;
        for (int idx = tid; idx < C * C / 2; idx += 256) {       // two complex numbers per 16 B load
            const f32x4 w = *reinterpret_cast<const f32x4*>(Wm + (long)idx * 4);
            const int i = (2 * idx) / C, o = 2 * idx - i * C;
            Ws[(0 * 64 + i) * MM_LD + o] = w[0];
            Ws[(1 * 64 + i) * MM_LD + o] = w[1];
            Ws[(0 * 64 + i) * MM_LD + o + 1] = w[2];
            Ws[(1 * 64 + i) * MM_LD + o + 1] = w[3];
        }
    }
    f32x16 accA = zero16(), accB = zero16();                     // wgrad: (re, im) of the wave's 32 x 32 tile, summed over the passes
    for (int b0 = 0; b0 < B; b0 += 32) {
        if (b0) __syncthreads();
        for (int idx = tid; idx < 2 * 32 * (C / 4); idx += 256) {
            const int c4 = idx % (C / 4), r = idx / (C / 4);     // r = bl * 2 + ri
            const int bl = r >> 1, ri = r & 1, b = b0 + bl;
            f32x4 v = {0.f, 0.f, 0.f, 0.f}, g = {0.f, 0.f, 0.f, 0.f};
            if (b < B) {
                v = *reinterpret_cast<const f32x4*>(src + (long)(b * 2 + ri) * plane + (long)m * C + 4 * c4);
                if (MODE == 2) g = *reinterpret_cast<const f32x4*>(GY + (long)(b * 2 + ri) * plane + (long)m * C + 4 * c4);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                Xs[(ri * 32 + bl) * MM_LD + 4 * c4 + t] = v[t];
                if (MODE == 2) Ws[(ri * 32 + bl) * MM_LD + 4 * c4 + t] = g[t];
            }
        }
        __syncthreads();
        if (MODE != 2) {
            // out[b][(ro, n)] for the wave's (ro = wave >> 1, 32 columns n0 ..): K = (ri, k) over both input planes
            const int ro = wave >> 1, n0 = (wave & 1) * 32;
            f32x16 acc = zero16();
#pragma unroll
            for (int ri = 0; ri < 2; ++ri) {
                // fwd:   Yr = Xr wr - Xi wi, Yi = Xr wi + Xi wr          -> weight plane (ri ^ ro), sign - for (ri, ro) = (1, 0)
                // dgrad: gXr = gYr wr + gYi wi, gXi = -gYr wi + gYi wr   -> weight plane (ri ^ ro), sign - for (ri, ro) = (0, 1)
                const int wp = ri ^ ro;
                const bool neg = MODE == 0 ? (ri == 1 && ro == 0) : (ri == 0 && ro == 1);
                const float* xa = Xs + (ri * 32 + col) * MM_LD + half;
                // fwd: B[k][n] = W[i = k][o = n0 + col];  dgrad: B[k][n] = W[i = n0 + col][o = k]
                const float* wb = MODE == 0 ? Ws + (wp * 64 + half) * MM_LD + n0 + col : Ws + (wp * 64 + n0 + col) * MM_LD + half;
                constexpr int kstep = MODE == 0 ? 2 * MM_LD : 2;
#pragma unroll 8
                for (int s = 0; s < 32; ++s) {
                    const float a = xa[2 * s];
                    acc = mfma32(neg ? -a : a, wb[s * kstep], acc);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int b = b0 + mfma_row(lane, r);
                if (b < B) OUT[(long)(b * 2 + ro) * plane + (long)m * C + n0 + col] = acc[r];
            }
        } else {
            // gW[i][o]: re = sum_b Xr gYr + Xi gYi, im = sum_b Xr gYi - Xi gYr;  wave = (i tile, o tile), K = the pass's 32 batch rows
            const int i0 = (wave >> 1) * 32, o0 = (wave & 1) * 32;
#pragma unroll 4
            for (int s = 0; s < 16; ++s) {
                const int bl = 2 * s + half;
                const float xr = Xs[(0 * 32 + bl) * MM_LD + i0 + col], xi = Xs[(1 * 32 + bl) * MM_LD + i0 + col];
                const float gr = Ws[(0 * 32 + bl) * MM_LD + o0 + col], gi = Ws[(1 * 32 + bl) * MM_LD + o0 + col];
                accA = mfma32(xr, gr, accA);
                accA = mfma32(xi, gi, accA);
                accB = mfma32(xr, gi, accB);
                accB = mfma32(-xi, gr, accB);
            }
        }
    }
    if (MODE == 2) {
        const int i0 = (wave >> 1) * 32, o0 = (wave & 1) * 32;
        float* Gm = OUT + (long)m * C * C * 2;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            f32x2* dst = reinterpret_cast<f32x2*>(Gm + ((long)(i0 + mfma_row(lane, r)) * C + o0 + col) * 2);
            f32x2 v = {accA[r], accB[r]};
            if (accumulate) v += *dst;
            *dst = v;
        }
    }
}

// ---- C = 64 on the bf16 matrix pipe from split fp32 operands (round 4; the fp32-grade arithmetic of rpb_cmx.hip: three truncation
// planes per operand, six products, fp32 accumulation).  The fp32-MFMA kernel above needs 128 matrix instructions of 64 cycles per wave
// and mode (8.2 k cycles) for a tile whose HBM time is ~2.5 k cycles: 0.06-0.095 ms per launch for 0.2 GB.  Here the same composite
// GEMMs take 96 instructions of 16 cycles per wave; the operands are gathered from the fp32 tiles staged in LDS straight into MFMA
// operand order (8 consecutive values of the contraction index per lane: 2 x ds_read_b128 where that index is contiguous in the tile,
// 8 x ds_read_b32 where it is the row index) and split in registers.
//   fwd    D[b][(ro,o)] = sum_(ri,i) X[b][(ri,i)] * s W[i][o][ri ^ ro]         s = -1 for (ri, ro) = (1, 0)      M = 32, N = 128, K = 128
//   dgrad  D[b][(ri,i)] = sum_(ro,o) gY[b][(ro,o)] * s W[i][o][ro ^ ri]        s = -1 for (ro, ri) = (0, 1)
//   wgrad  D[i][(part,o)] = sum_(ri,b) X[b][ri][i] * G(part)[(ri,b)][o]        G(re) = (gYr, gYi), G(im) = (gYi, -gYr)   M = 64, N = 128, K = 64
typedef __attribute__((ext_vector_type(8))) __bf16 mbf16x8;
typedef unsigned mu32x4 __attribute__((ext_vector_type(4)));
namespace {
__device__ __forceinline__ float m_trunc(float v) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & 0xffff0000u); }
__device__ __forceinline__ unsigned m_pack(float a, float b) {
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
struct Planes {
    mbf16x8 h, m, l;
};
__device__ __forceinline__ Planes m_split(const float (&v)[8]) {
    mu32x4 uh, um, ul;
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
    return Planes{__builtin_bit_cast(mbf16x8, uh), __builtin_bit_cast(mbf16x8, um), __builtin_bit_cast(mbf16x8, ul)};
}
__device__ __forceinline__ Planes m_neg(Planes p) {            // -x: flip the sign of every plane (exact)
    const mu32x4 sgn = {0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};
    return Planes{__builtin_bit_cast(mbf16x8, __builtin_bit_cast(mu32x4, p.h) ^ sgn), __builtin_bit_cast(mbf16x8, __builtin_bit_cast(mu32x4, p.m) ^ sgn),
                  __builtin_bit_cast(mbf16x8, __builtin_bit_cast(mu32x4, p.l) ^ sgn)};
}
__device__ __forceinline__ f32x4 m_mac6(const Planes& a, const Planes& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.l, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.l, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, c, 0, 0, 0);
    return c;
}
}  // namespace
// MB_NT: non-temporal loads of the mode's weight tile (A/B: profiles/r05_kbench_valu_variants.txt)
#ifndef MB_NT
#define MB_NT 0
#endif
#define MB_XP 68        // row pitch (floats) of the coefficient tiles [ri][b 32][c 64]
#define MB_WP 133       // row pitch (floats) of the weight tile [i 64][o 64][2]: odd, so that neither the row-strided gather of the forward
                        // (8 rows per lane group: 8 * 133 = 8 mod 32) nor the column-strided one of the data gradient piles onto a few banks
// PERSIST (B <= 32, the default there): a workgroup walks modes m = blockIdx.x, + gridDim.x, ... and requests the NEXT mode's weight tile (8 x 16 B
// per thread) and coefficient rows (4, wgrad 8) into registers before it computes the current one from LDS.  One workgroup per mode left
// every global load's latency exposed -- the launch ran at 1-2 TB/s of a 100 MB weight read (0.095 ms cold, 0.046 warm, against 0.02 at the
// copy rate) with three workgroups of 51 KB per CU and nothing in flight while they computed.
template <int MODE, bool PERSIST = false>
__global__ __launch_bounds__(256) void mode_bf16_kernel(const float* __restrict__ X, const float* __restrict__ Wt, const float* __restrict__ GY,
                                                         float* __restrict__ OUT, int B, int M, int accumulate) {
    constexpr int C = 64;
    __shared__ __attribute__((aligned(16))) float Xs[2 * 32 * MB_XP];                      // fwd / wgrad: X, dgrad: gY
    __shared__ __attribute__((aligned(16))) float Ws[MODE == 2 ? 2 * 32 * MB_XP : 64 * MB_WP];   // fwd / dgrad: the mode's weight tile; wgrad: gY
    int m = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kg = lane >> 4;
    const long plane = (long)M * C;
    const float* src = MODE == 1 ? GY : X;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 wq[PERSIST && MODE != 2 ? 8 : 1], xq[PERSIST ? 4 : 1], gq[PERSIST && MODE == 2 ? 4 : 1];      // PERSIST: the next mode's tiles in flight
    auto fetch = [&](int mm) {                                   // PERSIST: issue the loads of mode mm (B <= 32: one batch pass)
        if (MODE != 2) {
            const float* Wm = Wt + (long)mm * C * C * 2;
#pragma unroll
            for (int j = 0; j < 8; ++j) wq[PERSIST && MODE != 2 ? j : 0] = *reinterpret_cast<const f32x4*>(Wm + (long)(tid + 256 * j) * 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = tid + 256 * j, c4 = idx % (C / 4), r = idx / (C / 4), bl = r >> 1, ri = r & 1;
            f32x4 v = z4, g = z4;
            if (bl < B) {
                v = *reinterpret_cast<const f32x4*>(src + (long)(bl * 2 + ri) * plane + (long)mm * C + 4 * c4);
                if (MODE == 2) g = *reinterpret_cast<const f32x4*>(GY + (long)(bl * 2 + ri) * plane + (long)mm * C + 4 * c4);
            }
            xq[PERSIST ? j : 0] = v;
            if (MODE == 2) gq[PERSIST && MODE == 2 ? j : 0] = g;
        }
    };
    auto park = [&]() {                                          // PERSIST: registers -> LDS (waits for the loads)
        if (MODE != 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = tid + 256 * j, i = (2 * idx) / C, o = 2 * idx - i * C;
                const f32x4 w = wq[PERSIST && MODE != 2 ? j : 0];
                float* d = Ws + i * MB_WP + 2 * o;
                d[0] = w[0], d[1] = w[1], d[2] = w[2], d[3] = w[3];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = tid + 256 * j, c4 = idx % (C / 4), r = idx / (C / 4), bl = r >> 1, ri = r & 1;
            *reinterpret_cast<f32x4*>(Xs + (ri * 32 + bl) * MB_XP + 4 * c4) = xq[PERSIST ? j : 0];
            if (MODE == 2) *reinterpret_cast<f32x4*>(Ws + (ri * 32 + bl) * MB_XP + 4 * c4) = gq[PERSIST && MODE == 2 ? j : 0];
        }
    };
    if (PERSIST) fetch(m);
  for (; m < M; m += (int)gridDim.x) {                           // (one pass without PERSIST: gridDim.x == M)
    if (!PERSIST && MODE != 2) {
        const float* Wm = Wt + (long)m * C * C * 2;
        for (int idx = tid; idx < C * C / 2; idx += 256) {       // two complex numbers per 16 B load: row i, columns o, o + 1
#if MB_NT
            const f32x4 w = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Wm + (long)idx * 4));      // streamed once per launch by ONE workgroup
#else
            const f32x4 w = *reinterpret_cast<const f32x4*>(Wm + (long)idx * 4);
#endif
            const int i = (2 * idx) / C, o = 2 * idx - i * C;
            float* d = Ws + i * MB_WP + 2 * o;
            d[0] = w[0], d[1] = w[1], d[2] = w[2], d[3] = w[3];
        }
    }
    f32x4 accg[4][2];                                            // wgrad: [i tile][part] of the wave's o tile, summed over the batch passes
    if (PERSIST) {
        park();
        __syncthreads();
        if (m + (int)gridDim.x < M) fetch(m + (int)gridDim.x);   // in flight during everything below
    }
    if (MODE == 2) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) accg[rt][0] = accg[rt][1] = z4;
    }
    for (int b0 = 0; b0 < B; b0 += 32) {
        if (b0) __syncthreads();
        for (int idx = tid; idx < (PERSIST ? 0 : 2 * 32 * (C / 4)); idx += 256) {
            const int c4 = idx % (C / 4), r = idx / (C / 4);     // r = bl * 2 + ri
            const int bl = r >> 1, ri = r & 1, b = b0 + bl;
            f32x4 v = z4, g = z4;
            if (b < B) {
                v = *reinterpret_cast<const f32x4*>(src + (long)(b * 2 + ri) * plane + (long)m * C + 4 * c4);
                if (MODE == 2) g = *reinterpret_cast<const f32x4*>(GY + (long)(b * 2 + ri) * plane + (long)m * C + 4 * c4);
            }
            *reinterpret_cast<f32x4*>(Xs + (ri * 32 + bl) * MB_XP + 4 * c4) = v;
            if (MODE == 2) *reinterpret_cast<f32x4*>(Ws + (ri * 32 + bl) * MB_XP + 4 * c4) = g;
        }
        if (!PERSIST) __syncthreads();
        if (MODE != 2) {
            // wave w: output columns n = 32 w .. 32 w + 31 of (plane, channel): plane po = w >> 1, channels 32 (w & 1) + 16 ct + n16
            const int po = wave >> 1;
            f32x4 acc[2][2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) acc[rt][0] = acc[rt][1] = z4;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int pi = ks >> 1, k0 = 32 * (ks & 1) + 8 * kg;       // input plane, first of the lane's 8 contraction channels
                Planes A[2], Bp[2];
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {                            // A: row b = 16 rt + n16, 8 consecutive channels
                    const float* xp = Xs + (pi * 32 + 16 * rt + n16) * MB_XP + k0;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(xp), v1 = *reinterpret_cast<const f32x4*>(xp + 4);
                    const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    A[rt] = m_split(v);
                }
                const int wp = pi ^ po;                                     // real / imaginary part of the weight that this (pi, po) pair uses
                const bool neg = MODE == 0 ? (pi == 1 && po == 0) : (pi == 0 && po == 1);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const int cn = 32 * (wave & 1) + 16 * ct + n16;         // output channel (fwd: o, dgrad: i)
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e)                             // fwd: W[i = k][o = cn];  dgrad: W[i = cn][o = k]
                        v[e] = MODE == 0 ? Ws[(k0 + e) * MB_WP + 2 * cn + wp] : Ws[cn * MB_WP + 2 * (k0 + e) + wp];
                    Bp[ct] = m_split(v);
                    if (neg) Bp[ct] = m_neg(Bp[ct]);
                }
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = m_mac6(A[rt], Bp[ct], acc[rt][ct]);
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int b = b0 + 16 * rt + 4 * kg + r;
                        if (b < B) OUT[(long)(b * 2 + po) * plane + (long)m * C + 32 * (wave & 1) + 16 * ct + n16] = acc[rt][ct][r];
                    }
        } else {
            // wave w: output columns o = 16 w + n16, both parts; rows i = 16 rt + ...; K = (ri, b): one K-step per input plane
#pragma unroll
            for (int ri = 0; ri < 2; ++ri) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = Ws[(ri * 32 + 8 * kg + e) * MB_XP + 16 * wave + n16];      // gY[b = 8 kg + e][ri][o]
                const Planes G = m_split(v);
                // part re: + X_ri^T gY_ri ;  part im: ri = 0 -> + Xr^T gYi, ri = 1 -> - Xi^T gYr: the planes of gY_ri feed (re, ri) and (im, 1 - ri)
                const Planes Gn = m_neg(G);
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    float a[8], a2[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        a[e] = Xs[(ri * 32 + 8 * kg + e) * MB_XP + 16 * rt + n16];               // X[b][ri][i]
                        a2[e] = Xs[((1 - ri) * 32 + 8 * kg + e) * MB_XP + 16 * rt + n16];        // X[b][1 - ri][i]
                    }
                    const Planes A = m_split(a), A2 = m_split(a2);
                    accg[rt][0] = m_mac6(A, G, accg[rt][0]);                                     // re += X_ri^T gY_ri
                    accg[rt][1] = m_mac6(A2, ri == 0 ? Gn : G, accg[rt][1]);                     // im += Xr^T gYi (ri = 1) - Xi^T gYr (ri = 0)
                }
            }
        }
    }
    if (MODE == 2) {
        float* Gm = OUT + (long)m * C * C * 2;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x2* dst = reinterpret_cast<f32x2*>(Gm + ((long)(16 * rt + 4 * kg + r) * C + 16 * wave + n16) * 2);
                f32x2 v = {accg[rt][0][r], accg[rt][1][r]};
                if (accumulate) v += *dst;
                *dst = v;
            }
    }
    if (PERSIST) __syncthreads();                                // every wave is done with this mode's tiles before the next ones are parked
  }
}

// resident workgroups per CU of the persistent instances (registers: 184 / 180 for fwd / dgrad -> two waves per SIMD; 288 for wgrad -> one)
static int mode_persist_wgs(int mode) {
    static const int f = getenv("RPB_MODE_PERSIST_WGS") ? atoi(getenv("RPB_MODE_PERSIST_WGS")) : 0;
    return f > 0 ? f : (mode == 2 ? 1 : 2);
}
// Measured (B = 32, profiles/r06b_mode_persist.txt, tools/mode_cold_probe.py): the persistent instances run 0.054 -> 0.048 ms (fwd) and
// 0.050 -> 0.044 ms (dgrad) with warm weights and 0.055 -> 0.047 / 0.055 -> 0.049 with COLD ones (six 100 MB weight buffers cycled through the
// 256 MB Infinity Cache).  (A first reading of "0.124 ms persistent against 0.095 one-shot inside the step" came from HIP events around 50 us
// launches: that is the host's launch gap, not device time -- rocprofv3's kernel trace of the step gives 57 us one-shot, 45 us persistent.)
// So: forward and dgrad persistent (RPB_MODE_PERSIST_FWD=0 / RPB_MODE_PERSIST=0: one workgroup per mode), wgrad never (288 registers: one
// workgroup per CU, 0.044 -> 0.046).
static bool mode_persist_fwd() {
    static const bool on = !(getenv("RPB_MODE_PERSIST_FWD") && atoi(getenv("RPB_MODE_PERSIST_FWD")) == 0);
    return on;
}
static bool mode_persist_wgrad() {
    static const bool on = getenv("RPB_MODE_PERSIST_WGRAD") && atoi(getenv("RPB_MODE_PERSIST_WGRAD")) == 1;
    return on;
}
static bool mode_persist_on() {
    static const bool on = !(getenv("RPB_MODE_PERSIST") && atoi(getenv("RPB_MODE_PERSIST")) == 0);
    return on;
}
// ---- C = 128 (configs/fsi/fno.yaml, the Galerkin regressor) on the same pipe: the composite per-mode GEMM is [B x 256] x [256 x 256];
// the weight tile (128 KB) does not fit next to the coefficients, so the workgroup walks the 64-wide output halves (fwd: o halves with
// all 128 rows i of the weights; dgrad: i halves with all 128 columns o), re-staging the weight half in LDS (66 KB) each time.  wgrad
// keeps sixteen 32 x 32 (re, im) tiles in registers, four per wave.  The VALU kernels above ran these at 18-25 TF/s (10.5 of the
// fsi step's 65 ms, tools/fsi_probe.py).
#define MM_LD128 129
template <int MODE>
__global__ __launch_bounds__(256) void mode_mfma128_kernel(const float* __restrict__ X, const float* __restrict__ Wt, const float* __restrict__ GY,
                                                            float* __restrict__ OUT, int B, int M, int accumulate) {
    constexpr int C = 128;
    extern __shared__ float lds[];
    float* Xs = lds;                               // [ri][b 32][c 128 (+1)]   fwd: X, dgrad: gY, wgrad: X
    float* Ws = lds + 2 * 32 * MM_LD128;           // fwd: [ri][i 128][o-half 64 (+1)];  dgrad: [ri][i-half 64][o 128 (+1)];  wgrad: [ri][b 32][o 128 (+1)] = gY
    const int m = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    const long plane = (long)M * C;
    const float* src = MODE == 1 ? GY : X;
    const float* Wm = Wt + (long)m * C * C * 2;
    f32x16 accA[4], accB[4];                       // wgrad: tile q = 4 * wave + j -> (i0, o0) = (32 (q >> 2), 32 (q & 3))
    if (MODE == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) accA[j] = accB[j] = zero16();
    }
    for (int b0 = 0; b0 < B; b0 += 32) {
        if (b0) __syncthreads();
        for (int idx = tid; idx < 2 * 32 * (C / 4); idx += 256) {
            const int c4 = idx % (C / 4), r = idx / (C / 4);     // r = bl * 2 + ri
            const int bl = r >> 1, ri = r & 1, b = b0 + bl;
            f32x4 v = {0.f, 0.f, 0.f, 0.f}, g = {0.f, 0.f, 0.f, 0.f};
            if (b < B) {
                v = *reinterpret_cast<const f32x4*>(src + (long)(b * 2 + ri) * plane + (long)m * C + 4 * c4);
                if (MODE == 2) g = *reinterpret_cast<const f32x4*>(GY + (long)(b * 2 + ri) * plane + (long)m * C + 4 * c4);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                Xs[(ri * 32 + bl) * MM_LD128 + 4 * c4 + t] = v[t];
                if (MODE == 2) Ws[(ri * 32 + bl) * MM_LD128 + 4 * c4 + t] = g[t];
            }
        }
        if (MODE == 2) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = 4 * wave + j, i0 = 32 * (q >> 2), o0 = 32 * (q & 3);
#pragma unroll 4
                for (int s = 0; s < 16; ++s) {
                    const int bl = 2 * s + half;
                    const float xr = Xs[(0 * 32 + bl) * MM_LD128 + i0 + col], xi = Xs[(1 * 32 + bl) * MM_LD128 + i0 + col];
                    const float gr = Ws[(0 * 32 + bl) * MM_LD128 + o0 + col], gi = Ws[(1 * 32 + bl) * MM_LD128 + o0 + col];
                    accA[j] = mfma32(xr, gr, accA[j]);
                    accA[j] = mfma32(xi, gi, accA[j]);
                    accB[j] = mfma32(xr, gi, accB[j]);
                    accB[j] = mfma32(-xi, gr, accB[j]);
                }
            }
            continue;
        }
        for (int hf = 0; hf < 2; ++hf) {           // output half: columns (fwd: o, dgrad: i) 64 hf .. 64 hf + 63
            __syncthreads();                       // Xs staged (first half) / the previous half's weight reads are done
            if (MODE == 0) {                       // Ws[ri][i][o - 64 hf], 65 floats per row
                for (int idx = tid; idx < C * 32; idx += 256) {          // 128 rows x 32 pairs of complex numbers
                    const int i = idx >> 5, p2 = idx & 31, o = 64 * hf + 2 * p2;
                    const f32x4 w = *reinterpret_cast<const f32x4*>(Wm + ((long)i * C + o) * 2);
                    Ws[(0 * C + i) * MM_LD + 2 * p2] = w[0];
                    Ws[(1 * C + i) * MM_LD + 2 * p2] = w[1];
                    Ws[(0 * C + i) * MM_LD + 2 * p2 + 1] = w[2];
                    Ws[(1 * C + i) * MM_LD + 2 * p2 + 1] = w[3];
                }
            } else {                               // Ws[ri][i - 64 hf][o], 129 floats per row
                for (int idx = tid; idx < 64 * 64; idx += 256) {         // 64 rows x 64 pairs of complex numbers
                    const int il = idx >> 6, p2 = idx & 63, o = 2 * p2;
                    const f32x4 w = *reinterpret_cast<const f32x4*>(Wm + ((long)(64 * hf + il) * C + o) * 2);
                    Ws[(0 * 64 + il) * MM_LD128 + o] = w[0];
                    Ws[(1 * 64 + il) * MM_LD128 + o] = w[1];
                    Ws[(0 * 64 + il) * MM_LD128 + o + 1] = w[2];
                    Ws[(1 * 64 + il) * MM_LD128 + o + 1] = w[3];
                }
            }
            __syncthreads();
            const int ro = wave >> 1, n0 = (wave & 1) * 32;              // the wave's 32 columns of plane ro of this half
            f32x16 acc = zero16();
#pragma unroll
            for (int ri = 0; ri < 2; ++ri) {
                const int wp = ri ^ ro;
                const bool neg = MODE == 0 ? (ri == 1 && ro == 0) : (ri == 0 && ro == 1);
                const float* xa = Xs + (ri * 32 + col) * MM_LD128 + half;
                // fwd: B[k][n] = W[i = k][o = n0 + col] (row pitch 65);  dgrad: B[k][n] = W[i = n0 + col][o = k] (row pitch 129)
                const float* wb = MODE == 0 ? Ws + (wp * C + half) * MM_LD + n0 + col : Ws + (wp * 64 + n0 + col) * MM_LD128 + half;
                constexpr int kstep = MODE == 0 ? 2 * MM_LD : 2;
#pragma unroll 8
                for (int s = 0; s < C / 2; ++s) {
                    const float a = xa[2 * s];
                    acc = mfma32(neg ? -a : a, wb[s * kstep], acc);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int b = b0 + mfma_row(lane, r);
                if (b < B) OUT[(long)(b * 2 + ro) * plane + (long)m * C + 64 * hf + n0 + col] = acc[r];
            }
        }
    }
    if (MODE == 2) {
        float* Gm = OUT + (long)m * C * C * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = 4 * wave + j, i0 = 32 * (q >> 2), o0 = 32 * (q & 3);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                f32x2* dst = reinterpret_cast<f32x2*>(Gm + ((long)(i0 + mfma_row(lane, r)) * C + o0 + col) * 2);
                f32x2 v = {accA[j][r], accB[j][r]};
                if (accumulate) v += *dst;
                *dst = v;
            }
        }
    }
}

// ---- C = 128, second organisation of fwd / dgrad (round 5): the weights never pass through LDS.  The first organisation staged a 64-wide
// half of the mode's 128 KB weight tile in LDS with four ds_write_b32 per 16 B load and ran ONE workgroup per CU (99 KB of LDS): 0.56 ms per
// launch at the fsi shape for 0.54 GB of weights (1.4 TB/s) -- neither the matrix pipe (0.13 ms) nor the bytes (0.1 ms) but the staging.
// Here a wave owns 32 output columns of BOTH planes and loads its B operands straight from global memory in MFMA layout: forward, lane
// (col, half) reads the complex number W[i = 2 s + half][o = n0 + col] (8 B; 256 B contiguous per row across the lanes) and feeds four
// products (re / im of both output planes); data gradient, the lane walks ITS row W[i = n0 + col][:] in 16 B pieces (two complex numbers =
// two K-steps).  Only the coefficient tile (33 KB) is staged, so four workgroups share a CU and hide each other's load latency.
template <int MODE>
__global__ __launch_bounds__(256) void mode_mfma128d_kernel(const float* __restrict__ X, const float* __restrict__ Wt, const float* __restrict__ GY,
                                                             float* __restrict__ OUT, int B, int M) {
    constexpr int C = 128;
    __shared__ float Xs[2 * 32 * MM_LD128];        // [ri][b 32][c 128 (+1)]   fwd: X, dgrad: gY
    const int m = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    const long plane = (long)M * C;
    const float* src = MODE == 1 ? GY : X;
    const float* Wm = Wt + (long)m * C * C * 2;
    const int n0 = 32 * wave;
    for (int b0 = 0; b0 < B; b0 += 32) {
        if (b0) __syncthreads();
        for (int idx = tid; idx < 2 * 32 * (C / 4); idx += 256) {
            const int c4 = idx % (C / 4), r = idx / (C / 4);     // r = bl * 2 + ri
            const int bl = r >> 1, ri = r & 1, b = b0 + bl;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (b < B) v = *reinterpret_cast<const f32x4*>(src + (long)(b * 2 + ri) * plane + (long)m * C + 4 * c4);
#pragma unroll
            for (int t = 0; t < 4; ++t) Xs[(ri * 32 + bl) * MM_LD128 + 4 * c4 + t] = v[t];
        }
        __syncthreads();
        f32x16 accR = zero16(), accI = zero16();
        const float* xr_p = Xs + (0 * 32 + col) * MM_LD128;
        const float* xi_p = Xs + (1 * 32 + col) * MM_LD128;
        if (MODE == 0) {
            // Yr = Xr Wr - Xi Wi,  Yi = Xr Wi + Xi Wr;  K-step s: i = 2 s + half
            const f32x2* wp = reinterpret_cast<const f32x2*>(Wm) + (long)half * C + n0 + col;
#pragma unroll 8
            for (int s = 0; s < C / 2; ++s) {
                const f32x2 w = wp[(long)(2 * s) * C];
                const float xr = xr_p[2 * s + half], xi = xi_p[2 * s + half];
                accR = mfma32(xr, w[0], accR);
                accR = mfma32(-xi, w[1], accR);
                accI = mfma32(xr, w[1], accI);
                accI = mfma32(xi, w[0], accI);
            }
        } else {
            // GXr = GYr Wr + GYi Wi,  GXi = -GYr Wi + GYi Wr;  K-steps 2 t, 2 t + 1: o = 4 t + 2 half + u
            const f32x4* wp = reinterpret_cast<const f32x4*>(Wm + ((long)(n0 + col) * C + 2 * half) * 2);
#pragma unroll 4
            for (int t = 0; t < C / 4; ++t) {
                const f32x4 w = wp[2 * t];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int o = 4 * t + 2 * half + u;
                    const float gr = xr_p[o], gi = xi_p[o];
                    accR = mfma32(gr, w[2 * u], accR);
                    accR = mfma32(gi, w[2 * u + 1], accR);
                    accI = mfma32(-gr, w[2 * u + 1], accI);
                    accI = mfma32(gi, w[2 * u], accI);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int b = b0 + mfma_row(lane, r);
            if (b < B) {
                OUT[(long)(b * 2 + 0) * plane + (long)m * C + n0 + col] = accR[r];
                OUT[(long)(b * 2 + 1) * plane + (long)m * C + n0 + col] = accI[r];
            }
        }
    }
}
static bool mode128_direct() {            // RPB_MODE128_LDS=1: the first organisation (weight halves staged in LDS)
    static const bool off = getenv("RPB_MODE128_LDS") && atoi(getenv("RPB_MODE128_LDS")) == 1;
    return !off;
}
static size_t mode128_lds(int mode) {
    const size_t xs = 2 * 32 * MM_LD128, ws = mode == 0 ? 2 * 128 * MM_LD : (mode == 1 ? 2 * 64 * MM_LD128 : 2 * 32 * MM_LD128);
    return (xs + ws) * 4;
}
template <int MODE>
static int launch_mode128(const float* X, const float* W, const float* GY, float* OUT, int B, int M, int accumulate, hipStream_t st) {
    if (MODE != 2 && mode128_direct()) {
        hipLaunchKernelGGL((mode_mfma128d_kernel<MODE>), dim3(M), dim3(256), 0, st, X, W, GY, OUT, B, M);
        return RPB_OK;
    }
    const size_t lds = mode128_lds(MODE);
    (void)hipFuncSetAttribute((const void*)mode_mfma128_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((mode_mfma128_kernel<MODE>), dim3(M), dim3(256), lds, st, X, W, GY, OUT, B, M, accumulate);
    return RPB_OK;
}

static bool mode_mfma_on(int C) {
    static const bool off = getenv("RPB_MODE_CONTRACT_VALU") && atoi(getenv("RPB_MODE_CONTRACT_VALU")) == 1;
    return !off && C == 64;
}
static bool mode_bf16_on(int C) {        // RPB_MODE_CONTRACT_F32=1: the fp32-MFMA kernel of round 2
    static const bool off = (getenv("RPB_MODE_CONTRACT_F32") && atoi(getenv("RPB_MODE_CONTRACT_F32")) == 1) ||
                            (getenv("RPB_MODE_CONTRACT_VALU") && atoi(getenv("RPB_MODE_CONTRACT_VALU")) == 1);
    return !off && C == 64;
}
static bool mode_mfma128_on(int C) {
    static const bool off = getenv("RPB_MODE_CONTRACT_VALU") && atoi(getenv("RPB_MODE_CONTRACT_VALU")) == 1;
    return !off && C == 128;
}

static int mc_check(const void* a, const void* b, const void* c, int B, int M, int C) {
    RPB_REQUIRE(a && b && c, "mode_contract: null pointer");
    RPB_REQUIRE(B > 0 && M > 0, "mode_contract: bad sizes B=%d M=%d", B, M);
    RPB_REQUIRE(C > 0 && C <= MC_THREADS && MC_THREADS % C == 0, "mode_contract: C=%d must divide %d", C, MC_THREADS);
    return RPB_OK;
}

extern "C" int rpb_mode_contract_fwd(const float* X, const float* W, float* Y, int B, int M, int C, void* stream) {
    if (int e = mc_check(X, W, Y, B, M, C)) return e;
    if (mode_bf16_on(C)) {
        if (B <= 32 && mode_persist_on() && mode_persist_fwd()) {
            const int grid = M < mode_persist_wgs(0) * rpb_num_cus() ? M : mode_persist_wgs(0) * rpb_num_cus();
            hipLaunchKernelGGL((mode_bf16_kernel<0, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, X, W, (const float*)nullptr, Y, B, M, 0);
        } else {
            hipLaunchKernelGGL((mode_bf16_kernel<0>), dim3(M), dim3(256), 0, (hipStream_t)stream, X, W, (const float*)nullptr, Y, B, M, 0);
        }
        RPB_CHECK_LAUNCH("mode_contract_fwd");
    }
    if (mode_mfma_on(C)) {
        hipLaunchKernelGGL((mode_mfma_kernel<0>), dim3(M), dim3(256), 0, (hipStream_t)stream, X, W, (const float*)nullptr, Y, B, M, 0);
        RPB_CHECK_LAUNCH("mode_contract_fwd");
    }
    if (mode_mfma128_on(C)) {
        (void)launch_mode128<0>(X, W, nullptr, Y, B, M, 0, (hipStream_t)stream);
        RPB_CHECK_LAUNCH("mode_contract_fwd");
    }
    const size_t lds = (size_t)B * 2 * C * 4;
    RPB_REQUIRE(lds <= 160 * 1024, "mode_contract_fwd: B*C too large for LDS");
    (void)hipFuncSetAttribute((const void*)mode_contract_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    hipLaunchKernelGGL((mode_contract_kernel<false>), dim3(M), dim3(MC_THREADS), lds, (hipStream_t)stream, X, W, Y, B,
                       M, C);
    RPB_CHECK_LAUNCH("mode_contract_fwd");
}

extern "C" int rpb_mode_contract_dgrad(const float* GY, const float* W, float* GX, int B, int M, int C, void* stream) {
    if (int e = mc_check(GY, W, GX, B, M, C)) return e;
    if (mode_bf16_on(C)) {
        if (B <= 32 && mode_persist_on()) {
            const int grid = M < mode_persist_wgs(0) * rpb_num_cus() ? M : mode_persist_wgs(0) * rpb_num_cus();
            hipLaunchKernelGGL((mode_bf16_kernel<1, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr, W, GY, GX, B, M, 0);
        } else {
            hipLaunchKernelGGL((mode_bf16_kernel<1>), dim3(M), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr, W, GY, GX, B, M, 0);
        }
        RPB_CHECK_LAUNCH("mode_contract_dgrad");
    }
    if (mode_mfma_on(C)) {
        hipLaunchKernelGGL((mode_mfma_kernel<1>), dim3(M), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr, W, GY, GX, B, M, 0);
        RPB_CHECK_LAUNCH("mode_contract_dgrad");
    }
    if (mode_mfma128_on(C)) {
        (void)launch_mode128<1>(nullptr, W, GY, GX, B, M, 0, (hipStream_t)stream);
        RPB_CHECK_LAUNCH("mode_contract_dgrad");
    }
    if (C % MD_ROWS == 0) {
        const size_t lds = ((size_t)B * 2 * C + (size_t)MD_ROWS * (C + 1) * 2) * 4;
        RPB_REQUIRE(lds <= 160 * 1024, "mode_contract_dgrad: tiles too large for LDS (B=%d C=%d)", B, C);
        (void)hipFuncSetAttribute((const void*)mode_dgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(mode_dgrad_kernel, dim3(M * (C / MD_ROWS)), dim3(MC_THREADS), lds, (hipStream_t)stream, GY, W, GX, B, M, C);
        RPB_CHECK_LAUNCH("mode_contract_dgrad");
    }
    const size_t lds = ((size_t)B * 2 * C + (size_t)C * (C + 1) * 2) * 4;
    RPB_REQUIRE(lds <= 160 * 1024, "mode_contract_dgrad: tiles too large for LDS (B=%d C=%d)", B, C);
    (void)hipFuncSetAttribute((const void*)mode_contract_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    hipLaunchKernelGGL((mode_contract_kernel<true>), dim3(M), dim3(MC_THREADS), lds, (hipStream_t)stream, GY, W, GX, B,
                       M, C);
    RPB_CHECK_LAUNCH("mode_contract_dgrad");
}

extern "C" int rpb_mode_contract_wgrad(const float* X, const float* GY, float* GW, int B, int M, int C, int accumulate,
                                       void* stream) {
    if (int e = mc_check(X, GY, GW, B, M, C)) return e;
    if (mode_bf16_on(C)) {
        if (B <= 32 && mode_persist_on() && mode_persist_wgrad()) {      // measured slower (288 registers: one workgroup per CU): off
            const int grid = M < mode_persist_wgs(2) * rpb_num_cus() ? M : mode_persist_wgs(2) * rpb_num_cus();
            hipLaunchKernelGGL((mode_bf16_kernel<2, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, X, (const float*)nullptr, GY, GW, B, M, accumulate);
        } else {
            hipLaunchKernelGGL((mode_bf16_kernel<2>), dim3(M), dim3(256), 0, (hipStream_t)stream, X, (const float*)nullptr, GY, GW, B, M, accumulate);
        }
        RPB_CHECK_LAUNCH("mode_contract_wgrad");
    }
    if (mode_mfma_on(C)) {
        hipLaunchKernelGGL((mode_mfma_kernel<2>), dim3(M), dim3(256), 0, (hipStream_t)stream, X, (const float*)nullptr, GY, GW, B, M, accumulate);
        RPB_CHECK_LAUNCH("mode_contract_wgrad");
    }
    if (mode_mfma128_on(C)) {
        (void)launch_mode128<2>(X, nullptr, GY, GW, B, M, accumulate, (hipStream_t)stream);
        RPB_CHECK_LAUNCH("mode_contract_wgrad");
    }
    const size_t lds = (size_t)B * 4 * C * 4;
    RPB_REQUIRE(lds <= 160 * 1024, "mode_contract_wgrad: B*C too large for LDS");
    (void)hipFuncSetAttribute((const void*)mode_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(mode_wgrad_kernel, dim3(M), dim3(MC_THREADS), lds, (hipStream_t)stream, X, GY, GW, B, M, C,
                       accumulate);
    RPB_CHECK_LAUNCH("mode_contract_wgrad");
}
