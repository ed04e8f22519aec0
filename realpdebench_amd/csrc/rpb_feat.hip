// Layer-0 algebra of FNO3d: the lifted input A0 = pad(fc0 [x, grid]) (fno.py:106-111) is LINEAR in F + 1 "feature fields"
// phi = (x_0 .. x_{Cin-1}, grid_t, grid_h, grid_w, 1) (zero in the pad margin), A0 = W0ext phi with W0ext = [fc0.weight | fc0.bias].
// The truncated DFT D of the first spectral layer is linear too, so
//
//   forward    X^0 = D(A0) = W0ext D(phi):     the three DFT stages run on the F + 1 fields (a [T][H][W][Cin*B + 4] tensor, ~2 % of
//                                              the 64-channel A0) and rpb_feat_mix expands the result to 64 channels per mode;
//   backward   d W0ext = sum_cells g_A0 (x) phi with g_A0 = Wc0^T gs0 + D^T g^x  splits into
//                 conv path      Wc0^T (sum_cells gs0 (x) phi)          = rpb_small_atb(Wc0, rpb_lift_bwd(gs0))
//                 spectral path  sum_modes g^x (x) D(phi)               = rpb_feat_mix_wgrad
//              so layer 0's inverse DFT stages, its data-gradient cell_mix and the 3.8 GB gradient tensor they produce
//              (whose only consumer was rpb_lift_bwd) disappear.
//
// Phi^ = D(phi) is laid out [2 (re, im)][M modes][NB columns]: columns b * Cin + j = field j of sample b, then the four
// sample-independent fields (grid_t, grid_h, grid_w, 1) at columns B * Cin .. + 3 (NB = that, padded to a multiple of 64).
#include "rpb_common.h"
#include <stdlib.h>

// Xh[b][ri][m][c] = sum_j W0[c][j] Phi[ri][m][b*Cin + j] + sum_j' W0[c][Cin + j'] Phi[ri][m][B*Cin + j']   (j' < 3)  + b0[c] Phi[..][B*Cin + 3]
__global__ __launch_bounds__(256) void feat_mix_kernel(const float* __restrict__ Phi, const float* __restrict__ w0,
                                                       const float* __restrict__ b0, float* __restrict__ Xh, int B, int M2,
                                                       int NB, int Cin, int C) {
    // one thread = 4 channels of one (b, row); rows = 2 * M.  [W0 | b0] sits transposed in LDS ([F + 1][C]): per feature one 16 B LDS
    // read and one (16-lane broadcast) global read, instead of five strided global reads
    extern __shared__ float wt[];
    const int c4n = C >> 2;
    const long total = (long)B * M2 * c4n;
    const int F = Cin + 3;
    for (int i = threadIdx.x; i < (F + 1) * C; i += blockDim.x) {
        const int j = i / C, c = i - j * C;
        wt[i] = j < F ? w0[c * F + j] : b0[c];
    }
    __syncthreads();
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % c4n);
        const long r = idx / c4n;
        const int row = (int)(r % M2);
        const int b = (int)(r / M2);
        const float* ph = Phi + (long)row * NB;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < F + 1; ++j) {
            const float v = j < Cin ? ph[b * Cin + j] : ph[B * Cin + (j - Cin)];
            const f32x4 w = *reinterpret_cast<const f32x4*>(wt + j * C + 4 * c4);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += v * w[k];
        }
        *reinterpret_cast<f32x4*>(Xh + ((long)b * M2 + row) * C + 4 * c4) = acc;
    }
}

extern "C" int rpb_feat_mix(const float* Phi, const float* w0, const float* b0, float* Xh, int B, int M2, int NB, int Cin, int C,
                            void* stream) {
    RPB_REQUIRE(Phi && w0 && b0 && Xh && B > 0 && M2 > 0 && C % 4 == 0 && NB >= B * Cin + 4, "feat_mix: bad arguments");
    const long total = (long)B * M2 * (C / 4);
    long grid = (total + 255) / 256;
    const long cap = (long)rpb_num_cus() * 8;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(feat_mix_kernel, dim3((unsigned)grid), dim3(256), (size_t)(Cin + 4) * C * 4, (hipStream_t)stream, Phi, w0, b0,
                       Xh, B, M2, NB, Cin, C);
    RPB_CHECK_LAUNCH("feat_mix");
}

// part[block][c * (F + 1) + j] = sum over the block's (b, row) of G[b][row][c] * phi_j(b, row)     (j == F: the ones field -> d b0)
// C = 64: a thread owns (channel c = tid & 63, feature group tid >> 6 of 4 threads); rows are walked block-cyclically.
#define FM_FMAX 24
__global__ __launch_bounds__(256) void feat_mix_wgrad_kernel(const float* __restrict__ G, const float* __restrict__ Phi,
                                                             float* __restrict__ part, int B, int M2, int NB, int Cin, int C) {
    const int c = threadIdx.x % C;
    const int grp = threadIdx.x / C, ngrp = blockDim.x / C;
    const int F1 = Cin + 4;
    float acc[FM_FMAX];
#pragma unroll
    for (int j = 0; j < FM_FMAX; ++j) acc[j] = 0.f;
    const long rows = (long)B * M2;
    for (long r = (long)blockIdx.x * ngrp + grp; r < rows; r += (long)gridDim.x * ngrp) {
        const int row = (int)(r % M2);
        const int b = (int)(r / M2);
        const float g = G[r * C + c];
        const float* ph = Phi + (long)row * NB;
#pragma unroll
        for (int j = 0; j < FM_FMAX; ++j) {
            if (j < F1) acc[j] += g * (j < Cin ? ph[b * Cin + j] : ph[B * Cin + (j - Cin)]);
        }
    }
    extern __shared__ float red[];          // [ngrp][C][F1]
    for (int j = 0; j < F1; ++j) red[(grp * C + c) * F1 + j] = acc[j];
    __syncthreads();
    for (int idx = threadIdx.x; idx < C * F1; idx += blockDim.x) {
        float s = 0.f;
        for (int g2 = 0; g2 < ngrp; ++g2) s += red[g2 * C * F1 + idx];
        part[(long)blockIdx.x * C * F1 + idx] = s;
    }
}

extern "C" int rpb_feat_mix_wgrad_rows(void) { return rpb_num_cus() * 2; }

// part [rpb_feat_mix_wgrad_rows()][C][Cin + 4]: per-block partial sums of sum_(b,row) G[b][row][c] * phi_j; column Cin + 3 is d bias
extern "C" int rpb_feat_mix_wgrad(const float* G, const float* Phi, float* part, int B, int M2, int NB, int Cin, int C,
                                  void* stream) {
    RPB_REQUIRE(G && Phi && part && B > 0 && M2 > 0 && C > 0 && 256 % C == 0 && Cin + 4 <= FM_FMAX && NB >= B * Cin + 4, "feat_mix_wgrad: bad arguments (C=%d Cin=%d)", C, Cin);
    const int grid = rpb_feat_mix_wgrad_rows();
    const size_t lds = (size_t)256 * (Cin + 4) * 4;
    hipLaunchKernelGGL(feat_mix_wgrad_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, G, Phi, part, B, M2, NB, Cin, C);
    RPB_CHECK_LAUNCH("feat_mix_wgrad");
}

// out[m][n] (+)= sum_k A[m*a_rs + k*a_cs] * Bm[k*b_rs + n*b_cs]   (strided operands: any transposition; matrices of at most a
// few thousand elements, one workgroup, fp64 accumulation)
__global__ __launch_bounds__(256) void small_gemm_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                         float* __restrict__ out, int M, int N, int K, int a_rs, int a_cs,
                                                         int b_rs, int b_cs, int ldo, int accumulate) {
    for (int idx = threadIdx.x; idx < M * N; idx += blockDim.x) {
        const int m = idx / N, n = idx - m * N;
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += (double)A[m * a_rs + k * a_cs] * (double)Bm[k * b_rs + n * b_cs];
        out[m * ldo + n] = accumulate ? out[m * ldo + n] + (float)s : (float)s;
    }
}

extern "C" int rpb_small_gemm(const float* A, const float* Bm, float* out, int M, int N, int K, int a_rs, int a_cs, int b_rs,
                              int b_cs, int ldo, int accumulate, void* stream) {
    RPB_REQUIRE(A && Bm && out && K > 0 && M > 0 && N > 0 && (long)M * N <= 65536, "small_gemm: bad arguments");
    hipLaunchKernelGGL(small_gemm_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, A, Bm, out, M, N, K, a_rs, a_cs, b_rs,
                       b_cs, ldo, accumulate);
    RPB_CHECK_LAUNCH("small_gemm");
}

// Phi_c[cell][FW]: the feature fields per padded cell, channels-last -- (x_0 .. x_{Cin-1}, grid_t, grid_h, grid_w, 1, 0 ..) on the
// cells of the data, all zeros in the pad margin.  It replaces the lifted tensor A0 = W0ext Phi_c (64 channels) as the input of
// layer 0's channel mixing and of its weight gradient: 8 (or 32) floats per cell instead of 64.
// One thread = one 16 B piece of the output, decoded with 32-bit divisions (the row-walking variant left most of a 256-thread block idle
// on the second pass over a 268-piece row; round 3 decoded every piece with three 64-bit divisions).
__global__ __launch_bounds__(256) void lift_feat_kernel(const float* __restrict__ x, const float* __restrict__ gt,
                                                        const float* __restrict__ gh, const float* __restrict__ gw,
                                                        float* __restrict__ out, unsigned total, int Cin, int FW, CropMap cm) {
    const unsigned q4 = FW >> 2, qs = FW == 8 ? 1 : 3;   // pieces per cell: 2 or 8
    const unsigned npiece = (unsigned)cm.Wp * q4;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const unsigned row = idx / npiece, i = idx - row * npiece;
        const unsigned bt = row / (unsigned)cm.Hp, h = row - bt * (unsigned)cm.Hp;
        const unsigned b = bt / (unsigned)cm.Tp, t = bt - b * (unsigned)cm.Tp;
        const int w = (int)(i >> qs), f0 = (int)(i & (q4 - 1)) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((int)h < cm.H && (int)t < cm.T && w < cm.W) {
            const float* xp = x + ((((long)b * cm.T + t) * cm.H + h) * (long)cm.W + w) * Cin;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int f = f0 + k;
                v[k] = f < Cin ? xp[f] : f == Cin ? gt[t] : f == Cin + 1 ? gh[h] : f == Cin + 2 ? gw[w] : f == Cin + 3 ? 1.f : 0.f;
            }
        }
        *reinterpret_cast<f32x4*>(out + (long)idx * 4) = v;
    }
}

// The headline instance (Cin = 2, FW = 8): a wave walks whole rows of the padded tensor -- the (b, t, h) decode is paid once per row in
// scalar registers, a lane's piece i of the row is cell i >> 1, half i & 1: (x0, x1, grid_t, grid_h) or (grid_w, 1, 0, 0), chosen with
// selects (no divergence), one 8 B load of x and one 16 B store per piece.  The generic kernel above decodes every piece with three
// divisions and a select chain per field: 0.29 ms for a 0.56 GB pass; this one is bound by its store.
__global__ __launch_bounds__(256) void lift_feat2_kernel(const float* __restrict__ x, const float* __restrict__ gt,
                                                         const float* __restrict__ gh, const float* __restrict__ gw,
                                                         float* __restrict__ out, int nrows, CropMap cm) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int nwaves = (int)gridDim.x * 4;
    const int npiece = 2 * cm.Wp;
    for (int row = wave; row < nrows; row += nwaves) {
        const unsigned bt = (unsigned)row / (unsigned)cm.Hp, h = (unsigned)row - bt * (unsigned)cm.Hp;
        const unsigned b = bt / (unsigned)cm.Tp, t = bt - b * (unsigned)cm.Tp;
        f32x4* orow = reinterpret_cast<f32x4*>(out) + (long)row * npiece;
        const bool live = (int)h < cm.H && (int)t < cm.T;                  // uniform
        const float* xrow = x + (((long)b * cm.T + (live ? t : 0)) * cm.H + (live ? h : 0)) * (long)cm.W * 2;
        const float ft = live ? gt[t] : 0.f, fh = live ? gh[h] : 0.f;
        for (int i = lane; i < npiece; i += 64) {
            const int w = i >> 1;
            const bool half = i & 1, in = live && w < cm.W;
            const int wc = w < cm.W ? w : 0;
            const float2 xv = *reinterpret_cast<const float2*>(xrow + 2 * wc);
            const float g = gw[wc];
            f32x4 v;
            v[0] = half ? g : xv.x;
            v[1] = half ? 1.f : xv.y;
            v[2] = half ? 0.f : ft;
            v[3] = half ? 0.f : fh;
            if (!in) v = f32x4{0.f, 0.f, 0.f, 0.f};
            orow[i] = v;
        }
    }
}

extern "C" int rpb_lift_feat(const float* x, const float* gt, const float* gh, const float* gw, float* out, int B, int T, int H,
                             int W, int Cin, int Tp, int Hp, int Wp, int FW, void* stream) {
    RPB_REQUIRE(x && gt && gh && gw && out && (FW == 8 || FW == 32) && Cin + 4 <= FW, "lift_feat: FW=%d must be 8 or 32 and hold Cin + 4 = %d fields", FW, Cin + 4);
    const long total = (long)B * Tp * Hp * Wp * (FW / 4);
    RPB_REQUIRE(total < (1L << 31), "lift_feat: too many cells");
    static const bool generic = getenv("RPB_LIFT_FEAT_GENERIC") && atoi(getenv("RPB_LIFT_FEAT_GENERIC")) == 1;
    if (Cin == 2 && FW == 8 && !generic) {
        const int nrows = B * Tp * Hp;
        long g2 = ((long)nrows + 3) / 4;
        const long cap2 = (long)rpb_num_cus() * 8;
        if (g2 > cap2) g2 = cap2;
        hipLaunchKernelGGL(lift_feat2_kernel, dim3((unsigned)g2), dim3(256), 0, (hipStream_t)stream, x, gt, gh, gw, out, nrows,
                           CropMap{T, H, W, Tp, Hp, Wp});
        RPB_CHECK_LAUNCH("lift_feat");
    }
    long grid = (total + 255) / 256;
    const long cap = (long)rpb_num_cus() * 32;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(lift_feat_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, gt, gh, gw, out, (unsigned)total, Cin,
                       FW, CropMap{T, H, W, Tp, Hp, Wp});
    RPB_CHECK_LAUNCH("lift_feat");
}
