// HBM-streaming kernels of the Galerkin Transformer path (reference realpdebench/model/galerkin_transformer_libs):
//   * per-head LayerNorm of the attention keys / values  -- SimpleAttention.forward, layers.py:844-856
//     (norm_K / norm_V = n_head x nn.LayerNorm(d_k = 64, eps = norm_eps)), forward and backward;
//   * the hand-over between the token tensor [B*n][C] of the encoder and the zero-padded channels-last cell tensor
//     [B][Tp][Hp][Wp][C] of the spectral regressor -- the `cat([x, grid]) -> fc -> permute -> F.pad` of
//     SpectralRegressor.forward, model.py:612-618 (the 256-wide part of fc is a token GEMM, rpb_gemm_nt; what is left
//     here is the grid term, the bias and the scatter into the padded layout), and its adjoint gather.
// The dense work of the path (Q/K/V, K^T V, Q P, FeedForward, fc) runs on rpb_gemm_nt / rpb_gemm_tn; the spectral
// layer and the regressor head reuse the FNO3d kernels (K2-K7).
#include "rpb_common.h"

#define HN_THREADS 256
#define HN_WAVES (HN_THREADS / 64)

// sum over the 16 lanes that hold one 64-channel head (lane = 16*head + q, 4 channels per lane)
__device__ __forceinline__ float seg16_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

struct HeadNormArgs {
    const float* x;       // [M][ldx], 256 channels used = 4 heads x 64
    const float* gamma;   // [256] = the 4 per-head LayerNorm weights back to back
    const float* beta;    // [256]
    float* out;           // fwd: [M][ldo]
    const float* gy;      // bwd: [M][ldg]
    float* gx;            // bwd: [M][ldgx]
    float* part;          // bwd: [rows][512] = d gamma | d beta partial sums
    long M;
    int ldx, ldo, ldg, ldgx;
    float eps;
};

// one wave per token: a single 1 KB coalesced load, two 16-lane reductions
template <bool BWD>
__global__ __launch_bounds__(HN_THREADS) void headnorm_kernel(HeadNormArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long w0 = (long)blockIdx.x * HN_WAVES + wave, stride = (long)gridDim.x * HN_WAVES;
    const f32x4 ga = *reinterpret_cast<const f32x4*>(a.gamma + 4 * lane);
    f32x4 be = {0.f, 0.f, 0.f, 0.f};
    if (!BWD) be = *reinterpret_cast<const f32x4*>(a.beta + 4 * lane);
    f32x4 dg = {0.f, 0.f, 0.f, 0.f}, db = {0.f, 0.f, 0.f, 0.f};
    for (long m = w0; m < a.M; m += stride) {
        const f32x4 v = RPB_SLD4(a.x + m * a.ldx + 4 * lane);
        f32x4 g4 = {0.f, 0.f, 0.f, 0.f};
        if (BWD) g4 = RPB_SLD4(a.gy + m * a.ldg + 4 * lane);
        const float mean = seg16_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 64.0f);
        const f32x4 d = v - mean;
        const float var = seg16_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.0f / 64.0f);
        const float inv = 1.0f / sqrtf(var + a.eps);
        const f32x4 xh = d * inv;
        if (!BWD) {
            RPB_SST4(a.out + m * a.ldo + 4 * lane, xh * ga + be);
        } else {
            const f32x4 dy = g4 * ga;
            const float m1 = seg16_sum(dy[0] + dy[1] + dy[2] + dy[3]) * (1.0f / 64.0f);
            const float m2 =
                seg16_sum(dy[0] * xh[0] + dy[1] * xh[1] + dy[2] * xh[2] + dy[3] * xh[3]) * (1.0f / 64.0f);
            RPB_SST4(a.gx + m * a.ldgx + 4 * lane, (dy - m1 - xh * m2) * inv);
            dg += g4 * xh;
            db += g4;
        }
    }
    if (BWD) {
        float* row = a.part + w0 * 512;
        *reinterpret_cast<f32x4*>(row + 4 * lane) = dg;
        *reinterpret_cast<f32x4*>(row + 256 + 4 * lane) = db;
    }
}

static int headnorm_grid(long M) {
    long g = (M + HN_WAVES - 1) / HN_WAVES;
    const long cap = (long)rpb_num_cus() * 8;
    if (g > cap) g = cap;
    return (int)(g < 1 ? 1 : g);
}

extern "C" long rpb_headnorm_bwd_rows(long M) { return (long)headnorm_grid(M) * HN_WAVES; }

static int headnorm_check(const HeadNormArgs& a, int C) {
    RPB_REQUIRE(C == 256, "headnorm: %d channels; the kernel handles 4 heads x 64 = 256 (every reference YAML)", C);
    RPB_REQUIRE(a.M > 0 && a.ldx % 4 == 0 && a.ldx >= C, "headnorm: bad M=%ld / ldx=%d", a.M, a.ldx);
    return RPB_OK;
}

extern "C" int rpb_headnorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, float* out, int ldo,
                                long M, int C, float eps, void* stream) {
    RPB_REQUIRE(x && gamma && beta && out, "headnorm_fwd: null pointer");
    HeadNormArgs a{};
    a.x = x; a.gamma = gamma; a.beta = beta; a.out = out; a.M = M; a.ldx = ldx; a.ldo = ldo; a.eps = eps;
    if (int e = headnorm_check(a, C)) return e;
    RPB_REQUIRE(ldo % 4 == 0 && ldo >= C, "headnorm_fwd: bad ldo=%d", ldo);
    hipLaunchKernelGGL(headnorm_kernel<false>, dim3(headnorm_grid(M)), dim3(HN_THREADS), 0, (hipStream_t)stream, a);
    RPB_CHECK_LAUNCH("headnorm_fwd");
}

extern "C" int rpb_headnorm_bwd(const float* x, int ldx, const float* gamma, const float* gy, int ldg, float* gx,
                                int ldgx, float* part, long M, int C, float eps, void* stream) {
    RPB_REQUIRE(x && gamma && gy && gx && part, "headnorm_bwd: null pointer");
    HeadNormArgs a{};
    a.x = x; a.gamma = gamma; a.gy = gy; a.gx = gx; a.part = part; a.M = M; a.ldx = ldx; a.ldg = ldg; a.ldgx = ldgx;
    a.eps = eps;
    if (int e = headnorm_check(a, C)) return e;
    RPB_REQUIRE(ldg % 4 == 0 && ldg >= C && ldgx % 4 == 0 && ldgx >= C, "headnorm_bwd: bad ldg=%d / ldgx=%d", ldg, ldgx);
    hipLaunchKernelGGL(headnorm_kernel<true>, dim3(headnorm_grid(M)), dim3(HN_THREADS), 0, (hipStream_t)stream, a);
    RPB_CHECK_LAUNCH("headnorm_bwd");
}

// ---------------------------------------------------------------------------------- tokens <-> padded cells
// out[b,t,h,w,:] = U[tok(b,t,h,w)][:] + Wg @ (gt[t], gh[h], gw[w]) + bias   for t<T, h<H, w<W, else 0
__global__ __launch_bounds__(256) void pad_grid_kernel(const float* __restrict__ U, const float* __restrict__ gt,
                                                       const float* __restrict__ gh, const float* __restrict__ gw,
                                                       const float* __restrict__ Wg, const float* __restrict__ bias,
                                                       float* __restrict__ out, long ncell_pad, int C, CropMap cm) {
    const int c4n = C >> 2;
    const long total = ncell_pad * c4n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long cell = idx / c4n;
        const int o = (int)(idx - cell * c4n) * 4;
        const int w = (int)(cell % cm.Wp);
        long r = cell / cm.Wp;
        const int h = (int)(r % cm.Hp);
        r /= cm.Hp;
        const int t = (int)(r % cm.Tp);
        const long b = r / cm.Tp;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (w < cm.W && h < cm.H && t < cm.T) {
            const long tok = ((b * cm.T + t) * cm.H + h) * (long)cm.W + w;
            v = *reinterpret_cast<const f32x4*>(U + tok * C + o) + *reinterpret_cast<const f32x4*>(bias + o);
            const float ft = gt[t], fh = gh[h], fw = gw[w];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float* wr = Wg + (o + k) * 3;
                v[k] += wr[0] * ft + wr[1] * fh + wr[2] * fw;
            }
        }
        *reinterpret_cast<f32x4*>(out + cell * C + o) = v;
    }
}

// out[tok][:] = g[pad(tok)][:]   (adjoint of the scatter above w.r.t. U)
__global__ __launch_bounds__(256) void crop_gather_kernel(const float* __restrict__ g, float* __restrict__ out,
                                                          long ncrop, int C, CropMap cm) {
    const int c4n = C >> 2;
    const long total = ncrop * c4n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long tok = idx / c4n;
        const int o = (int)(idx - tok * c4n) * 4;
        *reinterpret_cast<f32x4*>(out + tok * C + o) =
            *reinterpret_cast<const f32x4*>(g + crop_to_pad(cm, tok) * C + o);
    }
}

static int stream_grid(long items) {
    long g = (items + 255) / 256;
    const long cap = (long)rpb_num_cus() * 16;
    if (g > cap) g = cap;
    return (int)(g < 1 ? 1 : g);
}

extern "C" int rpb_pad_grid_fwd(const float* U, const float* gt, const float* gh, const float* gw, const float* Wg,
                                const float* bias, float* out, int B, int T, int H, int W, int C, int Tp, int Hp,
                                int Wp, void* stream) {
    RPB_REQUIRE(U && gt && gh && gw && Wg && bias && out, "pad_grid_fwd: null pointer");
    RPB_REQUIRE(B > 0 && C % 4 == 0 && T <= Tp && H <= Hp && W <= Wp, "pad_grid_fwd: bad sizes");
    const long ncell = (long)B * Tp * Hp * Wp;
    hipLaunchKernelGGL(pad_grid_kernel, dim3(stream_grid(ncell * (C / 4))), dim3(256), 0, (hipStream_t)stream, U, gt, gh,
                       gw, Wg, bias, out, ncell, C, CropMap{T, H, W, Tp, Hp, Wp});
    RPB_CHECK_LAUNCH("pad_grid_fwd");
}

extern "C" int rpb_crop_gather(const float* g, float* out, int B, int T, int H, int W, int C, int Tp, int Hp, int Wp,
                               void* stream) {
    RPB_REQUIRE(g && out, "crop_gather: null pointer");
    RPB_REQUIRE(B > 0 && C % 4 == 0 && T <= Tp && H <= Hp && W <= Wp, "crop_gather: bad sizes");
    const long ncrop = (long)B * T * H * W;
    hipLaunchKernelGGL(crop_gather_kernel, dim3(stream_grid(ncrop * (C / 4))), dim3(256), 0, (hipStream_t)stream, g, out,
                       ncrop, C, CropMap{T, H, W, Tp, Hp, Wp});
    RPB_CHECK_LAUNCH("crop_gather");
}

// ---------------------------------------------------------------------------------- per-head products of the attention
// linear_attention (layers.py:708-734) per sample and head: scores = K^T V (64 x 64, reduced over the n tokens) and
// out = Q (scores / n).  Both are HBM-streaming (a head's 64 x 64 matrix against 256 B of each token), so they get
// their own kernels instead of 256-wide dense GEMMs that would multiply the off-diagonal (cross-head) zeros.
//
// head_scores: part[chunk][b][h][i][j] = sum_{m in chunk} G[b,m][64h+i] * A[b,m][64h+j]   (TN product; also used for
// the backward dP = Q^T g).  Workgroup = 4 waves = the 4 heads of one token chunk (so it streams whole 1 KB rows);
// operands go straight from HBM into MFMA layout: a lane loads the channel PAIR (2c, 2c+1) of one token with one
// float2 (256 B per half-wave), which makes MFMA tile t hold the channels == t (mod 2).  16-token halves are
// double-buffered in registers.  rpb_reduce_partials finishes the sum over chunks in fp64.
struct HeadScoreArgs {
    const float* G;
    const float* A;
    float* part;
    long n;
    int ldg, lda, B, nh;
};

__global__ __launch_bounds__(256) void head_scores_kernel(HeadScoreArgs a) {
    const int lane = threadIdx.x & 63;
    const int h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 31, half = lane >> 5;
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const long per = ((a.n + nchunk - 1) / nchunk + 31) / 32 * 32;
    const long mb = (long)chunk * per;
    long me = mb + per;
    if (me > a.n) me = a.n;
    const float* gp = a.G + ((long)b * a.n) * a.ldg + h * 64 + col * 2;
    const float* xp = a.A + ((long)b * a.n) * a.lda + h * 64 + col * 2;
    f32x16 acc[2][2];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[o][i] = zero16();
    f32x2 ga[8], gb[8], xa[8], xb[8];
    auto load_half = [&](long m0, int hh, f32x2 (&gv)[8], f32x2 (&xv)[8]) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const long m = m0 + hh * 16 + 2 * s + half;
            f32x2 gq = {0.f, 0.f}, xq = {0.f, 0.f};
            if (m < me) {
                gq = *reinterpret_cast<const f32x2*>(gp + m * a.ldg);
                xq = *reinterpret_cast<const f32x2*>(xp + m * a.lda);
            }
            gv[s] = gq;
            xv[s] = xq;
        }
    };
    auto compute_half = [&](const f32x2 (&gv)[8], const f32x2 (&xv)[8]) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[o][i] = mfma32(gv[s][o], xv[s][i], acc[o][i]);
    };
    long m = mb;
    if (m < me) load_half(m, 0, ga, xa);
    for (; m < me; m += 32) {
        load_half(m, 1, gb, xb);
        compute_half(ga, xa);
        if (m + 32 < me) load_half(m + 32, 0, ga, xa);
        compute_half(gb, xb);
    }
    float* part = a.part + (((long)chunk * a.B + b) * a.nh + h) * 4096;
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) part[(mfma_row(lane, r) * 2 + o) * 64 + col * 2 + i] = acc[o][i][r];
}

extern "C" int rpb_head_scores_chunks(int B, long n) {
    long c = ((long)rpb_num_cus() * 6 + B - 1) / (B > 0 ? B : 1);
    const long cap = (n + 511) / 512;               // at least 512 tokens per chunk
    if (c > cap) c = cap;
    if (c < 1) c = 1;
    return (int)c;
}

extern "C" int rpb_head_scores(const float* G, int ldg, const float* A, int lda, float* part, int B, long n, int nheads,
                               void* stream) {
    RPB_REQUIRE(G && A && part && B > 0 && n > 0, "head_scores: bad arguments");
    RPB_REQUIRE(nheads >= 1 && nheads <= 4, "head_scores: %d heads of 64 channels (1..4)", nheads);
    RPB_REQUIRE(ldg % 2 == 0 && lda % 2 == 0 && ldg >= 64 * nheads && lda >= 64 * nheads, "head_scores: bad leading dimensions %d %d", ldg, lda);
    HeadScoreArgs a{G, A, part, n, ldg, lda, B, nheads};
    hipLaunchKernelGGL(head_scores_kernel, dim3(rpb_head_scores_chunks(B, n), B), dim3(64 * nheads), 0, (hipStream_t)stream, a);
    RPB_CHECK_LAUNCH("head_scores");
}

// head_apply: out[b,m][64h+j] = (sum_i X[b,m][64h+i] * Wm[b][h][i][j]) * mask + residual.
// Workgroup = 8 waves over one sample: wave w works on head w&3 and every second 32-token tile; the sample's four
// 64x64 matrices sit in LDS in MFMA-B order, the activation tile goes registers -> wave-private +1-padded LDS tile ->
// transposed reads (the cell_mix scheme), with the next tile's loads in flight during the 64 MFMAs of this one.
struct HeadApplyArgs {
    const float* X;
    const float* Wm;
    float* out;
    const float* residual;
    const float* mask;
    long n;
    int ldx, ldo, ldr, ldm, nh;
    DropSpec drop;       // in-kernel dropout instead of `mask`: element index = row * (64*nh) + channel
};

#define HA_WAVES 8
#define HA_XS 65

__global__ __launch_bounds__(HA_WAVES * 64) void head_apply_kernel(HeadApplyArgs a) {
    extern __shared__ float lds[];
    float* Wl = lds;                                   // [nh heads][64 k][32 col][2 tiles]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nh = a.nh;
    const int h = wave % nh, sub = wave / nh, nsub = HA_WAVES / nh;      // nh in {1, 2, 4}
    float* xl = lds + nh * 4096 + wave * 32 * HA_XS;   // wave-private [32][65]
    const int col = lane & 31, half = lane >> 5;
    const int b = blockIdx.y;
    const float* wsrc = a.Wm + (long)b * nh * 4096;
    for (int idx = threadIdx.x; idx < nh * 4096; idx += blockDim.x) {
        const int hh = idx >> 12, k = (idx >> 6) & 63, j = idx & 63;          // W[hh][k][j], coalesced read
        Wl[hh * 4096 + (k * 32 + (j & 31)) * 2 + (j >> 5)] = wsrc[idx];
    }
    __syncthreads();
    const long ntiles = (a.n + 31) / 32;
    const long per = (ntiles + gridDim.x - 1) / gridDim.x;
    const long t0 = (long)blockIdx.x * per;
    long t1 = t0 + per;
    if (t1 > ntiles) t1 = ntiles;
    const float* xb = a.X + ((long)b * a.n) * a.ldx + h * 64;
    const int lrow = lane >> 4, lc4 = (lane & 15) * 4;
    f32x4 xr[8];
    auto issue_x = [&](long tile) {
        const long m0 = tile * 32;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long m = m0 + j * 4 + lrow;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < a.n) v = *reinterpret_cast<const f32x4*>(xb + m * a.ldx + lc4);
            xr[j] = v;
        }
    };
    long tile = t0 + sub;
    if (tile < t1) issue_x(tile);
    const float* wp = Wl + h * 4096 + (half * 32 + col) * 2;
    for (; tile < t1; tile += nsub) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float* d = xl + (j * 4 + lrow) * HA_XS + lc4;
            d[0] = xr[j][0];
            d[1] = xr[j][1];
            d[2] = xr[j][2];
            d[3] = xr[j][3];
        }
        if (tile + nsub < t1) issue_x(tile + nsub);
        __builtin_amdgcn_wave_barrier();
        f32x16 acc[2] = {zero16(), zero16()};
        const float* ap = xl + col * HA_XS + half;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const float av = ap[2 * s];
            const f32x2 bv = *reinterpret_cast<const f32x2*>(wp + 2 * s * 64);
            acc[0] = mfma32(av, bv[0], acc[0]);
            acc[1] = mfma32(av, bv[1], acc[1]);
        }
        const long m0 = tile * 32;
        const long rowbase = (long)b * a.n + m0;
        // epilogue through the (dead) wave-private x tile: a lane then owns 4 consecutive channels of a row, so the mask /
        // residual reads and the store are 16 B per lane (4 rows x 256 B per wave instruction)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) xl[mfma_row(lane, r) * HA_XS + t * 32 + col] = acc[t][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int rr = it * 4 + lrow;
            if (m0 + rr < a.n) {
                const float* sp = xl + rr * HA_XS + lc4;
                f32x4 v = {sp[0], sp[1], sp[2], sp[3]};
                const int ch = h * 64 + lc4;
                if (a.mask) v = v * *reinterpret_cast<const f32x4*>(a.mask + (rowbase + rr) * a.ldm + ch);
                if (a.drop.thr) v = v * dropout4(a.drop, (unsigned long long)((rowbase + rr) * (64 * nh) + ch) >> 2);
                if (a.residual) v += *reinterpret_cast<const f32x4*>(a.residual + (rowbase + rr) * a.ldr + ch);
                *reinterpret_cast<f32x4*>(a.out + (rowbase + rr) * a.ldo + ch) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

extern "C" int rpb_head_apply(const float* X, int ldx, const float* Wm, float* out, int ldo, const float* residual,
                              int ldr, const float* mask, int ldm, int B, long n, int nheads, long drop_seed,
                              float drop_keep, void* stream) {
    RPB_REQUIRE(X && Wm && out && B > 0 && n > 0, "head_apply: bad arguments");
    RPB_REQUIRE(nheads == 1 || nheads == 2 || nheads == 4, "head_apply: %d heads of 64 channels (1, 2 or 4)", nheads);
    RPB_REQUIRE(ldx % 4 == 0 && ldx >= 64 * nheads && ldo % 4 == 0 && ldo >= 64 * nheads && ldr % 4 == 0 && ldm % 4 == 0,
                "head_apply: leading dimensions %d %d %d %d must be multiples of 4 and cover the heads", ldx, ldo, ldr, ldm);
    HeadApplyArgs a{X, Wm, out, residual, mask, n, ldx, ldo, ldr, ldm, nheads, make_drop(drop_seed, drop_keep)};
    if (a.drop.thr) RPB_REQUIRE(!mask, "head_apply: pass a mask tensor or a dropout seed, not both");
    const long ntiles = (n + 31) / 32;
    long chunks = ((long)rpb_num_cus() * 2 + B - 1) / B;
    const int nsub = HA_WAVES / nheads;
    if (chunks > (ntiles + nsub - 1) / nsub) chunks = (ntiles + nsub - 1) / nsub;
    if (chunks < 1) chunks = 1;
    const size_t lds = (size_t)(nheads * 4096 + HA_WAVES * 32 * HA_XS) * 4;
    (void)hipFuncSetAttribute((const void*)head_apply_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(head_apply_kernel, dim3((unsigned)chunks, B), dim3(HA_WAVES * 64), lds, (hipStream_t)stream, a);
    RPB_CHECK_LAUNCH("head_apply");
}

// ---------------------------------------------------------------------------------- dropout regenerated in the backward pass
// out[i] = g[i] * mask(seed, i) over a dense [rows][C] tensor: the gradient through an nn.Dropout whose forward mask was
// generated in a GEMM / head_apply epilogue from the same (seed, element index).
__global__ __launch_bounds__(256) void dropout_mul_kernel(const float* __restrict__ g, float* __restrict__ out, long n4,
                                                          DropSpec d) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        RPB_SST4(out + 4 * i, RPB_SLD4(g + 4 * i) * dropout4(d, (unsigned long long)i));
}

extern "C" int rpb_dropout_mul(const float* g, float* out, long n, long seed, float keep, void* stream) {
    RPB_REQUIRE(g && out && n > 0 && n % 4 == 0, "dropout_mul: n must be a positive multiple of 4");
    RPB_REQUIRE(keep > 0.f && keep < 1.f, "dropout_mul: keep probability %f outside (0, 1)", (double)keep);
    hipLaunchKernelGGL(dropout_mul_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, g, out, n / 4,
                       make_drop(seed, keep));
    RPB_CHECK_LAUNCH("dropout_mul");
}
