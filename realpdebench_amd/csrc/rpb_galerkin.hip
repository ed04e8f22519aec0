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
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.x + m * a.ldx + 4 * lane);
        f32x4 g4 = {0.f, 0.f, 0.f, 0.f};
        if (BWD) g4 = *reinterpret_cast<const f32x4*>(a.gy + m * a.ldg + 4 * lane);
        const float mean = seg16_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 64.0f);
        const f32x4 d = v - mean;
        const float var = seg16_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.0f / 64.0f);
        const float inv = 1.0f / sqrtf(var + a.eps);
        const f32x4 xh = d * inv;
        if (!BWD) {
            *reinterpret_cast<f32x4*>(a.out + m * a.ldo + 4 * lane) = xh * ga + be;
        } else {
            const f32x4 dy = g4 * ga;
            const float m1 = seg16_sum(dy[0] + dy[1] + dy[2] + dy[3]) * (1.0f / 64.0f);
            const float m2 =
                seg16_sum(dy[0] * xh[0] + dy[1] * xh[1] + dy[2] * xh[2] + dy[3] * xh[3]) * (1.0f / 64.0f);
            *reinterpret_cast<f32x4*>(a.gx + m * a.ldgx + 4 * lane) = (dy - m1 - xh * m2) * inv;
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
