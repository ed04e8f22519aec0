// The [B x C]-sized algebra of the U-Net between its token-sized kernels (rounds 1-2 ran it as host glue under torch autograd):
//
//   rpb_gn_affine_fwd / _bwd   GroupNorm(8) statistics -> the per-(sample, channel) affine  y = A x + Bc  that rpb_affine_silu_* apply,
//                              with the time-embedding scale / shift folded in (unet.py:200-208 + 223-229), and its backward: d gamma,
//                              d beta, d scale|shift and the (P, Q) pair with which rpb_affine_silu_bwd_apply adds the gradient that
//                              flows through the group statistics
//   rpb_silu_fwd / _bwd        SiLU on the time embedding (unet.py:223)
//   rpb_relpos_bias_fwd / _bwd T5 relative-position bias: bias[h][i][j] = table[bucket(i, j)][h] (unet.py:78-116); the bucket index map
//                              is integer bookkeeping computed by the host exactly as the reference computes it
// One workgroup per (sample, group); sums over the group's channels go through LDS in a fixed order: bit-reproducible.
#include "rpb_common.h"

namespace {
__device__ __forceinline__ double block_sum(double v, double* red, int tid, int nthreads) {
    red[tid] = v;
    __syncthreads();
    double s = 0.0;
    if (tid == 0) {
        for (int i = 0; i < nthreads; ++i) s += red[i];
        red[0] = s;
    }
    __syncthreads();
    s = red[0];
    __syncthreads();
    return s;
}

// sums [B][2][C] fp64 (sum x, sum x^2 per sample and channel over n positions).  A, Bc [B][C]; stat [B][G][2] = (mean, invstd)
__global__ void gn_affine_fwd_kernel(const double* __restrict__ sums, const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ ss, float cnt, float eps, float* __restrict__ A, float* __restrict__ Bc,
                                     float* __restrict__ stat, int C, int G) {
    __shared__ double red[256];
    const int b = blockIdx.x / G, g = blockIdx.x % G, cg = C / G, tid = threadIdx.x;
    const bool on = tid < cg;
    const int c = g * cg + (on ? tid : 0);
    const double s0 = block_sum(on ? sums[((long)b * 2 + 0) * C + c] : 0.0, red, tid, blockDim.x);
    const double s1 = block_sum(on ? sums[((long)b * 2 + 1) * C + c] : 0.0, red, tid, blockDim.x);
    const float mean = (float)s0 / cnt;
    const float inv = 1.0f / sqrtf((float)s1 / cnt - mean * mean + eps);
    if (tid == 0) {
        stat[((long)b * G + g) * 2] = mean;
        stat[((long)b * G + g) * 2 + 1] = inv;
    }
    if (!on) return;
    float a = inv * gamma[c];
    float bb = beta[c] - mean * a;
    if (ss) {
        const float s = ss[(long)b * 2 * C + c] + 1.0f;
        a *= s;
        bb = bb * s + ss[(long)b * 2 * C + C + c];
    }
    A[(long)b * C + c] = a;
    Bc[(long)b * C + c] = bb;
}

// d [B][2][C] = (dL/dA, dL/dBc).  dgam, dbet [B][C] per-sample parts; dss [B][2C] or null; P, Q [B][C]: gradient w.r.t. x through the
// statistics is P + Q x
__global__ void gn_affine_bwd_kernel(const float* __restrict__ d, const float* __restrict__ stat, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, const float* __restrict__ ss, float cnt, float* __restrict__ dgam,
                                     float* __restrict__ dbet, float* __restrict__ dss, float* __restrict__ P, float* __restrict__ Q,
                                     int C, int G) {
    __shared__ double red[256];
    const int b = blockIdx.x / G, g = blockIdx.x % G, cg = C / G, tid = threadIdx.x;
    const bool on = tid < cg;
    const int c = g * cg + (on ? tid : 0);
    const float mean = stat[((long)b * G + g) * 2], inv = stat[((long)b * G + g) * 2 + 1];
    const float a0 = inv * gamma[c], b0 = beta[c] - mean * a0;
    const float dA = on ? d[((long)b * 2 + 0) * C + c] : 0.f, dB = on ? d[((long)b * 2 + 1) * C + c] : 0.f;
    float s1 = 1.0f;
    if (ss) {
        s1 = ss[(long)b * 2 * C + c] + 1.0f;
        if (on) {
            dss[(long)b * 2 * C + c] = dA * a0 + dB * b0;
            dss[(long)b * 2 * C + C + c] = dB;
        }
    }
    const float dB0 = dB * s1;
    const float dA0 = dA * s1 - dB0 * mean;
    if (on) {
        dbet[(long)b * C + c] = dB0;
        dgam[(long)b * C + c] = dA0 * inv;
    }
    const double dinv = block_sum(on ? (double)dA0 * gamma[c] : 0.0, red, tid, blockDim.x);
    const double dmean = block_sum(on ? -(double)dB0 * a0 : 0.0, red, tid, blockDim.x);
    const double inv3 = (double)inv * inv * inv;
    const float dS0 = (float)((dmean + dinv * inv3 * mean) / cnt);
    const float dS1 = (float)(dinv * (-0.5 * inv3) / cnt);
    if (on) {
        P[(long)b * C + c] = dS0;
        Q[(long)b * C + c] = 2.0f * dS1;
    }
}

__global__ void silu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = x[i];
        y[i] = v / (1.0f + __expf(-v));
    }
}
__global__ void silu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = x[i];
        const float sg = 1.0f / (1.0f + __expf(-v));
        gx[i] = gy[i] * sg * (1.0f + v * (1.0f - sg));
    }
}

// bias [heads][n2] = table[idx[p]][h];  table [nb][heads]
__global__ void relpos_fwd_kernel(const float* __restrict__ table, const int* __restrict__ idx, float* __restrict__ bias, int n2,
                                  int heads) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2 * heads) return;
    const int h = i / n2, p = i - h * n2;
    bias[i] = table[idx[p] * heads + h];
}
// gtable [nb][heads] = sum over positions p with idx[p] == k of gbias[h][p]: one workgroup per (bucket, head), fixed order
__global__ void relpos_bwd_kernel(const float* __restrict__ gbias, const int* __restrict__ idx, float* __restrict__ gtable, int n2,
                                  int heads) {
    __shared__ double red[256];
    const int k = blockIdx.x / heads, h = blockIdx.x % heads, tid = threadIdx.x;
    double s = 0.0;
    for (int p = tid; p < n2; p += blockDim.x)
        if (idx[p] == k) s += gbias[(long)h * n2 + p];
    s = block_sum(s, red, tid, blockDim.x);
    if (tid == 0) gtable[k * heads + h] = (float)s;
}
}  // namespace

extern "C" int rpb_gn_affine_fwd(const double* sums, const float* gamma, const float* beta, const float* ss, double count, float eps,
                                 float* A, float* Bc, float* stat, int B, int C, int G, void* stream) {
    RPB_REQUIRE(sums && gamma && beta && A && Bc && stat && B > 0 && G > 0 && C % G == 0 && C / G <= 256 && count > 0, "gn_affine_fwd: bad arguments");
    const int block = (C / G + 63) / 64 * 64;
    hipLaunchKernelGGL(gn_affine_fwd_kernel, dim3(B * G), dim3(block), 0, (hipStream_t)stream, sums, gamma, beta, ss, (float)count, eps, A,
                       Bc, stat, C, G);
    RPB_CHECK_LAUNCH("gn_affine_fwd");
}
extern "C" int rpb_gn_affine_bwd(const float* d, const float* stat, const float* gamma, const float* beta, const float* ss, double count,
                                 float* dgam, float* dbet, float* dss, float* P, float* Q, int B, int C, int G, void* stream) {
    RPB_REQUIRE(d && stat && gamma && beta && dgam && dbet && P && Q && (!ss || dss) && B > 0 && G > 0 && C % G == 0 && C / G <= 256 && count > 0,
                "gn_affine_bwd: bad arguments");
    const int block = (C / G + 63) / 64 * 64;
    hipLaunchKernelGGL(gn_affine_bwd_kernel, dim3(B * G), dim3(block), 0, (hipStream_t)stream, d, stat, gamma, beta, ss, (float)count, dgam,
                       dbet, dss, P, Q, C, G);
    RPB_CHECK_LAUNCH("gn_affine_bwd");
}
extern "C" int rpb_silu_fwd(const float* x, float* y, long n, void* stream) {
    RPB_REQUIRE(x && y && n > 0, "silu_fwd: bad arguments");
    hipLaunchKernelGGL(silu_fwd_kernel, dim3((unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    RPB_CHECK_LAUNCH("silu_fwd");
}
extern "C" int rpb_silu_bwd(const float* x, const float* gy, float* gx, long n, void* stream) {
    RPB_REQUIRE(x && gy && gx && n > 0, "silu_bwd: bad arguments");
    hipLaunchKernelGGL(silu_bwd_kernel, dim3((unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, gy, gx, n);
    RPB_CHECK_LAUNCH("silu_bwd");
}
extern "C" int rpb_relpos_bias_fwd(const float* table, const int* idx, float* bias, int n2, int heads, void* stream) {
    RPB_REQUIRE(table && idx && bias && n2 > 0 && heads > 0, "relpos_bias_fwd: bad arguments");
    hipLaunchKernelGGL(relpos_fwd_kernel, dim3((n2 * heads + 255) / 256), dim3(256), 0, (hipStream_t)stream, table, idx, bias, n2, heads);
    RPB_CHECK_LAUNCH("relpos_bias_fwd");
}
extern "C" int rpb_relpos_bias_bwd(const float* gbias, const int* idx, float* gtable, int n2, int heads, int nbuckets, void* stream) {
    RPB_REQUIRE(gbias && idx && gtable && n2 > 0 && heads > 0 && nbuckets > 0, "relpos_bias_bwd: bad arguments");
    hipLaunchKernelGGL(relpos_bwd_kernel, dim3(nbuckets * heads), dim3(256), 0, (hipStream_t)stream, gbias, idx, gtable, n2, heads);
    RPB_CHECK_LAUNCH("relpos_bias_bwd");
}
