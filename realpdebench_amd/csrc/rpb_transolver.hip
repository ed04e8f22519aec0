// Transolver (Physics-Attention on a structured 3-D mesh) -- the per-token kernels around the GEMMs of rpb_gemm.hip.
// Reference: realpdebench/model/TRANSOLVER_libs/Physics_Attention.py:148-176 and
//            realpdebench/model/TRANSOLVER_libs/Transolver_Structured_Mesh_3D.py:31-39,71-77,170-196.
// Tokens are channels-last rows [token][C]; heads are contiguous 32-channel groups of a row, so the reference's
// reshape/permute chains ('B N (H D) -> B H N D' and back) are pure index arithmetic here.
#include "rpb_common.h"

#define TS_THREADS 256

// ---------------------------------------------------------------------------------- tiny-K linear (+GELU)
// out[m][n] = act(sum_{k<K} x[m][k] W[n][k] + b[n]),  K <= 8  (preprocess.linear_pre: C_in = 3 -> 512, GELU)
__global__ __launch_bounds__(TS_THREADS) void tokens_lift_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                 const float* __restrict__ b, float* __restrict__ out,
                                                                 long M, int K, int N, int act) {
    extern __shared__ float wl[];   // [K][N] then bias[N]
    for (int idx = threadIdx.x; idx < K * N; idx += blockDim.x) {
        const int k = idx / N, n = idx - k * N;
        wl[idx] = W[n * K + k];
    }
    for (int idx = threadIdx.x; idx < N; idx += blockDim.x) wl[K * N + idx] = b[idx];
    __syncthreads();
    const int n4 = N >> 2;
    const long total = M * n4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long m = idx / n4;
        const int n = (int)(idx - m * n4) * 4;
        const float* xp = x + m * K;
        f32x4 v = *reinterpret_cast<const f32x4*>(wl + K * N + n);
        for (int k = 0; k < K; ++k) {
            const float f = xp[k];
            const f32x4 w = *reinterpret_cast<const f32x4*>(wl + k * N + n);
            v += w * f;
        }
        if (act) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = gelu_f(v[i]);
        }
        *reinterpret_cast<f32x4*>(out + m * N + n) = v;
    }
}

extern "C" int rpb_tokens_lift(const float* x, const float* W, const float* b, float* out, long M, int K, int N, int act,
                               void* stream) {
    RPB_REQUIRE(x && W && b && out && M > 0 && K > 0 && K <= 32 && N % 4 == 0, "tokens_lift: bad arguments (K=%d N=%d)", K, N);
    const size_t lds = (size_t)(K + 1) * N * 4;
    long grid = (M * (N / 4) + TS_THREADS - 1) / TS_THREADS;
    const long cap = (long)rpb_num_cus() * 8;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(tokens_lift_kernel, dim3((unsigned)grid), dim3(TS_THREADS), lds, (hipStream_t)stream, x, W, b, out,
                       M, K, N, act);
    RPB_CHECK_LAUNCH("tokens_lift");
}

// ---------------------------------------------------------------------------------- LayerNorm (one wave per token)
// nn.LayerNorm(C, eps=1e-5) with C = 64*V floats per token (V = 1..8 => C up to 512); biased variance.
template <int V>
__global__ __launch_bounds__(TS_THREADS) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ out,
                                                               long M, float eps) {
    constexpr int C = 64 * V;
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    float ga[V], be[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        ga[v] = gamma[v * 64 + lane];
        be[v] = beta[v * 64 + lane];
    }
    for (long m = wave; m < M; m += nwaves) {
        float xv[V], s = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            xv[v] = x[m * C + v * 64 + lane];
            s += xv[v];
        }
        const float mean = wave_sum(s) * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float dlt = xv[v] - mean;
            q += dlt * dlt;
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / C) + eps);
#pragma unroll
        for (int v = 0; v < V; ++v) out[m * C + v * 64 + lane] = (xv[v] - mean) * rstd * ga[v] + be[v];
    }
}

extern "C" int rpb_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* out, long M, int C,
                                 float eps, void* stream) {
    RPB_REQUIRE(x && gamma && beta && out && M > 0, "layernorm: bad arguments");
    RPB_REQUIRE(C % 64 == 0 && C >= 64 && C <= 512, "layernorm: C=%d must be a multiple of 64 up to 512", C);
    long grid = (M + 3) / 4;
    const long cap = (long)rpb_num_cus() * 8;
    if (grid > cap) grid = cap;
    hipStream_t st = (hipStream_t)stream;
#define RPB_LN(V_) \
    if (C == 64 * V_) hipLaunchKernelGGL((layernorm_kernel<V_>), dim3((unsigned)grid), dim3(TS_THREADS), 0, st, x, gamma, beta, out, M, eps);
    RPB_LN(1) RPB_LN(2) RPB_LN(3) RPB_LN(4) RPB_LN(5) RPB_LN(6) RPB_LN(7) RPB_LN(8)
#undef RPB_LN
    RPB_CHECK_LAUNCH("layernorm");
}

// ---------------------------------------------------------------------------------- slice: weights + token sums
// Physics_Attention.py:158-162 for dim_head = 32, heads = C/32, G <= 32 slices:
//   w[m][h][g]   = softmax_g( (xmid[m][h*32:+32] . Ws[g] + bs[g]) / clamp(temp[h], 0.1, 5) )
//   norm[b][h][g] = sum_{m in b} w,     tokS[b][h][g][c] = sum_{m in b} fx[m][h*32+c] * w[m][h][g]
// xf: [M][ldx] rows holding fx_mid at column 0 and x_mid at column C (the dual convolution writes them side by side).
// One block owns 64-token tiles of one sample; per tile: phase 1 = logits+softmax (thread = token x head group),
// phase 2 = the [G x 64] . [64 x 32] token sums (thread = head x channel, G accumulators in registers).
__global__ __launch_bounds__(TS_THREADS) void slice_fwd_kernel(const float* __restrict__ xf, const float* __restrict__ Ws,
                                                               const float* __restrict__ bs, const float* __restrict__ temp,
                                                               float* __restrict__ w_out, float* __restrict__ part,
                                                               int ntok, int heads, int G, int ldx, int blocks_per_sample,
                                                               const float* __restrict__ w_in) {
    extern __shared__ float lds[];
    const int C = heads * 32;
    float* Wsl = lds;                               // [G][33]
    float* wl = Wsl + G * 33;                       // [heads][64][G]
    float* fl = wl + heads * 64 * G;                // [64][C + 1]
    const int b = blockIdx.x / blocks_per_sample, blk = blockIdx.x % blocks_per_sample;
    for (int idx = threadIdx.x; idx < G * 32; idx += blockDim.x) Wsl[(idx >> 5) * 33 + (idx & 31)] = Ws[idx];
    __syncthreads();

    const int tid = threadIdx.x;
    const int ph = tid >> 5, pc = tid & 31;         // phase-2 role: head ph (if < heads), channel pc
    float accT[32];
#pragma unroll
    for (int g = 0; g < 32; ++g) accT[g] = 0.f;

    const long base = (long)b * ntok;
    for (int t0 = blk * 64; t0 < ntok; t0 += blocks_per_sample * 64) {
        // ---- stage fx tile (coalesced) : 64 tokens x C floats
        for (int idx = tid; idx < 64 * (C / 4); idx += blockDim.x) {
            const int r = idx / (C / 4), c4 = idx - r * (C / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (t0 + r < ntok) v = *reinterpret_cast<const f32x4*>(xf + (base + t0 + r) * ldx + 4 * c4);
            float* dst = fl + r * (C + 1) + 4 * c4;
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
        }
        // ---- phase 1: thread -> (token = tid & 63, heads hq, hq + 4, ...)
        {
            const int r = tid & 63;
            const bool ok = t0 + r < ntok;
            for (int h = tid >> 6; h < heads; h += TS_THREADS / 64) {
                if (w_in) {        // weights given (backward of deslice): just stage them
#pragma unroll
                    for (int g = 0; g < 32; ++g)
                        if (g < G) wl[(h * 64 + r) * G + g] = ok ? w_in[((base + t0 + r) * heads + h) * G + g] : 0.f;
                    continue;
                }
                float xv[32];
                const float* xp = xf + (base + t0 + r) * ldx + C + h * 32;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (ok) v = *reinterpret_cast<const f32x4*>(xp + 4 * k);
                    xv[4 * k] = v[0]; xv[4 * k + 1] = v[1]; xv[4 * k + 2] = v[2]; xv[4 * k + 3] = v[3];
                }
                const float inv_t = 1.0f / fminf(fmaxf(temp[h], 0.1f), 5.0f);
                float lg[32], mx = -3.0e38f;
#pragma unroll
                for (int g = 0; g < 32; ++g) {
                    float s = -3.0e38f;
                    if (g < G) {
                        s = bs[g];
#pragma unroll
                        for (int k = 0; k < 32; ++k) s += xv[k] * Wsl[g * 33 + k];
                        s *= inv_t;
                    }
                    lg[g] = s;
                    mx = fmaxf(mx, s);
                }
                float den = 0.f;
#pragma unroll
                for (int g = 0; g < 32; ++g) {
                    lg[g] = (g < G) ? expf(lg[g] - mx) : 0.f;
                    den += lg[g];
                }
                const float inv = 1.0f / den;
#pragma unroll
                for (int g = 0; g < 32; ++g) {
                    if (g < G) {
                        const float wv = ok ? lg[g] * inv : 0.f;
                        wl[(h * 64 + r) * G + g] = wv;
                        if (ok) w_out[((base + t0 + r) * heads + h) * G + g] = wv;
                    }
                }
            }
        }
        __syncthreads();
        // ---- phase 2: thread (head ph, channel pc): accT[g] += sum_r w[ph][r][g] * fx[r][ph*32+pc]
        if (ph < heads) {
            for (int r = 0; r < 64; ++r) {
                const float f = fl[r * (C + 1) + ph * 32 + pc];
                const float* wr = wl + (ph * 64 + r) * G;
#pragma unroll
                for (int g = 0; g < 32; ++g)
                    if (g < G) accT[g] += wr[g] * f;
            }
        }
        __syncthreads();
    }
    // partial row: [heads][G][32] token sums (the norms are a column sum of w, see slice_norm_kernel)
    float* prow = part + (long)blockIdx.x * ((long)heads * G * 32);
    if (ph < heads) {
#pragma unroll
        for (int g = 0; g < 32; ++g)
            if (g < G) prow[((long)ph * G + g) * 32 + pc] = accT[g];
    }
}

// norms: norm[b][h][g] = sum_m w[m][h][g]  -- a column sum of w_out (cheap second pass over the weights only)
__global__ __launch_bounds__(TS_THREADS) void slice_norm_kernel(const float* __restrict__ w, float* __restrict__ part,
                                                                int ntok, int HG, int blocks_per_sample) {
    extern __shared__ float red[];   // [nsub][HG]
    const int b = blockIdx.x / blocks_per_sample, blk = blockIdx.x % blocks_per_sample;
    const int c = threadIdx.x % HG, sub = threadIdx.x / HG, nsub = blockDim.x / HG;
    float s = 0.f;
    if (sub < nsub)
        for (int t = blk * nsub + sub; t < ntok; t += blocks_per_sample * nsub) s += w[((long)b * ntok + t) * HG + c];
    if (sub < nsub) red[sub * HG + c] = s;
    __syncthreads();
    if (threadIdx.x < HG) {
        float t = 0.f;
        for (int k = 0; k < nsub; ++k) t += red[k * HG + threadIdx.x];
        part[(long)blockIdx.x * HG + threadIdx.x] = t;
    }
}

extern "C" int rpb_slice_blocks_per_sample(int B) {
    int n = (rpb_num_cus() * 2 + B - 1) / B;
    return n < 1 ? 1 : n;
}

extern "C" int rpb_slice_fwd(const float* xf, const float* Ws, const float* bs, const float* temp, float* w_out,
                             float* tok_part, float* norm_part, int B, int ntok, int heads, int G, int ldx,
                             const float* w_in, void* stream) {
    RPB_REQUIRE(xf && tok_part && (w_in || (Ws && bs && temp && w_out && norm_part)), "slice_fwd: null pointer");
    RPB_REQUIRE(heads >= 1 && heads <= 8 && G >= 1 && G <= 32 && (heads * G) <= TS_THREADS && TS_THREADS % (heads * G) == 0,
                "slice_fwd: heads=%d G=%d unsupported (dim_head must be 32)", heads, G);
    const int bps = rpb_slice_blocks_per_sample(B);
    const int C = heads * 32;
    const size_t lds = ((size_t)G * 33 + (size_t)heads * 64 * G + (size_t)64 * (C + 1)) * 4;
    RPB_REQUIRE(lds <= 160 * 1024, "slice_fwd: LDS");
    (void)hipFuncSetAttribute((const void*)slice_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(slice_fwd_kernel, dim3(B * bps), dim3(TS_THREADS), lds, st, xf, w_in ? xf : Ws, bs, temp, w_out,
                       tok_part, ntok, heads, G, ldx, bps, w_in);
    const int HG = heads * G;
    if (!w_in)
        hipLaunchKernelGGL(slice_norm_kernel, dim3(B * bps), dim3(TS_THREADS), (size_t)(TS_THREADS / HG) * HG * 4, st, w_out,
                       norm_part, ntok, HG, bps);
    RPB_CHECK_LAUNCH("slice_fwd");
}

// ---------------------------------------------------------------------------------- attention among slice tokens
// Physics_Attention.py:164-171 (eval: dropout off).  One block per (b, h); tok[b][h][G][32] -> out same shape.
__global__ __launch_bounds__(TS_THREADS) void slice_attn_kernel(const float* __restrict__ tokS, const float* __restrict__ norm,
                                                                const float* __restrict__ Wq, const float* __restrict__ Wk,
                                                                const float* __restrict__ Wv, float* __restrict__ out,
                                                                int G, float scale) {
    __shared__ float t[32][33], q[32][33], k[32][33], v[32][33], p[32][33];
    const int bh = blockIdx.x;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < G * 32; idx += blockDim.x) {
        const int g = idx >> 5, c = idx & 31;
        t[g][c] = tokS[((long)bh * G + g) * 32 + c] / (norm[(long)bh * G + g] + 1e-5f);     // Physics_Attention.py:161-162
    }
    __syncthreads();
    for (int idx = tid; idx < G * 32; idx += blockDim.x) {
        const int g = idx >> 5, c = idx & 31;
        float sq = 0.f, sk = 0.f, sv = 0.f;
        for (int j = 0; j < 32; ++j) {
            const float tv = t[g][j];
            sq += tv * Wq[c * 32 + j];
            sk += tv * Wk[c * 32 + j];
            sv += tv * Wv[c * 32 + j];
        }
        q[g][c] = sq; k[g][c] = sk; v[g][c] = sv;
    }
    __syncthreads();
    for (int idx = tid; idx < G * G; idx += blockDim.x) {
        const int i = idx / G, j = idx - i * G;
        float s = 0.f;
        for (int c = 0; c < 32; ++c) s += q[i][c] * k[j][c];
        p[i][j] = s * scale;
    }
    __syncthreads();
    if (tid < G) {
        float mx = -3.0e38f;
        for (int j = 0; j < G; ++j) mx = fmaxf(mx, p[tid][j]);
        float den = 0.f;
        for (int j = 0; j < G; ++j) {
            p[tid][j] = expf(p[tid][j] - mx);
            den += p[tid][j];
        }
        const float inv = 1.f / den;
        for (int j = 0; j < G; ++j) p[tid][j] *= inv;
    }
    __syncthreads();
    for (int idx = tid; idx < G * 32; idx += blockDim.x) {
        const int g = idx >> 5, c = idx & 31;
        float s = 0.f;
        for (int j = 0; j < G; ++j) s += p[g][j] * v[j][c];
        out[((long)bh * G + g) * 32 + c] = s;
    }
}

extern "C" int rpb_slice_attn(const float* tokS, const float* norm, const float* Wq, const float* Wk, const float* Wv,
                              float* out, int BH, int G, void* stream) {
    RPB_REQUIRE(tokS && norm && Wq && Wk && Wv && out && BH > 0 && G >= 1 && G <= 32, "slice_attn: bad arguments");
    hipLaunchKernelGGL(slice_attn_kernel, dim3(BH), dim3(TS_THREADS), 0, (hipStream_t)stream, tokS, norm, Wq, Wk, Wv, out,
                       G, 1.0f / sqrtf(32.f));
    RPB_CHECK_LAUNCH("slice_attn");
}

// ---------------------------------------------------------------------------------- deslice
// out[m][h*32+c] = sum_g w[m][h][g] * tok2[b][h][g][c]     (Physics_Attention.py:173-175)
__global__ __launch_bounds__(TS_THREADS) void deslice_kernel(const float* __restrict__ w, const float* __restrict__ tok2,
                                                             float* __restrict__ out, int ntok, int heads, int G,
                                                             int blocks_per_sample) {
    extern __shared__ float tl[];   // [heads][G][32]
    const int C = heads * 32;
    const int b = blockIdx.x / blocks_per_sample, blk = blockIdx.x % blocks_per_sample;
    for (int idx = threadIdx.x; idx < heads * G * 32; idx += blockDim.x) tl[idx] = tok2[(long)b * heads * G * 32 + idx];
    __syncthreads();
    const int c4n = C / 4;
    const int c4 = threadIdx.x % c4n, sub = threadIdx.x / c4n, nsub = blockDim.x / c4n;
    const int h = (c4 * 4) / 32, c = (c4 * 4) % 32;
    for (int t = blk * nsub + sub; t < ntok; t += blocks_per_sample * nsub) {
        const long m = (long)b * ntok + t;
        const float* wr = w + (m * heads + h) * G;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < G; ++g) {
            const float wv = wr[g];
            const f32x4 tv = *reinterpret_cast<const f32x4*>(tl + (h * G + g) * 32 + c);
            acc += tv * wv;
        }
        *reinterpret_cast<f32x4*>(out + m * C + c4 * 4) = acc;
    }
}

extern "C" int rpb_deslice_fwd(const float* w, const float* tok2, float* out, int B, int ntok, int heads, int G,
                               void* stream) {
    RPB_REQUIRE(w && tok2 && out && heads >= 1 && heads <= 8 && G <= 32 && TS_THREADS % (heads * 8) == 0, "deslice: bad arguments");
    const int bps = rpb_slice_blocks_per_sample(B) * 4;
    hipLaunchKernelGGL(deslice_kernel, dim3(B * bps), dim3(TS_THREADS), (size_t)heads * G * 32 * 4, (hipStream_t)stream, w,
                       tok2, out, ntok, heads, G, bps);
    RPB_CHECK_LAUNCH("deslice");
}


// ---------------------------------------------------------------------------------- LayerNorm backward
// gx = rstd * (gy*gamma - mean(gy*gamma) - xhat * mean(gy*gamma*xhat)) [+ gadd];  per-wave partials of
// dgamma = sum gy*xhat and dbeta = sum gy  ->  part[nwaves][2][C].  mean / rstd are recomputed from x.
template <int V>
__global__ __launch_bounds__(TS_THREADS) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                   const float* __restrict__ gy, const float* gadd, float* gx,
                                                                   float* __restrict__ part, long M, float eps) {
    constexpr int C = 64 * V;
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    float ga[V], dg[V], db[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        ga[v] = gamma[v * 64 + lane];
        dg[v] = db[v] = 0.f;
    }
    for (long m = wave; m < M; m += nwaves) {
        float xv[V], gv[V], s = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            xv[v] = x[m * C + v * 64 + lane];
            gv[v] = gy[m * C + v * 64 + lane];
            s += xv[v];
        }
        const float mean = wave_sum(s) * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            xv[v] -= mean;
            q += xv[v] * xv[v];
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / C) + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            xv[v] *= rstd;                       // xhat
            dg[v] += gv[v] * xv[v];
            db[v] += gv[v];
            gv[v] *= ga[v];                      // gy * gamma
            s1 += gv[v];
            s2 += gv[v] * xv[v];
        }
        s1 = wave_sum(s1) * (1.0f / C);
        s2 = wave_sum(s2) * (1.0f / C);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float o = rstd * (gv[v] - s1 - xv[v] * s2);
            if (gadd) o += gadd[m * C + v * 64 + lane];
            gx[m * C + v * 64 + lane] = o;
        }
    }
    if (wave < nwaves) {
        float* pr = part + wave * 2 * C;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            pr[v * 64 + lane] = dg[v];
            pr[C + v * 64 + lane] = db[v];
        }
    }
}

extern "C" long rpb_layernorm_bwd_rows(long M) {
    long grid = (M + 3) / 4;
    const long cap = (long)rpb_num_cus() * 8;
    if (grid > cap) grid = cap;
    return grid * (TS_THREADS / 64);
}

extern "C" int rpb_layernorm_bwd(const float* x, const float* gamma, const float* gy, const float* gadd, float* gx,
                                 float* part, long M, int C, float eps, void* stream) {
    RPB_REQUIRE(x && gamma && gy && gx && part && M > 0, "layernorm_bwd: bad arguments");
    RPB_REQUIRE(C % 64 == 0 && C >= 64 && C <= 512, "layernorm_bwd: C=%d must be a multiple of 64 up to 512", C);
    const unsigned grid = (unsigned)(rpb_layernorm_bwd_rows(M) / (TS_THREADS / 64));
    hipStream_t st = (hipStream_t)stream;
#define RPB_LNB(V_) \
    if (C == 64 * V_) hipLaunchKernelGGL((layernorm_bwd_kernel<V_>), dim3(grid), dim3(TS_THREADS), 0, st, x, gamma, gy, gadd, gx, part, M, eps);
    RPB_LNB(1) RPB_LNB(2) RPB_LNB(3) RPB_LNB(4) RPB_LNB(5) RPB_LNB(6) RPB_LNB(7) RPB_LNB(8)
#undef RPB_LNB
    RPB_CHECK_LAUNCH("layernorm_bwd");
}

// ---------------------------------------------------------------------------------- slice backward
// Per (token, head), with w the saved slice weights (softmax output), T2 = attention output tokens, gT = dL/d(tokS)
// (slice-token sums, before the division by the norm), gN = dL/d(norm):
//   gw[g]   = sum_c gox[c] T2[g][c]  +  sum_c fx[c] gT[g][c]  +  gN[g]
//   gl[g]   = w[g] (gw[g] - sum_g' w[g'] gw[g'])                      gradient w.r.t. the temperature-scaled logits
//   g_xmid  = (1/tau) sum_g gl[g] Ws[g][:],      g_fxmid = sum_g w[g] gT[g][:]
//   dWs[g][c] += (1/tau) gl[g] xmid[c],  dbs[g] += (1/tau) gl[g],  dtau[h] += -(1/tau) sum_g gl[g] log w[g]
// (sum_g gl = 0, so log-sum-exp drops out of dtau).  Output g_xf rows = [g_fxmid | g_xmid] = the gradient of the dual
// convolution's output; partial row = [G*32 dWs | G dbs | heads dtau].
__global__ __launch_bounds__(TS_THREADS) void slice_bwd_kernel(const float* __restrict__ xf, const float* __restrict__ w,
                                                               const float* __restrict__ gox, const float* __restrict__ tok2,
                                                               const float* __restrict__ gT, const float* __restrict__ gN,
                                                               const float* __restrict__ Ws, const float* __restrict__ temp,
                                                               float* __restrict__ gxf, float* __restrict__ part, int ntok,
                                                               int heads, int G, int blocks_per_sample) {
    extern __shared__ float lds[];
    const int C = heads * 32;
    float* Wsl = lds;                               // [G][33]
    float* t2l = Wsl + G * 33;                      // [heads][G][33]
    float* gtl = t2l + heads * G * 33;              // [heads][G][33]
    float* gll = gtl + heads * G * 33;              // [heads][64][G]   scaled logit gradients of the tile
    float* xml = gll + heads * 64 * G;              // [64][C + 1]      x_mid tile
    float* red = xml + 64 * (C + 1);                // [G + heads] block sums of dbs, dtau
    const int b = blockIdx.x / blocks_per_sample, blk = blockIdx.x % blocks_per_sample;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < G * 32; idx += blockDim.x) Wsl[(idx >> 5) * 33 + (idx & 31)] = Ws[idx];
    for (int idx = tid; idx < heads * G * 32; idx += blockDim.x) {
        const int hg = idx >> 5, c = idx & 31;
        t2l[hg * 33 + c] = tok2[(long)b * heads * G * 32 + idx];
        gtl[hg * 33 + c] = gT[(long)b * heads * G * 32 + idx];
    }
    for (int idx = tid; idx < G + heads; idx += blockDim.x) red[idx] = 0.f;
    __syncthreads();

    const int pg = tid >> 5, pc = tid & 31;         // phase-2 role: slices pg, pg + 8, ... ; channel pc
    float accW[4] = {0.f, 0.f, 0.f, 0.f};           // G <= 32 -> at most 4 slices per thread
    const long base = (long)b * ntok;
    for (int t0 = blk * 64; t0 < ntok; t0 += blocks_per_sample * 64) {
        for (int idx = tid; idx < 64 * (C / 4); idx += blockDim.x) {           // stage the x_mid tile
            const int r = idx / (C / 4), c4 = idx - r * (C / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (t0 + r < ntok) v = *reinterpret_cast<const f32x4*>(xf + (base + t0 + r) * 2 * C + C + 4 * c4);
            float* dst = xml + r * (C + 1) + 4 * c4;
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
        }
        {
            const int r = tid & 63;
            const bool ok = t0 + r < ntok;
            const long m = base + t0 + r;
            for (int h = tid >> 6; h < heads; h += TS_THREADS / 64) {
                float fxv[32], gov[32], wv[32], gw[32];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
                    if (ok) {
                        a = *reinterpret_cast<const f32x4*>(xf + m * 2 * C + h * 32 + 4 * k);
                        c = *reinterpret_cast<const f32x4*>(gox + m * C + h * 32 + 4 * k);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        fxv[4 * k + i] = a[i];
                        gov[4 * k + i] = c[i];
                    }
                }
                float dot = 0.f;
#pragma unroll
                for (int g = 0; g < 32; ++g) {
                    float s = 0.f, wg = 0.f;
                    if (g < G) {
                        wg = ok ? w[(m * heads + h) * G + g] : 0.f;
                        s = gN[((long)b * heads + h) * G + g];
                        const float* t2 = t2l + (h * G + g) * 33;
                        const float* gt = gtl + (h * G + g) * 33;
#pragma unroll
                        for (int c = 0; c < 32; ++c) s += gov[c] * t2[c] + fxv[c] * gt[c];
                    }
                    wv[g] = wg;
                    gw[g] = s;
                    dot += wg * s;
                }
                const float inv_t = 1.0f / fminf(fmaxf(temp[h], 0.1f), 5.0f);
                float gfx[32], gxm[32], dtau = 0.f;
#pragma unroll
                for (int c = 0; c < 32; ++c) gfx[c] = gxm[c] = 0.f;
#pragma unroll
                for (int g = 0; g < 32; ++g) {
                    if (g < G) {
                        const float gl = wv[g] * (gw[g] - dot);            // d/d(scaled logit)
                        const float glr = gl * inv_t;                       // d/d(raw logit)
                        gll[(h * 64 + r) * G + g] = glr;
                        if (wv[g] > 0.f) dtau -= gl * logf(wv[g]);
                        const float* gt = gtl + (h * G + g) * 33;
#pragma unroll
                        for (int c = 0; c < 32; ++c) {
                            gfx[c] += wv[g] * gt[c];
                            gxm[c] += glr * Wsl[g * 33 + c];
                        }
                        if (ok) atomicAdd(&red[g], glr);
                    }
                }
                if (ok) {
                    atomicAdd(&red[G + h], dtau * inv_t);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        f32x4 a, c;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            a[i] = gfx[4 * k + i];
                            c[i] = gxm[4 * k + i];
                        }
                        *reinterpret_cast<f32x4*>(gxf + m * 2 * C + h * 32 + 4 * k) = a;
                        *reinterpret_cast<f32x4*>(gxf + m * 2 * C + C + h * 32 + 4 * k) = c;
                    }
                }
            }
        }
        __syncthreads();
        // phase 2: dWs[g][c] += sum_{r,h} glr[h][r][g] * xmid[r][h*32+c]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int g = pg + 8 * q;
            if (g < G) {
                float s = 0.f;
                for (int h = 0; h < heads; ++h)
                    for (int r = 0; r < 64; ++r) s += gll[(h * 64 + r) * G + g] * xml[r * (C + 1) + h * 32 + pc];
                accW[q] += s;
            }
        }
        __syncthreads();
    }
    float* prow = part + (long)blockIdx.x * ((long)G * 32 + G + heads);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int g = pg + 8 * q;
        if (g < G) prow[g * 32 + pc] = accW[q];
    }
    for (int idx = tid; idx < G + heads; idx += blockDim.x) prow[G * 32 + idx] = red[idx];
}

extern "C" int rpb_slice_bwd(const float* xf, const float* w, const float* gox, const float* tok2, const float* gT,
                             const float* gN, const float* Ws, const float* temp, float* gxf, float* part, int B, int ntok,
                             int heads, int G, void* stream) {
    RPB_REQUIRE(xf && w && gox && tok2 && gT && gN && Ws && temp && gxf && part, "slice_bwd: null pointer");
    RPB_REQUIRE(heads >= 1 && heads <= 8 && G >= 1 && G <= 32, "slice_bwd: heads=%d G=%d unsupported", heads, G);
    const int bps = rpb_slice_blocks_per_sample(B);
    const int C = heads * 32;
    const size_t lds = ((size_t)G * 33 + 2 * (size_t)heads * G * 33 + (size_t)heads * 64 * G + (size_t)64 * (C + 1) + G + heads) * 4;
    RPB_REQUIRE(lds <= 160 * 1024, "slice_bwd: LDS");
    (void)hipFuncSetAttribute((const void*)slice_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(slice_bwd_kernel, dim3(B * bps), dim3(TS_THREADS), lds, (hipStream_t)stream, xf, w, gox, tok2, gT, gN,
                       Ws, temp, gxf, part, ntok, heads, G, bps);
    RPB_CHECK_LAUNCH("slice_bwd");
}

// column sums: out_part[block][n] = sum over this block's rows of x[m][n]  (bias / placeholder gradients)
__global__ __launch_bounds__(TS_THREADS) void colsum_kernel(const float* __restrict__ x, float* __restrict__ part, long M,
                                                            int N, int ld) {
    extern __shared__ float red[];     // [nsub][N]
    const int n4n = N / 4;
    const int c4 = threadIdx.x % n4n, sub = threadIdx.x / n4n, nsub = blockDim.x / n4n;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (sub < nsub)
        for (long m = (long)blockIdx.x * nsub + sub; m < M; m += (long)gridDim.x * nsub)
            acc += *reinterpret_cast<const f32x4*>(x + m * ld + 4 * c4);
    if (sub < nsub) *reinterpret_cast<f32x4*>(red + sub * N + 4 * c4) = acc;
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nsub; ++k) s += red[k * N + n];
        part[(long)blockIdx.x * N + n] = s;
    }
}

extern "C" int rpb_colsum_rows(void) { return rpb_num_cus() * 4; }

extern "C" int rpb_colsum(const float* x, float* part, long M, int N, int ld, void* stream) {
    RPB_REQUIRE(x && part && M > 0 && N % 4 == 0 && N / 4 <= TS_THREADS && TS_THREADS % (N / 4) == 0 && ld % 4 == 0,
                "colsum: N=%d unsupported", N);
    const int nsub = TS_THREADS / (N / 4);
    hipLaunchKernelGGL(colsum_kernel, dim3(rpb_colsum_rows()), dim3(TS_THREADS), (size_t)nsub * N * 4, (hipStream_t)stream, x,
                       part, M, N, ld);
    RPB_CHECK_LAUNCH("colsum");
}
