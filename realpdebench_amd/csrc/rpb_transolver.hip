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
// nn.LayerNorm(C, eps=1e-5) with C = 64*V floats per token (V = 1..8, 12, 16 => C up to 512, 768, 1024); biased variance.
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
    RPB_REQUIRE(C % 64 == 0 && C >= 64 && (C <= 512 || C == 768 || C == 1024), "layernorm: C=%d must be a multiple of 64 up to 512, 768 or 1024", C);
    long grid = (M + 3) / 4;
    const long cap = (long)rpb_num_cus() * 8;
    if (grid > cap) grid = cap;
    hipStream_t st = (hipStream_t)stream;
#define RPB_LN(V_) \
    if (C == 64 * V_) hipLaunchKernelGGL((layernorm_kernel<V_>), dim3((unsigned)grid), dim3(TS_THREADS), 0, st, x, gamma, beta, out, M, eps);
    RPB_LN(1) RPB_LN(2) RPB_LN(3) RPB_LN(4) RPB_LN(5) RPB_LN(6) RPB_LN(7) RPB_LN(8) RPB_LN(12) RPB_LN(16)
#undef RPB_LN
    RPB_CHECK_LAUNCH("layernorm");
}

// ---------------------------------------------------------------------------------- slice: weights + token sums
// Physics_Attention.py:158-162 for dim_head = 32, heads = C/32, G <= 32 slices (G % 4 == 0):
//   w[m][h][g]   = softmax_g( (xmid[m][h*32:+32] . Ws[g] + bs[g]) / clamp(temp[h], 0.1, 5) )
//   norm[b][h][g] = sum_{m in b} w,     tokS[b][h][g][c] = sum_{m in b} fx[m][h*32+c] * w[m][h][g]
// xf: [M][ldx] rows holding fx_mid at column 0 and x_mid at column C (the dual convolution writes them side by side).
// Wave = head, 32-token tiles, both products on the fp32 MFMA (a first VALU version spent an LDS read per FMA: 0.67 TB/s):
//   logits^T[g][tok] = Ws x^T is accumulated transposed (row = slice, column = token) so that a lane owns one token and the
//   softmax over slices is register-local (+ one cross-half shuffle); the weights go through a wave-private LDS tile once,
//   which yields both the coalesced 16 B stores of w and the A operand (row = slice, k = token) of the token sums
//   tokS[g][c] += w^T fx, whose accumulator lives in registers across the tiles a wave walks.
#define SL_XS 33

struct SliceArgs {
    const float* xf;
    const float* Ws;
    const float* bs;
    const float* temp;
    float* w_out;
    float* tok_part;      // [B*bps][heads*G*32]
    float* norm_part;     // [B*bps][heads*G]
    const float* w_in;    // given weights (backward of deslice): no logits, token sums of xf[:, 0:C] only
    int ntok, heads, G, ldx, bps;
};

__global__ __launch_bounds__(512) void slice_fwd_kernel(SliceArgs a) {
    extern __shared__ float lds[];
    const int heads = a.heads, G = a.G, C = heads * 32;
    float* Wsl = lds;                                            // [32][33] slice projection, rows g >= G zero
    const int lane = threadIdx.x & 63;
    const int h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* xl = lds + 32 * SL_XS + h * 3 * 32 * SL_XS;           // wave-private: x_mid tile [32 tok][33]
    float* fl = xl + 32 * SL_XS;                                 //               fx tile    [32 tok][33]
    float* wl = fl + 32 * SL_XS;                                 //               weights    [32 tok][33] (columns g >= G zero)
    const int col = lane & 31, half = lane >> 5;
    const int b = blockIdx.x / a.bps, blk = blockIdx.x % a.bps;
    for (int idx = threadIdx.x; idx < 32 * SL_XS; idx += blockDim.x) {
        const int g = idx / SL_XS, c = idx - g * SL_XS;
        Wsl[idx] = (!a.w_in && g < G && c < 32) ? a.Ws[g * 32 + c] : 0.f;
    }
    for (int idx = lane; idx < 32 * SL_XS; idx += 64) wl[idx] = 0.f;
    __syncthreads();
    int jrow[16];
    float bsv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        jrow[r] = mfma_row(lane, r);
        bsv[r] = (!a.w_in && jrow[r] < G) ? a.bs[jrow[r]] : 0.f;
    }
    const float inv_t = a.w_in ? 1.f : 1.0f / fminf(fmaxf(a.temp[h], 0.1f), 5.0f);
    f32x16 tacc = zero16(), nacc = zero16();
    const long base = (long)b * a.ntok;
    const int g4 = G >> 2;
    for (int t0 = blk * 32; t0 < a.ntok; t0 += a.bps * 32) {
        // ---- stage the head's x_mid and fx rows of the tile (128 B per token and tensor)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = j * 64 + lane, row = idx >> 3, c4 = (idx & 7) * 4;
            const bool ok = t0 + row < a.ntok;
            const float* rp = a.xf + (base + t0 + row) * a.ldx + h * 32 + c4;
            f32x4 vf = {0.f, 0.f, 0.f, 0.f}, vx = vf;
            if (ok) vf = *reinterpret_cast<const f32x4*>(rp);
            if (ok && !a.w_in) vx = *reinterpret_cast<const f32x4*>(rp + C);
            float* df = fl + row * SL_XS + c4;
            df[0] = vf[0]; df[1] = vf[1]; df[2] = vf[2]; df[3] = vf[3];
            if (!a.w_in) {
                float* dx = xl + row * SL_XS + c4;
                dx[0] = vx[0]; dx[1] = vx[1]; dx[2] = vx[2]; dx[3] = vx[3];
            }
        }
        if (a.w_in) {                                             // given weights -> wl[tok][g]
            for (int idx = lane; idx < 32 * g4; idx += 64) {
                const int row = idx / g4, q = idx - row * g4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (t0 + row < a.ntok) v = *reinterpret_cast<const f32x4*>(a.w_in + ((base + t0 + row) * heads + h) * G + 4 * q);
                float* d = wl + row * SL_XS + 4 * q;
                d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
            }
            __builtin_amdgcn_wave_barrier();
        } else {
            __builtin_amdgcn_wave_barrier();
            // ---- logits^T[g][tok] = sum_d Ws[g][d] x[tok][d]; softmax over g in my registers (+ the other half-wave)
            f32x16 p = zero16();
#pragma unroll
            for (int s = 0; s < 16; ++s) p = mfma32(Wsl[col * SL_XS + 2 * s + half], xl[col * SL_XS + 2 * s + half], p);
            const bool tok_ok = t0 + col < a.ntok;
            float mx = -3.0e38f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] = (jrow[r] < G) ? (p[r] + bsv[r]) * inv_t : -3.0e38f;
                mx = fmaxf(mx, p[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float z = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] = (jrow[r] < G) ? expf(p[r] - mx) : 0.f;
                z += p[r];
            }
            z += __shfl_xor(z, 32, 64);
            const float iz = tok_ok ? 1.0f / z : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] *= iz;
                nacc[r] += p[r];
                if (jrow[r] < G) wl[col * SL_XS + jrow[r]] = p[r];
            }
            __builtin_amdgcn_wave_barrier();
            // ---- coalesced store of the tile's weights: 16 B per lane
            for (int idx = lane; idx < 32 * g4; idx += 64) {
                const int row = idx / g4, q = idx - row * g4;
                if (t0 + row < a.ntok) {
                    const float* sp = wl + row * SL_XS + 4 * q;
                    const f32x4 v = {sp[0], sp[1], sp[2], sp[3]};
                    *reinterpret_cast<f32x4*>(a.w_out + ((base + t0 + row) * heads + h) * G + 4 * q) = v;
                }
            }
        }
        // ---- token sums: tokS[g][c] += sum_tok w[tok][g] fx[tok][c]
#pragma unroll
        for (int s = 0; s < 16; ++s)
            tacc = mfma32(wl[(2 * s + half) * SL_XS + col], fl[(2 * s + half) * SL_XS + col], tacc);
        __builtin_amdgcn_wave_barrier();
    }
    float* prow = a.tok_part + (long)blockIdx.x * ((long)heads * G * 32);
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (jrow[r] < G) prow[((long)h * G + jrow[r]) * 32 + col] = tacc[r];
    if (!a.w_in) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {                            // norm[g] = sum over my half-wave's 32 token columns
            float v = nacc[r];
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 8, 64);
            v += __shfl_xor(v, 16, 64);
            if (col == 0 && jrow[r] < G) a.norm_part[(long)blockIdx.x * heads * G + h * G + jrow[r]] = v;
        }
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
    RPB_REQUIRE(heads >= 1 && heads <= 8 && G >= 4 && G <= 32 && G % 4 == 0 && ldx % 4 == 0,
                "slice_fwd: heads=%d G=%d ldx=%d unsupported (dim_head 32, slice_num a multiple of 4 up to 32)", heads, G, ldx);
    SliceArgs a{xf, Ws, bs, temp, w_out, tok_part, norm_part, w_in, ntok, heads, G, ldx, rpb_slice_blocks_per_sample(B)};
    const size_t lds = (size_t)(32 * SL_XS + heads * 3 * 32 * SL_XS) * 4;
    (void)hipFuncSetAttribute((const void*)slice_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(slice_fwd_kernel, dim3(B * a.bps), dim3(heads * 64), lds, (hipStream_t)stream, a);
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

// ---------------------------------------------------------------------------------- slice-token attention: training
// Physics_Attention.py:164-171 with nn.Dropout on the attention map (a given inverted-dropout mask amask[b][h][G][G], or none)
// and its full backward.  One block per (b, h); everything lives in LDS (G <= 32 tokens of 32 channels).
//   t = tokS / (norm + 1e-5);  q, k, v = t Wq^T, t Wk^T, t Wv^T;  P = softmax(q k^T / sqrt(32));  A = P * amask;  o = A v
__global__ __launch_bounds__(TS_THREADS) void slice_attn_train_kernel(const float* __restrict__ tokS, const float* __restrict__ norm,
                                                                      const float* __restrict__ Wq, const float* __restrict__ Wk,
                                                                      const float* __restrict__ Wv, const float* __restrict__ amask,
                                                                      const float* __restrict__ go, float* __restrict__ out,
                                                                      float* __restrict__ gT, float* __restrict__ gN,
                                                                      float* __restrict__ gW, int G, float scale) {
    // forward values, then (go != null) the gradients; gW [BH][3][32][32] per-block partials of d to_q / to_k / to_v weights
    __shared__ float t[32][33], q[32][33], k[32][33], v[32][33], p[32][33], a[32][33];
    __shared__ float g1[32][33], g2[32][33], g3[32][33], gs[32][33];
    const int bh = blockIdx.x;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < G * 32; idx += blockDim.x) {
        const int g = idx >> 5, c = idx & 31;
        t[g][c] = tokS[((long)bh * G + g) * 32 + c] / (norm[(long)bh * G + g] + 1e-5f);
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
        float sacc = 0.f;
        for (int c = 0; c < 32; ++c) sacc += q[i][c] * k[j][c];
        p[i][j] = sacc * scale;
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
        for (int j = 0; j < G; ++j) {
            p[tid][j] *= inv;
            a[tid][j] = amask ? p[tid][j] * amask[((long)bh * G + tid) * G + j] : p[tid][j];
        }
    }
    __syncthreads();
    if (out) {
        for (int idx = tid; idx < G * 32; idx += blockDim.x) {
            const int g = idx >> 5, c = idx & 31;
            float sacc = 0.f;
            for (int j = 0; j < G; ++j) sacc += a[g][j] * v[j][c];
            out[((long)bh * G + g) * 32 + c] = sacc;
        }
    }
    if (!go) return;
    // ---- backward.  g1 = go (then gq), g2 = gv (then gk), gs = dS
    for (int idx = tid; idx < G * 32; idx += blockDim.x) g1[idx >> 5][idx & 31] = go[(long)bh * G * 32 + idx];
    __syncthreads();
    for (int idx = tid; idx < G * G; idx += blockDim.x) {               // dA = go v^T, dP = dA * mask
        const int i = idx / G, j = idx - i * G;
        float sacc = 0.f;
        for (int c = 0; c < 32; ++c) sacc += g1[i][c] * v[j][c];
        gs[i][j] = amask ? sacc * amask[((long)bh * G + i) * G + j] : sacc;
    }
    for (int idx = tid; idx < G * 32; idx += blockDim.x) {              // gv = A^T go
        const int g = idx >> 5, c = idx & 31;
        float sacc = 0.f;
        for (int i = 0; i < G; ++i) sacc += a[i][g] * g1[i][c];
        g3[g][c] = sacc;
    }
    __syncthreads();
    if (tid < G) {                                                       // dS = scale * P * (dP - sum_j dP P)
        float dot = 0.f;
        for (int j = 0; j < G; ++j) dot += gs[tid][j] * p[tid][j];
        for (int j = 0; j < G; ++j) gs[tid][j] = scale * p[tid][j] * (gs[tid][j] - dot);
    }
    __syncthreads();
    for (int idx = tid; idx < G * 32; idx += blockDim.x) {              // gq = dS k,  gk = dS^T q
        const int g = idx >> 5, c = idx & 31;
        float sq = 0.f, sk = 0.f;
        for (int j = 0; j < G; ++j) {
            sq += gs[g][j] * k[j][c];
            sk += gs[j][g] * q[j][c];
        }
        g1[g][c] = sq;                                                   // go is dead (consumed above, barrier passed)
        g2[g][c] = sk;
    }
    __syncthreads();
    // d W*[c][j] = sum_g g*[g][c] t[g][j]   (per-block partials)
    for (int idx = tid; idx < 32 * 32; idx += blockDim.x) {
        const int c = idx >> 5, j = idx & 31;
        float sq = 0.f, sk = 0.f, sv = 0.f;
        for (int g = 0; g < G; ++g) {
            const float tv = t[g][j];
            sq += g1[g][c] * tv;
            sk += g2[g][c] * tv;
            sv += g3[g][c] * tv;
        }
        float* w = gW + (long)bh * 3 * 1024;
        w[idx] = sq;
        w[1024 + idx] = sk;
        w[2048 + idx] = sv;
    }
    // gt = gq Wq + gk Wk + gv Wv;  t = tokS / (norm + eps):  d tokS = gt / (norm + eps),  d norm = -sum_j gt t / (norm + eps)
    for (int idx = tid; idx < G * 32; idx += blockDim.x) {
        const int g = idx >> 5, j = idx & 31;
        float sacc = 0.f;
        for (int c = 0; c < 32; ++c) sacc += g1[g][c] * Wq[c * 32 + j] + g2[g][c] * Wk[c * 32 + j] + g3[g][c] * Wv[c * 32 + j];
        a[g][j] = sacc;                                                  // A is dead: reuse as gt
    }
    __syncthreads();
    for (int idx = tid; idx < G * 32; idx += blockDim.x) {
        const int g = idx >> 5, j = idx & 31;
        gT[((long)bh * G + g) * 32 + j] = a[g][j] / (norm[(long)bh * G + g] + 1e-5f);
    }
    if (tid < G) {
        float sacc = 0.f;
        for (int j = 0; j < 32; ++j) sacc += a[tid][j] * t[tid][j];
        gN[(long)bh * G + tid] = -sacc / (norm[(long)bh * G + tid] + 1e-5f);
    }
}

// out (optional) = the attended slice tokens under the attention-map dropout mask amask (optional).  With go: the backward --
// gT [BH][G][32], gN [BH][G] and gW [BH][3][32][32] (per-(b,h) partials of d to_q / to_k / to_v, summed by the caller).
extern "C" int rpb_slice_attn_train(const float* tokS, const float* norm, const float* Wq, const float* Wk, const float* Wv,
                                    const float* amask, const float* go, float* out, float* gT, float* gN, float* gW, int BH,
                                    int G, void* stream) {
    RPB_REQUIRE(tokS && norm && Wq && Wk && Wv && BH > 0 && G >= 1 && G <= 32, "slice_attn_train: bad arguments");
    RPB_REQUIRE(out || go, "slice_attn_train: nothing to compute");
    if (go) RPB_REQUIRE(gT && gN && gW, "slice_attn_train: the backward needs gT, gN and gW");
    hipLaunchKernelGGL(slice_attn_train_kernel, dim3(BH), dim3(TS_THREADS), 0, (hipStream_t)stream, tokS, norm, Wq, Wk, Wv,
                       amask, go, out, gT, gN, gW, G, 1.0f / sqrtf(32.f));
    RPB_CHECK_LAUNCH("slice_attn_train");
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
    RPB_REQUIRE(C % 64 == 0 && C >= 64 && (C <= 512 || C == 768 || C == 1024), "layernorm_bwd: C=%d must be a multiple of 64 up to 512, 768 or 1024", C);
    const unsigned grid = (unsigned)(rpb_layernorm_bwd_rows(M) / (TS_THREADS / 64));
    hipStream_t st = (hipStream_t)stream;
#define RPB_LNB(V_) \
    if (C == 64 * V_) hipLaunchKernelGGL((layernorm_bwd_kernel<V_>), dim3(grid), dim3(TS_THREADS), 0, st, x, gamma, gy, gadd, gx, part, M, eps);
    RPB_LNB(1) RPB_LNB(2) RPB_LNB(3) RPB_LNB(4) RPB_LNB(5) RPB_LNB(6) RPB_LNB(7) RPB_LNB(8) RPB_LNB(12) RPB_LNB(16)
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
__global__ __launch_bounds__(512) void slice_bwd_kernel(const float* __restrict__ xf, const float* __restrict__ w,
                                                        const float* __restrict__ gox, const float* __restrict__ tok2,
                                                        const float* __restrict__ gT, const float* __restrict__ gN,
                                                        const float* __restrict__ Ws, const float* __restrict__ temp,
                                                        float* __restrict__ gxf, float* __restrict__ part, int ntok,
                                                        int heads, int G, int blocks_per_sample) {
    // wave = head, 32-token tiles, every product on the fp32 MFMA; transposed accumulators (row = slice, column = token)
    // keep the softmax Jacobian register-local.  B operands whose contraction index is the channel are read straight from
    // global memory as 64 B per lane (the contraction order is free: step s of half-wave q contracts channel 16 q + s), the
    // ones whose contraction index is the token as coalesced 128 B rows; only the slice weights and their logit gradients
    // (both [32 tok][G]) pass through LDS, to be re-read as A operands.
    extern __shared__ float lds[];
    const int C = heads * 32, GS = G + 1;
    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
    const int h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* Wsl = lds;                                           // [G][33]
    float* t2l = Wsl + G * 33 + h * (2 * G * 33 + 2 * 32 * GS); // wave-private: tok2[h] [G][33]
    float* gtl = t2l + G * 33;                                  //               gT[h]   [G][33]
    float* wl = gtl + G * 33;                                   //               w tile  [32][GS]
    float* gll = wl + 32 * GS;                                  //               raw-logit gradients [32][GS]
    const int b = blockIdx.x / blocks_per_sample, blk = blockIdx.x % blocks_per_sample;
    for (int idx = threadIdx.x; idx < G * 32; idx += blockDim.x) Wsl[(idx >> 5) * 33 + (idx & 31)] = Ws[idx];
    for (int idx = lane; idx < G * 32; idx += 64) {
        const long src = ((long)b * heads + h) * G * 32 + idx;
        t2l[(idx >> 5) * 33 + (idx & 31)] = tok2[src];
        gtl[(idx >> 5) * 33 + (idx & 31)] = gT[src];
    }
    __syncthreads();
    int jrow[16];
    float gNv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        jrow[r] = mfma_row(lane, r);
        gNv[r] = jrow[r] < G ? gN[((long)b * heads + h) * G + jrow[r]] : 0.f;
    }
    const bool arow = col < G;                                  // A operands with row = slice: rows >= G are zero
    const int colg = arow ? col : G - 1;
    const float inv_t = 1.0f / fminf(fmaxf(temp[h], 0.1f), 5.0f);
    const int g4 = G >> 2, gh = G >> 1;
    f32x16 wacc = zero16(), racc = zero16();
    float dtau = 0.f;
    const long base = (long)b * ntok;
    for (int t0 = blk * 32; t0 < ntok; t0 += blocks_per_sample * 32) {
        const bool ok = t0 + col < ntok;
        const long m = base + t0 + col;
        float gov[16], fxv[16], xmv[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = a;
            if (ok) {
                a = *reinterpret_cast<const f32x4*>(gox + m * C + h * 32 + half * 16 + 4 * k);
                c = *reinterpret_cast<const f32x4*>(xf + m * 2 * C + h * 32 + half * 16 + 4 * k);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                gov[4 * k + i] = a[i];
                fxv[4 * k + i] = c[i];
            }
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int tr = t0 + 2 * s + half;
            xmv[s] = tr < ntok ? xf[(base + tr) * 2 * C + C + h * 32 + col] : 0.f;
        }
        for (int idx = lane; idx < 32 * g4; idx += 64) {
            const int row = idx / g4, q = idx - row * g4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (t0 + row < ntok) v = *reinterpret_cast<const f32x4*>(w + ((base + t0 + row) * heads + h) * G + 4 * q);
            float* d = wl + row * GS + 4 * q;
            d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        }
        __builtin_amdgcn_wave_barrier();
        // ---- gw^T[g][tok] = gN[g] + sum_c tok2[g][c] go[tok][c] + gT[g][c] fx[tok][c]
        f32x16 p = zero16();
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float a1 = t2l[colg * 33 + half * 16 + s], a2 = gtl[colg * 33 + half * 16 + s];
            p = mfma32(arow ? a1 : 0.f, gov[s], p);
            p = mfma32(arow ? a2 : 0.f, fxv[s], p);
        }
        float wv[16], dot = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            wv[r] = jrow[r] < G ? wl[col * GS + jrow[r]] : 0.f;
            p[r] += gNv[r];
            dot += wv[r] * p[r];
        }
        dot += __shfl_xor(dot, 32, 64);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float gl = wv[r] * (p[r] - dot);                // d / d(scaled logit)
            const float glr = gl * inv_t;                          // d / d(raw logit)
            if (wv[r] > 0.f) dtau -= gl * logf(wv[r]);
            racc[r] += glr;
            if (jrow[r] < G) gll[col * GS + jrow[r]] = glr;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- g_fx[tok][c] = sum_g w[tok][g] gT[g][c];  g_xmid[tok][c] = sum_g glr[tok][g] Ws[g][c]
        f32x16 gfx = zero16(), gxm = zero16();
        for (int s = 0; s < gh; ++s) {
            gfx = mfma32(wl[col * GS + 2 * s + half], gtl[(2 * s + half) * 33 + col], gfx);
            gxm = mfma32(gll[col * GS + 2 * s + half], Wsl[(2 * s + half) * 33 + col], gxm);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (t0 + jrow[r] < ntok) {
                float* dst = gxf + (base + t0 + jrow[r]) * 2 * C + h * 32 + col;
                dst[0] = gfx[r];
                dst[C] = gxm[r];
            }
        }
        // ---- dWs[g][c] += sum_tok glr[tok][g] xmid[tok][c]
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float a1 = gll[(2 * s + half) * GS + colg];
            wacc = mfma32(arow ? a1 : 0.f, xmv[s], wacc);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // ---- block sums in a fixed order (deterministic): each wave parks its partials in its own tiles
    float* wpark = wl;                                            // [G][32]   (64 (G + 1) >= 32 G floats)
    float* rpark = t2l;                                           // [G] + [1]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (jrow[r] < G) wpark[jrow[r] * 32 + col] = wacc[r];
        float v = racc[r];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        if (col == 0 && jrow[r] < G) rpark[jrow[r]] = v;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) dtau += __shfl_xor(dtau, o, 64);
    if (lane == 0) rpark[G] = dtau * inv_t;
    __syncthreads();
    const int wstride = 2 * G * 33 + 2 * 32 * GS;
    float* prow = part + (long)blockIdx.x * ((long)G * 32 + G + heads);
    const float* w0 = lds + G * 33 + 2 * G * 33;                 // wave 0's wl
    const float* r0 = lds + G * 33;                               // wave 0's t2l
    for (int idx = threadIdx.x; idx < G * 32; idx += blockDim.x) {
        float sacc = 0.f;
        for (int hh = 0; hh < heads; ++hh) sacc += w0[hh * wstride + idx];
        prow[idx] = sacc;
    }
    for (int idx = threadIdx.x; idx < G; idx += blockDim.x) {
        float sacc = 0.f;
        for (int hh = 0; hh < heads; ++hh) sacc += r0[hh * wstride + idx];
        prow[G * 32 + idx] = sacc;
    }
    for (int idx = threadIdx.x; idx < heads; idx += blockDim.x) prow[G * 32 + G + idx] = r0[idx * wstride + G];
}

extern "C" int rpb_slice_bwd(const float* xf, const float* w, const float* gox, const float* tok2, const float* gT,
                             const float* gN, const float* Ws, const float* temp, float* gxf, float* part, int B, int ntok,
                             int heads, int G, void* stream) {
    RPB_REQUIRE(xf && w && gox && tok2 && gT && gN && Ws && temp && gxf && part, "slice_bwd: null pointer");
    RPB_REQUIRE(heads >= 1 && heads <= 8 && G >= 4 && G <= 32 && G % 4 == 0, "slice_bwd: heads=%d G=%d unsupported", heads, G);
    const int bps = rpb_slice_blocks_per_sample(B);
    const size_t lds = ((size_t)G * 33 + (size_t)heads * (2 * G * 33 + 2 * 32 * (G + 1))) * 4;
    RPB_REQUIRE(lds <= 160 * 1024, "slice_bwd: LDS");
    (void)hipFuncSetAttribute((const void*)slice_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(slice_bwd_kernel, dim3(B * bps), dim3(heads * 64), lds, (hipStream_t)stream, xf, w, gox, tok2, gT, gN,
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
