// DPOT (AFNO patch transformer) kernels -- SURVEY.md section 8 row f4: the pieces of
// realpdebench/model/dpot_libs/models/dpot.py that are not plain token GEMMs (those run on rpb_gemm_nt / rpb_gemm3x / rpb_gemm_tn)
// or DFT stages (rpb_axis_gemm):
//
//   rpb_dpot_patch_tokens      PatchEmbed input gather (dpot.py:183-211 fed by DPOTNet.forward :366-373 and the wrapper's channel
//                              padding, model/dpot.py:213-221): one token row per (sample, patch, frame)
//   rpb_rowtable_add / _grad   x + pos_embed (dpot.py:375) on that token order, and d pos_embed
//   rpb_dpot_tagg_prep/_finish TimeAggregator 'exp_mlp' (dpot.py:227-241): weights scaled by cos(t gamma) in both GEMM layouts; d w, d gamma
//   rpb_gn_tokens_fwd / _bwd   torch.nn.GroupNorm(8, width) on channels-last tokens (dpot.py:143,151,166,174): wave / workgroup reductions
//   rpb_afno_wprep / _mlp / _wgrad / _wfinish
//                              AFNO2D's block-diagonal complex two-layer MLP on the kept modes (dpot.py:72-94) as a real GEMM pair per
//                              block on the fp32 MFMA, its data gradient (same kernel, transposed weights) and its weight gradient
//   rpb_dpot_unpatch(_bwd)     out_layer output rows (pixel-major) -> [B][T_out][H][W][C_data] (dpot.py:395-396, model/dpot.py:227,235)
#include "rpb_common.h"
#include <math.h>

namespace {

// ------------------------------------------------------------------------------------------------ PatchEmbed gather
// P[((b*nx + px)*ny + py)*T + t][(c*ps + i)*ps + j]:  c < Cd data, Cd <= c < Cm ones, then grid x, grid y, grid t
__global__ void patch_tokens_kernel(const float* __restrict__ u, const float* __restrict__ gx, const float* __restrict__ gy,
                                    const float* __restrict__ gt, float* __restrict__ P, int B, int T, int H, int W, int Cd, int Cm,
                                    int ps, long total) {
    const int Kp = (Cm + 3) * ps * ps;
    const int nx = H / ps, ny = W / ps;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int f = (int)(idx % Kp);
        long r = idx / Kp;
        const int t = (int)(r % T);
        r /= T;
        const int py = (int)(r % ny);
        r /= ny;
        const int px = (int)(r % nx);
        const int b = (int)(r / nx);
        const int j = f % ps, i = (f / ps) % ps, c = f / (ps * ps);
        const int x = px * ps + i, y = py * ps + j;
        float v;
        if (c < Cd) v = u[((((long)b * T + t) * H + x) * W + y) * Cd + c];
        else if (c < Cm) v = 1.0f;
        else if (c == Cm) v = gx[x];
        else if (c == Cm + 1) v = gy[y];
        else v = gt[t];
        P[idx] = v;
    }
}

// gradient of the gather above w.r.t. the data channels: patches do not overlap, so every input element has exactly one token entry
__global__ void patch_tokens_bwd_kernel(const float* __restrict__ gP, float* __restrict__ gu, int B, int T, int H, int W, int Cd, int Cm,
                                        int ps, long total) {
    const int Kp = (Cm + 3) * ps * ps;
    const int nx = H / ps, ny = W / ps;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % Cd);
        long r = idx / Cd;
        const int y = (int)(r % W);
        r /= W;
        const int x = (int)(r % H);
        r /= H;
        const int t = (int)(r % T);
        const int b = (int)(r / T);
        const long row = (((long)b * nx + x / ps) * ny + y / ps) * T + t;
        gu[idx] = gP[row * Kp + (c * ps + x % ps) * ps + y % ps];
    }
}

// x[r][c] += table[(r / rpe) % nent][c]
__global__ void rowtable_add_kernel(float* __restrict__ x, const float* __restrict__ table, long M, int C, int rpe, int nent) {
    const long n4 = M * (C / 4);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n4; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / (C / 4);
        const int c4 = (int)(idx % (C / 4));
        const int e = (int)((r / rpe) % nent);
        f32x4 v = reinterpret_cast<f32x4*>(x)[idx];
        v += reinterpret_cast<const f32x4*>(table)[(long)e * (C / 4) + c4];
        reinterpret_cast<f32x4*>(x)[idx] = v;
    }
}
// dtable[e][c] = sum_{b, t} g[((b*nent + e)*rpe + t)][c]   (fp64 accumulation, fixed order)
__global__ void rowtable_grad_kernel(const float* __restrict__ g, float* __restrict__ dt, int B, int C, int rpe, int nent) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)nent * C) return;
    const int c = (int)(idx % C), e = (int)(idx / C);
    double s = 0.0;
    for (int b = 0; b < B; ++b) {
        const float* p = g + (((long)b * nent + e) * rpe) * C + c;
        for (int t = 0; t < rpe; ++t) s += (double)p[(long)t * C];
    }
    dt[idx] = (float)s;
}

// ------------------------------------------------------------------------------------------------ TimeAggregator
// e[t][i] = cos(tt[t] * gamma[i]);  Wb[(t,i)][j] = e w[t][i][j]  (data-gradient GEMM, W = [N = T*C][K = C]);
// Wf[j][(t,i)] = the same transposed (forward GEMM, W = [N = C][K = T*C])
__global__ void tagg_prep_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ tt,
                                 float* __restrict__ Wf, float* __restrict__ Wb, float* __restrict__ e_out, int T, int C) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;                  // 256 threads: 32 x 8
    for (int r = ly; r < 32; r += 8) {
        const int i = i0 + r, j = j0 + lx;
        const float e = cosf(tt[t] * gamma[i]);
        const float v = e * w[((long)t * C + i) * C + j];
        Wb[((long)t * C + i) * C + j] = v;
        tile[r][lx] = v;
        if (blockIdx.x == 0 && lx == 0) e_out[t * C + i] = e;
    }
    __syncthreads();
    for (int r = ly; r < 32; r += 8) {
        const int j = j0 + r, i = i0 + lx;
        Wf[(long)j * T * C + (long)t * C + i] = tile[lx][r];
    }
}
// dWb [(t,i)][j] (+ dWsum [i][j] for every t) -> dw[t][i][j] = e[t][i] dWb;  dgamma[i] = sum_t -sin(tt[t] gamma[i]) tt[t] sum_j dWb[(t,i)][j] w[t][i][j]
__global__ void tagg_finish_kernel(const float* __restrict__ dWb, const float* __restrict__ dWsum, const float* __restrict__ w,
                                   const float* __restrict__ gamma, const float* __restrict__ tt, float* __restrict__ dw,
                                   float* __restrict__ dgamma, int T, int C) {
    const int i = blockIdx.x;
    const float ga = gamma[i];
    double acc = 0.0;
    for (int t = 0; t < T; ++t) {
        const float arg = tt[t] * ga;
        const float e = cosf(arg), de = -sinf(arg) * tt[t];
        const long base = ((long)t * C + i) * C;
        double s = 0.0;
        for (int j = threadIdx.x; j < C; j += blockDim.x) {
            const float d = dWb[base + j] + (dWsum ? dWsum[(long)i * C + j] : 0.f);   // + the gradient through sum_t Wb_t (composite path)
            dw[base + j] = e * d;
            s += (double)d * (double)w[base + j];
        }
        acc += s * (double)de;
    }
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = blockDim.x / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) dgamma[i] = (float)red[0];
}

// ------------------------------------------------------------------------------------------------ GroupNorm on tokens
// x (+ x2) [B][P][C], groups of cg = C / G consecutive channels; workgroup = (b, g), thread = (channel c = tid % cg, slice s = tid / cg)
__device__ __forceinline__ double block_sum(double v, double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}

__global__ void gn_tokens_fwd_kernel(const float* __restrict__ x, const float* __restrict__ x2, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ stat, int P, int C,
                                     int G, int S, float eps) {
    __shared__ double red[16];
    const int cg = C / G;
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const int c = threadIdx.x % cg, s = threadIdx.x / cg;
    const bool act = s < S;
    const long base = (long)b * P * C + g * cg + c;
    double sum = 0.0;
    if (act)
#pragma unroll 4
        for (int p = s; p < P; p += S) {
            float v = x[base + (long)p * C];
            if (x2) v += x2[base + (long)p * C];
            sum += (double)v;
        }
    const double n = (double)P * cg;
    const double mean = block_sum(sum, red) / n;
    double sq = 0.0;
    if (act)
#pragma unroll 4
        for (int p = s; p < P; p += S) {
            float v = x[base + (long)p * C];
            if (x2) v += x2[base + (long)p * C];
            const double d = (double)v - mean;
            sq += d * d;
        }
    const double var = block_sum(sq, red) / n;
    const float mu = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) {
        stat[2 * blockIdx.x] = mu;
        stat[2 * blockIdx.x + 1] = rstd;
    }
    if (act) {
        const float ga = gamma[g * cg + c], be = beta[g * cg + c];
#pragma unroll 4
        for (int p = s; p < P; p += S) {
            float v = x[base + (long)p * C];
            if (x2) v += x2[base + (long)p * C];
            y[base + (long)p * C] = (v - mu) * rstd * ga + be;
        }
    }
}

// gx = rstd * (gh - mean_g(gh) - xhat mean_g(gh xhat)) (+ gadd), gh = gy * gamma;  pg[b][c] = sum_p gy xhat, pb[b][c] = sum_p gy
__global__ void gn_tokens_bwd_kernel(const float* __restrict__ x, const float* __restrict__ x2, const float* __restrict__ gamma,
                                     const float* __restrict__ stat, const float* __restrict__ gy, const float* __restrict__ gadd,
                                     float* __restrict__ gx, float* __restrict__ pg, float* __restrict__ pb, int P, int C, int G,
                                     int S) {
    __shared__ double red[16];
    extern __shared__ float chan[];                  // [2][S][cg]
    const int cg = C / G;
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const int c = threadIdx.x % cg, s = threadIdx.x / cg;
    const bool act = s < S;
    const long base = (long)b * P * C + g * cg + c;
    const float mu = stat[2 * blockIdx.x], rstd = stat[2 * blockIdx.x + 1];
    const float ga = act ? gamma[g * cg + c] : 0.f;
    double s1 = 0.0, s2 = 0.0;
    float dg = 0.f, db = 0.f;
    if (act)
#pragma unroll 4
        for (int p = s; p < P; p += S) {
            float v = x[base + (long)p * C];
            if (x2) v += x2[base + (long)p * C];
            const float xh = (v - mu) * rstd;
            const float gyv = gy[base + (long)p * C];
            const float gh = gyv * ga;
            s1 += (double)gh;
            s2 += (double)gh * (double)xh;
            dg += gyv * xh;
            db += gyv;
        }
    const double n = (double)P * cg;
    const float m1 = (float)(block_sum(s1, red) / n);
    const float m2 = (float)(block_sum(s2, red) / n);
    if (act) {
        chan[s * cg + c] = dg;
        chan[(S + s) * cg + c] = db;
    }
    __syncthreads();
    if (act && s == 0) {
        float a = 0.f, bb = 0.f;
        for (int q = 0; q < S; ++q) {
            a += chan[q * cg + c];
            bb += chan[(S + q) * cg + c];
        }
        pg[(long)b * C + g * cg + c] = a;
        pb[(long)b * C + g * cg + c] = bb;
    }
    if (act)
#pragma unroll 4
        for (int p = s; p < P; p += S) {
            float v = x[base + (long)p * C];
            if (x2) v += x2[base + (long)p * C];
            const float xh = (v - mu) * rstd;
            const float gh = gy[base + (long)p * C] * ga;
            float o = rstd * (gh - m1 - xh * m2);
            if (gadd) o += gadd[base + (long)p * C];
            gx[base + (long)p * C] = o;
        }
}

// ------------------------------------------------------------------------------------------------ AFNO block MLP
// Complex weights w [2][nb][bs_in][bs_out] (re, im; einsum '...bi,bio->...bo') -> real composite per block
//   Wc[k][(ri_in, i)][(ri_out, o)]:  (0,i)->(0,o) = wr, (1,i)->(0,o) = -wi, (0,i)->(1,o) = wi, (1,i)->(1,o) = wr
// transpose != 0 stores Wc^T (the data-gradient operand).
__global__ void afno_wprep_kernel(const float* __restrict__ w, float* __restrict__ Wc, int nb, int bsi, int bso, int transpose) {
    const long total = (long)nb * 4 * bsi * bso;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int o = (int)(idx % bso);
        long r = idx / bso;
        const int ro = (int)(r % 2);
        r /= 2;
        const int i = (int)(r % bsi);
        r /= bsi;
        const int ri = (int)(r % 2);
        const int k = (int)(r / 2);
        const float wr = w[(((long)0 * nb + k) * bsi + i) * bso + o], wi = w[(((long)1 * nb + k) * bsi + i) * bso + o];
        const float v = (ri == ro) ? wr : (ri == 1 ? -wi : wi);
        const int K = 2 * bsi, N = 2 * bso;
        const int kk = ri * bsi + i, nn = ro * bso + o;
        if (transpose) Wc[((long)k * N + nn) * K + kk] = v;
        else Wc[((long)k * K + kk) * N + nn] = v;
    }
}

// One workgroup = (32-token tile, block k); wave w owns output columns [32 w, 32 w + 32) of both layers.
//   mode 0 (forward):  h = X Wa + ba;  pre_out <- h (if given);  H = gelu(h);  out = H Wb + bb
//   mode 1 (backward): h = (X Wa) * gelu'(aux);  mid_out <- h;  out = h Wb
// X, aux, pre_out / mid_out, out: [ntok][2][C] rows, the block's slice = columns [k*bs, +bs) of both halves.
// Layer widths: K1 = 2*bs1 -> N1 = 2*bs2 -> N2 = 2*bs3 (forward bs, bs*f, bs; the reference uses hidden_size_factor f = 1).
struct AfnoArgs {
    const float* X;
    const float* Wa;
    const float* ba;
    const float* Wb;
    const float* bb;
    const float* aux;
    float* mid;
    float* out;
    long ntok;
    int nb, bs, C, mode;
};

__global__ __launch_bounds__(1024) void afno_mlp_kernel(AfnoArgs a) {
    extern __shared__ float lds[];
    const int bs = a.bs, K = 2 * bs, LD = K + 1;
    float* Xs = lds;                    // [32][LD]
    float* Hs = lds + 32 * LD;          // [32][LD]
    const int k = blockIdx.y;
    const long tok0 = (long)blockIdx.x * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    const long rowld = 2L * a.C;
    // ---- stage the token tile: Xs[m][ri*bs + i] = X[tok0+m][ri][k*bs + i]
    for (int idx = tid; idx < 32 * K; idx += blockDim.x) {
        const int m = idx / K, kk = idx % K;
        const int ri = kk / bs, i = kk % bs;
        const long tok = tok0 + m;
        Xs[m * LD + kk] = tok < a.ntok ? a.X[tok * rowld + (long)ri * a.C + k * bs + i] : 0.f;
    }
    __syncthreads();
    const int n0 = wave * 32;
    const int ro = (n0 + col) / bs, oo = (n0 + col) % bs;        // this lane's output column -> (re/im half, channel in block)
    // ---- layer A
    f32x16 acc = zero16();
    {
        const float* Wp = a.Wa + (long)k * K * K + n0 + col;
        for (int s = 0; s < K; s += 16) {              // K is a multiple of 32: eight steps' operands are requested before the first MFMA
            float xa[8], wb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                xa[u] = Xs[col * LD + s + 2 * u + half];
                wb[u] = Wp[(long)(s + 2 * u + half) * K];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = mfma32(xa[u], wb[u], acc);
        }
    }
    const float bav = (a.mode == 0 && a.ba) ? a.ba[(long)ro * a.C + k * bs + oo] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mfma_row(lane, r);
        const long tok = tok0 + m;
        const long gi = tok * rowld + (long)ro * a.C + k * bs + oo;
        float h = acc[r] + bav;
        if (a.mode == 0) {
            if (a.mid && tok < a.ntok) a.mid[gi] = h;
            h = gelu_f(h);
        } else {
            h = tok < a.ntok ? h * gelu_grad_f(a.aux[gi]) : 0.f;
            if (tok < a.ntok) a.mid[gi] = h;
        }
        Hs[m * LD + n0 + col] = h;
    }
    __syncthreads();
    // ---- layer B
    acc = zero16();
    {
        const float* Wp = a.Wb + (long)k * K * K + n0 + col;
        for (int s = 0; s < K; s += 16) {
            float xa[8], wb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                xa[u] = Hs[col * LD + s + 2 * u + half];
                wb[u] = Wp[(long)(s + 2 * u + half) * K];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = mfma32(xa[u], wb[u], acc);
        }
    }
    const float bbv = (a.mode == 0 && a.bb) ? a.bb[(long)ro * a.C + k * bs + oo] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long tok = tok0 + mfma_row(lane, r);
        if (tok < a.ntok) a.out[tok * rowld + (long)ro * a.C + k * bs + oo] = acc[r] + bbv;
    }
}

// dWc partials: part[split][k][kk][nn] = sum_{tok in split} A[tok][kk] G[tok][nn]  (A optionally through GELU: the hidden layer)
// grid = (tiles = (K/32)^2, nb, splits), one wave per workgroup
__global__ __launch_bounds__(64) void afno_wgrad_kernel(const float* __restrict__ A, const float* __restrict__ G, float* __restrict__ part,
                                                         long ntok, int nb, int bs, int C, int a_gelu, int splits) {
    const int K = 2 * bs, nt = K / 32;
    const int ti = blockIdx.x / nt, tj = blockIdx.x % nt;
    const int k = blockIdx.y, sp = blockIdx.z;
    const int lane = threadIdx.x, col = lane & 31, half = lane >> 5;
    const int ka = ti * 32 + col, kb = tj * 32 + col;
    const long offa = (long)(ka / bs) * C + k * bs + ka % bs, offb = (long)(kb / bs) * C + k * bs + kb % bs;
    const long per = ((ntok + splits - 1) / splits + 1) & ~1L;
    const long t0 = sp * per, t1 = (t0 + per < ntok) ? t0 + per : ntok;
    const long rowld = 2L * C;
    f32x16 acc = zero16();
    for (long t = t0; t < t1; t += 16) {               // eight MFMA steps (16 tokens) per trip: all loads first
        float av[8], gv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long tok = t + 2 * u + half;
            av[u] = 0.f;
            gv[u] = 0.f;
            if (tok < t1) {
                av[u] = A[tok * rowld + offa];
                gv[u] = G[tok * rowld + offb];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = mfma32(a_gelu ? gelu_f(av[u]) : av[u], gv[u], acc);      // gelu(0) == 0 for the masked tail
    }
    float* p = part + (((long)sp * nb + k) * K) * K;
#pragma unroll
    for (int r = 0; r < 16; ++r) p[(long)(ti * 32 + mfma_row(lane, r)) * K + tj * 32 + col] = acc[r];
}
// dw[0][k][i][o] = sum_s P[(0,i)][(0,o)] + P[(1,i)][(1,o)];  dw[1][k][i][o] = sum_s P[(0,i)][(1,o)] - P[(1,i)][(0,o)]
__global__ void afno_wfinish_kernel(const float* __restrict__ part, float* __restrict__ dw, int nb, int bs, int splits) {
    const long total = (long)nb * bs * bs;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int o = (int)(idx % bs), i = (int)((idx / bs) % bs), k = (int)(idx / ((long)bs * bs));
    const int K = 2 * bs;
    double re = 0.0, im = 0.0;
    for (int s = 0; s < splits; ++s) {
        const float* p = part + (((long)s * nb + k) * K) * K;
        re += (double)p[(long)i * K + o] + (double)p[(long)(bs + i) * K + bs + o];
        im += (double)p[(long)i * K + bs + o] - (double)p[(long)(bs + i) * K + o];
    }
    dw[idx] = (float)re;
    dw[total + idx] = (float)im;
}

// ------------------------------------------------------------------------------------------------ output re-layout
// O[(((b*nx + px)*ny + py)*ps + i)*ps + j][ldo] (column t*Co + c) <-> pred[b][t][px*ps + i][py*ps + j][c], c < Cd
__global__ void unpatch_kernel(const float* __restrict__ O, float* __restrict__ pred, int B, int T, int H, int W, int Cd, int Co, int ps,
                               int ldo, long total, int backward, float* __restrict__ gO) {
    const int nx = H / ps, ny = W / ps;
    if (!backward) {
        for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
            const int c = (int)(idx % Cd);
            long r = idx / Cd;
            const int y = (int)(r % W);
            r /= W;
            const int x = (int)(r % H);
            r /= H;
            const int t = (int)(r % T);
            const int b = (int)(r / T);
            const long row = ((((long)b * nx + x / ps) * ny + y / ps) * ps + x % ps) * ps + y % ps;
            pred[idx] = O[row * ldo + t * Co + c];
        }
    } else {            // gO[row][col] = gpred[...] for col = t*Co + c with c < Cd, else 0  (total = rows * ldo)
        for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
            const int colo = (int)(idx % ldo);
            long row = idx / ldo;
            const int t = colo / Co, c = colo % Co;
            float v = 0.f;
            if (t < T && c < Cd) {
                const int j = (int)(row % ps);
                row /= ps;
                const int i = (int)(row % ps);
                row /= ps;
                const int py = (int)(row % ny);
                row /= ny;
                const int px = (int)(row % nx);
                const int b = (int)(row / nx);
                v = pred[((((long)b * T + t) * H + px * ps + i) * W + py * ps + j) * Cd + c];
            }
            gO[idx] = v;
        }
    }
}

int grid_for(long n, int block) {
    long g = (n + block - 1) / block;
    const long cap = (long)rpb_num_cus() * 16;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}
}  // namespace

extern "C" int rpb_dpot_patch_tokens(const float* u, const float* gx, const float* gy, const float* gt, float* P, int B, int T, int H,
                                     int W, int Cd, int Cm, int ps, void* stream) {
    RPB_REQUIRE(u && gx && gy && gt && P, "dpot_patch_tokens: null pointer");
    RPB_REQUIRE(B > 0 && T > 0 && ps > 0 && H % ps == 0 && W % ps == 0 && Cd > 0 && Cd <= Cm, "dpot_patch_tokens: bad sizes");
    const long total = (long)B * (H / ps) * (W / ps) * T * (Cm + 3) * ps * ps;
    hipLaunchKernelGGL(patch_tokens_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, u, gx, gy, gt, P, B, T, H, W,
                       Cd, Cm, ps, total);
    RPB_CHECK_LAUNCH("dpot_patch_tokens");
}

// gu [B][T][H][W][Cd] = gradient w.r.t. the data frames from gP [B*nx*ny*T][(Cm+3)*ps*ps], the gradient of the token rows
extern "C" int rpb_dpot_patch_tokens_bwd(const float* gP, float* gu, int B, int T, int H, int W, int Cd, int Cm, int ps, void* stream) {
    RPB_REQUIRE(gP && gu, "dpot_patch_tokens_bwd: null pointer");
    RPB_REQUIRE(B > 0 && T > 0 && ps > 0 && H % ps == 0 && W % ps == 0 && Cd > 0 && Cd <= Cm, "dpot_patch_tokens_bwd: bad sizes");
    const long total = (long)B * T * H * W * Cd;
    hipLaunchKernelGGL(patch_tokens_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, gP, gu, B, T, H, W, Cd, Cm,
                       ps, total);
    RPB_CHECK_LAUNCH("dpot_patch_tokens_bwd");
}

extern "C" int rpb_rowtable_add(float* x, const float* table, long M, int C, int rows_per_entry, int nent, void* stream) {
    RPB_REQUIRE(x && table && M > 0 && C > 0 && C % 4 == 0 && rows_per_entry > 0 && nent > 0, "rowtable_add: bad arguments");
    hipLaunchKernelGGL(rowtable_add_kernel, dim3(grid_for(M * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, x, table, M, C,
                       rows_per_entry, nent);
    RPB_CHECK_LAUNCH("rowtable_add");
}
extern "C" int rpb_rowtable_grad(const float* g, float* dtable, int B, int C, int rows_per_entry, int nent, void* stream) {
    RPB_REQUIRE(g && dtable && B > 0 && C > 0 && rows_per_entry > 0 && nent > 0, "rowtable_grad: bad arguments");
    const long n = (long)nent * C;
    hipLaunchKernelGGL(rowtable_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, dtable, B, C,
                       rows_per_entry, nent);
    RPB_CHECK_LAUNCH("rowtable_grad");
}

extern "C" int rpb_dpot_tagg_prep(const float* w, const float* gamma, const float* tt, float* Wf, float* Wb, float* e_out, int T, int C,
                                  void* stream) {
    RPB_REQUIRE(w && gamma && tt && Wf && Wb && e_out && T > 0 && C > 0 && C % 32 == 0, "dpot_tagg_prep: bad arguments (C %% 32)");
    hipLaunchKernelGGL(tagg_prep_kernel, dim3(C / 32, C / 32, T), dim3(256), 0, (hipStream_t)stream, w, gamma, tt, Wf, Wb, e_out, T, C);
    RPB_CHECK_LAUNCH("dpot_tagg_prep");
}
extern "C" int rpb_dpot_tagg_finish(const float* dWb, const float* dWsum, const float* w, const float* gamma, const float* tt, float* dw,
                                    float* dgamma, int T, int C, void* stream) {
    RPB_REQUIRE(dWb && w && gamma && tt && dw && dgamma && T > 0 && C > 0, "dpot_tagg_finish: bad arguments");
    hipLaunchKernelGGL(tagg_finish_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, dWb, dWsum, w, gamma, tt, dw, dgamma, T, C);
    RPB_CHECK_LAUNCH("dpot_tagg_finish");
}

// slices per channel: workgroup = cg * S threads rounded up to whole waves (the pad threads only take part in the reductions)
static int gn_slices(int cg) {
    int S = 1024 / cg;
    return S > 16 ? 16 : S;
}
extern "C" int rpb_gn_tokens_fwd(const float* x, const float* x2, const float* gamma, const float* beta, float* y, float* stat, int B,
                                 int P, int C, int G, float eps, void* stream) {
    RPB_REQUIRE(x && gamma && beta && y && stat && B > 0 && P > 0 && G > 0 && C % G == 0, "gn_tokens_fwd: bad arguments");
    const int S = gn_slices(C / G);
    RPB_REQUIRE(S > 0, "gn_tokens_fwd: %d channels per group unsupported (max 1024)", C / G);
    const int block = (C / G * S + 63) / 64 * 64;
    hipLaunchKernelGGL(gn_tokens_fwd_kernel, dim3(B * G), dim3(block), 0, (hipStream_t)stream, x, x2, gamma, beta, y, stat, P, C, G, S, eps);
    RPB_CHECK_LAUNCH("gn_tokens_fwd");
}
extern "C" int rpb_gn_tokens_bwd(const float* x, const float* x2, const float* gamma, const float* stat, const float* gy,
                                 const float* gadd, float* gx, float* pg, float* pb, int B, int P, int C, int G, void* stream) {
    RPB_REQUIRE(x && gamma && stat && gy && gx && pg && pb && B > 0 && P > 0 && G > 0 && C % G == 0, "gn_tokens_bwd: bad arguments");
    const int S = gn_slices(C / G);
    RPB_REQUIRE(S > 0, "gn_tokens_bwd: %d channels per group unsupported (max 1024)", C / G);
    const int block = (C / G * S + 63) / 64 * 64;
    hipLaunchKernelGGL(gn_tokens_bwd_kernel, dim3(B * G), dim3(block), (size_t)2 * block * 4, (hipStream_t)stream, x, x2, gamma, stat, gy,
                       gadd, gx, pg, pb, P, C, G, S);
    RPB_CHECK_LAUNCH("gn_tokens_bwd");
}

extern "C" int rpb_afno_wprep(const float* w, float* Wc, int nb, int bs_in, int bs_out, int transpose, void* stream) {
    RPB_REQUIRE(w && Wc && nb > 0 && bs_in > 0 && bs_out > 0, "afno_wprep: bad arguments");
    const long total = (long)nb * 4 * bs_in * bs_out;
    hipLaunchKernelGGL(afno_wprep_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, w, Wc, nb, bs_in, bs_out, transpose);
    RPB_CHECK_LAUNCH("afno_wprep");
}

extern "C" int rpb_afno_mlp(const float* X, const float* Wa, const float* ba, const float* Wb, const float* bb, const float* aux,
                            float* mid, float* out, long ntok, int nb, int bs, int mode, void* stream) {
    RPB_REQUIRE(X && Wa && Wb && out && ntok > 0 && nb > 0, "afno_mlp: bad arguments");
    RPB_REQUIRE(bs % 16 == 0 && 2 * bs <= 512, "afno_mlp: block size %d unsupported (multiple of 16, <= 256)", bs);      // 2 bs % 32 == 0: the K loops step by 16
    RPB_REQUIRE(mode == 0 || (mode == 1 && aux && mid), "afno_mlp: backward needs the saved pre-activation and a gradient buffer");
    AfnoArgs a{X, Wa, ba, Wb, bb, aux, mid, out, ntok, nb, bs, nb * bs, mode};
    const int waves = 2 * bs / 32;
    const size_t lds = (size_t)2 * 32 * (2 * bs + 1) * 4;
    hipLaunchKernelGGL(afno_mlp_kernel, dim3((unsigned)((ntok + 31) / 32), nb), dim3(waves * 64), lds, (hipStream_t)stream, a);
    RPB_CHECK_LAUNCH("afno_mlp");
}

extern "C" int rpb_afno_wgrad_splits(long ntok) {
    long s = (ntok + 255) / 256;
    return (int)(s < 1 ? 1 : (s > 16 ? 16 : s));
}
extern "C" int rpb_afno_wgrad(const float* A, const float* G, float* part, float* dw, long ntok, int nb, int bs, int a_gelu, void* stream) {
    RPB_REQUIRE(A && G && part && dw && ntok > 0 && nb > 0 && bs % 16 == 0, "afno_wgrad: bad arguments");
    const int splits = rpb_afno_wgrad_splits(ntok);
    const int nt = 2 * bs / 32;
    hipLaunchKernelGGL(afno_wgrad_kernel, dim3(nt * nt, nb, splits), dim3(64), 0, (hipStream_t)stream, A, G, part, ntok, nb, bs, nb * bs,
                       a_gelu, splits);
    const long total = (long)nb * bs * bs;
    hipLaunchKernelGGL(afno_wfinish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part, dw, nb, bs, splits);
    RPB_CHECK_LAUNCH("afno_wgrad");
}

extern "C" int rpb_dpot_unpatch(const float* O, float* pred, int B, int T, int H, int W, int Cd, int Co, int ps, int ldo, void* stream) {
    RPB_REQUIRE(O && pred && B > 0 && T > 0 && ps > 0 && H % ps == 0 && W % ps == 0 && Cd <= Co && T * Co <= ldo, "dpot_unpatch: bad arguments");
    const long total = (long)B * T * H * W * Cd;
    hipLaunchKernelGGL(unpatch_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, O, pred, B, T, H, W, Cd, Co, ps, ldo,
                       total, 0, (float*)nullptr);
    RPB_CHECK_LAUNCH("dpot_unpatch");
}
extern "C" int rpb_dpot_unpatch_bwd(const float* gpred, float* gO, int B, int T, int H, int W, int Cd, int Co, int ps, int ldo, void* stream) {
    RPB_REQUIRE(gpred && gO && B > 0 && T > 0 && ps > 0 && H % ps == 0 && W % ps == 0 && Cd <= Co && T * Co <= ldo, "dpot_unpatch_bwd: bad arguments");
    const long total = (long)B * H * W * ldo;
    hipLaunchKernelGGL(unpatch_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr,
                       const_cast<float*>(gpred), B, T, H, W, Cd, Co, ps, ldo, total, 1, gO);
    RPB_CHECK_LAUNCH("dpot_unpatch_bwd");
}
