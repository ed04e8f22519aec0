// HBM-streaming kernels of the U-Net path (reference realpdebench/model/unet.py), channels-last tokens [B][n][C]:
//
//   * GroupNorm(8) + time-conditioning scale/shift + SiLU of `Block` (unet.py:193-208) and their backward.
//     Everything that depends on the group structure is a function of per-(sample, channel) numbers, so the kernels are
//     group-agnostic "per-sample channel" passes and the B x C sized algebra between them is host glue:
//       fwd : chan_stats  -> (sum x, sum x^2)[b][c]   => mean / invstd per (b, group) => A[b][c], Bc[b][c]
//             affine_silu : y = silu(x * A[b][c] + Bc[b][c])
//       bwd : affine_silu_bwd_reduce -> (sum dz*x, sum dz)[b][c],  dz = gy * silu'(x*A + Bc)
//             => d gamma, d beta, d scale, d shift and, through mean/invstd(x), P[b][c], Q[b][c]
//             affine_silu_bwd_apply : gx = dz * A[b][c] + P[b][c] + Q[b][c] * x
//     Partials are one row per block, finished by rpb_reduce_partials in fp64 (deterministic, no atomics).
#include "rpb_common.h"

#define UN_THREADS 256

struct ChanArgs {
    const float* x;      // [B][n][C]
    const float* gy;     // bwd
    const float* A;      // [B][C]
    const float* Bc;     // [B][C]
    const float* P;      // [B][C]
    const float* Q;      // [B][C]
    const float* res;    // fwd: optional residual added after the SiLU (ResnetBlock identity shortcut)
    float* out;          // fwd: y ; bwd apply: gx
    float* part;         // [nblk][B][2][C]
    long n;
    int B, C;
};

__device__ __forceinline__ float silu_f(float z) { return z / (1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * z)); }
__device__ __forceinline__ float silu_grad_f(float z) {
    const float sig = 1.0f / (1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * z));
    return sig * (1.0f + z * (1.0f - sig));
}

// MODE 0: sums of x, x^2 | 1: y = silu(xA + Bc) | 2: sums of dz*x, dz | 3: gx = dz*A + P + Q*x
template <int MODE>
__global__ __launch_bounds__(UN_THREADS) void chan_kernel(ChanArgs a) {
    __shared__ f32x4 red[2][UN_THREADS];
    const int c4n = a.C >> 2;
    const int c4 = threadIdx.x % c4n, sub = threadIdx.x / c4n, nsub = UN_THREADS / c4n;
    const int b = blockIdx.y;
    const long per = (a.n + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * per;
    long r1 = r0 + per;
    if (r1 > a.n) r1 = a.n;
    const long base = ((long)b * a.n) * a.C + 4 * c4;
    f32x4 A4 = {0.f, 0.f, 0.f, 0.f}, B4 = A4, P4 = A4, Q4 = A4;
    if (MODE != 0) {
        A4 = *reinterpret_cast<const f32x4*>(a.A + (long)b * a.C + 4 * c4);
        B4 = *reinterpret_cast<const f32x4*>(a.Bc + (long)b * a.C + 4 * c4);
    }
    if (MODE == 3) {
        P4 = *reinterpret_cast<const f32x4*>(a.P + (long)b * a.C + 4 * c4);
        Q4 = *reinterpret_cast<const f32x4*>(a.Q + (long)b * a.C + 4 * c4);
    }
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll 4
    for (long r = r0 + sub; r < r1; r += nsub) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(a.x + base + r * a.C);
        if (MODE == 0) {
            s0 += x;
            s1 += x * x;
        } else if (MODE == 1) {
            const f32x4 z = x * A4 + B4;
            f32x4 y;
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = silu_f(z[k]);
            if (a.res) y += *reinterpret_cast<const f32x4*>(a.res + base + r * a.C);
            *reinterpret_cast<f32x4*>(a.out + base + r * a.C) = y;
        } else {
            const f32x4 g = *reinterpret_cast<const f32x4*>(a.gy + base + r * a.C);
            const f32x4 z = x * A4 + B4;
            f32x4 dz;
#pragma unroll
            for (int k = 0; k < 4; ++k) dz[k] = g[k] * silu_grad_f(z[k]);
            if (MODE == 2) {
                s0 += dz * x;
                s1 += dz;
            } else {
                *reinterpret_cast<f32x4*>(a.out + base + r * a.C) = dz * A4 + P4 + Q4 * x;
            }
        }
    }
    if (MODE == 0 || MODE == 2) {
        red[0][threadIdx.x] = s0;
        red[1][threadIdx.x] = s1;
        __syncthreads();
        if (sub == 0) {
            for (int k = 1; k < nsub; ++k) {
                s0 += red[0][k * c4n + c4];
                s1 += red[1][k * c4n + c4];
            }
            float* row = a.part + (((long)blockIdx.x * a.B + b) * 2) * a.C + 4 * c4;
            *reinterpret_cast<f32x4*>(row) = s0;
            *reinterpret_cast<f32x4*>(row + a.C) = s1;
        }
    }
}

extern "C" int rpb_chan_blocks(int B, long n) {
    long g = ((long)rpb_num_cus() * 8 + B - 1) / (B > 0 ? B : 1);
    const long cap = (n + 63) / 64;                  // at least 64 rows per block
    if (g > cap) g = cap;
    return (int)(g < 1 ? 1 : g);
}

static int chan_check(const ChanArgs& a) {
    RPB_REQUIRE(a.x && a.B > 0 && a.n > 0, "chan kernels: bad arguments");
    RPB_REQUIRE(a.C % 4 == 0 && a.C >= 4 && UN_THREADS % (a.C / 4) == 0, "chan kernels: C=%d must divide %d*4", a.C, UN_THREADS);
    return RPB_OK;
}

/* (sum x, sum x^2) per (sample, channel): part[rpb_chan_blocks(B, n)][B][2][C]. */
extern "C" int rpb_chan_stats(const float* x, float* part, int B, long n, int C, void* stream) {
    ChanArgs a{};
    a.x = x; a.part = part; a.B = B; a.n = n; a.C = C;
    if (int e = chan_check(a)) return e;
    RPB_REQUIRE(part, "chan_stats: null partial buffer");
    hipLaunchKernelGGL(chan_kernel<0>, dim3(rpb_chan_blocks(B, n), B), dim3(UN_THREADS), 0, (hipStream_t)stream, a);
    RPB_CHECK_LAUNCH("chan_stats");
}

extern "C" int rpb_affine_silu_fwd(const float* x, const float* A, const float* Bc, const float* res, float* y, int B,
                                   long n, int C, void* stream) {
    ChanArgs a{};
    a.x = x; a.A = A; a.Bc = Bc; a.res = res; a.out = y; a.B = B; a.n = n; a.C = C;
    if (int e = chan_check(a)) return e;
    RPB_REQUIRE(A && Bc && y, "affine_silu_fwd: null pointer");
    hipLaunchKernelGGL(chan_kernel<1>, dim3(rpb_chan_blocks(B, n), B), dim3(UN_THREADS), 0, (hipStream_t)stream, a);
    RPB_CHECK_LAUNCH("affine_silu_fwd");
}

/* (sum dz*x, sum dz) per (sample, channel), dz = gy * silu'(x*A + Bc): part[rpb_chan_blocks(B, n)][B][2][C]. */
extern "C" int rpb_affine_silu_bwd_reduce(const float* x, const float* gy, const float* A, const float* Bc, float* part,
                                          int B, long n, int C, void* stream) {
    ChanArgs a{};
    a.x = x; a.gy = gy; a.A = A; a.Bc = Bc; a.part = part; a.B = B; a.n = n; a.C = C;
    if (int e = chan_check(a)) return e;
    RPB_REQUIRE(gy && A && Bc && part, "affine_silu_bwd_reduce: null pointer");
    hipLaunchKernelGGL(chan_kernel<2>, dim3(rpb_chan_blocks(B, n), B), dim3(UN_THREADS), 0, (hipStream_t)stream, a);
    RPB_CHECK_LAUNCH("affine_silu_bwd_reduce");
}

extern "C" int rpb_affine_silu_bwd_apply(const float* x, const float* gy, const float* A, const float* Bc, const float* P,
                                         const float* Q, float* gx, int B, long n, int C, void* stream) {
    ChanArgs a{};
    a.x = x; a.gy = gy; a.A = A; a.Bc = Bc; a.P = P; a.Q = Q; a.out = gx; a.B = B; a.n = n; a.C = C;
    if (int e = chan_check(a)) return e;
    RPB_REQUIRE(gy && A && Bc && P && Q && gx, "affine_silu_bwd_apply: null pointer");
    hipLaunchKernelGGL(chan_kernel<3>, dim3(rpb_chan_blocks(B, n), B), dim3(UN_THREADS), 0, (hipStream_t)stream, a);
    RPB_CHECK_LAUNCH("affine_silu_bwd_apply");
}

// ---------------------------------------------------------------------------------- init_conv as im2col + token GEMM
// nn.Conv3d(C_in, dim, 7, padding 3) (unet.py:404) has C_in = 2..16 input channels: too few for the implicit-GEMM gather
// (32-channel chunks), so its im2col matrix col[m][tap*C_in + ci] (taps row-major over (dt, dh, dw), zero outside the
// mesh, zero-padded to ldc columns) is materialised once per step and both the forward product and the weight
// gradient are plain token GEMMs (rpb_gemm_nt / rpb_gemm_tn) on it.  The input needs no data gradient.
// A block walks a contiguous range of rows; a thread owns 4 fixed columns (one 16 B store per row) whose tap offsets are
// decoded once, and the row's mesh coordinates advance by +1 with carries -- no divisions in the row loop.
__global__ __launch_bounds__(320) void im2col_kernel(const float* __restrict__ x, float* __restrict__ col, long M, int T,
                                                     int H, int W, int Cin, int KS, int ldc, int q0) {
    const int R = KS / 2, ncol = KS * KS * KS * Cin;
    const int q = q0 + threadIdx.x;                                  // my float4 column (a launch covers 320 of them)
    const bool active = q < ldc / 4;
    int dt[4], dh[4], dw[4];
    long delta[4];
    bool kok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = 4 * q + e;
        kok[e] = active && k < ncol;
        const int tap = kok[e] ? k / Cin : 0, ci = kok[e] ? k - tap * Cin : 0;
        dw[e] = tap % KS - R;
        dh[e] = (tap / KS) % KS - R;
        dt[e] = tap / (KS * KS) - R;
        delta[e] = (((long)dt[e] * H + dh[e]) * W + dw[e]) * Cin + ci;
    }
    const long per = (M + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * per;
    long r1 = r0 + per;
    if (r1 > M) r1 = M;
    if (r0 >= r1) return;
    int w = (int)(r0 % W);
    long r = r0 / W;
    int h = (int)(r % H);
    int t = (int)((r / H) % T);
    for (long m = r0; m < r1; ++m) {
        if (active) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int tt = t + dt[e], hh = h + dh[e], ww = w + dw[e];
                if (kok[e] && tt >= 0 && tt < T && hh >= 0 && hh < H && ww >= 0 && ww < W) v[e] = x[m * Cin + delta[e]];
            }
            *reinterpret_cast<f32x4*>(col + m * ldc + 4 * q) = v;
        }
        if (++w >= W) {
            w = 0;
            if (++h >= H) {
                h = 0;
                if (++t >= T) t = 0;
            }
        }
    }
}

extern "C" int rpb_im2col(const float* x, float* col, int B, int T, int H, int W, int Cin, int KS, int ldc, void* stream) {
    RPB_REQUIRE(x && col && B > 0 && T > 0 && H > 0 && W > 0 && Cin > 0, "im2col: bad arguments");
    RPB_REQUIRE(KS % 2 == 1 && ldc >= KS * KS * KS * Cin && ldc % 4 == 0, "im2col: KS=%d must be odd and ldc=%d >= taps*Cin, a multiple of 4", KS, ldc);
    const long M = (long)B * T * H * W;
    long grid = (long)rpb_num_cus() * 16;
    if (grid > M) grid = M;
    for (int q0 = 0; q0 < ldc / 4; q0 += 320)                        // 320 float4 columns per launch (C_in = 3: one launch)
        hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)grid), dim3(320), 0, (hipStream_t)stream, x, col, M, T, H, W, Cin, KS,
                           ldc, q0);
    RPB_CHECK_LAUNCH("im2col");
}

// ---------------------------------------------------------------------------------- temporal attention
// Attention over the f = T <= 32 frames of one spatial location (unet.py:280-356 under EinopsToAndFrom 'b c f h w' ->
// 'b (h w) f c', :388): 4 heads x 32, q scaled by 32^-1/2, rotary position embedding on q and k, T5 relative-position
// bias added to the logits, softmax.  qkv is the token tensor [B][T][HW][3*128] produced by the to_qkv GEMM.
//
// One wave per (location, head) unit, every product on the fp32 MFMA with the T x 32 operands zero-padded to 32 x 32 in
// +1-padded LDS tiles (a first VALU version with one lane per query row ran at 5 TF/s: 55 k cycles per unit):
//   S^T = K Q^T is accumulated TRANSPOSED (row = key j, column = query i), so that a lane owns one query and the softmax
//   over keys is a reduction over its 16 registers plus one cross-half shuffle;
//   O = P V and d Q = dS K take P / dS straight from those registers as the MFMA A operand: at step s a lane supplies
//   key j = row(s, half) and the B operand reads the matching V / K row (any bijection of the contraction index works);
//   d K = dS^T Q and d V = P^T gO need the other orientation and go through LDS once.
// The backward recomputes the probabilities, accumulates the bias gradient in registers across the units a wave walks
// (waves keep a fixed head) and writes one partial row per wave.
#define TA_D 32
#define TA_XS 33
#define TA_TILE (32 * TA_XS)

struct TAttnArgs {
    const float* qkv;    // [B][T][HW][384]
    const float* rcos;   // [T][32] rotary tables (angle[t][d] = t * freq[d/2])
    const float* rsin;
    const float* bias;   // [4][T][T]
    float* out;          // fwd: [B][T][HW][128]
    const float* go;     // bwd: gradient of out
    float* gqkv;         // bwd: [B][T][HW][384]
    float* part;         // bwd: [gridDim.x * 4][T*T] bias-gradient partials; row % 4 = head
    long nloc;           // B * HW locations
    int T, HW;
};

// NI = frames per half-wave held in the prefetch registers (T <= 2 * NI): the q / k / v (/ gO) rows of the NEXT location a wave
// will process are requested right after the current one is staged, so their latency hides behind its MFMA and softmax work
// (two workgroups of four waves per CU cannot hide it by occupancy): 0.61 -> 0.47 ms forward, 2.35 -> 1.00 ms backward per call
// at the cylinder shape (the backward also went from one to two workgroups per CU).
template <bool BWD, int NI>
__global__ __launch_bounds__(256, 2) void tattn_kernel(TAttnArgs a) {
    extern __shared__ float lds[];
    const int T = a.T;
    const int lane = threadIdx.x & 63;
    const int head = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // 4 waves = 4 heads of one location
    const int col = lane & 31, half = lane >> 5;
    float* Ql = lds + head * (BWD ? 4 : 3) * TA_TILE;
    float* Kl = Ql + TA_TILE;
    float* Vl = Kl + TA_TILE;
    float* Gl = Vl + TA_TILE;                                     // BWD: go rows
    float* Pl = Vl;                                               // BWD: P^T  [j][i] -- takes V's tile once dP is done
    float* Sl = Kl;                                               // BWD: dS^T [j][i] -- takes K's tile once dQ is done
                                                                  // (4 tiles per wave = 68 KB + 8 KB of rotary tables per workgroup;
                                                                  //  __launch_bounds__(256, 2) keeps the backward at 256 registers so
                                                                  //  that two workgroups fit a CU: 268 before = one)
    for (int idx = lane; idx < (BWD ? 4 : 3) * TA_TILE; idx += 64) Ql[idx] = 0.f;     // rows t >= T stay zero
    float* rotc = lds + 4 * (BWD ? 4 : 3) * TA_TILE;              // rotary tables [32][32] (rows t >= T zero), shared by the waves
    float* rots = rotc + 32 * TA_D;
    for (int idx = threadIdx.x; idx < 32 * TA_D; idx += 256) {
        rotc[idx] = idx < T * TA_D ? a.rcos[idx] : 0.f;
        rots[idx] = idx < T * TA_D ? a.rsin[idx] : 0.f;
    }
    __syncthreads();
    const float scale = 0.17677669529663687f;                     // 32^-1/2
    const float sg = (col & 1) ? 1.f : -1.f;                      // rot(x)[2p] = -x[2p+1], rot(x)[2p+1] = x[2p]
    const bool qlive = col < T;                                   // my query column
    int jrow[16];
    float biasv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        jrow[r] = mfma_row(lane, r);
        biasv[r] = (qlive && jrow[r] < T) ? a.bias[(head * T + col) * T + jrow[r]] : 0.f;
    }
    f32x16 dbias = zero16();
    __builtin_amdgcn_wave_barrier();
    float rq[NI], rk[NI], rv[NI], rg[BWD ? NI : 1];
    auto fetch = [&](long loc) __attribute__((always_inline)) {
        const long b = loc / a.HW;
        const int hw = (int)(loc - b * a.HW);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int t = half + 2 * i;
            if (t < T) {
                const long tok = (b * T + t) * a.HW + hw;
                const float* src = a.qkv + tok * 384 + head * TA_D + col;
                rq[i] = src[0];
                rk[i] = src[128];
                rv[i] = src[256];
                if (BWD) rg[i] = a.go[tok * 128 + head * TA_D + col];
            }
        }
    };
    if ((long)blockIdx.x < a.nloc) fetch(blockIdx.x);
    for (long loc = blockIdx.x; loc < a.nloc; loc += gridDim.x) {
        const long b = loc / a.HW;
        const int hw = (int)(loc - b * a.HW);
        // ---- stage rows: lane = channel, the two half-waves take even / odd frames; rotary (+ the q scale) on the way in
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int t = half + 2 * i;
            if (t < T) {
                const float c = rotc[t * TA_D + col], sn = rots[t * TA_D + col];
                const float q0 = rq[i], k0 = rk[i];
                const float q1 = __shfl_xor(q0, 1, 64), k1 = __shfl_xor(k0, 1, 64);
                Ql[t * TA_XS + col] = (q0 * c + sg * q1 * sn) * scale;
                Kl[t * TA_XS + col] = k0 * c + sg * k1 * sn;
                Vl[t * TA_XS + col] = rv[i];
                if (BWD) Gl[t * TA_XS + col] = rg[i];
            }
        }
        if (loc + gridDim.x < a.nloc) fetch(loc + gridDim.x);     // in flight during everything below
        __builtin_amdgcn_wave_barrier();
        // ---- S^T[j][i] = sum_d K[j][d] Q[i][d]  (+ bias), softmax over j = my registers + the other half-wave
        f32x16 p = zero16();
#pragma unroll
        for (int s = 0; s < 16; ++s) p = mfma32(Kl[col * TA_XS + 2 * s + half], Ql[col * TA_XS + 2 * s + half], p);
        float mx = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = (jrow[r] < T) ? p[r] + biasv[r] : -3.0e38f;
            mx = fmaxf(mx, p[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float z = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = (jrow[r] < T) ? __expf(p[r] - mx) : 0.f;
            z += p[r];
        }
        z += __shfl_xor(z, 32, 64);
        const float iz = 1.0f / z;
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] *= iz;
        if (!BWD) {
            // ---- O[i][d] = sum_j P[i][j] V[j][d]: A = my P registers (key j = row(s, half)), B = the matching V row
            f32x16 o = zero16();
#pragma unroll
            for (int s = 0; s < 16; ++s) o = mfma32(p[s], Vl[jrow[s] * TA_XS + col], o);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (jrow[r] < T) a.out[((b * T + jrow[r]) * a.HW + hw) * 128 + head * TA_D + col] = o[r];
        } else {
            // ---- dP^T[j][i] = sum_d V[j][d] gO[i][d] ; dS = P (dP - sum_j P dP)
            f32x16 ds = zero16();
#pragma unroll
            for (int s = 0; s < 16; ++s) ds = mfma32(Vl[col * TA_XS + 2 * s + half], Gl[col * TA_XS + 2 * s + half], ds);
            float dot = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) dot += p[r] * ds[r];
            dot += __shfl_xor(dot, 32, 64);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                ds[r] = p[r] * (ds[r] - dot);
                if (qlive) dbias[r] += ds[r];
            }
            // ---- d q~[i][d] = sum_j dS[i][j] k~[j][d]  (A = my dS registers, B = the matching K row)
            f32x16 dq = zero16(), dk = zero16(), dv = zero16();
#pragma unroll
            for (int s = 0; s < 16; ++s) dq = mfma32(ds[s], Kl[jrow[s] * TA_XS + col], dq);
            __builtin_amdgcn_wave_barrier();                      // K and V are dead: their tiles take dS^T and P^T
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                Pl[jrow[r] * TA_XS + col] = p[r];
                Sl[jrow[r] * TA_XS + col] = ds[r];
            }
            __builtin_amdgcn_wave_barrier();
            // ---- d k~[j][d] = sum_i dS[i][j] q~[i][d] ; d v[j][d] = sum_i P[i][j] gO[i][d]  (A from the transposed tiles)
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                dk = mfma32(Sl[col * TA_XS + 2 * s + half], Ql[(2 * s + half) * TA_XS + col], dk);
                dv = mfma32(Pl[col * TA_XS + 2 * s + half], Gl[(2 * s + half) * TA_XS + col], dv);
            }
            // the rotation R_t is orthogonal: d q = scale * R_t^T d q~, d k = R_t^T d k~ ; R^T (y0, y1) = (y0 c + y1 s, -y0 s + y1 c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = jrow[r];
                const float nq = __shfl_xor(dq[r], 1, 64), nk = __shfl_xor(dk[r], 1, 64);
                if (t < T) {
                    const float c = rotc[t * TA_D + col], sn = rots[t * TA_D + col];
                    float* gq = a.gqkv + ((b * T + t) * a.HW + hw) * 384 + head * TA_D + col;
                    gq[0] = (dq[r] * c - sg * nq * sn) * scale;
                    gq[128] = dk[r] * c - sg * nk * sn;
                    gq[256] = dv[r];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (BWD && qlive) {
        float* row = a.part + ((long)blockIdx.x * 4 + head) * T * T + col * T;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (jrow[r] < T) row[jrow[r]] = dbias[r];
    }
}

extern "C" int rpb_tattn_blocks(long nloc) {
    long g = nloc;
    const long cap = (long)rpb_num_cus() * 6;
    if (g > cap) g = cap;
    return (int)(g < 1 ? 1 : g);
}

static int tattn_launch(bool bwd, TAttnArgs& a, hipStream_t st) {
    RPB_REQUIRE(a.qkv && a.rcos && a.rsin && a.bias && a.nloc > 0 && a.HW > 0, "tattn: bad arguments");
    RPB_REQUIRE(a.T >= 1 && a.T <= 32, "tattn: T=%d frames, the kernel holds up to 32", a.T);
    const size_t lds = ((size_t)(bwd ? 4 : 3) * TA_TILE * 4 + 2 * 32 * TA_D) * 4;
    const int grid = rpb_tattn_blocks(a.nloc);
#define TA_LAUNCH(B_, NI_)                                                                                                \
    do {                                                                                                                  \
        (void)hipFuncSetAttribute((const void*)tattn_kernel<B_, NI_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((tattn_kernel<B_, NI_>), dim3(grid), dim3(256), lds, st, a);                                   \
    } while (0)
    if (bwd) {
        if (a.T <= 10) TA_LAUNCH(true, 5);
        else if (a.T <= 20) TA_LAUNCH(true, 10);
        else TA_LAUNCH(true, 16);
    } else {
        if (a.T <= 10) TA_LAUNCH(false, 5);
        else if (a.T <= 20) TA_LAUNCH(false, 10);
        else TA_LAUNCH(false, 16);
    }
#undef TA_LAUNCH
    RPB_CHECK_LAUNCH("tattn");
}

extern "C" int rpb_tattn_fwd(const float* qkv, const float* rcos, const float* rsin, const float* bias, float* out, int B,
                             int T, int HW, void* stream) {
    RPB_REQUIRE(out, "tattn_fwd: null out");
    TAttnArgs a{};
    a.qkv = qkv; a.rcos = rcos; a.rsin = rsin; a.bias = bias; a.out = out; a.nloc = (long)B * HW; a.T = T; a.HW = HW;
    return tattn_launch(false, a, (hipStream_t)stream);
}

/* part[rpb_tattn_blocks(B*HW) * 4][T*T]: row r belongs to head r % 4 (d bias[head] = sum of its rows). */
extern "C" int rpb_tattn_bwd(const float* qkv, const float* rcos, const float* rsin, const float* bias, const float* go,
                             float* gqkv, float* part, int B, int T, int HW, void* stream) {
    RPB_REQUIRE(go && gqkv && part, "tattn_bwd: null pointer");
    TAttnArgs a{};
    a.qkv = qkv; a.rcos = rcos; a.rsin = rsin; a.bias = bias; a.go = go; a.gqkv = gqkv; a.part = part;
    a.nloc = (long)B * HW; a.T = T; a.HW = HW;
    return tattn_launch(true, a, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------- bottleneck spatial attention
// Softmax attention over the n = h*w tokens of one frame at the lowest resolution (unet.py:455-457: Attention under
// EinopsToAndFrom 'b c f h w' -> 'b f (h w) c'; 4 heads x 32, q scaled, no rotary, no bias), flash-attention style on the
// fp32 MFMA for any n (nothing n x n is ever stored; the forward keeps the row log-sum-exp, the backward recomputes the
// probabilities).  A wave owns 32 rows of one side, the other side streams through LDS in 64-row chunks:
//   MODE 0 forward       own = queries, stream K / V:  S^T[key][query] = K Q^T, online softmax, O^T[d][query] += V^T P
//   MODE 1 backward d q  own = queries, stream K / V:  P = exp(S^T - lse), dS = P (dP^T - D), dQ^T[d][query] += K^T dS
//   MODE 2 backward d k, d v  own = keys, stream Q / gO:  S[query][key], dV^T[d][key] += gO^T P, dK^T[d][key] += Q^T dS
// The transposed accumulators put "own" in the MFMA column = lane, so the softmax statistics (running max, sum, lse, D)
// are per-lane scalars, and a D-layout register tile feeds the next MFMA as its B operand when the A operand reads row
// mfma_row(lane, r) of the LDS tile (the contraction order of an MFMA chain is free).  Contractions over the head
// dimension use d = 16 * half + s, so a lane's own-side operand is 64 contiguous bytes of its row.
#define SA_D 32
#define SA_XS 33
#define SA_CH 64
#define SA_NW 4

struct SAttnArgs {
    const float* qkv;    // [F][n][384]
    float* out;          // fwd: [F][n][128]
    float* lse;          // [F][4][n] row log-sum-exp (natural log) of the scaled logits
    const float* o;      // bwd: forward output
    const float* go;     // bwd
    float* gqkv;         // bwd: [F][n][384]
    int n;
};

template <int MODE>
__global__ __launch_bounds__(256) void sattn_kernel(SAttnArgs a) {
    __shared__ float Al[SA_CH * SA_XS];        // MODE 0/1: K chunk      MODE 2: scaled Q chunk
    __shared__ float Bl[SA_CH * SA_XS];        // MODE 0/1: V chunk      MODE 2: gO chunk
    __shared__ float Ll[2][SA_CH];             // MODE 2: lse and D = gO . O of the chunk's queries
    __shared__ float El[SA_NW][32 * SA_XS];    // wave-private epilogue transposition
    const int n = a.n, tid = threadIdx.x, lane = tid & 63, col = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = blockIdx.x >> 2, head = blockIdx.x & 3;
    const int r0 = (blockIdx.y * SA_NW + wave) * 32;
    const float scale = 0.17677669529663687f;
    const float* qb = a.qkv + (long)f * n * 384 + head * SA_D;
    const float* gob = MODE ? a.go + (long)f * n * 128 + head * SA_D : nullptr;
    const float* ob = MODE ? a.o + (long)f * n * 128 + head * SA_D : nullptr;
    const long lrow = ((long)f * 4 + head) * n;
    const int own = r0 + col;
    const bool live = own < n;
    int jrow[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) jrow[r] = mfma_row(lane, r);

    // ---- own-side operands: lane (col, half) holds head dims 16 * half + s of row `own`
    float u[16], v[16];
    float Di = 0.f, lse_own = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f32x4 x = {0.f, 0.f, 0.f, 0.f}, y = x, z = x;
        if (live) {
            x = *reinterpret_cast<const f32x4*>(qb + (long)own * 384 + (MODE == 2 ? 128 : 0) + 16 * half + 4 * k);
            if (MODE == 2) y = *reinterpret_cast<const f32x4*>(qb + (long)own * 384 + 256 + 16 * half + 4 * k);
            if (MODE == 1) {
                y = *reinterpret_cast<const f32x4*>(gob + (long)own * 128 + 16 * half + 4 * k);
                z = *reinterpret_cast<const f32x4*>(ob + (long)own * 128 + 16 * half + 4 * k);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u[4 * k + i] = MODE == 2 ? x[i] : x[i] * scale;
            v[4 * k + i] = y[i];
            Di += y[i] * z[i];
        }
    }
    if (MODE == 1) {
        Di += __shfl_xor(Di, 32, 64);
        lse_own = live ? a.lse[lrow + own] : 0.f;
    }

    float m = -3.0e38f, l = 0.f;
    f32x16 acc0 = zero16(), acc1 = zero16();       // MODE 0: O^T;  MODE 1: dQ^T;  MODE 2: dK^T, dV^T
    f32x4 pa[2], pb[2], po[2];
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    auto prefetch = [&](int c0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + j * 256, row = idx >> 3, c4 = (idx & 7) * 4;
            const long r = c0 + row;
            pa[j] = pb[j] = po[j] = z4;
            if (r < n) {
                if (MODE == 2) {
                    pa[j] = *reinterpret_cast<const f32x4*>(qb + r * 384 + c4) * scale;
                    pb[j] = *reinterpret_cast<const f32x4*>(gob + r * 128 + c4);
                    po[j] = *reinterpret_cast<const f32x4*>(ob + r * 128 + c4);
                } else {
                    pa[j] = *reinterpret_cast<const f32x4*>(qb + r * 384 + 128 + c4);
                    pb[j] = *reinterpret_cast<const f32x4*>(qb + r * 384 + 256 + c4);
                }
            }
        }
    };
    prefetch(0);
    for (int c0 = 0; c0 < n; c0 += SA_CH) {
        __syncthreads();                                            // the previous chunk's LDS reads are done
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + j * 256, row = idx >> 3, c4 = (idx & 7) * 4;
            float* da = Al + row * SA_XS + c4;
            float* db = Bl + row * SA_XS + c4;
            da[0] = pa[j][0]; da[1] = pa[j][1]; da[2] = pa[j][2]; da[3] = pa[j][3];
            db[0] = pb[j][0]; db[1] = pb[j][1]; db[2] = pb[j][2]; db[3] = pb[j][3];
            if (MODE == 2) {
                float dd = pb[j][0] * po[j][0] + pb[j][1] * po[j][1] + pb[j][2] * po[j][2] + pb[j][3] * po[j][3];
                dd += __shfl_xor(dd, 1, 64);
                dd += __shfl_xor(dd, 2, 64);
                dd += __shfl_xor(dd, 4, 64);
                if ((tid & 7) == 0) {
                    Ll[1][row] = dd;
                    Ll[0][row] = (c0 + row < n) ? a.lse[lrow + c0 + row] : 3.0e38f;
                }
            }
        }
        __syncthreads();
        if (c0 + SA_CH < n) prefetch(c0 + SA_CH);                   // in flight during the MFMAs below
#pragma unroll
        for (int t = 0; t < SA_CH / 32; ++t) {
            if (c0 + 32 * t >= n) break;
            const float* At = Al + t * 32 * SA_XS;
            const float* Bt = Bl + t * 32 * SA_XS;
            f32x16 p = zero16();
#pragma unroll
            for (int s = 0; s < 16; ++s) p = mfma32(At[col * SA_XS + 16 * half + s], u[s], p);
            if (MODE == 0) {
                float tmax = -3.0e38f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (c0 + 32 * t + jrow[r] >= n) p[r] = -3.0e38f;
                    tmax = fmaxf(tmax, p[r]);
                }
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float mn = fmaxf(m, tmax);
                const float corr = __expf(m - mn);
                m = mn;
                float psum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    p[r] = __expf(p[r] - mn);
                    psum += p[r];
                    acc0[r] *= corr;
                }
                l = l * corr + psum;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc0 = mfma32(Bt[jrow[r] * SA_XS + col], p[r], acc0);
            } else {
                f32x16 dp = zero16();
#pragma unroll
                for (int s = 0; s < 16; ++s) dp = mfma32(Bt[col * SA_XS + 16 * half + s], v[s], dp);
                if (MODE == 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float pr = (c0 + 32 * t + jrow[r] < n) ? __expf(p[r] - lse_own) : 0.f;
                        p[r] = pr * (dp[r] - Di);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc0 = mfma32(At[jrow[r] * SA_XS + col], p[r], acc0);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int qi = 32 * t + jrow[r];
                        const float pr = __expf(p[r] - Ll[0][qi]);
                        dp[r] = pr * (dp[r] - Ll[1][qi]);
                        p[r] = pr;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        acc0 = mfma32(At[jrow[r] * SA_XS + col], dp[r], acc0);      // dK^T += Q_s^T dS
                        acc1 = mfma32(Bt[jrow[r] * SA_XS + col], p[r], acc1);       // dV^T += gO^T P
                    }
                }
            }
        }
    }
    // ---- epilogue: own-major rows through the wave's LDS tile, 16 B stores
    float* Et = El[wave];
    float mul = MODE == 1 ? scale : 1.f;
    if (MODE == 0) {
        const float lt = l + __shfl_xor(l, 32, 64);
        mul = 1.0f / lt;
        if (live && half == 0) a.lse[lrow + own] = m + __logf(lt);
    }
    float* dst = MODE == 0 ? a.out + (long)f * n * 128 + head * SA_D : a.gqkv + (long)f * n * 384 + head * SA_D + (MODE == 2 ? 128 : 0);
    const int ldo = MODE == 0 ? 128 : 384;
#pragma unroll
    for (int pass = 0; pass < (MODE == 2 ? 2 : 1); ++pass) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Et[col * SA_XS + jrow[r]] = (pass ? acc1[r] : acc0[r]) * mul;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = j * 64 + lane, row = idx >> 3, c4 = (idx & 7) * 4;
            if (r0 + row < n) {
                const float* sp = Et + row * SA_XS + c4;
                const f32x4 x = {sp[0], sp[1], sp[2], sp[3]};
                *reinterpret_cast<f32x4*>(dst + (long)(r0 + row) * ldo + pass * 128 + c4) = x;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int MODE>
static void sattn_launch(const SAttnArgs& a, int F, hipStream_t st) {
    hipLaunchKernelGGL(sattn_kernel<MODE>, dim3(F * 4, (a.n + 32 * SA_NW - 1) / (32 * SA_NW)), dim3(64 * SA_NW), 0, st, a);
}

extern "C" int rpb_sattn_fwd(const float* qkv, float* out, float* lse, int F, int n, void* stream) {
    RPB_REQUIRE(qkv && out && lse && F > 0 && n > 0, "sattn_fwd: bad arguments");
    SAttnArgs a{};
    a.qkv = qkv; a.out = out; a.lse = lse; a.n = n;
    sattn_launch<0>(a, F, (hipStream_t)stream);
    RPB_CHECK_LAUNCH("sattn_fwd");
}

extern "C" int rpb_sattn_bwd(const float* qkv, const float* o, const float* go, float* lse, float* gqkv, int F, int n,
                             void* stream) {
    RPB_REQUIRE(qkv && o && go && lse && gqkv && F > 0 && n > 0, "sattn_bwd: bad arguments");
    SAttnArgs a{};
    a.qkv = qkv; a.o = o; a.go = go; a.lse = lse; a.gqkv = gqkv; a.n = n;
    sattn_launch<1>(a, F, (hipStream_t)stream);
    sattn_launch<2>(a, F, (hipStream_t)stream);
    RPB_CHECK_LAUNCH("sattn_bwd");
}

// ---------------------------------------------------------------------------------- spatial linear attention
// SpatialLinearAttention (unet.py:236-261) per frame f = (b, t) with n = h*w tokens, 4 heads x 32:
//   q' = softmax_d(q) * 32^-1/2 ;  ksoft = softmax_n(k) ;  context[d][e] = sum_n ksoft[n][d] v[n][e] ;
//   out[n][e] = sum_d context[d][e] q'[n][d]
// ksoft = E / Z with E = exp(k - kmax[f][c]) and Z[f][c] = sum_n E, so the two token-sized tensors the dense kernels
// need are q' and E: linattn_prep writes QE[m] = [q' (128) | E (128)] in one pass over the to_qkv output; the 32 x 32
// per-head products run on rpb_head_scores / rpb_head_apply (the 128 channels taken as 2 x 64, the cross-head 32 x 32
// blocks are discarded / zero-filled by the host), Z comes from rpb_chan_stats, kmax from rpb_chan_max.
// Backward: with d q' and d E' (= head_apply(v, dS^T)) in dQE, d Z[f][c] from the host:
//   d q = q' * (d q' - <q', d q'>_head / scale) ;  d k = (d E' + d Z) * E   (kmax is a constant shift of a softmax).
struct LinPrepArgs {
    const float* qkv;    // [F*n][384]
    const float* kmax;   // [F][128]
    float* qe;           // fwd out / bwd in: [F*n][256]
    const float* dqe;    // bwd: [F*n][256]
    const float* dz;     // bwd: [F][128]
    float* gqkv;         // bwd: columns 0..255 of [F*n][384]
    long M;
    int n;
};

__device__ __forceinline__ float seg8_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}
__device__ __forceinline__ float seg8_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1, 64));
    v = fmaxf(v, __shfl_xor(v, 2, 64));
    v = fmaxf(v, __shfl_xor(v, 4, 64));
    return v;
}

// one wave per token: lanes 0..31 hold the 128 q channels (8 lanes = one 32-wide head), lanes 32..63 the 128 k channels
template <bool BWD>
__global__ __launch_bounds__(256) void linprep_kernel(LinPrepArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const bool isq = lane < 32;
    const int c = (lane & 31) * 4;
    const float scale = 0.17677669529663687f;
    for (long m = (long)blockIdx.x * 4 + wave; m < a.M; m += (long)gridDim.x * 4) {
        const long f = m / a.n;
        if (!BWD) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(a.qkv + m * 384 + (isq ? 0 : 128) + c);
            f32x4 o;
            if (isq) {
                const float mx = seg8_max(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = __expf(v[k] - mx);
                const float inv = scale / seg8_sum(o[0] + o[1] + o[2] + o[3]);
                o = o * inv;
            } else {
                const f32x4 km = *reinterpret_cast<const f32x4*>(a.kmax + f * 128 + c);
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = __expf(v[k] - km[k]);
            }
            *reinterpret_cast<f32x4*>(a.qe + m * 256 + (isq ? 0 : 128) + c) = o;
        } else {
            const f32x4 y = *reinterpret_cast<const f32x4*>(a.qe + m * 256 + (isq ? 0 : 128) + c);
            const f32x4 g = *reinterpret_cast<const f32x4*>(a.dqe + m * 256 + (isq ? 0 : 128) + c);
            f32x4 o;
            if (isq) {
                const float t = seg8_sum(y[0] * g[0] + y[1] * g[1] + y[2] * g[2] + y[3] * g[3]) * (1.0f / scale);
                o = y * (g - t);
            } else {
                const f32x4 dz = *reinterpret_cast<const f32x4*>(a.dz + f * 128 + c);
                o = (g + dz) * y;
            }
            *reinterpret_cast<f32x4*>(a.gqkv + m * 384 + (isq ? 0 : 128) + c) = o;
        }
    }
}

static int linprep_grid(long M) {
    long g = (M + 3) / 4;
    const long cap = (long)rpb_num_cus() * 8;
    if (g > cap) g = cap;
    return (int)(g < 1 ? 1 : g);
}

extern "C" int rpb_linattn_prep_fwd(const float* qkv, const float* kmax, float* qe, int F, int n, void* stream) {
    RPB_REQUIRE(qkv && kmax && qe && F > 0 && n > 0, "linattn_prep_fwd: bad arguments");
    LinPrepArgs a{};
    a.qkv = qkv; a.kmax = kmax; a.qe = qe; a.M = (long)F * n; a.n = n;
    hipLaunchKernelGGL(linprep_kernel<false>, dim3(linprep_grid(a.M)), dim3(256), 0, (hipStream_t)stream, a);
    RPB_CHECK_LAUNCH("linattn_prep_fwd");
}

extern "C" int rpb_linattn_prep_bwd(const float* qe, const float* dqe, const float* dz, float* gqkv, int F, int n,
                                    void* stream) {
    RPB_REQUIRE(qe && dqe && dz && gqkv && F > 0 && n > 0, "linattn_prep_bwd: bad arguments");
    LinPrepArgs a{};
    a.qe = const_cast<float*>(qe); a.dqe = dqe; a.dz = dz; a.gqkv = gqkv; a.M = (long)F * n; a.n = n;
    hipLaunchKernelGGL(linprep_kernel<true>, dim3(linprep_grid(a.M)), dim3(256), 0, (hipStream_t)stream, a);
    RPB_CHECK_LAUNCH("linattn_prep_bwd");
}

// column maxima / sums of a strided token tensor per frame: part[rpb_chan_blocks(F, n)][F][C]
// (the host finishes with a max / the fp64 partial reducer).  MODE 0 = max, 1 = sum.
template <int MODE>
__global__ __launch_bounds__(UN_THREADS) void colred_kernel(const float* __restrict__ x, float* __restrict__ part, long n,
                                                            int F, int C, int ldx) {
    __shared__ f32x4 red[UN_THREADS];
    const int c4n = C >> 2;
    const int c4 = threadIdx.x % c4n, sub = threadIdx.x / c4n, nsub = UN_THREADS / c4n;
    const int f = blockIdx.y;
    const long per = (n + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * per;
    long r1 = r0 + per;
    if (r1 > n) r1 = n;
    const float init = MODE == 0 ? -3.0e38f : 0.f;
    f32x4 acc = {init, init, init, init};
    for (long r = r0 + sub; r < r1; r += nsub) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((long)f * n + r) * ldx + 4 * c4);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = MODE == 0 ? fmaxf(acc[k], v[k]) : acc[k] + v[k];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (sub == 0) {
        for (int s = 1; s < nsub; ++s) {
            const f32x4 o = red[s * c4n + c4];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = MODE == 0 ? fmaxf(acc[k], o[k]) : acc[k] + o[k];
        }
        *reinterpret_cast<f32x4*>(part + ((long)blockIdx.x * F + f) * C + 4 * c4) = acc;
    }
}

extern "C" int rpb_col_reduce(const float* x, int ldx, float* part, int F, long n, int C, int mode, void* stream) {
    RPB_REQUIRE(x && part && F > 0 && n > 0, "col_reduce: bad arguments");
    RPB_REQUIRE(C % 4 == 0 && UN_THREADS % (C / 4) == 0 && ldx >= C && ldx % 4 == 0, "col_reduce: C=%d ldx=%d unsupported", C, ldx);
    RPB_REQUIRE(mode == 0 || mode == 1, "col_reduce: mode 0 (max) or 1 (sum)");
    const dim3 grid(rpb_chan_blocks(F, n), F);
    if (mode == 0) hipLaunchKernelGGL(colred_kernel<0>, grid, dim3(UN_THREADS), 0, (hipStream_t)stream, x, part, n, F, C, ldx);
    else hipLaunchKernelGGL(colred_kernel<1>, grid, dim3(UN_THREADS), 0, (hipStream_t)stream, x, part, n, F, C, ldx);
    RPB_CHECK_LAUNCH("col_reduce");
}
