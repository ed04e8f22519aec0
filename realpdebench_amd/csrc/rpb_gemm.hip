// Token GEMM on fp32 MFMA:  out[m][n] = epilogue( sum_k A(m,k) * W[n][k] )      (W in nn.Linear layout [N][K])
//
// Used by the Transolver path (reference realpdebench/model/TRANSOLVER_libs):
//   * nn.Linear over tokens -- preprocess.linear_post, Attn.to_out, mlp.linear_pre/post, mlp2
//     (Transolver_Structured_Mesh_3D.py:27-39,71-77), with bias / GELU / residual / broadcast-vector epilogues fused;
//   * nn.Conv3d(256, 256, 3, padding=1) x2 of Physics_Attention_Structured_Mesh_3D (Physics_Attention.py:138-139,
//     154-157) as ONE implicit GEMM: A(m, tap*Ci + ci) = x[neighbour(m, tap)][ci] (zero outside the mesh) gathered on
//     the fly from the channels-last token tensor, N = 512 = both convolutions, K = 27*Ci.  The reference's
//     permute(0,4,1,2,3) / permute(0,2,3,4,1) pairs around the convolutions disappear.
//
// Tiling: workgroup = WM x WN waves, each wave TM x TN MFMA tiles of 32x32 (v_mfma_f32_32x32x2_f32), K chunks of 32.
// Both operands are staged through +1-padded LDS tiles ([row][BK+1], conflict-free transposed ds_read_b32); the next
// chunk is prefetched into registers (float4, coalesced along k) while the current one is multiplied.
#include "rpb_common.h"

#define G_BK 32

struct GemmArgs {
    const float* A;        // [rows][lda]
    const float* W;        // [N][K]
    const float* bias;     // [N] or null
    const float* addvec;   // [N] or null (e.g. the Transolver `placeholder`)
    const float* residual; // [M][ldo] or null
    float* out;            // [M][ldo]
    long M;
    int N, K, lda, ldo;
    int act;               // 0 none, 1 exact GELU, 2 multiply by gelu'(aux[m][n]) (backward through a GELU),
                           // 3 ReLU (Galerkin FeedForward, layers.py:980), 4 zero where aux[m][n] <= 0 (backward through a ReLU)
    const float* aux;      // act == 2: the saved pre-activation [M][ldo]; act == 4: the saved ReLU output
    float* pre_out;        // act == 1: optional copy of the pre-activation (saved for the backward pass)
    const float* mask;     // optional [M][ldo] multiplier applied before addvec/residual (inverted-dropout mask)
    DropSpec drop;         // or the same dropout regenerated in-kernel from (seed, element index m*ldo + n)
    // conv = 1: implicit 3x3x3 convolution (padding 1) over a (Hc, Wc, Dc) mesh, tokens row-major in (h, w, d); Ci = K / 27
    // conv = 2: nn.Conv3d(C, C, (1,4,4), stride (1,2,2), padding (0,1,1)) -- U-Net Downsample, unet.py:166-167: the INPUT
    //           lives on the mesh (Hc, Wc, Dc) = (T, H, W), rows m enumerate the output mesh (T, H/2, W/2); K = 16*Ci,
    //           k = (kh*4 + kw)*Ci + ci
    // conv = 3: one output-parity class (ph, pw) = (cls >> 1, cls & 1) of nn.ConvTranspose3d(C, C, (1,4,4), (1,2,2),
    //           (0,1,1)) -- U-Net Upsample, unet.py:163-164: rows m enumerate the INPUT mesh (T, H, W); output row
    //           (t, 2h+ph, 2w+pw) of the (T, 2H, 2W) mesh; K = 4*Ci, k = (jh*2 + jw)*Ci + ci with input offsets
    //           dh = ph ? (jh ? 0 : +1) : (jh ? -1 : 0)  (kernel rows kh = ph ? (0, 2) : (1, 3)), same for w
    int conv, Hc, Wc, Dc, cls;
};

// CM = implicit-GEMM gather mode (GemmArgs::conv) as a compile-time constant: the plain and 3x3x3 instantiations carry none of
// the strided / transposed index arithmetic
template <int WM, int WN, int TM, int TN, int CM>
__global__ __launch_bounds__(WM* WN * 64) void gemm_nt_kernel(GemmArgs g) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NTHR = WM * WN * 64;
    constexpr int XS = G_BK + 1;
    constexpr int A4 = BM * (G_BK / 4) / NTHR;      // float4 loads per thread per chunk (A)
    constexpr int W4 = BN * (G_BK / 4) / NTHR;      //                                   (W)
    static_assert(A4 >= 1 && W4 >= 1, "tile too small for the thread count");
    extern __shared__ float lds[];
    float* Al = lds;                  // [BM][XS]
    float* Wl = lds + BM * XS;        // [BN][XS]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int col = lane & 31, half = lane >> 5;
    const int ntn = (g.N + BN - 1) / BN;
    const long tile_m = blockIdx.x / ntn;
    const int tile_n = blockIdx.x % ntn;
    const long m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int Ci = CM == 1 ? g.K / 27 : (CM == 2 ? g.K / 16 : (CM == 3 ? g.K / 4 : g.K));

    // per-thread rows of the A tile (fixed across chunks) and, for the convolution, their mesh coordinates
    int arow[A4];          // row within tile
    long am[A4];           // global token index (or -1)
    int ah[A4], aw[A4], ad[A4];
#pragma unroll
    for (int j = 0; j < A4; ++j) {
        const int idx = tid + j * NTHR;
        arow[j] = idx / (G_BK / 4);
        const long m = m0 + arow[j];
        am[j] = (m < g.M) ? m : -1;      // validity is tracked in arow_ok (a conv = 2 base token may be negative)
        // row indices fit 32 bits (M < 2^31 is checked by the launcher): unsigned 32-bit divisions, not the 64-bit library ones
        if (CM == 2 && am[j] >= 0) {                 // m over the output mesh (Hc, Wc/2, Dc/2)
            const unsigned Wo = g.Wc / 2, Do = g.Dc / 2;
            const unsigned per = (unsigned)g.Hc * Wo * Do;
            const unsigned mu = (unsigned)m;
            const unsigned bb = mu / per;
            unsigned r = mu - bb * per;
            const int dq = (int)(r % Do);
            r /= Do;
            const int wq = (int)(r % Wo);
            const int hq = (int)(r / Wo);
            ah[j] = hq;
            aw[j] = 2 * wq - 1;
            ad[j] = 2 * dq - 1;
            am[j] = (((long)bb * g.Hc + hq) * g.Wc + aw[j]) * g.Dc + ad[j];     // token of tap (0,0); only dereferenced for in-bounds taps
        } else if (CM && am[j] >= 0) {
            const unsigned per = (unsigned)g.Hc * g.Wc * g.Dc;
            unsigned r = (unsigned)m % per;
            ad[j] = (int)(r % (unsigned)g.Dc);
            r /= (unsigned)g.Dc;
            aw[j] = (int)(r % (unsigned)g.Wc);
            ah[j] = (int)(r / (unsigned)g.Wc);
        } else {
            ah[j] = aw[j] = ad[j] = 0;
        }
    }
    bool arow_ok[A4];
#pragma unroll
    for (int j = 0; j < A4; ++j) arow_ok[j] = (m0 + arow[j]) < g.M;
    const int c4 = tid % (G_BK / 4);           // float4 column within the chunk (same for every j: NTHR % 8 == 0)

    f32x4 pa[A4], pw[W4];
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const int nchunk = g.K / G_BK;

    auto prefetch = [&](int ch) {
        const int k0 = ch * G_BK;
        int tap = 0, kk = k0;
        int dh = 0, dw = 0, dd = 0;
        long noff = 0;
        if (CM) {
            tap = k0 / Ci;
            kk = k0 - tap * Ci;
            if (CM == 1) {
                dh = tap / 9 - 1;
                dw = (tap / 3) % 3 - 1;
                dd = tap % 3 - 1;
            } else if (CM == 2) {                 // taps (kw', kd') of the (1,4,4) kernel, relative to tap (0,0)
                dw = tap >> 2;
                dd = tap & 3;
            } else {                                  // transposed class: 2 x 2 taps
                const int ph = g.cls >> 1, pw = g.cls & 1, jh = tap >> 1, jw = tap & 1;
                dw = ph ? (jh ? 0 : 1) : (jh ? -1 : 0);
                dd = pw ? (jw ? 0 : 1) : (jw ? -1 : 0);
            }
            noff = ((long)dh * g.Wc + dw) * g.Dc + dd;
        }
#pragma unroll
        for (int j = 0; j < A4; ++j) {
            f32x4 v = z4;
            if (arow_ok[j]) {
                if (CM) {
                    const int hh = ah[j] + dh, ww = aw[j] + dw, d2 = ad[j] + dd;
                    if (hh >= 0 && hh < g.Hc && ww >= 0 && ww < g.Wc && d2 >= 0 && d2 < g.Dc)
                        v = *reinterpret_cast<const f32x4*>(g.A + (am[j] + noff) * g.lda + kk + 4 * c4);
                } else {
                    v = *reinterpret_cast<const f32x4*>(g.A + am[j] * g.lda + k0 + 4 * c4);
                }
            }
            pa[j] = v;
        }
#pragma unroll
        for (int j = 0; j < W4; ++j) {
            const int idx = tid + j * NTHR;
            const int n = n0 + idx / (G_BK / 4);
            pw[j] = (n < g.N) ? *reinterpret_cast<const f32x4*>(g.W + (long)n * g.K + k0 + 4 * c4) : z4;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = zero16();

    prefetch(0);
    for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();                               // previous chunk's LDS reads are done
#pragma unroll
        for (int j = 0; j < A4; ++j) {
            float* d = Al + arow[j] * XS + 4 * c4;
            d[0] = pa[j][0];
            d[1] = pa[j][1];
            d[2] = pa[j][2];
            d[3] = pa[j][3];
        }
#pragma unroll
        for (int j = 0; j < W4; ++j) {
            const int idx = tid + j * NTHR;
            float* d = Wl + (idx / (G_BK / 4)) * XS + 4 * c4;
            d[0] = pw[j][0];
            d[1] = pw[j][1];
            d[2] = pw[j][2];
            d[3] = pw[j][3];
        }
        __syncthreads();
        if (ch + 1 < nchunk) prefetch(ch + 1);          // in flight during the MFMAs below
        const float* ap = Al + (wm * TM * 32 + col) * XS + half;
        const float* bp = Wl + (wn * TN * 32 + col) * XS + half;
#pragma unroll
        for (int s = 0; s < G_BK / 2; ++s) {
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = ap[i * 32 * XS + 2 * s];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = bp[j * 32 * XS + 2 * s];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(av[i], bv[j], acc[i][j]);
        }
    }
    // epilogue: bias, broadcast vector, activation, mask, residual.
    // Fast path (N, ldo multiples of 4): every 32 x 32 accumulator tile goes through a wave-private LDS tile (the operand
    // tiles are dead by now) so that a lane owns 4 CONSECUTIVE columns of a row: bias / aux / mask / residual are read
    // and the result is written with 16 B per lane (a wave instruction covers 8 rows x 128 B) instead of 16 scalar,
    // latency-serialised accesses per tile and operand.
    const bool vec4 = (g.N % 4 == 0) && (g.ldo % 4 == 0);
    if (vec4) {
        __syncthreads();                                     // all waves are done with Al / Wl
        float* tb = lds + wave * (32 * 36);                   // wave-private [32][36]
        const int er = lane >> 3, ec = (lane & 7) * 4;        // row within a group of 8, first of my 4 columns
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nb = n0 + (wn * TN + j) * 32;
            if (nb >= g.N) continue;
            const int n = nb + ec;
            const bool nok = n < g.N;
            f32x4 badd = {0.f, 0.f, 0.f, 0.f}, vadd = badd;
            if (nok && g.bias) badd = *reinterpret_cast<const f32x4*>(g.bias + n);
            if (nok && g.addvec) vadd = *reinterpret_cast<const f32x4*>(g.addvec + n);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) tb[mfma_row(lane, r) * 36 + col] = acc[i][j][r];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = q * 8 + er;
                    const long mi = m0 + (wm * TM + i) * 32 + row;
                    if (mi < g.M && nok) {
                        long m = mi;
                        if (CM == 3) {                   // (b, t, h, w) -> row (b, t, 2h+ph, 2w+pw) of the up-sampled mesh
                            const unsigned per = (unsigned)g.Wc * g.Dc;
                            const unsigned bt = (unsigned)mi / per;
                            const unsigned r2 = (unsigned)mi - bt * per;
                            const unsigned hh = r2 / (unsigned)g.Dc, ww = r2 - hh * g.Dc;
                            m = ((long)bt * (2 * g.Wc) + 2 * hh + (g.cls >> 1)) * (2L * g.Dc) + 2 * ww + (g.cls & 1);
                        }
                        const long off = m * g.ldo + n;
                        f32x4 v = *reinterpret_cast<const f32x4*>(tb + row * 36 + ec) + badd;
                        if (g.act == 1) {
                            if (g.pre_out) *reinterpret_cast<f32x4*>(g.pre_out + off) = v;
#pragma unroll
                            for (int k = 0; k < 4; ++k) v[k] = gelu_f(v[k]);
                        } else if (g.act == 2) {
                            const f32x4 ax = *reinterpret_cast<const f32x4*>(g.aux + off);
#pragma unroll
                            for (int k = 0; k < 4; ++k) v[k] *= gelu_grad_f(ax[k]);
                        } else if (g.act == 3) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
                        } else if (g.act == 4) {
                            const f32x4 ax = *reinterpret_cast<const f32x4*>(g.aux + off);
#pragma unroll
                            for (int k = 0; k < 4; ++k) v[k] = ax[k] > 0.f ? v[k] : 0.f;
                        }
                        if (g.mask) v = v * *reinterpret_cast<const f32x4*>(g.mask + off);
                        if (g.drop.thr) v = v * dropout4(g.drop, (unsigned long long)off >> 2);
                        v += vadd;
                        if (g.residual) v += *reinterpret_cast<const f32x4*>(g.residual + off);
                        *reinterpret_cast<f32x4*>(g.out + off) = v;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + col;
        if (n >= g.N) continue;
        const float badd = (g.bias ? g.bias[n] : 0.f);
        const float vadd = (g.addvec ? g.addvec[n] : 0.f);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long mi = m0 + (wm * TM + i) * 32 + mfma_row(lane, r);
                if (mi < g.M) {
                    long m = mi;
                    if (CM == 3) {                       // (b, t, h, w) -> row (b, t, 2h+ph, 2w+pw) of the up-sampled mesh
                        const long per = (long)g.Wc * g.Dc;
                        const long bt = mi / per;
                        const int r2 = (int)(mi - bt * per);
                        const int hh = r2 / g.Dc, ww = r2 - hh * g.Dc;
                        m = (bt * (2 * g.Wc) + 2 * hh + (g.cls >> 1)) * (2L * g.Dc) + 2 * ww + (g.cls & 1);
                    }
                    float v = acc[i][j][r] + badd;
                    if (g.act == 1) {
                        if (g.pre_out) g.pre_out[m * g.ldo + n] = v;
                        v = gelu_f(v);
                    } else if (g.act == 2) {
                        v *= gelu_grad_f(g.aux[m * g.ldo + n]);
                    } else if (g.act == 3) {
                        v = fmaxf(v, 0.f);
                    } else if (g.act == 4) {
                        v = g.aux[m * g.ldo + n] > 0.f ? v : 0.f;
                    }
                    if (g.mask) v *= g.mask[m * g.ldo + n];
                    v += vadd;
                    if (g.residual) v += g.residual[m * g.ldo + n];
                    g.out[m * g.ldo + n] = v;
                }
            }
        }
    }
}

template <int WM, int WN, int TM, int TN, int CM>
static void launch_gemm_cm(const GemmArgs& g, unsigned tiles, size_t lds, hipStream_t st) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<WM, WN, TM, TN, CM>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, TM, TN, CM>), dim3(tiles), dim3(WM * WN * 64), lds, st, g);
}

template <int WM, int WN, int TM, int TN>
static int launch_gemm(const GemmArgs& g, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const size_t lds = (size_t)(BM + BN) * (G_BK + 1) * 4;
    const long tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    RPB_REQUIRE(tiles > 0 && tiles < (1L << 31), "gemm: bad tile count");
    switch (g.conv) {
        case 0: launch_gemm_cm<WM, WN, TM, TN, 0>(g, (unsigned)tiles, lds, st); break;
        case 1: launch_gemm_cm<WM, WN, TM, TN, 1>(g, (unsigned)tiles, lds, st); break;
        case 2: launch_gemm_cm<WM, WN, TM, TN, 2>(g, (unsigned)tiles, lds, st); break;
        default: launch_gemm_cm<WM, WN, TM, TN, 3>(g, (unsigned)tiles, lds, st); break;
    }
    RPB_CHECK_LAUNCH("gemm_nt");
}

extern "C" int rpb_gemm_nt(const float* A, const float* W, const float* bias, const float* addvec, const float* residual,
                           float* out, long M, int N, int K, int lda, int ldo, int act, const float* aux, float* pre_out,
                           const float* mask, int conv, int Hc, int Wc, int Dc, int cls, long drop_seed, float drop_keep,
                           void* stream) {
    RPB_REQUIRE(A && W && out, "gemm_nt: null pointer");
    RPB_REQUIRE((act != 2 && act != 4) || aux, "gemm_nt: act=2/4 needs the saved activation tensor");
    RPB_REQUIRE(act >= 0 && act <= 4, "gemm_nt: unknown act=%d", act);
    RPB_REQUIRE(M > 0 && M < (1L << 31) && N > 0 && K > 0 && K % G_BK == 0,
                "gemm_nt: bad sizes M=%ld N=%d K=%d (M < 2^31, K a multiple of %d)", M, N, K, G_BK);
    RPB_REQUIRE(lda % 4 == 0 && ldo >= N, "gemm_nt: lda=%d must be a multiple of 4 and ldo=%d >= N", lda, ldo);
    RPB_REQUIRE(conv >= 0 && conv <= 3 && cls >= 0 && cls <= 3, "gemm_nt: bad conv=%d / cls=%d", conv, cls);
    if (conv == 1) {
        RPB_REQUIRE(K % 27 == 0 && (K / 27) % G_BK == 0 && Hc > 0 && Wc > 0 && Dc > 0 && M % ((long)Hc * Wc * Dc) == 0,
                    "gemm_nt: bad convolution geometry");
    } else if (conv == 2) {
        RPB_REQUIRE(K % 16 == 0 && (K / 16) % G_BK == 0 && Hc > 0 && Wc > 0 && Dc > 0 && Wc % 2 == 0 && Dc % 2 == 0 &&
                        M % ((long)Hc * (Wc / 2) * (Dc / 2)) == 0,
                    "gemm_nt: bad strided-convolution geometry");
    } else if (conv == 3) {
        RPB_REQUIRE(K % 4 == 0 && (K / 4) % G_BK == 0 && Hc > 0 && Wc > 0 && Dc > 0 && M % ((long)Hc * Wc * Dc) == 0,
                    "gemm_nt: bad transposed-convolution geometry");
        RPB_REQUIRE(!aux && !pre_out && !mask && !residual, "gemm_nt: conv=3 supports the bias epilogue only");
    }
    GemmArgs g;
    g.A = A; g.W = W; g.bias = bias; g.addvec = addvec; g.residual = residual; g.out = out;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldo = ldo; g.act = act; g.aux = aux; g.pre_out = pre_out; g.mask = mask;
    g.conv = conv; g.Hc = Hc; g.Wc = Wc; g.Dc = Dc; g.cls = cls;
    g.drop = make_drop(drop_seed, drop_keep);
    if (g.drop.thr) RPB_REQUIRE(!mask && N % 4 == 0 && ldo % 4 == 0 && conv != 3, "gemm_nt: in-kernel dropout needs N, ldo multiples of 4 and no mask tensor");
    hipStream_t st = (hipStream_t)stream;
    if (N > 64) return launch_gemm<2, 2, 2, 2>(g, st);      // 128 x 128 tile
    if (N > 32) return launch_gemm<2, 2, 2, 1>(g, st);      // 128 x 64
    return launch_gemm<4, 1, 1, 1>(g, st);                   // 128 x 32 (small heads, e.g. mlp2: 256 -> 3)
}

// ---------------------------------------------------------------------------------- TN GEMM (weight gradients)
//   dW[n][k] = sum_m G[m][n] * A(m,k),   db[n] = sum_m G[m][n]          (autograd of nn.Linear / nn.Conv3d weights)
// Both operands are "lane = channel" in HBM, exactly what the MFMA wants when the reduction index is the token:
// they are loaded straight into registers (G: float2 = channel pair col*2+t; A: float2/float4), no LDS.  Workgroup =
// 2x2 waves x (64 x 64|128) = a 128 x 128|256 tile of dW; the token range is split across gridDim.y and each split writes a partial that
// rpb_reduce_partials sums in fp64.  With conv=1, A(m, tap*Ci+ci) = x[neighbour(m,tap)][ci] (a 128-column k tile never
// crosses a tap because Ci % 64 == 0 and every wave derives the tap from its own 64-column sub-tile).
struct GemmTnArgs {
    const float* G;      // [M][ldg]
    const float* A;      // [M][lda]
    float* part;         // [splits][N*K + N]
    long M;
    int N, K, ldg, lda;
    int conv, Hc, Wc, Dc;
};

template <int N>
struct TnVec;
template <>
struct TnVec<2> {
    typedef f32x2 T;
};
template <>
struct TnVec<4> {
    typedef f32x4 T;
};

// NTI = k-tiles (of 32 channels) per wave: wave tile = 64 (n) x 32*NTI (k), workgroup = 2 x 2 waves.
// WN = waves along n per workgroup (2 or 4): workgroup tile = 64*WN (n) x 64*NTI (k); the bigger tile halves the
// re-reads of A (every n-tile streams the whole gathered activation tensor once per tap)
template <int NTI, int WN, int CM>
__global__ __launch_bounds__(WN * 128) void gemm_tn_kernel(GemmTnArgs g) {
    typedef typename TnVec<NTI>::T veci;
    constexpr int WK = 32 * NTI;                       // k columns per wave
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wn = wave >> 1, wk = wave & 1;              // WN x 2 waves
    const int col = lane & 31, half = lane >> 5;
    const int tk = (g.K + 2 * WK - 1) / (2 * WK);
    // XCD-aware tile order: workgroup b runs on XCD b % 8, so give every XCD a CONTIGUOUS range of tiles.  The tk
    // k-tiles of one n-tile re-read the same G rows (and neighbouring taps the same A rows): on one XCD they hit its L2
    int tile = blockIdx.x;
    if (gridDim.x % 8 == 0) tile = (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;
    const int tile_n = tile / tk, tile_k = tile % tk;
    const int n0 = tile_n * 64 * WN + wn * 64, k0 = tile_k * 2 * WK + wk * WK;     // this wave's sub-tile
    const int nsplit = gridDim.y, split = blockIdx.y;
    const long per = ((g.M + nsplit - 1) / nsplit + 31) / 32 * 32;
    const long mb = (long)split * per;
    long me = mb + per;
    if (me > g.M) me = g.M;
    const int ntap = CM == 2 ? 16 : 27;
    const int Ci = CM ? g.K / ntap : g.K;
    int tap = 0, kk0 = k0, dh = 0, dw = 0, dd = 0;
    long noff = 0;
    if (CM) {
        tap = k0 / Ci;
        kk0 = k0 - tap * Ci;
        if (CM == 1) {
            dh = tap / 9 - 1;
            dw = (tap / 3) % 3 - 1;
            dd = tap % 3 - 1;
            noff = ((long)dh * g.Wc + dw) * g.Dc + dd;
        } else {                       // conv = 2: (1,4,4) kernel, stride (1,2,2), padding (0,1,1); G rows = output tokens
            dw = (tap >> 2) - 1;
            dd = (tap & 3) - 1;
        }
    }
    const bool tap_ok = !CM || tap < ntap;
    const int Wo = g.Wc / 2, Do = g.Dc / 2;              // conv = 2: output mesh (Hc, Wo, Do)
    const bool n_ok0 = n0 + col * 2 < g.N, n_ok1 = n0 + col * 2 + 1 < g.N;      // N may be < 64 (e.g. mlp2: 3)
    const int klim = CM ? Ci : g.K;
    const long mesh = (long)g.Hc * g.Wc * g.Dc;

    f32x16 acc[2][NTI];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int i = 0; i < NTI; ++i) acc[o][i] = zero16();
    float bsum[2] = {0.f, 0.f};

    f32x2 ga[8], gb[8];
    veci xa[8], xb[8];
    const veci zi = {};
    // m0 is wave-uniform; the mesh coordinates of m0 + j (j < 32) follow from ONE scalar decomposition of m0 plus a
    // division-free carry (float reciprocal of the small extents), instead of three integer divisions per load
    // A lane's tokens form ONE arithmetic progression mb + half, +2, +4, ... across steps, halves and chunks, so its
    // mesh coordinates are carried incrementally (add 2 with carry): no divisions and ~10 VALU ops per load
    int ld = 0, lw = 0, lh = 0;
    long lbt = 0;                                        // conv = 2: frame index b*T + t of this lane's token
    if (CM == 1) {
        const long r = (mb + half) % mesh;
        ld = (int)(r % g.Dc);
        lw = (int)((r / g.Dc) % g.Wc);
        lh = (int)(r / ((long)g.Dc * g.Wc));
    } else if (CM == 2) {
        const long r = mb + half;
        ld = (int)(r % Do);
        lw = (int)((r / Do) % Wo);
        lbt = r / ((long)Do * Wo);
    }
    // running row pointers of this lane's token progression (mb + half, +2, +4, ...): one live 64-bit pointer per operand
    // instead of a (m * ld) product per step -- the plain (CM = 0) instantiation spilled 64 registers without this
    const float* gcur = g.G + (mb + half) * g.ldg + n0 + col * 2;
    const float* xcur = g.A + (mb + half + noff) * g.lda + kk0 + col * NTI;
    auto load_half = [&](long m0, int h, f32x2 (&gv)[8], veci (&xv)[8]) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int j = h * 16 + 2 * s + half;
            const long m = m0 + j;
            f32x2 gq = {0.f, 0.f};
            veci xq = zi;
            bool ok = tap_ok;
            long atok = 0;
            if (CM == 1) {
                const int hh = lh + dh, ww = lw + dw, d2 = ld + dd;
                ok = ok && hh >= 0 && hh < g.Hc && ww >= 0 && ww < g.Wc && d2 >= 0 && d2 < g.Dc;
                ld += 2;                                             // advance this lane to its next token
                while (ld >= g.Dc) {
                    ld -= g.Dc;
                    if (++lw >= g.Wc) {
                        lw = 0;
                        if (++lh >= g.Hc) lh = 0;
                    }
                }
            } else if (CM == 2) {
                const int ww = 2 * lw + dw, d2 = 2 * ld + dd;
                ok = ok && ww >= 0 && ww < g.Wc && d2 >= 0 && d2 < g.Dc;
                atok = (lbt * g.Wc + ww) * g.Dc + d2;
                ld += 2;
                while (ld >= Do) {
                    ld -= Do;
                    if (++lw >= Wo) {
                        lw = 0;
                        ++lbt;
                    }
                }
            }
            if (m < me) {
                const float* gp = gcur;
                if (n_ok1) gq = *reinterpret_cast<const f32x2*>(gp);
                else if (n_ok0) gq[0] = gp[0];
                if (ok) {
                    const float* xp = (CM == 2) ? g.A + atok * g.lda + kk0 + col * NTI : xcur;
                    if (kk0 + col * NTI + NTI <= klim) xq = *reinterpret_cast<const veci*>(xp);
                    else {
#pragma unroll
                        for (int i = 0; i < NTI; ++i)
                            if (kk0 + col * NTI + i < klim) xq[i] = xp[i];
                    }
                }
            }
            gv[s] = gq;
            xv[s] = xq;
            gcur += 2 * g.ldg;
            xcur += 2 * g.lda;
        }
    };
    auto compute_half = [&](const f32x2 (&gv)[8], const veci (&xv)[8]) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                bsum[o] += gv[s][o];
#pragma unroll
                for (int i = 0; i < NTI; ++i) acc[o][i] = mfma32(gv[s][o], xv[s][i], acc[o][i]);
            }
    };
    long m = mb;
    if (m < me) load_half(m, 0, ga, xa);
    for (; m < me; m += 32) {
        load_half(m, 1, gb, xb);                  // NOTE: load_half must be called in token order (lane state above)
        compute_half(ga, xa);
        if (m + 32 < me) load_half(m + 32, 0, ga, xa);
        compute_half(gb, xb);
    }
    float* part = g.part + (long)split * ((long)g.N * g.K + g.N);
#pragma unroll
    for (int o = 0; o < 2; ++o) {
#pragma unroll
        for (int i = 0; i < NTI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + mfma_row(lane, r) * 2 + o;
                const int k = k0 + col * NTI + i;
                if (n < g.N && k < g.K && kk0 + col * NTI + i < klim) part[(long)n * g.K + k] = acc[o][i][r];
            }
        if (tile_k == 0 && wk == 0) {
            const float b = bsum[o] + __shfl_xor(bsum[o], 32, 64);
            const int n = n0 + col * 2 + o;
            if (half == 0 && n < g.N) part[(long)g.N * g.K + n] = b;
        }
    }
}


// ---------------------------------------------------------------------------------- 3x3x3 convolution weight gradient, LDS-tiled
//   dW[co][((kt*3+kh)*3+kw)*Ci + ci] = sum_tok G[tok][co] * X[neighbour(tok, kt, kh, kw)][ci]
// The register-operand TN kernel above re-reads both operands from L2 for every 64 x 128 wave tile (21 flop/B) and runs
// the convolution gradient at half the speed of the forward GEMM.  Here a workgroup owns a 64 (co) x [3 kh][3 kw][64 ci]
// block of dW for one kt and a token split: per 32-token chunk it stages the G rows (8 KB) and, per kh, ONE run of 34
// consecutive x rows (tokens m0-1 .. m0+32 shifted by (kt*H + kh)*W) that serves all three kw taps -- 34 KB of L2 traffic
// for 576 MFMAs (69 flop/B).  Both MFMA operands are read from LDS "row = token, lanes across channels" (conflict-free,
// no transposition); mesh-boundary validity is a 0/1 factor per (token, tap) applied to the x operand.
struct ConvWgArgs {
    const float* G;      // [M][ldg]
    const float* X;      // [M][ldx]
    float* part;         // [splits][Co*27*Ci + Co]
    long M;
    int Co, Ci, ldg, ldx, T, H, W;
};

#define CW_TOK 32

// Workgroup = 4 waves, one per (co half o, ci half c) of the 64 x 64 channel block; every wave accumulates all nine
// (kh, kw) taps of its 32 x 32 sub-block (9 MFMA tiles = 144 accumulator registers), so all four SIMDs of the CU are busy
// (a 3-wave "wave = kh" mapping with 64 x 192 wave tiles left one SIMD idle: 67-73 TF/s instead of 75-81).  The chunk tiles
// are single-buffered on purpose: the double-buffered, one-barrier-per-chunk variant measured 4 % slower.
__global__ __launch_bounds__(256) void conv3_wgrad_kernel(ConvWgArgs a) {
    // Zero padding without per-MFMA masks: a tap (kt, kh, kw) of token (t, h, w) is valid iff t+kt-1, h+kh-1, w+kw-1 are in
    // range.  The w condition depends on the token only -> three copies of the G tile (kw = 0: zero where w = 0, kw = 1: plain,
    // kw = 2: zero where w = W-1).  Given it, the X row q = tok + shift has no carry out of w, and the t / h conditions
    // become properties of the ROW (0 <= h_q - kh + 1 < H, 0 <= t_q - kt + 1 < T) -> invalid rows are simply not loaded.
    // (The first version multiplied every B operand by a validity table: 2 LDS reads + 1 v_mul per MFMA, 91 TF/s.)
    __shared__ float Gs[3][CW_TOK][64];
    __shared__ float Xs[3][CW_TOK + 2][64];
    __shared__ int Rm[2][3 * (CW_TOK + 2)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int o = wave & 1, c = wave >> 1;
    const int col = lane & 31, half = lane >> 5;
    const int ncb = a.Ci / 64;
    int bx = blockIdx.x;
    const int kt = bx % 3;
    bx /= 3;
    const int cib = bx % ncb, cob = bx / ncb;
    const int n0 = cob * 64, ci0 = cib * 64;
    const int nsplit = gridDim.y, split = blockIdx.y;
    const long per = ((a.M + nsplit - 1) / nsplit + CW_TOK - 1) / CW_TOK * CW_TOK;
    const long mb = (long)split * per;
    long me = mb + per;
    if (me > a.M) me = a.M;

    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = zero16();
    float bsum = 0.f;
    const bool do_bias = (cib == 0 && kt == 1 && c == 0);

    // staging map: G tile = 512 float4, X tiles = 3 * 34 * 16 = 1632 float4 over 256 threads
    constexpr int NG = (CW_TOK * 16 + 255) / 256;                     // 2
    constexpr int NXL = (3 * (CW_TOK + 2) * 16 + 255) / 256;          // 7
    constexpr int NROW = 3 * (CW_TOK + 2);                            // 102 staged X rows
    f32x4 pg[NG], px[NXL];
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    auto prefetch = [&](long m0, int buf) {
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int idx = tid + j * 256;
            f32x4 v = z4;
            if (idx < CW_TOK * 16) {
                const long m = m0 + (idx >> 4);
                if (m < me) v = *reinterpret_cast<const f32x4*>(a.G + m * a.ldg + n0 + (idx & 15) * 4);
            }
            pg[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NXL; ++j) {
            const int idx = tid + j * 256;
            f32x4 v = z4;
            if (idx < NROW * 16) {
                const int k2 = idx / ((CW_TOK + 2) * 16), rem = idx - k2 * (CW_TOK + 2) * 16;
                const long q = m0 - 1 + (rem >> 4) + ((long)(kt - 1) * a.H + (k2 - 1)) * a.W;
                if (Rm[buf][idx >> 4] && q >= 0 && q < a.M) v = *reinterpret_cast<const f32x4*>(a.X + q * a.ldx + ci0 + (rem & 15) * 4);
            }
            px[j] = v;
        }
    };
    // mesh coordinates are decoded once (64-bit divisions) and then carried chunk to chunk with 32-bit adds:
    //   threads 0..101 own one staged X row each (its t / h validity), every thread the w of its two G-tile tokens
    int qw = 0, qh = 0, qt = 0;
    const int rk2 = tid / (CW_TOK + 2);
    if (tid < NROW) {
        const long q = mb - 1 + (tid - rk2 * (CW_TOK + 2)) + ((long)(kt - 1) * a.H + (rk2 - 1)) * a.W + 3L * a.T * a.H * a.W;
        qw = (int)(q % a.W);
        const long r = q / a.W;
        qh = (int)(r % a.H);
        qt = (int)((r / a.H) % a.T);
    }
    auto row_mask = [&](int buf) {
        if (tid < NROW) {
            const int tt = qt - kt + 1, hh = qh - rk2 + 1;
            Rm[buf][tid] = (tt >= 0 && tt < a.T && hh >= 0 && hh < a.H) ? 1 : 0;
            qw += CW_TOK;                                             // the same slot of the next chunk: + 32 with carries
            while (qw >= a.W) {
                qw -= a.W;
                if (++qh >= a.H) {
                    qh = 0;
                    if (++qt >= a.T) qt = 0;
                }
            }
        }
    };
    int gw[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) gw[j] = (int)((mb + ((tid + j * 256) >> 4)) % a.W);
    row_mask(0);
    __syncthreads();
    if (mb < me) prefetch(mb, 0);
    int it = 0;
    for (long m0 = mb; m0 < me; m0 += CW_TOK, ++it) {
        __syncthreads();                                              // previous chunk's LDS reads are done
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int idx = tid + j * 256;
            if (idx < CW_TOK * 16) {
                *reinterpret_cast<f32x4*>(&Gs[0][idx >> 4][(idx & 15) * 4]) = gw[j] >= 1 ? pg[j] : z4;
                *reinterpret_cast<f32x4*>(&Gs[1][idx >> 4][(idx & 15) * 4]) = pg[j];
                *reinterpret_cast<f32x4*>(&Gs[2][idx >> 4][(idx & 15) * 4]) = gw[j] <= a.W - 2 ? pg[j] : z4;
            }
            gw[j] += CW_TOK;
            while (gw[j] >= a.W) gw[j] -= a.W;
        }
#pragma unroll
        for (int j = 0; j < NXL; ++j) {
            const int idx = tid + j * 256;
            if (idx < NROW * 16) {
                const int k2 = idx / ((CW_TOK + 2) * 16), rem = idx - k2 * (CW_TOK + 2) * 16;
                *reinterpret_cast<f32x4*>(&Xs[k2][rem >> 4][(rem & 15) * 4]) = px[j];
            }
        }
        row_mask((it + 1) & 1);                                       // for the prefetch below
        __syncthreads();
        if (m0 + CW_TOK < me) prefetch(m0 + CW_TOK, (it + 1) & 1);    // in flight during the MFMAs below
        // X rows tk + {0, 1, 2}: the kw = 2 operand of step s is the kw = 0 operand of step s + 1 (rows advance by 2)
        float x0[3], x1[3];
#pragma unroll
        for (int k2 = 0; k2 < 3; ++k2) {
            x0[k2] = Xs[k2][half][c * 32 + col];
            x1[k2] = Xs[k2][half + 1][c * 32 + col];
        }
#pragma unroll
        for (int s = 0; s < CW_TOK / 2; ++s) {
            const int tk = 2 * s + half;
            const float a0 = Gs[0][tk][o * 32 + col], a1 = Gs[1][tk][o * 32 + col], a2 = Gs[2][tk][o * 32 + col];
            if (do_bias) bsum += a1;
#pragma unroll
            for (int k2 = 0; k2 < 3; ++k2) {
                const float x2 = Xs[k2][tk + 2][c * 32 + col];
                acc[k2 * 3 + 0] = mfma32(a0, x0[k2], acc[k2 * 3 + 0]);
                acc[k2 * 3 + 1] = mfma32(a1, x1[k2], acc[k2 * 3 + 1]);
                acc[k2 * 3 + 2] = mfma32(a2, x2, acc[k2 * 3 + 2]);
                x0[k2] = x2;
                if (s + 1 < CW_TOK / 2) x1[k2] = Xs[k2][tk + 3][c * 32 + col];
            }
        }
    }
    const long K = 27L * a.Ci;
    float* part = a.part + (long)split * ((long)a.Co * K + a.Co);
#pragma unroll
    for (int k2 = 0; k2 < 3; ++k2)
#pragma unroll
        for (int k3 = 0; k3 < 3; ++k3)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + o * 32 + mfma_row(lane, r);
                const long k = ((long)(kt * 3 + k2) * 3 + k3) * a.Ci + ci0 + c * 32 + col;
                part[(long)n * K + k] = acc[k2 * 3 + k3][r];
            }
    if (do_bias) {
        const float bb = bsum + __shfl_xor(bsum, 32, 64);
        if (half == 0) part[(long)a.Co * K + n0 + o * 32 + col] = bb;
    }
}

// Split count for equal-length workgroups: a CU that receives one more of them than its neighbours sets the kernel time, so
// among 1..smax the count that makes tiles * splits (nearly) a multiple of the CU count wins, larger counts on ties
// (Transolver's conv weight gradient: 96 tiles x 11 splits = 4.1 per CU ran 5 rounds at 105 TF/s; x 8 = 3.0 per CU: 124).
static int balanced_splits(long tiles, long per_cu, long cap, long hard_max) {
    const long ncu = rpb_num_cus();
    long smax = (ncu * per_cu + tiles - 1) / tiles;
    if (smax > cap) smax = cap;
    if (smax > hard_max) smax = hard_max;
    if (smax < 1) smax = 1;
    long best = 1;
    double best_fill = 0.0;
    for (long sp = 1; sp <= smax; ++sp) {
        const long blocks = tiles * sp, rounds = (blocks + ncu - 1) / ncu;
        const double fill = (double)blocks / (double)(rounds * ncu);
        if (fill >= best_fill) {
            best_fill = fill;
            best = sp;
        }
    }
    return (int)best;
}

static int conv3_wgrad_splits(long M, int Co, int Ci) {
    return balanced_splits((long)(Co / 64) * (Ci / 64) * 3, 4, (M + 1023) / 1024 /* >= 1024 tokens per split */, 512);
}


// ---------------------------------------------------------------------------------- projection weight gradient, LDS-tiled
//   dW[n][k] = sum_tok G[tok][n] * A[tok][k]   for small N x K (the U-Net / Galerkin projections: N, K in 64..384)
// These are HBM-streaming problems (a token contributes N + K floats and 2*N*K flops): the register-operand kernel below
// re-reads G once per k-tile column and runs them at 11-22 TF/s / 0.5 TB/s.  Here a workgroup owns the WHOLE N range and a
// 64*WK-wide k block for one token split, so G and A are read once (grid.x = K / (64*WK) is 1 for K <= 128): per
// 32-token chunk both row blocks are staged in LDS and every wave multiplies its NTW x 2 tiles.
struct TnSmallArgs {
    const float* G;      // [M][ldg]
    const float* A;      // [M][lda]
    float* part;         // [splits][N*K + N]
    long M;
    int N, K, ldg, lda;
};

template <int WN, int NTW>
__global__ __launch_bounds__(256) void tn_small_kernel(TnSmallArgs a) {
    constexpr int WK = 4 / WN, BN = WN * NTW * 32, BK = WK * 64;
    __shared__ float Gs[CW_TOK][BN];
    __shared__ float As[CW_TOK][BK];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WK, wk = wave % WK;
    const int col = lane & 31, half = lane >> 5;
    const int k0 = blockIdx.x * BK;
    const int nsplit = gridDim.y, split = blockIdx.y;
    const long per = ((a.M + nsplit - 1) / nsplit + CW_TOK - 1) / CW_TOK * CW_TOK;
    const long mb = (long)split * per;
    long me = mb + per;
    if (me > a.M) me = a.M;
    f32x16 acc[NTW][2];
#pragma unroll
    for (int i = 0; i < NTW; ++i) acc[i][0] = acc[i][1] = zero16();
    float bsum[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) bsum[i] = 0.f;
    const bool do_bias = (blockIdx.x == 0 && wk == 0);
    constexpr int G4 = BN / 4, A4 = BK / 4;
    constexpr int NGL = (CW_TOK * G4 + 255) / 256, NAL = (CW_TOK * A4 + 255) / 256;
    f32x4 pg[NGL], pa[NAL];
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    auto prefetch = [&](long m0) {
#pragma unroll
        for (int j = 0; j < NGL; ++j) {
            const int idx = tid + j * 256;
            f32x4 v = z4;
            if (idx < CW_TOK * G4) {
                const long m = m0 + idx / G4;
                if (m < me) v = *reinterpret_cast<const f32x4*>(a.G + m * a.ldg + (idx % G4) * 4);
            }
            pg[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NAL; ++j) {
            const int idx = tid + j * 256;
            f32x4 v = z4;
            if (idx < CW_TOK * A4) {
                const long m = m0 + idx / A4;
                if (m < me) v = *reinterpret_cast<const f32x4*>(a.A + m * a.lda + k0 + (idx % A4) * 4);
            }
            pa[j] = v;
        }
    };
    if (mb < me) prefetch(mb);
    for (long m0 = mb; m0 < me; m0 += CW_TOK) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NGL; ++j) {
            const int idx = tid + j * 256;
            if (idx < CW_TOK * G4) *reinterpret_cast<f32x4*>(&Gs[idx / G4][(idx % G4) * 4]) = pg[j];
        }
#pragma unroll
        for (int j = 0; j < NAL; ++j) {
            const int idx = tid + j * 256;
            if (idx < CW_TOK * A4) *reinterpret_cast<f32x4*>(&As[idx / A4][(idx % A4) * 4]) = pa[j];
        }
        __syncthreads();
        if (m0 + CW_TOK < me) prefetch(m0 + CW_TOK);
#pragma unroll
        for (int s = 0; s < CW_TOK / 2; ++s) {
            const int tk = 2 * s + half;
            const float b0 = As[tk][wk * 64 + col], b1 = As[tk][wk * 64 + 32 + col];
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                const float av = Gs[tk][(wn * NTW + i) * 32 + col];
                if (do_bias) bsum[i] += av;
                acc[i][0] = mfma32(av, b0, acc[i][0]);
                acc[i][1] = mfma32(av, b1, acc[i][1]);
            }
        }
    }
    float* part = a.part + (long)split * ((long)a.N * a.K + a.N);
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = (wn * NTW + i) * 32 + mfma_row(lane, r);
                part[(long)n * a.K + k0 + wk * 64 + c * 32 + col] = acc[i][c][r];
            }
        if (do_bias) {
            const float bb = bsum[i] + __shfl_xor(bsum[i], 32, 64);
            if (half == 0) part[(long)a.N * a.K + (wn * NTW + i) * 32 + col] = bb;
        }
    }
}

// shapes served by tn_small_kernel: N in {64, 128, 256, 384}, K a multiple of the k block; N = 64: K <= 1280, otherwise
// K <= 384 and N*K <= 48 tiles (256 x 256 and beyond run faster on the register-operand kernel: 100-121 TF/s)
static bool tn_small_ok(int N, int K, int ldg, int lda) {
    if (!(N == 64 || N == 128 || N == 256 || N == 384) || ldg % 4 || lda % 4) return false;
    const int bk = (N == 64) ? 128 : 64;
    if (K % bk) return false;
    if (N == 64) return K <= 1280;                      // narrow G: re-reading it once per k block is cheap (U-Net init_conv: K = 1152)
    return K <= 384 && (long)N * K <= 49152;
}

// ---------------------------------------------------------------------------------- weight gradient of a lift layer (K <= 4)
//   dW[n][k] = sum_tok G[tok][n] * A[tok][k],  db[n] = sum_tok G[tok][n]   for K <= 4 input features (coordinates / channels of
// the first nn.Linear): a pure stream over G (the MFMA kernels above spent 1.75 ms on 1.3 GB: one 32-wide k tile of zeros).
// Thread = 4 consecutive columns, workgroup = TKT_ROWS row lanes x N/4 column groups; partial layout of rpb_gemm_tn.
struct TnTinyArgs {
    const float* G;
    const float* A;
    float* part;      // [gridDim.x][N*K + N]
    long M;
    int N, K, ldg, lda;
};

__global__ __launch_bounds__(256) void tn_tiny_kernel(TnTinyArgs a) {
    extern __shared__ float red[];                                      // [nsub][(K + 1) * N]
    const int n4n = a.N >> 2;
    const int c4 = threadIdx.x % n4n, sub = threadIdx.x / n4n, nsub = blockDim.x / n4n;
    f32x4 acc[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (sub < nsub)
        for (long m = (long)blockIdx.x * nsub + sub; m < a.M; m += (long)gridDim.x * nsub) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(a.G + m * a.ldg + 4 * c4);
            const f32x4 x = *reinterpret_cast<const f32x4*>(a.A + m * a.lda);      // lda % 4 == 0, columns >= K are padding
            acc[0] += g * x[0];
            acc[1] += g * x[1];
            acc[2] += g * x[2];
            acc[3] += g * x[3];
            acc[4] += g;
        }
    const int L = (a.K + 1) * a.N;
    if (sub < nsub) {
#pragma unroll
        for (int k = 0; k < 5; ++k)
            if (k < a.K || k == 4) {
                const int slot = k == 4 ? a.K : k;
                *reinterpret_cast<f32x4*>(red + sub * L + slot * a.N + 4 * c4) = acc[k];
            }
    }
    __syncthreads();
    float* prow = a.part + (long)blockIdx.x * ((long)a.N * a.K + a.N);
    for (int idx = threadIdx.x; idx < L; idx += blockDim.x) {
        float sacc = 0.f;
        for (int q = 0; q < nsub; ++q) sacc += red[q * L + idx];
        const int slot = idx / a.N, n = idx - slot * a.N;
        if (slot < a.K) prow[(long)n * a.K + slot] = sacc;
        else prow[(long)a.N * a.K + n] = sacc;
    }
}

static bool tn_tiny_ok(int N, int K, int ldg, int lda) {
    return K <= 4 && N % 4 == 0 && N / 4 <= 256 && 256 % (N / 4) == 0 && ldg % 4 == 0 && lda % 4 == 0 && lda >= 4;
}
static int tn_tiny_splits() { return rpb_num_cus() * 4; }

static int tn_small_splits(long M, int N, int K) {
    const int bk = (N == 64) ? 128 : 64;
    return balanced_splits(K / bk, 3, (M + 1023) / 1024, 1024);
}

static int gemm_tn_nti(int K, int conv) {
    const int Ci = conv == 1 ? K / 27 : (conv == 2 ? K / 16 : K);
    return (Ci % 128 == 0) ? 4 : 2;
}

static int gemm_tn_wn(int N) { return N > 128 ? 4 : (N > 64 ? 2 : 1); }      // waves along n: no idle half for N <= 64

extern "C" int rpb_gemm_tn_splits(long M, int N, int K, int conv) {
    if (conv == 1 && N % 64 == 0 && K % 27 == 0 && (K / 27) % 64 == 0) return conv3_wgrad_splits(M, N, K / 27);
    if (conv == 0 && tn_small_ok(N, K, 4, 4)) return tn_small_splits(M, N, K);
    if (conv == 0 && K <= 4 && N % 4 == 0 && N / 4 <= 256 && 256 % (N / 4) == 0) return tn_tiny_splits();
    const int wk2 = 64 * gemm_tn_nti(K, conv);
    const int bn = 64 * gemm_tn_wn(N);
    const long tiles = (long)((N + bn - 1) / bn) * ((K + wk2 - 1) / wk2);
    return balanced_splits(tiles, 3, (M + 511) / 512 /* >= 512 tokens per split */, 256);
}

extern "C" int rpb_gemm_tn(const float* G, const float* A, float* part, long M, int N, int K, int ldg, int lda, int conv,
                           int Hc, int Wc, int Dc, void* stream) {
    RPB_REQUIRE(G && A && part && M > 0 && N > 0 && K > 0, "gemm_tn: bad arguments");
    RPB_REQUIRE(ldg % 2 == 0 && lda % 2 == 0, "gemm_tn: leading dimensions must be even (float2 loads)");
    RPB_REQUIRE(conv >= 0 && conv <= 2, "gemm_tn: conv=%d", conv);
    if (conv == 1) RPB_REQUIRE(K % 27 == 0 && (K / 27) % 64 == 0 && Hc > 0 && Wc > 0 && Dc > 0 && M % ((long)Hc * Wc * Dc) == 0,
                               "gemm_tn: bad convolution geometry (Ci must be a multiple of 64)");
    if (conv == 2) RPB_REQUIRE(K % 16 == 0 && (K / 16) % 64 == 0 && Hc > 0 && Wc > 0 && Dc > 0 && Wc % 2 == 0 && Dc % 2 == 0 &&
                                   M % ((long)Hc * (Wc / 2) * (Dc / 2)) == 0,
                               "gemm_tn: bad strided-convolution geometry (Ci must be a multiple of 64)");
    GemmTnArgs a;
    a.G = G; a.A = A; a.part = part; a.M = M; a.N = N; a.K = K; a.ldg = ldg; a.lda = lda;
    a.conv = conv; a.Hc = Hc; a.Wc = Wc; a.Dc = Dc;
    if (conv == 1 && N % 64 == 0) {                     // LDS-tiled convolution weight gradient
        RPB_REQUIRE(ldg % 4 == 0 && lda % 4 == 0, "gemm_tn: conv leading dimensions must be multiples of 4");
        ConvWgArgs c{G, A, part, M, N, K / 27, ldg, lda, Hc, Wc, Dc};
        const int sp = conv3_wgrad_splits(M, N, K / 27);
        hipLaunchKernelGGL(conv3_wgrad_kernel, dim3((N / 64) * (K / 27 / 64) * 3, sp), dim3(256), 0, (hipStream_t)stream, c);
        RPB_CHECK_LAUNCH("gemm_tn(conv3)");
    }
    if (conv == 0 && K <= 4 && N % 4 == 0 && N / 4 <= 256 && 256 % (N / 4) == 0) {      // lift layers: stream G once
        RPB_REQUIRE(tn_tiny_ok(N, K, ldg, lda), "gemm_tn: K=%d needs ldg %% 4 == 0 and lda %% 4 == 0 (lda >= 4, zero padding)", K);
        TnTinyArgs t{G, A, part, M, N, K, ldg, lda};
        const int nsub = 256 / (N / 4);
        hipLaunchKernelGGL(tn_tiny_kernel, dim3(tn_tiny_splits()), dim3(256), (size_t)nsub * (K + 1) * N * 4, (hipStream_t)stream, t);
        RPB_CHECK_LAUNCH("gemm_tn(tiny)");
    }
    if (conv == 0 && tn_small_ok(N, K, 4, 4)) {         // LDS-tiled projection weight gradient: G and A are read once
        RPB_REQUIRE(ldg % 4 == 0 && lda % 4 == 0, "gemm_tn: leading dimensions %d / %d must be multiples of 4 for N=%d K=%d", ldg, lda, N, K);
        TnSmallArgs t{G, A, part, M, N, K, ldg, lda};
        const int sp = tn_small_splits(M, N, K);
        hipStream_t st2 = (hipStream_t)stream;
        if (N == 64) hipLaunchKernelGGL((tn_small_kernel<2, 1>), dim3(K / 128, sp), dim3(256), 0, st2, t);
        else if (N == 128) hipLaunchKernelGGL((tn_small_kernel<4, 1>), dim3(K / 64, sp), dim3(256), 0, st2, t);
        else if (N == 256) hipLaunchKernelGGL((tn_small_kernel<4, 2>), dim3(K / 64, sp), dim3(256), 0, st2, t);
        else hipLaunchKernelGGL((tn_small_kernel<4, 3>), dim3(K / 64, sp), dim3(256), 0, st2, t);
        RPB_CHECK_LAUNCH("gemm_tn(small)");
    }
    const int nti = gemm_tn_nti(K, conv);
    const int wk2 = 64 * nti;
    RPB_REQUIRE(lda % nti == 0, "gemm_tn: lda=%d must be a multiple of %d", lda, nti);
    const int wn = gemm_tn_wn(N), bn = 64 * wn;
    const int tiles = ((N + bn - 1) / bn) * ((K + wk2 - 1) / wk2);
    const int splits = rpb_gemm_tn_splits(M, N, K, conv);
    hipStream_t st = (hipStream_t)stream;
#define RPB_TN(NTI_, WN_, CM_) hipLaunchKernelGGL((gemm_tn_kernel<NTI_, WN_, CM_>), dim3(tiles, splits), dim3(WN_ * 128), 0, st, a)
#define RPB_TN_CM(NTI_, WN_)                      \
    do {                                          \
        if (conv == 0) RPB_TN(NTI_, WN_, 0);      \
        else if (conv == 1) RPB_TN(NTI_, WN_, 1); \
        else RPB_TN(NTI_, WN_, 2);                \
    } while (0)
    if (nti == 4 && wn == 4) RPB_TN_CM(4, 4);
    else if (nti == 4 && wn == 2) RPB_TN_CM(4, 2);
    else if (nti == 4 && wn == 1) RPB_TN_CM(4, 1);
    else if (wn == 4) RPB_TN_CM(2, 4);
    else if (wn == 1) RPB_TN_CM(2, 1);
    else RPB_TN_CM(2, 2);
#undef RPB_TN_CM
#undef RPB_TN
    RPB_CHECK_LAUNCH("gemm_tn");
}
