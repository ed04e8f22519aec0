// Channels-last "cell" kernels (activations are [cells][channels], channels contiguous = one 256 B line
// at width 64):
//
//   cell_mix     out[cell][o] = sum_k GW[w(cell)][k] * z2[g(cell)][k][o]      (last inverse-DFT stage, K4)
//                             + sum_i x[cell][i] * W[o][i] + bias[o]          (1x1x1 Conv3d, K5)
//                             (+ per-channel sum / sum-of-squares partials for BatchNorm3d)
//                replaces  `x1 + x2` of fno.py:114-116 (and its dgrad with W transposed).
//   cell_wgrad   dW[o][i] = sum_cell gs[cell][o] * x[cell][i],  db[o] = sum_cell gs[cell][o]
//                replaces the Conv3d / Linear weight gradients autograd produces for fno.py:115,123.
//
// Both run on v_mfma_f32_32x32x2_f32 with persistent waves.  A tile = 32 consecutive cells of the
// flattened (b,t,h,w) index, so no MFMA rows are wasted on the 134-wide rows; a tile that straddles two
// w-rows does the (cheap, K=2*m3) spectral part twice with the foreign rows masked to zero.
// The activation tile is transposed through a wave-private, +1-padded LDS tile (conflict-free
// ds_read_b32); weights / DFT matrix sit in block-shared LDS with the MFMA column index contiguous.
#include "rpb_common.h"
#include "rpb_cmx.h"
#include "rpb_pjx.h"
#include "rpb_bwr.h"

// ---------------------------------------------------------------------------------- cell_mix
struct CellMixArgs {
    const float* x;       // [rows_in][KC]
    const float* Wm;      // transpose_w == 0: [CO][KC] (out = x W^T) ; == 1: [KC][CO] (out = x W)
    const float* bias;    // [CO] or null
    const float* z2;      // [G][K2][CO] or null
    const float* GW;      // [K2][Wp]  (transposed stage matrix)
    float* out;           // [ncell][CO]
    float* stats_part;    // [gridDim.x*waves][2][CO] or null
    long ncell;
    int KC, CO, K2, Wp;
    int transpose_w;
    int gather;           // 1: out cells are padded cells, input row = pad_to_crop(cell) (zeros in the margin)
    CropMap cm;
    XForm xf;             // lazy BatchNorm(+GELU) applied to x on its way into LDS
    const float* bnb_s;   // STATS == 2: pre-BN tensor of the layer whose output gradient this launch produces
    XForm bnb;            //             and that layer's BatchNorm (mean, invstd, gamma, beta, gelu)
};

#ifndef CM_MAX_THREADS
#define CM_MAX_THREADS 768      // 12 waves per CU = 3 per SIMD -> <=168 VGPRs
#endif
#ifndef CM_PREF1
#define CM_PREF1 0
#endif
#ifndef CM_UNROLL
#define CM_UNROLL 8
#endif
#define CM_STR2(x) #x
#define CM_STR(x) CM_STR2(x)
#define CM_PRAGMA_UNROLL _Pragma(CM_STR(unroll CM_UNROLL))

template <int N>
struct VecT;
template <>
struct VecT<1> {
    typedef float T;
};
template <>
struct VecT<2> {
    typedef f32x2 T;
};
template <>
struct VecT<4> {
    typedef f32x4 T;
};
template <int N>
__device__ __forceinline__ float vget(const typename VecT<N>::T& v, int i) {
    if constexpr (N == 1) return v;
    else return v[i];
}

// NT   : output tiles of 32 channels (CO = 32*NT)
// KC   : input channels (compile time: the K loops are fully unrolled so LDS reads are software-pipelined)
// K2S  : spectral MFMA steps (2*K2S >= K2 rows of z2; 0 = no spectral term)
// Software pipeline per wave: the x tile of the NEXT item and the z2 rows of THIS item are in flight in
// registers while the 64..128 MFMAs of this item run, so HBM latency is hidden with 2 waves per SIMD.
// STATS: 0 none | 1 per-channel sum / sum-of-squares of the output (BatchNorm forward statistics) |
//        2 the output is dL/d(act(BN(s))) of a layer: accumulate sum gz and sum gz*shat (gz = out * act'(z)), i.e. the
//          first pass of the BatchNorm backward, fused into the kernel that produces the gradient
template <int NT, int KC, int K2S, int STATS>
__global__ __launch_bounds__((KC >= 128 || STATS == 2) ? 512 : CM_MAX_THREADS) void cell_mix_kernel(CellMixArgs a) {
    constexpr bool SPEC = K2S > 0;
    constexpr int CO = NT * 32;
    constexpr int XS = KC + 1;
    constexpr int NX = KC / 8;                       // float4 loads per lane per tile
    constexpr bool PREF1 = CM_PREF1 && NT <= 2;      // prefetch the second z2 row of a straddling tile too
    typedef typename VecT<NT>::T vecb;
    extern __shared__ float lds[];
    const int K2 = a.K2, Wp = a.Wp;
    float* GWl = lds;                                            // [2*K2S][Wp]
    float* Wl = GWl + (SPEC ? 2 * K2S * Wp : 0);                 // [KC][32][NT]
    // wave index / tile index are wave-uniform: keep them in SGPRs so that every global address below is
    // "uniform base + small per-lane offset + immediate" (saves ~100 VGPRs of 64-bit addresses)
    const int waves = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    float* xl = Wl + KC * CO + wave * 32 * XS;                   // wave-private [32][KC+1]
    int* srow = reinterpret_cast<int*>(Wl + KC * CO + waves * 32 * XS) + wave * 32;

    if (SPEC)
        for (int idx = threadIdx.x; idx < 2 * K2S * Wp; idx += blockDim.x) {
            const int k = idx / Wp, w = idx - k * Wp;
            GWl[idx] = (k < K2) ? a.GW[idx] : 0.f;            // GW is passed TRANSPOSED ([K2][Wp]): coalesced fill
        }
    for (int idx = threadIdx.x; idx < KC * CO; idx += blockDim.x) {
        const int k = idx / CO, n = idx - k * CO;            // n = t*32 + col
        const float v = a.transpose_w ? a.Wm[idx] : a.Wm[(long)n * KC + k];
        Wl[(k * 32 + (n & 31)) * NT + (n >> 5)] = v;
    }
    __syncthreads();

    const int col = lane & 31, half = lane >> 5;
    const long ntiles = (a.ncell + 31) / 32;
    const long tstride = (long)gridDim.x * waves;
    const bool k2_exact = (K2 == 2 * K2S);
    float ssum[NT], ssq[NT], bv[NT];
    XParam bp[NT];
    // STATS == 0 with bnb vectors but no bnb_s: OUTPUT transform -- the tile is stored as act(BN(out)) (eval mode: the
    // running statistics are known, so the activation is materialised once instead of lazily by both consumers)
    const bool oxf = STATS == 0 && a.bnb_s == nullptr && a.bnb.mean != nullptr;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        ssum[t] = ssq[t] = 0.f;
        bv[t] = a.bias ? a.bias[t * 32 + col] : 0.f;
        if (STATS == 2 || oxf) bp[t] = xf_load(a.bnb, t * 32 + col);
    }

    const bool has_xf = a.xf.mean != nullptr;          // only used with the contiguous (non-gather) x layout
    XParam xp4[4];
    if (has_xf) {
#pragma unroll
        for (int k = 0; k < 4; ++k) xp4[k] = xf_load(a.xf, 4 * (lane % (KC / 4)) + k);
    }
    f32x4 xr[NX];
    auto issue_x = [&](long tile) {
        const long cell0 = tile * 32;
        if (a.gather) {
            if (lane < 32) {
                const long c = cell0 + lane;
                srow[lane] = (c < a.ncell) ? (int)pad_to_crop(a.cm, c) : -1;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                const int idx = j * 64 + lane;
                const int row = idx / (KC / 4), c4 = idx - row * (KC / 4);
                const int sr = srow[row];
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (sr >= 0) v = *reinterpret_cast<const f32x4*>(a.x + (long)sr * KC + 4 * c4);
                xr[j] = v;
            }
            __builtin_amdgcn_wave_barrier();
        } else {
            const float* base = a.x + cell0 * KC;                 // uniform
            if (cell0 + 32 <= a.ncell) {
#pragma unroll
                for (int j = 0; j < NX; ++j) xr[j] = *reinterpret_cast<const f32x4*>(base + lane * 4 + j * 256);
            } else {
                const int lim = (int)(a.ncell - cell0) * KC;
#pragma unroll
                for (int j = 0; j < NX; ++j) {
                    const int off = lane * 4 + j * 256;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (off < lim) v = *reinterpret_cast<const f32x4*>(base + off);
                    xr[j] = v;
                }
            }
        }
    };

    long tile = (long)blockIdx.x * waves + wave;
    if (tile < ntiles) issue_x(tile);
    for (; tile < ntiles; tile += tstride) {
        const long cell0 = tile * 32;
        const bool full = cell0 + 32 <= a.ncell;                  // uniform
        // ---- 1. registers -> wave-private transposing LDS tile (lazy BN+GELU of the producer applied here)
        {
            float* d0 = xl + (lane / (KC / 4)) * XS + 4 * (lane % (KC / 4));
            if (has_xf) {
#pragma unroll
                for (int j = 0; j < NX; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k) xr[j][k] = xf_apply(xr[j][k], xp4[k], a.xf.gelu != 0);
            }
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                float* d = d0 + j * (64 / (KC / 4)) * XS;         // 64 float4 = 256/KC rows further down
                d[0] = xr[j][0];
                d[1] = xr[j][1];
                d[2] = xr[j][2];
                d[3] = xr[j][3];
            }
        }
        // ---- 2. z2 rows of this tile -> registers (land while the conv MFMAs run)
        float zr0[SPEC ? K2S : 1][NT], zr1[(SPEC && PREF1) ? K2S : 1][NT];
        int g0 = 0, g1 = 0, myg = 0, myw = 0;
        bool valid = false;
        if (SPEC) {
            const long mycell = cell0 + col;
            valid = mycell < a.ncell;
            g0 = (int)(cell0 / Wp);                               // uniform
            long lastc = cell0 + 31;
            if (lastc >= a.ncell) lastc = a.ncell - 1;
            g1 = (int)(lastc / Wp);                               // uniform
            const int w0 = (int)(cell0 - (long)g0 * Wp);          // uniform
            myw = w0 + col;
            myg = g0;
            while (myw >= Wp) {
                myw -= Wp;
                ++myg;
            }
            const float* zp = a.z2 + (long)g0 * K2 * CO;          // uniform
            const int lo = half * CO + col;
            if (k2_exact) {
#pragma unroll
                for (int s = 0; s < K2S; ++s)
#pragma unroll
                    for (int t = 0; t < NT; ++t) zr0[s][t] = zp[lo + 2 * s * CO + t * 32];
            } else {
#pragma unroll
                for (int s = 0; s < K2S; ++s)
#pragma unroll
                    for (int t = 0; t < NT; ++t) zr0[s][t] = (2 * s + half < K2) ? zp[lo + 2 * s * CO + t * 32] : 0.f;
            }
            if (PREF1 && g1 > g0) {
                const float* zq = zp + (long)K2 * CO;
#pragma unroll
                for (int s = 0; s < K2S; ++s)
#pragma unroll
                    for (int t = 0; t < NT; ++t) zr1[s][t] = (2 * s + half < K2) ? zq[lo + 2 * s * CO + t * 32] : 0.f;
            }
        }
        // ---- 2b. STATS == 2: this tile's pre-BN values (same positions as the outputs) for the BN-backward sums
        float spre[STATS == 2 ? NT : 1][STATS == 2 ? 16 : 1];
        if (STATS == 2) {
            const rsrc_t sr = make_rsrc(a.bnb_s + cell0 * CO, tile_bytes(a.ncell - cell0, 32, CO * 4));
            const int vo = (4 * half * CO + col) * 4;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    spre[t][r] = buf_load_f32(sr, vo + ((8 * (r >> 2) + (r & 3)) * CO + t * 32) * 4, 0);
        }
        // ---- 3. next tile's x loads go out now and stay in flight during the MFMAs below
        if (tile + tstride < ntiles) issue_x(tile + tstride);
        __builtin_amdgcn_wave_barrier();

        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = zero16();
        // ---- 4. channel mixing: A = x tile (LDS, transposed read), B = W (LDS, NT values per read)
        {
            const float* ap = xl + col * XS + half;
            const float* bp = Wl + (half * 32 + col) * NT;
            CM_PRAGMA_UNROLL
            for (int s = 0; s < KC / 2; ++s) {
                const float av = ap[2 * s];
                const vecb b = *reinterpret_cast<const vecb*>(bp + 2 * s * 32 * NT);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = mfma32(av, vget<NT>(b, t), acc[t]);
            }
        }
        // ---- 5. last inverse-DFT stage: A = GW[w(cell)][k] (LDS), B = z2 row (registers)
        if (SPEC) {
            {
                const bool mine = valid && (myg == g0);
                const float* gp = GWl + half * Wp + myw;
#pragma unroll
                for (int s = 0; s < K2S; ++s) {
                    const float av = mine ? gp[2 * s * Wp] : 0.f;
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = mfma32(av, zr0[s][t], acc[t]);
                }
            }
            for (int gg = g0 + 1; gg <= g1; ++gg) {            // tile straddles w-rows (rare beyond one)
                const bool mine = valid && (myg == gg);
                const float* gp = GWl + half * Wp + myw;
                const float* zq = a.z2 + (long)gg * K2 * CO;
                const int lo = half * CO + col;
                const bool pre = PREF1 && (gg == g0 + 1);
                float zb[SPEC ? K2S : 1][NT];
                if (!pre) {
#pragma unroll
                    for (int s = 0; s < K2S; ++s)
#pragma unroll
                        for (int t = 0; t < NT; ++t) zb[s][t] = (2 * s + half < K2) ? zq[lo + 2 * s * CO + t * 32] : 0.f;
                }
#pragma unroll
                for (int s = 0; s < K2S; ++s) {
                    const float av = mine ? gp[2 * s * Wp] : 0.f;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        float b;
                        if (PREF1) b = pre ? zr1[s][t] : zb[s][t];
                        else b = zb[s][t];
                        acc[t] = mfma32(av, b, acc[t]);
                    }
                }
            }
        }
        // ---- 6. epilogue: out[cell0 + row][t*32 + col], row = 8*(r>>2) + 4*half + (r&3)
        {
            float* ob = a.out + cell0 * CO;                       // uniform
            const int lo = 4 * half * CO + col;
            const int rows_left = (int)(a.ncell - cell0);         // only used on the tail tile
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = 8 * (r >> 2) + (r & 3);
                    float v = acc[t][r] + bv[t];
                    if (STATS == 0 && oxf) v = xf_apply(v, bp[t], a.bnb.gelu != 0);
                    if (full || rr + 4 * half < rows_left) {
                        ob[lo + rr * CO + t * 32] = v;
                        if (STATS == 1) {
                            ssum[t] += v;
                            ssq[t] += v * v;
                        } else if (STATS == 2) {
                            const float sh = (spre[t][r] - bp[t].mu) * bp[t].is;
                            const float gz = a.bnb.gelu ? v * gelu_grad_f(sh * bp[t].ga + bp[t].be) : v;
                            ssum[t] += gz;
                            ssq[t] += gz * sh;
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (STATS != 0) {
        float* part = a.stats_part + ((long)blockIdx.x * waves + wave) * 2 * CO;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float s1 = ssum[t] + __shfl_xor(ssum[t], 32, 64);
            const float s2 = ssq[t] + __shfl_xor(ssq[t], 32, 64);
            if (half == 0) {
                part[t * 32 + col] = s1;
                part[CO + t * 32 + col] = s2;
            }
        }
    }
}

static int cell_mix_k2s(int K2, bool spec) {
    if (!spec) return 0;
    if (K2 <= 16) return 8;
    if (K2 <= 32) return 16;
    if (K2 <= 40) return 20;            // Galerkin cylinder regressor: fourier_modes_y = 20
    return -1;
}

static size_t cell_mix_lds(int KC, int CO, int K2, int Wp, bool spec, int waves) {
    const int k2s = cell_mix_k2s(K2, spec);
    return ((size_t)(spec ? 2 * k2s * Wp : 0) + (size_t)KC * CO + (size_t)waves * 32 * (KC + 1) + (size_t)waves * 32) * 4;
}

static int cell_mix_waves(int KC, int CO, int K2, int Wp, bool spec, bool bnb = false) {
    if (cell_mix_k2s(K2, spec) < 0) return 0;
    const int wmax = ((KC >= 128 || bnb) ? 512 : CM_MAX_THREADS) / 64;
    static const int cand[] = {16, 14, 12, 10, 8, 7, 6, 4, 2, 1};   // KC = 128 tiles are 16.5 KB per wave: 7 waves fit next to a 128 x 64
                                                                   // weight tile where the power-of-two list stopped at 4 (-10 %); five
                                                                   // waves (128 x 128 weights) measured 4 % slower than four
    for (int w : cand)
        if (w <= wmax && cell_mix_lds(KC, CO, K2, Wp, spec, w) <= 160 * 1024) return w;
    return 0;
}

template <int NT, int KC, int K2S, int STATS>
static int launch_cell_mix(const CellMixArgs& a, int waves, int grid, hipStream_t st) {
    const size_t lds = cell_mix_lds(a.KC, a.CO, a.K2, a.Wp, K2S > 0, waves);
    (void)hipFuncSetAttribute((const void*)cell_mix_kernel<NT, KC, K2S, STATS>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((cell_mix_kernel<NT, KC, K2S, STATS>), dim3(grid), dim3(waves * 64), lds, st, a);
    RPB_CHECK_LAUNCH("cell_mix");
}

// number of [2][CO] stat partial rows cell_mix writes for this problem (== grid * waves)
extern "C" long rpb_cell_mix_stat_rows(long ncell, int KC, int CO, int K2, int Wp, int has_spec, int bn_bwd_stats) {
    if (rpb_cmx_supported(ncell, KC, CO, K2, Wp, has_spec != 0, false)) return rpb_cmx_stat_rows(ncell, Wp, bn_bwd_stats ? 2 : 1);
    if (rpb_cmx128_supported(ncell, KC, CO, K2, Wp, has_spec != 0, false)) return rpb_cmx128_stat_rows(ncell, Wp);
    const int waves = cell_mix_waves(KC, CO, K2, Wp, has_spec != 0, bn_bwd_stats != 0);
    if (waves == 0) return -1;
    const long ntiles = (ncell + 31) / 32;
    long grid = rpb_num_cus();
    const long need = (ntiles + waves - 1) / waves;
    if (grid > need) grid = need;
    return grid * waves;
}

extern "C" int rpb_cell_mix_writes_gz(long ncell, int KC, int CO, int K2, int Wp, int has_spec, int gather) {
    return (rpb_cmx_supported(ncell, KC, CO, K2, Wp, has_spec != 0, gather != 0) || rpb_cmx128_supported(ncell, KC, CO, K2, Wp, has_spec != 0, gather != 0)) ? 1 : 0;
}

extern "C" int rpb_cell_mix(const float* x, const float* Wm, const float* bias, const float* z2, const float* GW,
                            float* out, float* stats_part, long ncell, int KC, int CO, int K2, int Wp, int transpose_w,
                            int gather, int T, int H, int W, int Tp, int Hp, int Wp_pad, const float* xf_mean,
                            const float* xf_invstd, const float* xf_gamma, const float* xf_beta, int xf_gelu,
                            const float* bnb_s, const float* bnb_mean, const float* bnb_invstd, const float* bnb_gamma,
                            const float* bnb_beta, int bnb_gelu, void* stream) {
    RPB_REQUIRE(x && Wm && out, "cell_mix: null pointer");
    if (xf_mean) RPB_REQUIRE(xf_invstd && xf_gamma && xf_beta && !gather, "cell_mix: bad input-transform arguments");
    RPB_REQUIRE(ncell > 0 && ncell < (1L << 31), "cell_mix: ncell=%ld out of range", ncell);
    RPB_REQUIRE(KC == 32 || KC == 64 || KC == 128, "cell_mix: KC=%d must be 32, 64 or 128", KC);
    RPB_REQUIRE(CO == 32 || CO == 64 || CO == 128, "cell_mix: CO=%d must be 32, 64 or 128", CO);
    const bool spec = z2 != nullptr;
    if (spec) RPB_REQUIRE(GW && K2 > 0 && K2 <= 40 && Wp > 0 && ncell % Wp == 0, "cell_mix: bad spectral arguments (K2=%d)", K2);
    if (bnb_s) RPB_REQUIRE(stats_part && bnb_mean && bnb_invstd && bnb_gamma && bnb_beta, "cell_mix: BN-backward statistics need stats_part and all four vectors");
    if (!bnb_s && bnb_mean) RPB_REQUIRE(!stats_part && bnb_invstd && bnb_gamma && bnb_beta, "cell_mix: the output transform needs all four vectors and no statistics");
    const bool c128 = rpb_cmx128_supported(ncell, KC, CO, K2, Wp, spec, gather != 0);
    if (c128 || rpb_cmx_supported(ncell, KC, CO, K2, Wp, spec, gather != 0)) {      // C = 64 / 128 spectral instances: bf16 matrix pipe, split operands
        CmxArgs c{};
        c.c128 = c128;
        c.x = x; c.Wm = Wm; c.bias = bias; c.z2 = z2; c.GW = GW; c.out = out; c.stats_part = stats_part;
        c.ncell = ncell; c.K2 = K2; c.Wp = Wp; c.transpose_w = transpose_w;
        c.xf = XForm{xf_mean, xf_invstd, xf_gamma, xf_beta, xf_gelu};
        c.bnb_s = bnb_s;
        c.bnb = XForm{bnb_mean, bnb_invstd, bnb_gamma, bnb_beta, bnb_gelu != 0};
        c.write_gz = bnb_s != nullptr && bnb_gelu == 2;
        c.bf16_io = 0;
        c.feat_w = 0;
        c.FWt = nullptr; c.y1out = nullptr; c.K2f = 0; c.gw_planes = nullptr;
        return rpb_cmx_launch(c, stats_part == nullptr ? 0 : (bnb_s ? 2 : 1), (hipStream_t)stream);
    }
    // the fc1 data gradient of the width-128 head (configs/fsi/fno.yaml): gu [ncrop][128] x fc1.weight [128][128] gathered into the padded
    // layout, no statistics -- csrc/rpb_pjh.hip's MODE 2 on the bf16 matrix pipe (round 6b; RPB_GATHER_128_PJH=0: the fp32-pipe kernel below)
    // (with bnb_* and stats_part and no GELU in that layer: the BatchNorm-backward sums ride in the same launch -- MODE 3)
    if (gather && !spec && KC == 128 && CO == 128 && transpose_w && !bias && !xf_mean && ((!stats_part && !bnb_s && !bnb_mean) || (stats_part && bnb_s && bnb_gelu == 0)) &&
        ncell % ((long)Tp * Hp * Wp_pad) == 0 && (long)Wp_pad * 512 < (1l << 31) &&
        !(getenv("RPB_GATHER_128_PJH") && atoi(getenv("RPB_GATHER_128_PJH")) == 0))
        return rpb_pjh_dgrad128_launch(x, Wm, out, (int)(ncell / ((long)Tp * Hp * Wp_pad)), T, H, W, Tp, Hp, Wp_pad, (hipStream_t)stream, bnb_s,
                                       bnb_mean, bnb_invstd, stats_part, stats_part ? rpb_cell_mix_stat_rows(ncell, KC, CO, 0, 1, 0, 1) : 0);
    RPB_REQUIRE(bnb_gelu != 2, "cell_mix: this shape runs on the fp32 kernel, which does not store gz (ask rpb_cell_mix_writes_gz)");
    const int waves = cell_mix_waves(KC, CO, K2, Wp, spec, bnb_s != nullptr);
    RPB_REQUIRE(waves > 0, "cell_mix: tiles do not fit LDS (KC=%d CO=%d K2=%d Wp=%d)", KC, CO, K2, Wp);
    CellMixArgs a;
    a.x = x; a.Wm = Wm; a.bias = bias; a.z2 = z2; a.GW = GW; a.out = out; a.stats_part = stats_part;
    a.ncell = ncell; a.KC = KC; a.CO = CO; a.K2 = spec ? K2 : 0; a.Wp = spec ? Wp : 1;
    a.transpose_w = transpose_w; a.gather = gather;
    a.cm = CropMap{T, H, W, Tp, Hp, Wp_pad};
    a.xf = XForm{xf_mean, xf_invstd, xf_gamma, xf_beta, xf_gelu};
    a.bnb_s = bnb_s;
    a.bnb = XForm{bnb_mean, bnb_invstd, bnb_gamma, bnb_beta, bnb_gelu};
    if (bnb_s) RPB_REQUIRE(stats_part && bnb_mean && bnb_invstd && bnb_gamma && bnb_beta, "cell_mix: BN-backward statistics need stats_part and all four vectors");
    if (!bnb_s && bnb_mean) RPB_REQUIRE(!stats_part && bnb_invstd && bnb_gamma && bnb_beta, "cell_mix: the output transform needs all four vectors and no statistics");
    const int grid = (int)(rpb_cell_mix_stat_rows(ncell, KC, CO, K2, Wp, spec, bnb_s != nullptr) / waves);
    hipStream_t st = (hipStream_t)stream;
    const int stats = stats_part == nullptr ? 0 : (bnb_s ? 2 : 1);
    const int NT = CO / 32;
    const int k2s = cell_mix_k2s(K2, spec);
#define RPB_CM(NT_, KC_, K2S_, ST_) \
    if (NT == NT_ && KC == KC_ && k2s == K2S_ && stats == ST_) return launch_cell_mix<NT_, KC_, K2S_, ST_>(a, waves, grid, st);
#define RPB_CM3(NT_, KC_, K2S_) RPB_CM(NT_, KC_, K2S_, 0) RPB_CM(NT_, KC_, K2S_, 1) RPB_CM(NT_, KC_, K2S_, 2)
    // square channel mixing (1x1x1 conv fwd / dgrad), with and without the spectral term
    RPB_CM3(1, 32, 8) RPB_CM3(1, 32, 16) RPB_CM(1, 32, 0, 0)
    RPB_CM3(2, 64, 8) RPB_CM3(2, 64, 16) RPB_CM(2, 64, 0, 0)
    RPB_CM3(4, 128, 8) RPB_CM3(4, 128, 16) RPB_CM(4, 128, 0, 0)
    RPB_CM3(1, 32, 20) RPB_CM3(4, 128, 20)
    // fc1 dgrad (128 hidden -> C), gather into the padded layout; STATS 2 = BN-backward sums of the last layer
    RPB_CM(1, 128, 0, 0) RPB_CM(2, 128, 0, 0) RPB_CM(1, 128, 0, 2) RPB_CM(2, 128, 0, 2) RPB_CM(4, 128, 0, 2)
#undef RPB_CM3
#undef RPB_CM
    RPB_FAIL(RPB_ERR_UNSUPPORTED, "cell_mix: unsupported configuration KC=%d CO=%d K2=%d stats=%d", KC, CO, K2, (int)stats);
}

// eval / rollout with bf16 activation storage (BASELINE.json configs[4]): x and out are bf16 [ncell][64]; the statistics are the
// running ones, so BatchNorm(+GELU) is applied to the tile before it is rounded and stored (oxf_* = that layer's vectors)
extern "C" int rpb_cell_mix_bf16(const void* x_bf16, const float* Wm, const float* bias, const void* z2, const float* GW,
                                 void* out_bf16, long ncell, int C, int K2, int Wp, const float* oxf_mean,
                                 const float* oxf_invstd, const float* oxf_gamma, const float* oxf_beta, int oxf_gelu,
                                 int spectra_bf16, void* stream) {
    RPB_REQUIRE(x_bf16 && Wm && z2 && GW && out_bf16, "cell_mix_bf16: null pointer");
    RPB_REQUIRE(rpb_cmx_supported(ncell, C, C, K2, Wp, true, false), "cell_mix_bf16: needs C = 64, K2 <= 32, Wp >= 32 (C=%d K2=%d Wp=%d)", C, K2, Wp);
    if (oxf_mean) RPB_REQUIRE(oxf_invstd && oxf_gamma && oxf_beta, "cell_mix_bf16: the output transform needs all four vectors");
    CmxArgs c{};
    c.x = (const float*)x_bf16; c.Wm = Wm; c.bias = bias; c.z2 = (const float*)z2; c.GW = GW; c.out = (float*)out_bf16; c.stats_part = nullptr;
    c.ncell = ncell; c.K2 = K2; c.Wp = Wp; c.transpose_w = 0;
    c.xf = XForm{nullptr, nullptr, nullptr, nullptr, 0};
    c.bnb_s = nullptr;
    c.bnb = XForm{oxf_mean, oxf_invstd, oxf_gamma, oxf_beta, oxf_gelu != 0};
    c.write_gz = 0;
    c.bf16_io = 1;
    c.spec_bf16 = spectra_bf16 != 0;
    c.feat_w = 0;
    c.FWt = nullptr; c.y1out = nullptr; c.K2f = 0; c.gw_planes = nullptr;
    return rpb_cmx_launch(c, 0, (hipStream_t)stream);
}

// layer 0 of FNO3d on the feature fields (csrc/rpb_feat.hip): out = GW z2 + Wcomp phi + bias with phi [ncell][FW] (rpb_lift_feat)
// and Wcomp = convs.0.weight [fc0.weight | fc0.bias] [64][FW]; stats_part: BatchNorm forward sums (training) or NULL with the
// oxf_* vectors (eval: store act(BN(out)))
extern "C" int rpb_cell_mix_feat(const float* phi, const float* Wcomp, const float* bias, const float* z2, const float* GW,
                                 float* out, float* stats_part, long ncell, int FW, int K2, int Wp, const float* oxf_mean,
                                 const float* oxf_invstd, const float* oxf_gamma, const float* oxf_beta, int oxf_gelu,
                                 void* stream) {
    RPB_REQUIRE(phi && Wcomp && z2 && GW && out && (FW == 8 || FW == 32), "cell_mix_feat: bad arguments (FW=%d)", FW);
    RPB_REQUIRE(rpb_cmx_supported(ncell, 64, 64, K2, Wp, true, false), "cell_mix_feat: needs K2 <= 32, Wp >= 32 (K2=%d Wp=%d)", K2, Wp);
    if (oxf_mean) RPB_REQUIRE(!stats_part && oxf_invstd && oxf_gamma && oxf_beta, "cell_mix_feat: output transform xor statistics");
    CmxArgs c{};
    c.x = phi; c.Wm = Wcomp; c.bias = bias; c.z2 = z2; c.GW = GW; c.out = out; c.stats_part = stats_part;
    c.ncell = ncell; c.K2 = K2; c.Wp = Wp; c.transpose_w = 0;
    c.xf = XForm{nullptr, nullptr, nullptr, nullptr, 0};
    c.bnb_s = nullptr;
    c.bnb = XForm{oxf_mean, oxf_invstd, oxf_gamma, oxf_beta, oxf_gelu != 0};
    c.write_gz = 0;
    c.bf16_io = 0;
    c.feat_w = FW;
    c.FWt = nullptr; c.y1out = nullptr; c.K2f = 0; c.gw_planes = nullptr;
    return rpb_cmx_launch(c, stats_part ? 1 : 0, (hipStream_t)stream);
}

// Eval cell_mix (output transform = this layer's BatchNorm (+GELU)) with the NEXT layer's forward W stage fused in (csrc/rpb_cmx.hip):
// out [ncell][64] as rpb_cell_mix / rpb_cell_mix_feat (feat_w > 0: x is the feature tensor, Wm the composite weight), and
// y1 [ncell / Wp][K2f][64] = sum_w FWt[w][k] out[line, w][c] -- what rpb_axis_gemm(out, y1, FWt, ...) would compute from a second read.
extern "C" int rpb_cell_mix_eval_dft(const float* x, const float* Wm, const float* bias, const float* z2, const float* GW, float* out,
                                     long ncell, int K2, int Wp, int feat_w, const float* oxf_mean, const float* oxf_invstd,
                                     const float* oxf_gamma, const float* oxf_beta, int oxf_gelu, const float* FWt, int K2f, float* y1,
                                     void* scratch, void* stream) {
    RPB_REQUIRE(x && Wm && z2 && GW && out && oxf_mean && oxf_invstd && oxf_gamma && oxf_beta && FWt && y1 && scratch, "cell_mix_eval_dft: null pointer");
    RPB_REQUIRE(feat_w == 0 || feat_w == 8 || feat_w == 32, "cell_mix_eval_dft: feat_w=%d", feat_w);
    RPB_REQUIRE(rpb_cmx_supported(ncell, 64, 64, K2, Wp, true, false) && rpb_cmx_dft_supported(Wp, K2f),
                "cell_mix_eval_dft: unsupported sizes (K2=%d Wp=%d K2f=%d)", K2, Wp, K2f);
    CmxArgs c{};
    c.x = x; c.Wm = Wm; c.bias = bias; c.z2 = z2; c.GW = GW; c.out = out; c.stats_part = nullptr;
    c.ncell = ncell; c.K2 = K2; c.Wp = Wp; c.transpose_w = 0;
    c.xf = XForm{nullptr, nullptr, nullptr, nullptr, 0};
    c.bnb_s = nullptr;
    c.bnb = XForm{oxf_mean, oxf_invstd, oxf_gamma, oxf_beta, oxf_gelu != 0};
    c.write_gz = 0;
    c.bf16_io = 0;
    c.feat_w = feat_w;
    c.FWt = FWt; c.y1out = y1; c.K2f = K2f; c.gw_planes = scratch;
    return rpb_cmx_launch(c, 0, (hipStream_t)stream);
}
// The same launch on the opt-in "f16x2" arithmetic (csrc/rpb_cmx.hip, template parameter H2): operands as two fp16 planes (round to nearest
// even: one fp32 unit in the last place), three products per fp32 product, dropped term <= 2^-22 |a b| -- below the fp32 grade of the
// default path, which is why it is a separate, explicitly named entry point.  spec_exp: floor(log2(Tp * Hp * Wp)) - 1 (the exact
// power-of-two rescaling of GWt / z2 that keeps both inside fp16's range).
extern "C" int rpb_cell_mix_eval_dft_f16x2(const float* x, const float* Wm, const float* bias, const float* z2, const float* GW, float* out,
                                           long ncell, int K2, int Wp, int feat_w, const float* oxf_mean, const float* oxf_invstd,
                                           const float* oxf_gamma, const float* oxf_beta, int oxf_gelu, const float* FWt, int K2f, float* y1,
                                           void* scratch, int spec_exp, void* stream) {
    RPB_REQUIRE(x && Wm && z2 && GW && out && oxf_mean && oxf_invstd && oxf_gamma && oxf_beta && FWt && y1 && scratch, "cell_mix_eval_dft_f16x2: null pointer");
    RPB_REQUIRE(feat_w == 0 || feat_w == 8 || feat_w == 32, "cell_mix_eval_dft_f16x2: feat_w=%d", feat_w);
    RPB_REQUIRE(spec_exp >= 0 && spec_exp <= 40, "cell_mix_eval_dft_f16x2: spec_exp=%d", spec_exp);
    RPB_REQUIRE(rpb_cmx_supported(ncell, 64, 64, K2, Wp, true, false) && rpb_cmx_dft_supported(Wp, K2f),
                "cell_mix_eval_dft_f16x2: unsupported sizes (K2=%d Wp=%d K2f=%d)", K2, Wp, K2f);
    CmxArgs c{};
    c.x = x; c.Wm = Wm; c.bias = bias; c.z2 = z2; c.GW = GW; c.out = out; c.stats_part = nullptr;
    c.ncell = ncell; c.K2 = K2; c.Wp = Wp; c.transpose_w = 0;
    c.xf = XForm{nullptr, nullptr, nullptr, nullptr, 0};
    c.bnb_s = nullptr;
    c.bnb = XForm{oxf_mean, oxf_invstd, oxf_gamma, oxf_beta, oxf_gelu != 0};
    c.feat_w = feat_w;
    c.FWt = FWt; c.y1out = y1; c.K2f = K2f; c.gw_planes = scratch;
    c.h2 = 1; c.spec_exp = spec_exp;
    return rpb_cmx_launch(c, 0, (hipStream_t)stream);
}
// ... and the crop-only last layer (fp32 storage)
extern "C" int rpb_cell_mix_eval_crop_f16x2(const float* x, const float* Wm, const float* bias, const float* z2, const float* GW, float* out,
                                            int B, int T, int H, int W, int Tp, int Hp, int Wp, int K2, const float* oxf_mean,
                                            const float* oxf_invstd, const float* oxf_gamma, const float* oxf_beta, int oxf_gelu,
                                            int spec_exp, void* stream) {
    RPB_REQUIRE(x && Wm && z2 && GW && out && oxf_mean && oxf_invstd && oxf_gamma && oxf_beta, "cell_mix_eval_crop_f16x2: null pointer");
    RPB_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0 && T <= Tp && H <= Hp && W <= Wp, "cell_mix_eval_crop_f16x2: bad crop (%d %d %d of %d %d %d)", T, H, W, Tp, Hp, Wp);
    RPB_REQUIRE(spec_exp >= 0 && spec_exp <= 40, "cell_mix_eval_crop_f16x2: spec_exp=%d", spec_exp);
    const long ncell = (long)B * Tp * Hp * Wp;
    RPB_REQUIRE((long)B * T * H < (1l << 31), "cell_mix_eval_crop_f16x2: too many lines");
    RPB_REQUIRE(rpb_cmx_supported(ncell, 64, 64, K2, Wp, true, false), "cell_mix_eval_crop_f16x2: needs C = 64, K2 <= 32, Wp >= 32 (K2=%d Wp=%d)", K2, Wp);
    CmxArgs c{};
    c.x = x; c.Wm = Wm; c.bias = bias; c.z2 = z2; c.GW = GW; c.out = out; c.stats_part = nullptr;
    c.ncell = ncell; c.K2 = K2; c.Wp = Wp; c.transpose_w = 0;
    c.xf = XForm{nullptr, nullptr, nullptr, nullptr, 0};
    c.bnb_s = nullptr;
    c.bnb = XForm{oxf_mean, oxf_invstd, oxf_gamma, oxf_beta, oxf_gelu != 0};
    c.crop_T = T; c.crop_H = H; c.crop_W = W; c.Tp = Tp; c.Hp = Hp;
    c.h2 = 1; c.spec_exp = spec_exp;
    return rpb_cmx_launch(c, 0, (hipStream_t)stream);
}
// Eval cell_mix of the LAST Fourier layer (fno.py:117-121: BatchNorm, no GELU, then x[..., :-6, :-6, :-6, :] -> fc1): only the
// B * T * H lines of the crop are produced, each up to the tile that holds cell W - 1; the pad cells of `out` keep whatever they held
// (nothing reads them: rpb_proj_fwd walks the crop).  bf16_io: x / out are bf16 [ncell][64].
extern "C" int rpb_cell_mix_eval_crop(const void* x, const float* Wm, const float* bias, const void* z2, const float* GW, void* out,
                                      int B, int T, int H, int W, int Tp, int Hp, int Wp, int K2, const float* oxf_mean,
                                      const float* oxf_invstd, const float* oxf_gamma, const float* oxf_beta, int oxf_gelu, int bf16_io,
                                      int spectra_bf16, void* stream) {
    RPB_REQUIRE(x && Wm && z2 && GW && out && oxf_mean && oxf_invstd && oxf_gamma && oxf_beta, "cell_mix_eval_crop: null pointer");
    RPB_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0 && T <= Tp && H <= Hp && W <= Wp, "cell_mix_eval_crop: bad crop (%d %d %d of %d %d %d)", T, H, W, Tp, Hp, Wp);
    const long ncell = (long)B * Tp * Hp * Wp;
    RPB_REQUIRE((long)B * T * H < (1l << 31), "cell_mix_eval_crop: too many lines");
    RPB_REQUIRE(rpb_cmx_supported(ncell, 64, 64, K2, Wp, true, false), "cell_mix_eval_crop: needs C = 64, K2 <= 32, Wp >= 32 (K2=%d Wp=%d)", K2, Wp);
    CmxArgs c{};
    c.x = (const float*)x; c.Wm = Wm; c.bias = bias; c.z2 = (const float*)z2; c.GW = GW; c.out = (float*)out; c.stats_part = nullptr;
    c.ncell = ncell; c.K2 = K2; c.Wp = Wp; c.transpose_w = 0;
    c.xf = XForm{nullptr, nullptr, nullptr, nullptr, 0};
    c.bnb_s = nullptr;
    c.bnb = XForm{oxf_mean, oxf_invstd, oxf_gamma, oxf_beta, oxf_gelu != 0};
    c.bf16_io = bf16_io != 0;
    c.spec_bf16 = bf16_io != 0 && spectra_bf16 != 0;
    c.crop_T = T; c.crop_H = H; c.crop_W = W; c.Tp = Tp; c.Hp = Hp;
    return rpb_cmx_launch(c, 0, (hipStream_t)stream);
}
// ... at width 128 (configs/fsi/fno.yaml, the Galerkin regressor; fp32 storage): the C = 128 instance of csrc/rpb_cmx.hip over the crop's lines
extern "C" int rpb_cell_mix_eval_crop_c128(const float* x, const float* Wm, const float* bias, const float* z2, const float* GW, float* out,
                                           int B, int T, int H, int W, int Tp, int Hp, int Wp, int K2, const float* oxf_mean,
                                           const float* oxf_invstd, const float* oxf_gamma, const float* oxf_beta, int oxf_gelu, void* stream) {
    RPB_REQUIRE(x && Wm && z2 && GW && out && oxf_mean && oxf_invstd && oxf_gamma && oxf_beta, "cell_mix_eval_crop_c128: null pointer");
    RPB_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0 && T <= Tp && H <= Hp && W <= Wp, "cell_mix_eval_crop_c128: bad crop (%d %d %d of %d %d %d)", T, H, W, Tp, Hp, Wp);
    const long ncell = (long)B * Tp * Hp * Wp;
    RPB_REQUIRE((long)B * T * H < (1l << 31), "cell_mix_eval_crop_c128: too many lines");
    RPB_REQUIRE(rpb_cmx128_supported(ncell, 128, 128, K2, Wp, true, false), "cell_mix_eval_crop_c128: needs K2 <= 32, Wp >= 32 (K2=%d Wp=%d)", K2, Wp);
    CmxArgs c{};
    c.c128 = 1;
    c.x = x; c.Wm = Wm; c.bias = bias; c.z2 = z2; c.GW = GW; c.out = out; c.stats_part = nullptr;
    c.ncell = ncell; c.K2 = K2; c.Wp = Wp; c.transpose_w = 0;
    c.xf = XForm{nullptr, nullptr, nullptr, nullptr, 0};
    c.bnb_s = nullptr;
    c.bnb = XForm{oxf_mean, oxf_invstd, oxf_gamma, oxf_beta, oxf_gelu != 0};
    c.crop_T = T; c.crop_H = H; c.crop_W = W; c.Tp = Tp; c.Hp = Hp;
    return rpb_cmx_launch(c, 0, (hipStream_t)stream);
}
extern "C" int rpb_cell_mix_eval_crop_c128_supported(long ncell, int K2, int Wp) { return rpb_cmx128_supported(ncell, 128, 128, K2, Wp, true, false) ? 1 : 0; }
// The same on bf16-stored activations (x, out bf16 [ncell][64]; y1 fp32): the stage sees the ROUNDED activations, i.e. exactly what
// rpb_axis_gemm_bf16in would read back from `out`.
extern "C" int rpb_cell_mix_eval_dft_bf16(const void* x_bf16, const float* Wm, const float* bias, const void* z2, const float* GW,
                                          void* out_bf16, long ncell, int K2, int Wp, const float* oxf_mean, const float* oxf_invstd,
                                          const float* oxf_gamma, const float* oxf_beta, int oxf_gelu, const float* FWt, int K2f,
                                          void* y1, void* scratch, int spectra_bf16, void* stream) {
    RPB_REQUIRE(x_bf16 && Wm && z2 && GW && out_bf16 && oxf_mean && oxf_invstd && oxf_gamma && oxf_beta && FWt && y1 && scratch,
                "cell_mix_eval_dft_bf16: null pointer");
    RPB_REQUIRE(rpb_cmx_supported(ncell, 64, 64, K2, Wp, true, false) && rpb_cmx_dft_supported(Wp, K2f),
                "cell_mix_eval_dft_bf16: unsupported sizes (K2=%d Wp=%d K2f=%d)", K2, Wp, K2f);
    CmxArgs c{};
    c.x = (const float*)x_bf16; c.Wm = Wm; c.bias = bias; c.z2 = (const float*)z2; c.GW = GW; c.out = (float*)out_bf16; c.stats_part = nullptr;
    c.ncell = ncell; c.K2 = K2; c.Wp = Wp; c.transpose_w = 0;
    c.xf = XForm{nullptr, nullptr, nullptr, nullptr, 0};
    c.bnb_s = nullptr;
    c.bnb = XForm{oxf_mean, oxf_invstd, oxf_gamma, oxf_beta, oxf_gelu != 0};
    c.bf16_io = 1;
    c.spec_bf16 = spectra_bf16 != 0;
    c.FWt = FWt; c.y1out = (float*)y1; c.K2f = K2f; c.gw_planes = scratch;
    return rpb_cmx_launch(c, 0, (hipStream_t)stream);
}
extern "C" int rpb_cell_mix_eval_dft_supported(long ncell, int K2, int Wp, int K2f) {
    return rpb_cmx_supported(ncell, 64, 64, K2, Wp, true, false) && rpb_cmx_dft_supported(Wp, K2f);
}

// Backward cell_mix of a Fourier layer with the layer's Conv3d weight gradient riding along (wave pairs of csrc/rpb_cmx.hip, WG; C = 64;
// the one-wave-per-SIMD organisation of round 4 lost every A/B since and is archived as tools/archive/rpb_cmw.hip):
//   out  = gs Wc + FW^T z2, stored as gz = out * act'(BN(s_prev)) when gelu == 2        (as rpb_cell_mix with bnb_*)
//   stats_part[slot][2][64] = partial (sum gz, sum gz * shat) of the layer below
//   wg_part[slot][64][64]   = partial dWc[co][ci] = sum_cells gs[cell][co] * act(BN(s_prev))[cell][ci]
// with slot < rpb_cell_mix_wgrad_slots(ncell, Wp).  Wc is convs.l.weight [co][ci]; FW the adjoint stage matrix [K2][Wp].
extern "C" long rpb_cell_mix_wgrad_slots(long ncell, int Wp) { return Wp > 0 ? rpb_cmx_wg_slots(ncell, Wp) : -1; }
extern "C" int rpb_cell_mix_wgrad_supported(long ncell, int K2, int Wp) {
    return rpb_cmx_supported(ncell, 64, 64, K2, Wp, true, false) && rpb_bwr_supported(64, Wp, K2, 0) ? 1 : 0;
}
extern "C" int rpb_cell_mix_wgrad(const float* gs, const float* Wc, const float* z2, const float* FW, float* out, float* stats_part,
                                  float* wg_part, long ncell, int K2, int Wp, const float* s_prev, const float* mean, const float* invstd,
                                  const float* gamma, const float* beta, int gelu, void* stream) {
    RPB_REQUIRE(gs && Wc && z2 && FW && out && stats_part && wg_part && s_prev && mean && invstd && gamma && beta, "cell_mix_wgrad: null pointer");
    RPB_REQUIRE(ncell > 0 && ncell < (1L << 31) && Wp > 0 && ncell % Wp == 0, "cell_mix_wgrad: bad sizes");
    RPB_REQUIRE(rpb_cmx_supported(ncell, 64, 64, K2, Wp, true, false), "cell_mix_wgrad: needs C = 64, K2 <= 32, Wp >= 32 (K2=%d Wp=%d)", K2, Wp);
    CmxArgs c{};
    c.x = gs; c.Wm = Wc; c.bias = nullptr; c.z2 = z2; c.GW = FW; c.out = out; c.stats_part = stats_part; c.wg_part = wg_part;
    c.ncell = ncell; c.K2 = K2; c.Wp = Wp; c.transpose_w = 1;
    c.xf = XForm{nullptr, nullptr, nullptr, nullptr, 0};
    c.bnb_s = s_prev;
    c.bnb = XForm{mean, invstd, gamma, beta, gelu != 0};
    c.write_gz = gelu == 2;
    return rpb_cmx_wg_launch(c, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------- cell_wgrad
// dW[o][i] = sum_cell gs[cell][o] * x[xrow(cell)][i] ; db[o] = sum_cell gs[cell][o].
// Both operands are read straight from HBM in MFMA layout (lane = channel, 128 B per half-wave).
// Each wave accumulates NTO x NTI 32x32 tiles and writes one partial; rpb_reduce_partials sums them.
struct WgradArgs {
    const float* gs;   // [ncell][CO]
    const float* x;    // [rows][CI]
    float* part;       // [nslots][CO*CI + CO]
    long ncell;
    int CO, CI;
    int crop;          // 1: x row = crop_to_pad(cell)
    CropMap cm;
    XForm xf;          // lazy BatchNorm(+GELU) applied to x before the MFMA
};

// Channel <-> MFMA index mapping: a lane loads NTO (resp. NTI) CONTIGUOUS channels of one cell with one vector
// load (lane col -> channels col*NT .. col*NT+NT-1, i.e. a full 128/256/512 B line per half-wave), so MFMA tile `t`
// holds the channels == t (mod NT): row i of o-tile `to` is channel i*NTO + to, column j of i-tile `ti` is channel
// j*NTI + ti.  Half tiles (8 MFMA steps = 16 cells) are double-buffered in registers: the loads of the next half
// are in flight while the MFMAs of the current half run.
template <int CO, int CI>
__global__ __launch_bounds__(512) void cell_wgrad_kernel(WgradArgs a) {
    constexpr int NTI = CI / 32;
    constexpr int NTO = (CO / 32) * NTI > 4 ? ((CO / 32) * NTI > 8 ? (CO / 32) / 4 : (CO / 32) / 2) : CO / 32;
    static_assert(NTO >= 1 && NTO * NTI <= 4, "accumulator budget");
    typedef typename VecT<NTO>::T veco;
    typedef typename VecT<NTI>::T veci;
    const int waves = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int col = lane & 31, half = lane >> 5;
    constexpr int nroles = CO / (32 * NTO);
    const long sid = (long)blockIdx.x * waves + wave;           // uniform
    const int role = (int)(sid % nroles);
    const long tslot = sid / nroles;
    const long ntslots = ((long)gridDim.x * waves) / nroles;
    const long ntiles = (a.ncell + 31) / 32;
    const bool rowfast = !a.crop || (a.cm.W % 32 == 0);          // a tile never leaves its (b,t,h) row

    f32x16 acc[NTO][NTI];
#pragma unroll
    for (int o = 0; o < NTO; ++o)
#pragma unroll
        for (int i = 0; i < NTI; ++i) acc[o][i] = zero16();
    float bsum[NTO];
#pragma unroll
    for (int o = 0; o < NTO; ++o) bsum[o] = 0.f;

    veco ga[8], gb[8];
    veci xa[8], xb[8];
    const veco zo = {};
    const veci zi = {};
    const bool has_xf = a.xf.mean != nullptr;
    XParam xp[NTI];
    if (has_xf) {
#pragma unroll
        for (int i = 0; i < NTI; ++i) xp[i] = xf_load(a.xf, col * NTI + i);
    }

    auto load_half = [&](long tile, int h, veco (&gv)[8], veci (&xv)[8]) {
        const long cell0 = tile * 32 + h * 16;                   // uniform
        const float* gp = a.gs + cell0 * CO + role * 32 * NTO;   // uniform
        const int go = half * CO + col * NTO;                    // per-lane, tile-invariant
        if (rowfast) {
            const long xrow0 = a.crop ? crop_to_pad(a.cm, tile * 32) + h * 16 : cell0;     // uniform
            const float* xp = a.x + xrow0 * CI;                  // uniform
            const int xo = half * CI + col * NTI;
            if (cell0 + 16 <= a.ncell) {
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    gv[s] = *reinterpret_cast<const veco*>(gp + go + 2 * s * CO);
                    xv[s] = *reinterpret_cast<const veci*>(xp + xo + 2 * s * CI);
                }
            } else {
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const bool ok = cell0 + 2 * s + half < a.ncell;
                    gv[s] = ok ? *reinterpret_cast<const veco*>(gp + go + 2 * s * CO) : zo;
                    xv[s] = ok ? *reinterpret_cast<const veci*>(xp + xo + 2 * s * CI) : zi;
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const long c = cell0 + 2 * s + half;
                const bool ok = c < a.ncell;
                gv[s] = ok ? *reinterpret_cast<const veco*>(gp + go + 2 * s * CO) : zo;
                xv[s] = ok ? *reinterpret_cast<const veci*>(a.x + crop_to_pad(a.cm, c) * CI + col * NTI) : zi;
            }
        }
    };
    // NOTE: cells past ncell load gs == 0, so whatever the transform makes of their x never reaches dW
    auto compute_half = [&](const veco (&gv)[8], const veci (&xv)[8]) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float xt[NTI];
#pragma unroll
            for (int i = 0; i < NTI; ++i) {
                xt[i] = vget<NTI>(xv[s], i);
                if (has_xf) xt[i] = xf_apply(xt[i], xp[i], a.xf.gelu != 0);
            }
#pragma unroll
            for (int o = 0; o < NTO; ++o) {
                const float av = vget<NTO>(gv[s], o);
                bsum[o] += av;
#pragma unroll
                for (int i = 0; i < NTI; ++i) acc[o][i] = mfma32(av, xt[i], acc[o][i]);
            }
        }
    };

    long tile = tslot;
    if (tile < ntiles) load_half(tile, 0, ga, xa);
    for (; tile < ntiles; tile += ntslots) {
        load_half(tile, 1, gb, xb);
        compute_half(ga, xa);
        if (tile + ntslots < ntiles) load_half(tile + ntslots, 0, ga, xa);
        compute_half(gb, xb);
    }
    float* part = a.part + tslot * ((long)CO * CI + CO);
#pragma unroll
    for (int o = 0; o < NTO; ++o) {
#pragma unroll
        for (int i = 0; i < NTI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int orow = role * 32 * NTO + mfma_row(lane, r) * NTO + o;
                part[(long)orow * CI + col * NTI + i] = acc[o][i][r];
            }
        const float b = bsum[o] + __shfl_xor(bsum[o], 32, 64);
        if (half == 0) part[(long)CO * CI + role * 32 * NTO + col * NTO + o] = b;
    }
}

static void wgrad_shape(int CO, int CI, int& NTO, int& NTI) {
    NTI = CI / 32;
    NTO = CO / 32;
    while (NTO * NTI > 4 && NTO > 1) NTO >>= 1;
}

extern "C" long rpb_cell_wgrad_slots(long ncell, int CO, int CI) {
    int NTO, NTI;
    wgrad_shape(CO, CI, NTO, NTI);
    const int nroles = CO / (32 * NTO);
    const long ntiles = (ncell + 31) / 32;
    long slots = (long)rpb_num_cus() * 8 / nroles;
    if (slots > ntiles) slots = ((ntiles * nroles + 7) / 8) * 8 / nroles;   // keep grid*8 % nroles == 0
    if (slots < 1) slots = 8 / nroles > 0 ? 8 / nroles : 1;
    return slots;
}

extern "C" int rpb_cell_wgrad(const float* gs, const float* x, float* part, long ncell, int CO, int CI, int crop,
                              int T, int H, int W, int Tp, int Hp, int Wp, const float* xf_mean, const float* xf_invstd,
                              const float* xf_gamma, const float* xf_beta, int xf_gelu, void* stream) {
    RPB_REQUIRE(gs && x && part, "cell_wgrad: null pointer");
    if (xf_mean) RPB_REQUIRE(xf_invstd && xf_gamma && xf_beta, "cell_wgrad: bad input-transform arguments");
    RPB_REQUIRE(ncell > 0 && ncell < (1L << 31), "cell_wgrad: ncell out of range");
    int NTO, NTI;
    wgrad_shape(CO, CI, NTO, NTI);
    const int nroles = CO / (32 * NTO);
    const long slots = rpb_cell_wgrad_slots(ncell, CO, CI);
    const int grid = (int)(slots * nroles / 8);
    RPB_REQUIRE(grid >= 1 && (long)grid * 8 == slots * nroles, "cell_wgrad: internal slot arithmetic");
    WgradArgs a;
    a.gs = gs; a.x = x; a.part = part; a.ncell = ncell; a.CO = CO; a.CI = CI; a.crop = crop;
    a.cm = CropMap{T, H, W, Tp, Hp, Wp};
    a.xf = XForm{xf_mean, xf_invstd, xf_gamma, xf_beta, xf_gelu};
    hipStream_t st = (hipStream_t)stream;
    if (rpb_cwx128_supported(ncell, CO, CI, crop, W)) return rpb_cwx128_launch(gs, x, part, ncell, slots, a.xf, crop, a.cm, st);     // csrc/rpb_cwx.hip
#define RPB_WG(O_, I_)                                                                         \
    if (CO == O_ && CI == I_) {                                                                \
        hipLaunchKernelGGL((cell_wgrad_kernel<O_, I_>), dim3(grid), dim3(512), 0, st, a);      \
        RPB_CHECK_LAUNCH("cell_wgrad");                                                        \
    }
    RPB_WG(32, 32) RPB_WG(64, 64) RPB_WG(128, 128) RPB_WG(128, 32) RPB_WG(128, 64) RPB_WG(64, 32)
#undef RPB_WG
    RPB_FAIL(RPB_ERR_UNSUPPORTED, "cell_wgrad: unsupported CO=%d CI=%d", CO, CI);
}
