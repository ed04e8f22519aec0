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

// ---------------------------------------------------------------------------------- cell_mix
struct CellMixArgs {
    const float* x;       // [rows_in][KC]
    const float* Wm;      // transpose_w == 0: [CO][KC] (out = x W^T) ; == 1: [KC][CO] (out = x W)
    const float* bias;    // [CO] or null
    const float* z2;      // [G][K2][CO] or null
    const float* GW;      // [Wp][K2]
    float* out;           // [ncell][CO]
    float* stats_part;    // [gridDim.x*waves][2][CO] or null
    long ncell;
    int KC, CO, K2, Wp;
    int transpose_w;
    int gather;           // 1: out cells are padded cells, input row = pad_to_crop(cell) (zeros in the margin)
    CropMap cm;
};

template <int NT, bool HAS_SPEC, bool STATS>
__global__ __launch_bounds__(512) void cell_mix_kernel(CellMixArgs a) {
    extern __shared__ float lds[];
    const int KC = a.KC, CO = a.CO, K2 = a.K2, Wp = a.Wp;
    const int K2p = (K2 + 1) & ~1;
    float* GWl = lds;                                         // [K2p][Wp]
    float* Wl = GWl + (HAS_SPEC ? K2p * Wp : 0);              // [KC][CO]
    const int waves = blockDim.x >> 6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int XS = KC + 1;
    float* xl = Wl + KC * CO + wave * 32 * XS;                // wave-private [32][KC+1]
    int* srow = reinterpret_cast<int*>(Wl + KC * CO + waves * 32 * XS) + wave * 32;

    if (HAS_SPEC)
        for (int idx = threadIdx.x; idx < K2p * Wp; idx += blockDim.x) {
            const int k = idx / Wp, w = idx - k * Wp;
            GWl[idx] = (k < K2) ? a.GW[w * K2 + k] : 0.f;
        }
    for (int idx = threadIdx.x; idx < KC * CO; idx += blockDim.x) {
        const int k = idx / CO, n = idx - k * CO;
        Wl[idx] = a.transpose_w ? a.Wm[idx] : a.Wm[(long)n * KC + k];
    }
    __syncthreads();

    const int col = lane & 31, half = lane >> 5;
    const long ntiles = (a.ncell + 31) / 32;
    float ssum[NT], ssq[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) ssum[t] = ssq[t] = 0.f;
    float bv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bv[t] = a.bias ? a.bias[t * 32 + col] : 0.f;

    for (long tile = (long)blockIdx.x * waves + wave; tile < ntiles; tile += (long)gridDim.x * waves) {
        const long cell0 = tile * 32;
        // ---- source row of each of the 32 cells (lane < 32 computes one)
        if (lane < 32) {
            const long c = cell0 + lane;
            long r = -1;
            if (c < a.ncell) r = a.gather ? pad_to_crop(a.cm, c) : c;
            srow[lane] = (int)r;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- stage x tile: 32 rows x KC floats, float4 per lane, coalesced
        const int v4_per_row = KC >> 2;
        for (int j = 0; j < (KC >> 3); ++j) {
            const int idx = j * 64 + lane;
            const int row = idx / v4_per_row, c4 = idx - row * v4_per_row;
            const int sr = srow[row];
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (sr >= 0) v = *reinterpret_cast<const f32x4*>(a.x + (long)sr * KC + 4 * c4);
            float* d = xl + row * XS + 4 * c4;
            d[0] = v[0];
            d[1] = v[1];
            d[2] = v[2];
            d[3] = v[3];
        }
        __builtin_amdgcn_wave_barrier();

        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = zero16();

        if (HAS_SPEC) {
            const long mycell = cell0 + col;
            const bool valid = mycell < a.ncell;
            const int myg = (int)(mycell / Wp);
            const int myw = (int)(mycell - (long)myg * Wp);
            const int g0 = (int)(cell0 / Wp);
            long lastc = cell0 + 31;
            if (lastc >= a.ncell) lastc = a.ncell - 1;
            const int g1 = (int)(lastc / Wp);
            for (int gg = g0; gg <= g1; ++gg) {
                const bool mine = valid && (myg == gg);
                const float* zp = a.z2 + (long)gg * K2 * CO + col;
#pragma unroll 4
                for (int s = 0; s < K2p / 2; ++s) {
                    const int k = 2 * s + half;
                    const float av = mine ? GWl[k * Wp + myw] : 0.f;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const float b = (k < K2) ? zp[(long)k * CO + t * 32] : 0.f;
                        acc[t] = mfma32(av, b, acc[t]);
                    }
                }
            }
        }
        // ---- channel mixing: A = x tile (LDS, transposed read), B = W (LDS)
#pragma unroll 4
        for (int s = 0; s < KC / 2; ++s) {
            const int k = 2 * s + half;
            const float av = xl[col * XS + k];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma32(av, Wl[k * CO + t * 32 + col], acc[t]);
        }
        // ---- epilogue
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long c = cell0 + mfma_row(lane, r);
                if (c < a.ncell) {
                    const float v = acc[t][r] + bv[t];
                    a.out[c * CO + t * 32 + col] = v;
                    if (STATS) {
                        ssum[t] += v;
                        ssq[t] += v * v;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (STATS) {
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

static size_t cell_mix_lds(int KC, int CO, int K2, int Wp, bool spec, int waves) {
    const int K2p = (K2 + 1) & ~1;
    return ((size_t)(spec ? K2p * Wp : 0) + (size_t)KC * CO + (size_t)waves * 32 * (KC + 1) + (size_t)waves * 32) * 4;
}

static int cell_mix_waves(int KC, int CO, int K2, int Wp, bool spec) {
    for (int w = 8; w >= 1; w >>= 1)
        if (cell_mix_lds(KC, CO, K2, Wp, spec, w) <= 160 * 1024) return w;
    return 0;
}

template <int NT, bool SPEC, bool STATS>
static int launch_cell_mix(const CellMixArgs& a, int waves, int grid, hipStream_t st) {
    const size_t lds = cell_mix_lds(a.KC, a.CO, a.K2, a.Wp, SPEC, waves);
    (void)hipFuncSetAttribute((const void*)cell_mix_kernel<NT, SPEC, STATS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)lds);
    hipLaunchKernelGGL((cell_mix_kernel<NT, SPEC, STATS>), dim3(grid), dim3(waves * 64), lds, st, a);
    RPB_CHECK_LAUNCH("cell_mix");
}

// number of [2][CO] stat partial rows cell_mix writes for this problem (== grid * waves)
extern "C" long rpb_cell_mix_stat_rows(long ncell, int KC, int CO, int K2, int Wp, int has_spec) {
    const int waves = cell_mix_waves(KC, CO, K2, Wp, has_spec != 0);
    if (waves == 0) return -1;
    const long ntiles = (ncell + 31) / 32;
    long grid = rpb_num_cus();
    const long need = (ntiles + waves - 1) / waves;
    if (grid > need) grid = need;
    return grid * waves;
}

extern "C" int rpb_cell_mix(const float* x, const float* Wm, const float* bias, const float* z2, const float* GW,
                            float* out, float* stats_part, long ncell, int KC, int CO, int K2, int Wp, int transpose_w,
                            int gather, int T, int H, int W, int Tp, int Hp, int Wp_pad, void* stream) {
    RPB_REQUIRE(x && Wm && out, "cell_mix: null pointer");
    RPB_REQUIRE(ncell > 0 && ncell < (1L << 31), "cell_mix: ncell=%ld out of range", ncell);
    RPB_REQUIRE(KC % 8 == 0 && KC > 0, "cell_mix: KC=%d must be a multiple of 8", KC);
    RPB_REQUIRE(CO == 32 || CO == 64 || CO == 128, "cell_mix: CO=%d must be 32, 64 or 128", CO);
    const bool spec = z2 != nullptr;
    if (spec) RPB_REQUIRE(GW && K2 > 0 && Wp > 0 && ncell % Wp == 0, "cell_mix: bad spectral arguments");
    const int waves = cell_mix_waves(KC, CO, K2, Wp, spec);
    RPB_REQUIRE(waves > 0, "cell_mix: tiles do not fit LDS (KC=%d CO=%d K2=%d Wp=%d)", KC, CO, K2, Wp);
    CellMixArgs a;
    a.x = x; a.Wm = Wm; a.bias = bias; a.z2 = z2; a.GW = GW; a.out = out; a.stats_part = stats_part;
    a.ncell = ncell; a.KC = KC; a.CO = CO; a.K2 = spec ? K2 : 0; a.Wp = spec ? Wp : 1;
    a.transpose_w = transpose_w; a.gather = gather;
    a.cm = CropMap{T, H, W, Tp, Hp, Wp_pad};
    const int grid = (int)(rpb_cell_mix_stat_rows(ncell, KC, CO, K2, Wp, spec) / waves);
    hipStream_t st = (hipStream_t)stream;
    const bool stats = stats_part != nullptr;
    const int NT = CO / 32;
#define RPB_CM(NT_, S_, ST_) \
    if (NT == NT_ && spec == S_ && stats == ST_) return launch_cell_mix<NT_, S_, ST_>(a, waves, grid, st);
    RPB_CM(1, true, true) RPB_CM(1, true, false) RPB_CM(1, false, true) RPB_CM(1, false, false)
    RPB_CM(2, true, true) RPB_CM(2, true, false) RPB_CM(2, false, true) RPB_CM(2, false, false)
    RPB_CM(4, true, true) RPB_CM(4, true, false) RPB_CM(4, false, true) RPB_CM(4, false, false)
#undef RPB_CM
    RPB_FAIL(RPB_ERR_UNSUPPORTED, "cell_mix: unsupported configuration");
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
};

template <int NTO, int NTI>
__global__ __launch_bounds__(512) void cell_wgrad_kernel(WgradArgs a) {
    const int waves = blockDim.x >> 6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int col = lane & 31, half = lane >> 5;
    const int CO = a.CO, CI = a.CI;
    const int nroles = CO / (32 * NTO);
    const long sid = (long)blockIdx.x * waves + wave;
    const int role = (int)(sid % nroles);
    const long tslot = sid / nroles;
    const long ntslots = ((long)gridDim.x * waves) / nroles;
    const long ntiles = (a.ncell + 31) / 32;

    f32x16 acc[NTO][NTI];
#pragma unroll
    for (int o = 0; o < NTO; ++o)
#pragma unroll
        for (int i = 0; i < NTI; ++i) acc[o][i] = zero16();
    float bsum[NTO];
#pragma unroll
    for (int o = 0; o < NTO; ++o) bsum[o] = 0.f;

    for (long tile = tslot; tile < ntiles; tile += ntslots) {
        const long cell0 = tile * 32;
        long myrow = -1;
        {
            const long c = cell0 + col;
            if (c < a.ncell) myrow = a.crop ? crop_to_pad(a.cm, c) : c;
        }
        const int myrow_i = (int)myrow;
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const int j = 2 * s + half;
            const long c = cell0 + j;
            const int xr = __shfl(myrow_i, j, 64);
            const bool ok = c < a.ncell;
            float av[NTO], bx[NTI];
#pragma unroll
            for (int o = 0; o < NTO; ++o) av[o] = ok ? a.gs[c * CO + (role * NTO + o) * 32 + col] : 0.f;
#pragma unroll
            for (int i = 0; i < NTI; ++i) bx[i] = ok ? a.x[(long)xr * CI + i * 32 + col] : 0.f;
#pragma unroll
            for (int o = 0; o < NTO; ++o) {
                bsum[o] += av[o];
#pragma unroll
                for (int i = 0; i < NTI; ++i) acc[o][i] = mfma32(av[o], bx[i], acc[o][i]);
            }
        }
    }
    float* part = a.part + tslot * ((long)CO * CI + CO);
#pragma unroll
    for (int o = 0; o < NTO; ++o) {
#pragma unroll
        for (int i = 0; i < NTI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int orow = (role * NTO + o) * 32 + mfma_row(lane, r);
                part[(long)orow * CI + i * 32 + col] = acc[o][i][r];
            }
        const float b = bsum[o] + __shfl_xor(bsum[o], 32, 64);
        if (half == 0) part[(long)CO * CI + (role * NTO + o) * 32 + col] = b;
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
                              int T, int H, int W, int Tp, int Hp, int Wp, void* stream) {
    RPB_REQUIRE(gs && x && part, "cell_wgrad: null pointer");
    RPB_REQUIRE(ncell > 0 && ncell < (1L << 31), "cell_wgrad: ncell out of range");
    RPB_REQUIRE((CO == 32 || CO == 64 || CO == 128) && (CI == 32 || CI == 64 || CI == 128),
                "cell_wgrad: CO=%d CI=%d must each be 32, 64 or 128", CO, CI);
    int NTO, NTI;
    wgrad_shape(CO, CI, NTO, NTI);
    const int nroles = CO / (32 * NTO);
    const long slots = rpb_cell_wgrad_slots(ncell, CO, CI);
    const int grid = (int)(slots * nroles / 8);
    RPB_REQUIRE(grid >= 1 && (long)grid * 8 == slots * nroles, "cell_wgrad: internal slot arithmetic");
    WgradArgs a;
    a.gs = gs; a.x = x; a.part = part; a.ncell = ncell; a.CO = CO; a.CI = CI; a.crop = crop;
    a.cm = CropMap{T, H, W, Tp, Hp, Wp};
    hipStream_t st = (hipStream_t)stream;
#define RPB_WG(O_, I_)                                                                         \
    if (NTO == O_ && NTI == I_) {                                                              \
        hipLaunchKernelGGL((cell_wgrad_kernel<O_, I_>), dim3(grid), dim3(512), 0, st, a);      \
        RPB_CHECK_LAUNCH("cell_wgrad");                                                        \
    }
    RPB_WG(1, 1) RPB_WG(2, 1) RPB_WG(4, 1) RPB_WG(1, 2) RPB_WG(2, 2) RPB_WG(1, 4)
#undef RPB_WG
    RPB_FAIL(RPB_ERR_UNSUPPORTED, "cell_wgrad: unsupported NTO=%d NTI=%d", NTO, NTI);
}
