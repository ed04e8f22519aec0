// K7: crop + fc1 (C -> HID=128) + exact-erf GELU + fc2 (128 -> DO = C_out*r) on channels-last cells.
// Replaces fno.py:121-125 (the crop `x[..., :-6, :-6, :-6]`, the permute, fc1, F.gelu, fc2).
//
// fc1 runs on v_mfma_f32_32x32x2_f32 (tile = 32 cropped cells x 128 hidden, K = C); fc2 is tiny
// (DO = 2..16 outputs) so it is a lane-local dot product over the hidden index followed by a recursive-halving
// butterfly over the 32 lanes that leaves exactly one (cell, output) sum per lane (31 shuffles instead of 160).
// The activation tile is gathered from the padded tensor (crop fused into the load), prefetched one tile ahead
// in registers and transposed through a wave-private +1-padded LDS tile.
//
// proj_bwd recomputes u = fc1 a + b1 instead of saving it (saves 5.4 GB of HBM traffic per step at
// B=32), produces gu = (fc2^T g) * gelu'(u) for the downstream dgrad / wgrad kernels and accumulates
// d fc2.weight, d fc2.bias, d fc1.bias in registers.
#include "rpb_common.h"
#include <stdlib.h>
#include "rpb_pjx.h"

#define HID 128
#define NTH (HID / 32)

struct ProjArgs {
    const float* a;      // padded activations [B*Tp*Hp*Wp][C]
    const float* w1;     // fc1.weight [HID][C]
    const float* b1;     // [HID]
    const float* w2;     // fc2.weight [DO][HID]
    const float* b2;     // [DO]
    float* out;          // fwd: [ncrop][DO]
    const float* gout;   // bwd: [ncrop][DO]
    float* gu;           // bwd: [ncrop][HID]
    float* part;         // bwd: [nslots][DO*HID + HID + DO]   (d w2, d b1, d b2)
    long ncrop;
    int C, DO;
    CropMap cm;
    XForm xf;            // lazy BatchNorm of the last Fourier layer (no GELU there, fno.py:118)
    int act;             // 0: exact GELU (fno.py:124); 1: SiLU (Galerkin SpectralRegressor, model.py:631-632)
    int a_bf16;          // forward only: `a` holds bf16 (BASELINE.json configs[4] activation storage)
};

// v = gelu(u), d = gelu'(u) with ONE erf evaluation
__device__ __forceinline__ void gelu_pair(float u, float& v, float& d) {
    const float cdf = 0.5f * (1.0f + fast_erf(u * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * u * u);
    v = u * cdf;
    d = cdf + u * pdf;
}

// v = silu(u) = u * sigmoid(u), d = silu'(u) = sig * (1 + u * (1 - sig))
__device__ __forceinline__ void silu_pair(float u, float& v, float& d) {
    const float sig = 1.0f / (1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * u));
    v = u * sig;
    d = sig * (1.0f + u * (1.0f - sig));
}

// DOT = compile-time bound on DO (register arrays must be statically indexed); ROWFAST = W % 32 == 0, i.e. a
// 32-cell tile never leaves its (b,t,h) row and the crop gather is one contiguous 32*C block (keeps the generic
// gather path, with its per-lane 64-bit addresses, out of the hot instantiation's register budget)
template <int C, bool BWD, int DOT, bool ROWFAST, bool SILU>
__global__ __launch_bounds__(512) void proj_kernel(ProjArgs p) {
    extern __shared__ float lds[];
    constexpr int XS = C + 1;
    constexpr int NX = C / 8;
    const int DO = p.DO;
    const int waves = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int col = lane & 31, half = lane >> 5;
    float* W1l = lds;                       // [C][32][4]   B[k=i][n=hid=t*32+col] = w1[hid][i], 4 tiles per 16 B read
    float* W2l = W1l + C * HID;             // [DO][HID]
    float* xl = W2l + DO * HID + wave * 32 * XS;
    int* srow = reinterpret_cast<int*>(W2l + DO * HID + waves * 32 * XS) + wave * 32;

    for (int idx = threadIdx.x; idx < C * HID; idx += blockDim.x) {
        const int k = idx / HID, n = idx - k * HID;
        W1l[(k * 32 + (n & 31)) * 4 + (n >> 5)] = p.w1[(long)n * C + k];
    }
    for (int idx = threadIdx.x; idx < DO * HID; idx += blockDim.x) W2l[idx] = p.w2[idx];
    __syncthreads();

    float b1v[NTH], w2r[DOT][NTH];
#pragma unroll
    for (int t = 0; t < NTH; ++t) b1v[t] = p.b1[t * 32 + col];
#pragma unroll
    for (int j = 0; j < DOT; ++j)
#pragma unroll
        for (int t = 0; t < NTH; ++t) w2r[j][t] = (j < DO) ? W2l[j * HID + t * 32 + col] : 0.f;
    float dw2[BWD ? DOT : 1][NTH], db1[NTH], db2[BWD ? DOT : 1];
    if (BWD) {
#pragma unroll
        for (int j = 0; j < DOT; ++j) {
            db2[j] = 0.f;
#pragma unroll
            for (int t = 0; t < NTH; ++t) dw2[j][t] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < NTH; ++t) db1[t] = 0.f;
    }

    const long ntiles = (p.ncrop + 31) / 32;
    const long tstride = (long)gridDim.x * waves;

    const bool has_xf = p.xf.mean != nullptr;
    XParam xp4[4];
    if (has_xf) {
#pragma unroll
        for (int k = 0; k < 4; ++k) xp4[k] = xf_load(p.xf, 4 * (lane % (C / 4)) + k);
    }
    f32x4 xr[NX];
    // 4 consecutive channels of a cell: 16 B of fp32, or 8 B of bf16 widened (exact) to fp32
    auto load4 = [&](long elem) -> f32x4 {
        if (!BWD && p.a_bf16) {
            const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p.a) + elem);
            return f32x4{__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xffff0000u),
                         __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xffff0000u)};
        }
        return *reinterpret_cast<const f32x4*>(p.a + elem);
    };
    auto issue_x = [&](long tile) {
        const long q0 = tile * 32;
        if (ROWFAST && q0 + 32 <= p.ncrop) {
            const long base = crop_to_pad(p.cm, q0) * C;                     // uniform: 32 consecutive padded cells
#pragma unroll
            for (int j = 0; j < NX; ++j) xr[j] = load4(base + lane * 4 + j * 256);
        } else {
            if (lane < 32) {
                const long q = q0 + lane;
                srow[lane] = (q < p.ncrop) ? (int)crop_to_pad(p.cm, q) : -1;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                const int idx = j * 64 + lane;
                const int row = idx / (C / 4), c4 = idx - row * (C / 4);
                const int sr = srow[row];
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (sr >= 0) v = load4((long)sr * C + 4 * c4);
                xr[j] = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
    };

    // forward: the next tile's x loads are prefetched in registers; backward: the epilogue needs the registers
    // (prefetching there makes hipcc spill the prefetch buffer to scratch and serialise it: 9.5 ms vs ~5 ms)
    constexpr bool PREFETCH_X = !BWD;
    long tile = (long)blockIdx.x * waves + wave;
    if (PREFETCH_X && tile < ntiles) issue_x(tile);
    for (; tile < ntiles; tile += tstride) {
        const long q0 = tile * 32;
        const bool full = q0 + 32 <= p.ncrop;
        if (!PREFETCH_X) issue_x(tile);
        {
            float* d0 = xl + (lane / (C / 4)) * XS + 4 * (lane % (C / 4));
            if (has_xf) {
#pragma unroll
                for (int j = 0; j < NX; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k) xr[j][k] = xf_apply(xr[j][k], xp4[k], p.xf.gelu != 0);
            }
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                float* d = d0 + j * (64 / (C / 4)) * XS;
                d[0] = xr[j][0];
                d[1] = xr[j][1];
                d[2] = xr[j][2];
                d[3] = xr[j][3];
            }
        }
        // backward: this tile's dLoss/dout rows go out before the MFMAs so their latency is hidden
        float gpre[BWD ? 16 : 1][BWD ? DOT : 1];
        if (BWD) {
            // rows beyond ncrop are out of the descriptor's range and read as 0
            const rsrc_t gr = make_rsrc(p.gout + q0 * DO, tile_bytes(p.ncrop - q0, 32, DO * 4));
            const int vo = 4 * half * DO * 4;
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int j = 0; j < DOT; ++j)
                    gpre[r][j] = (j < DO) ? buf_load_f32(gr, vo + ((8 * (r >> 2) + (r & 3)) * DO + j) * 4, 0) : 0.f;
        }
        if (PREFETCH_X && tile + tstride < ntiles) issue_x(tile + tstride);
        __builtin_amdgcn_wave_barrier();

        f32x16 acc[NTH];
#pragma unroll
        for (int t = 0; t < NTH; ++t) acc[t] = zero16();
        {
            const float* ap = xl + col * XS + half;
            const float* bp = W1l + (half * 32 + col) * 4;
#pragma unroll 8
            for (int s = 0; s < C / 2; ++s) {
                const float av = ap[2 * s];
                const f32x4 b = *reinterpret_cast<const f32x4*>(bp + 2 * s * 32 * 4);
#pragma unroll
                for (int t = 0; t < NTH; ++t) acc[t] = mfma32(av, b[t], acc[t]);
            }
        }

        if (!BWD) {
            // out[row][j] = b2[j] + sum_hid gelu(u[row][hid]) * w2[j][hid]
            constexpr int NVAL = 16 * DOT;
            float val[NVAL];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v[NTH];
#pragma unroll
                for (int t = 0; t < NTH; ++t) {
                    if (SILU) {
                        float dd;
                        silu_pair(acc[t][r] + b1v[t], v[t], dd);
                    } else {
                        v[t] = gelu_f(acc[t][r] + b1v[t]);
                    }
                }
#pragma unroll
                for (int j = 0; j < DOT; ++j) {
                    float s = 0.f;
#pragma unroll
                    for (int t = 0; t < NTH; ++t) s += v[t] * w2r[j][t];
                    val[r * DOT + j] = s;
                }
            }
            // recursive halving over the 32 lanes of each half-wave: afterwards a lane owns NVAL/32 complete sums
            int base_idx = 0;
#pragma unroll
            for (int step = 0; step < 5; ++step) {
                const int off = 16 >> step;
                const int n = NVAL >> (step + 1);
                const bool hi = (lane & off) != 0;
#pragma unroll
                for (int i = 0; i < n; ++i) {
                    const float lo_v = val[i], hi_v = val[i + n];
                    const float send = hi ? lo_v : hi_v;
                    const float keep = hi ? hi_v : lo_v;
                    val[i] = keep + __shfl_xor(send, off, 64);
                }
                base_idx += hi ? n : 0;
            }
#pragma unroll
            for (int i = 0; i < NVAL / 32; ++i) {
                const int idx = base_idx + i;
                const int r = idx / DOT, j = idx - r * DOT;
                const long q = q0 + 8 * (r >> 2) + 4 * half + (r & 3);
                if (j < DO && (full || q < p.ncrop)) p.out[q * DO + j] = val[i] + p.b2[j];
            }
        } else {
            // gu[row][hid] = (sum_j g[row][j] w2[j][hid]) * gelu'(u);  accumulate d w2, d b1, d b2
            const rsrc_t ur = make_rsrc(p.gu + q0 * HID, tile_bytes(p.ncrop - q0, 32, HID * 4));
            const int vo = (4 * half * HID + col) * 4;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float g[DOT];
#pragma unroll
                for (int j = 0; j < DOT; ++j) {
                    g[j] = gpre[r][j];
                    db2[j] += g[j];
                }
#pragma unroll
                for (int t = 0; t < NTH; ++t) {
                    float v, d;
                    if (SILU) silu_pair(acc[t][r] + b1v[t], v, d);
                    else gelu_pair(acc[t][r] + b1v[t], v, d);
                    float gvs = 0.f;
#pragma unroll
                    for (int j = 0; j < DOT; ++j) {
                        gvs += g[j] * w2r[j][t];
                        dw2[j][t] += g[j] * v;
                    }
                    const float guv = gvs * d;          // rows past ncrop have g == 0 -> guv == 0 (and the store is dropped)
                    buf_store_f32(guv, ur, vo + ((8 * (r >> 2) + (r & 3)) * HID + t * 32) * 4, 0);
                    db1[t] += guv;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (BWD) {
        float* part = p.part + ((long)blockIdx.x * waves + wave) * ((long)DO * HID + HID + DO);
#pragma unroll
        for (int j = 0; j < DOT; ++j) {
            if (j < DO) {
#pragma unroll
                for (int t = 0; t < NTH; ++t) {
                    const float v = dw2[j][t] + __shfl_xor(dw2[j][t], 32, 64);
                    if (half == 0) part[j * HID + t * 32 + col] = v;
                }
                // every lane of a half holds the same row sums: lanes 0 and 32 carry the two halves
                const float b = db2[j] + __shfl_xor(db2[j], 32, 64);
                if (lane == 0) part[DO * HID + HID + j] = b;
            }
        }
#pragma unroll
        for (int t = 0; t < NTH; ++t) {
            const float v = db1[t] + __shfl_xor(db1[t], 32, 64);
            if (half == 0) part[DO * HID + t * 32 + col] = v;
        }
    }
}

static size_t proj_lds(int C, int DO, int waves) {
    return ((size_t)C * HID + (size_t)DO * HID + (size_t)waves * 32 * (C + 1) + (size_t)waves * 32) * 4;
}
static int proj_waves(int C, int DO) {
    for (int w = 8; w >= 1; w >>= 1)
        if (proj_lds(C, DO, w) <= 160 * 1024) return w;
    return 0;
}

extern "C" long rpb_proj_slots(long ncrop, int C, int DO) {
    const int waves = proj_waves(C, DO);
    if (!waves) return -1;
    const long ntiles = (ncrop + 31) / 32;
    long grid = rpb_num_cus();
    const long need = (ntiles + waves - 1) / waves;
    if (grid > need) grid = need;
    return grid * waves;
}

template <int C, bool BWD, int DOT, bool ROWFAST, bool SILU>
static void proj_launch_k(ProjArgs& p, int grid, int waves, size_t lds, hipStream_t st) {
    (void)hipFuncSetAttribute((const void*)proj_kernel<C, BWD, DOT, ROWFAST, SILU>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((proj_kernel<C, BWD, DOT, ROWFAST, SILU>), dim3(grid), dim3(waves * 64), lds, st, p);
}

template <int C, bool BWD, int DOT>
static void proj_launch_t(ProjArgs& p, int grid, int waves, size_t lds, hipStream_t st) {
    const bool rowfast = p.cm.W % 32 == 0;
    if (p.act) {
        if (rowfast) proj_launch_k<C, BWD, DOT, true, true>(p, grid, waves, lds, st);
        else proj_launch_k<C, BWD, DOT, false, true>(p, grid, waves, lds, st);
    } else {
        if (rowfast) proj_launch_k<C, BWD, DOT, true, false>(p, grid, waves, lds, st);
        else proj_launch_k<C, BWD, DOT, false, false>(p, grid, waves, lds, st);
    }
}

template <int C>
static void proj_launch_c(bool bwd, ProjArgs& p, int grid, int waves, size_t lds, hipStream_t st) {
    const int d = p.DO <= 2 ? 2 : p.DO <= 4 ? 4 : p.DO <= 8 ? 8 : 16;
#define RPB_PJ(D_)                                                     \
    if (d == D_) {                                                     \
        if (bwd) proj_launch_t<C, true, D_>(p, grid, waves, lds, st);  \
        else proj_launch_t<C, false, D_>(p, grid, waves, lds, st);     \
    }
    RPB_PJ(2) RPB_PJ(4) RPB_PJ(8) RPB_PJ(16)
#undef RPB_PJ
}

static int proj_launch(bool bwd, ProjArgs& p, hipStream_t st) {
    RPB_REQUIRE(p.a && p.w1 && p.b1 && p.w2 && p.b2, "proj: null pointer");
    RPB_REQUIRE(p.C == 32 || p.C == 64 || p.C == 128, "proj: C=%d must be 32, 64 or 128", p.C);
    RPB_REQUIRE(p.DO >= 1 && p.DO <= 16, "proj: fc2 out features %d not in [1,16]", p.DO);
    RPB_REQUIRE(p.act == 0 || p.act == 1, "proj: act=%d must be 0 (GELU) or 1 (SiLU)", p.act);
    RPB_REQUIRE(p.ncrop > 0 && p.ncrop < (1L << 31), "proj: ncrop out of range");
    const int waves = proj_waves(p.C, p.DO);
    RPB_REQUIRE(waves > 0, "proj: does not fit LDS");
    const int grid = (int)(rpb_proj_slots(p.ncrop, p.C, p.DO) / waves);
    const size_t lds = proj_lds(p.C, p.DO, waves);
    if (p.C == 32) proj_launch_c<32>(bwd, p, grid, waves, lds, st);
    else if (p.C == 64) proj_launch_c<64>(bwd, p, grid, waves, lds, st);
    else proj_launch_c<128>(bwd, p, grid, waves, lds, st);
    RPB_CHECK_LAUNCH("proj");
}

extern "C" int rpb_proj_fwd(const float* a, const float* w1, const float* b1, const float* w2, const float* b2,
                            float* out, long ncrop, int C, int DO, int T, int H, int W, int Tp, int Hp, int Wp,
                            const float* xf_mean, const float* xf_invstd, const float* xf_gamma, const float* xf_beta,
                            int xf_gelu, int act, void* stream) {
    RPB_REQUIRE(out, "proj_fwd: null out");
    ProjArgs p{};
    p.act = act;
    p.xf = XForm{xf_mean, xf_invstd, xf_gamma, xf_beta, xf_gelu};
    p.a = a; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.out = out; p.ncrop = ncrop; p.C = C; p.DO = DO;
    p.cm = CropMap{T, H, W, Tp, Hp, Wp};
    if (rpb_pjx_head_supported(C, DO, false)) {      // bf16 matrix pipe, fp32-grade split operands (csrc/rpb_pjx.hip)
        RPB_REQUIRE(a && w1 && b1 && w2 && b2, "proj: null pointer");
        return rpb_pjx_head_launch(false, a, w1, b1, w2, b2, nullptr, out, nullptr, nullptr, 0, DO, T, H, W, Tp, Hp, Wp, ncrop, p.xf, act,
                                   (hipStream_t)stream);
    }
    if (C == 128 && a && w1 && b1 && w2 && b2 && ncrop % ((long)T * H * W) == 0 && rpb_pjh_supported(128, DO, act, p.xf, false))
        // width 128 (configs/fsi/fno.yaml), at most four outputs, GELU: csrc/rpb_pjh.hip's C = 128 instance (round 6b; the fp32-pipe kernel
        // below ran it at 0.68 TB/s)
        return rpb_pjh_launch(a, w1, b1, w2, b2, out, (int)(ncrop / ((long)T * H * W)), DO, T, H, W, Tp, Hp, Wp, p.xf, (hipStream_t)stream, false,
                              false, 128, act == 1);
    return proj_launch(false, p, (hipStream_t)stream);
}

// The evaluation head on the opt-in "f16x2" arithmetic (csrc/rpb_pjh.hip, H2; the contract of rpb_cell_mix_eval_dft_f16x2): C = 64, at most
// four fc2 outputs, exact-erf GELU, plain fp32 activations (the last layer's BatchNorm was applied by its eval cell_mix).
extern "C" int rpb_proj_fwd_f16x2(const float* a, const float* w1, const float* b1, const float* w2, const float* b2, float* out, long ncrop,
                                  int DO, int T, int H, int W, int Tp, int Hp, int Wp, void* stream) {
    RPB_REQUIRE(a && w1 && b1 && w2 && b2 && out, "proj_fwd_f16x2: null pointer");
    RPB_REQUIRE(T > 0 && H > 0 && W > 0 && ncrop > 0 && ncrop % ((long)T * H * W) == 0, "proj_fwd_f16x2: bad sizes");
    const XForm xf{nullptr, nullptr, nullptr, nullptr, 0};
    RPB_REQUIRE(rpb_pjh_supported(64, DO, 0, xf, false), "proj_fwd_f16x2: needs 1 <= DO <= 4 (DO=%d)", DO);
    return rpb_pjh_launch(a, w1, b1, w2, b2, out, (int)(ncrop / ((long)T * H * W)), DO, T, H, W, Tp, Hp, Wp, xf, (hipStream_t)stream, true);
}

extern "C" int rpb_proj_fwd_bf16(const void* a_bf16, const float* w1, const float* b1, const float* w2, const float* b2,
                                 float* out, long ncrop, int C, int DO, int T, int H, int W, int Tp, int Hp, int Wp, int act,
                                 void* stream) {
    RPB_REQUIRE(out && a_bf16, "proj_fwd_bf16: null pointer");
    ProjArgs p{};
    p.act = act;
    p.a_bf16 = 1;
    p.xf = XForm{nullptr, nullptr, nullptr, nullptr, 0};
    p.a = (const float*)a_bf16; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.out = out; p.ncrop = ncrop; p.C = C; p.DO = DO;
    p.cm = CropMap{T, H, W, Tp, Hp, Wp};
    if (rpb_pjx_head_supported(C, DO, false)) {
        RPB_REQUIRE(w1 && b1 && w2 && b2, "proj: null pointer");
        return rpb_pjx_head_launch(false, (const float*)a_bf16, w1, b1, w2, b2, nullptr, out, nullptr, nullptr, 0, DO, T, H, W, Tp, Hp, Wp,
                                   ncrop, p.xf, act, (hipStream_t)stream, true);
    }
    return proj_launch(false, p, (hipStream_t)stream);
}

extern "C" int rpb_proj_bwd(const float* a, const float* w1, const float* b1, const float* w2, const float* b2,
                            const float* gout, float* gu, float* part, long ncrop, int C, int DO, int T, int H, int W,
                            int Tp, int Hp, int Wp, const float* xf_mean, const float* xf_invstd, const float* xf_gamma,
                            const float* xf_beta, int xf_gelu, int act, void* stream) {
    RPB_REQUIRE(gout && gu && part, "proj_bwd: null pointer");
    ProjArgs p{};
    p.act = act;
    p.xf = XForm{xf_mean, xf_invstd, xf_gamma, xf_beta, xf_gelu};
    p.a = a; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.gout = gout; p.gu = gu; p.part = part;
    p.ncrop = ncrop; p.C = C; p.DO = DO;
    p.cm = CropMap{T, H, W, Tp, Hp, Wp};
    if (rpb_pjx_head_supported(C, DO, true)) {
        RPB_REQUIRE(a && w1 && b1 && w2 && b2, "proj: null pointer");
        return rpb_pjx_head_launch(true, a, w1, b1, w2, b2, gout, nullptr, gu, part, rpb_proj_slots(ncrop, C, DO), DO, T, H, W, Tp, Hp, Wp,
                                   ncrop, p.xf, act, (hipStream_t)stream);
    }
    if (C == 128 && a && w1 && b1 && w2 && b2 && ncrop % ((long)T * H * W) == 0 && rpb_pjh_supported(128, DO, act, p.xf, false) &&
        !(getenv("RPB_HEAD_PJH_128_BWD") && atoi(getenv("RPB_HEAD_PJH_128_BWD")) == 0))
        return rpb_pjh_bwd128_launch(a, w1, b1, w2, b2, gout, gu, part, rpb_proj_slots(ncrop, C, DO), (int)(ncrop / ((long)T * H * W)), DO, T, H, W,
                                     Tp, Hp, Wp, p.xf, (hipStream_t)stream, act == 1);
    return proj_launch(true, p, (hipStream_t)stream);
}
