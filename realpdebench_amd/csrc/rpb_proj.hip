// K7: crop + fc1 (C -> HID=128) + exact-erf GELU + fc2 (128 -> DO = C_out*r) on channels-last cells.
// Replaces fno.py:121-125 (the crop `x[..., :-6, :-6, :-6]`, the permute, fc1, F.gelu, fc2).
//
// fc1 runs on v_mfma_f32_32x32x2_f32 (tile = 32 cropped cells x 128 hidden, K = C); fc2 is tiny
// (DO = 2..16 outputs) so it is a lane-local dot product over the hidden index followed by a 32-lane
// butterfly sum.  The activation tile is gathered from the padded tensor (crop fused into the load) and
// transposed through a wave-private +1-padded LDS tile.
//
// proj_bwd recomputes u = fc1 a + b1 instead of saving it (saves 5.4 GB of HBM traffic per step at
// B=32), produces gu = (fc2^T g) * gelu'(u) for the downstream dgrad / wgrad kernels and accumulates
// d fc2.weight, d fc2.bias, d fc1.bias in registers.
#include "rpb_common.h"

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
};

template <bool BWD, int DOT>   // DOT = compile-time bound on DO (register arrays must be statically indexed)
__global__ __launch_bounds__(512) void proj_kernel(ProjArgs p) {
    extern __shared__ float lds[];
    const int C = p.C, DO = p.DO;
    const int waves = blockDim.x >> 6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int col = lane & 31, half = lane >> 5;
    const int XS = C + 1;
    float* W1l = lds;                       // [C][HID]   B[k=i][n=hid] = w1[hid][i]
    float* W2l = W1l + C * HID;             // [DO][HID]
    float* xl = W2l + DO * HID + wave * 32 * XS;
    int* srow = reinterpret_cast<int*>(W2l + DO * HID + waves * 32 * XS) + wave * 32;

    for (int idx = threadIdx.x; idx < C * HID; idx += blockDim.x) {
        const int k = idx / HID, n = idx - k * HID;
        W1l[idx] = p.w1[(long)n * C + k];
    }
    for (int idx = threadIdx.x; idx < DO * HID; idx += blockDim.x) W2l[idx] = p.w2[idx];
    __syncthreads();

    float b1v[NTH];
#pragma unroll
    for (int t = 0; t < NTH; ++t) b1v[t] = p.b1[t * 32 + col];

    // backward accumulators (per lane: hidden index = t*32+col, rows of this lane's half)
    float dw2[DOT][NTH], w2r[DOT][NTH];
    float db1[NTH], db2[DOT];
    if (BWD) {
#pragma unroll
        for (int j = 0; j < DOT; ++j) {
            db2[j] = 0.f;
#pragma unroll
            for (int t = 0; t < NTH; ++t) {
                dw2[j][t] = 0.f;
                w2r[j][t] = (j < DO) ? W2l[j * HID + t * 32 + col] : 0.f;
            }
        }
#pragma unroll
        for (int t = 0; t < NTH; ++t) db1[t] = 0.f;
    }

    const long ntiles = (p.ncrop + 31) / 32;
    for (long tile = (long)blockIdx.x * waves + wave; tile < ntiles; tile += (long)gridDim.x * waves) {
        const long q0 = tile * 32;
        if (lane < 32) {
            const long q = q0 + lane;
            srow[lane] = (q < p.ncrop) ? (int)crop_to_pad(p.cm, q) : -1;
        }
        __builtin_amdgcn_wave_barrier();
        const int v4_per_row = C >> 2;
        for (int j = 0; j < (C >> 3); ++j) {
            const int idx = j * 64 + lane;
            const int row = idx / v4_per_row, c4 = idx - row * v4_per_row;
            const int sr = srow[row];
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (sr >= 0) v = *reinterpret_cast<const f32x4*>(p.a + (long)sr * C + 4 * c4);
            float* d = xl + row * XS + 4 * c4;
            d[0] = v[0];
            d[1] = v[1];
            d[2] = v[2];
            d[3] = v[3];
        }
        __builtin_amdgcn_wave_barrier();

        f32x16 acc[NTH];
#pragma unroll
        for (int t = 0; t < NTH; ++t) acc[t] = zero16();
#pragma unroll 2
        for (int s = 0; s < C / 2; ++s) {
            const int k = 2 * s + half;
            const float av = xl[col * XS + k];
#pragma unroll
            for (int t = 0; t < NTH; ++t) acc[t] = mfma32(av, W1l[k * HID + t * 32 + col], acc[t]);
        }

        if (!BWD) {
            // v = gelu(u); out[row][j] = b2[j] + sum_hid v[row][hid] * w2[j][hid]
#pragma unroll
            for (int t = 0; t < NTH; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = gelu_f(acc[t][r] + b1v[t]);
            for (int j = 0; j < DO; ++j) {
                float w2v[NTH];
#pragma unroll
                for (int t = 0; t < NTH; ++t) w2v[t] = W2l[j * HID + t * 32 + col];
                const float bj = p.b2[j];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float s = 0.f;
#pragma unroll
                    for (int t = 0; t < NTH; ++t) s += acc[t][r] * w2v[t];
#pragma unroll
                    for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
                    const long q = q0 + mfma_row(lane, r);
                    if (col == 0 && q < p.ncrop) p.out[q * DO + j] = s + bj;
                }
            }
        } else {
            // gu[row][hid] = (sum_j g[row][j] w2[j][hid]) * gelu'(u);  accumulate d w2, d b1, d b2
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long q = q0 + mfma_row(lane, r);
                const bool ok = q < p.ncrop;
                float g[DOT];
#pragma unroll
                for (int j = 0; j < DOT; ++j) {
                    g[j] = (ok && j < DO) ? p.gout[q * DO + j] : 0.f;
                    db2[j] += g[j];
                }
#pragma unroll
                for (int t = 0; t < NTH; ++t) {
                    const float u = acc[t][r] + b1v[t];
                    const float v = gelu_f(u);
                    float gvs = 0.f;
#pragma unroll
                    for (int j = 0; j < DOT; ++j) {
                        gvs += g[j] * w2r[j][t];
                        dw2[j][t] += g[j] * v;
                    }
                    const float guv = gvs * gelu_grad_f(u);
                    if (ok) {
                        p.gu[q * HID + t * 32 + col] = guv;
                        db1[t] += guv;
                    }
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

static int proj_launch(bool bwd, ProjArgs& p, hipStream_t st) {
    RPB_REQUIRE(p.a && p.w1 && p.b1 && p.w2 && p.b2, "proj: null pointer");
    RPB_REQUIRE(p.C % 8 == 0 && p.C >= 8 && p.C <= 256, "proj: C=%d unsupported", p.C);
    RPB_REQUIRE(p.DO >= 1 && p.DO <= 16, "proj: fc2 out features %d not in [1,16]", p.DO);
    RPB_REQUIRE(p.ncrop > 0 && p.ncrop < (1L << 31), "proj: ncrop out of range");
    const int waves = proj_waves(p.C, p.DO);
    RPB_REQUIRE(waves > 0, "proj: does not fit LDS");
    const int grid = (int)(rpb_proj_slots(p.ncrop, p.C, p.DO) / waves);
    const size_t lds = proj_lds(p.C, p.DO, waves);
#define RPB_PJ(B_, D_)                                                                                          \
    {                                                                                                           \
        (void)hipFuncSetAttribute((const void*)proj_kernel<B_, D_>, hipFuncAttributeMaxDynamicSharedMemorySize,      \
                            (int)lds);                                                                          \
        hipLaunchKernelGGL((proj_kernel<B_, D_>), dim3(grid), dim3(waves * 64), lds, st, p);                    \
    }
    if (!bwd) RPB_PJ(false, 1)
    else if (p.DO <= 2) RPB_PJ(true, 2)
    else if (p.DO <= 4) RPB_PJ(true, 4)
    else if (p.DO <= 8) RPB_PJ(true, 8)
    else RPB_PJ(true, 16)
#undef RPB_PJ
    RPB_CHECK_LAUNCH("proj");
}

extern "C" int rpb_proj_fwd(const float* a, const float* w1, const float* b1, const float* w2, const float* b2,
                            float* out, long ncrop, int C, int DO, int T, int H, int W, int Tp, int Hp, int Wp,
                            void* stream) {
    RPB_REQUIRE(out, "proj_fwd: null out");
    ProjArgs p{};
    p.a = a; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.out = out; p.ncrop = ncrop; p.C = C; p.DO = DO;
    p.cm = CropMap{T, H, W, Tp, Hp, Wp};
    return proj_launch(false, p, (hipStream_t)stream);
}

extern "C" int rpb_proj_bwd(const float* a, const float* w1, const float* b1, const float* w2, const float* b2,
                            const float* gout, float* gu, float* part, long ncrop, int C, int DO, int T, int H, int W,
                            int Tp, int Hp, int Wp, void* stream) {
    RPB_REQUIRE(gout && gu && part, "proj_bwd: null pointer");
    ProjArgs p{};
    p.a = a; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.gout = gout; p.gu = gu; p.part = part;
    p.ncrop = ncrop; p.C = C; p.DO = DO;
    p.cm = CropMap{T, H, W, Tp, Hp, Wp};
    return proj_launch(true, p, (hipStream_t)stream);
}
