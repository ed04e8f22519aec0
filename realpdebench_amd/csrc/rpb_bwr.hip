// The fused backward "row" kernel of one Fourier layer at C = 64 on the bf16 matrix pipe ("bwr"; rpb_bwd_row.hip holds the entry points
// and the round-1 fp32-MFMA kernel that still serves C = 32):
//
//   gs   = BatchNorm3d(+GELU) backward apply                                   autograd of fno.py:117-119
//   Y1   = GW^T gs      adjoint of the last inverse-DFT stage (the W stage of the spectral backward)     autograd of fno.py:63
//   dWc += gs^T x,  dbc += sum gs      (x = the layer input, lazily activated; layer 0: the feature fields)   autograd of fno.py:115
//
// All three consumers contract over CELLS with channels as the free index, so one register image serves them: a lane holds 4
// consecutive channels (16 B) of 8 cells -- a "B-layout" load, 4 whole 256 B cell rows = 1 KB contiguous per load instruction, the
// access shape with which the DFT stages already reach the chip's copy ceiling (the round-1 kernel moved 8 B per lane and kept the
// fp32 MFMA pipe 40 % busy at 64 cycles per instruction).  The split planes of gs are the B operand of the Y1 product (A = GW^T
// planes from LDS) AND the A operand of the weight gradient (B = the planes of x): one split each for gs and x per 32-cell step,
// 48 + 96 bf16 MFMAs instead of 96 fp32 ones (6144 -> 2304 matrix-pipe cycles).  One wave per SIMD (accumulators 64 + 32, two steps
// of three input streams in flight: ~96 KB of loads per CU), rows of Wp cells walked in 32-cell steps.
#include "rpb_bwr.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

#define BW_WAVES 4

namespace {
// (cache policy of the streaming loads / stores: RPB_STREAM_AUX, rpb_common.h -- nt by default since round 5)
__device__ __forceinline__ u32x4 ld16(rsrc_t r, int voff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, RPB_STREAM_AUX));
}
__device__ __forceinline__ void st16(f32x4v v, rsrc_t r, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, RPB_STREAM_AUX);
}
__device__ __forceinline__ float trunc_bf16(float v) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_hi(float a, float b) {
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
    u32x4 uh, um, ul;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = v[2 * q], b = v[2 * q + 1];
        {
            unsigned ph_, pm_, pl_;
            rpb_split_pair(a, b, ph_, pm_, pl_);
            uh[q] = ph_;
            um[q] = pm_;
            ul[q] = pl_;
        }
    }
    h = __builtin_bit_cast(bf16x8, uh);
    m = __builtin_bit_cast(bf16x8, um);
    l = __builtin_bit_cast(bf16x8, ul);
}
__device__ __forceinline__ f32x4v mfma16(bf16x8 a, bf16x8 b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
}  // namespace

// six products of the three-plane split, small terms first, four independent accumulation chains advancing together
#define BW_MAC6(ACC, AH, AM, AL, BH, BM, BL)                                            \
    _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) ACC(c_) = mfma16(AH(c_), BL(c_), ACC(c_)); \
    _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) ACC(c_) = mfma16(AL(c_), BH(c_), ACC(c_)); \
    _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) ACC(c_) = mfma16(AM(c_), BM(c_), ACC(c_)); \
    _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) ACC(c_) = mfma16(AH(c_), BM(c_), ACC(c_)); \
    _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) ACC(c_) = mfma16(AM(c_), BH(c_), ACC(c_)); \
    _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) ACC(c_) = mfma16(AH(c_), BH(c_), ACC(c_));

// GELU: gy is multiplied by gelu'(z) here (the producer did not store gz);  XGELU / XBN: the layer input is act(BN(x));  FEAT: x is the
// feature tensor (layer 0): the weight-gradient columns are its FW <= 16 fields;  NOX: no weight gradient here (a.x == null: the
// backward cell_mix of the same layer forms it, csrc/rpb_cmx.hip WG) -- the layer input is not read at all: 12.4 instead of 16.2 GB
// CS: floats per cell row (64; 128 = one 64-channel half of a width-128 layer per launch, NOX only: BwrArgs::CS / coff)
// MT: 16-mode row tiles of Y1 (2: K2 <= 32; 3: K2 <= 48, the Galerkin regressor's modes (4, 16, 20) -> K2 = 40)
template <bool GELU, bool XBN, bool XGELU, bool FEAT, bool NOX = false, int CS = 64, int MT = 2>
__global__ __launch_bounds__(BW_WAVES * 64, 1) void bwr_kernel(BwrArgs a) {
    static_assert(CS == 64 || NOX, "width 128: the launch without the weight gradient");
    extern __shared__ u32x4 lds4[];                     // GW^T planes [q][plane 3][mt MT][lane]: A operand of Y1 = GW^T gs  (rows = mode 16 mt + n16)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kg = lane >> 4;
    const int Wp = a.Wp, K2 = a.K2;
    const int nq = (Wp + 31) >> 5;
    for (int idx = tid; idx < nq * MT * 64; idx += blockDim.x) {
        const int l = idx & 63, mt = (idx >> 6) % MT, q = (idx >> 6) / MT;
        const int o = 16 * mt + (l & 15);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int cell = 32 * q + 4 * e + (l >> 4);
            v[e] = (cell < Wp && o < K2) ? a.GW[cell * K2 + o] : 0.f;
        }
        bf16x8 h, m, lo;
        split8(v, h, m, lo);
        lds4[((q * 3 + 0) * MT + mt) * 64 + l] = __builtin_bit_cast(u32x4, h);
        lds4[((q * 3 + 1) * MT + mt) * 64 + l] = __builtin_bit_cast(u32x4, m);
        lds4[((q * 3 + 2) * MT + mt) * 64 + l] = __builtin_bit_cast(u32x4, lo);
    }
    __syncthreads();

    // per-lane constants of the lane's channels 4 n16 .. 4 n16 + 3
    const int c0 = 4 * n16;
    const f32x4v mu = *reinterpret_cast<const f32x4v*>(a.mean + c0), is = *reinterpret_cast<const f32x4v*>(a.invstd + c0);
    const f32x4v ga = *reinterpret_cast<const f32x4v*>(a.gamma + c0), be = *reinterpret_cast<const f32x4v*>(a.beta + c0);
    const f32x4v m1 = *reinterpret_cast<const f32x4v*>(a.sums + c0) * a.inv_count;
    const f32x4v m2 = *reinterpret_cast<const f32x4v*>(a.sums + CS + c0) * a.inv_count;
    const f32x4v gis = ga * is;
    f32x4v xmu = {0.f, 0.f, 0.f, 0.f}, xis = xmu, xga = xmu, xbe = xmu;
    if (XBN) {
        xmu = *reinterpret_cast<const f32x4v*>(a.xf.mean + c0);
        xis = *reinterpret_cast<const f32x4v*>(a.xf.invstd + c0);
        xga = *reinterpret_cast<const f32x4v*>(a.xf.gamma + c0);
        xbe = *reinterpret_cast<const f32x4v*>(a.xf.beta + c0);
    }
    const f32x4v z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4v accW[4][(FEAT || NOX) ? 1 : 4];               // d conv weight: tile (uo, ui): row 4 mg + r <-> out channel 4 (4 mg + r) + uo, column n16 <-> in channel 4 n16 + ui (FEAT: field n16)
    f32x4v accY[MT][4];                                   // Y1 of the current row: row 16 mt + 4 mg + r = mode, column n16 of tile u <-> channel 4 n16 + u
    f32x4v bsum = z4;
#pragma unroll
    for (int uo = 0; uo < 4; ++uo)
#pragma unroll
        for (int ui = 0; ui < ((FEAT || NOX) ? 1 : 4); ++ui) accW[uo][ui] = z4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int u = 0; u < 4; ++u) accY[mt][u] = z4;

    const long nslots = (long)gridDim.x * BW_WAVES;
    const long slot = (long)blockIdx.x * BW_WAVES + wave;
    constexpr int CB = CS * 4;                           // width 128: rows of 512 B, this launch's 256 B half at byte 4 coff
    const int coff = CS == 64 ? 0 : a.coff;
    const unsigned row_bytes = (unsigned)Wp * (unsigned)CB - 4u * (unsigned)coff;
    const int FW = a.FW;

    // one step = 32 cells of a row: cell (kg, e) = 32 q + 4 e + kg.  Loads of step k + 1 are issued before step k is computed.
    u32x4 sA[8], yA[8], xA[8], sB[8], yB[8], xB[8];
    float fA[8], fB[8];                                  // FEAT: field n16 of the lane's 8 cells
    auto issue = [&](long g, int q, u32x4 (&sv)[8], u32x4 (&yv)[8], u32x4 (&xv)[8], float (&fv)[8]) {
        const bool ok = g < a.G;
        const long off = (ok ? g : 0) * (long)Wp * CS + coff;
        const unsigned nb = ok ? row_bytes : 0u;         // past the wave's last row: an empty descriptor, the loads return 0 without traffic
        const rsrc_t rs = make_rsrc(a.s + off, nb), ry = make_rsrc(a.gy + off, nb);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int vo = (32 * q + 4 * e + kg) * CB + n16 * 16;
            sv[e] = ld16(rs, vo);
            yv[e] = ld16(ry, vo);
        }
        if (NOX) {
        } else if (FEAT) {
            const rsrc_t rx = make_rsrc(a.x + (ok ? g : 0) * (long)Wp * FW, ok ? (unsigned)(Wp * FW) * 4u : 0u);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                fv[e] = buf_load_f32(rx, n16 < FW ? ((32 * q + 4 * e + kg) * FW + n16) * 4 : 0x7ffffff0, 0);
        } else {
            const rsrc_t rx = make_rsrc(a.x + off, nb);
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[e] = ld16(rx, (32 * q + 4 * e + kg) * 256 + n16 * 16);
        }
    };
    auto compute = [&](long g, int q, const u32x4 (&sv)[8], const u32x4 (&yv)[8], const u32x4 (&xv)[8], const float (&fv)[8]) {
        const rsrc_t ro = make_rsrc(a.gs ? a.gs + g * (long)Wp * CS + coff : a.s, a.gs ? row_bytes : 0u);   // gs == NULL: stores dropped
        f32x4v gsv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const f32x4v sh = (__builtin_bit_cast(f32x4v, sv[e]) - mu) * is;
            f32x4v gz = __builtin_bit_cast(f32x4v, yv[e]);
            if (GELU) gz = gz * gelu_grad4(sh * ga + be);
            f32x4v v = gis * ((gz - m1) - sh * m2);
            v = (32 * q + 4 * e + kg < Wp) ? v : z4;     // cells past the row end read zeros, which BatchNorm does not map to zero
            gsv[e] = v;
            st16(v, ro, (32 * q + 4 * e + kg) * CB + n16 * 16);          // past the row end: dropped by the descriptor
            bsum += v;
        }
        bf16x8 Gh[4], Gm[4], Gl[4];                      // gs planes: column / row n16 of tile u <-> channel 4 n16 + u, K = the lane group's 8 cells
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gsv[e][u];
            split8(v, Gh[u], Gm[u], Gl[u]);
        }
        // ---- Y1[mode][channel] += GW^T gs
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const bf16x8 ah = __builtin_bit_cast(bf16x8, lds4[((q * 3 + 0) * MT + mt) * 64 + lane]);
            const bf16x8 am = __builtin_bit_cast(bf16x8, lds4[((q * 3 + 1) * MT + mt) * 64 + lane]);
            const bf16x8 al = __builtin_bit_cast(bf16x8, lds4[((q * 3 + 2) * MT + mt) * 64 + lane]);
#define BW_ACC(c) accY[mt][c]
#define BW_A1(c) ah
#define BW_A2(c) am
#define BW_A3(c) al
#define BW_B1(c) Gh[c]
#define BW_B2(c) Gm[c]
#define BW_B3(c) Gl[c]
            BW_MAC6(BW_ACC, BW_A1, BW_A2, BW_A3, BW_B1, BW_B2, BW_B3)
#undef BW_ACC
#undef BW_A1
#undef BW_A2
#undef BW_A3
#undef BW_B1
#undef BW_B2
#undef BW_B3
        }
        // ---- dWc[out][in] += gs^T x
        if (NOX) {
        } else if (FEAT) {
            bf16x8 Xh, Xm, Xl;
            split8(fv, Xh, Xm, Xl);
#define BW_ACC(c) accW[c][0]
#define BW_A1(c) Gh[c]
#define BW_A2(c) Gm[c]
#define BW_A3(c) Gl[c]
#define BW_B1(c) Xh
#define BW_B2(c) Xm
#define BW_B3(c) Xl
            BW_MAC6(BW_ACC, BW_A1, BW_A2, BW_A3, BW_B1, BW_B2, BW_B3)
#undef BW_ACC
#undef BW_A1
#undef BW_A2
#undef BW_A3
#undef BW_B1
#undef BW_B2
#undef BW_B3
        } else {
            bf16x8 Xh[4], Xm[4], Xl[4];
            f32x4v xt[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                f32x4v xx = __builtin_bit_cast(f32x4v, xv[e]);
                if (XBN) {
                    xx = bn4(xx, xmu, xis, xga, xbe);
                    if (XGELU) xx = gelu4(xx);
                }
                xt[e] = xx;                              // rows past the end carry act(BN(0)) != 0, but gs is 0 there
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = xt[e][u];
                split8(v, Xh[u], Xm[u], Xl[u]);
            }
#pragma unroll
            for (int uo = 0; uo < 4; ++uo) {
#define BW_ACC(c) accW[uo][c]
#define BW_A1(c) Gh[uo]
#define BW_A2(c) Gm[uo]
#define BW_A3(c) Gl[uo]
#define BW_B1(c) Xh[c]
#define BW_B2(c) Xm[c]
#define BW_B3(c) Xl[c]
                BW_MAC6(BW_ACC, BW_A1, BW_A2, BW_A3, BW_B1, BW_B2, BW_B3)
#undef BW_ACC
#undef BW_A1
#undef BW_A2
#undef BW_A3
#undef BW_B1
#undef BW_B2
#undef BW_B3
            }
        }
        if (q == nq - 1) {                               // row complete: Y1[g][mode][channel], 16 B per lane
            float* yp = a.Y1 + g * (long)K2 * CS + coff + 4 * n16;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = 16 * mt + 4 * kg + r;
                    if (o < K2) *reinterpret_cast<f32x4v*>(yp + (long)o * CS) = f32x4v{accY[mt][0][r], accY[mt][1][r], accY[mt][2][r], accY[mt][3][r]};
                }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int u = 0; u < 4; ++u) accY[mt][u] = z4;
        }
    };

    long g = slot;
    int q = 0;
    if (g < a.G) issue(g, 0, sA, yA, xA, fA);
    while (g < a.G) {
        // next step of this wave: the same row's next 32 cells, or the first cells of its next row
        long gn = g;
        int qn = q + 1;
        if (qn == nq) {
            qn = 0;
            gn = g + nslots;
        }
        asm volatile("" ::: "memory");
        issue(gn, qn, sB, yB, xB, fB);
        compute(g, q, sA, yA, xA, fA);
        g = gn;
        q = qn;
        if (g >= a.G) break;
        gn = g;
        qn = q + 1;
        if (qn == nq) {
            qn = 0;
            gn = g + nslots;
        }
        asm volatile("" ::: "memory");
        issue(gn, qn, sA, yA, xA, fA);
        compute(g, q, sB, yB, xB, fB);
        g = gn;
        q = qn;
    }

    // ---- the wave's partial row: [64 out][64 in] (FEAT: columns 0 .. FW-1), then [64] sum gs
    float* part = a.part + slot * (long)(64 * 64 + 64);
#pragma unroll
    for (int uo = 0; uo < 4; ++uo)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = 4 * (4 * kg + r) + uo;
            if (NOX) {
            } else if (FEAT) {
                if (n16 < FW) part[o * 64 + n16] = accW[uo][0][r];
            } else {
                *reinterpret_cast<f32x4v*>(part + o * 64 + 4 * n16) =
                    f32x4v{accW[uo][0][r], accW[uo][FEAT ? 0 : 1][r], accW[uo][FEAT ? 0 : 2][r], accW[uo][FEAT ? 0 : 3][r]};
            }
        }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float b = bsum[c];
        b += __shfl_xor(b, 16, 64);
        b += __shfl_xor(b, 32, 64);
        if (kg == 0) part[64 * 64 + 4 * n16 + c] = b;
    }
}

static bool bwr_off() {
    static const bool off = getenv("RPB_BWD_ROW_F32") && atoi(getenv("RPB_BWD_ROW_F32")) == 1;        // round-1 fp32-MFMA kernel
    return off;
}

// K2 in 33 .. 48: three 16-mode row tiles, the width-128 half launches only (rpb_bwr_supported_c128)
bool rpb_bwr_supported_c128(int Wp, int K2) {
    return rpb_bwr_supported(64, Wp, K2 <= 32 ? K2 : 32, 0) && K2 >= 1 && K2 <= 48 && (size_t)((Wp + 31) / 32) * 9 * 1024 <= 150 * 1024;
}
bool rpb_bwr_supported(int C, int Wp, int K2, int FW) {
    return !bwr_off() && C == 64 && K2 >= 1 && K2 <= 32 && Wp >= 1 && (long)Wp * 256 < (1L << 30) && FW >= 0 && FW <= 16 &&
           (size_t)((Wp + 31) / 32) * 6 * 1024 <= 150 * 1024;
}

long rpb_bwr_slots(int G) {
    long grid = rpb_num_cus();
    const long need = ((long)G + BW_WAVES - 1) / BW_WAVES;
    if (grid > need) grid = need;
    return grid * BW_WAVES;
}

// part has `part_rows` rows (what rpb_bn_bwd_row_slots promised the caller); rows beyond this launch's waves are zeroed
int rpb_bwr_launch(const BwrArgs& a, long part_rows, hipStream_t st) {
    const long slots = rpb_bwr_slots(a.G);
    RPB_REQUIRE(slots <= part_rows, "bn_bwd_row (bf16 pipe): %ld partial rows needed, %ld allocated", slots, part_rows);
    if (part_rows > slots)
        (void)hipMemsetAsync(a.part + slots * (64 * 64 + 64), 0, (size_t)(part_rows - slots) * (64 * 64 + 64) * 4, st);
    const int grid = (int)(slots / BW_WAVES);
    RPB_REQUIRE(a.CS == 64 || (a.CS == 128 && !a.x && (a.coff == 0 || a.coff == 64)), "bn_bwd_row (bf16 pipe): row stride %d", a.CS);
    const int mt = a.K2 > 32 ? 3 : 2;
    RPB_REQUIRE(mt == 2 || (a.CS == 128 && a.K2 <= 48), "bn_bwd_row (bf16 pipe): K2 = %d", a.K2);
    const size_t lds = (size_t)((a.Wp + 31) / 32) * 3 * mt * 64 * 16;
    const bool gelu = a.gelu != 0, xbn = a.xf.mean != nullptr, xgelu = xbn && a.xf.gelu != 0, feat = a.FW > 0, nox = a.x == nullptr;
    if (nox && a.CS == 128 && mt == 3) {
        (void)hipFuncSetAttribute((const void*)bwr_kernel<true, false, false, false, true, 128, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)bwr_kernel<false, false, false, false, true, 128, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (gelu) hipLaunchKernelGGL((bwr_kernel<true, false, false, false, true, 128, 3>), dim3(grid), dim3(BW_WAVES * 64), lds, st, a);
        else hipLaunchKernelGGL((bwr_kernel<false, false, false, false, true, 128, 3>), dim3(grid), dim3(BW_WAVES * 64), lds, st, a);
        RPB_CHECK_LAUNCH("bn_bwd_row (bf16 pipe, width 128, one 64-channel half, K2 <= 48)");
    }
    if (nox && a.CS == 128) {
        (void)hipFuncSetAttribute((const void*)bwr_kernel<true, false, false, false, true, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)bwr_kernel<false, false, false, false, true, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (gelu) hipLaunchKernelGGL((bwr_kernel<true, false, false, false, true, 128>), dim3(grid), dim3(BW_WAVES * 64), lds, st, a);
        else hipLaunchKernelGGL((bwr_kernel<false, false, false, false, true, 128>), dim3(grid), dim3(BW_WAVES * 64), lds, st, a);
        RPB_CHECK_LAUNCH("bn_bwd_row (bf16 pipe, width 128, one 64-channel half)");
    }
    if (nox) {
        (void)hipFuncSetAttribute((const void*)bwr_kernel<true, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)bwr_kernel<false, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (gelu) hipLaunchKernelGGL((bwr_kernel<true, false, false, false, true>), dim3(grid), dim3(BW_WAVES * 64), lds, st, a);
        else hipLaunchKernelGGL((bwr_kernel<false, false, false, false, true>), dim3(grid), dim3(BW_WAVES * 64), lds, st, a);
        RPB_CHECK_LAUNCH("bn_bwd_row (bf16 pipe, no weight gradient)");
    }
#define RPB_BWR(G_, B_, X_, F_)                                                                                                \
    if (gelu == G_ && xbn == B_ && xgelu == X_ && feat == F_) {                                                                \
        (void)hipFuncSetAttribute((const void*)bwr_kernel<G_, B_, X_, F_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((bwr_kernel<G_, B_, X_, F_>), dim3(grid), dim3(BW_WAVES * 64), lds, st, a);                          \
    }
    RPB_BWR(false, false, false, false) RPB_BWR(true, false, false, false)
    RPB_BWR(false, true, false, false) RPB_BWR(true, true, false, false)
    RPB_BWR(false, true, true, false) RPB_BWR(true, true, true, false)
    RPB_BWR(false, false, false, true) RPB_BWR(true, false, false, true)
#undef RPB_BWR
    RPB_CHECK_LAUNCH("bn_bwd_row (bf16 pipe)");
}
