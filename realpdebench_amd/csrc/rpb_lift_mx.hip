// lift + zero-pad with bf16 OUTPUT for wide inputs on the matrix pipe (BASELINE.json configs[4]: the combustion volume, C_in = 16):
//   out[b,t,h,w,:] = bf16( fc0.weight @ [x[b,t,h,w,:], gt[t], gh[h], gw[w]] + fc0.bias )  inside T x H x W, 0 in the pad
// (fno.py:106-111: get_grid, cat, fc0, permute, F.pad), the instance rpb_lift_pad_fwd_bf16 dispatches to at C_in = 16, C = 64.
//
// Why: the vector kernel (rpb_pointwise.hip, lift_pad_kernel<19>) runs this shape at 2.1 TB/s (0.46 ms of a 4.85 ms forward): a block
// stages one (b,t,h) row in LDS between two barriers and every thread then issues 64 FMAs per cell from LDS reads -- neither FMA-bound
// (packed FMAs measured no change) nor byte-bound, it waits on itself.  As a GEMM the lift is one K-step: K = 32 features per cell
// (16 inputs, 3 coordinates, a constant 1 that carries the bias, 12 zeros) x 64 outputs.
//
//  * a wave walks whole (b,t,h) lines of the padded tensor; a wave tile = 32 consecutive cells = two 16-row MFMA tiles.  Lane
//    (m = lane & 15, kg = lane >> 4) holds features 8 kg .. 8 kg + 7 of cell m: lane groups 0 / 1 load 32 B of the cell's inputs each
//    (every load instruction reads 16 whole 64 B cell rows), group 2 forms (gt, gh, gw[w], 1, 0, 0, 0, 0), group 3 is zero;
//  * both operands as three bf16 planes, six products (fp32 grade, rpb_cmx.hip): 48 MFMAs per 32 cells hide under the tile's 6 KB of HBM
//    traffic, and the result equals the fp32 lift's to ~1e-7, i.e. the stored bf16 values differ from round(fp32 lift) only where the fp32
//    result sits within ~1e-5 of a rounding boundary;
//  * the MFMA column n of output tile t stands for channel 4 n + t (as rpb_cmx.hip): a lane's accumulators are 4 consecutive channels of
//    4 cells -> one 8 B store per (tile, row) = 16 whole 128 B cell rows per instruction;
//  * cells w >= W of an inside line and all cells of a pad line are written as zeros (the consumers read the whole padded tensor).
#include "rpb_common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

namespace {
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
    u32x4 uh, um, ul;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned ph, pm, pl;
        rpb_split_pair(v[2 * q], v[2 * q + 1], ph, pm, pl);
        uh[q] = ph;
        um[q] = pm;
        ul[q] = pl;
    }
    h = __builtin_bit_cast(bf16x8, uh);
    m = __builtin_bit_cast(bf16x8, um);
    l = __builtin_bit_cast(bf16x8, ul);
}
__device__ __forceinline__ f32x4v mfma16(bf16x8 a, bf16x8 b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
}  // namespace

struct LiftMxArgs {
    const float* x;        // [B][T][H][W][16]
    const float *gt, *gh, *gw;
    const float* w0;       // fc0.weight [64][19]
    const float* b0;       // [64]
    void* out;             // bf16 [B][Tp][Hp][Wp][64]
    int B, T, H, W, Tp, Hp, Wp;
};

#define LMX_WAVES 4
#define LMX_CIN 16
#define LMX_F 19

__global__ __launch_bounds__(LMX_WAVES * 64) void lift_mx_kernel(LiftMxArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int m = lane & 15, kg = lane >> 4;
    // ---- B operand: W0ext[co = 4 n + t][k = 8 kg + e] (k < 16: fc0.weight, 16 .. 18: the coordinate columns, 19: the bias, else 0), three planes
    bf16x8 Bh[4], Bm[4], Bl[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int co = 4 * m + t;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 8 * kg + e;
            v[e] = k < LMX_F ? a.w0[co * LMX_F + k] : (k == LMX_F ? a.b0[co] : 0.f);
        }
        split8(v, Bh[t], Bm[t], Bl[t]);
    }
    const long nlines = (long)a.B * a.Tp * a.Hp;
    const long nslots = (long)gridDim.x * LMX_WAVES;
    const unsigned out_line_bytes = (unsigned)a.Wp * 128u;
    const unsigned x_line_bytes = (unsigned)a.W * (LMX_CIN * 4u);
    const int TQ = (a.Wp + 31) >> 5;                     // tiles per padded line (cells >= W store zeros, cells >= Wp are dropped)
    const int TQX = (a.W + 31) >> 5;                     // tiles that hold real cells
    for (long g = (long)blockIdx.x * LMX_WAVES + wave; g < nlines; g += nslots) {
        const int h = (int)(g % a.Hp);
        const long r2 = g / a.Hp;
        const int t = (int)(r2 % a.Tp);
        const long b = r2 / a.Tp;
        const rsrc_t ro = make_rsrc(reinterpret_cast<const char*>(a.out) + g * (long)out_line_bytes, out_line_bytes);
        const bool inside = h < a.H && t < a.T;          // uniform
        if (!inside) {                                   // a pad line: zeros, 1 KB per instruction
            for (int off = lane * 16; off < (int)out_line_bytes; off += 64 * 16)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, ro, off, 0, RPB_STREAM_AUX);
            continue;
        }
        const rsrc_t rx = make_rsrc(a.x + (((b * a.T + t) * a.H + h) * (long)a.W) * LMX_CIN, x_line_bytes);
        const float gtv = a.gt[t], ghv = a.gh[h];
        u32x4 xa[2][2];
        if (kg < 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                xa[j][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (16 * j + m) * 64 + kg * 32, 0, RPB_STREAM_AUX));
                xa[j][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (16 * j + m) * 64 + kg * 32 + 16, 0, RPB_STREAM_AUX));
            }
        }
        for (int q = 0; q < TQ; ++q) {
            if (q >= TQX) {                              // the tile lies in the w pad: zeros (cells >= Wp dropped by the descriptor)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, ro, (32 * q + 16 * j + 4 * kg + r) * 128 + m * 8, 0, RPB_STREAM_AUX);
                continue;
            }
            bf16x8 Ah[2], Am[2], Al[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int w = 32 * q + 16 * j + m;
                float v[8];
                if (kg < 2) {
                    const f32x4v v0 = __builtin_bit_cast(f32x4v, xa[j][0]), v1 = __builtin_bit_cast(f32x4v, xa[j][1]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = v0[e];
                        v[4 + e] = v1[e];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = 0.f;
                    if (kg == 2) {
                        v[0] = gtv;
                        v[1] = ghv;
                        v[2] = w < a.W ? a.gw[w] : 0.f;
                        v[3] = 1.0f;
                    }
                }
                split8(v, Ah[j], Am[j], Al[j]);
            }
            if (kg < 2 && q + 1 < TQX) {                 // the next tile's inputs: in flight during the products and the stores
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    xa[j][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (32 * (q + 1) + 16 * j + m) * 64 + kg * 32, 0, RPB_STREAM_AUX));
                    xa[j][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (32 * (q + 1) + 16 * j + m) * 64 + kg * 32 + 16, 0, RPB_STREAM_AUX));
                }
            }
            f32x4v acc[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    f32x4v c = {0.f, 0.f, 0.f, 0.f};
                    c = mfma16(Ah[j], Bl[tt], c);
                    c = mfma16(Al[j], Bh[tt], c);
                    c = mfma16(Am[j], Bm[tt], c);
                    c = mfma16(Ah[j], Bm[tt], c);
                    c = mfma16(Am[j], Bh[tt], c);
                    c = mfma16(Ah[j], Bh[tt], c);
                    acc[j][tt] = c;
                }
            // ---- cell 32 q + 16 j + 4 kg + r, channels 4 m .. 4 m + 3: round to nearest even, 8 B per lane
            typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
            typedef float f32x2v __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cell = 32 * q + 16 * j + 4 * kg + r;
                    u32x2 pk;
                    pk[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{acc[j][0][r], acc[j][1][r]}, bf16x2v));
                    pk[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{acc[j][2][r], acc[j][3][r]}, bf16x2v));
                    if (cell >= a.W) pk = u32x2{0u, 0u};        // the w pad
                    __builtin_amdgcn_raw_buffer_store_b64(pk, ro, cell * 128 + m * 8, 0, RPB_STREAM_AUX);
                }
        }
    }
}

bool rpb_lift_mx_supported(int Cin, int C) {
    static const bool off = getenv("RPB_LIFT_MX") && atoi(getenv("RPB_LIFT_MX")) == 0;
    return !off && Cin == LMX_CIN && C == 64;
}

int rpb_lift_mx_launch(const float* x, const float* gt, const float* gh, const float* gw, const float* w0, const float* b0, void* out_bf16,
                       int B, int T, int H, int W, int Tp, int Hp, int Wp, hipStream_t st) {
    RPB_REQUIRE((long)Wp * 128 < (1l << 31) && (long)W * 64 < (1l << 31), "lift (matrix pipe): line too long");
    LiftMxArgs a{x, gt, gh, gw, w0, b0, out_bf16, B, T, H, W, Tp, Hp, Wp};
    const long nlines = (long)B * Tp * Hp;
    long grid = (long)rpb_num_cus() * 4;
    const long need = (nlines + LMX_WAVES - 1) / LMX_WAVES;
    if (grid > need) grid = need;
    hipLaunchKernelGGL(lift_mx_kernel, dim3((unsigned)grid), dim3(LMX_WAVES * 64), 0, st, a);
    RPB_CHECK_LAUNCH("lift_pad (matrix pipe, bf16 out)");
}
