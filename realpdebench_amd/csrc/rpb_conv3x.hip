// 3x3x3 convolution (padding 1) as an implicit GEMM on the bf16 MFMA with fp32-grade accuracy ("bf16x3 split"):
// every fp32 operand x is written as hi + mid + lo with three bf16 numbers (8 + 8 + 8 significand bits = all 24 of x; the
// residuals are exact in fp32), and a product a*b is accumulated in fp32 from the six bf16 x bf16 products
//   hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi          (dropped: mid*lo, lo*mid, lo*lo <= 2^-23 |a b|)
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of v_mfma_f32_32x32x2_f32, so six of them cost 0.375 of the fp32 MFMA the
// first version of these convolutions used (122 TF/s = the fp32 pipe's sustained rate; they are 70-78 % of the U-Net and
// Transolver steps).  Replaces nn.Conv3d(Ci, Co, 3, padding=1) forward and its data gradient (flipped taps) --
// Physics_Attention.py:154-157, unet.py:196,201 -- behind rpb_conv3x; the exact-fp32 path (rpb_gemm_nt conv mode 1) stays.
//
// Layout.  Activations are split once per tensor into three bf16 planes P[3][M][Ci] (rpb_split3).  Weights are split and
// stored in MFMA B-operand order (rpb_conv3x_wprep): for every (tap, 16-channel chunk, plane, 32-wide co tile) the 64 lanes'
// 16 B operands are contiguous (one coalesced 1 KB load straight into registers; the 21 MB stay L2 / MALL resident).
// A workgroup owns 128 consecutive tokens; for each (kt, kh) and 64-channel chunk it stages the 130 rows
// [m0 - 1 + shift, m0 + 129 + shift) of the three planes in LDS (full 128 B lines from HBM; rows whose t / h neighbour falls
// outside the mesh are not loaded), which serves the three kw taps (LDS row offset 0 / 1 / 2; the w boundary is a property
// of the OUTPUT token and is applied to the A operand).  LDS order [plane][16-ch chunk][k half][row][8 ch]: the A operand of
// lane (row i, half) is one ds_read_b128 and 32 lanes read 512 contiguous bytes.  Each of the 4 waves computes a
// 128 token x 64 co tile (4 x 2 MFMA tiles, 128 accumulator registers); waves are spread over co first (WN of them) and the
// remaining factor KS = 4 / WN splits the 16-channel chunks of a stage, reduced through LDS at the end.
#include "rpb_common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // native vector: selects stay in registers (HIP's uint4 struct did not)
template <int V>
struct IC {
    static constexpr int value = V;
};

#define CX_BM 128
#define CX_ROWS (CX_BM + 2)

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// round-to-nearest-even bf16 of x (as the upper 16 bits of an fp32)
__device__ __forceinline__ unsigned bf16_rne(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = bf16_rne(x);
    const float r1 = x - __builtin_bit_cast(float, h << 16);          // exact
    m = bf16_rne(r1);
    const float r2 = r1 - __builtin_bit_cast(float, m << 16);         // exact
    l = bf16_rne(r2);
}

// ---------------------------------------------------------------------------------- activation planes
// x [M][ldx] fp32 (C channels used) -> P[3][M][C] bf16; one thread = 8 channels (32 B in, 3 x 16 B out)
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, uint16_t* __restrict__ P, long M, int C,
                                                     int ldx) {
    const int c8n = C >> 3;
    const long total = M * c8n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long m = idx / c8n;
        const int c8 = (int)(idx - m * c8n);
        const f32x4 v0 = RPB_SLD4(x + m * ldx + c8 * 8);                 // (the planes are re-read nine times by the convolution: default policy)
        const f32x4 v1 = RPB_SLD4(x + m * ldx + c8 * 8 + 4);
        unsigned h[8], md[8], lo[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            split3(v0[i], h[i], md[i], lo[i]);
            split3(v1[i], h[4 + i], md[4 + i], lo[4 + i]);
        }
        uint4 oh, om, ol;
        oh.x = h[0] | (h[1] << 16); oh.y = h[2] | (h[3] << 16); oh.z = h[4] | (h[5] << 16); oh.w = h[6] | (h[7] << 16);
        om.x = md[0] | (md[1] << 16); om.y = md[2] | (md[3] << 16); om.z = md[4] | (md[5] << 16); om.w = md[6] | (md[7] << 16);
        ol.x = lo[0] | (lo[1] << 16); ol.y = lo[2] | (lo[3] << 16); ol.z = lo[4] | (lo[5] << 16); ol.w = lo[6] | (lo[7] << 16);
        const long o = m * C + c8 * 8;
        *reinterpret_cast<uint4*>(P + o) = oh;
        *reinterpret_cast<uint4*>(P + M * C + o) = om;
        *reinterpret_cast<uint4*>(P + 2 * M * C + o) = ol;
    }
}

extern "C" int rpb_split3(const float* x, void* planes, long M, int C, int ldx, void* stream) {
    RPB_REQUIRE(x && planes && M > 0 && C > 0 && C % 8 == 0 && ldx % 4 == 0 && ldx >= C, "split3: bad arguments (C=%d ldx=%d)", C, ldx);
    const long total = M * (C / 8);
    long grid = (total + 255) / 256;
    const long cap = (long)rpb_num_cus() * 16;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)planes, M, C, ldx);
    RPB_CHECK_LAUNCH("split3");
}

// ---------------------------------------------------------------------------------- weights in B-operand order
// W [N][27 * Ci] fp32 (tap-major, channel-minor rows: the layout rpb_gemm_nt conv mode 1 takes) ->
// Wz[tap][Ci/16][3 planes][N/32][64 lanes][8] bf16, lane = (co & 31) + 32 * k-half, element e <-> ci = 16 cc + 8 half + e
__global__ __launch_bounds__(256) void conv3x_wprep_kernel(const float* __restrict__ W, uint16_t* __restrict__ Wz, int N, int Ci) {
    const int NT = N >> 5, NCC = Ci >> 4;
    const long total = 27L * NCC * NT * 64;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    long r = idx >> 6;
    const int nt = (int)(r % NT);
    r /= NT;
    const int cc = (int)(r % NCC);
    const int tap = (int)(r / NCC);
    const int co = nt * 32 + (lane & 31), half = lane >> 5;
    const float* src = W + (long)co * 27 * Ci + (long)tap * Ci + cc * 16 + half * 8;
    unsigned h[8], md[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) split3(src[e], h[e], md[e], lo[e]);
    uint4 o[3];
    o[0].x = h[0] | (h[1] << 16); o[0].y = h[2] | (h[3] << 16); o[0].z = h[4] | (h[5] << 16); o[0].w = h[6] | (h[7] << 16);
    o[1].x = md[0] | (md[1] << 16); o[1].y = md[2] | (md[3] << 16); o[1].z = md[4] | (md[5] << 16); o[1].w = md[6] | (md[7] << 16);
    o[2].x = lo[0] | (lo[1] << 16); o[2].y = lo[2] | (lo[3] << 16); o[2].z = lo[4] | (lo[5] << 16); o[2].w = lo[6] | (lo[7] << 16);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const long dst = (((((long)tap * NCC + cc) * 3 + p) * NT + nt) * 64 + lane) * 8;
        *reinterpret_cast<uint4*>(Wz + dst) = o[p];
    }
}

extern "C" int rpb_conv3x_wprep(const float* W, void* Wz, int N, int Ci, void* stream) {
    RPB_REQUIRE(W && Wz && N > 0 && N % 32 == 0 && Ci > 0 && Ci % 16 == 0, "conv3x_wprep: N=%d Ci=%d unsupported", N, Ci);
    const long total = 27L * (Ci / 16) * (N / 32) * 64;
    hipLaunchKernelGGL(conv3x_wprep_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W,
                       (uint16_t*)Wz, N, Ci);
    RPB_CHECK_LAUNCH("conv3x_wprep");
}

// ---------------------------------------------------------------------------------- the convolution
struct Conv3xArgs {
    const uint16_t* P;     // [3][M][Ci] bf16 planes of the input tokens
    const uint16_t* Wz;    // B-operand order, see above
    const float* bias;     // [N] or null
    float* out;            // [M][ldo]
    long M;
    int N, Ci, ldo, T, H, W;
};

template <int WN>
__global__ __launch_bounds__(256, 1) void conv3x_kernel(Conv3xArgs a) {
    constexpr int KS = 4 / WN;
    extern __shared__ u32x4 lds4[];
    // two stage buffers of [3 planes][4 chunks][2 halves][CX_ROWS] x 16 B, then [9][CX_ROWS] row validity per (kt, kh)
    unsigned char* rv = reinterpret_cast<unsigned char*>(lds4 + 2 * 24 * CX_ROWS);
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = wave % WN, kp = wave / WN;
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so XCD x walks the
    // contiguous tile range [x * chunk, (x + 1) * chunk): the h +- 1 rows a tile stages are its neighbours' own rows and hit L2
    const unsigned chunk = gridDim.x >> 3;
    const long tile = (long)(blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    const long m0 = tile * CX_BM;
    if (m0 >= a.M) return;
    const int n0 = blockIdx.y * (64 * WN) + nw * 64;
    const int NT = a.N >> 5, NCC = a.Ci >> 4;
    const unsigned uT = a.T, uH = a.H, uW = a.W;

    for (int idx = tid; idx < 9 * CX_ROWS; idx += 256) {
        const int g = idx / CX_ROWS, j = idx - g * CX_ROWS;
        const int kt = g / 3, kh = g - kt * 3;
        const long q = m0 - 1 + j + ((long)(kt - 1) * a.H + (kh - 1)) * a.W;
        bool ok = q >= 0 && q < a.M;
        if (ok) {
            const unsigned uq = (unsigned)q, r = uq / uW;
            const int hq = (int)(r % uH), tq = (int)((r / uH) % uT);
            const int tt = tq - kt + 1, hh = hq - kh + 1;
            ok = tt >= 0 && tt < a.T && hh >= 0 && hh < a.H;
        }
        rv[idx] = ok ? 1 : 0;
    }
    bool wlo[4], whi[4];                                                // my output tokens at the w boundaries
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
        const unsigned w = (unsigned)(m0 + tm * 32 + col) % uW;
        wlo[tm] = (w == 0);
        whi[tm] = (w == uW - 1);
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) acc[tm][0] = acc[tm][1] = zero16();
    const u32x4 z4 = {0u, 0u, 0u, 0u};
    const long MC = a.M * a.Ci;

    // One workgroup per CU, one wave per SIMD with the whole 512-register file, everything software-pipelined by hand:
    //   * the next stage's 13 A loads per thread are issued (unconditionally: invalid rows read a dummy address and are zeroed on
    //     the way into LDS) before the current stage's MFMAs and land in the other LDS buffer afterwards -- one barrier per stage;
    //   * the B operands of the NEXT tap step (kw, then the next 16-channel chunk / stage) are requested before the 48 MFMAs of
    //     the current one (left to itself the compiler put every load next to its use: an L2 latency per tap, 57 % MFMA busy).
    constexpr int SPS = 4 / KS;                                        // my 16-channel chunks per stage
    const int nc64 = a.Ci >> 6;
    const uint16_t* wbase = a.Wz + ((long)(n0 >> 5) * 64 + lane) * 8;
    const long wplane = (long)NT * 512;                                // bf16 elements between the planes of one (tap, chunk)
    const long wchunk = 3 * wplane, wtap = (long)NCC * wchunk;
    auto bload = [&](const uint16_t* src, u32x4 (&b)[2][3]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) b[tn][p] = *reinterpret_cast<const u32x4*>(src + p * wplane + tn * 512);
    };
    u32x4 bc[2][3], bn[2][3];
    const u32x4* As = lds4;
    u32x4 ac[3], an[3];                                                 // A operands of the current / next 32-token row tile
    auto lda = [&](int s, int kw, int tm, u32x4 (&av)[3]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p) av[p] = As[((p * 4 + s) * 2 + half) * CX_ROWS + tm * 32 + col + kw];
    };
    // one tap step = 4 row tiles x 12 MFMAs; the next tile's A operands (the next step's first tile when `more`) are read from
    // LDS before the current tile's MFMAs are issued -- with one wave per SIMD nobody else hides that latency
    // (the A tiles ping-pong between ac and an -- four tiles per step, so every step starts on ac -- and the B sets between bc and
    // bn by step parity: no register copies)
    auto mfma12 = [&](int tm, int kw, u32x4 (&av)[3], const u32x4 (&b)[2][3]) __attribute__((always_inline)) {
        if ((kw == 0 && wlo[tm]) || (kw == 2 && whi[tm])) av[0] = av[1] = av[2] = z4;
        const bf16x8 ah = __builtin_bit_cast(bf16x8, av[0]), am = __builtin_bit_cast(bf16x8, av[1]),
                     al = __builtin_bit_cast(bf16x8, av[2]);
        const bf16x8 b0h = __builtin_bit_cast(bf16x8, b[0][0]), b0m = __builtin_bit_cast(bf16x8, b[0][1]),
                     b0l = __builtin_bit_cast(bf16x8, b[0][2]), b1h = __builtin_bit_cast(bf16x8, b[1][0]),
                     b1m = __builtin_bit_cast(bf16x8, b[1][1]), b1l = __builtin_bit_cast(bf16x8, b[1][2]);
        // small terms first; the two co tiles alternate so consecutive MFMAs are independent
        acc[tm][0] = mfma_bf16(al, b0h, acc[tm][0]);
        acc[tm][1] = mfma_bf16(al, b1h, acc[tm][1]);
        acc[tm][0] = mfma_bf16(ah, b0l, acc[tm][0]);
        acc[tm][1] = mfma_bf16(ah, b1l, acc[tm][1]);
        acc[tm][0] = mfma_bf16(am, b0m, acc[tm][0]);
        acc[tm][1] = mfma_bf16(am, b1m, acc[tm][1]);
        acc[tm][0] = mfma_bf16(am, b0h, acc[tm][0]);
        acc[tm][1] = mfma_bf16(am, b1h, acc[tm][1]);
        acc[tm][0] = mfma_bf16(ah, b0m, acc[tm][0]);
        acc[tm][1] = mfma_bf16(ah, b1m, acc[tm][1]);
        acc[tm][0] = mfma_bf16(ah, b0h, acc[tm][0]);
        acc[tm][1] = mfma_bf16(ah, b1h, acc[tm][1]);
    };
    auto tap_step = [&](int s, int kw, bool more, const u32x4 (&b)[2][3]) __attribute__((always_inline)) {   // kw: compile-time constant
        lda(s, kw, 1, an);
        __builtin_amdgcn_sched_barrier(0);
        mfma12(0, kw, ac, b);
        __builtin_amdgcn_sched_barrier(0);
        lda(s, kw, 2, ac);
        __builtin_amdgcn_sched_barrier(0);
        mfma12(1, kw, an, b);
        __builtin_amdgcn_sched_barrier(0);
        lda(s, kw, 3, an);
        __builtin_amdgcn_sched_barrier(0);
        mfma12(2, kw, ac, b);
        __builtin_amdgcn_sched_barrier(0);
        if (more) lda(kw == 2 ? s + KS : s, kw == 2 ? 0 : kw + 1, 0, ac);
        __builtin_amdgcn_sched_barrier(0);
        mfma12(3, kw, an, b);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto bsel = [&](auto pc) -> u32x4(&)[2][3] {
        if constexpr (decltype(pc)::value) return bn;
        else return bc;
    };
    // A staging registers: 13 x 16 B per thread, named individually (as an array indexed from helper lambdas they were demoted to
    // scratch memory: load, wait, spill -- one exposed HBM latency per load)
    constexpr int NLD = (CX_ROWS * 24 + 255) / 256;
    static_assert(NLD == 13, "staging macros below are written for 13 loads per thread");
#define CX_FOR13(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12)
#define CX_DECL(J) u32x4 sv##J = z4; bool ok##J = false;
    CX_FOR13(CX_DECL)
#define CX_LOAD(J)                                                                                            \
    if constexpr (J >= j0 && J < j1) {                                                                        \
        const int idx = tid + J * 256;                                                                        \
        const int row = idx / 24, rem = idx - row * 24;                                                       \
        const int p = rem >> 3, sh = rem & 7; /* sh = 2 * chunk + half: 8 x 16 B = one 128 B line */          \
        ok##J = idx < CX_ROWS * 24 && rv[g * CX_ROWS + (row < CX_ROWS ? row : 0)];                            \
        const long off = ok##J ? (long)p * MC + (rowbase + row) * a.Ci + c * 64 + sh * 8 : 0;                 \
        sv##J = *reinterpret_cast<const u32x4*>(a.P + off);                                                   \
    }
#define CX_STORE(J)                                                                                           \
    {                                                                                                         \
        const int idx = tid + J * 256;                                                                        \
        const int row = idx / 24, rem = idx - row * 24;                                                       \
        if (idx < CX_ROWS * 24) dst[rem * CX_ROWS + row] = ok##J ? sv##J : z4;                                \
    }
    // loads j0 <= j < j1 of the 13 a thread contributes to stage (g, c)
    auto stage_load = [&](int g, int c, auto j0c, auto j1c) __attribute__((always_inline)) {
        constexpr int j0 = decltype(j0c)::value, j1 = decltype(j1c)::value;
        const int kt = g / 3, kh = g - kt * 3;
        const long rowbase = m0 - 1 + ((long)(kt - 1) * a.H + (kh - 1)) * a.W;
        CX_FOR13(CX_LOAD)
    };
    auto stage_store = [&](u32x4* dst) __attribute__((always_inline)) { CX_FOR13(CX_STORE) };
    bload(wbase + (long)kp * wchunk, bc);                               // (g 0, c 0, s = kp, kw 0)
    stage_load(0, 0, IC<0>{}, IC<NLD>{});
    stage_store(lds4);
    __syncthreads();
    int buf = 0;
    constexpr int NST = 3 * SPS;                                        // tap steps per stage
    for (int g = 0; g < 9; ++g) {
        for (int c = 0; c < nc64; ++c) {
            int gn = g, cn = c + 1;                                     // next stage
            if (cn == nc64) {
                cn = 0;
                ++gn;
            }
            const bool more = gn < 9;
            As = lds4 + buf * 24 * CX_ROWS;
            lda(kp, 0, 0, ac);
            // the next stage's A loads are spread over this stage's tap steps (13 MB at once from every CU of the chip in
            // lock-step is a burst that the in-order vmcnt of the next B operands would have to wait out)
#define CX_SI_BLOCK(SI)                                                                                        \
    if constexpr (SI < SPS) {                                                                                  \
        const int s = kp + SI * KS;                                                                            \
        const uint16_t* w0 = wbase + (long)(g * 3) * wtap + (long)(c * 4 + s) * wchunk;                        \
        bload(w0 + wtap, bsel(IC<(SI * 3 + 1) & 1>{}));                                                        \
        if (more) stage_load(gn, cn, IC<(SI * 3 + 0) * NLD / NST>{}, IC<(SI * 3 + 1) * NLD / NST>{});          \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        tap_step(s, 0, true, bsel(IC<(SI * 3 + 0) & 1>{}));                                                    \
        bload(w0 + 2 * wtap, bsel(IC<(SI * 3 + 2) & 1>{}));                                                    \
        if (more) stage_load(gn, cn, IC<(SI * 3 + 1) * NLD / NST>{}, IC<(SI * 3 + 2) * NLD / NST>{});          \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        tap_step(s, 1, true, bsel(IC<(SI * 3 + 1) & 1>{}));                                                    \
        /* the step after (s, kw 2): next chunk of this stage, else the next stage's first chunk */            \
        if (SI + 1 < SPS) bload(w0 + (long)KS * wchunk, bsel(IC<(SI * 3 + 3) & 1>{}));                         \
        else if (more) bload(wbase + (long)(gn * 3) * wtap + (long)(cn * 4 + kp) * wchunk, bsel(IC<(SI * 3 + 3) & 1>{})); \
        if (more) stage_load(gn, cn, IC<(SI * 3 + 2) * NLD / NST>{}, IC<(SI * 3 + 3) * NLD / NST>{});          \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        tap_step(s, 2, SI + 1 < SPS, bsel(IC<(SI * 3 + 2) & 1>{}));                                            \
    }
            CX_SI_BLOCK(0) CX_SI_BLOCK(1) CX_SI_BLOCK(2) CX_SI_BLOCK(3)
            if constexpr (NST & 1) {                                    // odd step count (N = 64): the next stage's first B set is in bn
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int p = 0; p < 3; ++p) bc[tn][p] = bn[tn][p];
            }
            if (more) stage_store(lds4 + (buf ^ 1) * 24 * CX_ROWS);
            __syncthreads();                                            // everyone is done with buf and has filled buf ^ 1
            buf ^= 1;
        }
    }
#undef CX_SI_BLOCK
#undef CX_STORE
#undef CX_LOAD
#undef CX_DECL
#undef CX_FOR13
    // ---- K-split partial sums through LDS in ONE round: the KS waves of a column group own the four 32-token row tiles round-robin
    // (tile tm belongs to wave tm % KS); every wave parks its partials of the tiles it does not own, one barrier, every wave sums
    // and stores its own tiles.  (A first version let the kp == 0 waves reduce and store all four tiles, one tile and two
    // barriers at a time: measured 38 k of the 116 k cycles of a Ci = 64, N = 64 workgroup.)
    float* red = reinterpret_cast<float*>(lds4);                        // [WN][4 tiles][KS - 1 sources][2 col tiles][16 regs][64 lanes]
    if (KS > 1) {
        __syncthreads();                                                // the last stage's LDS reads are done
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            const int owner = tm % KS;
            if (kp != owner) {
                const int rank = kp < owner ? kp : kp - 1;
                float* slot = red + ((nw * 4 + tm) * (KS - 1) + rank) * 2048;
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) slot[(tn * 16 + r) * 64 + lane] = acc[tm][tn][r];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
        if (KS > 1 && kp != tm % KS) continue;
        if (KS > 1) {
#pragma unroll
            for (int k2 = 0; k2 < KS - 1; ++k2) {
                const float* slot = red + ((nw * 4 + tm) * (KS - 1) + k2) * 2048;
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tm][tn][r] += slot[(tn * 16 + r) * 64 + lane];
            }
        }
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int n = n0 + tn * 32 + col;
            const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long m = m0 + tm * 32 + mfma_row(lane, r);
                if (m < a.M) a.out[m * a.ldo + n] = acc[tm][tn][r] + bv;
            }
        }
    }
}

extern "C" int rpb_conv3x(const void* planes, const void* Wz, const float* bias, float* out, long M, int N, int Ci, int ldo,
                          int Hc, int Wc, int Dc, void* stream) {
    RPB_REQUIRE(planes && Wz && out && M > 0 && M < (1L << 31), "conv3x: bad arguments");
    RPB_REQUIRE(Ci % 64 == 0 && (N == 64 || N == 128 || N % 256 == 0) && ldo >= N, "conv3x: N=%d Ci=%d unsupported (Ci %% 64, N = 64, 128 or a multiple of 256)", N, Ci);
    RPB_REQUIRE(Hc > 0 && Wc > 0 && Dc > 0 && M % ((long)Hc * Wc * Dc) == 0, "conv3x: bad mesh");
    Conv3xArgs a{(const uint16_t*)planes, (const uint16_t*)Wz, bias, out, M, N, Ci, ldo, Hc, Wc, Dc};
    const size_t lds = (size_t)2 * 24 * CX_ROWS * 16 + 9 * CX_ROWS + 16;
    const unsigned gx = (unsigned)(((M + CX_BM - 1) / CX_BM + 7) / 8 * 8);
    hipStream_t st = (hipStream_t)stream;
    (void)hipFuncSetAttribute((const void*)conv3x_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)conv3x_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)conv3x_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (N == 64) {
        hipLaunchKernelGGL(conv3x_kernel<1>, dim3(gx, 1), dim3(256), lds, st, a);
    } else if (N == 128) {
        hipLaunchKernelGGL(conv3x_kernel<2>, dim3(gx, 1), dim3(256), lds, st, a);
    } else {
        hipLaunchKernelGGL(conv3x_kernel<4>, dim3(gx, N / 256), dim3(256), lds, st, a);
    }
    RPB_CHECK_LAUNCH("conv3x");
}

// ====================================================================================== weight gradient
// dW[co][tap][ci] = sum_tok G[tok][co] * X[tok + shift(tap)][ci] contracts over TOKENS, so the MFMA operands are runs of 8
// consecutive tokens of one channel: both tensors are split into bf16 planes Pt[3][M/8][C][8] (rpb_split3t).
// Requires W % 8 == 0 (every mesh of the reference's configs): an aligned group of 8 tokens then lies in one w-row, so
//   * the t / h validity of a tap is a property of the whole X group -> invalid groups are zeroed when they are staged;
//   * the w validity touches one element: for kw = 0 the first token of a G group that starts a w-row, for kw = 2 the last token
//     of a group that ends one -> a single v_and on the A operand;
//   * only the kw = 0 / 2 operands are misaligned (by one token): they are assembled from the aligned 16 B and one neighbouring
//     dword with four v_alignbit.
// A workgroup owns a 64 co x 64 ci tile of one kt (9 taps, 4 waves = co half x ci half, 144 accumulator registers each) for one
// token split; grid.x = splits (a multiple of 8: split s always runs on XCD s % 8, so the 3 * Co/64 * Ci/64 tiles that stream the
// same tokens share that XCD's L2), grid.y = tiles.  Chunks of 32 tokens are double-buffered in LDS (conflict-free row strides),
// the next chunk's loads are in flight during the 108 MFMAs per wave of the current one.
#define WX_KT 32
#define WX_GROW 28                       // dwords per G^T row   (48 tok = 24 dwords + 4 pad: conflict-free b128 accesses)
#define WX_XROW 20                       // dwords per X^T row   (32 tok = 16 dwords + 4 pad)
#define WX_GS (3 * 64 * WX_GROW)         // dwords: [3 planes][64 co]
#define WX_XS (9 * 64 * WX_XROW)         // dwords: [3 kh][3 planes][64 ci]
#define WX_BUF (WX_GS + WX_XS)

// x [M][ldx] fp32 -> Pt[3][M/8][C][8] bf16 (runs of 8 tokens per channel, channels next: the 16 B operands of 64 channels are
// 1 KB contiguous, so a wave stages one token group of 64 channels with one coalesced load); block = 64 tokens x 64 channels
// rev: the planes are written in the token order of the REVERSED mesh (d2, d1, d0) -- a 3x3x3 convolution is symmetric in its
// three axes, so a mesh whose innermost dimension is not a multiple of 8 (Transolver's 128 x 64 x 20) is handed to the weight
// gradient kernel with its outermost one innermost, and the taps of the result are transposed back.
__global__ __launch_bounds__(256) void split3t_kernel(const float* __restrict__ x, uint16_t* __restrict__ Pt, long M, int C,
                                                      int ldx, int rev, int d0, int d1, int d2) {
    __shared__ uint16_t tl[3][64][72];                                 // [plane][channel][token], rows padded to 144 B
    const int tid = threadIdx.x;
    const long m0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {                                       // 64 tokens x 16 float4
        const int idx = tid + j * 256, tok = idx >> 4, c4 = (idx & 15) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        long src = m0 + tok;
        if (rev && src < M) {                                           // token (b, i2, i1, i0) of the reversed mesh <- (b, i0, i1, i2)
            const long n = src;
            const int i0 = (int)(n % d0);
            long r = n / d0;
            const int i1 = (int)(r % d1);
            r /= d1;
            const int i2 = (int)(r % d2);
            src = (((r / d2) * d0 + i0) * d1 + i1) * d2 + i2;
        }
        if (m0 + tok < M) v = *reinterpret_cast<const f32x4*>(x + src * ldx + c0 + c4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned h, md, lo;
            split3(v[i], h, md, lo);
            tl[0][c4 + i][tok] = (uint16_t)h;
            tl[1][c4 + i][tok] = (uint16_t)md;
            tl[2][c4 + i][tok] = (uint16_t)lo;
        }
    }
    __syncthreads();
    const long MG = M >> 3;
#pragma unroll
    for (int j = 0; j < 6; ++j) {                                       // 3 planes x 8 token groups x 64 channels (fastest: 1 KB runs)
        const int idx = tid + j * 256, ch = idx & 63, g8 = (idx >> 6) & 7, p = idx >> 9;
        if (m0 + g8 * 8 < M)
            *reinterpret_cast<uint4*>(Pt + (((long)p * MG + (m0 >> 3) + g8) * C + c0 + ch) * 8) = *reinterpret_cast<const uint4*>(&tl[p][ch][g8 * 8]);
    }
}

extern "C" int rpb_split3t(const float* x, void* planes_t, long M, int C, int ldx, int rev, int d0, int d1, int d2,
                           void* stream) {
    RPB_REQUIRE(x && planes_t && M > 0 && M % 8 == 0 && C > 0 && C % 64 == 0 && ldx % 4 == 0 && ldx >= C,
                "split3t: bad arguments (M=%ld C=%d ldx=%d; M %% 8, C %% 64)", M, C, ldx);
    if (rev) RPB_REQUIRE(d0 > 0 && d1 > 0 && d2 > 0 && M % ((long)d0 * d1 * d2) == 0, "split3t: bad mesh for the reversed token order");
    hipLaunchKernelGGL(split3t_kernel, dim3((unsigned)((M + 63) / 64), C / 64), dim3(256), 0, (hipStream_t)stream, x,
                       (uint16_t*)planes_t, M, C, ldx, rev, d0, d1, d2);
    RPB_CHECK_LAUNCH("split3t");
}

struct WgxArgs {
    const uint16_t* Gt;    // [3][M/8][Co][8] bf16 planes of the output gradient
    const uint16_t* Xt;    // [3][M/8][Ci][8] bf16 planes of the input
    float* part;           // [splits][Co * 27 * Ci + Co]
    long M;
    int Co, Ci, T, H, W;
};

__global__ __launch_bounds__(256, 1) void conv3x_wgrad_kernel(WgxArgs a) {
    // The sum runs over u = the X token (so the X windows of the three kh are ALIGNED 32-token runs, no halo) and pairs it with
    // G[u - (kw - 1)]: the one-token misalignment of kw = 0 / 2 is taken on the G side, whose shifted operands serve all three kh
    // (24 v_alignbit per 16-token step instead of 72 when X carried it).
    extern __shared__ unsigned ldsw[];                                  // 2 x WX_BUF dwords, then the group validity tables
    int* Vt = reinterpret_cast<int*>(ldsw + 2 * WX_BUF);                // [2][12]: (kh, group of 8 tokens of the chunk)
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int o = wave & 1, c = wave >> 1;
    const int ncb = a.Ci >> 6;
    // tile order: co tile fastest -- the workgroups of one split that are resident together then stream the SAME X windows
    // (one (ci tile, kt) pair is shared by all Co / 64 of them; with kt fastest every workgroup had its own stream: 30 % L2 hits)
    int tile = blockIdx.y;
    const int nob = a.Co >> 6;
    const int cob = tile % nob;
    tile /= nob;
    const int kt = tile % 3, cib = tile / 3;
    const int n0 = cob * 64, ci0 = cib * 64;
    const int nsplit = gridDim.x, split = blockIdx.x;
    const long per = ((a.M + nsplit - 1) / nsplit + WX_KT - 1) / WX_KT * WX_KT;
    const long mb = (long)split * per;
    long me = mb + per;
    if (me > a.M) me = a.M;
    const long tshift = (long)(kt - 1) * a.H * a.W;                     // token shift of this kt (the kh shift is added per window)
    const long MG = a.M >> 3;

    f32x16 acc[9], accb = zero16();
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = zero16();
    const bool do_bias = (cib == 0 && kt == 1 && c == 0);
    const u32x4 z4 = {0u, 0u, 0u, 0u};
    const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};   // bf16 1.0 x 8

    // ---- X group validity (t / h neighbours inside the mesh), carried chunk to chunk by threads 0..11
    int vh = 0, vt = 0, vw = 0;
    const int vkh = tid >> 2, vgrp = tid & 3;
    if (tid < 12) {
        const long q = mb + vgrp * 8 + tshift + (long)(vkh - 1) * a.W + 3L * a.T * a.H * a.W;
        vw = (int)(q % a.W);
        const long r = q / a.W;
        vh = (int)(r % a.H);
        vt = (int)((r / a.H) % a.T);
    }
    auto group_mask = [&](int buf) __attribute__((always_inline)) {
        if (tid < 12) {
            const int tt = vt - kt + 1, hh = vh - vkh + 1;
            Vt[buf * 12 + tid] = (tt >= 0 && tt < a.T && hh >= 0 && hh < a.H) ? 1 : 0;
            vw += WX_KT;
            while (vw >= a.W) {
                vw -= a.W;
                if (++vh >= a.H) {
                    vh = 0;
                    if (++vt >= a.T) vt = 0;
                }
            }
        }
    };

    // ---- staging: 1152 G pieces (3 planes x 6 groups x 64 co) + 2304 X pieces (9 (kh, plane) x 4 groups x 64 ci) of 16 B over
    // 256 threads = 14 per thread; a wave's 64 lanes are the 64 channels of one (plane, group): one coalesced 1 KB load.
    // Per-thread constants of every piece are computed once; registers are named individually (see conv3x_kernel).
#define WX_FOR14(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13)
#define WX_DECL(J)                                                                                             \
    u32x4 sv##J = z4;                                                                                          \
    bool ok##J = false;                                                                                        \
    const uint16_t* pp##J; /* the piece of the next chunk to load; advances by st##J elements per chunk */     \
    int qr##J, ds##J, vi##J, st##J; /* token offset rel. to t0, LDS dword offset, validity slot, 32 * C */     \
    {                                                                                                          \
        const int idx = tid + J * 256;                                                                         \
        if (idx < 1152) {                                                                                      \
            const int u = idx >> 6, ch = idx & 63, p = u / 6, grp = u - p * 6;                                 \
            qr##J = grp * 8 - 8;                                                                               \
            pp##J = a.Gt + ((long)p * MG * a.Co + n0 + ch) * 8 + ((mb + qr##J) >> 3) * (a.Co * 8L);            \
            ds##J = (p * 64 + ch) * WX_GROW + grp * 4;                                                         \
            vi##J = -1;                                                                                        \
            st##J = a.Co * 32;                                                                                 \
        } else {                                                                                               \
            const int xi = idx - 1152, u = xi >> 6, ch = xi & 63, kp = u >> 2, grp = u & 3;                    \
            const int kh = kp / 3, p = kp - kh * 3;                                                            \
            qr##J = (int)(grp * 8 + tshift + (long)(kh - 1) * a.W);                                            \
            pp##J = a.Xt + ((long)p * MG * a.Ci + ci0 + ch) * 8 + ((mb + qr##J) >> 3) * (a.Ci * 8L);           \
            ds##J = WX_GS + (kp * 64 + ch) * WX_XROW + grp * 4;                                                \
            vi##J = xi < 2304 ? kh * 4 + grp : -2;                                                             \
            st##J = a.Ci * 32;                                                                                 \
        }                                                                                                      \
    }
    WX_FOR14(WX_DECL)
#define WX_LOAD(J)                                                                                             \
    {                                                                                                          \
        const long q = t0 + qr##J;                                                                             \
        bool okp = vi##J != -2 && q >= 0 && q + 8 <= a.M;                                                      \
        if (vi##J >= 0) okp = okp && Vt[vbuf * 12 + vi##J];                                                    \
        sv##J = *reinterpret_cast<const u32x4*>(okp ? pp##J : a.Gt); /* selected at the store: no vmcnt here */ \
        ok##J = okp;                                                                                           \
        pp##J += st##J;                                                                                        \
    }
#define WX_STORE(J) \
    if (vi##J != -2) *reinterpret_cast<u32x4*>(dst + ds##J) = ok##J ? sv##J : z4;
    auto stage_load = [&](long t0, int vbuf) __attribute__((always_inline)) { WX_FOR14(WX_LOAD) };
    auto stage_store = [&](unsigned* dst) __attribute__((always_inline)) { WX_FOR14(WX_STORE) };

    // ---- one k-step = 16 tokens: 9 taps x 6 split products (+ 3 MFMAs against ones for the bias gradient)
    int wg = (int)((mb + 8 * half) % a.W);                              // w of my first token group of the chunk
    auto kstep = [&](const unsigned* buf, int ks) __attribute__((always_inline)) {
        int wpos = wg + ks * 16;
        while (wpos >= a.W) wpos -= a.W;
        const unsigned m0 = wpos + 8 == a.W ? 0x0000FFFFu : 0xFFFFFFFFu;   // kw = 0 pairs X[u] with G[u + 1]: none for w_u = W - 1
        const unsigned m2 = wpos == 0 ? 0xFFFF0000u : 0xFFFFFFFFu;         // kw = 2 pairs X[u] with G[u - 1]: none for w_u = 0
        u32x4 ga[3], g0[3], g2[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const unsigned* gr = buf + (p * 64 + o * 32 + col) * WX_GROW + 4 + ks * 8 + half * 4;   // window starts at t0 - 8
            const u32x4 x = *reinterpret_cast<const u32x4*>(gr);
            const unsigned xm = gr[-1], xn = gr[4];
            ga[p] = x;
            g0[p].x = __builtin_amdgcn_alignbit(x.y, x.x, 16);          // G[u + 1]
            g0[p].y = __builtin_amdgcn_alignbit(x.z, x.y, 16);
            g0[p].z = __builtin_amdgcn_alignbit(x.w, x.z, 16);
            g0[p].w = __builtin_amdgcn_alignbit(xn, x.w, 16) & m0;
            g2[p].x = __builtin_amdgcn_alignbit(x.x, xm, 16) & m2;      // G[u - 1]
            g2[p].y = __builtin_amdgcn_alignbit(x.y, x.x, 16);
            g2[p].z = __builtin_amdgcn_alignbit(x.z, x.y, 16);
            g2[p].w = __builtin_amdgcn_alignbit(x.w, x.z, 16);
        }
        if (do_bias) {
#pragma unroll
            for (int p = 0; p < 3; ++p) accb = mfma_bf16(__builtin_bit_cast(bf16x8, ga[p]), __builtin_bit_cast(bf16x8, ones), accb);
        }
        u32x4 xb[3][3];                                                 // all nine X operands of the step before its MFMAs
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                xb[kh][p] = *reinterpret_cast<const u32x4*>(buf + WX_GS + ((kh * 3 + p) * 64 + c * 32 + col) * WX_XROW + ks * 8 + half * 4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            // (A plane, B plane): lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi; the three kw taps alternate
#define WX_MF(PA, PB)                                                                                                             \
    acc[kh * 3 + 0] = mfma_bf16(__builtin_bit_cast(bf16x8, g0[PA]), __builtin_bit_cast(bf16x8, xb[kh][PB]), acc[kh * 3 + 0]);    \
    acc[kh * 3 + 1] = mfma_bf16(__builtin_bit_cast(bf16x8, ga[PA]), __builtin_bit_cast(bf16x8, xb[kh][PB]), acc[kh * 3 + 1]);    \
    acc[kh * 3 + 2] = mfma_bf16(__builtin_bit_cast(bf16x8, g2[PA]), __builtin_bit_cast(bf16x8, xb[kh][PB]), acc[kh * 3 + 2]);
            WX_MF(2, 0) WX_MF(0, 2) WX_MF(1, 1) WX_MF(1, 0) WX_MF(0, 1) WX_MF(0, 0)
#undef WX_MF
        }
    };

    group_mask(0);
    group_mask(1);
    __syncthreads();
    if (mb < me) {
        stage_load(mb, 0);
        stage_store(ldsw);
    }
    __syncthreads();
    int it = 0;
    for (long t0 = mb; t0 < me; t0 += WX_KT, ++it) {
        const unsigned* cur = ldsw + (it & 1) * WX_BUF;
        const bool more = t0 + WX_KT < me;
        if (more) stage_load(t0 + WX_KT, (it + 1) & 1);                 // in flight during the MFMAs below
        __builtin_amdgcn_sched_barrier(0);
        kstep(cur, 0);
        kstep(cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (more) stage_store(ldsw + ((it + 1) & 1) * WX_BUF);
        group_mask(it & 1);                                             // for chunk it + 2
        wg += WX_KT;
        while (wg >= a.W) wg -= a.W;
        __syncthreads();
    }
#undef WX_STORE
#undef WX_LOAD
#undef WX_DECL
#undef WX_FOR14
    const long K = 27L * a.Ci;
    float* part = a.part + (long)split * ((long)a.Co * K + a.Co);
#pragma unroll
    for (int k9 = 0; k9 < 9; ++k9)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + o * 32 + mfma_row(lane, r);
            const long k = ((long)(kt * 9 + k9)) * a.Ci + ci0 + c * 32 + col;
            part[(long)n * K + k] = acc[k9][r];
        }
    if (do_bias && col == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) part[(long)a.Co * K + n0 + o * 32 + mfma_row(lane, r)] = accb[r];
    }
}

// ---------------------------------------------------------------------------------- weight gradient, X rows in an LDS ring
// The three kh windows of a chunk are the same X rows W tokens apart: window kh of chunk i is window kh - 1 of the chunk W / 32
// later.  When W % 32 == 0 and 2 W + 32 tokens of the 64 ci fit LDS (W <= 128), the X planes are kept in a ring over the tokens
// [t0 - W, t0 + W + 32) and every chunk loads only its 32 NEW tokens: 8 instead of 14 staging pieces per thread (the staging
// path, not the MFMA pipe, bounds the kernel above: 30 % L2 hits, 11 of its 29 ms).  The t / h validity of a group then
// depends on the window that reads it, so it is applied to the B operand when it is read (v_and with 0 / ~0).
struct WgrArgs {
    const uint16_t* Gt;
    const uint16_t* Xt;
    float* part;
    long M;
    int Co, Ci, T, H, W;
    int RL, RS;            // ring length in tokens (2 W + 32), row stride in dwords (RL / 2 + 4)
};

__global__ __launch_bounds__(256, 1) void conv3x_wgrad_ring_kernel(WgrArgs a) {
    extern __shared__ unsigned ldsw[];
    unsigned* Gs = ldsw;                                                // [3 planes][64 co][WX_GROW]: 48-token window at t0 - 8
    unsigned* Xr = ldsw + WX_GS;                                        // [3 planes][64 ci][RS]: the ring
    int* Vt = reinterpret_cast<int*>(Xr + 3 * 64 * a.RS);               // [2][12]
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int o = wave & 1, c = wave >> 1;
    int tile = blockIdx.y;
    const int nob = a.Co >> 6;
    const int cob = tile % nob;
    tile /= nob;
    const int kt = tile % 3, cib = tile / 3;
    const int n0 = cob * 64, ci0 = cib * 64;
    const int nsplit = gridDim.x, split = blockIdx.x;
    const long per = ((a.M + nsplit - 1) / nsplit + WX_KT - 1) / WX_KT * WX_KT;
    const long mb = (long)split * per;
    long me = mb + per;
    if (me > a.M) me = a.M;
    const long tshift = (long)(kt - 1) * a.H * a.W;
    const long MG = a.M >> 3;
    const int RL = a.RL, RS = a.RS;

    f32x16 acc[9], accb = zero16();
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = zero16();
    const bool do_bias = (cib == 0 && kt == 1 && c == 0);
    const u32x4 z4 = {0u, 0u, 0u, 0u};
    const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};

    // ---- X group validity per (kh, group of the chunk), carried chunk to chunk by threads 0..11 (as above)
    int vh = 0, vt = 0, vw = 0;
    const int vkh = tid >> 2, vgrp = tid & 3;
    if (tid < 12) {
        const long q = mb + vgrp * 8 + tshift + (long)(vkh - 1) * a.W + 3L * a.T * a.H * a.W;
        vw = (int)(q % a.W);
        const long r = q / a.W;
        vh = (int)(r % a.H);
        vt = (int)((r / a.H) % a.T);
    }
    auto group_mask = [&](int buf) __attribute__((always_inline)) {
        if (tid < 12) {
            const int tt = vt - kt + 1, hh = vh - vkh + 1;
            Vt[buf * 12 + tid] = (tt >= 0 && tt < a.T && hh >= 0 && hh < a.H) ? -1 : 0;
            vw += WX_KT;
            while (vw >= a.W) {
                vw -= a.W;
                if (++vh >= a.H) {
                    vh = 0;
                    if (++vt >= a.T) vt = 0;
                }
            }
        }
    };

    // ---- staging pieces per chunk: G 3 planes x 6 groups x 64 co = 1152 (J = 0..4), new X 3 planes x 4 groups x 64 ci = 768
    // (J = 5..7); lanes = channels: one coalesced 1 KB load per wave and piece
#define WR_FOR8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define WR_DECL(J)                                                                                             \
    u32x4 sv##J = z4;                                                                                          \
    bool ok##J = false;                                                                                        \
    const uint16_t* pp##J;                                                                                     \
    int qr##J, ds##J, st##J, kd##J; /* token offset rel. to t0, LDS dword offset (G) / row offset (X), 32 C, kind */ \
    {                                                                                                          \
        const int idx = tid + J * 256;                                                                         \
        if (J < 5) {                                                                                           \
            const int u = idx >> 6, ch = idx & 63, p = u / 6, grp = u - p * 6;                                 \
            kd##J = idx < 1152 ? 0 : -1;                                                                       \
            qr##J = grp * 8 - 8;                                                                               \
            pp##J = a.Gt + ((long)p * MG * a.Co + n0 + ch) * 8 + ((mb + qr##J) >> 3) * (a.Co * 8L);            \
            ds##J = (p * 64 + ch) * WX_GROW + grp * 4;                                                         \
            st##J = a.Co * 32;                                                                                 \
        } else {                                                                                               \
            const int xi = idx - 5 * 256, u = xi >> 6, ch = xi & 63, p = u >> 2, grp = u & 3;                  \
            kd##J = 1;                                                                                         \
            qr##J = (int)(grp * 8 + tshift); /* + the ring offset of the chunk being loaded */                 \
            pp##J = a.Xt + ((long)p * MG * a.Ci + ci0 + ch) * 8 + ((mb - a.W + qr##J) >> 3) * (a.Ci * 8L);     \
            ds##J = (p * 64 + ch) * RS + grp * 4;                                                              \
            st##J = a.Ci * 32;                                                                                 \
        }                                                                                                      \
    }
    WR_FOR8(WR_DECL)
    // load the pieces of the G window of chunk t0 (J < 5) and of the X tokens [xq, xq + 32) + tshift (J >= 5)
#define WR_LOAD(J)                                                                                             \
    {                                                                                                          \
        const long q = (kd##J == 1 ? xq : t0) + qr##J;                                                         \
        const bool okp = kd##J >= 0 && q >= 0 && q + 8 <= a.M && (kd##J == 1 ? xon : gon);                     \
        sv##J = *reinterpret_cast<const u32x4*>(okp ? pp##J : a.Gt);                                           \
        ok##J = okp;                                                                                           \
        if (kd##J == 1 ? xon : gon) pp##J += st##J;                                                            \
    }
#define WR_STORE(J)                                                                                            \
    if (kd##J == 0 && gon) *reinterpret_cast<u32x4*>(Gs + ds##J) = ok##J ? sv##J : z4;                         \
    else if (kd##J == 1 && xon) *reinterpret_cast<u32x4*>(Xr + ds##J + (xpos >> 1)) = ok##J ? sv##J : z4;
    auto stage_load = [&](long t0, long xq, bool gon, bool xon) __attribute__((always_inline)) { WR_FOR8(WR_LOAD) };
    auto stage_store = [&](int xpos, bool gon, bool xon) __attribute__((always_inline)) { WR_FOR8(WR_STORE) };

    int wg = (int)((mb + 8 * half) % a.W);
    // one k-step = 16 tokens, in two halves so that the operands of step i + 1 (LDS reads, alignbit / and fix-ups) are prepared
    // in the same scheduling region as the 54 MFMAs of step i (in-kernel cycle counters: 5.8 k cycles per 32-token chunk for
    // 3.5 k cycles of MFMA issue when every step first prepared and then multiplied)
    struct Ops {
        u32x4 ga[3], g0[3], g2[3], xb[3][3];
    };
    auto prep = [&](int ks, int vb, int xp0, int xp1, int xp2, Ops& o_) __attribute__((always_inline)) {
        int wpos = wg + ks * 16;
        while (wpos >= a.W) wpos -= a.W;
        const unsigned m0 = wpos + 8 == a.W ? 0x0000FFFFu : 0xFFFFFFFFu;
        const unsigned m2 = wpos == 0 ? 0xFFFF0000u : 0xFFFFFFFFu;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const unsigned* gr = Gs + (p * 64 + o * 32 + col) * WX_GROW + 4 + ks * 8 + half * 4;
            const u32x4 x = *reinterpret_cast<const u32x4*>(gr);
            const unsigned xm = gr[-1], xn = gr[4];
            o_.ga[p] = x;
            o_.g0[p].x = __builtin_amdgcn_alignbit(x.y, x.x, 16);
            o_.g0[p].y = __builtin_amdgcn_alignbit(x.z, x.y, 16);
            o_.g0[p].z = __builtin_amdgcn_alignbit(x.w, x.z, 16);
            o_.g0[p].w = __builtin_amdgcn_alignbit(xn, x.w, 16) & m0;
            o_.g2[p].x = __builtin_amdgcn_alignbit(x.x, xm, 16) & m2;
            o_.g2[p].y = __builtin_amdgcn_alignbit(x.y, x.x, 16);
            o_.g2[p].z = __builtin_amdgcn_alignbit(x.z, x.y, 16);
            o_.g2[p].w = __builtin_amdgcn_alignbit(x.w, x.z, 16);
        }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int xp = kh == 0 ? xp0 : (kh == 1 ? xp1 : xp2);
            const unsigned vm = (unsigned)Vt[vb * 12 + kh * 4 + ks * 2 + half];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                u32x4 x = *reinterpret_cast<const u32x4*>(Xr + (p * 64 + c * 32 + col) * RS + (xp >> 1) + ks * 8 + half * 4);
                x.x &= vm; x.y &= vm; x.z &= vm; x.w &= vm;
                o_.xb[kh][p] = x;
            }
        }
    };
    auto mma = [&](const Ops& o_) __attribute__((always_inline)) {
        if (do_bias) {
#pragma unroll
            for (int p = 0; p < 3; ++p) accb = mfma_bf16(__builtin_bit_cast(bf16x8, o_.ga[p]), __builtin_bit_cast(bf16x8, ones), accb);
        }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
#define WX_MF(PA, PB)                                                                                                                   \
    acc[kh * 3 + 0] = mfma_bf16(__builtin_bit_cast(bf16x8, o_.g0[PA]), __builtin_bit_cast(bf16x8, o_.xb[kh][PB]), acc[kh * 3 + 0]);    \
    acc[kh * 3 + 1] = mfma_bf16(__builtin_bit_cast(bf16x8, o_.ga[PA]), __builtin_bit_cast(bf16x8, o_.xb[kh][PB]), acc[kh * 3 + 1]);    \
    acc[kh * 3 + 2] = mfma_bf16(__builtin_bit_cast(bf16x8, o_.g2[PA]), __builtin_bit_cast(bf16x8, o_.xb[kh][PB]), acc[kh * 3 + 2]);
            WX_MF(2, 0) WX_MF(0, 2) WX_MF(1, 1) WX_MF(1, 0) WX_MF(0, 1) WX_MF(0, 0)
#undef WX_MF
        }
    };

    // ---- prologue: validity of chunks 0 / 1, ring filled with the tokens [mb - W, mb + W + 32), G window of chunk 0
    group_mask(0);
    group_mask(1);
    const int nfill = RL / WX_KT;
    for (int f = 0; f < nfill; ++f) {
        stage_load(mb, mb - a.W + (long)f * WX_KT, f == 0, true);
        stage_store(f * WX_KT, f == 0, true);
    }
    __syncthreads();
    // ring position of token u: (u - mb + W) mod RL; the chunk's windows start at xw + kh * W, the next new tokens go to xn
    int xw = 0, xn = 0;                                                 // (t0 - mb) mod RL ; (t0 - mb + 2 W + 32) mod RL == xw
    int it = 0;
    for (long t0 = mb; t0 < me; t0 += WX_KT, ++it) {
        const bool more = t0 + WX_KT < me;
        // new X tokens of the NEXT chunk: [t0 + 32 + W, + 32); its G window starts at t0 + 32 - 8
        if (more) stage_load(t0 + WX_KT, t0 + WX_KT + a.W, true, true);
        __builtin_amdgcn_sched_barrier(0);
        int xp1 = xw + a.W, xp2 = xw + 2 * a.W;
        if (xp1 >= RL) xp1 -= RL;
        if (xp2 >= RL) xp2 -= RL;
        Ops o0, o1;
        prep(0, it & 1, xw, xp1, xp2, o0);
        __builtin_amdgcn_sched_barrier(0);
        prep(1, it & 1, xw, xp1, xp2, o1);                              // same region as the MFMAs of step 0
        mma(o0);
        __builtin_amdgcn_sched_barrier(0);
        mma(o1);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                                // everyone is done with G, window 0 and the validity table
        // the next chunk's new tokens overwrite this chunk's window 0: ring slot (xw + 2 W + 32) mod RL == xw
        if (more) stage_store(xw, true, true);
        group_mask(it & 1);                                             // for chunk it + 2
        xw += WX_KT;
        if (xw >= RL) xw -= RL;
        wg += WX_KT;
        while (wg >= a.W) wg -= a.W;
        __syncthreads();
    }
    (void)xn;
#undef WR_STORE
#undef WR_LOAD
#undef WR_DECL
#undef WR_FOR8
    const long K = 27L * a.Ci;
    float* part = a.part + (long)split * ((long)a.Co * K + a.Co);
#pragma unroll
    for (int k9 = 0; k9 < 9; ++k9)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + o * 32 + mfma_row(lane, r);
            const long k = ((long)(kt * 9 + k9)) * a.Ci + ci0 + c * 32 + col;
            part[(long)n * K + k] = acc[k9][r];
        }
    if (do_bias && col == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) part[(long)a.Co * K + n0 + o * 32 + mfma_row(lane, r)] = accb[r];
    }
}

static int conv3x_wgrad_nsplit(long M, int Co, int Ci) {
    // splits: a multiple of 8 (XCD affinity, see above) that makes workgroups per CU integral, >= 1024 tokens per split
    const long tiles = (long)(Co / 64) * (Ci / 64) * 3, ncu = rpb_num_cus();
    long cap = M / 1024;
    if (cap < 8) cap = 8;
    long best = 8;
    double best_fill = 0.0;
    for (long sp = 8; sp <= cap && sp * tiles <= 6 * ncu + 8 * tiles; sp += 8) {
        const long blocks = tiles * sp, rounds = (blocks + ncu - 1) / ncu;
        const double fill = (double)blocks / (double)(rounds * ncu);
        if (fill >= best_fill) {
            best_fill = fill;
            best = sp;
        }
    }
    return (int)best;
}

extern "C" int rpb_conv3x_wgrad_splits(long M, int Co, int Ci) { return conv3x_wgrad_nsplit(M, Co, Ci); }

extern "C" int rpb_conv3x_wgrad(const void* Gt, const void* Xt, float* part, long M, int Co, int Ci, int Hc, int Wc, int Dc,
                                void* stream) {
    RPB_REQUIRE(Gt && Xt && part && M > 0 && M % 8 == 0 && M < (1L << 40), "conv3x_wgrad: bad arguments");
    RPB_REQUIRE(Co % 64 == 0 && Ci % 64 == 0 && Dc % 8 == 0 && Dc >= 16, "conv3x_wgrad: Co=%d Ci=%d W=%d unsupported (Co, Ci %% 64; innermost mesh dimension %% 8)", Co, Ci, Dc);
    RPB_REQUIRE(Hc > 0 && Wc > 0 && M % ((long)Hc * Wc * Dc) == 0, "conv3x_wgrad: bad mesh");
    WgxArgs a{(const uint16_t*)Gt, (const uint16_t*)Xt, part, M, Co, Ci, Hc, Wc, Dc};
    const size_t lds = (size_t)2 * WX_BUF * 4 + 2 * 12 * 4;
    RPB_REQUIRE(lds <= 160 * 1024, "conv3x_wgrad: LDS");
    (void)hipFuncSetAttribute((const void*)conv3x_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int sp = conv3x_wgrad_nsplit(M, Co, Ci);
    const int RL = 2 * Dc + WX_KT, RS = RL / 2 + 4;
    const size_t lds_ring = ((size_t)WX_GS + 3 * 64 * RS) * 4 + 2 * 12 * 4;
    if (Dc % 32 == 0 && lds_ring <= 160 * 1024 && !getenv("RPB_WGRAD_NORING")) {      // X rows in an LDS ring: each loaded once
        WgrArgs r{(const uint16_t*)Gt, (const uint16_t*)Xt, part, M, Co, Ci, Hc, Wc, Dc, RL, RS};
        (void)hipFuncSetAttribute((const void*)conv3x_wgrad_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ring);
        hipLaunchKernelGGL(conv3x_wgrad_ring_kernel, dim3(sp, (Co / 64) * (Ci / 64) * 3), dim3(256), lds_ring, (hipStream_t)stream, r);
        RPB_CHECK_LAUNCH("conv3x_wgrad");
    }
    hipLaunchKernelGGL(conv3x_wgrad_kernel, dim3(sp, (Co / 64) * (Ci / 64) * 3), dim3(256), lds, (hipStream_t)stream, a);
    RPB_CHECK_LAUNCH("conv3x_wgrad");
}
