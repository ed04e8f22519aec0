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
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + m * ldx + c8 * 8);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(x + m * ldx + c8 * 8 + 4);
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
    auto tap_step = [&](int s, int kw, bool more) __attribute__((always_inline)) {                    // kw is a compile-time constant at every call site
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            if (tm < 3) lda(s, kw, tm + 1, an);
            else if (more) lda(kw == 2 ? s + KS : s, kw == 2 ? 0 : kw + 1, 0, an);
            __builtin_amdgcn_sched_barrier(0);
            if ((kw == 0 && wlo[tm]) || (kw == 2 && whi[tm])) ac[0] = ac[1] = ac[2] = z4;
            const bf16x8 ah = __builtin_bit_cast(bf16x8, ac[0]), am = __builtin_bit_cast(bf16x8, ac[1]),
                         al = __builtin_bit_cast(bf16x8, ac[2]);
            const bf16x8 b0h = __builtin_bit_cast(bf16x8, bc[0][0]), b0m = __builtin_bit_cast(bf16x8, bc[0][1]),
                         b0l = __builtin_bit_cast(bf16x8, bc[0][2]), b1h = __builtin_bit_cast(bf16x8, bc[1][0]),
                         b1m = __builtin_bit_cast(bf16x8, bc[1][1]), b1l = __builtin_bit_cast(bf16x8, bc[1][2]);
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
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 3; ++p) ac[p] = an[p];
        }
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int p = 0; p < 3; ++p) bc[tn][p] = bn[tn][p];
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
        bload(w0 + wtap, bn);                                                                                  \
        if (more) stage_load(gn, cn, IC<(SI * 3 + 0) * NLD / NST>{}, IC<(SI * 3 + 1) * NLD / NST>{});          \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        tap_step(s, 0, true);                                                                                  \
        bload(w0 + 2 * wtap, bn);                                                                              \
        if (more) stage_load(gn, cn, IC<(SI * 3 + 1) * NLD / NST>{}, IC<(SI * 3 + 2) * NLD / NST>{});          \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        tap_step(s, 1, true);                                                                                  \
        /* the step after (s, kw 2): next chunk of this stage, else the next stage's first chunk */            \
        if (SI + 1 < SPS) bload(w0 + (long)KS * wchunk, bn);                                                   \
        else if (more) bload(wbase + (long)(gn * 3) * wtap + (long)(cn * 4 + kp) * wchunk, bn);                \
        if (more) stage_load(gn, cn, IC<(SI * 3 + 2) * NLD / NST>{}, IC<(SI * 3 + 3) * NLD / NST>{});          \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        tap_step(s, 2, SI + 1 < SPS);                                                                          \
    }
            CX_SI_BLOCK(0) CX_SI_BLOCK(1) CX_SI_BLOCK(2) CX_SI_BLOCK(3)
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
    // ---- K-split partial sums through LDS (one 32-token row tile at a time), then bias + store by the kp == 0 waves
    float* red = reinterpret_cast<float*>(lds4);                        // [(KS-1) * WN slots][2 tiles][16 regs][64 lanes]
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
        if (KS > 1) {
            __syncthreads();
            if (kp > 0) {
                float* slot = red + ((kp - 1) * WN + nw) * 2048;
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) slot[(tn * 16 + r) * 64 + lane] = acc[tm][tn][r];
            }
            __syncthreads();
            if (kp == 0) {
#pragma unroll
                for (int k2 = 1; k2 < KS; ++k2) {
                    const float* slot = red + ((k2 - 1) * WN + nw) * 2048;
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[tm][tn][r] += slot[(tn * 16 + r) * 64 + lane];
                }
            }
        }
        if (kp == 0) {
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
