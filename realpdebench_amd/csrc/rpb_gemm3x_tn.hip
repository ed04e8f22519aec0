// Weight-gradient ("TN") token GEMM on the bf16 MFMA from split fp32 operands:
//     dW[n][k] = sum_m G[m][n] * A[m][k],   db[n] = sum_m G[m][n]          (autograd of nn.Linear: weight.grad = g^T x, bias.grad = colsum g)
// for M = 0.65 M .. 2.6 M tokens and N, K multiples of 256 -- the nn.Linear weight gradients of the Transolver / Galerkin / DPOT token
// MLPs and projections (reference call sites listed at rpb_gemm_tn in include/rpb.h).  rpb_gemm_tn runs them on the fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32: 107-117 TF/s); here every product is six bf16 products of three-plane splits with fp32 accumulation
// (fp32-grade, as rpb_gemm3x / rpb_conv3x), whose sustained ceiling on this chip is ~270-290 TF/s fp32-equivalent.
//
// The contraction index is the TOKEN, so both MFMA operands need 8 consecutive tokens of one channel per lane -- the transpose of the
// row-major token tensors.  No transposed copy is made: per 16-token step thread t of the workgroup loads column n0 + t of G and
// column k0 + t of A for the 16 tokens (32 dword loads, each instruction 1 KB contiguous across the workgroup), splits its 2 x 16 values
// in registers (every element is split ONCE per workgroup) and writes them as 16 B operand units [plane][token half][channel][8] to LDS,
// where any wave reads any 32-channel block of either operand conflict-free.  Workgroup = 256 x 256 outputs (4 waves x 128 x 128 =
// 256 accumulator registers, ONE workgroup per CU), token ranges are dealt to workgroups (split-K) and the partials are finished by
// rpb_reduce_partials like rpb_gemm_tn's.  Workgroups that share a token range (different output tiles) sit on the same XCD, so the
// operand they share is read from HBM once and from that XCD's L2 afterwards.
#include "rpb_common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 t3_bf16x8;
typedef __bf16 t3_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned t3_u32x4 __attribute__((ext_vector_type(4)));

#define T3_TILE 256       // outputs per workgroup: T3_TILE x T3_TILE
#define T3_STEP 16        // tokens per step = K of one v_mfma_f32_32x32x16_bf16

__device__ __forceinline__ f32x16 t3_mfma(t3_u32x4 a, t3_u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(t3_bf16x8, a), __builtin_bit_cast(t3_bf16x8, b), c, 0, 0, 0);
}
// (x0, x1) -> one dword per plane (round-to-nearest-even bf16; the residuals are exact, so hi + mid + lo == x to 2^-24 relative)
__device__ __forceinline__ void t3_split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const f32x2 v = {x0, x1};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, t3_bf16x2));
    const f32x2 hf = {__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xFFFF0000u)};
    const f32x2 r1 = v - hf;
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, t3_bf16x2));
    const f32x2 mf = {__builtin_bit_cast(float, m << 16), __builtin_bit_cast(float, m & 0xFFFF0000u)};
    const f32x2 r2 = r1 - mf;
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, t3_bf16x2));
}

struct T3Args {
    const float* G;     // [M][ldg]
    const float* A;     // [M][lda]
    float* part;        // [splits][N * K + N]
    long M;
    int N, K, ldg, lda;
    int splits;         // token ranges (multiple of 8)
    long range;         // tokens per range (multiple of T3_STEP)
};

// TN = output rows (channels of G) per workgroup.  256: 4 waves x (128 x 128), 256 accumulator registers, one workgroup per CU.
// 128: 4 waves x (64 x 128), 128 accumulator registers and 72 KB of LDS, so TWO workgroups share a CU and one's staging waits and
// barriers fall into the other's MFMAs (3-5 % faster).
template <int TN>
__global__ __launch_bounds__(256, TN == 256 ? 1 : 2) void gemm3x_tn_kernel(T3Args a) {
    constexpr int NI = TN / 64;                                          // 32-row blocks of a wave's G range (wave grid 2 x 2)
    constexpr int GE = TN * T3_STEP / 256;                               // G values a thread stages per step: 16 (all tokens of its channel) or 8
    constexpr int BUF = 3 * 2 * (TN + T3_TILE);                          // 16 B units of one step buffer: [G: plane][half][TN] then [A: plane][half][256]
    extern __shared__ t3_u32x4 lds4[];
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wk = wave & 1;                             // the wave's (TN / 2) x 128 quadrant of the tile
    const int tiles_k = a.K / T3_TILE, ntiles = (a.N / TN) * tiles_k;
    // XCD-aware deal: workgroup ids i, i + 8, i + 16, .. run on one XCD; the `ntiles` output tiles of a token range are consecutive there
    const int id = blockIdx.x, xcd = id & 7, q = id >> 3;
    const int tile = q % ntiles, rng = (q / ntiles) * 8 + xcd;
    const int n0 = (tile / tiles_k) * TN, k0 = (tile % tiles_k) * T3_TILE;
    const long mb = (long)rng * a.range;
    float* prow = a.part + (long)rng * ((long)a.N * a.K + a.N);
    long rows = a.M - mb;
    if (rows > a.range) rows = a.range;
    if (rows < 0) rows = 0;
    const int nsteps = (int)((rows + T3_STEP - 1) / T3_STEP);

    f32x16 acc[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = zero16();
    float dbsum = 0.f;
    const bool do_db = k0 == 0;

    // descriptors clipped to the range's valid rows: a load past them returns 0 (no tail predicates)
    const rsrc_t rg = make_rsrc(a.G + mb * a.ldg + n0, rows > 0 ? (unsigned)(((rows - 1) * a.ldg + TN) * 4) : 0u);
    const rsrc_t ra = make_rsrc(a.A + mb * a.lda + k0, rows > 0 ? (unsigned)(((rows - 1) * a.lda + T3_TILE) * 4) : 0u);
    const int gb = a.ldg * 4, ab = a.lda * 4;
    const int gch = tid & (TN - 1), ghf = TN == 256 ? 0 : tid >> 7;      // G staging: channel, and (TN = 128) which 8 of the 16 tokens
    float gv[GE], av[T3_STEP];
    auto load_step = [&](int s) __attribute__((always_inline)) {
        const int gofs = gch * 4 + (s * T3_STEP + ghf * 8) * gb, aofs = tid * 4 + s * T3_STEP * ab;
#pragma unroll
        for (int j = 0; j < GE; ++j) gv[j] = buf_load_f32(rg, gofs + j * gb, 0);
#pragma unroll
        for (int j = 0; j < T3_STEP; ++j) av[j] = buf_load_f32(ra, aofs + j * ab, 0);
    };
    // operand unit (plane p, token half hf, channel c) of operand o in step buffer b
    auto unit = [&](int b, int o, int p, int hf, int c) __attribute__((always_inline)) -> t3_u32x4* {
        return lds4 + b * BUF + (o == 0 ? (p * 2 + hf) * TN : 6 * TN + (p * 2 + hf) * T3_TILE) + c;
    };
    auto split_store = [&](const float* v, int b, int o, int hf, int c) __attribute__((always_inline)) {
        unsigned h[4], m[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) t3_split_pair(v[2 * e], v[2 * e + 1], h[e], m[e], l[e]);
        *unit(b, o, 0, hf, c) = t3_u32x4{h[0], h[1], h[2], h[3]};
        *unit(b, o, 1, hf, c) = t3_u32x4{m[0], m[1], m[2], m[3]};
        *unit(b, o, 2, hf, c) = t3_u32x4{l[0], l[1], l[2], l[3]};
    };
    auto store_step = [&](int b) __attribute__((always_inline)) {
        if (TN == 256) {
            split_store(gv, b, 0, 0, gch);
            split_store(gv + (GE > 8 ? 8 : 0), b, 0, 1, gch);
        } else {
            split_store(gv, b, 0, ghf, gch);
        }
        split_store(av, b, 1, 0, tid);
        split_store(av + 8, b, 1, 1, tid);
        if (do_db) {
#pragma unroll
            for (int j = 0; j < GE; ++j) dbsum += gv[j];
        }
    };
    auto compute = [&](int b) __attribute__((always_inline)) {
        t3_u32x4 bo[4][3];                                              // the wave's four 32-column blocks of A, three planes
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) bo[j][p] = *unit(b, 1, p, half, wk * 128 + j * 32 + col);
        t3_u32x4 ao[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) ao[0][p] = *unit(b, 0, p, half, wn * (TN / 2) + col);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (i + 1 < NI) {                                            // the next row block's planes: in flight during this block's MFMAs
#pragma unroll
                for (int p = 0; p < 3; ++p) ao[(i + 1) & 1][p] = *unit(b, 0, p, half, wn * (TN / 2) + (i + 1) * 32 + col);
            }
            // (G plane, A plane): lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi -- small terms first; the four column blocks alternate
#define T3_MF(PA, PB)                                                   \
    acc[i][0] = t3_mfma(ao[i & 1][PA], bo[0][PB], acc[i][0]);           \
    acc[i][1] = t3_mfma(ao[i & 1][PA], bo[1][PB], acc[i][1]);           \
    acc[i][2] = t3_mfma(ao[i & 1][PA], bo[2][PB], acc[i][2]);           \
    acc[i][3] = t3_mfma(ao[i & 1][PA], bo[3][PB], acc[i][3]);
            T3_MF(2, 0) T3_MF(0, 2) T3_MF(1, 1) T3_MF(1, 0) T3_MF(0, 1) T3_MF(0, 0)
#undef T3_MF
        }
    };

    if (nsteps > 0) {
        load_step(0);
        store_step(0);
        if (nsteps > 1) load_step(1);
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            const int b = s & 1;
            if (s + 1 < nsteps) store_step(b ^ 1);                     // registers hold step s + 1 (loaded one step ago)
            if (s + 2 < nsteps) load_step(s + 2);                       // in flight during this step's MFMAs
            compute(b);
            __syncthreads();
        }
    }

    // ---- partial tile: acc[i][j] register r = dW[n0 + wn*(TN/2) + i*32 + mfma_row(lane, r)][k0 + wk*128 + j*32 + col]
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * (TN / 2) + i * 32 + mfma_row(lane, r), k = k0 + wk * 128 + j * 32 + col;
                prow[(long)n * a.K + k] = acc[i][j][r];
            }
    if (do_db) {
        if (TN == 256) {
            prow[(long)a.N * a.K + n0 + tid] = dbsum;
        } else {                                                         // two threads (token halves) per channel
            float* red = reinterpret_cast<float*>(lds4);
            if (ghf == 1) red[gch] = dbsum;
            __syncthreads();
            if (ghf == 0) prow[(long)a.N * a.K + n0 + gch] = dbsum + red[gch];
        }
    }
}

// one workgroup per CU with 256-row tiles when that already fills the chip with few, long token ranges; else 128-row tiles, two per CU
static int t3_tn(int N, int K) {
    static const int force = getenv("RPB_GEMM3X_TN_TILE") ? atoi(getenv("RPB_GEMM3X_TN_TILE")) : 0;
    if (force == 128 || force == 256) return (N % force) ? 128 : force;
    return 128;
}

static long t3_plan(long M, int N, int K, int* splits_out) {
    const int tn = t3_tn(N, K);
    const int ntiles = (N / tn) * (K / T3_TILE);
    int splits = (rpb_num_cus() * (tn == 256 ? 1 : 2) / ntiles) / 8 * 8;
    if (splits < 8) splits = 8;
    long range = ((M + splits - 1) / splits + T3_STEP - 1) / T3_STEP * T3_STEP;
    *splits_out = splits;
    return range;
}

static long t3_min_rows() {
    static const long v = getenv("RPB_GEMM3X_TN_MIN_ROWS") ? atol(getenv("RPB_GEMM3X_TN_MIN_ROWS")) : 4096;
    return v;
}

extern "C" int rpb_gemm3x_tn_supported(long M, int N, int K, int ldg, int lda) {
    static const bool off = getenv("RPB_GEMM_EXACT") && atoi(getenv("RPB_GEMM_EXACT")) == 1;
    static const bool off2 = getenv("RPB_GEMM_TN_F32") && atoi(getenv("RPB_GEMM_TN_F32")) == 1;
    if (off || off2 || N <= 0 || K <= 0 || N % T3_TILE || K % T3_TILE || ldg < N || lda < K || M < t3_min_rows()) return 0;
    const int tn = t3_tn(N, K);
    const int ntiles = (N / tn) * (K / T3_TILE);
    if (ntiles > rpb_num_cus() * (tn == 256 ? 1 : 2) / 8) return 0;
    int splits;
    const long range = t3_plan(M, N, K, &splits);
    // a range's byte offsets must fit the descriptors' 32 bits
    return range * (long)(ldg > lda ? ldg : lda) * 4 < (1l << 31);
}

extern "C" int rpb_gemm3x_tn_splits(long M, int N, int K) {
    int splits;
    t3_plan(M, N, K, &splits);
    return splits;
}

extern "C" int rpb_gemm3x_tn(const float* G, const float* A, float* part, long M, int N, int K, int ldg, int lda, void* stream) {
    RPB_REQUIRE(G && A && part, "gemm3x_tn: null pointer");
    RPB_REQUIRE(rpb_gemm3x_tn_supported(M, N, K, ldg, lda), "gemm3x_tn: M=%ld N=%d K=%d ldg=%d lda=%d unsupported (N, K multiples of 256, M >= 4096)",
                M, N, K, ldg, lda);
    T3Args a{G, A, part, M, N, K, ldg, lda, 0, 0};
    a.range = t3_plan(M, N, K, &a.splits);
    const int tn = t3_tn(N, K);
    const int ntiles = (N / tn) * (K / T3_TILE);
    const size_t lds = (size_t)2 * 3 * 2 * (tn + T3_TILE) * 16;
    if (tn == 256) {
        (void)hipFuncSetAttribute((const void*)gemm3x_tn_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(gemm3x_tn_kernel<256>, dim3((unsigned)(ntiles * a.splits)), dim3(256), lds, (hipStream_t)stream, a);
    } else {
        (void)hipFuncSetAttribute((const void*)gemm3x_tn_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(gemm3x_tn_kernel<128>, dim3((unsigned)(ntiles * a.splits)), dim3(256), lds, (hipStream_t)stream, a);
    }
    RPB_CHECK_LAUNCH("gemm3x_tn");
}
