// Token GEMM on the bf16 MFMA from split fp32 operands:  out[m][n] = epilogue( sum_k A[m][k] * W[n][k] )   (nn.Linear layout)
// The dense projections of the Galerkin Transformer / Transolver (K = 256 .. 1024, 2.6 M / 0.66 M tokens) run at the sustained
// rate of the fp32 MFMA pipe in rpb_gemm_nt (100-120 TF/s) although their arithmetic intensity would allow twice that from HBM.
// Same split as rpb_conv3x.hip (x = hi + mid + lo in bf16, six products per fp32 product, fp32 accumulate: fp32-grade), same
// workgroup organisation (128 rows x 64 WN columns, 4 waves x (128 x 64) tiles, ONE workgroup per CU with the whole register
// file, double-buffered LDS stage of 64 k-columns, B operands from operand-ordered planes one step ahead) -- but the A operand
// is split ON THE WAY INTO LDS: the workgroup reads fp32 rows (a separate split pass over a K = 256 operand would cost as much
// HBM time as the GEMM saves) and every staged element is split once (5 VALU ops) for the 8 column tiles that use it.
// Epilogue = rpb_gemm_nt's (bias, broadcast vector, GELU / GELU' / ReLU / ReLU', dropout mask or in-kernel Philox dropout,
// residual), 16 B per lane through a wave-private LDS tile.  Replaces, for K % 64 == 0 and N in {64, 128, 256 k}, the nn.Linear
// calls listed at rpb_gemm_nt in include/rpb.h.
#include "rpb_common.h"
#include "rpb_gemm3x2.h"

typedef __attribute__((ext_vector_type(8))) __bf16 g3_bf16x8;
typedef __bf16 g3_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned g3_u32x4 __attribute__((ext_vector_type(4)));
template <int V>
struct G3IC {
    static constexpr int value = V;
};

#define G3_BM 128

__device__ __forceinline__ f32x16 g3_mfma(g3_u32x4 a, g3_u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(g3_bf16x8, a), __builtin_bit_cast(g3_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void g3_split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const f32x2 v = {x0, x1};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, g3_bf16x2));                 // v_cvt_pk_bf16_f32 (RNE)
    const f32x2 hf = {__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xFFFF0000u)};
    const f32x2 r1 = v - hf;                                                                  // exact
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, g3_bf16x2));
    const f32x2 mf = {__builtin_bit_cast(float, m << 16), __builtin_bit_cast(float, m & 0xFFFF0000u)};
    const f32x2 r2 = r1 - mf;                                                                 // exact
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, g3_bf16x2));
}
__device__ __forceinline__ void g3_split8(f32x4 v0, f32x4 v1, g3_u32x4& h, g3_u32x4& m, g3_u32x4& l) {
    unsigned hh[4], mm[4], ll[4];
    g3_split_pair(v0[0], v0[1], hh[0], mm[0], ll[0]);
    g3_split_pair(v0[2], v0[3], hh[1], mm[1], ll[1]);
    g3_split_pair(v1[0], v1[1], hh[2], mm[2], ll[2]);
    g3_split_pair(v1[2], v1[3], hh[3], mm[3], ll[3]);
    h = g3_u32x4{hh[0], hh[1], hh[2], hh[3]};
    m = g3_u32x4{mm[0], mm[1], mm[2], mm[3]};
    l = g3_u32x4{ll[0], ll[1], ll[2], ll[3]};
}

// ---------------------------------------------------------------------------------- weights in B-operand order
// W [N][K] fp32 -> Wz[K/16][3 planes][N/32][64 lanes][8] bf16, lane = (n & 31) + 32 * k-half, element e <-> k = 16 cc + 8 half + e
__global__ __launch_bounds__(256) void gemm3x_wprep_kernel(const float* __restrict__ W, uint16_t* __restrict__ Wz, int N, int K) {
    const int NT = N >> 5, NCC = K >> 4;
    const long total = (long)NCC * NT * 64;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const long r = idx >> 6;
    const int nt = (int)(r % NT), cc = (int)(r / NT);
    const float* src = W + (long)(nt * 32 + (lane & 31)) * K + cc * 16 + (lane >> 5) * 8;
    g3_u32x4 h, m, l;
    g3_split8(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4), h, m, l);
    const long dst = ((((long)cc * 3) * NT + nt) * 64 + lane) * 8;
    *reinterpret_cast<g3_u32x4*>(Wz + dst) = h;
    *reinterpret_cast<g3_u32x4*>(Wz + dst + (long)NT * 512) = m;
    *reinterpret_cast<g3_u32x4*>(Wz + dst + 2L * NT * 512) = l;
}

extern "C" int rpb_gemm3x_wprep(const float* W, void* Wz, int N, int K, void* stream) {
    RPB_REQUIRE(W && Wz && N > 0 && N % 32 == 0 && K > 0 && K % 16 == 0, "gemm3x_wprep: N=%d K=%d unsupported", N, K);
    const long total = (long)(K / 16) * (N / 32) * 64;
    hipLaunchKernelGGL(gemm3x_wprep_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W,
                       (uint16_t*)Wz, N, K);
    RPB_CHECK_LAUNCH("gemm3x_wprep");
}

// ---------------------------------------------------------------------------------- the GEMM
struct Gemm3xArgs {
    const float* A;        // [M][lda] fp32
    const uint16_t* Wz;    // operand-ordered planes of W[N][K]
    const float* bias;     // [N] or null
    const float* addvec;   // [N] or null
    const float* residual; // [M][ldo] or null
    float* out;            // [M][ldo]
    long M;
    int N, K, lda, ldo;
    int act;               // as rpb_gemm_nt: 0 none, 1 GELU (pre_out optional), 2 * gelu'(aux), 3 ReLU, 4 zero where aux <= 0
    const float* aux;
    float* pre_out;
    const float* mask;
    DropSpec drop;
};

template <int WN>
__global__ __launch_bounds__(256, 1) void gemm3x_kernel(Gemm3xArgs a) {
    constexpr int KS = 4 / WN, SPS = 4 / KS;                            // K-split factor; my 16-column chunks per stage
    extern __shared__ g3_u32x4 lds4[];                                  // two stage buffers [3 planes][4 chunks][2 halves][128 rows] x 16 B
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = wave % WN, kp = wave / WN;
    // Persistent workgroups (one per CU): tiles of 128 rows are dealt round-robin, and the stage pipeline runs ACROSS tiles -- the
    // first stage of the next tile is loaded during the last stage of the current one and is in LDS before its epilogue starts
    // (with one short-lived workgroup per tile, K = 256 gave 4 stages of work per launch + prologue + epilogue: 96 TF/s).
    const long ntiles = (a.M + G3_BM - 1) / G3_BM;
    long tile = blockIdx.x;
    if (tile >= ntiles) return;
    long m0 = tile * G3_BM;
    const int n0 = blockIdx.y * (64 * WN) + nw * 64;
    const int NT = a.N >> 5;
    f32x16 acc[4][2];
    const f32x4 zf = {0.f, 0.f, 0.f, 0.f};

    const int nc64 = a.K >> 6;
    const uint16_t* wbase = a.Wz + ((long)(n0 >> 5) * 64 + lane) * 8;
    const long wplane = (long)NT * 512, wchunk = 3 * wplane;
    auto bload = [&](const uint16_t* src, g3_u32x4 (&b)[2][3]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) b[tn][p] = *reinterpret_cast<const g3_u32x4*>(src + p * wplane + tn * 512);
    };
    g3_u32x4 bc[2][3], bn[2][3];
    const g3_u32x4* As = lds4;
    g3_u32x4 ac[3], an[3];
    auto lda_ = [&](int s, int tm, g3_u32x4 (&av)[3]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p) av[p] = As[((p * 4 + s) * 2 + half) * G3_BM + tm * 32 + col];
    };
    auto step = [&](int s, bool more) __attribute__((always_inline)) {  // 16 k-columns: 4 row tiles x 12 MFMAs
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            if (tm < 3) lda_(s, tm + 1, an);
            else if (more) lda_(s + KS, 0, an);
            __builtin_amdgcn_sched_barrier(0);
            acc[tm][0] = g3_mfma(ac[2], bc[0][0], acc[tm][0]);          // small terms first; the two column tiles alternate
            acc[tm][1] = g3_mfma(ac[2], bc[1][0], acc[tm][1]);
            acc[tm][0] = g3_mfma(ac[0], bc[0][2], acc[tm][0]);
            acc[tm][1] = g3_mfma(ac[0], bc[1][2], acc[tm][1]);
            acc[tm][0] = g3_mfma(ac[1], bc[0][1], acc[tm][0]);
            acc[tm][1] = g3_mfma(ac[1], bc[1][1], acc[tm][1]);
            acc[tm][0] = g3_mfma(ac[1], bc[0][0], acc[tm][0]);
            acc[tm][1] = g3_mfma(ac[1], bc[1][0], acc[tm][1]);
            acc[tm][0] = g3_mfma(ac[0], bc[0][1], acc[tm][0]);
            acc[tm][1] = g3_mfma(ac[0], bc[1][1], acc[tm][1]);
            acc[tm][0] = g3_mfma(ac[0], bc[0][0], acc[tm][0]);
            acc[tm][1] = g3_mfma(ac[0], bc[1][0], acc[tm][1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 3; ++p) ac[p] = an[p];
        }
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int p = 0; p < 3; ++p) bc[tn][p] = bn[tn][p];
    };
    // ---- A staging: 128 rows x 8 pieces of 8 columns per stage = 4 pieces (2 float4 loads) per thread, named registers
#define G3_FOR4(X) X(0) X(1) X(2) X(3)
#define G3_DECL(J) f32x4 sa##J = zf, sb##J = zf; bool oks##J = false;
    G3_FOR4(G3_DECL)
#define G3_LOAD(J)                                                                              \
    if constexpr (J >= j0 && J < j1) {                                                          \
        const int idx = tid + J * 256, row = idx >> 3, sh = idx & 7;                            \
        const bool okr = mt + row < a.M;                                                        \
        const float* src = a.A + (okr ? (mt + row) * a.lda + c * 64 + sh * 8 : 0);              \
        sa##J = *reinterpret_cast<const f32x4*>(src);                                           \
        sb##J = *reinterpret_cast<const f32x4*>(src + 4);                                       \
        oks##J = okr;                                                                           \
    }
#define G3_STORE(J)                                                                             \
    {                                                                                           \
        const int idx = tid + J * 256, row = idx >> 3, sh = idx & 7;                            \
        g3_u32x4 h, m, l;                                                                       \
        g3_split8(oks##J ? sa##J : zf, oks##J ? sb##J : zf, h, m, l);                                 \
        dst[(0 * 8 + sh) * G3_BM + row] = h;                                                    \
        dst[(1 * 8 + sh) * G3_BM + row] = m;                                                    \
        dst[(2 * 8 + sh) * G3_BM + row] = l;                                                    \
    }
    auto stage_load = [&](long mt, int c, auto j0c, auto j1c) __attribute__((always_inline)) {
        constexpr int j0 = decltype(j0c)::value, j1 = decltype(j1c)::value;
        G3_FOR4(G3_LOAD)
    };
    auto stage_store = [&](g3_u32x4* dst) __attribute__((always_inline)) { G3_FOR4(G3_STORE) };

    // epilogue scratch lives behind the two stage buffers (the next tile's first stage is already in one of them)
    float* red = reinterpret_cast<float*>(lds4 + 2 * 24 * G3_BM);       // [(KS-1) * WN slots][2 tiles][16 regs][64 lanes] <= 24 KB
    const int er = lane >> 3, ec = (lane & 7) * 4;
    // ---- K-split partial sums through LDS, then the epilogue by the kp == 0 waves (wave-private [32][36] tile, 16 B per lane)
    auto reduce_tile = [&](f32x16& c0, f32x16& c1) __attribute__((always_inline)) {
        if (KS > 1) {
            __syncthreads();
            if (kp > 0) {
                float* slot = red + ((kp - 1) * WN + nw) * 2048;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    slot[r * 64 + lane] = c0[r];
                    slot[(16 + r) * 64 + lane] = c1[r];
                }
            }
            __syncthreads();
            if (kp == 0) {
#pragma unroll
                for (int k2 = 1; k2 < KS; ++k2) {
                    const float* slot = red + ((k2 - 1) * WN + nw) * 2048;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        c0[r] += slot[r * 64 + lane];
                        c1[r] += slot[(16 + r) * 64 + lane];
                    }
                }
            }
        }
    };
    // Epilogue in two phases of four 32 x 32 tiles: the accumulators are parked in wave-private LDS tiles ([32][36] floats) and ONE
    // rolled loop applies bias / activation / dropout / residual and stores 16 B per lane.  (Fully unrolled, the 32 copies of the
    // activation + Philox code made the kernel 80 KB -- more than the 64 KB instruction cache two CUs share -- and an epilogue took
    // 31 k cycles per tile; the kernel is ~25 KB now.)  Parking space: the stage buffer that does NOT hold the next tile's first
    // stage (waves 0, 1) and the scratch area behind the buffers (waves 2, 3).
    auto park = [&](float* pt, const f32x16& cv) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) pt[mfma_row(lane, r) * 36 + col] = cv[r];
    };
    auto epilogue_phase = [&](const float* pbase, int tm0) __attribute__((always_inline)) {   // tiles (tm0 + i / 2, i % 2), i = 0..3
#pragma unroll 1
        for (int it = 0; it < 16; ++it) {
            const int ti = it >> 2, q = it & 3;
            const int tm = tm0 + (ti >> 1), tn = ti & 1;
            const int row = q * 8 + er;
            const long m = m0 + tm * 32 + row;
            if (m >= a.M) continue;
            const int n = n0 + tn * 32 + ec;
            const long off = m * a.ldo + n;
            f32x4 v = *reinterpret_cast<const f32x4*>(pbase + ti * (32 * 36) + row * 36 + ec);
            if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + n);
            if (a.act == 1) {
                if (a.pre_out) *reinterpret_cast<f32x4*>(a.pre_out + off) = v;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = gelu_f(v[k]);
            } else if (a.act == 2) {
                const f32x4 ax = *reinterpret_cast<const f32x4*>(a.aux + off);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] *= gelu_grad_f(ax[k]);
            } else if (a.act == 3) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
            } else if (a.act == 4) {
                const f32x4 ax = *reinterpret_cast<const f32x4*>(a.aux + off);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = ax[k] > 0.f ? v[k] : 0.f;
            }
            if (a.mask) v = v * *reinterpret_cast<const f32x4*>(a.mask + off);
            if (a.drop.thr) v = v * dropout4(a.drop, (unsigned long long)off >> 2);
            if (a.addvec) v += *reinterpret_cast<const f32x4*>(a.addvec + n);
            if (a.residual) v += *reinterpret_cast<const f32x4*>(a.residual + off);
            *reinterpret_cast<f32x4*>(a.out + off) = v;
        }
    };
#define G3_EPILOGUE                                                                                              \
    {                                                                                                            \
        reduce_tile(acc[0][0], acc[0][1]);                                                                       \
        reduce_tile(acc[1][0], acc[1][1]);                                                                       \
        reduce_tile(acc[2][0], acc[2][1]);                                                                       \
        reduce_tile(acc[3][0], acc[3][1]);                                                                       \
        if (KS > 1) __syncthreads(); /* the K-split slots (scratch area) are free again */                       \
        float* pbase = wave < 2 ? reinterpret_cast<float*>(lds4 + (buf ^ 1) * 24 * G3_BM) + wave * (4 * 32 * 36)  \
                                : red + (wave - 2) * (4 * 32 * 36);                                              \
        if (kp == 0) {                                                                                           \
            park(pbase + 0 * (32 * 36), acc[0][0]);                                                              \
            park(pbase + 1 * (32 * 36), acc[0][1]);                                                              \
            park(pbase + 2 * (32 * 36), acc[1][0]);                                                              \
            park(pbase + 3 * (32 * 36), acc[1][1]);                                                              \
            __builtin_amdgcn_wave_barrier();                                                                     \
            epilogue_phase(pbase, 0);                                                                            \
            __builtin_amdgcn_wave_barrier();                                                                     \
            park(pbase + 0 * (32 * 36), acc[2][0]);                                                              \
            park(pbase + 1 * (32 * 36), acc[2][1]);                                                              \
            park(pbase + 2 * (32 * 36), acc[3][0]);                                                              \
            park(pbase + 3 * (32 * 36), acc[3][1]);                                                              \
            __builtin_amdgcn_wave_barrier();                                                                     \
            epilogue_phase(pbase, 2);                                                                            \
        }                                                                                                        \
        __syncthreads(); /* parking space is a stage buffer: free it before the next tile's second stage lands */ \
    }
    bload(wbase + (long)kp * wchunk, bc);
    stage_load(m0, 0, G3IC<0>{}, G3IC<4>{});
    stage_store(lds4);
    __syncthreads();
    int buf = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        m0 = tile * G3_BM;
        const long m0n = (tile + gridDim.x) * G3_BM;                    // next tile of this workgroup
        const bool next_tile = tile + gridDim.x < ntiles;
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) acc[tm][0] = acc[tm][1] = zero16();
        for (int c = 0; c < nc64; ++c) {
            const bool last = c + 1 == nc64;
            const bool more = !last || next_tile;                       // a next stage exists (of this tile or of the next one)
            const long mtn = last ? m0n : m0;
            const int cn = last ? 0 : c + 1;
            As = lds4 + buf * 24 * G3_BM;
            lda_(kp, 0, ac);
#define G3_SI_BLOCK(SI)                                                                              \
    if constexpr (SI < SPS) {                                                                        \
        const int s = kp + SI * KS;                                                                  \
        if (SI + 1 < SPS) bload(wbase + (long)(c * 4 + s + KS) * wchunk, bn);                        \
        else if (more) bload(wbase + (long)(cn * 4 + kp) * wchunk, bn);                              \
        if (more) stage_load(mtn, cn, G3IC<SI * 4 / SPS>{}, G3IC<(SI + 1) * 4 / SPS>{});             \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        step(s, SI + 1 < SPS);                                                                       \
    }
            G3_SI_BLOCK(0) G3_SI_BLOCK(1) G3_SI_BLOCK(2) G3_SI_BLOCK(3)
            if (more) stage_store(lds4 + (buf ^ 1) * 24 * G3_BM);
            __syncthreads();
            buf ^= 1;
        }
        G3_EPILOGUE
    }
#undef G3_EPILOGUE
#undef G3_SI_BLOCK
#undef G3_STORE
#undef G3_LOAD
#undef G3_DECL
#undef G3_FOR4
}

extern "C" int rpb_gemm3x(const float* A, const void* Wz, const float* bias, const float* addvec, const float* residual, float* out,
                          long M, int N, int K, int lda, int ldo, int act, const float* aux, float* pre_out, const float* mask,
                          long drop_seed, float drop_keep, void* stream) {
    RPB_REQUIRE(A && Wz && out && M > 0, "gemm3x: bad arguments");
    const DropSpec dsp = make_drop(drop_seed, drop_keep);
    const bool v2 = rpb_gemm3x2_supported(M, N, K, mask != nullptr, dsp.thr != 0);
    RPB_REQUIRE(K % 64 == 0 && (N == 64 || N == 128 || N % 256 == 0 || (v2 && N % 128 == 0)) && lda % 4 == 0 && lda >= K && ldo % 4 == 0 && ldo >= N,
                "gemm3x: N=%d K=%d lda=%d ldo=%d unsupported (K %% 64; N = 64, 128 or a multiple of 256 -- of 128 without mask / dropout; leading dimensions %% 4)", N, K, lda, ldo);
    RPB_REQUIRE(act >= 0 && act <= 4 && ((act != 2 && act != 4) || aux) && (!pre_out || act == 1), "gemm3x: bad activation arguments");
    if (v2) {   // N % 256 == 0, no mask tensor: 64-row tiles, two workgroups per CU
        const G2Args g{A, (const uint16_t*)Wz, bias, addvec, residual, out, M, N, K, lda, ldo, act, aux, pre_out, dsp};
        return rpb_gemm3x2_launch(g, (hipStream_t)stream);
    }
    Gemm3xArgs a{A, (const uint16_t*)Wz, bias, addvec, residual, out, M, N, K, lda, ldo, act, aux, pre_out, mask, dsp};
    const size_t lds = (size_t)2 * 24 * G3_BM * 16 + 24576 + 4 * 32 * 36 * 4;
    long gxl = (M + G3_BM - 1) / G3_BM;
    const long cus = rpb_num_cus() / (N > 256 ? N / 256 : 1);          // one workgroup per CU in total
    if (gxl > cus) gxl = cus < 1 ? 1 : cus;
    const unsigned gx = (unsigned)gxl;
    hipStream_t st = (hipStream_t)stream;
    (void)hipFuncSetAttribute((const void*)gemm3x_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm3x_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm3x_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (N == 64) {
        hipLaunchKernelGGL(gemm3x_kernel<1>, dim3(gx, 1), dim3(256), lds, st, a);
    } else if (N == 128) {
        hipLaunchKernelGGL(gemm3x_kernel<2>, dim3(gx, 1), dim3(256), lds, st, a);
    } else {
        hipLaunchKernelGGL(gemm3x_kernel<4>, dim3(gx, N / 256), dim3(256), lds, st, a);
    }
    RPB_CHECK_LAUNCH("gemm3x");
}
