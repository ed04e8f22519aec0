// K2/K4: truncated DFT stage = small dense real matrix applied along one strided axis.
//
//   out[g][o][n] = sum_k M[o][k] * in[g][k][n]        n contiguous, k/o strided, g = batch
//   (the matrix argument is M^T, i.e. [K][O] row-major, so that staging it into LDS is a coalesced copy)
//
// Replaces torch.fft.rfftn / irfftn in SpectralConv3d.forward (reference realpdebench/model/fno.py:48,63):
// only the retained modes are ever produced, so the full [B,C,Tp,Hp,Wp/2+1] spectrum (97 % discarded
// by fno.py:51-60) is never materialised.  Complex data is planar (re/im adjacent to the transformed
// axis) so every stage -- forward, inverse and both adjoints -- is this one real kernel.
//
// Mapping: one wave owns OT o-tiles (32 rows each) x NV interleaved n-tiles (column j of n-tile v is
// n = n0 + NV*j + v, so a lane loads/stores NV contiguous floats: 128/256/512 B per half-wave).
// B operand (the data) is loaded straight from HBM in MFMA layout; A operand (the matrix) sits in LDS
// as Mlds[k][o] (o contiguous -> conflict-free ds_read_b32).  Persistent grid, one item per wave.
#include "rpb_common.h"
#include "rpb_axg.h"

template <int NV>
struct VecN;
template <>
struct VecN<1> {
    typedef float T;
};
template <>
struct VecN<2> {
    typedef f32x2 T;
};
template <>
struct VecN<4> {
    typedef f32x4 T;
};

template <int NV>
__device__ __forceinline__ void load_vec(const float* p, bool ok, float (&v)[NV]) {
    if (ok) {
        typename VecN<NV>::T t = *reinterpret_cast<const typename VecN<NV>::T*>(p);
        if constexpr (NV == 1) {
            v[0] = t;
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = t[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = 0.f;
    }
}

// __launch_bounds__(512) although the launch uses 256 threads: under a 256-register budget hipcc hoists more of the chunk loads
// above the MFMA block (OT2 x NV4: 256 registers instead of 200 under the 512-register budget of a 256-thread bound; same two
// waves per SIMD) and the H stages run 25 % faster (0.42 -> 0.31 ms).  Eight waves per workgroup or 256 B rows were slower.
template <int OT, int NV, bool XF>
__global__ __launch_bounds__(512) void axis_gemm_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        const float* __restrict__ M, int G, int K, int O, int N,
                                                        long in_g, long in_k, long out_g, long out_o, int k_valid,
                                                        int accumulate, XForm xf) {
    extern __shared__ float Mlds[];   // [Kp][Op]
    typedef typename VecN<NV>::T vec;
    const int ot_total = (O + 31) / 32;
    const int och = (ot_total + OT - 1) / OT;
    const int Op = och * OT * 32;
    const int Kp = (K + 15) & ~15;                      // whole 8-step chunks; the pad rows are zero
    for (int idx = threadIdx.x; idx < Kp * Op; idx += blockDim.x) {
        const int k = idx / Op, o = idx - k * Op;
        Mlds[idx] = (k < K && o < O) ? M[(long)k * O + o] : 0.f;     // M is passed TRANSPOSED ([K][O]): coalesced fill
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform -> SGPR addressing
    const int waves = blockDim.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int nch = N / (32 * NV);
    const long nitems = (long)G * nch * och;
    int kch = (k_valid + 15) / 16;                      // chunks of 8 MFMA steps (16 k) that hold non-zero input
    if (kch < 1) kch = 1;                               // k_valid == 0: one all-masked chunk -> zeros are written
    const long istride = (long)gridDim.x * waves;
    const vec vz = {};
    constexpr bool has_xf = XF;

    // The wave walks ONE flattened stream of (item, chunk) pairs with two register buffers in ping-pong: the first chunk
    // of the next item is already in flight while the last chunk of the current item is multiplied and stored, so short
    // items (the H / T stages: 3-17 chunks) do not expose a load latency per item.
    struct Item {
        const float* ip;   // uniform input base of the item
        float* op;         // output base (lane part added at store time)
        int oc, nc;
    };
    auto decode = [&](long item) {
        Item it;
        it.oc = (int)(item % och);
        const long r = item / och;
        it.nc = (int)(r % nch);
        const long g = r / nch;
        it.ip = in + g * in_g + (long)it.nc * 32 * NV;
        it.op = out + g * out_g + (long)it.nc * 32 * NV;
        return it;
    };
    const long lane_in = (long)half * in_k + NV * col;
    auto load_chunk = [&](const Item& it, int c, vec (&b)[8]) {
        const float* cp = it.ip + (long)(16 * c) * in_k;                   // uniform
        if (16 * c + 16 <= k_valid) {
#pragma unroll
            for (int s = 0; s < 8; ++s) b[s] = *reinterpret_cast<const vec*>(cp + lane_in + (long)(2 * s) * in_k);
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s)
                b[s] = (16 * c + 2 * s + half < k_valid)
                           ? *reinterpret_cast<const vec*>(cp + lane_in + (long)(2 * s) * in_k) : vz;
        }
    };
    f32x16 acc[OT][NV];
#pragma unroll
    for (int a = 0; a < OT; ++a)
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[a][v] = zero16();
    XParam xp[NV];

    auto compute_chunk = [&](const Item& it, int c, vec (&b)[8]) {
        const float* mp = Mlds + (16 * c + half) * Op + it.oc * OT * 32 + col;
        if (has_xf) {
            // lazy BatchNorm(+GELU) of the producing layer on the data operand (N == C: the n index is the channel)
            if (c == 0) {
#pragma unroll
                for (int v = 0; v < NV; ++v) xp[v] = xf_load(xf, it.nc * 32 * NV + NV * col + v);
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if constexpr (NV == 1) b[s] = xf_apply(b[s], xp[0], xf.gelu != 0);
                else {
#pragma unroll
                    for (int v = 0; v < NV; ++v) b[s][v] = xf_apply(b[s][v], xp[v], xf.gelu != 0);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int a = 0; a < OT; ++a) {
                const float av = mp[2 * s * Op + a * 32];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    float bv;
                    if constexpr (NV == 1) bv = b[s];
                    else bv = b[s][v];
                    acc[a][v] = mfma32(av, bv, acc[a][v]);
                }
            }
        }
    };
    auto store_item = [&](const Item& it) {
        float* op = it.op + NV * col;
#pragma unroll
        for (int a = 0; a < OT; ++a) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int o = (it.oc * OT + a) * 32 + mfma_row(lane, rr);
                if (o < O) {
                    float* dst = op + (long)o * out_o;
                    if constexpr (NV == 1) {
                        dst[0] = accumulate ? dst[0] + acc[a][0][rr] : acc[a][0][rr];
                    } else {
                        vec t;
#pragma unroll
                        for (int v = 0; v < NV; ++v) t[v] = acc[a][v][rr];
                        if (accumulate) {
                            const vec old = *reinterpret_cast<vec*>(dst);
#pragma unroll
                            for (int v = 0; v < NV; ++v) t[v] += old[v];
                        }
                        *reinterpret_cast<vec*>(dst) = t;
                    }
                }
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[a][v] = zero16();
        }
    };

    long item = (long)blockIdx.x * waves + wave;
    if (item >= nitems) return;
    Item cur = decode(item);
    int c = 0;
    vec ba[8], bb[8];
    load_chunk(cur, 0, ba);
    // one pipeline step: prefetch the successor (same item next chunk, or next item chunk 0) into `nb`, multiply `cb`
    auto step = [&](vec (&cb)[8], vec (&nb)[8]) -> bool {
        long nitem = item;
        int ncnk = c + 1;
        Item nxt = cur;
        if (ncnk == kch) {
            nitem = item + istride;
            ncnk = 0;
            if (nitem < nitems) nxt = decode(nitem);
        }
        const bool more = nitem < nitems;
        if (more) load_chunk(nxt, ncnk, nb);
        compute_chunk(cur, c, cb);
        if (c == kch - 1) store_item(cur);
        item = nitem;
        c = ncnk;
        cur = nxt;
        return more;
    };
    while (true) {
        if (!step(ba, bb)) break;
        if (!step(bb, ba)) break;
    }
}

template <int OT, int NV>
static int launch_axis(const float* in, float* out, const float* M, int G, int K, int O, int N, long in_g, long in_k,
                       long out_g, long out_o, int k_valid, int accumulate, XForm xf, hipStream_t stream) {
    const int ot_total = (O + 31) / 32;
    const int och = (ot_total + OT - 1) / OT;
    const int Op = och * OT * 32;
    const int Kp = (K + 15) & ~15;
    const size_t lds = (size_t)Kp * Op * sizeof(float);
    RPB_REQUIRE(lds <= 160 * 1024, "axis_gemm: matrix %dx%d does not fit LDS", O, K);
    if (xf.mean)
        (void)hipFuncSetAttribute((const void*)axis_gemm_kernel<OT, NV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    else
        (void)hipFuncSetAttribute((const void*)axis_gemm_kernel<OT, NV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int waves = 4;
    const long nitems = (long)G * (N / (32 * NV)) * och;
    const int per_cu = lds > 0 ? (int)((160 * 1024) / lds) : 8;
    // workgroups per CU: the light variants (<= 84 registers: the W stage, K = 134 -> O = 32) gain 13 % from six instead of four
    // (their waves sit on s_waitcnt ~40 % of the time, profiles/r01_pmc_axis_gemm.txt); the T stages (OT * NV = 4) lose 10 %
    const int cap = (OT * NV <= 2) ? 6 : 4;
    long grid = (long)rpb_num_cus() * (per_cu < 1 ? 1 : (per_cu > cap ? cap : per_cu));
    const long need = (nitems + waves - 1) / waves;
    if (grid > need) grid = need;
    if (grid < 1) grid = 1;
    if (xf.mean)
        hipLaunchKernelGGL((axis_gemm_kernel<OT, NV, true>), dim3((unsigned)grid), dim3(waves * 64), lds, stream, in, out,
                           M, G, K, O, N, in_g, in_k, out_g, out_o, k_valid, accumulate, xf);
    else
        hipLaunchKernelGGL((axis_gemm_kernel<OT, NV, false>), dim3((unsigned)grid), dim3(waves * 64), lds, stream, in, out,
                           M, G, K, O, N, in_g, in_k, out_g, out_o, k_valid, accumulate, xf);
    RPB_CHECK_LAUNCH("axis_gemm");
}

extern "C" int rpb_axis_gemm(const float* in, float* out, const float* M, int G, int K, int O, int N, long in_g,
                             long in_k, long out_g, long out_o, int k_valid, int accumulate, const float* xf_mean,
                             const float* xf_invstd, const float* xf_gamma, const float* xf_beta, int xf_gelu,
                             void* stream) {
    RPB_REQUIRE(in && out && M, "axis_gemm: null pointer");
    const XForm xf{xf_mean, xf_invstd, xf_gamma, xf_beta, xf_gelu};
    if (xf_mean) RPB_REQUIRE(xf_invstd && xf_gamma && xf_beta && N <= 128, "axis_gemm: input transform needs all four vectors and N == channels");
    RPB_REQUIRE(G > 0 && K > 0 && O > 0 && N > 0, "axis_gemm: bad sizes G=%d K=%d O=%d N=%d", G, K, O, N);
    RPB_REQUIRE(N % 32 == 0, "axis_gemm: N=%d must be a multiple of 32", N);
    RPB_REQUIRE(k_valid >= 0 && k_valid <= K, "axis_gemm: k_valid=%d out of range", k_valid);
    hipStream_t st = (hipStream_t)stream;
    if (rpb_axg_supported(G, K, O, N, in_g, in_k, out_g, out_o, k_valid, accumulate, xf_mean != nullptr)) {
        AxgArgs a{in, out, M, G, K, O, N, in_g, in_k, out_g, out_o, k_valid, xf, 0};      // bf16 matrix pipe, split operands
        return rpb_axg_launch(a, st);
    }
    const int ot_total = (O + 31) / 32;
    int OT = ot_total >= 3 ? 3 : ot_total;
    int NV = (N % 128 == 0) ? 4 : (N % 64 == 0) ? 2 : 1;
    if (ot_total >= 3 && NV == 4) OT = 2;     // wide rows (512 B per half-wave) beat a third o-tile (measured: inverse H stage -9 %)
    while (OT * NV > 8) NV >>= 1;
    // 16-byte vector access needs aligned strides
    if (NV == 4 && ((in_g | in_k | out_g | out_o) & 3)) NV = 2;
    if (NV == 2 && ((in_g | in_k | out_g | out_o) & 1)) NV = 1;
#define RPB_AX(OT_, NV_)                                                                                        \
    if (OT == OT_ && NV == NV_)                                                                                 \
        return launch_axis<OT_, NV_>(in, out, M, G, K, O, N, in_g, in_k, out_g, out_o, k_valid, accumulate, xf, st);
    RPB_AX(1, 1) RPB_AX(1, 2) RPB_AX(1, 4) RPB_AX(2, 1) RPB_AX(2, 2) RPB_AX(2, 4) RPB_AX(3, 1) RPB_AX(3, 2)
#undef RPB_AX
    RPB_FAIL(RPB_ERR_UNSUPPORTED, "axis_gemm: no instantiation OT=%d NV=%d", OT, NV);
}

// forward W stage reading bf16 activations (BASELINE.json configs[4] storage): same contraction, `in` holds bf16 and in_g / in_k
// count bf16 elements; the output (truncated spectrum rows) stays fp32
extern "C" int rpb_axis_gemm_bf16in(const void* in_bf16, float* out, const float* M, int G, int K, int O, int N, long in_g,
                                    long in_k, long out_g, long out_o, int k_valid, void* stream) {
    RPB_REQUIRE(in_bf16 && out && M, "axis_gemm_bf16in: null pointer");
    RPB_REQUIRE(G > 0 && K > 0 && O > 0 && O <= 64 && N > 0 && k_valid >= 1 && k_valid <= K, "axis_gemm_bf16in: bad sizes G=%d K=%d O=%d N=%d", G, K, O, N);
    RPB_REQUIRE(rpb_axg_supported(G, K, O, N, in_g, in_k, out_g, out_o, k_valid, 0, false), "axis_gemm_bf16in: unsupported layout (N=%d must be a multiple of 64, strides 16 B aligned)", N);
    AxgArgs a{(const float*)in_bf16, out, M, G, K, O, N, in_g, in_k, out_g, out_o, k_valid, XForm{nullptr, nullptr, nullptr, nullptr, 0}, 1, 0};
    return rpb_axg_launch(a, (hipStream_t)stream);
}

// inverse H stage writing the rows that cell_mix reads as bf16 (BASELINE.json configs[4] storage, spectra included): `out` holds bf16,
// out_g / out_o count bf16 elements; short contractions with more than 64 output rows (the "resident" kernel)
extern "C" int rpb_axis_gemm_bf16out(const float* in, void* out_bf16, const float* M, int G, int K, int O, int N, long in_g, long in_k,
                                     long out_g, long out_o, int k_valid, void* stream) {
    RPB_REQUIRE(in && out_bf16 && M, "axis_gemm_bf16out: null pointer");
    RPB_REQUIRE(G > 0 && K > 0 && K <= 64 && O > 64 && N > 0 && k_valid >= 1 && k_valid <= K, "axis_gemm_bf16out: bad sizes G=%d K=%d O=%d N=%d", G, K, O, N);
    RPB_REQUIRE(rpb_axg_supported(G, K, O, N, in_g, in_k, out_g, out_o, k_valid, 0, false), "axis_gemm_bf16out: unsupported layout (N=%d must be a multiple of 64, strides 16 B aligned)", N);
    AxgArgs a{in, (float*)out_bf16, M, G, K, O, N, in_g, in_k, out_g, out_o, k_valid, XForm{nullptr, nullptr, nullptr, nullptr, 0}, 0, 1};
    return rpb_axg_launch(a, (hipStream_t)stream);
}
