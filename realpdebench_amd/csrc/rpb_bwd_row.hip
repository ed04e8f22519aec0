// Fused backward "row" kernel of one Fourier layer (C = 32 or 64):
//
//   gs   = BatchNorm3d(+GELU) backward apply         (was rpb_bn_bwd_apply:  read s, gy        write gs)
//   Y1   = GW^T gs   -- adjoint of the last inverse-DFT stage, i.e. the W stage of the spectral backward
//                                                    (was rpb_axis_gemm:      read gs           write Y1)
//   dWc += gs^T x,  dbc += sum gs                    (was rpb_cell_wgrad:     read gs, x)
//
// autograd of fno.py:115-119 + the first stage of the autograd of fno.py:63.  One pass reads s, gy, x once and
// writes gs once: 4 activation passes instead of 6 per layer.  This works because all three consumers want gs in the
// SAME register layout: "lane = channel pair, MFMA k index = cell" is the A operand of the weight-gradient MFMA and
// the B operand of the DFT-stage MFMA (channels-last, read straight from HBM, no LDS staging).
//
// Work item = one w-row of Wp cells (the DFT runs along w); persistent waves; the three input streams are
// double-buffered in registers 4 MFMA steps (8 cells) ahead through SRSRC descriptors (per-row bounds: the pad
// steps of the last chunk read zeros and their stores are dropped by the hardware).
#include "rpb_common.h"
#include "rpb_bwr.h"
// (cache policy of the streaming loads / stores: RPB_STREAM_AUX, rpb_common.h -- nt by default since round 5)
#include <stdlib.h>

template <int N>
struct RowVec;
template <>
struct RowVec<1> {
    typedef float T;
    typedef unsigned U;
};
template <>
struct RowVec<2> {
    typedef f32x2 T;
    typedef unsigned U __attribute__((ext_vector_type(2)));
};

template <int NT>
__device__ __forceinline__ typename RowVec<NT>::T row_load(rsrc_t r, int voff, int soff) {
    if constexpr (NT == 1) return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, RPB_STREAM_AUX));
    else return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, RPB_STREAM_AUX));
}
template <int NT>
__device__ __forceinline__ void row_store(typename RowVec<NT>::T v, rsrc_t r, int voff, int soff) {
    if constexpr (NT == 1) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, RPB_STREAM_AUX);
    else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(typename RowVec<2>::U, v), r, voff, soff, RPB_STREAM_AUX);
}
template <int NT>
__device__ __forceinline__ float rget(const typename RowVec<NT>::T& v, int i) {
    if constexpr (NT == 1) return v;
    else return v[i];
}
template <int NT>
__device__ __forceinline__ void rset(typename RowVec<NT>::T& v, int i, float x) {
    if constexpr (NT == 1) v = x;
    else v[i] = x;
}

struct BwdRowArgs {
    const float* s;        // [G*Wp][C] pre-BatchNorm output of this layer
    const float* gy;       // [G*Wp][C] gradient w.r.t. the layer output
    const float* x;        // [G*Wp][C] layer input (plain, or pre-BN of the previous layer with xf)
    float* gs;             // [G*Wp][C] out (may alias gy)
    const float* mean;     // BatchNorm of THIS layer
    const float* invstd;
    const float* gamma;
    const float* beta;
    const float* sums;     // [2C] global (sum gz | sum gz*shat)
    float inv_count;
    int gelu;
    XForm xf;              // lazy activation of the layer input x
    const float* GWt;      // adjoint W-stage matrix GW^T [K2][Wp], passed TRANSPOSED, i.e. as GW [Wp][K2]
    float* Y1;             // [G][K2][C]
    float* part;           // [nslots][C*C + C]
    int G, Wp, K2;
    int feat_w;            // > 0 (layer 0): x is the feature tensor Phi_c [G*Wp][feat_w]; dW comes out as [C][feat_w] field moments
};

// FEAT (layer 0 of FNO3d, csrc/rpb_feat.hip): the layer input A0 = W0ext Phi_c is never materialised; the weight-gradient operand
// is the feature tensor itself (lane col < FW holds field col of the cell), so the partial holds sum_cells gs (x) phi in columns
// 0 .. FW-1 of its C x C block, from which d convs.0.weight = (.) W0ext^T and the conv path of d fc0 follow by tiny GEMMs.
template <int C, bool FEAT = false>
__global__ __launch_bounds__(512) void bwd_row_kernel(BwdRowArgs a) {
    constexpr int NT = C / 32;
    typedef typename RowVec<NT>::T vec;
    extern __shared__ float Ml[];                      // [Kp][32]   Ml[k][o] = GWt[o][k]
    const int Wp = a.Wp, K2 = a.K2;
    const int nchunk = (Wp + 7) / 8;
    const int Kp = nchunk * 8;
    for (int idx = threadIdx.x; idx < Kp * 32; idx += blockDim.x) {
        const int k = idx >> 5, o = idx & 31;
        Ml[idx] = (k < Wp && o < K2) ? a.GWt[k * K2 + o] : 0.f;      // passed as [Wp][K2] = (GW^T)^T: coalesced fill
    }
    __syncthreads();

    const int waves = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int col = lane & 31, half = lane >> 5;
    const long slot = (long)blockIdx.x * waves + wave;
    const long nslots = (long)gridDim.x * waves;
    const bool gelu = a.gelu != 0;
    const bool has_xf = !FEAT && a.xf.mean != nullptr;
    const int FW = a.feat_w;

    float mu[NT], is[NT], ga[NT], be[NT], m1[NT], m2[NT];
    XParam xp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ch = col * NT + t;
        mu[t] = a.mean[ch];
        is[t] = a.invstd[ch];
        ga[t] = a.gamma[ch];
        be[t] = a.beta[ch];
        m1[t] = a.sums[ch] * a.inv_count;
        m2[t] = a.sums[C + ch] * a.inv_count;
        if (has_xf) xp[t] = xf_load(a.xf, ch);
    }
    f32x16 dW[NT][NT];
    float bsum[NT];
#pragma unroll
    for (int o = 0; o < NT; ++o) {
        bsum[o] = 0.f;
#pragma unroll
        for (int i = 0; i < NT; ++i) dW[o][i] = zero16();
    }

    const int voff = (half * C + col * NT) * 4;
    const unsigned row_bytes = (unsigned)Wp * C * 4;
    vec sa[4], ya[4], xa[4], sb[4], yb[4], xb[4];

    for (long g_ = slot; g_ < a.G; g_ += nslots) {
        // the row index is wave-uniform, but the compiler carries it (and every descriptor derived from it) in vector registers and
        // wraps each buffer access of the row in a waterfall loop (4 v_readfirstlane + 2 v_cmp + a branch per access, 76 per row in the
        // layer-0 instance): pin it to a scalar register
        const long g = (long)__builtin_amdgcn_readfirstlane((int)g_);
        const long off = g * (long)Wp * C;
        const rsrc_t rs = make_rsrc(a.s + off, row_bytes);
        const rsrc_t ry = make_rsrc(a.gy + off, row_bytes);
        const rsrc_t rx = FEAT ? make_rsrc(a.x + g * (long)Wp * FW, (unsigned)Wp * FW * 4u) : make_rsrc(a.x + off, row_bytes);
        const rsrc_t ro = make_rsrc(a.gs ? a.gs + off : a.s, a.gs ? row_bytes : 0u);      // gs == NULL: an empty descriptor drops the stores (no traffic)
        f32x16 accw[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) accw[t] = zero16();

        auto load_chunk = [&](int c, vec (&sv)[4], vec (&yv)[4], vec (&xv)[4]) {
            // the whole row-relative offset goes through voffset: only voffset+imm is bounds-checked (soffset is not)
            const int vo = voff + c * 8 * C * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sv[j] = row_load<NT>(rs, vo + 2 * j * C * 4, 0);
                yv[j] = row_load<NT>(ry, vo + 2 * j * C * 4, 0);
                if (FEAT) {        // field `col` of cell 8 c + 2 j + half (lanes past FW: an offset outside the descriptor -> 0)
                    const int fo = col < FW ? ((c * 8 + 2 * j + half) * FW + col) * 4 : 0x7ffffff0;
                    rset<NT>(xv[j], 0, buf_load_f32(rx, fo, 0));
                } else {
                    xv[j] = row_load<NT>(rx, vo + 2 * j * C * 4, 0);
                }
            }
        };
        auto compute_chunk = [&](int c, const vec (&sv)[4], const vec (&yv)[4], const vec (&xv)[4]) {
            const int vo = voff + c * 8 * C * 4;
            const float* mp = Ml + (c * 8 + half) * 32 + col;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool valid = c * 8 + 2 * j + half < Wp;
                float gsv[NT], xt[NT];
                vec gout;
                if constexpr (NT == 2) {           // the lane's two channels are a pair: packed fp32 math
                    const f32x2 isv = f32x2{is[0], is[1]}, gav = f32x2{ga[0], ga[1]};
                    const f32x2 sh = (sv[j] - f32x2{mu[0], mu[1]}) * isv;
                    f32x2 gz = yv[j];
                    if (gelu) gz = gz * gelu_grad2(pk_fma(sh, gav, f32x2{be[0], be[1]}));
                    f32x2 v = (gav * isv) * ((gz - f32x2{m1[0], m1[1]}) - sh * f32x2{m2[0], m2[1]});
                    v = valid ? v : pk2(0.f);
                    gout = v;
                    gsv[0] = v[0];
                    gsv[1] = v[1];
                    if (FEAT) {
                        xt[0] = xv[j][0];
                        xt[1] = 0.f;
                    } else {
                        f32x2 xx = xv[j];
                        if (has_xf) {
                            xx = pk_fma((xx - f32x2{xp[0].mu, xp[1].mu}) * f32x2{xp[0].is, xp[1].is}, f32x2{xp[0].ga, xp[1].ga},
                                        f32x2{xp[0].be, xp[1].be});
                            if (a.xf.gelu != 0) xx = gelu2(xx);
                        }
                        xt[0] = xx[0];
                        xt[1] = xx[1];
                    }
                } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float sh = (rget<NT>(sv[j], t) - mu[t]) * is[t];
                    const float gyv = rget<NT>(yv[j], t);
                    const float gz = gelu ? gyv * gelu_grad_f(sh * ga[t] + be[t]) : gyv;
                    const float v = ga[t] * is[t] * (gz - m1[t] - sh * m2[t]);
                    gsv[t] = valid ? v : 0.f;
                    rset<NT>(gout, t, gsv[t]);
                    const float xv0 = (FEAT && t > 0) ? 0.f : rget<NT>(xv[j], t);
                    xt[t] = has_xf ? xf_apply(xv0, xp[t], a.xf.gelu != 0) : xv0;
                }
                }
                row_store<NT>(gout, ro, vo + 2 * j * C * 4, 0);          // pad steps fall outside the descriptor
                const float aw = mp[2 * j * 32];
#pragma unroll
                for (int t = 0; t < NT; ++t) accw[t] = mfma32(aw, gsv[t], accw[t]);
#pragma unroll
                for (int o = 0; o < NT; ++o) {
                    bsum[o] += gsv[o];
#pragma unroll
                    for (int i = 0; i < (FEAT ? 1 : NT); ++i) dW[o][i] = mfma32(gsv[o], xt[i], dW[o][i]);
                }
            }
        };
        load_chunk(0, sa, ya, xa);
        for (int c = 0; c < nchunk; c += 2) {
            if (c + 1 < nchunk) load_chunk(c + 1, sb, yb, xb);
            compute_chunk(c, sa, ya, xa);
            if (c + 2 < nchunk) load_chunk(c + 2, sa, ya, xa);
            if (c + 1 < nchunk) compute_chunk(c + 1, sb, yb, xb);
        }
        // Y1[g][o][channel]
        float* yp = a.Y1 + g * (long)K2 * C + col * NT;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = mfma_row(lane, r);
            if (o < K2) {
                vec v;
#pragma unroll
                for (int t = 0; t < NT; ++t) rset<NT>(v, t, accw[t][r]);
                *reinterpret_cast<vec*>(yp + (long)o * C) = v;
            }
        }
    }
    float* part = a.part + slot * ((long)C * C + C);
#pragma unroll
    for (int o = 0; o < NT; ++o) {
#pragma unroll
        for (int i = 0; i < (FEAT ? 1 : NT); ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                part[(long)(mfma_row(lane, r) * NT + o) * C + (FEAT ? col : col * NT + i)] = dW[o][i][r];
        const float b = bsum[o] + __shfl_xor(bsum[o], 32, 64);
        if (half == 0) part[(long)C * C + col * NT + o] = b;
    }
}

extern "C" long rpb_bn_bwd_row_slots(int G) {
    static const int per_cu = getenv("RPB_BWD_ROW_WG_PER_CU") ? atoi(getenv("RPB_BWD_ROW_WG_PER_CU")) : 1;
    long slots = (long)rpb_num_cus() * 8 * (per_cu > 0 ? per_cu : 1);
    const long need = ((long)G + 7) / 8 * 8;
    return slots < need ? slots : need;
}

extern "C" int rpb_bn_bwd_row(const float* s, const float* gy, const float* x, float* gs, const float* mean,
                              const float* invstd, const float* gamma, const float* beta, const float* sums,
                              double count, int gelu, const float* xf_mean, const float* xf_invstd,
                              const float* xf_gamma, const float* xf_beta, int xf_gelu, const float* GWt, float* Y1,
                              float* part, int G, int Wp, int C, int K2, void* stream) {
    RPB_REQUIRE(s && gy && gs && mean && invstd && gamma && beta && sums && GWt && Y1 && part,
                "bn_bwd_row: null pointer");
    RPB_REQUIRE(C == 32 || C == 64, "bn_bwd_row: C=%d not supported by the fused kernel (use the unfused kernels)", C);
    // x == NULL: no weight gradient in this launch (rpb_cell_mix_wgrad of the same layer forms it): the C x C block of the partial
    // rows is left unwritten, only [C] sum gs follows it
    RPB_REQUIRE(x || (rpb_bwr_supported(C, Wp, K2, 0) && !xf_mean), "bn_bwd_row: x == NULL needs the C = 64 bf16-pipe kernel and no input transform");
    RPB_REQUIRE(G > 0 && Wp > 0 && K2 > 0 && K2 <= 32 && count > 0, "bn_bwd_row: bad sizes G=%d Wp=%d K2=%d", G, Wp, K2);
    RPB_REQUIRE((long)Wp * C * 4 < (1L << 31), "bn_bwd_row: row too long");
    if (xf_mean) RPB_REQUIRE(xf_invstd && xf_gamma && xf_beta, "bn_bwd_row: bad input-transform arguments");
    if (rpb_bwr_supported(C, Wp, K2, 0)) {               // C = 64: bf16 matrix pipe, B-layout loads (csrc/rpb_bwr.hip)
        BwrArgs b;
        b.s = s; b.gy = gy; b.x = x; b.gs = gs; b.mean = mean; b.invstd = invstd; b.gamma = gamma; b.beta = beta; b.sums = sums;
        b.inv_count = (float)(1.0 / count); b.gelu = gelu; b.xf = XForm{xf_mean, xf_invstd, xf_gamma, xf_beta, xf_gelu};
        b.GW = GWt; b.Y1 = Y1; b.part = part; b.G = G; b.Wp = Wp; b.K2 = K2; b.FW = 0; b.CS = 64; b.coff = 0;
        return rpb_bwr_launch(b, rpb_bn_bwd_row_slots(G), (hipStream_t)stream);
    }
    BwdRowArgs a;
    a.s = s; a.gy = gy; a.x = x; a.gs = gs; a.mean = mean; a.invstd = invstd; a.gamma = gamma; a.beta = beta;
    a.sums = sums; a.inv_count = (float)(1.0 / count); a.gelu = gelu;
    a.xf = XForm{xf_mean, xf_invstd, xf_gamma, xf_beta, xf_gelu};
    a.GWt = GWt; a.Y1 = Y1; a.part = part; a.G = G; a.Wp = Wp; a.K2 = K2; a.feat_w = 0;
    const int grid = (int)(rpb_bn_bwd_row_slots(G) / 8);
    const size_t lds = (size_t)((Wp + 7) / 8 * 8) * 32 * 4;
    hipStream_t st = (hipStream_t)stream;
    if (C == 32) hipLaunchKernelGGL((bwd_row_kernel<32>), dim3(grid), dim3(512), lds, st, a);
    else hipLaunchKernelGGL((bwd_row_kernel<64>), dim3(grid), dim3(512), lds, st, a);
    RPB_CHECK_LAUNCH("bn_bwd_row");
}

// Width 128 (configs/fsi/fno.yaml, the Galerkin regressor): BatchNorm3d(+GELU) backward apply and the adjoint W stage in ONE pass over gy / s
// instead of rpb_bn_bwd_apply + an rpb_axis_gemm that reads gs back -- the C = 64 row kernel run on each 64-channel half of the 512 B rows
// (both consumers are per channel; a half row is 256 contiguous bytes).  No weight gradient here (rpb_cell_wgrad forms it, with the bias
// gradient); `part`: 2 x rpb_bn_bwd_row_slots(G) rows of 64 * 64 + 64 floats of scratch (per half: [64] sum gs behind an unwritten block).
extern "C" int rpb_bn_bwd_row_c128_supported(int Wp, int K2) { return rpb_bwr_supported_c128(Wp, K2) ? 1 : 0; }
extern "C" int rpb_bn_bwd_row_c128(const float* s, const float* gy, float* gs, const float* mean, const float* invstd, const float* gamma,
                                   const float* beta, const float* sums, double count, int gelu, const float* GWt, float* Y1, float* part,
                                   int G, int Wp, int K2, void* stream) {
    RPB_REQUIRE(s && gy && gs && mean && invstd && gamma && beta && sums && GWt && Y1 && part, "bn_bwd_row_c128: null pointer");
    RPB_REQUIRE(G > 0 && count > 0 && rpb_bwr_supported_c128(Wp, K2), "bn_bwd_row_c128: unsupported G=%d Wp=%d K2=%d", G, Wp, K2);
    RPB_REQUIRE((long)Wp * 512 < (1L << 31), "bn_bwd_row_c128: row too long");
    const long rows = rpb_bn_bwd_row_slots(G);
    for (int h = 0; h < 2; ++h) {
        BwrArgs b;
        b.s = s; b.gy = gy; b.x = nullptr; b.gs = gs;
        b.mean = mean + 64 * h; b.invstd = invstd + 64 * h; b.gamma = gamma + 64 * h; b.beta = beta + 64 * h; b.sums = sums + 64 * h;
        b.inv_count = (float)(1.0 / count); b.gelu = gelu; b.xf = XForm{nullptr, nullptr, nullptr, nullptr, 0};
        b.GW = GWt; b.Y1 = Y1; b.part = part + h * rows * (64 * 64 + 64); b.G = G; b.Wp = Wp; b.K2 = K2; b.FW = 0; b.CS = 128; b.coff = 64 * h;
        const int rc = rpb_bwr_launch(b, rows, (hipStream_t)stream);
        if (rc != RPB_OK) return rc;
    }
    return RPB_OK;
}

// layer 0 of FNO3d on the feature fields: as rpb_bn_bwd_row with x = Phi_c [G*Wp][FW] (rpb_lift_feat, FW = 8 or 32); the partial's
// C x C block holds the field moments sum_cells gs (x) phi in its columns 0 .. FW-1 (the rest is not written), then [C] sum gs
extern "C" int rpb_bn_bwd_row_feat(const float* s, const float* gy, const float* phi, float* gs, const float* mean,
                                   const float* invstd, const float* gamma, const float* beta, const float* sums, double count,
                                   int gelu, const float* GWt, float* Y1, float* part, int G, int Wp, int C, int K2, int FW,
                                   void* stream) {
    // gs == NULL: the BatchNorm-backward result itself is not stored -- layer 0's data gradient is never formed (csrc/rpb_feat.hip), so
    // nothing reads gs_0: only Y1 and the field moments leave the kernel (9.1 instead of 12.9 GB at B = 32)
    RPB_REQUIRE(s && gy && phi && mean && invstd && gamma && beta && sums && GWt && Y1 && part, "bn_bwd_row_feat: null pointer");
    RPB_REQUIRE(C == 64 && (FW == 8 || FW == 32), "bn_bwd_row_feat: C=%d FW=%d unsupported", C, FW);
    RPB_REQUIRE(G > 0 && Wp > 0 && K2 > 0 && K2 <= 32 && count > 0, "bn_bwd_row_feat: bad sizes G=%d Wp=%d K2=%d", G, Wp, K2);
    // (measured at B = 32: the feature-field variant of the bf16-pipe kernel 2.72 ms, this fp32 kernel 2.47 ms -- scalar field loads; off unless asked for)
    if (getenv("RPB_BWR_FEAT") && atoi(getenv("RPB_BWR_FEAT")) == 1 && rpb_bwr_supported(C, Wp, K2, FW)) {
        BwrArgs b;
        b.s = s; b.gy = gy; b.x = phi; b.gs = gs; b.mean = mean; b.invstd = invstd; b.gamma = gamma; b.beta = beta; b.sums = sums;
        b.inv_count = (float)(1.0 / count); b.gelu = gelu; b.xf = XForm{nullptr, nullptr, nullptr, nullptr, 0};
        b.GW = GWt; b.Y1 = Y1; b.part = part; b.G = G; b.Wp = Wp; b.K2 = K2; b.FW = FW; b.CS = 64; b.coff = 0;
        return rpb_bwr_launch(b, rpb_bn_bwd_row_slots(G), (hipStream_t)stream);
    }
    BwdRowArgs a;
    a.s = s; a.gy = gy; a.x = phi; a.gs = gs; a.mean = mean; a.invstd = invstd; a.gamma = gamma; a.beta = beta;
    a.sums = sums; a.inv_count = (float)(1.0 / count); a.gelu = gelu;
    a.xf = XForm{nullptr, nullptr, nullptr, nullptr, 0};
    a.GWt = GWt; a.Y1 = Y1; a.part = part; a.G = G; a.Wp = Wp; a.K2 = K2; a.feat_w = FW;
    const int grid = (int)(rpb_bn_bwd_row_slots(G) / 8);
    const size_t lds = (size_t)((Wp + 7) / 8 * 8) * 32 * 4;
    hipLaunchKernelGGL((bwd_row_kernel<64, true>), dim3(grid), dim3(512), lds, (hipStream_t)stream, a);
    RPB_CHECK_LAUNCH("bn_bwd_row_feat");
}
