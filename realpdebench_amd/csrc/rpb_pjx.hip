// Backward of the projection head on the bf16 matrix pipe, without the gu round trips ("pjx"):
//
//   fno.py:121-125    u = fc1 a + b1 (64 -> 128),  v = gelu(u),  out = fc2 v + b2        a = crop(act(BN(s_{L-1})))
//   autograd          gh = (fc2^T gout) * gelu'(u);   g_a = fc1^T gh (scattered into the padded layout, + the BatchNorm-backward
//                     sums of the last Fourier layer);   d fc1 = gh^T a,  d b1 = sum gh,  d fc2 = gout^T v,  d b2 = sum gout
//
// Round 1 ran this as three kernels chained through gu = gh in HBM ([ncrop][128] fp32 = 5.4 GB at B = 32: written by proj_bwd with
// dword stores, read by the fc1-dgrad cell_mix and again by cell_wgrad -- 16 GB of traffic and 8.5 ms, all on the fp32 matrix
// pipe).  gh is cheap to RECOMPUTE from (a, gout) once fc1 runs on v_mfma_f32_16x16x32_bf16 from split operands (the
// fp32-grade arithmetic of rpb_cmx.hip), so the backward is two kernels that each rebuild it in the operand orientation they
// need and never store it:
//
//   pjx_dgrad  (lane = cell):    u^T = W1 a^T -> gh^T in the accumulators = the B operand of  g_a^T = W1^T gh^T  (chained GEMM:
//                                accumulator rows {16 mt + 4 mg + r} of two row tiles are the 8 contraction values of lane-group
//                                mg); 16 B stores of g_a into the padded tensor (zeros in the pad margin), BatchNorm-backward sums
//                                of the last layer from the x registers the kernel already holds.
//   pjx_wgrad  (lane = hidden):  u = a W1^T -> gh in the accumulators = the A operand of  d fc1 = gh^T a  (the two 16-cell tiles
//                                of a wave tile are the 8 contraction values {4 kg + r, 16 + 4 kg + r}); d fc2, d b1, d b2 are
//                                per-lane sums over cells.  A wave owns one half of the hidden units (64): 64 accumulator registers.
#include "rpb_common.h"
#include "rpb_pjx.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

#define PJ_HID 128
#define PJ_WAVES 8
#define PJ_DOMAX 4

namespace {
__device__ __forceinline__ u32x4 ld16(rsrc_t r, int voff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
__device__ __forceinline__ void st16(f32x4v v, rsrc_t r, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, 0);
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
// six products of the three-plane split, small terms first
#define PJ_MAC6(ACC, AH, AM, AL, BH, BM, BL) \
    ACC = mfma16(AH, BL, ACC);                \
    ACC = mfma16(AL, BH, ACC);                \
    ACC = mfma16(AM, BM, ACC);                \
    ACC = mfma16(AH, BM, ACC);                \
    ACC = mfma16(AM, BH, ACC);                \
    ACC = mfma16(AH, BH, ACC);

// d/du of the head's activation: exact-erf GELU (fno.py:124) or SiLU (Galerkin SpectralRegressor); v = act(u) as well
__device__ __forceinline__ void act_pair(float u, bool silu, float& v, float& d) {
    if (silu) {
        const float sig = 1.0f / (1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * u));
        v = u * sig;
        d = sig * (1.0f + u * (1.0f - sig));
    } else {
        const float cdf = 0.5f * (1.0f + fast_erf(u * 0.70710678118654752440f));
        const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * u * u);
        v = u * cdf;
        d = cdf + u * pdf;
    }
}
// channel of contraction index (ks, kg, e) under the 4 x 16 B-per-cell load pattern (see rpb_cmx.hip)
__device__ __forceinline__ int chan_of(int ks, int kg, int e) { return 16 * (2 * ks + (e >> 2)) + 4 * kg + (e & 3); }
}  // namespace

struct PjxArgs {
    const float* s;       // padded pre-BN tensor of the last Fourier layer [B*Tp*Hp*Wp][64]  (a = xf(s) on the cropped cells)
    const float* w1;      // fc1.weight [128][64]
    const float* b1;      // [128]
    const float* w2;      // fc2.weight [DO][128]
    const float* gout;    // [ncrop][DO]
    const float* gu;      // dgrad, optional: gh [ncrop][128] as written by rpb_proj_bwd (then gout is not read)
    float* g;             // dgrad: [ncell][64] gradient w.r.t. the layer output, padded layout
    float* stats_part;    // dgrad: [slots][2][64] BatchNorm-backward sums (sum gz, sum gz*shat)
    float* part;          // wgrad: [slots][128*64 + DO*128 + 128 + DO]   (d fc1 | d fc2 | d b1 | d b2), hidden-half blocks
    int B, DO, act;
    CropMap cm;
    XForm xf;             // BatchNorm (+GELU flag) of the last layer: mean, invstd, gamma, beta
};

// ------------------------------------------------------------------------------------------------------------ dgrad
// LOADGH: gh is READ from p.gu ([ncrop][128] fp32, written by rpb_proj_bwd) instead of being recomputed: the kernel is then the
//         fc1-dgrad "gather" alone (one GEMM, no activation), with the lane's 2 x 16 B per K-step as its contraction values
//         (k = (s, kg, e) <-> hidden 32 s + 8 kg + e).  Measured at B = 32: recompute 3.4 ms (the second evaluation of act' on
//         128 hidden units per cell costs more VALU time than the 5.4 GB it saves), load 2.2 ms, round-1 fp32 kernel 3.3 ms.
template <bool LOADGH>
__global__ __launch_bounds__(PJ_WAVES * 64) void pjx_dgrad_kernel(PjxArgs p) {
    extern __shared__ u32x4 lds4[];
    u32x4* W1T = lds4;                              // [s 4][plane 3][mt2 4][lane]   A of g^T = W1^T gh^T   (rows = channel)
    float* xfl = reinterpret_cast<float*>(W1T + 4 * 3 * 4 * 64);     // [4][64] mean, invstd, gamma, beta
    float* b1l = xfl + 256;                         // [128]                         (recompute flavour only from here on)
    float* w2l = b1l + PJ_HID;                      // [DO][128]
    u32x4* W1A = reinterpret_cast<u32x4*>(w2l + PJ_DOMAX * PJ_HID);  // [ks 2][plane 3][mt 8][lane]   A of u^T = W1 a^T (rows = hidden)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kg = lane >> 4;
    const int DO = p.DO;
    for (int idx = tid; idx < (LOADGH ? 0 : 2 * 8 * 64); idx += blockDim.x) {
        const int l = idx & 63, mt = (idx >> 6) & 7, ks = idx >> 9;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p.w1[(16 * mt + (l & 15)) * 64 + chan_of(ks, l >> 4, e)];
        bf16x8 h, m, lo;
        split8(v, h, m, lo);
        W1A[((ks * 3 + 0) * 8 + mt) * 64 + l] = __builtin_bit_cast(u32x4, h);
        W1A[((ks * 3 + 1) * 8 + mt) * 64 + l] = __builtin_bit_cast(u32x4, m);
        W1A[((ks * 3 + 2) * 8 + mt) * 64 + l] = __builtin_bit_cast(u32x4, lo);
    }
    for (int idx = tid; idx < 4 * 4 * 64; idx += blockDim.x) {
        const int l = idx & 63, mt2 = (idx >> 6) & 3, s = idx >> 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int hid = LOADGH ? 32 * s + 8 * (l >> 4) + e : 32 * s + 16 * (e >> 2) + 4 * (l >> 4) + (e & 3);
            v[e] = p.w1[hid * 64 + 16 * mt2 + (l & 15)];
        }
        bf16x8 h, m, lo;
        split8(v, h, m, lo);
        W1T[((s * 3 + 0) * 4 + mt2) * 64 + l] = __builtin_bit_cast(u32x4, h);
        W1T[((s * 3 + 1) * 4 + mt2) * 64 + l] = __builtin_bit_cast(u32x4, m);
        W1T[((s * 3 + 2) * 4 + mt2) * 64 + l] = __builtin_bit_cast(u32x4, lo);
    }
    if (!LOADGH) {
        for (int idx = tid; idx < PJ_HID; idx += blockDim.x) b1l[idx] = p.b1[idx];
        for (int idx = tid; idx < PJ_DOMAX * PJ_HID; idx += blockDim.x) w2l[idx] = idx < DO * PJ_HID ? p.w2[idx] : 0.f;
    }
    if (tid < 64) {
        xfl[tid] = p.xf.mean[tid];
        xfl[64 + tid] = p.xf.invstd[tid];
        xfl[128 + tid] = p.xf.gamma[tid];
        xfl[192 + tid] = p.xf.beta[tid];
    }
    __syncthreads();

    const CropMap cm = p.cm;
    const long G = (long)p.B * cm.Tp * cm.Hp;
    const long nslots = (long)gridDim.x * PJ_WAVES;
    const long slot = (long)blockIdx.x * PJ_WAVES + wave;
    const int TQ = (cm.Wp + 15) >> 4;
    const unsigned line_bytes = (unsigned)cm.Wp * 256u;
    const bool xgelu = p.xf.gelu != 0, silu = p.act == 1;
    // the lane's 16 channels 16 i + 4 kg + c (as loaded) == 16 mt2 + 4 mg + r (as produced): sums live per lane, reduced at the end
    f32x4v ssum[4], ssq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ssum[i] = ssq[i] = f32x4v{0.f, 0.f, 0.f, 0.f};

    for (long g = slot; g < G; g += nslots) {
        const int h = (int)(g % cm.Hp);
        const long r2 = g / cm.Hp;
        const int t = (int)(r2 % cm.Tp);
        const long b = r2 / cm.Tp;
        const rsrc_t ro = make_rsrc(p.g + g * cm.Wp * 64, line_bytes);
        if (h >= cm.H || t >= cm.T) {                            // uniform: the whole line is padding -> zeros
            const f32x4v z = {0.f, 0.f, 0.f, 0.f};
            for (int off = lane * 16; off < (int)line_bytes; off += 1024) st16(z, ro, off);
            continue;
        }
        const long rc = (b * cm.T + t) * cm.H + h;               // cropped line
        const rsrc_t rx = make_rsrc(p.s + g * cm.Wp * 64, (unsigned)cm.W * 256u);            // cells >= W read as 0
        const rsrc_t rg = make_rsrc(LOADGH ? p.gu + rc * cm.W * PJ_HID : p.gout + rc * cm.W * DO,
                                    (unsigned)(cm.W * (LOADGH ? PJ_HID : DO)) * 4u);            // cells >= W read as 0
        for (int q = 0; q < TQ; ++q) {
            asm volatile("" ::: "memory");
            const bool live = 16 * q < cm.W;                     // uniform: the tile holds cropped cells
            if (!live) {                                         // margin cells W .. Wp-1: zeros
                const f32x4v z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) st16(z, ro, (16 * q + n16) * 256 + i * 64 + kg * 16);
                continue;
            }
            u32x4 xa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xa[i] = ld16(rx, (16 * q + n16) * 256 + i * 64 + kg * 16);
            f32x4v shat[4];                                       // (s - mean) * invstd at the lane's 16 channels: reused by the sums
            f32x4v acc[8];
            if (LOADGH) {                                         // gh rows of the lane's cell: 32 B per K-step
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    acc[i] = __builtin_bit_cast(f32x4v, ld16(rg, (16 * q + n16) * 512 + (i >> 1) * 128 + kg * 32 + (i & 1) * 16));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4v xv = __builtin_bit_cast(f32x4v, xa[i]);
                    const f32x4v mu = *reinterpret_cast<const f32x4v*>(xfl + 16 * i + 4 * kg);
                    const f32x4v is = *reinterpret_cast<const f32x4v*>(xfl + 64 + 16 * i + 4 * kg);
                    shat[i] = (xv - mu) * is;
                }
            }
            float go[PJ_DOMAX];
#pragma unroll
            for (int j = 0; j < PJ_DOMAX; ++j) go[j] = (!LOADGH && j < DO) ? buf_load_f32(rg, ((16 * q + n16) * DO + j) * 4, 0) : 0.f;
            // ---- u^T = W1 a^T + b1: B planes of a^T per K-step, 8 row tiles of 16 hidden units
            if (!LOADGH) {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) acc[mt] = *reinterpret_cast<const f32x4v*>(b1l + 16 * mt + 4 * kg);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float v[8];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int i = 2 * ks + hf;
                    const f32x4v xv = __builtin_bit_cast(f32x4v, xa[i]);
                    const f32x4v mu = *reinterpret_cast<const f32x4v*>(xfl + 16 * i + 4 * kg);
                    const f32x4v is = *reinterpret_cast<const f32x4v*>(xfl + 64 + 16 * i + 4 * kg);
                    const f32x4v ga = *reinterpret_cast<const f32x4v*>(xfl + 128 + 16 * i + 4 * kg);
                    const f32x4v be = *reinterpret_cast<const f32x4v*>(xfl + 192 + 16 * i + 4 * kg);
                    f32x4v z = bn4(xv, mu, is, ga, be, &shat[i]);            // channel pairs: packed fp32 math
                    if (xgelu) z = gelu4(z);
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[4 * hf + c] = z[c];
                }
                bf16x8 Bh, Bm, Bl;
                split8(v, Bh, Bm, Bl);
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) {
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, W1A[((ks * 3 + 0) * 8 + mt) * 64 + lane]);
                    const bf16x8 am = __builtin_bit_cast(bf16x8, W1A[((ks * 3 + 1) * 8 + mt) * 64 + lane]);
                    const bf16x8 al = __builtin_bit_cast(bf16x8, W1A[((ks * 3 + 2) * 8 + mt) * 64 + lane]);
                    PJ_MAC6(acc[mt], ah, am, al, Bh, Bm, Bl)
                }
            }
            // ---- gh^T = (fc2^T gout) * act'(u), in place: row 16 mt + 4 mg + r, column = the lane's cell
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                f32x4v gp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < PJ_DOMAX; ++j) {
                    if (j < DO) {                                 // uniform
                        const f32x4v w = *reinterpret_cast<const f32x4v*>(w2l + j * PJ_HID + 16 * mt + 4 * kg);
                        gp += w * go[j];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float vv, d;
                    act_pair(acc[mt][r], silu, vv, d);
                    acc[mt][r] = gp[r] * d;
                }
            }
            }   // !LOADGH
            // ---- g^T = W1^T gh^T: the contraction values of lane-group mg in K-step s are rows {32 s + 4 mg + r, 32 s + 16 + 4 mg + r}
            f32x4v acc2[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc2[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[2 * s][r];
                    v[4 + r] = acc[2 * s + 1][r];
                }
                bf16x8 Bh, Bm, Bl;
                split8(v, Bh, Bm, Bl);
#pragma unroll
                for (int mt2 = 0; mt2 < 4; ++mt2) {
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, W1T[((s * 3 + 0) * 4 + mt2) * 64 + lane]);
                    const bf16x8 am = __builtin_bit_cast(bf16x8, W1T[((s * 3 + 1) * 4 + mt2) * 64 + lane]);
                    const bf16x8 al = __builtin_bit_cast(bf16x8, W1T[((s * 3 + 2) * 4 + mt2) * 64 + lane]);
                    PJ_MAC6(acc2[mt2], ah, am, al, Bh, Bm, Bl)
                }
            }
            // ---- store 4 x 16 B (channels 16 mt2 + 4 mg ..) of the lane's cell; BatchNorm-backward sums (cells >= W carry g == 0)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                st16(acc2[i], ro, (16 * q + n16) * 256 + i * 64 + kg * 16);
                ssum[i] += acc2[i];
                ssq[i] += acc2[i] * shat[i];
            }
        }
    }
    // per-channel totals: sum over the 16 cell lanes of a lane group, then lane n16 == 0 of group mg writes channels 16 i + 4 mg + c
    float* part = p.stats_part + slot * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = ssum[i][c], q2 = ssq[i][c];
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                a += __shfl_xor(a, off, 64);
                q2 += __shfl_xor(q2, off, 64);
            }
            if (n16 == 0) {
                part[16 * i + 4 * kg + c] = a;
                part[64 + 16 * i + 4 * kg + c] = q2;
            }
        }
}

static size_t pjx_dgrad_lds(bool loadgh) {
    const size_t base = (size_t)(4 * 3 * 4 * 64) * 16 + 256 * 4;
    return loadgh ? base : base + (PJ_HID + PJ_DOMAX * PJ_HID) * 4 + (size_t)(2 * 3 * 8 * 64) * 16;
}

static bool pjx_off() {
    static const bool off = getenv("RPB_PROJ_F32") && atoi(getenv("RPB_PROJ_F32")) == 1;       // round-1 fp32 kernels through gu
    return off;
}

extern "C" int rpb_proj_bwd_fused_supported(int C, int DO, int W, int Wp) {
    return !pjx_off() && C == 64 && DO >= 1 && DO <= PJ_DOMAX && W >= 16 && Wp >= W;
}

extern "C" long rpb_proj_dgrad_slots(int B, int Tp, int Hp) {
    const long G = (long)B * Tp * Hp;
    long grid = 2L * rpb_num_cus();                  // the load flavour needs 128 registers and 50 KB of LDS: two workgroups per CU
    const long need = (G + PJ_WAVES - 1) / PJ_WAVES;
    if (grid > need) grid = need;
    return grid * PJ_WAVES;
}

// g [ncell][64] = crop-scatter( fc1^T ((fc2^T gout) * act'(fc1 a + b1)) ),  a = xf(s) on the cropped cells; stats_part
// [rpb_proj_dgrad_slots][2][64] = partial (sum g, sum g * shat) over all cells, shat = (s - mean) * invstd  (the last layer has
// no GELU after its BatchNorm, so gz = g)
extern "C" int rpb_proj_dgrad(const float* s, const float* w1, const float* b1, const float* w2, const float* gout,
                              const float* gu, float* g, float* stats_part, int B, int DO, int T, int H, int W, int Tp, int Hp, int Wp,
                              const float* xf_mean, const float* xf_invstd, const float* xf_gamma, const float* xf_beta,
                              int xf_gelu, int act, void* stream) {
    RPB_REQUIRE(s && w1 && b1 && w2 && (gout || gu) && g && stats_part && xf_mean && xf_invstd && xf_gamma && xf_beta, "proj_dgrad: null pointer");
    RPB_REQUIRE(rpb_proj_bwd_fused_supported(64, DO, W, Wp), "proj_dgrad: unsupported shape (DO=%d W=%d Wp=%d)", DO, W, Wp);
    PjxArgs p{};
    p.s = s; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.gout = gout; p.gu = gu; p.g = g; p.stats_part = stats_part; p.B = B; p.DO = DO; p.act = act;
    p.cm = CropMap{T, H, W, Tp, Hp, Wp};
    p.xf = XForm{xf_mean, xf_invstd, xf_gamma, xf_beta, xf_gelu};
    const int grid = (int)(rpb_proj_dgrad_slots(B, Tp, Hp) / PJ_WAVES);
    const size_t lds = pjx_dgrad_lds(gu != nullptr);
    if (p.gu) {
        (void)hipFuncSetAttribute((const void*)pjx_dgrad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(pjx_dgrad_kernel<true>, dim3(grid), dim3(PJ_WAVES * 64), lds, (hipStream_t)stream, p);
    } else {
        (void)hipFuncSetAttribute((const void*)pjx_dgrad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(pjx_dgrad_kernel<false>, dim3(grid), dim3(PJ_WAVES * 64), lds, (hipStream_t)stream, p);
    }
    RPB_CHECK_LAUNCH("proj_dgrad");
}


// ------------------------------------------------------------------------------------------------------------ head forward / gu
// The projection head itself on the bf16 matrix pipe (replaces the fp32-MFMA kernels of rpb_proj.hip for C = 64):
//   BWD = false:  out[cell][j] = b2[j] + sum_hid w2[j][hid] act(u[hid]),  u = fc1 a + b1           (fno.py:123-125)
//   BWD = true :  gu[cell][hid] = (sum_j gout[cell][j] w2[j][hid]) act'(u[hid])  + partial sums of d fc2.weight, d fc1.bias, d fc2.bias
// u^T = W1 a^T is computed as in the data-gradient kernel above (hidden units = MFMA rows, the lane's cell = MFMA column), so a lane
// holds the 32 hidden units {16 mt + 4 mg + r} of ONE cell: the fc2 contraction is 32 FMAs per output + two cross-group adds, the
// gu rows are 16 B stores, and the parameter-gradient sums are per-lane accumulators reduced over the 16 cell lanes once at the end.
struct PjhArgs {
    const float* s;
    const float* w1;
    const float* b1;
    const float* w2;
    const float* b2;
    const float* gout;    // BWD: [ncrop][DO]
    float* out;           // FWD: [ncrop][DO]
    float* gu;            // BWD: [ncrop][128]
    float* part;          // BWD: [slots][DO*128 + 128 + DO]
    int B, DO, act;
    CropMap cm;
    XForm xf;
};

// DOT = 16 (forward only: the combustion configurations have 16 output features): fc2 runs on the matrix pipe as well -- the
//   activated hidden units in the accumulators are the B operand of out^T = W2 v^T (rows {4 mg + r, 16 + 4 mg + r} of a 32-row
//   K-step, the same accumulator-as-operand trick as the data gradient), W2 sits in LDS in A-operand order.
// BFIN: the input is STORED as bf16 [ncell][64] (BASELINE.json configs[4]): one 16 B load is the exact B operand of a K-step
//   (channel 32 ks + 8 kg + e), three products instead of six, no lazy transform.
template <bool BWD, int DOT, bool BFIN = false>
__global__ __launch_bounds__(PJ_WAVES * 64) void pjx_head_kernel(PjhArgs p) {
    static_assert(!(BWD && (BFIN || DOT > 4)), "gu producer: fp32 input, <= 4 outputs");
    constexpr bool FC2M = DOT > 4;
    extern __shared__ u32x4 lds4[];
    u32x4* W1A = lds4;                                              // [ks 2][plane 3][mt 8][lane]
    float* xfl = reinterpret_cast<float*>(W1A + 2 * 3 * 8 * 64);    // [4][64]
    float* b1l = xfl + 256;                                         // [128]
    float* w2l = b1l + PJ_HID;                                      // [DOT][128]            (FC2M: unused)
    u32x4* W2A = reinterpret_cast<u32x4*>(w2l + (FC2M ? 0 : DOT * PJ_HID));   // FC2M: [s 4][plane 3][lane]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kg = lane >> 4;
    const int DO = p.DO;
    for (int idx = tid; idx < 2 * 8 * 64; idx += blockDim.x) {
        const int l = idx & 63, mt = (idx >> 6) & 7, ks = idx >> 9;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p.w1[(16 * mt + (l & 15)) * 64 + (BFIN ? 32 * ks + 8 * (l >> 4) + e : chan_of(ks, l >> 4, e))];
        bf16x8 h, m, lo;
        split8(v, h, m, lo);
        W1A[((ks * 3 + 0) * 8 + mt) * 64 + l] = __builtin_bit_cast(u32x4, h);
        W1A[((ks * 3 + 1) * 8 + mt) * 64 + l] = __builtin_bit_cast(u32x4, m);
        W1A[((ks * 3 + 2) * 8 + mt) * 64 + l] = __builtin_bit_cast(u32x4, lo);
    }
    for (int idx = tid; idx < PJ_HID; idx += blockDim.x) b1l[idx] = p.b1[idx];
    if (FC2M) {
        for (int idx = tid; idx < 4 * 64; idx += blockDim.x) {
            const int l = idx & 63, s = idx >> 6;
            const int j = l & 15;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = j < DO ? p.w2[j * PJ_HID + 32 * s + 16 * (e >> 2) + 4 * (l >> 4) + (e & 3)] : 0.f;
            bf16x8 h, m, lo;
            split8(v, h, m, lo);
            W2A[(s * 3 + 0) * 64 + l] = __builtin_bit_cast(u32x4, h);
            W2A[(s * 3 + 1) * 64 + l] = __builtin_bit_cast(u32x4, m);
            W2A[(s * 3 + 2) * 64 + l] = __builtin_bit_cast(u32x4, lo);
        }
    } else {
        for (int idx = tid; idx < DOT * PJ_HID; idx += blockDim.x) w2l[idx] = idx < DO * PJ_HID ? p.w2[idx] : 0.f;
    }
    const bool has_xf = !BFIN && p.xf.mean != nullptr;
    if (tid < 64) {
        xfl[tid] = has_xf ? p.xf.mean[tid] : 0.f;
        xfl[64 + tid] = has_xf ? p.xf.invstd[tid] : 1.f;
        xfl[128 + tid] = has_xf ? p.xf.gamma[tid] : 1.f;
        xfl[192 + tid] = has_xf ? p.xf.beta[tid] : 0.f;
    }
    __syncthreads();

    const CropMap cm = p.cm;
    const long GL = (long)p.B * cm.T * cm.H;                        // cropped lines
    const long nslots = (long)gridDim.x * PJ_WAVES;
    const long slot = (long)blockIdx.x * PJ_WAVES + wave;
    const int TQ = (cm.W + 15) >> 4;
    const bool xgelu = has_xf && p.xf.gelu != 0, silu = p.act == 1;
    float b2v[DOT];
#pragma unroll
    for (int j = 0; j < DOT; ++j) b2v[j] = (!BWD && j < DO) ? p.b2[j] : 0.f;
    f32x4v db1[BWD ? 8 : 1], dw2[BWD ? DOT : 1][BWD ? 8 : 1];
    float db2[DOT];
#pragma unroll
    for (int j = 0; j < DOT; ++j) db2[j] = 0.f;
    if (BWD) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            db1[mt] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < DOT; ++j) dw2[j][mt] = f32x4v{0.f, 0.f, 0.f, 0.f};
        }
    }
    auto line_of = [&](long gl) {                                   // cropped line -> padded line
        const int h = (int)(gl % cm.H);
        const long r2 = gl / cm.H;
        return ((r2 / cm.T) * cm.Tp + r2 % cm.T) * cm.Hp + h;
    };
    u32x4 xa[BFIN ? 2 : 4];
    auto issue = [&](long gl, int q) {
        if (BFIN) {                                                  // 128 B per cell: loads ks = 0, 1 are the two K-steps
            const rsrc_t rx = make_rsrc(p.s + line_of(gl) * cm.Wp * 32, (unsigned)cm.W * 128u);
#pragma unroll
            for (int i = 0; i < 2; ++i) xa[i] = ld16(rx, (16 * q + n16) * 128 + i * 64 + kg * 16);
        } else {
            const rsrc_t rx = make_rsrc(p.s + line_of(gl) * cm.Wp * 64, (unsigned)cm.W * 256u);      // cells >= W read as 0
#pragma unroll
            for (int i = 0; i < 4; ++i) xa[i] = ld16(rx, (16 * q + n16) * 256 + i * 64 + kg * 16);
        }
    };
    if (slot < GL) issue(slot, 0);
    for (long gl = slot; gl < GL; gl += nslots) {
        const rsrc_t rgo = make_rsrc((BWD ? p.gout : p.out) + gl * cm.W * DO, (unsigned)(cm.W * DO) * 4u);
        const rsrc_t rgu = make_rsrc(BWD ? p.gu + gl * cm.W * PJ_HID : p.s, BWD ? (unsigned)(cm.W * PJ_HID) * 4u : 0u);
        for (int q = 0; q < TQ; ++q) {
            asm volatile("" ::: "memory");
            const bool last = q + 1 == TQ;
            const long gn = last ? gl + nslots : gl;
            f32x4v acc[8];
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) acc[mt] = *reinterpret_cast<const f32x4v*>(b1l + 16 * mt + 4 * kg);
            float go[DOT];
#pragma unroll
            for (int j = 0; j < DOT; ++j) go[j] = (BWD && j < DO) ? buf_load_f32(rgo, ((16 * q + n16) * DO + j) * 4, 0) : 0.f;
            bf16x8 Bh[2], Bm[2], Bl[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (BFIN) {
                    Bh[ks] = __builtin_bit_cast(bf16x8, xa[ks]);
                    continue;
                }
                float v[8];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int i = BFIN ? 0 : 2 * ks + hf;
                    const f32x4v xv = __builtin_bit_cast(f32x4v, xa[i]);
                    const f32x4v mu = *reinterpret_cast<const f32x4v*>(xfl + 16 * i + 4 * kg);
                    const f32x4v is = *reinterpret_cast<const f32x4v*>(xfl + 64 + 16 * i + 4 * kg);
                    const f32x4v ga = *reinterpret_cast<const f32x4v*>(xfl + 128 + 16 * i + 4 * kg);
                    const f32x4v be = *reinterpret_cast<const f32x4v*>(xfl + 192 + 16 * i + 4 * kg);
                    f32x4v z = bn4(xv, mu, is, ga, be);
                    if (xgelu) z = gelu4(z);
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[4 * hf + c] = z[c];
                }
                split8(v, Bh[ks], Bm[ks], Bl[ks]);
            }
            if (gn < GL) issue(gn, last ? 0 : q + 1);               // the next tile's loads fly during the products below
            // MG row tiles advance together: MG independent accumulation chains per product keep the matrix pipe issuing back to back
            constexpr int MG = BWD ? 2 : 4;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int m0 = 0; m0 < 8; m0 += MG) {
                    bf16x8 ah[MG], am[MG], al[MG];
#pragma unroll
                    for (int u = 0; u < MG; ++u) {
                        ah[u] = __builtin_bit_cast(bf16x8, W1A[((ks * 3 + 0) * 8 + m0 + u) * 64 + lane]);
                        am[u] = __builtin_bit_cast(bf16x8, W1A[((ks * 3 + 1) * 8 + m0 + u) * 64 + lane]);
                        al[u] = __builtin_bit_cast(bf16x8, W1A[((ks * 3 + 2) * 8 + m0 + u) * 64 + lane]);
                    }
#define PJ_ROW(AP, BP) _Pragma("unroll") for (int u = 0; u < MG; ++u) acc[m0 + u] = mfma16(AP[u], BP[ks], acc[m0 + u]);
                    if (BFIN) {
                        if (RPB_BF16_CONST_PLANES > 2) { PJ_ROW(al, Bh) }
                        PJ_ROW(am, Bh) PJ_ROW(ah, Bh)
                    } else {
                        PJ_ROW(ah, Bl) PJ_ROW(al, Bh) PJ_ROW(am, Bm) PJ_ROW(ah, Bm) PJ_ROW(am, Bh) PJ_ROW(ah, Bh)
                    }
#undef PJ_ROW
                }
            if (!BWD && FC2M) {
                // out^T [j][cell] = W2 v^T + b2: K-step s takes hidden rows {32 s + 4 mg + r, 32 s + 16 + 4 mg + r} = acc[2 s], acc[2 s + 1]
                f32x4v o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = 4 * kg + r < DO ? p.b2[4 * kg + r] : 0.f;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    f32x4v v0, v1;
                    if (silu) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float a0, a1, d;
                            act_pair(acc[2 * s4][r], true, a0, d);
                            act_pair(acc[2 * s4 + 1][r], true, a1, d);
                            v0[r] = a0;
                            v1[r] = a1;
                        }
                    } else {
                        v0 = gelu4(acc[2 * s4]);
                        v1 = gelu4(acc[2 * s4 + 1]);
                    }
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = v0[r];
                        v[4 + r] = v1[r];
                    }
                    bf16x8 Vh, Vm, Vl;
                    split8(v, Vh, Vm, Vl);
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, W2A[(s4 * 3 + 0) * 64 + lane]);
                    const bf16x8 am = __builtin_bit_cast(bf16x8, W2A[(s4 * 3 + 1) * 64 + lane]);
                    const bf16x8 al = __builtin_bit_cast(bf16x8, W2A[(s4 * 3 + 2) * 64 + lane]);
                    PJ_MAC6(o, ah, am, al, Vh, Vm, Vl)
                }
                // lane (cell n16, group mg) holds outputs j = 4 mg + r of its cell
                if ((DO & 3) == 0) {
                    if (4 * kg < DO) st16(o, rgo, ((16 * q + n16) * DO + 4 * kg) * 4);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * kg + r < DO) buf_store_f32(o[r], rgo, ((16 * q + n16) * DO + 4 * kg + r) * 4, 0);
                }
            } else if (!BWD) {
                float po[DOT];
#pragma unroll
                for (int j = 0; j < DOT; ++j) po[j] = 0.f;
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) {
                    f32x4v vv;
                    if (silu) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v1, d;
                            act_pair(acc[mt][r], true, v1, d);
                            vv[r] = v1;
                        }
                    } else {
                        vv = gelu4(acc[mt]);
                    }
#pragma unroll
                    for (int j = 0; j < DOT; ++j) {
                        const f32x4v w = *reinterpret_cast<const f32x4v*>(w2l + j * PJ_HID + 16 * mt + 4 * kg);
                        const f32x4v pr = w * vv;
                        po[j] += (pr[0] + pr[1]) + (pr[2] + pr[3]);
                    }
                }
#pragma unroll
                for (int j = 0; j < DOT; ++j) {
                    float t = po[j];
                    t += __shfl_xor(t, 16, 64);
                    t += __shfl_xor(t, 32, 64);
                    if (kg == 0 && j < DO) buf_store_f32(t + b2v[j], rgo, ((16 * q + n16) * DO + j) * 4, 0);   // cells >= W: dropped
                }
            } else {
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) {
                    f32x4v gp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < DOT; ++j) gp += *reinterpret_cast<const f32x4v*>(w2l + j * PJ_HID + 16 * mt + 4 * kg) * go[j];
                    f32x4v vv, dd;
                    if (silu) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v1, d1;
                            act_pair(acc[mt][r], true, v1, d1);
                            vv[r] = v1;
                            dd[r] = d1;
                        }
                    } else {                                        // v = u Phi(u), d = Phi(u) + u phi(u): one erf, packed pairs
                        const f32x4v u = acc[mt];
                        const f32x2 e0 = fast_erf2(u.lo * pk2(0.70710678118654752440f)), e1 = fast_erf2(u.hi * pk2(0.70710678118654752440f));
                        const f32x4v cdf = join4(pk2(0.5f) * (pk2(1.0f) + e0), pk2(0.5f) * (pk2(1.0f) + e1));
                        const f32x4v q2 = (f32x4v{-0.72134752044448170368f, -0.72134752044448170368f, -0.72134752044448170368f, -0.72134752044448170368f} * u) * u;
                        f32x4v ex;
#pragma unroll
                        for (int r = 0; r < 4; ++r) ex[r] = __builtin_amdgcn_exp2f(q2[r]);
                        vv = u * cdf;
                        dd = cdf + u * (ex * 0.39894228040143267794f);
                    }
                    const f32x4v guv = gp * dd;                     // cells >= W carry gout == 0 -> 0, and their store is dropped
                    st16(guv, rgu, (16 * q + n16) * 512 + (16 * mt + 4 * kg) * 4);
                    db1[mt] += guv;
#pragma unroll
                    for (int j = 0; j < DOT; ++j) dw2[j][mt] += vv * go[j];
                }
#pragma unroll
                for (int j = 0; j < DOT; ++j) db2[j] += kg == 0 ? go[j] : 0.f;
            }
        }
    }
    if (BWD) {      // sums over the 16 cell lanes of each lane group; lane n16 == 0 of group mg writes hidden 16 mt + 4 mg + r
        float* part = p.part + slot * ((long)DO * PJ_HID + PJ_HID + DO);
        auto red16 = [&](float v) {
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) v += __shfl_xor(v, off, 64);
            return v;
        };
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float b = red16(db1[mt][r]);
                if (n16 == 0) part[DO * PJ_HID + 16 * mt + 4 * kg + r] = b;
#pragma unroll
                for (int j = 0; j < DOT; ++j) {
                    const float w = red16(dw2[j][mt][r]);
                    if (n16 == 0 && j < DO) part[j * PJ_HID + 16 * mt + 4 * kg + r] = w;
                }
            }
#pragma unroll
        for (int j = 0; j < DOT; ++j) {
            float b = red16(db2[j]);                                 // only group 0 counted the cells
            if (lane == 0 && j < DO) part[DO * PJ_HID + PJ_HID + j] = b;
        }
    }
}

static size_t pjx_head_lds(int dot) {
    return (size_t)(2 * 3 * 8 * 64) * 16 + (256 + PJ_HID) * 4 + (dot > 4 ? (size_t)(4 * 3 * 64) * 16 : (size_t)dot * PJ_HID * 4);
}

// ------------------------------------------------------------------------------------------------------------ wgrad
// A wave owns PJ_NT 16-wide hidden tiles = PJ_HB hidden units (role = slot % PJ_ROLES); part row of a slot:
//   [PJ_HB * 64]  d fc1.weight[PJ_HB role + hl][ch]
//   [DO * PJ_HB]  d fc2.weight[j][PJ_HB role + hl]
//   [PJ_HB]       d fc1.bias[PJ_HB role + hl]
//   [DO]          d fc2.bias[j]                    (role 0 rows only; the other roles' rows hold zeros)
#define PJ_NT 2
#define PJ_HB (16 * PJ_NT)
#define PJ_ROLES (PJ_HID / PJ_HB)
#define PJ_WROW(DO_) (PJ_HB * 64 + (DO_) * PJ_HB + PJ_HB + (DO_))

template <int DOT>
__global__ __launch_bounds__(PJ_WAVES * 64) void pjx_wgrad_kernel(PjxArgs p) {
    extern __shared__ u32x4 lds4[];
    u32x4* W1B = lds4;                              // [role][ks 2][plane 3][t PJ_NT][lane]   B of u = a W1^T   (columns = hidden)
    float* xfl = reinterpret_cast<float*>(W1B + PJ_ROLES * 2 * 3 * PJ_NT * 64);       // [4][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kg = lane >> 4;
    const int DO = p.DO;
    for (int idx = tid; idx < PJ_ROLES * 2 * PJ_NT * 64; idx += blockDim.x) {
        const int l = idx & 63, t = (idx >> 6) % PJ_NT, ks = ((idx >> 6) / PJ_NT) & 1, role = (idx >> 6) / (2 * PJ_NT);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p.w1[(PJ_HB * role + PJ_NT * (l & 15) + t) * 64 + chan_of(ks, l >> 4, e)];
        bf16x8 h, m, lo;
        split8(v, h, m, lo);
        W1B[(((role * 2 + ks) * 3 + 0) * PJ_NT + t) * 64 + l] = __builtin_bit_cast(u32x4, h);
        W1B[(((role * 2 + ks) * 3 + 1) * PJ_NT + t) * 64 + l] = __builtin_bit_cast(u32x4, m);
        W1B[(((role * 2 + ks) * 3 + 2) * PJ_NT + t) * 64 + l] = __builtin_bit_cast(u32x4, lo);
    }
    if (tid < 64) {
        xfl[tid] = p.xf.mean[tid];
        xfl[64 + tid] = p.xf.invstd[tid];
        xfl[128 + tid] = p.xf.gamma[tid];
        xfl[192 + tid] = p.xf.beta[tid];
    }
    __syncthreads();

    const CropMap cm = p.cm;
    const long slot = (long)blockIdx.x * PJ_WAVES + wave;
    const int role = (int)(slot % PJ_ROLES);
    const long wslot = slot / PJ_ROLES, nwslots = ((long)gridDim.x * PJ_WAVES) / PJ_ROLES;
    const long nlines = (long)p.B * cm.T * cm.H;
    const int TQ = (cm.W + 31) >> 5;
    const bool xgelu = p.xf.gelu != 0, silu = p.act == 1;
    const u32x4* Wb = W1B + role * (2 * 3 * PJ_NT * 64) + lane;

    // per-lane constants of the lane's hidden units PJ_HB role + PJ_NT n16 + t and of its 4 channels 4 n16 + u (B' operand)
    float b1v[PJ_NT], w2v[DOT][PJ_NT], dw2[DOT][PJ_NT], db1[PJ_NT], db2[DOT];
#pragma unroll
    for (int t = 0; t < PJ_NT; ++t) {
        b1v[t] = p.b1[PJ_HB * role + PJ_NT * n16 + t];
        db1[t] = 0.f;
#pragma unroll
        for (int j = 0; j < DOT; ++j) {
            w2v[j][t] = j < DO ? p.w2[j * PJ_HID + PJ_HB * role + PJ_NT * n16 + t] : 0.f;
            dw2[j][t] = 0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < DOT; ++j) db2[j] = 0.f;
    f32x4v acc3[PJ_NT][4];                          // d fc1: [hidden tile t][channel tile u]
#pragma unroll
    for (int t = 0; t < PJ_NT; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc3[t][u] = f32x4v{0.f, 0.f, 0.f, 0.f};

    for (long rc = wslot; rc < nlines; rc += nwslots) {
        const int h = (int)(rc % cm.H);
        const long r2 = rc / cm.H;
        const int t_ = (int)(r2 % cm.T);
        const long b = r2 / cm.T;
        const long gline = (b * cm.Tp + t_) * cm.Hp + h;
        const rsrc_t rx = make_rsrc(p.s + gline * cm.Wp * 64, (unsigned)cm.W * 256u);          // cells >= W read as 0
        const rsrc_t rg = make_rsrc(p.gout + rc * cm.W * DO, (unsigned)(cm.W * DO) * 4u);
        for (int q = 0; q < TQ; ++q) {
            asm volatile("" ::: "memory");
            // ---- a tile in A-operand layout (lane = cell 32 q + 16 j + n16): u = a W1^T + b1 for the wave's hidden units
            f32x4v acc[2][PJ_NT];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < PJ_NT; ++t) acc[j][t] = f32x4v{b1v[t], b1v[t], b1v[t], b1v[t]};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 Ah[2], Am[2], Al[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float v[8];
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int i = 2 * ks + hf;
                        const f32x4v xv = __builtin_bit_cast(f32x4v, ld16(rx, (32 * q + 16 * j + n16) * 256 + i * 64 + kg * 16));
                        const f32x4v mu = *reinterpret_cast<const f32x4v*>(xfl + 16 * i + 4 * kg);
                        const f32x4v is = *reinterpret_cast<const f32x4v*>(xfl + 64 + 16 * i + 4 * kg);
                        const f32x4v ga = *reinterpret_cast<const f32x4v*>(xfl + 128 + 16 * i + 4 * kg);
                        const f32x4v be = *reinterpret_cast<const f32x4v*>(xfl + 192 + 16 * i + 4 * kg);
                        f32x4v z = bn4(xv, mu, is, ga, be);
                        if (xgelu) z = gelu4(z);
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[4 * hf + c] = z[c];
                    }
                    split8(v, Ah[j], Am[j], Al[j]);
                }
#pragma unroll
                for (int t = 0; t < PJ_NT; ++t) {
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, Wb[((ks * 3 + 0) * PJ_NT + t) * 64]);
                    const bf16x8 bm = __builtin_bit_cast(bf16x8, Wb[((ks * 3 + 1) * PJ_NT + t) * 64]);
                    const bf16x8 bl = __builtin_bit_cast(bf16x8, Wb[((ks * 3 + 2) * PJ_NT + t) * 64]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) { PJ_MAC6(acc[j][t], Ah[j], Am[j], Al[j], bh, bm, bl) }
                }
            }
            // ---- gh = (fc2^T gout) * act'(u) in place; d fc2, d b1, d b2 are per-lane sums over the lane's cells 32 q + 16 j + 4 mg + r
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float go[DOT];
#pragma unroll
                    for (int jj = 0; jj < DOT; ++jj) {
                        go[jj] = jj < DO ? buf_load_f32(rg, ((32 * q + 16 * j + 4 * kg + r) * DO + jj) * 4, 0) : 0.f;
                        db2[jj] += go[jj];
                    }
#pragma unroll
                    for (int t = 0; t < PJ_NT; ++t) {
                        float vv, d;
                        act_pair(acc[j][t][r], silu, vv, d);
                        float gp = 0.f;
#pragma unroll
                        for (int jj = 0; jj < DOT; ++jj) {
                            gp += go[jj] * w2v[jj][t];
                            dw2[jj][t] += go[jj] * vv;
                        }
                        const float ghv = gp * d;                  // cells >= W: gout == 0 -> gh == 0
                        acc[j][t][r] = ghv;
                        db1[t] += ghv;
                    }
                }
            // ---- d fc1 += gh^T a: contraction over the 32 cells; lane group kg holds cells {4 kg + r, 16 + 4 kg + r}
            bf16x8 Gh[PJ_NT], Gm[PJ_NT], Gl[PJ_NT];
#pragma unroll
            for (int t = 0; t < PJ_NT; ++t) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[0][t][r];
                    v[4 + r] = acc[1][t][r];
                }
                split8(v, Gh[t], Gm[t], Gl[t]);
            }
            {
                f32x4v xr[8];                                      // a in B-operand layout: column = channel 4 n16 + u
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    xr[e] = __builtin_bit_cast(f32x4v, ld16(rx, (32 * q + 16 * (e >> 2) + 4 * kg + (e & 3)) * 256 + n16 * 16));
                const f32x4v bmu = *reinterpret_cast<const f32x4v*>(xfl + 4 * n16), bis = *reinterpret_cast<const f32x4v*>(xfl + 64 + 4 * n16);
                const f32x4v bga = *reinterpret_cast<const f32x4v*>(xfl + 128 + 4 * n16), bbe = *reinterpret_cast<const f32x4v*>(xfl + 192 + 4 * n16);
#pragma unroll
                for (int e = 0; e < 8; ++e) {       // rows of cells >= W read as 0 but xf(0) != 0: their gh is 0, so the product vanishes anyway
                    xr[e] = bn4(xr[e], bmu, bis, bga, bbe);
                    if (xgelu) xr[e] = gelu4(xr[e]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = xr[e][u];
                    bf16x8 Xh, Xm, Xl;
                    split8(v, Xh, Xm, Xl);
#pragma unroll
                    for (int t = 0; t < PJ_NT; ++t) { PJ_MAC6(acc3[t][u], Gh[t], Gm[t], Gl[t], Xh, Xm, Xl) }
                }
            }
        }
    }
    // ---- partial row of this wave.  acc3[t][u][r]: hidden PJ_HB role + PJ_NT (4 mg + r) + t (row), channel 4 n16 + u (column)
    float* part = p.part + slot * (long)PJ_WROW(DO);
#pragma unroll
    for (int t = 0; t < PJ_NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x4v o;
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] = acc3[t][u][r];
            *reinterpret_cast<f32x4v*>(part + (PJ_NT * (4 * kg + r) + t) * 64 + 4 * n16) = o;
        }
    // per-lane sums over cells: lanes of different mg hold different cells of the same hidden units -> add over mg
#pragma unroll
    for (int t = 0; t < PJ_NT; ++t) {
        float s1 = db1[t];
        s1 += __shfl_xor(s1, 16, 64);
        s1 += __shfl_xor(s1, 32, 64);
        if (kg == 0) part[PJ_HB * 64 + DO * PJ_HB + PJ_NT * n16 + t] = s1;
#pragma unroll
        for (int j = 0; j < DOT; ++j) {
            if (j < DO) {
                float s2 = dw2[j][t];
                s2 += __shfl_xor(s2, 16, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (kg == 0) part[PJ_HB * 64 + j * PJ_HB + PJ_NT * n16 + t] = s2;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < DOT; ++j) {
        if (j < DO) {
            float s3 = db2[j];                       // identical on the 16 lanes of a group; groups hold different cells
            s3 += __shfl_xor(s3, 16, 64);
            s3 += __shfl_xor(s3, 32, 64);
            if (lane == 0) part[PJ_HB * 64 + DO * PJ_HB + PJ_HB + j] = role == 0 ? s3 : 0.f;
        }
    }
}

static size_t pjx_wgrad_lds() { return (size_t)(PJ_ROLES * 2 * 3 * PJ_NT * 64) * 16 + 256 * 4; }

extern "C" long rpb_proj_wgrad_slots(int B, int T, int H) {
    const long nlines = (long)B * T * H;
    long grid = rpb_num_cus();
    const long need = (PJ_ROLES * nlines + PJ_WAVES - 1) / PJ_WAVES;
    if (grid > need) grid = need;
    return grid * PJ_WAVES;                          // a multiple of PJ_ROLES: PJ_WAVES is
}

extern "C" int rpb_proj_wgrad_row(int DO) { return PJ_WROW(DO); }
extern "C" int rpb_proj_wgrad_roles(void) { return PJ_ROLES; }

// part [rpb_proj_wgrad_slots][rpb_proj_wgrad_row(DO)]: per-wave partial sums; row `slot` covers hidden units
// [PJ_HB * (slot % roles), + PJ_HB) (layout above); the caller sums rows of equal role (rpb_reduce_partials with a row stride of
// `roles` rows)
extern "C" int rpb_proj_wgrad(const float* s, const float* w1, const float* b1, const float* w2, const float* gout, float* part,
                              int B, int DO, int T, int H, int W, int Tp, int Hp, int Wp, const float* xf_mean,
                              const float* xf_invstd, const float* xf_gamma, const float* xf_beta, int xf_gelu, int act,
                              void* stream) {
    RPB_REQUIRE(s && w1 && b1 && w2 && gout && part && xf_mean && xf_invstd && xf_gamma && xf_beta, "proj_wgrad: null pointer");
    RPB_REQUIRE(rpb_proj_bwd_fused_supported(64, DO, W, Wp), "proj_wgrad: unsupported shape (DO=%d W=%d Wp=%d)", DO, W, Wp);
    PjxArgs p{};
    p.s = s; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.gout = gout; p.part = part; p.B = B; p.DO = DO; p.act = act;
    p.cm = CropMap{T, H, W, Tp, Hp, Wp};
    p.xf = XForm{xf_mean, xf_invstd, xf_gamma, xf_beta, xf_gelu};
    const int grid = (int)(rpb_proj_wgrad_slots(B, T, H) / PJ_WAVES);
    const size_t lds = pjx_wgrad_lds();
    if (DO <= 2) {
        (void)hipFuncSetAttribute((const void*)pjx_wgrad_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(pjx_wgrad_kernel<2>, dim3(grid), dim3(PJ_WAVES * 64), lds, (hipStream_t)stream, p);
    } else {
        (void)hipFuncSetAttribute((const void*)pjx_wgrad_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(pjx_wgrad_kernel<4>, dim3(grid), dim3(PJ_WAVES * 64), lds, (hipStream_t)stream, p);
    }
    RPB_CHECK_LAUNCH("proj_wgrad");
}

// ------------------------------------------------------------------------------------------------------------ head entry points
// (called by rpb_proj_fwd / rpb_proj_bwd in rpb_proj.hip when the shape is covered; same arguments, same partial-row layout)
// (the gu producer keeps 32 (1 + DO) gradient accumulators per lane: two fc2 outputs fit the register file, four spill)
bool rpb_pjx_head_supported(int C, int DO, bool bwd) { return !pjx_off() && C == 64 && DO >= 1 && DO <= (bwd ? 2 : 16); }

long rpb_pjx_head_slots(int B, int T, int H, bool bwd) {
    const long GL = (long)B * T * H;
    long grid = (bwd ? 1L : 2L) * rpb_num_cus();          // forward: 122 registers, 60 KB of LDS -> two workgroups per CU
    const long need = (GL + PJ_WAVES - 1) / PJ_WAVES;
    if (grid > need) grid = need;
    return grid * PJ_WAVES;
}

int rpb_pjx_head_launch(bool bwd, const float* s, const float* w1, const float* b1, const float* w2, const float* b2, const float* gout,
                        float* out, float* gu, float* part, long part_rows, int DO, int T, int H, int W, int Tp, int Hp, int Wp, long ncrop,
                        const XForm& xf, int act, hipStream_t st, bool a_bf16) {
    if (!bwd && rpb_pjh_supported(64, DO, act, xf, a_bf16))                       // the evaluation forward, third organisation (csrc/rpb_pjh.hip)
        return rpb_pjh_launch(s, w1, b1, w2, b2, out, (int)(ncrop / ((long)T * H * W)), DO, T, H, W, Tp, Hp, Wp, xf, st, false, a_bf16);
    PjhArgs p{};
    p.s = s; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.gout = gout; p.out = out; p.gu = gu; p.part = part;
    p.B = (int)(ncrop / ((long)T * H * W)); p.DO = DO; p.act = act;
    p.cm = CropMap{T, H, W, Tp, Hp, Wp};
    p.xf = xf;
    long slots = rpb_pjx_head_slots(p.B, T, H, bwd);
    if (bwd && slots > part_rows) slots = part_rows / PJ_WAVES * PJ_WAVES;       // never more partial rows than the caller allocated
    RPB_REQUIRE(slots >= PJ_WAVES, "proj (bf16 pipe): no partial rows");
    const int grid = (int)(slots / PJ_WAVES);
    if (bwd && part_rows > slots)                                                 // rows this launch does not write
        (void)hipMemsetAsync(part + slots * ((long)DO * PJ_HID + PJ_HID + DO), 0, (size_t)(part_rows - slots) * ((long)DO * PJ_HID + PJ_HID + DO) * 4, st);
    const int dot = DO <= 2 ? 2 : (DO <= 4 ? 4 : 16);
    const size_t lds = pjx_head_lds(dot);
    if (a_bf16 && bwd) RPB_FAIL(RPB_ERR_UNSUPPORTED, "proj (bf16 pipe): bf16 activation storage is a forward path");
#define RPB_PJH(BWD_, D_)                                                                                                    \
    if (bwd == BWD_ && dot == D_) {                                                                                          \
        (void)hipFuncSetAttribute((const void*)pjx_head_kernel<BWD_, D_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((pjx_head_kernel<BWD_, D_>), dim3(grid), dim3(PJ_WAVES * 64), lds, st, p);                        \
    }
#define RPB_PJHB(D_)                                                                                                        \
    if (a_bf16 && dot == D_) {                                                                                               \
        (void)hipFuncSetAttribute((const void*)pjx_head_kernel<false, D_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((pjx_head_kernel<false, D_, true>), dim3(grid), dim3(PJ_WAVES * 64), lds, st, p);                 \
        RPB_CHECK_LAUNCH("proj (bf16 pipe, bf16 storage)");                                                                  \
    }
    RPB_PJHB(2) RPB_PJHB(4) RPB_PJHB(16)
#undef RPB_PJHB
    RPB_PJH(false, 2) RPB_PJH(false, 4) RPB_PJH(false, 16) RPB_PJH(true, 2)
#undef RPB_PJH
    RPB_CHECK_LAUNCH("proj (bf16 pipe)");
}
