// Weight / bias gradient of the 1x1x1 Conv3d at C = 128 on the bf16 matrix pipe ("cwx"; configs/fsi/fno.yaml: width 128):
//     dW[co][ci] = sum_cells gs[cell][co] * a[cell][ci],   db[co] = sum_cells gs[cell][co],   a = act(BN(x)) applied on load
// (autograd of fno.py:115 for layers l >= 1; rpb_cell_wgrad's (128, 128) instance without the crop -- the fp32-MFMA kernel of
// csrc/rpb_cell.hip runs it at 61-69 TF/s, 1.8-2.0 ms per launch at the fsi shape, five launches per step).
//
// The product contracts over CELLS, so both factors are wanted as "8 cells of one channel per lane".  A lane (n = lane & 15,
// mg = lane >> 4) loads, for the 32 cells of a tile, 16 B = channels 4 n .. 4 n + 3 (of one 64-channel half) of cells 16 j + 4 mg + r:
// 8 loads per factor, each instruction 4 whole 256 B half rows -- the accumulator layout of csrc/rpb_cmx.hip, whose values for one channel
// sub-index u are exactly an MFMA operand with k <-> cell.  Both factors are split into three bf16 planes in registers (truncation
// splits, six products per fp32 product, fp32 accumulate: the arithmetic of rpb_cmx.hip) and a wave accumulates ONE 64 x 64 quadrant of
// dW (16 tiles of 16 x 16, 96 MFMAs per 32 cells, 64 accumulator registers).  A workgroup is two tile streams of four quadrant waves;
// the four waves of a stream read the same two cell tiles (the second to fourth read out of L1 / L2) and each factor half is split by
// two of them.  Partial rows as rpb_cell_wgrad's: [stream][128 * 128 + 128], summed by rpb_reduce_partials.
#include "rpb_cmx.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

#ifndef RPB_CWX_AUX
#define RPB_CWX_AUX 0   /* cache policy of the tile loads (2 = nt): experiment switch */
#endif
__device__ __forceinline__ u32x4 ld16(rsrc_t r, int voff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, RPB_CWX_AUX));
}
__device__ __forceinline__ float asf(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ float trunc_bf16(float v) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_hi(float a, float b) {
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
// 8 fp32 -> three bf16x8 planes (exact: hi + mid + lo == v)
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

struct CwxArgs {
    const float* gs;   // [ncell][128]
    const float* x;    // [ncell][128]
    float* part;       // [2 * gridDim.x][128 * 128 + 128]
    long ncell;
    XForm xf;          // lazy BatchNorm (+ GELU) of x, or mean == null
    int crop;          // cwx128s only: gs is [ncell = B T H W][128] over the CROP, x the padded [B Tp Hp Wp][128] tensor (fc1, fno.py:121); W % 32 == 0
    CropMap cm;
};

__global__ __launch_bounds__(512) void cwx128_kernel(CwxArgs a) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, mg = lane >> 4;
    const int stream = wave >> 2, oh = (wave >> 1) & 1, ih = wave & 1;     // quadrant (out half, in half) of this wave
    const long ntiles = (a.ncell + 31) >> 5;
    const long nstreams = (long)gridDim.x * 2, slot = (long)blockIdx.x * 2 + stream;
    const bool has_xf = a.xf.mean != nullptr, xgelu = a.xf.gelu != 0;
    f32x2 sc[2], be[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int c = 64 * ih + 4 * n + 2 * p;
        sc[p] = has_xf ? f32x2{a.xf.invstd[c] * a.xf.gamma[c], a.xf.invstd[c + 1] * a.xf.gamma[c + 1]} : pk2(1.f);
        be[p] = has_xf ? f32x2{a.xf.beta[c], a.xf.beta[c + 1]} - f32x2{a.xf.mean[c], a.xf.mean[c + 1]} * sc[p] : pk2(0.f);     // x * sc + be
    }
    const f32x4v z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4v accW[4][4];          // tile (uo, ui): row 4 mg + r <-> out channel 64 oh + 4 (4 mg + r) + uo, column n <-> in channel 64 ih + 4 n + ui
#pragma unroll
    for (int uo = 0; uo < 4; ++uo)
#pragma unroll
        for (int ui = 0; ui < 4; ++ui) accW[uo][ui] = z4;
    f32x4v bs = z4;             // sum of gs over this lane's cells, channels 64 oh + 4 n + u
    u32x4 gb[2][4], xb[2][4];
    const int ooff = (4 * mg) * 512 + n * 16;
    // rows past the end of the tensor lie outside the descriptors -> 0 (gs = 0 removes them from every sum)
    // Register budget (two waves per SIMD: 256): the X planes (48) stay resident for the 96 products; the G planes are split per channel
    // sub-index just before their 24 products (12 at a time), so the raw gs image is live until the end and ITS next tile is requested
    // last -- the x image of the next tile is requested as soon as the X planes exist and flies during all 96 products, and the next
    // iteration starts with the ~400 vector instructions on x, which cover the gs round trip.
    auto issue_x = [&](long t) {
        const long left = a.ncell - t * 32;
        const unsigned bytes = (unsigned)(left < 32 ? left : 32) * 512u;
        const rsrc_t rx = make_rsrc(a.x + t * (32 * 128) + 64 * ih, bytes - 256u * (unsigned)ih);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) xb[j][r] = ld16(rx, ooff + j * (16 * 512) + r * 512);
    };
    auto issue_g = [&](long t) {
        const long left = a.ncell - t * 32;
        const unsigned bytes = (unsigned)(left < 32 ? left : 32) * 512u;
        const rsrc_t rg = make_rsrc(a.gs + t * (32 * 128) + 64 * oh, bytes - 256u * (unsigned)oh);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) gb[j][r] = ld16(rg, ooff + j * (16 * 512) + r * 512);
    };
    long t = slot;
    if (t < ntiles) {
        issue_x(t);
        issue_g(t);
    }
    while (t < ntiles) {
        bf16x8 Xh[4], Xm[4], Xl[4];
        {                                   // channel pairs: packed fp32 math for the lazy BatchNorm + GELU, the two pairs' erf in lock-step
            float v0[8], v1[8], v2[8], v3[8];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const f32x4v xv = __builtin_bit_cast(f32x4v, xb[j][r]);
                    f32x2 za = f32x2{xv[0], xv[1]}, zb = f32x2{xv[2], xv[3]};
                    if (has_xf) {
                        za = pk_fma(za, sc[0], be[0]);
                        zb = pk_fma(zb, sc[1], be[1]);
                        if (xgelu) gelu2x2(za, zb);
                    }
                    v0[4 * j + r] = za[0];
                    v1[4 * j + r] = za[1];
                    v2[4 * j + r] = zb[0];
                    v3[4 * j + r] = zb[1];
                }
            split8(v0, Xh[0], Xm[0], Xl[0]);
            split8(v1, Xh[1], Xm[1], Xl[1]);
            split8(v2, Xh[2], Xm[2], Xl[2]);
            split8(v3, Xh[3], Xm[3], Xl[3]);
        }
        const long tn = t + nstreams;
        if (tn < ntiles) issue_x(tn);
#pragma unroll
        for (int uo = 0; uo < 4; ++uo) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * j + r] = __builtin_bit_cast(f32x4v, gb[j][r])[uo];
            if (ih == 0) bs[uo] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            bf16x8 gh, gm, gl;
            split8(v, gh, gm, gl);
#define CWX_W(AP, BP) _Pragma("unroll") for (int ui = 0; ui < 4; ++ui) accW[uo][ui] = mfma16(AP, BP[ui], accW[uo][ui]);
            CWX_W(gh, Xl) CWX_W(gl, Xh) CWX_W(gm, Xm) CWX_W(gh, Xm) CWX_W(gm, Xh) CWX_W(gh, Xh)
#undef CWX_W
        }
        if (tn < ntiles) issue_g(tn);
        t = tn;
    }
    float* wp = a.part + slot * (long)(128 * 128 + 128);
#pragma unroll
    for (int uo = 0; uo < 4; ++uo)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = 64 * oh + 4 * (4 * mg + r) + uo;
            *reinterpret_cast<f32x4v*>(wp + (long)o * 128 + 64 * ih + 4 * n) = f32x4v{accW[uo][0][r], accW[uo][1][r], accW[uo][2][r], accW[uo][3][r]};
        }
    if (ih == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float s = bs[u];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            bs[u] = s;
        }
        if (mg == 0) *reinterpret_cast<f32x4v*>(wp + 128 * 128 + 64 * oh + 4 * n) = bs;
    }
}

// ---- second organisation (default; RPB_CWX_SHARED=0 keeps the kernel above): ONE tile stream per workgroup of four quadrant waves, two
// workgroups per CU.  In the kernel above the two waves that share an input half (oh = 0, 1) each run the lazy BatchNorm + GELU and the
// three-plane split over the same 32 x 64 values, and the two that share an output half split the same gs values: 750 of a wave's ~780
// vector instructions per tile are done twice, on a kernel that is vector-issue bound (two waves per SIMD: 2.6 cycles per scalar, 4.5
// per packed instruction -> ~5.3 k cycles per tile pair and SIMD next to 3.1 k matrix cycles; measured 10 k).  Here wave (oh, ih)
//   * transforms and splits the channel PAIR oh of each lane's four x channels (8-byte loads: channels 4 n + 2 oh, + 1 of half ih),
//   * splits the channel pair ih of each lane's four gs channels (8-byte loads: channels 4 n + 2 ih, + 1 of half oh),
// stores its 12 plane registers to LDS in operand order ([ih][ui][plane][lane] and [oh][uo][plane][lane], 16 B per lane: conflict-free)
// and reads the other pairs' 12 from its partners (1 - oh, ih) and (oh, 1 - ih).  One buffer of 48 KB: a barrier between the stores and
// the reads, and one behind the reads (the waves have just met, it costs its own latency); both wait for LDS only -- the next tile's
// loads, issued as soon as the images are consumed, stay in flight across them.  Same products in the same order: results bit-equal.
__global__ __launch_bounds__(256, 2) void cwx128s_kernel(CwxArgs a) {
    extern __shared__ u32x4 xpl[];           // X planes [ih 2][ui 4][plane 3][lane 64], then G planes [oh 2][uo 4][plane 3][lane 64]
    typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, mg = lane >> 4;
    const int oh = (wave >> 1) & 1, ih = wave & 1;
    const long ntiles = (a.ncell + 31) >> 5;
    const long nstreams = (long)gridDim.x, slot = (long)blockIdx.x;
    const bool has_xf = a.xf.mean != nullptr, xgelu = a.xf.gelu != 0;
    f32x2 sc, be;
    {
        const int c = 64 * ih + 4 * n + 2 * oh;
        sc = has_xf ? f32x2{a.xf.invstd[c] * a.xf.gamma[c], a.xf.invstd[c + 1] * a.xf.gamma[c + 1]} : pk2(1.f);
        be = has_xf ? f32x2{a.xf.beta[c], a.xf.beta[c + 1]} - f32x2{a.xf.mean[c], a.xf.mean[c + 1]} * sc : pk2(0.f);
    }
    const f32x4v z4 = {0.f, 0.f, 0.f, 0.f};
    // tile (p, q): rows <-> out channel 64 oh + 4 (4 mg + r) + uo(p), columns <-> in channel 64 ih + 4 n + ui(q); p, q = 0, 1: the pair this
    // wave prepared (uo = 2 ih + p, ui = 2 oh + q), p, q = 2, 3: the partner's (uo = 2 (1 - ih) + p - 2, ui = 2 (1 - oh) + q - 2)
    f32x4v accW[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) accW[p][q] = z4;
    float bs[2] = {0.f, 0.f};  // sum of gs over this lane's cells, channels 64 oh + 4 n + 2 ih + (0, 1)
    u32x2v gb[2][4], xb[2][4];
    const int ooff = (4 * mg) * 512 + n * 16;
    auto issue = [&](long t) {
        const long left = a.ncell - t * 32;
        const unsigned bytes = (unsigned)(left < 32 ? left : 32) * 512u;
        long xc = t * 32;                        // first cell of the tile in x
        if (a.crop) {                            // a tile of 32 crop cells lies inside one w-row (W % 32 == 0): its cells are consecutive in the padded tensor
            const unsigned q = (unsigned)xc;     // ncell < 2^31 (rpb_cell_wgrad)
            const unsigned row = q / (unsigned)a.cm.W, w0 = q - row * (unsigned)a.cm.W;
            const unsigned r2 = row / (unsigned)a.cm.H, h = row - r2 * (unsigned)a.cm.H;
            const unsigned b = r2 / (unsigned)a.cm.T, tt = r2 - b * (unsigned)a.cm.T;
            xc = (((long)b * a.cm.Tp + tt) * a.cm.Hp + h) * (long)a.cm.Wp + w0;
        }
        const rsrc_t rx = make_rsrc(a.x + xc * 128 + 64 * ih, bytes - 256u * (unsigned)ih);
        const rsrc_t rg = make_rsrc(a.gs + t * (32 * 128) + 64 * oh, bytes - 256u * (unsigned)oh);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xb[j][r] = __builtin_amdgcn_raw_buffer_load_b64(rx, ooff + 8 * oh + j * (16 * 512) + r * 512, 0, 0);
                gb[j][r] = __builtin_amdgcn_raw_buffer_load_b64(rg, ooff + 8 * ih + j * (16 * 512) + r * 512, 0, 0);
            }
    };
    u32x4* xw = xpl + (ih * 4) * 3 * 64 + lane;
    u32x4* gw = xpl + (8 + oh * 4) * 3 * 64 + lane;
    long t = slot;
    if (t < ntiles) issue(t);
    while (t < ntiles) {                     // the four waves of a workgroup walk the same tiles: the barriers below are uniform
        bf16x8 Xh[4], Xm[4], Xl[4], Gh[4], Gm[4], Gl[4];
        {
            float v0[8], v1[8], g0[8], g1[8];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {         // two cells in lock-step (the erf polynomials of two packed pairs interleave)
                    // (by-value helper: __builtin_bit_cast(float, v[1]) on a vector ELEMENT reads element 0 with hipcc 7.2)
                    f32x2 za = f32x2{asf(xb[j][r][0]), asf(xb[j][r][1])};
                    f32x2 zb = f32x2{asf(xb[j][r + 1][0]), asf(xb[j][r + 1][1])};
                    if (has_xf) {
                        za = pk_fma(za, sc, be);
                        zb = pk_fma(zb, sc, be);
                        if (xgelu) gelu2x2(za, zb);
                    }
                    v0[4 * j + r] = za[0];
                    v1[4 * j + r] = za[1];
                    v0[4 * j + r + 1] = zb[0];
                    v1[4 * j + r + 1] = zb[1];
                }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    g0[4 * j + r] = asf(gb[j][r][0]);
                    g1[4 * j + r] = asf(gb[j][r][1]);
                }
            bs[0] += ((g0[0] + g0[1]) + (g0[2] + g0[3])) + ((g0[4] + g0[5]) + (g0[6] + g0[7]));
            bs[1] += ((g1[0] + g1[1]) + (g1[2] + g1[3])) + ((g1[4] + g1[5]) + (g1[6] + g1[7]));
            split8(v0, Xh[0], Xm[0], Xl[0]);
            split8(v1, Xh[1], Xm[1], Xl[1]);
            split8(g0, Gh[0], Gm[0], Gl[0]);
            split8(g1, Gh[1], Gm[1], Gl[1]);
        }
        const long tn = t + nstreams;
        if (tn < ntiles) issue(tn);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            xw[((2 * oh + e) * 3 + 0) * 64] = __builtin_bit_cast(u32x4, Xh[e]);
            xw[((2 * oh + e) * 3 + 1) * 64] = __builtin_bit_cast(u32x4, Xm[e]);
            xw[((2 * oh + e) * 3 + 2) * 64] = __builtin_bit_cast(u32x4, Xl[e]);
            gw[((2 * ih + e) * 3 + 0) * 64] = __builtin_bit_cast(u32x4, Gh[e]);
            gw[((2 * ih + e) * 3 + 1) * 64] = __builtin_bit_cast(u32x4, Gm[e]);
            gw[((2 * ih + e) * 3 + 2) * 64] = __builtin_bit_cast(u32x4, Gl[e]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // LDS only: the loads of the next tile stay in flight
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            Xh[2 + e] = __builtin_bit_cast(bf16x8, xw[((2 * (1 - oh) + e) * 3 + 0) * 64]);
            Xm[2 + e] = __builtin_bit_cast(bf16x8, xw[((2 * (1 - oh) + e) * 3 + 1) * 64]);
            Xl[2 + e] = __builtin_bit_cast(bf16x8, xw[((2 * (1 - oh) + e) * 3 + 2) * 64]);
            Gh[2 + e] = __builtin_bit_cast(bf16x8, gw[((2 * (1 - ih) + e) * 3 + 0) * 64]);
            Gm[2 + e] = __builtin_bit_cast(bf16x8, gw[((2 * (1 - ih) + e) * 3 + 1) * 64]);
            Gl[2 + e] = __builtin_bit_cast(bf16x8, gw[((2 * (1 - ih) + e) * 3 + 2) * 64]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // the planes are in registers: the buffer is free for the next tile
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#define CWX_W(AP, BP) _Pragma("unroll") for (int q = 0; q < 4; ++q) accW[p][q] = mfma16(AP[p], BP[q], accW[p][q]);
            CWX_W(Gh, Xl) CWX_W(Gl, Xh) CWX_W(Gm, Xm) CWX_W(Gh, Xm) CWX_W(Gm, Xh) CWX_W(Gh, Xh)
#undef CWX_W
        }
        t = tn;
    }
    float* wp = a.part + slot * (long)(128 * 128 + 128);
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = 64 * oh + 4 * (4 * mg + r) + (p ^ (2 * ih));
            const f32x4v own_first = {accW[p][0][r], accW[p][1][r], accW[p][2][r], accW[p][3][r]};
            const f32x4v partner_first = {accW[p][2][r], accW[p][3][r], accW[p][0][r], accW[p][1][r]};
            *reinterpret_cast<f32x4v*>(wp + (long)o * 128 + 64 * ih + 4 * n) = oh ? partner_first : own_first;
        }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        float s_ = bs[u];
        s_ += __shfl_xor(s_, 16, 64);
        s_ += __shfl_xor(s_, 32, 64);
        bs[u] = s_;
    }
    if (mg == 0) {
        wp[128 * 128 + 64 * oh + 4 * n + 2 * ih] = bs[0];
        wp[128 * 128 + 64 * oh + 4 * n + 2 * ih + 1] = bs[1];
    }
}

static bool cwx_shared_off() {
    static const bool off = getenv("RPB_CWX_SHARED") && atoi(getenv("RPB_CWX_SHARED")) == 0;
    return off;
}
// crop (the fc1 weight gradient: gs over the crop, x padded): the shared-plane kernel only, and only when a 32-cell tile cannot straddle w-rows
bool rpb_cwx128_supported(long ncell, int CO, int CI, int crop, int W) {
    static const bool off = getenv("RPB_CELL_WGRAD_128_F32") && atoi(getenv("RPB_CELL_WGRAD_128_F32")) == 1;     // the fp32-MFMA kernel
    if (crop && (cwx_shared_off() || W <= 0 || W % 32 != 0)) return false;
    return !off && CO == 128 && CI == 128 && ncell > 0;
}

// slots = partial rows the caller allocated (rpb_cell_wgrad_slots: even, two streams per workgroup)
int rpb_cwx128_launch(const float* gs, const float* x, float* part, long ncell, long slots, const XForm& xf, int crop, const CropMap& cm,
                      hipStream_t st) {
    RPB_REQUIRE(slots >= 2 && slots % 2 == 0, "cell_wgrad (bf16 pipe, C = 128): %ld partial rows", slots);
    CwxArgs a;
    a.gs = gs; a.x = x; a.part = part; a.ncell = ncell; a.xf = xf; a.crop = crop; a.cm = cm;
    if (!cwx_shared_off()) {
        const int lds = 2 * 2 * 4 * 3 * 64 * 16;      // X and G planes of one tile
        hipLaunchKernelGGL(cwx128s_kernel, dim3((unsigned)slots), dim3(256), lds, st, a);
        RPB_CHECK_LAUNCH("cell_wgrad(bf16x3, C = 128, shared planes)");
    }
    hipLaunchKernelGGL(cwx128_kernel, dim3((unsigned)(slots / 2)), dim3(512), 0, st, a);
    RPB_CHECK_LAUNCH("cell_wgrad(bf16x3, C = 128)");
}
