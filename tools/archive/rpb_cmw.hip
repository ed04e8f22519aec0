// Backward cell_mix of one Fourier layer at C = 64 WITH the layer's 1x1-conv weight gradient ("cmw"): the STATS == 2 launch of
// csrc/rpb_cmx.hip -- g_x = gs Wc + FW^T z2, gz = g_x * act'(BN(s_prev)), the BatchNorm-backward sums of the layer below
// (autograd of fno.py:63,115-119) -- plus
//
//   dWc[co][ci] = sum_cells gs[cell][co] * act(BN(s_prev))[cell][ci]            (autograd of fno.py:115, Conv3d weight)
//
// Why here: rounds 1-3 formed dWc inside bn_bwd_row, which read the layer input s_prev (3.8 GB at B = 32) for nothing else;
// this launch streams both factors anyway -- gs as its contraction operand, s_prev for act' -- so the row kernel drops a whole
// tensor read (16.2 -> 12.4 GB per layer) and one evaluation of erf per element and step disappears (act and act' share it here).
//
// The weight gradient contracts over CELLS, the channel mixing over CHANNELS, so gs is needed in two register images:
//  * A layout (lane = cell, 16 B = 4 channels of the cell's row; csrc/rpb_cmx.hip): the A operand of the channel mixing;
//  * accumulator layout (lane (n, mg) = channels 4 n .. 4 n + 3 of cells {16 j + 4 mg + r}): 8 cells per lane = the K = 32 operand of
//    the weight gradient.  s_prev is loaded in exactly that layout for the epilogue already (it multiplies the accumulators), so
//    act(BN(s_prev)) IS the B operand; gs is fetched a second time in this layout (the tile was requested a few microseconds earlier
//    by the same wave: L2 hits, rpb_pjf.hip does the same with its activation tile) and split once more.
//
// One wave per SIMD: the 64 x 64 fp32 accumulator of dWc takes 64 registers per lane on top of cmx's ~250, so a wave gets the
// whole 512-register budget and hides latency by itself: all three input images of the NEXT tile (24 KB per wave, 96 KB per CU)
// are in flight while the current tile computes.  Four waves per CU walk whole (b,t,h) lines like cmx.
#include "rpb_cmx.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

#define CMW_WAVES 4

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
// 8 fp32 -> three bf16x8 planes (exact: hi + mid + lo == v)
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
    u32x4 uh, um, ul;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = v[2 * q], b = v[2 * q + 1];
        uh[q] = pack_hi(a, b);
        const float ra = a - trunc_bf16(a), rb = b - trunc_bf16(b);
        um[q] = pack_hi(ra, rb);
        const float sa = ra - trunc_bf16(ra), sb = rb - trunc_bf16(rb);
        ul[q] = pack_hi(sa, sb);
    }
    h = __builtin_bit_cast(bf16x8, uh);
    m = __builtin_bit_cast(bf16x8, um);
    l = __builtin_bit_cast(bf16x8, ul);
}
__device__ __forceinline__ f32x4v mfma16(bf16x8 a, bf16x8 b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
}  // namespace

// GELU: the layer below has the GELU (every layer but the last);  WGZ: store gz = g_x * act'(z) (the row kernel below then skips act')
// The tile body is straight-line code (one wave per SIMD: nobody hides a branch's drain): cells past the line end are handled by
// data, not by control flow -- their gs / s_prev loads return 0 (line-clipped descriptors), their spectral operand row is a zero row
// appended to the stage matrix, so g_x = gz = 0 there and neither the sums nor the weight gradient see them; their stores are dropped.
template <bool GELU, bool WGZ>
__global__ __launch_bounds__(CMW_WAVES * 64, 1) void cmw_kernel(CmxArgs a) {
    extern __shared__ u32x4 lds4[];
    const int Wp = a.Wp, K2 = a.K2;
    u32x4* Bw = lds4;                        // [ks 2][plane 3][t 4][lane 64]   conv weights, B-operand order
    u32x4* GWs = Bw + 24 * 64;               // [plane 3][w Wp + 1][kg 4]       last-stage DFT matrix, A-operand rows; row Wp = zeros
    u32x4* Zs = GWs + 3 * (Wp + 1) * 4;      // [wave][plane 3][t 4][lane 64]   the current line's z2 row, B-operand order
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kg = lane >> 4;         // A role: cell row m, k group kg;  B / D role: column n = m, row group mg = kg

    // ---- per-workgroup operand preparation (as csrc/rpb_cmx.hip)
    for (int idx = tid; idx < 2 * 4 * 64; idx += blockDim.x) {
        const int l = idx & 63, t = (idx >> 6) & 3, ks = idx >> 8;
        const int n = l & 15, kgb = l >> 4;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = 16 * (2 * ks + (e >> 2)) + 4 * kgb + (e & 3);
            const int co = 4 * n + t;
            v[e] = a.transpose_w ? a.Wm[ci * 64 + co] : a.Wm[co * 64 + ci];
        }
        bf16x8 h, md, lo;
        split8(v, h, md, lo);
        Bw[((ks * 3 + 0) * 4 + t) * 64 + l] = __builtin_bit_cast(u32x4, h);
        Bw[((ks * 3 + 1) * 4 + t) * 64 + l] = __builtin_bit_cast(u32x4, md);
        Bw[((ks * 3 + 2) * 4 + t) * 64 + l] = __builtin_bit_cast(u32x4, lo);
    }
    for (int idx = tid; idx < (Wp + 1) * 4; idx += blockDim.x) {
        const int w = idx >> 2, kgw = idx & 3;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 8 * kgw + e;
            v[e] = (k < K2 && w < Wp) ? a.GW[k * Wp + w] : 0.f;
        }
        bf16x8 h, md, lo;
        split8(v, h, md, lo);
        GWs[(0 * (Wp + 1) + w) * 4 + kgw] = __builtin_bit_cast(u32x4, h);
        GWs[(1 * (Wp + 1) + w) * 4 + kgw] = __builtin_bit_cast(u32x4, md);
        GWs[(2 * (Wp + 1) + w) * 4 + kgw] = __builtin_bit_cast(u32x4, lo);
    }
    __syncthreads();

    // ---- per-lane constants of the output channels 4 n .. 4 n + 3 (the BatchNorm of the layer below)
    XParam bp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bp[t] = xf_load(a.bnb, 4 * m + t);
    f32x2 ssum[2], ssq[2];
    ssum[0] = ssum[1] = ssq[0] = ssq[1] = pk2(0.f);
    const f32x4v z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4v accW[4][4];                       // dWc tile (uo, ui): row 4 mg + r <-> out channel 4 (4 mg + r) + uo, column n <-> in channel 4 n + ui
#pragma unroll
    for (int uo = 0; uo < 4; ++uo)
#pragma unroll
        for (int ui = 0; ui < 4; ++ui) accW[uo][ui] = z4;

    // every line / tile index is wave-uniform: pinned to SGPRs (the 64-bit division lands in VGPRs, and a descriptor built from a
    // VGPR-resident index makes the compiler wrap every buffer access in a waterfall loop)
    const int G = __builtin_amdgcn_readfirstlane((int)((unsigned)a.ncell / (unsigned)Wp));
    const int TQ = (Wp + 31) >> 5;
    const int nslots = (int)gridDim.x * CMW_WAVES;
    const int slot = (int)blockIdx.x * CMW_WAVES + wave;
    const unsigned line_bytes = (unsigned)Wp * 256u;
    const long line_floats = (long)Wp * 64;
    const int xoff = m * 256 + kg * 16;                  // A layout: byte offset of the lane's first 16 B inside a 16-cell block
    const int ooff = (4 * kg) * 256 + m * 16;            // accumulator layout: cell 4 mg + r, channels 4 n ..

    u32x4 xa[2][4];                                      // gs, A layout
    u32x4 sp[2][4], gb[2][4];                            // s_prev and gs, accumulator layout (rows r of MFMA tile j)
    // ok == false (past the wave's last tile): an empty descriptor -- the loads return 0 without traffic, no branch in the tile body
    auto issue_x = [&](int g_, int q, int j, int ks, bool ok) {
        const long g = __builtin_amdgcn_readfirstlane(g_);
        const rsrc_t rx = make_rsrc(a.x + g * line_floats, ok ? line_bytes : 0u);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) xa[j][2 * ks + hf] = ld16(rx, q * 8192 + xoff + j * 4096 + (2 * ks + hf) * 64);
    };
    auto issue_sg = [&](int g_, int q, bool ok) {
        const long g = __builtin_amdgcn_readfirstlane(g_);
        const rsrc_t rs = make_rsrc(a.bnb_s + g * line_floats, ok ? line_bytes : 0u);
        const rsrc_t rx = make_rsrc(a.x + g * line_floats, ok ? line_bytes : 0u);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sp[j][r] = ld16(rs, q * 8192 + ooff + j * 4096 + r * 256);
                gb[j][r] = ld16(rx, q * 8192 + ooff + j * 4096 + r * 256);
            }
    };
    u32x4 zr[8];
    auto issue_z = [&](int g_) {        // z2 row in B-operand layout: lane (n, kg) holds k = 8 kg + e, channels 4 n .. 4 n + 3
        const long g = __builtin_amdgcn_readfirstlane(g_);
        const rsrc_t rz = make_rsrc(a.z2 + g * K2 * 64, (unsigned)K2 * 256u);
#pragma unroll
        for (int e = 0; e < 8; ++e) zr[e] = ld16(rz, (8 * kg + e) * 256 + m * 16);
    };

    if (slot < G) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            issue_x(slot, 0, j, 0, true);
            issue_x(slot, 0, j, 1, true);
        }
        issue_z(slot);
        issue_sg(slot, 0, true);
    }
    u32x4* Zw = Zs + wave * 12 * 64 + lane;
    for (int g = slot; g < G; g += nslots) {
        const int g_next = g + nslots < G ? g + nslots : 0;
        const rsrc_t ro = make_rsrc(a.out + (long)__builtin_amdgcn_readfirstlane(g) * line_floats, line_bytes);
        for (int q = 0; q < TQ; ++q) {
            const bool last = q + 1 == TQ;
            const int gn = last ? g_next : g;                                // next wave tile: (gn, qn)
            const int qn = last ? 0 : q + 1;
            const bool more = !last || g + nslots < G;
            asm volatile("" ::: "memory");   // keep the (tile-invariant) LDS operand reads inside the loop

            f32x4v acc[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[j][t] = z4;
            // ---- channel mixing: K = 64 = 2 steps of 32; the freed x registers take the next wave tile's loads
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 Ah[2], Am[2], Al[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float v[8];
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const f32x4v xv = __builtin_bit_cast(f32x4v, xa[j][2 * ks + hf]);
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[4 * hf + c] = xv[c];
                    }
                    split8(v, Ah[j], Am[j], Al[j]);
                }
                issue_x(gn, qn, 0, ks, more);
                issue_x(gn, qn, 1, ks, more);
                if (ks == 0) {
                    if (q == 0) {                  // new line: its z2 row (requested one tile ago) -> three bf16 planes per channel
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            float v[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = __builtin_bit_cast(f32x4v, zr[e])[t];
                            bf16x8 zh, zm, zl;
                            split8(v, zh, zm, zl);
                            Zw[(0 * 4 + t) * 64] = __builtin_bit_cast(u32x4, zh);
                            Zw[(1 * 4 + t) * 64] = __builtin_bit_cast(u32x4, zm);
                            Zw[(2 * 4 + t) * 64] = __builtin_bit_cast(u32x4, zl);
                        }
                    }
                    if (last && more) issue_z(gn);                           // next line's row: in flight for a whole tile
                }
#pragma unroll
                for (int tp = 0; tp < 2; ++tp) {
                    bf16x8 Bh[2], Bm[2], Bl[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int t = 2 * tp + u;
                        Bh[u] = __builtin_bit_cast(bf16x8, Bw[((ks * 3 + 0) * 4 + t) * 64 + lane]);
                        Bm[u] = __builtin_bit_cast(bf16x8, Bw[((ks * 3 + 1) * 4 + t) * 64 + lane]);
                        Bl[u] = __builtin_bit_cast(bf16x8, Bw[((ks * 3 + 2) * 4 + t) * 64 + lane]);
                    }
#define CMW_PROD(AP, BP)                                                     \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                            \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) acc[j][2 * tp + u] = mfma16(AP[j], BP[u], acc[j][2 * tp + u]);
                    CMW_PROD(Ah, Bl) CMW_PROD(Al, Bh) CMW_PROD(Am, Bm) CMW_PROD(Ah, Bm) CMW_PROD(Am, Bh) CMW_PROD(Ah, Bh)
#undef CMW_PROD
                }
            }
            // ---- adjoint of the forward W stage: A = FW^T row of the cell's w (LDS), B = the line's z2 planes (the wave's LDS slice)
            {
                bf16x8 ah[2], am[2], al[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    int wl = 32 * q + 16 * j + m;
                    wl = wl < Wp ? wl : Wp;                                   // past the line end: the zero row
                    ah[j] = __builtin_bit_cast(bf16x8, GWs[(0 * (Wp + 1) + wl) * 4 + kg]);
                    am[j] = __builtin_bit_cast(bf16x8, GWs[(1 * (Wp + 1) + wl) * 4 + kg]);
                    al[j] = __builtin_bit_cast(bf16x8, GWs[(2 * (Wp + 1) + wl) * 4 + kg]);
                }
#pragma unroll
                for (int tp = 0; tp < 2; ++tp) {
                    bf16x8 Zh[2], Zm[2], Zl[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        Zh[u] = __builtin_bit_cast(bf16x8, Zw[(0 * 4 + 2 * tp + u) * 64]);
                        Zm[u] = __builtin_bit_cast(bf16x8, Zw[(1 * 4 + 2 * tp + u) * 64]);
                        Zl[u] = __builtin_bit_cast(bf16x8, Zw[(2 * 4 + 2 * tp + u) * 64]);
                    }
#define CMW_SPEC(AP, ZP)                                                                                    \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                           \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) acc[j][2 * tp + u] = mfma16(AP[j], ZP[u], acc[j][2 * tp + u]);
                    CMW_SPEC(ah, Zl) CMW_SPEC(al, Zh) CMW_SPEC(am, Zm) CMW_SPEC(ah, Zm) CMW_SPEC(am, Zh) CMW_SPEC(ah, Zh)
#undef CMW_SPEC
                }
            }
            // ---- epilogue: cell 32 q + 16 j + 4 mg + r, channels 4 n + t: gz = g_x * act'(z), the BatchNorm-backward sums, one 16 B
            //      store per (j, r); the layer input act(z) of the same cells and channels stays in registers for the weight gradient
            f32x4v av[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const f32x4v spv = __builtin_bit_cast(f32x4v, sp[j][r]);
                    f32x4v o;
#pragma unroll
                    for (int t = 0; t < 4; t += 2) {             // channel pairs: packed fp32 math
                        const f32x2 v = f32x2{acc[j][t][r], acc[j][t + 1][r]};
                        const f32x2 sh = (f32x2{spv[t], spv[t + 1]} - f32x2{bp[t].mu, bp[t + 1].mu}) * f32x2{bp[t].is, bp[t + 1].is};
                        const f32x2 z = pk_fma(sh, f32x2{bp[t].ga, bp[t + 1].ga}, f32x2{bp[t].be, bp[t + 1].be});
                        f32x2 act = z, gz = v;
                        if (GELU) {
                            f32x2 gp;
                            gelu_both2(z, act, gp);
                            gz = v * gp;
                        }
                        av[j][r][t] = act[0];
                        av[j][r][t + 1] = act[1];
                        const f32x2 ov = WGZ ? gz : v;
                        o[t] = ov[0];
                        o[t + 1] = ov[1];
                        ssum[t >> 1] += gz;                      // cells past the line end: g_x = 0 exactly
                        ssq[t >> 1] = pk_fma(gz, sh, ssq[t >> 1]);
                    }
                    st16(o, ro, q * 8192 + ooff + j * 4096 + r * 256);
                }
            }
            // ---- weight gradient: dWc[co][ci] += sum over the tile's 32 cells of gs[cell][co] * act[cell][ci].  A = gs planes (row n of
            //      tile uo <-> co = 4 n + uo), B = act planes (column n of tile ui <-> ci = 4 n + ui), K = (mg, e = 4 j + r) <-> cell
            //      16 j + 4 mg + r on both sides.  Cells past the line end: their gs loads returned 0.
            {
                bf16x8 Xh[4], Xm[4], Xl[4], Gh[4], Gm[4], Gl[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float va[8], vg[8];
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            va[4 * j + r] = av[j][r][u];
                            vg[4 * j + r] = __builtin_bit_cast(f32x4v, gb[j][r])[u];
                        }
                    split8(va, Xh[u], Xm[u], Xl[u]);
                    split8(vg, Gh[u], Gm[u], Gl[u]);
                }
                issue_sg(gn, qn, more);              // sp / gb are consumed: the next tile's images go out now
#pragma unroll
                for (int uo = 0; uo < 4; ++uo) {
#define CMW_W(AP, BP) _Pragma("unroll") for (int ui = 0; ui < 4; ++ui) accW[uo][ui] = mfma16(AP[uo], BP[ui], accW[uo][ui]);
                    CMW_W(Gh, Xl) CMW_W(Gl, Xh) CMW_W(Gm, Xm) CMW_W(Gh, Xm) CMW_W(Gm, Xh) CMW_W(Gh, Xh)
#undef CMW_W
                }
            }
        }
    }
    // ---- partial rows of this wave: BatchNorm-backward sums [2][64], then the weight gradient [64 out][64 in]
    {
        float* part = a.stats_part + (long)slot * 128;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float s1 = ssum[t >> 1][t & 1], s2 = ssq[t >> 1][t & 1];
            s1 += __shfl_xor(s1, 16, 64);
            s2 += __shfl_xor(s2, 16, 64);
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (kg == 0) {
                part[4 * m + t] = s1;
                part[64 + 4 * m + t] = s2;
            }
        }
        float* wp = a.wg_part + (long)slot * (64 * 64);
#pragma unroll
        for (int uo = 0; uo < 4; ++uo)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 4 * (4 * kg + r) + uo;
                *reinterpret_cast<f32x4v*>(wp + o * 64 + 4 * m) = f32x4v{accW[uo][0][r], accW[uo][1][r], accW[uo][2][r], accW[uo][3][r]};
            }
    }
}

static size_t cmw_lds(int Wp) { return (size_t)(24 * 64 + 3 * (Wp + 1) * 4 + CMW_WAVES * 12 * 64) * 16; }

static bool cmw_one_wave() {
    static const bool v = getenv("RPB_CMW_VARIANT") && atoi(getenv("RPB_CMW_VARIANT")) == 1;
    return v;
}

long rpb_cmw_slots(long ncell, int Wp) {
    if (!cmw_one_wave()) return rpb_cmx_wg_slots(ncell, Wp);
    const long G = ncell / Wp;
    long grid = rpb_num_cus();
    const long need = (G + CMW_WAVES - 1) / CMW_WAVES;
    if (grid > need) grid = need;
    return grid * CMW_WAVES;
}

int rpb_cmw_launch(const CmxArgs& a, hipStream_t st) {
    RPB_REQUIRE(a.x && a.Wm && a.z2 && a.GW && a.out && a.stats_part && a.wg_part && a.bnb_s && a.bnb.mean, "cell_mix_wgrad: null pointer");
    RPB_REQUIRE(!a.bias && !a.xf.mean && !a.bf16_io && !a.feat_w && !a.y1out && a.crop_T == 0, "cell_mix_wgrad: plain fp32 backward launch only");
    if (!cmw_one_wave()) return rpb_cmx_wg_launch(a, st);
    const size_t lds = cmw_lds(a.Wp);
    RPB_REQUIRE(lds <= 160 * 1024, "cell_mix_wgrad: Wp=%d does not fit LDS", a.Wp);
    const int grid = (int)(rpb_cmw_slots(a.ncell, a.Wp) / CMW_WAVES);
    const bool gelu = a.bnb.gelu != 0, wgz = gelu && a.write_gz != 0;
#define RPB_CMW(G_, W_)                                                                                                     \
    if (gelu == G_ && wgz == W_) {                                                                                          \
        (void)hipFuncSetAttribute((const void*)cmw_kernel<G_, W_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
        hipLaunchKernelGGL((cmw_kernel<G_, W_>), dim3(grid), dim3(CMW_WAVES * 64), lds, st, a);                             \
    }
    RPB_CMW(true, true) RPB_CMW(true, false) RPB_CMW(false, false)
#undef RPB_CMW
    RPB_CHECK_LAUNCH("cell_mix_wgrad(bf16x3)");
}
