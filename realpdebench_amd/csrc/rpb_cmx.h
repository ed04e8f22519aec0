// Interface between rpb_cell.hip (dispatch) and rpb_cmx.hip (cell_mix on the bf16 matrix pipe, C = 64).
#pragma once
#include "rpb_common.h"

struct CmxArgs {
    const float* x;       // [ncell][64]
    const float* Wm;      // transpose_w == 0: [CO][KC] (out = x W^T) ; == 1: [KC][CO] (out = x W)
    const float* bias;    // [64] or null
    const float* z2;      // [G][K2][64]
    const float* GW;      // [K2][Wp]  (transposed stage matrix)
    float* out;           // [ncell][64]
    float* stats_part;    // [gridDim.x*8][2][64] or null
    long ncell;
    int K2, Wp;
    int transpose_w;
    XForm xf;             // lazy BatchNorm(+GELU) applied to x on load
    const float* bnb_s;   // STATS == 2: pre-BN tensor of the layer whose output gradient this launch produces
    XForm bnb;            //             and that layer's BatchNorm; STATS == 0 with bnb.mean: OUTPUT transform (eval)
    int bf16_io;          // x and out are bf16 [ncell][64] (passed through the float pointers); eval path only
    int feat_w;           // > 0: x is the feature tensor Phi_c [ncell][feat_w] and Wm the composite weight [64][feat_w] (layer 0, forward)
    int write_gz;         // STATS == 2: store gz = out * act'(z) instead of out (the consumer then skips act')
    const float* FWt;     // eval (STATS == 0 with the output transform): forward W-stage matrix [Wp][K2f] of the NEXT layer's spectral branch ...
    float* y1out;         // ... and its result [G][K2f][64] = that stage applied to the activated line this launch writes (fused: the
    int K2f;              //     activations are not read again for it); null = off
    int crop_T, crop_H, crop_W, Tp, Hp;   // crop_T > 0 (eval, no statistics, no fused stage): only lines t < crop_T, h < crop_H of each [Tp][Hp] sample
                                          // are produced, and of each line the tiles up to cell crop_W - 1 (the projection head reads nothing else)
    void* gw_planes;      //     scratch of 3 * Wp * 64 bytes: GW as bf16 planes in operand order (written by the launch)
    int spec_bf16;        // with bf16_io: z2 (input rows) and y1out (fused W stage) hold bf16 as well
    int c128;             // x / out / z2 / bnb_s rows hold 128 channels, Wm is [128][128] (rpb_cmx128_supported); 0: the C = 64 instance
    int claim_mode;       // how the (b,t,h) lines reach the waves: 0 dealt round-robin | 1 claimed from a workgroup counter (LDS) | 2 claimed chip-wide
    int* claim_ctr;       //   mode 2: zero on entry, left zero (the last claim resets it)
    unsigned long long* wave_times;   // diagnostics (rpb_cmx_debug_wave_times): [block][wave][2] constant-clock ticks at wave start / end, or null
    int h2;               // eval launches on fp32 storage: the opt-in f16x2 arithmetic (two fp16 planes per operand, three products; rpb_cmx.hip "H2")
    int spec_exp;         //   with h2: GW is multiplied by 2^spec_exp and the z2 rows by 2^-spec_exp before they are split (GW carries 1 / (Tp Hp Wp):
                          //   subnormal in fp16; the caller passes floor(log2(Tp Hp Wp)) - 1 so that max |GW| 2^spec_exp lies in [0.5, 2))
    float* wg_part;       // WG launch only: [slots][64 x 64] partial rows of the 1x1-conv weight gradient  x^T act(BN(bnb_s))
};

bool rpb_cmx_dft_supported(int Wp, int K2f);
bool rpb_cmx_supported(long ncell, int KC, int CO, int K2, int Wp, bool spec, bool gather);
long rpb_cmx_stat_rows(long ncell, int Wp, int stats);   // stats: 0 / 1 / 2 as in the kernel template
bool rpb_cmx128_supported(long ncell, int KC, int CO, int K2, int Wp, bool spec, bool gather);   // the C = 128 instance (set CmxArgs::c128)
long rpb_cmx128_stat_rows(long ncell, int Wp);           // partial rows of [2][128]
int rpb_cmx_launch(const CmxArgs& a, int stats, hipStream_t st);

// The STATS == 2 launch with the 1x1-conv weight gradient of the same layer riding along (x = gs of the layer, bnb_s = the pre-BN tensor
// whose activation is the layer input): dWc[co][ci] = sum_cells x[cell][co] * act(BN(bnb_s))[cell][ci] -- wave pairs inside the
// two-waves-per-SIMD kernel (rpb_cmx.hip, template parameter WG).  (The one-wave-per-SIMD organisation of round 4, measured slower in
// every A/B of rounds 4 and 5, is archived under tools/archive/rpb_cmw.hip and no longer compiled.)
long rpb_cmx_wg_slots(long ncell, int Wp);         // partial rows of stats_part ([2][64]) and wg_part ([64][64])
int rpb_cmx_wg_launch(const CmxArgs& a, hipStream_t st);

// csrc/rpb_cwx.hip: rpb_cell_wgrad's (CO, CI) = (128, 128) instance without the crop on the bf16 matrix pipe (width-128 Fourier layers)
bool rpb_cwx128_supported(long ncell, int CO, int CI, int crop, int W);
int rpb_cwx128_launch(const float* gs, const float* x, float* part, long ncell, long slots, const XForm& xf, int crop, const CropMap& cm,
                      hipStream_t st);
