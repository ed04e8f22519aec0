// Arguments of the one-pass projection-head backward, shared by its two kernels: rpb_pjf.hip (16x16x32 tiles, every fc2 width <= 4) and
// rpb_pjg.hip (32x32x16 tiles, scalar vector work, fc2 width 2: the default where it applies).
#pragma once
#include "rpb_common.h"

struct PjfArgs {
    const float* s;       // padded pre-BN tensor of the last Fourier layer [B*Tp*Hp*Wp][64]
    const float* w1;      // fc1.weight [128][64]
    const float* b1;      // [128]
    const float* w2;      // fc2.weight [DO][128]
    const float* gout;    // [ncrop][DO]  (LOSS: the TARGET y instead -- the kernel forms out = fc2 gelu(u) + b2 and gout = gscale (out - y) itself)
    const float* b2;      // LOSS: fc2.bias [DO]
    float gscale;         // LOSS: 2 / (number of output elements over all ranks)
    float* loss_part;     // LOSS: [slots] partial sums of (out - y)^2
    float* g;             // [ncell][64] gradient w.r.t. the layer output, padded layout
    float* part;          // [slots][128*64 + DO*128 + 128 + DO]   (M = gh^T shat | d fc2 | d b1 | d b2)
    int B, DO;
    CropMap cm;
    XForm xf;             // BatchNorm of the last layer: mean, invstd, gamma, beta (gelu must be 0)
};

// rpb_pjg.hip: 1 when the 32x32x16 kernel takes this shape (DO == 2 and not switched off with RPB_HEAD_PJG=0)
int pjg_supported(int DO);
size_t pjg_lds();
int pjg_launch(PjfArgs& p, bool loss, int grid, hipStream_t st);
