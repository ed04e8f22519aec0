// Interface between rpb_bwd_row.hip (entry points, fp32-MFMA kernel) and rpb_bwr.hip (the C = 64 row kernel on the bf16 matrix pipe).
#pragma once
#include "rpb_common.h"

struct BwrArgs {
    const float* s;        // [G*Wp][64] pre-BatchNorm output of this layer
    const float* gy;       // [G*Wp][64] gradient w.r.t. the layer output (gz when the producer already applied gelu')
    const float* x;        // [G*Wp][64] layer input (plain, or pre-BN of the previous layer with xf); FEAT: [G*Wp][FW] feature fields
    float* gs;             // [G*Wp][64] out (may alias gy)
    const float* mean;     // BatchNorm of THIS layer
    const float* invstd;
    const float* gamma;
    const float* beta;
    const float* sums;     // [2*64] global (sum gz | sum gz*shat)
    float inv_count;
    int gelu;
    XForm xf;              // lazy activation of the layer input x
    const float* GW;       // adjoint W-stage matrix as [Wp][K2]
    float* Y1;             // [G][K2][64]
    float* part;           // [slots][64*64 + 64]
    int G, Wp, K2, FW;
    int CS, coff;          // floats per cell row of s / gy / gs / Y1 (64; 128: one 64-channel half of a width-128 layer per launch, NOX only) and the half's first channel;
                           // the per-channel vectors (mean .. sums) are passed already offset, the second moment sits CS floats behind the first
};

bool rpb_bwr_supported(int C, int Wp, int K2, int FW);
bool rpb_bwr_supported_c128(int Wp, int K2);
long rpb_bwr_slots(int G);
int rpb_bwr_launch(const BwrArgs& a, long part_rows, hipStream_t st);
