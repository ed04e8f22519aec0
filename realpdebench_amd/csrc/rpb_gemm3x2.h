// Interface between rpb_gemm3x.hip (C ABI, dispatch) and rpb_gemm3x2.hip (the two-workgroups-per-CU variant for N % 256 == 0).
#pragma once
#include "rpb_common.h"

struct G2Args {
    const float* A;        // [M][lda] fp32
    const uint16_t* Wz;    // operand-ordered planes of W[N][K] (rpb_gemm3x_wprep)
    const float* bias;     // [N] or null
    const float* addvec;   // [N] or null
    const float* residual; // [M][ldo] or null
    float* out;            // [M][ldo]
    long M;
    int N, K, lda, ldo;
    int act;               // as rpb_gemm_nt: 0 none, 1 GELU (pre_out optional), 2 * gelu'(aux), 3 ReLU, 4 zero where aux <= 0
    const float* aux;
    float* pre_out;
    DropSpec drop;         // thr != 0: in-kernel inverted dropout, same Philox counters as rpb_gemm_nt / rpb_dropout_mul (element index >> 2)
};

bool rpb_gemm3x2_supported(long M, int N, int K, bool has_mask, bool has_drop);
int rpb_gemm3x2_launch(const G2Args& a, hipStream_t st);
