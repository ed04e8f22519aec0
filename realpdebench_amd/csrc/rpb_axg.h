// Interface between rpb_axis_gemm.hip (dispatch) and rpb_axg.hip (DFT stage on the bf16 matrix pipe).
#pragma once
#include "rpb_common.h"

struct AxgArgs {
    const float* in;
    float* out;
    const float* Mt;     // [K][O] row-major (the stage matrix transposed)
    int G, K, O, N;
    long in_g, in_k, out_g, out_o;
    int k_valid;
    XForm xf;            // lazy BatchNorm(+GELU) on the input, channel = n (N <= 128)
    int in_bf16;         // `in` holds bf16 (strides in bf16 elements); plain stage, O <= 64
    int out_bf16;        // `out` is stored as bf16, round to nearest even (strides in bf16 elements); the short-K "resident" kernel only
};

bool rpb_axg_supported(int G, int K, int O, int N, long in_g, long in_k, long out_g, long out_o, int k_valid, int accumulate,
                       bool has_xf);
int rpb_axg_launch(const AxgArgs& a, hipStream_t st);
