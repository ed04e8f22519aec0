// Projection head (fc1 -> act -> fc2, fno.py:121-125) on the bf16 matrix pipe: internal entry points of csrc/rpb_pjx.hip that
// rpb_proj_fwd / rpb_proj_bwd (csrc/rpb_proj.hip) dispatch to when the shape is covered (C = 64, fc2 out features <= 4, fp32 input).
#pragma once
#include "rpb_common.h"

bool rpb_pjx_head_supported(int C, int DO, bool bwd);
long rpb_pjx_head_slots(int B, int T, int H, bool bwd);
int rpb_pjx_head_launch(bool bwd, const float* s, const float* w1, const float* b1, const float* w2, const float* b2, const float* gout,
                        float* out, float* gu, float* part, long part_rows, int DO, int T, int H, int W, int Tp, int Hp, int Wp, long ncrop,
                        const XForm& xf, int act, hipStream_t st, bool a_bf16 = false);

// rpb_pjh.hip: the evaluation forward on 32x32x16 tiles at two waves per SIMD (C = 64, <= 4 fc2 outputs, GELU, fp32 storage; RPB_HEAD_PJH=0
// keeps the kernel above)
bool rpb_pjh_supported(int C, int DO, int act, const XForm& xf, bool a_bf16);
int rpb_pjh_launch(const float* s, const float* w1, const float* b1, const float* w2, const float* b2, float* out, int B, int DO, int T, int H,
                   int W, int Tp, int Hp, int Wp, const XForm& xf, hipStream_t st, bool f16x2 = false,    // f16x2: the opt-in two-fp16-plane arithmetic
                   bool a_bf16 = false,                                   // a_bf16: `s` holds bf16 [cells][64] (bf16 activation storage)
                   int C = 64,                                            // C = 128: the width-128 instance (fp32 storage, default arithmetic)
                   bool silu = false);                                    //   ... with SiLU instead of GELU (the Galerkin regressor's head)
// ... and rpb_proj_bwd's job at C = 128 (gu + the fc2 / bias partial rows) on the same organisation
int rpb_pjh_bwd128_launch(const float* s, const float* w1, const float* b1, const float* w2, const float* b2, const float* gout, float* gu,
                          float* part, long part_rows, int B, int DO, int T, int H, int W, int Tp, int Hp, int Wp, const XForm& xf, hipStream_t st,
                          bool silu = false);
// ... and the fc1 data gradient at C = 128, gathered into the padded layout (rpb_cell_mix with gather = 1, no statistics)
// (stats_part != NULL: + the BatchNorm-backward sums (sum g, sum g * shat) of the layer whose pre-BN tensor is bnb_s, one [2][128] row per wave)
int rpb_pjh_dgrad128_launch(const float* gu, const float* w1, float* g, int B, int T, int H, int W, int Tp, int Hp, int Wp, hipStream_t st,
                            const float* bnb_s = nullptr, const float* mean = nullptr, const float* invstd = nullptr, float* stats_part = nullptr,
                            long stats_rows = 0);
