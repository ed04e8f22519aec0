/* rpb.h -- C ABI of librpb_hip.so: the MI355X (gfx950) kernels behind RealPDEBench's FNO3d hot path.
 *
 * Boundary contract:
 *   - plain C, no torch / pybind types; device pointers are raw `float*` into caller-owned HBM;
 *   - every function is stream-ordered on the `hipStream_t` passed as `void* stream`, never synchronises,
 *     never allocates; scratch ("partial rows") is caller-allocated, sized by the `*_rows/_slots` queries;
 *   - returns 0 on success, <0 on error (RPB_ERR_*); `rpb_last_error()` returns a thread-local message;
 *   - fp32 STORAGE everywhere unless an entry point says bf16 (the reference is fp32).  Arithmetic of the contractions: at the shapes
 *     the reference's YAMLs use (C = 64, token GEMMs with K >= 256, 3x3x3 convolutions) v_mfma_f32_{16x16x32,32x32x16}_bf16 on operands split
 *     into three bf16 planes, six products per fp32 product with fp32 accumulation (dropped terms <= 2^-24 |a b|: fp32-grade, Rel-L2 ~2e-7
 *     against fp64); other shapes and the RPB_*_F32 / RPB_*_EXACT switches use v_mfma_f32_32x32x2_f32 = exact fp32 FMA chains.
 *
 * Tensor layouts (MI355X-first, not the reference's):
 *   activations   [B][Tp][Hp][Wp][C]       channels-last, one cell = C contiguous floats (256 B at C=64)
 *   spectra       [B][2 (re,im)][M][C]     planar complex, M = (2*m1)*(2*m2)*m3 retained modes
 *   spectral W    [M][Ci][Co][2]           mode-major interleaved complex (reference: 4 x [Ci][Co][m1][m2][m3] c64)
 *
 * Each entry point names the reference code it replaces (paths relative to the reference repo root).
 */
#ifndef RPB_H
#define RPB_H
#ifdef __cplusplus
extern "C" {
#endif

#define RPB_OK 0
#define RPB_ERR_ARG (-1)
#define RPB_ERR_LAUNCH (-2)
#define RPB_ERR_UNSUPPORTED (-3)

/* ABI version of THIS header.  rpb_abi_version() returns the version the library was built with: a caller compiled against another
 * header must refuse to run (signatures changed under unchanged symbol names between versions -- see INTEGRATION.md "ABI versions").
 *   1  rounds 1-3
 *   2  round 4: rpb_cell_mix_bf16, rpb_cell_mix_eval_dft_bf16 and rpb_cell_mix_eval_crop gained `int spectra_bf16` before `stream`;
 *      round 5: rpb_dp_reduce_scatter_enqueue / rpb_dp_allgather_enqueue / rpb_dp_mark / rpb_dp_wait_mark / rpb_dp_set_model /
 *      rpb_adam_step_ranges added (additions only);
 *      round 6: rpb_dp_p2p_*, rpb_cell_mix_eval_dft_f16x2, rpb_cell_mix_eval_crop_f16x2, rpb_proj_fwd_f16x2 added (additions only) */
#define RPB_ABI_VERSION 2
const char* rpb_last_error(void);
int rpb_abi_version(void);
/* bf16 activation STORAGE (BASELINE.json configs[4]; the opt-in rollout path): how many bf16 planes of the fp32 constants (conv / fc1 weights,
 * DFT stage matrices) are multiplied with a bf16-stored operand -- 2 (default build: the third plane is 1 / 128 of the error the stored operand
 * already carries) or 3 (-DRPB_BF16_CONST_PLANES=3).  The fp32-storage path always uses three planes of both operands. */
int rpb_bf16_const_planes(void);

/* K1  lift + zero-pad.  realpdebench/model/fno.py:106-111 (get_grid :135-143, cat, fc0, permute, F.pad).
 *     out[b,t,h,w,:] = fc0_w @ [x[b,t,h,w,:], gt[t], gh[h], gw[w]] + fc0_b inside T x H x W, 0 in the pad. */
int rpb_lift_pad_fwd(const float* x, const float* gt, const float* gh, const float* gw, const float* fc0_w,
                     const float* fc0_b, float* out, int B, int T, int H, int W, int Cin, int C, int Tp, int Hp, int Wp,
                     void* stream);
/*     autograd of the above w.r.t. fc0.{weight,bias}: part[rows][C*(Cin+3) + C], rows = rpb_lift_bwd_rows(). */
int rpb_lift_bwd_rows(void);
int rpb_lift_bwd(const float* g_out, const float* x, const float* gt, const float* gh, const float* gw, float* part,
                 int B, int T, int H, int W, int Cin, int C, int Tp, int Hp, int Wp, void* stream);

/* K2/K4  one truncated-DFT stage: out[g][o][n] (+)= sum_k M[o][k] * in[g][k][n], n contiguous.
 *     `Mt` is the stage matrix TRANSPOSED ([K][O] row-major) so that staging it into LDS is a coalesced copy.
 *     Replaces torch.fft.rfftn / torch.fft.irfftn of SpectralConv3d.forward (fno.py:48, :63) without ever
 *     materialising the discarded 97 % of the spectrum (fno.py:51-60).  Inputs k >= k_valid are treated as 0. */
/*     Lazy activation (all `xf_*` argument groups below): when xf_mean != NULL the input tensor is the PRE-BatchNorm
 *     output s of the producing layer and the kernel applies act(xf_gamma*(s-xf_mean)*xf_invstd+xf_beta), act = exact
 *     GELU if xf_gelu else identity, per channel while loading -- fno.py:117-119 fused into the consumer, so the
 *     normalised/activated tensor is never written to HBM.  For rpb_axis_gemm this requires N == channels. */
int rpb_axis_gemm(const float* in, float* out, const float* Mt, int G, int K, int O, int N, long in_g_stride,
                  long in_k_stride, long out_g_stride, long out_o_stride, int k_valid, int accumulate,
                  const float* xf_mean, const float* xf_invstd, const float* xf_gamma, const float* xf_beta, int xf_gelu,
                  void* stream);

/* K3  per-mode complex channel contraction.  compl_mul3d / the four corner-block einsums, fno.py:41-43, 53-60,
 *     and their autograd (dgrad: gX = gY conj(W); wgrad: gW = conj(X) gY). */
int rpb_mode_contract_fwd(const float* X, const float* W, float* Y, int B, int M, int C, void* stream);
int rpb_mode_contract_dgrad(const float* GY, const float* W, float* GX, int B, int M, int C, void* stream);
int rpb_mode_contract_wgrad(const float* X, const float* GY, float* GW, int B, int M, int C, int accumulate,
                            void* stream);

/* K4+K5  last inverse-DFT stage fused with the 1x1x1 Conv3d, the `x1 + x2` add and the BatchNorm statistics.
 *     fno.py:63 (W-axis c2r part of irfftn), :115 (Conv3d), :116 (add), batch statistics of :117.
 *     out[cell][o] = sum_k GW[w(cell)][k] z2[g(cell)][k][o] + sum_i x[row(cell)][i] Wm(o,i) + bias[o]
 *     (`GWt` is GW transposed: [K2][Wp] row-major)
 *     transpose_w=0: Wm is [CO][KC] (forward); =1: Wm is [KC][CO] (dgrad).  z2 == NULL drops the spectral term.
 *     gather=1: cells are padded cells and row(cell) is the cropped index (zero rows in the pad margin).
 *     stats_part (optional): [rpb_cell_mix_stat_rows(...)][2][CO] partial sums of out and out^2; when bnb_s != NULL
 *     the output is the gradient w.r.t. act(BN(bnb_s)) and the partials are instead (sum gz, sum gz*shat), the
 *     first pass of that layer's BatchNorm backward (bnb_* = its mean, invstd, gamma, beta, gelu flag). */
long rpb_cell_mix_stat_rows(long ncell, int KC, int CO, int K2, int Wp, int has_spec, int bn_bwd_stats);
/*     1 when the kernel this shape dispatches to honours bnb_gelu == 2: "GELU, and store gz = out * gelu'(z) instead of out", so
 *     that the BatchNorm-backward apply that consumes the tensor (rpb_bn_bwd_row with gelu = 0) does not evaluate gelu' again. */
int rpb_cell_mix_writes_gz(long ncell, int KC, int CO, int K2, int Wp, int has_spec, int gather);
/*     bnb_s == NULL with the four bnb vectors given (and no stats_part): OUTPUT transform, the tile is stored as
 *     act(BN(out)) -- eval mode, where the running statistics are known before the launch (fno.py:117-119). */
int rpb_cell_mix(const float* x, const float* Wm, const float* bias, const float* z2, const float* GWt, float* out,
                 float* stats_part, long ncell, int KC, int CO, int K2, int Wp, int transpose_w, int gather, int T,
                 int H, int W, int Tp, int Hp, int Wp_pad, const float* xf_mean, const float* xf_invstd,
                 const float* xf_gamma, const float* xf_beta, int xf_gelu, const float* bnb_s, const float* bnb_mean,
                 const float* bnb_invstd, const float* bnb_gamma, const float* bnb_beta, int bnb_gelu, void* stream);

/*     The backward launch of a Fourier layer at C = 64 with the layer's Conv3d weight gradient riding along (csrc/rpb_cmx.hip, wave pairs;
 *     autograd of fno.py:63,115-119): out = gs Wc + FW^T z2 (Wc = convs.l.weight [co][ci], FWt = the adjoint stage matrix [K2][Wp]),
 *     stored as gz = out * act'(BN(s_prev)) when gelu == 2 (gelu == 1: act' only enters the sums, 0: identity activation);
 *     stats_part[slot][2][64] = (sum gz, sum gz * shat) of the layer below;  wg_part[slot][64][64] = partial
 *     d convs.l.weight[co][ci] = sum_cells gs[cell][co] * act(BN(s_prev))[cell][ci];  slot < rpb_cell_mix_wgrad_slots(ncell, Wp).
 *     With it rpb_bn_bwd_row runs with x == NULL (no weight gradient there: the layer input is read by one kernel less). */
long rpb_cell_mix_wgrad_slots(long ncell, int Wp);
int rpb_cell_mix_wgrad_supported(long ncell, int K2, int Wp);
int rpb_cell_mix_wgrad(const float* gs, const float* Wc, const float* z2, const float* FWt, float* out, float* stats_part,
                       float* wg_part, long ncell, int K2, int Wp, const float* s_prev, const float* mean, const float* invstd,
                       const float* gamma, const float* beta, int gelu, void* stream);
/*     diagnostics: buf != NULL makes every C = 64 cell_mix launch record, per line-walking wave, the constant-clock tick (100 MHz) at
 *     its start and end in buf[(block * waves + wave) * 2 + {0, 1}] (8 B each; >= 256 * 8 * 2 entries); NULL switches it off.
 *     tools/wave_times.py turns the records into the residency profile of a launch (how long the last wave runs past the mean). */
int rpb_cmx_debug_wave_times(void* buf);
/*     How the (b,t,h) lines of a C = 64 / 128 cell_mix launch reach its waves: 0 dealt round-robin (every partial sum in a fixed order:
 *     bit-reproducible training runs), 1 claimed by the waves of a workgroup from a counter in LDS (default), 2 claimed chip-wide from a
 *     counter in HBM; -1 returns to the RPB_LINE_CLAIM environment variable / the default.  Outputs are identical in every mode; the
 *     per-wave partial rows (BatchNorm sums, weight-gradient rows) differ in summation order only. */
int rpb_line_claim_set(int mode);

/*     weight / bias gradient of a per-cell linear layer (Conv3d 1x1x1 fno.py:115, fc1 fno.py:123):
 *     part[rpb_cell_wgrad_slots(...)][CO*CI + CO];  crop=1: x row = padded index of cropped cell. */
long rpb_cell_wgrad_slots(long ncell, int CO, int CI);
int rpb_cell_wgrad(const float* gs, const float* x, float* part, long ncell, int CO, int CI, int crop, int T, int H,
                   int W, int Tp, int Hp, int Wp, const float* xf_mean, const float* xf_invstd, const float* xf_gamma,
                   const float* xf_beta, int xf_gelu, void* stream);

/* K6  BatchNorm3d (+ exact-erf GELU).  fno.py:117-119; training statistics include the padded cells. */
/*     out[j] (+)= scale * sum_r part[r*row_stride + j], j < L, accumulated in fp64 (deterministic, no atomics). */
int rpb_reduce_partials(const float* part, long rows, long L, long row_stride, float* out_f32, double* out_f64,
                        double scale, int accumulate, void* stream);
/*     nbatch independent reductions in one launch: out[b][j] = sum_r part[b*batch_stride + r*row_stride + j] (fp64 accumulate). */
int rpb_reduce_partials_batched(const float* part, int nbatch, long rows, long L, long row_stride, long batch_stride,
                                float* outf, void* stream);
/*     n reductions of different shapes in one launch: items [n][6] int64 ON THE DEVICE = { part pointer, out pointer (fp32), rows, L,
 *     row_stride in floats, first grid block of the item }; a block covers rpb_reduce_partials_grouped_cols() columns, total_chunks =
 *     sum over items of ceil(L / that).  out[j] = sum_r part[r * row_stride + j] in fp64, fixed order (as rpb_reduce_partials). */
int rpb_reduce_partials_grouped_cols(void);
int rpb_reduce_partials_grouped(const void* items, int n, long total_chunks, void* stream);
int rpb_bn_finalize(const double* sums, double count, float eps, float momentum, float* mean, float* invstd,
                    float* running_mean, float* running_var, int C, void* stream);
int rpb_bn_eval_prep(const float* running_var, float eps, float* invstd, int C, void* stream);
int rpb_bn_act_fwd(const float* s, const float* mean, const float* invstd, const float* gamma, const float* beta,
                   float* y, long ncell, int C, int gelu, void* stream);
int rpb_bn_bwd_rows(void);
int rpb_bn_bwd_reduce(const float* s, const float* gy, const float* mean, const float* invstd, const float* gamma,
                      const float* beta, float* part, long ncell, int C, int gelu, void* stream);
int rpb_bn_bwd_apply(const float* s, const float* gy, const float* mean, const float* invstd, const float* gamma,
                     const float* beta, const float* sums, double count, float* gs, long ncell, int C, int gelu,
                     void* stream);

/*     Fused backward row pass of one Fourier layer (C = 32 or 64): BatchNorm(+GELU) backward apply -> gs (may alias
 *     gy), adjoint W stage Y1[g][K2][C] = GWt gs, and the Conv3d weight/bias gradient partials -- one pass over
 *     s, gy, x instead of three kernels (autograd of fno.py:115-119 + first stage of the autograd of fno.py:63).
 *     part[rpb_bn_bwd_row_slots(G)][C*C + C]; G = B*Tp*Hp rows of Wp cells.  `M_wk` is the adjoint W-stage
 *     matrix transposed, i.e. [Wp][K2] row-major.  x == NULL (C = 64, no xf_*): no weight gradient in this launch -- the
 *     C*C block of the partial rows stays unwritten (rpb_cell_mix_wgrad forms it), only the [C] bias sums follow it. */
long rpb_bn_bwd_row_slots(int G);
int rpb_bn_bwd_row(const float* s, const float* gy, const float* x, float* gs, const float* mean, const float* invstd,
                   const float* gamma, const float* beta, const float* sums, double count, int gelu,
                   const float* xf_mean, const float* xf_invstd, const float* xf_gamma, const float* xf_beta, int xf_gelu,
                   const float* M_wk, float* Y1, float* part, int G, int Wp, int C, int K2, void* stream);

/* K7  crop + fc1 + activation + fc2.  fno.py:121-125 (act = 0: exact GELU); the same head of the Galerkin
 *     SpectralRegressor, galerkin_transformer_libs/model.py:626-633 (act = 1: SiLU).  proj_bwd recomputes fc1 and emits gu = dL/d(fc1 pre-activation)
 *     [ncrop][128] plus partial rows [rpb_proj_slots(...)][DO*128 + 128 + DO] for d fc2.weight, d fc1.bias, d fc2.bias. */
long rpb_proj_slots(long ncrop, int C, int DO);
int rpb_proj_fwd(const float* a, const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b,
                 float* out, long ncrop, int C, int DO, int T, int H, int W, int Tp, int Hp, int Wp, const float* xf_mean,
                 const float* xf_invstd, const float* xf_gamma, const float* xf_beta, int xf_gelu, int act, void* stream);
int rpb_proj_bwd(const float* a, const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b,
                 const float* gout, float* gu, float* part, long ncrop, int C, int DO, int T, int H, int W, int Tp,
                 int Hp, int Wp, const float* xf_mean, const float* xf_invstd, const float* xf_gamma, const float* xf_beta,
                 int xf_gelu, int act, void* stream);

/*     MSE.  realpdebench/utils/metrics.py:11-13 + `.mean()` of train.py:328 and its gradient. */
int rpb_mse_rows(void);
int rpb_mse(const float* pred, const float* target, float* elem, float* gout, float* part, long n, float gscale,
            void* stream);

/* K8  Adam (torch.optim.Adam defaults, train.py:290,333; complex weights as 2 x fp32). */
int rpb_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                  long step, float gscale, void* stream);
/*     the same update on a list of ranges of the arena: tab [nr][2] (device, int64) = (first element, float4 groups before the range),
 *     starts and counts multiples of 4, total = elements over all ranges (the sharded optimizer step of the data-parallel path) */
int rpb_adam_step_ranges(float* p, const float* g, float* m, float* v, const long* tab, int nr, long total, float lr, float beta1,
                         float beta2, float eps, long step, float gscale, void* stream);

/* K9  rollout step glue.  realpdebench/eval.py:316-318 + data/data_normalizer.py:50-62. */
int rpb_rollout_affine(const float* pred, const float* para, float* out, long ncell, int Cp, int Cx,
                       const float* mean_t, const float* std_t, const float* mean_i, const float* std_i, void* stream);
int rpb_channel_affine(const float* in, float* out, long n, int C, const float* mean, const float* stdv, int inverse,
                       void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Transolver (Physics-Attention, structured 3-D mesh) -- forward path.  Tokens are channels-last rows [token][C].
 * Reference: realpdebench/model/TRANSOLVER_libs/{Physics_Attention.py:148-176, Transolver_Structured_Mesh_3D.py:170-196}
 * ------------------------------------------------------------------------------------------------------------------ */

/*     out = epilogue(A W^T): nn.Linear over tokens with fused bias / GELU / broadcast vector / residual
 *     (Transolver_Structured_Mesh_3D.py:31-39,71-77) and, with conv=1, the two nn.Conv3d(C, C, 3, padding=1) of
 *     Physics_Attention.py:154-157 as one implicit GEMM over the (Hc, Wc, Dc) mesh: W is [N][27*Ci] with
 *     column = ((kh*3 + kw)*3 + kd)*Ci + ci.  K must be a multiple of 32. */
/*     act: 0 none | 1 exact GELU (pre_out, if given, receives the pre-activation for the backward pass) |
 *          2 multiply by gelu'(aux[m][n]) (backward through a GELU; aux = that saved pre-activation) |
 *          3 ReLU (Galerkin FeedForward, galerkin_transformer_libs/layers.py:979-981) |
 *          4 zero where aux[m][n] <= 0 (backward through that ReLU; aux = the saved ReLU output). */
int rpb_gemm_nt(const float* A, const float* W, const float* bias, const float* addvec, const float* residual, float* out,
                long M, int N, int K, int lda, int ldo, int act, const float* aux, float* pre_out, const float* mask,
                int conv, int Hc, int Wc, int Dc, int cls, long drop_seed, float drop_keep, void* stream);
/*     mask: optional [M][ldo] inverted-dropout multiplier; or drop_keep in (0,1): the same nn.Dropout generated in the
 *     epilogue by a counter-based RNG (Philox4x32-10) keyed on (drop_seed, element index m*ldo + n), so that the backward
 *     pass regenerates it (rpb_dropout_mul, or the same seed on the data-gradient GEMM) and no mask tensor exists. */
/*     conv = 2 / 3: the U-Net's Downsample nn.Conv3d(C, C, (1,4,4), (1,2,2), (0,1,1)) (realpdebench/model/unet.py:166-167; rows =
 *     output tokens, K = 16*Ci, k = (kh*4 + kw)*Ci + ci) and one output-parity class cls = 2*ph + pw of its Upsample
 *     nn.ConvTranspose3d (unet.py:163-164; rows = input tokens, written to row (t, 2h+ph, 2w+pw), K = 4*Ci); the data gradient
 *     of either is the other with the weights re-laid-out.  rpb_gemm_tn conv = 2 is the matching weight-gradient gather. */
/*     weight gradients: part[rpb_gemm_tn_splits(M,N,K)][N*K + N] partials of dW[n][k] = sum_m G[m][n] A(m,k) and
 *     db[n] = sum_m G[m][n] (autograd of the nn.Linear / nn.Conv3d weights above); conv=1 gathers A like rpb_gemm_nt. */
int rpb_gemm_tn_splits(long M, int N, int K, int conv);
int rpb_gemm_tn(const float* G, const float* A, float* part, long M, int N, int K, int ldg, int lda, int conv, int Hc,
                int Wc, int Dc, void* stream);
/*     the same weight gradients (conv = 0) on the bf16 matrix pipe from three-plane splits of both operands (fp32-grade, six products,
 *     csrc/rpb_gemm3x_tn.hip): N, K multiples of 256, M >= 4096; part[rpb_gemm3x_tn_splits(M,N,K)][N*K + N] as above.  The token
 *     tensors are read row-major as they are (the transposition to MFMA operand order happens in registers / LDS). */
int rpb_gemm3x_tn_supported(long M, int N, int K, int ldg, int lda);
int rpb_gemm3x_tn_splits(long M, int N, int K);
int rpb_gemm3x_tn(const float* G, const float* A, float* part, long M, int N, int K, int ldg, int lda, void* stream);
/*     tiny-K linear (+GELU): preprocess.linear_pre, C_in -> 2*n_hidden (Transolver_Structured_Mesh_3D.py:27,32). */
int rpb_tokens_lift(const float* x, const float* W, const float* b, float* out, long M, int K, int N, int act,
                    void* stream);
/*     nn.LayerNorm(C) per token, one wavefront per token (Transolver_Structured_Mesh_3D.py:56,60,68). */
int rpb_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* out, long M, int C, float eps,
                      void* stream);
/*     its backward: gx = LN'(gy) (+ gadd, the residual branch's gradient); part[rpb_layernorm_bwd_rows(M)][2][C]
 *     partials of (d gamma, d beta). */
long rpb_layernorm_bwd_rows(long M);
int rpb_layernorm_bwd(const float* x, const float* gamma, const float* gy, const float* gadd, float* gx, float* part,
                      long M, int C, float eps, void* stream);
/*     slice weights (temperature softmax over G slices per head) + per-sample slice-token sums and norms
 *     (Physics_Attention.py:158-162).  xf rows: fx_mid at column 0, x_mid at column heads*32.
 *     tok_part [B*bps][heads*G*32], norm_part [B*bps][heads*G], bps = rpb_slice_blocks_per_sample(B). */
int rpb_slice_blocks_per_sample(int B);
/*     w_in != NULL: skip the softmax, use the given weights and only produce the token sums of xf[:, :heads*32]
 *     (that is the backward of deslice w.r.t. the slice tokens). */
int rpb_slice_fwd(const float* xf, const float* Ws, const float* bs, const float* temp, float* w_out, float* tok_part,
                  float* norm_part, int B, int ntok, int heads, int G, int ldx, const float* w_in, void* stream);
/*     backward of slice + deslice w.r.t. the dual-convolution output: gxf[m] = [g_fx_mid | g_x_mid];
 *     part[B*bps][G*32 + G + heads] partials of (d in_project_slice.weight, d in_project_slice.bias, d tau). */
int rpb_slice_bwd(const float* xf, const float* w, const float* gox, const float* tok2, const float* gT, const float* gN,
                  const float* Ws, const float* temp, float* gxf, float* part, int B, int ntok, int heads, int G,
                  void* stream);
/*     out = a * b elementwise (dropout masks in the backward pass). */
int rpb_mul(const float* a, const float* b, float* out, long n, void* stream);
/*     out = a + b (gradient sum where the U-Net's tape forks) and strided row-block copies
 *     dst[m][doff..doff+C) = src[m][soff..soff+C) (the skip-connection concat torch.cat(dim=1) of unet.py:463,479 and its split). */
int rpb_add(const float* a, const float* b, float* out, long n, void* stream);
int rpb_copy_cols(const float* src, float* dst, long M, int C, int lds, int ldd, int soff, int doff, void* stream);
/*     column sums (bias / placeholder gradients): part[rpb_colsum_rows()][N]. */
int rpb_colsum_rows(void);
int rpb_colsum(const float* x, float* part, long M, int N, int ld, void* stream);
/*     attention among the G slice tokens of every (sample, head) (Physics_Attention.py:164-171, eval mode). */
int rpb_slice_attn(const float* tokS, const float* norm, const float* Wq, const float* Wk, const float* Wv, float* out,
                   int BH, int G, void* stream);
/*     the same attention in TRAINING: nn.Dropout on the attention map as a given inverted-dropout mask amask [BH][G][G] (or NULL),
 *     and -- with go = dLoss/d(out) -- its whole backward: gT [BH][G][32] (w.r.t. tokS), gN [BH][G] (w.r.t. norm) and
 *     gW [BH][3][32][32] (per-(b,h) partials of d to_q / to_k / to_v weights).  out may be NULL in a backward-only call. */
int rpb_slice_attn_train(const float* tokS, const float* norm, const float* Wq, const float* Wk, const float* Wv,
                         const float* amask, const float* go, float* out, float* gT, float* gN, float* gW, int BH, int G,
                         void* stream);
/*     deslice: out[m][h*32+c] = sum_g w[m][h][g] tok2[b][h][g][c] (Physics_Attention.py:173-175). */
int rpb_deslice_fwd(const float* w, const float* tok2, float* out, int B, int ntok, int heads, int G, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Galerkin Transformer.  Token rows [token][256]; the dense products run on
 * rpb_gemm_nt / rpb_gemm_tn, the spectral regressor on K2-K7 above.
 * Reference: realpdebench/model/galerkin_transformer_libs/{layers.py:708-734,829-899,954-987, model.py:93-129,600-638}
 * ------------------------------------------------------------------------------------------------------------------ */

/*     per-head LayerNorm of keys / values: rows of 4 heads x 64 channels, gamma/beta = the 4 nn.LayerNorm(64) affines
 *     back to back (SimpleAttention norm_K / norm_V, layers.py:848-856, 921-935). */
int rpb_headnorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, float* out, int ldo, long M, int C,
                     float eps, void* stream);
/*     its backward; part[rpb_headnorm_bwd_rows(M)][512] partials of (d gamma[256] | d beta[256]). */
long rpb_headnorm_bwd_rows(long M);
int rpb_headnorm_bwd(const float* x, int ldx, const float* gamma, const float* gy, int ldg, float* gx, int ldgx,
                     float* part, long M, int C, float eps, void* stream);
/*     linear_attention (layers.py:708-734) per sample b and head h (4 heads x 64 channels):
 *     head_scores: part[chunk][b][h][i][j] = sum_{m in chunk} G[b,m][64h+i] A[b,m][64h+j]  (K^T V; also dP = Q^T g),
 *     chunk < rpb_head_scores_chunks(B, n); rpb_reduce_partials(rows = chunks, L = B*4*64*64) finishes the sum.
 *     head_apply: out[b,m][64h+j] = (sum_i X[b,m][64h+i] Wm[b][h][i][j]) * mask + residual  (Q (K^T V / n), its
 *     dropout + residual of model.py:112-116 fused; with transposed Wm the three data gradients). */
int rpb_head_scores_chunks(int B, long n);
int rpb_head_scores(const float* G, int ldg, const float* A, int lda, float* part, int B, long n, int nheads, void* stream);
int rpb_head_apply(const float* X, int ldx, const float* Wm, float* out, int ldo, const float* residual, int ldr,
                   const float* mask, int ldm, int B, long n, int nheads, long drop_seed, float drop_keep, void* stream);
/*     out = g * dropout_mask(seed, element index) on a dense tensor (backward of the in-kernel dropout). */
int rpb_dropout_mul(const float* g, float* out, long n, long seed, float keep, void* stream);
/*     nheads = number of 64-channel heads per row (Galerkin: 4; the U-Net's spatial linear attention: 2). */
/*     SpectralRegressor.forward model.py:612-618 after the 256-wide token GEMM U = x fc.weight[:, :256]^T:
 *     out[b,t,h,w,:] = U[token] + fc.weight[:, 256:259] (gt[t], gh[h], gw[w]) + fc.bias inside the mesh, 0 in the
 *     6-cell pad; Wg = that [C][3] slice, contiguous. */
int rpb_pad_grid_fwd(const float* U, const float* gt, const float* gh, const float* gw, const float* Wg,
                     const float* bias, float* out, int B, int T, int H, int W, int C, int Tp, int Hp, int Wp,
                     void* stream);
/*     adjoint w.r.t. U: out[token][:] = g[padded cell of token][:]. */
int rpb_crop_gather(const float* g, float* out, int B, int T, int H, int W, int C, int Tp, int Hp, int Wp, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * U-Net.  Channels-last tokens [B][T*H*W][C]; the 3x3x3 / strided / transposed
 * convolutions and every nn.Linear / 1x1 convolution run on rpb_gemm_nt / rpb_gemm_tn.
 * Reference: realpdebench/model/unet.py
 * ------------------------------------------------------------------------------------------------------------------ */

/*     Block = conv -> GroupNorm(8) -> x*(scale+1)+shift -> SiLU (unet.py:193-208) as per-(sample, channel) passes:
 *     chan_stats -> part[rpb_chan_blocks(B,n)][B][2][C] = (sum x, sum x^2); affine_silu_fwd: y = silu(x*A[b][c] + Bc[b][c]);
 *     bwd_reduce -> part[...][B][2][C] = (sum dz*x, sum dz), dz = gy*silu'(x*A+Bc); bwd_apply: gx = dz*A + P + Q*x. */
int rpb_chan_blocks(int B, long n);
int rpb_chan_stats(const float* x, float* part, int B, long n, int C, void* stream);
int rpb_affine_silu_fwd(const float* x, const float* A, const float* Bc, const float* res, float* y, int B, long n, int C,
                        void* stream);   /* y = silu(x*A + Bc) (+ res) */
int rpb_affine_silu_bwd_reduce(const float* x, const float* gy, const float* A, const float* Bc, float* part, int B, long n,
                               int C, void* stream);
int rpb_affine_silu_bwd_apply(const float* x, const float* gy, const float* A, const float* Bc, const float* P,
                              const float* Q, float* gx, int B, long n, int C, void* stream);
/*     3x3x3 convolution (padding 1) forward / data gradient on the bf16 MFMA with fp32-grade accuracy: fp32 operands are split
 *     into three bf16 terms (hi + mid + lo = all 24 significand bits) and six bf16 x bf16 products per pair are accumulated in
 *     fp32 (csrc/rpb_conv3x.hip).  Same contract as rpb_gemm_nt conv = 1 with a bias-only epilogue -- nn.Conv3d(Ci, Co, 3,
 *     padding=1), realpdebench/model/transolver_libs/Physics_Attention.py:154-157, realpdebench/model/unet.py:196,201 (and their
 *     autograd data gradient with flipped taps): rpb_split3 writes the planes P[3][M][C] (bf16) of x[M][ldx];
 *     rpb_conv3x_wprep turns W[N][27*Ci] (tap-major rows, as rpb_gemm_nt takes them) into MFMA operand order
 *     (3 * N * 27 * Ci bf16); rpb_conv3x computes out[M][ldo] = conv(P, Wz) + bias.  Ci % 64 == 0, N in {64, 128, 256 k}. */
int rpb_split3(const float* x, void* planes, long M, int C, int ldx, void* stream);
int rpb_conv3x_wprep(const float* W, void* Wz, int N, int Ci, void* stream);
int rpb_conv3x(const void* planes, const void* Wz, const float* bias, float* out, long M, int N, int Ci, int ldo, int Hc, int Wc,
               int Dc, void* stream);
/*     ... and its weight / bias gradient (autograd of the same nn.Conv3d): the contraction runs over tokens, so both tensors are
 *     split into planes Pt[3][M/8][C][8] (runs of 8 tokens per channel; rpb_split3t; M % 8 == 0, C % 64 == 0); part[rpb_conv3x_wgrad_splits()][Co*27*Ci
 *     + Co] receives per-split partials of dW[co][tap][ci] and db[co] in the layout of rpb_gemm_tn (finish with
 *     rpb_reduce_partials).  Co % 64 == 0, Ci % 64 == 0, innermost mesh dimension % 8 == 0. */
int rpb_split3t(const float* x, void* planes_t, long M, int C, int ldx, int rev, int d0, int d1, int d2, void* stream);
/*     rev != 0: planes in the token order of the reversed mesh (d2, d1, d0) of x's (d0, d1, d2) mesh -- pass the reversed mesh to
 *     rpb_conv3x_wgrad and transpose the three tap axes of its result (for meshes whose innermost dimension is not % 8). */
int rpb_conv3x_wgrad_splits(long M, int Co, int Ci);
int rpb_conv3x_wgrad(const void* Gt, const void* Xt, float* part, long M, int Co, int Ci, int Hc, int Wc, int Dc, void* stream);
/*     input pipeline: one pass from the full-resolution time slabs as they lie in the
 *     reference's Arrow cells -- planar [B][Cp][horizon][Hf][Wf] ((u, v, p): Cp = 3; combustion's `observed`: Cp = 1) and an
 *     optional channels-last cell cl [B][horizon][Hf][Wf][Cl] (combustion's 15 `numerical` channels) -- to the model's channels-last
 *     input [B][in_step][H][W][Cp + Cl + n_para] and target [B][horizon - in_step][H][W][Cp + Cl]: spatial sub-sampling
 *     [::sub_s, ::sub_s], channel stack, masking (flags[b][c] == 0 -> planar channel c is zeros, flags[b][3] == 0 -> the cl block is),
 *     ControlledCylinder's parameter channels (flags[b][4 + k]) -- realpdebench/data/fluid_hf_dataset.py:280-335,
 *     data/combustion_hf_dataset.py:268-320 -- and GaussianNormalizer.preprocess (data/data_normalizer.py:50-55) fused.
 *     flags is [B][4 + max(n_para, 1)].  rows_subsampled != 0: the staged slabs already hold every sub_s-th row (Hf = the
 *     sub-sampled height; the host copies whole rows, which halves its traffic for sub_s = 2), only columns are strided. */
int rpb_window_pack(const float* planar, const float* cl, const float* flags, float* inp, float* tgt, int B, int horizon,
                    int in_step, int Hf, int Wf, int sub_s, int rows_subsampled, int n_para, int Cp, int Cl, const float* mean_in,
                    const float* mean_tgt, const float* std_in, const float* std_tgt, void* stream);
/*     combustion surrogate samples (realpdebench/data/combustion_surrogate_hf_dataset.py:213-243 +
 *     data/data_normalizer.py:50-55 / :126-131): num [B][ntok][Cl] (the `numerical` windows, channels last) + para [B][n_para]
 *     (numbers parsed from sim_id, appended as constant channels) -> inp [B][ntok][Cl + n_para]; real [B][ntok] -> tgt [B][ntok][1];
 *     both (x - mean) / std per channel (RangeNormalizer: mean = 0, std = max). */
int rpb_pair_pack(const float* num, const float* real, const float* para, float* inp, float* tgt, int B, long ntok, int Cl,
                  int n_para, const float* mean_in, const float* mean_tgt, const float* std_in, const float* std_tgt, void* stream);
/*     rpb_gemm_nt without the convolution modes on the bf16 MFMA from split fp32 operands (csrc/rpb_gemm3x.hip; fp32-grade: hi + mid +
 *     lo, six products per fp32 product): out[M][ldo] = epilogue(A[M][lda] W^T) with W prepared once by rpb_gemm3x_wprep
 *     (W[N][K] -> 3*N*K bf16 in MFMA operand order) and A split on its way into LDS (no extra pass).  Same epilogue arguments and
 *     semantics as rpb_gemm_nt.  K % 64 == 0, N in {64, 128, 256 k}. */
int rpb_gemm3x_wprep(const float* W, void* Wz, int N, int K, void* stream);
int rpb_gemm3x(const float* A, const void* Wz, const float* bias, const float* addvec, const float* residual, float* out, long M,
               int N, int K, int lda, int ldo, int act, const float* aux, float* pre_out, const float* mask, long drop_seed,
               float drop_keep, void* stream);
/*     im2col of init_conv = nn.Conv3d(C_in, dim, KS, padding KS/2) (unet.py:404): col[m][tap*C_in + ci], ldc columns. */
int rpb_im2col(const float* x, float* col, int B, int T, int H, int W, int Cin, int KS, int ldc, void* stream);
/*     temporal attention over the T frames of a location (unet.py:280-356,388): qkv [B][T][HW][384], 4 heads x 32,
 *     rotary tables [T][32], relative-position bias [4][T][T].  bwd: part[rpb_tattn_blocks(B*HW)*4][T*T] bias-gradient
 *     partials, row r belongs to head r % 4. */
int rpb_tattn_blocks(long nloc);
int rpb_tattn_fwd(const float* qkv, const float* rcos, const float* rsin, const float* bias, float* out, int B, int T,
                  int HW, void* stream);
int rpb_tattn_bwd(const float* qkv, const float* rcos, const float* rsin, const float* bias, const float* go, float* gqkv,
                  float* part, int B, int T, int HW, void* stream);
/*     bottleneck softmax attention over the n tokens of a frame (unet.py:455-457; any n, streaming MFMA kernel):
 *     qkv [F][n][384] -> out [F][n][128], lse [F][4][n]; the backward recomputes the probabilities. */
int rpb_sattn_fwd(const float* qkv, float* out, float* lse, int F, int n, void* stream);
int rpb_sattn_bwd(const float* qkv, const float* o, const float* go, float* lse, float* gqkv, int F, int n, void* stream);
/*     SpatialLinearAttention (unet.py:236-261): qe[m] = [softmax_d(q)*32^-1/2 | exp(k - kmax[f])] and its backward
 *     (gqkv columns 0..255 from dqe = [d q' | d E'] and d Z[f][128]); rpb_col_reduce: per-frame column max (mode 0) /
 *     sum (mode 1) partials part[rpb_chan_blocks(F,n)][F][C] of a strided token tensor. */
int rpb_linattn_prep_fwd(const float* qkv, const float* kmax, float* qe, int F, int n, void* stream);
int rpb_linattn_prep_bwd(const float* qe, const float* dqe, const float* dz, float* gqkv, int F, int n, void* stream);
int rpb_col_reduce(const float* x, int ldx, float* part, int F, long n, int C, int mode, void* stream);

/* ---- layer-0 algebra (fno.py:106-111 feeding :113-116): A0 = pad(fc0 [x, grid]) is linear in the F + 1 = Cin + 4 feature fields
 *      phi = (x_j, grid_t, grid_h, grid_w, 1), so the first spectral layer transforms the FIELDS (rpb_axis_gemm on a
 *      [T][H][W][NB] tensor) and these kernels move between the field spectra Phi [2][M][NB] (columns b*Cin + j, then the four
 *      shared fields at B*Cin..) and the 64-channel spectra: Xh[b][r][c] = sum_j W0ext[c][j] Phi[r][col(b, j)] (r = (re/im, mode),
 *      M2 = 2 M rows) and the adjoint sum over (b, r) for d fc0 (part [rpb_feat_mix_wgrad_rows()][C][Cin + 4], last column = d bias).
 *      rpb_small_gemm: out[m][n] (+)= sum_k A[m*a_rs + k*a_cs] B[k*b_rs + n*b_cs] for tiny matrices (the composite weight
 *      Wc0 W0ext and the conv paths of d fc0 / d convs.0.weight from the field moments sum_cells gs0 (x) phi). */
int rpb_feat_mix(const float* Phi, const float* w0, const float* b0, float* Xh, int B, int M2, int NB, int Cin, int C, void* stream);
int rpb_feat_mix_wgrad_rows(void);
int rpb_feat_mix_wgrad(const float* G, const float* Phi, float* part, int B, int M2, int NB, int Cin, int C, void* stream);
int rpb_small_gemm(const float* A, const float* Bm, float* out, int M, int N, int K, int a_rs, int a_cs, int b_rs, int b_cs,
                   int ldo, int accumulate, void* stream);
/*     Phi_c [ncell][FW] (FW = 8 or 32 >= Cin + 4): the feature fields per padded cell (x_j, grid_t, grid_h, grid_w, 1, 0..; zeros in
 *     the pad margin).  Layer 0's channel mixing and conv weight gradient read it instead of the 64-channel lifted tensor:
 *     rpb_cell_mix_feat / rpb_bn_bwd_row_feat. */
int rpb_lift_feat(const float* x, const float* gt, const float* gh, const float* gw, float* out, int B, int T, int H, int W,
                  int Cin, int Tp, int Hp, int Wp, int FW, void* stream);
int rpb_cell_mix_feat(const float* phi, const float* Wcomp, const float* bias, const float* z2, const float* GWt, float* out,
                      float* stats_part, long ncell, int FW, int K2, int Wp, const float* oxf_mean, const float* oxf_invstd,
                      const float* oxf_gamma, const float* oxf_beta, int oxf_gelu, void* stream);
/*     rpb_bn_bwd_row_feat with gs == NULL: the BatchNorm-backward tensor itself is not stored (layer 0 of the fused trainer: its data
 *     gradient is never formed, so nothing reads gs_0; Y1 and the field moments are all that leaves the kernel). */
int rpb_bn_bwd_row_feat(const float* s, const float* gy, const float* phi, float* gs, const float* mean, const float* invstd,
                        const float* gamma, const float* beta, const float* sums, double count, int gelu, const float* GWt,
                        float* Y1, float* part, int G, int Wp, int C, int K2, int FW, void* stream);

/*     Width 128 (configs/fsi/fno.yaml; the Galerkin SpectralRegressor): BatchNorm3d(+GELU) backward apply + adjoint W stage in one pass,
 *     run as the C = 64 row kernel on each 64-channel half of the 512-byte rows (autograd of fno.py:117-119 and of the last inverse
 *     stage, fno.py:63).  s, gy, gs: [G*Wp][128] (gs may alias gy); per-channel vectors [128]; sums [2*128]; Y1 [G][K2][128];
 *     part: scratch of 2 * rpb_bn_bwd_row_slots(G) rows of 64*64+64 floats.  No weight gradient (rpb_cell_wgrad).  Added in round 5
 *     (additions do not change RPB_ABI_VERSION). */
int rpb_bn_bwd_row_c128_supported(int Wp, int K2);
int rpb_bn_bwd_row_c128(const float* s, const float* gy, float* gs, const float* mean, const float* invstd, const float* gamma,
                        const float* beta, const float* sums, double count, int gelu, const float* GWt, float* Y1, float* part,
                        int G, int Wp, int K2, void* stream);

/* ---- backward of the projection head without the gu round trips (fno.py:121-125 autograd; C = 64, DO <= 4, W >= 16):
 *      gh = (fc2^T gout) * act'(fc1 a + b1) is recomputed on the bf16 matrix pipe by each consumer instead of being written once
 *      ([ncrop][128] fp32) and read twice.  `s` is the PADDED pre-BatchNorm tensor of the last Fourier layer, a = xf(s) on the
 *      cropped cells (xf_* = that layer's mean, invstd, gamma, beta, gelu flag); act 0 = exact GELU, 1 = SiLU.
 *      rpb_proj_dgrad: g [ncell][64] = gradient w.r.t. the layer output in the padded layout (zeros in the margin) and
 *        (gu != NULL: gh is READ from gu [ncrop][128] as written by rpb_proj_bwd instead of recomputed -- the faster choice, see
 *        csrc/rpb_pjx.hip -- and gout may be NULL)
 *        stats_part [rpb_proj_dgrad_slots][2][64] = partial (sum g, sum g * shat), shat = (s - mean) * invstd.
 *      rpb_proj_wgrad: part [rpb_proj_wgrad_slots][rpb_proj_wgrad_row(DO)], row `slot` = partial sums for the HB = 128 / roles hidden
 *        units [HB * (slot % roles), + HB): [HB*64] d fc1.weight | [DO*HB] d fc2.weight | [HB] d fc1.bias | [DO] d fc2.bias. */
int rpb_proj_bwd_fused_supported(int C, int DO, int W, int Wp);
long rpb_proj_dgrad_slots(int B, int Tp, int Hp);
int rpb_proj_dgrad(const float* s, const float* w1, const float* b1, const float* w2, const float* gout, const float* gu,
                   float* g, float* stats_part, int B, int DO, int T, int H, int W, int Tp, int Hp, int Wp, const float* xf_mean,
                   const float* xf_invstd, const float* xf_gamma, const float* xf_beta, int xf_gelu, int act, void* stream);
long rpb_proj_wgrad_slots(int B, int T, int H);
int rpb_proj_wgrad_row(int DO);
int rpb_proj_wgrad_roles(void);
int rpb_proj_wgrad(const float* s, const float* w1, const float* b1, const float* w2, const float* gout, float* part, int B,
                   int DO, int T, int H, int W, int Tp, int Hp, int Wp, const float* xf_mean, const float* xf_invstd,
                   const float* xf_gamma, const float* xf_beta, int xf_gelu, int act, void* stream);

/* ---- the whole backward of the projection head in ONE pass (round 3; fno.py:121-125 autograd + the BatchNorm-backward sums of the last
 *      Fourier layer, fno.py:117; C = 64, DO <= 4, exact-GELU head after a BatchNorm without GELU):  csrc/rpb_pjf.hip.
 *      rpb_head_bwd reads the PADDED pre-BatchNorm tensor s of the last layer (a = gamma * shat + beta, shat = (s - mean) * invstd on the
 *      cropped cells) and gout [ncrop][DO]; writes g [ncell][64] = crop-scatter(gh fc1), gh = (fc2^T gout) * gelu'(fc1 a + b1) (zeros in
 *      the pad margin) and per-wave partial rows part [rpb_head_bwd_slots][rpb_head_bwd_row(DO)] =
 *      [128*64] M = gh^T shat | [DO*128] d fc2.weight | [128] d fc1.bias | [DO] d fc2.bias.  gh never reaches HBM.
 *      rpb_head_bwd_finalize takes the row-reduced partials `tot` and writes d fc1.weight = gamma_c M + beta_c (d fc1.bias) [128][64],
 *      d fc2.weight, d fc1.bias, d fc2.bias and bn_sums [2][64] = (sum_cells g, sum_cells g * shat) = (W1^T db1, sum_h W1 .* M). */
int rpb_head_bwd_supported(int C, int DO, int W, int Wp, int xf_gelu, int act);
long rpb_head_bwd_slots(int B, int T, int H);
int rpb_head_bwd_row(int DO);
int rpb_head_bwd(const float* s, const float* w1, const float* b1, const float* w2, const float* gout, float* g, float* part, int B,
                 int DO, int T, int H, int W, int Tp, int Hp, int Wp, const float* xf_mean, const float* xf_invstd,
                 const float* xf_gamma, const float* xf_beta, void* stream);
/*      rpb_head_fwd_bwd (fused trainer): the same pass with the head's FORWARD, the squared-error loss and dLoss/dout formed inside --
 *      out = fc2 gelu(fc1 a + b1) + b2, loss_part [rpb_head_bwd_slots] = per-wave sums of (out - target)^2, gout = gscale (out - target);
 *      replaces rpb_proj_fwd + rpb_mse + rpb_head_bwd of a training step (fno.py:121-125, utils/metrics.py:11-13, train.py:328-329). */
int rpb_head_fwd_bwd(const float* s, const float* w1, const float* b1, const float* w2, const float* b2, const float* target, float gscale,
                     float* g, float* part, float* loss_part, int B, int DO, int T, int H, int W, int Tp, int Hp, int Wp,
                     const float* xf_mean, const float* xf_invstd, const float* xf_gamma, const float* xf_beta, void* stream);
int rpb_head_bwd_finalize(const float* tot, const float* w1, const float* gamma, const float* beta, int DO, float* dw1, float* dw2,
                          float* db1, float* db2, float* bn_sums, void* stream);

/* ---- U-Net: the [B x C]-sized algebra between the token kernels (csrc/rpb_unet_glue.hip; torch autograd glue in rounds 1-2).
 *      rpb_gn_affine_fwd: sums [B][2][C] fp64 (per-channel sum x, sum x^2 over `count / (C/G)` positions; count = elements per group) ->
 *        GroupNorm(G) statistics stat [B][G][2] = (mean, invstd) and the affine A, Bc [B][C] of y = A x + Bc, with the time-embedding
 *        scale|shift ss [B][2C] (or NULL) folded in: A = invstd gamma (1 + scale), Bc = (beta - mean invstd gamma)(1 + scale) + shift
 *        (unet.py:200-208, 223-229).  rpb_gn_affine_bwd: d [B][2][C] = (dL/dA, dL/dBc) -> per-sample parts dgam, dbet [B][C], dss
 *        [B][2C], and P, Q [B][C] (the gradient through the statistics is P + Q x, applied by rpb_affine_silu_bwd_apply).
 *      rpb_silu_fwd / _bwd: SiLU on the time embedding (unet.py:223).  rpb_relpos_bias_fwd / _bwd: bias [heads][n2] =
 *        table[idx[p]][h] and its table gradient; idx [n2] = the reference's T5 bucket of every (i, j) (unet.py:78-116), host-computed. */
int rpb_gn_affine_fwd(const double* sums, const float* gamma, const float* beta, const float* ss, double count, float eps, float* A,
                      float* Bc, float* stat, int B, int C, int G, void* stream);
int rpb_gn_affine_bwd(const float* d, const float* stat, const float* gamma, const float* beta, const float* ss, double count, float* dgam,
                      float* dbet, float* dss, float* P, float* Q, int B, int C, int G, void* stream);
int rpb_silu_fwd(const float* x, float* y, long n, void* stream);
int rpb_silu_bwd(const float* x, const float* gy, float* gx, long n, void* stream);
int rpb_relpos_bias_fwd(const float* table, const int* idx, float* bias, int n2, int heads, void* stream);
int rpb_relpos_bias_bwd(const float* gbias, const int* idx, float* gtable, int n2, int heads, int nbuckets, void* stream);

/* ---- measurement aid (bench.py roofline.copy_ceiling; not on the model path): out = a (* b (+ c)) over n floats, `nread` tensors read
 *      once + one written once with 16 B per lane -- the streaming ceiling of the chip for the read / write mix of the FNO kernels. */
int rpb_stream_probe(const float* a, const float* b, const float* c, float* out, long n, int nread, int threads, void* stream);
/*      rpb_mfma_probe: the matrix pipe's sustained bf16 rate (v_mfma_f32_32x32x16_bf16, register operands built from seed4096[4096]:
 *      random values -> the power-limited rate real data sees, zeros -> the datasheet-like rate); out = scratch of
 *      256 * waves_per_simd * CUs floats; *flops = operations executed by the launch. */
int rpb_mfma_probe(const float* seed4096, float* out, int iters, int waves_per_simd, double* flops, void* stream);

/* ---- eval_metrics (realpdebench/utils/metrics.py:71-100): |F|^2 of the truncated spectrum corner accumulated by radial bin
 *      floor(sqrt(i^2+j^2+k^2)) < R.  Y [R][R][R][2][NB] (re, im planes; columns = (channel, sample)), out [R][NB].  The three
 *      truncated DFT stages in front of it are rpb_axis_gemm launches (realpdebench_amd/metrics.py). */
int rpb_spectrum_bin(const float* Y, float* out, int R, int NB, void* stream);

/* ---- bf16 ACTIVATION STORAGE for the eval / rollout forward (BASELINE.json configs[4]: FNO3d on the combustion volume, "bf16").
 *      The reference has no reduced-precision path; this variant keeps weights, spectra, accumulation and BatchNorm in fp32 and
 *      stores only the [cells][C] activations between kernels as bf16 (round to nearest even): lift -> W stage -> ... ->
 *      cell_mix -> projection read / write 2 bytes per element (SURVEY.md section 8d byte model 0.445 GB*B + 0.537 GB per step).
 *      Replaces the same reference lines as their fp32 twins (fno.py:106-129). */
int rpb_lift_pad_fwd_bf16(const float* x, const float* gt, const float* gh, const float* gw, const float* w0, const float* b0,
                          void* out_bf16, int B, int T, int H, int W, int Cin, int C, int Tp, int Hp, int Wp, void* stream);
int rpb_axis_gemm_bf16in(const void* in_bf16, float* out, const float* Mt, int G, int K, int O, int N, long in_g, long in_k,
                         long out_g, long out_o, int k_valid, void* stream);
/*      spectra_bf16 != 0 (round 4): the SPECTRAL intermediates next to the activations are bf16 too -- the z2 rows these launches read
 *      (written by rpb_axis_gemm_bf16out, the inverse H stage) and the Y1 rows the fused W stage writes (read by rpb_axis_gemm_bf16in);
 *      `z2` / `y1` then point to bf16 rows of 128 B.  Stated tolerance of the rollout unchanged (tests/test_gpu_configs.py). */
int rpb_axis_gemm_bf16out(const float* in, void* out_bf16, const float* Mt, int G, int K, int O, int N, long in_g, long in_k,
                          long out_g, long out_o, int k_valid, void* stream);
int rpb_cell_mix_bf16(const void* x_bf16, const float* Wm, const float* bias, const void* z2, const float* GWt, void* out_bf16,
                      long ncell, int C, int K2, int Wp, const float* oxf_mean, const float* oxf_invstd, const float* oxf_gamma,
                      const float* oxf_beta, int oxf_gelu, int spectra_bf16, void* stream);
int rpb_proj_fwd_bf16(const void* a_bf16, const float* w1, const float* b1, const float* w2, const float* b2, float* out,
                      long ncrop, int C, int DO, int T, int H, int W, int Tp, int Hp, int Wp, int act, void* stream);

/* ---- data-parallel gradient exchange: RCCL all-reduce over xGMI on a side HIP stream (the reference is
 *      single-process, realpdebench/train.py:63).  rank 0 creates 128 opaque bytes with rpb_dp_unique_id and the launcher hands them to
 *      every rank; _init builds the communicator on the CURRENT device plus a side stream; _enqueue sums `buf` in place across ranks
 *      (dtype 0 = fp32, 1 = fp64) once everything queued on producer_stream so far has run, and returns immediately; _wait makes
 *      consumer_stream wait for every bucket enqueued so far; _inline runs the reduction on the given stream (SyncBN statistics).
 *      rpb_dp_available() == 0 when no librccl.so can be loaded (everything else in this header still works). */
int rpb_dp_available(void);
int rpb_dp_unique_id(void* id128);
int rpb_dp_allreduce_init(const void* id128, int rank, int world, void** handle);
int rpb_dp_allreduce_enqueue(void* handle, void* buf, long count, int dtype, void* producer_stream);
int rpb_dp_allreduce_wait(void* handle, void* consumer_stream);
int rpb_dp_allreduce_inline(void* handle, void* buf, long count, int dtype, void* stream);
/*      Sharded optimizer step: rpb_dp_reduce_scatter_enqueue sums the gradient chunk buf[count] over the ranks IN PLACE so that rank r
 *      holds the sum of its piece [r * count / world, (r + 1) * count / world) (count a multiple of world; side stream, ordered after
 *      producer_stream like rpb_dp_allreduce_enqueue); rpb_adam_step_ranges updates the owned pieces; rpb_dp_allgather_enqueue hands every
 *      rank's piece of the parameter chunk to all ranks, in place.  rpb_dp_mark(h, idx) records event idx (0 .. 15) on the side stream,
 *      rpb_dp_wait_mark(h, idx, stream) makes a stream wait for it (the next forward waits per layer for its weights).
 *      rpb_dp_set_model(h, world, gbps, lat_us): one-GPU proxy -- every collective of the handle idles its stream for the modelled ring
 *      transfer over `world` ranks (0 = off). */
int rpb_dp_reduce_scatter_enqueue(void* handle, void* buf, long count, int dtype, void* producer_stream);
int rpb_dp_allgather_enqueue(void* handle, void* buf, long count, int dtype, void* producer_stream);
int rpb_dp_mark(void* handle, int idx);
int rpb_dp_wait_mark(void* handle, int idx, void* stream);
int rpb_dp_set_model(void* handle, int model_world, float gbps, float lat_us);
int rpb_dp_allreduce_destroy(void* handle);
int rpb_dp_allreduce_abort(void* handle);   /* process exit: ncclCommAbort, never blocks on a collective whose peer is gone */
/*      Instrumentation (the N > 1 bench line): rpb_dp_set_timing(h, 1) brackets every bucket / inline reduction with timing events and
 *      restarts the records; after a device synchronisation rpb_dp_step_times fills out[] = { nb, ni, exposed ms (how long after the
 *      consumer stream reached rpb_dp_allreduce_wait the last bucket finished), ms from the first announcement to the last bucket's end,
 *      nb x (start ms since the first announcement, duration ms, bytes), ni x inline duration ms } and returns the count. */
int rpb_dp_set_timing(void* handle, int on);
int rpb_dp_step_times(void* handle, float* out, int max_out);
/*      Opt-in: the optimizer step over PEER POINTERS instead of a collective library (csrc/rpb_p2p.hip; SURVEY.md section 5.8: direct
 *      reduce-scatter + all-gather over the fully connected xGMI mesh).  The host exchanges IPC handles of every rank's gradient arena,
 *      parameter arena and flag block (2 x 16 zeroed 8-byte words) and hands `world` device pointers each -- valid in the calling process,
 *      entry `rank` its own -- to rpb_dp_p2p_init (status: one zeroed int of this rank; total: arena elements, a multiple of 4).
 *      rpb_dp_p2p_adam(h, m, v, ..., step, gscale, stream), called where the all-reduce path calls rpb_adam_step: announces this rank's
 *      gradients (flag GRAD_READY = step at every peer), waits for all peers', then for the slice rpb_dp_p2p_slice names sums the W
 *      gradient arenas in rank order, applies rpb_adam_step's update to p / m / v and stores the new parameters into all W parameter
 *      arenas, and announces PARAM_DONE = step.  rpb_dp_p2p_wait(h, 1, step, stream) before the next read of the parameters (and before
 *      the next backward overwrites the gradient arena).  kind: 0 GRAD_READY, 1 PARAM_DONE.  A wait that sees no flag for timeout_ms sets
 *      *status = 1 + the silent rank and returns control to the stream (the update kernel then leaves the parameters untouched). */
int rpb_dp_p2p_init(int rank, int world, void* const* grads, void* const* params, void* const* flags, void* status, long total,
                    int timeout_ms, void** handle);
int rpb_dp_p2p_slice(void* handle, long* first, long* count);
int rpb_dp_p2p_signal(void* handle, int kind, long step, void* stream);
int rpb_dp_p2p_wait(void* handle, int kind, long step, void* stream);
int rpb_dp_p2p_adam(void* handle, float* m, float* v, float lr, float beta1, float beta2, float eps, long step, float gscale,
                    void* stream);
int rpb_dp_p2p_destroy(void* handle);

/* ---- rollout: eval cell_mix (output = act(BatchNorm(.)) of THIS layer, as rpb_cell_mix with oxf_* / rpb_cell_mix_feat) with the NEXT
 *      layer's forward W stage fused in: y1 [ncell / Wp][K2f][64] = sum_w FWt[w][k] out[line, w][c], i.e. what
 *      rpb_axis_gemm(out, y1, FWt, ...) of fno.py:48 would compute from a second read of the activations (one of the three activation
 *      passes per layer of the evaluation forward).  feat_w > 0: x is the feature tensor [ncell][feat_w], Wm the composite weight.
 *      scratch: 3 * Wp * 64 bytes the launch fills with GWt's bf16 planes in operand order (read back through L1 by the kernel). */
int rpb_cell_mix_eval_dft_supported(long ncell, int K2, int Wp, int K2f);
int rpb_cell_mix_eval_dft(const float* x, const float* Wm, const float* bias, const float* z2, const float* GWt, float* out, long ncell,
                          int K2, int Wp, int feat_w, const float* oxf_mean, const float* oxf_invstd, const float* oxf_gamma,
                          const float* oxf_beta, int oxf_gelu, const float* FWt, int K2f, float* y1, void* scratch, void* stream);

/* the same on bf16-stored activations (BASELINE.json configs[4]): x, out bf16 [ncell][64]; the fused stage is applied to the ROUNDED
 * activations, i.e. y1 is what rpb_axis_gemm_bf16in(out, ...) would compute. */
int rpb_cell_mix_eval_dft_bf16(const void* x_bf16, const float* Wm, const float* bias, const void* z2, const float* GWt, void* out_bf16,
                               long ncell, int K2, int Wp, const float* oxf_mean, const float* oxf_invstd, const float* oxf_gamma,
                               const float* oxf_beta, int oxf_gelu, const float* FWt, int K2f, void* y1, void* scratch, int spectra_bf16,
                               void* stream);

/* ---- rollout: eval cell_mix of the LAST Fourier layer (reference fno.py:117-121: BatchNorm without GELU, then the crop
 *      x[..., :-6, :-6, :-6, :] feeds fc1): only the B * T * H lines of the crop are produced and of each line the 32-cell tiles up to
 *      cell W - 1; pad cells of `out` are left untouched (rpb_proj_fwd reads the crop only).  bf16_io != 0: x / out bf16 [ncell][64]. */
int rpb_cell_mix_eval_crop(const void* x, const float* Wm, const float* bias, const void* z2, const float* GWt, void* out, int B, int T,
                           int H, int W, int Tp, int Hp, int Wp, int K2, const float* oxf_mean, const float* oxf_invstd,
                           const float* oxf_gamma, const float* oxf_beta, int oxf_gelu, int bf16_io, int spectra_bf16, void* stream);

/* the same at width 128 (configs/fsi/fno.yaml, the Galerkin regressor; fp32 storage, K2 <= 32): x / out [cells][128], Wm [128][128] */
int rpb_cell_mix_eval_crop_c128_supported(long ncell, int K2, int Wp);
int rpb_cell_mix_eval_crop_c128(const float* x, const float* Wm, const float* bias, const float* z2, const float* GWt, float* out, int B, int T,
                                int H, int W, int Tp, int Hp, int Wp, int K2, const float* oxf_mean, const float* oxf_invstd,
                                const float* oxf_gamma, const float* oxf_beta, int oxf_gelu, void* stream);

/* ---- rollout, OPT-IN arithmetic "f16x2" (FNO3d.set_arith("f16x2"); never the default, never used by training): the two launches above
 *      on fp32 storage with every operand as TWO fp16 planes rounded to nearest even (x = hi + lo to one fp32 unit in the last place) and
 *      three products hi*lo + lo*hi + hi*hi per fp32 product on v_mfma_f32_16x16x32_f16 -- the dropped lo*lo term is <= 2^-22 |a b| (the
 *      grade of 3xTF32; the default path's six bf16 products drop <= 2^-24).  Half the matrix-pipe time, 2.5 instead of 5.5 vector
 *      instructions per split value.  fp16's range: the conv weights / bias carry 2^4 (undone in the output transform's scale), GWt is
 *      multiplied by 2^spec_exp and the z2 rows by 2^-spec_exp before they are split -- all exact.  spec_exp = floor(log2(Tp Hp Wp)) - 1
 *      (max |GWt| 2^spec_exp in (0.5, 1]).  With feat_w > 0 the K = feat_w mixing of the raw feature fields stays on bf16 planes.
 *      Activations must lie inside fp16's range (|a| < 65504: they are BatchNorm (+GELU) outputs). */
int rpb_cell_mix_eval_dft_f16x2(const float* x, const float* Wm, const float* bias, const float* z2, const float* GWt, float* out,
                                long ncell, int K2, int Wp, int feat_w, const float* oxf_mean, const float* oxf_invstd,
                                const float* oxf_gamma, const float* oxf_beta, int oxf_gelu, const float* FWt, int K2f, float* y1,
                                void* scratch, int spec_exp, void* stream);
int rpb_cell_mix_eval_crop_f16x2(const float* x, const float* Wm, const float* bias, const float* z2, const float* GWt, float* out, int B,
                                 int T, int H, int W, int Tp, int Hp, int Wp, int K2, const float* oxf_mean, const float* oxf_invstd,
                                 const float* oxf_gamma, const float* oxf_beta, int oxf_gelu, int spec_exp, void* stream);

/* the projection head's evaluation forward on the same opt-in arithmetic (C = 64, 1 <= DO <= 4, exact-erf GELU, plain fp32 activations;
 * fc1.weight carries 2^4, undone inside GELU's two affine uses of the accumulator) */
int rpb_proj_fwd_f16x2(const float* a, const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b, float* out,
                       long ncrop, int DO, int T, int H, int W, int Tp, int Hp, int Wp, void* stream);

/* ---- DPOT: AFNO patch transformer (realpdebench/model/dpot.py + dpot_libs/models/dpot.py).  Tokens are
 *      channels-last rows; the dense layers run on rpb_gemm_nt / rpb_gemm_tn, the 2-D DFT stages on rpb_axis_gemm.
 *      rpb_dpot_patch_tokens: PatchEmbed's input gather -- P[((b*nx + px)*ny + py)*T + t][(c*ps + i)*ps + j] for the conv weight
 *        [E][Cm + 3][ps][ps] (dpot.py:199-203): c < Cd data channel of u [B][T][H][W][Cd], Cd <= c < Cm the wrapper's ones padding
 *        (model/dpot.py:213-221), then get_grid_3d's x / y / t coordinates (dpot.py:352-363,370-372).
 *      rpb_rowtable_add / _grad: x[r][:] += table[(r / rows_per_entry) % nent][:] (x + pos_embed, dpot.py:375) and d table.
 *      rpb_dpot_tagg_prep / _finish: TimeAggregator 'exp_mlp' (dpot.py:227-241): e[t][i] = cos(tt[t] gamma[i]); Wb [(t,i)][j] = e w and
 *        its transpose Wf [j][(t,i)], the W operands of the data-gradient / forward token GEMMs (K = T*C resp. N = T*C);
 *        finish: dWb (+ dWsum [C][C] added to every frame's block, optional) -> d w, d gamma.
 *      rpb_gn_tokens_fwd / _bwd: torch.nn.GroupNorm(G, C) (dpot.py:143,151) of x (+ x2) [B][P][C]; stat [B*G][2] = (mean, rstd);
 *        bwd: gx (+ gadd), pg / pb [B][C] per-sample partials of d weight / d bias.
 *      rpb_afno_wprep / _mlp / _wgrad: AFNO2D's block-diagonal complex MLP on the kept modes (dpot.py:72-94).  Spectral rows
 *        [ntok][2 (re, im)][C = nb*bs]; wprep turns w [2][nb][bs][bs] into the real composite [nb][2 bs][2 bs] (transpose != 0: its
 *        transpose); mlp mode 0: out = (gelu(X Wa + ba)) Wb + bb with the pre-activation saved to `mid` (optional),
 *        mode 1: mid = (X Wa) * gelu'(aux), out = mid Wb (data gradient with the transposed composites);
 *        wgrad: dw [2][nb][bs][bs] = complex-structured sum_tok A^T G (A through GELU when a_gelu), part [rpb_afno_wgrad_splits][nb][2bs][2bs].
 *      rpb_dpot_unpatch(_bwd): out_layer rows O[(((b*nx + px)*ny + py)*ps + i)*ps + j][ldo], column t*Co + c, <-> [B][T][H][W][Cd]
 *        (dpot.py:395-396 and the channel slice of model/dpot.py:227). */
int rpb_dpot_patch_tokens(const float* u, const float* gx, const float* gy, const float* gt, float* P, int B, int T, int H, int W,
                          int Cd, int Cm, int ps, void* stream);
/*      gradient of that gather w.r.t. the data frames (sliding-window training feeds predictions back in, model/dpot.py:256-309) */
int rpb_dpot_patch_tokens_bwd(const float* gP, float* gu, int B, int T, int H, int W, int Cd, int Cm, int ps, void* stream);
int rpb_rowtable_add(float* x, const float* table, long M, int C, int rows_per_entry, int nent, void* stream);
int rpb_rowtable_grad(const float* g, float* dtable, int B, int C, int rows_per_entry, int nent, void* stream);
int rpb_dpot_tagg_prep(const float* w, const float* gamma, const float* tt, float* Wf, float* Wb, float* e_out, int T, int C,
                       void* stream);
int rpb_dpot_tagg_finish(const float* dWb, const float* dWsum, const float* w, const float* gamma, const float* tt, float* dw,
                         float* dgamma, int T, int C, void* stream);
int rpb_gn_tokens_fwd(const float* x, const float* x2, const float* gamma, const float* beta, float* y, float* stat, int B, int P,
                      int C, int G, float eps, void* stream);
int rpb_gn_tokens_bwd(const float* x, const float* x2, const float* gamma, const float* stat, const float* gy, const float* gadd,
                      float* gx, float* pg, float* pb, int B, int P, int C, int G, void* stream);
int rpb_afno_wprep(const float* w, float* Wc, int nb, int bs_in, int bs_out, int transpose, void* stream);
int rpb_afno_mlp(const float* X, const float* Wa, const float* ba, const float* Wb, const float* bb, const float* aux, float* mid,
                 float* out, long ntok, int nb, int bs, int mode, void* stream);
int rpb_afno_wgrad_splits(long ntok);
int rpb_afno_wgrad(const float* A, const float* G, float* part, float* dw, long ntok, int nb, int bs, int a_gelu, void* stream);
int rpb_dpot_unpatch(const float* O, float* pred, int B, int T, int H, int W, int Cd, int Co, int ps, int ldo, void* stream);
int rpb_dpot_unpatch_bwd(const float* gpred, float* gO, int B, int T, int H, int W, int Cd, int Co, int ps, int ldo, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RPB_H */
