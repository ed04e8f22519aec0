// HBM-streaming kernels of the FNO3d step: lift+pad (K1), BatchNorm3d + GELU forward/backward (K6),
// MSE, Adam (K8), rollout affine (K9) and the deterministic partial-sum reducers.
// All are float4 / grid-stride; reductions go per-thread -> LDS -> one partial row per block and are
// finished by rpb_reduce_partials* in fp64 (no atomics: bit-reproducible run to run).
#include "rpb_common.h"

#define PW_THREADS 256
#ifndef RPB_ADAM_NT
#define RPB_ADAM_NT (RPB_STREAM_AUX == 2)     /* Adam streams 2.8 GB once: nontemporal like the row kernels (rpb_common.h) */
#endif

static inline int pw_grid(long work_items, int per_cu = 8) {
    long g = (work_items + PW_THREADS - 1) / PW_THREADS;
    const long cap = (long)rpb_num_cus() * per_cu;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// ---------------------------------------------------------------------------------- K1 lift + pad
// out[b,t,h,w,:] = fc0.weight @ [x[b,t,h,w,:], gt[t], gh[h], gw[w]] + fc0.bias  for t<T,h<H,w<W, else 0
// (fno.py:106-111: get_grid, cat, fc0, permute, F.pad -- the permute disappears: we stay channels-last)
#define LIFT_FMAX 24
// A block walks whole (b,t,h) rows of the padded tensor: the row decode is scalar, the row's features (C_in inputs of
// W cells + the three grid coordinates) are staged once in LDS, and a thread produces 4 channels (one 16 B store) of
// every (256/(C/4))-th cell.  Rows in the pad margin are plain zero fills.  F = C_in + 3 is a template parameter (the
// reference's datasets have C_in = 2, 3, 5, 16; other counts take the generic instantiation).
template <int FT>
__global__ __launch_bounds__(PW_THREADS) void lift_pad_kernel(const float* __restrict__ x, const float* __restrict__ gt,
                                                              const float* __restrict__ gh, const float* __restrict__ gw,
                                                              const float* __restrict__ w0, const float* __restrict__ b0,
                                                              float* __restrict__ out, long nrows_pad, int Cin, int C,
                                                              CropMap cm, int out_bf16) {
    extern __shared__ float wl[];   // [F][C] transposed fc0.weight, bias [C], then the feature row [W][Cin]
    const int F = FT > 0 ? FT : Cin + 3;
    float* xrow = wl + (F + 1) * C;
    for (int idx = threadIdx.x; idx < F * C; idx += blockDim.x) {
        const int j = idx / C, o = idx - j * C;
        wl[idx] = w0[o * F + j];
    }
    for (int idx = threadIdx.x; idx < C; idx += blockDim.x) wl[F * C + idx] = b0[idx];
    const int c4n = C >> 2;
    const int c4 = threadIdx.x % c4n, sub = threadIdx.x / c4n, nsub = blockDim.x / c4n;
    const int o = c4 * 4;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    // wide inputs (C_in = 16: the combustion volume): the thread's 4 x C_in weights live in registers, the cell's inputs are read
    // from LDS as 16 B vectors -- per cell 4 LDS reads instead of 32 (the kernel was LDS-bound at 1.4 TB/s)
    constexpr bool WREG = FT > 8 && (FT - 3) % 4 == 0;
    f32x4 wreg[WREG ? FT - 3 : 1];
    if (WREG) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < (WREG ? FT - 3 : 0); ++j) wreg[j] = *reinterpret_cast<const f32x4*>(wl + j * C + o);
    }
    float pre[8];
    bool have = false;
    for (long row = blockIdx.x; row < nrows_pad; row += gridDim.x) {
        const int h = (int)(row % cm.Hp);
        const long r2 = row / cm.Hp;
        const int t = (int)(r2 % cm.Tp);
        const long b = r2 / cm.Tp;
        float* op = out + row * cm.Wp * C + o;
        // bf16 activation storage (BASELINE.json configs[4]): the same cell rows at 2 bytes per channel, rounded to nearest even
        __bf16* ob = reinterpret_cast<__bf16*>(out) + row * cm.Wp * C + o;
        typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
        typedef float f32x4w __attribute__((ext_vector_type(4)));
        if (h >= cm.H || t >= cm.T) {                                   // uniform: whole row is padding
            for (int w = sub; w < cm.Wp; w += nsub) {
                if (out_bf16) *reinterpret_cast<uint2*>(ob + (long)w * C) = uint2{0u, 0u};
                else *reinterpret_cast<f32x4*>(op + (long)w * C) = z4;
            }
            continue;
        }
        const float* xp = x + (((b * cm.T + t) * cm.H + h) * (long)cm.W) * Cin;
        const int rowlen = cm.W * Cin;
        const bool pf = rowlen <= PW_THREADS * 8;                       // rows of <= 2048 inputs: the next row is prefetched into registers
        if (pf && !have) {
#pragma unroll
            for (int i = 0; i < 8; ++i) pre[i] = threadIdx.x + PW_THREADS * i < rowlen ? xp[threadIdx.x + PW_THREADS * i] : 0.f;
        }
        __syncthreads();                                               // previous row's readers are done (and wl is filled)
        if (pf) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (threadIdx.x + PW_THREADS * i < rowlen) xrow[threadIdx.x + PW_THREADS * i] = pre[i];
        } else {
            for (int idx = threadIdx.x; idx < rowlen; idx += blockDim.x) xrow[idx] = xp[idx];
        }
        __syncthreads();
        have = false;
        if (pf && row + gridDim.x < nrows_pad) {                        // this block's next row, unless it is a pad row (those read nothing)
            const long rn = row + gridDim.x;
            const int hn = (int)(rn % cm.Hp);
            const long r2n = rn / cm.Hp;
            const int tn = (int)(r2n % cm.Tp);
            if (hn < cm.H && tn < cm.T) {
                const float* xn = x + ((((r2n / cm.Tp) * cm.T + tn) * cm.H + hn) * (long)cm.W) * Cin;
#pragma unroll
                for (int i = 0; i < 8; ++i) pre[i] = threadIdx.x + PW_THREADS * i < rowlen ? xn[threadIdx.x + PW_THREADS * i] : 0.f;
                have = true;
            }
        }
        // row-constant part: bias + gt[t] * W[:, Cin] + gh[h] * W[:, Cin+1]
        const int Ci = F - 3;
        f32x4 base = *reinterpret_cast<const f32x4*>(wl + F * C + o);
        base += *reinterpret_cast<const f32x4*>(wl + Ci * C + o) * gt[t];
        base += *reinterpret_cast<const f32x4*>(wl + (Ci + 1) * C + o) * gh[h];
        const f32x4 wwv = *reinterpret_cast<const f32x4*>(wl + (Ci + 2) * C + o);
        for (int w = sub; w < cm.Wp; w += nsub) {
            f32x4 v = z4;
            if (w < cm.W) {
                v = base + wwv * gw[w];
                if (WREG) {
#pragma unroll
                    for (int j4 = 0; j4 < (WREG ? (FT - 3) / 4 : 0); ++j4) {
                        const f32x4 xv = *reinterpret_cast<const f32x4*>(xrow + w * (FT - 3) + 4 * j4);
                        v += wreg[4 * j4] * xv.x;
                        v += wreg[4 * j4 + 1] * xv.y;
                        v += wreg[4 * j4 + 2] * xv.z;
                        v += wreg[4 * j4 + 3] * xv.w;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < (FT > 0 ? FT - 3 : LIFT_FMAX - 3); ++j)
                        if (FT > 0 || j < Ci) v += *reinterpret_cast<const f32x4*>(wl + j * C + o) * xrow[w * Ci + j];
                }
            }
            if (out_bf16) {
                const bf16x4v b = __builtin_convertvector(f32x4w{v[0], v[1], v[2], v[3]}, bf16x4v);
                *reinterpret_cast<bf16x4v*>(ob + (long)w * C) = b;
            } else {
                *reinterpret_cast<f32x4*>(op + (long)w * C) = v;
            }
        }
    }
}

template <int FT>
static void lift_pad_launch(int grid, size_t lds, hipStream_t st, const float* x, const float* gt, const float* gh,
                            const float* gw, const float* w0, const float* b0, float* out, long nrows, int Cin, int C,
                            CropMap cm, int out_bf16) {
    hipLaunchKernelGGL(lift_pad_kernel<FT>, dim3(grid), dim3(PW_THREADS), lds, st, x, gt, gh, gw, w0, b0, out, nrows, Cin, C,
                       cm, out_bf16);
}

// csrc/rpb_lift_mx.hip: the bf16-output lift at C_in = 16, C = 64 as one MFMA K-step (RPB_LIFT_MX=0 keeps the vector kernel below)
bool rpb_lift_mx_supported(int Cin, int C);
int rpb_lift_mx_launch(const float* x, const float* gt, const float* gh, const float* gw, const float* w0, const float* b0, void* out_bf16,
                       int B, int T, int H, int W, int Tp, int Hp, int Wp, hipStream_t st);

static int lift_pad_impl(const float* x, const float* gt, const float* gh, const float* gw, const float* w0,
                         const float* b0, float* out, int B, int T, int H, int W, int Cin, int C, int Tp, int Hp,
                         int Wp, int out_bf16, void* stream) {
    RPB_REQUIRE(x && gt && gh && gw && w0 && b0 && out, "lift_pad: null pointer");
    if (out_bf16 && rpb_lift_mx_supported(Cin, C)) return rpb_lift_mx_launch(x, gt, gh, gw, w0, b0, out, B, T, H, W, Tp, Hp, Wp, (hipStream_t)stream);
    RPB_REQUIRE(C % 4 == 0 && PW_THREADS % (C / 4) == 0 && Cin >= 0 && Cin + 3 <= LIFT_FMAX, "lift_pad: C=%d Cin=%d unsupported", C, Cin);
    const long nrows = (long)B * Tp * Hp;
    const int F = Cin + 3;
    const size_t lds = ((size_t)(F + 1) * C + (size_t)W * Cin) * 4;
    long grid = (long)rpb_num_cus() * 8;
    if (grid > nrows) grid = nrows;
    const CropMap cm{T, H, W, Tp, Hp, Wp};
    hipStream_t st = (hipStream_t)stream;
    switch (F) {
        case 5: lift_pad_launch<5>((int)grid, lds, st, x, gt, gh, gw, w0, b0, out, nrows, Cin, C, cm, out_bf16); break;
        case 6: lift_pad_launch<6>((int)grid, lds, st, x, gt, gh, gw, w0, b0, out, nrows, Cin, C, cm, out_bf16); break;
        case 8: lift_pad_launch<8>((int)grid, lds, st, x, gt, gh, gw, w0, b0, out, nrows, Cin, C, cm, out_bf16); break;
        case 19: lift_pad_launch<19>((int)grid, lds, st, x, gt, gh, gw, w0, b0, out, nrows, Cin, C, cm, out_bf16); break;
        default: lift_pad_launch<0>((int)grid, lds, st, x, gt, gh, gw, w0, b0, out, nrows, Cin, C, cm, out_bf16); break;
    }
    RPB_CHECK_LAUNCH("lift_pad");
}

extern "C" int rpb_lift_pad_fwd(const float* x, const float* gt, const float* gh, const float* gw, const float* w0,
                                const float* b0, float* out, int B, int T, int H, int W, int Cin, int C, int Tp, int Hp,
                                int Wp, void* stream) {
    return lift_pad_impl(x, gt, gh, gw, w0, b0, out, B, T, H, W, Cin, C, Tp, Hp, Wp, 0, stream);
}

extern "C" int rpb_lift_pad_fwd_bf16(const float* x, const float* gt, const float* gh, const float* gw, const float* w0,
                                     const float* b0, void* out_bf16, int B, int T, int H, int W, int Cin, int C, int Tp,
                                     int Hp, int Wp, void* stream) {
    return lift_pad_impl(x, gt, gh, gw, w0, b0, (float*)out_bf16, B, T, H, W, Cin, C, Tp, Hp, Wp, 1, stream);
}

// d fc0.weight[o][j] = sum_cells g[cell][o] * feat[cell][j],  d fc0.bias[o] = sum_cells g[cell][o]
// part row layout: [C*F] weight grad (o*F + j) then [C] bias grad.
// Same row walk as lift_pad: the feature row sits in LDS, a thread owns 4 channels (one 16 B load per cell) and every
// (256/(C/4))-th cell of the row; the two row-constant grid features (t, h) are accumulated once per row from the
// row's bias sum instead of once per cell.
template <int FT>
__global__ __launch_bounds__(PW_THREADS) void lift_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                              const float* __restrict__ gt, const float* __restrict__ gh,
                                                              const float* __restrict__ gw, float* __restrict__ part,
                                                              long ncrop, int Cin, int C, CropMap cm) {
    extern __shared__ float red[];  // [nsub][F+1][C], then the feature row [W][Cin]
    const int F = FT > 0 ? FT : Cin + 3;
    const int Ci = F - 3;
    const int c4n = C >> 2;
    const int c4 = threadIdx.x % c4n, sub = threadIdx.x / c4n, nsub = blockDim.x / c4n;
    const int c = c4 * 4;
    float* xrow = red + (long)nsub * (F + 1) * C;
    constexpr int NA = FT > 0 ? FT : LIFT_FMAX;
    f32x4 acc[NA + 1];              // [0, Ci): inputs, Ci: t, Ci+1: h, Ci+2: w, NA: bias
#pragma unroll
    for (int j = 0; j <= NA; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long nrows = ncrop / cm.W;
    for (long row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int h = (int)(row % cm.H);
        const long r2 = row / cm.H;
        const int t = (int)(r2 % cm.T);
        const long b = r2 / cm.T;
        const float* gp = g + (((b * cm.Tp + t) * cm.Hp + h) * (long)cm.Wp) * C + c;
        const float* xp = x + row * cm.W * Cin;
        if (Ci > 0) {
            __syncthreads();
            for (int idx = threadIdx.x; idx < cm.W * Ci; idx += blockDim.x) xrow[idx] = xp[idx];
            __syncthreads();
        }
        f32x4 rs = {0.f, 0.f, 0.f, 0.f}, rw = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int w = sub; w < cm.W; w += nsub) {
            const f32x4 gv = *reinterpret_cast<const f32x4*>(gp + (long)w * C);
#pragma unroll
            for (int j = 0; j < NA - 3; ++j)
                if (FT > 0 || j < Ci) acc[j] += gv * xrow[w * Ci + j];
            rw += gv * gw[w];
            rs += gv;
        }
        if (FT > 0) {
            acc[FT - 3] += rs * gt[t];
            acc[FT - 2] += rs * gh[h];
            acc[FT - 1] += rw;
        } else {
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                if (j == Ci) acc[j] += rs * gt[t];
                else if (j == Ci + 1) acc[j] += rs * gh[h];
                else if (j == Ci + 2) acc[j] += rw;
            }
        }
        acc[NA] += rs;
    }
#pragma unroll
    for (int j = 0; j < NA; ++j)
        if (j < F) *reinterpret_cast<f32x4*>(red + ((long)(sub * (F + 1) + j)) * C + c) = acc[j];
    *reinterpret_cast<f32x4*>(red + ((long)(sub * (F + 1) + F)) * C + c) = acc[NA];
    __syncthreads();
    float* prow = part + (long)blockIdx.x * ((long)C * F + C);
    for (int idx = threadIdx.x; idx < C * (F + 1); idx += blockDim.x) {
        const int j = idx / C, oo = idx - j * C;
        float s = 0.f;
        for (int k = 0; k < nsub; ++k) s += red[((long)(k * (F + 1) + j)) * C + oo];
        if (j < F) prow[oo * F + j] = s;
        else prow[C * F + oo] = s;
    }
}

extern "C" int rpb_lift_bwd_rows() { return rpb_num_cus() * 4; }

extern "C" int rpb_lift_bwd(const float* g, const float* x, const float* gt, const float* gh, const float* gw,
                            float* part, int B, int T, int H, int W, int Cin, int C, int Tp, int Hp, int Wp,
                            void* stream) {
    RPB_REQUIRE(g && x && gt && gh && gw && part, "lift_bwd: null pointer");
    RPB_REQUIRE(Cin >= 0 && Cin + 3 <= LIFT_FMAX && C % 4 == 0 && PW_THREADS % (C / 4) == 0, "lift_bwd: C=%d Cin=%d unsupported", C,
                Cin);
    const long ncrop = (long)B * T * H * W;
    const int nsub = PW_THREADS / (C / 4);
    const size_t lds = ((size_t)nsub * (Cin + 4) * C + (size_t)W * Cin) * 4;
    RPB_REQUIRE(lds <= 160 * 1024, "lift_bwd: C=%d Cin=%d W=%d does not fit LDS", C, Cin, W);
    const CropMap cm{T, H, W, Tp, Hp, Wp};
    hipStream_t st = (hipStream_t)stream;
    const int grid = rpb_lift_bwd_rows();
#define RPB_LB(FT_)                                                                                                       \
    (void)hipFuncSetAttribute((const void*)lift_bwd_kernel<FT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
    hipLaunchKernelGGL(lift_bwd_kernel<FT_>, dim3(grid), dim3(PW_THREADS), lds, st, g, x, gt, gh, gw, part, ncrop, Cin, C, cm)
    switch (Cin + 3) {
        case 3: RPB_LB(3); break;
        case 5: RPB_LB(5); break;
        case 6: RPB_LB(6); break;
        case 8: RPB_LB(8); break;
        case 19: RPB_LB(19); break;
        default: RPB_LB(0); break;
    }
#undef RPB_LB
    RPB_CHECK_LAUNCH("lift_bwd");
}

// ---------------------------------------------------------------------------------- reducers
// out[j] (+)= scale * sum_r part[r*row_stride + j], j < L   (fp64 accumulation)
// CL columns x RG = 1024 / CL row groups per block.  Wide reductions (many columns) use 64 x 16: 256 B rows per wave; TALL ones (the
// BatchNorm / weight-gradient partials: thousands of rows, a few hundred columns) use 16 x 64 -- a thread's serial chain of dependent
// fp64 adds is a quarter as long and four times as many blocks share the work (2048 rows x 128 columns: ~20 us -> ~6 us, and most
// reductions of a train step are of that shape).
template <int CL>
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ part, long rows, long L,
                                                               long row_stride, float* __restrict__ outf,
                                                               double* __restrict__ outd, double scale,
                                                               int accumulate, long batch_stride) {
    constexpr int RG = 1024 / CL;
    // 4 independent fp64 chains per thread keep the loads pipelined
    // blockIdx.y = batch: an independent reduction over `rows` rows starting batch_stride floats further, out + y * L
    __shared__ double red[RG][CL];
    part += (long)blockIdx.y * batch_stride;
    if (outf) outf += (long)blockIdx.y * L;
    if (outd) outd += (long)blockIdx.y * L;
    const int cl = threadIdx.x % CL, rg = threadIdx.x / CL;
    const long j = (long)blockIdx.x * CL + cl;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (j < L) {
        long r = rg;
        for (; r + 3 * RG < rows; r += 4 * RG) {
            s0 += (double)part[r * row_stride + j];
            s1 += (double)part[(r + RG) * row_stride + j];
            s2 += (double)part[(r + 2 * RG) * row_stride + j];
            s3 += (double)part[(r + 3 * RG) * row_stride + j];
        }
        for (; r < rows; r += RG) s0 += (double)part[r * row_stride + j];
    }
    red[rg][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && j < L) {
        double v = 0.0;
#pragma unroll 8
        for (int k = 0; k < RG; ++k) v += red[k][cl];
        v *= scale;
        if (outd) outd[j] = accumulate ? outd[j] + v : v;
        if (outf) outf[j] = accumulate ? (float)((double)outf[j] + v) : (float)v;
    }
}

static bool reduce_tall(long rows, long L) { return rows >= 256 && L <= 16384; }

extern "C" int rpb_reduce_partials(const float* part, long rows, long L, long row_stride, float* outf, double* outd,
                                   double scale, int accumulate, void* stream) {
    RPB_REQUIRE(part && (outf || outd) && rows > 0 && L > 0 && row_stride >= L, "reduce_partials: bad arguments");
    if (reduce_tall(rows, L))
        hipLaunchKernelGGL(reduce_partials_kernel<16>, dim3((unsigned)((L + 15) / 16)), dim3(1024), 0, (hipStream_t)stream,
                           part, rows, L, row_stride, outf, outd, scale, accumulate, 0L);
    else
        hipLaunchKernelGGL(reduce_partials_kernel<64>, dim3((unsigned)((L + 63) / 64)), dim3(1024), 0, (hipStream_t)stream,
                           part, rows, L, row_stride, outf, outd, scale, accumulate, 0L);
    RPB_CHECK_LAUNCH("reduce_partials");
}

// nbatch independent reductions in one launch: out[b][j] = sum_r part[b*batch_stride + r*row_stride + j]
extern "C" int rpb_reduce_partials_batched(const float* part, int nbatch, long rows, long L, long row_stride, long batch_stride,
                                           float* outf, void* stream) {
    RPB_REQUIRE(part && outf && nbatch > 0 && nbatch < 65536 && rows > 0 && L > 0 && row_stride >= L, "reduce_partials_batched: bad arguments");
    hipLaunchKernelGGL(reduce_partials_kernel<64>, dim3((unsigned)((L + 63) / 64), (unsigned)nbatch), dim3(1024), 0,
                       (hipStream_t)stream, part, rows, L, row_stride, outf, (double*)nullptr, 1.0, 0, batch_stride);
    RPB_CHECK_LAUNCH("reduce_partials_batched");
}

// n independent reductions of DIFFERENT shapes in one launch (the ~10 weight / bias / norm gradients of every transformer block end in
// a partial reduction of a few microseconds each: 68 launches = 1.0 of DPOT-S's 9.8 ms step).  items [n][6] int64 on the device:
// { part pointer, out pointer (fp32), rows, L, row_stride (floats), first 64-column chunk of this item in the launch's grid }.
// Same arithmetic as rpb_reduce_partials (fp64 accumulation, fixed order: bit-reproducible).
// A block owns RG_SUB consecutive 64-column chunks of ONE item (the item lookup -- a scan of the table by one lane -- is paid once per
// 1024 columns, not once per 64: the first version spent 0.8 of its 1.6 ms on it).
#define RG_SUB 16
__global__ __launch_bounds__(1024) void reduce_grouped_kernel(const long* __restrict__ items, int n) {
    constexpr int CL = 64, RG = 1024 / CL;
    __shared__ double red[RG][CL];
    __shared__ int it_s;
    if (threadIdx.x == 0) {
        int lo = 0;
        for (int i = 1; i < n; ++i)
            if (items[6 * i + 5] <= (long)blockIdx.x) lo = i;
        it_s = lo;
    }
    __syncthreads();
    const long* it = items + 6 * it_s;
    const float* part = reinterpret_cast<const float*>(it[0]);
    float* outf = reinterpret_cast<float*>(it[1]);
    const long rows = it[2], L = it[3], row_stride = it[4];
    const int cl = threadIdx.x % CL, rg = threadIdx.x / CL;
    const long j0 = ((long)blockIdx.x - it[5]) * (CL * RG_SUB);
    for (int sub = 0; sub < RG_SUB; ++sub) {
        const long j = j0 + sub * CL + cl;
        if (j0 + sub * CL >= L) break;                     // uniform
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (j < L) {
            long r = rg;
            for (; r + 3 * RG < rows; r += 4 * RG) {
                s0 += (double)part[r * row_stride + j];
                s1 += (double)part[(r + RG) * row_stride + j];
                s2 += (double)part[(r + 2 * RG) * row_stride + j];
                s3 += (double)part[(r + 3 * RG) * row_stride + j];
            }
            for (; r < rows; r += RG) s0 += (double)part[r * row_stride + j];
        }
        if (sub) __syncthreads();                          // the previous sub-chunk's reads of red[] are done
        red[rg][cl] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (rg == 0 && j < L) {
            double v = 0.0;
#pragma unroll 8
            for (int k = 0; k < RG; ++k) v += red[k][cl];
            outf[j] = (float)v;
        }
    }
}
extern "C" int rpb_reduce_partials_grouped(const void* items, int n, long total_chunks, void* stream) {
    RPB_REQUIRE(items && n > 0 && n <= 4096 && total_chunks > 0 && total_chunks < (1L << 31), "reduce_partials_grouped: bad arguments");
    hipLaunchKernelGGL(reduce_grouped_kernel, dim3((unsigned)total_chunks), dim3(1024), 0, (hipStream_t)stream, (const long*)items, n);
    RPB_CHECK_LAUNCH("reduce_partials_grouped");
}
extern "C" int rpb_reduce_partials_grouped_cols(void) { return 64 * RG_SUB; }      // columns per grid block: chunk0 counts blocks of this size

// ---------------------------------------------------------------------------------- K6 BatchNorm3d (+GELU)
// sums = [sum_c, sumsq_c] in fp64 over `count` cells (all ranks, after the optional SyncBN all-reduce).
// Produces batch mean / invstd and updates the running statistics exactly like nn.BatchNorm3d(momentum=0.1):
// biased variance to normalise, unbiased variance into running_var (fno.py:117).
__global__ void bn_finalize_kernel(const double* __restrict__ sums, double count, float eps, float momentum,
                                   float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ rmean,
                                   float* __restrict__ rvar, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = sums[c] / count;
    double var = sums[C + c] / count - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        rmean[c] = (float)((1.0 - momentum) * (double)rmean[c] + momentum * m);
        rvar[c] = (float)((1.0 - momentum) * (double)rvar[c] + momentum * unb);
    }
}

extern "C" int rpb_bn_finalize(const double* sums, double count, float eps, float momentum, float* mean, float* invstd,
                               float* rmean, float* rvar, int C, void* stream) {
    RPB_REQUIRE(sums && mean && invstd && count > 0 && C > 0, "bn_finalize: bad arguments");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, sums, count, eps,
                       momentum, mean, invstd, rmean, rvar, C);
    RPB_CHECK_LAUNCH("bn_finalize");
}

__global__ void bn_eval_prep_kernel(const float* __restrict__ rvar, float eps, float* __restrict__ invstd, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) invstd[c] = 1.0f / sqrtf(rvar[c] + eps);
}
extern "C" int rpb_bn_eval_prep(const float* rvar, float eps, float* invstd, int C, void* stream) {
    RPB_REQUIRE(rvar && invstd && C > 0, "bn_eval_prep: bad arguments");
    hipLaunchKernelGGL(bn_eval_prep_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, rvar, eps, invstd, C);
    RPB_CHECK_LAUNCH("bn_eval_prep");
}

// y = act(gamma * (s - mean) * invstd + beta),  act = exact-erf GELU or identity (last layer, fno.py:118)
template <bool GELU>
__global__ __launch_bounds__(PW_THREADS) void bn_act_kernel(const float* __restrict__ s, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            long n4, int C) {
    // the grid stride is a multiple of C/4, so every thread keeps ONE group of 4 channels: parameters live in registers
    const int c4n = C >> 2;
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)(i0 % c4n) * 4;
    float mu[4], sc[4], be[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mu[k] = mean[c + k];
        sc[k] = invstd[c + k];
        be[k] = beta[c + k];
    }
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
    for (long idx = i0; idx < n4; idx += (long)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4*>(s)[idx];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float z = (v[k] - mu[k]) * sc[k] * ga[k] + be[k];
            o[k] = GELU ? gelu_f(z) : z;
        }
        reinterpret_cast<f32x4*>(y)[idx] = o;
    }
}

extern "C" int rpb_bn_act_fwd(const float* s, const float* mean, const float* invstd, const float* gamma,
                              const float* beta, float* y, long ncell, int C, int gelu, void* stream) {
    RPB_REQUIRE(s && mean && invstd && gamma && beta && y && C % 4 == 0 && PW_THREADS % (C / 4) == 0,
                "bn_act_fwd: bad arguments");
    const long n4 = ncell * (C / 4);
    if (gelu)
        hipLaunchKernelGGL(bn_act_kernel<true>, dim3(pw_grid(n4)), dim3(PW_THREADS), 0, (hipStream_t)stream, s, mean,
                           invstd, gamma, beta, y, n4, C);
    else
        hipLaunchKernelGGL(bn_act_kernel<false>, dim3(pw_grid(n4)), dim3(PW_THREADS), 0, (hipStream_t)stream, s, mean,
                           invstd, gamma, beta, y, n4, C);
    RPB_CHECK_LAUNCH("bn_act_fwd");
}

// backward pass 1: per-channel  sum gz  and  sum gz*shat,  gz = gy * act'(z)
template <bool GELU>
__global__ __launch_bounds__(PW_THREADS) void bn_bwd_reduce_kernel(const float* __restrict__ s,
                                                                   const float* __restrict__ gy,
                                                                   const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta,
                                                                   float* __restrict__ part, long ncell, int C) {
    extern __shared__ float red[];   // [nsub][2][C]
    const int c4n = C >> 2;
    const int c4 = threadIdx.x % c4n, sub = threadIdx.x / c4n, nsub = blockDim.x / c4n;
    const int c = c4 * 4;
    float mu[4], is[4], ga[4], be[4], a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mu[k] = mean[c + k]; is[k] = invstd[c + k]; ga[k] = gamma[c + k]; be[k] = beta[c + k];
    }
    if (sub < nsub) {
        for (long cell = (long)blockIdx.x * nsub + sub; cell < ncell; cell += (long)gridDim.x * nsub) {
            const f32x4 sv = *reinterpret_cast<const f32x4*>(s + cell * C + c);
            const f32x4 gv = *reinterpret_cast<const f32x4*>(gy + cell * C + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float sh = (sv[k] - mu[k]) * is[k];
                const float gz = GELU ? gv[k] * gelu_grad_f(sh * ga[k] + be[k]) : gv[k];
                a1[k] += gz;
                a2[k] += gz * sh;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            red[(sub * 2 + 0) * C + c + k] = a1[k];
            red[(sub * 2 + 1) * C + c + k] = a2[k];
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 2 * C; idx += blockDim.x) {
        float t = 0.f;
        for (int k = 0; k < nsub; ++k) t += red[k * 2 * C + idx];
        part[(long)blockIdx.x * 2 * C + idx] = t;
    }
}

extern "C" int rpb_bn_bwd_rows() { return rpb_num_cus() * 8; }

extern "C" int rpb_bn_bwd_reduce(const float* s, const float* gy, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, float* part, long ncell, int C, int gelu,
                                 void* stream) {
    RPB_REQUIRE(s && gy && mean && invstd && gamma && beta && part, "bn_bwd_reduce: null pointer");
    RPB_REQUIRE(C % 4 == 0 && PW_THREADS % (C / 4) == 0, "bn_bwd_reduce: C=%d unsupported", C);
    const int nsub = PW_THREADS / (C / 4);
    const size_t lds = (size_t)nsub * 2 * C * 4;
    const int grid = rpb_bn_bwd_rows();
    if (gelu)
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<true>, dim3(grid), dim3(PW_THREADS), lds, (hipStream_t)stream, s, gy,
                           mean, invstd, gamma, beta, part, ncell, C);
    else
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<false>, dim3(grid), dim3(PW_THREADS), lds, (hipStream_t)stream, s, gy,
                           mean, invstd, gamma, beta, part, ncell, C);
    RPB_CHECK_LAUNCH("bn_bwd_reduce");
}

// backward pass 2: gs = gamma*invstd * (gz - dbeta/N - shat * dgamma/N);  sums = [dbeta | dgamma] (fp32, global)
template <bool GELU>
__global__ __launch_bounds__(PW_THREADS) void bn_bwd_apply_kernel(const float* __restrict__ s, const float* gy,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta,
                                                                  const float* __restrict__ sums, float inv_count,
                                                                  float* gs, long n4, int C) {
    const int c4n = C >> 2;
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)(i0 % c4n) * 4;
    float mu[4], is[4], ga[4], be[4], m1[4], m2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mu[k] = mean[c + k];
        is[k] = invstd[c + k];
        ga[k] = gamma[c + k];
        be[k] = beta[c + k];
        m1[k] = sums[c + k] * inv_count;
        m2[k] = sums[C + c + k] * inv_count;
    }
    for (long idx = i0; idx < n4; idx += (long)gridDim.x * blockDim.x) {
        const f32x4 sv = reinterpret_cast<const f32x4*>(s)[idx];
        const f32x4 gv = reinterpret_cast<const f32x4*>(gy)[idx];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float sh = (sv[k] - mu[k]) * is[k];
            const float gz = GELU ? gv[k] * gelu_grad_f(sh * ga[k] + be[k]) : gv[k];
            o[k] = ga[k] * is[k] * (gz - m1[k] - sh * m2[k]);
        }
        reinterpret_cast<f32x4*>(gs)[idx] = o;
    }
}

extern "C" int rpb_bn_bwd_apply(const float* s, const float* gy, const float* mean, const float* invstd,
                                const float* gamma, const float* beta, const float* sums, double count, float* gs,
                                long ncell, int C, int gelu, void* stream) {
    RPB_REQUIRE(s && gy && mean && invstd && gamma && beta && sums && gs && C % 4 == 0 && PW_THREADS % (C / 4) == 0,
                "bn_bwd_apply: bad arguments");
    const long n4 = ncell * (C / 4);
    const float ic = (float)(1.0 / count);
    if (gelu)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(pw_grid(n4)), dim3(PW_THREADS), 0, (hipStream_t)stream, s, gy,
                           mean, invstd, gamma, beta, sums, ic, gs, n4, C);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(pw_grid(n4)), dim3(PW_THREADS), 0, (hipStream_t)stream, s,
                           gy, mean, invstd, gamma, beta, sums, ic, gs, n4, C);
    RPB_CHECK_LAUNCH("bn_bwd_apply");
}

// ---------------------------------------------------------------------------------- MSE (metrics.py:11-13 + .mean())
// elem = (pred-target)^2 (optional), gout = 2*(pred-target)*gscale, part[block] = sum elem
__global__ __launch_bounds__(PW_THREADS) void mse_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                         float* __restrict__ elem, float* __restrict__ gout,
                                                         float* __restrict__ part, long n, float gscale) {
    __shared__ float red[PW_THREADS / 64];
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = pred[i] - tgt[i];
        const float e = d * d;
        if (elem) elem[i] = e;
        if (gout) gout[i] = 2.f * d * gscale;
        acc += e;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < PW_THREADS / 64; ++k) t += red[k];
        part[blockIdx.x] = t;
    }
}

extern "C" int rpb_mse_rows() { return rpb_num_cus() * 4; }

extern "C" int rpb_mse(const float* pred, const float* tgt, float* elem, float* gout, float* part, long n, float gscale,
                       void* stream) {
    RPB_REQUIRE(pred && tgt && part && n > 0, "mse: bad arguments");
    hipLaunchKernelGGL(mse_kernel, dim3(rpb_mse_rows()), dim3(PW_THREADS), 0, (hipStream_t)stream, pred, tgt, elem, gout,
                       part, n, gscale);
    RPB_CHECK_LAUNCH("mse");
}

// ---------------------------------------------------------------------------------- K8 Adam
// torch.optim.Adam defaults (train.py:290): complex parameters are updated as 2x fp32 (view_as_real).
// lr / bias corrections are host scalars computed from the step count (no device sync, no .item()).
// one element's update, shared by both kernels and compiled without FMA contraction: the plain and the ranged launch must agree bit
// for bit (the sharded optimizer step of the data-parallel path is tested against the plain one), and the separate roundings are
// torch.optim.Adam's own (it runs the update as a chain of element-wise kernels)
__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, float gscale, float b1, float b2, float eps,
                                            float step_size, float inv_sqrt_bc2) {
#pragma clang fp contract(off)
    const float gk = g * gscale;
    m = b1 * m + (1.f - b1) * gk;
    v = b2 * v + (1.f - b2) * gk * gk;
    const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
    p -= step_size * (m / denom);
}

__global__ __launch_bounds__(PW_THREADS) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v, long n,
                                                          float gscale, float b1, float b2, float eps, float step_size,
                                                          float inv_sqrt_bc2) {
    const long n4 = n >> 2;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n4; idx += (long)gridDim.x * blockDim.x) {
#if RPB_ADAM_NT     /* nontemporal loads / stores */
        f32x4 pv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(p) + idx);
        const f32x4 gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + idx);
        f32x4 mv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m) + idx);
        f32x4 vv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v) + idx);
#else
        f32x4 pv = reinterpret_cast<f32x4*>(p)[idx];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[idx];
        f32x4 mv = reinterpret_cast<f32x4*>(m)[idx];
        f32x4 vv = reinterpret_cast<f32x4*>(v)[idx];
#endif
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pk = pv[k], mk = mv[k], vk = vv[k];
            adam_update(pk, gv[k], mk, vk, gscale, b1, b2, eps, step_size, inv_sqrt_bc2);
            pv[k] = pk, mv[k] = mk, vv[k] = vk;
        }
#if RPB_ADAM_NT
        __builtin_nontemporal_store(pv, reinterpret_cast<f32x4*>(p) + idx);
        __builtin_nontemporal_store(mv, reinterpret_cast<f32x4*>(m) + idx);
        __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v) + idx);
#else
        reinterpret_cast<f32x4*>(p)[idx] = pv;
        reinterpret_cast<f32x4*>(m)[idx] = mv;
        reinterpret_cast<f32x4*>(v)[idx] = vv;
#endif
    }
    // tail
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float pk = p[i], mk = m[i], vk = v[i];
        adam_update(pk, g[i], mk, vk, gscale, b1, b2, eps, step_size, inv_sqrt_bc2);
        m[i] = mk;
        v[i] = vk;
        p[i] = pk;
    }
}

extern "C" int rpb_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                             float eps, long step, float gscale, void* stream) {
    RPB_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam_step: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float isb2 = (float)(1.0 / sqrt(bc2));
    hipLaunchKernelGGL(adam_kernel, dim3(pw_grid((n + 3) / 4)), dim3(PW_THREADS), 0, (hipStream_t)stream, p, g, m, v, n,
                       gscale, beta1, beta2, eps, step_size, isb2);
    RPB_CHECK_LAUNCH("adam_step");
}

// The same update on a LIST of ranges of the arena (the sharded optimizer step of the data-parallel path: a rank updates the pieces of
// the parameter arena it owns).  tab [nr][2] (device, int64) = (first element, float4 groups before this range); starts and counts are
// multiples of 4; total4 = float4 groups over all ranges.
__global__ __launch_bounds__(PW_THREADS) void adam_ranges_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                                 float* __restrict__ v, const long* __restrict__ tab, int nr, long total4,
                                                                 float gscale, float b1, float b2, float eps, float step_size,
                                                                 float inv_sqrt_bc2) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += (long)gridDim.x * blockDim.x) {
        int lo = 0, hi = nr - 1;                         // last range whose prefix <= idx
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tab[2 * mid + 1] <= idx) lo = mid;
            else hi = mid - 1;
        }
        const long e4 = (tab[2 * lo] >> 2) + (idx - tab[2 * lo + 1]);
        f32x4 pv = reinterpret_cast<f32x4*>(p)[e4];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[e4];
        f32x4 mv = reinterpret_cast<f32x4*>(m)[e4];
        f32x4 vv = reinterpret_cast<f32x4*>(v)[e4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pk = pv[k], mk = mv[k], vk = vv[k];
            adam_update(pk, gv[k], mk, vk, gscale, b1, b2, eps, step_size, inv_sqrt_bc2);
            pv[k] = pk, mv[k] = mk, vv[k] = vk;
        }
        reinterpret_cast<f32x4*>(p)[e4] = pv;
        reinterpret_cast<f32x4*>(m)[e4] = mv;
        reinterpret_cast<f32x4*>(v)[e4] = vv;
    }
}

extern "C" int rpb_adam_step_ranges(float* p, const float* g, float* m, float* v, const long* tab, int nr, long total, float lr,
                                    float beta1, float beta2, float eps, long step, float gscale, void* stream) {
    RPB_REQUIRE(p && g && m && v && tab && nr > 0 && total > 0 && total % 4 == 0 && step >= 1, "adam_step_ranges: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float isb2 = (float)(1.0 / sqrt(bc2));
    hipLaunchKernelGGL(adam_ranges_kernel, dim3(pw_grid(total / 4)), dim3(PW_THREADS), 0, (hipStream_t)stream, p, g, m, v, tab, nr,
                       total / 4, gscale, beta1, beta2, eps, step_size, isb2);
    RPB_CHECK_LAUNCH("adam_step_ranges");
}

// ---------------------------------------------------------------------------------- K9 rollout affine
// eval.py:316-318 between two autoregressive steps, with data_normalizer.py:50-62 arithmetic kept in the
// reference's order:  t = p*std_t + mean_t ; [cat control channels] ; out = (t - mean_i)/std_i
__global__ __launch_bounds__(PW_THREADS) void rollout_affine_kernel(const float* __restrict__ pred,
                                                                    const float* __restrict__ para,
                                                                    float* __restrict__ out, long ncell, int Cp, int Cx,
                                                                    const float* __restrict__ mean_t,
                                                                    const float* __restrict__ std_t,
                                                                    const float* __restrict__ mean_i,
                                                                    const float* __restrict__ std_i) {
    const int Ct = Cp + Cx;
    const long n = ncell * Ct;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long cell = i / Ct;
        const int c = (int)(i - cell * Ct);
        float t;
        if (c < Cp) {
            t = pred[cell * Cp + c];
            if (mean_t) t = t * std_t[c] + mean_t[c];
        } else {
            t = para[cell * Cx + (c - Cp)];
        }
        out[i] = mean_i ? (t - mean_i[c]) / std_i[c] : t;
    }
}

extern "C" int rpb_rollout_affine(const float* pred, const float* para, float* out, long ncell, int Cp, int Cx,
                                  const float* mean_t, const float* std_t, const float* mean_i, const float* std_i,
                                  void* stream) {
    RPB_REQUIRE(pred && out && ncell > 0 && Cp > 0 && Cx >= 0, "rollout_affine: bad arguments");
    RPB_REQUIRE(Cx == 0 || para, "rollout_affine: control channels requested without a source");
    RPB_REQUIRE((mean_t == nullptr) == (std_t == nullptr) && (mean_i == nullptr) == (std_i == nullptr),
                "rollout_affine: mean/std must come in pairs");
    hipLaunchKernelGGL(rollout_affine_kernel, dim3(pw_grid(ncell * (Cp + Cx))), dim3(PW_THREADS), 0, (hipStream_t)stream,
                       pred, para, out, ncell, Cp, Cx, mean_t, std_t, mean_i, std_i);
    RPB_CHECK_LAUNCH("rollout_affine");
}

// per-channel normalise / denormalise of a channels-last tensor (GaussianNormalizer.preprocess/postprocess)
__global__ __launch_bounds__(PW_THREADS) void channel_affine_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                    long n, int C, const float* __restrict__ mean,
                                                                    const float* __restrict__ stdv, int inverse) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        out[i] = inverse ? in[i] * stdv[c] + mean[c] : (in[i] - mean[c]) / stdv[c];
    }
}
extern "C" int rpb_channel_affine(const float* in, float* out, long n, int C, const float* mean, const float* stdv,
                                  int inverse, void* stream) {
    RPB_REQUIRE(in && out && mean && stdv && n > 0 && C > 0, "channel_affine: bad arguments");
    hipLaunchKernelGGL(channel_affine_kernel, dim3(pw_grid(n)), dim3(PW_THREADS), 0, (hipStream_t)stream, in, out, n, C,
                       mean, stdv, inverse);
    RPB_CHECK_LAUNCH("channel_affine");
}

// out = a * b elementwise (dropout-mask application in the Transolver backward)
__global__ __launch_bounds__(PW_THREADS) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ out, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(a)[i] * reinterpret_cast<const f32x4*>(b)[i];
}
extern "C" int rpb_mul(const float* a, const float* b, float* out, long n, void* stream) {
    RPB_REQUIRE(a && b && out && n > 0 && n % 4 == 0, "mul: n must be a positive multiple of 4");
    hipLaunchKernelGGL(mul_kernel, dim3(pw_grid(n / 4)), dim3(PW_THREADS), 0, (hipStream_t)stream, a, b, out, n / 4);
    RPB_CHECK_LAUNCH("mul");
}

// out = a + b elementwise: the sum of two gradients where the U-Net's tape forks (residual / skip connections)
__global__ __launch_bounds__(PW_THREADS) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ out, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        RPB_SST4(out + 4 * i, RPB_SLD4(a + 4 * i) + RPB_SLD4(b + 4 * i));
}
extern "C" int rpb_add(const float* a, const float* b, float* out, long n, void* stream) {
    RPB_REQUIRE(a && b && out && n > 0 && n % 4 == 0, "add: n must be a positive multiple of 4");
    hipLaunchKernelGGL(add_kernel, dim3(pw_grid(n / 4)), dim3(PW_THREADS), 0, (hipStream_t)stream, a, b, out, n / 4);
    RPB_CHECK_LAUNCH("add");
}

// dst[m][doff .. doff + C) = src[m][soff .. soff + C) for m < M (row strides ldd / lds floats): the skip-connection concat of
// the U-Net (torch.cat((x, skip), dim=1), unet.py:463,479) and its backward split, as strided row copies (C % 4 == 0)
__global__ __launch_bounds__(PW_THREADS) void copy_cols_kernel(const float* __restrict__ src, float* __restrict__ dst, long M,
                                                               int C4, int lds4, int ldd4, int soff4, int doff4) {
    const long total = M * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / C4;
        const int c = (int)(i - m * C4);
        reinterpret_cast<f32x4*>(dst)[m * ldd4 + doff4 + c] = reinterpret_cast<const f32x4*>(src)[m * lds4 + soff4 + c];
    }
}
extern "C" int rpb_copy_cols(const float* src, float* dst, long M, int C, int lds, int ldd, int soff, int doff, void* stream) {
    RPB_REQUIRE(src && dst && M > 0 && C > 0 && ((C | lds | ldd | soff | doff) & 3) == 0, "copy_cols: sizes must be multiples of 4");
    hipLaunchKernelGGL(copy_cols_kernel, dim3(pw_grid(M * (C / 4))), dim3(PW_THREADS), 0, (hipStream_t)stream, src, dst, M,
                       C / 4, lds / 4, ldd / 4, soff / 4, doff / 4);
    RPB_CHECK_LAUNCH("copy_cols");
}
