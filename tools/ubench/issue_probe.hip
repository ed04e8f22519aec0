// What does one gfx950 SIMD issue per cycle when matrix and vector instructions share it?  Prices the design questions of the projection-head
// rewrite:
//   A  one wave per SIMD, NM MFMAs per iteration with K independent vector instructions pinned between every two MFMAs (sched_barrier keeps
//      the order): cycles per iteration against K, for the 16x16x32 and the 32x32x16 bf16 shapes, scalar v_fma_f32 and packed v_pk_fma_f32 fillers
//   B  vector instructions alone at 1 / 2 waves per SIMD: cycles per v_fma_f32, per v_pk_fma_f32, per v_exp_f32
//   C  two waves per SIMD with ROLES: waves 0-3 only MFMAs, waves 4-7 only vector work -- does the pair run in max(a, b) or in a + b?
//   D  two waves per SIMD, both running the interleaved body of A
// hipcc --offload-arch=gfx950 -O3 -o issue_probe issue_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define SB() __builtin_amdgcn_sched_barrier(0)

// FILL: 0 = v_fma_f32, 1 = v_pk_fma_f32, 2 = v_exp_f32
template <int FILL>
__device__ __forceinline__ void filler(float (&v)[16], f2 (&w)[8], int idx, float s) {
    // asm volatile: the order written is the order issued (the IR optimisers otherwise sink, merge and SLP-pack these across sched_barriers)
    if (FILL == 0) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(v[idx & 15]) : "v"(s));
    else if (FILL == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(w[idx & 7]) : "v"(f2{s, s}));
    else asm volatile("v_exp_f32 %0, %0" : "+v"(v[idx & 15]));
}

// SHAPE 0: 16x16x32 (8 accumulators of 4), 1: 32x32x16 (4 accumulators of 16);  ROLE: 0 = every wave runs MFMA + fillers, 1 = waves < 4 MFMA only,
// waves >= 4 fillers only (NM * K of them per iteration)
template <int SHAPE, int NM, int K, int FILL, int ROLE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float s, long long* cyc) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i * 0.37f); b[i] = (__bf16)(s * 0.01f + i * 0.11f); }
    f4 acc[8];
    f16v big[4];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
    float v[16];
    f2 w[8];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.5f + i;
    for (int i = 0; i < 8; ++i) w[i] = f2{threadIdx.x * 0.25f + i, threadIdx.x * 0.125f - i};
    const int wave = threadIdx.x >> 6;
    const bool do_m = ROLE == 0 || wave < 4, do_v = ROLE == 0 || wave >= 4;
    const long long t0 = clock64();
    if (do_m && do_v) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                if (SHAPE == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a), "v"(b));
                else if (SHAPE == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(big[m & 3]) : "v"(a), "v"(b));
                SB();
#pragma unroll
                for (int j = 0; j < K; ++j) filler<FILL>(v, w, m * K + j, s);
                SB();
            }
        }
    } else if (do_m) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                if (SHAPE == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(big[m & 3]) : "v"(a), "v"(b));
                SB();
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < NM * K; ++j) {
                filler<FILL>(v, w, j, s);
                SB();
            }
        }
    }
    const long long t1 = clock64();
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) r += big[i][j];
    for (int i = 0; i < 16; ++i) r += v[i];
    for (int i = 0; i < 8; ++i) r += w[i][0] + w[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static float* out;
static long long* cyc;
static hipEvent_t e0, e1;
template <int SHAPE, int NM, int K, int FILL, int ROLE>
static void run(int threads, const char* what) {
    const int iters = 4000;
    long long hc[8] = {0};
    hipLaunchKernelGGL((k<SHAPE, NM, K, FILL, ROLE>), dim3(256), dim3(threads), 0, 0, out, 50, 1.0001f, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, NM, K, FILL, ROLE>), dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hc, cyc, 64, hipMemcpyDeviceToHost);
    const double c0 = (double)hc[0] / iters, c4 = threads > 256 ? (double)hc[4] / iters : 0.0;
    printf("%-34s %s NM %2d K %2d fill %s waves/SIMD %d: %8.1f cyc/iter (wave 0)  %8.1f (wave 4)  = %6.2f cyc/MFMA-slot  %.3f ms  %.2f GHz\n", what,
           SHAPE == 2 ? "no MFMA " : SHAPE ? "32x32x16" : "16x16x32", NM, K, FILL == 0 ? "fma " : FILL == 1 ? "pkfma" : "exp ", threads / 256, c0, c4, c0 / NM, ms,
           hc[0] / (ms * 1e6));
}

int main() {
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 64);
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("## A: one wave per SIMD, K fillers pinned after every MFMA\n");
#define ROW(S, F)                                                                                                         \
    run<S, 8, 0, F, 0>(256, "A");  run<S, 8, 1, F, 0>(256, "A");  run<S, 8, 2, F, 0>(256, "A");  run<S, 8, 3, F, 0>(256, "A");     \
    run<S, 8, 4, F, 0>(256, "A");  run<S, 8, 6, F, 0>(256, "A");  run<S, 8, 8, F, 0>(256, "A");  run<S, 8, 12, F, 0>(256, "A");
    ROW(0, 0) ROW(1, 0) ROW(0, 1) ROW(1, 1) ROW(1, 2)
    printf("## B: vector instructions alone (SHAPE 2 = no MFMA in the body), 64 per iteration, at one and two waves per SIMD\n");
    run<2, 8, 8, 0, 0>(256, "B");  run<2, 8, 8, 1, 0>(256, "B");  run<2, 8, 8, 2, 0>(256, "B");
    run<2, 8, 8, 0, 0>(512, "B");  run<2, 8, 8, 1, 0>(512, "B");  run<2, 8, 8, 2, 0>(512, "B");
    printf("## C: wave roles at two waves per SIMD: waves 0-3 MFMA only, waves 4-7 NM*K fillers only\n");
    run<0, 8, 2, 0, 1>(512, "C");  run<0, 8, 4, 0, 1>(512, "C");  run<0, 8, 8, 0, 1>(512, "C");
    run<1, 8, 4, 0, 1>(512, "C");  run<1, 8, 8, 0, 1>(512, "C");  run<1, 8, 16, 0, 1>(512, "C");
    run<0, 8, 4, 1, 1>(512, "C");  run<1, 8, 8, 1, 1>(512, "C");
    printf("## D: two waves per SIMD, both interleaved\n");
    run<0, 8, 2, 0, 0>(512, "D");  run<0, 8, 4, 0, 0>(512, "D");  run<0, 8, 8, 0, 0>(512, "D");
    run<1, 8, 4, 0, 0>(512, "D");  run<1, 8, 8, 0, 0>(512, "D");  run<1, 8, 16, 0, 0>(512, "D");
    run<0, 8, 4, 1, 0>(512, "D");  run<1, 8, 8, 1, 0>(512, "D");
    return 0;
}
