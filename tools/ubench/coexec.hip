// Do MFMA and VALU work overlap on one gfx950 SIMD?  Per loop iteration: NM independent bf16 MFMAs (16x16x32) and NV fp32 FMAs on
// other registers, either BLOCKED (all MFMAs, then all FMAs) or INTERLEAVED (one MFMA, then NV/NM FMAs, ...), at 1 / 2 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NM, int NV, int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s, long long* cyc) {
    const long long t0 = clock64();
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(s + i); }
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.5f + i;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 7], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j & 15] = __builtin_fmaf(v[j & 15], s, 1.0f);
        } else {
            constexpr int PER = NM ? NV / NM : NV;
#pragma unroll
            for (int m = 0; m < (NM ? NM : 1); ++m) {
                if (NM) acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 7], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < PER; ++j) v[(m * PER + j) & 15] = __builtin_fmaf(v[(m * PER + j) & 15], s, 1.0f);
                if (NM) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, PER, 0);   // PER VALU
                }
            }
        }
    }
    const long long t1 = clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) r += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
    float* out; hipMalloc(&out, 256 * 4096 * 4); long long* cyc; hipMalloc(&cyc, 8); long long hc = 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
#define RUN(NM, NV, MODE, WPS)                                                                                          \
    {                                                                                                                   \
        hipLaunchKernelGGL((k<NM, NV, MODE>), dim3(256 * WPS), dim3(256), 0, 0, out, 100, 1.0001f, cyc);                       \
        hipEventRecord(e0);                                                                                             \
        hipLaunchKernelGGL((k<NM, NV, MODE>), dim3(256 * WPS), dim3(256), 0, 0, out, iters, 1.0001f, cyc);                     \
        hipEventRecord(e1); hipEventSynchronize(e1);                                                                    \
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);                                                               \
        hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);                                                                  \
        printf("NM %2d NV %3d %s waves/SIMD %d: %.3f ms  %.1f ns/iter  %.1f shader cycles/iter  (%.2f GHz)\n", NM, NV, MODE ? "interleaved" : "blocked    ", WPS, ms, ms * 1e6 / iters, (double)hc / iters, hc / (ms * 1e6)); \
    }
    RUN(8, 0, 0, 1) RUN(0, 32, 0, 1) RUN(8, 32, 0, 1) RUN(8, 32, 1, 1)
    RUN(8, 0, 0, 2) RUN(0, 32, 0, 2) RUN(8, 32, 0, 2) RUN(8, 32, 1, 2)
    RUN(8, 0, 0, 4) RUN(0, 32, 0, 4) RUN(8, 32, 0, 4) RUN(8, 32, 1, 4)
    RUN(8, 16, 0, 1) RUN(8, 16, 1, 1) RUN(8, 64, 0, 1) RUN(8, 64, 1, 1) RUN(8, 64, 0, 2) RUN(8, 64, 1, 2)
    return 0;
}
