// Sustained bf16 MFMA rate of the chip on NON-TRIVIAL operands (the power-limited "practical peak" the split-bf16 kernels are priced
// against): independent accumulators, register operands, 1 / 2 / 4 waves per SIMD; operands random (high toggle rate) or zero.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int SHAPE>
__global__ __launch_bounds__(256) void k(float* out, int iters, const float* seed) {
    bf16x8 a[4], b[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 8; ++i) {
            a[j][i] = (__bf16)seed[(threadIdx.x * 8 + i + 64 * j) & 4095];
            b[j][i] = (__bf16)seed[(threadIdx.x * 8 + i + 1000 + 64 * j) & 4095];
        }
    if (SHAPE == 16) {
        f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[2], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[3], b[3], c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], c4, 0, 0, 0);
            c5 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[2], c5, 0, 0, 0);
            c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[3], c6, 0, 0, 0);
            c7 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[3], b[0], c7, 0, 0, 0);
        }
        f4 r = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
        out[blockIdx.x * blockDim.x + threadIdx.x] = r[0] + r[1] + r[2] + r[3];
    } else {
        f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[2], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[3], b[3], c3, 0, 0, 0);
        }
        f16v r = c0 + c1 + c2 + c3;
        float s = 0;
        for (int i = 0; i < 16; ++i) s += r[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}
int main() {
    float *out, *seed;
    hipMalloc(&out, 256 * 8192 * 4);
    hipMalloc(&seed, 4096 * 4);
    float h[4096];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int data = 0; data < 2; ++data) {
        srand(1);
        for (int i = 0; i < 4096; ++i) h[i] = data ? (rand() / (float)RAND_MAX - 0.5f) * 4.f : 0.f;
        hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
#define RUN(SHAPE, WPS, PER)                                                                                  \
        {                                                                                                       \
            const int iters = 40000;                                                                            \
            hipLaunchKernelGGL((k<SHAPE>), dim3(256 * WPS), dim3(256), 0, 0, out, 100, seed);                   \
            hipEventRecord(e0);                                                                                 \
            hipLaunchKernelGGL((k<SHAPE>), dim3(256 * WPS), dim3(256), 0, 0, out, iters, seed);                 \
            hipEventRecord(e1); hipEventSynchronize(e1);                                                        \
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);                                                   \
            const double fl = 2.0 * (SHAPE == 16 ? 16 * 16 * 32 : 32 * 32 * 16) * PER * (double)iters * 1024.0 * WPS; \
            printf("%s data, %dx%d, %d waves/SIMD: %.3f ms  %.0f TFLOP/s\n", data ? "random" : "zero  ", SHAPE, SHAPE, WPS, ms, fl / ms / 1e9); \
        }
        RUN(16, 1, 8) RUN(16, 2, 8) RUN(16, 4, 8) RUN(32, 1, 4) RUN(32, 2, 4) RUN(32, 4, 4)
    }
    return 0;
}
