#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    u2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
    int y = threadIdx.x;
    out[128 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, y, 0x140, 0xF, 0xF, true);
    out[192 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, y, 0x141, 0xF, 0xF, true);
    out[256 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, y, 0x4E, 0xF, 0xF, true);
    out[320 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, y, 0xB1, 0xF, 0xF, true);
}
int main() {
    unsigned* d; hipMalloc(&d, 384 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[6] = {"swap r0", "swap r1", "row_mirror", "half_mirror", "qp 2301", "qp 1032"};
    for (int j = 0; j < 6; ++j) { printf("%s:", nm[j]); for (int i = 0; i < 64; ++i) printf(" %u", h[64 * j + i]); printf("\n"); }
    return 0;
}
