// Probe of ds_read_b64_tr_b16 on gfx950: LDS holds the element index (u16), lane l supplies address 8 * l (4 contiguous elements);
// prints what each lane receives.     hipcc --offload-arch=gfx950 -O2 tools/ubench/tr_probe.hip -o tools/ubench/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned short* out, int stride_bytes) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)lds + threadIdx.x * stride_bytes;
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    out[threadIdx.x * 4 + 0] = v[0] & 0xffff;
    out[threadIdx.x * 4 + 1] = v[0] >> 16;
    out[threadIdx.x * 4 + 2] = v[1] & 0xffff;
    out[threadIdx.x * 4 + 3] = v[1] >> 16;
}
int main() {
    unsigned short* d;
    hipMalloc(&d, 64 * 4 * 2);
    for (int stride : {8, 32}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
        unsigned short h[256];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("stride %d bytes per lane: lane -> 4 received element indices\n", stride);
        for (int l = 0; l < 64; ++l) printf("%2d: %4d %4d %4d %4d%s", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3], (l % 4 == 3) ? "\n" : "   ");
    }
    return 0;
}
