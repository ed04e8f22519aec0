// Which streaming PATTERN reaches the chip's copy ceiling?  R reads + 1 write of 3.82 GB tensors, 16 B per lane.
//   mode 0: grid-stride (thread i, i + grid*block, ...)      mode 1: contiguous chunk per workgroup
//   mode 2: wave walks whole 34 KB rows round-robin (cmx / bwd_row pattern)     nt: nontemporal loads / stores
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NR, int MODE, bool NT, int UNR>
__global__ void k(const f4* __restrict__ a, const f4* __restrict__ b, const f4* __restrict__ c, f4* __restrict__ o, long n4) {
    auto ld = [&](const f4* p, long i) { return NT ? __builtin_nontemporal_load(p + i) : p[i]; };
    auto st = [&](f4* p, long i, f4 v) { if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v; };
    long i0, i1, step;
    if (MODE == 0) { step = (long)gridDim.x * blockDim.x; i0 = (long)blockIdx.x * blockDim.x + threadIdx.x; i1 = n4; }
    else if (MODE == 1) { const long per = (n4 / gridDim.x + 63) / 64 * 64; i0 = blockIdx.x * per + threadIdx.x; i1 = min(n4, (blockIdx.x + 1) * per); step = blockDim.x; }
    else { step = 0; i0 = 0; i1 = 0; }
    if (MODE == 2) {
        const int ROW4 = 134 * 16;   // 134 cells x 256 B / 16
        const long nrows = n4 / ROW4, nw = (long)gridDim.x * (blockDim.x >> 6);
        const int lane = threadIdx.x & 63;
        for (long r = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < nrows; r += nw) {
            const long base = r * ROW4;
            for (int j = lane; j < ROW4; j += 64 * UNR) {
                f4 va[UNR], vb[UNR], vc[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) if (j + 64 * u < ROW4) { va[u] = ld(a, base + j + 64 * u); if (NR > 1) vb[u] = ld(b, base + j + 64 * u); if (NR > 2) vc[u] = ld(c, base + j + 64 * u); }
#pragma unroll
                for (int u = 0; u < UNR; ++u) if (j + 64 * u < ROW4) { f4 r2 = va[u]; if (NR > 1) r2 = r2 * vb[u]; if (NR > 2) r2 = r2 + vc[u]; st(o, base + j + 64 * u, r2); }
            }
        }
        return;
    }
    long i = i0;
    for (; i + (UNR - 1) * step < i1; i += UNR * step) {
        f4 va[UNR], vb[UNR], vc[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) { va[u] = ld(a, i + u * step); if (NR > 1) vb[u] = ld(b, i + u * step); if (NR > 2) vc[u] = ld(c, i + u * step); }
#pragma unroll
        for (int u = 0; u < UNR; ++u) { f4 r = va[u]; if (NR > 1) r = r * vb[u]; if (NR > 2) r = r + vc[u]; st(o, i + u * step, r); }
    }
    for (; i < i1; i += step) st(o, i, ld(a, i));
}
int main() {
    const long n = 14939392L * 64;
    float *a, *b, *c, *o;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4); hipMalloc(&o, n * 4);
    hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4); hipMemset(c, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(NR, MODE, NT, UNR, GRID, BLK)                                                                    \
    {                                                                                                        \
        for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k<NR, MODE, NT, UNR>), dim3(GRID), dim3(BLK), 0, 0, (f4*)a, (f4*)b, (f4*)c, (f4*)o, n / 4); \
        hipEventRecord(e0);                                                                                  \
        for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k<NR, MODE, NT, UNR>), dim3(GRID), dim3(BLK), 0, 0, (f4*)a, (f4*)b, (f4*)c, (f4*)o, n / 4); \
        hipEventRecord(e1); hipEventSynchronize(e1);                                                         \
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;                                           \
        printf("R%d mode %d nt %d unroll %d grid %5d x %4d: %.3f ms  %.0f GB/s\n", NR, MODE, (int)NT, UNR, GRID, BLK, ms, (NR + 1) * n * 4.0 / ms / 1e6); \
    }
    RUN(1, 0, false, 4, 256, 512) RUN(1, 0, false, 4, 256, 1024) RUN(1, 0, false, 4, 256, 256) RUN(1, 0, false, 8, 256, 256) RUN(1, 0, false, 2, 256, 1024)
    RUN(1, 0, true, 4, 256, 512) RUN(1, 0, true, 4, 512, 512) RUN(1, 0, true, 4, 1024, 256)
    RUN(1, 1, false, 4, 256, 512) RUN(1, 1, false, 4, 512, 512) RUN(1, 1, false, 4, 2048, 256) RUN(1, 1, true, 4, 256, 512) RUN(1, 1, false, 4, 8192, 256) RUN(1, 1, false, 4, 65536, 256)
    RUN(1, 2, false, 4, 256, 512) RUN(1, 2, false, 8, 256, 512) RUN(1, 2, true, 8, 256, 512) RUN(1, 2, false, 8, 512, 256) RUN(1, 2, false, 8, 256, 256)
    RUN(3, 0, false, 4, 256, 512) RUN(3, 0, true, 4, 256, 512) RUN(3, 1, false, 4, 256, 512) RUN(3, 2, false, 4, 256, 512) RUN(3, 2, false, 8, 256, 512) RUN(3, 2, true, 8, 256, 512) RUN(3, 2, false, 8, 256, 256)
    RUN(2, 0, false, 4, 256, 512) RUN(2, 2, false, 8, 256, 512) RUN(2, 2, true, 8, 256, 512)
    return 0;
}
