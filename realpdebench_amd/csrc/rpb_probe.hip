// Measurement aid, not part of the model path: the chip's streaming ceiling for the access MIX of the FNO kernels -- `nread` fp32
// tensors read once and one written once, 16 B per lane, persistent grid-stride waves, two flops per element.  bench.py runs it on
// buffers of one activation tensor's size and reports the dominant kernel family's rate against it (roofline.copy_ceiling): on the
// MI355X boxes of this project a plain copy reaches 5.4-5.7 TB/s of the 8 TB/s HBM3E peak, whatever the pattern
// (tools/ubench/stream_pat.hip), so that rate -- not 8 TB/s -- is the bound a kernel that moves its algorithmic bytes exactly
// once can approach.
#include "rpb_common.h"

template <int NR, int UNR>
__global__ void stream_probe_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b, const f32x4* __restrict__ c,
                                    f32x4* __restrict__ o, long n4) {
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNR - 1) * stride < n4; i += UNR * stride) {
        f32x4 va[UNR], vb[UNR], vc[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            va[u] = a[i + u * stride];
            if (NR > 1) vb[u] = b[i + u * stride];
            if (NR > 2) vc[u] = c[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            f32x4 r = va[u];
            if (NR > 1) r = r * vb[u];
            if (NR > 2) r = r + vc[u];
            o[i + u * stride] = r;
        }
    }
    for (; i < n4; i += stride) o[i] = a[i];
}

// out[i] = a[i] (* b[i] (+ c[i])) over n floats (n % 4 == 0); nread in 1..3; threads = 256 or 512
extern "C" int rpb_stream_probe(const float* a, const float* b, const float* c, float* out, long n, int nread, int threads,
                                void* stream) {
    RPB_REQUIRE(a && out && n > 0 && n % 4 == 0 && nread >= 1 && nread <= 3 && (threads == 256 || threads == 512) &&
                (nread < 2 || b) && (nread < 3 || c), "stream_probe: bad arguments");
    const dim3 grid(rpb_num_cus()), block(threads);
    hipStream_t st = (hipStream_t)stream;
    const f32x4 *a4 = (const f32x4*)a, *b4 = (const f32x4*)b, *c4 = (const f32x4*)c;
    if (nread == 1) hipLaunchKernelGGL((stream_probe_kernel<1, 8>), grid, block, 0, st, a4, b4, c4, (f32x4*)out, n / 4);
    else if (nread == 2) hipLaunchKernelGGL((stream_probe_kernel<2, 8>), grid, block, 0, st, a4, b4, c4, (f32x4*)out, n / 4);
    else hipLaunchKernelGGL((stream_probe_kernel<3, 8>), grid, block, 0, st, a4, b4, c4, (f32x4*)out, n / 4);
    RPB_CHECK_LAUNCH("stream_probe");
}
