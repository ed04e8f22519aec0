// Measurement aid, not part of the model path: the chip's streaming ceiling for the access MIX of the FNO kernels -- `nread` fp32
// tensors read once and one written once, 16 B per lane, persistent grid-stride waves, two flops per element.  bench.py runs it on
// buffers of one activation tensor's size and reports the dominant kernel family's rate against it (roofline.copy_ceiling): on the
// MI355X boxes of this project a plain copy reaches 5.4-5.7 TB/s of the 8 TB/s HBM3E peak, whatever the pattern
// (tools/ubench/stream_pat.hip), so that rate -- not 8 TB/s -- is the bound a kernel that moves its algorithmic bytes exactly
// once can approach.
// Round 5: the probe streams with the policy the product kernels stream with (RPB_STREAM_AUX == 2: nontemporal loads and stores,
// +5 % on these boxes), so that the ceiling stays the rate of a plain copy written the way the kernels are.
#include "rpb_common.h"

namespace {
__device__ __forceinline__ f32x4 pld(const f32x4* p, long i) {
#if RPB_STREAM_AUX == 2
    return __builtin_nontemporal_load(p + i);
#else
    return p[i];
#endif
}
__device__ __forceinline__ void pst(f32x4* p, long i, f32x4 v) {
#if RPB_STREAM_AUX == 2
    __builtin_nontemporal_store(v, p + i);
#else
    p[i] = v;
#endif
}
}  // namespace

template <int NR, int UNR>
__global__ void stream_probe_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b, const f32x4* __restrict__ c,
                                    f32x4* __restrict__ o, long n4) {
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNR - 1) * stride < n4; i += UNR * stride) {
        f32x4 va[UNR], vb[UNR], vc[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            va[u] = pld(a, i + u * stride);
            if (NR > 1) vb[u] = pld(b, i + u * stride);
            if (NR > 2) vc[u] = pld(c, i + u * stride);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            f32x4 r = va[u];
            if (NR > 1) r = r * vb[u];
            if (NR > 2) r = r + vc[u];
            pst(o, i + u * stride, r);
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

// ---- the matrix pipe's SUSTAINED bf16 rate on non-trivial operands: 4 independent v_mfma_f32_32x32x16_bf16 accumulation chains per wave on
// register operands taken from `seed` (4096 floats: random -> the power-limited rate real data sees; zeros -> the datasheet-like rate),
// `waves_per_simd` waves on every SIMD.  Measured on the MI355X boxes of this project (tools/ubench/mfma_peak.hip): 1.76 PFLOP/s with
// random operands (1.6 for the 16x16x32 shape) against 2.4-2.45 with zeros and the 2.5 PFLOP/s datasheet peak -- the bound the
// split-bf16 convolutions / token GEMMs (six bf16 products per fp32 product) are priced against in bench.py.
typedef __attribute__((ext_vector_type(8))) __bf16 probe_bf16x8;
__global__ __launch_bounds__(256) void mfma_probe_kernel(const float* __restrict__ seed, float* __restrict__ out, int iters) {
    probe_bf16x8 a[4], b[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 8; ++i) {
            a[j][i] = (__bf16)seed[(threadIdx.x * 8 + i + 64 * j) & 4095];
            b[j][i] = (__bf16)seed[(threadIdx.x * 8 + i + 1000 + 64 * j) & 4095];
        }
    f32x16 c0 = zero16(), c1 = zero16(), c2 = zero16(), c3 = zero16();
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[2], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[3], b[3], c3, 0, 0, 0);
    }
    const f32x16 r = c0 + c1 + c2 + c3;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// out: >= 256 * waves_per_simd * CUs floats of scratch; returns through *flops the bf16 floating-point operations the launch executes
extern "C" int rpb_mfma_probe(const float* seed4096, float* out, int iters, int waves_per_simd, double* flops, void* stream) {
    RPB_REQUIRE(seed4096 && out && iters > 0 && waves_per_simd >= 1 && waves_per_simd <= 2, "mfma_probe: bad arguments");
    const int grid = rpb_num_cus() * waves_per_simd;
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, seed4096, out, iters);
    if (flops) *flops = 2.0 * 32 * 32 * 16 * 4.0 * (double)iters * 4.0 * grid;
    RPB_CHECK_LAUNCH("mfma_probe");
}
