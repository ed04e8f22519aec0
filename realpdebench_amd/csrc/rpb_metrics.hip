// eval_metrics support (realpdebench/utils/metrics.py:71-100): radial binning of the truncated error spectrum.
//
// The reference takes fftn of prediction and target and walks (t/2)(h/2)(w/2) bins in a Python triple loop (twice) adding
// |F|^2 of bin (i,j,k) to radial bin floor(sqrt(i^2+j^2+k^2)) when that is < R = min(t,h,w)//2.  Only the corner i,j,k < R of
// the spectrum can land in a kept bin, so the three DFT stages are truncated GEMMs (rpb_axis_gemm: the same kernels as the FNO
// spectral layer) over a batch-innermost layout, and this kernel does the |.|^2 + radial accumulation:
//
//   Y   [R][R][R][2 (re, im)][NB]     truncated spectrum, columns = (channel, sample) pairs, NB a multiple of 64
//   out [R][NB]                       out[r][n] = sum over corner bins q with rad(q) == r of re^2 + im^2
//
// One workgroup per radial bin r walks the R^3 corner in a fixed order (deterministic, no atomics); a thread owns columns.
#include "rpb_common.h"

__global__ __launch_bounds__(256) void spectrum_bin_kernel(const float* __restrict__ Y, float* __restrict__ out, int R, int NB) {
    const int r = blockIdx.x;
    for (int n = threadIdx.x; n < NB; n += blockDim.x) {
        float acc = 0.f;
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < R; ++j)
                for (int k = 0; k < R; ++k) {
                    const int rad = (int)floorf(sqrtf((float)(i * i + j * j + k * k)));       // exact for these small integers
                    if (rad != r) continue;                                                     // uniform across the block
                    const long q = ((long)i * R + j) * R + k;
                    const float re = Y[(q * 2 + 0) * NB + n], im = Y[(q * 2 + 1) * NB + n];
                    acc += re * re + im * im;
                }
        out[(long)r * NB + n] = acc;
    }
}

extern "C" int rpb_spectrum_bin(const float* Y, float* out, int R, int NB, void* stream) {
    RPB_REQUIRE(Y && out && R > 0 && R <= 64 && NB > 0, "spectrum_bin: bad arguments (R=%d NB=%d)", R, NB);
    hipLaunchKernelGGL(spectrum_bin_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, Y, out, R, NB);
    RPB_CHECK_LAUNCH("spectrum_bin");
}
