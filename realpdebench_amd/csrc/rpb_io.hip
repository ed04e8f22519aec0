// Input pipeline kernels: what the reference does on CPU worker processes with numpy between
// the Arrow row and the model input (realpdebench/data/fluid_hf_dataset.py:280-335) and then on the device with four broadcast
// ops (data/data_normalizer.py:50-55), as ONE pass over HBM.
#include "rpb_common.h"

struct WindowPackArgs {
    const float* planar;   // [B][3][horizon][Hf][Wf]  full-resolution time slabs of (u, v, p) as they lie in the Arrow cell
    const float* flags;    // [B][4 + max(n_para, 1)]  [c] != 0: channel c present (else zeros: masked / real-data pressure);
                           //                          [4 + k]: k-th parameter parsed from sim_id (ControlledCylinder)
    float* inp;            // [B][in_step][H][W][3 + n_para]
    float* tgt;            // [B][horizon - in_step][H][W][3]
    const float* mean_in;  // [3 + n_para]
    const float* mean_tgt; // [3]
    const float* std_in;
    const float* std_tgt;
    long ntok;             // B * horizon * H * W
    int horizon, in_step, Hf, Wf, H, W, sub_s, n_para, nflag;
};

__global__ __launch_bounds__(256) void window_pack_kernel(WindowPackArgs a) {
    const long per_b = (long)a.horizon * a.H * a.W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.ntok; i += (long)gridDim.x * blockDim.x) {
        const long b = i / per_b;
        long r = i - b * per_b;
        const int t = (int)(r / ((long)a.H * a.W));
        r -= (long)t * a.H * a.W;
        const int h = (int)(r / a.W), w = (int)(r - (long)h * a.W);
        const float* fl = a.flags + b * a.nflag;
        const float* src = a.planar + ((b * 3 * a.horizon + t) * (long)a.Hf + (long)h * a.sub_s) * a.Wf + (long)w * a.sub_s;
        const long cstride = (long)a.horizon * a.Hf * a.Wf;
        float x[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) x[c] = fl[c] != 0.f ? src[c * cstride] : 0.f;
        if (t < a.in_step) {
            const int C = 3 + a.n_para;
            float* dst = a.inp + (((b * a.in_step + t) * a.H + h) * (long)a.W + w) * C;
#pragma unroll
            for (int c = 0; c < 3; ++c) dst[c] = (x[c] - a.mean_in[c]) / a.std_in[c];
            for (int k = 0; k < a.n_para; ++k) dst[3 + k] = (fl[4 + k] - a.mean_in[3 + k]) / a.std_in[3 + k];
        } else {
            float* dst = a.tgt + (((b * (a.horizon - a.in_step) + (t - a.in_step)) * a.H + h) * (long)a.W + w) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) dst[c] = (x[c] - a.mean_tgt[c]) / a.std_tgt[c];
        }
    }
}

extern "C" int rpb_window_pack(const float* planar, const float* flags, float* inp, float* tgt, int B, int horizon, int in_step,
                               int Hf, int Wf, int sub_s, int n_para, const float* mean_in, const float* mean_tgt,
                               const float* std_in, const float* std_tgt, void* stream) {
    RPB_REQUIRE(planar && flags && inp && tgt && mean_in && mean_tgt && std_in && std_tgt, "window_pack: null pointer");
    RPB_REQUIRE(B > 0 && horizon > in_step && in_step > 0 && Hf > 0 && Wf > 0 && sub_s >= 1 && n_para >= 0 && n_para <= 8,
                "window_pack: bad sizes (B=%d horizon=%d in_step=%d sub_s=%d n_para=%d)", B, horizon, in_step, sub_s, n_para);
    WindowPackArgs a;
    a.planar = planar; a.flags = flags; a.inp = inp; a.tgt = tgt;
    a.mean_in = mean_in; a.mean_tgt = mean_tgt; a.std_in = std_in; a.std_tgt = std_tgt;
    a.horizon = horizon; a.in_step = in_step; a.Hf = Hf; a.Wf = Wf; a.sub_s = sub_s; a.n_para = n_para;
    a.H = (Hf + sub_s - 1) / sub_s;                                      // len(range(0, Hf, sub_s)) = numpy's [::sub_s]
    a.W = (Wf + sub_s - 1) / sub_s;
    a.nflag = 4 + (n_para > 1 ? n_para : 1);
    a.ntok = (long)B * horizon * a.H * a.W;
    long grid = (a.ntok + 255) / 256;
    const long cap = (long)rpb_num_cus() * 16;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(window_pack_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    RPB_CHECK_LAUNCH("window_pack");
}
