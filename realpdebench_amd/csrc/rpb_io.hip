// Input pipeline kernels: what the reference does on CPU worker processes with numpy between
// the Arrow row and the model input (realpdebench/data/fluid_hf_dataset.py:280-335) and then on the device with four broadcast
// ops (data/data_normalizer.py:50-55), as ONE pass over HBM.
#include "rpb_common.h"

struct WindowPackArgs {
    const float* planar;   // [B][Cp][horizon][Hf][Wf]  full-resolution time slabs of the planar cells: (u, v, p) of the fluid
                           //                           scenarios (Cp = 3), `observed` of combustion (Cp = 1)
    const float* cl;       // [B][horizon][Hf][Wf][Cl]  channels-last cell (combustion's 15 `numerical` channels) or null
    const float* flags;    // [B][4 + max(n_para, 1)]  [c] != 0, c < 3: planar channel c present (else zeros: masked / real-data
                           //                          pressure); [3] != 0: the channels-last block is present;
                           //                          [4 + k]: k-th parameter parsed from sim_id (ControlledCylinder)
    float* inp;            // [B][in_step][H][W][Cp + Cl + n_para]
    float* tgt;            // [B][horizon - in_step][H][W][Cp + Cl]
    const float* mean_in;  // [Cp + Cl + n_para]
    const float* mean_tgt; // [Cp + Cl]
    const float* std_in;
    const float* std_tgt;
    long ntok;             // B * horizon * H * W
    int horizon, in_step, Hf, Wf, H, W, sub_s, sub_h, n_para, nflag, Cp, Cl;   // sub_h: row stride in the staged slab (1 if the host already kept every sub_s-th row)
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
        const long pix = (long)h * a.sub_h * a.Wf + (long)w * a.sub_s;                      // [::sub_s, ::sub_s]
        const float* src = a.planar + (b * a.Cp * a.horizon + t) * (long)a.Hf * a.Wf + pix;
        const long cstride = (long)a.horizon * a.Hf * a.Wf;
        const float* srcl = a.cl ? a.cl + ((b * a.horizon + t) * (long)a.Hf * a.Wf + pix) * a.Cl : nullptr;
        const bool lon = a.cl && fl[3] != 0.f;
        const bool is_in = t < a.in_step;
        const int C = a.Cp + a.Cl + (is_in ? a.n_para : 0);
        float* dst = is_in ? a.inp + (((b * a.in_step + t) * a.H + h) * (long)a.W + w) * C
                           : a.tgt + (((b * (a.horizon - a.in_step) + (t - a.in_step)) * a.H + h) * (long)a.W + w) * C;
        const float* mean = is_in ? a.mean_in : a.mean_tgt;
        const float* sd = is_in ? a.std_in : a.std_tgt;
        for (int c = 0; c < a.Cp; ++c) dst[c] = ((fl[c] != 0.f ? src[c * cstride] : 0.f) - mean[c]) / sd[c];
        for (int c = 0; c < a.Cl; ++c) dst[a.Cp + c] = ((lon ? srcl[c] : 0.f) - mean[a.Cp + c]) / sd[a.Cp + c];
        if (is_in)
            for (int k = 0; k < a.n_para; ++k) dst[a.Cp + a.Cl + k] = (fl[4 + k] - mean[a.Cp + a.Cl + k]) / sd[a.Cp + a.Cl + k];
    }
}

extern "C" int rpb_window_pack(const float* planar, const float* cl, const float* flags, float* inp, float* tgt, int B, int horizon,
                               int in_step, int Hf, int Wf, int sub_s, int rows_subsampled, int n_para, int Cp, int Cl, const float* mean_in,
                               const float* mean_tgt, const float* std_in, const float* std_tgt, void* stream) {
    RPB_REQUIRE(planar && flags && inp && tgt && mean_in && mean_tgt && std_in && std_tgt, "window_pack: null pointer");
    RPB_REQUIRE(Cp >= 1 && Cp <= 3 && Cl >= 0 && (Cl == 0) == (cl == nullptr), "window_pack: Cp=%d Cl=%d (planar channels 1..3; a channels-last block iff Cl > 0)", Cp, Cl);
    RPB_REQUIRE(B > 0 && horizon > in_step && in_step > 0 && Hf > 0 && Wf > 0 && sub_s >= 1 && n_para >= 0 && n_para <= 8,
                "window_pack: bad sizes (B=%d horizon=%d in_step=%d sub_s=%d n_para=%d)", B, horizon, in_step, sub_s, n_para);
    WindowPackArgs a;
    a.planar = planar; a.cl = cl; a.flags = flags; a.inp = inp; a.tgt = tgt; a.Cp = Cp; a.Cl = Cl;
    a.mean_in = mean_in; a.mean_tgt = mean_tgt; a.std_in = std_in; a.std_tgt = std_tgt;
    a.horizon = horizon; a.in_step = in_step; a.Hf = Hf; a.Wf = Wf; a.sub_s = sub_s; a.n_para = n_para;
    a.sub_h = rows_subsampled ? 1 : sub_s;                               // Hf = rows present in the staged slab
    a.H = rows_subsampled ? Hf : (Hf + sub_s - 1) / sub_s;               // len(range(0, Hf, sub_s)) = numpy's [::sub_s]
    a.W = (Wf + sub_s - 1) / sub_s;
    a.nflag = 4 + (n_para > 1 ? n_para : 1);
    a.ntok = (long)B * horizon * a.H * a.W;
    long grid = (a.ntok + 255) / 256;
    const long cap = (long)rpb_num_cus() * 16;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(window_pack_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    RPB_CHECK_LAUNCH("window_pack");
}

// Combustion surrogate samples (realpdebench/data/combustion_surrogate_hf_dataset.py:213-243): input = the channels-last
// `numerical` window + one constant channel per number parsed from sim_id (gas ratio, equivalence ratio), target = the `real`
// window with a singleton channel; both normalised ((x - mean) / std; a RangeNormalizer passes mean = 0, std = max) in the same pass.
struct PairPackArgs {
    const float* num;      // [B][ntok][Cl]
    const float* real;     // [B][ntok]
    const float* para;     // [B][n_para]
    float* inp;            // [B][ntok][Cl + n_para]
    float* tgt;            // [B][ntok][1]
    const float* mean_in;  // [Cl + n_para]
    const float* std_in;
    const float* mean_tgt; // [1]
    const float* std_tgt;
    long ntok, total;      // T * H * W, B * ntok
    int Cl, n_para;
};

__global__ __launch_bounds__(256) void pair_pack_kernel(PairPackArgs a) {
    const int C = a.Cl + a.n_para;
    // one thread per input element: consecutive lanes write consecutive floats of `inp`
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.total * C; i += (long)gridDim.x * blockDim.x) {
        const long tok = i / C;
        const int c = (int)(i - tok * C);
        const float v = c < a.Cl ? a.num[tok * a.Cl + c] : a.para[(tok / a.ntok) * a.n_para + (c - a.Cl)];
        a.inp[i] = (v - a.mean_in[c]) / a.std_in[c];
        if (c == 0) a.tgt[tok] = (a.real[tok] - a.mean_tgt[0]) / a.std_tgt[0];
    }
}

extern "C" int rpb_pair_pack(const float* num, const float* real, const float* para, float* inp, float* tgt, int B, long ntok, int Cl,
                             int n_para, const float* mean_in, const float* mean_tgt, const float* std_in, const float* std_tgt,
                             void* stream) {
    RPB_REQUIRE(num && real && inp && tgt && mean_in && mean_tgt && std_in && std_tgt, "pair_pack: null pointer");
    RPB_REQUIRE(B > 0 && ntok > 0 && Cl >= 1 && n_para >= 0 && n_para <= 8 && (n_para == 0 || para), "pair_pack: bad sizes (B=%d Cl=%d n_para=%d)",
                B, Cl, n_para);
    PairPackArgs a;
    a.num = num; a.real = real; a.para = para; a.inp = inp; a.tgt = tgt; a.mean_in = mean_in; a.std_in = std_in;
    a.mean_tgt = mean_tgt; a.std_tgt = std_tgt; a.ntok = ntok; a.total = (long)B * ntok; a.Cl = Cl; a.n_para = n_para;
    long grid = (a.total * (Cl + n_para) + 255) / 256;
    const long cap = (long)rpb_num_cus() * 16;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(pair_pack_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    RPB_CHECK_LAUNCH("pair_pack");
}
