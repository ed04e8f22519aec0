// Data-parallel optimizer step over PEER POINTERS (opt-in; SURVEY.md section 5.8 / 8e: direct reduce-scatter + all-gather over the fully
// connected xGMI mesh, 0.66 ms for the cylinder FNO's 403 MB against 4.6 ms for one ring): no collective library in the data path.
//
// Every rank maps the gradient arena, the parameter arena and a small flag block of every other rank (IPC handles exchanged by the
// host; this file sees plain device pointers).  Rank r owns the contiguous slice [r S, (r + 1) S) of the arena (S a multiple of 4):
//
//   backward done            rpb_dp_p2p_signal(GRAD_READY, k)          one wave: flag[q][GRAD_READY][r] = k for every peer q   (after the
//                                                                       kernel boundary that wrote this rank's gradients back to memory)
//   rpb_dp_p2p_adam          one wave waits until flag[r][GRAD_READY][q] >= k for every q, then the update kernel:
//                              g = ((g_0 + g_1) + g_2) + ...  read from the W gradient arenas (W - 1 of them over xGMI) -- the reduce-scatter
//                              p, m, v of the OWNED slice updated (torch.optim.Adam arithmetic, rpb_adam_step's)
//                              p stored into all W parameter arenas (W - 1 over xGMI)                                  -- the all-gather
//                            then rpb_dp_p2p_signal(PARAM_DONE, k)
//   next forward             rpb_dp_p2p_wait(PARAM_DONE, k): every slice of MY parameter arena has been written, and every peer has finished
//                            READING my gradient arena (its update kernel ended before its signal) -- so the next backward may overwrite it.
//
// Flags are monotonic step numbers (never reset), one 8-byte word per (kind, source rank) in the destination's own memory, written with
// system-scope release stores and polled with system-scope acquire loads; a poll gives up after `timeout_ms` and raises the handle's
// status word (a caller-owned device int the host reads back) instead of hanging the device.  The sum order is fixed (rank 0 first) and independent of the
// rank that evaluates it: every element is computed by exactly one rank, so all ranks hold bit-identical parameters; for two ranks the
// step equals the all-reduce step bit for bit (a + b is commutative), for more the usual fp32 reordering differences apply.
// Not here: gradient-norm clipping (the norm needs the reduced gradient before the update: two more flag rounds), overlap with backward
// (the exchange starts when backward ends; it is short, not hidden).  Unmeasured across devices: no multi-GPU box has run it; two
// processes sharing one GPU exercise the protocol (tests/test_gpu_p2p.py).
#include "rpb_common.h"
#include <math.h>

#define P2P_MAX_WORLD 16
#define P2P_GRAD_READY 0
#define P2P_PARAM_DONE 1

namespace {
struct P2pPtrs {
    float* grad[P2P_MAX_WORLD];
    float* param[P2P_MAX_WORLD];
    unsigned long long* flags[P2P_MAX_WORLD];       // [2 kinds][P2P_MAX_WORLD sources]
};
struct P2pHandle {
    P2pPtrs p;
    int rank, world;
    long total, slice;                              // arena elements, elements per rank (multiple of 4)
    int* status;                                    // device word: 0 ok, 1 + q = timed out waiting for rank q
    long timeout_ticks;                             // 100 MHz ticks
};

__global__ void p2p_signal_kernel(P2pPtrs p, int kind, int rank, int world, unsigned long long value) {
    const int q = threadIdx.x;
    if (q < world)
        __hip_atomic_store(p.flags[q] + kind * P2P_MAX_WORLD + rank, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void p2p_wait_kernel(const unsigned long long* mine, int kind, int world, unsigned long long value, long timeout_ticks,
                                int* status) {
    const int q = threadIdx.x;
    if (q >= world) return;
    const unsigned long long* f = mine + kind * P2P_MAX_WORLD + q;
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < value) {
        __builtin_amdgcn_s_sleep(64);
        if ((long)(wall_clock64() - t0) > timeout_ticks) {
            atomicCAS(status, 0, 1 + q);
            return;
        }
    }
}

__device__ __forceinline__ void p2p_adam_update(float& p, float g, float& m, float& v, float gscale, float b1, float b2, float eps,
                                                float step_size, float inv_sqrt_bc2) {
#pragma clang fp contract(off)
    const float gk = g * gscale;                    // (the arithmetic of adam_update in rpb_pointwise.hip, statement for statement)
    m = b1 * m + (1.f - b1) * gk;
    v = b2 * v + (1.f - b2) * gk * gk;
    const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
    p -= step_size * (m / denom);
}

__global__ __launch_bounds__(256) void p2p_adam_kernel(P2pPtrs pp, float* __restrict__ m, float* __restrict__ v, int rank, int world,
                                                       long e0_4, long n4, float gscale, float b1, float b2, float eps, float step_size,
                                                       float inv_sqrt_bc2, const int* __restrict__ status) {
    if (*status != 0) return;                       // a peer never announced its gradients: leave the parameters alone
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long e4 = e0_4 + i;
        f32x4 g = reinterpret_cast<const f32x4*>(pp.grad[0])[e4];
        for (int q = 1; q < world; ++q) {
#pragma clang fp contract(off)
            g += reinterpret_cast<const f32x4*>(pp.grad[q])[e4];
        }
        f32x4 pv = reinterpret_cast<const f32x4*>(pp.param[rank])[e4];
        f32x4 mv = reinterpret_cast<f32x4*>(m)[e4];
        f32x4 vv = reinterpret_cast<f32x4*>(v)[e4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pk = pv[k], mk = mv[k], vk = vv[k];
            p2p_adam_update(pk, g[k], mk, vk, gscale, b1, b2, eps, step_size, inv_sqrt_bc2);
            pv[k] = pk, mv[k] = mk, vv[k] = vk;
        }
        reinterpret_cast<f32x4*>(m)[e4] = mv;
        reinterpret_cast<f32x4*>(v)[e4] = vv;
        for (int q = 0; q < world; ++q) reinterpret_cast<f32x4*>(pp.param[q])[e4] = pv;
    }
}
}  // namespace

// grads / params / flags: `world` device pointers each, valid in THIS process (entry `rank` is this rank's own allocation, the others
// are the peers' allocations opened from their IPC handles); every flag block holds 2 * 16 zero-initialised 8-byte words; `status` is one
// zero-initialised int of this rank.  total: elements of the arena (a multiple of 4).
extern "C" int rpb_dp_p2p_init(int rank, int world, void* const* grads, void* const* params, void* const* flags, void* status,
                               long total, int timeout_ms, void** handle) {
    RPB_REQUIRE(handle && grads && params && flags && status && world >= 1 && world <= P2P_MAX_WORLD && rank >= 0 && rank < world,
                "dp_p2p_init: bad arguments (rank %d of %d, at most %d ranks)", rank, world, P2P_MAX_WORLD);
    RPB_REQUIRE(total > 0 && total % 4 == 0 && timeout_ms > 0, "dp_p2p_init: total=%ld must be a positive multiple of 4", total);
    for (int q = 0; q < world; ++q) RPB_REQUIRE(grads[q] && params[q] && flags[q], "dp_p2p_init: null pointer for rank %d", q);
    P2pHandle* h = new P2pHandle();
    for (int q = 0; q < world; ++q) {
        h->p.grad[q] = (float*)grads[q];
        h->p.param[q] = (float*)params[q];
        h->p.flags[q] = (unsigned long long*)flags[q];
    }
    h->rank = rank;
    h->world = world;
    h->total = total;
    h->slice = ((total / 4 + world - 1) / world) * 4;
    h->status = (int*)status;
    h->timeout_ticks = (long)timeout_ms * 100000L;
    *handle = h;
    return RPB_OK;
}

// [first, first + count) of the arena this rank updates (count may be 0 for the last ranks of a tiny arena)
extern "C" int rpb_dp_p2p_slice(void* handle, long* first, long* count) {
    RPB_REQUIRE(handle && first && count, "dp_p2p_slice: null pointer");
    const P2pHandle* h = (const P2pHandle*)handle;
    const long a = (long)h->rank * h->slice;
    *first = a < h->total ? a : h->total;
    *count = a < h->total ? (a + h->slice <= h->total ? h->slice : h->total - a) : 0;
    return RPB_OK;
}

extern "C" int rpb_dp_p2p_signal(void* handle, int kind, long step, void* stream) {
    RPB_REQUIRE(handle && (kind == P2P_GRAD_READY || kind == P2P_PARAM_DONE) && step >= 1, "dp_p2p_signal: bad arguments");
    const P2pHandle* h = (const P2pHandle*)handle;
    hipLaunchKernelGGL(p2p_signal_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, h->p, kind, h->rank, h->world, (unsigned long long)step);
    RPB_CHECK_LAUNCH("dp_p2p_signal");
}

extern "C" int rpb_dp_p2p_wait(void* handle, int kind, long step, void* stream) {
    RPB_REQUIRE(handle && (kind == P2P_GRAD_READY || kind == P2P_PARAM_DONE) && step >= 1, "dp_p2p_wait: bad arguments");
    const P2pHandle* h = (const P2pHandle*)handle;
    hipLaunchKernelGGL(p2p_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, h->p.flags[h->rank], kind, h->world,
                       (unsigned long long)step, h->timeout_ticks, h->status);
    RPB_CHECK_LAUNCH("dp_p2p_wait");
}

// The whole exchange + update of step `step` on `stream` (which ran this rank's backward pass): announce the gradients, wait for every
// peer's, reduce + Adam + broadcast of the owned slice, announce the parameters.  m, v: this rank's moment arenas (only the owned slice is
// touched).  gscale as in rpb_adam_step.
extern "C" int rpb_dp_p2p_adam(void* handle, float* m, float* v, float lr, float beta1, float beta2, float eps, long step, float gscale,
                               void* stream) {
    RPB_REQUIRE(handle && m && v && step >= 1, "dp_p2p_adam: bad arguments");
    const P2pHandle* h = (const P2pHandle*)handle;
    hipStream_t st = (hipStream_t)stream;
    int rc = rpb_dp_p2p_signal(handle, P2P_GRAD_READY, step, stream);
    if (rc != RPB_OK) return rc;
    rc = rpb_dp_p2p_wait(handle, P2P_GRAD_READY, step, stream);
    if (rc != RPB_OK) return rc;
    long first = 0, count = 0;
    (void)rpb_dp_p2p_slice(handle, &first, &count);
    if (count > 0) {
        const double bc1 = 1.0 - pow((double)beta1, (double)step);
        const double bc2 = 1.0 - pow((double)beta2, (double)step);
        const float step_size = (float)((double)lr / bc1);
        const float isb2 = (float)(1.0 / sqrt(bc2));
        long grid = (count / 4 + 255) / 256;
        const long cap = (long)rpb_num_cus() * 8;
        if (grid > cap) grid = cap;
        hipLaunchKernelGGL(p2p_adam_kernel, dim3((unsigned)grid), dim3(256), 0, st, h->p, m, v, h->rank, h->world, first / 4, count / 4,
                           gscale, beta1, beta2, eps, step_size, isb2, h->status);
        const hipError_t e_ = hipGetLastError();
        if (e_ != hipSuccess) RPB_FAIL(RPB_ERR_LAUNCH, "dp_p2p_adam: %s", hipGetErrorString(e_));
    }
    return rpb_dp_p2p_signal(handle, P2P_PARAM_DONE, step, stream);
}

extern "C" int rpb_dp_p2p_destroy(void* handle) {
    if (handle) delete (P2pHandle*)handle;
    return RPB_OK;
}
