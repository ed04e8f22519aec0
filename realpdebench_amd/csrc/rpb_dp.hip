// Data-parallel gradient exchange behind the C ABI: RCCL all-reduce over xGMI on a SIDE HIP stream, ordered against the
// compute stream with events -- SURVEY.md section 8(b)/(e).  The reference has no distributed path at all (single process,
// realpdebench/train.py:63); this is the collective a maintainer binds next to the kernels:
//
//   rank 0: rpb_dp_unique_id(id)            -> 128 opaque bytes, handed to every rank by whatever launcher is in use
//   all   : rpb_dp_allreduce_init(id, rank, world, &h)       (communicator on the CURRENT device, side stream, events)
//   per bucket, from inside the backward pass:
//           rpb_dp_allreduce_enqueue(h, buf, count, dtype, producer_stream)
//              the side stream waits for everything enqueued on producer_stream so far, then sums `buf` in place across
//              ranks; the call returns immediately and the producer stream keeps running backward kernels meanwhile
//   before the optimizer:
//           rpb_dp_allreduce_wait(h, consumer_stream)        (consumer_stream waits for every bucket enqueued so far)
//   small synchronous reductions (SyncBN statistics): rpb_dp_allreduce_inline(h, buf, count, dtype, stream)
//
// RCCL is resolved at run time (dlopen of the librccl already loaded into the process -- PyTorch-ROCm brings one -- else the
// system one), so librpb_hip.so itself loads on hosts without RCCL and two RCCL copies never coexist in one process.
#include "rpb_common.h"
#include <stdlib.h>
#include <dlfcn.h>

namespace {
typedef struct {
    char internal[128];
} nccl_uid_t;
typedef void* nccl_comm_t;
typedef int (*fn_get_uid)(nccl_uid_t*);
typedef int (*fn_init_rank)(nccl_comm_t*, int, nccl_uid_t, int);
typedef int (*fn_destroy)(nccl_comm_t);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t);
typedef int (*fn_reducescatter)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t);
typedef int (*fn_allgather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t);
typedef const char* (*fn_errstr)(int);

struct Rccl {
    void* lib = nullptr;
    fn_get_uid get_uid = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_destroy destroy = nullptr;
    fn_destroy abort = nullptr;
    fn_allreduce allreduce = nullptr;
    fn_reducescatter reducescatter = nullptr;
    fn_allgather allgather = nullptr;
    fn_errstr errstr = nullptr;
};

Rccl* rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names)                      // the copy already mapped into the process, if any
            if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char* n : names)
            if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) r.lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) {
            r.get_uid = (fn_get_uid)dlsym(r.lib, "ncclGetUniqueId");
            r.init_rank = (fn_init_rank)dlsym(r.lib, "ncclCommInitRank");
            r.destroy = (fn_destroy)dlsym(r.lib, "ncclCommDestroy");
            r.abort = (fn_destroy)dlsym(r.lib, "ncclCommAbort");
            r.allreduce = (fn_allreduce)dlsym(r.lib, "ncclAllReduce");
            r.reducescatter = (fn_reducescatter)dlsym(r.lib, "ncclReduceScatter");
            r.allgather = (fn_allgather)dlsym(r.lib, "ncclAllGather");
            r.errstr = (fn_errstr)dlsym(r.lib, "ncclGetErrorString");
        }
    }
    return (r.lib && r.get_uid && r.init_rank && r.destroy && r.allreduce) ? &r : nullptr;
}

#define DP_MAX_MARKS 16
#define DP_MAX_TIMED 96        /* a step of the cylinder FNO: 4 x 7 chunks of 16 MB + the small buckets; 7-8 inline reductions */
struct DpHandle {
    nccl_comm_t comm;
    hipStream_t side;
    hipEvent_t ready, done;
    int rank, world;
    long enqueued;
    // optional instrumentation (rpb_dp_set_timing): per-bucket start / end events on the side stream, the moment the consumer
    // stream reaches its wait, and start / end of every inline reduction since the last rpb_dp_allreduce_wait
    int timing, events_made;
    int nb, ni;
    hipEvent_t b0[DP_MAX_TIMED], b1[DP_MAX_TIMED], i0[DP_MAX_TIMED], i1[DP_MAX_TIMED], cwait, first;
    long bbytes[DP_MAX_TIMED];
    int have_wait;
    hipEvent_t marks[DP_MAX_MARKS];                      // rpb_dp_mark / rpb_dp_wait_mark: points of the side stream other streams can wait for
    int marks_made;
    // instrumentation of the marks (sharded step: the parameter all-gathers the NEXT forward waits for, per bucket): the moment the side
    // stream passed mark idx (mk1) and the moment a consumer stream reached its wait for it (mw0); nb_at_wait = buckets recorded when the
    // consumer's rpb_dp_allreduce_wait was issued (the gathers of a sharded step are recorded after it)
    hipEvent_t mk1[DP_MAX_MARKS], mw0[DP_MAX_MARKS];
    int mk_set[DP_MAX_MARKS], mw_set[DP_MAX_MARKS], nb_at_wait;
    // modelled transfers (one-GPU proxy of an N-rank run, rpb_dp_set_model): after every collective the side stream idles for the time
    // the operation would take over the interconnect of `model_world` ranks at `model_gbps` per direction and rank
    int model_world;
    float model_gbps, model_lat_us;
};

const int kNcclSum = 0, kNcclF32 = 7, kNcclF64 = 8;
}  // namespace

#define RPB_NCCL(call, what)                                                                                  \
    do {                                                                                                      \
        int rc_ = (call);                                                                                     \
        if (rc_ != 0) RPB_FAIL(RPB_ERR_LAUNCH, "%s: RCCL error %d (%s)", what, rc_, R->errstr ? R->errstr(rc_) : "?"); \
    } while (0)
#define RPB_HIP(call, what)                                                                        \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) RPB_FAIL(RPB_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e_));     \
    } while (0)

// the side stream idles for `us` microseconds (s_memrealtime: the constant 100 MHz counter) -- the modelled duration of a transfer
__global__ void dp_delay_kernel(long ticks) {
    const unsigned long long s0 = wall_clock64();
    while ((long)(wall_clock64() - s0) < ticks) __builtin_amdgcn_s_sleep(32);
}
// modelled time of a ring reduce-scatter or all-gather of `bytes` (the whole buffer) over `world` ranks: (world - 1) / world of it crosses
// each link at `gbps`; an all-reduce is the two in sequence
static void dp_model_delay(DpHandle* h, double bytes, int phases, hipStream_t st) {
    if (h->model_world <= 1 || h->model_gbps <= 0.f) return;
    const double us = phases * (bytes * (h->model_world - 1) / h->model_world / (h->model_gbps * 1e3) + h->model_lat_us);
    hipLaunchKernelGGL(dp_delay_kernel, dim3(1), dim3(1), 0, st, (long)(us * 100.0));
}

extern "C" int rpb_dp_available(void) { return rccl() != nullptr; }

extern "C" int rpb_dp_unique_id(void* id128) {
    RPB_REQUIRE(id128, "dp_unique_id: null pointer");
    Rccl* R = rccl();
    if (!R) RPB_FAIL(RPB_ERR_UNSUPPORTED, "dp: librccl.so could not be loaded");
    nccl_uid_t id;
    RPB_NCCL(R->get_uid(&id), "ncclGetUniqueId");
    memcpy(id128, &id, sizeof(id));
    return RPB_OK;
}

extern "C" int rpb_dp_allreduce_init(const void* id128, int rank, int world, void** handle) {
    RPB_REQUIRE(id128 && handle && world >= 1 && rank >= 0 && rank < world, "dp_allreduce_init: bad arguments (rank %d of %d)", rank, world);
    Rccl* R = rccl();
    if (!R) RPB_FAIL(RPB_ERR_UNSUPPORTED, "dp: librccl.so could not be loaded");
    DpHandle* h = new DpHandle();
    h->rank = rank;
    h->world = world;
    h->enqueued = 0;
    h->timing = h->events_made = 0;
    h->nb = h->ni = h->have_wait = h->nb_at_wait = 0;
    h->marks_made = 0;
    for (int i = 0; i < DP_MAX_MARKS; ++i) h->mk_set[i] = h->mw_set[i] = 0;
    h->model_world = 0;
    h->model_gbps = h->model_lat_us = 0.f;
    nccl_uid_t id;
    memcpy(&id, id128, sizeof(id));
    int rc = R->init_rank(&h->comm, world, id, rank);
    if (rc != 0) {
        delete h;
        RPB_FAIL(RPB_ERR_LAUNCH, "ncclCommInitRank: RCCL error %d (%s)", rc, R->errstr ? R->errstr(rc) : "?");
    }
    {
        // RPB_DP_SIDE_PRIORITY=1: the side stream gets the highest stream priority (when a CU slot frees up between two compute kernels the
        // collective's kernel takes it first).  OFF by default -- measured (round 6): in a fresh process it changes
        // nothing (B = 4 step with the DP path 5.98 vs 6.09 ms), but inside bench.py, after the single-GPU sections have run in the same
        // process, every DP step costs +4.2 ms with it (10.5 vs 6.1 ms at B = 4): a high-priority queue next to the queues the process
        // already owns is not free on this runtime.
        int pr_least = 0, pr_greatest = 0;
        const char* e = getenv("RPB_DP_SIDE_PRIORITY");
        if (!(e && atoi(e) == 1) || hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest) != hipSuccess) pr_greatest = 0;
        RPB_HIP(hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, pr_greatest), "dp side stream");
    }
    RPB_HIP(hipEventCreateWithFlags(&h->ready, hipEventDisableTiming), "dp event");
    RPB_HIP(hipEventCreateWithFlags(&h->done, hipEventDisableTiming), "dp event");
    *handle = h;
    return RPB_OK;
}

static int dp_dtype(int dtype) { return dtype == 0 ? kNcclF32 : (dtype == 1 ? kNcclF64 : -1); }

extern "C" int rpb_dp_allreduce_enqueue(void* handle, void* buf, long count, int dtype, void* producer_stream) {
    RPB_REQUIRE(handle && buf && count > 0 && dp_dtype(dtype) >= 0, "dp_allreduce_enqueue: bad arguments");
    Rccl* R = rccl();
    DpHandle* h = (DpHandle*)handle;
    if (h->timing && h->have_wait) {                     // first bucket of a new step: forget the previous step's records
        h->nb = h->ni = 0;
        h->have_wait = 0;
    }
    RPB_HIP(hipEventRecord(h->ready, (hipStream_t)producer_stream), "dp record");
    RPB_HIP(hipStreamWaitEvent(h->side, h->ready, 0), "dp wait");
    const bool timed = h->timing && h->nb < DP_MAX_TIMED;
    if (timed) {
        if (h->nb == 0) RPB_HIP(hipEventRecord(h->first, (hipStream_t)producer_stream), "dp record");
        RPB_HIP(hipEventRecord(h->b0[h->nb], h->side), "dp record");
    }
    RPB_NCCL(R->allreduce(buf, buf, (size_t)count, dp_dtype(dtype), kNcclSum, h->comm, h->side), "ncclAllReduce");
    dp_model_delay(h, (double)count * (dtype == 0 ? 4 : 8), 2, h->side);
    if (timed) {
        RPB_HIP(hipEventRecord(h->b1[h->nb], h->side), "dp record");
        h->bbytes[h->nb] = count * (dtype == 0 ? 4 : 8);
        h->nb++;
    }
    h->enqueued++;
    return RPB_OK;
}

extern "C" int rpb_dp_allreduce_wait(void* handle, void* consumer_stream) {
    RPB_REQUIRE(handle, "dp_allreduce_wait: null handle");
    DpHandle* h = (DpHandle*)handle;
    if (h->timing) {
        RPB_HIP(hipEventRecord(h->cwait, (hipStream_t)consumer_stream), "dp record");
        h->have_wait = 1;
        h->nb_at_wait = h->nb;
    }
    RPB_HIP(hipEventRecord(h->done, h->side), "dp record");
    RPB_HIP(hipStreamWaitEvent((hipStream_t)consumer_stream, h->done, 0), "dp wait");
    return RPB_OK;
}

extern "C" int rpb_dp_allreduce_inline(void* handle, void* buf, long count, int dtype, void* stream) {
    RPB_REQUIRE(handle && buf && count > 0 && dp_dtype(dtype) >= 0, "dp_allreduce_inline: bad arguments");
    Rccl* R = rccl();
    DpHandle* h = (DpHandle*)handle;
    const bool timed = h->timing && h->ni < DP_MAX_TIMED;
    if (timed) RPB_HIP(hipEventRecord(h->i0[h->ni], (hipStream_t)stream), "dp record");
    RPB_NCCL(R->allreduce(buf, buf, (size_t)count, dp_dtype(dtype), kNcclSum, h->comm, (hipStream_t)stream), "ncclAllReduce");
    dp_model_delay(h, (double)count * (dtype == 0 ? 4 : 8), 2, (hipStream_t)stream);
    if (timed) {
        RPB_HIP(hipEventRecord(h->i1[h->ni], (hipStream_t)stream), "dp record");
        h->ni++;
    }
    return RPB_OK;
}

// Sharded optimizer step (ZeRO-1 shape): the gradient chunk `buf` [count] is reduce-scattered IN PLACE -- rank r ends up with the sum
// of its piece buf[r * count / world, (r + 1) * count / world) -- on the side stream, after everything enqueued on producer_stream so far.
// count must be a multiple of world.  The matching rpb_dp_allgather_enqueue sends every rank's piece of the PARAMETER chunk to all ranks,
// in place.  Same bytes on the wire as the all-reduce of the chunk; Adam runs on 1 / world of the arena in between.
static int dp_side_op(void* handle, void* buf, long count, int dtype, void* producer_stream, int op) {
    RPB_REQUIRE(handle && buf && count > 0 && dp_dtype(dtype) >= 0, "dp_%s_enqueue: bad arguments", op ? "allgather" : "reduce_scatter");
    Rccl* R = rccl();
    DpHandle* h = (DpHandle*)handle;
    RPB_REQUIRE(R && R->reducescatter && R->allgather, "dp: this librccl.so has no ncclReduceScatter / ncclAllGather");
    RPB_REQUIRE(count % h->world == 0, "dp_%s_enqueue: count %ld is not a multiple of the %d ranks", op ? "allgather" : "reduce_scatter", count, h->world);
    if (h->timing && h->have_wait && op == 0) {          // first chunk of a new step: forget the previous step's records
        h->nb = h->ni = 0;
        h->have_wait = 0;
    }
    RPB_HIP(hipEventRecord(h->ready, (hipStream_t)producer_stream), "dp record");
    RPB_HIP(hipStreamWaitEvent(h->side, h->ready, 0), "dp wait");
    const bool timed = h->timing && h->nb < DP_MAX_TIMED;
    if (timed) {
        if (h->nb == 0) RPB_HIP(hipEventRecord(h->first, (hipStream_t)producer_stream), "dp record");
        RPB_HIP(hipEventRecord(h->b0[h->nb], h->side), "dp record");
    }
    const long piece = count / h->world;
    const size_t esz = dtype == 0 ? 4 : 8;
    char* mine = (char*)buf + (size_t)h->rank * piece * esz;
    if (op == 0) RPB_NCCL(R->reducescatter(buf, mine, (size_t)piece, dp_dtype(dtype), kNcclSum, h->comm, h->side), "ncclReduceScatter");
    else RPB_NCCL(R->allgather(mine, buf, (size_t)piece, dp_dtype(dtype), h->comm, h->side), "ncclAllGather");
    dp_model_delay(h, (double)count * esz, 1, h->side);
    if (timed) {
        RPB_HIP(hipEventRecord(h->b1[h->nb], h->side), "dp record");
        h->bbytes[h->nb] = count * (long)esz;
        h->nb++;
    }
    h->enqueued++;
    return RPB_OK;
}
extern "C" int rpb_dp_reduce_scatter_enqueue(void* handle, void* buf, long count, int dtype, void* producer_stream) {
    return dp_side_op(handle, buf, count, dtype, producer_stream, 0);
}
extern "C" int rpb_dp_allgather_enqueue(void* handle, void* buf, long count, int dtype, void* producer_stream) {
    return dp_side_op(handle, buf, count, dtype, producer_stream, 1);
}

// rpb_dp_mark(h, idx): event idx (0 .. 15) is recorded on the side stream behind everything enqueued so far;
// rpb_dp_wait_mark(h, idx, stream): `stream` waits for it -- the next forward pass waits per layer for the all-gather of that layer's weights
extern "C" int rpb_dp_mark(void* handle, int idx) {
    RPB_REQUIRE(handle && idx >= 0 && idx < DP_MAX_MARKS, "dp_mark: bad arguments");
    DpHandle* h = (DpHandle*)handle;
    if (!h->marks_made) {
        for (int i = 0; i < DP_MAX_MARKS; ++i) RPB_HIP(hipEventCreateWithFlags(&h->marks[i], hipEventDisableTiming), "dp event");
        h->marks_made = 1;
    }
    RPB_HIP(hipEventRecord(h->marks[idx], h->side), "dp record");
    if (h->timing && h->events_made && !h->mk_set[idx]) {   // one-shot per rpb_dp_set_timing(1): the first mark and the first wait for it
        RPB_HIP(hipEventRecord(h->mk1[idx], h->side), "dp record");
        h->mk_set[idx] = 1;
    }
    return RPB_OK;
}
extern "C" int rpb_dp_wait_mark(void* handle, int idx, void* stream) {
    RPB_REQUIRE(handle && idx >= 0 && idx < DP_MAX_MARKS, "dp_wait_mark: bad arguments");
    DpHandle* h = (DpHandle*)handle;
    if (!h->marks_made) return RPB_OK;                   // nothing was ever marked
    if (h->timing && h->events_made && h->mk_set[idx] && !h->mw_set[idx]) {
        RPB_HIP(hipEventRecord(h->mw0[idx], (hipStream_t)stream), "dp record");
        h->mw_set[idx] = 1;
    }
    RPB_HIP(hipStreamWaitEvent((hipStream_t)stream, h->marks[idx], 0), "dp wait");
    return RPB_OK;
}

// One-GPU proxy of an N-rank run: model_world > 1 makes every collective of this handle idle its stream for the modelled transfer time
// (ring schedule, gbps per direction and rank, lat_us per phase); 0 turns the model off.  The collectives themselves still run
// (no-ops on a one-rank communicator).
extern "C" int rpb_dp_set_model(void* handle, int model_world, float gbps, float lat_us) {
    RPB_REQUIRE(handle && model_world >= 0 && gbps >= 0.f && lat_us >= 0.f, "dp_set_model: bad arguments");
    DpHandle* h = (DpHandle*)handle;
    h->model_world = model_world;
    h->model_gbps = gbps;
    h->model_lat_us = lat_us;
    return RPB_OK;
}

// Instrumentation for the N > 1 bench line.  on = 1: every bucket / inline reduction from now on is bracketed by timing events.
extern "C" int rpb_dp_set_timing(void* handle, int on) {
    RPB_REQUIRE(handle, "dp_set_timing: null handle");
    DpHandle* h = (DpHandle*)handle;
    if (on && !h->events_made) {
        for (int i = 0; i < DP_MAX_TIMED; ++i) {
            RPB_HIP(hipEventCreate(&h->b0[i]), "dp event");
            RPB_HIP(hipEventCreate(&h->b1[i]), "dp event");
            RPB_HIP(hipEventCreate(&h->i0[i]), "dp event");
            RPB_HIP(hipEventCreate(&h->i1[i]), "dp event");
        }
        for (int i = 0; i < DP_MAX_MARKS; ++i) {
            RPB_HIP(hipEventCreate(&h->mk1[i]), "dp event");
            RPB_HIP(hipEventCreate(&h->mw0[i]), "dp event");
        }
        RPB_HIP(hipEventCreate(&h->cwait), "dp event");
        RPB_HIP(hipEventCreate(&h->first), "dp event");
        h->events_made = 1;
    }
    h->timing = on ? 1 : 0;
    h->nb = h->ni = h->have_wait = h->nb_at_wait = 0;    // (re)start the records
    for (int i = 0; i < DP_MAX_MARKS; ++i) h->mk_set[i] = h->mw_set[i] = 0;
    return RPB_OK;
}

// Times of the step that ended with the last rpb_dp_allreduce_wait (call after synchronising the device).  out[0] = number of
// buckets nb, out[1] = number of inline reductions ni, out[2] = exposed milliseconds (how long after the consumer stream reached its
// wait the last bucket finished; 0 when the reduction was fully hidden), out[3] = milliseconds from the first bucket's
// announcement to the end of the last bucket, then nb triples (start since the first announcement, duration, bytes) and ni inline
// durations.  The exposed time counts the buckets enqueued BEFORE the wait (a sharded step records its parameter all-gathers after it:
// those end after the Adam launch by construction and are not "exposure" of the wait).  When there is room, one more float follows the
// inline durations: the sum over the marks of how long a consumer stream that reached rpb_dp_wait_mark(idx) had to wait for the side stream
// to pass the mark (the sharded step's exposed all-gather time in the NEXT forward pass: the first mark / first wait of every index since
// rpb_dp_set_timing(1) -- turn timing on, run two steps, synchronise, read).
// Returns the number of floats written, or a negative status.
extern "C" int rpb_dp_step_times(void* handle, float* out, int max_out) {
    RPB_REQUIRE(handle && out && max_out >= 4, "dp_step_times: bad arguments");
    DpHandle* h = (DpHandle*)handle;
    const int nb = h->nb, ni = h->ni;
    RPB_REQUIRE(4 + 3 * nb + ni <= max_out, "dp_step_times: need %d floats", 4 + 3 * nb + ni);
    out[0] = (float)nb;
    out[1] = (float)ni;
    out[2] = out[3] = 0.f;
    float ms = 0.f;
    const int nw = h->nb_at_wait > 0 && h->nb_at_wait <= nb ? h->nb_at_wait : nb;
    if (nb > 0 && h->have_wait) {
        if (hipEventElapsedTime(&ms, h->cwait, h->b1[nw - 1]) == hipSuccess) out[2] = ms > 0.f ? ms : 0.f;
        if (hipEventElapsedTime(&ms, h->first, h->b1[nw - 1]) == hipSuccess) out[3] = ms;
    }
    for (int i = 0; i < nb; ++i) {
        out[4 + 3 * i] = hipEventElapsedTime(&ms, h->first, h->b0[i]) == hipSuccess ? ms : -1.f;
        out[5 + 3 * i] = hipEventElapsedTime(&ms, h->b0[i], h->b1[i]) == hipSuccess ? ms : -1.f;
        out[6 + 3 * i] = (float)h->bbytes[i];
    }
    for (int i = 0; i < ni; ++i) out[4 + 3 * nb + i] = hipEventElapsedTime(&ms, h->i0[i], h->i1[i]) == hipSuccess ? ms : -1.f;
    if (4 + 3 * nb + ni + 1 <= max_out) {
        float gsum = 0.f;
        if (h->events_made)
            for (int i = 0; i < DP_MAX_MARKS; ++i)
                if (h->mk_set[i] && h->mw_set[i] && hipEventElapsedTime(&ms, h->mw0[i], h->mk1[i]) == hipSuccess && ms > 0.f) gsum += ms;
        out[4 + 3 * nb + ni] = gsum;
        return 4 + 3 * nb + ni + 1;
    }
    return 4 + 3 * nb + ni;
}

static int dp_teardown(void* handle, bool abort);
extern "C" int rpb_dp_allreduce_destroy(void* handle) { return dp_teardown(handle, false); }
// Process-exit teardown: ncclCommAbort instead of ncclCommDestroy and no wait for the side stream -- destroy blocks until outstanding
// collectives finish, which never happens once a peer rank has died.
extern "C" int rpb_dp_allreduce_abort(void* handle) { return dp_teardown(handle, true); }

static int dp_teardown(void* handle, bool abort) {
    if (!handle) return RPB_OK;
    Rccl* R = rccl();
    DpHandle* h = (DpHandle*)handle;
    if (abort && R && R->abort) {
        (void)R->abort(h->comm);
    } else {
        (void)hipStreamSynchronize(h->side);
        if (R) (void)R->destroy(h->comm);
    }
    (void)hipEventDestroy(h->ready);
    (void)hipEventDestroy(h->done);
    if (h->marks_made)
        for (int i = 0; i < DP_MAX_MARKS; ++i) (void)hipEventDestroy(h->marks[i]);
    if (h->events_made) {
        for (int i = 0; i < DP_MAX_TIMED; ++i) {
            (void)hipEventDestroy(h->b0[i]);
            (void)hipEventDestroy(h->b1[i]);
            (void)hipEventDestroy(h->i0[i]);
            (void)hipEventDestroy(h->i1[i]);
        }
        for (int i = 0; i < DP_MAX_MARKS; ++i) {
            (void)hipEventDestroy(h->mk1[i]);
            (void)hipEventDestroy(h->mw0[i]);
        }
        (void)hipEventDestroy(h->cwait);
        (void)hipEventDestroy(h->first);
    }
    (void)hipStreamDestroy(h->side);
    delete h;
    return RPB_OK;
}
