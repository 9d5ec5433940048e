// comm.hip — the collective half of the C ABI: gradient all-reduce and the initial parameter broadcast of the data-parallel
// path, on RCCL over xGMI, one communicator per process (= per GPU).  The reference has no distributed code; this is the new
// component BASELINE configs 4-5 ask for.
//
// RCCL is bound at run time (dlopen of librccl.so.1 at mi_comm_init): a single-GPU process never loads it, and a process that already
// carries an RCCL (PyTorch's) shares that copy instead of mapping a second one.  The communicator owns one extra HIP stream: the `_async`
// all-reduce of a gradient bucket waits there for the producing kernels (event on the caller's stream) and runs under the next part of the
// backward pass; mi_comm_wait makes the caller's stream wait for every bucket before the optimiser step.  Nothing here synchronises the host.
// Rendezvous (who is rank 0, how the 128-byte id reaches the others) is the caller's business: mi_comm_unique_id on rank 0, any
// out-of-band channel (the Python side uses the torch.distributed store it already has), mi_comm_init everywhere.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include "mi_internal.hpp"
#include "mi355_carla.h"

namespace {

struct Rccl {
    void* lib;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)(void);
    ncclResult_t (*GroupEnd)(void);
    const char* (*GetErrorString)(ncclResult_t);
};

Rccl g_rccl;                                              // function table, filled once (idempotent; same values from any thread)

int load_rccl() {
    if (g_rccl.lib) return MI_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
    if (!lib) return mi_fail(MI_ERR_STATE, "mi_comm: librccl.so.1 not found");
    Rccl r = {};
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(lib, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(lib, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(lib, "ncclAllReduce");
    r.Broadcast = (decltype(r.Broadcast))dlsym(lib, "ncclBroadcast");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(lib, "ncclGetErrorString");
    r.ReduceScatter = (decltype(r.ReduceScatter))dlsym(lib, "ncclReduceScatter");      // optional: the two-phase schedule below is skipped without them
    r.AllGather = (decltype(r.AllGather))dlsym(lib, "ncclAllGather");
    r.GroupStart = (decltype(r.GroupStart))dlsym(lib, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(lib, "ncclGroupEnd");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.Broadcast || !r.GetErrorString)
        return mi_fail(MI_ERR_STATE, "mi_comm: librccl.so.1 lacks an entry point");
    r.lib = lib;
    g_rccl = r;
    return MI_OK;
}

struct MiComm {
    ncclComm_t comm;
    int rank, world;
    hipStream_t side;                                     // the all-reduce stream of the `_async` form
    hipEvent_t ready, done;
    int pending;                                          // buckets queued on `side` since the last mi_comm_wait
};

// Gradient-bucket schedule: 0 = ncclAllReduce (RCCL picks ring / tree / one-shot itself), 1 = reduce-scatter + all-gather: on the
// fully connected xGMI mesh of one node every rank owns 1/W of the bucket, receives the other ranks' pieces of ITS slice over the seven direct
// links in one hop, sums, and sends its finished slice back over the same links -- 2 (W-1)/W of the bucket per link direction, no multi-hop ring.
// MI355_COMM_ALGO=rsag selects it; both forms sit behind mi_allreduce_sum_f32 so callers never see the difference.  (Validated at world size 1
// only -- no multi-GPU box in this build environment -- hence off by default.)
int comm_algo() {
    static int algo = -1;
    if (algo < 0) { const char* e = getenv("MI355_COMM_ALGO"); algo = (e && strcmp(e, "rsag") == 0) ? 1 : 0; }
    return algo;
}

int rccl_fail(const char* what, ncclResult_t r) {
    static thread_local char msg[256];
    snprintf(msg, sizeof(msg), "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
    return mi_fail(MI_ERR_LAUNCH, msg);
}

// in-place sum of buf[0..n) over the ranks on stream `st`
int allreduce_on(MiComm* c, float* buf, long long n, hipStream_t st) {
    const long long chunk = n / c->world;
    if (comm_algo() == 1 && c->world > 1 && chunk >= 1024 && g_rccl.ReduceScatter && g_rccl.AllGather && g_rccl.GroupStart && g_rccl.GroupEnd) {
        // slice r of the first chunk * W elements belongs to rank r (both calls in RCCL's in-place form); the n % W tail rides along as a tiny all-reduce
        float* mine = buf + (long long)c->rank * chunk;
        ncclResult_t r = g_rccl.ReduceScatter(buf, mine, (size_t)chunk, ncclFloat, ncclSum, c->comm, st);
        if (r != ncclSuccess) return rccl_fail("ncclReduceScatter", r);
        r = g_rccl.AllGather(mine, buf, (size_t)chunk, ncclFloat, c->comm, st);
        if (r != ncclSuccess) return rccl_fail("ncclAllGather", r);
        const long long tail = n - chunk * c->world;
        if (tail > 0) {
            r = g_rccl.AllReduce(buf + chunk * c->world, buf + chunk * c->world, (size_t)tail, ncclFloat, ncclSum, c->comm, st);
            if (r != ncclSuccess) return rccl_fail("ncclAllReduce (tail)", r);
        }
        return MI_OK;
    }
    ncclResult_t r = g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat, ncclSum, c->comm, st);
    return r == ncclSuccess ? MI_OK : rccl_fail("ncclAllReduce", r);
}

}  // namespace

extern "C" {

int mi_comm_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

// binds RCCL (dlopen + every entry point) WITHOUT creating anything: every rank calls this before the collective mi_comm_init, so that a
// rank whose library is missing can tell the others instead of leaving them inside ncclCommInitRank (mi355/dist.py, step 1)
int mi_comm_probe(void) { return load_rccl(); }

// rank 0: a fresh rendezvous id (mi_comm_id_bytes() = 128 bytes) to hand to every other rank
int mi_comm_unique_id(unsigned char* id_out) {
    if (!id_out) return mi_fail(MI_ERR_ARG, "mi_comm_unique_id: null buffer");
    int rc = load_rccl();
    if (rc != MI_OK) return rc;
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
    memcpy(id_out, &id, sizeof(id));
    return MI_OK;
}

// collective over all `world` processes; the current HIP device is the one this rank reduces on
int mi_comm_init(void** comm_out, int rank, int world, const unsigned char* id) {
    if (!comm_out || !id || world < 1 || rank < 0 || rank >= world) return mi_fail(MI_ERR_ARG, "mi_comm_init: bad arguments");
    int rc = load_rccl();
    if (rc != MI_OK) return rc;
    MiComm* c = (MiComm*)calloc(1, sizeof(MiComm));
    if (!c) return mi_fail(MI_ERR_STATE, "mi_comm_init: out of host memory");
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) { free(c); return rccl_fail("ncclCommInitRank", r); }
    c->rank = rank; c->world = world;
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
        g_rccl.CommDestroy(c->comm); free(c);
        return mi_fail(MI_ERR_STATE, "mi_comm_init: stream / event creation failed");
    }
    *comm_out = c;
    return MI_OK;
}

int mi_comm_destroy(void* comm) {
    MiComm* c = (MiComm*)comm;
    if (!c) return MI_OK;
    (void)hipStreamSynchronize(c->side);
    ncclResult_t r = g_rccl.CommDestroy(c->comm);
    (void)hipEventDestroy(c->ready); (void)hipEventDestroy(c->done); (void)hipStreamDestroy(c->side);
    free(c);
    return r == ncclSuccess ? MI_OK : rccl_fail("ncclCommDestroy", r);
}

// buf[i] <- sum over ranks of buf[i], in stream order on `stream`
int mi_allreduce_sum_f32(void* comm, void* stream, float* buf, long long n) {
    MiComm* c = (MiComm*)comm;
    if (!c || !buf || n < 0) return mi_fail(MI_ERR_ARG, "mi_allreduce_sum_f32: bad arguments");
    if (n == 0) return MI_OK;
    return allreduce_on(c, buf, n, (hipStream_t)stream);
}

// the same sum on the communicator's own stream, after everything queued on `stream` so far; `stream` itself goes on (the next part of the
// backward pass runs under the all-reduce).  mi_comm_wait(comm, stream) joins.
int mi_allreduce_sum_f32_async(void* comm, void* stream, float* buf, long long n) {
    MiComm* c = (MiComm*)comm;
    if (!c || !buf || n < 0) return mi_fail(MI_ERR_ARG, "mi_allreduce_sum_f32_async: bad arguments");
    if (n == 0) return MI_OK;
    if (hipEventRecord(c->ready, (hipStream_t)stream) != hipSuccess || hipStreamWaitEvent(c->side, c->ready, 0) != hipSuccess)
        return mi_fail(MI_ERR_LAUNCH, "mi_allreduce_sum_f32_async: event chaining failed");
    const int rc = allreduce_on(c, buf, n, c->side);
    if (rc != MI_OK) return rc;
    c->pending += 1;
    return MI_OK;
}

// `stream` waits (on the device) for every `_async` all-reduce queued so far
int mi_comm_wait(void* comm, void* stream) {
    MiComm* c = (MiComm*)comm;
    if (!c) return mi_fail(MI_ERR_ARG, "mi_comm_wait: null communicator");
    if (!c->pending) return MI_OK;
    if (hipEventRecord(c->done, c->side) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, c->done, 0) != hipSuccess)
        return mi_fail(MI_ERR_LAUNCH, "mi_comm_wait: event chaining failed");
    c->pending = 0;
    return MI_OK;
}

// rank `root`'s bytes to every rank (the initial parameter replica), in stream order on `stream`
int mi_broadcast(void* comm, void* stream, void* buf, long long bytes, int root) {
    MiComm* c = (MiComm*)comm;
    if (!c || !buf || bytes < 0 || root < 0 || root >= c->world) return mi_fail(MI_ERR_ARG, "mi_broadcast: bad arguments");
    if (bytes == 0) return MI_OK;
    ncclResult_t r = g_rccl.Broadcast(buf, buf, (size_t)bytes, ncclChar, root, c->comm, (hipStream_t)stream);
    return r == ncclSuccess ? MI_OK : rccl_fail("ncclBroadcast", r);
}

}  // extern "C"
