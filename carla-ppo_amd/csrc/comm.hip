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
#include <stdint.h>
#include <stdio.h>
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
    const char* (*GetErrorString)(ncclResult_t);
    ncclResult_t (*CommCount)(const ncclComm_t, int*);    // optional (mi_comm_ranks)
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
    r.CommCount = (decltype(r.CommCount))dlsym(lib, "ncclCommCount");
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
    int algo;                                             // gradient-bucket schedule: 0 ncclAllReduce, 1 reduce-scatter + all-gather (mi_comm_set_algo: the SAME value on every rank)
    // RECORDING communicator (mi_comm_init_recording, round 5): no RCCL, no stream, no device -- every collective this communicator would issue is appended to the
    // caller's host log as {op, floats (bytes for a broadcast), async?, buffer address} and the data is left alone (the sum over ONE rank).  Test infrastructure of the
    // data-parallel step (tests assert that mi_vae_train_step_dp issues the schedule the host path issues, on every rank of 2 / 4 / 8, without a second GPU).
    long long* rec; int rec_cap, rec_n;
};
enum { REC_ALLREDUCE = 1, REC_REDUCE_SCATTER = 2, REC_ALLGATHER = 3, REC_BROADCAST = 4, REC_WAIT = 5 };
void rec_add(MiComm* c, int op, long long count, int async, const void* buf) {
    if (c->rec_n < c->rec_cap) { long long* e = c->rec + 4ll * c->rec_n; e[0] = op; e[1] = count; e[2] = async; e[3] = (long long)(uintptr_t)buf; }
    if (c->rec_n < (1 << 30)) c->rec_n += 1;              // (counts past the capacity too: the caller sees that its log was too short; saturates -- a recording communicator kept
}                                                         //  for a whole benchmark run adds ~7 entries per step)

// Gradient-bucket schedule: 0 = ncclAllReduce (RCCL picks ring / tree / one-shot itself), 1 = reduce-scatter + all-gather: on the
// fully connected xGMI mesh of one node every rank owns 1/W of the bucket, receives the other ranks' pieces of ITS slice over the seven direct
// links in one hop, sums, and sends its finished slice back over the same links -- 2 (W-1)/W of the bucket per link direction, no multi-hop ring.
// Both forms sit behind mi_allreduce_sum_f32 so callers never see the difference.  The schedule is a property of the COMMUNICATOR, set by
// mi_comm_set_algo after the ranks have agreed on it (mi355/dist.py: MI355_COMM_ALGO is read per process and a mismatch would pair one rank's
// reduce-scatter with another's all-reduce -- a hang).  (Never timed: no multi-GPU box in this build environment -- hence off by default.)
// What one call issues is a pure function of (algo, world, rank, n): mi_comm_allreduce_plan, checked on the CPU for every rank of 2 / 4 / 8.
struct Plan { int rsag; long long chunk, mine, tail_off, tail_n; };
Plan plan_of(int algo, int world, int rank, long long n) {
    Plan p = {0, 0, 0, 0, 0};
    const long long chunk = world > 0 ? n / world : 0;
    if (algo == 1 && world > 1 && chunk >= 1024) {
        p.rsag = 1; p.chunk = chunk; p.mine = (long long)rank * chunk; p.tail_off = chunk * world; p.tail_n = n - chunk * world;
    }
    return p;
}

int rccl_fail(const char* what, ncclResult_t r) {
    static thread_local char msg[256];
    snprintf(msg, sizeof(msg), "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
    return mi_fail(MI_ERR_LAUNCH, msg);
}

// in-place sum of buf[0..n) over the ranks on stream `st`
int allreduce_on(MiComm* c, float* buf, long long n, hipStream_t st, int async = 0) {
    const Plan p = plan_of(c->algo, c->world, c->rank, n);
    if (c->rec) {
        if (p.rsag) {
            rec_add(c, REC_REDUCE_SCATTER, p.chunk, async, buf); rec_add(c, REC_ALLGATHER, p.chunk, async, buf + p.mine);
            if (p.tail_n > 0) rec_add(c, REC_ALLREDUCE, p.tail_n, async, buf + p.tail_off);
        } else rec_add(c, REC_ALLREDUCE, n, async, buf);
        return MI_OK;
    }
    if (p.rsag && g_rccl.ReduceScatter && g_rccl.AllGather) {
        // slice r of the first chunk * W elements belongs to rank r (both calls in RCCL's in-place form); the n % W tail rides along as a tiny all-reduce
        float* mine = buf + p.mine;
        ncclResult_t r = g_rccl.ReduceScatter(buf, mine, (size_t)p.chunk, ncclFloat, ncclSum, c->comm, st);
        if (r != ncclSuccess) return rccl_fail("ncclReduceScatter", r);
        r = g_rccl.AllGather(mine, buf, (size_t)p.chunk, ncclFloat, c->comm, st);
        if (r != ncclSuccess) return rccl_fail("ncclAllGather", r);
        if (p.tail_n > 0) {
            r = g_rccl.AllReduce(buf + p.tail_off, buf + p.tail_off, (size_t)p.tail_n, ncclFloat, ncclSum, c->comm, st);
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
// 1: the bound RCCL has ncclReduceScatter and ncclAllGather (the reduce-scatter + all-gather schedule can be set); 0: not, or nothing bound yet.  The host agrees on the
// MINIMUM of this over the ranks before any rank calls mi_comm_set_algo(1) (one rank failing set_algo alone would leave the others in a collective)
int mi_comm_has_rsag(void) { return (g_rccl.lib && g_rccl.ReduceScatter && g_rccl.AllGather) ? 1 : 0; }

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

// A communicator that RECORDS instead of communicating (see MiComm::rec): rank / world are only what the schedule is computed for.  log: HOST memory, 4 long long per
// entry {op (1 all-reduce, 2 reduce-scatter, 3 all-gather, 4 broadcast, 5 wait), floats (bytes for op 4), 1 if issued by the _async form, buffer address};
// mi_comm_recorded = entries issued so far (may exceed the capacity: the log was too short).  Needs neither RCCL nor a GPU.
int mi_comm_init_recording(void** comm_out, int rank, int world, long long* log, int log_capacity) {
    if (!comm_out || !log || log_capacity < 1 || world < 1 || rank < 0 || rank >= world) return mi_fail(MI_ERR_ARG, "mi_comm_init_recording: bad arguments");
    MiComm* c = (MiComm*)calloc(1, sizeof(MiComm));
    if (!c) return mi_fail(MI_ERR_STATE, "mi_comm_init_recording: out of host memory");
    c->rank = rank; c->world = world; c->rec = log; c->rec_cap = log_capacity; c->rec_n = 0;
    *comm_out = c;
    return MI_OK;
}
int mi_comm_recorded(void* comm) { MiComm* c = (MiComm*)comm; return (c && c->rec) ? c->rec_n : -1; }
// how many ranks the communicator spans as RCCL itself reports it (ncclCommCount; a recording communicator: the world it was created for); < 0: error / entry point missing.
// bench.py prints it next to the data-parallel timings: the all-reduce really went over that many devices.
int mi_comm_ranks(void* comm) {
    MiComm* c = (MiComm*)comm;
    if (!c) return mi_fail(MI_ERR_ARG, "mi_comm_ranks: null communicator");
    if (c->rec) return c->world;
    if (!g_rccl.CommCount) return mi_fail(MI_ERR_STATE, "mi_comm_ranks: the bound RCCL lacks ncclCommCount");
    int n = 0;
    const ncclResult_t r = g_rccl.CommCount(c->comm, &n);
    return r == ncclSuccess ? n : rccl_fail("ncclCommCount", r);
}

// the gradient-bucket schedule of this communicator: 0 = ncclAllReduce, 1 = reduce-scatter + all-gather (needs both entry points in the bound library).  Every rank
// must set the SAME value (the caller agrees on it first: mi355/dist.py, which also agrees on whether EVERY rank's library has both entry points); returns MI_OK
int mi_comm_set_algo(void* comm, int algo) {
    MiComm* c = (MiComm*)comm;
    if (!c || algo < 0 || algo > 1) return mi_fail(MI_ERR_ARG, "mi_comm_set_algo: bad arguments");
    if (algo == 1 && !c->rec && !(g_rccl.ReduceScatter && g_rccl.AllGather)) return mi_fail(MI_ERR_STATE, "mi_comm_set_algo: the bound RCCL lacks ncclReduceScatter / ncclAllGather");
    c->algo = algo;
    return MI_OK;
}

// what one all-reduce of n floats issues on rank `rank` of `world` under schedule `algo` (no GPU, no RCCL needed): out[0] = 1 reduce-scatter + all-gather / 0 one
// ncclAllReduce of n; out[1] = floats per rank slice; out[2] = this rank's slice offset; out[3], out[4] = offset and length of the tail all-reduce (n % world floats)
int mi_comm_allreduce_plan(int algo, int world, int rank, long long n, long long* out5) {
    if (!out5 || world < 1 || rank < 0 || rank >= world || n < 0 || algo < 0 || algo > 1) return mi_fail(MI_ERR_ARG, "mi_comm_allreduce_plan: bad arguments");
    const Plan p = plan_of(algo, world, rank, n);
    out5[0] = p.rsag; out5[1] = p.chunk; out5[2] = p.mine; out5[3] = p.tail_off; out5[4] = p.tail_n;
    return MI_OK;
}

int mi_comm_destroy(void* comm) {
    MiComm* c = (MiComm*)comm;
    if (!c) return MI_OK;
    if (c->rec) { free(c); return MI_OK; }
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
    if (c->rec) { c->pending += 1; return allreduce_on(c, buf, n, nullptr, 1); }
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
    if (c->rec) { rec_add(c, REC_WAIT, c->pending, 0, nullptr); c->pending = 0; return MI_OK; }
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
    if (c->rec) { rec_add(c, REC_BROADCAST, bytes, 0, buf); return MI_OK; }
    ncclResult_t r = g_rccl.Broadcast(buf, buf, (size_t)bytes, ncclChar, root, c->comm, (hipStream_t)stream);
    return r == ncclSuccess ? MI_OK : rccl_fail("ncclBroadcast", r);
}

}  // extern "C"
