// Data-parallel exchange step behind the C ABI: RCCL over xGMI, called directly.
//
// The reference trains on ONE device (`training_GPUs=[0]`, ecog2txt/trainers.py:131); sharding utterances over the
// GPUs of a node adds exactly one exchange per optimisation step: the sum of the flat fp32 gradient buffer, range by
// range in the order the backward pass completes the ranges.  One communicator per process (= per GPU).  The
// collectives run on a stream the communicator owns; ordering against the caller's streams is by events created once
// at e2t_comm_init (no allocation afterwards), so every call is asynchronous and legal inside a hipGraph capture
// (the communicator's stream then joins the capture through the event edge and must be joined back with
// e2t_comm_wait before the capture ends).
//
// librccl is loaded with dlopen at e2t_comm_init: the compute entry points of this library do not depend on it, and a
// process that already carries an RCCL (torch ships one) keeps exactly that copy.
#include "common.h"
#include "ecog2txt_hip.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <mutex>
#include <string.h>

namespace {
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
RcclApi g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
    // a copy that is already mapped (torch's) wins; then the ROCm install's
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) { g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (g_rccl.handle) break; }
    if (!g_rccl.handle)
        for (const char* n : names) { g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (g_rccl.handle) break; }
    if (!g_rccl.handle) return;
#define E2T_SYM(field, name) g_rccl.field = (decltype(g_rccl.field))dlsym(g_rccl.handle, name)
    E2T_SYM(GetUniqueId, "ncclGetUniqueId"); E2T_SYM(CommInitRank, "ncclCommInitRank"); E2T_SYM(CommDestroy, "ncclCommDestroy");
    E2T_SYM(AllReduce, "ncclAllReduce"); E2T_SYM(Broadcast, "ncclBroadcast"); E2T_SYM(GetErrorString, "ncclGetErrorString");
#undef E2T_SYM
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.Broadcast && g_rccl.GetErrorString;
}
int need_rccl() {
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.ok) {
        const char* why = dlerror();             // (the call clears the state: read it once)
        e2t_set_error("librccl could not be loaded (%s)", why ? why : "symbols missing");
        return E2T_ERR_HIP;
    }
    return E2T_OK;
}
}  // namespace

#define E2T_NEVENTS 64
struct e2t_comm {
    ncclComm_t comm;
    hipStream_t stream;                    // the collectives' stream
    hipEvent_t ev[E2T_NEVENTS];            // ring: "caller's work so far" edges and per-collective completion tickets
    unsigned next;
    int rank, nranks, device;
};

#define E2T_NCCL(call)                                                                                         \
    do { ncclResult_t r_ = (call);                                                                              \
         if (r_ != ncclSuccess) { e2t_set_error("%s:%d: rccl: %s", __FILE__, __LINE__, g_rccl.GetErrorString(r_)); return E2T_ERR_HIP; } \
    } while (0)
#define E2T_HIP(call)                                                                                          \
    do { hipError_t e_ = (call);                                                                                \
         if (e_ != hipSuccess) { e2t_set_error("%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e_)); return E2T_ERR_HIP; } \
    } while (0)

extern "C" int e2t_comm_unique_id(void* id128) {
    E2T_CHECK_ARG(id128);
    if (int rc = need_rccl()) return rc;
    static_assert(sizeof(ncclUniqueId) == E2T_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    E2T_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return E2T_OK;
}

extern "C" int e2t_comm_init(e2t_comm** out, int rank, int nranks, const void* id128, int device) {
    E2T_CHECK_ARG(out && id128 && nranks >= 1 && rank >= 0 && rank < nranks);
    if (int rc = need_rccl()) return rc;
    E2T_HIP(hipSetDevice(device));
    e2t_comm* c = new e2t_comm();
    c->rank = rank; c->nranks = nranks; c->device = device; c->next = 0;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) { e2t_set_error("ncclCommInitRank: %s", g_rccl.GetErrorString(r)); delete c; return E2T_ERR_HIP; }
    // (a failure below must not leak the communicator, the stream or the events made so far)
    // (Round 6: a HIGH-PRIORITY stream here -- so that the communicator never shares a hardware queue with the engine's side stream,
    //  profiles/r06_dp_timeline_graph_per_stage.txt -- was tried and withdrawn: with a priority stream in the process, hipGraphLaunch
    //  of ROCm 7.0.2 crashed (SIGSEGV inside libamdhip64) in the next multi-branch graph replay of the test suite,
    //  tests/test_gpu_e2e.py::test_sequential_transfer_and_resume behind the data-parallel tests; it bought nothing on one rank.)
    hipError_t he = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    int nev = 0;
    for (; he == hipSuccess && nev < E2T_NEVENTS; ++nev) he = hipEventCreateWithFlags(&c->ev[nev], hipEventDisableTiming);
    if (he != hipSuccess) {
        e2t_set_error("%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(he));
        for (int i = 0; i + 1 < nev; ++i) (void)hipEventDestroy(c->ev[i]);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        (void)g_rccl.CommDestroy(c->comm);
        delete c;
        return E2T_ERR_HIP;
    }
    *out = c;
    return E2T_OK;
}

extern "C" int e2t_comm_destroy(e2t_comm* c) {
    if (!c) return E2T_OK;
    (void)hipStreamSynchronize(c->stream);
    if (g_rccl.ok) (void)g_rccl.CommDestroy(c->comm);
    for (int i = 0; i < E2T_NEVENTS; ++i) (void)hipEventDestroy(c->ev[i]);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return E2T_OK;
}

extern "C" int e2t_comm_rank(const e2t_comm* c) { return c ? c->rank : -1; }
extern "C" int e2t_comm_size(const e2t_comm* c) { return c ? c->nranks : 0; }

static int order_after(e2t_comm* c, void* after_stream) {
    // the communicator's stream waits for everything enqueued so far on the caller's stream
    hipEvent_t e = c->ev[c->next++ % E2T_NEVENTS];
    E2T_HIP(hipEventRecord(e, (hipStream_t)after_stream));
    E2T_HIP(hipStreamWaitEvent(c->stream, e, 0));
    return E2T_OK;
}
// a ticket is the RUNNING index of its event (not the ring slot), so that e2t_comm_wait can tell a ticket whose slot has been
// reused since -- two events are taken per collective, i.e. after 32 later collectives -- from a live one
static int ticket(e2t_comm* c, int* out) {
    const unsigned t = c->next++;
    E2T_HIP(hipEventRecord(c->ev[t % E2T_NEVENTS], c->stream));
    if (out) *out = (int)(t & 0x7FFFFFFFu);
    return E2T_OK;
}

// The communicator's stream waits for everything enqueued so far on `after_stream` and issues nothing.  Inside a stream capture
// this is how the communicator's stream ENTERS the capture from the capture's origin stream: ROCm 7.0's hipGraphInstantiate crashes
// on a stream that was pulled into a capture by a stream that was itself pulled in (a fork off a forked stream), so a captured
// step calls this once on its main stream before the first collective is ordered behind a side stream's work.
// One rank: RCCL enqueues nothing for an in-place collective, which leaves the communicator's stream without a node of its own
// inside a captured step -- a stream that only forwards dependencies (ROCm 7.0's graph instantiation then recurses without end on
// the merged edges).  A one-thread marker kernel stands in for the collective's kernel, so that a one-rank communicator (tests,
// one-GPU runs of the data-parallel schedule) records the same graph shape as a multi-rank one.
__global__ void k_comm_marker(int) {}
static void one_rank_marker(e2t_comm* c) { if (c->nranks == 1) hipLaunchKernelGGL(k_comm_marker, dim3(1), dim3(1), 0, c->stream, 0); }

extern "C" int e2t_comm_order_after(e2t_comm* c, void* after_stream) {
    E2T_CHECK_ARG(c);
    return order_after(c, after_stream);
}

extern "C" int e2t_comm_allreduce_f32(e2t_comm* c, float* buf, size_t n, void* after_stream, int* ticket_out) {
    E2T_CHECK_ARG(c && (buf || n == 0));
    if (int rc = order_after(c, after_stream)) return rc;
    if (n) E2T_NCCL(g_rccl.AllReduce(buf, buf, n, ncclFloat32, ncclSum, c->comm, c->stream));
    one_rank_marker(c);
    return ticket(c, ticket_out);
}

extern "C" int e2t_comm_allreduce_i32(e2t_comm* c, int32_t* buf, size_t n, void* after_stream, int* ticket_out) {
    E2T_CHECK_ARG(c && (buf || n == 0));
    if (int rc = order_after(c, after_stream)) return rc;
    if (n) E2T_NCCL(g_rccl.AllReduce(buf, buf, n, ncclInt32, ncclSum, c->comm, c->stream));
    one_rank_marker(c);
    return ticket(c, ticket_out);
}

// ABI 8: the 'this step is invalid' word of the persistent recurrences is agreed on by MAX, not by SUM -- the word is only cleared
// when the host looks at it (once per epoch), a sum taken at every step in between multiplies a raised word by the number of ranks
// per step (and wraps to zero after 32 / log2(ranks) steps: the optimiser would silently resume); the maximum is idempotent and keeps
// the code of the error.
extern "C" int e2t_comm_allreduce_max_i32(e2t_comm* c, int32_t* buf, size_t n, void* after_stream, int* ticket_out) {
    E2T_CHECK_ARG(c && (buf || n == 0));
    if (int rc = order_after(c, after_stream)) return rc;
    if (n) E2T_NCCL(g_rccl.AllReduce(buf, buf, n, ncclInt32, ncclMax, c->comm, c->stream));
    one_rank_marker(c);
    return ticket(c, ticket_out);
}

extern "C" int e2t_comm_broadcast(e2t_comm* c, void* buf, size_t bytes, int root, void* after_stream, int* ticket_out) {
    E2T_CHECK_ARG(c && (buf || bytes == 0) && root >= 0 && root < c->nranks);
    if (int rc = order_after(c, after_stream)) return rc;
    if (bytes) E2T_NCCL(g_rccl.Broadcast(buf, buf, bytes, ncclUint8, root, c->comm, c->stream));
    return ticket(c, ticket_out);
}

extern "C" int e2t_comm_wait(e2t_comm* c, int ticket_id, void* stream) {
    E2T_CHECK_ARG(c);
    if (ticket_id < 0) {                                   // everything issued so far
        int t;
        if (int rc = ticket(c, &t)) return rc;
        ticket_id = t;
    }
    const unsigned age = ((c->next & 0x7FFFFFFFu) - (unsigned)ticket_id) & 0x7FFFFFFFu;      // events taken since, this one included
    if (age == 0 || age > E2T_NEVENTS) {
        e2t_set_error("e2t_comm_wait: ticket %d is stale (its event was reused: at most %d collectives may be outstanding)", ticket_id, E2T_NEVENTS / 2);
        return E2T_ERR_ARG;
    }
    E2T_HIP(hipStreamWaitEvent((hipStream_t)stream, c->ev[(unsigned)ticket_id % E2T_NEVENTS], 0));
    return E2T_OK;
}
