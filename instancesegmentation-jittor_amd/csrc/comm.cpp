// comm.cpp -- RCCL all-gather of fixed-size detection records over xGMI.
//
// The reference tree has no communication code; upstream maskrcnn-benchmark
// all-gathers pickled {image_id: BoxList} dicts in engine/inference.py.  Here images shard by batch
// across ranks (one process per GPU) and the ONLY exchange is one ncclAllGather of fixed-capacity
// records per batch, issued on the engine's results stream behind the kernel that packed them (it overlaps the next batch's backbone).
// librccl.so is dlopen'ed lazily: single-GPU use never touches it.
#include <dlfcn.h>
#include <string.h>

#include "../../include/isegmi.h"
#include "common.h"
#include <rccl/rccl.h>

namespace isegmi {

// Types and prototypes come from RCCL's own header (round 3; rounds 1-2 re-declared them by hand); the library itself is still resolved at
// run time, so a single-GPU process never loads it.
typedef ncclUniqueId ncclUniqueId_t;
typedef ncclComm_t ncclComm_tt;
typedef decltype(&ncclGetUniqueId) fn_getuid;
typedef decltype(&ncclCommInitRank) fn_initrank;
typedef decltype(&ncclAllGather) fn_allgather;
typedef decltype(&ncclCommDestroy) fn_destroy;
typedef decltype(&ncclGetErrorString) fn_errstr;
static_assert(sizeof(ncclUniqueId) == 128, "isegmi_comm_unique_id hands out 128 bytes");

static struct {
    void* so = nullptr;
    fn_getuid getuid = nullptr;
    fn_initrank initrank = nullptr;
    fn_allgather allgather = nullptr;
    fn_destroy destroy = nullptr;
    fn_errstr errstr = nullptr;
} R;

static int load_rccl() {
    if (R.so) return ISEGMI_OK;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) { R.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (R.so) break; }
    if (!R.so) { set_error(std::string("cannot dlopen librccl.so: ") + dlerror()); return ISEGMI_ERR_RCCL; }
    R.getuid = (fn_getuid)dlsym(R.so, "ncclGetUniqueId");
    R.initrank = (fn_initrank)dlsym(R.so, "ncclCommInitRank");
    R.allgather = (fn_allgather)dlsym(R.so, "ncclAllGather");
    R.destroy = (fn_destroy)dlsym(R.so, "ncclCommDestroy");
    R.errstr = (fn_errstr)dlsym(R.so, "ncclGetErrorString");
    if (!R.getuid || !R.initrank || !R.allgather || !R.destroy) { set_error("librccl.so lacks expected symbols"); return ISEGMI_ERR_RCCL; }
    return ISEGMI_OK;
}

#define RCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t _r = (expr);                                                                        \
        if (_r != ncclSuccess) {                                                                                   \
            set_error(std::string(#expr) + " -> " + (R.errstr ? R.errstr(_r) : "rccl error"));          \
            return ISEGMI_ERR_RCCL;                                                                      \
        }                                                                                                \
    } while (0)

}  // namespace isegmi
using namespace isegmi;

// Two record slots: step t gathers out of / into slot t % 2 while the producer already packs step t+1 into the other one.
// done[s] is recorded behind slot s's all-gather; isegmi_comm_fence_producer makes the producer stream wait on it before it
// overwrites send[s] / before a new gather overwrites recv[s] (WAR), so no host synchronisation is needed between steps.
constexpr int COMM_SLOTS = 2;
struct isegmi_comm {
    ncclComm_tt comm = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr;
    hipEvent_t done[COMM_SLOTS] = {nullptr, nullptr};
    bool used[COMM_SLOTS] = {false, false};
    int rank = 0, world = 1;
    // One communicator = ONE order of collectives.  RCCL launches a collective on the stream it is given; two collectives of one communicator on two
    // streams are ordered only by the runtime's good will.  last_slot / last_stream remember where the previous collective went: a collective that goes
    // to ANOTHER stream first makes that stream wait (on the device) for the previous one's done event, so the order in which the host issued them is
    // the order in which they run, whatever streams the caller mixes (a rank's empty step or a control word between two data steps).
    int last_slot = -1;
    hipStream_t last_stream = nullptr;
    int64_t collectives = 0, stream_switches = 0;
};

extern "C" int isegmi_comm_unique_id(void* out128) {
    ARG_CHECK(out128, "null");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId_t id;
    RCCL_TRY(R.getuid(&id));
    memcpy(out128, &id, sizeof(id));
    return ISEGMI_OK;
}

extern "C" int isegmi_comm_create(const void* uid128, int rank, int world, isegmi_comm** out) {
    ARG_CHECK(uid128 && out && world > 0 && rank >= 0 && rank < world, "comm args");
    int rc = load_rccl();
    if (rc) return rc;
    isegmi_comm* c = new isegmi_comm();
    c->rank = rank; c->world = world;
    ncclUniqueId_t id;
    memcpy(&id, uid128, sizeof(id));
    ncclResult_t r = R.initrank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { set_error(std::string("ncclCommInitRank -> ") + (R.errstr ? R.errstr(r) : "?")); delete c; return ISEGMI_ERR_RCCL; }
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
    for (int i = 0; i < COMM_SLOTS; ++i) HIP_TRY(hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming));
    *out = c;
    return ISEGMI_OK;
}

extern "C" int isegmi_comm_destroy(isegmi_comm* c) {
    if (!c) return ISEGMI_OK;
    (void)hipStreamSynchronize(c->stream);
    if (c->comm) R.destroy(c->comm);
    (void)hipEventDestroy(c->ready);
    for (int i = 0; i < COMM_SLOTS; ++i) (void)hipEventDestroy(c->done[i]);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return ISEGMI_OK;
}

// WAR fence of slot `slot`: everything enqueued on `producer_stream` after this call runs after the slot's last all-gather has
// finished reading its send block and writing its receive block.  Call it before re-packing records into the slot's send buffer.
extern "C" int isegmi_comm_fence_producer(isegmi_comm* c, int slot, void* producer_stream) {
    ARG_CHECK(c && slot >= 0 && slot < COMM_SLOTS, "fence args");
    if (c->used[slot]) HIP_TRY(hipStreamWaitEvent((hipStream_t)producer_stream, c->done[slot], 0));
    return ISEGMI_OK;
}

// All-gather `bytes` bytes from every rank through slot `slot` (0 or 1): d_recv holds world*bytes, rank r's block at r*bytes.
// Runs ON `producer_stream` (the engine's results stream), right behind the kernel that packed the block; on the comm's own stream only when
// there is no producer (a control word; isegmi.dist hands a rank's empty step the stream of its data steps).  Rounds 1-2 always used the comm's stream: one more stream for the runtime to fold onto its four
// hardware queues -- on the measured box it shared the MAIN stream's queue, so the collective of step i (enqueued after forward i, waiting for
// step i's last kernel) sat in front of forward i+1's backbone in that in-order queue and the cross-step overlap was gone: bench.py through its
// N > 1 code path on one GPU (world-1 communicators, tools/forced_dist_bench.sh) read 926 img/s against 961 without the collective.  On the
// results stream the collective only holds back the NEXT step's Detect / postprocess chain, which starts a whole backbone + heads later.
extern "C" int isegmi_comm_allgather_slot(isegmi_comm* c, int slot, const void* d_send, void* d_recv, int64_t bytes, void* producer_stream) {
    ARG_CHECK(c && d_send && d_recv && bytes > 0 && slot >= 0 && slot < COMM_SLOTS, "allgather args");
    hipStream_t s = producer_stream ? (hipStream_t)producer_stream : c->stream;
    if (c->last_slot >= 0 && c->last_stream != s) {   // the previous collective of this communicator went to another stream: order behind it
        HIP_TRY(hipStreamWaitEvent(s, c->done[c->last_slot], 0));
        ++c->stream_switches;
    }
    RCCL_TRY(R.allgather(d_send, d_recv, (size_t)bytes, ncclInt8, c->comm, s));
    HIP_TRY(hipEventRecord(c->done[slot], s));
    c->used[slot] = true;
    c->last_slot = slot; c->last_stream = s;
    ++c->collectives;
    return ISEGMI_OK;
}

extern "C" int isegmi_comm_allgather(isegmi_comm* c, const void* d_send, void* d_recv, int64_t bytes, void* producer_stream) {
    return isegmi_comm_allgather_slot(c, 0, d_send, d_recv, bytes, producer_stream);
}

extern "C" int isegmi_comm_info(isegmi_comm* c, int64_t* out4) {
    ARG_CHECK(c && out4, "null");
    out4[0] = c->rank; out4[1] = c->world; out4[2] = c->collectives; out4[3] = c->stream_switches;
    return ISEGMI_OK;
}

extern "C" int isegmi_comm_wait_slot(isegmi_comm* c, int slot) {
    ARG_CHECK(c && slot >= 0 && slot < COMM_SLOTS, "wait args");
    if (c->used[slot]) HIP_TRY(hipEventSynchronize(c->done[slot]));
    return ISEGMI_OK;
}

extern "C" int isegmi_comm_wait(isegmi_comm* c) {
    ARG_CHECK(c, "null");
    for (int i = 0; i < COMM_SLOTS; ++i) if (c->used[i]) HIP_TRY(hipEventSynchronize(c->done[i]));
    return ISEGMI_OK;
}
