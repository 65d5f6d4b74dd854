// comm.cpp -- RCCL all-gather of fixed-size detection records over xGMI.
//
// The reference tree has no communication code; upstream maskrcnn-benchmark
// all-gathers pickled {image_id: BoxList} dicts in engine/inference.py.  Here images shard by batch
// across ranks (one process per GPU) and the ONLY exchange is one ncclAllGather of fixed-capacity
// records per batch, issued on its own stream so it overlaps the next batch's backbone.
// librccl.so is dlopen'ed lazily: single-GPU use never touches it.
#include <dlfcn.h>
#include <string.h>

#include "../../include/isegmi.h"
#include "common.h"

namespace isegmi {

typedef struct { char internal[128]; } ncclUniqueId_t;
typedef void* ncclComm_tt;
typedef int (*fn_getuid)(ncclUniqueId_t*);
typedef int (*fn_initrank)(ncclComm_tt*, int, ncclUniqueId_t, int);
typedef int (*fn_allgather)(const void*, void*, size_t, int, ncclComm_tt, hipStream_t);
typedef int (*fn_destroy)(ncclComm_tt);
typedef const char* (*fn_errstr)(int);

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
        int _r = (expr);                                                                                 \
        if (_r != 0) {                                                                                   \
            set_error(std::string(#expr) + " -> " + (R.errstr ? R.errstr(_r) : "rccl error"));          \
            return ISEGMI_ERR_RCCL;                                                                      \
        }                                                                                                \
    } while (0)

}  // namespace isegmi
using namespace isegmi;

struct isegmi_comm {
    ncclComm_tt comm = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr, done = nullptr;
    int rank = 0, world = 1;
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
    int r = R.initrank(&c->comm, world, id, rank);
    if (r != 0) { set_error(std::string("ncclCommInitRank -> ") + (R.errstr ? R.errstr(r) : "?")); delete c; return ISEGMI_ERR_RCCL; }
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
    *out = c;
    return ISEGMI_OK;
}

extern "C" int isegmi_comm_destroy(isegmi_comm* c) {
    if (!c) return ISEGMI_OK;
    (void)hipStreamSynchronize(c->stream);
    if (c->comm) R.destroy(c->comm);
    (void)hipEventDestroy(c->ready);
    (void)hipEventDestroy(c->done);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return ISEGMI_OK;
}

// All-gather `bytes` bytes from every rank: d_recv holds world*bytes, rank r's block at r*bytes.
// Ordered after everything already enqueued on `producer_stream`; runs on the comm's own stream.
extern "C" int isegmi_comm_allgather(isegmi_comm* c, const void* d_send, void* d_recv, int64_t bytes, void* producer_stream) {
    ARG_CHECK(c && d_send && d_recv && bytes > 0, "allgather args");
    HIP_TRY(hipEventRecord(c->ready, (hipStream_t)producer_stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ready, 0));
    RCCL_TRY(R.allgather(d_send, d_recv, (size_t)bytes, /*ncclInt8*/ 0, c->comm, c->stream));
    HIP_TRY(hipEventRecord(c->done, c->stream));
    return ISEGMI_OK;
}

extern "C" int isegmi_comm_wait(isegmi_comm* c) {
    ARG_CHECK(c, "null");
    HIP_TRY(hipEventSynchronize(c->done));
    return ISEGMI_OK;
}
