// Gradient exchange inside the library: RCCL looked up at run time, the bucketed all-reduce, fsmg_comm_*.
// Host-side C++ only (part of the C-ABI of libfsmg, include/fsmg.h); every kernel lives in gemm.hip / lstm_*.hip / elementwise.hip.
#include "fsmg_model.h"

using namespace fsmg;
using namespace fsmg_host;

namespace fsmg_host {

// ---- RCCL, looked up at run time (the library has no link-time dependency on it; a process that already loaded torch's
// librccl.so gets that one)
struct Rccl {
    typedef struct { char internal[128]; } UniqueId;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr; int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
    bool ok = false;
    Rccl() {
        static const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        void* lib = nullptr;
        for (const char* name : names) if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);   // the copy already in the process first
        for (const char* name : names) if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (!lib) { why = "librccl.so not found (dlopen)"; return; }
        GetUniqueId = (int (*)(UniqueId*))dlsym(lib, "ncclGetUniqueId");
        CommInitRank = (int (*)(void**, int, UniqueId, int))dlsym(lib, "ncclCommInitRank");
        CommDestroy = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
        AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(lib, "ncclAllReduce");
        Broadcast = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(lib, "ncclBroadcast");
        GroupStart = (int (*)())dlsym(lib, "ncclGroupStart");
        GroupEnd = (int (*)())dlsym(lib, "ncclGroupEnd");
        GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
        ok = GetUniqueId && CommInitRank && CommDestroy && AllReduce && Broadcast && GroupStart && GroupEnd && GetErrorString;
        if (!ok) why = "librccl.so lacks an expected symbol";
    }
};
inline Rccl& rccl() { static Rccl r; return r; }
constexpr int NCCL_FLOAT = 7, NCCL_SUM = 0, NCCL_CHAR = 0;
#define NCCLCK(h, call)                                                                                   \
    do {                                                                                                  \
        const int e_ = (call);                                                                            \
        if (e_ != 0) return fail(h, FSMG_ERR_HIP, std::string(#call) + ": " + rccl().GetErrorString(e_)); \
    } while (0)

// sum of the gradient buffer over the ranks: three buckets on the communication stream, each behind its readiness event
// (bucket 0 = softmax gradients: final behind the projection-gradient GEMMs when the backward pass is cut there); the compute
// stream (not the host) then waits for the communication stream
void comm_destroy(fsmg_model* h) {
    if (h->comm && h->own_comm && rccl().ok) rccl().CommDestroy(h->comm);
    if (h->ev_comm) hipEventDestroy(h->ev_comm);
    if (h->comm_stream) hipStreamDestroy(h->comm_stream);
    h->comm = nullptr; h->ev_comm = nullptr; h->comm_stream = nullptr;
}

int exchange_gradients(fsmg_model* h) {
    Rccl& r = rccl();
    HIPCK(h, hipStreamWaitEvent(h->comm_stream, h->ev_bucket[0], 0));
    NCCLCK(h, r.AllReduce(h->G + h->off_w, h->G + h->off_w, (size_t)(h->n_flat - h->off_w), NCCL_FLOAT, NCCL_SUM, h->comm, h->comm_stream));
    HIPCK(h, hipStreamWaitEvent(h->comm_stream, h->ev_bucket[1], 0));
    NCCLCK(h, r.GroupStart());
    NCCLCK(h, r.AllReduce(h->G, h->G, (size_t)h->off_w, NCCL_FLOAT, NCCL_SUM, h->comm, h->comm_stream));
    NCCLCK(h, r.AllReduce(h->G + h->n_flat, h->G + h->n_flat, (size_t)FSMG_GRAD_TAIL, NCCL_FLOAT, NCCL_SUM, h->comm, h->comm_stream));
    NCCLCK(h, r.GroupEnd());
    HIPCK(h, hipEventRecord(h->ev_comm, h->comm_stream));
    HIPCK(h, hipStreamWaitEvent(h->stream, h->ev_comm, 0));
    return FSMG_OK;
}

}  // namespace fsmg_host

// =========================================================================== C ABI
extern "C" {

// ---- the gradient exchange inside the library
static int comm_prepare(fsmg_handle h) {
    if (!rccl().ok) return fail(h, FSMG_ERR_STATE, "RCCL is not available: " + rccl().why);
    BEGIN_CALL(h);
    if (!h->comm_stream) HIPCK(h, hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
    if (!h->ev_comm) HIPCK(h, hipEventCreateWithFlags(&h->ev_comm, hipEventDisableTiming));
    return FSMG_OK;
}

int fsmg_comm_unique_id(char id[FSMG_COMM_ID_BYTES]) {
    if (!id) return FSMG_ERR_INVALID;
    if (!rccl().ok) return fail(nullptr, FSMG_ERR_STATE, "RCCL is not available: " + rccl().why);
    Rccl::UniqueId u;
    const int e = rccl().GetUniqueId(&u);
    if (e != 0) return fail(nullptr, FSMG_ERR_HIP, std::string("ncclGetUniqueId: ") + rccl().GetErrorString(e));
    std::memcpy(id, u.internal, FSMG_COMM_ID_BYTES);
    return FSMG_OK;
}

int fsmg_comm_init(fsmg_handle h, const char id[FSMG_COMM_ID_BYTES], int32_t world_size, int32_t rank) {
    if (!h || !id || world_size < 1 || rank < 0 || rank >= world_size) return FSMG_ERR_INVALID;
    if (h->comm) return fail(h, FSMG_ERR_STATE, "a communicator is already attached");
    int rc = comm_prepare(h);
    if (rc != FSMG_OK) return rc;
    Rccl::UniqueId u;
    std::memcpy(u.internal, id, FSMG_COMM_ID_BYTES);
    void* c = nullptr;
    NCCLCK(h, rccl().CommInitRank(&c, world_size, u, rank));
    h->comm = c; h->own_comm = true; h->world = world_size; h->rank = rank;
    drop_graphs(h);
    return FSMG_OK;
}

int fsmg_comm_attach(fsmg_handle h, void* nccl_comm, int32_t world_size, int32_t rank) {
    if (!h || !nccl_comm || world_size < 1 || rank < 0 || rank >= world_size) return FSMG_ERR_INVALID;
    if (h->comm) return fail(h, FSMG_ERR_STATE, "a communicator is already attached");
    int rc = comm_prepare(h);
    if (rc != FSMG_OK) return rc;
    h->comm = nccl_comm; h->own_comm = false; h->world = world_size; h->rank = rank;
    drop_graphs(h);
    return FSMG_OK;
}

int fsmg_comm_broadcast_state(fsmg_handle h, int32_t root) {
    if (!h) return FSMG_ERR_INVALID;
    if (!h->comm || root < 0 || root >= h->world) return fail(h, FSMG_ERR_STATE, "no communicator attached / bad root");
    BEGIN_CALL(h);
    HIPCK(h, hipStreamSynchronize(h->stream));
    NCCLCK(h, rccl().GroupStart());
    NCCLCK(h, rccl().Broadcast(h->P, h->P, (size_t)h->n_flat, NCCL_FLOAT, root, h->comm, h->comm_stream));
    NCCLCK(h, rccl().Broadcast(h->M, h->M, (size_t)2 * h->n_flat, NCCL_FLOAT, root, h->comm, h->comm_stream));     // m and v are adjacent
    NCCLCK(h, rccl().Broadcast(h->d_step, h->d_step, sizeof(long long), NCCL_CHAR, root, h->comm, h->comm_stream));
    NCCLCK(h, rccl().GroupEnd());
    HIPCK(h, hipStreamSynchronize(h->comm_stream));
    h->khf_dirty = true;
    return FSMG_OK;
}

int fsmg_comm_release(fsmg_handle h) {
    if (!h) return FSMG_ERR_INVALID;
    BEGIN_CALL(h);
    HIPCK(h, hipStreamSynchronize(h->stream));
    if (h->comm_stream) HIPCK(h, hipStreamSynchronize(h->comm_stream));
    if (h->comm && h->own_comm && rccl().ok) rccl().CommDestroy(h->comm);
    h->comm = nullptr; h->own_comm = false; h->world = 1; h->rank = 0;
    drop_graphs(h);
    return FSMG_OK;
}

}  // extern "C"
