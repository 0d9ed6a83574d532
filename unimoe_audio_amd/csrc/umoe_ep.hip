// Expert-parallel exchange over RCCL (xGMI): the all-to-all of DeepSpeed's _AllToAll in the reference
// (utils/UniMoE_Audio_core.py:467,480; utils/UniMoE_Audio_utils.py:332-335).  The communicator is an opaque pointer
// (ncclComm_t) owned by the caller; librccl is resolved at first use (dlopen), so the library loads on hosts without it.
// One process per GPU; every rank passes slabs of equal size (the fixed per-destination capacity of unimoe_audio_amd/ep.py:
// no MAX all-reduce, static buffers): rank r sends send + p*bytes to peer p and receives peer p's slab at recv + p*bytes.
#include "umoe_common.h"
#include <dlfcn.h>
#include <string.h>

namespace {
typedef int (*fn_void)();
typedef int (*fn_uid)(void*);
struct uid128 { char b[128]; };   // ncclUniqueId (passed BY VALUE to ncclCommInitRank)
typedef int (*fn_init_t)(void**, int, uid128, int);
typedef int (*fn_destroy)(void*);
typedef int (*fn_sendrecv)(const void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_recv)(void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*fn_err)(int);

struct Rccl {
    void* h = nullptr;
    fn_void group_start = nullptr, group_end = nullptr;
    fn_uid get_uid = nullptr;
    fn_init_t init_rank = nullptr;
    fn_destroy destroy = nullptr;
    fn_sendrecv send = nullptr;
    fn_recv recv = nullptr;
    fn_err err = nullptr;
};
Rccl g_rccl;

int load_rccl() {
    if (g_rccl.h) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    UMOE_REQUIRE(h, "umoe_ep: librccl not found (%s)", dlerror());
    g_rccl.group_start = (fn_void)dlsym(h, "ncclGroupStart");
    g_rccl.group_end = (fn_void)dlsym(h, "ncclGroupEnd");
    g_rccl.get_uid = (fn_uid)dlsym(h, "ncclGetUniqueId");
    g_rccl.init_rank = (fn_init_t)dlsym(h, "ncclCommInitRank");
    g_rccl.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
    g_rccl.send = (fn_sendrecv)dlsym(h, "ncclSend");
    g_rccl.recv = (fn_recv)dlsym(h, "ncclRecv");
    g_rccl.err = (fn_err)dlsym(h, "ncclGetErrorString");
    UMOE_REQUIRE(g_rccl.group_start && g_rccl.group_end && g_rccl.get_uid && g_rccl.init_rank && g_rccl.destroy && g_rccl.send && g_rccl.recv,
                 "umoe_ep: librccl lacks a required symbol");
    g_rccl.h = h;
    return 0;
}
#define UMOE_NCCL(expr)                                                                                       \
    do {                                                                                                      \
        int _r = (expr);                                                                                      \
        if (_r != 0) {                                                                                        \
            umoe_set_error("%s failed: %s", #expr, g_rccl.err ? g_rccl.err(_r) : "rccl error");               \
            return -3;                                                                                        \
        }                                                                                                     \
    } while (0)
}  // namespace

extern "C" int umoe_ep_unique_id(void* out128) {
    UMOE_REQUIRE(out128, "umoe_ep_unique_id: null argument");
    if (int rc = load_rccl()) return rc;
    UMOE_NCCL(g_rccl.get_uid(out128));
    return 0;
}

extern "C" int umoe_ep_comm_create(const void* uid128_bytes, int rank, int nranks, void** comm_out) {
    UMOE_REQUIRE(uid128_bytes && comm_out && nranks >= 1 && rank >= 0 && rank < nranks, "umoe_ep_comm_create: bad argument");
    if (int rc = load_rccl()) return rc;
    uid128 id;
    memcpy(id.b, uid128_bytes, sizeof(id.b));
    UMOE_NCCL(g_rccl.init_rank(comm_out, nranks, id, rank));
    return 0;
}

extern "C" int umoe_ep_comm_destroy(void* comm) {
    if (!comm) return 0;
    if (int rc = load_rccl()) return rc;
    UMOE_NCCL(g_rccl.destroy(comm));
    return 0;
}

extern "C" int umoe_ep_all_to_all(void* comm, const void* send, void* recv, size_t bytes_per_peer, int nranks, umoe_stream_t stream) {
    UMOE_REQUIRE(comm && send && recv && nranks >= 1, "umoe_ep_all_to_all: null argument");
    if (bytes_per_peer == 0) return 0;
    if (int rc = load_rccl()) return rc;
    hipStream_t s = (hipStream_t)stream;
    UMOE_NCCL(g_rccl.group_start());
    for (int p = 0; p < nranks; ++p) {
        UMOE_NCCL(g_rccl.send(reinterpret_cast<const char*>(send) + (size_t)p * bytes_per_peer, bytes_per_peer, /*ncclInt8*/ 0, p, comm, s));
        UMOE_NCCL(g_rccl.recv(reinterpret_cast<char*>(recv) + (size_t)p * bytes_per_peer, bytes_per_peer, 0, p, comm, s));
    }
    UMOE_NCCL(g_rccl.group_end());
    return 0;
}
