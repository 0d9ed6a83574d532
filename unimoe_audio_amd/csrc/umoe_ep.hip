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


namespace {
typedef int (*fn_allgather)(const void*, void*, size_t, int, void*, hipStream_t);
fn_allgather g_allgather = nullptr;
}  // namespace

int umoe_ep_rccl_allgather(void* comm, const void* send, void* recv, size_t bytes, hipStream_t s) {
    UMOE_REQUIRE(comm && send && recv, "umoe_ep_rccl_allgather: null argument");
    if (int rc = load_rccl()) return rc;
    if (!g_allgather) g_allgather = (fn_allgather)dlsym(g_rccl.h, "ncclAllGather");
    UMOE_REQUIRE(g_allgather, "umoe_ep: librccl lacks ncclAllGather");
    UMOE_NCCL(g_allgather(send, recv, bytes, /*ncclInt8*/ 0, comm, s));
    return 0;
}

// ------------------------------------------------------------------------------------------------ HIP IPC helpers
extern "C" int umoe_ep_ipc_export(const void* dev_ptr, void* handle64_out) {
    UMOE_REQUIRE(dev_ptr && handle64_out, "umoe_ep_ipc_export: null argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    hipIpcMemHandle_t h;
    UMOE_HIP(hipIpcGetMemHandle(&h, const_cast<void*>(dev_ptr)));
    memcpy(handle64_out, &h, sizeof(h));
    return 0;
}

extern "C" int umoe_ep_ipc_open(const void* handle64, void** dev_ptr_out) {
    UMOE_REQUIRE(handle64 && dev_ptr_out, "umoe_ep_ipc_open: null argument");
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    UMOE_HIP(hipIpcOpenMemHandle(dev_ptr_out, h, hipIpcMemLazyEnablePeerAccess));
    return 0;
}

extern "C" int umoe_ep_ipc_close(void* dev_ptr) {
    if (!dev_ptr) return 0;
    UMOE_HIP(hipIpcCloseMemHandle(dev_ptr));
    return 0;
}

// ------------------------------------------------------------------------------------------------ peer push / pull
// Hand-off form (cdna_hip_programming.md, Guideline 16 R1, at SYSTEM scope because the consumer is another GPU): every payload
// byte is a 16-byte write-through store (sc0 sc1), every storing wave drains (s_waitcnt vmcnt(0)), the workgroup meets at its
// barrier, ONE lane publishes the epoch with a system-scope store; the consumer polls that word with a system-scope load (one
// lane, bounded), the other waves wait at the barrier, and EVERY load of the payload is an sc0 sc1 load -- no stale line of any
// cache level can be read, whatever the memory type of the region turns out to be on the peer.
typedef __attribute__((address_space(1))) uint32_t gu32;
#define UMOE_SYS_AUX 17   // raw buffer aux bits on gfx950: sc0 (1) | sc1 (16) = system scope

__device__ __forceinline__ gu32* ep_flag(char* base, int kind, int tile, int part) {
    return reinterpret_cast<gu32*>(reinterpret_cast<uintptr_t>(base + ((size_t)(kind * UMOE_MAX_EP + tile) * UMOE_EP_PARTS + part) * 64));
}

__global__ __launch_bounds__(256) void ep_push_kernel(const umoe_ep_xfer x) {
    const int part = blockIdx.x, j = blockIdx.y, tid = threadIdx.x;
    const int p = (x.rank + 1 + j) % x.size;                 // destination rank
    const int tile = x.loopback ? p : x.rank;                // where my rows live in its slab
    const uint32_t epoch = *x.step * (uint32_t)x.layers + (uint32_t)x.layer + 1u;
    const size_t per = x.chunk / UMOE_EP_PARTS;
    const char* src = x.src + (long)p * x.src_stride + (size_t)part * per;
    char* dst = x.peer_base[p] + x.data_off + (size_t)tile * x.chunk + (size_t)part * per;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)per, 0x00020000);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    for (size_t i = (size_t)tid * 16; i < per; i += 256 * 16) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + i);
        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)i, 0, UMOE_SYS_AUX);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every storing wave drains its write-through stores
    __syncthreads();
    if (tid == 0) __hip_atomic_store(ep_flag(x.peer_base[p], x.kind, tile, part), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void ep_pull_kernel(const umoe_ep_xfer x) {
    const int part = blockIdx.x, j = blockIdx.y, tid = threadIdx.x;
    const int p = (x.rank + 1 + j) % x.size;                 // source rank = tile index in my slabs
    const uint32_t epoch = *x.step * (uint32_t)x.layers + (uint32_t)x.layer + 1u;
    char* own = x.peer_base[x.rank];
    if (tid == 0) {
        gu32* f = ep_flag(own, x.kind, p, part);
        gu32* err = reinterpret_cast<gu32*>(reinterpret_cast<uintptr_t>(x.err));
        const unsigned long long t0 = wall_clock64();        // 100 MHz
        for (unsigned spins = 0;; ++spins) {
            const uint32_t v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((int32_t)(v - epoch) >= 0) break;
            __builtin_amdgcn_s_sleep(8);
            if ((spins & 255u) == 255u) {
                // exit condition every wave reaches: a peer that never arrives (or an earlier timeout anywhere) ends the wait
                if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                if (wall_clock64() - t0 > 1000000000ull) {    // 10 s
                    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
    }
    __syncthreads();
    const size_t per = x.chunk / UMOE_EP_PARTS;
    char* src = own + x.data_off + (size_t)p * x.chunk + (size_t)part * per;
    char* dst = x.dst + (size_t)p * x.chunk + (size_t)part * per;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(src, 0, (int)per, 0x00020000);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    for (size_t i = (size_t)tid * 16; i < per; i += 256 * 16) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)i, 0, UMOE_SYS_AUX);
        *reinterpret_cast<u32x4*>(dst + i) = v;
    }
}

int umoe_ep_push(const umoe_ep_xfer& x, hipStream_t s) {
    UMOE_REQUIRE(x.size >= 2 && x.size <= UMOE_MAX_EP && x.chunk % (UMOE_EP_PARTS * 16) == 0 && x.chunk / UMOE_EP_PARTS < (1u << 30),
                 "umoe_ep_push: bad geometry (size %d, chunk %zu)", x.size, x.chunk);
    ep_push_kernel<<<dim3(UMOE_EP_PARTS, (unsigned)(x.size - 1)), 256, 0, s>>>(x);
    UMOE_LAUNCH_CHECK();
    return 0;
}

int umoe_ep_pull(const umoe_ep_xfer& x, hipStream_t s) {
    UMOE_REQUIRE(x.size >= 2 && x.size <= UMOE_MAX_EP && x.chunk % (UMOE_EP_PARTS * 16) == 0 && x.chunk / UMOE_EP_PARTS < (1u << 30),
                 "umoe_ep_pull: bad geometry (size %d, chunk %zu)", x.size, x.chunk);
    ep_pull_kernel<<<dim3(UMOE_EP_PARTS, (unsigned)(x.size - 1)), 256, 0, s>>>(x);
    UMOE_LAUNCH_CHECK();
    return 0;
}
