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
// One workgroup moves one part = row s of every sub-block of one tile; all of a thread's loads are in flight before its stores.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define EP_MAXC 8   // 16-byte chunks per thread: n_sub * row_bytes / 16 / 256 <= 8 (4 local experts x D 4096)

__global__ __launch_bounds__(256) void ep_push_kernel(const umoe_ep_xfer x) {
    const int srow = blockIdx.x, j = blockIdx.y, tid = threadIdx.x;
    const int p = (x.rank + 1 + j) % x.size;                 // destination rank
    const int tile = x.loopback ? p : x.rank;                // where my rows live in its slab
    const uint32_t epoch = umoe_ep_epoch(x);
    const int cpr = x.row_bytes >> 4, total = x.n_sub * cpr; // chunks per row, chunks of this part
    const char* src = x.src + (long)p * x.src_stride;
    char* dst = x.peer_base[p] + x.data_off + (size_t)tile * x.chunk;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)x.chunk, 0x00020000);
    u32x4 v[EP_MAXC];
#pragma unroll
    for (int k = 0; k < EP_MAXC; ++k) {
        const int c = tid + k * 256;
        if (c < total) v[k] = *reinterpret_cast<const u32x4*>(src + ((size_t)(c / cpr) * x.rows + srow) * x.row_bytes + (size_t)(c % cpr) * 16);
    }
#pragma unroll
    for (int k = 0; k < EP_MAXC; ++k) {
        const int c = tid + k * 256;
        if (c < total)
            __builtin_amdgcn_raw_buffer_store_b128(v[k], rsrc, (int)(((size_t)(c / cpr) * x.rows + srow) * x.row_bytes + (size_t)(c % cpr) * 16), 0, UMOE_SYS_AUX);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every storing wave drains its write-through stores
    __syncthreads();
    if (tid == 0) __hip_atomic_store(umoe_ep_flag(x.peer_base[p], x.kind, tile, srow), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// PACK: the destination tile is written in MFMA operand order (dispatch slab only: n_sub 1); blockIdx.y == size - 1 packs the own rows
template <bool PACK>
__global__ __launch_bounds__(256) void ep_pull_kernel(const umoe_ep_xfer x, const uint16_t* own_rows) {
    const int srow = blockIdx.x, j = blockIdx.y, tid = threadIdx.x;
    const bool own = PACK && j == x.size - 1;
    const int p = own ? x.rank : (x.rank + 1 + j) % x.size;  // source rank = tile index in my slabs
    char* region = x.peer_base[x.rank];
    if (!own) {
        if (tid == 0) umoe_ep_wait(umoe_ep_flag(region, x.kind, p, srow), umoe_ep_epoch(x), x.err);
        __syncthreads();
    }
    const int cpr = x.row_bytes >> 4, total = x.n_sub * cpr;
    char* src = region + x.data_off + (size_t)p * x.chunk;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(src, 0, (int)x.chunk, 0x00020000);
    u32x4 v[EP_MAXC];
#pragma unroll
    for (int k = 0; k < EP_MAXC; ++k) {
        const int c = tid + k * 256;
        if (c < total) {
            const size_t off = ((size_t)(c / cpr) * x.rows + srow) * x.row_bytes + (size_t)(c % cpr) * 16;
            if (own) v[k] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(own_rows) + off);
            else v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, UMOE_SYS_AUX);
        }
    }
#pragma unroll
    for (int k = 0; k < EP_MAXC; ++k) {
        const int c = tid + k * 256;
        if (c < total) {
            size_t off = ((size_t)(c / cpr) * x.rows + srow) * x.row_bytes + (size_t)(c % cpr) * 16;
            if (PACK) {   // chunk c of row srow: K-quarter q, k-step i  ->  fragment (i, lane = q*16 + row)
                const int q4 = cpr >> 2, q = c / q4, i = c % q4;
                off = ((size_t)i * 64 + q * 16 + srow) * 16;
            }
            // (operand-order tiles are 16 rows apart whatever the row count is)
            *reinterpret_cast<u32x4*>(x.dst + (size_t)p * (PACK ? (size_t)16 * x.row_bytes : x.chunk) + off) = v[k];
        }
    }
}

static int xfer_check(const umoe_ep_xfer& x, const char* who) {
    UMOE_REQUIRE(x.size >= 2 && x.size <= UMOE_MAX_EP && x.rows >= 1 && x.rows <= UMOE_EP_PARTS && x.row_bytes % 16 == 0 && x.n_sub >= 1 &&
                     x.n_sub * (x.row_bytes / 16) <= 256 * EP_MAXC && x.chunk == (size_t)x.n_sub * x.rows * x.row_bytes && x.chunk < (1u << 30),
                 "%s: bad geometry (size %d rows %d row_bytes %d n_sub %d chunk %zu)", who, x.size, x.rows, x.row_bytes, x.n_sub, x.chunk);
    return 0;
}

int umoe_ep_push(const umoe_ep_xfer& x, hipStream_t s) {
    if (int rc = xfer_check(x, "umoe_ep_push")) return rc;
    ep_push_kernel<<<dim3((unsigned)x.rows, (unsigned)(x.size - 1)), 256, 0, s>>>(x);
    UMOE_LAUNCH_CHECK();
    return 0;
}

int umoe_ep_pull(const umoe_ep_xfer& x, hipStream_t s) {
    if (int rc = xfer_check(x, "umoe_ep_pull")) return rc;
    ep_pull_kernel<false><<<dim3((unsigned)x.rows, (unsigned)(x.size - 1)), 256, 0, s>>>(x, nullptr);
    UMOE_LAUNCH_CHECK();
    return 0;
}

int umoe_ep_pull_pack(const umoe_ep_xfer& x, const uint16_t* own_rows, hipStream_t s) {
    if (int rc = xfer_check(x, "umoe_ep_pull_pack")) return rc;
    UMOE_REQUIRE(x.n_sub == 1 && own_rows && (x.row_bytes / 16) % 4 == 0, "umoe_ep_pull_pack: one sub-block, K %% 32 == 0");
    ep_pull_kernel<true><<<dim3((unsigned)x.rows, (unsigned)x.size), 256, 0, s>>>(x, own_rows);
    UMOE_LAUNCH_CHECK();
    return 0;
}
