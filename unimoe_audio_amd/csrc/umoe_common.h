// Shared device/host helpers for libumoe_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "umoe.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;  // one MFMA 16x16x32 operand fragment
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define UMOE_MAXE 16  // router columns kept on lanes 0..15

// ---- error plumbing ----------------------------------------------------------------------
void umoe_set_error(const char* fmt, ...);
#define UMOE_HIP(expr)                                                                      \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            umoe_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return -2;                                                                      \
        }                                                                                   \
    } while (0)
#define UMOE_REQUIRE(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            umoe_set_error(__VA_ARGS__); \
            return -1;                   \
        }                                \
    } while (0)
#define UMOE_LAUNCH_CHECK() UMOE_HIP(hipGetLastError())

// ---- optional in-kernel timeline (diagnostic build only: make tl -> libumoe_hip_tl.so, scripts/timeline.py) -----------
// slots per kernel id: 0 = min entry over workgroups, 1 = max entry, 2 = max exit, 3.. = marks of workgroup (0,0,0)
#ifdef UMOE_TIMELINE
// buffer: [layer][kernel id][16 slots] of u64, then one word counting finished combine workgroups and one word holding
// the number of workgroups per combine launch
#define UMOE_TL_CTR (64 * 256)
static __device__ unsigned long long* g_tl;
// time stamps stay in registers until the kernel's exit: nothing but s_memrealtime is added to the measured path
struct tl_state { unsigned long long m[10]; };
__device__ __forceinline__ void tl_exit(const tl_state& st, int kid, unsigned long long t_exit = 0) {
    if (threadIdx.x != 0) return;
    const unsigned long long t = t_exit ? t_exit : wall_clock64();
    // layer = finished combine workgroups / workgroups per combine launch (second word, set by the harness)
    const int div = (int)g_tl[UMOE_TL_CTR + 1];
    int lay = (int)__atomic_load_n(&g_tl[UMOE_TL_CTR], __ATOMIC_RELAXED) / (div > 0 ? div : 1);
    lay = lay < 0 ? 0 : (lay > 63 ? 63 : lay);
    const int b = lay * 256 + kid * 16;
    atomicMin(&g_tl[b + 0], st.m[3]);
    atomicMax(&g_tl[b + 1], st.m[3]);
    atomicMax(&g_tl[b + 2], t);
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
#pragma unroll
        for (int k = 3; k < 10; ++k) g_tl[b + k] = st.m[k];
    }
    // per-workgroup dump of ONE kernel id in ONE layer (words 2 / 3 behind the counter: kernel id + 1, layer): entry, marks, exit,
    // and where the workgroup ran (XCC_ID, HW_ID)
    if ((int)g_tl[UMOE_TL_CTR + 2] == kid + 1 && (int)g_tl[UMOE_TL_CTR + 3] == lay) {
        const unsigned wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (wg < 1024) {
            unsigned long long* d = g_tl + UMOE_TL_CTR + 8 + wg * 12;
#pragma unroll
            for (int k = 3; k < 10; ++k) d[k - 3] = st.m[k];
            d[7] = t;
            d[8] = ((unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) << 32) | __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
            d[9] = 1;
        }
    }
    if (kid == 9) atomicAdd(&g_tl[UMOE_TL_CTR], 1ull);
}
#define UMOE_TL_SETTER(name) \
    extern "C" int umoe_tl_set_##name(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tl), &p, sizeof(p)); }
#define TL_ENTER(kid) tl_state tl_st; for (int tl_i = 0; tl_i < 10; ++tl_i) tl_st.m[tl_i] = 0; tl_st.m[3] = wall_clock64()
#define TL_MARK(kid, k) tl_st.m[k] = wall_clock64()
#define TL_EXIT(kid) tl_exit(tl_st, kid)
#define TL_PARAM , tl_state& tl_st
#define TL_PASS , tl_st
#else
#define UMOE_TL_SETTER(name)
#define TL_PARAM
#define TL_PASS
#define TL_ENTER(kid)
#define TL_MARK(kid, k)
#define TL_EXIT(kid)
#endif

// ---- bf16 <-> f32 (round to nearest even; NaN stays NaN) -----------------------------------
__host__ __device__ __forceinline__ float bf2f(uint16_t h) {
    union { uint32_t u; float f; } c;
    c.u = ((uint32_t)h) << 16;
    return c.f;
}
__host__ __device__ __forceinline__ uint16_t f2bf(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    // plain cast: hipcc emits v_cvt_pk_bf16_f32 (round to nearest even, NaN stays NaN; MI355X_MICROARCH.md)
    return __builtin_bit_cast(uint16_t, (__bf16)f);
#else
    union { uint32_t u; float f; } c;
    c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
#endif
}
__host__ __device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }
__device__ __forceinline__ float round_t(float v, int is_bf16) { return is_bf16 ? rbf(v) : v; }

// 8 bf16 <-> 8 floats via one 16-byte vector
struct alignas(16) bf16x8_raw { uint16_t v[8]; };
__device__ __forceinline__ uint4 ld16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void st16(void* p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }
__device__ __forceinline__ void unpack8(uint4 u, float* f) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 u;
    u.x = (uint32_t)f2bf(f[0]) | ((uint32_t)f2bf(f[1]) << 16);
    u.y = (uint32_t)f2bf(f[2]) | ((uint32_t)f2bf(f[3]) << 16);
    u.z = (uint32_t)f2bf(f[4]) | ((uint32_t)f2bf(f[5]) << 16);
    u.w = (uint32_t)f2bf(f[6]) | ((uint32_t)f2bf(f[7]) << 16);
    return u;
}

// ---- deterministic exp: the SAME fp64 fma sequence as oracle/router_oracle.c (exp_det) ------
// Written independently of the oracle source; the op sequence is the arithmetic contract
// documented in DESIGN.md ("router arithmetic").
__device__ __forceinline__ float umoe_exp_det(float xf) {
    if (!(xf > -110.0f)) return (xf != xf) ? xf : 0.0f;
    if (xf > 88.0f) xf = 88.0f + (xf - xf);
    const double x = (double)xf;
    const double k = rint(x * 1.4426950408889634074);
    double r = fma(k, -6.93147180369123816490e-01, x);
    r = fma(k, -1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const long long ki = (long long)k;
    const double scale = __longlong_as_double((ki + 1023) << 52);
    return (float)(p * scale);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// fixed-order sum over a 256-thread workgroup (sh: 4 floats of LDS)
#if defined(__HIPCC__)
__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}
#endif
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// side stream + events of the backward composites (umoe_bwd.hip): independent kernels of ONE call run beside each other, forked and joined
// inside the call (UMOE_BWD_OVERLAP=0: ok == false, everything on the caller's stream)
struct BwdSide {
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr, mid = nullptr, join = nullptr;
    bool ok = false;
};
BwdSide& bwd_side();

// ---- expert-parallel peer exchange (umoe_ep.hip; used by the decode engine) -----------------------------------------
// Region of one rank (uncached device memory, mapped by every peer through HIP IPC):
//   [flag block: 2 kinds x UMOE_MAX_EP tiles x UMOE_EP_PARTS words, one 64-byte line each][dispatch slab][return slab]
// A hand-off unit ("part") is ONE token row: part s of tile t carries row s of rank t (dispatch: one row of D; return: row s
// of each of the n_sub local experts' outputs), published by one flag word.
#define UMOE_EP_PARTS 16
#define UMOE_EP_FLAG_BYTES (2 * UMOE_MAX_EP * UMOE_EP_PARTS * 64)
struct umoe_ep_xfer {          // one push (local rows -> every peer's slab) or pull (own slab -> local buffer); by value
    char* peer_base[UMOE_MAX_EP];  // region base of every rank as mapped in this process
    const char* src;           // push: local source; the chunk for peer p starts at src + p * src_stride
    long src_stride;
    char* dst;                 // pull: local destination of tile p = dst + p * chunk (row-major), or the packed tiles (pull_pack)
    size_t chunk;              // bytes per tile = n_sub * rows * row_bytes
    size_t data_off;           // slab offset inside a region
    int kind;                  // 0 dispatch, 1 return: selects the flag block half
    int rank, size, loopback;
    int rows, row_bytes, n_sub;    // rows per tile (<= UMOE_EP_PARTS), bytes per row (multiple of 16), sub-blocks [n_sub][rows][row_bytes]
    const uint32_t* step;      // device word: decode steps taken so far
    int layer, layers;         // epoch = *step * layers + layer + 1
    uint32_t* err;             // device word, sticky: 1 = a receive timed out
    // return slab filled by umoe_moe_ep.hip: n_cwg > 0 = wait for COUNTERS (cumulative: word (kind 1, source rank, part 0) >=
    // ((*round - 1) * layers + layer + 1) * n_cwg, *round = decode steps taken incl. this one) instead of the row flags
    const uint32_t* round;
    int n_cwg;
};
int umoe_ep_push(const umoe_ep_xfer& x, hipStream_t s);
int umoe_ep_pull(const umoe_ep_xfer& x, hipStream_t s);
// pull of the dispatch slab that also re-lays every 16-row tile (the own one from `own_rows`, row-major) into MFMA operand order
// (WP16 of a [16][K = row_bytes / 2] matrix, include/umoe.h): x.dst = packed tiles [size][K * 16]
int umoe_ep_pull_pack(const umoe_ep_xfer& x, const uint16_t* own_rows, hipStream_t s);
// RMSNorm-only router launch (umoe_router_args.norm_only) that also pushes each normalised row to every peer (umoe_router.hip)
int umoe_router_norm_push(const umoe_router_args* a, const umoe_ep_xfer& x, hipStream_t s);
int umoe_ep_rccl_allgather(void* comm, const void* send, void* recv, size_t bytes, hipStream_t s);

// ---- riders publish the normalised rows inside the gate/up launch (umoe_gemm_args.rider_pub -> this, host side) ------------
// The row flags are REPLICATED: UMOE_FLAG_REPL copies of the 16-word line, 64 bytes apart; a rider stores its epoch into every copy,
// a polling workgroup reads copy (its id % UMOE_FLAG_REPL) -- 226 workgroups polling ONE line serialised at that line's home
// (scripts/timeline_wgs.py: 3 us from flag store to detection).
#define UMOE_FLAG_REPL 8
struct umoe_rider_pub {
    uint32_t* flags;           // device [UMOE_FLAG_REPL][16] words, one per token row and replica, monotonic epochs
    const uint32_t* step;      // device word: decode steps taken so far
    int layer, layers;         // epoch = *step * layers + layer + 1
    uint32_t* err;             // device word, sticky: 2 = a workgroup gave up waiting for the riders
    unsigned long long* rs;    // second hand-off form (fused expert launch with the RMSNorm prologue): device [16] granules {rs, epoch}
};

// The two expert GEMMs of a dense decode layer (gate/up SwiGLU with riders + rider_pub, then the down projections) in ONE launch
// (umoe_gemm.hip moe_fused_kernel; decode engine only).  `gu` as for umoe_grouped_gemm with nt 14, fused_router and rider_pub; `dn`
// with nt 6, dn->a == gu->out; `flags`: device words, one per gate/up workgroup of the launch box (>= num_groups * ceil(max pairs / 7)).
// Returns 1 (nothing launched) when the shapes do not allow the fusion: the caller then issues the two launches.
int umoe_moe_fused(const umoe_gemm_args* gu, const umoe_gemm_args* dn, uint32_t* flags, int flag_words, hipStream_t s);
// The same two GEMMs as ONE workgroup per CU with a static, byte-balanced schedule (umoe_moe_flat.hip): `n_wg` workgroups (<= the
// device's CU count: every workgroup must be resident), the riders are the first S of them.  `gu` / `dn` as for umoe_moe_fused (any
// nt); `flags`: >= n_wg device words.  Returns 1 (nothing launched) when the shapes or n_wg do not allow a schedule.
// `oproj` (optional): the o_proj + residual GEMM whose output rows ARE gu's raw rows (fused_router->x): it is computed inside the launch, half
// a 16-feature tile per workgroup, handed over through `o_flags` (8 x 256 device words, monotonic epochs); K = D = 2048 only.
int umoe_moe_flat(const umoe_gemm_args* gu, const umoe_gemm_args* dn, uint32_t* flags, int flag_words, int n_wg, hipStream_t s,
                  const umoe_gemm_args* oproj = nullptr, uint32_t* o_flags = nullptr);
bool umoe_moe_flat_feasible(int n_wg, int S, int D, int I_dyn, int I_sh, int n_real, int n_fix);

// A small decode GEMM with row riders in front (umoe_gemm.hip wstream_gemm_rk, umoe_riders_dev.h; decode engine only).  kind 2: the
// MoE combine of the previous layer rides in the QKV launch; kind 4: the same over the return slab of the expert-parallel exchange.
// Returns 1 (nothing launched) when the shapes do not fit.
struct umoe_rider2;
int umoe_gemm_riders(const umoe_gemm_args* a, int kind, const umoe_rider2* r2, const umoe_rider_pub* pub, hipStream_t s);

// ---- weight-streaming GEMM over several 16-row tiles per weight pass (umoe_gemm_mt.hip; expert parallel decode) --------
#define UMOE_MT_MAXG 4
#define UMOE_MT_MAXT UMOE_MAX_EP
struct umoe_mt_args {
    const uint16_t* w[UMOE_MT_MAXG];   // WP16 weights of group g = local expert (gate/up blocks interleaved for SwiGLU)
    int num_groups, n_blocks, k;       // 16-feature blocks per group (SwiGLU: 2*I/16), contraction length (k % 32 == 0)
    int tiles, n_rows;                 // row tiles per group (= ep_size: one per source rank), valid rows per tile (<= 16)
    const uint16_t* b;                 // activation tiles in MFMA operand order (WP16 of a [16][k] matrix): tile (g, t) at
    int b_group_tiles;                 //   b + (g * b_group_tiles + t) * 16 * k; 0 = every group reads the same tiles (gate/up)
    uint16_t* h_out;                   // UMOE_EPI_SWIGLU: tile (g, t) of silu(g)*u in operand order for the down projection,
                                       //   h_out + (g * tiles + t) * 16 * I, I = n_blocks * 8
    uint16_t* y_out[UMOE_MT_MAXG][UMOE_MT_MAXT];   // UMOE_EPI_BF16: row-major [16][ldo] output of tile (g, t)
    int ldo;
    int epilogue;
    const umoe_router_args* fused_router;   // optional HOST pointer: S router workgroups ride as an extra z-slice (see umoe_gemm_args)
};
int umoe_gemm_mt(const umoe_mt_args* a, hipStream_t s);

// ---- the MoE half of an expert-parallel decode layer in ONE launch (umoe_moe_ep.hip; decode engine only) ------------------------
#define UMOE_EPF_MAXT 24       // task words per workgroup (the last one stays 0 = end)
struct umoe_epf_desc {
    const umoe_router_args* router;    // x = raw rows x1 [S][D] (post-attention residual stream), norm_w / rms_eps = post-attention RMSNorm, gate, tables
    const umoe_rider_pub* pub;         // step / layer / layers / err (its row flags are not used)
    int R, rank, loopback, E_loc, S, D, I_dyn, I_sh, n_fix, n_wg;
    char* const* peer_base;            // [R] exchange regions as mapped here
    size_t disp_off, ret_off;          // dispatch slab [R][S][D] (raw rows), return slab [n_real][16][D] inside a region
    const uint16_t* const* w_lgu;      // [E_loc] WP16 gate/up, [E_loc] WP16 down of the local experts
    const uint16_t* const* w_ldn;
    const uint16_t* const* w_sgu;      // [n_fix] shared experts
    const uint16_t* const* w_sdn;
    uint16_t* xgp;                     // [R][16 * D]: normalised row tiles in MFMA operand order
    uint16_t* hpk;                     // [E_loc][R][16 * I_dyn]: silu(g)*u tiles in operand order
    uint16_t* h_sh; int ldh, h_row0;   // shared experts' silu(g)*u rows: expert i at rows h_row0 + i*S of h_sh [.][ldh]
    uint16_t* y_sh; int ldy, y_row0;   // shared experts' outputs: expert i at rows y_row0 + i*S of y_sh [.][ldy]
    uint32_t* flags; int flag_words;   // >= 2 * n_wg + UMOE_FLAG_REPL * 16 device words (monotonic epochs)
    const uint32_t* round;             // device word: decode steps taken (no prefill bumps): base of the return counters' rounds
    uint32_t* tasks_dev;               // n_wg * UMOE_EPF_MAXT device words (umoe_moe_ep_prepare fills them)
    // filled by umoe_moe_ep_prepare
    int n_cwg;                         // workgroups that count themselves in on every return slab per round
    int proda_base[4], proda_n[4], prodb_base[UMOE_MT_MAXG], prodb_n[UMOE_MT_MAXG];
};
int umoe_moe_ep_prepare(umoe_epf_desc* d, hipStream_t s);     // 0 ok, 1 no plan for the shape; NOT inside a stream capture
int umoe_moe_ep(const umoe_epf_desc* d, hipStream_t s);

// device side of the hand-off (shared by umoe_ep.hip, umoe_router.hip, umoe_misc.hip)
#if defined(__HIPCC__)
typedef __attribute__((address_space(1))) uint32_t umoe_gu32;
#define UMOE_SYS_AUX 17   // raw buffer aux bits on gfx950: sc0 (1) | sc1 (16) = system scope
__device__ __forceinline__ umoe_gu32* umoe_ep_flag(char* base, int kind, int tile, int part) {
    return reinterpret_cast<umoe_gu32*>(reinterpret_cast<uintptr_t>(base + ((size_t)(kind * UMOE_MAX_EP + tile) * UMOE_EP_PARTS + part) * 64));
}
__device__ __forceinline__ uint32_t umoe_ep_epoch(const umoe_ep_xfer& x) { return *x.step * (uint32_t)x.layers + (uint32_t)x.layer + 1u; }
// ONE lane: bounded wait until `flag` carries `epoch` (or a later one).  A peer that never arrives, or a timeout anywhere
// earlier (sticky error word), ends the wait: every wave reaches its exit.
__device__ __forceinline__ void umoe_ep_wait(umoe_gu32* flag, uint32_t epoch, uint32_t* err_word) {
    umoe_gu32* err = reinterpret_cast<umoe_gu32*>(reinterpret_cast<uintptr_t>(err_word));
    const unsigned long long t0 = wall_clock64();        // 100 MHz
    for (unsigned spins = 0;; ++spins) {
        const uint32_t v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int32_t)(v - epoch) >= 0) break;
        __builtin_amdgcn_s_sleep(2);
        if ((spins & 255u) == 255u) {
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            if (wall_clock64() - t0 > 1000000000ull) {    // 10 s
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
}
#endif
