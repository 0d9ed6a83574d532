// Top-P dynamic router + ragged dispatch tables.
//
// Replaces the reference's per-k Python loop of tiny torch kernels and its >= 15 host syncs per
// layer (utils/UniMoE_Audio_core.py:246-291, 94-167, 178-193; dispatch: utils/UniMoE_Audio_utils.py
// :436-523) with two launches and no host sync.
//
// router_kernel: one 64-lane wave per token (decode: one wave per workgroup, so each token's serial
// chain owns a SIMD).  The gate GEMV (11 dot products of length D) is spread over the wave and
// reduced with a fixed reduce-scatter tree; afterwards lane e owns router column e and the Top-P
// count / iterative arg-max mixer run on v_readlane gathers with compile-time sizes --
// every fp32 operation happens in the order fixed by the arithmetic contract of
// oracle/router_oracle.c (sequential softmax sum, reciprocal multiply, deterministic exp, cumsum
// accumulator type, thresholds cast to T), so integer outputs are bit-exact given equal logits.
//
// dispatch_kernel: one workgroup per routed expert; wavefront ballot + popcount prefix sums give
// every (token, expert) pair its slot in token order -- no [S,E,D] expansion, no padding to the
// max capacity, no argsort.
// Roofline: latency-bound at decode (S = 16); HBM-bound on [S,D] reads for large S.
#include "umoe_common.h"

#include "umoe_router_dev.h"

template <int ND, int NF, int TB>
__global__ __launch_bounds__(256) void router_kernel(const umoe_router_args a) {
    TL_ENTER(5);
    const int s = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (s >= a.S) return;
    route_token<ND, NF, TB>(a, s, threadIdx.x & 63 TL_PASS);
    TL_EXIT(5);
}
UMOE_TL_SETTER(router)

// Decode shapes (S <= 256, D = 2048 or 4096): one 256-thread workgroup per token.  The four waves split the RMSNorm and
// the gate GEMV over D (13 loads per lane instead of 52, a quarter of the VALU work each), partial sums meet in LDS in
// fixed order, wave 0 then walks the serial routing chain.
template <int ND, int NF, int TB>
__global__ __launch_bounds__(256) void router_kernel4(const umoe_router_args a) {
    __shared__ float lds[4 + 4 * UMOE_MAXE];
    TL_ENTER(5);
    router4_body<ND, NF, TB, false>(a, blockIdx.x, threadIdx.x, lds TL_PASS);
    TL_EXIT(5);
}
// RMSNorm only (umoe_router_args.norm_only): h_out = norm_w * bf16(x * rsqrt(mean(x^2) + eps)), the router's own arithmetic
template <int ND, int NF, int TB>
__global__ __launch_bounds__(256) void router_norm_kernel(const umoe_router_args a) {
    __shared__ float lds[4 + 4 * UMOE_MAXE];
    TL_ENTER(5);
    router4_body<ND, NF, TB, true>(a, blockIdx.x, threadIdx.x, lds TL_PASS);
    TL_EXIT(5);
}

// Expert parallel decode: the RMSNorm-only launch also PUSHES its row to every peer's dispatch slab (first exchange of
// AudioMOELayer.forward, core.py:467, in its dense form) -- each thread re-reads the 16 bytes it just stored and writes them
// through to the peers (sc0 sc1), every wave drains, lanes 0..size-2 publish the row's epoch, one peer each (umoe_common.h).
__global__ __launch_bounds__(256) void router_norm_push_kernel(const umoe_router_args a, const umoe_ep_xfer x) {
    __shared__ float lds[4 + 4 * UMOE_MAXE];
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    TL_ENTER(5);
    router4_body<9, 2, 1, true>(a, s, tid, lds TL_PASS);
    const int nch = a.D >> 11;
    const uint32_t epoch = umoe_ep_epoch(x);
    u32x4 v[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
        if (n < nch) v[n] = *reinterpret_cast<const u32x4*>(a.h_out + (size_t)s * a.D + ((n * 4 + wave) * 64 + lane) * 8);   // own stores
    for (int j = 0; j + 1 < x.size; ++j) {
        const int p = (x.rank + 1 + j) % x.size;
        const int tile = x.loopback ? p : x.rank;
        char* dst = x.peer_base[p] + x.data_off + (size_t)tile * x.chunk + (size_t)s * x.row_bytes;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, x.row_bytes, 0x00020000);
#pragma unroll
        for (int n = 0; n < 2; ++n)
            if (n < nch) __builtin_amdgcn_raw_buffer_store_b128(v[n], rsrc, ((n * 4 + wave) * 64 + lane) * 16, 0, UMOE_SYS_AUX);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid + 1 < x.size) {
        const int p = (x.rank + 1 + tid) % x.size;
        __hip_atomic_store(umoe_ep_flag(x.peer_base[p], x.kind, x.loopback ? p : x.rank, s), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int umoe_router_norm_push(const umoe_router_args* a, const umoe_ep_xfer& x, hipStream_t s) {
    UMOE_REQUIRE(a && a->x && a->norm_w && a->h_out && (a->D == 2048 || a->D == 4096) && a->S >= 1 && a->S <= UMOE_EP_PARTS,
                 "umoe_router_norm_push: needs x, norm_w, h_out, D 2048 / 4096, S <= %d", UMOE_EP_PARTS);
    UMOE_REQUIRE(x.size >= 2 && x.size <= UMOE_MAX_EP && x.rows == a->S && x.row_bytes == a->D * 2 && x.n_sub == 1 && x.kind == 0,
                 "umoe_router_norm_push: exchange geometry does not match the rows");
    router_norm_push_kernel<<<dim3((unsigned)a->S), 256, 0, s>>>(*a, x);
    UMOE_LAUNCH_CHECK();
    return 0;
}

extern "C" int umoe_router_fwd(const umoe_router_args* a, umoe_stream_t stream);
extern "C" int umoe_dispatch_build(const int32_t*, int, int, int, int32_t*, int32_t*, int32_t*, int32_t*, umoe_stream_t);

template <int ND, int NF>
static void launch_router(const umoe_router_args* a, dim3 grid, hipStream_t st) {
    (void)grid;
    if constexpr (ND > 0) {
        if (a->S <= 256 && !a->logits_in && !a->x_noise && (a->D == 2048 || a->D == 4096)) {
            if (a->logits_bf16) router_kernel4<ND, NF, 1><<<dim3((unsigned)a->S), 256, 0, st>>>(*a);
            else router_kernel4<ND, NF, 0><<<dim3((unsigned)a->S), 256, 0, st>>>(*a);
            return;
        }
    }
    // few tokens (decode): one wave per workgroup so every token's serial routing chain owns a SIMD on its own CU
    const unsigned threads = a->S <= 256 ? 64u : 256u;
    grid = dim3((unsigned)ceil_div(a->S, (int)(threads / 64)));
    if (a->logits_bf16) router_kernel<ND, NF, 1><<<grid, threads, 0, st>>>(*a);
    else router_kernel<ND, NF, 0><<<grid, threads, 0, st>>>(*a);
}
static int router_check(const umoe_router_args* a) {
    UMOE_REQUIRE(a && a->expert_mask, "umoe_router_fwd: null argument");
    const int E = a->n_dyn + a->n_fix;
    UMOE_REQUIRE(a->S >= 0 && a->n_dyn >= 1 && E <= UMOE_MAXE && a->n_real <= a->n_dyn,
                 "umoe_router_fwd: bad sizes S=%d n_dyn=%d n_real=%d n_fix=%d (E <= %d)", a->S, a->n_dyn, a->n_real,
                 a->n_fix, UMOE_MAXE);
    UMOE_REQUIRE(a->logits_in || (a->x && a->gate_w && a->D > 0 && a->D % 8 == 0),
                 "umoe_router_fwd: need logits_in or (x, gate_w, D %% 8 == 0)");
    UMOE_REQUIRE(!a->gumbel || a->rand_u, "umoe_router_fwd: the training branch of the mixer needs both noise tensors (gumbel, rand_u)");
    UMOE_REQUIRE(!a->x_noise || (a->x && !a->logits_in && !a->norm_w && !a->h_out && !a->norm_only),
                 "umoe_router_fwd: x_noise (input jitter) goes with (x, gate_w) only: no logits_in / norm_w / h_out / norm_only");
    return 0;
}

// S <= 16: ragged dispatch tables by ballot + popcount, 16 lanes (tokens) per routed expert, token order preserved
__global__ __launch_bounds__(256) void dispatch_small_kernel(const int32_t* __restrict__ mask, int S, int ld, int n_real,
                                                             int32_t* counts, int32_t* offsets, int32_t* slot_token,
                                                             int32_t* slot_of) {
    __shared__ int cnt_s[UMOE_MAXE];
    TL_ENTER(6);
    const int e = threadIdx.x >> 4, t = threadIdx.x & 15;
    const bool mine = e < n_real;
    const bool on = mine && t < S && mask[(size_t)t * ld + e] != 0;
    const unsigned long long bal = __ballot(on);
    const unsigned bits = (unsigned)((bal >> (16 * (e & 3))) & 0xffffull);
    const int cnt = __popc(bits), pos = __popc(bits & ((1u << t) - 1u));
    if (mine && t == 0) cnt_s[e] = cnt;
    __syncthreads();
    if (mine) {
        int off = 0;
        for (int ee = 0; ee < e; ++ee) off += cnt_s[ee];
        if (t < S) slot_of[(size_t)t * n_real + e] = on ? off + pos : -1;
        if (on) slot_token[off + pos] = t;
        if (t == 0) {
            counts[e] = cnt;
            offsets[e] = off;
            if (e == n_real - 1) offsets[n_real] = off + cnt;
        }
    }
    TL_EXIT(6);
}

// router + dispatch tables (decode, S <= 16: one-wave-per-token router + a 256-thread ballot dispatch)
extern "C" int umoe_router_dispatch_fwd(const umoe_router_args* a, int32_t* counts, int32_t* offsets, int32_t* slot_token,
                                        int32_t* slot_of, umoe_stream_t stream) {
    if (int rc = router_check(a)) return rc;
    UMOE_REQUIRE(counts && offsets && slot_token && slot_of, "umoe_router_dispatch_fwd: null table");
    if (a->S == 0) return 0;
    if (int rc = umoe_router_fwd(a, stream)) return rc;
    if (a->S > 16)
        return umoe_dispatch_build(a->expert_mask, a->S, a->n_dyn + a->n_fix, a->n_real, counts, offsets, slot_token, slot_of, stream);
    dispatch_small_kernel<<<1, 256, 0, (hipStream_t)stream>>>(a->expert_mask, a->S, a->n_dyn + a->n_fix, a->n_real, counts, offsets,
                                                              slot_token, slot_of);
    UMOE_LAUNCH_CHECK();
    return 0;
}

extern "C" int umoe_router_fwd(const umoe_router_args* a, umoe_stream_t stream) {
    if (a && a->norm_only) {
        UMOE_REQUIRE(a->x && a->norm_w && a->h_out && a->S > 0 && a->S <= 256 && (a->D == 2048 || a->D == 4096),
                     "umoe_router_fwd(norm_only): need x, norm_w, h_out, S <= 256, D 2048 or 4096 (S=%d D=%d)", a->S, a->D);
        router_norm_kernel<9, 2, 1><<<dim3((unsigned)a->S), 256, 0, (hipStream_t)stream>>>(*a);   // (ND / NF / TB do not enter the norm)
        UMOE_LAUNCH_CHECK();
        return 0;
    }
    if (int rc = router_check(a)) return rc;
    if (a->S == 0) return 0;
    const dim3 grid((unsigned)ceil_div(a->S, 4));
    hipStream_t st = (hipStream_t)stream;
    // the shipped shapes (utils/config.json: 8 routed + 1 null + 2 shared) get fully unrolled, guard-free code
    if (a->n_dyn == 9 && a->n_fix == 2) launch_router<9, 2>(a, grid, st);
    else if (a->n_dyn == 8 && a->n_fix == 2) launch_router<8, 2>(a, grid, st);
    else launch_router<0, 0>(a, grid, st);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ dispatch
// grid = n_real workgroups of 256.  Every workgroup counts all experts (to know its own offset),
// then assigns slots for its expert in token order with ballot/popcount prefix sums.
// `align` (power of two): every expert's slot range starts on a multiple of it (training: the transposed slot buffers
// of the weight-gradient GEMMs are read in 16-byte chunks); the padding slots point at token 0 and are never counted
__global__ __launch_bounds__(256) void dispatch_kernel(const int32_t* __restrict__ mask, int S, int ld, int n_real, int align,
                                                       int32_t* counts, int32_t* offsets, int32_t* slot_token,
                                                       int32_t* slot_of) {
    __shared__ int wave_cnt[4];
    __shared__ int tot[UMOE_MAXE];
    const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < UMOE_MAXE) tot[tid] = 0;
    __syncthreads();
    // pass 1: counts of experts < e (offset) and of e itself -- one atomic per wave per expert
    for (int ee = 0; ee <= e; ++ee) {
        int c = 0;
        for (int s = tid; s < S; s += 256) c += (mask[(size_t)s * ld + ee] != 0);
        c = (int)wave_sum((float)c);  // exact for counts < 2^24
        if (lane == 0) atomicAdd(&tot[ee], c);
    }
    __syncthreads();
    int off = 0;
    for (int ee = 0; ee < e; ++ee) off += (tot[ee] + align - 1) & ~(align - 1);
    const int padded = (tot[e] + align - 1) & ~(align - 1);
    if (tid == 0) {
        counts[e] = tot[e];
        offsets[e] = off;
        if (e == n_real - 1) offsets[n_real] = off + padded;
    }
    if (tid < padded - tot[e]) slot_token[off + tot[e] + tid] = 0;
    // pass 2: slots in token order
    int base = off;
    for (int s0 = 0; s0 < S; s0 += 256) {
        const int s = s0 + tid;
        const bool on = (s < S) && (mask[(size_t)s * ld + e] != 0);
        const unsigned long long b = __ballot(on);
        const int before = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(b);
        __syncthreads();
        int wbase = 0;
        for (int w2 = 0; w2 < wave; ++w2) wbase += wave_cnt[w2];
        const int chunk_total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        if (s < S) {
            const int slot = on ? base + wbase + before : -1;
            slot_of[(size_t)s * n_real + e] = slot;
            if (on) slot_token[slot] = s;
        }
        base += chunk_total;
        __syncthreads();
    }
}

extern "C" int umoe_dispatch_build(const int32_t* expert_mask, int S, int ld_mask, int n_real, int32_t* counts,
                                   int32_t* offsets, int32_t* slot_token, int32_t* slot_of, umoe_stream_t stream) {
    UMOE_REQUIRE(expert_mask && counts && offsets && slot_token && slot_of, "umoe_dispatch_build: null argument");
    UMOE_REQUIRE(n_real >= 1 && n_real <= UMOE_MAXE && ld_mask >= n_real && S >= 0 && S < (1 << 24),
                 "umoe_dispatch_build: bad sizes S=%d n_real=%d ld=%d", S, n_real, ld_mask);
    dispatch_kernel<<<dim3((unsigned)n_real), 256, 0, (hipStream_t)stream>>>(expert_mask, S, ld_mask, n_real, 1, counts,
                                                                            offsets, slot_token, slot_of);
    UMOE_LAUNCH_CHECK();
    return 0;
}

extern "C" int umoe_dispatch_build_aligned(const int32_t* expert_mask, int S, int ld_mask, int n_real, int align,
                                           int32_t* counts, int32_t* offsets, int32_t* slot_token, int32_t* slot_of,
                                           umoe_stream_t stream) {
    UMOE_REQUIRE(expert_mask && counts && offsets && slot_token && slot_of, "umoe_dispatch_build_aligned: null argument");
    UMOE_REQUIRE(n_real >= 1 && n_real <= UMOE_MAXE && ld_mask >= n_real && S >= 0 && S < (1 << 24),
                 "umoe_dispatch_build_aligned: bad sizes S=%d n_real=%d ld=%d", S, n_real, ld_mask);
    UMOE_REQUIRE(align >= 1 && align <= 256 && (align & (align - 1)) == 0, "umoe_dispatch_build_aligned: align must be a power of two <= 256");
    dispatch_kernel<<<dim3((unsigned)n_real), 256, 0, (hipStream_t)stream>>>(expert_mask, S, ld_mask, n_real, align, counts,
                                                                            offsets, slot_token, slot_of);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ permute
__global__ __launch_bounds__(256) void permute_kernel(const uint16_t* __restrict__ x, int D, const int32_t* slot_token,
                                                      const int32_t* total, uint16_t* __restrict__ out) {
    const int slot = blockIdx.x;
    if (slot >= *total) return;
    const uint16_t* src = x + (size_t)slot_token[slot] * D;
    uint16_t* dst = out + (size_t)slot * D;
    for (int c = threadIdx.x; c < (D >> 3); c += 256) st16(dst + c * 8, ld16(src + c * 8));
}

extern "C" int umoe_permute_fwd(const uint16_t* x, int D, const int32_t* slot_token, const int32_t* total_slots,
                                int max_slots, uint16_t* out, umoe_stream_t stream) {
    UMOE_REQUIRE(x && slot_token && total_slots && out && D % 8 == 0, "umoe_permute_fwd: bad argument");
    if (max_slots <= 0) return 0;
    permute_kernel<<<dim3((unsigned)max_slots), 256, 0, (hipStream_t)stream>>>(x, D, slot_token, total_slots, out);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ token drop
// Replaces the token-drop branch of UniMoEAudioSparseMoeBlock.forward (core.py:302-329; capacity :170-175):
//   "probs"    per dynamic column keep the `capacity` selected tokens with the largest logits (reference: topk(dim=0) over the
//              column with unselected entries filled with finfo.min, :305-314);
//   "position" per column -- ALL E columns, the shared ones too, as the reference's cumsum over the whole mask does -- keep the
//              first `capacity` selected tokens in token order (:321-323);
// then routing weights are zeroed on dropped entries and renormalised (:328-329) and the global weights recomputed over the
// kept columns (:178-193).  Two launches: one workgroup per column selects (radix select on the order-preserving integer image
// of the logit + index-ordered tie break, or a prefix count), one wave per token finishes with the router's lane-per-column
// arithmetic.  torch.topk leaves the order among EQUAL values unspecified; here ties at the capacity boundary keep the lowest
// token indices (identical to the reference whenever the capacity-th and the next logit of a column differ).
__device__ __forceinline__ uint32_t drop_key(const void* logits, int is_bf16, size_t idx) {
    if (is_bf16) {
        const uint32_t u = reinterpret_cast<const uint16_t*>(logits)[idx];
        return (u & 0x8000u) ? (~u & 0xffffu) : (u | 0x8000u);          // 16-bit key, larger logit = larger key
    }
    const uint32_t u = __float_as_uint(reinterpret_cast<const float*>(logits)[idx]);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void token_drop_select_kernel(const void* __restrict__ logits, int is_bf16, const int32_t* __restrict__ mask_in,
                                                                int S, int E, int n_dyn, int capacity, int policy,
                                                                int32_t* __restrict__ mask_out) {
    __shared__ unsigned hist[256];
    __shared__ unsigned scan[256];
    __shared__ unsigned bcast[2];
    const int e = blockIdx.x, tid = threadIdx.x;
    const int per = (S + 255) / 256, t0 = tid * per, t1 = min(t0 + per, S);   // contiguous token range per thread: index order
    if (policy == 0 && e >= n_dyn) {       // shared columns are never dropped by "probs" (keep[:, n_dyn:] = 1)
        for (int s = tid; s < S; s += 256) mask_out[(size_t)s * E + e] = mask_in[(size_t)s * E + e];
        return;
    }
    unsigned thr_key = 0;      // probs: kept = key > thr_key, plus the first `need_eq` tokens with key == thr_key
    unsigned need_eq = 0;
    bool keep_all = false;
    if (policy == 0) {
        unsigned cnt = 0;
        for (int s = t0; s < t1; ++s) cnt += mask_in[(size_t)s * E + e] != 0;
        scan[tid] = cnt;
        __syncthreads();
        unsigned total = 0;
        for (int i = 0; i < 256; ++i) total += scan[i];
        __syncthreads();
        if ((int)total <= capacity) {
            keep_all = true;
        } else {
            // radix select, most significant byte first: find the key of the capacity-th largest selected logit
            const int bits = is_bf16 ? 16 : 32;
            unsigned prefix = 0, want = (unsigned)capacity;     // `want` largest remain to be found among keys matching `prefix`
            for (int shift = bits - 8; shift >= 0; shift -= 8) {
                hist[tid] = 0;
                __syncthreads();
                const unsigned hi_mask = (shift + 8 >= 32) ? 0u : (0xffffffffu << (shift + 8));
                for (int s = t0; s < t1; ++s)
                    if (mask_in[(size_t)s * E + e] != 0) {
                        const unsigned k = drop_key(logits, is_bf16, (size_t)s * E + e);
                        if ((k & hi_mask) == (prefix & hi_mask)) atomicAdd(&hist[(k >> shift) & 255u], 1u);
                    }
                __syncthreads();
                if (tid == 0) {
                    unsigned acc = 0;
                    int b = 255;
                    for (; b > 0; --b) {
                        if (acc + hist[b] >= want) break;
                        acc += hist[b];
                    }
                    bcast[0] = (unsigned)b;
                    bcast[1] = want - acc;           // still to take inside bin b
                }
                __syncthreads();
                prefix |= bcast[0] << shift;
                want = bcast[1];
                __syncthreads();
            }
            thr_key = prefix;
            need_eq = want;
        }
    }
    // index-ordered prefix over the candidates of the tie break (probs) or over the selected tokens (position)
    unsigned mine = 0;
    for (int s = t0; s < t1; ++s) {
        const bool sel = mask_in[(size_t)s * E + e] != 0;
        if (policy == 1) mine += sel;
        else if (!keep_all) mine += sel && drop_key(logits, is_bf16, (size_t)s * E + e) == thr_key;
    }
    scan[tid] = mine;
    __syncthreads();
    unsigned before = 0;
    for (int i = 0; i < tid; ++i) before += scan[i];
    for (int s = t0; s < t1; ++s) {
        const int m = mask_in[(size_t)s * E + e];
        int out = m;
        if (m != 0) {
            if (policy == 1) {
                out = before < (unsigned)capacity ? m : 0;
                before += 1;
            } else if (!keep_all) {
                const unsigned k = drop_key(logits, is_bf16, (size_t)s * E + e);
                if (k > thr_key) out = m;
                else if (k == thr_key) { out = before < need_eq ? m : 0; before += 1; }
                else out = 0;
            }
        }
        mask_out[(size_t)s * E + e] = out;
    }
}

template <int TB>
__global__ __launch_bounds__(256) void token_drop_finish_kernel(const void* __restrict__ logits, const int32_t* __restrict__ mask, const float* __restrict__ routing_in,
                                                                int S, int n_dyn, int n_real, int n_fix, float* __restrict__ routing_out,
                                                                float* __restrict__ global_w, float* __restrict__ moe_w) {
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= S) return;
    const int E = n_dyn + n_fix;
    float full = -INFINITY;
    int m = 0;
    if (lane < E) {
        full = TB ? bf2f(reinterpret_cast<const uint16_t*>(logits)[(size_t)s * E + lane]) : reinterpret_cast<const float*>(logits)[(size_t)s * E + lane];
        m = mask[(size_t)s * E + lane];
    }
    float w = (lane < n_dyn && m != 0) ? routing_in[(size_t)s * n_dyn + lane] : 0.f;      // masked_fill(~mask, 0)
    float ws[UMOE_MAXE];
    gather16<0>(w, n_dyn, 0.f, ws);
    float sum = ws[0];
#pragma unroll
    for (int j = 1; j < UMOE_MAXE; ++j)
        if (j < n_dyn) sum = sum + ws[j];
    sum = round_t(sum, TB);
    const float den = round_t(sum + 1e-6f, TB);
    w = (lane < n_dyn) ? round_t(w / den, TB) : 0.f;                                        // core.py:328-329
    float gw = w;
    {
        const float gl = (lane < E && m) ? full : -INFINITY;                                 // core.py:188
        gw = lane_softmax<0, TB>(gl, E, lane);
        float gs[UMOE_MAXE];
        gather16<0>(gw, n_dyn, 0.f, gs);
        float ds = gs[0];
#pragma unroll
        for (int j = 1; j < UMOE_MAXE; ++j)
            if (j < n_dyn) ds = ds + gs[j];
        ds = round_t(ds, TB);
        if (lane < n_dyn) gw = round_t(w * ds, TB);
    }
    if (lane < E && global_w) global_w[(size_t)s * E + lane] = gw;
    if (lane < n_dyn && routing_out) routing_out[(size_t)s * n_dyn + lane] = w;
    if (lane < n_real && moe_w) moe_w[(size_t)s * n_real + lane] = gw * (float)(m != 0);
}

extern "C" int umoe_token_drop(const void* logits, int logits_bf16, const int32_t* expert_mask_in, const float* routing_w_in, int S,
                               int n_dyn, int n_real, int n_fix, int capacity, int policy, int32_t* expert_mask_out,
                               float* routing_w_out, float* global_w, float* moe_w, umoe_stream_t stream) {
    UMOE_REQUIRE(logits && expert_mask_in && routing_w_in && expert_mask_out && expert_mask_out != expert_mask_in,
                 "umoe_token_drop: null argument (the mask is not updated in place)");
    UMOE_REQUIRE(policy == 0 || policy == 1, "umoe_token_drop: policy must be 0 (probs) or 1 (position)");     // core.py:325 raises ValueError
    UMOE_REQUIRE(n_dyn >= 1 && n_dyn + n_fix <= UMOE_MAXE && n_real <= n_dyn && capacity >= 0, "umoe_token_drop: bad sizes");
    if (S == 0) return 0;
    const int E = n_dyn + n_fix;
    hipStream_t st = (hipStream_t)stream;
    token_drop_select_kernel<<<dim3((unsigned)E), 256, 0, st>>>(logits, logits_bf16, expert_mask_in, S, E, n_dyn, policy == 0 ? min(capacity, S) : capacity, policy, expert_mask_out);
    UMOE_LAUNCH_CHECK();
    if (logits_bf16) token_drop_finish_kernel<1><<<dim3((unsigned)ceil_div(S, 4)), 256, 0, st>>>(logits, expert_mask_out, routing_w_in, S, n_dyn, n_real, n_fix, routing_w_out, global_w, moe_w);
    else token_drop_finish_kernel<0><<<dim3((unsigned)ceil_div(S, 4)), 256, 0, st>>>(logits, expert_mask_out, routing_w_in, S, n_dyn, n_real, n_fix, routing_w_out, global_w, moe_w);
    UMOE_LAUNCH_CHECK();
    return 0;
}
