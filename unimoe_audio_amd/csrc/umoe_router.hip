// Top-P dynamic router + ragged dispatch tables.
//
// Replaces the reference's per-k Python loop of tiny torch kernels and its >= 15 host syncs per
// layer (utils/UniMoE_Audio_core.py:246-291, 94-167, 178-193; dispatch: utils/UniMoE_Audio_utils.py
// :436-523) with two launches and no host sync.
//
// router_kernel: one 64-lane wave per token.  The gate GEMV (11 dot products of length D) is
// spread over the wave and reduced with a fixed xor butterfly; afterwards lane e owns router
// column e and the Top-P count / iterative arg-max mixer run lane-parallel with wave shuffles --
// every fp32 operation happens in the order fixed by the arithmetic contract of
// oracle/router_oracle.c (sequential softmax sum, reciprocal multiply, deterministic exp, cumsum
// accumulator type, thresholds cast to T), so integer outputs are bit-exact given equal logits.
//
// dispatch_kernel: one workgroup per routed expert; wavefront ballot + popcount prefix sums give
// every (token, expert) pair its slot in token order -- no [S,E,D] expansion, no padding to the
// max capacity, no argsort.
// Roofline: latency-bound at decode (S = 16); HBM-bound on [S,D] reads for large S.
#include "umoe_common.h"

struct RouterDev {
    umoe_router_args a;
};

// broadcast helpers: value held by lane `src` (lanes 0..15 hold router columns)
__device__ __forceinline__ float bcast(float v, int src) { return __shfl(v, src, 64); }

// softmax over lanes [0, n): sequential fp32 sum in index order, reciprocal multiply, round to T
__device__ __forceinline__ float lane_softmax(float x, int n, int lane, int is_bf16) {
    float xm = (lane < n) ? x : -INFINITY;
    float m = xm;
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));  // lanes 0..15 (16..63 hold -inf too)
    m = __shfl(m, 0, 64);
    const float e = (lane < n) ? umoe_exp_det(xm - m) : 0.f;
    float s = bcast(e, 0);
    for (int j = 1; j < n; ++j) s = s + bcast(e, j);
    const float r = 1.0f / s;
    return round_t(e * r, is_bf16);
}

__global__ __launch_bounds__(256) void router_kernel(const umoe_router_args a) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= a.S) return;
    const int E = a.n_dyn + a.n_fix, n_dyn = a.n_dyn, T = a.logits_bf16;

    // ---- logits: lane e <- column e ----------------------------------------------------------
    float full = -INFINITY;  // this lane's logit (valid for lane < E)
    if (a.logits_in) {
        if (lane < E)
            full = T ? bf2f(reinterpret_cast<const uint16_t*>(a.logits_in)[(size_t)s * E + lane])
                     : reinterpret_cast<const float*>(a.logits_in)[(size_t)s * E + lane];
    } else {
        const uint16_t* xr = a.x + (size_t)s * a.D;
        const int nchunk = a.D >> 3;
        float rs = 1.f;
        if (a.norm_w) {
            float ss = 0.f;
            for (int c = lane; c < nchunk; c += 64) {
                float f[8];
                unpack8(ld16(xr + c * 8), f);
#pragma unroll
                for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
            }
            ss = wave_sum(ss);
            rs = rsqrtf(ss / (float)a.D + a.rms_eps);
        }
        float acc[UMOE_MAXE];
#pragma unroll
        for (int e = 0; e < UMOE_MAXE; ++e) acc[e] = 0.f;
        for (int c = lane; c < nchunk; c += 64) {
            float f[8];
            uint4 u = ld16(xr + c * 8);
            unpack8(u, f);
            if (a.norm_w) {
                float w[8];
                unpack8(ld16(a.norm_w + c * 8), w);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = rbf(w[j] * rbf(f[j] * rs));
                u = pack8(f);
            }
            if (a.h_out) st16(a.h_out + (size_t)s * a.D + c * 8, u);
#pragma unroll
            for (int e = 0; e < UMOE_MAXE; ++e)
                if (e < E) {
                    float w[8];
                    unpack8(ld16(a.gate_w + (size_t)e * a.D + c * 8), w);
                    float d = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) d += f[j] * w[j];
                    acc[e] += d;
                }
        }
        float mine = 0.f;
#pragma unroll
        for (int e = 0; e < UMOE_MAXE; ++e) {
            const float t = wave_sum(acc[e]);
            if (lane == e) mine = t;
        }
        if (lane < E) full = round_t(mine, T);
    }
    if (a.logits_out && lane < E) {
        if (T) reinterpret_cast<uint16_t*>(a.logits_out)[(size_t)s * E + lane] = f2bf(full);
        else reinterpret_cast<float*>(a.logits_out)[(size_t)s * E + lane] = full;
    }

    // ---- Top-P count (core.py:157-167) -------------------------------------------------------
    int k = a.fixed_top_k;
    if (a.top_p != 0.0f) {
        const float p = lane_softmax(full, n_dyn, lane, T);
        // rank in descending order (ties: lower index first); values only matter
        int rank = 0;
        for (int j = 0; j < n_dyn; ++j) {
            const float pj = bcast(p, j);
            rank += (pj > p) || (pj == p && j < lane);
        }
        // sorted[t] = p of the lane whose rank is t
        int src = 0;
        for (int j = 0; j < n_dyn; ++j)
            if (__shfl(rank, j, 64) == lane) src = j;
        const float sorted = __shfl(p, src, 64);
        const float thr = round_t(a.top_p, T);
        int below = 0;
        if (T) {
            float acc = 0.f;
            for (int t = 0; t < n_dyn; ++t) {
                acc = acc + bcast(sorted, t);
                below += !(rbf(acc) >= thr);
            }
        } else {
            double acc = 0.0;
            for (int t = 0; t < n_dyn; ++t) {
                acc = acc + (double)bcast(sorted, t);
                below += !((float)acc >= thr);
            }
        }
        k = below + 1;
    }
    if (k > n_dyn) k = n_dyn;

    // ---- iterative arg-max mixer, eval branch (core.py:94-154, 262-282) -----------------------
    const float two_eps = round_t((float)(2.0 * a.jitter_eps), T);
    float masked = (lane < n_dyn) ? full : -INFINITY;
    float w = 0.f;
    int m = 0;
    for (int j = 0; j < k; ++j) {
        // max + lowest-index arg-max over lanes 0..15
        float bv = masked;
        int bi = lane;
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        const float thr = __shfl(bv, 0, 64);
        const int ind = __shfl(bi, 0, 64);
        const float af = fabsf(full), at = fabsf(thr);
        const float factor = af > at ? af : at;
        const float d = round_t(thr - full, T);
        const float q = round_t(d / factor, T);
        const float gate = (q > two_eps) ? -INFINITY : masked;
        const float gsm = lane_softmax(gate, n_dyn, lane, T);
        if (lane == ind) {
            w = gsm;
            m += 1;
            masked = -INFINITY;
        }
        if (a.sel && lane == 0) a.sel[(size_t)s * n_dyn + j] = ind;
    }
    if (a.sel && lane >= k && lane < n_dyn) a.sel[(size_t)s * n_dyn + lane] = -1;

    // ---- renormalise / padding / shared always on (core.py:284-291) ---------------------------
    float sum = bcast(w, 0);
    for (int j = 1; j < n_dyn; ++j) sum = sum + bcast(w, j);
    sum = round_t(sum, T);
    const float den = round_t(sum + 1e-6f, T);
    w = (lane < n_dyn) ? round_t(w / den, T) : 0.f;
    if (a.attn_mask) m *= (int)(a.attn_mask[s] != 0);
    if (lane >= n_dyn && lane < E) m = 1;

    // ---- global routing weight (core.py:178-193) ----------------------------------------------
    float gw = w;
    if (a.n_fix > 0) {
        const float gl = (lane < E && m) ? full : -INFINITY;
        gw = lane_softmax(gl, E, lane, T);
        float ds = bcast(gw, 0);
        for (int j = 1; j < n_dyn; ++j) ds = ds + bcast(gw, j);
        ds = round_t(ds, T);
        if (lane < n_dyn) gw = round_t(w * ds, T);
    }
    if (lane < E) {
        a.expert_mask[(size_t)s * E + lane] = m;
        if (a.global_w) a.global_w[(size_t)s * E + lane] = gw;
    }
    if (lane < n_dyn && a.routing_w) a.routing_w[(size_t)s * n_dyn + lane] = w;
    if (lane < a.n_real && a.moe_w) a.moe_w[(size_t)s * a.n_real + lane] = gw * (float)m;
    if (a.top_k && lane == 0) a.top_k[s] = k;
}

extern "C" int umoe_router_fwd(const umoe_router_args* a, umoe_stream_t stream) {
    UMOE_REQUIRE(a && a->expert_mask, "umoe_router_fwd: null argument");
    const int E = a->n_dyn + a->n_fix;
    UMOE_REQUIRE(a->S >= 0 && a->n_dyn >= 1 && E <= UMOE_MAXE && a->n_real <= a->n_dyn,
                 "umoe_router_fwd: bad sizes S=%d n_dyn=%d n_real=%d n_fix=%d (E <= %d)", a->S, a->n_dyn, a->n_real,
                 a->n_fix, UMOE_MAXE);
    UMOE_REQUIRE(a->logits_in || (a->x && a->gate_w && a->D > 0 && a->D % 8 == 0),
                 "umoe_router_fwd: need logits_in or (x, gate_w, D %% 8 == 0)");
    if (a->S == 0) return 0;
    router_kernel<<<dim3((unsigned)ceil_div(a->S, 4)), 256, 0, (hipStream_t)stream>>>(*a);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ dispatch
// grid = n_real workgroups of 256.  Every workgroup counts all experts (to know its own offset),
// then assigns slots for its expert in token order with ballot/popcount prefix sums.
__global__ __launch_bounds__(256) void dispatch_kernel(const int32_t* __restrict__ mask, int S, int ld, int n_real,
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
    for (int ee = 0; ee < e; ++ee) off += tot[ee];
    if (tid == 0) {
        counts[e] = tot[e];
        offsets[e] = off;
        if (e == n_real - 1) offsets[n_real] = off + tot[e];
    }
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
    dispatch_kernel<<<dim3((unsigned)n_real), 256, 0, (hipStream_t)stream>>>(expert_mask, S, ld_mask, n_real, counts,
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
