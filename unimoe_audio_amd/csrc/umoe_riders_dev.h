// Riders of the small decode GEMM launches (umoe_gemm.hip wstream_gemm_rk): the latency-bound row kernels in FRONT of a GEMM run as
// the first 16 workgroups of the GEMM's own launch and HAND their output rows to the GEMM workgroups (sc1 stores, drain, one flag per
// row; umoe_common.h umoe_rider_pub) -- the GEMM workgroups request their whole weight slice first and wait for the flags, so the
// launch boundary between the two kernels and the second kernel's cold start leave the chain.
//   combine_row_dense: combine_kernel<false, true> of umoe_misc.hip for ONE row (dense decode layout, D = 2048: one 16-byte chunk per
//     thread), same loads, same arithmetic order, same fixed-order RMSNorm sum -> bit-identical rows          (core.py:488,342-351)
//     EPC (expert parallel, umoe_moe_ep.hip): the routed experts' rows sit in this rank's RETURN slab; lane e of wave 0 waits until the
//     owner of expert e has counted all its workgroups in for this round, and every load of the slab is a system-scope load
// (Measured and removed: the merge of the attention key splits riding in the o_proj launch the same way -- 3.18 vs 3.13 ms/step: the
//  merge is 1 us of work, its hand-off costs more than the launch boundary it removes.)
#pragma once
#include "umoe_common.h"

struct umoe_rider2 {
    int n_riders;                  // rows (= rider workgroups in front of the GEMM tiles)
    umoe_combine_args cb;          // kind 2: MoE combine + residual + next RMSNorm; cb.norm_out is the GEMM's activation operand
    // kind 4 (expert parallel): cb.y_slots = this rank's return slab [n_real][S][D] inside `ep_region`
    char* ep_region;               // own exchange region (counters: umoe_ep_flag(ep_region, 1, source rank, 0))
    const uint32_t* ep_round;      // device word: rounds base (decode steps taken)
    int ep_layer, ep_layers;       // the layer whose experts produced the rows: round = *ep_round * ep_layers + ep_layer + 1
    int ep_size, ep_n_cwg;         // ranks; workgroups that count themselves in per round and source rank
    uint32_t* ep_err;
};

__device__ __forceinline__ void rider_publish(uint32_t* flag, uint32_t epoch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x < UMOE_FLAG_REPL)      // `flag` = this row's word in replica 0; the replicas are 16 words apart
        __hip_atomic_store(reinterpret_cast<umoe_gu32*>(reinterpret_cast<uintptr_t>(flag + threadIdx.x * 16)), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void st16_sc1(uint16_t* base, long elem_off, uint4 v, long bytes) {
    typedef uint32_t u32x4_pub __attribute__((ext_vector_type(4)));
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
    const u32x4_pub v4 = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(v4, rsrc, (int)(elem_off * 2), 0, 16);      // aux 16 = sc1 (agent-scope write-through)
}

template <bool EPC>
__device__ __forceinline__ void combine_row_dense(const umoe_combine_args& a, const umoe_rider2& r2, const int s, float* sh, uint32_t* flag, const uint32_t epoch) {
    const int tid = threadIdx.x, lane = tid & 63, c = tid;          // D = 2048: chunk c of the row
    const int E = a.n_dyn + a.n_fix;
    // tables (lane e <- entry e; clamped, straight-line) and every row this token needs, all in one round trip
    const float wgt_l = a.moe_w[(size_t)s * a.n_real + min(lane, a.n_real - 1)];
    const float sw_l = a.global_w[(size_t)s * E + a.n_dyn + min(lane, a.n_fix - 1)];
    const int tab_v = a.expert_mask[(size_t)s * a.mask_ld + min(lane, a.n_real - 1)];
    uint4 yv[UMOE_MAXE], sv[4];
    if constexpr (EPC) {
        // expert e's rows come from rank e / E_loc: wait until that rank has counted every one of its phase C workgroups in (bounded)
        if (tid < a.n_real) {
            const int src = tid / (a.n_real / r2.ep_size);
            // (the counters are cumulative: ep_n_cwg per layer of every decode step taken so far; *ep_round counts the steps incl. this one)
            const uint32_t target = ((__builtin_nontemporal_load(r2.ep_round) - 1u) * (uint32_t)r2.ep_layers + (uint32_t)r2.ep_layer + 1u) * (uint32_t)r2.ep_n_cwg;
            umoe_ep_wait(umoe_ep_flag(r2.ep_region, 1, src, 0), target, r2.ep_err);
        }
        __syncthreads();
        typedef uint32_t u32x4_ep __attribute__((ext_vector_type(4)));
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.y_slots), 0, a.n_real * a.dense_rows * a.D * 2, 0x00020000);
#pragma unroll
        for (int e = 0; e < UMOE_MAXE; ++e)
            if (e < a.n_real) {
                const u32x4_ep t4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ((e * a.dense_rows + s) * a.D + c * 8) * 2, 0, UMOE_SYS_AUX);
                yv[e] = make_uint4(t4[0], t4[1], t4[2], t4[3]);
            }
    } else {
#pragma unroll
        for (int e = 0; e < UMOE_MAXE; ++e)
            if (e < a.n_real) yv[e] = ld16(a.y_slots + (size_t)(e * a.dense_rows + s) * a.D + c * 8);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < a.n_fix) sv[i] = ld16(a.y_shared + ((size_t)i * a.S + s) * a.D + c * 8);
    const uint4 rv = ld16(a.resid + (size_t)s * a.D + c * 8);
    const uint4 nw = ld16(a.norm_w + c * 8);
    __builtin_amdgcn_sched_barrier(0);
    int sel[UMOE_MAXE];
    float wgt[UMOE_MAXE], swgt[4];
    const int sel_l = lane < a.n_real ? (tab_v != 0) : 0;
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e) {
        sel[e] = __builtin_amdgcn_readlane(sel_l, e);
        wgt[e] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wgt_l), e));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) swgt[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sw_l), i));
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e)
        if (e < a.n_real && sel[e]) {
            float y[8];
            unpack8(yv[e], y);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += wgt[e] * y[j];
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = rbf(acc[j]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < a.n_fix) {
            float y[8];
            unpack8(sv[i], y);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = rbf(acc[j] + rbf(y[j] * swgt[i]));
        }
    {
        float r[8];
        unpack8(rv, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = r[j] + acc[j];
    }
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        acc[j] = rbf(acc[j]);
        ss += acc[j] * acc[j];
    }
    st16(a.out + (size_t)s * a.D + c * 8, pack8(acc));        // the residual stream: read by LATER launches only
    ss = block_sum_256(ss, sh);
    const float rs = rsqrtf(ss / (float)a.D + a.rms_eps);
    float w[8], f[8];
    unpack8(nw, w);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = w[j] * rbf(acc[j] * rs);
    st16_sc1(a.norm_out, (long)s * a.D + c * 8, pack8(f), (long)a.S * a.D * 2);
    rider_publish(flag, epoch);
}
