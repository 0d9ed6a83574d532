// Small fused kernels of the decode path: RMSNorm(+residual), unpermute+combine, codec embedding sum,
// CFG + masks + sampler, on-device EOS/delay bookkeeping, RVQ lookup / nearest neighbour.
// All are HBM/latency-bound elementwise or row-reduction kernels: 16-byte vector accesses, one
// workgroup per row, no atomics, fixed summation orders.
#include "umoe_common.h"
#include <string.h>

static thread_local char g_err[512] = "";
void umoe_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* umoe_last_error(void) { return g_err; }
extern "C" int umoe_abi_version(void) { return 1; }
extern "C" size_t umoe_struct_size(const char* name) {
    if (!name) return 0;
#define UMOE_SZ(t) if (!strcmp(name, #t)) return sizeof(t);
    UMOE_SZ(umoe_router_args) UMOE_SZ(umoe_group_t) UMOE_SZ(umoe_gemm_args) UMOE_SZ(umoe_tgroup_t) UMOE_SZ(umoe_tgemm_args) UMOE_SZ(umoe_tn_group_t) UMOE_SZ(umoe_tgemm_tn_args)
    UMOE_SZ(umoe_swiglu_bwd_args) UMOE_SZ(umoe_attn_bwd_args) UMOE_SZ(umoe_combine_args) UMOE_SZ(umoe_rope_args) UMOE_SZ(umoe_attn_args)
    UMOE_SZ(umoe_sample_args) UMOE_SZ(umoe_engine_cfg) UMOE_SZ(umoe_layer_weights) UMOE_SZ(umoe_decode_io)
#undef UMOE_SZ
    return 0;
}


// ------------------------------------------------------------------------------------ rmsnorm
// transformers Qwen2RMSNorm (reference model.py:206-207,307): w * bf16(x * rsqrt(mean(x^2) + eps))
__global__ __launch_bounds__(256) void rmsnorm_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ r,
                                                      const uint16_t* __restrict__ w, float eps, int D,
                                                      uint16_t* __restrict__ sum_out, uint16_t* __restrict__ y) {
    __shared__ float sh[4];
    const size_t base = (size_t)blockIdx.x * D;
    float ss = 0.f;
    for (int c = threadIdx.x; c < (D >> 3); c += 256) {
        float f[8];
        unpack8(ld16(x + base + c * 8), f);
        if (r) {
            float g[8];
            unpack8(ld16(r + base + c * 8), g);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = rbf(f[j] + g[j]);
            if (sum_out) st16(sum_out + base + c * 8, pack8(f));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
    }
    ss = block_sum_256(ss, sh);
    const float rs = rsqrtf(ss / (float)D + eps);
    for (int c = threadIdx.x; c < (D >> 3); c += 256) {
        float f[8], wv[8];
        unpack8(ld16(x + base + c * 8), f);
        if (r) {
            float g[8];
            unpack8(ld16(r + base + c * 8), g);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = rbf(f[j] + g[j]);
        }
        unpack8(ld16(w + c * 8), wv);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = wv[j] * rbf(f[j] * rs);
        st16(y + base + c * 8, pack8(f));
    }
}

extern "C" int umoe_rmsnorm_residual_fwd(const uint16_t* x, const uint16_t* r, const uint16_t* w, float eps, int S,
                                         int D, uint16_t* sum_out, uint16_t* y, umoe_stream_t stream) {
    UMOE_REQUIRE(x && w && y && D % 8 == 0 && S >= 0, "umoe_rmsnorm_residual_fwd: bad argument (D=%d)", D);
    if (S == 0) return 0;
    rmsnorm_kernel<<<dim3((unsigned)S), 256, 0, (hipStream_t)stream>>>(x, r, w, eps, D, sum_out, y);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ combine
// reference: einsum("se,sem->sm") core.py:488 (fp32 accumulate, one rounding), `final + current`
// core.py:342, shared experts `expert(x) * w` then add core.py:349-351, residual model.py:242.
// EP (expert parallel decode, dense layout): y_slots is this rank's RETURN slab (uncached region memory); the rows of expert e
// arrive from rank e / E_loc (second all-to-all, core.py:480): lane e of wave 0 waits for that rank's flag of THIS token row,
// the workgroup meets at its barrier, and every load of the slab is a system-scope (sc0 sc1) load (hand-off form: umoe_common.h).
template <bool EP, bool DENSE = EP>     // DENSE: the dense layout (no slot table, no partial slabs) known at compile time
__global__ __launch_bounds__(256) void combine_kernel(const umoe_combine_args a, const umoe_ep_xfer x) {
    __shared__ float sh[4];
    TL_ENTER(9);
    const int s = blockIdx.x;
    const int E = a.n_dyn + a.n_fix;
    float ss = 0.f;
    // one expert-output row chunk: bf16 slots, or the fixed-order sum of the K-split fp32 partial slabs rounded ONCE
    // to bf16 (the rounding point of the reference's down_proj output)
    auto load_y = [&](long row, int c, float* y) {
        if (a.y_parts) {
            float sacc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) sacc[j] = 0.f;
            for (int pz = 0; pz < a.n_parts; ++pz) {
                const float4* q = reinterpret_cast<const float4*>(a.y_parts + (size_t)pz * a.part_stride + row * a.D + c * 8);
                const float4 u0 = q[0], u1 = q[1];
                sacc[0] += u0.x; sacc[1] += u0.y; sacc[2] += u0.z; sacc[3] += u0.w;
                sacc[4] += u1.x; sacc[5] += u1.y; sacc[6] += u1.z; sacc[7] += u1.w;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = rbf(sacc[j]);
        } else {
            unpack8(ld16(a.y_slots + (size_t)row * a.D + c * 8), y);
        }
    };
    // routing of this token: all table entries first (one round trip), then every selected expert row, the shared
    // rows and the residual are requested together; the accumulation order below is the reference's
    int slot[UMOE_MAXE];
    float wgt[UMOE_MAXE], swgt[UMOE_MAXE];
    const bool fastp = DENSE || (!a.y_parts && a.n_real <= UMOE_MAXE && a.n_fix <= 4);
    // dense layout (every expert computed every row): the row of expert e is e * dense_rows + s whatever the mask says, so the rows are
    // requested TOGETHER with the tables (one round trip, not two) and the mask only decides which of them are added
    const bool dense_rows_known = DENSE;
    int slot_l = -1, tab_v = 0;      // lane e <- table entry e (raw value: consumed in broadcast_tables, not before)
    float wgt_l = 0.f, sw_l = 0.f;
    if (fastp) {
        // one vector load per table (lane e <- entry e), then broadcast: a single memory round trip instead of a chain
        // of dependent scalar loads
        const int lane = threadIdx.x & 63;
        if (DENSE) {
            // straight-line, clamped (lanes beyond the tables re-read their last entry; never used): no branch around a load, so the
            // compiler does not drain the queue in front of the row requests
            wgt_l = a.moe_w[(size_t)s * a.n_real + min(lane, a.n_real - 1)];
            sw_l = a.global_w[(size_t)s * E + a.n_dyn + min(lane, a.n_fix - 1)];
            tab_v = a.expert_mask[(size_t)s * a.mask_ld + min(lane, a.n_real - 1)];
        } else {
            if (lane < a.n_real) wgt_l = a.moe_w[(size_t)s * a.n_real + lane];
            if (a.y_shared && lane < a.n_fix) sw_l = a.global_w[(size_t)s * E + a.n_dyn + lane];
            if (lane < a.n_real) {   // both table flavours are ONE int per (token, expert): same load, different meaning
                const int32_t* tab = a.slot_of ? a.slot_of + (size_t)s * a.n_real : a.expert_mask + (size_t)s * a.mask_ld;
                tab_v = tab[lane];
            }
        }
    }
    auto broadcast_tables = [&]() {
        const int lane = threadIdx.x & 63;
        slot_l = lane < a.n_real ? ((!DENSE && a.slot_of) ? tab_v : (tab_v != 0 ? lane * a.dense_rows + s : -1)) : -1;
#pragma unroll
        for (int e = 0; e < UMOE_MAXE; ++e) {
            slot[e] = __builtin_amdgcn_readlane(slot_l, e);
            wgt[e] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wgt_l), e));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) swgt[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sw_l), i));
    };
    bool tables_ready = false;
    if (fastp && !dense_rows_known) {
        broadcast_tables();
        tables_ready = true;
    }
    if constexpr (EP) {
        if (threadIdx.x < (unsigned)a.n_real) {
            const int src = (int)threadIdx.x / (a.n_real / x.size);
            if (x.n_cwg > 0) {      // slab filled by the one-launch exchange (umoe_moe_ep.hip): every source counts its workgroups in
                const uint32_t target = ((*x.round - 1u) * (uint32_t)x.layers + (uint32_t)x.layer + 1u) * (uint32_t)x.n_cwg;      // cumulative
                umoe_ep_wait(umoe_ep_flag(x.peer_base[x.rank], 1, src, 0), target, x.err);
            } else if (src != x.rank) {
                umoe_ep_wait(umoe_ep_flag(x.peer_base[x.rank], 1, src, s), umoe_ep_epoch(x), x.err);
            }
        }
        __syncthreads();
    }
    TL_MARK(9, 5);
    for (int c = threadIdx.x; c < (a.D >> 3); c += 256) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        if (fastp) {
            uint4 yv[UMOE_MAXE], sv[4], rv = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < UMOE_MAXE; ++e)   // unselected experts re-read row 0 (valid memory, value never used): no
                if (e < a.n_real) {                // data-dependent branch between the loads, all of them in flight at once
                    if constexpr (EP) {
                        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.y_slots), 0, a.n_real * a.dense_rows * a.D * 2, 0x00020000);
                        const u32x4 t4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ((e * a.dense_rows + s) * a.D + c * 8) * 2, 0, UMOE_SYS_AUX);
                        yv[e] = make_uint4(t4[0], t4[1], t4[2], t4[3]);
                    } else {
                        yv[e] = ld16(a.y_slots + (size_t)(dense_rows_known ? e * a.dense_rows + s : (slot[e] >= 0 ? slot[e] : 0)) * a.D + c * 8);
                    }
                }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (a.y_shared && i < a.n_fix) sv[i] = ld16(a.y_shared + ((size_t)i * a.S + s) * a.D + c * 8);
            if (a.resid) rv = ld16(a.resid + (size_t)s * a.D + c * 8);
            if (!tables_ready) {     // dense layout: the tables are consumed only now, behind the row requests
                __builtin_amdgcn_sched_barrier(0);
                broadcast_tables();
                tables_ready = true;
            }
            TL_MARK(9, 6);
#pragma unroll
            for (int e = 0; e < UMOE_MAXE; ++e)
                if (e < a.n_real && slot[e] >= 0) {
                    float y[8];
                    unpack8(yv[e], y);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += wgt[e] * y[j];
                }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = rbf(acc[j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (a.y_shared && i < a.n_fix) {
                    float y[8];
                    unpack8(sv[i], y);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = rbf(acc[j] + rbf(y[j] * swgt[i]));
                }
            if (a.resid) {
                float r[8];
                unpack8(rv, r);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = r[j] + acc[j];
            }
        } else {
            for (int e = 0; e < a.n_real; ++e) {
                const int sl = a.slot_of ? a.slot_of[(size_t)s * a.n_real + e]
                                         : (a.expert_mask[(size_t)s * a.mask_ld + e] != 0 ? e * a.dense_rows + s : -1);
                if (sl >= 0) {
                    const float wg = a.moe_w[(size_t)s * a.n_real + e];
                    float y[8];
                    load_y(sl, c, y);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += wg * y[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = rbf(acc[j]);
            if (a.y_shared || (a.y_parts && a.shared_row0 >= 0))
                for (int i = 0; i < a.n_fix; ++i) {
                    const float wg = a.global_w[(size_t)s * E + a.n_dyn + i];
                    float y[8];
                    if (a.y_shared) unpack8(ld16(a.y_shared + ((size_t)i * a.S + s) * a.D + c * 8), y);
                    else load_y((long)a.shared_row0 + (long)i * a.S + s, c, y);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = rbf(acc[j] + rbf(y[j] * wg));
                }
            if (a.resid) {
                float r[8];
                unpack8(ld16(a.resid + (size_t)s * a.D + c * 8), r);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = r[j] + acc[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j] = rbf(acc[j]);
            ss += acc[j] * acc[j];
        }
        TL_MARK(9, 7);
        st16(a.out + (size_t)s * a.D + c * 8, pack8(acc));
    }
    TL_MARK(9, 4);
    if (!a.norm_w) return;
    // fused RMSNorm of the row just produced (input_layernorm of the NEXT layer / final norm): saves every QKV
    // workgroup from recomputing it (model.py:227,428)
    ss = block_sum_256(ss, sh);
    const float rs = rsqrtf(ss / (float)a.D + a.rms_eps);
    for (int c = threadIdx.x; c < (a.D >> 3); c += 256) {
        float f[8], w[8];
        unpack8(ld16(a.out + (size_t)s * a.D + c * 8), f);   // this thread's own stores
        unpack8(ld16(a.norm_w + c * 8), w);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = w[j] * rbf(f[j] * rs);
        st16(a.norm_out + (size_t)s * a.D + c * 8, pack8(f));
    }
    TL_EXIT(9);
}
UMOE_TL_SETTER(misc)

extern "C" int umoe_unpermute_combine_fwd(const umoe_combine_args* a, umoe_stream_t stream) {
    UMOE_REQUIRE(a && (a->y_slots || (a->y_parts && a->n_parts > 0)) && (a->slot_of || (a->expert_mask && a->dense_rows >= a->S)) &&
                     a->moe_w && a->out && a->D % 8 == 0,
                 "umoe_unpermute_combine_fwd: bad argument");
    UMOE_REQUIRE(!(a->y_shared || (a->y_parts && a->shared_row0 >= 0)) || a->global_w,
                 "umoe_unpermute_combine_fwd: shared experts need global_w");
    if (a->S == 0) return 0;
    umoe_ep_xfer x;
    memset(&x, 0, sizeof(x));
    if (a->ep_xfer) {
        x = *reinterpret_cast<const umoe_ep_xfer*>(a->ep_xfer);
        UMOE_REQUIRE(!a->slot_of && !a->y_parts && a->expert_mask && a->n_real <= UMOE_MAXE && a->n_fix >= 1 && a->n_fix <= 4 && a->y_shared && a->global_w &&
                         x.size >= 2 && a->n_real % x.size == 0 &&
                         a->S <= UMOE_EP_PARTS && a->dense_rows == a->S,
                     "umoe_unpermute_combine_fwd: the expert-parallel form needs the dense layout over <= %d rows", UMOE_EP_PARTS);
        combine_kernel<true><<<dim3((unsigned)a->S), 256, 0, (hipStream_t)stream>>>(*a, x);
    } else {
        if (!a->slot_of && !a->y_parts && a->expert_mask && a->n_real >= 1 && a->n_real <= UMOE_MAXE && a->y_shared && a->global_w && a->n_fix >= 1 && a->n_fix <= 4)
            combine_kernel<false, true><<<dim3((unsigned)a->S), 256, 0, (hipStream_t)stream>>>(*a, x);
        else
            combine_kernel<false, false><<<dim3((unsigned)a->S), 256, 0, (hipStream_t)stream>>>(*a, x);
    }
    UMOE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ codec embedding
__global__ __launch_bounds__(256) void codec_embed_kernel(const int32_t* __restrict__ tok, const uint16_t* __restrict__ emb,
                                                          int C, int V, int D, uint16_t* __restrict__ out) {
    const int r = blockIdx.x;
    for (int c8 = threadIdx.x; c8 < (D >> 3); c8 += 256) {
        float acc[8];
        for (int ch = 0; ch < C; ++ch) {
            int t = tok[(size_t)r * C + ch];
            t = min(max(t, 0), V - 1);
            float e[8];
            unpack8(ld16(emb + ((size_t)ch * V + t) * D + c8 * 8), e);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = (ch == 0) ? e[j] : rbf(acc[j] + e[j]);
        }
        st16(out + (size_t)r * D + c8 * 8, pack8(acc));
    }
}

extern "C" int umoe_codec_embed_sum(const int32_t* tok, const uint16_t* emb, int rows, int C, int V, int D,
                                    uint16_t* out, umoe_stream_t stream) {
    UMOE_REQUIRE(tok && emb && out && D % 8 == 0 && C > 0, "umoe_codec_embed_sum: bad argument");
    if (rows == 0) return 0;
    codec_embed_kernel<<<dim3((unsigned)rows), 256, 0, (hipStream_t)stream>>>(tok, emb, C, V, D, out);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// backward: one workgroup per (vocabulary id v, channel c).  It scans the channel's token column once (thread i <- rows i, i + 256,
// ..), keeps the match bits of every 256-row segment in LDS, then all threads walk the bits in ascending row order and add that
// row of d_out to their 8 columns: deterministic, no atomics, no sort.  Ids nobody chose (most of a 1027-entry table at 6 k rows
// are chosen a handful of times) cost the scan only.
#define EMB_BWD_SEG 64     // 256-row segments per pass (16 K rows)
__global__ __launch_bounds__(256) void codec_embed_bwd_kernel(const int32_t* __restrict__ tok, const uint16_t* __restrict__ d_out, int rows, int C,
                                                              int V, int D, uint16_t* __restrict__ d_emb) {
    __shared__ unsigned long long bits[EMB_BWD_SEG][4];
    const int v = blockIdx.x, c = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nc8 = D >> 3;
    float acc[8][8];     // up to 8 column chunks per thread (D <= 16384)
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[q][j] = 0.f;
    for (int r0 = 0; r0 < rows; r0 += EMB_BWD_SEG * 256) {
        const int nseg = min(EMB_BWD_SEG, (rows - r0 + 255) / 256);
        for (int sg = 0; sg < nseg; ++sg) {
            const int r = r0 + sg * 256 + tid;
            const bool hit = r < rows && tok[(size_t)r * C + c] == v;
            const unsigned long long m = __ballot(hit);
            if (lane == 0) bits[sg][wave] = m;
        }
        __syncthreads();
        for (int sg = 0; sg < nseg; ++sg)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                unsigned long long m = bits[sg][w];
                while (m) {
                    const int b = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const size_t r = (size_t)r0 + sg * 256 + w * 64 + b;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int c8 = tid + q * 256;
                        if (c8 < nc8) {
                            float f[8];
                            unpack8(ld16(d_out + r * D + c8 * 8), f);
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[q][j] += f[j];
                        }
                    }
                }
            }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c8 = tid + q * 256;
        if (c8 < nc8) st16(d_emb + ((size_t)c * V + v) * D + c8 * 8, pack8(acc[q]));
    }
}

// Large tables (V much larger than the row count: the 151 936-entry text table at 6 240 rows): one workgroup per ROW instead of
// one per id.  It scans the id column in ascending segments; the first match it meets is the smallest row with its id -- if that
// is an earlier row, this workgroup is not the id's owner and leaves; the owner adds every matching row in ascending order (the same
// order, hence the same bits, as the per-id kernel) and writes the table row.  Rows of the table nobody chose are zeroed by a memset
// in front of the launch.
__global__ __launch_bounds__(256) void codec_embed_bwd_rows_kernel(const int32_t* __restrict__ tok, const uint16_t* __restrict__ d_out, int rows,
                                                                   int C, int V, int D, uint16_t* __restrict__ d_emb) {
    __shared__ unsigned long long bits[EMB_BWD_SEG][4];
    __shared__ int first_seen;
    const int me = blockIdx.x, c = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int v = tok[(size_t)me * C + c];
    if (v < 0 || v >= V) return;
    const int nc8 = D >> 3;
    float acc[8][8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[q][j] = 0.f;
    if (tid == 0) first_seen = 0x7fffffff;
    __syncthreads();
    for (int r0 = 0; r0 < rows; r0 += EMB_BWD_SEG * 256) {
        const int nseg = min(EMB_BWD_SEG, (rows - r0 + 255) / 256);
        for (int sg = 0; sg < nseg; ++sg) {
            const int r = r0 + sg * 256 + tid;
            const bool hit = r < rows && tok[(size_t)r * C + c] == v;
            const unsigned long long m = __ballot(hit);
            if (lane == 0) {
                bits[sg][wave] = m;
                if (m) atomicMin(&first_seen, r0 + sg * 256 + wave * 64 + (__ffsll((long long)m) - 1));
            }
        }
        __syncthreads();
        if (first_seen < me) return;          // an earlier row owns this id (uniform: every thread reads the same word)
        for (int sg = 0; sg < nseg; ++sg)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                unsigned long long m = bits[sg][w];
                while (m) {
                    const int b = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const size_t r = (size_t)r0 + sg * 256 + w * 64 + b;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int c8 = tid + q * 256;
                        if (c8 < nc8) {
                            float f[8];
                            unpack8(ld16(d_out + r * D + c8 * 8), f);
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[q][j] += f[j];
                        }
                    }
                }
            }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c8 = tid + q * 256;
        if (c8 < nc8) st16(d_emb + ((size_t)c * V + v) * D + c8 * 8, pack8(acc[q]));
    }
}

extern "C" int umoe_codec_embed_sum_bwd(const int32_t* tok, const uint16_t* d_out, int rows, int C, int V, int D, uint16_t* d_emb,
                                        umoe_stream_t stream) {
    UMOE_REQUIRE(tok && d_out && d_emb && D % 8 == 0 && D <= 2048 * 8 && C > 0 && V > 0 && rows >= 0 && C <= 65535,
                 "umoe_codec_embed_sum_bwd: bad argument (D %% 8 == 0, D <= 16384)");
    if ((long)V > 4L * rows && rows > 0 && rows <= 0x7fffffff / 2) {
        UMOE_HIP(hipMemsetAsync(d_emb, 0, (size_t)C * V * D * 2, (hipStream_t)stream));
        codec_embed_bwd_rows_kernel<<<dim3((unsigned)rows, (unsigned)C), 256, 0, (hipStream_t)stream>>>(tok, d_out, rows, C, V, D, d_emb);
    } else {
        codec_embed_bwd_kernel<<<dim3((unsigned)V, (unsigned)C), 256, 0, (hipStream_t)stream>>>(tok, d_out, rows, C, V, D, d_emb);
    }
    UMOE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ input jitter
__global__ __launch_bounds__(256) void mul_noise_kernel(const uint16_t* __restrict__ x, const float* __restrict__ nz, long n8, uint16_t* __restrict__ y) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float f[8];
        unpack8(ld16(x + i * 8), f);
        const float4 a = *reinterpret_cast<const float4*>(nz + i * 8), b = *reinterpret_cast<const float4*>(nz + i * 8 + 4);
        f[0] *= a.x; f[1] *= a.y; f[2] *= a.z; f[3] *= a.w; f[4] *= b.x; f[5] *= b.y; f[6] *= b.z; f[7] *= b.w;
        st16(y + i * 8, pack8(f));
    }
}

extern "C" int umoe_mul_noise(const uint16_t* x, const float* noise, long n, uint16_t* y, umoe_stream_t stream) {
    UMOE_REQUIRE(x && noise && y && n >= 0 && n % 8 == 0 && ((size_t)x & 15) == 0 && ((size_t)noise & 15) == 0 && ((size_t)y & 15) == 0,
                 "umoe_mul_noise: n %% 8 == 0 and 16-byte aligned pointers");
    if (n == 0) return 0;
    const long n8 = n / 8;
    const unsigned wgs = (unsigned)((n8 + 255) / 256 > 4096 ? 4096 : (n8 + 255) / 256);
    mul_noise_kernel<<<wgs, 256, 0, (hipStream_t)stream>>>(x, noise, n8, y);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ CFG + sampler
__device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// one workgroup per (b, c) row of V logits; dynamic LDS: 3 * V floats
__global__ __launch_bounds__(256) void cfg_sample_kernel(const umoe_sample_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float shf[4];
    __shared__ int shi[4];
    __shared__ float bc_f;
    __shared__ int bc_i;
    float* x = reinterpret_cast<float*>(smem);
    float* pr = x + a.V;
    float* tmp = pr + a.V;
    const int b = blockIdx.x / a.C, c = blockIdx.x - b * a.C, tid = threadIdx.x;
    const int V = a.V, eos = a.eos;
    const int step = a.step ? *a.step : 0;
    const bool enable_eos = a.step ? (a.min_tokens < 0 || step >= a.min_tokens) : (a.enable_eos != 0);
    const float* un = a.logits + ((size_t)(2 * b) * a.C + c) * V;
    const float* co = a.logits + ((size_t)(2 * b + 1) * a.C + c) * V;
    for (int v = tid; v < V; v += 256) {
        float g = (a.cfg_scale != 0.f) ? co[v] + a.cfg_scale * (co[v] - un[v]) : co[v];
        if (enable_eos) {
            if (v > eos || (c >= 1 && v >= eos)) g = -INFINITY;
        } else if (v >= eos) {
            g = -INFINITY;
        }
        if (c == 0 && v == eos) g *= a.eos_mul;
        x[v] = g;
    }
    __syncthreads();
    auto block_argmax = [&](const float* arr) -> int {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int v = tid; v < V; v += 256) {
            const float t = arr[v];
            if (t > bv || (t == bv && v < bi)) {
                bv = t;
                bi = v;
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if ((tid & 63) == 0) {
            shf[tid >> 6] = bv;
            shi[tid >> 6] = bi;
        }
        __syncthreads();
        float rv = shf[0];
        int ri = shi[0];
        for (int w = 1; w < 4; ++w)
            if (shf[w] > rv || (shf[w] == rv && shi[w] < ri)) {
                rv = shf[w];
                ri = shi[w];
            }
        __syncthreads();
        return ri;
    };
    if (!a.do_sample || a.temperature == 0.0f) {
        const int am = block_argmax(x);
        if (tid == 0) a.pred[(size_t)b * a.C + c] = am;
        return;
    }
    // temperature, EOS-unless-arg-max (model.py:884-891)
    for (int v = tid; v < V; v += 256) x[v] = x[v] / a.temperature;
    __syncthreads();
    if (eos >= 0) {
        const int top = block_argmax(x);
        if (tid == 0 && top != eos) x[eos] = -INFINITY;
        __syncthreads();
    }
    // ---- fast path (0 < top_k <= 64): radix-select the k-th largest, then everything on <= 64 candidates -------
    if (a.top_k > 0 && a.top_k <= 64) {
        __shared__ unsigned hist[256];
        __shared__ unsigned sel_prefix, sel_remaining;
        __shared__ int wsum[4];
        __shared__ float cand_val[64];
        __shared__ int cand_idx[64];
        __shared__ int n_cand;
        auto key_of = [](float f) -> unsigned {  // monotonic: larger float -> larger key
            unsigned u = __float_as_uint(f);
            return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        };
        constexpr int PER = 8;  // contiguous indices per thread (V <= 2048)
        const int v0 = tid * PER;
        unsigned keys[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) keys[j] = (v0 + j < V) ? key_of(x[v0 + j]) : 0u;
        if (tid == 0) {
            sel_prefix = 0u;
            sel_remaining = (unsigned)min(a.top_k, V);
        }
        for (int pass = 3; pass >= 0; --pass) {
            hist[tid] = 0u;
            __syncthreads();
            const unsigned prefix = sel_prefix, himask = (pass == 3) ? 0u : (0xffffffffu << (8 * (pass + 1)));
#pragma unroll
            for (int j = 0; j < PER; ++j)
                if (v0 + j < V && (keys[j] & himask) == prefix) atomicAdd(&hist[(keys[j] >> (8 * pass)) & 255u], 1u);
            __syncthreads();
            if (tid < 64) {  // wave 0: suffix counts over the 256 bins, 4 bins per lane
                const unsigned c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
                unsigned above = c0 + c1 + c2 + c3;  // inclusive suffix sum over lanes >= tid
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned t = __shfl_down(above, o, 64);
                    if (tid + o < 64) above += t;
                }
                const unsigned rem = sel_remaining;
                const unsigned higher = above - (c0 + c1 + c2 + c3);  // elements in bins of higher lanes
                if (above >= rem && higher < rem) {  // the k-th largest lives in one of this lane's bins
                    unsigned acc2 = higher;
                    int d = 3;
                    const unsigned cs[4] = {c0, c1, c2, c3};
                    for (; d >= 0; --d) {
                        if (acc2 + cs[d] >= rem) break;
                        acc2 += cs[d];
                    }
                    sel_prefix = prefix | ((unsigned)(4 * tid + d) << (8 * pass));
                    sel_remaining = rem - acc2;  // how many of the selected bin are still needed
                }
            }
            __syncthreads();
        }
        const unsigned tkey = sel_prefix;     // key of the k-th largest value
        const int need_eq = (int)sel_remaining;  // number of == tkey elements to keep (lowest indices first)
        // exclusive scans over threads (index order): equal-to-threshold count, then kept count
        auto block_excl = [&](int v) -> int {
            int inc = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(inc, o, 64);
                if ((tid & 63) >= o) inc += t;
            }
            if ((tid & 63) == 63) wsum[tid >> 6] = inc;
            __syncthreads();
            int base = 0;
            for (int w2 = 0; w2 < (tid >> 6); ++w2) base += wsum[w2];
            __syncthreads();
            return base + inc - v;
        };
        int eq = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) eq += (v0 + j < V && keys[j] == tkey);
        int eq_before = block_excl(eq);
        int keep = 0;
        bool kj[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const bool in = v0 + j < V;
            const bool is_eq = in && keys[j] == tkey;
            kj[j] = in && (keys[j] > tkey || (is_eq && eq_before < need_eq));
            eq_before += is_eq;
            keep += kj[j];
        }
        int pos = block_excl(keep);
        if (tid == 255) n_cand = pos + keep;
#pragma unroll
        for (int j = 0; j < PER; ++j)
            if (kj[j]) {
                if (pos < 64) {
                    cand_val[pos] = x[v0 + j];
                    cand_idx[pos] = v0 + j;
                }
                ++pos;
            }
        if (a.probs_out)
            for (int v = tid; v < V; v += 256) a.probs_out[((size_t)b * a.C + c) * V + v] = 0.f;
        __syncthreads();
        if (tid >= 64) return;
        // ---- wave 0: lane = candidate, candidates are in index order ----------------------------------------
        const int n = min(n_cand, 64);
        float val = (tid < n) ? cand_val[tid] : -INFINITY;
        auto lane_softmax64 = [&](float vv) -> float {
            const float mx = wave_max(vv);
            const float e = (vv == -INFINITY) ? 0.f : expf(vv - mx);
            return e / wave_sum(e);
        };
        if (a.top_p < 1.0f) {  // model.py:899-910 restricted to the survivors of top-k
            const float pme = lane_softmax64(val);
            float before = 0.f;
            for (int j = 0; j < n; ++j) {
                const float pj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pme), j));
                if (pj > pme || (pj == pme && j < tid)) before += pj;
            }
            if (before > a.top_p) val = -INFINITY;
        }
        const float pr_me = lane_softmax64(val);
        if (a.probs_out && tid < n) a.probs_out[((size_t)b * a.C + c) * V + cand_idx[tid]] = pr_me;
        const uint64_t hsh = mix64(a.seed ^ mix64(((uint64_t)(uint32_t)step << 32) | (uint32_t)blockIdx.x));
        const float u = (float)((hsh >> 40) + 0.5) * (1.0f / 16777216.0f);
        float cum = pr_me;  // inclusive prefix in index order
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float t = __shfl_up(cum, o, 64);
            if (tid >= o) cum += t;
        }
        const unsigned long long hitm = __ballot(pr_me > 0.f && cum > u);
        const unsigned long long posm = __ballot(pr_me > 0.f);
        int pick;
        if (hitm) pick = __ffsll((long long)hitm) - 1;
        else pick = posm ? 63 - __clzll((long long)posm) : 0;
        if (tid == 0) a.pred[(size_t)b * a.C + c] = cand_idx[pick];
        return;
    }
    // top-k by rank (ties: lower index first) (model.py:893-897)
    if (a.top_k > 0) {
        for (int v = tid; v < V; v += 256) {
            const float t = x[v];
            int rank = 0;
            for (int j = 0; j < V; ++j) {
                const float u = x[j];
                rank += (u > t) || (u == t && j < v);
            }
            tmp[v] = (rank < a.top_k) ? t : -INFINITY;
        }
        __syncthreads();
        for (int v = tid; v < V; v += 256) x[v] = tmp[v];
        __syncthreads();
    }
    auto block_softmax = [&](const float* in, float* out) {
        float mx = -INFINITY;
        for (int v = tid; v < V; v += 256) mx = fmaxf(mx, in[v]);
        mx = wave_max(mx);
        if ((tid & 63) == 0) shf[tid >> 6] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(shf[0], shf[1]), fmaxf(shf[2], shf[3]));
        __syncthreads();
        float sm = 0.f;
        for (int v = tid; v < V; v += 256) {
            const float e = expf(in[v] - mx);
            out[v] = e;
            sm += e;
        }
        sm = block_sum_256(sm, shf);
        for (int v = tid; v < V; v += 256) out[v] = out[v] / sm;
        __syncthreads();
    };
    if (a.top_p < 1.0f) {  // model.py:899-910
        block_softmax(x, pr);
        for (int v = tid; v < V; v += 256) {
            const float t = pr[v];
            float before = 0.f;
            for (int j = 0; j < V; ++j) {
                const float u = pr[j];
                if ((u > t) || (u == t && j < v)) before += u;
            }
            tmp[v] = (before > a.top_p) ? -INFINITY : x[v];
        }
        __syncthreads();
        for (int v = tid; v < V; v += 256) x[v] = tmp[v];
        __syncthreads();
    }
    block_softmax(x, pr);
    if (a.probs_out)
        for (int v = tid; v < V; v += 256) a.probs_out[((size_t)b * a.C + c) * V + v] = pr[v];
    // inverse-CDF draw in index order
    if (tid == 0) {
        const uint64_t h = mix64(a.seed ^ mix64(((uint64_t)(uint32_t)step << 32) | (uint32_t)blockIdx.x));
        const float u = (float)((h >> 40) + 0.5) * (1.0f / 16777216.0f);
        float cum = 0.f;
        int pick = -1, last = 0;
        for (int v = 0; v < V; ++v) {
            const float p = pr[v];
            if (p > 0.f) {
                last = v;
                cum += p;
                if (cum > u) {
                    pick = v;
                    break;
                }
            }
        }
        a.pred[(size_t)b * a.C + c] = pick >= 0 ? pick : last;
    }
}

extern "C" int umoe_codec_head_cfg_sample(const umoe_sample_args* a, umoe_stream_t stream) {
    UMOE_REQUIRE(a && a->logits && a->pred && a->B > 0 && a->C > 0 && a->V > 1, "umoe_codec_head_cfg_sample: bad argument");
    UMOE_REQUIRE(a->eos >= 0 && a->eos < a->V, "umoe_codec_head_cfg_sample: eos %d outside vocabulary %d", a->eos, a->V);
    const size_t lds = (size_t)3 * a->V * sizeof(float);
    UMOE_REQUIRE(lds <= 60 * 1024 && a->V <= 2048, "umoe_codec_head_cfg_sample: vocabulary %d too large for the LDS sampler", a->V);
    cfg_sample_kernel<<<dim3((unsigned)(a->B * a->C)), 256, lds, (hipStream_t)stream>>>(*a);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ delay / EOS step
// state layout: eos_detected[B], countdown[B], finished[B], prefill_step[B], {step, max_tokens, all_done, bos_over}
__global__ __launch_bounds__(256) void delay_step_kernel(int64_t* pred, int32_t* tokens, int32_t* state,
                                                         const int32_t* __restrict__ delay, int B, int C, int Tmax,
                                                         int eos, int pad, int md) {
    int32_t* eos_det = state;
    int32_t* countdown = state + B;
    int32_t* finished = state + 2 * B;
    const int32_t* prefill = state + 3 * B;
    int32_t* sc = state + 4 * B;
    __shared__ int s_all_done, s_bos_over;
    const int tid = threadIdx.x;
    const int dec_step = sc[0], max_tokens = sc[1];
    // the reference checks `(eos_countdown == 0).all()` and `dec_step < max_tokens` at the loop head
    if (tid == 0) {
        int done = 1;
        for (int b = 0; b < B; ++b) done &= (countdown[b] == 0);
        s_all_done = done || (dec_step >= max_tokens);
        s_bos_over = sc[3];
    }
    __syncthreads();
    if (s_all_done) {
        if (tid == 0) sc[2] = 1;
        return;
    }
    const int cur = dec_step + 1;
    __syncthreads();
    if (tid < B) {  // per-sample EOS logic (model.py:1173-1183)
        const int b = tid;
        const int cd = countdown[b];
        const bool active = cd != 0;
        const bool trig = active && ((!eos_det[b] && pred[(size_t)b * C] == eos) || (cur >= max_tokens - md));
        if (trig) eos_det[b] = 1;
        if (trig && cd < 0) {
            countdown[b] = md;
            finished[b] = cur;
        }
    }
    __syncthreads();
    for (int i = tid; i < B * C; i += blockDim.x) {  // forced EOS / PAD by delay pattern (model.py:1185-1196)
        const int b = i / C, c = i - b * C;
        const int cd = countdown[b];
        if (cd > 0) {
            const int after = md - cd;
            if (after == delay[c]) pred[i] = eos;
            else if (after > delay[c]) pred[i] = pad;
        }
    }
    __syncthreads();
    if (tid < B && countdown[tid] > 0) countdown[tid] -= 1;
    if (tid == 0 && !s_bos_over) {  // model.py:1199-1200
        int over = 1;
        for (int b = 0; b < B; ++b) over &= (cur - prefill[b] >= md);
        sc[3] = over;
    }
    __syncthreads();
    // DecoderOutput.update_one (utils.py:290-298): keep prompt/BOS entries, fill the -1 ones
    if (cur < Tmax)
        for (int i = tid; i < B * C; i += blockDim.x) {
            const int b = i / C, c = i - b * C;
            int32_t* t = tokens + ((size_t)b * Tmax + cur) * C + c;
            if (*t == -1) *t = (int32_t)pred[i];
        }
    __syncthreads();
    if (tid == 0) {
        sc[0] = dec_step + 1;
        int done = 1;
        for (int b = 0; b < B; ++b) done &= (countdown[b] == 0);
        sc[2] = (done || (dec_step + 1 >= max_tokens)) ? 1 : 0;
    }
}

extern "C" int umoe_delay_step(int64_t* pred, int32_t* tokens, int32_t* state, const int32_t* delay, int B, int C,
                               int Tmax, int eos, int pad, int max_delay, umoe_stream_t stream) {
    UMOE_REQUIRE(pred && tokens && state && delay && B > 0 && B <= 256 && C > 0, "umoe_delay_step: bad argument (B=%d)", B);
    delay_step_kernel<<<1, 256, 0, (hipStream_t)stream>>>(pred, tokens, state, delay, B, C, Tmax, eos, pad, max_delay);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ RVQ
// from_codes: z[d][t] = sum_q ( out_w[q][d][:] . codebook[q][codes[q][t]][:] + out_b[q][d] )
// third-party descript-audio-codec 1.0.0 ResidualVectorQuantize.from_codes (reference call site
// utils/UniMoE_Audio_utils.py:123).  codes [NQ][T], codebooks [NQ][CB][cd], out_w [NQ][Dl][cd], z [Dl][T].
__global__ __launch_bounds__(256) void rvq_from_codes_kernel(const int32_t* __restrict__ codes, const float* __restrict__ cb,
                                                             const float* __restrict__ ow, const float* __restrict__ ob,
                                                             int NQ, int CB, int cd, int Dl, int T, float* __restrict__ z) {
    const int t = blockIdx.x;
    for (int d = threadIdx.x; d < Dl; d += blockDim.x) {
        float acc = 0.f;
        for (int q = 0; q < NQ; ++q) {
            int code = codes[(size_t)q * T + t];
            code = min(max(code, 0), CB - 1);
            const float* e = cb + ((size_t)q * CB + code) * cd;
            const float* w = ow + ((size_t)q * Dl + d) * cd;
            float s = ob ? ob[(size_t)q * Dl + d] : 0.f;
            for (int j = 0; j < cd; ++j) s += w[j] * e[j];
            acc += s;
        }
        z[(size_t)d * T + t] = acc;
    }
}

extern "C" int umoe_rvq_from_codes(const int32_t* codes, const float* codebooks, const float* out_w, const float* out_b,
                                   int NQ, int CB, int cd, int Dl, int T, float* z, umoe_stream_t stream) {
    UMOE_REQUIRE(codes && codebooks && out_w && z && NQ > 0 && CB > 0 && cd > 0 && Dl > 0, "umoe_rvq_from_codes: bad argument");
    if (T == 0) return 0;
    rvq_from_codes_kernel<<<dim3((unsigned)T), 256, 0, (hipStream_t)stream>>>(codes, codebooks, out_w, out_b, NQ, CB, cd, Dl, T, z);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// nearest: per level q: e = in_proj_q(resid) [cd]; code = argmax_j <normalize(e), normalize(cb_j)> (== arg-min of the
// L2 distance between the normalised vectors; ties: lowest index); resid -= out_proj_q(cb[code]).
// One workgroup per frame t; resid_ws [T][Dl] fp32 scratch.  z [Dl][T].
__global__ __launch_bounds__(256) void rvq_nearest_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                          const float* __restrict__ iw, const float* __restrict__ ib,
                                                          const float* __restrict__ ow, const float* __restrict__ ob,
                                                          int NQ, int CB, int cd, int Dl, int T, int32_t* __restrict__ codes,
                                                          float* __restrict__ resid_ws) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float shf[4];
    __shared__ int shi[4];
    float* e = reinterpret_cast<float*>(smem);  // [cd]
    const int t = blockIdx.x, tid = threadIdx.x;
    float* res = resid_ws + (size_t)t * Dl;
    for (int d = tid; d < Dl; d += 256) res[d] = z[(size_t)d * T + t];
    __syncthreads();
    for (int q = 0; q < NQ; ++q) {
        for (int j = tid; j < cd; j += 256) {
            const float* w = iw + ((size_t)q * cd + j) * Dl;
            float s = ib ? ib[(size_t)q * cd + j] : 0.f;
            for (int d = 0; d < Dl; ++d) s += w[d] * res[d];
            e[j] = s;
        }
        __syncthreads();
        float en = 0.f;
        for (int j = 0; j < cd; ++j) en += e[j] * e[j];
        en = fmaxf(sqrtf(en), 1e-12f);
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int j = tid; j < CB; j += 256) {
            const float* cj = cb + ((size_t)q * CB + j) * cd;
            float dot = 0.f, cn = 0.f;
            for (int i = 0; i < cd; ++i) {
                dot += (e[i] / en) * cj[i];
                cn += cj[i] * cj[i];
            }
            cn = fmaxf(sqrtf(cn), 1e-12f);
            // -(|a|^2 - 2 a.b + |b|^2) with a, b normalised
            const float sc = -(1.0f - 2.0f * dot / cn + 1.0f);
            if (sc > bv || (sc == bv && j < bi)) {
                bv = sc;
                bi = j;
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if ((tid & 63) == 0) {
            shf[tid >> 6] = bv;
            shi[tid >> 6] = bi;
        }
        __syncthreads();
        float rv = shf[0];
        int ri = shi[0];
        for (int w = 1; w < 4; ++w)
            if (shf[w] > rv || (shf[w] == rv && shi[w] < ri)) {
                rv = shf[w];
                ri = shi[w];
            }
        if (tid == 0) codes[(size_t)q * T + t] = ri;
        const float* cq = cb + ((size_t)q * CB + ri) * cd;
        for (int d = tid; d < Dl; d += 256) {
            const float* w = ow + ((size_t)q * Dl + d) * cd;
            float s = ob ? ob[(size_t)q * Dl + d] : 0.f;
            for (int j = 0; j < cd; ++j) s += w[j] * cq[j];
            res[d] -= s;
        }
        __syncthreads();
    }
}

extern "C" int umoe_rvq_nearest(const float* z, const float* codebooks, const float* in_w, const float* in_b,
                                const float* out_w, const float* out_b, int NQ, int CB, int cd, int Dl, int T,
                                int32_t* codes, float* resid_ws, umoe_stream_t stream) {
    UMOE_REQUIRE(z && codebooks && in_w && out_w && codes && resid_ws && NQ > 0 && CB > 0 && cd > 0 && Dl > 0,
                 "umoe_rvq_nearest: bad argument");
    if (T == 0) return 0;
    rvq_nearest_kernel<<<dim3((unsigned)T), 256, (size_t)cd * sizeof(float), (hipStream_t)stream>>>(
        z, codebooks, in_w, in_b, out_w, out_b, NQ, CB, cd, Dl, T, codes, resid_ws);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ codec cross-entropy
// reference UniMoE_Audio_model.py:830-847: for every codec channel c an nn.CrossEntropyLoss (mean over labels != -100)
// of the already shifted logits/labels; channels c != 0 without any valid label are skipped; the channel losses are
// summed.  logits [N][C][V] fp32, labels [N][C] int64.
// ce_row_kernel: one workgroup per (n, c) row -> nll[n][c] (0 where ignored); optionally the softmax row for backward.
__global__ __launch_bounds__(256) void ce_row_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, int C,
                                                     int V, float* __restrict__ nll, float* __restrict__ probs) {
    __shared__ float sh[4];
    const size_t row = blockIdx.x;  // n * C + c
    const float* x = logits + row * V;
    const int64_t lab = labels[row];
    float mx = -INFINITY;
    for (int v = threadIdx.x; v < V; v += 256) mx = fmaxf(mx, x[v]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    float sm = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) sm += expf(x[v] - mx);
    sm = block_sum_256(sm, sh);
    const float lse = mx + logf(sm);
    if (threadIdx.x == 0) nll[row] = (lab >= 0 && lab < V) ? lse - x[lab] : 0.f;
    if (probs)
        for (int v = threadIdx.x; v < V; v += 256) probs[row * V + v] = expf(x[v] - lse);
}

// ce_reduce_kernel: one workgroup per channel; fixed-order sum over n -> channel mean; thread 0 of channel 0 waits for
// nobody: the total is assembled by a second tiny launch (ce_total_kernel) to stay deterministic without atomics.
__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* __restrict__ nll, const int64_t* __restrict__ labels, int N,
                                                        int C, int V, float* __restrict__ ch_loss, int32_t* __restrict__ ch_count) {
    __shared__ float sh[4];
    const int c = blockIdx.x;
    float s = 0.f, cnt = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) {
        const int64_t lab = labels[(size_t)n * C + c];
        if (lab >= 0 && lab < V) {
            s += nll[(size_t)n * C + c];
            cnt += 1.f;
        }
    }
    s = block_sum_256(s, sh);
    cnt = block_sum_256(cnt, sh);
    if (threadIdx.x == 0) {
        ch_count[c] = (int)cnt;
        ch_loss[c] = cnt > 0.f ? s / cnt : (c == 0 ? NAN : 0.f);  // torch: mean over zero elements is NaN (channel 0 is never skipped)
    }
}

__global__ void ce_total_kernel(const float* __restrict__ ch_loss, const int32_t* __restrict__ ch_count, int C, float* total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float t = ch_loss[0];
        for (int c = 1; c < C; ++c)
            if (ch_count[c] > 0) t += ch_loss[c];
        *total = t;
    }
}

// dlogits = grad * (softmax - onehot) / count_c for valid rows, 0 elsewhere (probs from ce_row_kernel)
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ probs, const int64_t* __restrict__ labels,
                                                     const int32_t* __restrict__ ch_count, int C, int V, float grad,
                                                     float* __restrict__ dlogits) {
    const size_t row = blockIdx.x;
    const int c = (int)(row % C);
    const int64_t lab = labels[row];
    const bool valid = lab >= 0 && lab < V;
    const float sc = valid ? grad / (float)ch_count[c] : 0.f;
    for (int v = threadIdx.x; v < V; v += 256) {
        float p = probs[row * V + v];
        if (v == lab) p -= 1.f;
        dlogits[row * V + v] = sc * p;
    }
}

extern "C" int umoe_codec_ce_fwd(const float* logits, const int64_t* labels, int N, int C, int V, float* nll_ws, float* probs,
                                 float* ch_loss, int32_t* ch_count, float* total, umoe_stream_t stream) {
    UMOE_REQUIRE(logits && labels && nll_ws && ch_loss && ch_count && total && N > 0 && C > 0 && V > 1, "umoe_codec_ce_fwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ce_row_kernel<<<dim3((unsigned)((size_t)N * C)), 256, 0, s>>>(logits, labels, C, V, nll_ws, probs);
    UMOE_LAUNCH_CHECK();
    ce_reduce_kernel<<<dim3((unsigned)C), 256, 0, s>>>(nll_ws, labels, N, C, V, ch_loss, ch_count);
    UMOE_LAUNCH_CHECK();
    ce_total_kernel<<<1, 64, 0, s>>>(ch_loss, ch_count, C, total);
    UMOE_LAUNCH_CHECK();
    return 0;
}

extern "C" int umoe_codec_ce_bwd(const float* probs, const int64_t* labels, const int32_t* ch_count, int N, int C, int V, float grad,
                                 float* dlogits, umoe_stream_t stream) {
    UMOE_REQUIRE(probs && labels && ch_count && dlogits && N > 0, "umoe_codec_ce_bwd: bad argument");
    ce_bwd_kernel<<<dim3((unsigned)((size_t)N * C)), 256, 0, (hipStream_t)stream>>>(probs, labels, ch_count, C, V, grad, dlogits);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ aux load-balancing loss
// reference audio_load_balancing_loss_func, core.py:361-389: softmax over the n_dyn columns of the logits with
// unselected experts filled with finfo.min, then n_dyn * sum_e mean_s(mask[s,e]) * mean_s(prob[s,e]) (optionally
// token-weighted).  One workgroup; each thread owns tokens s = tid, tid+256, ...; fixed-order block reductions.
// T-arithmetic follows the logits dtype (softmax in T as torch does); the means accumulate in fp32.
__global__ __launch_bounds__(256) void aux_loss_kernel(const void* __restrict__ logits, int logits_bf16, const int32_t* __restrict__ mask,
                                                       const float* __restrict__ tok_w, int S, int E, int n_dyn, float* out) {
    __shared__ float sh[4];
    float fm[UMOE_MAXE], fp[UMOE_MAXE];
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e) fm[e] = fp[e] = 0.f;
    float wsum = 0.f;
    for (int s = threadIdx.x; s < S; s += 256) {
        float x[UMOE_MAXE];
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < UMOE_MAXE; ++e)
            if (e < n_dyn) {
                const float l = logits_bf16 ? bf2f(reinterpret_cast<const uint16_t*>(logits)[(size_t)s * E + e])
                                            : reinterpret_cast<const float*>(logits)[(size_t)s * E + e];
                // masked_fill(mask == 0, finfo(dtype).min)
                x[e] = mask[(size_t)s * E + e] ? l : (logits_bf16 ? -3.3895313892515355e38f : -3.4028234663852886e38f);
                mx = fmaxf(mx, x[e]);
            }
        float sm = 0.f, ex[UMOE_MAXE];
#pragma unroll
        for (int e = 0; e < UMOE_MAXE; ++e)
            if (e < n_dyn) {
                ex[e] = expf(x[e] - mx);
                sm += ex[e];
            }
        const float w = tok_w ? tok_w[s] : 1.f;
        wsum += w;
#pragma unroll
        for (int e = 0; e < UMOE_MAXE; ++e)
            if (e < n_dyn) {
                const float p = round_t(ex[e] / sm, logits_bf16);
                fm[e] += w * (float)mask[(size_t)s * E + e];
                fp[e] += w * p;
            }
    }
    wsum = block_sum_256(wsum, sh);
    float total = 0.f;
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e)
        if (e < n_dyn) {
            const float a = block_sum_256(fm[e], sh), b = block_sum_256(fp[e], sh);
            // torch.mean of the bf16 probabilities returns bf16 (fp32 accumulate, ONE rounding, core.py:378); the token-weighted
            // branch multiplies by the fp32 weights first and stays in fp32 (core.py:384-385)
            const float pm = (!tok_w && logits_bf16) ? rbf(b / wsum) : b / wsum;
            total += (a / wsum) * pm;
        }
    if (threadIdx.x == 0) *out = total * (float)n_dyn;
}

// Many tokens (training): the same sums in two launches -- AUX_PARTS workgroups take contiguous token ranges and leave fixed-order
// partial sums in the caller's workspace, one small workgroup adds the parts in order and finishes the formula.  (The one-workgroup
// kernel above took 74 us at 6 240 tokens: 25 dependent rounds of loads per thread.)
#define AUX_PARTS 64
__global__ __launch_bounds__(256) void aux_loss_part_kernel(const void* __restrict__ logits, int logits_bf16, const int32_t* __restrict__ mask,
                                                            const float* __restrict__ tok_w, int S, int E, int n_dyn, float* __restrict__ ws) {
    __shared__ float sh[4];
    float fm[UMOE_MAXE], fp[UMOE_MAXE];
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e) fm[e] = fp[e] = 0.f;
    float wsum = 0.f;
    const int per = (S + AUX_PARTS - 1) / AUX_PARTS, s0 = blockIdx.x * per, s1 = min(S, s0 + per);
    for (int s = s0 + threadIdx.x; s < s1; s += 256) {
        float x[UMOE_MAXE];
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < UMOE_MAXE; ++e)
            if (e < n_dyn) {
                const float l = logits_bf16 ? bf2f(reinterpret_cast<const uint16_t*>(logits)[(size_t)s * E + e])
                                            : reinterpret_cast<const float*>(logits)[(size_t)s * E + e];
                x[e] = mask[(size_t)s * E + e] ? l : (logits_bf16 ? -3.3895313892515355e38f : -3.4028234663852886e38f);
                mx = fmaxf(mx, x[e]);
            }
        float sm = 0.f, ex[UMOE_MAXE];
#pragma unroll
        for (int e = 0; e < UMOE_MAXE; ++e)
            if (e < n_dyn) {
                ex[e] = expf(x[e] - mx);
                sm += ex[e];
            }
        const float w = tok_w ? tok_w[s] : 1.f;
        wsum += w;
#pragma unroll
        for (int e = 0; e < UMOE_MAXE; ++e)
            if (e < n_dyn) {
                const float p = round_t(ex[e] / sm, logits_bf16);
                fm[e] += w * (float)mask[(size_t)s * E + e];
                fp[e] += w * p;
            }
    }
    float* o = ws + (size_t)blockIdx.x * (2 * UMOE_MAXE + 1);
    wsum = block_sum_256(wsum, sh);
    if (threadIdx.x == 0) o[2 * UMOE_MAXE] = wsum;
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e)
        if (e < n_dyn) {
            const float a = block_sum_256(fm[e], sh), b = block_sum_256(fp[e], sh);
            if (threadIdx.x == 0) {
                o[e] = a;
                o[UMOE_MAXE + e] = b;
            }
        }
}
__global__ __launch_bounds__(64) void aux_loss_fin_kernel(const float* __restrict__ ws, int logits_bf16, int has_w, int n_dyn, float* out) {
    const int e = threadIdx.x;
    float a = 0.f, b = 0.f, wsum = 0.f;
    for (int q = 0; q < AUX_PARTS; ++q) {       // fixed order
        const float* o = ws + (size_t)q * (2 * UMOE_MAXE + 1);
        wsum += o[2 * UMOE_MAXE];
        if (e < n_dyn) {
            a += o[e];
            b += o[UMOE_MAXE + e];
        }
    }
    float term = 0.f;
    if (e < n_dyn) {
        const float pm = (!has_w && logits_bf16) ? rbf(b / wsum) : b / wsum;
        term = (a / wsum) * pm;
    }
    // ascending expert order, as the one-workgroup kernel adds them
    float total = 0.f;
    for (int q = 0; q < n_dyn; ++q) total += __shfl(term, q, 64);
    if (e == 0) *out = total * (float)n_dyn;
}

extern "C" int umoe_aux_loss_fwd_ws(const void* logits, int logits_bf16, const int32_t* expert_mask, const float* token_weight, int S,
                                    int E, int n_dyn, float* out, float* ws, umoe_stream_t stream) {
    UMOE_REQUIRE(logits && expert_mask && out && ws && S > 0 && n_dyn >= 1 && n_dyn <= E && E <= UMOE_MAXE, "umoe_aux_loss_fwd_ws: bad argument");
    aux_loss_part_kernel<<<AUX_PARTS, 256, 0, (hipStream_t)stream>>>(logits, logits_bf16, expert_mask, token_weight, S, E, n_dyn, ws);
    aux_loss_fin_kernel<<<1, 64, 0, (hipStream_t)stream>>>(ws, logits_bf16, token_weight != nullptr, n_dyn, out);
    UMOE_LAUNCH_CHECK();
    return 0;
}
extern "C" size_t umoe_aux_loss_workspace_floats(void) { return (size_t)AUX_PARTS * (2 * UMOE_MAXE + 1); }

extern "C" int umoe_aux_loss_fwd(const void* logits, int logits_bf16, const int32_t* expert_mask, const float* token_weight, int S,
                                 int E, int n_dyn, float* out, umoe_stream_t stream) {
    UMOE_REQUIRE(logits && expert_mask && out && S > 0 && n_dyn >= 1 && n_dyn <= E && E <= UMOE_MAXE, "umoe_aux_loss_fwd: bad argument");
    aux_loss_kernel<<<1, 256, 0, (hipStream_t)stream>>>(logits, logits_bf16, expert_mask, token_weight, S, E, n_dyn, out);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ Infinity Cache warm-up
__global__ __launch_bounds__(256) void prefetch_kernel(const uint4* __restrict__ p, size_t n16, uint32_t* sink) {
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += 8 * stride) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (i + j * stride < n16) ? p[i + j * stride] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x9e3779b9u && sink) *sink = acc;   // never true in practice; keeps the loads alive
}

extern "C" int umoe_prefetch(const void* p, size_t bytes, int wgs, umoe_stream_t stream) {
    UMOE_REQUIRE(p && ((size_t)p & 15) == 0 && wgs > 0 && wgs <= 65535, "umoe_prefetch: bad argument");
    if (bytes < 16) return 0;
    prefetch_kernel<<<dim3((unsigned)wgs), 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const uint4*>(p), bytes / 16, nullptr);
    UMOE_LAUNCH_CHECK();
    return 0;
}
