// Backward pieces of the DCMoE block and of the decoder layer's norms (the *_bwd entry points).
// The contractions (dX = dY W, dW = dY^T X) run on umoe_tiled_gemm: its operands are K-contiguous, so the weight
// gradients contract over SLOT COLUMNS of transposed buffers built here (8-aligned slot ranges, zero padded), and the
// input gradients use transposed weight copies.  Everything else -- SwiGLU, combine, permute, router, RMSNorm -- is
// elementwise / per-token work: HBM bound, fp32 arithmetic, one rounding to bf16 where autograd would round.
#include "umoe_common.h"
#include <string.h>

// ------------------------------------------------------------------------------------ transposes
// dst[c][off_g + r] = src[row(off_g + r)][c] for r < cnt_g, 0 for cnt_g <= r < roundup8(cnt_g); row(s) = rows ? rows[s] : s.
// grid = (slot tiles of TR_TS, column tiles of 64, groups).  counts == NULL: one group of `static_rows` rows at offset 0.
// Tile = TR_TS slots x 64 columns.  Measured in the training step (scripts/train_bench.py, same box): TR_TS 64 (128-byte segments on both
// sides) 241.9 / 245.6 ms, TR_TS 128 (256 contiguous bytes per destination row and tile, 18 KB of LDS) 249.0 / 249.1 ms -- 64 stays.
#ifndef TR_TS
#define TR_TS 64
#endif
__device__ __forceinline__ void transpose_tile(const uint16_t* __restrict__ src, const int ld_src, const int C,
                                               const int32_t* __restrict__ rows, const int cnt, const int off,
                                               uint16_t* __restrict__ dst, const int ld_dst, const bool compact = false) {
    __shared__ uint16_t tile[TR_TS][72];   // [slot][col], 144-byte rows: 16-byte aligned chunks, bank spread
    const int pad = (cnt + 7) & ~7;
    const int r0 = blockIdx.x * TR_TS, c0 = blockIdx.y * 64;
    if (r0 >= pad) return;
    const int tid = threadIdx.x;
    // load: thread (tr = tid / 8, ch = tid % 8) reads 8 columns of slots tr, tr + 32, ...; every load is requested before the first
    // LDS write (straight-line: rows beyond the count re-read the group's first row and are zeroed)
    uint4 v[TR_TS / 32];
    const int ch = tid & 7, c = c0 + ch * 8;
    const bool vec = c + 8 <= C && (ld_src & 7) == 0;
#pragma unroll
    for (int ps = 0; ps < TR_TS / 32; ++ps) {
        const int r = (tid >> 3) + 32 * ps;
        const bool live = r0 + r < cnt && c < C;
        const int rr = live ? r0 + r : 0;
        v[ps] = make_uint4(0, 0, 0, 0);
        if (vec) {
            const long srow = rows ? (long)rows[off + rr] : (long)(off + rr);
            const uint4 t4 = ld16(src + srow * ld_src + c);
            if (live) v[ps] = t4;
        } else if (live) {
            const long srow = rows ? (long)rows[off + rr] : (long)(off + rr);
            const uint16_t* sp = src + srow * ld_src + c;
            uint16_t t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = (c + q < C) ? sp[q] : (uint16_t)0;
            v[ps] = make_uint4((uint32_t)t[0] | ((uint32_t)t[1] << 16), (uint32_t)t[2] | ((uint32_t)t[3] << 16),
                               (uint32_t)t[4] | ((uint32_t)t[5] << 16), (uint32_t)t[6] | ((uint32_t)t[7] << 16));
        }
    }
#pragma unroll
    for (int ps = 0; ps < TR_TS / 32; ++ps) *reinterpret_cast<uint4*>(&tile[(tid >> 3) + 32 * ps][ch * 8]) = v[ps];
    __syncthreads();
    // store: thread (c = tid / 16 (+16 per pass), sc = tid % 16) writes 8 consecutive slots of column c: 256 contiguous bytes per column
    constexpr int SPC = TR_TS / 8;          // 16-byte chunks per column of the tile
#pragma unroll
    for (int ps = 0; ps < 64 / (256 / SPC); ++ps) {
        const int cc = tid / SPC + (256 / SPC) * ps, sc = tid % SPC;
        if (c0 + cc >= C || r0 + sc * 8 >= pad) continue;
        uint16_t t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = tile[sc * 8 + q][cc];
        const uint4 o = make_uint4((uint32_t)t[0] | ((uint32_t)t[1] << 16), (uint32_t)t[2] | ((uint32_t)t[3] << 16),
                                   (uint32_t)t[4] | ((uint32_t)t[5] << 16), (uint32_t)t[6] | ((uint32_t)t[7] << 16));
        // compact: the group's own block (C rows of `pad` elements) at dst + off * C; else column window [off, off + pad) of wide rows
        if (compact) st16(dst + (size_t)off * C + (size_t)(c0 + cc) * pad + r0 + sc * 8, o);
        else st16(dst + (size_t)(c0 + cc) * ld_dst + off + r0 + sc * 8, o);
    }
}

__global__ __launch_bounds__(256) void transpose_slots_kernel(const uint16_t* __restrict__ src, int ld_src, int C,
                                                              const int32_t* __restrict__ rows, const int32_t* __restrict__ counts,
                                                              const int32_t* __restrict__ offsets, int static_rows,
                                                              uint16_t* __restrict__ dst, int ld_dst, int compact) {
    const int g = blockIdx.z;
    transpose_tile(src, ld_src, C, rows, counts ? counts[g] : static_rows, offsets ? offsets[g] : 0, dst, ld_dst, compact != 0);
}

// the same plain transpose [R][C] -> [C][r8(R)] of up to 12 separate matrices of one shape in ONE launch (the per-expert weight
// copies of the backward composites: 30 launches of ~10 us each per layer became 5)
struct trans_ptrs { const uint16_t* src[12]; uint16_t* dst[12]; };
__global__ __launch_bounds__(256) void transpose_multi_kernel(const trans_ptrs p, int ld_src, int C, int R, int ld_dst) {
    transpose_tile(p.src[blockIdx.z], ld_src, C, nullptr, R, 0, p.dst[blockIdx.z], ld_dst);
}
static int transpose_multi(const uint16_t* const* src, uint16_t* const* dst, int n, int ld_src, int C, int R, int ld_dst, umoe_stream_t stream) {
    UMOE_REQUIRE(n >= 1 && n <= 12 && (ld_dst & 7) == 0, "transpose_multi: 1..12 matrices, ld_dst %% 8");
    trans_ptrs p{};
    for (int i = 0; i < n; ++i) { p.src[i] = src[i]; p.dst[i] = dst[i]; }
    dim3 grid((unsigned)ceil_div(((R + 7) & ~7), TR_TS), (unsigned)ceil_div(C, 64), (unsigned)n);
    transpose_multi_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(p, ld_src, C, R, ld_dst);
    UMOE_LAUNCH_CHECK();
    return 0;
}

extern "C" int umoe_transpose_slots(const uint16_t* src, int ld_src, int C, const int32_t* rows, const int32_t* counts,
                                    const int32_t* offsets, int n_groups, int max_rows, uint16_t* dst, int ld_dst,
                                    umoe_stream_t stream) {
    UMOE_REQUIRE(src && dst && C > 0 && max_rows >= 0 && (ld_dst & 7) == 0, "umoe_transpose_slots: bad argument (ld_dst %% 8)");
    UMOE_REQUIRE((counts == nullptr) == (offsets == nullptr), "umoe_transpose_slots: counts and offsets come together");
    if (max_rows == 0) return 0;
    const int G = counts ? n_groups : 1;
    UMOE_REQUIRE(G >= 1 && G <= 65535, "umoe_transpose_slots: bad group count");
    dim3 grid((unsigned)ceil_div(((max_rows + 7) & ~7), TR_TS), (unsigned)ceil_div(C, 64), (unsigned)G);
    transpose_slots_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(src, ld_src, C, rows, counts, offsets, max_rows, dst, ld_dst, 0);
    UMOE_LAUNCH_CHECK();
    return 0;
}

extern "C" int umoe_transpose_slots_compact(const uint16_t* src, int ld_src, int C, const int32_t* rows, const int32_t* counts,
                                            const int32_t* offsets, int n_groups, int max_rows, uint16_t* dst, umoe_stream_t stream) {
    UMOE_REQUIRE(src && dst && counts && offsets && C > 0 && max_rows >= 0 && n_groups >= 1 && n_groups <= 65535,
                 "umoe_transpose_slots_compact: bad argument (counts and offsets are required)");
    if (max_rows == 0) return 0;
    dim3 grid((unsigned)ceil_div(((max_rows + 7) & ~7), TR_TS), (unsigned)ceil_div(C, 64), (unsigned)n_groups);
    transpose_slots_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(src, ld_src, C, rows, counts, offsets, max_rows, dst, 0, 1);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ SwiGLU backward
// h = silu(g) * u (core.py:31,49).  autograd on bf16 tensors: d_silu = bf16(dh * u), du = bf16(dh * silu(g)),
// dg = bf16(d_silu * sigmoid(g) * (1 + g * (1 - sigmoid(g)))) (torch's silu backward, fp32 inside).
// gu [rows][2I] = (g | u) saved by the forward; dgu [rows][2I] = (dg | du).  Rows are the slot rows [0, *total).
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const uint16_t* __restrict__ dh, int ld_dh, const uint16_t* __restrict__ gu,
                                                         int ld_gu, int I, const int32_t* __restrict__ total, int static_rows,
                                                         uint16_t* __restrict__ dgu, int ld_dgu) {
    const int rows = total ? *total : static_rows;
    const int nch = I >> 3;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < (long)rows * nch; idx += (long)gridDim.x * 256) {
        const long r = idx / nch;
        const int c = (int)(idx - r * nch) * 8;
        float d[8], g[8], u[8], og[8], ou[8];
        unpack8(ld16(dh + r * ld_dh + c), d);
        unpack8(ld16(gu + r * ld_gu + c), g);
        unpack8(ld16(gu + r * ld_gu + I + c), u);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float sg = 1.0f / (1.0f + expf(-g[j]));
            const float silu = rbf(g[j] * sg);
            ou[j] = d[j] * silu;
            const float ds = rbf(d[j] * u[j]);
            og[j] = ds * (sg * (1.0f + g[j] * (1.0f - sg)));
        }
        st16(dgu + r * ld_dgu + c, pack8(og));
        st16(dgu + r * ld_dgu + I + c, pack8(ou));
    }
}

extern "C" int umoe_swiglu_bwd(const uint16_t* dh, int ld_dh, const uint16_t* gu, int ld_gu, int I, const int32_t* total_rows,
                               int max_rows, uint16_t* dgu, int ld_dgu, umoe_stream_t stream) {
    UMOE_REQUIRE(dh && gu && dgu && I > 0 && I % 8 == 0 && (ld_dh & 7) == 0 && (ld_gu & 7) == 0 && (ld_dgu & 7) == 0,
                 "umoe_swiglu_bwd: bad argument (I and leading dimensions must be multiples of 8)");
    if (max_rows <= 0) return 0;
    const long work = (long)max_rows * (I >> 3);
    const unsigned blocks = (unsigned)((work + 255) / 256 > 4096 ? 4096 : (work + 255) / 256);
    swiglu_bwd_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(dh, ld_dh, gu, ld_gu, I, total_rows, max_rows, dgu, ld_dgu);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ combine backward
// forward (umoe_unpermute_combine_fwd): out[s] = sum_e w[s][e] * y[slot(s,e)] + sum_i gw[s][n_dyn+i] * ysh[i][s] (+ resid).
// backward per token s: dy[slot] = bf16(w * dout[s]), dw[s][e] = <dout[s], y[slot]>, same for the shared experts.
__global__ __launch_bounds__(256) void combine_bwd_kernel(const uint16_t* __restrict__ dout, const umoe_combine_args a,
                                                          uint16_t* __restrict__ dy_slots, uint16_t* __restrict__ dy_shared,
                                                          float* __restrict__ d_moe_w, float* __restrict__ d_gw_shared) {
    __shared__ float sh[4];
    const int s = blockIdx.x, E = a.n_dyn + a.n_fix;
    const int nch = a.D >> 3;
    if (nch == 256 && a.n_real <= UMOE_MAXE - 4 && a.n_fix <= 4 && (a.y_shared || a.n_fix == 0)) {
        // D = 2048 (one 16-byte chunk per thread), every expert row of the token requested at once: the loop below walks the experts one
        // dependent round trip (slot -> row) and two barriers at a time (78 us per launch at 6 240 tokens).  Same products, same
        // reduction (wave sums, then the four wave partials in order: block_sum_256).
        __shared__ float part[4][UMOE_MAXE];
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = tid;
        const int NE = a.n_real + (a.y_shared ? a.n_fix : 0);
        int slot_l = -1;
        float w_l = 0.f;
        if (lane < a.n_real) {
            slot_l = a.slot_of[(size_t)s * a.n_real + lane];
            w_l = a.moe_w[(size_t)s * a.n_real + lane];
        } else if (lane < NE) {
            slot_l = (lane - a.n_real) * a.S + s;
            w_l = a.global_w[(size_t)s * E + a.n_dyn + (lane - a.n_real)];
        }
        float d[8];
        unpack8(ld16(dout + (size_t)s * a.D + c * 8), d);
        float dot[UMOE_MAXE];
#pragma unroll
        for (int e = 0; e < UMOE_MAXE; ++e) {
            dot[e] = 0.f;
            if (e < NE) {
                const int row = __builtin_amdgcn_readlane(slot_l, e);
                const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w_l), e));
                if (row >= 0) {
                    const bool shd = e >= a.n_real;
                    float yv[8], o[8];
                    unpack8(ld16((shd ? a.y_shared : a.y_slots) + (size_t)row * a.D + c * 8), yv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        dot[e] += d[j] * yv[j];
                        o[j] = w * d[j];
                    }
                    st16((shd ? dy_shared : dy_slots) + (size_t)row * a.D + c * 8, pack8(o));
                }
            }
        }
#pragma unroll
        for (int e = 0; e < UMOE_MAXE; ++e)
            if (e < NE) {
                const float v = wave_sum(dot[e]);
                if (lane == 0) part[wave][e] = v;
            }
        __syncthreads();
        if (tid < NE) {
            const float t = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
            if (tid < a.n_real) d_moe_w[(size_t)s * a.n_real + tid] = t;        // (an expert the token did not pick: no products, 0)
            else d_gw_shared[(size_t)s * a.n_fix + (tid - a.n_real)] = t;
        }
        return;
    }
    for (int e = 0; e < a.n_real + a.n_fix; ++e) {
        const bool shd = e >= a.n_real;
        const int i = e - a.n_real;
        long row;
        float w;
        if (!shd) {
            const int slot = a.slot_of[(size_t)s * a.n_real + e];
            if (slot < 0) {
                if (threadIdx.x == 0) d_moe_w[(size_t)s * a.n_real + e] = 0.f;
                continue;
            }
            row = slot;
            w = a.moe_w[(size_t)s * a.n_real + e];
        } else {
            if (!a.y_shared) break;
            row = (long)i * a.S + s;
            w = a.global_w[(size_t)s * E + a.n_dyn + i];
        }
        const uint16_t* y = (shd ? a.y_shared : a.y_slots) + row * a.D;
        uint16_t* dy = (shd ? dy_shared : dy_slots) + row * a.D;
        float dot = 0.f;
        for (int c = threadIdx.x; c < nch; c += 256) {
            float d[8], yv[8], o[8];
            unpack8(ld16(dout + (size_t)s * a.D + c * 8), d);
            unpack8(ld16(y + c * 8), yv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                dot += d[j] * yv[j];
                o[j] = w * d[j];
            }
            st16(dy + c * 8, pack8(o));
        }
        dot = block_sum_256(dot, sh);
        if (threadIdx.x == 0) {
            if (!shd) d_moe_w[(size_t)s * a.n_real + e] = dot;
            else d_gw_shared[(size_t)s * a.n_fix + i] = dot;
        }
        __syncthreads();
    }
}

extern "C" int umoe_unpermute_combine_bwd(const uint16_t* dout, const umoe_combine_args* a, uint16_t* dy_slots,
                                          uint16_t* dy_shared, float* d_moe_w, float* d_gw_shared, umoe_stream_t stream) {
    UMOE_REQUIRE(dout && a && a->y_slots && a->slot_of && a->moe_w && dy_slots && d_moe_w && a->D % 8 == 0,
                 "umoe_unpermute_combine_bwd: bad argument");
    UMOE_REQUIRE(!a->y_shared || (a->global_w && dy_shared && d_gw_shared), "umoe_unpermute_combine_bwd: shared experts need global_w and outputs");
    if (a->S == 0) return 0;
    combine_bwd_kernel<<<dim3((unsigned)a->S), 256, 0, (hipStream_t)stream>>>(dout, *a, dy_slots, dy_shared, d_moe_w, d_gw_shared);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ permute backward
// dx[s] = bf16( sum_{e: slot(s,e) >= 0} dxe[slot] + sum_i dxsh[i][s] + extra[s] ), fp32 accumulation in expert order.
__global__ __launch_bounds__(256) void permute_bwd_kernel(const uint16_t* __restrict__ dxe, const int32_t* __restrict__ slot_of,
                                                          int n_real, const uint16_t* __restrict__ dxsh, int n_fix, int S, int D,
                                                          const uint16_t* __restrict__ extra, uint16_t* __restrict__ dx) {
    const int s = blockIdx.x;
    if ((D >> 3) == 256 && n_real <= 12 && n_fix <= 4) {
        // D = 2048: every row of the token requested at once (the loop below reads slot -> row one dependent round trip at a time);
        // the same additions in the same order
        const int tid = threadIdx.x, lane = tid & 63, c = tid;
        int slot_l = -1;
        if (lane < n_real) slot_l = slot_of[(size_t)s * n_real + lane];
        uint4 v[16];
#pragma unroll
        for (int e = 0; e < 12; ++e)
            if (e < n_real) {
                const int slot = __builtin_amdgcn_readlane(slot_l, e);
                v[e] = ld16(dxe + (size_t)(slot >= 0 ? slot : 0) * D + c * 8);
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < n_fix) v[12 + i] = ld16(dxsh + ((size_t)i * S + s) * D + c * 8);
        uint4 xv = make_uint4(0, 0, 0, 0);
        if (extra) xv = ld16(extra + (size_t)s * D + c * 8);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int e = 0; e < 12; ++e)
            if (e < n_real && __builtin_amdgcn_readlane(slot_l, e) >= 0) {
                float f[8];
                unpack8(v[e], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += f[j];
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < n_fix) {
                float f[8];
                unpack8(v[12 + i], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += f[j];
            }
        if (extra) {
            float f[8];
            unpack8(xv, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += f[j];
        }
        st16(dx + (size_t)s * D + c * 8, pack8(acc));
        return;
    }
    for (int c = threadIdx.x; c < (D >> 3); c += 256) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int e = 0; e < n_real; ++e) {
            const int slot = slot_of[(size_t)s * n_real + e];
            if (slot < 0) continue;
            float v[8];
            unpack8(ld16(dxe + (size_t)slot * D + c * 8), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += v[j];
        }
        for (int i = 0; i < n_fix; ++i) {
            float v[8];
            unpack8(ld16(dxsh + ((size_t)i * S + s) * D + c * 8), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += v[j];
        }
        if (extra) {
            float v[8];
            unpack8(ld16(extra + (size_t)s * D + c * 8), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += v[j];
        }
        st16(dx + (size_t)s * D + c * 8, pack8(acc));
    }
}

extern "C" int umoe_permute_bwd(const uint16_t* dxe, const int32_t* slot_of, int n_real, const uint16_t* dx_shared, int n_fix,
                                int S, int D, const uint16_t* extra, uint16_t* dx, umoe_stream_t stream) {
    UMOE_REQUIRE(dxe && slot_of && dx && D % 8 == 0 && n_real >= 0 && (n_fix == 0 || dx_shared), "umoe_permute_bwd: bad argument");
    if (S == 0) return 0;
    permute_bwd_kernel<<<dim3((unsigned)S), 256, 0, (hipStream_t)stream>>>(dxe, slot_of, n_real, dx_shared, n_fix, S, D, extra, dx);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ router backward
// Gradient of (moe_w, shared global weights) with respect to the router logits, shipped configuration
// (ignore_differentiable_router: the mixer runs its eval branch, core.py:115-119,272; gradients flow through the
// per-round softmax multipliers, the renormalisation core.py:284 and the global softmax core.py:178-193).
// One thread per token; the forward chain is recomputed in fp32 from the saved logits and selection order.
//   rounds j < k: A_j = {unselected a : (max_j - z_a) / max(|z_a|, |max_j|) <= 2 eps}, p_j = softmax_{A_j}(z), rw[e_j] = p_j[e_j]
//   r = rw / (sum rw + 1e-6);  G = softmax_{mask != 0}(z over all E);  Gd = sum_{e < n_dyn} G[e]
//   gw[e < n_dyn] = r[e] * Gd,  gw[n_dyn + i] = G[n_dyn + i];  moe_w[e < n_real] = gw[e] * mask[e]
__global__ __launch_bounds__(256) void router_bwd_kernel(const void* __restrict__ logits, int logits_bf16, const int32_t* __restrict__ sel,
                                                         const int64_t* __restrict__ top_k, const int32_t* __restrict__ mask,
                                                         const float* __restrict__ d_moe_w, const float* __restrict__ d_gw_shared,
                                                         const float* __restrict__ d_logits_in, int S, int n_dyn, int n_real, int n_fix,
                                                         float jitter_eps, float* __restrict__ d_logits, int drop,
                                                         const float* __restrict__ round_factor) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    const int E = n_dyn + n_fix;
    float z[UMOE_MAXE], dz[UMOE_MAXE], rw[UMOE_MAXE], drw[UMOE_MAXE];
    int m[UMOE_MAXE];
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e) {
        z[e] = -INFINITY; dz[e] = 0.f; rw[e] = 0.f; drw[e] = 0.f; m[e] = 0;
        if (e < E) {
            z[e] = logits_bf16 ? bf2f(reinterpret_cast<const uint16_t*>(logits)[(size_t)s * E + e])
                               : reinterpret_cast<const float*>(logits)[(size_t)s * E + e];
            m[e] = mask[(size_t)s * E + e];
        }
    }
    const int k = (int)top_k[s];
    // ---- forward recompute: routing weights of the selected columns ----
    {
        bool taken[UMOE_MAXE];
#pragma unroll
        for (int e = 0; e < UMOE_MAXE; ++e) taken[e] = false;
        for (int j = 0; j < k; ++j) {
            const int ej = sel[(size_t)s * n_dyn + j];
            float mx = -INFINITY;
#pragma unroll
            for (int e = 0; e < UMOE_MAXE; ++e)
                if (e < n_dyn && !taken[e]) mx = fmaxf(mx, z[e]);
            float den = 0.f;
#pragma unroll
            for (int e = 0; e < UMOE_MAXE; ++e)
                if (e < n_dyn && !taken[e]) {
                    const float f = fmaxf(fabsf(z[e]), fabsf(mx));
                    if (!((mx - z[e]) / f > 2.f * jitter_eps)) den += expf(z[e] - mx);
                }
#pragma unroll
            for (int e = 0; e < UMOE_MAXE; ++e)
                if (e == ej) {
                    // differentiable router (training branch of the mixer, core.py:111-137): the forward multiplies the softmax
                    // multiplier by mask_for_one (1 or 0.3333); AudioMoERoutingFunction.backward (core.py:64-91) ignores that factor
                    // and is otherwise the softmax gradient used below
                    rw[e] = expf(z[e] - mx) / den * (round_factor ? round_factor[(size_t)s * n_dyn + j] : 1.f);
                    taken[e] = true;
                }
        }
    }
    float R = 0.f;
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e)
        if (e < n_dyn) R += rw[e];
    const float Rinv = 1.f / (R + 1e-6f);
    // global softmax over the active columns
    float G[UMOE_MAXE];
    float gmx = -INFINITY;
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e)
        if (e < E && m[e]) gmx = fmaxf(gmx, z[e]);
    float gden = 0.f;
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e) {
        G[e] = (e < E && m[e]) ? expf(z[e] - gmx) : 0.f;
        gden += G[e];
    }
    float Gd = 0.f;
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e) {
        G[e] = gden > 0.f ? G[e] / gden : 0.f;
        if (e < n_dyn) Gd += G[e];
    }
    // ---- backward ----
    float dG[UMOE_MAXE], dr[UMOE_MAXE];
    float dGd = 0.f;
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e) {
        dr[e] = 0.f;
        dG[e] = 0.f;
        if (e < n_real) {
            const float dgw = d_moe_w[(size_t)s * n_real + e] * (float)(m[e] != 0);   // moe_w = gw * mask
            dr[e] = dgw * Gd;
            dGd += dgw * rw[e] * Rinv;
        }
    }
    float Q = 0.f;       // token drop (core.py:328-329): r2 = q / (Q + 1e-6), q = r * mask_after_drop; gw = r2 * Gd
    if (drop) {
#pragma unroll
        for (int e = 0; e < UMOE_MAXE; ++e)
            if (e < n_dyn && m[e] != 0) Q += rw[e] * Rinv;
        const float Qinv = 1.f / (Q + 1e-6f);
        dGd = 0.f;
        float qdot = 0.f;   // sum_j dr2[j] * q[j]
#pragma unroll
        for (int e = 0; e < UMOE_MAXE; ++e)
            if (e < n_real && m[e] != 0) {
                const float dgw = d_moe_w[(size_t)s * n_real + e];
                const float q = rw[e] * Rinv;
                dGd += dgw * q * Qinv;
                qdot += dgw * Gd * q;
            }
#pragma unroll
        for (int e = 0; e < UMOE_MAXE; ++e) {
            // dr2[e] = dgw[e] * Gd on kept real experts; dq = dr2 * Qinv - qdot * Qinv^2 on every kept dynamic column; dr = dq * mask
            float dq = 0.f;
            if (e < n_dyn && m[e] != 0) dq = (e < n_real ? d_moe_w[(size_t)s * n_real + e] * Gd : 0.f) * Qinv - qdot * Qinv * Qinv;
            dr[e] = dq;
        }
    }
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e) {
        if (e < n_dyn) dG[e] = dGd;
        else if (e < E) dG[e] = d_gw_shared ? d_gw_shared[(size_t)s * n_fix + (e - n_dyn)] : 0.f;
    }
    float gdot = 0.f;
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e) gdot += G[e] * dG[e];
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e) dz[e] += G[e] * (dG[e] - gdot);
    // renormalisation: r = rw * Rinv
    float rdot = 0.f;
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e) rdot += dr[e] * rw[e];
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e)
        if (e < n_dyn) drw[e] = dr[e] * Rinv - rdot * Rinv * Rinv;
    // per-round softmax multipliers
    {
        bool taken[UMOE_MAXE];
#pragma unroll
        for (int e = 0; e < UMOE_MAXE; ++e) taken[e] = false;
        for (int j = 0; j < k; ++j) {
            const int ej = sel[(size_t)s * n_dyn + j];
            float mx = -INFINITY;
#pragma unroll
            for (int e = 0; e < UMOE_MAXE; ++e)
                if (e < n_dyn && !taken[e]) mx = fmaxf(mx, z[e]);
            float pj[UMOE_MAXE];
            float den = 0.f;
#pragma unroll
            for (int e = 0; e < UMOE_MAXE; ++e) {
                pj[e] = 0.f;
                if (e < n_dyn && !taken[e]) {
                    const float f = fmaxf(fabsf(z[e]), fabsf(mx));
                    if (!((mx - z[e]) / f > 2.f * jitter_eps)) pj[e] = expf(z[e] - mx);
                }
                den += pj[e];
            }
            float dwj = 0.f, psel = 0.f;
#pragma unroll
            for (int e = 0; e < UMOE_MAXE; ++e) {
                pj[e] /= den;
                if (e == ej) {
                    dwj = drw[e];
                    psel = pj[e];
                }
            }
#pragma unroll
            for (int e = 0; e < UMOE_MAXE; ++e) {
                dz[e] += dwj * psel * ((e == ej ? 1.f : 0.f) - pj[e]);
                if (e == ej) taken[e] = true;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e)
        if (e < E) d_logits[(size_t)s * E + e] = dz[e] + (d_logits_in ? d_logits_in[(size_t)s * E + e] : 0.f);
}

extern "C" int umoe_router_bwd(const void* logits, int logits_bf16, const int32_t* sel, const int64_t* top_k,
                               const int32_t* expert_mask, const float* d_moe_w, const float* d_gw_shared,
                               const float* d_logits_in, int S, int n_dyn, int n_real, int n_fix, double jitter_eps,
                               float* d_logits, umoe_stream_t stream) {
    UMOE_REQUIRE(logits && sel && top_k && expert_mask && d_moe_w && d_logits, "umoe_router_bwd: null argument");
    UMOE_REQUIRE(n_dyn >= 1 && n_dyn + n_fix <= UMOE_MAXE && n_real <= n_dyn && (n_fix == 0 || d_gw_shared),
                 "umoe_router_bwd: bad sizes n_dyn=%d n_real=%d n_fix=%d", n_dyn, n_real, n_fix);
    if (S == 0) return 0;
    router_bwd_kernel<<<dim3((unsigned)ceil_div(S, 256)), 256, 0, (hipStream_t)stream>>>(
        logits, logits_bf16, sel, top_k, expert_mask, d_moe_w, d_gw_shared, d_logits_in, S, n_dyn, n_real, n_fix, (float)jitter_eps,
        d_logits, 0, nullptr);
    UMOE_LAUNCH_CHECK();
    return 0;
}

extern "C" int umoe_router_bwd_drop(const void* logits, int logits_bf16, const int32_t* sel, const int64_t* top_k, const int32_t* expert_mask,
                                    const float* d_moe_w, const float* d_gw_shared, const float* d_logits_in, int S, int n_dyn, int n_real,
                                    int n_fix, double jitter_eps, float* d_logits, umoe_stream_t stream) {
    UMOE_REQUIRE(logits && sel && top_k && expert_mask && d_moe_w && d_logits, "umoe_router_bwd_drop: null argument");
    UMOE_REQUIRE(n_dyn >= 1 && n_dyn + n_fix <= UMOE_MAXE && n_real <= n_dyn && (n_fix == 0 || d_gw_shared),
                 "umoe_router_bwd_drop: bad sizes n_dyn=%d n_real=%d n_fix=%d", n_dyn, n_real, n_fix);
    if (S == 0) return 0;
    router_bwd_kernel<<<dim3((unsigned)ceil_div(S, 256)), 256, 0, (hipStream_t)stream>>>(
        logits, logits_bf16, sel, top_k, expert_mask, d_moe_w, d_gw_shared, d_logits_in, S, n_dyn, n_real, n_fix, (float)jitter_eps,
        d_logits, 1, nullptr);
    UMOE_LAUNCH_CHECK();
    return 0;
}

extern "C" int umoe_router_bwd_ex(const void* logits, int logits_bf16, const int32_t* sel, const int64_t* top_k, const int32_t* expert_mask,
                                  const float* d_moe_w, const float* d_gw_shared, const float* d_logits_in, int S, int n_dyn, int n_real,
                                  int n_fix, double jitter_eps, int token_drop, const float* round_factor, float* d_logits,
                                  umoe_stream_t stream) {
    UMOE_REQUIRE(logits && sel && top_k && expert_mask && d_moe_w && d_logits, "umoe_router_bwd_ex: null argument");
    UMOE_REQUIRE(n_dyn >= 1 && n_dyn + n_fix <= UMOE_MAXE && n_real <= n_dyn && (n_fix == 0 || d_gw_shared),
                 "umoe_router_bwd_ex: bad sizes n_dyn=%d n_real=%d n_fix=%d", n_dyn, n_real, n_fix);
    if (S == 0) return 0;
    router_bwd_kernel<<<dim3((unsigned)ceil_div(S, 256)), 256, 0, (hipStream_t)stream>>>(
        logits, logits_bf16, sel, top_k, expert_mask, d_moe_w, d_gw_shared, d_logits_in, S, n_dyn, n_real, n_fix, (float)jitter_eps,
        d_logits, token_drop ? 1 : 0, round_factor);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ RMSNorm (+ residual) backward
// forward: h = r + x (optional residual), y = w * bf16(h * rs), rs = rsqrt(mean(h^2) + eps)   (Qwen2RMSNorm, model.py:206-207)
// backward: with xh = h * rs, gy = dy * w:  dh = rs * (gy - xh * mean(gy * xh)) (+ dsum, the gradient arriving on h itself)
//           dw[c] = sum_s dy[s][c] * bf16(xh[s][c])   (two stages: per-workgroup partials, then a fixed-order reduction)
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const uint16_t* __restrict__ h, const uint16_t* __restrict__ w,
                                                          const uint16_t* __restrict__ dy, const uint16_t* __restrict__ dsum, float eps,
                                                          int S, int D, int rows_per_wg, uint16_t* __restrict__ dh,
                                                          float* __restrict__ dw_part) {
    __shared__ float sh[4];
    const int nch = D >> 3;
    const int s0 = blockIdx.x * rows_per_wg;
    // per-thread column chunks stay fixed over the rows of this workgroup: dw partials accumulate in registers
    float dwacc[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) dwacc[q][j] = 0.f;
    for (int s = s0; s < min(s0 + rows_per_wg, S); ++s) {
        float ss = 0.f, dot = 0.f;
        float hv[4][8], gv[4][8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = threadIdx.x + 256 * q;
#pragma unroll
            for (int j = 0; j < 8; ++j) hv[q][j] = gv[q][j] = 0.f;
            if (c < nch) {
                float wv[8], d[8];
                unpack8(ld16(h + (size_t)s * D + c * 8), hv[q]);
                unpack8(ld16(w + c * 8), wv);
                unpack8(ld16(dy + (size_t)s * D + c * 8), d);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    ss += hv[q][j] * hv[q][j];
                    gv[q][j] = d[j] * wv[j];
                }
            }
        }
        ss = block_sum_256(ss, sh);
        __syncthreads();
        const float rs = rsqrtf(ss / (float)D + eps);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) dot += gv[q][j] * hv[q][j] * rs;
        dot = block_sum_256(dot, sh);
        __syncthreads();
        const float mean = dot / (float)D;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = threadIdx.x + 256 * q;
            if (c < nch) {
                float o[8], d[8], ds[8];
                unpack8(ld16(dy + (size_t)s * D + c * 8), d);
#pragma unroll
                for (int j = 0; j < 8; ++j) ds[j] = 0.f;
                if (dsum) unpack8(ld16(dsum + (size_t)s * D + c * 8), ds);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = hv[q][j] * rs;
                    o[j] = rs * (gv[q][j] - xh * mean) + ds[j];
                    dwacc[q][j] += d[j] * rbf(xh);
                }
                st16(dh + (size_t)s * D + c * 8, pack8(o));
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = threadIdx.x + 256 * q;
        if (c < nch) {
#pragma unroll
            for (int j = 0; j < 8; ++j) dw_part[(size_t)blockIdx.x * D + c * 8 + j] = dwacc[q][j];
        }
    }
}

// column sums of the per-workgroup partial rows: 64 columns per workgroup, wave w adds the parts p = w (mod 4) on four
// independent chains (16 loads in flight per column instead of one dependent load per part: 111 -> ~10 us at 512 parts),
// fixed association: ((chain sums of wave 0) + wave 1) + wave 2) + wave 3
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ part, int n_part, int D, uint16_t* __restrict__ out) {
    __shared__ float sh[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < D) {
        int p = wave;
        for (; p + 12 < n_part; p += 16) {
            a0 += part[(size_t)p * D + c];
            a1 += part[(size_t)(p + 4) * D + c];
            a2 += part[(size_t)(p + 8) * D + c];
            a3 += part[(size_t)(p + 12) * D + c];
        }
        for (; p < n_part; p += 4) a0 += part[(size_t)p * D + c];
    }
    sh[wave][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (wave == 0 && c < D) out[c] = f2bf(((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane]);
}

extern "C" int umoe_rmsnorm_residual_bwd(const uint16_t* h, const uint16_t* w, const uint16_t* dy, const uint16_t* dsum, float eps,
                                         int S, int D, uint16_t* dh, uint16_t* dw, float* ws, size_t ws_floats,
                                         umoe_stream_t stream) {
    UMOE_REQUIRE(h && w && dy && dh && dw && ws && D % 8 == 0 && D <= 8192, "umoe_rmsnorm_residual_bwd: bad argument (D %% 8, D <= 8192)");
    if (S == 0) return 0;
    const int n_wg = S < 512 ? S : 512;
    const int rpw = ceil_div(S, n_wg);
    const int used = ceil_div(S, rpw);
    UMOE_REQUIRE(ws_floats >= (size_t)used * D, "umoe_rmsnorm_residual_bwd: workspace too small (%zu < %zu floats)", ws_floats, (size_t)used * D);
    rmsnorm_bwd_kernel<<<dim3((unsigned)used), 256, 0, (hipStream_t)stream>>>(h, w, dy, dsum, eps, S, D, rpw, dh, ws);
    UMOE_LAUNCH_CHECK();
    colsum_kernel<<<dim3((unsigned)ceil_div(D, 64)), 256, 0, (hipStream_t)stream>>>(ws, used, D, dw);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ aux loss backward
// aux = n_dyn * sum_e f_e * P_e,  f_e = sum_s w_s mask[s][e] / W (no gradient),  P_e = sum_s w_s p[s][e] / W,
// p[s] = softmax over the n_dyn columns of masked_fill(logits, mask == 0, finfo.min)  (core.py:361-389)
// d aux / d z[s][e] = n_dyn * (w_s / W) * p[s][e] * (f_e - sum_a f_a p[s][a]) on the columns that kept their logit.
__global__ __launch_bounds__(256) void aux_frac_kernel(const int32_t* __restrict__ mask, const float* __restrict__ tok_w, int S, int E,
                                                       int n_dyn, float* __restrict__ ws) {
    __shared__ float sh[4];
    float fm[UMOE_MAXE];
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e) fm[e] = 0.f;
    float wsum = 0.f;
    for (int s = threadIdx.x; s < S; s += 256) {
        const float w = tok_w ? tok_w[s] : 1.f;
        wsum += w;
#pragma unroll
        for (int e = 0; e < UMOE_MAXE; ++e)
            if (e < n_dyn) fm[e] += w * (float)mask[(size_t)s * E + e];
    }
    wsum = block_sum_256(wsum, sh);
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e)
        if (e < n_dyn) {
            const float a = block_sum_256(fm[e], sh);
            if (threadIdx.x == 0) ws[e] = a / wsum;
        }
    if (threadIdx.x == 0) ws[UMOE_MAXE] = wsum;
}

__global__ __launch_bounds__(256) void aux_bwd_kernel(const void* __restrict__ logits, int logits_bf16, const int32_t* __restrict__ mask,
                                                      const float* __restrict__ tok_w, const float* __restrict__ ws,
                                                      const float* __restrict__ d_aux, int S, int E, int n_dyn,
                                                      float* __restrict__ d_logits) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    float x[UMOE_MAXE], p[UMOE_MAXE];
    bool keep[UMOE_MAXE];
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e) {
        keep[e] = false;
        x[e] = -INFINITY;
        if (e < n_dyn) {
            const float l = logits_bf16 ? bf2f(reinterpret_cast<const uint16_t*>(logits)[(size_t)s * E + e])
                                        : reinterpret_cast<const float*>(logits)[(size_t)s * E + e];
            keep[e] = mask[(size_t)s * E + e] != 0;
            x[e] = keep[e] ? l : (logits_bf16 ? -3.3895313892515355e38f : -3.4028234663852886e38f);
            mx = fmaxf(mx, x[e]);
        }
    }
    float sm = 0.f;
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e) {
        p[e] = e < n_dyn ? expf(x[e] - mx) : 0.f;
        sm += p[e];
    }
    float fdot = 0.f;
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e) {
        p[e] /= sm;
        if (e < n_dyn) fdot += ws[e] * p[e];
    }
    const float scale = (*d_aux) * (float)n_dyn * (tok_w ? tok_w[s] : 1.f) / ws[UMOE_MAXE];
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e)
        if (e < E) d_logits[(size_t)s * E + e] = (e < n_dyn && keep[e]) ? scale * p[e] * (ws[e] - fdot) : 0.f;
}

extern "C" int umoe_aux_loss_bwd(const void* logits, int logits_bf16, const int32_t* expert_mask, const float* token_weight, int S,
                                 int E, int n_dyn, const float* d_aux, float* d_logits, float* ws, umoe_stream_t stream) {
    UMOE_REQUIRE(logits && expert_mask && d_aux && d_logits && ws && n_dyn >= 1 && n_dyn <= E && E <= UMOE_MAXE,
                 "umoe_aux_loss_bwd: bad argument");
    if (S == 0) return 0;
    aux_frac_kernel<<<1, 256, 0, (hipStream_t)stream>>>(expert_mask, token_weight, S, E, n_dyn, ws);
    UMOE_LAUNCH_CHECK();
    aux_bwd_kernel<<<dim3((unsigned)ceil_div(S, 256)), 256, 0, (hipStream_t)stream>>>(logits, logits_bf16, expert_mask, token_weight, ws,
                                                                                      d_aux, S, E, n_dyn, d_logits);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ attention backward pieces
// Training-path attention backward recomputes P = softmax(scale * Q K^T + causal / left-padding mask) from materialised
// score tiles (umoe_tiled_gemm, fp32 raw epilogue) -- "unfused" on purpose for the first version: every contraction runs
// on the verified tiled GEMM, these two kernels do the row-wise part (eager_attention_forward semantics of the
// reference's transformers dependency: fp32 softmax, probabilities cast to bf16).
// scores [rows][ld] fp32, row = (head-in-group, query t) at h * Tp + t; keys [kv_start, t] are visible.
__global__ __launch_bounds__(256) void attn_softmax_kernel(const float* __restrict__ sc, int ld, int T, int Tp, int kv_start, float scale,
                                                           uint16_t* __restrict__ p_out, int ld_p) {
    __shared__ float sh[4];
    const int row = blockIdx.x;                 // h * Tp + t
    const int t = row % Tp;
    if (t >= T) return;
    const float* s = sc + (size_t)row * ld;
    uint16_t* po = p_out + (size_t)row * ld_p;
    float mx = -INFINITY;
    for (int j = kv_start + threadIdx.x; j <= t; j += 256) mx = fmaxf(mx, s[j] * scale);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    float sum = 0.f;
    for (int j = kv_start + threadIdx.x; j <= t; j += 256) sum += expf(s[j] * scale - mx);
    sum = block_sum_256(sum, sh);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;
    for (int j = threadIdx.x; j < ld_p; j += 256) {
        float v = 0.f;
        if (j >= kv_start && j <= t) v = expf(s[j] * scale - mx) * inv;
        po[j] = f2bf(v);
    }
}

// dS = scale * P o (dP - sum_j dP_j P_j)   (softmax backward in fp32, rounded to bf16; masked keys stay 0)
__global__ __launch_bounds__(256) void attn_softmax_bwd_kernel(const uint16_t* __restrict__ p, const uint16_t* __restrict__ dp, int ld,
                                                               int T, int Tp, float scale, uint16_t* __restrict__ ds) {
    __shared__ float sh[4];
    const int row = blockIdx.x;
    const int t = row % Tp;
    if (t >= T) return;
    const uint16_t* pr = p + (size_t)row * ld;
    const uint16_t* dr = dp + (size_t)row * ld;
    float dot = 0.f;
    for (int j = threadIdx.x; j <= t; j += 256) dot += bf2f(pr[j]) * bf2f(dr[j]);
    dot = block_sum_256(dot, sh);
    uint16_t* o = ds + (size_t)row * ld;
    for (int j = threadIdx.x; j < ld; j += 256) {
        float v = 0.f;
        if (j <= t) v = scale * bf2f(pr[j]) * (bf2f(dr[j]) - dot);
        o[j] = f2bf(v);
    }
}

extern "C" int umoe_attn_softmax_fwd(const float* scores, int ld, int heads, int T, int Tp, int kv_start, float scale, uint16_t* p_out,
                                     int ld_p, umoe_stream_t stream) {
    UMOE_REQUIRE(scores && p_out && heads > 0 && T > 0 && Tp >= T && ld >= T && ld_p >= T, "umoe_attn_softmax_fwd: bad argument");
    attn_softmax_kernel<<<dim3((unsigned)(heads * Tp)), 256, 0, (hipStream_t)stream>>>(scores, ld, T, Tp, kv_start, scale, p_out, ld_p);
    UMOE_LAUNCH_CHECK();
    return 0;
}

extern "C" int umoe_attn_softmax_bwd(const uint16_t* p, const uint16_t* dp, int ld, int heads, int T, int Tp, float scale, uint16_t* ds,
                                     umoe_stream_t stream) {
    UMOE_REQUIRE(p && dp && ds && heads > 0 && T > 0 && Tp >= T && ld >= T, "umoe_attn_softmax_bwd: bad argument");
    attn_softmax_bwd_kernel<<<dim3((unsigned)(heads * Tp)), 256, 0, (hipStream_t)stream>>>(p, dp, ld, T, Tp, scale, ds);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ mRoPE backward
// forward (umoe_qkv_mrope_kvappend): q_r = q*cos + rot(q)*sin, k likewise into the cache, v copied.  The rotation is
// orthogonal: d(x) = dy*cos - rot(dy)*sin.  Inputs: dq [n_tok][H*hd], dk / dv in cache layout [rows][KVH][Lmax][hd];
// output d_qkv [n_tok][(H + 2 KVH)*hd] (the gradient of the bias-added QKV projection).
__global__ __launch_bounds__(256) void rope_bwd_kernel(const umoe_rope_args a, const uint16_t* __restrict__ dq, const uint16_t* __restrict__ dk,
                                                       const uint16_t* __restrict__ dv, uint16_t* __restrict__ dqkv) {
    const int tok = blockIdx.x;
    const int row = tok / a.T, t = tok - row * a.T;
    const int hd = a.hd, half = hd >> 1;
    const int QKV = (a.H + 2 * a.KVH) * hd;
    const int p0 = a.pos3[tok], p1 = a.pos3[a.n_tok + tok], p2 = a.pos3[2 * a.n_tok + tok];
    const int slot = a.kv_pos[tok];
    for (int idx = threadIdx.x; idx < (a.H + 2 * a.KVH) * hd; idx += 256) {
        const int head = idx / hd, d = idx - head * hd;
        float out;
        if (head < a.H + a.KVH) {
            const uint16_t* src = head < a.H ? dq + (size_t)tok * a.H * hd + (size_t)head * hd
                                             : dk + (((size_t)row * a.KVH + (head - a.H)) * a.Lmax + slot) * hd;
            const int i = d % half;
            const int pos = (i < a.sec0) ? p0 : (i < a.sec0 + a.sec1 ? p1 : p2);
            const float c = bf2f(a.cos_tab[(size_t)pos * half + i]), sn = bf2f(a.sin_tab[(size_t)pos * half + i]);
            const float y = bf2f(src[d]);
            const float yr = d < half ? -bf2f(src[d + half]) : bf2f(src[d - half]);   // rot(dy)
            out = rbf(y * c) - rbf(yr * sn);
        } else {
            out = bf2f(dv[(((size_t)row * a.KVH + (head - a.H - a.KVH)) * a.Lmax + slot) * hd + d]);
        }
        dqkv[(size_t)tok * QKV + idx] = f2bf(out);
    }
    (void)t;
}

extern "C" int umoe_qkv_mrope_bwd(const umoe_rope_args* a, const uint16_t* dq, const uint16_t* dk_cache, const uint16_t* dv_cache,
                                  uint16_t* dqkv, umoe_stream_t stream) {
    UMOE_REQUIRE(a && a->cos_tab && a->sin_tab && a->pos3 && a->kv_pos && dq && dk_cache && dv_cache && dqkv, "umoe_qkv_mrope_bwd: null argument");
    UMOE_REQUIRE(a->hd % 2 == 0 && a->sec0 + a->sec1 + a->sec2 == a->hd / 2 && a->T >= 1 && a->n_tok % a->T == 0,
                 "umoe_qkv_mrope_bwd: bad head_dim/sections/T");
    if (a->n_tok == 0) return 0;
    rope_bwd_kernel<<<dim3((unsigned)a->n_tok), 256, 0, (hipStream_t)stream>>>(*a, dq, dk_cache, dv_cache, dqkv);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// ====================================================================================================================
// Composite backward entry points: host-side sequences of the kernels above and of umoe_tiled_gemm, with
// a caller-provided workspace -- nothing is allocated or synchronised here.
// ====================================================================================================================
namespace {
struct WsCarver {
    char* base;
    size_t off = 0, cap;
    WsCarver(void* b, size_t c) : base(reinterpret_cast<char*>(b)), cap(c) {}
    template <typename T>
    T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};
inline int r8(int n) { return (n + 7) & ~7; }
}  // namespace

// layout of the workspace of umoe_grouped_swiglu_bwd / umoe_shared_swiglu_bwd (sizes in bf16 elements)
static size_t swiglu_bwd_carve(const umoe_swiglu_bwd_args* a, void* ws, size_t cap, uint16_t** wdT, uint16_t** wguT, uint16_t** dh,
                               uint16_t** dgu, uint16_t** xe, float** tn_ws) {
    WsCarver k(ws, cap);
    const int G = a->num_groups, D = a->D, I = a->I;
    // (transposed weight copies only where the k-major weight read of umoe_tiled_gemm does not apply: I % 32 != 0)
    *wdT = I % 32 ? k.take<uint16_t>((size_t)G * I * r8(D)) : nullptr;        // per group [I][r8(D)]
    *wguT = I % 32 ? k.take<uint16_t>((size_t)G * D * 2 * I) : nullptr;       // per group [D][2I]
    *dh = k.take<uint16_t>((size_t)a->slot_rows * I);
    *dgu = k.take<uint16_t>((size_t)a->slot_rows * 2 * I);
    // routed experts: the gathered input rows in slot order (the weight-gradient product reads its operands by row window);
    // shared experts: fp32 partial outputs of the K-split weight-gradient products (up to 8 parts of the (dWg | dWu) slab)
    *xe = a->counts ? k.take<uint16_t>((size_t)a->slot_rows * D) : nullptr;
    *tn_ws = a->counts ? nullptr : k.take<float>((size_t)8 * 2 * G * I * D);
    return (k.off + 255) & ~(size_t)255;
}

extern "C" size_t umoe_swiglu_bwd_workspace_bytes(const umoe_swiglu_bwd_args* a) {
    if (!a) return 0;
    uint16_t *p0, *p1, *p2, *p3, *p4;
    float* p5;
    return swiglu_bwd_carve(a, nullptr, 0, &p0, &p1, &p2, &p3, &p4, &p5);
}

// xe[off_g + r] = x[slot_token[off_g + r]], r < counts[g]: the routed experts' input rows in slot order (rows behind a count are never read)
__global__ __launch_bounds__(256) void gather_slots_kernel(const uint16_t* __restrict__ x, const int ldx, const int D, const int32_t* __restrict__ slot_token,
                                                           const int32_t* __restrict__ counts, const int32_t* __restrict__ offsets, uint16_t* __restrict__ out) {
    const int g = blockIdx.y, cnt = counts[g], off = offsets[g];
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < cnt; r += gridDim.x * 4) {
        const uint16_t* src = x + (size_t)slot_token[off + r] * ldx;
        uint16_t* dst = out + (size_t)(off + r) * D;
        for (int c = threadIdx.x & 63; c < (D >> 3); c += 64) st16(dst + c * 8, ld16(src + c * 8));
    }
}

// The weight-gradient products are off the dependency chain of a backward pass (nothing in the pass reads dW): they run on a SIDE stream
// beside the input-gradient chain, forked and joined INSIDE the call (events), so the HBM-bound kernels of the chain (SwiGLU backward, the
// row gather) and the round tails of the GEMMs overlap with them.  UMOE_BWD_OVERLAP=0: everything on the caller's stream.
BwdSide& bwd_side() {
    static BwdSide b;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char* v = getenv("UMOE_BWD_OVERLAP");
        if (!(v && atoi(v) == 0)) {
            b.ok = hipStreamCreateWithFlags(&b.side, hipStreamNonBlocking) == hipSuccess &&
                   hipEventCreateWithFlags(&b.fork, hipEventDisableTiming) == hipSuccess &&
                   hipEventCreateWithFlags(&b.mid, hipEventDisableTiming) == hipSuccess &&
                   hipEventCreateWithFlags(&b.join, hipEventDisableTiming) == hipSuccess;
        }
    }
    return b;
}

// backward of down(silu(gate x) * up x) over groups of rows (core.py:16-49,406-416):
//   dH = dY Wd ; (dG | dU) = swiglu'(G, U, dH) ; dX_slots = dG Wg + dU Wu ; dWd = dY^T H ; dWg = dG^T X ; dWu = dU^T X
// ragged groups (routed experts): counts/offsets (8-aligned, umoe_dispatch_build_aligned) + optional gather list for x;
// static groups (shared experts): counts == NULL, group g owns rows [row_base + g*max_rows, +max_rows).
static int swiglu_bwd_impl(const umoe_swiglu_bwd_args* a, umoe_stream_t stream) {
    UMOE_REQUIRE(a && a->num_groups > 0 && a->num_groups <= 12 && a->w_gate && a->w_up && a->w_down && a->x && a->h && a->gu && a->dy &&
                     a->dx_slots && a->dw_gate && a->dw_up && a->dw_down && a->ws,
                 "umoe_*_swiglu_bwd: null argument or more than 12 groups");
    UMOE_REQUIRE(a->D % 8 == 0 && a->I % 8 == 0 && a->max_rows > 0 && a->slot_rows > 0, "umoe_*_swiglu_bwd: D and I must be multiples of 8");
    UMOE_REQUIRE((a->counts == nullptr) == (a->offsets == nullptr), "umoe_*_swiglu_bwd: counts and offsets come together");
    const bool ragged = a->counts != nullptr;
    uint16_t *wdT, *wguT, *dh, *dgu, *xe;
    float* tn_ws;
    const size_t need = swiglu_bwd_carve(a, a->ws, a->ws_bytes, &wdT, &wguT, &dh, &dgu, &xe, &tn_ws);
    UMOE_REQUIRE(a->ws_bytes >= need, "umoe_*_swiglu_bwd: workspace too small (%zu < %zu bytes)", a->ws_bytes, need);
    const int G = a->num_groups, D = a->D, I = a->I, S = a->max_rows;
    int rc;
    // transposed weight copies: Wd^T [I][r8(D)], (Wg^T | Wu^T) [D][2I] -- the caller's, when it kept them (unchanged weights)
    UMOE_REQUIRE((a->w_down_T == nullptr) == (a->w_gateup_T == nullptr), "umoe_*_swiglu_bwd: w_down_T and w_gateup_T come together");
    // input gradients on the weights AS STORED (umoe_tgroup_t.w_kmajor: the contraction index is the weight's row) -- no transposed copies;
    // (dG | dU) runs against Wg then Wu in one contraction, which needs I % 32 == 0 (K tiles of 32 rows); otherwise the transposed copies
    const bool kmaj = I % 32 == 0;
    if (kmaj) {
    } else if (a->w_down_T) {
        wdT = const_cast<uint16_t*>(a->w_down_T);
        wguT = const_cast<uint16_t*>(a->w_gateup_T);
    } else {
        uint16_t* d1[12];
        uint16_t* d2[12];
        uint16_t* d3[12];
        for (int g = 0; g < G; ++g) {
            d1[g] = wdT + (size_t)g * I * r8(D);
            d2[g] = wguT + (size_t)g * D * 2 * I;
            d3[g] = d2[g] + I;
        }
        if ((rc = transpose_multi(a->w_down, d1, G, I, I, D, r8(D), stream))) return rc;      // Wd [D][I] -> [I][r8(D)]
        if ((rc = transpose_multi(a->w_gate, d2, G, D, D, I, 2 * I, stream))) return rc;      // Wg [I][D] -> [D][2I] left half
        if ((rc = transpose_multi(a->w_up, d3, G, D, D, I, 2 * I, stream))) return rc;        // Wu        -> right half
    }
    umoe_tgroup_t tg[12];
    auto rows_of = [&](umoe_tgroup_t& t, int g) {
        if (ragged) { t.row_off = a->offsets + g; t.count = a->counts + g; }
        else { t.static_count = S; t.a_row_base = a->row_base + g * S; t.out_row_base = a->row_base + g * S; }
    };
    // ---- weight gradients straight from the row-major slot buffers (umoe_tiled_gemm_tn; round 2 wrote dY^T, H^T, (dG|dU)^T and X^T first):
    //   dWd_g [D][I] = dY_g^T H_g ; (dWg_g ; dWu_g) [I][D] = (dG_g | dU_g)^T X_g
    // routed experts: ONE grouped launch per product kind with the experts' slot windows read on the device (gate and up together:
    // 2 G groups); shared experts: static windows, the K split chosen by the library when the outputs of a launch are one stacked slab
    // (ops.experts_swiglu_bwd allocates them so).
    auto stacked = [&](uint16_t* const* ptrs, size_t elems) {
        for (int g = 1; g < G; ++g)
            if (ptrs[g] != ptrs[0] + (size_t)g * elems) return false;
        return true;
    };
    umoe_tn_group_t tn[24];
    auto wgrad_down = [&](umoe_stream_t st) -> int {
        umoe_tgemm_tn_args b{};
        memset(tn, 0, sizeof(tn));
        if (ragged) {
            for (int g = 0; g < G; ++g) {
                tn[g].m = D; tn[g].n = I; tn[g].k_off_dev = a->offsets + g; tn[g].k_count_dev = a->counts + g; tn[g].out = a->dw_down[g];
            }
        } else {       // shared experts: group g owns the slot rows [row_base + g S, + S) of dY / H
            const bool slab = stacked(a->dw_down, (size_t)D * I);
            for (int g = 0; g < G; ++g) {
                const size_t r0 = (size_t)a->row_base + (size_t)g * S;
                tn[g].m = D; tn[g].n = I; tn[g].k = S;
                tn[g].p = a->dy + r0 * a->lddy; tn[g].ldp = a->lddy; tn[g].q = a->h + r0 * a->ldh; tn[g].ldq = a->ldh;
                if (slab) tn[g].out_row_base = g * D; else tn[g].out = a->dw_down[g];
            }
            if (slab) { b.k_split = -1; b.ws = tn_ws; b.part_stride = (long)G * D * I; }
        }
        b.groups = tn; b.num_groups = G; b.p = a->dy; b.ldp = a->lddy; b.q = a->h; b.ldq = a->ldh; b.out = a->dw_down[0]; b.ldo = I;
        return umoe_tiled_gemm_tn(&b, st);
    };
    auto wgrad_gateup = [&](umoe_stream_t st) -> int {
        umoe_tgemm_tn_args b{};
        memset(tn, 0, sizeof(tn));
        if (ragged) {
            gather_slots_kernel<<<dim3((unsigned)(S < 2048 ? ceil_div(S, 4) : 512), (unsigned)G), 256, 0, (hipStream_t)st>>>(a->x, a->ldx, D, a->slot_token, a->counts, a->offsets, xe);
            UMOE_LAUNCH_CHECK();
            for (int g = 0; g < G; ++g) {
                tn[g].m = I; tn[g].n = D; tn[g].k_off_dev = a->offsets + g; tn[g].k_count_dev = a->counts + g; tn[g].out = a->dw_gate[g];
                tn[G + g] = tn[g];
                tn[G + g].p_col_off = I; tn[G + g].out = a->dw_up[g];
            }
            b.q = xe; b.ldq = D;
        } else {       // ... and of (dG | dU); x is read by identity (rows 0 .. S-1)
            const bool slab = stacked(a->dw_gate, (size_t)I * D) && stacked(a->dw_up, (size_t)I * D) && a->dw_up[0] == a->dw_gate[0] + (size_t)G * I * D;
            for (int g = 0; g < G; ++g) {
                const size_t r0 = (size_t)a->row_base + (size_t)g * S;
                tn[g].m = I; tn[g].n = D; tn[g].k = S;
                tn[g].p = dgu + r0 * 2 * I; tn[g].ldp = 2 * I;
                if (slab) tn[g].out_row_base = g * I; else tn[g].out = a->dw_gate[g];
                tn[G + g] = tn[g];
                tn[G + g].p_col_off = I;
                if (slab) tn[G + g].out_row_base = (G + g) * I; else tn[G + g].out = a->dw_up[g];
            }
            if (slab) { b.k_split = -1; b.ws = tn_ws; b.part_stride = (long)2 * G * I * D; }
            b.q = a->x; b.ldq = a->ldx;
        }
        b.groups = tn; b.num_groups = 2 * G; b.p = dgu; b.ldp = 2 * I; b.out = a->dw_gate[0]; b.ldo = D;
        return umoe_tiled_gemm_tn(&b, st);
    };
    // fork: dWd needs only the call's inputs
    BwdSide& bs = bwd_side();
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    bool overlap = false;
    if (bs.ok) {
        if (hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess) overlap = cap == hipStreamCaptureStatusNone;
        else (void)hipGetLastError();       // (a stream that cannot be queried: no overlap, and no stale error for the launch checks below)
    }
    if (overlap) {
        UMOE_HIP(hipEventRecord(bs.fork, (hipStream_t)stream));
        UMOE_HIP(hipStreamWaitEvent(bs.side, bs.fork, 0));
        if ((rc = wgrad_down(bs.side))) return rc;
    }
    // dH = dY * Wd
    memset(tg, 0, sizeof(tg));
    for (int g = 0; g < G; ++g) {
        if (kmaj) { tg[g].w = a->w_down[g]; tg[g].w_kmajor = 1; tg[g].n = I; tg[g].k = D; tg[g].ldw = I; }       // Wd [D][I]: row = contraction index
        else { tg[g].w = wdT + (size_t)g * I * r8(D); tg[g].n = I; tg[g].k = D; tg[g].ldw = r8(D); }
        rows_of(tg[g], g);
    }
    umoe_tgemm_args ta{};
    ta.groups = tg; ta.num_groups = G; ta.max_rows = S; ta.a = a->dy; ta.lda = a->lddy; ta.out = dh; ta.ldo = I; ta.epilogue = UMOE_EPI_BF16;
    if ((rc = umoe_tiled_gemm(&ta, stream))) return rc;
    // SwiGLU backward over the slot rows
    if (ragged) {
        if ((rc = umoe_swiglu_bwd(dh, I, a->gu, a->ldgu, I, a->offsets + G, a->slot_rows, dgu, 2 * I, stream))) return rc;
    } else {
        const size_t r0 = (size_t)a->row_base;
        if ((rc = umoe_swiglu_bwd(dh + r0 * I, I, a->gu + r0 * a->ldgu, a->ldgu, I, nullptr, G * S, dgu + r0 * 2 * I, 2 * I, stream))) return rc;
    }
    if (overlap) {      // (dG | dU) exists: the side stream takes dWg / dWu while this one computes dX
        UMOE_HIP(hipEventRecord(bs.mid, (hipStream_t)stream));
        UMOE_HIP(hipStreamWaitEvent(bs.side, bs.mid, 0));
        if ((rc = wgrad_gateup(bs.side))) return rc;
        UMOE_HIP(hipEventRecord(bs.join, bs.side));
    }
    // dX_slots = (dG | dU) * (Wg^T | Wu^T)^T
    memset(tg, 0, sizeof(tg));
    for (int g = 0; g < G; ++g) {
        if (kmaj) { tg[g].w = a->w_gate[g]; tg[g].w2 = a->w_up[g]; tg[g].w_kmajor = 1; tg[g].k_w1 = I; tg[g].n = D; tg[g].k = 2 * I; tg[g].ldw = D; }
        else { tg[g].w = wguT + (size_t)g * D * 2 * I; tg[g].n = D; tg[g].k = 2 * I; tg[g].ldw = 2 * I; }
        rows_of(tg[g], g);
    }
    ta = umoe_tgemm_args{};
    ta.groups = tg; ta.num_groups = G; ta.max_rows = S; ta.a = dgu; ta.lda = 2 * I; ta.out = a->dx_slots; ta.ldo = a->lddx; ta.epilogue = UMOE_EPI_BF16;
    if ((rc = umoe_tiled_gemm(&ta, stream))) return rc;
    if (overlap) {
        UMOE_HIP(hipStreamWaitEvent((hipStream_t)stream, bs.join, 0));      // join: the caller's stream continues behind the weight gradients
        return 0;
    }
    if ((rc = wgrad_down(stream))) return rc;
    return wgrad_gateup(stream);
}

extern "C" int umoe_grouped_swiglu_bwd(const umoe_swiglu_bwd_args* a, umoe_stream_t stream) {
    UMOE_REQUIRE(a && a->counts && a->offsets, "umoe_grouped_swiglu_bwd: routed experts need counts/offsets (umoe_dispatch_build_aligned)");
    return swiglu_bwd_impl(a, stream);
}
extern "C" int umoe_shared_swiglu_bwd(const umoe_swiglu_bwd_args* a, umoe_stream_t stream) {
    UMOE_REQUIRE(a && !a->counts && !a->offsets, "umoe_shared_swiglu_bwd: shared experts take every row (counts/offsets must be NULL)");
    return swiglu_bwd_impl(a, stream);
}

// ------------------------------------------------------------------------------------ attention backward (composite)
static size_t attn_bwd_carve(const umoe_attn_bwd_args* a, void* ws, size_t cap, float** sc, uint16_t** P, uint16_t** dP, uint16_t** dS,
                             uint16_t** qsT, uint16_t** dosT, uint16_t** kT, uint16_t** dST, uint16_t** PT) {
    WsCarver k(ws, cap);
    const int G = a->H / a->KVH, Tp = r8(a->T);
    *sc = k.take<float>((size_t)G * Tp * Tp);
    *P = k.take<uint16_t>((size_t)G * Tp * Tp);
    *dP = k.take<uint16_t>((size_t)G * Tp * Tp);
    *dS = k.take<uint16_t>((size_t)G * Tp * Tp);
    *qsT = k.take<uint16_t>((size_t)a->hd * G * Tp);
    *dosT = k.take<uint16_t>((size_t)a->hd * G * Tp);
    *kT = k.take<uint16_t>((size_t)a->hd * Tp);
    *dST = k.take<uint16_t>((size_t)Tp * G * Tp);
    *PT = k.take<uint16_t>((size_t)Tp * G * Tp);
    return (k.off + 255) & ~(size_t)255;
}

extern "C" size_t umoe_attn_prefill_bwd_workspace_bytes(const umoe_attn_bwd_args* a) {
    if (!a || a->KVH <= 0) return 0;
    float* f;
    uint16_t *p1, *p2, *p3, *p4, *p5, *p6, *p7, *p8;
    const size_t unfused = attn_bwd_carve(a, nullptr, 0, &f, &p1, &p2, &p3, &p4, &p5, &p6, &p7, &p8);
    // fused path: D [rows*T*H] fp32, kv_start [rows], then up to 4 head-split slabs of dK and dV in fp32 (umoe_attn_bwd.hip)
    const size_t nD = (size_t)a->rows * a->T * a->H;
    const size_t fused = ((nD * 4 + 255) & ~(size_t)255) + (((size_t)a->rows * 4 + 255) & ~(size_t)255) +
                         2 * 4 * (size_t)a->rows * a->KVH * a->T * 128 * 4 + 256;
    return unfused > fused ? unfused : fused;
}

// Backward of causal GQA attention over full sequences (one query per key position, left padding via kv_start):
// per (row, kv head) group the scores of its G query heads are materialised (rows h*Tp + t), P is recomputed in fp32,
// dP = dO V^T, dS = scale * P o (dP - rowsum(dP o P)), dQ = dS K, dK = sum_h dS_h^T Q_h, dV = sum_h P_h^T dO_h -- every
// contraction on umoe_tiled_gemm ("unfused" first version).
int umoe_attn_bwd_fused(const umoe_attn_bwd_args* a, umoe_stream_t stream);   // umoe_attn_bwd.hip

extern "C" int umoe_attn_prefill_bwd(const umoe_attn_bwd_args* a, umoe_stream_t stream) {
    UMOE_REQUIRE(a && a->q && a->k_cache && a->v_cache && a->kv_start_host && a->d_out && a->dq && a->dk_cache && a->dv_cache && a->ws,
                 "umoe_attn_prefill_bwd: null argument");
    {
        const int rcf = umoe_attn_bwd_fused(a, stream);   // flash-style path when the forward's output and log-sum-exp are given
        if (rcf <= 0) return rcf;
    }
    UMOE_REQUIRE(a->KVH > 0 && a->H % a->KVH == 0 && a->H / a->KVH <= 12 && a->hd % 8 == 0 && a->T > 0 && a->T <= a->Lmax && a->rows > 0,
                 "umoe_attn_prefill_bwd: bad sizes (H=%d KVH=%d hd=%d T=%d Lmax=%d)", a->H, a->KVH, a->hd, a->T, a->Lmax);
    float* sc;
    uint16_t *P, *dP, *dS, *qsT, *dosT, *kT, *dST, *PT;
    const size_t need = attn_bwd_carve(a, a->ws, a->ws_bytes, &sc, &P, &dP, &dS, &qsT, &dosT, &kT, &dST, &PT);
    UMOE_REQUIRE(a->ws_bytes >= need, "umoe_attn_prefill_bwd: workspace too small (%zu < %zu bytes)", a->ws_bytes, need);
    const int G = a->H / a->KVH, T = a->T, Tp = r8(T), hd = a->hd, HD = a->H * hd;
    hipStream_t s = (hipStream_t)stream;
    // padding rows / columns of the stacked buffers must be finite zeros (they take part in the stacked contractions)
    UMOE_HIP(hipMemsetAsync(P, 0, (size_t)G * Tp * Tp * 2, s));
    UMOE_HIP(hipMemsetAsync(dP, 0, (size_t)G * Tp * Tp * 2, s));
    UMOE_HIP(hipMemsetAsync(dS, 0, (size_t)G * Tp * Tp * 2, s));
    int rc;
    umoe_tgroup_t tg[12];
    for (int b = 0; b < a->rows; ++b)
        for (int g = 0; g < a->KVH; ++g) {
            const uint16_t* K_ = a->k_cache + (((size_t)b * a->KVH + g) * a->Lmax) * hd;
            const uint16_t* V_ = a->v_cache + (((size_t)b * a->KVH + g) * a->Lmax) * hd;
            uint16_t* dK_ = a->dk_cache + (((size_t)b * a->KVH + g) * a->Lmax) * hd;
            uint16_t* dV_ = a->dv_cache + (((size_t)b * a->KVH + g) * a->Lmax) * hd;
            // S_h = Q_h K^T (raw fp32)
            memset(tg, 0, sizeof(tg));
            for (int j = 0; j < G; ++j) {
                tg[j].w = K_; tg[j].n = T; tg[j].k = hd; tg[j].ldw = hd; tg[j].static_count = T; tg[j].a_row_base = b * T;
                tg[j].a_col_off = (g * G + j) * hd; tg[j].out_row_base = j * Tp;
            }
            umoe_tgemm_args ta{};
            ta.groups = tg; ta.num_groups = G; ta.max_rows = T; ta.a = a->q; ta.lda = HD; ta.out = sc; ta.ldo = Tp; ta.epilogue = UMOE_EPI_F32_RAW;
            if ((rc = umoe_tiled_gemm(&ta, stream))) return rc;
            if ((rc = umoe_attn_softmax_fwd(sc, Tp, G, T, Tp, a->kv_start_host[b], a->scale, P, Tp, stream))) return rc;
            // dP_h = dO_h V^T
            for (int j = 0; j < G; ++j) tg[j].w = V_;
            ta.a = a->d_out; ta.out = dP; ta.epilogue = UMOE_EPI_BF16;
            if ((rc = umoe_tiled_gemm(&ta, stream))) return rc;
            if ((rc = umoe_attn_softmax_bwd(P, dP, Tp, G, T, Tp, a->scale, dS, stream))) return rc;
            // dQ_h = dS_h K
            if ((rc = umoe_transpose_slots(K_, hd, hd, nullptr, nullptr, nullptr, 1, T, kT, Tp, stream))) return rc;
            memset(tg, 0, sizeof(tg));
            for (int j = 0; j < G; ++j) {
                tg[j].w = kT; tg[j].n = hd; tg[j].k = Tp; tg[j].ldw = Tp; tg[j].static_count = T; tg[j].a_row_base = j * Tp;
                tg[j].out_row_base = b * T; tg[j].out_col_off = (g * G + j) * hd;
            }
            ta = umoe_tgemm_args{};
            ta.groups = tg; ta.num_groups = G; ta.max_rows = T; ta.a = dS; ta.lda = Tp; ta.out = a->dq; ta.ldo = HD; ta.epilogue = UMOE_EPI_BF16;
            if ((rc = umoe_tiled_gemm(&ta, stream))) return rc;
            // stacked contractions over (head, query): dK = [dS_h]^T [Q_h], dV = [P_h]^T [dO_h]
            for (int j = 0; j < G; ++j) {
                const size_t off = (size_t)b * T * HD + (size_t)(g * G + j) * hd;
                if ((rc = umoe_transpose_slots(a->q + off, HD, hd, nullptr, nullptr, nullptr, 1, T, qsT + (size_t)j * Tp, G * Tp, stream))) return rc;
                if ((rc = umoe_transpose_slots(a->d_out + off, HD, hd, nullptr, nullptr, nullptr, 1, T, dosT + (size_t)j * Tp, G * Tp, stream))) return rc;
            }
            if ((rc = umoe_transpose_slots(dS, Tp, Tp, nullptr, nullptr, nullptr, 1, G * Tp, dST, G * Tp, stream))) return rc;
            if ((rc = umoe_transpose_slots(P, Tp, Tp, nullptr, nullptr, nullptr, 1, G * Tp, PT, G * Tp, stream))) return rc;
            umoe_tgroup_t t1{};
            t1.w = qsT; t1.n = hd; t1.k = G * Tp; t1.ldw = G * Tp; t1.static_count = T;
            ta = umoe_tgemm_args{};
            ta.groups = &t1; ta.num_groups = 1; ta.max_rows = T; ta.a = dST; ta.lda = G * Tp; ta.out = dK_; ta.ldo = hd; ta.epilogue = UMOE_EPI_BF16;
            if ((rc = umoe_tiled_gemm(&ta, stream))) return rc;
            t1.w = dosT;
            ta.a = PT; ta.out = dV_;
            if ((rc = umoe_tiled_gemm(&ta, stream))) return rc;
        }
    return 0;
}
