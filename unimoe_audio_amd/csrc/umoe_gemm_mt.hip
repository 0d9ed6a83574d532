// Weight-streaming GEMM for the expert-parallel decode step: the local experts of a rank see the rows of EVERY rank
// (ep_size tiles of 16 rows), so one pass over an expert's weights must serve several row tiles.
//
//   Y[tile t][16, N] = epilogue( A[tile t][16, K] * W_g^T )   for every tile t of group g, weights read ONCE
//
// Design (MI355X / gfx950), relative to wstream_gemm (umoe_gemm.hip):
//  * the same WP16 weight stream straight into VGPRs (non-temporal 1 KiB wave-loads) and the same K split over the 8 waves
//    of a workgroup with the same fixed-order LDS reduction -- every (tile, 16-feature block) product is therefore
//    BIT-IDENTICAL to the one the ep_size = 1 launch computes (tests/test_gpu_ep.py);
//  * the activations arrive already in MFMA operand order (the pull kernel of the exchange re-lays each 16-row tile while it
//    copies it out of the uncached slab; the gate/up epilogue writes h in operand order for the down projection), so a B
//    fragment is ONE contiguous 1 KiB wave-load from L2: no LDS staging pass, no barrier in front of the stream;
//  * per k-step a wave issues NT weight loads (HBM) + MT fragment loads (L2) for NT x MT MFMAs; NT x MT = 16 keeps the
//    accumulators at 64 VGPRs and the workgroup count at 172 for every ep_size (2 / 4 / 8 tiles: 8 / 4 / 2 blocks per workgroup);
//  * the router's workgroups ride as an extra z-slice exactly as in wstream_gemm (its results feed the combine only).
// Roofline: HBM.  Algorithmic bytes per launch = sum over local experts of N*K*2.
#include "umoe_common.h"
#include "umoe_router_dev.h"
#include <string.h>

template <int NT, int MT, int U, int EPI, bool FR>
__global__ __launch_bounds__(512, 1) void wstream_mt(const umoe_mt_args p, const umoe_router_args ra) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WV = 8;
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    if constexpr (FR) {
        if (blockIdx.z == gridDim.z - 1) {     // riders: one workgroup per token, threads 0..255 (umoe_router_dev.h)
            const int token = (int)blockIdx.x;
            if (token < ra.S && threadIdx.x < 256) {
                if (ra.logits_bf16) router4_body<9, 2, 1, false>(ra, token, threadIdx.x, reinterpret_cast<float*>(smem));
                else router4_body<9, 2, 0, false>(ra, token, threadIdx.x, reinterpret_cast<float*>(smem));
            }
            return;
        }
    }
    const int g = blockIdx.z, nb0 = blockIdx.x * NT;
    if (nb0 >= p.n_blocks) return;
    const int K = p.k, KB = K >> 5;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: every guard around an MFMA must be a scalar branch
    // K split of wstream_gemm<.., U, .., 8>: whole U-step chunks per wave when they divide, single steps otherwise
    int i0, i1;
    if (KB % U == 0) {
        const int units = KB / U;
        i0 = U * ((units * wave) / WV);
        i1 = U * ((units * (wave + 1)) / WV);
    } else {
        i0 = (KB * wave) / WV;
        i1 = (KB * (wave + 1)) / WV;
    }
    i0 = __builtin_amdgcn_readfirstlane(i0);
    i1 = __builtin_amdgcn_readfirstlane(i1);
    const u32x4_t* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int nb = min(nb0 + t, p.n_blocks - 1);   // tail blocks re-read the last one; never stored
        wp[t] = reinterpret_cast<const u32x4_t*>(p.w[g]) + ((size_t)nb * KB) * 64 + lane;
    }
    const u32x4_t* bp[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int mm = min(m, p.tiles - 1);
        bp[m] = reinterpret_cast<const u32x4_t*>(p.b) + ((size_t)(g * p.b_group_tiles + mm) * KB) * 64 + lane;
    }
    f32x4_t acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    u32x4_t w0[NT][U], w1[NT][U], b0[MT][U], b1[MT][U];
    auto load_chunk = [&](u32x4_t (&dw)[NT][U], u32x4_t (&db)[MT][U], int ibase) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ii = min(ibase + u, i1 - 1);
#pragma unroll
            for (int t = 0; t < NT; ++t) dw[t][u] = __builtin_nontemporal_load(wp[t] + (size_t)ii * 64);
#pragma unroll
            for (int m = 0; m < MT; ++m) db[m][u] = bp[m][(size_t)ii * 64];
        }
    };
    auto compute_chunk = [&](const u32x4_t (&sw)[NT][U], const u32x4_t (&sb)[MT][U], int ibase) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (ibase + u < i1) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const bf16x8_t bfrag = __builtin_bit_cast(bf16x8_t, sb[m][u]);
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, sw[t][u]), bfrag, acc[m][t], 0, 0, 0);
                }
            }
        }
    };
    if (i0 < i1) load_chunk(w0, b0, i0);
    for (int i = i0; i < i1; i += 2 * U) {
        if (i + U < i1) load_chunk(w1, b1, i + U);
        compute_chunk(w0, b0, i);
        if (i + 2 * U < i1) load_chunk(w0, b0, i + 2 * U);
        if (i + U < i1) compute_chunk(w1, b1, i + U);
    }
    // ---- fixed-order cross-wave reduction (wave 0 first, as in wstream_gemm) --------------------------------------------
    f32x4_t* red = reinterpret_cast<f32x4_t*>(smem);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) red[((wave * MT + m) * NT + t) * 64 + lane] = acc[m][t];
    __syncthreads();
    auto reduced = [&](int m, int t) -> f32x4_t {
        f32x4_t s = red[(m * NT + t) * 64 + lane];
#pragma unroll
        for (int w = 1; w < WV; ++w) {
            const f32x4_t v = red[((w * MT + m) * NT + t) * 64 + lane];
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        }
        return s;
    };
    const int h = lane >> 4, mm = lane & 15;   // lane (h, mm) owns features 4h..4h+3 of token row mm of every tile
    if (EPI == UMOE_EPI_SWIGLU) {
        const int I = p.n_blocks * 8, Q = I >> 2;           // intermediate size and its K-quarter for the down projection
        for (int q = wave; q < MT * (NT / 2); q += WV) {
            const int m = q / (NT / 2), pq = q % (NT / 2);
            if (m >= p.tiles || nb0 + 2 * pq >= p.n_blocks) continue;
            const f32x4_t ga = reduced(m, 2 * pq), ua = reduced(m, 2 * pq + 1);
            uint16_t y[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gt = rbf(ga[j]);
                const float up = rbf(ua[j]);
                const float si = rbf(gt / (1.0f + expf(-gt)));
                y[j] = f2bf(si * up);
            }
            // feature f..f+3 of row mm -> operand order of the [16][I] tile: fragment (k-step i, lane = quarter*16 + row), element j
            const int f = (nb0 / 2 + pq) * 16 + 4 * h;
            const int qq = f / Q, r = f % Q;
            uint16_t* o = p.h_out + (size_t)(g * p.tiles + m) * 16 * I + ((size_t)(r >> 3) * 64 + qq * 16 + mm) * 8 + (r & 7);
            *reinterpret_cast<uint2*>(o) = make_uint2((uint32_t)y[0] | ((uint32_t)y[1] << 16), (uint32_t)y[2] | ((uint32_t)y[3] << 16));
        }
        return;
    }
    for (int q = wave; q < MT * NT; q += WV) {
        const int m = q / NT, t = q % NT;
        if (m >= p.tiles || nb0 + t >= p.n_blocks || mm >= p.n_rows) continue;
        const f32x4_t a4 = reduced(m, t);
        const int n = (nb0 + t) * 16 + 4 * h;
        uint16_t y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = f2bf(a4[j]);
        uint16_t* o = p.y_out[g][m] + (size_t)mm * p.ldo + n;
        *reinterpret_cast<uint2*>(o) = make_uint2((uint32_t)y[0] | ((uint32_t)y[1] << 16), (uint32_t)y[2] | ((uint32_t)y[3] << 16));
    }
}

template <int NT, int MT, int U, int EPI, bool FR>
static int launch_mt(const umoe_mt_args* a, hipStream_t s) {
    const size_t lds = (size_t)8 * MT * NT * 64 * 16;
    static bool configured = false;
    if (!configured) {
        UMOE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wstream_mt<NT, MT, U, EPI, FR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    umoe_router_args ra;
    memset(&ra, 0, sizeof(ra));
    dim3 grid((unsigned)ceil_div(a->n_blocks, NT), 1, (unsigned)a->num_groups);
    if (FR) {
        ra = *a->fused_router;
        UMOE_REQUIRE(ra.S <= (int)grid.x && ra.n_dyn == 9 && ra.n_fix == 2 && (ra.D == 2048 || ra.D == 4096) && ra.x && ra.gate_w && ra.expert_mask &&
                         !ra.logits_in && !ra.norm_only,
                     "umoe_gemm_mt: fused_router needs n_dyn 9 / n_fix 2, D 2048 / 4096, S <= %u", grid.x);
        grid.z += 1;
    }
    wstream_mt<NT, MT, U, EPI, FR><<<grid, 512, lds, s>>>(*a, ra);
    UMOE_LAUNCH_CHECK();
    return 0;
}

int umoe_gemm_mt(const umoe_mt_args* a, hipStream_t s) {
    UMOE_REQUIRE(a && a->b && a->num_groups >= 1 && a->num_groups <= UMOE_MT_MAXG && a->k > 0 && a->k % 32 == 0 && a->n_blocks > 0,
                 "umoe_gemm_mt: bad argument");
    UMOE_REQUIRE(a->tiles == 2 || a->tiles == 4 || a->tiles == 8, "umoe_gemm_mt: 2, 4 or 8 row tiles (got %d)", a->tiles);
    UMOE_REQUIRE(a->n_rows >= 1 && a->n_rows <= 16, "umoe_gemm_mt: 1..16 rows per tile");
    for (int g = 0; g < a->num_groups; ++g) UMOE_REQUIRE(a->w[g], "umoe_gemm_mt: group %d has no weights", g);
    if (a->epilogue == UMOE_EPI_SWIGLU) {
        // the K split of the dense decode gate/up launch: 8 waves, 1-step chunks
        UMOE_REQUIRE(a->h_out && a->n_blocks % 2 == 0 && (a->n_blocks * 8) % 32 == 0, "umoe_gemm_mt: SwiGLU needs h_out and gate/up block pairs, I %% 32 == 0");
        const bool fr = a->fused_router != nullptr;
        if (a->tiles == 2) return fr ? launch_mt<8, 2, 1, UMOE_EPI_SWIGLU, true>(a, s) : launch_mt<8, 2, 1, UMOE_EPI_SWIGLU, false>(a, s);
        if (a->tiles == 4) return fr ? launch_mt<4, 4, 1, UMOE_EPI_SWIGLU, true>(a, s) : launch_mt<4, 4, 1, UMOE_EPI_SWIGLU, false>(a, s);
        return fr ? launch_mt<2, 8, 1, UMOE_EPI_SWIGLU, true>(a, s) : launch_mt<2, 8, 1, UMOE_EPI_SWIGLU, false>(a, s);
    }
    UMOE_REQUIRE(a->epilogue == UMOE_EPI_BF16 && !a->fused_router && a->ldo % 4 == 0, "umoe_gemm_mt: epilogue must be SwiGLU or bf16 (ldo %% 4 == 0)");
    for (int g = 0; g < a->num_groups; ++g)
        for (int t = 0; t < a->tiles; ++t) UMOE_REQUIRE(a->y_out[g][t], "umoe_gemm_mt: tile (%d, %d) has no output", g, t);
    // the K split of the dense decode down launch: 8 waves, 2-step chunks
    if (a->tiles == 2) return launch_mt<4, 2, 2, UMOE_EPI_BF16, false>(a, s);
    if (a->tiles == 4) return launch_mt<2, 4, 2, UMOE_EPI_BF16, false>(a, s);
    return launch_mt<1, 8, 2, UMOE_EPI_BF16, false>(a, s);
}
