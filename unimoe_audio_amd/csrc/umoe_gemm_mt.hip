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

template <int NT, int MT, int U, int EPI, bool FR, int RW, int RB, bool WREFILL>
__global__ __launch_bounds__(512, 1) void wstream_mt(const umoe_mt_args p, const umoe_router_args ra) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WV = 8;
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    if constexpr (FR) {
        if (blockIdx.z == gridDim.z - 1) {     // riders: one workgroup per token, threads 0..255 (umoe_router_dev.h)
            const int token = (int)blockIdx.x;
            if (token < ra.S && threadIdx.x < 256) {
                TL_ENTER(5);
                if (ra.logits_bf16) router4_body<9, 2, 1, false>(ra, token, threadIdx.x, reinterpret_cast<float*>(smem) TL_PASS);
                else router4_body<9, 2, 0, false>(ra, token, threadIdx.x, reinterpret_cast<float*>(smem) TL_PASS);
                TL_EXIT(5);
            }
            return;
        }
    }
    const int g = blockIdx.z, nb0 = blockIdx.x * NT;
    const int tile0 = blockIdx.y * MT;          // a workgroup serves MT of the group's row tiles (grid.y covers the rest)
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
        const int mm = min(tile0 + m, p.tiles - 1);
        bp[m] = reinterpret_cast<const u32x4_t*>(p.b) + ((size_t)(g * p.b_group_tiles + mm) * KB) * 64 + lane;
    }
    f32x4_t acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // Register rings, one k-step per slot.  A workgroup streams so few bytes (2 blocks x K = 128 KiB of weights at ep_size 8) that
    // the launch is bound by memory LATENCY, not bandwidth: with the double-buffered chunks of wstream_gemm every second k-step
    // waited a full HBM round trip (8 k-steps per wave: ~4 round trips, 12-15 us per launch measured).  So the weight ring holds
    // RW k-steps (the wave's whole K slice where it fits: every weight load is in flight before the first MFMA) and the
    // fragment ring RB k-steps of L2-resident activations; within an iteration the fragment refill is issued BEFORE the weight
    // refill (loads return in issue order: a wait for fragments must not drag in the younger weight loads).
    u32x4_t wr[RW][NT], br[RB][MT];
    auto load_w = [&](u32x4_t (&d)[NT], int ii) {
#pragma unroll
        for (int t = 0; t < NT; ++t) d[t] = __builtin_nontemporal_load(wp[t] + (size_t)ii * 64);
    };
    auto load_b = [&](u32x4_t (&d)[MT], int ii) {
#pragma unroll
        for (int m = 0; m < MT; ++m) d[m] = bp[m][(size_t)ii * 64];
    };
    // every load is UNCONDITIONAL (index clamped to the slice; a refill past the end re-reads the last k-step, at most RW - 1 + RB - 1
    // L2-resident fragments): a branch around a load makes hipcc wait for vmcnt(0) at the join, which serialises the rings
    const int il = i1 - 1;
#pragma unroll
    for (int r = 0; r < RB; ++r) load_b(br[r], min(i0 + r, il));
#pragma unroll
    for (int r = 0; r < RW; ++r) load_w(wr[r], min(i0 + r, il));
    __builtin_amdgcn_sched_barrier(0);   // (hipcc otherwise sinks the loads next to their first use: one round trip per fragment)
    for (int base = i0; base < i1; base += RW) {
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int ii = base + r;
            if (ii < i1) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const bf16x8_t bfrag = __builtin_bit_cast(bf16x8_t, br[r % RB][m]);
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wr[r][t]), bfrag, acc[m][t], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            load_b(br[r % RB], min(ii + RB, il));
            if constexpr (WREFILL) load_w(wr[r], min(ii + RW, il));   // (WREFILL false: the whole K slice of a wave fits the ring)
            __builtin_amdgcn_sched_barrier(0);
        }
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
            if (tile0 + m >= p.tiles || nb0 + 2 * pq >= p.n_blocks) continue;
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
            uint16_t* o = p.h_out + (size_t)(g * p.tiles + tile0 + m) * 16 * I + ((size_t)(r >> 3) * 64 + qq * 16 + mm) * 8 + (r & 7);
            *reinterpret_cast<uint2*>(o) = make_uint2((uint32_t)y[0] | ((uint32_t)y[1] << 16), (uint32_t)y[2] | ((uint32_t)y[3] << 16));
        }
        return;
    }
    for (int q = wave; q < MT * NT; q += WV) {
        const int m = q / NT, t = q % NT;
        if (tile0 + m >= p.tiles || nb0 + t >= p.n_blocks || mm >= p.n_rows) continue;
        const f32x4_t a4 = reduced(m, t);
        const int n = (nb0 + t) * 16 + 4 * h;
        uint16_t y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = f2bf(a4[j]);
        uint16_t* o = p.y_out[g][tile0 + m] + (size_t)mm * p.ldo + n;
        *reinterpret_cast<uint2*>(o) = make_uint2((uint32_t)y[0] | ((uint32_t)y[1] << 16), (uint32_t)y[2] | ((uint32_t)y[3] << 16));
    }
}

template <int NT, int MT, int U, int EPI, bool FR, int RW, int RB, bool WREFILL>
static int launch_mt_v(const umoe_mt_args* a, hipStream_t s) {
    const size_t lds = (size_t)8 * MT * NT * 64 * 16;
    static bool configured = false;
    if (!configured) {
        UMOE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wstream_mt<NT, MT, U, EPI, FR, RW, RB, WREFILL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    umoe_router_args ra;
    memset(&ra, 0, sizeof(ra));
    dim3 grid((unsigned)ceil_div(a->n_blocks, NT), (unsigned)ceil_div(a->tiles, MT), (unsigned)a->num_groups);
    if (FR) {
        ra = *a->fused_router;
        UMOE_REQUIRE(ra.S <= (int)grid.x && ra.n_dyn == 9 && ra.n_fix == 2 && (ra.D == 2048 || ra.D == 4096) && ra.x && ra.gate_w && ra.expert_mask &&
                         !ra.logits_in && !ra.norm_only,
                     "umoe_gemm_mt: fused_router needs n_dyn 9 / n_fix 2, D 2048 / 4096, S <= %u", grid.x);
        grid.z += 1;
    }
    wstream_mt<NT, MT, U, EPI, FR, RW, RB, WREFILL><<<grid, 512, lds, s>>>(*a, ra);
    UMOE_LAUNCH_CHECK();
    return 0;
}

template <int NT, int MT, int U, int EPI, bool FR, int RW, int RB>
static int launch_mt(const umoe_mt_args* a, hipStream_t s) {
    // k-steps of the longest wave slice (U-step chunks dealt to 8 waves): no weight refill when the ring holds them all
    const int KB = a->k >> 5;
    const int longest = (KB % U == 0) ? U * ceil_div(KB / U, 8) : ceil_div(KB, 8);
    if (longest <= RW) return launch_mt_v<NT, MT, U, EPI, FR, RW, RB, false>(a, s);
    return launch_mt_v<NT, MT, U, EPI, FR, RW, RB, true>(a, s);
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
        if (a->tiles == 2) return fr ? launch_mt<8, 2, 1, UMOE_EPI_SWIGLU, true, 4, 2>(a, s) : launch_mt<8, 2, 1, UMOE_EPI_SWIGLU, false, 4, 2>(a, s);
        if (a->tiles == 4) return fr ? launch_mt<4, 4, 1, UMOE_EPI_SWIGLU, true, 8, 2>(a, s) : launch_mt<4, 4, 1, UMOE_EPI_SWIGLU, false, 8, 2>(a, s);
        return fr ? launch_mt<2, 8, 1, UMOE_EPI_SWIGLU, true, 8, 2>(a, s) : launch_mt<2, 8, 1, UMOE_EPI_SWIGLU, false, 8, 2>(a, s);
    }
    UMOE_REQUIRE(a->epilogue == UMOE_EPI_BF16 && !a->fused_router && a->ldo % 4 == 0, "umoe_gemm_mt: epilogue must be SwiGLU or bf16 (ldo %% 4 == 0)");
    for (int g = 0; g < a->num_groups; ++g)
        for (int t = 0; t < a->tiles; ++t) UMOE_REQUIRE(a->y_out[g][t], "umoe_gemm_mt: tile (%d, %d) has no output", g, t);
    // the K split of the dense decode down launch: 8 waves, 2-step chunks.  Shapes: a workgroup takes in its weight blocks AND
    // the fragments of its row tiles through the vector memory path at ~50 GB/s per CU (measured: 128 workgroups of 1 block x 8
    // tiles = 792 KB each took 18 us), so the launch is cut into 256 workgroups of <= 4 tiles
    if (a->tiles == 2) return launch_mt<2, 2, 2, UMOE_EPI_BF16, false, 8, 4>(a, s);
    return launch_mt<1, 4, 2, UMOE_EPI_BF16, false, 8, 4>(a, s);
}
