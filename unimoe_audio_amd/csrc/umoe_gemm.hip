// Weight-streaming grouped GEMM for M <= 16 rows per tile (decode) and ragged expert groups.
//
//   Y[rows(g), N(g)] = epilogue( prologue(A[rows(g), K(g)]) * W_g^T )
//
// Design (MI355X / gfx950):
//  * weights are pre-packed (WP16, see include/umoe.h) so every wave-instruction reads one
//    contiguous 1 KiB fragment straight into VGPRs -- no LDS round trip for the operand that is
//    streamed once (cdna_hip_programming.md "GEMV / M <= 16 decode weights" row);
//  * the 16-row activation tile is staged once per workgroup in LDS, 256-byte segments rotated by
//    the row index so both the staging writes and the ds_read_b128 fragment reads are bank-conflict
//    free (lane-group h reads K-quarter h: its bank offset is a multiple of 256 B);
//  * v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the A operand: each lane ends with 4 consecutive
//    output features of one token -> one 8-byte store per lane;
//  * the 4 waves of a workgroup split K; partial tiles are reduced through LDS in fixed order
//    (bitwise reproducible, no atomics);
//  * groups (routed experts with ragged row lists, shared experts, dense layers) share one launch:
//    grid = (n-tiles, row-tiles, groups); row counts / offsets are read on device, so the host
//    never synchronises on the router's result.
// Roofline: HBM. Algorithmic bytes per launch = sum over groups hit of N*K*2 (+ activations).
#include "umoe_common.h"
#include "umoe_router_dev.h"
#include "umoe_riders_dev.h"
#include <string.h>
#ifndef UMOE_POLLER_WAVE
#define UMOE_POLLER_WAVE 0     // build-time A/B of the poller wave (see wstream_body): measured 3.094-3.11 vs 3.075-3.087 ms/step -- off
#endif
#include <stdlib.h>

// ------------------------------------------------------------------------------------ packing
__global__ void pack_kernel(const uint16_t* __restrict__ Wa, const uint16_t* __restrict__ Wb, int N, int K, int KB,
                            long total, uint16_t* __restrict__ out) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const long t = idx >> 6;
    const int i = (int)(t % KB);
    int nb = (int)(t / KB);
    const uint16_t* W = Wa;
    if (Wb) {  // interleaved gate/up blocks
        W = (nb & 1) ? Wb : Wa;
        nb >>= 1;
    }
    const int row = nb * 16 + (lane & 15);
    const int col = (lane >> 4) * (K >> 2) + i * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < N) v = ld16(W + (size_t)row * K + col);
    st16(out + idx * 8, v);
}

extern "C" size_t umoe_packed_elems(int N, int K) { return (size_t)ceil_div(N, 16) * 16 * (size_t)K; }

extern "C" int umoe_pack_weight(const uint16_t* W, int N, int K, uint16_t* packed, umoe_stream_t stream) {
    UMOE_REQUIRE(W && packed && N > 0 && K > 0 && K % 32 == 0, "umoe_pack_weight: need K %% 32 == 0 (N=%d K=%d)", N, K);
    const int NB = ceil_div(N, 16), KB = K / 32;
    const long total = (long)NB * KB * 64;
    pack_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, (hipStream_t)stream>>>(W, nullptr, N, K, KB, total, packed);
    UMOE_LAUNCH_CHECK();
    return 0;
}

extern "C" int umoe_pack_gate_up(const uint16_t* Wg, const uint16_t* Wu, int I, int K, uint16_t* packed,
                                 umoe_stream_t stream) {
    UMOE_REQUIRE(Wg && Wu && packed && I > 0 && I % 16 == 0 && K % 32 == 0,
                 "umoe_pack_gate_up: need I %% 16 == 0 and K %% 32 == 0 (I=%d K=%d)", I, K);
    const int KB = K / 32;
    const long total = (long)(2 * I / 16) * KB * 64;
    pack_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, (hipStream_t)stream>>>(Wg, Wu, I, K, KB, total, packed);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ kernel
__device__ __forceinline__ int lds_chunk_off(int QS, int h, int i, int m) {
    // 16-byte chunk i of K-quarter h, row m: 256-byte segments, slot rotated by the row index
    return h * QS + (i >> 4) * 256 + (((i & 15) + m) & 15) * 16;
}

struct umoe_group_pack { umoe_group_t g[UMOE_GROUPS_INLINE]; };

// FR ("fused router", umoe_gemm_args.fused_router): the z-slice in front of the first group holds one workgroup per token that runs the
// Top-P router (threads 0..255; umoe_router_dev.h) instead of a GEMM tile -- 16 workgroups on CUs the 226 GEMM workgroups leave
// idle.  The GEMM of the dense-expert decode layout does not read the router's outputs, the combine launch after it does.
// PUB (umoe_gemm_args.rider_pub, with FR): the riders also WRITE the normalised rows this GEMM stages (ra.h_out == p.a) and publish
// one flag per row; the GEMM workgroups request their first weight chunk, then wait for the 16 flags (one lane each, bounded), then
// stage the rows with sc1 loads -- the hand-off hides behind the weight stream's first round trip and the RMSNorm launch disappears.
// XW (fused expert launch, moe_fused_kernel below): 1 = this GEMM PUBLISHES its output tile to workgroups of the same launch (SwiGLU
// epilogue: write-through stores, drain, one flag per workgroup); 2 = this GEMM's activation rows were published that way (weights
// first, then wait for the producers' flags, then sc1 loads; the second register stage of the weight stream is requested right
// behind the rows).  Returns 0 when the workgroup finished a tile, 1 when it had none (riders, tile-less workgroups of the box).
struct wg_coord { unsigned x, y, z, gx; };     // workgroup coordinates inside this GEMM's grid box and the box's x extent
struct umoe_fuse_x {
    uint32_t* flags;                        // one word per workgroup of the PRODUCING GEMM (z * gx + x); epochs as in umoe_rider_pub
    int prod_base[UMOE_GROUPS_INLINE];      // per group of the CONSUMING GEMM: first flag and number of flags it waits for
    int prod_n[UMOE_GROUPS_INLINE];
};
template <int NT, int U, int PRO, int EPI, int WV, bool FR, bool PUB, int XW, bool BV = false>
__device__ __forceinline__ int wstream_body(const umoe_gemm_args& p, const umoe_group_pack& gp, const umoe_router_args& ra, const int rider_mode,
                                            const umoe_rider_pub& pub, const umoe_fuse_x& fx, const wg_coord blk, char* smem
#ifdef UMOE_TIMELINE
                                            , tl_state* tl_carry = nullptr      // XW 1: the stamps leave with the caller, written after the second GEMM
#endif
) {
    // riders (FR).  rider_mode 1: an extra z-slice in front of the first group (x = token); 2: the launch's DEAD workgroups (x beyond
    // a short group's tiles -- the grid is a box over the widest group) take the tokens in (z, x) order: no extra workgroups, the
    // launch still fits the chip in one wave (with the extra slice 275 workgroups were launched on 256 CUs and 7 real GEMM tiles
    // waited for a CU).
    if constexpr (FR) {
        int token = -1;
        if (rider_mode == 1) {
            if (blk.z == 0) token = (int)blk.x;
        } else {
            const int live = (gp.g[blk.z].n_blocks + NT - 1) / NT;
            if ((int)blk.x >= live) {
                token = (int)blk.x - live;
                for (unsigned i = 0; i < blk.z; ++i) token += (int)blk.gx - (gp.g[i].n_blocks + NT - 1) / NT;
            }
        }
        if (token >= 0) {
            if (token < ra.S && blk.y == 0 && threadIdx.x < 256) {
                TL_ENTER(5);
                uint32_t* pf = nullptr;
                uint32_t pe = 0;
                if constexpr (PUB) {
                    pf = pub.flags;
                    pe = *pub.step * (uint32_t)pub.layers + (uint32_t)pub.layer + 1u;
                }
                unsigned long long* rsp = nullptr;
                if constexpr (PUB && PRO == UMOE_PRO_RMSNORM) {   // the scale only (see router4_body rs_pub); no row flags
                    rsp = pub.rs;
                    pf = nullptr;
                }
                if (ra.logits_bf16) router4_body<9, 2, 1, false>(ra, token, threadIdx.x, reinterpret_cast<float*>(smem) TL_PASS, pf, pe, rsp);
                else router4_body<9, 2, 0, false>(ra, token, threadIdx.x, reinterpret_cast<float*>(smem) TL_PASS, pf, pe, rsp);
                TL_EXIT(5);
            }
            return 1;
        }
    }
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    constexpr int KID = EPI == UMOE_EPI_SWIGLU ? 2 : (EPI == UMOE_EPI_F32 ? 4 : (EPI == UMOE_EPI_BF16_RESID ? 1 : (NT == 1 ? 0 : 3)));
    (void)KID;
    TL_ENTER(KID);
    // group descriptor: from the kernel arguments when the host passed them by value (scalar loads, no HBM round trip)
    // (measured and removed: blockIdx.x enumerating equal slices of ALL groups' gate/up pairs inside THIS kernel -- 42.4 vs 37.3 us: a
    //  6-pair slice re-read its 7th register slot; the byte-balanced form is its own launch now, umoe_moe_flat.hip)
    const unsigned zg = blk.z - ((FR && rider_mode == 1) ? 1u : 0u);      // group index
    // BV (descriptors known to be by value: riders / the fused launch need them, the small decode launches are dispatched on it): plain
    // kernel-argument reads = scalar loads.  Left to a run-time choice the compiler selects between the two ADDRESSES and reads the
    // descriptor with flat loads through the vector memory path -- two dependent round trips in front of the first request.
    const umoe_group_t g = (BV || FR || XW != 0) ? gp.g[zg] : (p.groups_host ? gp.g[zg] : p.groups[zg]);
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int ks = ksplit > 1 ? (int)(blk.x % ksplit) : 0;      // K-slice of this workgroup (fp32 partial slab `ks`)
    const int nb0 = (ksplit > 1 ? (int)(blk.x / ksplit) : (int)blk.x) * NT;
    if (nb0 >= g.n_blocks) return 1;

    const int K = g.k, KB = K >> 5;
    // this workgroup covers MFMA k-steps [ia, ib) of every K-quarter; only those activation chunks are staged
    const int ia = (KB * ks) / ksplit, ib = (KB * (ks + 1)) / ksplit;
    const int QS = (((ib - ia) * 16) + 255) & ~255;  // bytes of one staged K-quarter slice of a row, padded to 256
    const int RS = QS * 4;
    const int tid = threadIdx.x, lane = tid & 63;
    // wave-uniform by construction: keep it (and the K split i0 / i1 derived from it) in SGPRs, so that every guard around an MFMA
    // is a SCALAR branch.  With a per-lane condition hipcc may predicate a short block through EXEC without a skip branch, and
    // MFMA ignores EXEC: the guarded MFMA of a partial chunk would then run on its clamped duplicate operands (seen in an
    // experiment with register-resident activations: last k-step counted twice; tests/test_abi_cpu.py scans the assembly).
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- weight stream set-up: the first chunk is requested BEFORE anything that depends on device-produced data
    //      (row counts, gather lists, activations), so the HBM latency of the stream overlaps the whole prologue ----
    // whole U-step chunks per wave when the slice divides (K = 2752: 43 chunks of 2 steps over 8 waves): a partial last
    // chunk would re-read its clamped step (measured: +7 % HBM traffic on the down projection, profiles/r01e_pmc_traffic.md)
    int i0, i1;
    if ((ib - ia) % U == 0) {
        const int units = (ib - ia) / U;
        i0 = ia + U * ((units * wave) / WV);
        i1 = ia + U * ((units * (wave + 1)) / WV);
    } else {
        i0 = ia + ((ib - ia) * wave) / WV;
        i1 = ia + ((ib - ia) * (wave + 1)) / WV;
    }
    // (K may have come from a descriptor in memory: pin the wave-uniform slice bounds into SGPRs, see `wave` above)
    i0 = __builtin_amdgcn_readfirstlane(i0);
    i1 = __builtin_amdgcn_readfirstlane(i1);
    f32x4_t acc[NT];
    const u32x4_t* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const int nb = min(nb0 + t, g.n_blocks - 1);  // tail tiles re-read the last block; never stored
        wp[t] = reinterpret_cast<const u32x4_t*>(g.w) + ((size_t)nb * KB) * 64 + lane;
    }
    u32x4_t w0[NT][U], w1[NT][U];
    auto load_chunk = [&](u32x4_t (&dst)[NT][U], int ibase) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ii = min(ibase + u, i1 - 1);
#pragma unroll
            for (int t = 0; t < NT; ++t) dst[t][u] = __builtin_nontemporal_load(wp[t] + (size_t)ii * 64);
        }
    };
    // Order of requests.  A wave's loads return in issue order, so whatever is requested first is usable first:
    //  * static groups (dense layers, shared experts, dense-expert decode): the activation rows (and norm weights) are
    //    requested FIRST, the weight stream right behind them -- the tile is staged while the first chunk is in flight;
    //  * ragged groups: the activation addresses hang on device-produced tables (count -> offset -> gather list), so the
    //    weight stream goes first and overlaps that chain.
    // (PUB / XW 2: weights first too -- the rows do not exist yet; PUB with the RMSNorm prologue: the RAW rows exist, only their scale
    //  is handed over, so the rows are requested first like any static group's)
    constexpr bool ROWS_HANDED = (PUB && PRO != UMOE_PRO_RMSNORM) || XW == 2;
    const bool ragged = ROWS_HANDED || g.count || g.row_off || g.rows;
    // second register stage requested before the staging too (static groups): HBM has work queued for the whole prologue.
    // Only where the registers allow it without spilling (checked per instantiation with -S: private_segment_fixed_size 0).
    constexpr bool DEEP = false;   // measured: gate/up NT 14 35.3 -> 42.6 us, down 23.8 -> 29.8 us -- MORE bytes in flight made it slower
    // The POLLER (experiment, off): a wave's loads return in issue order, so a flag poll issued behind the wave's own weight chunk
    // comes back only when that chunk has landed; the last wave polled FIRST, with nothing in its queue, and requested its first chunk
    // only when the rows were there.  Result (scripts/timeline_wgs.py): detection did NOT get earlier (rows staged 7.4 vs 6.6 us after
    // entry, riders' flag stored at 3.3 us either way) -- the 3 us between flag store and detection are the memory system's latency
    // under the weight stream (write-through of the flag, the poll's own round trip), not the wave's queue.
    const bool poller = ROWS_HANDED && UMOE_POLLER_WAVE && wave == WV - 1;
    if (ragged && i0 < i1 && !poller) load_chunk(w0, i0);
    // (XW 2: measured on the fused expert launch, 300 decode steps: second register stage requested here, in front of the wait for
    //  the producers, 3.26-3.29 ms/step; right behind the rows 3.26-3.28; a THIRD stage in front of the wait 3.40-3.43 -- more bytes
    //  in flight per CU made the launch slower, as in the two-launch form)
    TL_MARK(KID, 4);

    const int count = g.count ? *g.count : g.static_count;
    const int roff = g.row_off ? *g.row_off : 0;
    const int row0 = blk.y * 16;
    if (row0 >= count) return 1;   // (an expert no row chose: its first chunk was requested for nothing -- rare at 16 rows)
    if constexpr (XW == 2) {
        // wait for the workgroups of THIS launch that produced this group's rows: lane i of wave 0 polls producer i's flag (bounded)
        const int np = fx.prod_n[zg];
        if (UMOE_POLLER_WAVE ? (poller && lane < np) : (tid < np)) {
            const int tid = lane;      // (flag index of this lane)
            const uint32_t epoch = *pub.step * (uint32_t)pub.layers + (uint32_t)pub.layer + 1u;
            umoe_gu32* f = reinterpret_cast<umoe_gu32*>(reinterpret_cast<uintptr_t>(fx.flags + fx.prod_base[zg] + tid));
            umoe_gu32* err = reinterpret_cast<umoe_gu32*>(reinterpret_cast<uintptr_t>(pub.err));
            const unsigned long long t0 = wall_clock64();
            for (unsigned spins = 0;; ++spins) {
                if ((int32_t)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) >= 0) break;
                __builtin_amdgcn_s_sleep(1);
                if ((spins & 1023u) == 1023u) {
                    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;      // an earlier cause (kept) ends every wait
                    if (wall_clock64() - t0 > 200000000ull) {      // this lane's OWN timeout: first cause wins
                        uint32_t zero = 0u;
                        __hip_atomic_compare_exchange_strong(err, &zero, 3u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
        }
        if (poller && i0 < i1) load_chunk(w0, i0);
        __syncthreads();
    }
    if constexpr (PUB && PRO != UMOE_PRO_RMSNORM) {
        // wait for the riders of THIS launch: lanes 0..count-1 of wave 0 poll one row flag each; bounded (a rider that never runs --
        // an admitted workgroup that is not resident -- ends the wait with the sticky error word set)
        if (UMOE_POLLER_WAVE ? (poller && lane < count) : (tid < count)) {
            const int tid = lane;      // (row of this lane)
            const uint32_t epoch = *pub.step * (uint32_t)pub.layers + (uint32_t)pub.layer + 1u;
            umoe_gu32* f = reinterpret_cast<umoe_gu32*>(reinterpret_cast<uintptr_t>(pub.flags + ((blk.x + blk.z * blk.gx) % UMOE_FLAG_REPL) * 16 + row0 + tid));
            umoe_gu32* err = reinterpret_cast<umoe_gu32*>(reinterpret_cast<uintptr_t>(pub.err));
            const unsigned long long t0 = wall_clock64();
            for (unsigned spins = 0;; ++spins) {
                if ((int32_t)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) >= 0) break;
                __builtin_amdgcn_s_sleep(1);
                if ((spins & 1023u) == 1023u) {
                    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;      // an earlier cause (kept) ends every wait
                    if (wall_clock64() - t0 > 200000000ull) {      // this lane's OWN timeout: first cause wins
                        uint32_t zero = 0u;
                        __hip_atomic_compare_exchange_strong(err, &zero, 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
        }
        if (poller && i0 < i1) load_chunk(w0, i0);
        __syncthreads();
    }

    // ---- stage the 16-row activation tile: every global load of a thread is in flight before the first LDS write ----
    {
        constexpr int TPR = WV * 4;  // threads per activation row
        const int m = tid / TPR, sub = tid % TPR;
        const int r = row0 + m;
        const bool valid = r < count;   // rows beyond the count stay unwritten: a token is one MFMA column, garbage
                                        // there never reaches another token's outputs and is never stored
        // EVERY thread loads (threads of rows beyond the count re-read the tile's first row, chunks beyond the slice its last chunk):
        // straight-line loads the compiler can count, so the LDS writes wait for the rows only (vmcnt(n)) while the weight chunk
        // requested behind them is still in flight.  With a branch around each load the wait was vmcnt(0) -- the staging ended when
        // the WEIGHTS had landed (scripts/timeline_wgs.py: loads issued over 1.0 us, tile staged 2.7 us after the first request).
        const int rl = valid ? r : row0;
        const long arow = g.rows ? (long)g.rows[roff + rl] : (long)(g.a_row_base + roff + rl);
        const uint16_t* src = p.a + arow * (long)p.lda + g.a_col_off;
        char* dst = smem + m * RS;
        const int Q8 = KB;       // 16-byte chunks per quarter of the full row
        const int QW = ib - ia;  // chunks per quarter staged by this workgroup
        // thread (m, sub) owns chunks i = sub + TPR*j (j = 0..3) of each K-quarter h per round: 16 loads, no division
        const bool single = PRO == UMOE_PRO_RMSNORM && QW == Q8 && Q8 <= 4 * TPR;   // whole row in one round
        float rs = 0.f;
        if (PRO == UMOE_PRO_RMSNORM && !single) {
            float ss = 0.f;
            if (valid)
                for (int c = sub; c < 4 * Q8; c += TPR) {
                    float f[8];
                    unpack8(ld16(src + c * 8), f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
                }
#pragma unroll
            for (int o = TPR / 2; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
            rs = rsqrtf(ss / (float)K + p.rms_eps);
        }
        for (int ib0 = 0; ib0 < QW; ib0 += 4 * TPR) {
            uint4 buf[16];
            // norm weights: one 16-byte chunk per thread into LDS (behind the tile), read back per chunk below --
            // 4 registers instead of 64 (single-round RMSNorm only: 4*Q8 <= threads)
            uint4 nw1 = make_uint4(0, 0, 0, 0);
            char* nw_lds = smem + 16 * RS;
            if (single && tid < 4 * Q8) nw1 = ld16(p.norm_w + tid * 8);
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const int h = n >> 2, i = min(ib0 + sub + TPR * (n & 3), QW - 1);
                if constexpr (ROWS_HANDED) {   // rows handed over inside this launch: every load of them is an sc1 load
                    typedef uint32_t u32x4_pub __attribute__((ext_vector_type(4)));
                    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.a), 0, XW == 2 ? 0x7fffffff : 16 * p.lda * 2, 0x00020000);
                    const u32x4_pub t4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((arow * (long)p.lda + g.a_col_off + (h * Q8 + ia + i) * 8) * 2), 0, 16);
                    buf[n] = make_uint4(t4[0], t4[1], t4[2], t4[3]);
                } else {
                    buf[n] = ld16(src + (h * Q8 + ia + i) * 8);
                }
            }
            if (ib0 == 0 && !ragged) {
                __builtin_amdgcn_sched_barrier(0);
                if (i0 < i1) load_chunk(w0, i0);
                if (DEEP && i0 + U < i1) load_chunk(w1, i0 + U);   // both register stages in flight while the tile is staged
                __builtin_amdgcn_sched_barrier(0);
            }

            if (XW == 2 && ib0 == 0) {     // the second register stage right behind the rows (returns: first stage, rows, second stage)
                __builtin_amdgcn_sched_barrier(0);
                if (i0 + U < i1) load_chunk(w1, i0 + U);
                __builtin_amdgcn_sched_barrier(0);
            }
            TL_MARK(KID, 9);
            if (single && PUB) {
                // the row's scale comes from its rider (one {rs, epoch} granule per row, umoe_router_dev.h rs_pub): every thread of the
                // row polls that granule (two addresses per wave), bounded; the raw row is already in registers
                if (tid < 4 * Q8) st16(nw_lds + tid * 16, nw1);
                const uint32_t epoch = *pub.step * (uint32_t)pub.layers + (uint32_t)pub.layer + 1u;
                typedef uint32_t u32x2_g __attribute__((ext_vector_type(2)));
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(pub.rs, 0, 8 * UMOE_EP_PARTS, 0x00020000);
                umoe_gu32* err = reinterpret_cast<umoe_gu32*>(reinterpret_cast<uintptr_t>(pub.err));
                const unsigned long long t0 = wall_clock64();
                u32x2_g gr = {0u, 0u};
                for (unsigned spins = 0;; ++spins) {
                    gr = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (valid ? r : row0) * 8, 0, 16);
                    if ((int32_t)(gr[1] - epoch) >= 0) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((spins & 1023u) == 1023u) {
                        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;      // an earlier cause (kept) ends every wait
                        if (wall_clock64() - t0 > 200000000ull) {      // this lane's OWN timeout: first cause wins
                            uint32_t zero = 0u;
                            __hip_atomic_compare_exchange_strong(err, &zero, 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                }
                rs = __int_as_float((int)gr[0]);
                __syncthreads();
            } else if (single) {
                float ss = 0.f;
                if (TPR == 32 && Q8 == 64) {
                    // K = 2048 on 8 waves: the SAME summation tree as the router body (umoe_router_dev.h router4_body: lane l of wave h
                    // sums the 8 squares of chunk 64 h + l, xor-butterfly 32, 16, .. 1, the four wave sums added in order) -- thread
                    // `sub` holds chunks sub and sub + 32 of every quarter, i.e. both operands of the butterfly's first level.  The
                    // normalised rows are then bit-identical to the rows the router launches write, whichever launch made them.
                    float q4[4];
#pragma unroll
                    for (int hq4 = 0; hq4 < 4; ++hq4) {
                        float c2[2];
#pragma unroll
                        for (int k2 = 0; k2 < 2; ++k2) {
                            float f[8];
                            unpack8(buf[hq4 * 4 + k2], f);
                            float cs = 0.f;
#pragma unroll
                            for (int j = 0; j < 8; ++j) cs += f[j] * f[j];
                            c2[k2] = cs;
                        }
                        float v = c2[0] + c2[1];
#pragma unroll
                        for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
                        q4[hq4] = v;
                    }
                    ss = ((q4[0] + q4[1]) + q4[2]) + q4[3];
                } else {
#pragma unroll
                    for (int n = 0; n < 16; ++n) {
                        float f[8];
                        unpack8(buf[n], f);
                        const bool own = ib0 + sub + TPR * (n & 3) < QW;     // (a clamped re-read beyond the slice does not count)
#pragma unroll
                        for (int j = 0; j < 8; ++j) ss += own ? f[j] * f[j] : 0.f;
                    }
#pragma unroll
                    for (int o = TPR / 2; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
                }
                rs = rsqrtf(ss / (float)K + p.rms_eps);
                if (tid < 4 * Q8) st16(nw_lds + tid * 16, nw1);
                __syncthreads();
                // keep the row slice packed (64 registers): without this the compiler carries all 128 unpacked floats
                // from the sum of squares to the scaling and spills beside the in-flight weight chunk
#pragma unroll
                for (int n = 0; n < 16; ++n)
                    asm volatile("" : "+v"(buf[n].x), "+v"(buf[n].y), "+v"(buf[n].z), "+v"(buf[n].w));
            }
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const int h = n >> 2, i = ib0 + sub + TPR * (n & 3);
                if (valid && i < QW) {
                    uint4 u = buf[n];
                    if (PRO == UMOE_PRO_RMSNORM) {
                        float f[8], w[8];
                        unpack8(u, f);
                        unpack8(single ? *reinterpret_cast<const uint4*>(nw_lds + (h * Q8 + i) * 16)
                                       : ld16(p.norm_w + (h * Q8 + ia + i) * 8), w);
#pragma unroll
                        for (int j = 0; j < 8; ++j) f[j] = w[j] * rbf(f[j] * rs);
                        u = pack8(f);
                    }
                    st16(dst + lds_chunk_off(QS, h, i, m), u);
                }
            }
        }
    }
    // ---- epilogue operands (bias / residual) requested now, consumed after the reduction ---------------------------
    constexpr int TPW = (NT + WV - 1) / WV;   // output tiles finished by one wave
    const int hq = lane >> 4, mq = lane & 15;
    float4 bias_r[TPW];
    uint2 resid_r[TPW];
    const bool bias_vec = g.bias && ((size_t)g.bias & 15) == 0;
    if (EPI != UMOE_EPI_SWIGLU) {
        const long orow_q = (long)g.out_row_base + roff + row0 + mq;
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            const int t = wave + q * WV;
            const int n = (nb0 + t) * 16 + 4 * hq;
            bias_r[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            resid_r[q] = make_uint2(0, 0);
            const bool live = t < NT && nb0 + t < g.n_blocks && row0 + mq < count && n + 3 < p.n_valid;
            if (live && bias_vec) bias_r[q] = *reinterpret_cast<const float4*>(g.bias + n);
            if (EPI == UMOE_EPI_BF16_RESID && live && (p.ldo & 3) == 0)
                resid_r[q] = *reinterpret_cast<const uint2*>(p.resid + orow_q * p.ldo + n);
        }
    }
    TL_MARK(KID, 5);
    __syncthreads();
    TL_MARK(KID, 6);

    // ---- stream weights (double-buffered in registers), 4 waves split K -------------------------------------
    const int h = lane >> 4, mm = lane & 15;
    const char* bbase = smem + mm * RS;
    auto compute_chunk = [&](const u32x4_t (&src)[NT][U], int ibase) {
        if (U >= 4 && ibase + U <= i1) {
            // whole chunk (scalar condition): every fragment read is issued before the first MFMA.  With the per-step guard below each
            // ds_read waited for its own round trip in front of its MFMA: 16 steps took 0.92 us in the QKV / o_proj launches
            uint4 bv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) bv[u] = *reinterpret_cast<const uint4*>(bbase + lds_chunk_off(QS, h, ibase + u - ia, mm));
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bf16x8_t bfrag = __builtin_bit_cast(bf16x8_t, bv[u]);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, src[t][u]), bfrag, acc[t], 0, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ii = ibase + u;
            if (ii < i1) {
                const uint4 bv = *reinterpret_cast<const uint4*>(bbase + lds_chunk_off(QS, h, ii - ia, mm));
                const bf16x8_t bfrag = __builtin_bit_cast(bf16x8_t, bv);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, src[t][u]), bfrag,
                                                                    acc[t], 0, 0, 0);
            }
        }
    };
    for (int i = i0; i < i1; i += 2 * U) {
        if (i + U < i1 && !(((DEEP && !ragged) || XW == 2) && i == i0)) load_chunk(w1, i + U);
        compute_chunk(w0, i);
        if (i + 2 * U < i1) load_chunk(w0, i + 2 * U);
        if (i + U < i1) compute_chunk(w1, i + U);
    }

    // ---- fixed-order cross-wave reduction through the (now free) staging area -------------------------------
    TL_MARK(KID, 7);
    __syncthreads();
    TL_MARK(KID, 8);
    f32x4_t* red = reinterpret_cast<f32x4_t*>(smem);
#pragma unroll
    for (int t = 0; t < NT; ++t) red[(wave * NT + t) * 64 + lane] = acc[t];
    __syncthreads();
    auto reduced = [&](int t) -> f32x4_t {
        f32x4_t s = red[t * 64 + lane];
#pragma unroll
        for (int w = 1; w < WV; ++w) {
            const f32x4_t v = red[(w * NT + t) * 64 + lane];
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        }
        return s;
    };

    // ---- epilogue: lane (h, mm) owns features 4h..4h+3 of token row mm; tiles are spread over the waves -----
    const int r = row0 + mm;
    if (XW != 1 && r >= count) return 0;
    const long orow = (long)g.out_row_base + roff + r;
    if (EPI == UMOE_EPI_SWIGLU) {
        for (int q = wave; q < NT / 2; q += WV) {
            if (nb0 + 2 * q >= g.n_blocks) break;
            const int col = (nb0 / 2 + q) * 16 + 4 * h;
            const long orow_q = orow;
            const f32x4_t ga = reduced(2 * q), ua = reduced(2 * q + 1);
            uint16_t y[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gt = rbf(ga[j]);
                const float up = rbf(ua[j]);
                const float si = rbf(gt / (1.0f + expf(-gt)));
                y[j] = f2bf(si * up);
            }
            uint16_t* o = reinterpret_cast<uint16_t*>(p.out) + orow_q * p.ldo + col;
            if constexpr (XW == 1) {     // handed to workgroups of this launch: write-through (sc1) store
                typedef uint32_t u32x2_pub __attribute__((ext_vector_type(2)));
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint16_t*>(p.out), 0, 0x7fffffff, 0x00020000);
                const u32x2_pub v2 = {(uint32_t)y[0] | ((uint32_t)y[1] << 16), (uint32_t)y[2] | ((uint32_t)y[3] << 16)};
                if (r < count) __builtin_amdgcn_raw_buffer_store_b64(v2, rsrc, (int)((orow_q * p.ldo + col) * 2), 0, 16);
            } else {
                *reinterpret_cast<uint2*>(o) = make_uint2((uint32_t)y[0] | ((uint32_t)y[1] << 16), (uint32_t)y[2] | ((uint32_t)y[3] << 16));
            }
        }
        if constexpr (XW == 1) {
            // publish: every storing wave drains its write-through stores, the workgroup meets, one lane raises this workgroup's flag
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const uint32_t epoch = *pub.step * (uint32_t)pub.layers + (uint32_t)pub.layer + 1u;
                __hip_atomic_store(reinterpret_cast<umoe_gu32*>(reinterpret_cast<uintptr_t>(fx.flags + blk.z * blk.gx + blk.x)), epoch, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#ifdef UMOE_TIMELINE
        if (XW == 1 && tl_carry) { tl_st.m[2] = wall_clock64(); *tl_carry = tl_st; return 0; }
#endif
        TL_EXIT(KID);
        return 0;
    }
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int t = wave + q * WV;
        if (t >= NT || nb0 + t >= g.n_blocks) break;
        const f32x4_t a4 = reduced(t);
        const int n = (nb0 + t) * 16 + 4 * h;
        const bool fast = n + 3 < p.n_valid;   // whole 4-feature group valid: operands were prefetched
        float v[4];
        const float bq[4] = {bias_r[q].x, bias_r[q].y, bias_r[q].z, bias_r[q].w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            v[j] = a4[j] + ((fast && (bias_vec || !g.bias)) ? bq[j] : ((g.bias && n + j < p.n_valid) ? g.bias[n + j] : 0.f));
        if (EPI == UMOE_EPI_F32 || EPI == UMOE_EPI_F32_RAW) {
            float* o = reinterpret_cast<float*>(p.out) + (size_t)ks * p.part_stride + orow * p.ldo + n;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (n + j < p.n_valid) o[j] = (EPI == UMOE_EPI_F32) ? rbf(v[j]) : v[j];
        } else {
            uint16_t* o = reinterpret_cast<uint16_t*>(p.out) + orow * p.ldo + n;
            uint16_t y[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = rbf(v[j]);
                if (EPI == UMOE_EPI_BF16_RESID && n + j < p.n_valid) {
                    const uint32_t rw = j < 2 ? resid_r[q].x : resid_r[q].y;
                    const float rv = (fast && (p.ldo & 3) == 0) ? __uint_as_float((j & 1) ? (rw & 0xffff0000u) : (rw << 16))
                                                                : bf2f(p.resid[orow * p.ldo + n + j]);
                    x = rv + x;
                }
                y[j] = f2bf(x);
            }
            if (n + 3 < p.n_valid && (p.ldo & 3) == 0) {
                *reinterpret_cast<uint2*>(o) = make_uint2((uint32_t)y[0] | ((uint32_t)y[1] << 16), (uint32_t)y[2] | ((uint32_t)y[3] << 16));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n + j < p.n_valid) o[j] = y[j];
            }
        }
    }
    TL_EXIT(KID);
    return 0;
}

template <int NT, int U, int PRO, int EPI, int WV, bool FR = false, bool PUB = false, bool BV = false>
__global__ __launch_bounds__(WV * 64, 2) void wstream_gemm(const umoe_gemm_args p, const umoe_group_pack gp, const umoe_router_args ra, const int rider_mode,
                                                            const umoe_rider_pub pub) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const umoe_fuse_x fx{};
    (void)wstream_body<NT, U, PRO, EPI, WV, FR, PUB, 0, BV>(p, gp, ra, rider_mode, pub, fx, wg_coord{blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x}, smem);
}

// A one-block-per-workgroup decode GEMM (QKV with bias / o_proj with residual: NT 1, 4 waves) with ROW riders in front
// (umoe_riders_dev.h): workgroups [0, n_riders) run the row kernel that produces this GEMM's activation rows (RK 2: MoE combine +
// residual + RMSNorm of the previous layer; RK 4: the same over the return slab of the expert-parallel exchange) and hand them over;
// the GEMM tiles follow.
template <int EPI, int RK>
__global__ __launch_bounds__(256, 2) void wstream_gemm_rk(const umoe_gemm_args p, const umoe_group_pack gp, const umoe_rider_pub pub, const umoe_rider2 r2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < r2.n_riders) {
        const uint32_t epoch = *pub.step * (uint32_t)pub.layers + (uint32_t)pub.layer + 1u;
        combine_row_dense<RK == 4>(r2.cb, r2, (int)blockIdx.x, reinterpret_cast<float*>(smem), pub.flags + blockIdx.x, epoch);
        return;
    }
    const umoe_fuse_x fx{};
    const umoe_router_args ra{};
    (void)wstream_body<1, 16, UMOE_PRO_PLAIN, EPI, 4, false, true, 0, true>(p, gp, ra, 0, pub, fx,
                                                                        wg_coord{blockIdx.x - (unsigned)r2.n_riders, 0u, 0u, gridDim.x - (unsigned)r2.n_riders}, smem);
}

// The two expert GEMMs of a dense decode layer in ONE launch (8 routed + 2 shared experts, 16 rows): every workgroup computes its
// gate/up slice (7 pairs, riders and their hand-off as in the gate/up launch), publishes it, then takes a down-projection slice (6
// blocks): it requests that slice's first weights, waits for the gate/up workgroups of ITS expert only, stages the rows and streams.
// What the fusion buys: no launch boundary, and the down projection's weight stream starts while other workgroups still finish
// gate/up -- as two launches the chip idled through the down projection's 7 us prologue.  Same tiles, same K split, same reduction
// order as the two launches: bit-identical outputs.
template <int GPRO>     // prologue of the gate/up GEMM: PLAIN = the riders hand the normalised rows over, RMSNORM = only their scales
__global__ __launch_bounds__(512, 1) void moe_fused_kernel(const umoe_gemm_args pg, const umoe_group_pack gg, const umoe_gemm_args pd, const umoe_group_pack gd,
                                                            const umoe_router_args ra, const int rider_mode, const umoe_rider_pub pub, const umoe_fuse_x fx,
                                                            const int dn_per_group) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const wg_coord b{blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x};
#ifdef UMOE_TIMELINE
    tl_state tl_first;
    if (wstream_body<14, 1, GPRO, UMOE_EPI_SWIGLU, 8, true, true, 1>(pg, gg, ra, rider_mode, pub, fx, b, smem, &tl_first)) return;
#else
    if (wstream_body<14, 1, GPRO, UMOE_EPI_SWIGLU, 8, true, true, 1>(pg, gg, ra, rider_mode, pub, fx, b, smem)) return;
#endif
    // live index of this workgroup among the gate/up tiles -> its down-projection slice
    int li = (int)b.x;
    for (unsigned i = 0; i < b.z; ++i) li += (gg.g[i].n_blocks + 13) / 14;
    const unsigned dz = (unsigned)(li / dn_per_group), dx = (unsigned)(li % dn_per_group);
    if ((int)dz < pd.num_groups) {
        __syncthreads();     // (the reduction slab of the first GEMM is the staging area of the second)
        (void)wstream_body<6, 2, UMOE_PRO_PLAIN, UMOE_EPI_BF16, 8, false, false, 2>(pd, gd, ra, 0, pub, fx, wg_coord{dx, 0u, dz, (unsigned)dn_per_group}, smem);
    }
#ifdef UMOE_TIMELINE
    tl_exit(tl_first, 2, tl_first.m[2]);
#endif
}

UMOE_TL_SETTER(gemm)
// ------------------------------------------------------------------------------------ launcher
static size_t gemm_lds_bytes(int max_k, int NT, int WV, int ksplit, int pro = UMOE_PRO_PLAIN) {
    const int KB = max_k >> 5;
    const int per = (KB + ksplit - 1) / ksplit;   // k-steps of the largest K-slice
    const size_t QS = (size_t)((per * 16 + 255) & ~255);
    const size_t a = 16 * 4 * QS + (pro == UMOE_PRO_RMSNORM ? (size_t)max_k * 2 : 0), red = (size_t)WV * NT * 64 * 16;
    return a > red ? a : red;
}

template <int NT, int U, int PRO, int EPI, int WV = 4, bool FR = false, bool PUB = false, bool BV = false>
static int launch_gemm(const umoe_gemm_args* a, hipStream_t s) {
    // the small decode launches (QKV, o_proj, codec head: one or two blocks per 4-wave workgroup, plain prologue) and the dense decode
    // down projection: by-value descriptors known at compile time when the host passed them (see wstream_body, BV)
    if constexpr (!BV && !FR && PRO == UMOE_PRO_PLAIN && ((NT <= 2 && WV == 4 && EPI != UMOE_EPI_SWIGLU) || (NT == 6 && WV == 8 && EPI == UMOE_EPI_BF16))) {
        if (a->groups_host && a->num_groups <= UMOE_GROUPS_INLINE) return launch_gemm<NT, U, PRO, EPI, WV, FR, PUB, true>(a, s);
    }
    const int ksplit = a->ksplit > 1 ? a->ksplit : 1;
    const size_t lds = gemm_lds_bytes(a->max_k, NT, WV, ksplit, PRO);
    UMOE_REQUIRE(lds <= 160 * 1024, "umoe_grouped_gemm: K=%d needs %zu bytes of LDS (> 160 KiB)", a->max_k, lds);
    static size_t configured = 0;  // per instantiation
    if (lds > configured) {
        UMOE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wstream_gemm<NT, U, PRO, EPI, WV, FR, PUB, BV>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
    }
    dim3 grid((unsigned)(ceil_div(a->max_n_blocks, NT) * ksplit), (unsigned)ceil_div(a->max_rows, 16), (unsigned)a->num_groups);
    umoe_group_pack gp;
    umoe_gemm_args b = *a;
    if (a->groups_host && a->num_groups <= UMOE_GROUPS_INLINE) {
        memcpy(gp.g, a->groups_host, sizeof(umoe_group_t) * a->num_groups);
    } else {
        b.groups_host = nullptr;
        memset(&gp, 0, sizeof(umoe_group_t));
    }
    umoe_router_args ra;
    memset(&ra, 0, sizeof(ra));
    int rider_mode = 0;
    if (FR) {
        ra = *a->fused_router;
        int dead = 0;                // workgroups of the box that have no tile (FR requires the host descriptors)
        for (int i = 0; i < a->num_groups; ++i) dead += (int)grid.x - ceil_div(a->groups_host[i].n_blocks, NT);
        const char* fv = getenv("UMOE_RIDER_MODE");      // (read per launch: the decode graph captures it once; tests toggle it)
        const int force = fv ? atoi(fv) : 0;
        rider_mode = (dead >= ra.S && ksplit == 1 && grid.y == 1 && force != 1) ? 2 : 1;
        if (rider_mode == 1) grid.z += 1;        // the router's workgroups: x = token, z = 0 (in front of the first group)
    }
    umoe_rider_pub pub;
    memset(&pub, 0, sizeof(pub));
    if (PUB) {
        pub = *reinterpret_cast<const umoe_rider_pub*>(a->rider_pub);
        UMOE_REQUIRE(pub.flags && pub.step && pub.err && ra.h_out == a->a && ra.norm_w && grid.y == 1 && a->max_rows <= 16 && a->groups_host,
                     "umoe_grouped_gemm: rider_pub needs flags / step / err, fused_router->h_out == a (with norm_w), <= 16 rows, host descriptors");
        for (int i = 0; i < a->num_groups; ++i)
            UMOE_REQUIRE(!a->groups_host[i].rows && !a->groups_host[i].count && !a->groups_host[i].row_off && a->groups_host[i].a_row_base == 0 &&
                             a->groups_host[i].a_col_off == 0 && a->groups_host[i].static_count == ra.S,
                         "umoe_grouped_gemm: rider_pub needs static groups over rows [0, S) of `a` (group %d)", i);
    }
    wstream_gemm<NT, U, PRO, EPI, WV, FR, PUB, BV><<<grid, WV * 64, lds, s>>>(b, gp, ra, rider_mode, pub);
    UMOE_LAUNCH_CHECK();
    return 0;
}

int umoe_gemm_riders(const umoe_gemm_args* a, int kind, const umoe_rider2* r2, const umoe_rider_pub* pub, hipStream_t s) {
    UMOE_REQUIRE(a && r2 && pub && pub->flags && pub->step && pub->err, "umoe_gemm_riders: null argument");
    // shapes the riders are written for; anything else: 1 = nothing launched (the caller issues the separate launches)
    if (!(a->groups_host && a->num_groups == 1 && a->max_rows <= 16 && a->max_rows == r2->n_riders && a->prologue == UMOE_PRO_PLAIN && a->ksplit <= 1 &&
          a->max_k % 512 == 0 && a->max_k <= 4096 && !a->fused_router && (a->lda & 7) == 0))
        return 1;
    const umoe_group_t& g = a->groups_host[0];
    if (g.rows || g.count || g.row_off || g.a_row_base || g.a_col_off || g.static_count != r2->n_riders) return 1;
    if (kind == 2 || kind == 4) {
        const umoe_combine_args& c = r2->cb;
        if (!(a->epilogue == UMOE_EPI_BF16 && c.D == 2048 && c.S == r2->n_riders && !c.slot_of && !c.y_parts && c.expert_mask && c.y_slots && c.y_shared &&
              c.global_w && c.moe_w && c.resid && c.out && c.norm_w && c.norm_out == a->a && a->lda == c.D && a->max_k == c.D && c.n_real >= 1 &&
              c.n_real <= UMOE_MAXE && c.n_fix >= 1 && c.n_fix <= 4 && c.dense_rows >= c.S && !c.ep_xfer))
            return 1;
        if (kind == 4 && !(r2->ep_region && r2->ep_round && r2->ep_err && r2->ep_n_cwg > 0 && r2->ep_size >= 2 && c.n_real % r2->ep_size == 0)) return 1;
    } else {
        return 1;
    }
    const size_t lds = gemm_lds_bytes(a->max_k, 1, 4, 1);
    umoe_group_pack gp;
    memset(&gp, 0, sizeof(gp));
    gp.g[0] = g;
    const dim3 grid((unsigned)(r2->n_riders + a->max_n_blocks), 1, 1);
    static size_t conf2 = 0, conf4 = 0;
    if (kind == 2) {
        if (lds > conf2) {
            UMOE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wstream_gemm_rk<UMOE_EPI_BF16, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            conf2 = lds;
        }
        wstream_gemm_rk<UMOE_EPI_BF16, 2><<<grid, 256, lds, s>>>(*a, gp, *pub, *r2);
    } else {
        if (lds > conf4) {
            UMOE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wstream_gemm_rk<UMOE_EPI_BF16, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            conf4 = lds;
        }
        wstream_gemm_rk<UMOE_EPI_BF16, 4><<<grid, 256, lds, s>>>(*a, gp, *pub, *r2);
    }
    UMOE_LAUNCH_CHECK();
    return 0;
}

int umoe_moe_fused(const umoe_gemm_args* gu, const umoe_gemm_args* dn, uint32_t* flags, int flag_words, hipStream_t s) {
    UMOE_REQUIRE(gu && dn && flags, "umoe_moe_fused: null argument");
    const int G = gu->num_groups;
    if (!(gu->fused_router && gu->rider_pub && gu->groups_host && dn->groups_host && G == dn->num_groups && G <= UMOE_GROUPS_INLINE && gu->nt == 14 &&
          dn->nt == 6 && gu->prologue == UMOE_PRO_PLAIN &&
          gu->epilogue == UMOE_EPI_SWIGLU && dn->prologue == UMOE_PRO_PLAIN &&
          dn->epilogue == UMOE_EPI_BF16 && gu->ksplit <= 1 && dn->ksplit <= 1 && gu->max_rows <= 16 && dn->max_rows <= 16 &&
          dn->a == gu->out && dn->lda == gu->ldo && !dn->fused_router && gu->max_k % 32 == 0 && dn->max_k % 32 == 0))
        return 1;
    const umoe_router_args* r = gu->fused_router;
    const int gx = ceil_div(gu->max_n_blocks, 14), per = ceil_div(dn->max_n_blocks, 6);
    int live = 0, dead = 0;
    for (int i = 0; i < G; ++i) {
        const umoe_group_t& a = gu->groups_host[i];
        const umoe_group_t& b = dn->groups_host[i];
        if (a.rows || a.count || a.row_off || a.a_row_base || a.a_col_off || a.static_count != r->S || b.rows || b.count || b.row_off || b.a_col_off ||
            b.static_count != r->S || b.bias || a.bias)
            return 1;
        live += ceil_div(a.n_blocks, 14);
        dead += gx - ceil_div(a.n_blocks, 14);
    }
    if (dead < r->S || live < G * per || G * gx > flag_words) return 1;
    if (!(r->S <= 16 && r->n_dyn == 9 && r->n_fix == 2 && (r->D == 2048 || r->D == 4096) && r->x && r->gate_w && r->expert_mask && !r->logits_in &&
          !r->norm_only && r->norm_w))
        return 1;
    // (measured and removed: the riders handing over only the rows' RMSNorm scales while the GEMM workgroups normalise the raw rows
    //  themselves -- 3.105-3.11 vs 3.075-3.087 ms/step; and no hand-off at all, every workgroup normalising in its prologue: 3.42 vs 3.36)
    if (!(r->h_out == gu->a)) return 1;
    umoe_fuse_x fx;
    memset(&fx, 0, sizeof(fx));
    fx.flags = flags;
    for (int i = 0; i < G; ++i) {      // the gate/up group whose output rows this down group reads
        const umoe_group_t& b = dn->groups_host[i];
        int j = -1;
        for (int t = 0; t < G; ++t)
            if (gu->groups_host[t].out_row_base == b.a_row_base) j = t;
        if (j < 0 || gu->groups_host[j].n_blocks * 8 != b.k || ceil_div(gu->groups_host[j].n_blocks, 14) > 64) return 1;
        fx.prod_base[i] = j * gx;
        fx.prod_n[i] = ceil_div(gu->groups_host[j].n_blocks, 14);
    }
    const size_t l1 = gemm_lds_bytes(gu->max_k, 14, 8, 1, gu->prologue), l2 = gemm_lds_bytes(dn->max_k, 6, 8, 1), lds = l1 > l2 ? l1 : l2;
    if (lds > 160 * 1024) return 1;
    static size_t configured = 0;
    if (lds > configured) {
        UMOE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&moe_fused_kernel<UMOE_PRO_PLAIN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
    }
    umoe_group_pack gg, gd;
    memset(&gg, 0, sizeof(gg));
    memset(&gd, 0, sizeof(gd));
    memcpy(gg.g, gu->groups_host, sizeof(umoe_group_t) * G);
    memcpy(gd.g, dn->groups_host, sizeof(umoe_group_t) * G);
    const umoe_rider_pub pub = *reinterpret_cast<const umoe_rider_pub*>(gu->rider_pub);
    UMOE_REQUIRE(pub.flags && pub.step && pub.err, "umoe_moe_fused: rider_pub needs flags / step / err");
    moe_fused_kernel<UMOE_PRO_PLAIN><<<dim3((unsigned)gx, 1, (unsigned)G), 512, lds, s>>>(*gu, gg, *dn, gd, *r, /*rider_mode*/ 2, pub, fx, per);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// 8 waves per workgroup when the staging tile leaves room for only one workgroup per CU (measured: +6 % on K=2752)
static bool use8(const umoe_gemm_args* a, int nt) {
    if (a->waves) return a->waves == 8;
    return gemm_lds_bytes(a->max_k, nt, 4, a->ksplit > 1 ? a->ksplit : 1) > 80 * 1024;
}

// n-blocks per workgroup: more blocks amortise the activation staging, fewer blocks give more workgroups
template <int PRO, int EPI>
static int launch_gemm_nt(const umoe_gemm_args* a, int nt, hipStream_t s) {
    // 8-wave / 2-step-chunk variants of the small tiles: the SAME K split as the 6-block down-projection launch of the dense
    // decode, so a product computed in a launch of its own (expert parallel: shared experts beside the exchange) is
    // bit-identical to the one computed inside the big launch
    if constexpr (PRO == UMOE_PRO_PLAIN && EPI == UMOE_EPI_BF16) {
        if (a->waves == 8 && nt == 1) return launch_gemm<1, 2, PRO, EPI, 8>(a, s);
        if (a->waves == 8 && nt == 2) return launch_gemm<2, 2, PRO, EPI, 8>(a, s);
        if (a->waves == 8 && nt == 4) return launch_gemm<4, 2, PRO, EPI, 8>(a, s);
    }
    switch (nt) {
        case 1: return launch_gemm<1, 16, PRO, EPI>(a, s);   // 16 k-steps per wave at K=2048: the whole stream is requested up front
        case 2: return launch_gemm<2, 8, PRO, EPI>(a, s);
        case 4: return launch_gemm<4, 4, PRO, EPI>(a, s);
        case 5: return use8(a, 5) ? launch_gemm<5, 3, PRO, EPI, 8>(a, s) : launch_gemm<5, 3, PRO, EPI>(a, s);
        case 6: return use8(a, 6) ? launch_gemm<6, 2, PRO, EPI, 8>(a, s) : launch_gemm<6, 2, PRO, EPI>(a, s);
        case 8: return use8(a, 8) ? launch_gemm<8, 2, PRO, EPI, 8>(a, s) : launch_gemm<8, 2, PRO, EPI>(a, s);
    }
    UMOE_REQUIRE(false, "umoe_grouped_gemm: nt must be 1, 2, 4, 5, 6 or 8 (SwiGLU: also 14) (got %d)", nt);
}

static int auto_nt(const umoe_gemm_args* a, bool swiglu) {
    if (a->nt) return a->nt;
    const long row_tiles = ceil_div(a->max_rows, 16);
    const long blocks = (long)a->max_n_blocks * a->num_groups * row_tiles;
    // measured on MI355X at 16 rows (scripts/kbench.py): gate/up 344-block groups and the 128-block down groups are
    // fastest at 8 blocks per workgroup (staging amortised 8x); small dense layers keep one block per workgroup
    if (swiglu) return blocks >= 1024 ? 8 : (blocks >= 512 ? 4 : 2);
    if (blocks >= 1024) return 8;
    if (blocks >= 512) return 2;
    return 1;
}

extern "C" int umoe_grouped_gemm(const umoe_gemm_args* a, umoe_stream_t stream) {
    UMOE_REQUIRE(a && (a->groups || (a->groups_host && a->num_groups <= UMOE_GROUPS_INLINE)) && a->a && a->out,
                 "umoe_grouped_gemm: null argument");
    UMOE_REQUIRE(a->num_groups > 0 && a->num_groups <= 65535, "umoe_grouped_gemm: bad num_groups %d", a->num_groups);
    UMOE_REQUIRE(a->max_k > 0 && a->max_k % 32 == 0, "umoe_grouped_gemm: K must be a multiple of 32 (got %d)", a->max_k);
    UMOE_REQUIRE(a->max_rows > 0 && a->max_n_blocks > 0, "umoe_grouped_gemm: empty problem");
    UMOE_REQUIRE(ceil_div(a->max_rows, 16) <= 65535, "umoe_grouped_gemm: too many rows (%d)", a->max_rows);
    UMOE_REQUIRE((a->lda & 7) == 0, "umoe_grouped_gemm: lda must be a multiple of 8 (16-byte rows)");
    UMOE_REQUIRE(a->ksplit <= 1 || (a->epilogue == UMOE_EPI_F32_RAW && a->prologue == UMOE_PRO_PLAIN && a->ksplit <= 4),
                 "umoe_grouped_gemm: ksplit > 1 needs the plain prologue and the raw fp32 partial-slab epilogue");
    UMOE_REQUIRE(!a->fused_router || (a->epilogue == UMOE_EPI_SWIGLU && a->nt == 14 &&
                                      (a->prologue == UMOE_PRO_PLAIN || (a->prologue == UMOE_PRO_RMSNORM && !a->rider_pub && a->max_k == 2048))),
                 "umoe_grouped_gemm: fused_router rides only in the SwiGLU launch with nt = 14 (RMSNorm prologue: K 2048, no rider_pub)");
    hipStream_t s = (hipStream_t)stream;
    const int pro = a->prologue, epi = a->epilogue;
    if (pro == UMOE_PRO_RMSNORM) {
        UMOE_REQUIRE(a->norm_w, "umoe_grouped_gemm: RMSNorm prologue needs norm_w");
        const int nt = auto_nt(a, false);
        if (epi == UMOE_EPI_BF16) return launch_gemm_nt<UMOE_PRO_RMSNORM, UMOE_EPI_BF16>(a, nt, s);
        if (epi == UMOE_EPI_F32) return launch_gemm_nt<UMOE_PRO_RMSNORM, UMOE_EPI_F32>(a, nt, s);
        if (epi == UMOE_EPI_SWIGLU) {  // shared experts straight from the residual stream (norm fused in the staging)
            UMOE_REQUIRE(a->max_n_blocks % 2 == 0, "umoe_grouped_gemm: SwiGLU needs gate/up block pairs");
            const int ns = auto_nt(a, true);
            if (ns <= 2) return launch_gemm<2, 8, UMOE_PRO_RMSNORM, UMOE_EPI_SWIGLU>(a, s);
            if (ns == 4) return launch_gemm<4, 4, UMOE_PRO_RMSNORM, UMOE_EPI_SWIGLU>(a, s);
            return launch_gemm<8, 2, UMOE_PRO_RMSNORM, UMOE_EPI_SWIGLU>(a, s);
        }
        UMOE_REQUIRE(false, "umoe_grouped_gemm: unsupported prologue/epilogue %d/%d", pro, epi);
    }
    UMOE_REQUIRE(pro == UMOE_PRO_PLAIN, "umoe_grouped_gemm: bad prologue %d", pro);
    switch (epi) {
        case UMOE_EPI_BF16: return launch_gemm_nt<UMOE_PRO_PLAIN, UMOE_EPI_BF16>(a, auto_nt(a, false), s);
        case UMOE_EPI_BF16_RESID:
            UMOE_REQUIRE(a->resid, "umoe_grouped_gemm: residual epilogue needs resid");
            return launch_gemm_nt<UMOE_PRO_PLAIN, UMOE_EPI_BF16_RESID>(a, auto_nt(a, false), s);
        case UMOE_EPI_SWIGLU: {
            UMOE_REQUIRE(a->max_n_blocks % 2 == 0, "umoe_grouped_gemm: SwiGLU needs gate/up block pairs");
            const int nt = auto_nt(a, true);
            UMOE_REQUIRE(nt >= 2, "umoe_grouped_gemm: SwiGLU needs nt >= 2");
            // (8 waves, 1-step chunks: the K split of the 14-block dense decode launch -- see launch_gemm_nt)
            if (nt == 2 && a->waves == 8) return launch_gemm<2, 1, UMOE_PRO_PLAIN, UMOE_EPI_SWIGLU, 8>(a, s);
            if (nt == 4 && a->waves == 8) return launch_gemm<4, 1, UMOE_PRO_PLAIN, UMOE_EPI_SWIGLU, 8>(a, s);
            if (nt == 2) return launch_gemm<2, 8, UMOE_PRO_PLAIN, UMOE_EPI_SWIGLU>(a, s);
            if (nt == 4) return launch_gemm<4, 4, UMOE_PRO_PLAIN, UMOE_EPI_SWIGLU>(a, s);
            if (nt == 6) return launch_gemm<6, 2, UMOE_PRO_PLAIN, UMOE_EPI_SWIGLU>(a, s);
            // 7 gate/up pairs per 8-wave workgroup: 226 workgroups for the dense decode shape = at most ONE per CU.  The
            // per-CU byte balance decides this kernel (scripts/kbench.py flat): NT 8 -> 387 workgroups, half of the CUs carry
            // two: 39.3 us; NT 12 -> 262 workgroups, six CUs carry two: 51.0 us; NT 14: 35.3-37.0 us; NT 16 (198 CUs): 43.0 us
            if (nt == 14) {
                if (a->fused_router) {
                    const umoe_router_args* r = a->fused_router;
                    UMOE_REQUIRE(a->max_rows <= 16 && r->S <= ceil_div(a->max_n_blocks, 14) && r->n_dyn == 9 && r->n_fix == 2 &&
                                     (r->D == 2048 || r->D == 4096) && r->x && r->gate_w && r->expert_mask && !r->logits_in && !r->norm_only &&
                                     a->groups_host && a->num_groups <= UMOE_GROUPS_INLINE,
                                 "umoe_grouped_gemm: fused_router needs <= 16 rows, n_dyn 9 / n_fix 2, D 2048 / 4096, S <= %d workgroups of the launch",
                                 ceil_div(a->max_n_blocks, 14));
                    if (a->rider_pub) return launch_gemm<14, 1, UMOE_PRO_PLAIN, UMOE_EPI_SWIGLU, 8, true, true>(a, s);
                    return launch_gemm<14, 1, UMOE_PRO_PLAIN, UMOE_EPI_SWIGLU, 8, true>(a, s);
                }
                return launch_gemm<14, 1, UMOE_PRO_PLAIN, UMOE_EPI_SWIGLU, 8>(a, s);
            }
            return use8(a, 8) ? launch_gemm<8, 2, UMOE_PRO_PLAIN, UMOE_EPI_SWIGLU, 8>(a, s)
                                 : launch_gemm<8, 2, UMOE_PRO_PLAIN, UMOE_EPI_SWIGLU>(a, s);
        }
        case UMOE_EPI_F32: return launch_gemm_nt<UMOE_PRO_PLAIN, UMOE_EPI_F32>(a, auto_nt(a, false), s);
        case UMOE_EPI_F32_RAW: return launch_gemm_nt<UMOE_PRO_PLAIN, UMOE_EPI_F32_RAW>(a, auto_nt(a, false), s);
    }
    UMOE_REQUIRE(false, "umoe_grouped_gemm: bad epilogue %d", epi);
}

extern "C" int umoe_grouped_swiglu_fwd(const umoe_group_t* gateup_groups, const umoe_group_t* down_groups,
                                       int num_groups, int max_rows, const uint16_t* x, int D, int I, uint16_t* h_ws,
                                       uint16_t* y_slots, umoe_stream_t stream) {
    umoe_gemm_args a{};
    a.groups = gateup_groups; a.num_groups = num_groups; a.max_rows = max_rows; a.max_n_blocks = 2 * I / 16;
    a.max_k = D; a.a = x; a.lda = D; a.out = h_ws; a.ldo = I; a.n_valid = I;
    a.prologue = UMOE_PRO_PLAIN; a.epilogue = UMOE_EPI_SWIGLU;
    int rc = umoe_grouped_gemm(&a, stream);
    if (rc) return rc;
    umoe_gemm_args b{};
    b.groups = down_groups; b.num_groups = num_groups; b.max_rows = max_rows; b.max_n_blocks = D / 16;
    b.max_k = I; b.a = h_ws; b.lda = I; b.out = y_slots; b.ldo = D; b.n_valid = D;
    b.prologue = UMOE_PRO_PLAIN; b.epilogue = UMOE_EPI_BF16;
    return umoe_grouped_gemm(&b, stream);
}

// shared experts (reference core.py:344-351 + :16-31): dense SwiGLU over all S rows for n_fix experts; the scaling by
// global_weight[:, n_dyn + i] and the accumulation happen in umoe_unpermute_combine_fwd with the reference's roundings.
extern "C" int umoe_shared_swiglu_fwd(const umoe_group_t* gateup_groups, const umoe_group_t* down_groups, int n_fix, int S,
                                      const uint16_t* x, int D, int I, uint16_t* h_ws, uint16_t* y_shared,
                                      umoe_stream_t stream) {
    return umoe_grouped_swiglu_fwd(gateup_groups, down_groups, n_fix, S, x, D, I, h_ws, y_shared, stream);
}
