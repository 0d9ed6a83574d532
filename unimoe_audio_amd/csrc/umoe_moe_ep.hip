// Expert-parallel decode layer, the MoE half in ONE launch of ONE workgroup per CU (decode engine only; reference: AudioMOELayer.forward,
// utils/UniMoE_Audio_core.py:446-493 -- the two all-to-alls :467,480 around the local experts :406-416 -- and the shared experts
// :344-351).  Replaces the eight launches of run_moe_ep (RMSNorm + push | shared gate/up | pull + re-lay | local gate/up | local down |
// push | shared down, and the combine's own polling launch): the exchange sits INSIDE the GEMM launch, as rider workgroups and as
// the down projection's epilogue, and the combine rides in the next layer's QKV launch as at ep_size 1 (umoe_riders_dev.h).
//
// Every workgroup walks a short TASK LIST the host computes once per engine (epf_plan); a task is a slice of one of four phases:
//   A  shared experts' gate/up SwiGLU on the OWN rows (flat_gateup: the workgroup normalises the raw rows itself)  -- hides the flight
//      of the dispatch
//   B  local experts' gate/up SwiGLU over the rows of EVERY rank (MT = ep_size tiles of 16 rows, each weight byte streamed once; the
//      tiles arrive in MFMA operand order from the tile riders below; arithmetic of umoe_gemm_mt.hip = of the ep_size 1 launch)
//   C  local experts' down projection over the same tiles; the epilogue stores each tile's rows STRAIGHT into the return slab of the
//      rank that owns them (system-scope write-through stores over xGMI) and counts the workgroup in on that rank
//   D  shared experts' down projection (flat_down)                                                     -- hides the return flight
// and the riders: workgroup j < ep_size pushes the raw own rows x1 to rank j's dispatch slab, then waits for rank j's rows in its own
// slab, normalises them (post-attention RMSNorm, model.py:240; the summation tree of the router body: the same bits as every other
// launch that makes these rows) and re-lays them into operand order for phase B; the next S workgroups run the Top-P router of one
// own row each (umoe_router_dev.h; its tables feed the combine only).  Riders take their GEMM tasks afterwards like everybody else.
//
// Hand-offs (cdna_hip_programming.md Guideline 16 R1; umoe_common.h "Peer exchange"): payload by write-through stores (agent scope
// inside the GPU, system scope to a peer), every storing wave drains, one word per part: an epoch FLAG where one producer writes
// (dispatch rows: one per source rank; operand-order tiles; a workgroup's phase A / B slices) and a COUNTER where many do (the
// return slab: every phase C workgroup adds 1 on every destination; the reader waits for rounds x workgroups).  Every poll is bounded
// and keeps the first error cause in the sticky word (1 a peer's rows / counters, 3 the shared experts' rows, 4 the operand-order tiles, 5 a
// local expert's h tiles did not arrive).  Bit-identical to ep_size 1: every (expert, row tile, 16-feature block) product
// keeps the K split over 8 waves and the fixed-order reduction of moe_flat_kernel / moe_fused_kernel.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "umoe_common.h"
#include "umoe_router_dev.h"
#include "umoe_flat_dev.h"

#define EPF_RIDER_LDS 512
typedef uint32_t epf_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t epf_u32x2 __attribute__((ext_vector_type(2)));

struct epf_args {
    flat_args S;                        // the shared experts' slices (groups 0 .. n_fix-1, own rows): phases A and D
    // local experts (phases B, C)
    const uint16_t* w_lgu[UMOE_MT_MAXG];
    const uint16_t* w_ldn[UMOE_MT_MAXG];
    uint16_t* xgp;                      // [R][16 * D] normalised row tiles in MFMA operand order (written by the tile riders)
    uint16_t* hpk;                      // [E_loc][R][16 * I] silu(g)*u tiles in operand order (phase B -> phase C)
    int D, I, E_loc, R, rank, loopback;
    char* peer_base[UMOE_MAX_EP];       // exchange regions (umoe_common.h)
    size_t disp_off, ret_off;           // slabs inside a region: dispatch [R][S][D] raw rows x1, return [n_real][16][D]
    uint32_t* flag_b;                   // [n_wg] phase B published (agent scope)
    uint32_t* tile_ready;               // [UMOE_FLAG_REPL][16] operand-order tile j is in xgp
    int prodb_base[UMOE_MT_MAXG], prodb_n[UMOE_MT_MAXG];      // workgroups that produce local expert g's h tiles
    const uint32_t* round;              // device word: expert-parallel rounds base (decode steps taken; no prefill bumps)
    const uint32_t* tasks;              // [n_wg][UMOE_EPF_MAXT]
};

static_assert(sizeof(epf_args) + sizeof(umoe_router_args) + sizeof(umoe_rider_pub) + 16 <= 4096, "moe_ep_kernel: kernel arguments exceed 4 KiB");

// ---- tile rider: push the own raw rows to rank j (REAL exchange: j != rank; loopback: into the own slab, tile j) and / or make tile j.
// With enough workgroups the two halves are DIFFERENT workgroups (a rank's push does not depend on its peers' rows: the re-lay workgroup
// polls from the first microsecond on); with few (ranks sharing a card) one workgroup does both, push first. ----
__device__ __forceinline__ void epf_tile_rider(const epf_args& P, const umoe_rider_pub& pub, const int j, char* smem, const int tid, const bool do_push,
                                               const bool do_pack) {
    constexpr int KB = 64, TPR = 32;
    const int m = tid / TPR, sub = tid % TPR;
    const bool valid = m < P.S.S;
    const uint32_t epoch = flat_epoch(pub);
    const size_t row_bytes = (size_t)P.D * 2, tile_bytes = (size_t)P.S.S * row_bytes;
    const bool own = !P.loopback && j == P.rank;
    const uint16_t* xsrc = P.S.a + (size_t)(valid ? m : 0) * P.S.lda;
    uint4 buf[8];
    if (!own && do_push) {
        // my rows -> tile (loopback: j, else my rank) of rank j's dispatch slab; one flag word per source tile
        const int dt = P.loopback ? j : P.rank;
        char* dst = P.peer_base[j] + P.disp_off + (size_t)dt * tile_bytes;
        const auto drs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)tile_bytes, 0x00020000);
#pragma unroll
        for (int n = 0; n < 8; ++n) buf[n] = ld16(xsrc + ((n >> 1) * KB + sub + TPR * (n & 1)) * 8);
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const epf_u32x4 v4 = {buf[n].x, buf[n].y, buf[n].z, buf[n].w};
            if (valid) __builtin_amdgcn_raw_buffer_store_b128(v4, drs, (int)((size_t)m * row_bytes + (size_t)((n >> 1) * KB + sub + TPR * (n & 1)) * 16), 0, UMOE_SYS_AUX);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(umoe_ep_flag(P.peer_base[j], 0, dt, 0), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (!do_pack) return;
    if (!own) {
        // rank j's rows (loopback: what this rank's push stored) arrive in MY slab, tile j
        if (tid == 0) umoe_ep_wait(umoe_ep_flag(P.peer_base[P.rank], 0, j, 0), epoch, pub.err);
        __syncthreads();
        char* src = P.peer_base[P.rank] + P.disp_off + (size_t)j * tile_bytes;
        const auto srs = __builtin_amdgcn_make_buffer_rsrc(src, 0, (int)tile_bytes, 0x00020000);
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const epf_u32x4 t4 = __builtin_amdgcn_raw_buffer_load_b128(srs, (int)((size_t)(valid ? m : 0) * row_bytes + (size_t)((n >> 1) * KB + sub + TPR * (n & 1)) * 16), 0, UMOE_SYS_AUX);
            buf[n] = make_uint4(t4[0], t4[1], t4[2], t4[3]);
        }
    } else {
#pragma unroll
        for (int n = 0; n < 8; ++n) buf[n] = ld16(xsrc + ((n >> 1) * KB + sub + TPR * (n & 1)) * 8);
    }
    // normalise (arithmetic of flat_gateup's staging = of router4_body) and store in operand order: chunk (quarter h, step i) of row m
    // is lane h*16 + m of fragment i
    char* nw_lds = smem;
    const uint4 nw1 = ld16(P.S.norm_w + (tid & (4 * KB - 1)) * 8);
    st16(nw_lds + (tid & (4 * KB - 1)) * 16, nw1);
    float q4[4];
#pragma unroll
    for (int hq = 0; hq < 4; ++hq) {
        float c2[2];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            float f[8];
            unpack8(buf[hq * 2 + k2], f);
            float cs = 0.f;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) cs += f[jj] * f[jj];
            c2[k2] = cs;
        }
        float v = c2[0] + c2[1];
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        q4[hq] = v;
    }
    const float ss = ((q4[0] + q4[1]) + q4[2]) + q4[3];
    const float rs = rsqrtf(ss / (float)(KB * 32) + P.S.rms_eps);
    __syncthreads();
    const auto ors = __builtin_amdgcn_make_buffer_rsrc(P.xgp + (size_t)j * 16 * P.D, 0, 16 * P.D * 2, 0x00020000);
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int h = n >> 1, i = sub + TPR * (n & 1);
        float f[8], w[8];
        unpack8(buf[n], f);
        unpack8(*reinterpret_cast<const uint4*>(nw_lds + (h * KB + i) * 16), w);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) f[jj] = w[jj] * rbf(f[jj] * rs);
        const uint4 pk = pack8(f);
        const epf_u32x4 v4 = {pk.x, pk.y, pk.z, pk.w};
        if (valid) __builtin_amdgcn_raw_buffer_store_b128(v4, ors, (i * 64 + h * 16 + m) * 16, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < UMOE_FLAG_REPL)
        __hip_atomic_store(reinterpret_cast<umoe_gu32*>(reinterpret_cast<uintptr_t>(P.tile_ready + tid * 16 + j)), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- phases B / C: NT weight blocks x MT row tiles per pass (umoe_gemm_mt.hip wstream_mt, fragments by sc1 loads: they were written
// inside this launch); K split over the 8 waves in whole U-step chunks, fixed-order LDS reduction ----
template <int NT, int MT, int U, int RW, int RB, bool SWIGLU>
__device__ __forceinline__ void epf_mt(const epf_args& P, const int grp, const int nb0, const int KB, const uint16_t* wbase, const uint16_t* fbase,
                                       const int ftile0, const int fbytes, char* smem, const int tid, const umoe_rider_pub& pub, uint32_t* wflag,
                                       const int nwait, const uint32_t wcode, const int tile0) {      // tile0: first of this pass's MT row tiles
    constexpr int WV = 8;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const epf_u32x4* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wp[t] = reinterpret_cast<const epf_u32x4*>(wbase) + ((size_t)(nb0 + t) * KB) * 64 + lane;
    const auto frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(fbase), 0, fbytes, 0x00020000);
    f32x4_t acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    epf_u32x4 wr[RW][NT], br[RB][MT];
    auto load_w = [&](epf_u32x4 (&d)[NT], int ii) {
#pragma unroll
        for (int t = 0; t < NT; ++t) d[t] = __builtin_nontemporal_load(wp[t] + (size_t)ii * 64);
    };
    auto load_b = [&](epf_u32x4 (&d)[MT], int ii) {
#pragma unroll
        for (int m = 0; m < MT; ++m) d[m] = __builtin_amdgcn_raw_buffer_load_b128(frs, (((ftile0 + m) * KB + ii) * 64 + lane) * 16, 0, 16);
    };
    const int il = i1 - 1;
    // the weights first: they depend on nobody.  Then (first pass of a phase only: nwait > 0) the hand-off this pass's fragments hang on --
    // the operand-order row tiles (phase B) or the h tiles of this expert's producers (phase C) -- lane i of wave 0 polls part i, bounded
#pragma unroll
    for (int r = 0; r < RW; ++r) load_w(wr[r], min(i0 + r, il));
    __builtin_amdgcn_sched_barrier(0);
    if (nwait > 0) {
        if (tid < nwait) flat_wait(wflag + tid, flat_epoch(pub), pub.err, wcode);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) load_b(br[r], min(i0 + r, il));
    __builtin_amdgcn_sched_barrier(0);
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
            load_w(wr[r], min(ii + RW, il));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
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
    const int h = lane >> 4, mm = lane & 15;
    if constexpr (SWIGLU) {
        // silu(g) * u of tile (grp, m) in operand order of the [16][I] matrix the down projection contracts over (write-through)
        const int I = P.I, Q = I >> 2;
        const auto hrs = __builtin_amdgcn_make_buffer_rsrc(P.hpk, 0, P.E_loc * P.R * 16 * I * 2, 0x00020000);
        for (int q = wave; q < MT * (NT / 2); q += WV) {
            const int m = q / (NT / 2), pq = q % (NT / 2);
            const f32x4_t ga = reduced(m, 2 * pq), ua = reduced(m, 2 * pq + 1);
            uint16_t y[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gt = rbf(ga[j]);
                const float up = rbf(ua[j]);
                const float si = rbf(gt / (1.0f + expf(-gt)));
                y[j] = f2bf(si * up);
            }
            const int f = (nb0 / 2 + pq) * 16 + 4 * h;
            const int qq = f / Q, r = f % Q;
            const long eo = (long)(grp * P.R + tile0 + m) * 16 * I + ((long)(r >> 3) * 64 + qq * 16 + mm) * 8 + (r & 7);
            const epf_u32x2 v2 = {(uint32_t)y[0] | ((uint32_t)y[1] << 16), (uint32_t)y[2] | ((uint32_t)y[3] << 16)};
            __builtin_amdgcn_raw_buffer_store_b64(v2, hrs, (int)(eo * 2), 0, 16);
        }
    } else {
        // rows of tile m belong to rank m: straight into that rank's return slab, slot (my rank, local expert grp)
        for (int q = wave; q < MT * NT; q += WV) {
            const int m = q / NT, t = q % NT;
            const f32x4_t a4 = reduced(m, t);
            const int n = (nb0 + t) * 16 + 4 * h;
            uint16_t y[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = f2bf(rbf(a4[j] + 0.f));
            const int gt = tile0 + m;      // the rows of tile gt belong to rank gt
            const int dr = P.loopback ? P.rank : gt, sr = P.loopback ? gt : P.rank;
            char* dst = P.peer_base[dr] + P.ret_off + ((size_t)(sr * P.E_loc + grp) * P.S.S) * (size_t)P.D * 2;      // slab [n_real][S][D]
            const auto yrs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, P.S.S * P.D * 2, 0x00020000);
            const epf_u32x2 v2 = {(uint32_t)y[0] | ((uint32_t)y[1] << 16), (uint32_t)y[2] | ((uint32_t)y[3] << 16)};
            if (mm < P.S.S) __builtin_amdgcn_raw_buffer_store_b64(v2, yrs, (mm * P.D + n) * 2, 0, UMOE_SYS_AUX);
        }
    }
}

// task word: kind << 28 | group << 24 | first unit << 8 | units
#define EPF_END 0u
#define EPF_A 1u          // shared gate/up: `first` = flat pair index over the shared groups, units = pairs (1..7)
#define EPF_PUB_A 2u
#define EPF_B 3u          // local gate/up: group = local expert | tile half << 2 (8 tiles: a pass takes 4), first = pair, units = pairs (NT / 2)
#define EPF_PUB_B 4u
#define EPF_C 5u          // local down: group = local expert | tile half << 2 (8 tiles: a pass takes 4), first = block, units = blocks (1..4)
#define EPF_SIG_C 6u
#define EPF_D 7u          // shared down: group = shared expert, first = block, units = blocks (1..10)
#define EPF_TILE 8u       // rider: group = tile / peer j, units = 1 push | 2 re-lay (3: both, push first)
#define EPF_ROUTER 9u     // rider: first = token

template <int MT>
__global__ __launch_bounds__(512, 1) void moe_ep_kernel(const epf_args P, const umoe_router_args ra, const umoe_rider_pub pub, const int lds_gemm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned b = blockIdx.x;
    const uint32_t* tl = P.tasks + (size_t)b * UMOE_EPF_MAXT;
    flat_stamps st;
#ifdef UMOE_TIMELINE
    // diagnostics (instrumented build only, scripts/ep_timeline.py): stamp 0 = entry, stamp k + 1 = end of task k, 15 = exit
    unsigned long long tstamp[16];
    if (P.S.dbg) {
#pragma unroll
        for (int k = 0; k < 16; ++k) tstamp[k] = 0;
        tstamp[0] = wall_clock64();
    }
#endif
    bool tiles_seen = false;
    unsigned seam_seen = 0u;
    for (int ti = 0; ti < UMOE_EPF_MAXT; ++ti) {
        // an OPAQUE copy of the thread index per task: with the plain threadIdx.x every per-lane address of every task kind is
        // loop-invariant, hipcc hoists them all in front of the loop and spills ~170 registers at kernel entry
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const uint32_t tw = __builtin_amdgcn_readfirstlane(tl[ti]);
        const unsigned kind = tw >> 28;
        if (kind == EPF_END) break;
        const int grp = (int)((tw >> 24) & 15u), first = (int)((tw >> 8) & 0xffffu), n = (int)(tw & 255u);
        if (ti) __syncthreads();        // (the reduction slab of the previous task is the staging area of this one)
        if (kind == EPF_TILE) {
            epf_tile_rider(P, pub, grp, smem, tid, (n & 1) != 0, (n & 2) != 0);      // units: bit 0 push, bit 1 re-lay
        } else if (kind == EPF_ROUTER) {
            float* rl = reinterpret_cast<float*>(smem + lds_gemm);
            if (tid < 256) {
#ifdef UMOE_TIMELINE
                TL_ENTER(5);
#endif
                if (ra.logits_bf16) router4_body<9, 2, 1, false>(ra, first, tid, rl TL_PASS, nullptr, 0u, nullptr);
                else router4_body<9, 2, 0, false>(ra, first, tid, rl TL_PASS, nullptr, 0u, nullptr);
            } else {
                __syncthreads();
                __syncthreads();
            }
        } else if (kind == EPF_A) {
            switch (n) {
                case 1: flat_gateup<1, false>(P.S, pub, first, b, smem, st, tid); break;
                case 2: flat_gateup<2, false>(P.S, pub, first, b, smem, st, tid); break;
                case 3: flat_gateup<3, false>(P.S, pub, first, b, smem, st, tid); break;
                case 4: flat_gateup<4, false>(P.S, pub, first, b, smem, st, tid); break;
                case 5: flat_gateup<5, false>(P.S, pub, first, b, smem, st, tid); break;
                case 6: flat_gateup<6, false>(P.S, pub, first, b, smem, st, tid); break;
                default: flat_gateup<7, false>(P.S, pub, first, b, smem, st, tid); break;
            }
        } else if (kind == EPF_PUB_A || kind == EPF_PUB_B) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0)      // (`first` = this workgroup's position among the producers of its role)
                __hip_atomic_store(reinterpret_cast<umoe_gu32*>(reinterpret_cast<uintptr_t>((kind == EPF_PUB_A ? P.S.flags : P.flag_b) + first)), flat_epoch(pub),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (kind == EPF_B) {
            // the operand-order tiles of every rank: waited for INSIDE the first pass, behind its weight requests (replicated flag lines)
            uint32_t* wf = P.tile_ready + (b % UMOE_FLAG_REPL) * 16;
            const int nw = tiles_seen ? 0 : MT;      // (all MT tile flags, whichever tiles this pass takes: one wait per workgroup)
            tiles_seen = true;
            const int ex = grp & 3, tile0 = (grp >> 2) * 4;      // (task group field: local expert | tile half << 2)
            const uint16_t* w = P.w_lgu[ex];
            const int fb = MT * 16 * P.D * 2;
            // one pass = NT weight blocks x MTB row tiles with NT * MTB <= 16 accumulator tiles.  At 8 row tiles a pass takes HALF of them
            // and two pairs instead of all eight and one: 512 KiB of intake per pass instead of 640 (a pass's time IS its intake, ~50 GB/s
            // per CU of L2-resident fragments), the weights of a pair are then read by two workgroups (the second from L2 / MALL)
            constexpr int MTB = MT >= 8 ? 4 : MT;
            if constexpr (MTB == 2) {
                switch (n) {
                    case 4: epf_mt<8, 2, 1, 4, 2, true>(P, ex, 2 * first, P.D >> 5, w, P.xgp, tile0, fb, smem, tid, pub, wf, nw, 4u, tile0); break;
                    case 3: epf_mt<6, 2, 1, 4, 2, true>(P, ex, 2 * first, P.D >> 5, w, P.xgp, tile0, fb, smem, tid, pub, wf, nw, 4u, tile0); break;
                    case 2: epf_mt<4, 2, 1, 8, 2, true>(P, ex, 2 * first, P.D >> 5, w, P.xgp, tile0, fb, smem, tid, pub, wf, nw, 4u, tile0); break;
                    default: epf_mt<2, 2, 1, 8, 2, true>(P, ex, 2 * first, P.D >> 5, w, P.xgp, tile0, fb, smem, tid, pub, wf, nw, 4u, tile0); break;
                }
            } else {
                // (rings: 4 k-steps of 4 weight blocks + 4 k-steps of 4 fragments = 128 VGPRs beside the 64 of the accumulators)
                if (n >= 2) epf_mt<4, 4, 1, 4, 4, true>(P, ex, 2 * first, P.D >> 5, w, P.xgp, tile0, fb, smem, tid, pub, wf, nw, 4u, tile0);
                else epf_mt<2, 4, 1, 8, 4, true>(P, ex, 2 * first, P.D >> 5, w, P.xgp, tile0, fb, smem, tid, pub, wf, nw, 4u, tile0);
            }
        } else if (kind == EPF_C) {
            // the workgroups that produced this expert's h tiles: waited for inside the first pass on this expert, behind its weight requests
            const int ex = grp & 3, tile0 = (grp >> 2) * 4;      // (task group field: local expert | tile half << 2)
            uint32_t* wf = P.flag_b + P.prodb_base[ex];
            const int nw = ((seam_seen >> ex) & 1u) ? 0 : P.prodb_n[ex];
            seam_seen |= 1u << ex;
            const uint16_t* w = P.w_ldn[ex];
            const int KB = P.I >> 5, fb = P.E_loc * MT * 16 * P.I * 2;
            // one pass = 1..4 blocks x MTC tiles (at 8 tiles a pass takes HALF of them: twice the tasks, so every workgroup of the launch has
            // one, and four k-steps of fragments fit the ring).  Rings: RW k-steps of NT weight blocks (<= 64 VGPRs), 4 k-steps of fragments
            constexpr int MTC = MT >= 8 ? 4 : MT;
            const int ft0 = ex * MT + tile0;
            switch (n) {
                case 1: epf_mt<1, MTC, 2, 8, 4, false>(P, ex, first, KB, w, P.hpk, ft0, fb, smem, tid, pub, wf, nw, 5u, tile0); break;
                case 2: epf_mt<2, MTC, 2, 8, 4, false>(P, ex, first, KB, w, P.hpk, ft0, fb, smem, tid, pub, wf, nw, 5u, tile0); break;
                case 3: epf_mt<3, MTC, 2, 4, 4, false>(P, ex, first, KB, w, P.hpk, ft0, fb, smem, tid, pub, wf, nw, 5u, tile0); break;
                default: epf_mt<4, MTC, 2, 4, 4, false>(P, ex, first, KB, w, P.hpk, ft0, fb, smem, tid, pub, wf, nw, 5u, tile0); break;
            }
        } else if (kind == EPF_SIG_C) {
            // every storing wave drains its system-scope stores, the workgroup meets, lane t counts this workgroup in on the owner of tile t
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid < MT) {
                const int dr = P.loopback ? P.rank : tid, sr = P.loopback ? tid : P.rank;
                __hip_atomic_fetch_add(umoe_ep_flag(P.peer_base[dr], 1, sr, 0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        } else if (kind == EPF_D) {
            switch (n) {
                case 1: flat_down<1, 1>(P.S, pub, grp, first, smem, st, 7, tid); break;
                case 2: flat_down<2, 1>(P.S, pub, grp, first, smem, st, 7, tid); break;
                case 3: flat_down<3, 1>(P.S, pub, grp, first, smem, st, 7, tid); break;
                case 4: flat_down<4, 1>(P.S, pub, grp, first, smem, st, 7, tid); break;
                case 5: flat_down<5, 1>(P.S, pub, grp, first, smem, st, 7, tid); break;
                case 6: flat_down<6, 1>(P.S, pub, grp, first, smem, st, 7, tid); break;
                case 7: flat_down<7, 1>(P.S, pub, grp, first, smem, st, 7, tid); break;
                case 8: flat_down<8, 1>(P.S, pub, grp, first, smem, st, 7, tid); break;
                case 9: flat_down<9, 1>(P.S, pub, grp, first, smem, st, 7, tid); break;
                default: flat_down<10, 1>(P.S, pub, grp, first, smem, st, 7, tid); break;
            }
        }
#ifdef UMOE_TIMELINE
        if (P.S.dbg) {
#pragma unroll
            for (int k = 1; k < 15; ++k)
                if (k == ti + 1) tstamp[k] = wall_clock64();
        }
#endif
    }
#ifdef UMOE_TIMELINE
    if (P.S.dbg && threadIdx.x == 0) {
        tstamp[15] = wall_clock64();
#pragma unroll
        for (int k = 0; k < 16; ++k) P.S.dbg[(size_t)b * 16 + k] = tstamp[k];
    }
#endif
}

// ------------------------------------------------------------------------------------ host: the task lists
// At the decode shape a phase of one workgroup is a LATENCY chain (stage / wait, first round trip, 8-11 k-steps of fragments, reduction,
// epilogue: 8-12 us whatever the slice), so a workgroup that walked all four phases took ~57 us for 67 MB at ep 8.  The lists keep the
// chains short and side by side (scripts/ep_timeline.py):
//   * phase A (shared gate/up, own rows, nobody to wait for) on EVERY workgroup that is not a rider: it fills the ~12 us the tile
//     riders need to push / re-lay the rows, which is when phase B can start at the earliest;
//   * then two roles: LOCAL workgroups take B (local gate/up over every rank's rows), SHARED workgroups -- the router riders among them
//     -- take D (shared down: it only needs phase A); the split follows a small intake model (HBM ~26 GB/s, L2-resident fragments
//     ~100 GB/s per CU; UMOE_EPF_NS overrides it);
//   * phase C (local down + return stores) on EVERY workgroup, one pass each where the units allow it: it hangs on ALL of phase B, so
//     its own duration sits fully on the critical path.
// Workgroup order: [0, R) re-lay riders and [R, 2R) push riders (dispatched first: everything hangs on them; a rank's push does not
// depend on its peers' rows, so the re-lay workgroups poll from the first microsecond on), [2R, 2R + n_s) shared role, its first S also
// route one own row each, then the local role.  With few workgroups (ranks sharing a card in the tests) every workgroup takes a share
// of every phase instead: rider | A | B | C | D.
struct EpfPlan {
    bool ok = false;
    int n_wg = 0, n_cwg = 0, n_s = 0;
    int proda_base[4], proda_n[4], prodb_base[UMOE_MT_MAXG], prodb_n[UMOE_MT_MAXG];
    std::vector<uint32_t> tasks;
};

static inline uint32_t epf_task(unsigned kind, int grp, int first, int n) { return (kind << 28) | ((uint32_t)grp << 24) | ((uint32_t)first << 8) | (uint32_t)n; }

// deals `units` over the workgroups wgs[0 .. nw) in equal shares, the remainder starting at position `rot`
static void epf_deal(int units, const std::vector<int>& wgs, int rot, std::vector<int>& first, std::vector<int>& count) {
    const int nw = (int)wgs.size();
    const int base = units / nw, extra = units % nw;
    std::vector<int> cnt(nw, base);
    for (int k = 0; k < extra; ++k) cnt[(rot + k) % nw] += 1;
    int acc = 0;
    for (int k = 0; k < nw; ++k) {
        first[wgs[k]] = acc;
        count[wgs[k]] = cnt[k];
        acc += cnt[k];
    }
}

static void epf_plan(int n_wg, int R, int E_loc, int S, int D, int I_dyn, int I_sh, int n_fix, EpfPlan& pl) {
    pl.ok = false;
    pl.n_wg = n_wg;
    const bool roles = n_wg >= 128;
    const int RT = roles ? 2 * R : R;            // tile riders: re-lay workgroups [0, R) and, with roles, push workgroups [R, 2R)
    const int riders = RT + S;
    if (n_wg < riders + 1 || n_wg > 256 || E_loc < 1 || E_loc > UMOE_MT_MAXG || n_fix < 1 || n_fix > 4 || !(R == 2 || R == 4 || R == 8) || D != 2048 ||
        I_dyn % 32 || I_sh % 32 || (I_sh / 32) % 2 == 0)       // (the shared experts' down slices run on flat_down<., 1>: odd k-steps)
        return;
    // ---- who takes which phase
    std::vector<int> wa, wb, wc, wd;      // workgroups of phases A, B, C, D in the order their units are dealt
    int n_s = 0;
    if (roles) {
        const double hbm = 0.0385, l2 = 0.01;      // us per KiB of intake of one CU
        const double dcost = n_fix * (D / 16) * (double)(I_sh / 32) * hbm;      // shared down
        const double bcost = E_loc * (I_dyn / 16) * (4.0 * (D / 32) * hbm + R * 2.0 * (D / 32) * l2);
        n_s = (int)(n_wg * dcost / (dcost + bcost) + 0.5);
        n_s = std::max(n_s, n_wg - (E_loc * (I_dyn / 16) + 1) / 2 * (R >= 8 ? 2 : 1));      // (no more B workgroups than there are full passes at 8 tiles)
        n_s = std::max(n_s, (n_fix * (D / 16) + 3) / 4);       // (<= 4 down blocks per shared workgroup: its chain must end before phase C starts)
        if (const char* v = getenv("UMOE_EPF_NS")) n_s = atoi(v);
        n_s = std::max(n_s, S);
        n_s = std::min(n_s, n_wg - RT - 1);
        for (int w = RT + S; w < n_wg; ++w) wa.push_back(w);                // everybody but the re-lay and the router riders ...
        for (int w = R; w < RT; ++w) wa.push_back(w);                        // ... the push riders last in the deal (they lose ~3 us first)
        for (int w = RT + n_s; w < n_wg; ++w) wb.push_back(w);
        for (int w = 0; w < RT; ++w) wb.push_back(w);                        // the tile riders last in the deal: remainder units go to the others first
        for (int w = RT + S; w < RT + n_s; ++w) wd.push_back(w);             // the router riders last in the deal, too
        for (int w = RT; w < RT + S; ++w) wd.push_back(w);
        for (int w = RT; w < n_wg; ++w) wc.push_back(w);
        for (int w = 0; w < RT; ++w) wc.push_back(w);
    } else {
        for (int w = 0; w < n_wg; ++w) { wa.push_back(w); wb.push_back(w); wc.push_back(w); wd.push_back(w); }
    }
    pl.n_s = n_s;
    std::vector<std::vector<uint32_t>> la(n_wg), lb(n_wg), lc(n_wg), ld(n_wg);
    std::vector<int> first(n_wg, 0), count(n_wg, 0);
    auto clear = [&]() { std::fill(first.begin(), first.end(), 0); std::fill(count.begin(), count.end(), 0); };
    // position of a workgroup among the producers of a phase (its word in that phase's flag array)
    std::vector<int> pos_a(n_wg, -1), pos_b(n_wg, -1);
    for (size_t k = 0; k < wa.size(); ++k) pos_a[wa[k]] = (int)k;
    for (size_t k = 0; k < wb.size(); ++k) pos_b[wb[k]] = (int)k;
    // ---- A: the shared experts' pairs as ONE flat list (a slice may straddle two experts)
    {
        const int P = n_fix * (I_sh / 16);
        clear();
        epf_deal(P, wa, 0, first, count);
        for (int g = 0; g < n_fix; ++g) { pl.proda_base[g] = -1; pl.proda_n[g] = 0; }
        for (int w : wa) {
            int f = first[w], c = count[w];
            while (c > 0) {
                const int k = std::min(c, 7);
                la[w].push_back(epf_task(EPF_A, 0, f, k));
                f += k; c -= k;
            }
            if (count[w] > 0) {
                la[w].push_back(epf_task(EPF_PUB_A, 0, pos_a[w], 0));
                for (int g = 0; g < n_fix; ++g) {
                    const int lo = g * (I_sh / 16), hi = lo + I_sh / 16;
                    if (first[w] < hi && first[w] + count[w] > lo) {
                        if (pl.proda_base[g] < 0) pl.proda_base[g] = pos_a[w];
                        pl.proda_n[g] = pos_a[w] - pl.proda_base[g] + 1;
                    }
                }
            }
        }
        for (int g = 0; g < n_fix; ++g)
            if (pl.proda_base[g] < 0 || pl.proda_n[g] > 512) return;
    }
    // ---- B: local experts' pairs as (expert, tile half, pair) units -- at 8 tiles a pass takes four of them and two pairs --; 1..4 pairs per
    // pass at 2 tiles, 1..2 at 4 and at 8
    {
        const int PP = I_dyn / 16, H = R >= 8 ? 2 : 1, P = E_loc * H * PP;
        clear();
        epf_deal(P, wb, 0, first, count);
        const int pmax = R >= 8 ? 2 : 8 / R;
        for (int g = 0; g < E_loc; ++g) { pl.prodb_base[g] = -1; pl.prodb_n[g] = 0; }
        for (int w : wb) {
            int f = first[w], c = count[w];
            bool any = false;
            while (c > 0) {
                const int g = f / (H * PP), hf = (f / PP) % H, lp = f % PP;
                const int k = std::min(std::min(c, pmax), PP - lp);
                lb[w].push_back(epf_task(EPF_B, g | (hf << 2), lp, k));
                if (pl.prodb_base[g] < 0) pl.prodb_base[g] = pos_b[w];
                pl.prodb_n[g] = pos_b[w] - pl.prodb_base[g] + 1;
                f += k; c -= k;
                any = true;
            }
            if (any) lb[w].push_back(epf_task(EPF_PUB_B, 0, pos_b[w], 0));
        }
        for (int g = 0; g < E_loc; ++g)
            if (pl.prodb_base[g] < 0 || pl.prodb_n[g] > 512) return;
    }
    // ---- C: local experts' down blocks as (expert, tile half, block) units -- at 8 tiles a pass takes four of them --, 1..4 blocks per pass
    {
        const int NB = D / 16, H = R >= 8 ? 2 : 1, P = E_loc * H * NB;
        clear();
        epf_deal(P, wc, (int)wc.size() / 3, first, count);
        pl.n_cwg = 0;
        for (int w : wc) {
            int f = first[w], c = count[w];
            bool any = false;
            while (c > 0) {
                const int g = f / (H * NB), hf = (f / NB) % H, lb0 = f % NB;
                const int k = std::min(std::min(c, 4), NB - lb0);
                lc[w].push_back(epf_task(EPF_C, g | (hf << 2), lb0, k));
                f += k; c -= k;
                any = true;
            }
            if (any) {
                lc[w].push_back(epf_task(EPF_SIG_C, 0, 0, 0));
                pl.n_cwg += 1;
            }
        }
    }
    // ---- D: shared experts' down blocks (43 k-steps each: half a routed block), per expert
    {
        const int NB = D / 16, P = n_fix * NB;
        clear();
        epf_deal(P, wd, 0, first, count);
        for (int w : wd) {
            int f = first[w], c = count[w];
            while (c > 0) {
                const int g = f / NB, lb0 = f % NB;
                const int k = std::min(std::min(c, 10), NB - lb0);
                ld[w].push_back(epf_task(EPF_D, g, lb0, k));
                f += k; c -= k;
            }
        }
    }
    pl.tasks.assign((size_t)n_wg * UMOE_EPF_MAXT, 0u);
    for (int w = 0; w < n_wg; ++w) {
        std::vector<uint32_t> l;
        if (w < R) l.push_back(epf_task(EPF_TILE, w, 0, roles ? 2 : 3));          // re-lay tile w (few workgroups: push to rank w first)
        else if (w < RT) l.push_back(epf_task(EPF_TILE, w - R, 0, 1));              // push the own rows to rank w - R
        else if (w < RT + S) l.push_back(epf_task(EPF_ROUTER, 0, w - RT, 0));
        // with roles: A | B or D | C (D only needs A; C hangs on every workgroup's B); without: A | B | C | D (D hides the return flight)
        l.insert(l.end(), la[w].begin(), la[w].end());
        l.insert(l.end(), lb[w].begin(), lb[w].end());
        if (roles) l.insert(l.end(), ld[w].begin(), ld[w].end());
        l.insert(l.end(), lc[w].begin(), lc[w].end());
        if (!roles) l.insert(l.end(), ld[w].begin(), ld[w].end());
        if ((int)l.size() >= UMOE_EPF_MAXT) return;      // (the last word stays EPF_END)
        for (size_t k = 0; k < l.size(); ++k) pl.tasks[(size_t)w * UMOE_EPF_MAXT + k] = l[k];
    }
    pl.ok = true;
}

static unsigned long long* g_epf_dbg = nullptr;
// diagnostics: NULL enables the stamps of the instrumented build (outside any capture); otherwise copies the last launch's [256][16] stamps
extern "C" int umoe_moe_ep_stamps(unsigned long long* host_out) {
    if (!host_out) {
        if (!g_epf_dbg && hipMalloc(&g_epf_dbg, sizeof(unsigned long long) * 16 * 256) != hipSuccess) return -2;
        return hipMemset(g_epf_dbg, 0, sizeof(unsigned long long) * 16 * 256) == hipSuccess ? 0 : -2;
    }
    if (!g_epf_dbg) return -1;
    return hipMemcpy(host_out, g_epf_dbg, sizeof(unsigned long long) * 16 * 256, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}

// test hook (no GPU needed): the task lists of a shape; out[0] = ok, out[1] = n_cwg, then n_wg * UMOE_EPF_MAXT task words
extern "C" int umoe_moe_ep_plan_probe(int n_wg, int R, int E_loc, int S, int D, int I_dyn, int I_sh, int n_fix, uint32_t* out, int out_len) {
    if (!out || n_wg < 1 || n_wg > 256 || out_len < 2 + n_wg * UMOE_EPF_MAXT) return -1;
    EpfPlan pl;
    epf_plan(n_wg, R, E_loc, S, D, I_dyn, I_sh, n_fix, pl);
    out[0] = pl.ok ? 1u : 0u;
    out[1] = (uint32_t)pl.n_cwg;
    if (pl.ok) memcpy(out + 2, pl.tasks.data(), pl.tasks.size() * sizeof(uint32_t));
    return 0;
}

// Builds the task table of an engine (host -> `tasks_dev`, n_wg * UMOE_EPF_MAXT words; NOT inside a stream capture).  Returns 0 and the
// number of counting workgroups, 1 when the shape has no plan.
int umoe_moe_ep_prepare(umoe_epf_desc* d, hipStream_t s) {
    UMOE_REQUIRE(d && d->tasks_dev, "umoe_moe_ep_prepare: null argument");
    EpfPlan pl;
    epf_plan(d->n_wg, d->R, d->E_loc, d->S, d->D, d->I_dyn, d->I_sh, d->n_fix, pl);
    if (!pl.ok) return 1;
    UMOE_HIP(hipMemcpyAsync(d->tasks_dev, pl.tasks.data(), pl.tasks.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    UMOE_HIP(hipStreamSynchronize(s));
    d->n_cwg = pl.n_cwg;
    for (int g = 0; g < 4; ++g) { d->proda_base[g] = pl.proda_base[g]; d->proda_n[g] = pl.proda_n[g]; }
    for (int g = 0; g < UMOE_MT_MAXG; ++g) { d->prodb_base[g] = pl.prodb_base[g]; d->prodb_n[g] = pl.prodb_n[g]; }
    if (getenv("UMOE_EPF_DEBUG")) {
        fprintf(stderr, "umoe_moe_ep: plan n_wg %d R %d E_loc %d n_cwg %d\n", d->n_wg, d->R, d->E_loc, pl.n_cwg);
        for (int w = 0; w < d->n_wg; ++w) {
            fprintf(stderr, "  wg %3d:", w);
            for (int k = 0; k < UMOE_EPF_MAXT && pl.tasks[(size_t)w * UMOE_EPF_MAXT + k]; ++k) {
                const uint32_t t = pl.tasks[(size_t)w * UMOE_EPF_MAXT + k];
                fprintf(stderr, " %u:%u:%u+%u", t >> 28, (t >> 24) & 15u, (t >> 8) & 0xffffu, t & 255u);
            }
            fprintf(stderr, "\n");
        }
    }
    return 0;
}

template <int MT>
static int epf_launch(const epf_args& P, const umoe_router_args& ra, const umoe_rider_pub& pub, int n_wg, size_t lds, hipStream_t s) {
    static size_t configured = 0;
    if (lds + EPF_RIDER_LDS > configured) {
        UMOE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&moe_ep_kernel<MT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + EPF_RIDER_LDS)));
        configured = lds + EPF_RIDER_LDS;
    }
    moe_ep_kernel<MT><<<dim3((unsigned)n_wg), 512, lds + EPF_RIDER_LDS, s>>>(P, ra, pub, (int)lds);
    UMOE_LAUNCH_CHECK();
    return 0;
}

int umoe_moe_ep(const umoe_epf_desc* d, hipStream_t s) {
    UMOE_REQUIRE(d && d->router && d->pub && d->tasks_dev && d->n_cwg > 0, "umoe_moe_ep: not prepared");
    const umoe_router_args* r = d->router;
    UMOE_REQUIRE(r->S == d->S && r->S >= 1 && r->S <= 16 && r->n_dyn == 9 && r->n_fix == 2 && r->D == 2048 && d->D == 2048 && r->x && r->gate_w && r->expert_mask &&
                     !r->logits_in && !r->norm_only && r->norm_w && !r->gumbel && !r->x_noise && !r->attn_mask && d->n_fix == 2,
                 "umoe_moe_ep: built for n_dyn 9 / n_fix 2, D 2048, <= 16 rows");
    epf_args P;
    memset(&P, 0, sizeof(P));
    flat_args& A = P.S;
    A.a = r->x; A.norm_w = r->norm_w; A.rms_eps = r->rms_eps; A.lda = d->D; A.S = d->S; A.G = d->n_fix; A.kb_gu = d->D / 32;
    A.h = d->h_sh; A.ldh = d->ldh; A.y = d->y_sh; A.ldy = d->ldy; A.flags = d->flags;
    A.dbg = g_epf_dbg;
    int PP = 0;
    for (int i = 0; i < d->n_fix; ++i) {
        A.w_gu[i] = d->w_sgu[i]; A.w_dn[i] = d->w_sdn[i];
        A.pair0[i] = PP; PP += d->I_sh / 16;
        A.h_row[i] = d->h_row0 + i * d->S;
        A.dn_kb[i] = d->I_sh / 32; A.dn_a_row[i] = d->h_row0 + i * d->S; A.dn_y_row[i] = d->y_row0 + i * d->S; A.dn_nb[i] = d->D / 16;
        A.prod_base[i] = d->proda_base[i]; A.prod_n[i] = d->proda_n[i];
    }
    for (int i = d->n_fix; i <= FLAT_MAXG; ++i) A.pair0[i] = PP;
    for (int q = 0; q < d->E_loc; ++q) { P.w_lgu[q] = d->w_lgu[q]; P.w_ldn[q] = d->w_ldn[q]; P.prodb_base[q] = d->prodb_base[q]; P.prodb_n[q] = d->prodb_n[q]; }
    P.xgp = d->xgp; P.hpk = d->hpk; P.D = d->D; P.I = d->I_dyn; P.E_loc = d->E_loc; P.R = d->R; P.rank = d->rank; P.loopback = d->loopback;
    for (int p = 0; p < d->R; ++p) P.peer_base[p] = d->peer_base[p];
    P.disp_off = d->disp_off; P.ret_off = d->ret_off;
    P.flag_b = d->flags + d->n_wg; P.tile_ready = d->flags + 2 * d->n_wg;
    P.round = d->round; P.tasks = d->tasks_dev;
    UMOE_REQUIRE(d->flag_words >= 2 * d->n_wg + UMOE_FLAG_REPL * 16, "umoe_moe_ep: %d flag words needed", 2 * d->n_wg + UMOE_FLAG_REPL * 16);
    const umoe_rider_pub pub = *d->pub;
    UMOE_REQUIRE(pub.step && pub.err, "umoe_moe_ep: pub needs step / err");
    auto stage_bytes = [](int kb) { return (size_t)16 * 4 * (size_t)((kb * 16 + 255) & ~255); };
    size_t lds = std::max(stage_bytes(d->D / 32) + 4096, stage_bytes(d->I_sh / 32));
    lds = std::max(lds, (size_t)8 * 16 * 1024);      // reduction slabs: 8 waves x (NT x MT = 16 | 14 | 10) KiB
    umoe_router_args rr = *r;
    rr.h_out = nullptr;
    if (d->R == 2) return epf_launch<2>(P, rr, pub, d->n_wg, lds, s);
    if (d->R == 4) return epf_launch<4>(P, rr, pub, d->n_wg, lds, s);
    return epf_launch<8>(P, rr, pub, d->n_wg, lds, s);
}
