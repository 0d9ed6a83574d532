// Weight-gradient GEMM that contracts over the ROW index of both operands ("TN"), so dW = dY^T X runs on the activations as autograd holds
// them -- no transposed copies (round 2: umoe_transpose_slots wrote dY^T, H^T, (dG|dU)^T and X^T of every layer first, 9 % of the step).
//
//   out_g[m][n] = sum over k in [k_off_g, k_off_g + k_count_g) of  P[k][p_col_off_g + m] * Q[k][q_col_off_g + n]      (bf16 in, fp32 sum)
//
// P, Q row-major with k (tokens / expert slots) as the row index.  Same tile, wave layout, LDS-DMA ring and barrier schedule as
// tgemm_pp_kernel (umoe_tgemm.hip: 256 x 256 output tile, 8 waves in two groups one barrier apart, K tiles of 32, ring of 4 slots, the
// DMA three tiles ahead, counted vmcnt) -- what differs is the LDS image and the operand read:
//  * a unit is 32 k-rows x 256 columns (512-byte rows).  One global_load_lds_dwordx4 of a wave writes 1 KiB = two k-rows; a row PAIR is
//    followed by 64 bytes of padding (pair stride 1088 B) and the 16-byte chunks of the odd row of a pair are stored with chunk index ^ 2
//    (on the SOURCE address: the DMA writes linearly);
//  * MFMA operands come out of the k-major image by ds_read_b64_tr_b16 (hardware transpose: per 16-lane group, lane 4q+p supplies the
//    address of row q, columns 4p..4p+3; lane i receives column i of the four rows): two reads per 16x16x32 operand.  Lane group h takes
//    LDS rows 4h..4h+3 with the first read and 16+4h.. with the second; a 32-lane half of a read then touches eight consecutive LDS rows
//    = four pairs whose 32-byte granules fall on bank groups {0,1}+2t (pad) with the XOR choosing inside the pair: every bank once.
//    The DMA puts window row 8h+e of a K tile into LDS row 4h+e (e < 4) / 16+4h+e-4, so k-slot (h, e) of the MFMA holds window row 8h+e
//    exactly as in tgemm_pp_kernel over a transposed copy;
//  * the 16-column fragments of a wave are STRIDED over the tile (Q side: columns 16 (4j + wc), P side: 16 (2i + wr)) so that the XOR bit
//    is a wave constant and every fragment is one base register + an immediate offset.
// Results: the fp32 sums run over the same 32-row tiles in the same order as tgemm_pp_kernel over transposed copies, and inside a tile
// every MFMA gets the same operands in the same k slots -- tests/test_gpu_ops.py compares the two bit for bit (no K split).
// Roofline: MFMA (2.5 PFLOP/s dense bf16).
#include "umoe_common.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#define TN_MAXG 24
struct tn_pack { umoe_tn_group_t g[TN_MAXG]; };
__device__ __attribute__((aligned(16))) uint4 tn_zero16;
typedef unsigned int tn_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int tn_u32x4 __attribute__((ext_vector_type(4)));

// One MFMA operand = two transposing reads (rows 4h.. and 16 + 4h..: eight row pairs = 8 * 1088 bytes further).  Inline assembly, not
// __builtin_amdgcn_ds_read_tr16_b64: hipcc (ROCm 7.2) puts s_waitcnt vmcnt(0) in front of the builtin while an LDS-DMA is in flight (it
// carries no memory operand, so the wait-count pass assumes it may read what the DMA writes) -- that drains the three-tile prefetch in
// every phase.  The asm reads are invisible to that pass: the s_waitcnt lgkmcnt(0) + sched_barrier in front of the MFMAs are explicit.
template <int OFF>
__device__ __forceinline__ bf16x8_t tn_frag(const unsigned addr) {
    tn_u32x2 a, b;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(a) : "v"(addr), "n"(OFF));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(b) : "v"(addr), "n"(OFF + 8 * 1088));
    return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(a, b, 0, 1, 2, 3));
}
template <int N, int STEP, int OFF0>
__device__ __forceinline__ void tn_frags(bf16x8_t (&f)[N], const unsigned addr) {
    if constexpr (N >= 1) f[0] = tn_frag<OFF0>(addr);
    if constexpr (N >= 2) f[1] = tn_frag<OFF0 + STEP>(addr);
    if constexpr (N >= 3) f[2] = tn_frag<OFF0 + 2 * STEP>(addr);
    if constexpr (N >= 4) f[3] = tn_frag<OFF0 + 3 * STEP>(addr);
}

template <int NS, bool F32OUT>
__global__ __launch_bounds__(512, 2) void tgemm_tn_kernel(const umoe_tgemm_tn_args p, const tn_pack gp, const int nx, const int ny, const int nz, const int ragged_order,
                                                           const unsigned total_wgs, const int ksplit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PAIR = 1088, UNIT = 16 * PAIR, SLOT = 2 * UNIT;
    constexpr int AH = NS - 1;
    // XCD-aware tile order as in tgemm_pp_kernel (1-D launch, lin % 8 = the workgroup's XCD)
    int bx, by, bz;
    {
        const unsigned lin = blockIdx.x, xcd = lin & 7, seq = lin >> 3;
        if (ragged_order) {
            const unsigned rr = seq / (unsigned)nx;
            bx = (int)(seq - rr * (unsigned)nx);
            const unsigned RR = rr * 8 + xcd;
            if (RR >= (unsigned)(ny * nz)) return;
            bz = (int)(RR / (unsigned)ny);
            by = (int)(RR - (unsigned)bz * (unsigned)ny);
        } else {
            const unsigned nwg = total_wgs, q = nwg >> 3, r = nwg & 7;
            const unsigned id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + seq;
            bx = (int)(id % (unsigned)nx);
            const unsigned RR = id / (unsigned)nx;
            bz = (int)(RR / (unsigned)ny);
            by = (int)(RR - (unsigned)bz * (unsigned)ny);
        }
    }
    // K split (static groups that would not fill the chip): z index = group * ksplit + part, part takes K tiles [part * per, ...)
    const int part = bz % ksplit;
    const umoe_tn_group_t g = gp.g[bz / ksplit];
    const int m0 = by * 256, n0 = bx * 256;
    if (m0 >= g.m || n0 >= g.n) return;
    int koff = g.k_off_dev ? *g.k_off_dev : g.k_off;
    int K = g.k_count_dev ? *g.k_count_dev : g.k;
    if (ksplit > 1) {
        const int per = (((K + 31) >> 5) + ksplit - 1) / ksplit * 32;      // whole 32-row tiles per part
        const int lo = part * per;
        const int hi = lo + per < K ? lo + per : K;
        koff += lo;
        K = hi > lo ? hi - lo : 0;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- LDS-DMA sources: this wave stages the row pairs `wave` and `wave + 8` of every unit
    const char* zero = reinterpret_cast<const char*>(&tn_zero16);
    const int rodd = lane >> 5;
    const int cl = (lane & 31) ^ (rodd << 1);               // logical 16-byte chunk (8 columns) this lane fetches
    const uint16_t* Pb = g.p ? g.p : p.p;
    const uint16_t* Qb = g.q ? g.q : p.q;
    const long ldp = g.p ? g.ldp : p.ldp, ldq = g.q ? g.ldq : p.ldq;
    long pdel[2], qdel[2];
    int rrow[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        rrow[q] = 8 * (wave >> 1) + 4 * q + 2 * (wave & 1) + rodd;      // LDS row 4h + e (pairs 0..7) / 16 + 4h + e - 4 (pairs 8..15) <- window row 8h + e
        const int mc = m0 + 8 * cl, nc = n0 + 8 * cl;
        pdel[q] = mc < g.m ? reinterpret_cast<const char*>(Pb + (long)(koff + rrow[q]) * ldp + g.p_col_off + mc) - zero : 0;
        qdel[q] = nc < g.n ? reinterpret_cast<const char*>(Qb + (long)(koff + rrow[q]) * ldq + g.q_col_off + nc) - zero : 0;
    }
    const long pstep = 64 * ldp, qstep = 64 * ldq;          // bytes per K tile (32 rows)
    auto stage = [&](const long (&del)[2], const long step, const int unit_off, const int tile) {
        char* base = smem + (tile % NS) * SLOT + unit_off;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const long live = (tile * 32 + rrow[q] < K) ? -1L : 0L;       // rows behind the window read zeros (also the tiles staged past the end)
            const long d = del[q] == 0 ? 0 : ((del[q] + (long)tile * step) & live);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(zero + d),
                                             (__attribute__((address_space(3))) void*)(base + (wave + 8 * q) * PAIR), 16, 0, 0);
        }
    };
    const char* pptr[2];
    const char* qptr[2];
    long pinc[2], qinc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        pinc[q] = pdel[q] ? pstep : 0;
        qinc[q] = qdel[q] ? qstep : 0;
        pptr[q] = zero + pdel[q] + AH * pinc[q];
        qptr[q] = zero + qdel[q] + AH * qinc[q];
    }
    auto stage_run = [&](const char* (&ptr)[2], const long (&inc)[2], const int unit_off, const int slot_off) {
        char* base = smem + slot_off + unit_off;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ptr[q],
                                             (__attribute__((address_space(3))) void*)(base + (wave + 8 * q) * PAIR), 16, 0, 0);
            ptr[q] += inc[q];
        }
    };

    f32x4_t acc[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // operand reads: lane (h, 4 rq + rp) supplies row 4h + rq, columns 4 rp .. 4 rp + 3 of the fragment
    const int h = lane >> 4, c16 = lane & 15, rq = c16 >> 2, rp = c16 & 3;
    const int rowoff = (2 * h + (rq >> 1)) * PAIR + (rq & 1) * 512 + (rp & 1) * 8;
    const int qbase = rowoff + (((2 * wc + (rp >> 1)) ^ ((rq & 1) << 1)) << 4);                // fragment j: + 128 j
    const int pbase = UNIT + rowoff + (((2 * wr + (rp >> 1)) ^ ((rq & 1) << 1)) << 4);         // fragment i: + 64 i

    const unsigned lds0 = (unsigned)reinterpret_cast<size_t>(smem);      // LDS byte offset of the ring (low half of the generic pointer)
    const int KT = (K + 31) >> 5;
#pragma unroll
    for (int u = 0; u < AH; ++u) {
        stage(qdel, qstep, 0, u);
        stage(pdel, pstep, UNIT, u);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (AH - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();      // group 1 runs one barrier behind group 0 from here on
    // ONE load segment and ONE MFMA segment of 32 per K tile (tgemm_pp_kernel: two of 16): half as many barriers per MFMA -- 5-7 % on
    // every shape of the training step (scripts/gemm_ab.py).  The fragment reads are waited for IN FRONT of the barrier that ends the load
    // segment (lgkmcnt(0)): the partner group restages the slot of tile v - 1 in its next load segment, one barrier behind this group's
    // last read of it, so those reads have to be complete when anybody passes that barrier (WAR with one phase of distance; RAW as in
    // tgemm_pp_kernel: the counted vmcnt in front of the same barrier retires every wave's pieces of tile v + 1, read one phase later).
    auto tile_step = [&](const int v, auto steady_tag) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        const int so = (v % NS) * SLOT;
        const int sn = ((v + AH) % NS) * SLOT;
        const unsigned Qf = lds0 + so + qbase;
        const unsigned Pf = lds0 + so + pbase;
        bf16x8_t wf[4], af[8];
        tn_frags<4, 128, 0>(wf, Qf);
        tn_frags<4, 64, 0>(*reinterpret_cast<bf16x8_t (*)[4]>(&af[0]), Pf);
        tn_frags<4, 64, 256>(*reinterpret_cast<bf16x8_t (*)[4]>(&af[4]), Pf);
        if (STEADY) { stage_run(qptr, qinc, 0, sn); stage_run(pptr, pinc, UNIT, sn); }
        else { stage(qdel, qstep, 0, v + AH); stage(pdel, pstep, UNIT, v + AH); }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (AH - 1)) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[j][i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
    };
    const int v_steady = KT - (AH + 1) > 0 ? KT - (AH + 1) : 0;
    int v = 0;
    for (; v < v_steady; ++v) tile_step(v, std::true_type{});
    for (; v < KT; ++v) tile_step(v, std::false_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-tile DMAs of the last iterations must not outlive the workgroup
    if (wr == 0) __builtin_amdgcn_s_barrier();

    // ---- store: lane holds 4 consecutive output columns (n) of one output row (m)
    const long orow_base = (long)g.out_row_base + m0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = 16 * (2 * i + wr) + c16;
        if (m0 + r >= g.m) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + 16 * (4 * j + wc) + 4 * h;
            if (col >= g.n) continue;
            if constexpr (F32OUT) {
                float* o = reinterpret_cast<float*>(g.out ? g.out : p.out) + (size_t)part * p.part_stride + (orow_base + r) * (long)p.ldo + g.out_col_off + col;
                *reinterpret_cast<float4*>(o) = make_float4(acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]);
            } else {
                uint16_t* o = reinterpret_cast<uint16_t*>(g.out ? g.out : p.out) + (orow_base + r) * (long)p.ldo + g.out_col_off + col;
                *reinterpret_cast<uint2*>(o) = make_uint2((uint32_t)f2bf(acc[j][i][0]) | ((uint32_t)f2bf(acc[j][i][1]) << 16),
                                                          (uint32_t)f2bf(acc[j][i][2]) | ((uint32_t)f2bf(acc[j][i][3]) << 16));
            }
        }
    }
}

// ------------------------------------------------------------------------------------ input gradients on the weights as stored ("NN")
//   Y[rows(g)][n] = sum_k A[rows(g)][k] * W_g[k][n],   W_g row-major [K][ldw] with the CONTRACTION index as its row
// -- dX = dY W for an nn.Linear weight W [N_out][K_in] as the optimizer holds it: no W^T copy (round 2 kept transposed copies of every
// weight while the parameters were unchanged, +10.5 GB, and rebuilt them in every step of a trainer that steps after each backward).
// The tile is tgemm_pp_kernel's; the token unit is its image (256 rows x 64 B, ds_read_b128), the weight unit is the k-major image of
// tgemm_tn_kernel (transposing reads).  `w2` continues the contraction behind the first k_w1 rows of `w` ((dG | dU) against Wg then Wu).
struct tn_tgpack { umoe_tgroup_t g[12]; };

__global__ __launch_bounds__(512, 2) void tgemm_nn_kernel(const umoe_tgemm_args p, const tn_tgpack gp, const int nx, const int ny, const int nz, const int ragged_order,
                                                           const unsigned total_wgs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NS = 4, AH = NS - 1;
    constexpr int PAIR = 1088, UNITW = 16 * PAIR, UNITT = 256 * 64, SLOT = UNITW + UNITT;
    int bx, by, bz;
    {
        const unsigned lin = blockIdx.x, xcd = lin & 7, seq = lin >> 3;
        if (ragged_order) {
            const unsigned rr = seq / (unsigned)nx;
            bx = (int)(seq - rr * (unsigned)nx);
            const unsigned RR = rr * 8 + xcd;
            if (RR >= (unsigned)(ny * nz)) return;
            bz = (int)(RR / (unsigned)ny);
            by = (int)(RR - (unsigned)bz * (unsigned)ny);
        } else {
            const unsigned nwg = total_wgs, q = nwg >> 3, r = nwg & 7;
            const unsigned id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + seq;
            bx = (int)(id % (unsigned)nx);
            const unsigned RR = id / (unsigned)nx;
            bz = (int)(RR / (unsigned)ny);
            by = (int)(RR - (unsigned)bz * (unsigned)ny);
        }
    }
    const umoe_tgroup_t g = gp.g[bz];
    const int count = g.count ? *g.count : g.static_count;
    const int roff = g.row_off ? *g.row_off : 0;
    const int row0 = by * 256, n0 = bx * 256;
    if (row0 >= count || n0 >= g.n) return;
    const int K = g.k;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const char* zero = reinterpret_cast<const char*>(&tn_zero16);

    // ---- weight unit (k-major): this wave stages the row pairs `wave`, `wave + 8`; two sources: w for tiles below sw, w2 behind
    const int rodd = lane >> 5;
    const int cl = (lane & 31) ^ (rodd << 1);
    const int sw = g.w2 ? g.k_w1 >> 5 : 0x7fffffff;          // first K tile that comes from w2
    long wdel[2], wdel2[2];
    int rrow[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        rrow[q] = 8 * (wave >> 1) + 4 * q + 2 * (wave & 1) + rodd;
        const int nc = n0 + 8 * cl;
        wdel[q] = nc < g.n ? reinterpret_cast<const char*>(g.w + (long)rrow[q] * g.ldw + nc) - zero : 0;
        wdel2[q] = (nc < g.n && g.w2) ? reinterpret_cast<const char*>(g.w2 + (long)rrow[q] * g.ldw + nc) - zero : 0;
    }
    const long wstep = 64 * (long)g.ldw;
    // ---- token unit (contraction-contiguous rows, as in tgemm_pp_kernel): row groups `wave`, `wave + 8` of 16 rows
    const int rl = lane >> 2, slot = lane & 3;
    const int gch = slot ^ ((0 - (rl >> 2)) & 3);
    long tdel[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int tr = (wave + 8 * q) * 16 + rl;
        const int r = row0 + tr;
        tdel[q] = 0;
        if (r < count) {
            const long arow = g.rows ? (long)g.rows[roff + r] : (long)(g.a_row_base + roff + r);
            tdel[q] = reinterpret_cast<const char*>(p.a + arow * (long)p.lda + g.a_col_off + gch * 8) - zero;
        }
    }
    auto stage_w = [&](const int tile) {
        char* base = smem + (tile % NS) * SLOT;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const long live = (tile * 32 + rrow[q] < K) ? -1L : 0L;
            const long src = tile < sw ? (wdel[q] == 0 ? 0 : wdel[q] + (long)tile * wstep) : (wdel2[q] == 0 ? 0 : wdel2[q] + (long)(tile - sw) * wstep);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(zero + (src & live)),
                                             (__attribute__((address_space(3))) void*)(base + (wave + 8 * q) * PAIR), 16, 0, 0);
        }
    };
    auto stage_t = [&](const int tile) {
        const int k0 = tile * 32;
        const long live = (k0 + gch * 8 < K) ? -1L : 0L;
        char* base = smem + (tile % NS) * SLOT + UNITW;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const long d = tdel[q] == 0 ? 0 : ((tdel[q] + 2L * k0) & live);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(zero + d),
                                             (__attribute__((address_space(3))) void*)(base + (wave + 8 * q) * 1024), 16, 0, 0);
        }
    };
    const char* tptr[2];
    const char* wptr[2];
    long tinc[2], winc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        tinc[q] = tdel[q] ? 64 : 0;
        winc[q] = wdel[q] ? wstep : 0;
        tptr[q] = zero + tdel[q] + AH * tinc[q];
        wptr[q] = zero + wdel[q] + AH * winc[q];          // (re-based at the switch to w2, below)
    }
    auto run_w = [&](const int tile, const int slot_off) {
        if (tile == sw) {
#pragma unroll
            for (int q = 0; q < 2; ++q) wptr[q] = zero + wdel2[q];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)wptr[q],
                                             (__attribute__((address_space(3))) void*)(smem + slot_off + (wave + 8 * q) * PAIR), 16, 0, 0);
            wptr[q] += winc[q];
        }
    };
    auto run_t = [&](const int slot_off) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)tptr[q],
                                             (__attribute__((address_space(3))) void*)(smem + slot_off + UNITW + (wave + 8 * q) * 1024), 16, 0, 0);
            tptr[q] += tinc[q];
        }
    };

    f32x4_t acc[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int h = lane >> 4, c16 = lane & 15, rq = c16 >> 2, rp = c16 & 3;
    const int wbase = (2 * h + (rq >> 1)) * PAIR + (rq & 1) * 512 + (rp & 1) * 8 + (((2 * wc + (rp >> 1)) ^ ((rq & 1) << 1)) << 4);   // fragment j: + 128 j
    const int trow = 128 * wr + c16;
    const int tbase = UNITW + trow * 64 + ((h ^ ((0 - (trow >> 2)) & 3)) << 4);                                                       // fragment i: + 1024 i
    const unsigned lds0 = (unsigned)reinterpret_cast<size_t>(smem);
    const int KT = (K + 31) >> 5;
#pragma unroll
    for (int u = 0; u < AH; ++u) {
        stage_w(u);
        stage_t(u);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (AH - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();
    // ONE load segment and ONE MFMA segment of 32 per K tile, as in tgemm_tn_kernel above (same hazard argument)
    auto tile_step = [&](const int v, auto steady_tag) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        const int so = (v % NS) * SLOT;
        const int sn = ((v + AH) % NS) * SLOT;
        const unsigned Wf = lds0 + so + wbase;
        const char* Tb = smem + so + tbase;
        bf16x8_t wf[4], af[8];
        tn_frags<4, 128, 0>(wf, Wf);
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Tb + i * 1024));
        if (STEADY) { run_w(v + AH, sn); run_t(sn); }
        else { stage_w(v + AH); stage_t(v + AH); }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (AH - 1)) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[j][i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
    };
    const int v_steady = KT - (AH + 1) > 0 ? KT - (AH + 1) : 0;
    int v = 0;
    for (; v < v_steady; ++v) tile_step(v, std::true_type{});
    for (; v < KT; ++v) tile_step(v, std::false_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wr == 0) __builtin_amdgcn_s_barrier();

    uint16_t* out = reinterpret_cast<uint16_t*>(p.out) + g.out_col_off;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = row0 + 128 * wr + 16 * i + c16;
        if (r >= count) continue;
        const long orow = (long)g.out_row_base + roff + r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + 16 * (4 * j + wc) + 4 * h;
            if (col >= g.n) continue;
            *reinterpret_cast<uint2*>(out + orow * (long)p.ldo + col) = make_uint2((uint32_t)f2bf(acc[j][i][0]) | ((uint32_t)f2bf(acc[j][i][1]) << 16),
                                                                                   (uint32_t)f2bf(acc[j][i][2]) | ((uint32_t)f2bf(acc[j][i][3]) << 16));
        }
    }
}

// called by umoe_tiled_gemm when the groups' weights are k-major (umoe_tgroup_t.w_kmajor)
int umoe_tiled_gemm_nn_launch(const umoe_tgemm_args* a, int max_n, hipStream_t s) {
    UMOE_REQUIRE(a->epilogue == UMOE_EPI_BF16, "umoe_tiled_gemm: k-major weights (w_kmajor) take the plain bf16 epilogue only");
    UMOE_REQUIRE((a->ldo & 3) == 0, "umoe_tiled_gemm: k-major weights: ldo must be a multiple of 4");
    int ragged = 0;
    for (int i = 0; i < a->num_groups; ++i) {
        const umoe_tgroup_t& g = a->groups[i];
        // (k need not be a multiple of 8: the weight rows behind k read as zeros; the activation's last 8-column chunk must lie inside
        //  its row -- lda >= roundup8(k) -- and hold finite values, zeros as train._pad8 writes them)
        UMOE_REQUIRE(g.w_kmajor && !g.bias && !g.k_off && !g.k_count && g.n % 8 == 0 && g.ldw % 8 == 0 && g.ldw >= g.n && (g.out_col_off & 3) == 0 &&
                         (reinterpret_cast<size_t>(g.w) & 15) == 0 && g.a_col_off + ((g.k + 7) & ~7) <= a->lda,
                     "umoe_tiled_gemm: group %d: k-major weights need n %% 8, ldw %% 8, no bias / K window, 16-byte aligned rows, lda >= roundup8(k) (all groups of a launch alike)", i);
        UMOE_REQUIRE(!g.w2 || (g.k_w1 > 0 && g.k_w1 % 32 == 0 && g.k_w1 < g.k && (reinterpret_cast<size_t>(g.w2) & 15) == 0),
                     "umoe_tiled_gemm: group %d: w2 continues the contraction behind k_w1 rows of w: 0 < k_w1 < k, k_w1 %% 32 == 0", i);
        ragged |= g.count != nullptr;
    }
    tn_tgpack gp;
    memset(&gp, 0, sizeof(gp));
    memcpy(gp.g, a->groups, sizeof(umoe_tgroup_t) * a->num_groups);
    constexpr int lds = 4 * (16 * 1088 + 256 * 64);
    static bool configured = false;
    if (!configured) {
        UMOE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tgemm_nn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        configured = true;
    }
    const int nx = ceil_div(max_n, 256), ny = ceil_div(a->max_rows, 256), nz = a->num_groups;
    const long nwg = ragged ? (long)nx * (((long)ny * nz + 7) & ~7L) : (long)nx * ny * nz;
    UMOE_REQUIRE(nwg < (1L << 31), "umoe_tiled_gemm: too many tiles (%ld)", nwg);
    tgemm_nn_kernel<<<dim3((unsigned)nwg), 512, lds, s>>>(*a, gp, nx, ny, nz, ragged, (unsigned)nwg);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// fixed-order sum of the K-split partials: out = bf16(part 0 + part 1 + ...)
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ parts, const long part_stride, const int nparts, const long n4, uint16_t* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 s = reinterpret_cast<const float4*>(parts)[i];
    for (int k = 1; k < nparts; ++k) {
        const float4 t = reinterpret_cast<const float4*>(parts + (size_t)k * part_stride)[i];
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    reinterpret_cast<uint2*>(out)[i] = make_uint2((uint32_t)f2bf(s.x) | ((uint32_t)f2bf(s.y) << 16), (uint32_t)f2bf(s.z) | ((uint32_t)f2bf(s.w) << 16));
}

template <bool F32OUT>
static int launch_tn(const umoe_tgemm_tn_args* a, int max_m, int max_n, int ragged, int ksplit, hipStream_t s) {
    constexpr int NS = 4;
    constexpr int lds = NS * 2 * 16 * 1088;
    tn_pack gp;
    memset(&gp, 0, sizeof(gp));
    memcpy(gp.g, a->groups, sizeof(umoe_tn_group_t) * a->num_groups);
    static bool configured = false;
    if (!configured) {
        UMOE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tgemm_tn_kernel<NS, F32OUT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        configured = true;
    }
    const int nx = ceil_div(max_n, 256), ny = ceil_div(max_m, 256), nz = a->num_groups * ksplit;
    const long nwg = ragged ? (long)nx * (((long)ny * nz + 7) & ~7L) : (long)nx * ny * nz;
    UMOE_REQUIRE(nwg < (1L << 31), "umoe_tiled_gemm_tn: too many tiles (%ld)", nwg);
    tgemm_tn_kernel<NS, F32OUT><<<dim3((unsigned)nwg), 512, lds, s>>>(*a, gp, nx, ny, nz, ragged, (unsigned)nwg, ksplit);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// K split of a launch (k_split < 0: chosen here).  Model fitted on MI355X (scripts/tn_bench.py): a workgroup needs 20 us + 0.55 us per
// 32-row K tile, the launch takes ceil(workgroups / CUs) rounds of that, the reduction 3 us + its bytes at 5 TB/s.
static int tn_effective_split(const umoe_tgemm_tn_args* a) {
    if (a->k_split >= 0) return a->k_split > 1 ? a->k_split : 1;
    long tiles = 0, elems = 0;
    int kmax = 0;
    for (int i = 0; i < a->num_groups; ++i) {
        const umoe_tn_group_t& g = a->groups[i];
        if (g.k_count_dev || g.out || g.out_col_off || g.n != a->ldo) return 1;      // device windows are balanced by their number; split needs one dense slab
        tiles += (long)ceil_div(g.m, 256) * ceil_div(g.n, 256);
        elems += (long)g.m * g.n;
        if (g.k > kmax) kmax = g.k;
    }
    static int cus = 0;
    if (!cus) {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        else (void)hipGetLastError();
        if (const char* v = getenv("UMOE_FAKE_CUS")) cus = atoi(v);      // (the co-residency guards' override: tests)
        if (cus <= 0) cus = 256;
    }
    const int KT = (kmax + 31) / 32;
    int best = 1;
    double best_t = 1e30;
    for (int ks = 1; ks <= 8; ++ks) {
        if (ks > 1 && KT / ks < 16) break;
        const double rounds = (double)((tiles * ks + cus - 1) / cus);
        const double t = rounds * (20.0 + 0.55 * ((KT + ks - 1) / ks)) + (ks > 1 ? 3.0 + (4.0 * ks + 2.0) * elems / 5.0e6 : 0.0);
        if (t < best_t - 1e-9) { best_t = t; best = ks; }
    }
    return best;
}

// the K split umoe_tiled_gemm_tn would use for these arguments (host logic only: tests/test_tn_plan_cpu.py pins the model's choices)
extern "C" int umoe_tiled_gemm_tn_split(const umoe_tgemm_tn_args* a) {
    if (!a || !a->groups || a->num_groups <= 0 || a->num_groups > TN_MAXG) return -1;
    return tn_effective_split(a);
}

extern "C" size_t umoe_tiled_gemm_tn_workspace_bytes(const umoe_tgemm_tn_args* a) {
    if (!a || !a->groups || a->num_groups <= 0 || a->num_groups > TN_MAXG) return 0;
    const int ks = a->k_split < 0 ? 8 : a->k_split;          // (auto: room for the largest split the model can choose)
    if (ks <= 1) return 0;
    return (size_t)ks * (size_t)a->part_stride * sizeof(float);
}

extern "C" int umoe_tiled_gemm_tn(const umoe_tgemm_tn_args* a, umoe_stream_t stream) {
    UMOE_REQUIRE(a && a->groups && a->out, "umoe_tiled_gemm_tn: null argument");
    UMOE_REQUIRE(a->num_groups > 0 && a->num_groups <= TN_MAXG, "umoe_tiled_gemm_tn: 1..%d groups per launch (got %d)", TN_MAXG, a->num_groups);
    UMOE_REQUIRE((a->ldo & 3) == 0, "umoe_tiled_gemm_tn: ldo must be a multiple of 4");
    int max_m = 0, max_n = 0, ragged = 0;
    for (int i = 0; i < a->num_groups; ++i) {
        const umoe_tn_group_t& g = a->groups[i];
        const uint16_t* P = g.p ? g.p : a->p;
        const uint16_t* Q = g.q ? g.q : a->q;
        const int ldp = g.p ? g.ldp : a->ldp, ldq = g.q ? g.ldq : a->ldq;
        UMOE_REQUIRE(P && Q && ldp % 8 == 0 && ldq % 8 == 0 && (reinterpret_cast<size_t>(P) & 15) == 0 && (reinterpret_cast<size_t>(Q) & 15) == 0,
                     "umoe_tiled_gemm_tn: group %d: operands must be 16-byte aligned with leading dimensions that are multiples of 8", i);
        // (the DMA fetches whole 8-column chunks: a last chunk that straddles m or n must lie inside the row -- its surplus columns only feed
        //  output rows / columns the stores skip)
        UMOE_REQUIRE(g.m > 0 && g.n > 0 && g.n % 4 == 0 && g.p_col_off % 8 == 0 && g.q_col_off % 8 == 0 && (g.out_col_off & 3) == 0 &&
                         g.p_col_off + ((g.m + 7) & ~7) <= ldp && g.q_col_off + ((g.n + 7) & ~7) <= ldq,
                     "umoe_tiled_gemm_tn: group %d: n %% 4, column offsets %% 8, and the 8-column chunks that hold m / n must lie inside a row (m=%d n=%d)", i, g.m, g.n);
        UMOE_REQUIRE((g.k_off_dev == nullptr) == (g.k_count_dev == nullptr), "umoe_tiled_gemm_tn: group %d: k_off_dev and k_count_dev come together", i);
        UMOE_REQUIRE(g.k_count_dev || (g.k >= 0 && g.k_off >= 0), "umoe_tiled_gemm_tn: group %d: bad static window", i);
        if (g.m > max_m) max_m = g.m;
        if (g.n > max_n) max_n = g.n;
        ragged |= g.k_count_dev != nullptr && a->num_groups > 1;
    }
    hipStream_t s = (hipStream_t)stream;
    const int ks = tn_effective_split(a);
    if (ks == 1) return launch_tn<false>(a, max_m, max_n, ragged, 1, s);
    // K split: fp32 partial slabs [k_split][part_stride] in the caller's workspace, summed in fixed order
    UMOE_REQUIRE(a->ws && ks <= 16, "umoe_tiled_gemm_tn: k_split needs a workspace (umoe_tiled_gemm_tn_workspace_bytes), k_split <= 16");
    UMOE_REQUIRE(!ragged, "umoe_tiled_gemm_tn: k_split needs static windows");
    UMOE_REQUIRE(a->part_stride % 4 == 0 && (reinterpret_cast<size_t>(a->ws) & 15) == 0, "umoe_tiled_gemm_tn: part_stride %% 4, 16-byte aligned workspace");
    umoe_tgemm_tn_args b = *a;
    umoe_tn_group_t gs[TN_MAXG];
    memcpy(gs, a->groups, sizeof(umoe_tn_group_t) * a->num_groups);
    for (int i = 0; i < a->num_groups; ++i) {
        UMOE_REQUIRE(!gs[i].out && !gs[i].k_count_dev && gs[i].out_col_off == 0 && gs[i].n == a->ldo,
                     "umoe_tiled_gemm_tn: k_split needs static windows and groups that are whole rows of ONE dense output (n == ldo)");
        UMOE_REQUIRE(((long)gs[i].out_row_base + gs[i].m) * (long)a->ldo <= a->part_stride, "umoe_tiled_gemm_tn: part_stride smaller than the output");
    }
    long covered = 0;
    for (int i = 0; i < a->num_groups; ++i) covered += (long)gs[i].m * a->ldo;
    UMOE_REQUIRE(covered == a->part_stride, "umoe_tiled_gemm_tn: k_split: the groups must cover the whole output slab (part_stride = %ld, covered %ld)", a->part_stride, covered);
    b.groups = gs;
    b.out = a->ws;
    if (int rc = launch_tn<true>(&b, max_m, max_n, 0, ks, s)) return rc;
    const long n4 = a->part_stride / 4;
    tn_reduce_kernel<<<dim3((unsigned)((n4 + 255) / 256)), 256, 0, s>>>(reinterpret_cast<const float*>(a->ws), a->part_stride, ks, n4, reinterpret_cast<uint16_t*>(a->out));
    UMOE_LAUNCH_CHECK();
    return 0;
}
