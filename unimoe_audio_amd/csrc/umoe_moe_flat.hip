// The two expert GEMMs of a dense decode layer (8 routed + 2 shared experts, <= 16 rows) in ONE launch of ONE workgroup per CU, with a
// STATIC SCHEDULE that balances the bytes every CU takes in (decode engine only; replaces core.py:406-416,34-49,344-351 for the
// decode shape, like moe_fused_kernel of umoe_gemm.hip whose tiles, K split and reduction order it keeps: bit-identical outputs).
//
// Why.  A weight-streaming workgroup takes in ~26 GB/s (~10.5 B/clk/CU: MI355X_MICROARCH.md "global_load_dwordx4 (HBM-bound)"), so a
// launch of one 8-wave workgroup per CU ends when its HEAVIEST CU has taken in its bytes: every timing of round 1/2 fits bytes-per-CU /
// 26 GB/s (7 pairs = 896 KiB -> 35-37 us; 8 pairs -> 43 us; two 512 KiB workgroups on one CU -> 39 us; 7 pairs + 6 down blocks =
// 1412 KiB -> 57 us).  The box grid of moe_fused_kernel gives 226 of the 256 CUs 7 gate/up pairs + 6 down blocks and leaves 30 idle;
// spread evenly the layer's 304 MB are 1161 KiB per CU.  Here every CU gets a slice of the FLAT list of gate/up pairs (5-7 pairs,
// possibly straddling two experts) and, behind it, a slice of ONE expert's down projection (1-6 blocks), both from a table the host
// computes once (flat_plan): the slices are sized so that every workgroup ends at the same time under a small timing model that knows
// that a down slice can start only when ALL producers of its expert's rows have published (the "seam" of that expert).
// The router riders (one token each: RMSNorm + gate GEMV + Top-P chain, umoe_router_dev.h) are the first S workgroups themselves:
// they route their row, then take a (lighter) slice like everybody else -- no extra workgroups, all n_wg are resident at once.  Nobody
// waits for them: every workgroup normalises the 16 rows itself while its first weight chunk is in flight (flat_gateup).
//
// Hand-offs as in moe_fused_kernel (cdna_hip_programming.md Guideline 16 R1): write-through payload, every storing wave drains, one
// flag per part, relaxed agent-scope poll (bounded, sticky error word), every load of handed-over bytes an sc1 load.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "umoe_common.h"
#include "umoe_router_dev.h"

#define FLAT_MAXG UMOE_GROUPS_INLINE
#define FLAT_MAXWG 256
#define FLAT_NP_MIN 4
#define FLAT_NP_MAX 7
#define FLAT_ND_MAX2 6      // blocks of one down slice, 2-step chunks (even number of k-steps: the routed experts)
#define FLAT_ND_MAX1 10     // ... 1-step chunks (odd number of k-steps: the shared experts, half the bytes per block)
#define FLAT_SLICES 2       // down slices per workgroup

struct flat_args {
    // (what the first weight request needs sits at the front: one batch of kernel-argument loads)
    const uint16_t* a;                  // raw rows x1 [S][lda] (the residual stream after attention); every workgroup normalises them itself
    const uint16_t* norm_w;             // post-attention RMSNorm weights [D]
    float rms_eps;
    int lda, S, G;                      // row stride, rows (<= 16), groups
    int pair0[FLAT_MAXG + 1];           // first flat pair of gate/up group i (pair0[G] = all pairs)
    const uint16_t* w_gu[FLAT_MAXG];    // WP16 gate/up weights (blocks interleaved) per group
    uint16_t* h;                        // silu(g)*u rows [.][ldh]: written by the gate/up phase, read by the down phase of this launch
    uint16_t* y;                        // down-projection outputs [.][ldy]
    uint32_t* flags;                    // one word per workgroup: its gate/up slice is published
    unsigned long long* dbg;            // diagnostics only (NULL otherwise): [workgroup][16] wall-clock stamps (100 MHz)
    int ldh, ldy, kb_gu;                // k-steps (K / 32) of the gate/up GEMMs (64: see flat_gateup)
    const uint16_t* w_dn[FLAT_MAXG];    // WP16 down weights per group
    int h_row[FLAT_MAXG];               // gate/up group i writes rows h_row[i] + r of h
    int dn_kb[FLAT_MAXG];               // k-steps of down group i
    int dn_a_row[FLAT_MAXG];            // down group i reads rows dn_a_row[i] + r of h ...
    int dn_y_row[FLAT_MAXG];            // ... and writes rows dn_y_row[i] + r of y
    int dn_nb[FLAT_MAXG];               // 16-feature blocks of down group i
    int prod_base[FLAT_MAXG];           // producers of down group i's rows: workgroups [prod_base, prod_base + prod_n)
    int prod_n[FLAT_MAXG];
    uint32_t gu[FLAT_MAXWG];            // per workgroup: first flat pair | pairs << 11 | (rider token + 1) << 16 (0: not a rider)
    uint32_t dn[FLAT_MAXWG];            // per workgroup: TWO down slices, 16 bits each (low half first): group | first block << 4 | blocks << 12
};                                      //   (blocks 0 = no slice; see FLAT_ND_*)

__device__ __forceinline__ int flat_lds_chunk_off(int QS, int h, int i, int m) {
    // 16-byte chunk i of K-quarter h, row m: 256-byte segments, slot rotated by the row index (umoe_gemm.hip lds_chunk_off)
    return h * QS + (i >> 4) * 256 + (((i & 15) + m) & 15) * 16;
}

// bounded wait of ONE lane for an epoch flag; `code` lands in the sticky error word only on this lane's OWN timeout (a word that is
// already set -- an earlier cause, e.g. an expert-parallel receive -- ends the wait and is kept)
__device__ __forceinline__ void flat_wait(uint32_t* flag, uint32_t epoch, uint32_t* err_word, uint32_t code) {
    umoe_gu32* f = reinterpret_cast<umoe_gu32*>(reinterpret_cast<uintptr_t>(flag));
    umoe_gu32* err = reinterpret_cast<umoe_gu32*>(reinterpret_cast<uintptr_t>(err_word));
    const unsigned long long t0 = wall_clock64();
    for (unsigned spins = 0;; ++spins) {
        if ((int32_t)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) >= 0) break;
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 1023u) == 1023u) {
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            if (wall_clock64() - t0 > 200000000ull) {      // 2 s
                uint32_t zero = 0u;
                __hip_atomic_compare_exchange_strong(err, &zero, code, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
}

// diagnostics: thread 0 keeps up to 16 stamps in registers and stores them at exit (scalar branch on a kernel argument)
// (the instrumented build only -- make tl, scripts/flat_timeline.py; the product kernel carries no stamp)
#ifdef UMOE_TIMELINE
struct flat_stamps { unsigned long long t[16]; };
#define FSTAMP(k) do { if (A.dbg) st.t[k] = wall_clock64(); } while (0)
#else
struct flat_stamps {};
#define FSTAMP(k) do { } while (0)
#endif

typedef uint32_t flat_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t flat_u32x2 __attribute__((ext_vector_type(2)));

// ---- gate/up SwiGLU slice: NP pairs of the flat list starting at fp0 (arithmetic of wstream_body<14, 1, PLAIN, SWIGLU, 8> per tile) ----
// The workgroup normalises the 16 rows ITSELF while it stages them (post-attention RMSNorm, model.py:240): the raw rows exist when the
// launch starts, so nothing is waited for -- rows requested first, the weight stream right behind them, and the ~2 us of row arithmetic
// hide under the first chunk's flight.  The sum of squares follows the router body's tree (umoe_router_dev.h router4_body: lane l of wave
// h sums the 8 squares of chunk 64 h + l, xor-butterfly 32 .. 1, the four wave sums added in order), so the rows are bit-identical to
// the rows the router launches write.  K = 2048 only (one staging round, thread (row m, sub) holds chunks sub and sub + 32 of a quarter).
__device__ __forceinline__ uint32_t flat_epoch(const umoe_rider_pub& pub) {
    // the step word lives in device memory: read it where it is first needed (a load the compiler may not hoist in front of the
    // kernel's first weight request -- it was one more dependent scalar round trip there)
    asm volatile("" ::: "memory");
    return __builtin_nontemporal_load(pub.step) * (uint32_t)pub.layers + (uint32_t)pub.layer + 1u;
}

template <int NP>
__device__ __forceinline__ void flat_gateup(const flat_args& A, const umoe_rider_pub& pub, const int fp0, const unsigned b, char* smem, flat_stamps& st) {
    constexpr int NT = 2 * NP, WV = 8, KB = 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: every guard around an MFMA is a scalar branch
    constexpr int QS = (KB * 16 + 255) & ~255, RS = QS * 4;
    const int i0 = __builtin_amdgcn_readfirstlane((KB * wave) / WV), i1 = __builtin_amdgcn_readfirstlane((KB * (wave + 1)) / WV);
    // the slice straddles at most two groups (a group has far more than 7 pairs).  Every table read is a kernel-argument read with a
    // compile-time offset + a scalar select: ONE batch of scalar loads in front of the first request, no dependent second one
    int g0 = 0;
#pragma unroll
    for (int i = 1; i < FLAT_MAXG; ++i) g0 += (fp0 >= A.pair0[i] && i < A.G) ? 1 : 0;      // (pair0 ascends: the count IS the index)
    const int p0 = A.pair0[g0], cut = A.pair0[g0 + 1];
    const uint16_t *wg0 = A.w_gu[g0], *wg1 = A.w_gu[g0 + 1 < FLAT_MAXG ? g0 + 1 : g0];
    const int g1 = min(g0 + 1, A.G - 1);      // pairs >= cut belong to group g1
    f32x4_t acc[NT];
    const flat_u32x4* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const int pp = fp0 + (t >> 1);
        const bool second = pp >= cut;
        const int lp = pp - (second ? cut : p0);
        wp[t] = reinterpret_cast<const flat_u32x4*>(second ? wg1 : wg0) + ((size_t)(2 * lp + (t & 1)) * KB) * 64 + lane;
    }
    flat_u32x4 w0[NT], w1[NT];
    auto load_chunk = [&](flat_u32x4 (&dst)[NT], int ii) {
        const int ic = min(ii, i1 - 1);
#pragma unroll
        for (int t = 0; t < NT; ++t) dst[t] = __builtin_nontemporal_load(wp[t] + (size_t)ic * 64);
    };
    const int count = A.S;
    // ---- rows first (they exist: the previous launch wrote them), then the weight stream; normalise while the chunk flies ----
    {
        constexpr int TPR = WV * 4;      // threads per row
        const int m = tid / TPR, sub = tid % TPR;
        const bool valid = m < count;
        const uint16_t* src = A.a + (size_t)(valid ? m : 0) * A.lda;
        char* dst = smem + m * RS;
        char* nw_lds = smem + 16 * RS;
        uint4 buf[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) buf[n] = ld16(src + ((n >> 1) * KB + sub + TPR * (n & 1)) * 8);
        // (straight-line loads only: a branch around a load makes hipcc wait for vmcnt(0), i.e. for the weight chunk behind the rows)
        const uint4 nw1 = ld16(A.norm_w + (tid & (4 * KB - 1)) * 8);
        // (measured and rejected: BOTH register stages requested here.  A CU issues about 1 KiB of vector loads per 100 cycles, so the
        //  second stage's 14 requests per wave only delayed the point where the rows are staged -- 8.6 -> 12.1 us -- and bought nothing:
        //  3.020 vs 3.015 ms/step.  The launch runs at the CU's request rate from its first request on.)
        __builtin_amdgcn_sched_barrier(0);
        load_chunk(w0, i0);
        __builtin_amdgcn_sched_barrier(0);
        FSTAMP(1);
        st16(nw_lds + (tid & (4 * KB - 1)) * 16, nw1);      // (both halves of the workgroup store the same 4 KiB)
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
                for (int j = 0; j < 8; ++j) cs += f[j] * f[j];
                c2[k2] = cs;
            }
            float v = c2[0] + c2[1];
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
            q4[hq] = v;
        }
        const float ss = ((q4[0] + q4[1]) + q4[2]) + q4[3];
        const float rs = rsqrtf(ss / (float)(KB * 32) + A.rms_eps);
        __syncthreads();
        FSTAMP(2);
        // keep the row slice packed (32 registers) between the sum of squares and the scaling
#pragma unroll
        for (int n = 0; n < 8; ++n) asm volatile("" : "+v"(buf[n].x), "+v"(buf[n].y), "+v"(buf[n].z), "+v"(buf[n].w));
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const int h = n >> 1, i = sub + TPR * (n & 1);
            float f[8], w[8];
            unpack8(buf[n], f);
            unpack8(*reinterpret_cast<const uint4*>(nw_lds + (h * KB + i) * 16), w);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = w[j] * rbf(f[j] * rs);
            if (valid) st16(dst + flat_lds_chunk_off(QS, h, i, m), pack8(f));
        }
    }
    __syncthreads();
    FSTAMP(3);
    // ---- stream: 1-step chunks, double-buffered in registers, the 8 waves split K ----
    const int h = lane >> 4, mm = lane & 15;
    const char* bbase = smem + mm * RS;
    auto compute_chunk = [&](const flat_u32x4 (&src)[NT], int ii) {
        if (ii < i1) {
            const uint4 bv = *reinterpret_cast<const uint4*>(bbase + flat_lds_chunk_off(QS, h, ii, mm));
            const bf16x8_t bfrag = __builtin_bit_cast(bf16x8_t, bv);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, src[t]), bfrag, acc[t], 0, 0, 0);
        }
    };
    for (int i = i0; i < i1; i += 2) {
        if (i + 1 < i1) load_chunk(w1, i + 1);
        compute_chunk(w0, i);
        if (i + 2 < i1) load_chunk(w0, i + 2);
        if (i + 1 < i1) compute_chunk(w1, i + 1);
    }
    // ---- fixed-order cross-wave reduction through the (now free) staging area ----
    FSTAMP(4);
    __syncthreads();
    FSTAMP(5);
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
    // ---- SwiGLU epilogue: lane (h, mm) owns features 4h..4h+3 of row mm; pairs spread over the waves; write-through stores ----
    const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(A.h, 0, 0x7fffffff, 0x00020000);
    for (int q = wave; q < NP; q += WV) {
        const int pp = fp0 + q;
        const int grp = pp >= cut ? g1 : g0;
        const int col = (pp - A.pair0[grp]) * 16 + 4 * h;
        const long orow = (long)A.h_row[grp] + mm;
        const f32x4_t ga = reduced(2 * q), ua = reduced(2 * q + 1);
        uint16_t yv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gt = rbf(ga[j]);
            const float up = rbf(ua[j]);
            const float si = rbf(gt / (1.0f + expf(-gt)));
            yv[j] = f2bf(si * up);
        }
        const flat_u32x2 v2 = {(uint32_t)yv[0] | ((uint32_t)yv[1] << 16), (uint32_t)yv[2] | ((uint32_t)yv[3] << 16)};
        if (mm < count) __builtin_amdgcn_raw_buffer_store_b64(v2, orsrc, (int)((orow * A.ldh + col) * 2), 0, 16);
    }
    // publish: every storing wave drains, the workgroup meets, one lane raises this workgroup's flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
        __hip_atomic_store(reinterpret_cast<umoe_gu32*>(reinterpret_cast<uintptr_t>(A.flags + b)), flat_epoch(pub), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    FSTAMP(6);
}

// ---- down-projection slice: blocks [nb0, nb0 + ND) of group grp (arithmetic of wstream_body<6, 2, PLAIN, BF16, 8> per tile) ----
// U = 2 for an even number of k-steps (whole 2-step chunks per wave), U = 1 for an odd one: the K split of the 2-step launch does not
// depend on U then, and a 1-step stream has no clamped duplicate step at the end of a wave's slice.
template <int ND, int U>
__device__ __forceinline__ void flat_down(const flat_args& A, const umoe_rider_pub& pub, const int grp, const int nb0, char* smem, flat_stamps& st,
                                          const int sb) {
    constexpr int NT = ND, WV = 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KB = A.dn_kb[grp];
    const int QS = (KB * 16 + 255) & ~255, RS = QS * 4;
    int i0, i1;
    if (KB % 2 == 0) {      // whole 2-step chunks per wave when the slice divides (U == 2 here)
        const int units = KB / 2;
        i0 = 2 * ((units * wave) / WV);
        i1 = 2 * ((units * (wave + 1)) / WV);
    } else {
        i0 = (KB * wave) / WV;
        i1 = (KB * (wave + 1)) / WV;
    }
    i0 = __builtin_amdgcn_readfirstlane(i0);
    i1 = __builtin_amdgcn_readfirstlane(i1);
    f32x4_t acc[NT];
    const flat_u32x4* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        wp[t] = reinterpret_cast<const flat_u32x4*>(A.w_dn[grp]) + ((size_t)(nb0 + t) * KB) * 64 + lane;
    }
    flat_u32x4 w0[NT][U], w1[NT][U];
    auto load_chunk = [&](flat_u32x4 (&dst)[NT][U], int ibase) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ii = min(ibase + u, i1 - 1);
#pragma unroll
            for (int t = 0; t < NT; ++t) dst[t][u] = __builtin_nontemporal_load(wp[t] + (size_t)ii * 64);
        }
    };
    if (i0 < i1) load_chunk(w0, i0);
    const int count = A.S;
    // wait for the workgroups of THIS launch that produced this group's rows: lane i of wave 0 polls producer i (bounded)
    if (tid < A.prod_n[grp]) flat_wait(A.flags + A.prod_base[grp] + tid, flat_epoch(pub), pub.err, 3u);
    __syncthreads();
    FSTAMP(sb);
    {
        constexpr int TPR = WV * 4;
        const int m = tid / TPR, sub = tid % TPR;
        const bool valid = m < count;
        const long arow = (long)A.dn_a_row[grp] + (valid ? m : 0);
        char* dst = smem + m * RS;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(A.h, 0, 0x7fffffff, 0x00020000);
        for (int ib0 = 0; ib0 < KB; ib0 += 4 * TPR) {
            uint4 buf[16];
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const int h = n >> 2, i = min(ib0 + sub + TPR * (n & 3), KB - 1);
                const flat_u32x4 t4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((arow * A.ldh + (h * KB + i) * 8) * 2), 0, 16);
                buf[n] = make_uint4(t4[0], t4[1], t4[2], t4[3]);
            }
            if (ib0 == 0) {     // the second register stage right behind the rows (returns: first stage, rows, second stage)
                __builtin_amdgcn_sched_barrier(0);
                if (i0 + U < i1) load_chunk(w1, i0 + U);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const int h = n >> 2, i = ib0 + sub + TPR * (n & 3);
                if (valid && i < KB) st16(dst + flat_lds_chunk_off(QS, h, i, m), buf[n]);
            }
        }
    }
    __syncthreads();
    FSTAMP(sb + 1);
    const int h = lane >> 4, mm = lane & 15;
    const char* bbase = smem + mm * RS;
    auto compute_chunk = [&](const flat_u32x4 (&src)[NT][U], int ibase) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ii = ibase + u;
            if (ii < i1) {
                const uint4 bv = *reinterpret_cast<const uint4*>(bbase + flat_lds_chunk_off(QS, h, ii, mm));
                const bf16x8_t bfrag = __builtin_bit_cast(bf16x8_t, bv);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, src[t][u]), bfrag, acc[t], 0, 0, 0);
            }
        }
    };
    for (int i = i0; i < i1; i += 2 * U) {
        if (i + U < i1 && i != i0) load_chunk(w1, i + U);
        compute_chunk(w0, i);
        if (i + 2 * U < i1) load_chunk(w0, i + 2 * U);
        if (i + U < i1) compute_chunk(w1, i + U);
    }
    FSTAMP(sb + 2);
    __syncthreads();
    FSTAMP(sb + 3);
    f32x4_t* red = reinterpret_cast<f32x4_t*>(smem);
#pragma unroll
    for (int t = 0; t < NT; ++t) red[(wave * NT + t) * 64 + lane] = acc[t];
    __syncthreads();
    if (mm >= count) return;
    // tile t is finished by wave t % 8
    for (int t = wave; t < NT; t += WV) {
        f32x4_t s = red[t * 64 + lane];
#pragma unroll
        for (int w = 1; w < WV; ++w) {
            const f32x4_t v = red[(w * NT + t) * 64 + lane];
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        }
        const int n = (nb0 + t) * 16 + 4 * h;
        uint16_t yv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) yv[j] = f2bf(rbf(s[j] + 0.f));      // (+ 0.f: the bias slot of the generic epilogue; -0 -> +0 like there)
        uint16_t* o = A.y + ((long)A.dn_y_row[grp] + mm) * A.ldy + n;
        *reinterpret_cast<uint2*>(o) = make_uint2((uint32_t)yv[0] | ((uint32_t)yv[1] << 16), (uint32_t)yv[2] | ((uint32_t)yv[3] << 16));
    }
}

#define FLAT_RIDER_LDS 512      // bytes of LDS behind the GEMM area for the riders' partial sums (router4_body: 4 + 4 * 16 floats)

__global__ __launch_bounds__(512, 1) void moe_flat_kernel(const flat_args A, const umoe_router_args ra, const umoe_rider_pub pub, const int lds_gemm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned b = blockIdx.x;
    const unsigned eg = A.gu[b], ed = A.dn[b];
    flat_stamps st;
#ifdef UMOE_TIMELINE
    if (A.dbg) {
#pragma unroll
        for (int k = 0; k < 16; ++k) st.t[k] = 0;
        st.t[0] = wall_clock64();
    }
#endif
    const int token = (int)(eg >> 16) - 1;
    if (token >= 0) {
        // rider: the Top-P router of row `token` (its own RMSNorm + gate GEMV on waves 0..3, then wave 0 alone walks the serial chain while
        // the other waves go on to the GEMM).  Nobody in this launch waits for it: its tables feed the combine of a LATER launch.
        // Waves 4..7 only keep the two barriers of router4_body company.
        float* rl = reinterpret_cast<float*>(smem + lds_gemm);
        if (threadIdx.x < 256) {
#ifdef UMOE_TIMELINE
            TL_ENTER(5);
#endif
            if (ra.logits_bf16) router4_body<9, 2, 1, false>(ra, token, threadIdx.x, rl TL_PASS, nullptr, 0u, nullptr);
            else router4_body<9, 2, 0, false>(ra, token, threadIdx.x, rl TL_PASS, nullptr, 0u, nullptr);
        } else {
            __syncthreads();
            __syncthreads();
        }
    }
    const int fp0 = (int)(eg & 2047u), np = (int)((eg >> 11) & 7u);
    switch (np) {
        case 4: flat_gateup<4>(A, pub, fp0, b, smem, st); break;
        case 5: flat_gateup<5>(A, pub, fp0, b, smem, st); break;
        case 6: flat_gateup<6>(A, pub, fp0, b, smem, st); break;
        case 7: flat_gateup<7>(A, pub, fp0, b, smem, st); break;
        default: break;
    }
    for (int sl = 0; sl < FLAT_SLICES; ++sl) {
        const unsigned e16 = (ed >> (16 * sl)) & 0xffffu;
        const int nd = (int)(e16 >> 12), grp = (int)(e16 & 15u), nb0 = (int)((e16 >> 4) & 255u);
        if (nd == 0) break;
        __syncthreads();     // (the reduction slab of the previous GEMM is the staging area of this one)
        if (A.dn_kb[grp] & 1) {
            switch (nd) {
                case 1: flat_down<1, 1>(A, pub, grp, nb0, smem, st, 7 + 4 * sl); break;
                case 2: flat_down<2, 1>(A, pub, grp, nb0, smem, st, 7 + 4 * sl); break;
                case 3: flat_down<3, 1>(A, pub, grp, nb0, smem, st, 7 + 4 * sl); break;
                case 4: flat_down<4, 1>(A, pub, grp, nb0, smem, st, 7 + 4 * sl); break;
                case 5: flat_down<5, 1>(A, pub, grp, nb0, smem, st, 7 + 4 * sl); break;
                case 6: flat_down<6, 1>(A, pub, grp, nb0, smem, st, 7 + 4 * sl); break;
                case 7: flat_down<7, 1>(A, pub, grp, nb0, smem, st, 7 + 4 * sl); break;
                case 8: flat_down<8, 1>(A, pub, grp, nb0, smem, st, 7 + 4 * sl); break;
                case 9: flat_down<9, 1>(A, pub, grp, nb0, smem, st, 7 + 4 * sl); break;
                default: flat_down<10, 1>(A, pub, grp, nb0, smem, st, 7 + 4 * sl); break;
            }
        } else {
            switch (nd) {
                case 1: flat_down<1, 2>(A, pub, grp, nb0, smem, st, 7 + 4 * sl); break;
                case 2: flat_down<2, 2>(A, pub, grp, nb0, smem, st, 7 + 4 * sl); break;
                case 3: flat_down<3, 2>(A, pub, grp, nb0, smem, st, 7 + 4 * sl); break;
                case 4: flat_down<4, 2>(A, pub, grp, nb0, smem, st, 7 + 4 * sl); break;
                case 5: flat_down<5, 2>(A, pub, grp, nb0, smem, st, 7 + 4 * sl); break;
                default: flat_down<6, 2>(A, pub, grp, nb0, smem, st, 7 + 4 * sl); break;
            }
        }
    }
#ifdef UMOE_TIMELINE
    if (A.dbg && threadIdx.x == 0) {
        st.t[15] = wall_clock64();
#pragma unroll
        for (int k = 0; k < 16; ++k) A.dbg[(size_t)b * 16 + k] = st.t[k];
    }
#endif
}

// ------------------------------------------------------------------------------------ host: the static schedule
// Model (units: KiB a workgroup takes in; 1 KiB ~ 38 ns at 26 GB/s): a pair costs 2 * kb KiB, a down block kb KiB, a rider loses
// `rider` KiB in front of its slice, a down slice `stage` KiB for the wait + the rows, a published slice is seen `flagc` KiB later.
struct FlatPlan {
    bool ok = false;
    int n_wg = 0;
    uint32_t gu[FLAT_MAXWG], dn[FLAT_MAXWG];
    int prod_base[FLAT_MAXG], prod_n[FLAT_MAXG];
    double makespan = 0, mean = 0;      // model time of the slowest workgroup / mean useful KiB per workgroup
};

struct FlatShape {
    int G, S, n_wg, kb_gu;
    int pairs[FLAT_MAXG], dn_nb[FLAT_MAXG], dn_kb[FLAT_MAXG], dn_src[FLAT_MAXG];
    int rider_less, heavy_at;           // experiment knobs: -1 = search
    double rider, stage, flagc, pair_scale;
    bool operator==(const FlatShape& o) const { return memcmp(this, &o, sizeof(*this)) == 0; }
};

// List scheduling of the down slices for a target makespan T.  Groups are taken in the order `ex` (big blocks first, by ascending
// seam; the small-block groups last: they fill the gaps); a slice goes to the workgroup that can START it first (free and the
// group's seam reached) and takes as many blocks as end in front of T.  A workgroup takes at most FLAT_SLICES slices.
struct FlatSlices { int n[FLAT_MAXWG]; int grp[FLAT_MAXWG][FLAT_SLICES], nb0[FLAT_MAXWG][FLAT_SLICES], nd[FLAT_MAXWG][FLAT_SLICES]; };
static bool flat_assign(const FlatShape& sh, const double* avail, const double* seam, const int* ex, double T, FlatSlices& o) {
    const int n = sh.n_wg;
    double fre[FLAT_MAXWG];
    for (int j = 0; j < n; ++j) { fre[j] = avail[j]; o.n[j] = 0; }
    for (int q = 0; q < sh.G; ++q) {
        const int i = ex[q];
        const double se = seam[sh.dn_src[i]], cb = (double)sh.dn_kb[i];
        const int ndmax = (sh.dn_kb[i] & 1) ? FLAT_ND_MAX1 : FLAT_ND_MAX2;
        int left = sh.dn_nb[i], next = 0;
        while (left > 0) {
            int bj = -1;
            double bs = 1e30;
            for (int j = 0; j < n; ++j) {
                if (o.n[j] >= FLAT_SLICES) continue;
                const double st = std::max(fre[j], se);
                if (st < bs && st + sh.stage + cb <= T) { bs = st; bj = j; }
            }
            if (bj < 0) return false;
            int nd = (int)((T - bs - sh.stage) / cb);
            nd = std::min(std::min(nd, ndmax), left);
            const int k = o.n[bj]++;
            o.grp[bj][k] = i; o.nb0[bj][k] = next; o.nd[bj][k] = nd;
            fre[bj] = bs + sh.stage + nd * cb;
            next += nd; left -= nd;
        }
    }
    return true;
}

static void flat_plan(const FlatShape& sh, FlatPlan& out) {
    out.ok = false;
    out.n_wg = sh.n_wg;
    const int n = sh.n_wg, G = sh.G, S = sh.S;
    if (n < 1 || n > FLAT_MAXWG || G > FLAT_MAXG || S > n) return;
    int P = 0, pair0[FLAT_MAXG + 1];
    for (int i = 0; i < G; ++i) { pair0[i] = P; P += sh.pairs[i]; }
    pair0[G] = P;
    if (P >= 2048) return;
    const double cp = 2.0 * sh.kb_gu;
    double total = P * cp + S * sh.rider + n * sh.stage;
    int kb_big = 0;
    for (int i = 0; i < G; ++i) { total += (double)sh.dn_nb[i] * sh.dn_kb[i]; kb_big = std::max(kb_big, sh.dn_kb[i]); }
    FlatPlan best;
    double best_T = 1e30;
    std::vector<int> np(n), fp0(n);
    std::vector<double> avail(n);
    FlatSlices sl;
    // The riders are workgroups 0 .. S-1 (`base` or `base - 1` pairs); the workgroups with one pair more than `base` form ONE block at
    // position `h0` of the flat order: groups whose producers include a late workgroup have a late seam.  Candidate positions: right
    // behind the riders, and every position that starts or ends the block on a group boundary.
    // (riders with one pair less measured best on MI355X -- 3.004-3.015 vs 3.026 ms/step; the other form only where that one has no schedule)
    for (int rider_less = 1; rider_less >= 0; --rider_less) {
        if (sh.rider_less >= 0 && rider_less != sh.rider_less) continue;
        if (best.ok) break;
        const int adj = rider_less ? S : 0;
        const int base = (P + adj) / n;
        const int extra = P + adj - base * n;
        const int np_max = extra ? base + 1 : base, np_min = rider_less ? base - 1 : base;
        if (np_max > FLAT_NP_MAX || np_min < FLAT_NP_MIN || extra > n - S) continue;
        std::vector<int> cand;
        cand.push_back(S);
        cand.push_back(n - extra);
        for (int g = 1; g < G; ++g) {
            // block [h0, h0 + extra) starts at the first workgroup whose pairs begin at or behind pair0[g] (with `base` pairs in front) ...
            const int rp = S * (base - rider_less);
            if (pair0[g] >= rp) {
                const int h0 = S + (pair0[g] - rp + base - 1) / base;
                cand.push_back(h0);
                cand.push_back(h0 - extra);      // ... or ends there
                cand.push_back(h0 - extra - 1);
                cand.push_back(h0 + 1);
            }
        }
        for (int h0 : cand) {
            if (h0 < S || h0 + extra > n) continue;
            if (sh.heavy_at >= 0 && h0 != std::min(std::max(sh.heavy_at, S), n - extra)) continue;
            for (int j = 0; j < n; ++j) np[j] = (j < S) ? base - rider_less : ((j >= h0 && j < h0 + extra) ? base + 1 : base);
            int acc = 0;
            for (int j = 0; j < n; ++j) { fp0[j] = acc; acc += np[j]; }
            if (acc != P) break;
            // (pair_scale < 1: the weight stream is HBM-bound and shared in proportion to what a workgroup has in flight, so a slice of
            //  7 pairs ends its gate/up phase hardly later than one of 6: measured, scripts/flat_timeline.py)
            for (int j = 0; j < n; ++j) avail[j] = (j < S ? sh.rider : 0.0) + base * cp + (np[j] - base) * cp * sh.pair_scale;
            double seam[FLAT_MAXG];
            int pb[FLAT_MAXG], pn[FLAT_MAXG], ex[FLAT_MAXG];
            bool fits = true;
            for (int g = 0; g < G; ++g) {
                int lo = -1, hi = -1;
                double sm = 0;
                for (int j = 0; j < n; ++j)
                    if (fp0[j] < pair0[g + 1] && fp0[j] + np[j] > pair0[g]) {
                        if (lo < 0) lo = j;
                        hi = j;
                        sm = std::max(sm, avail[j]);
                    }
                if (lo < 0 || hi - lo + 1 > 64) fits = false;
                pb[g] = lo; pn[g] = hi - lo + 1;
                seam[g] = sm + sh.flagc;
            }
            if (!fits) continue;
            for (int i = 0; i < G; ++i) ex[i] = i;
            std::stable_sort(ex, ex + G, [&](int x, int y) {
                const bool bx = sh.dn_kb[x] == kb_big, by = sh.dn_kb[y] == kb_big;
                if (bx != by) return bx;
                return seam[sh.dn_src[x]] < seam[sh.dn_src[y]];
            });
            double lo = total / n - 1.0, hi = std::min(best_T, 3.0 * total / n + 4096.0);
            if (!flat_assign(sh, avail.data(), seam, ex, hi, sl)) continue;     // cannot beat the best so far
            for (int it = 0; it < 24 && hi - lo > 0.5; ++it) {
                const double mid = 0.5 * (lo + hi);
                if (flat_assign(sh, avail.data(), seam, ex, mid, sl)) hi = mid;
                else lo = mid;
            }
            if (hi >= best_T) continue;
            flat_assign(sh, avail.data(), seam, ex, hi, sl);
            best_T = hi;
            best.ok = true;
            best.n_wg = n;
            for (int j = 0; j < n; ++j) {
                best.gu[j] = (uint32_t)fp0[j] | ((uint32_t)np[j] << 11) | (j < S ? (uint32_t)(j + 1) << 16 : 0u);
                uint32_t d = 0;
                for (int k = 0; k < sl.n[j]; ++k) d |= ((uint32_t)sl.grp[j][k] | ((uint32_t)sl.nb0[j][k] << 4) | ((uint32_t)sl.nd[j][k] << 12)) << (16 * k);
                best.dn[j] = d;
            }
            for (int i = 0; i < G; ++i) { best.prod_base[i] = pb[sh.dn_src[i]]; best.prod_n[i] = pn[sh.dn_src[i]]; }
            best.makespan = hi;
            best.mean = total / n;
        }
    }
    if (best.ok) out = best;
}

extern "C" int umoe_moe_flat_plan_probe(int n_wg, int S, int D, int I_dyn, int I_sh, int n_real, int n_fix, double* out, int out_len);
static unsigned long long* g_flat_dbg = nullptr;
// diagnostics: copy the stamps of the last stamped launch to the host ([256][16] u64); -1 when none was taken
extern "C" int umoe_moe_flat_stamps(unsigned long long* host_out) {
    if (!host_out) {       // enable (outside any stream capture): later launches of the instrumented build write their stamps
        if (!g_flat_dbg && hipMalloc(&g_flat_dbg, sizeof(unsigned long long) * 16 * FLAT_MAXWG) != hipSuccess) return -2;
        return hipMemset(g_flat_dbg, 0, sizeof(unsigned long long) * 16 * FLAT_MAXWG) == hipSuccess ? 0 : -2;
    }
    if (!g_flat_dbg) return -1;
    return hipMemcpy(host_out, g_flat_dbg, sizeof(unsigned long long) * 16 * FLAT_MAXWG, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
static double flat_env(const char* name, double dflt) {
    const char* v = getenv(name);
    return v ? atof(v) : dflt;
}
static void flat_knobs(FlatShape& sh) {
    // model constants (KiB of weight stream a workgroup forgoes; calibrated on MI355X) and experiment knobs
    sh.rider = flat_env("UMOE_FLAT_RIDER_KIB", 110.0);
    sh.stage = flat_env("UMOE_FLAT_STAGE_KIB", 40.0);
    sh.flagc = flat_env("UMOE_FLAT_FLAG_KIB", 0.0);
    sh.pair_scale = flat_env("UMOE_FLAT_PAIR_SCALE", 0.3);
    sh.rider_less = (int)flat_env("UMOE_FLAT_RIDER_LESS", -1.0);
    sh.heavy_at = (int)flat_env("UMOE_FLAT_HEAVY_AT", -1.0);
}

// Returns 0 (launched), 1 (shapes / CU count do not allow it: nothing launched), < 0 error.
int umoe_moe_flat(const umoe_gemm_args* gu, const umoe_gemm_args* dn, uint32_t* flags, int flag_words, int n_wg, hipStream_t s) {
    UMOE_REQUIRE(gu && dn && flags, "umoe_moe_flat: null argument");
    const int G = gu->num_groups;
    if (!(gu->fused_router && gu->rider_pub && gu->groups_host && dn->groups_host && G == dn->num_groups && G <= FLAT_MAXG && gu->prologue == UMOE_PRO_PLAIN &&
          gu->epilogue == UMOE_EPI_SWIGLU && dn->prologue == UMOE_PRO_PLAIN && dn->epilogue == UMOE_EPI_BF16 && gu->ksplit <= 1 && dn->ksplit <= 1 &&
          gu->max_rows <= 16 && dn->max_rows <= 16 && dn->a == gu->out && dn->lda == gu->ldo && !dn->fused_router && gu->max_k % 32 == 0 &&
          dn->max_k % 32 == 0 && (gu->lda & 7) == 0 && (gu->ldo & 7) == 0 && (dn->ldo & 3) == 0 && n_wg >= 1 && n_wg <= FLAT_MAXWG && n_wg <= flag_words))
        return 1;
    const umoe_router_args* r = gu->fused_router;
    if (!(r->S >= 1 && r->S <= 16 && r->n_dyn == 9 && r->n_fix == 2 && r->D == 2048 && r->x && r->gate_w && r->expert_mask && !r->logits_in &&
          !r->norm_only && r->norm_w && !r->gumbel && !r->x_noise && !r->attn_mask))
        return 1;
    FlatShape sh;
    memset(&sh, 0, sizeof(sh));
    sh.G = G; sh.S = r->S; sh.n_wg = n_wg; sh.kb_gu = gu->groups_host[0].k / 32;
    for (int i = 0; i < G; ++i) {
        const umoe_group_t& a = gu->groups_host[i];
        const umoe_group_t& b = dn->groups_host[i];
        if (a.rows || a.count || a.row_off || a.a_row_base || a.a_col_off || a.static_count != r->S || a.bias || (a.n_blocks & 1) || a.k != gu->groups_host[0].k ||
            a.k != r->D || b.rows || b.count || b.row_off || b.a_col_off || b.static_count != r->S || b.bias || b.k % 32 || b.n_blocks * 16 != dn->n_valid ||
            b.n_blocks > 255 || a.n_blocks / 2 < 2 * FLAT_NP_MAX)
            return 1;
        sh.pairs[i] = a.n_blocks / 2;
        sh.dn_nb[i] = b.n_blocks;
        sh.dn_kb[i] = b.k / 32;
        int j = -1;
        for (int t = 0; t < G; ++t)
            if (gu->groups_host[t].out_row_base == b.a_row_base) j = t;
        if (j < 0 || gu->groups_host[j].n_blocks * 8 != b.k) return 1;
        sh.dn_src[i] = j;
    }
    flat_knobs(sh);
    static FlatShape cached_shape;
    static FlatPlan cached_plan;
    static bool have = false;
    if (!have || !(cached_shape == sh)) {
        flat_plan(sh, cached_plan);
        cached_shape = sh;
        have = true;
        if (getenv("UMOE_FLAT_DEBUG")) {
            fprintf(stderr, "umoe_moe_flat: plan ok=%d n_wg=%d makespan %.0f KiB mean %.0f KiB\n", (int)cached_plan.ok, n_wg, cached_plan.makespan, cached_plan.mean);
            if (cached_plan.ok && atoi(getenv("UMOE_FLAT_DEBUG")) > 1)
                for (int j = 0; j < n_wg; ++j)
                    fprintf(stderr, "  wg %3d: pairs %4d +%d | down group %2d blocks %3d +%d | group %2d blocks %3d +%d\n", j, cached_plan.gu[j] & 2047,
                            (cached_plan.gu[j] >> 11) & 7, cached_plan.dn[j] & 15, (cached_plan.dn[j] >> 4) & 255, (cached_plan.dn[j] >> 12) & 15,
                            (cached_plan.dn[j] >> 16) & 15, (cached_plan.dn[j] >> 20) & 255, cached_plan.dn[j] >> 28);
        }
    }
    const FlatPlan& pl = cached_plan;
    if (!pl.ok) return 1;
    flat_args A;
    memset(&A, 0, sizeof(A));
    A.a = r->x; A.norm_w = r->norm_w; A.rms_eps = r->rms_eps; A.h = reinterpret_cast<uint16_t*>(gu->out); A.y = reinterpret_cast<uint16_t*>(dn->out); A.flags = flags;
    A.dbg = g_flat_dbg;      // diagnostics (scripts/flat_timeline.py; NULL unless umoe_moe_flat_stamps(NULL) enabled them): stamps of the LAST launch
    A.lda = r->D; A.ldh = gu->ldo; A.ldy = dn->ldo; A.S = r->S; A.G = G; A.kb_gu = sh.kb_gu;
    int P = 0, kb_dn_max = 0;
    for (int i = 0; i < G; ++i) {
        A.w_gu[i] = gu->groups_host[i].w; A.w_dn[i] = dn->groups_host[i].w;
        A.pair0[i] = P; P += sh.pairs[i];
        A.h_row[i] = gu->groups_host[i].out_row_base;
        A.dn_kb[i] = sh.dn_kb[i]; A.dn_a_row[i] = dn->groups_host[i].a_row_base; A.dn_y_row[i] = dn->groups_host[i].out_row_base; A.dn_nb[i] = sh.dn_nb[i];
        A.prod_base[i] = pl.prod_base[i]; A.prod_n[i] = pl.prod_n[i];
        kb_dn_max = std::max(kb_dn_max, sh.dn_kb[i]);
    }
    for (int i = G; i <= FLAT_MAXG; ++i) A.pair0[i] = P;
    memcpy(A.gu, pl.gu, sizeof(uint32_t) * n_wg);
    memcpy(A.dn, pl.dn, sizeof(uint32_t) * n_wg);
    const umoe_rider_pub pub = *reinterpret_cast<const umoe_rider_pub*>(gu->rider_pub);
    UMOE_REQUIRE(pub.flags && pub.step && pub.err, "umoe_moe_flat: rider_pub needs flags / step / err");
    auto stage_bytes = [](int kb) { return (size_t)16 * 4 * (size_t)((kb * 16 + 255) & ~255); };
    size_t lds = std::max(stage_bytes(sh.kb_gu), stage_bytes(kb_dn_max));
    lds = std::max(lds, (size_t)8 * 2 * FLAT_NP_MAX * 1024);
    lds = std::max(lds, (size_t)8 * FLAT_ND_MAX1 * 1024);
    if (lds + FLAT_RIDER_LDS > 160 * 1024) return 1;
    static size_t configured = 0;
    if (lds + FLAT_RIDER_LDS > configured) {
        UMOE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&moe_flat_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + FLAT_RIDER_LDS)));
        configured = lds + FLAT_RIDER_LDS;
    }
    umoe_router_args rr = *r;
    rr.h_out = nullptr;          // nobody reads normalised rows from memory: every workgroup makes its own copy in LDS
    moe_flat_kernel<<<dim3((unsigned)n_wg), 512, lds + FLAT_RIDER_LDS, s>>>(A, rr, pub, (int)lds);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// Does a schedule exist for this decode shape on n_wg workgroups?  (the engine asks before it drops the RMSNorm launch)
bool umoe_moe_flat_feasible(int n_wg, int S, int D, int I_dyn, int I_sh, int n_real, int n_fix) {
    static int key[7] = {-1, -1, -1, -1, -1, -1, -1};
    static bool ans = false;
    const int k[7] = {n_wg, S, D, I_dyn, I_sh, n_real, n_fix};
    if (memcmp(k, key, sizeof(k)) == 0) return ans;
    std::vector<double> out(3 + 9 * (size_t)std::max(n_wg, 1));
    const int rc = (n_wg >= 1 && n_wg <= FLAT_MAXWG && I_dyn % 32 == 0 && I_sh % 32 == 0 && I_dyn / 16 >= 2 * FLAT_NP_MAX && I_sh / 16 >= 2 * FLAT_NP_MAX && D / 16 <= 255)
                       ? umoe_moe_flat_plan_probe(n_wg, S, D, I_dyn, I_sh, n_real, n_fix, out.data(), (int)out.size()) : -1;
    memcpy(key, k, sizeof(k));
    ans = rc == 0 && out[0] != 0.0;
    return ans;
}

// test hook (tests/test_abi_cpu.py, no GPU needed): the plan for a decode shape on `n_wg` workgroups (group order of the engine's
// hand-off launch: shared experts first); out[0] = ok, out[1] = model makespan (KiB), out[2] = mean KiB per workgroup, then per
// workgroup {first pair, pairs, rider token + 1, then {down group, first block, blocks} of its two slices}
extern "C" int umoe_moe_flat_plan_probe(int n_wg, int S, int D, int I_dyn, int I_sh, int n_real, int n_fix, double* out, int out_len) {
    FlatShape sh;
    memset(&sh, 0, sizeof(sh));
    sh.G = n_real + n_fix; sh.S = S; sh.n_wg = n_wg; sh.kb_gu = D / 32;
    if (sh.G > FLAT_MAXG || out_len < 3 + 9 * n_wg || n_wg < 1 || n_wg > FLAT_MAXWG) return -1;
    for (int i = 0; i < sh.G; ++i) {
        const bool shd = i < n_fix;
        sh.pairs[i] = (shd ? I_sh : I_dyn) / 16; sh.dn_nb[i] = D / 16; sh.dn_kb[i] = (shd ? I_sh : I_dyn) / 32; sh.dn_src[i] = i;
    }
    flat_knobs(sh);
    FlatPlan pl;
    flat_plan(sh, pl);
    out[0] = pl.ok ? 1.0 : 0.0; out[1] = pl.makespan; out[2] = pl.mean;
    if (pl.ok)
        for (int j = 0; j < n_wg; ++j) {
            double* o = out + 3 + 9 * j;
            o[0] = pl.gu[j] & 2047; o[1] = (pl.gu[j] >> 11) & 7; o[2] = pl.gu[j] >> 16;
            for (int k = 0; k < 2; ++k) {
                const uint32_t e16 = (pl.dn[j] >> (16 * k)) & 0xffffu;
                o[3 + 3 * k] = e16 & 15; o[4 + 3 * k] = (e16 >> 4) & 255; o[5 + 3 * k] = e16 >> 12;
            }
        }
    return 0;
}
