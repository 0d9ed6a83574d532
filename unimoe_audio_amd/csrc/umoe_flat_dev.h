// Device side of the flat expert launch (umoe_moe_flat.hip) shared with the expert-parallel flat launch (umoe_moe_ep.hip): the argument
// block, the LDS tile layout, the bounded flag wait and the two slice bodies (gate/up SwiGLU, down projection) whose tiles, K split and
// reduction order are those of moe_fused_kernel (umoe_gemm.hip) -- bit-identical outputs whichever launch computes a tile.
#pragma once
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

// ---- o_proj half tiles (arithmetic of wstream_body<1, 16, PLAIN, BF16_RESID, 4> per feature: K split in four 16-step slices summed in order,
// bf16 rounding, residual add, bf16 rounding).  Half tile ht = features [8 ht, 8 ht + 8): the A operand's 16 rows are those 8 features twice
// (an output row depends on its own A row only), the valid outputs are rows 0..7. ----
// o_proj INSIDE the launch (half > 0; umoe_moe_flat with an o_proj argument): the raw rows x1 = x + o_proj(attention rows) do not exist when the
// launch starts -- every workgroup computes HALF of a 16-feature tile of them first, hands it over, and requests its first expert weights in
// front of the wait: the o_proj launch, its boundary and its cold start leave the chain.  A kernel argument of its OWN (growing flat_args by
// these fields made hipcc copy that whole 2.7 KB block into scratch).
struct flat_o {
    const uint16_t* rows;      // merged attention rows [S][lda_rows]
    const uint16_t* w;         // WP16 o_proj weights [D / 16 blocks][64 k-steps]
    const uint16_t* resid;     // residual stream x [S][lda]
    uint16_t* x1;              // = flat_args.a: the raw rows, written here
    uint32_t* flags;           // [8 replicas][256] words: half tile ht published (epochs)
    int lda_rows, lda, S, half, n_wg;      // half = half tiles (2 * D / 16; 0 = o_proj is its own launch), n_wg = workgroups of the launch
};
// PREFETCH: the caller's first weight chunk (pw0 <- pwp at k-step ic0) is requested inside: by waves 1..7 as soon as the half tile's own loads
// have landed, by wave 0 -- whose drain in front of the publish would wait for it (vmcnt counts loads and stores, in order) -- behind the
// publish.  The weight stream then runs through the half tile, its hand-off and the wait.  (Both register stages here: 68-102 spilled
// registers -- the normalisation of the rows runs with the stages live.)
template <int NT, bool PREFETCH>
__device__ __forceinline__ void flat_oproj_half(const flat_o O, const umoe_rider_pub& pub, const unsigned b, char* smem, const int tid,
                                                const flat_u32x4* const (&pwp)[NT], flat_u32x4 (&pw0)[NT], const int ic0) {
    auto prefetch = [&]() {
#pragma unroll
        for (int t = 0; t < NT; ++t) pw0[t] = __builtin_nontemporal_load(pwp[t] + (size_t)ic0 * 64);
    };
    bool first = true;
    constexpr int KB = 64, TPR = 32;
    constexpr int QS = (KB * 16 + 255) & ~255, RS = QS * 4;
    const int lane = tid & 63, h = lane >> 4, mm = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ws = wave & 3;                                         // waves 4..7 repeat the work of 0..3 (no branch around a load); only 0..3 count
    const uint32_t epoch = flat_epoch(pub);
    const int m = tid / TPR, sub = tid % TPR;
    const bool valid = m < O.S;
    for (int ht = (int)b; ht < O.half; ht += O.n_wg) {
        const int t = ht >> 1, half = ht & 1;
        const uint16_t* src = O.rows + (size_t)(valid ? m : 0) * O.lda_rows;
        uint4 buf[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) buf[n] = ld16(src + ((n >> 1) * KB + sub + TPR * (n & 1)) * 8);
        const flat_u32x4* wp = reinterpret_cast<const flat_u32x4*>(O.w) + ((size_t)t * KB) * 64 + (h * 16 + 8 * half + (mm & 7));
        flat_u32x4 wo[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) wo[u] = __builtin_nontemporal_load(wp + (size_t)(16 * ws + u) * 64);
        const int col = 16 * t + 8 * half + 4 * (h & 1);
        const uint2 rv2 = *reinterpret_cast<const uint2*>(O.resid + (size_t)(mm < O.S ? mm : 0) * O.lda + col);
        uint32_t rvx, rvy;
#pragma unroll
        for (int n = 0; n < 8; ++n)
            if (valid) st16(smem + m * RS + flat_lds_chunk_off(QS, n >> 1, sub + TPR * (n & 1), m), buf[n]);
        if constexpr (PREFETCH) {
            // everything this half tile loaded has to be in its registers HERE (the asm operands make hipcc place its wait in front of the
            // prefetch: behind it, a wait for these older loads would be a wait for the younger weight chunks too)
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("" : "+v"(wo[u]));
            uint32_t r0 = rv2.x, r1 = rv2.y;
            asm volatile("" : "+v"(r0), "+v"(r1));
            rvx = r0; rvy = r1;
            if (first && wave != 0) prefetch();
        } else {
            rvx = rv2.x; rvy = rv2.y;
        }
        __syncthreads();
        f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const char* bbase = smem + mm * RS;
        // (fragments four at a time: the two weight stages of the caller are live here, all sixteen at once spilled)
#pragma unroll
        for (int u0 = 0; u0 < 16; u0 += 4) {
            uint4 bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) bv[u] = *reinterpret_cast<const uint4*>(bbase + flat_lds_chunk_off(QS, h, 16 * ws + u0 + u, mm));
#pragma unroll
            for (int u = 0; u < 4; ++u)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wo[u0 + u]), __builtin_bit_cast(bf16x8_t, bv[u]), acc, 0, 0, 0);
        }
        __syncthreads();
        f32x4_t* red = reinterpret_cast<f32x4_t*>(smem);
        if (wave < 4) red[wave * 64 + lane] = acc;
        __syncthreads();
        if (wave == 0 && h < 2 && mm < O.S) {
            f32x4_t s4 = red[lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const f32x4_t v = red[w * 64 + lane];
                s4[0] += v[0]; s4[1] += v[1]; s4[2] += v[2]; s4[3] += v[3];
            }
            uint16_t y[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = rbf(s4[j] + 0.f);
                const uint32_t rw = j < 2 ? rvx : rvy;
                const float rv = __uint_as_float((j & 1) ? (rw & 0xffff0000u) : (rw << 16));
                x = rv + x;
                y[j] = f2bf(x);
            }
            const auto xrs = __builtin_amdgcn_make_buffer_rsrc(O.x1, 0, O.S * O.lda * 2, 0x00020000);
            const flat_u32x2 v2 = {(uint32_t)y[0] | ((uint32_t)y[1] << 16), (uint32_t)y[2] | ((uint32_t)y[3] << 16)};
            __builtin_amdgcn_raw_buffer_store_b64(v2, xrs, (mm * O.lda + col) * 2, 0, 16);
        }
        if (wave == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // only wave 0 stored (and only it has nothing else in flight)
        __syncthreads();
        if (tid < 8)
            __hip_atomic_store(reinterpret_cast<umoe_gu32*>(reinterpret_cast<uintptr_t>(O.flags + tid * 256 + ht)), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (PREFETCH) {
            if (first && wave == 0) prefetch();
        }
        first = false;
    }
    if constexpr (PREFETCH) {
        if (first) prefetch();      // (a workgroup without a half tile)
    }
}

// every row of x1 is complete when all half tiles are.  ONE wave polls, four flags per lane (one 16-byte sc1 load per lane: the 256 flags of a
// replica are one 1 KiB wave-load), eight replicas: the first version -- every thread of every workgroup polling its own word of four replicas
// -- was 65 k polling loads per round on 16 cache lines and cost more than the launch boundary it replaced.  Bounded like flat_wait.
__device__ __forceinline__ void flat_oproj_wait(const flat_o O, const umoe_rider_pub& pub, const unsigned b, const int tid) {
    if (tid < 64) {
        const uint32_t epoch = flat_epoch(pub);
        const auto frs = __builtin_amdgcn_make_buffer_rsrc(O.flags + (b & 7u) * 256, 0, 1024, 0x00020000);
        umoe_gu32* err = reinterpret_cast<umoe_gu32*>(reinterpret_cast<uintptr_t>(pub.err));
        const bool mine = 4 * tid < O.half;
        const unsigned long long t0 = wall_clock64();
        for (unsigned spins = 0;; ++spins) {
            const flat_u32x4 f4 = __builtin_amdgcn_raw_buffer_load_b128(frs, tid * 16, 0, 16);
            const bool ok = !mine || ((int32_t)(f4[0] - epoch) >= 0 && (int32_t)(f4[1] - epoch) >= 0 && (int32_t)(f4[2] - epoch) >= 0 && (int32_t)(f4[3] - epoch) >= 0);
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(2);
            if ((spins & 1023u) == 1023u) {
                if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                if (wall_clock64() - t0 > 200000000ull) {      // 2 s
                    uint32_t zero = 0u;
                    if (tid == 0) __hip_atomic_compare_exchange_strong(err, &zero, 6u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
    }
    __syncthreads();
}

// oph (a kernel argument: scalar branches) 0: the raw rows `a` exist when the launch starts; 1: they are made INSIDE this launch -- the o_proj
// half tile, the hand-off and the wait sit between this slice's first weight request and its row loads; 2: made inside the launch, half tile
// and wait already done by the caller (the router riders).  ONE instantiation per NP serves all three: a second set of instantiations made
// hipcc copy the whole 2.7 KB kernel-argument block into scratch (private_segment_fixed_size 36 -> 2736).
template <int NP, bool PUBLISH = true>
__device__ __forceinline__ void flat_gateup(const flat_args& A, const umoe_rider_pub& pub, const int fp0, const unsigned b, char* smem, flat_stamps& st,
                                            const int tid_in, const int oph = 0, const flat_o O = flat_o{}) {      // tid_in = threadIdx.x (a caller that loops over slices passes an opaque copy: see umoe_moe_ep.hip)
    constexpr int NT = 2 * NP, WV = 8, KB = 64;
    const int tid = tid_in, lane = tid & 63;
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
        // normalise + stage the 16 rows from the slices the threads hold (`buf`, `nw1`: the row chunks and the norm weights of this thread).
        // Called at the end of BOTH request orders below, so that each keeps its own counted waits: a common tail behind the join made
        // the path that requests the rows first wait for the weight chunk behind them.
        auto norm_stage = [&](uint4 (&buf)[8], const uint4 nw1) {
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
        };
        if (oph != 0) {
            // o_proj inside the launch: the first weight chunk goes out FIRST (nothing it needs is missing), the half tile, its hand-off and
            // the wait for everybody's follow while it flies; then the rows (every load of handed-over bytes an sc1 load)
            // oph 1: the first register stage is requested inside the half tile (flat_oproj_half); oph 2 (router riders: half tile and
            // wait done by the caller): here
            if (oph == 1) {
                flat_oproj_half<NT, true>(O, pub, b, smem, tid, wp, w0, min(i0, i1 - 1));
                flat_oproj_wait(O, pub, b, tid);
            } else {
                load_chunk(w0, i0);
            }
            uint4 buf[8];
            const auto rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(A.a), 0, A.S * A.lda * 2, 0x00020000);
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                const flat_u32x4 t4 = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)(((size_t)(valid ? m : 0) * A.lda + ((n >> 1) * KB + sub + TPR * (n & 1)) * 8) * 2), 0, 16);
                buf[n] = make_uint4(t4[0], t4[1], t4[2], t4[3]);
            }
            const uint4 nw1 = ld16(A.norm_w + (tid & (4 * KB - 1)) * 8);
            FSTAMP(1);
            norm_stage(buf, nw1);
        } else {
            // the rows exist (the previous launch wrote them): rows first, the weight stream right behind them; normalise while the chunk flies
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
            norm_stage(buf, nw1);
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
    // (PUBLISH false: a workgroup with several slices of one phase raises its flag once, behind the last of them -- the caller does)
    if constexpr (PUBLISH) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0)
            __hip_atomic_store(reinterpret_cast<umoe_gu32*>(reinterpret_cast<uintptr_t>(A.flags + b)), flat_epoch(pub), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    FSTAMP(6);
}

// ---- down-projection slice: blocks [nb0, nb0 + ND) of group grp (arithmetic of wstream_body<6, 2, PLAIN, BF16, 8> per tile) ----
// U = 2 for an even number of k-steps (whole 2-step chunks per wave), U = 1 for an odd one: the K split of the 2-step launch does not
// depend on U then, and a 1-step stream has no clamped duplicate step at the end of a wave's slice.
template <int ND, int U>
__device__ __forceinline__ void flat_down(const flat_args& A, const umoe_rider_pub& pub, const int grp, const int nb0, char* smem, flat_stamps& st,
                                          const int sb, const int tid_in) {
    constexpr int NT = ND, WV = 8;
    const int tid = tid_in, lane = tid & 63;
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
