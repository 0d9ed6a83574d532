// Fused (flash-style) backward of causal GQA attention over full sequences, matrix cores only (no score matrices in HBM):
//   D[t][head]  = sum_d dO o O                                  (attn_bwd_d_kernel)
//   dQ          : query tile stationary, K / V tiles streamed   (attn_bwd_dq_kernel)   S^T, dP^T, dS^T -> dQ^T += K^T dS^T
//   dK, dV      : key tile stationary, Q / dO tiles streamed    (attn_bwd_dkv_kernel)  S, dP, dS -> dV^T += dO^T P, dK^T += Q^T dS
// P is recomputed from the forward's log-sum-exp (p = exp(scale * s - lse)); two kernels instead of one with atomics: every
// output element has ONE writer and a fixed summation order.  Operand tricks as in attn_prefill_kernel (umoe_attn.hip): the
// tile that is a row-major [row][128] image in LDS (256-byte rows, chunks XOR-swizzled by row & 15) serves as MFMA A operand
// by ds_read_b128 when the contraction runs over its 128 columns and by ds_read_b64_tr_b16 (hardware transpose) when the
// contraction runs over its rows; the probability / score-gradient tiles never leave the registers: the lane that holds 4
// consecutive rows of two 16-row MFMA outputs holds exactly one 8-deep slice of the next product's B operand.
// Reference graph: autograd through eager attention of the transformers dependency (called at model.py:228-237).
#include "umoe_common.h"
#include <stdlib.h>

typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef short v8s_t __attribute__((ext_vector_type(8)));

// hipcc (ROCm 7.2) moved __builtin_amdgcn_ds_read_tr16_b64 calls BELOW a following __syncthreads() (seen in the ISA of the
// dK/dV kernel: the transposing reads of the second half-step sat after the barrier, racing with the next tile's stores --
// wrong dV in the last d-block only, timing dependent; the cause is the compiler's placement, not a missing lgkmcnt wait: the
// builtin is not ordered against the barrier the way ordinary LDS loads are).  Pinned at the source, by construction: in front
// of every barrier behind which a tile is restaged, the ACCUMULATORS fed by transposing reads pass through an `asm volatile`
// (TR_PIN8).  Volatile asm and s_barrier keep their program order, the MFMAs that write the accumulators must precede the asm,
// and their operand reads must precede the MFMAs -- whatever the compiler thinks of the builtin.
#define TR_FENCE() asm volatile("" ::: "memory")
#define TR_PIN8(a) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]))

__device__ __forceinline__ int kvo(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }

// A operand [16 rows of the tile starting at r0][k = 32 columns starting at 32 kb]: lane (h, c16)
__device__ __forceinline__ bf16x8_t frag_rows(const char* tile, int r0, int kb, int h, int c16) {
    return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(tile + kvo(r0 + c16, kb * 4 + h)));
}
// A operand of the TRANSPOSED tile: rows = 16 columns of the tile starting at 16 db, k = 8 tile rows per lane group:
// rows rA + (4h .. 4h+3) and the same 16 rows further (rA + 16 + ...), matching the register layout of two MFMA outputs
__device__ __forceinline__ bf16x8_t frag_cols(const char* tile, int rA, int db, int h, int c16) {
    const int tq = c16 >> 2, tp = c16 & 3;
    const int cch = db * 2 + (tp >> 1), sub = (tp & 1) * 8;
    const v4s_t va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(tile + kvo(rA + 4 * h + tq, cch) + sub));
    const v4s_t vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(tile + kvo(rA + 16 + 4 * h + tq, cch) + sub));
    const v8s_t v8 = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
    return __builtin_bit_cast(bf16x8_t, v8);
}
__device__ __forceinline__ bf16x8_t pack_bf8(const float (&v)[8]) {
    uint4 u;
    u.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16); u.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    u.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16); u.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
    return __builtin_bit_cast(bf16x8_t, u);
}

// ------------------------------------------------------------------------------------ D = rowsum(dO o O)
__global__ __launch_bounds__(256) void attn_bwd_d_kernel(const uint16_t* __restrict__ d_out, const uint16_t* __restrict__ out, long n_rows,
                                                         float* __restrict__ D) {
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);     // one wave per (token, head) row of 128 values
    if (r >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const uint32_t a = *reinterpret_cast<const uint32_t*>(d_out + r * 128 + 2 * lane);
    const uint32_t b = *reinterpret_cast<const uint32_t*>(out + r * 128 + 2 * lane);
    float s = __uint_as_float(a << 16) * __uint_as_float(b << 16) + __uint_as_float(a & 0xffff0000u) * __uint_as_float(b & 0xffff0000u);
    s = wave_sum(s);
    if (lane == 0) D[r] = s;
}

// ------------------------------------------------------------------------------------ dQ
template <int GP>
__global__ __launch_bounds__(64 * GP) void attn_bwd_dq_kernel(const umoe_attn_bwd_args a, const int32_t* __restrict__ kv_start,
                                                              const float* __restrict__ Dv) {
    constexpr int HD = 128, KT = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][K 16 KiB | V 16 KiB]
    const int t0 = blockIdx.x * 16, kvh = blockIdx.y, row = blockIdx.z;
    const int G = a.H / a.KVH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 4, c16 = lane & 15;
    const int head = kvh * G + wave;
    const bool live_wave = wave < G;
    const int kv0 = kv_start[row];
    const int kmax = min(t0 + 15, a.T - 1);
    const uint16_t* Kc = a.k_cache + ((size_t)row * a.KVH + kvh) * a.Lmax * HD;
    const uint16_t* Vc = a.v_cache + ((size_t)row * a.KVH + kvh) * a.Lmax * HD;
    const int t = min(t0 + c16, a.T - 1);
    const bool q_ok = t0 + c16 < a.T;
    const size_t qrow = ((size_t)row * a.T + t) * a.H + (live_wave ? head : 0);
    bf16x8_t qf[4], dof[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        qf[kb] = __builtin_bit_cast(bf16x8_t, ld16(a.q + qrow * HD + kb * 32 + h * 8));
        dof[kb] = __builtin_bit_cast(bf16x8_t, ld16(a.d_out + qrow * HD + kb * 32 + h * 8));
    }
    const float lse_q = a.lse[qrow], D_q = Dv[qrow];
    const int my_qpos = t0 + c16;
    f32x4_t acc[8];
#pragma unroll
    for (int db = 0; db < 8; ++db) acc[db] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    constexpr int NTH = 64 * GP, NLD = 2048 / NTH;
    uint4 stg[NLD];
    auto gload = [&](int k0) {
#pragma unroll
        for (int n = 0; n < NLD; ++n) {
            const int idx = tid + n * NTH;
            const int r = (idx >> 4) & 63, c = idx & 15;
            const int key = k0 + r;
            const uint16_t* base = (idx >> 10) ? Vc : Kc;
            stg[n] = make_uint4(0, 0, 0, 0);
            if (key <= kmax) stg[n] = ld16(base + (size_t)key * HD + c * 8);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int n = 0; n < NLD; ++n) {
            const int idx = tid + n * NTH;
            st16(smem + buf * 32768 + (idx >> 10) * 16384 + kvo((idx >> 4) & 63, idx & 15), stg[n]);
        }
    };
    const int kfirst = kv0 & ~(KT - 1);
    if (kfirst <= kmax) {
        gload(kfirst);
        lstore(0);
    }
    __syncthreads();
    int it = 0;
    for (int k0 = kfirst; k0 <= kmax; k0 += KT, ++it) {
        const int buf = it & 1;
        const bool more = k0 + KT <= kmax;
        if (more) gload(k0 + KT);
        const char* Kt = smem + buf * 32768;
        const char* Vt = Kt + 16384;
        if (live_wave) {
            bf16x8_t dsf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float dsv[8];
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    const int kt = 2 * ks + sub;
                    f32x4_t sacc = {0.f, 0.f, 0.f, 0.f}, dpacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) {
                        sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Kt, kt * 16, kb, h, c16), qf[kb], sacc, 0, 0, 0);
                        dpacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Vt, kt * 16, kb, h, c16), dof[kb], dpacc, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = k0 + kt * 16 + 4 * h + r;
                        const bool ok = q_ok && key >= kv0 && key <= my_qpos;
                        const float p = ok ? __expf(sacc[r] * a.scale - lse_q) : 0.f;
                        dsv[sub * 4 + r] = p * (dpacc[r] - D_q) * a.scale;
                    }
                }
                dsf[ks] = pack_bf8(dsv);
            }
#pragma unroll
            for (int db = 0; db < 8; ++db)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    acc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(Kt, 32 * ks, db, h, c16), dsf[ks], acc[db], 0, 0, 0);
        }
        if (more) lstore(buf ^ 1);
        TR_FENCE();
        TR_PIN8(acc);
        __syncthreads();
    }
    if (live_wave && q_ok) {
        uint16_t* o = a.dq + qrow * HD + 4 * h;
#pragma unroll
        for (int db = 0; db < 8; ++db) {
            const uint32_t lo = (uint32_t)f2bf(acc[db][0]) | ((uint32_t)f2bf(acc[db][1]) << 16);
            const uint32_t hi = (uint32_t)f2bf(acc[db][2]) | ((uint32_t)f2bf(acc[db][3]) << 16);
            *reinterpret_cast<uint2*>(o + db * 16) = make_uint2(lo, hi);
        }
    }
}

// ------------------------------------------------------------------------------------ dK, dV
// grid = (64-key tiles, kv heads x P, rows), 4 waves: wave w owns keys K0 + 16 w .. +15 (stationary K / V fragments and the
// dK^T / dV^T accumulators in registers), the workgroup streams the Q / dO tiles (64 queries) of G / P query heads of the group.
// Causality makes the work of a key tile proportional to the queries behind it (the first tile: every query tile of every head,
// the last: one per head), so the kernel lasts as long as its first key tile: splitting the group's heads over P workgroups
// (P = 4 at G = 8) cuts that critical path by P and quadruples the waves in flight; the P partial sums go to fp32 slabs and
// attn_bwd_dkv_sum_kernel adds them in fixed order (one rounding to bf16, as before).  P = 1: direct bf16 stores.
// NW waves = 16 NW keys per workgroup: every key tile re-streams the Q / dO tiles of the queries behind it, so the L2 -> LDS
// traffic (5.3 GB per layer at 4 x 1560 tokens with 64-key tiles: the kernel ran at L2 bandwidth) halves with 128-key tiles.
template <int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dkv_kernel(const umoe_attn_bwd_args a, const int32_t* __restrict__ kv_start,
                                                               const float* __restrict__ Dv, const int P, float* __restrict__ slab_k,
                                                               float* __restrict__ slab_v) {
    constexpr int NT = NW * 64, NLD = 2048 / NT;               // threads, 16-byte chunks per thread and stage
    constexpr int HD = 128;
    constexpr int STAGE = 2 * 16384 + 512;                     // Q tile | dO tile | lse[64] | D[64]
    extern __shared__ __attribute__((aligned(16))) char smem[];   // two stages
    const int K0 = blockIdx.x * (NW * 16), kvh = blockIdx.y / P, part = blockIdx.y - kvh * P, row = blockIdx.z;
    const int G = a.H / a.KVH, GW = G / P;                      // query heads of this workgroup: kvh G + part GW .. + GW
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 4, c16 = lane & 15;
    const int kv0 = kv_start[row];
    const int key = K0 + wave * 16 + c16;                       // this lane's key (column of S)
    const bool key_ok = key < a.T && key >= kv0;
    const uint16_t* Kc = a.k_cache + ((size_t)row * a.KVH + kvh) * a.Lmax * HD;
    const uint16_t* Vc = a.v_cache + ((size_t)row * a.KVH + kvh) * a.Lmax * HD;
    bf16x8_t kf[4], vf[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        uint4 ku = make_uint4(0, 0, 0, 0), vu = make_uint4(0, 0, 0, 0);
        if (key < a.T) {
            ku = ld16(Kc + (size_t)key * HD + kb * 32 + h * 8);
            vu = ld16(Vc + (size_t)key * HD + kb * 32 + h * 8);
        }
        kf[kb] = __builtin_bit_cast(bf16x8_t, ku);
        vf[kb] = __builtin_bit_cast(bf16x8_t, vu);
    }
    f32x4_t dkT[8], dvT[8];
#pragma unroll
    for (int db = 0; db < 8; ++db) dkT[db] = dvT[db] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // steps: every query head of the group x every 64-query tile that can see a key of this workgroup
    const int q_first = (K0 > kv0 ? K0 : kv0) & ~63;
    const int nq_steps = q_first < a.T ? (a.T - q_first + 63) >> 6 : 0;
    const int n_steps = GW * nq_steps;
    uint4 stg[NLD];
    float stg_l = 0.f;
    auto gload = [&](int step) {
        const int head = kvh * G + part * GW + step / nq_steps, q0 = q_first + (step % nq_steps) * 64;
#pragma unroll
        for (int n = 0; n < NLD; ++n) {
            const int idx = tid + n * NT;
            const int tq = q0 + ((idx >> 4) & 63);
            stg[n] = make_uint4(0, 0, 0, 0);
            if (tq < a.T) stg[n] = ld16(((idx >> 10) ? a.d_out : a.q) + (((size_t)row * a.T + tq) * a.H + head) * HD + (idx & 15) * 8);
        }
        if (tid < 128) {                                        // threads 0..63: lse, 64..127: D
            const int tq = q0 + (tid & 63);
            const size_t qr = ((size_t)row * a.T + min(tq, a.T - 1)) * a.H + head;
            stg_l = tid < 64 ? (tq < a.T ? a.lse[qr] : INFINITY) : (tq < a.T ? Dv[qr] : 0.f);
        }
    };
    auto lstore = [&](int buf) {
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int n = 0; n < NLD; ++n) {
            const int idx = tid + n * NT;
            st16(base + (idx >> 10) * 16384 + kvo((idx >> 4) & 63, idx & 15), stg[n]);
        }
        if (tid < 128) reinterpret_cast<float*>(base + 32768)[tid] = stg_l;
    };
    if (n_steps > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    for (int step = 0; step < n_steps; ++step) {
        const int buf = step & 1;
        if (step + 1 < n_steps) gload(step + 1);
        const char* Qt = smem + buf * STAGE;
        const char* dOt = Qt + 16384;
        const float* lse_s = reinterpret_cast<const float*>(Qt + 32768);
        const float* D_s = lse_s + 64;
        const int q0 = q_first + (step % nq_steps) * 64;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float pv[8], dsv[8];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int qt = 2 * ks + sub;
                f32x4_t sacc = {0.f, 0.f, 0.f, 0.f}, dpacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Qt, qt * 16, kb, h, c16), kf[kb], sacc, 0, 0, 0);
                    dpacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(dOt, qt * 16, kb, h, c16), vf[kb], dpacc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ql = qt * 16 + 4 * h + r, tq = q0 + ql;      // query of this accumulator row
                    const bool ok = key_ok && tq < a.T && key <= tq;
                    const float p = ok ? __expf(sacc[r] * a.scale - lse_s[ql]) : 0.f;
                    pv[sub * 4 + r] = p;
                    dsv[sub * 4 + r] = p * (dpacc[r] - D_s[ql]) * a.scale;
                }
            }
            const bf16x8_t pf = pack_bf8(pv), dsf = pack_bf8(dsv);
#pragma unroll
            for (int db = 0; db < 8; ++db) {
                dvT[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(dOt, 32 * ks, db, h, c16), pf, dvT[db], 0, 0, 0);
                dkT[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(Qt, 32 * ks, db, h, c16), dsf, dkT[db], 0, 0, 0);
            }
        }
        if (step + 1 < n_steps) lstore(buf ^ 1);
        TR_FENCE();
        TR_PIN8(dvT);
        TR_PIN8(dkT);
        __syncthreads();
    }
    if (P > 1) {
        if (key < a.T) {
            const size_t o = ((((size_t)part * a.rows + row) * a.KVH + kvh) * a.T + key) * HD + 4 * h;
#pragma unroll
            for (int db = 0; db < 8; ++db) {
                *reinterpret_cast<float4*>(slab_k + o + db * 16) = make_float4(dkT[db][0], dkT[db][1], dkT[db][2], dkT[db][3]);
                *reinterpret_cast<float4*>(slab_v + o + db * 16) = make_float4(dvT[db][0], dvT[db][1], dvT[db][2], dvT[db][3]);
            }
        }
        return;
    }
    if (key < a.T) {
        uint16_t* ok = a.dk_cache + (((size_t)row * a.KVH + kvh) * a.Lmax + key) * HD + 4 * h;
        uint16_t* ov = a.dv_cache + (((size_t)row * a.KVH + kvh) * a.Lmax + key) * HD + 4 * h;
#pragma unroll
        for (int db = 0; db < 8; ++db) {
            *reinterpret_cast<uint2*>(ok + db * 16) = make_uint2((uint32_t)f2bf(dkT[db][0]) | ((uint32_t)f2bf(dkT[db][1]) << 16),
                                                                 (uint32_t)f2bf(dkT[db][2]) | ((uint32_t)f2bf(dkT[db][3]) << 16));
            *reinterpret_cast<uint2*>(ov + db * 16) = make_uint2((uint32_t)f2bf(dvT[db][0]) | ((uint32_t)f2bf(dvT[db][1]) << 16),
                                                                 (uint32_t)f2bf(dvT[db][2]) | ((uint32_t)f2bf(dvT[db][3]) << 16));
        }
    }
}

// dK / dV = sum of the P head-split slabs, parts in ascending order, one rounding to bf16; 8 elements per thread
__global__ __launch_bounds__(256) void attn_bwd_dkv_sum_kernel(const umoe_attn_bwd_args a, const int P, const float* __restrict__ slab_k,
                                                               const float* __restrict__ slab_v) {
    constexpr int HD = 128;
    const size_t per = (size_t)a.rows * a.KVH * a.T * HD;        // elements of one slab
    const size_t i8 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i8 >= per) return;
    const size_t rk = i8 / HD;                                   // (row * KVH + kvh) * T + key
    const size_t rowkv = rk / a.T, key = rk - rowkv * a.T;
    const size_t o = (rowkv * a.Lmax + key) * HD + (i8 & (HD - 1));
    for (int which = 0; which < 2; ++which) {
        const float* sl = which ? slab_v : slab_k;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int p = 0; p < P; ++p) {
            const float4 x = *reinterpret_cast<const float4*>(sl + (size_t)p * per + i8);
            const float4 y = *reinterpret_cast<const float4*>(sl + (size_t)p * per + i8 + 4);
            acc[0] += x.x; acc[1] += x.y; acc[2] += x.z; acc[3] += x.w;
            acc[4] += y.x; acc[5] += y.y; acc[6] += y.z; acc[7] += y.w;
        }
        *reinterpret_cast<uint4*>((which ? a.dv_cache : a.dk_cache) + o) = pack8(acc);
    }
}

template <int GP>
static int launch_dq(const umoe_attn_bwd_args* a, const int32_t* kv_start_dev, const float* D, hipStream_t s) {
    static bool configured = false;
    if (!configured) {
        UMOE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<GP>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        configured = true;
    }
    attn_bwd_dq_kernel<GP><<<dim3((unsigned)ceil_div(a->T, 16), (unsigned)a->KVH, (unsigned)a->rows), 64 * GP, 65536, s>>>(*a, kv_start_dev, D);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// returns 1 when the shape is not covered (the caller then runs the unfused composite).  ws: fp32 D [rows*T*H] then rows int32.
int umoe_attn_bwd_fused(const umoe_attn_bwd_args* a, umoe_stream_t stream) {
    static int enabled = -1;
    if (enabled < 0) {
        const char* e = getenv("UMOE_ATTN_BWD_FUSED");
        enabled = e ? atoi(e) : 1;
    }
    const int G = a->KVH > 0 ? a->H / a->KVH : 0;
    if (!enabled || !a->out || !a->lse || a->hd != 128 || G < 1 || G > 8 || a->T < 1 || ceil_div(a->T, 16) > 65535 || a->rows > 65535) return 1;
    const size_t nD = (size_t)a->rows * a->T * a->H;
    UMOE_REQUIRE(a->ws && a->ws_bytes >= nD * 4 + (size_t)a->rows * 4 + 256, "umoe_attn_prefill_bwd: workspace too small for the fused path");
    hipStream_t s = (hipStream_t)stream;
    float* D = reinterpret_cast<float*>(a->ws);
    int32_t* kvs = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(a->ws) + ((nD * 4 + 255) & ~(size_t)255));
    UMOE_HIP(hipMemcpyAsync(kvs, a->kv_start_host, sizeof(int32_t) * a->rows, hipMemcpyHostToDevice, s));
    attn_bwd_d_kernel<<<dim3((unsigned)ceil_div((int)nD, 4)), 256, 0, s>>>(a->d_out, a->out, (long)nD, D);
    UMOE_LAUNCH_CHECK();
    // the dQ pass and the dK / dV pass are independent (both read Q, K, V, dO, D; disjoint outputs) and each leaves the chip half empty
    // (784 / 416 workgroups of 64 KiB LDS, causal imbalance): the dQ pass runs on the side stream beside the other, joined below
    BwdSide& bs = bwd_side();
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    bool overlap = false;
    if (bs.ok) {
        if (hipStreamIsCapturing(s, &cap) == hipSuccess) overlap = cap == hipStreamCaptureStatusNone;
        else (void)hipGetLastError();
    }
    hipStream_t sq = s;
    if (overlap) {
        UMOE_HIP(hipEventRecord(bs.fork, s));
        UMOE_HIP(hipStreamWaitEvent(bs.side, bs.fork, 0));
        sq = bs.side;
    }
    int rc;
    if (G <= 1) rc = launch_dq<1>(a, kvs, D, sq);
    else if (G <= 2) rc = launch_dq<2>(a, kvs, D, sq);
    else if (G <= 4) rc = launch_dq<4>(a, kvs, D, sq);
    else rc = launch_dq<8>(a, kvs, D, sq);
    if (rc) return rc;
    if (overlap) UMOE_HIP(hipEventRecord(bs.join, bs.side));
    // head split of the dK / dV pass: as many parts as divide the group (up to 4) and fit the workspace behind D and kv_start
    static int psplit = -1;
    if (psplit < 0) {
        const char* e = getenv("UMOE_ATTN_BWD_SPLIT");
        psplit = e ? atoi(e) : 4;
    }
    int P = 1;
    const size_t base = ((nD * 4 + 255) & ~(size_t)255) + (((size_t)a->rows * 4 + 255) & ~(size_t)255);
    const size_t per = (size_t)a->rows * a->KVH * a->T * 128;
    for (int cand = 4; cand >= 2; cand >>= 1)
        if (cand <= psplit && G % cand == 0 && a->ws_bytes >= base + 2 * (size_t)cand * per * 4 && (size_t)a->KVH * cand <= 65535) { P = cand; break; }
    float* slab_k = reinterpret_cast<float*>(reinterpret_cast<char*>(a->ws) + base);
    float* slab_v = slab_k + (size_t)P * per;
    static int key_waves = -1;
    if (key_waves < 0) {
        const char* e = getenv("UMOE_ATTN_BWD_KEYWAVES");
        key_waves = e ? atoi(e) : 8;
    }
    constexpr int LDS = 2 * (2 * 16384 + 512);
    static bool cfg4 = false, cfg8 = false;
    if (key_waves == 8 && a->T > 64) {     // 128-key tiles
        if (!cfg8) {
            UMOE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
            cfg8 = true;
        }
        attn_bwd_dkv_kernel<8><<<dim3((unsigned)ceil_div(a->T, 128), (unsigned)(a->KVH * P), (unsigned)a->rows), 512, LDS, s>>>(*a, kvs, D, P, slab_k, slab_v);
    } else {
        if (!cfg4) {
            UMOE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
            cfg4 = true;
        }
        attn_bwd_dkv_kernel<4><<<dim3((unsigned)ceil_div(a->T, 64), (unsigned)(a->KVH * P), (unsigned)a->rows), 256, LDS, s>>>(*a, kvs, D, P, slab_k, slab_v);
    }
    UMOE_LAUNCH_CHECK();
    if (P > 1) {
        attn_bwd_dkv_sum_kernel<<<dim3((unsigned)((per / 8 + 255) / 256)), 256, 0, s>>>(*a, P, slab_k, slab_v);
        UMOE_LAUNCH_CHECK();
    }
    if (overlap) UMOE_HIP(hipStreamWaitEvent(s, bs.join, 0));
    return 0;
}
