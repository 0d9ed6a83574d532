// mRoPE + KV append, and GQA attention over a preallocated KV cache (decode and causal prefill).
//
// Replaces transformers' Qwen2_5_VLAttention internals as used by the reference
// (utils/UniMoE_Audio_model.py:204,228-237): apply_multimodal_rotary_pos_emb, DynamicCache.update
// (an O(L) torch.cat per layer per step in the reference, model.py:353-354,1109) and the sdpa core.
//
// attn_kernel: grid = (key-splits, kv-heads, rows*nq).  A workgroup owns one (query token, kv head)
// and one contiguous slice of the keys; its 4 waves take 16-key tiles round-robin:
//   * QK^T on v_mfma_f32_16x16x32_bf16 with the K tile as the A operand (16 keys x 128 dims, read
//     straight from HBM, 64 contiguous bytes per lane) and the GQA query group as the B operand
//     (up to 16 query heads that share this kv head) -- KV is read once for all heads of the group;
//   * online softmax per head in registers (keys live on lane-groups, heads on lanes);
//   * P.V on the VALU: each lane owns two of the 128 value columns, V rows are read coalesced.
// Partials (m, l, O) per split are merged by attn_combine_kernel (flash-decoding).
// Roofline: HBM (KV bytes = 2 * L * KVH * hd * 2 B per row per layer).
#include "umoe_common.h"
#include <stdlib.h>

// ------------------------------------------------------------------------------------ rope + append
__global__ __launch_bounds__(256) void rope_append_kernel(const umoe_rope_args a) {
    const int tok = blockIdx.x;
    const int row = tok / a.T;
    const int hd = a.hd, half = hd >> 1;
    const int nheads = a.H + 2 * a.KVH;
    const int ld = nheads * hd;
    const uint16_t* src = a.qkv + (size_t)tok * ld;
    const int slot = a.kv_pos[tok];
    const int p0 = a.pos3[tok], p1 = a.pos3[a.n_tok + tok], p2 = a.pos3[2 * a.n_tok + tok];
    // work items: (head, i) for i < half, rotating q and k heads; v heads are copied
    for (int w = threadIdx.x; w < nheads * half; w += blockDim.x) {
        const int head = w / half, i = w - head * half;
        const uint16_t* hs = src + head * hd;
        if (head < a.H + a.KVH) {
            const int pos = (i < a.sec0) ? p0 : (i < a.sec0 + a.sec1 ? p1 : p2);
            const float c = bf2f(a.cos_tab[(size_t)pos * half + i]);
            const float s = bf2f(a.sin_tab[(size_t)pos * half + i]);
            const float x1 = bf2f(hs[i]), x2 = bf2f(hs[i + half]);
            // q*cos + rotate_half(q)*sin, every op rounded to bf16 as torch does on bf16 tensors
            const uint16_t o1 = f2bf(rbf(x1 * c) + rbf(-x2 * s));
            const uint16_t o2 = f2bf(rbf(x2 * c) + rbf(x1 * s));
            if (head < a.H) {
                uint16_t* d = a.q_out + (size_t)tok * a.H * hd + head * hd;
                d[i] = o1;
                d[i + half] = o2;
            } else {
                const int kh = head - a.H;
                uint16_t* d = a.k_cache + (((size_t)row * a.KVH + kh) * a.Lmax + slot) * hd;
                d[i] = o1;
                d[i + half] = o2;
            }
        } else {
            const int vh = head - a.H - a.KVH;
            uint16_t* d = a.v_cache + (((size_t)row * a.KVH + vh) * a.Lmax + slot) * hd;
            d[i] = hs[i];
            d[i + half] = hs[i + half];
        }
    }
}

extern "C" int umoe_qkv_mrope_kvappend(const umoe_rope_args* a, umoe_stream_t stream) {
    UMOE_REQUIRE(a && a->qkv && a->cos_tab && a->sin_tab && a->pos3 && a->kv_pos && a->q_out && a->k_cache && a->v_cache,
                 "umoe_qkv_mrope_kvappend: null argument");
    UMOE_REQUIRE(a->hd % 2 == 0 && a->sec0 + a->sec1 + a->sec2 == a->hd / 2 && a->T >= 1 && a->n_tok % a->T == 0,
                 "umoe_qkv_mrope_kvappend: bad head_dim/sections/T (hd=%d sections=%d+%d+%d T=%d)", a->hd, a->sec0,
                 a->sec1, a->sec2, a->T);
    if (a->n_tok == 0) return 0;
    rope_append_kernel<<<dim3((unsigned)a->n_tok), 256, 0, (hipStream_t)stream>>>(*a);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------ attention
// mRoPE of 32 consecutive dims [32*h4, 32*h4+32) of one head, straight from the raw (bias-added) QKV row:
// out = x*cos + rotate_half(x)*sin with every product and the sum rounded to bf16 (as torch does on bf16 tensors).
// `own` = the 32 dims, `par` = their rotate_half partners (dims +-64); chunks of 8 dims never straddle an mRoPE section
// (host checks sec0 % 8 == 0 and (sec0+sec1) % 8 == 0).
struct rope_regs { uint4 x[4], y[4], c[4], s[4]; };
__device__ __forceinline__ void rope32_load(const uint16_t* head_raw, int h4, const umoe_attn_args& a, int p0, int p1, int p2,
                                            rope_regs& r) {
    const bool first = h4 < 2;
    const uint16_t* own = head_raw + h4 * 32;
    const uint16_t* par = head_raw + (first ? h4 * 32 + 64 : h4 * 32 - 64);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const int i0 = (h4 & 1) * 32 + kb * 8;  // index into the half-dim cos/sin row
        const int pos = (i0 < a.sec0) ? p0 : (i0 < a.sec0 + a.sec1 ? p1 : p2);
        r.x[kb] = ld16(own + kb * 8);
        r.y[kb] = ld16(par + kb * 8);
        r.c[kb] = ld16(a.cos_tab + (size_t)pos * 64 + i0);
        r.s[kb] = ld16(a.sin_tab + (size_t)pos * 64 + i0);
    }
}
__device__ __forceinline__ void rope32_math(const rope_regs& r, int h4, uint4 (&out)[4]) {
    const bool first = h4 < 2;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        float x[8], y[8], c[8], sn[8], o[8];
        unpack8(r.x[kb], x);
        unpack8(r.y[kb], y);
        unpack8(r.c[kb], c);
        unpack8(r.s[kb], sn);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rbf(x[j] * c[j]) + rbf((first ? -y[j] : y[j]) * sn[j]);
        out[kb] = pack8(o);
    }
}
__device__ __forceinline__ void rope32(const uint16_t* head_raw, int h4, const umoe_attn_args& a, int p0, int p1, int p2,
                                       uint4 (&out)[4]) {
    const bool first = h4 < 2;
    const uint16_t* own = head_raw + h4 * 32;
    const uint16_t* par = head_raw + (first ? h4 * 32 + 64 : h4 * 32 - 64);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const int i0 = (h4 & 1) * 32 + kb * 8;  // index into the half-dim cos/sin row
        const int pos = (i0 < a.sec0) ? p0 : (i0 < a.sec0 + a.sec1 ? p1 : p2);
        float x[8], y[8], c[8], sn[8], o[8];
        unpack8(ld16(own + kb * 8), x);
        unpack8(ld16(par + kb * 8), y);
        unpack8(ld16(a.cos_tab + (size_t)pos * 64 + i0), c);
        unpack8(ld16(a.sin_tab + (size_t)pos * 64 + i0), sn);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rbf(x[j] * c[j]) + rbf((first ? -y[j] : y[j]) * sn[j]);
        out[kb] = pack8(o);
    }
}

// hd == 128 only (4 MFMA k-steps; lane owns 2 value columns).  GP = GQA group size padded to a power of two.
// With a.qkv_raw set (decode, nq == 1) the kernel also applies mRoPE and appends the new K/V to the cache.
template <int GP>
__global__ __launch_bounds__(256) void attn_kernel(const umoe_attn_args a) {
    constexpr int HD = 128;
    __shared__ float p_lds[4][16][16];       // per wave: [key in tile][head]
    __shared__ float al_lds[4][16];          // per wave: rescale factor per head
    __shared__ float red_o[4][16][HD];       // cross-wave merge
    __shared__ float red_ml[4][16][2];
    TL_ENTER(7);
    const int split = blockIdx.x, kvh = blockIdx.y, qi = blockIdx.z;
    const int row = qi / a.nq, t = qi - row * a.nq;
    const int G = a.H / a.KVH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h4 = lane >> 4, c = lane & 15;

    const bool fuse = a.qkv_raw != nullptr;      // decode: rope + append fused, the new key is NOT read from the cache
    // every per-row scalar of the step state in ONE round trip (they are independent of each other; read where they were first used,
    // the three rope positions cost a round trip of their own behind the key range)
    int p0 = 0, p1 = 0, p2 = 0;
    if (fuse) {
        const int ntok = a.rows * a.nq;
        p0 = a.pos3[qi]; p1 = a.pos3[ntok + qi]; p2 = a.pos3[2 * ntok + qi];
    }
    const int kbeg_all = a.kv_start[row];
    const int kend_all = a.q_pos0[row] + t + (fuse ? 0 : 1);  // exclusive
    const int nkeys = max(kend_all - kbeg_all, 0);
    int chunk = (nkeys + a.splits - 1) / a.splits;
    chunk = (chunk + 15) & ~15;
    const int kbeg = kbeg_all + split * chunk;
    const int kend = min(kbeg + chunk, kend_all);

    const uint16_t* Kc = a.k_cache + ((size_t)row * a.KVH + kvh) * a.Lmax * HD;
    const uint16_t* Vc = a.v_cache + ((size_t)row * a.KVH + kvh) * a.Lmax * HD;

    // Q fragments (B operand): lane (h4, c = head in group) holds q[c][h4*32 + kb*8 .. +8]
    bf16x8_t qf[4];
    const int QKV_LD = (a.H + 2 * a.KVH) * HD;
    // first key tile of this wave: requested right behind the rope operands, so the (HBM-cold) K/V rows are in flight
    // while the rope arithmetic runs (a wave's loads return in issue order: operands first, then the tile)
    uint4 kfr_first[4];
    uint32_t vraw_first[16];
    const int k0_first = kbeg + wave * 16;
    auto load_tile = [&](uint4 (&kfr)[4], uint32_t (&vraw)[16], const int k0) {
        // unconditional, clamped into the cache row: a tile past the slice re-reads valid memory and is masked (or never
        // consumed) -- no branch around the loads, so the compiler can count them and wait for exactly what it needs
        const int key = max(min(k0 + c, kend - 1), 0);  // A operand row = lane&15 -> key index
        const uint16_t* kp = Kc + (size_t)key * HD + h4 * 32;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) kfr[kb] = ld16(kp + kb * 8);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
            vraw[kk] = *reinterpret_cast<const uint32_t*>(Vc + (size_t)max(min(k0 + kk, kend - 1), 0) * HD + 2 * lane);
    };
    if (fuse) {
        uint4 u[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) u[kb] = make_uint4(0, 0, 0, 0);
        // straight-line code (no branch around any load: lanes beyond the group re-read head 0 and are zeroed below), so
        // the compiler counts the loads and the rope arithmetic waits for its 16 operands only, not for the K/V tile
        rope_regs rr;
        rope32_load(a.qkv_raw + (size_t)qi * QKV_LD + (size_t)(kvh * G + (c < G ? c : 0)) * HD, h4, a, p0, p1, p2, rr);
        __builtin_amdgcn_sched_barrier(0);
        load_tile(kfr_first, vraw_first, k0_first);
        __builtin_amdgcn_sched_barrier(0);
        rope32_math(rr, h4, u);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) qf[kb] = __builtin_bit_cast(bf16x8_t, c < G ? u[kb] : make_uint4(0, 0, 0, 0));
    } else {
        load_tile(kfr_first, vraw_first, k0_first);
        const uint16_t* qp = a.q + ((size_t)qi * a.H + kvh * G + c) * HD + h4 * 32;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            uint4 u = make_uint4(0, 0, 0, 0);
            if (c < G) u = ld16(qp + kb * 8);
            qf[kb] = __builtin_bit_cast(bf16x8_t, u);
        }
    }
    TL_MARK(7, 4);
    // running state: this lane's head is c (for m, l); O for ALL heads of the group on 2 columns
    float m_run = -INFINITY, l_run = 0.f;
    float o[GP][2];
#pragma unroll
    for (int g = 0; g < GP; ++g) o[g][0] = o[g][1] = 0.f;

    auto process_tile = [&](const uint4 (&kfr)[4], const uint32_t (&vraw)[16], const int k0, const int kend) {
        // S tile: D[key = 4*h4 + r][head = c]
        f32x4_t sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
            sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kfr[kb]), qf[kb], sacc, 0, 0, 0);
        float sv[4];
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = k0 + 4 * h4 + r;
            sv[r] = (key < kend) ? sacc[r] * a.scale : -INFINITY;
            tmax = fmaxf(tmax, sv[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pv = (sv[r] == -INFINITY) ? 0.f : __expf(sv[r] - m_new);
            psum += pv;
            p_lds[wave][4 * h4 + r][c] = pv;
        }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (h4 == 0) al_lds[wave][c] = alpha;
        __builtin_amdgcn_wave_barrier();
        // P.V: lane owns value columns 2*lane, 2*lane+1
#pragma unroll
        for (int g = 0; g < GP; ++g) {
            const float al = al_lds[wave][g];
            o[g][0] *= al;
            o[g][1] *= al;
        }
        // keys beyond the slice carry p = 0 (masked above), so all 16 rows can be accumulated unconditionally;
        // P comes back from LDS as 4 x 16-byte broadcasts per key (all lanes read the same address)
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float v0 = __uint_as_float(vraw[kk] << 16), v1 = __uint_as_float(vraw[kk] & 0xffff0000u);
            float pv[GP];
            if (GP >= 4) {
                const float4* pr4 = reinterpret_cast<const float4*>(&p_lds[wave][kk][0]);
#pragma unroll
                for (int q4 = 0; q4 < GP / 4; ++q4) {
                    const float4 t4 = pr4[q4];
                    pv[4 * q4] = t4.x; pv[4 * q4 + 1] = t4.y; pv[4 * q4 + 2] = t4.z; pv[4 * q4 + 3] = t4.w;
                }
            } else {
#pragma unroll
                for (int g = 0; g < GP; ++g) pv[g] = p_lds[wave][kk][g];
            }
#pragma unroll
            for (int g = 0; g < GP; ++g) {   // heads >= G carry zero queries -> p is finite garbage-free, o unused
                o[g][0] += pv[g] * v0;
                o[g][1] += pv[g] * v1;
            }
            if ((kk & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // keep the unrolled body from hoisting all LDS reads
        }
        __builtin_amdgcn_wave_barrier();
    };
    // two tiles in flight: the next tile is requested before the current one is consumed
    {
        int k0 = k0_first;
        bool have = k0 < kend;
        while (have) {
            const int k1 = k0 + 64;
            const bool nb = k1 < kend;
            uint4 kB[4];
            uint32_t vB[16];
            load_tile(kB, vB, k1);
            process_tile(kfr_first, vraw_first, k0, kend);
            if (!nb) break;
            const int k2 = k1 + 64;
            have = k2 < kend;
            load_tile(kfr_first, vraw_first, k2);
            process_tile(kB, vB, k1, kend);
            k0 = k2;
        }
    }
    if (fuse && split == a.splits - 1 && wave == 0) {
        // the new token: K roped from the raw QKV row (lanes c == 0 hold its 4 x 32 dims), V raw; one extra 1-key tile,
        // and this wave is the single writer of cache slot q_pos0 for (row, kvh)
        const int slot = a.q_pos0[row];
        const uint16_t* kraw = a.qkv_raw + (size_t)qi * QKV_LD + (size_t)(a.H + kvh) * HD;
        const uint16_t* vrow = a.qkv_raw + (size_t)qi * QKV_LD + (size_t)(a.H + a.KVH + kvh) * HD;
        uint4 kfr[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) kfr[kb] = make_uint4(0, 0, 0, 0);
        if (c == 0) {
            rope32(kraw, h4, a, p0, p1, p2, kfr);
            uint16_t* kd = const_cast<uint16_t*>(a.k_cache) + (((size_t)row * a.KVH + kvh) * a.Lmax + slot) * HD + h4 * 32;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) st16(kd + kb * 8, kfr[kb]);
        }
        uint32_t vraw[16];
        const uint32_t vnew = *reinterpret_cast<const uint32_t*>(vrow + 2 * lane);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) vraw[kk] = vnew;
        *reinterpret_cast<uint32_t*>(const_cast<uint16_t*>(a.v_cache) + (((size_t)row * a.KVH + kvh) * a.Lmax + slot) * HD + 2 * lane) = vnew;
        process_tile(kfr, vraw, slot, slot + 1);   // keys slot+1.. are masked (p = 0)
    }

    TL_MARK(7, 5);
    // ---- merge the 4 waves --------------------------------------------------------------------
    if (h4 == 0) {
        red_ml[wave][c][0] = m_run;
        red_ml[wave][c][1] = l_run;
    }
#pragma unroll
    for (int g = 0; g < GP; ++g) {
        red_o[wave][g][2 * lane] = o[g][0];
        red_o[wave][g][2 * lane + 1] = o[g][1];
    }
    __syncthreads();
    // thread -> (head g, 8 columns): 16 heads x 16 column-groups
    const int g = tid >> 4, cg = tid & 15;
    if (g < G) {
        float mm = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) mm = fmaxf(mm, red_ml[w][g][0]);
        float L = 0.f, acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = red_ml[w][g][0];
            const float sc = (mw == -INFINITY) ? 0.f : __expf(mw - mm);
            L += sc * red_ml[w][g][1];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += sc * red_o[w][g][cg * 8 + j];
        }
        const int head = kvh * G + g;
        float* po = a.part_o + (((size_t)qi * a.H + head) * a.splits + split) * HD + cg * 8;
        float* pm = a.part_ml + (((size_t)qi * a.H + head) * a.splits + split) * 2;
        if (a.splits == 1) {
            // one key split: this workgroup holds the whole softmax -- the merge of umoe_attn_combine over ONE partial is o / l
            // (its scale exp(m - m) is exactly 1), so the output is written here and the combine launch is skipped
            uint16_t y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = f2bf(L > 0.f ? acc[j] / L : 0.f);
            st16(a.out + ((size_t)qi * a.H + head) * HD + cg * 8,
                 make_uint4((uint32_t)y[0] | ((uint32_t)y[1] << 16), (uint32_t)y[2] | ((uint32_t)y[3] << 16),
                            (uint32_t)y[4] | ((uint32_t)y[5] << 16), (uint32_t)y[6] | ((uint32_t)y[7] << 16)));
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) po[j] = acc[j];
            if (cg == 0) {
                pm[0] = mm;
                pm[1] = L;
            }
        }
    }
    // (Measured and removed, DESIGN.md 4a: the LAST split to finish merging in this same launch -- write-through partials, a ticket,
    //  a coherent re-read are three dependent trips to the coherence point, 6.5 us against the 4.9 us combine launch; and a form with
    //  no key split across workgroups at all -- 8-wave workgroups splitting the keys over their waves: two dependent HBM round trips
    //  per wave where the split kernel makes one, 10.7 / 17.6 us against 8.3 / 10.3 us at 305 / 814 cached keys.)
    TL_EXIT(7);
}

// SP > 0: the split count is a compile-time constant and EVERY partial is requested before the first use
// (the partials were written by other XCDs: each dependent load is a trip to the Infinity Cache)
template <int SP>
__global__ __launch_bounds__(128) void attn_combine_kernel(const umoe_attn_args a) {
    constexpr int HD = 128;
    TL_ENTER(8);
    const int head = blockIdx.x, qi = blockIdx.y, d = threadIdx.x;
    const int splits = SP > 0 ? SP : a.splits;
    const float* pm = a.part_ml + ((size_t)qi * a.H + head) * splits * 2;
    const float* po = a.part_o + ((size_t)qi * a.H + head) * splits * HD;
    float L = 0.f, acc = 0.f;
    if (SP > 0) {
        float2 ml[SP > 0 ? SP : 1];
        float ov[SP > 0 ? SP : 1];
#pragma unroll
        for (int s = 0; s < SP; ++s) {
            ml[s] = *reinterpret_cast<const float2*>(pm + 2 * s);
            ov[s] = po[(size_t)s * HD + d];
        }
        float mm = -INFINITY;
#pragma unroll
        for (int s = 0; s < SP; ++s) mm = fmaxf(mm, ml[s].x);
#pragma unroll
        for (int s = 0; s < SP; ++s) {
            const float sc = (ml[s].x == -INFINITY) ? 0.f : __expf(ml[s].x - mm);
            L += sc * ml[s].y;
            acc += sc * ov[s];
        }
    } else {
        float mm = -INFINITY;
        for (int s = 0; s < splits; ++s) mm = fmaxf(mm, pm[2 * s]);
        for (int s = 0; s < splits; ++s) {
            const float ms = pm[2 * s];
            const float sc = (ms == -INFINITY) ? 0.f : __expf(ms - mm);
            L += sc * pm[2 * s + 1];
            acc += sc * po[(size_t)s * HD + d];
        }
    }
    a.out[((size_t)qi * a.H + head) * HD + d] = f2bf(L > 0.f ? acc / L : 0.f);
    TL_EXIT(8);
}
UMOE_TL_SETTER(attn)
static void launch_attn_combine(const umoe_attn_args* a, dim3 grid, hipStream_t s) {
    switch (a->splits) {
        case 1: attn_combine_kernel<1><<<grid, 128, 0, s>>>(*a); break;
        case 2: attn_combine_kernel<2><<<grid, 128, 0, s>>>(*a); break;
        case 4: attn_combine_kernel<4><<<grid, 128, 0, s>>>(*a); break;
        case 8: attn_combine_kernel<8><<<grid, 128, 0, s>>>(*a); break;
        default: attn_combine_kernel<0><<<grid, 128, 0, s>>>(*a);
    }
}
static void launch_attn(const umoe_attn_args* a, dim3 grid, hipStream_t s) {
    const int G = a->H / a->KVH;
    if (G <= 1) attn_kernel<1><<<grid, 256, 0, s>>>(*a);
    else if (G <= 2) attn_kernel<2><<<grid, 256, 0, s>>>(*a);
    else if (G <= 4) attn_kernel<4><<<grid, 256, 0, s>>>(*a);
    else if (G <= 8) attn_kernel<8><<<grid, 256, 0, s>>>(*a);
    else attn_kernel<16><<<grid, 256, 0, s>>>(*a);
}

extern "C" int umoe_attn_decode(const umoe_attn_args* a, umoe_stream_t stream) {
    UMOE_REQUIRE(a && (a->q || a->qkv_raw) && a->k_cache && a->v_cache && a->kv_start && a->q_pos0 && a->part_o && a->part_ml && a->out,
                 "umoe_attn_decode: null argument");
    if (a->qkv_raw) {
        UMOE_REQUIRE(a->nq == 1 && a->cos_tab && a->sin_tab && a->pos3, "umoe_attn_decode: fused rope needs nq == 1 and rope tables");
        UMOE_REQUIRE(a->sec0 % 8 == 0 && (a->sec0 + a->sec1) % 8 == 0 && a->sec0 + a->sec1 + a->sec2 == 64,
                     "umoe_attn_decode: fused rope needs mRoPE sections on multiples of 8 (got %d,%d,%d)", a->sec0, a->sec1, a->sec2);
    }
    UMOE_REQUIRE(a->hd == 128, "umoe_attn_decode: head_dim must be 128 (got %d)", a->hd);
    UMOE_REQUIRE(a->KVH > 0 && a->H % a->KVH == 0 && a->H / a->KVH <= 16,
                 "umoe_attn_decode: GQA group must be <= 16 (H=%d KVH=%d)", a->H, a->KVH);
    UMOE_REQUIRE(a->splits >= 1 && a->rows > 0 && a->nq > 0, "umoe_attn_decode: bad splits/rows/nq");
    UMOE_REQUIRE((long)a->rows * a->nq <= 65535 * 1L * 65535, "umoe_attn_decode: too many queries");
    hipStream_t s = (hipStream_t)stream;
    const unsigned nqi = (unsigned)(a->rows * a->nq);
    // grid.z <= 65535: fold large query counts
    UMOE_REQUIRE(nqi <= 65535u * 32u, "umoe_attn_decode: too many query tokens (%u)", nqi);
    if (nqi <= 65535u) {
        launch_attn(a, dim3((unsigned)a->splits, (unsigned)a->KVH, nqi), s);
        UMOE_LAUNCH_CHECK();
        if (a->splits > 1) {
            launch_attn_combine(a, dim3((unsigned)a->H, nqi), s);
            UMOE_LAUNCH_CHECK();
        }
    } else {
        // process row by row (prefill with very long prompts)
        for (int r = 0; r < a->rows; ++r) {
            umoe_attn_args b = *a;
            b.rows = 1;
            b.q = a->q + (size_t)r * a->nq * a->H * a->hd;
            b.out = a->out + (size_t)r * a->nq * a->H * a->hd;
            b.kv_start = a->kv_start + r;
            b.q_pos0 = a->q_pos0 + r;
            b.k_cache = a->k_cache + (size_t)r * a->KVH * a->Lmax * a->hd;
            b.v_cache = a->v_cache + (size_t)r * a->KVH * a->Lmax * a->hd;
            b.part_o = a->part_o;
            b.part_ml = a->part_ml;
            UMOE_REQUIRE(a->nq <= 65535, "umoe_attn_decode: nq too large");
            launch_attn(&b, dim3((unsigned)a->splits, (unsigned)a->KVH, (unsigned)a->nq), s);
            UMOE_LAUNCH_CHECK();
            if (a->splits > 1) launch_attn_combine(&b, dim3((unsigned)a->H, (unsigned)a->nq), s);
            UMOE_LAUNCH_CHECK();
        }
    }
    return 0;
}

// causal prefill over the cache (nq = T query tokens per row): same kernel family, key slices of one split by default


// ------------------------------------------------------------------------------------ prefill / training attention (MFMA)
// Causal GQA attention with many queries per row (nq >= 16): flash-attention forward on the matrix cores.
//   grid = (16-query tiles, kv heads, rows); one wave per query head of the GQA group (G <= 8 waves), all waves share the
//   K / V tiles of their kv head in LDS (64 keys per step, double buffered, 256-byte rows with the 16-byte chunks XOR-swizzled
//   by row & 15: the ds_read_b128 K operand reads are conflict-free).
//   S^T[key][query] = K_tile (A operand) x Q (B operand, registers) -> online softmax per query (fp32; row max / sum across
//   the four 16-lane groups by two shuffles) -> P^T stays in registers AS the B operand of the second product (a lane holds
//   4 consecutive keys of each 16-key tile = one 8-key slice of a 32-key MFMA step) -> O^T[d][query] += V^T x P^T with the
//   V^T operand read straight out of the row-major V tile by ds_read_b64_tr_b16 (hardware transpose; probe:
//   scripts/probe/tr_read_probe.hip).  Reference arithmetic: eager attention of the transformers dependency (fp32 softmax,
//   probabilities cast to bf16 before P V), as oracle/decode.py restates it.
typedef short v4s_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int kv_off(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }

template <int GP>
__global__ __launch_bounds__(64 * GP) void attn_prefill_kernel(const umoe_attn_args a) {
    constexpr int HD = 128, KT = 64;                       // keys per step
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][K 16 KiB | V 16 KiB]
    const int t0 = blockIdx.x * 16, kvh = blockIdx.y, row = blockIdx.z;
    const int G = a.H / a.KVH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 4, c16 = lane & 15;
    const int head = kvh * G + wave;
    const bool live_wave = wave < G;
    const int kv0 = a.kv_start[row];
    const int qp0 = a.q_pos0[row] + t0;                    // absolute position of this tile's first query
    const int kmax = min(qp0 + 15, a.q_pos0[row] + a.nq - 1);   // last key any query of the tile may see
    const uint16_t* Kc = a.k_cache + ((size_t)row * a.KVH + kvh) * a.Lmax * HD;
    const uint16_t* Vc = a.v_cache + ((size_t)row * a.KVH + kvh) * a.Lmax * HD;

    // Q fragments: lane (h, q = c16) holds q[t0 + q][kb*32 + h*8 .. +8]
    bf16x8_t qf[4];
    {
        const int t = min(t0 + c16, a.nq - 1);
        const uint16_t* qp = a.q + (((size_t)row * a.nq + t) * a.H + (live_wave ? head : 0)) * HD + h * 8;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) qf[kb] = __builtin_bit_cast(bf16x8_t, ld16(qp + kb * 32));
    }
    const int my_qpos = qp0 + c16;                          // this lane's query position (column of S^T)
    const bool q_ok = t0 + c16 < a.nq;

    f32x4_t acc_o[8];
#pragma unroll
    for (int db = 0; db < 8; ++db) acc_o[db] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    // K / V tile staging through registers: 2 tiles x 64 rows x 16 chunks = 2048 chunks over 64*GP threads
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
            if (key <= kmax) stg[n] = ld16(base + (size_t)key * HD + c * 8);   // kmax < Lmax
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int n = 0; n < NLD; ++n) {
            const int idx = tid + n * NTH;
            const int r = (idx >> 4) & 63, c = idx & 15;
            st16(smem + buf * 32768 + (idx >> 10) * 16384 + kv_off(r, c), stg[n]);
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
            // ---- S^T = K Q^T for the four 16-key tiles ----
            f32x4_t sacc[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                sacc[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const bf16x8_t kf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Kt + kv_off(kt * 16 + c16, kb * 4 + h)));
                    sacc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kb], sacc[kt], 0, 0, 0);
                }
            }
            // ---- online softmax over the 64 keys of this step (lane: query c16, keys kt*16 + 4h + r) ----
            float sv[4][4];
            float tmax = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + kt * 16 + 4 * h + r;
                    const bool ok = q_ok && key >= kv0 && key <= my_qpos;
                    sv[kt][r] = ok ? sacc[kt][r] * a.scale : -INFINITY;
                    tmax = fmaxf(tmax, sv[kt][r]);
                }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m_run, tmax);
            const float alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_new);
            float psum = 0.f;
            bf16x8_t pf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint16_t pb[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float s = sv[2 * ks + (j >> 2)][j & 3];
                    const float pv = (s == -INFINITY) ? 0.f : __expf(s - m_new);
                    psum += pv;
                    pb[j] = f2bf(pv);
                }
                uint4 u;
                u.x = (uint32_t)pb[0] | ((uint32_t)pb[1] << 16); u.y = (uint32_t)pb[2] | ((uint32_t)pb[3] << 16);
                u.z = (uint32_t)pb[4] | ((uint32_t)pb[5] << 16); u.w = (uint32_t)pb[6] | ((uint32_t)pb[7] << 16);
                pf[ks] = __builtin_bit_cast(bf16x8_t, u);
            }
            psum += __shfl_xor(psum, 16, 64);
            psum += __shfl_xor(psum, 32, 64);
            l_run = l_run * alpha + psum;
            m_run = m_new;
            // ---- O^T = O^T * alpha + V^T P^T ----
            const int tq = c16 >> 2, tp = c16 & 3;   // transposing read: this lane supplies row tq, columns 4 tp .. 4 tp + 3 of its group's block
#pragma unroll
            for (int db = 0; db < 8; ++db) {
                acc_o[db][0] *= alpha; acc_o[db][1] *= alpha; acc_o[db][2] *= alpha; acc_o[db][3] *= alpha;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int rA = (2 * ks) * 16 + 4 * h + tq, rB = rA + 16;
                    const int cch = db * 2 + (tp >> 1), sub = (tp & 1) * 8;
                    const v4s_t va = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) v4s_t*)(Vt + kv_off(rA, cch) + sub));
                    const v4s_t vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) v4s_t*)(Vt + kv_off(rB, cch) + sub));
                    typedef short v8s_t __attribute__((ext_vector_type(8)));
                    const v8s_t v8 = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
                    acc_o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, v8), pf[ks], acc_o[db], 0, 0, 0);
                }
            }
        }
        if (more) lstore(buf ^ 1);
        // keeps the transposing-read builtins above the barrier (cause and argument: umoe_attn_bwd.hip, TR_PIN8): the accumulators
        // they feed pass through a volatile asm, which cannot move across s_barrier
        asm volatile("" ::: "memory");
        asm volatile("" : "+v"(acc_o[0]), "+v"(acc_o[1]), "+v"(acc_o[2]), "+v"(acc_o[3]), "+v"(acc_o[4]), "+v"(acc_o[5]), "+v"(acc_o[6]), "+v"(acc_o[7]));
        __syncthreads();
    }
    // ---- output: lane (h, q): d = db*16 + 4h + r of query t0 + q ----
    if (live_wave && q_ok) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        if (a.lse_out && h == 0) a.lse_out[((size_t)row * a.nq + t0 + c16) * a.H + head] = l_run > 0.f ? m_run + __logf(l_run) : INFINITY;
        uint16_t* o = a.out + (((size_t)row * a.nq + t0 + c16) * a.H + head) * HD + 4 * h;
#pragma unroll
        for (int db = 0; db < 8; ++db) {
            const uint32_t lo = (uint32_t)f2bf(acc_o[db][0] * inv) | ((uint32_t)f2bf(acc_o[db][1] * inv) << 16);
            const uint32_t hi = (uint32_t)f2bf(acc_o[db][2] * inv) | ((uint32_t)f2bf(acc_o[db][3] * inv) << 16);
            *reinterpret_cast<uint2*>(o + db * 16) = make_uint2(lo, hi);
        }
    }
}

static int prefill_mfma_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("UMOE_ATTN_PREFILL_MFMA");
        v = e ? atoi(e) : 1;
    }
    return v;
}

template <int GP>
static int launch_attn_prefill(const umoe_attn_args* a, hipStream_t s) {
    static bool configured = false;
    if (!configured) {
        UMOE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_prefill_kernel<GP>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        configured = true;
    }
    dim3 grid((unsigned)ceil_div(a->nq, 16), (unsigned)a->KVH, (unsigned)a->rows);
    attn_prefill_kernel<GP><<<grid, 64 * GP, 65536, s>>>(*a);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// returns 1 when the shape is not handled here (caller falls back to the split-key kernel)
static int attn_prefill_mfma(const umoe_attn_args* a, hipStream_t s) {
    const int G = a->KVH > 0 ? a->H / a->KVH : 0;
    if (!prefill_mfma_enabled() || a->nq < 16 || a->hd != 128 || G < 1 || G > 8 || a->qkv_raw || ceil_div(a->nq, 16) > 65535 || a->rows > 65535)
        return 1;
    if (G <= 1) return launch_attn_prefill<1>(a, s);
    if (G <= 2) return launch_attn_prefill<2>(a, s);
    if (G <= 4) return launch_attn_prefill<4>(a, s);
    return launch_attn_prefill<8>(a, s);
}

// causal prefill over the cache (nq = T query tokens per row): MFMA flash kernel for >= 16 queries per row, else the
// split-key kernel family of the decode path
extern "C" int umoe_attn_prefill_fwd(const umoe_attn_args* a, umoe_stream_t stream) {
    UMOE_REQUIRE(a && !a->qkv_raw, "umoe_attn_prefill_fwd: rope fusion is decode-only; run umoe_qkv_mrope_kvappend first");
    UMOE_REQUIRE(a->q && a->k_cache && a->v_cache && a->kv_start && a->q_pos0 && a->out, "umoe_attn_prefill_fwd: null argument");
    const int rc = attn_prefill_mfma(a, (hipStream_t)stream);
    if (rc <= 0) return rc;
    return umoe_attn_decode(a, stream);
}
