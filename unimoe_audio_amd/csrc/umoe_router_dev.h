// Device side of the Top-P router (shared by umoe_router.hip and the fused launch in umoe_gemm.hip); see umoe_router.hip for
// the arithmetic contract.  Include after umoe_common.h.
#pragma once

// ---- uniform gathers: lane j's value read into a scalar with v_readlane (no LDS crossbar traffic) -------------
#define RLF(v, j) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (j)))
#define RLI(v, j) __builtin_amdgcn_readlane((v), (j))

// all[j] = value of lane j for j < n, `fill` otherwise (j is a compile-time constant after unrolling).
// NC > 0: n is the compile-time constant NC (no guards, NC readlanes); NC == 0: runtime n, 16-wide with guards.
template <int NC>
__device__ __forceinline__ void gather16(float v, int n, float fill, float (&all)[UMOE_MAXE]) {
#pragma unroll
    for (int j = 0; j < UMOE_MAXE; ++j) {
        if (NC > 0) all[j] = (j < NC) ? RLF(v, j) : fill;
        else all[j] = (j < n) ? RLF(v, j) : fill;
    }
}
#define LIM(NC, n) ((NC) > 0 ? (NC) : (n))

// softmax over lanes [0, n): max, deterministic exp, SEQUENTIAL fp32 sum in index order, reciprocal multiply,
// round to T -- the arithmetic contract shared with oracle/router_oracle.c
template <int NC, int TB>
__device__ __forceinline__ float lane_softmax(float x, int n, int lane) {
    const float xm = (lane < LIM(NC, n)) ? x : -INFINITY;
    float xs[UMOE_MAXE];
    gather16<NC>(xm, n, -INFINITY, xs);
    float m = xs[0];
#pragma unroll
    for (int j = 1; j < UMOE_MAXE; ++j)
        if (NC == 0 || j < NC) m = (xs[j] > m) ? xs[j] : m;
    const float e = (lane < LIM(NC, n)) ? umoe_exp_det(xm - m) : 0.f;
    float es[UMOE_MAXE];
    gather16<NC>(e, n, 0.f, es);
    float s = es[0];
#pragma unroll
    for (int j = 1; j < UMOE_MAXE; ++j)
        if (j < LIM(NC, n)) s = s + es[j];
    const float r = 1.0f / s;
    return TB ? rbf(e * r) : e * r;
}

// sum over the 64 lanes of 16 per-lane accumulators in 17 exchanges (reduce-scatter, then 2 butterfly steps).
// Returns, on lane e < 16, the total of acc[e].  Fixed tree => run-to-run deterministic.
__device__ __forceinline__ float reduce16_to_lanes(float (&acc)[UMOE_MAXE], int lane) {
    {
        const bool hi = lane & 32;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float send = hi ? acc[j] : acc[j + 8];
            const float keep = hi ? acc[j + 8] : acc[j];
            acc[j] = keep + __shfl_xor(send, 32, 64);
        }
    }
    {
        const bool hi = lane & 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float send = hi ? acc[j] : acc[j + 4];
            const float keep = hi ? acc[j + 4] : acc[j];
            acc[j] = keep + __shfl_xor(send, 16, 64);
        }
    }
    {
        const bool hi = lane & 8;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float send = hi ? acc[j] : acc[j + 2];
            const float keep = hi ? acc[j + 2] : acc[j];
            acc[j] = keep + __shfl_xor(send, 8, 64);
        }
    }
    float v;
    {
        const bool hi = lane & 4;
        const float send = hi ? acc[0] : acc[1];
        const float keep = hi ? acc[1] : acc[0];
        v = keep + __shfl_xor(send, 4, 64);
    }
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    // lane L now holds column ((L>>5)&1)*8 + ((L>>4)&1)*4 + ((L>>3)&1)*2 + ((L>>2)&1); fetch column `lane`
    const int src = (((lane >> 3) & 1) << 5) | (((lane >> 2) & 1) << 4) | (((lane >> 1) & 1) << 3) | ((lane & 1) << 2);
    return __shfl(v, src, 64);
}

// ND / NF: compile-time n_dyn / n_fix (ND == 0: generic runtime sizes); TB: 1 = bf16 arithmetic type, 0 = fp32
template <int ND, int NF, int TB>
__device__ __forceinline__ int route_from_logits(const umoe_router_args& a, const int s, const int lane, float full TL_PARAM);

template <int ND, int NF, int TB>
__device__ __forceinline__ int route_token(const umoe_router_args& a, const int s, const int lane TL_PARAM) {
    const int n_dyn = ND > 0 ? ND : a.n_dyn;
    const int E = ND > 0 ? ND + NF : a.n_dyn + a.n_fix;
    constexpr int T = TB;
    constexpr int NE = ND > 0 ? ND + NF : 0;  // compile-time E (0 = runtime)

    // ---- logits: lane e <- column e ----------------------------------------------------------
    float full = -INFINITY;  // this lane's logit (valid for lane < E)
    if (a.logits_in) {
        if (lane < E)
            full = T ? bf2f(reinterpret_cast<const uint16_t*>(a.logits_in)[(size_t)s * E + lane])
                     : reinterpret_cast<const float*>(a.logits_in)[(size_t)s * E + lane];
    } else if (ND > 0 && a.D <= 2048 && (a.D & 511) == 0 && !a.x_noise) {
        // decode fast path: EVERY load of this token (row, norm weights, all E gate rows) is in flight before the
        // first use -- one memory latency instead of one per chunk (the gate weights are cold in HBM every layer)
        constexpr int NEc = ND > 0 ? ND + NF : 1;  // (the generic instantiation never takes this branch)
        const uint16_t* xr = a.x + (size_t)s * a.D;
        const int nch = a.D >> 9;  // 16-byte chunks per lane
        uint4 xv[4], nw[4], gwv[NEc][4];
#pragma unroll
        for (int n = 0; n < 4; ++n)
            if (n < nch) {
                xv[n] = ld16(xr + (lane + 64 * n) * 8);
                if (a.norm_w) nw[n] = ld16(a.norm_w + (lane + 64 * n) * 8);
            }
#pragma unroll
        for (int e = 0; e < NEc; ++e)
#pragma unroll
            for (int n = 0; n < 4; ++n)
                if (n < nch) gwv[e][n] = ld16(a.gate_w + (size_t)e * a.D + (lane + 64 * n) * 8);
        TL_MARK(5, 4);
        float rs = 1.f;
        if (a.norm_w) {
            float ss = 0.f;
#pragma unroll
            for (int n = 0; n < 4; ++n)
                if (n < nch) {
                    float f[8];
                    unpack8(xv[n], f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
                }
            ss = wave_sum(ss);
            rs = rsqrtf(ss / (float)a.D + a.rms_eps);
        }
        TL_MARK(5, 5);
        float acc[UMOE_MAXE];
#pragma unroll
        for (int e = 0; e < UMOE_MAXE; ++e) acc[e] = 0.f;
#pragma unroll
        for (int n = 0; n < 4; ++n)
            if (n < nch) {
                float f[8];
                unpack8(xv[n], f);
                if (a.norm_w) {
                    float w[8];
                    unpack8(nw[n], w);
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = rbf(w[j] * rbf(f[j] * rs));
                    xv[n] = pack8(f);
                }
                if (a.h_out) st16(a.h_out + (size_t)s * a.D + (lane + 64 * n) * 8, xv[n]);
#pragma unroll
                for (int e = 0; e < NEc; ++e) {
                    float w[8];
                    unpack8(gwv[e][n], w);
                    float d = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) d += f[j] * w[j];
                    acc[e] += d;
                }
            }
        const float mine = reduce16_to_lanes(acc, lane);
        if (lane < E) full = round_t(mine, T);
    } else {
        const uint16_t* xr = a.x + (size_t)s * a.D;
        const int nchunk = a.D >> 3;
        float rs = 1.f;
        if (a.norm_w) {
            float ss = 0.f;
            for (int c = lane; c < nchunk; c += 64) {
                float f[8];
                unpack8(ld16(xr + c * 8), f);
#pragma unroll
                for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
            }
            ss = wave_sum(ss);
            rs = rsqrtf(ss / (float)a.D + a.rms_eps);
        }
        float acc[UMOE_MAXE];
#pragma unroll
        for (int e = 0; e < UMOE_MAXE; ++e) acc[e] = 0.f;
        for (int c = lane; c < nchunk; c += 64) {
            float f[8];
            uint4 u = ld16(xr + c * 8);
            unpack8(u, f);
            if (a.norm_w) {
                float w[8];
                unpack8(ld16(a.norm_w + c * 8), w);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = rbf(w[j] * rbf(f[j] * rs));
                u = pack8(f);
            }
            if (a.x_noise) {   // input jitter in front of the fp32 gate (core.py:240-249): the product stays fp32
                const float4* nz = reinterpret_cast<const float4*>(a.x_noise + (size_t)s * a.D + c * 8);
                const float4 n0 = nz[0], n1 = nz[1];
                f[0] *= n0.x; f[1] *= n0.y; f[2] *= n0.z; f[3] *= n0.w;
                f[4] *= n1.x; f[5] *= n1.y; f[6] *= n1.z; f[7] *= n1.w;
            }
            if (a.h_out) st16(a.h_out + (size_t)s * a.D + c * 8, u);
#pragma unroll
            for (int e = 0; e < UMOE_MAXE; ++e)
                if (e < E) {
                    float w[8];
                    unpack8(ld16(a.gate_w + (size_t)e * a.D + c * 8), w);
                    float d = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) d += f[j] * w[j];
                    acc[e] += d;
                }
        }
        const float mine = reduce16_to_lanes(acc, lane);
        if (lane < E) full = round_t(mine, T);
    }
    TL_MARK(5, 6);
    return route_from_logits<ND, NF, TB>(a, s, lane, full TL_PASS);
}


template <int ND, int NF, int TB>
__device__ __forceinline__ int route_from_logits(const umoe_router_args& a, const int s, const int lane, float full TL_PARAM) {
    const int n_dyn = ND > 0 ? ND : a.n_dyn;
    const int E = ND > 0 ? ND + NF : a.n_dyn + a.n_fix;
    constexpr int T = TB;
    constexpr int NE = ND > 0 ? ND + NF : 0;  // compile-time E (0 = runtime)
    if (a.logits_out && lane < E) {
        if (T) reinterpret_cast<uint16_t*>(a.logits_out)[(size_t)s * E + lane] = f2bf(full);
        else reinterpret_cast<float*>(a.logits_out)[(size_t)s * E + lane] = full;
    }

    // ---- Top-P count (core.py:157-167) -------------------------------------------------------
    int k = a.fixed_top_k;
    if (a.top_p != 0.0f) {
        const float p = lane_softmax<ND, TB>(full, n_dyn, lane);
        float ps[UMOE_MAXE];
        gather16<ND>(p, n_dyn, -INFINITY, ps);
        // rank in descending order (ties: lower index first); only the sorted VALUES matter
        int rank = 0;
#pragma unroll
        for (int j = 0; j < UMOE_MAXE; ++j)
            if (j < LIM(ND, n_dyn)) rank += (ps[j] > p) || (ps[j] == p && j < lane);
        // value at sorted position t = p of the lane whose rank is t
        float mineS = 0.f;
#pragma unroll
        for (int j = 0; j < UMOE_MAXE; ++j)
            if (j < LIM(ND, n_dyn) && RLI(rank, j) == lane) mineS = ps[j];
        float sorted[UMOE_MAXE];
        gather16<ND>(mineS, n_dyn, 0.f, sorted);
        const float thr = round_t(a.top_p, T);
        int below = 0;
        if (T) {
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < UMOE_MAXE; ++t)
                if (t < LIM(ND, n_dyn)) {
                    acc = acc + sorted[t];
                    below += !(rbf(acc) >= thr);
                }
        } else {
            double acc = 0.0;
#pragma unroll
            for (int t = 0; t < UMOE_MAXE; ++t)
                if (t < LIM(ND, n_dyn)) {
                    acc = acc + (double)sorted[t];
                    below += !((float)acc >= thr);
                }
        }
        k = below + 1;
    }
    if (k > n_dyn) k = n_dyn;
    TL_MARK(5, 7);

    // ---- iterative arg-max mixer, eval branch (core.py:94-154, 262-282) -----------------------
    const float two_eps = round_t((float)(2.0 * a.jitter_eps), T);
    float masked = (lane < n_dyn) ? full : -INFINITY;
    float w = 0.f;
    int m = 0;
    for (int j = 0; j < k; ++j) {
        // max + lowest-index arg-max over the n_dyn columns (uniform scan of the gathered values)
        float ms[UMOE_MAXE];
        gather16<ND>(masked, n_dyn, -INFINITY, ms);
        float thr = ms[0];
        int ind = 0;
#pragma unroll
        for (int e = 1; e < UMOE_MAXE; ++e)
            if ((ND == 0 || e < ND) && ms[e] > thr) {
                thr = ms[e];
                ind = e;
            }
        const float af = fabsf(full), at = fabsf(thr);
        const float factor = af > at ? af : at;
        const float d = round_t(thr - full, T);
        const float q = round_t(d / factor, T);
        const float gate = (q > two_eps) ? -INFINITY : masked;
        const float gsm = lane_softmax<ND, TB>(gate, n_dyn, lane);
        if (a.gumbel) {
            // training branch (core.py:111-137): selected = arg-max(masked_gates + Gumbel noise) -- the sum is fp32 (fp32 noise) --,
            // weight = softmax multiplier of the SELECTED column * mask_for_one; lowest index on ties, like torch.max
            const float noisy = (lane < n_dyn) ? gate + a.gumbel[((size_t)s * n_dyn + j) * n_dyn + lane] : -INFINITY;
            float ns[UMOE_MAXE], ps[UMOE_MAXE];
            gather16<ND>(noisy, n_dyn, -INFINITY, ns);
            gather16<ND>(gsm, n_dyn, -INFINITY, ps);
            float nb = ns[0], pb = ps[0];
            int sel_j = 0, mi = 0;
#pragma unroll
            for (int e = 1; e < UMOE_MAXE; ++e)
                if (ND == 0 || e < ND) {
                    if (ns[e] > nb) { nb = ns[e]; sel_j = e; }
                    if (ps[e] > pb) { pb = ps[e]; mi = e; }
                }
            float mo = 0.f;
#pragma unroll
            for (int e = 0; e < UMOE_MAXE; ++e)
                if (e == sel_j) mo = ps[e];
            const bool one = (sel_j == mi) || (a.rand_u[(size_t)s * n_dyn + j] > 0.75f);
            const float factor = round_t(one ? (0.3333f + 0.6667f) : 0.3333f, T);     // torch.add(0.3333, mask, alpha=0.6667).type_as(gates)
            if (lane == sel_j) {
                w = round_t(mo * factor, T);
                m += 1;
                masked = -INFINITY;
            }
            if (lane == 0) {
                if (a.sel) a.sel[(size_t)s * n_dyn + j] = sel_j;
                if (a.round_factor) a.round_factor[(size_t)s * n_dyn + j] = factor;
            }
            continue;
        }
        if (lane == ind) {
            w = gsm;
            m += 1;
            masked = -INFINITY;
        }
        if (a.sel && lane == 0) a.sel[(size_t)s * n_dyn + j] = ind;
    }
    if (a.gumbel && a.round_factor && lane >= k && lane < n_dyn) a.round_factor[(size_t)s * n_dyn + lane] = 0.f;
    if (a.sel && lane >= k && lane < n_dyn) a.sel[(size_t)s * n_dyn + lane] = -1;
    TL_MARK(5, 8);

    // ---- renormalise / padding / shared always on (core.py:284-291) ---------------------------
    float ws[UMOE_MAXE];
    gather16<ND>(w, n_dyn, 0.f, ws);
    float sum = ws[0];
#pragma unroll
    for (int j = 1; j < UMOE_MAXE; ++j)
        if (j < LIM(ND, n_dyn)) sum = sum + ws[j];
    sum = round_t(sum, T);
    const float den = round_t(sum + 1e-6f, T);
    w = (lane < n_dyn) ? round_t(w / den, T) : 0.f;
    if (a.attn_mask) m *= (int)(a.attn_mask[s] != 0);
    if (lane >= n_dyn && lane < E) m = 1;

    // ---- global routing weight (core.py:178-193) ----------------------------------------------
    float gw = w;
    if ((ND > 0 ? NF : a.n_fix) > 0) {
        const float gl = (lane < E && m) ? full : -INFINITY;
        gw = lane_softmax<NE, TB>(gl, E, lane);
        float gs[UMOE_MAXE];
        gather16<ND>(gw, n_dyn, 0.f, gs);
        float ds = gs[0];
#pragma unroll
        for (int j = 1; j < UMOE_MAXE; ++j)
            if (j < LIM(ND, n_dyn)) ds = ds + gs[j];
        ds = round_t(ds, T);
        if (lane < n_dyn) gw = round_t(w * ds, T);
    }
    if (lane < E) {
        a.expert_mask[(size_t)s * E + lane] = m;
        if (a.global_w) a.global_w[(size_t)s * E + lane] = gw;
    }
    if (lane < n_dyn && a.routing_w) a.routing_w[(size_t)s * n_dyn + lane] = w;
    if (lane < a.n_real && a.moe_w) a.moe_w[(size_t)s * a.n_real + lane] = gw * (float)m;
    if (a.top_k && lane == 0) a.top_k[s] = k;
    return m;  // this lane's mask entry (lane < E)
}


// One token per 4 waves (threads 0..255 of the workgroup): the four waves split the RMSNorm and the gate GEMV over D, partial sums
// meet in LDS in fixed order, wave 0 then walks the serial routing chain.  D = 2048 or 4096.
// NORM_ONLY: stop after h_out = RMSNorm(x) (the same arithmetic, bit for bit, as the full body): the decode engine runs this as its
// own small launch in front of the expert GEMM and lets the full body (h_out = NULL) ride INSIDE the gate/up launch as extra
// workgroups -- the routing chain (4.4 us of serial work per token) leaves the critical path.
// pub_flag != NULL (riders of the dense decode gate/up launch, umoe_gemm_args.rider_pub): the normalised row h_out[s] is HANDED to
// the GEMM workgroups of the SAME launch -- stored write-through (sc1), drained by every storing wave in front of the body's own
// barrier, then lane 0 publishes `pub_epoch` in pub_flag[s] (hand-off form: cdna_hip_programming.md Guideline 16 R1).
template <int ND, int NF, int TB, bool NORM_ONLY>
__device__ __forceinline__ void router4_body(const umoe_router_args& a, const int s, const int tid, float* lds TL_PARAM,
                                             uint32_t* pub_flag = nullptr, uint32_t pub_epoch = 0, unsigned long long* rs_pub = nullptr) {

    constexpr int NEc = ND + NF;
    constexpr int T = TB;
    // caller-provided LDS, (4 + 4 * UMOE_MAXE) floats: the fused launch passes a piece of its ONE dynamic array -- a second
    // __shared__ object in that kernel made hipcc serialise the GEMM's weight stream against its LDS reads (+4.5 us per launch)
    float* ss_part = lds;
    float (*lg_part)[UMOE_MAXE] = reinterpret_cast<float (*)[UMOE_MAXE]>(lds + 4);
    const int lane = tid & 63, wave = tid >> 6;
    const int nch = a.D >> 11;  // 16-byte chunks per lane: 1 or 2
    const uint16_t* xr = a.x + (size_t)s * a.D;
    uint4 xv[2], nw[2], gwv[NEc][2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
        if (n < nch) {
            const int c = (n * 4 + wave) * 64 + lane;
            xv[n] = ld16(xr + c * 8);
            if (a.norm_w) nw[n] = ld16(a.norm_w + c * 8);
            if (!NORM_ONLY) {
#pragma unroll
                for (int e = 0; e < NEc; ++e) gwv[e][n] = ld16(a.gate_w + (size_t)e * a.D + c * 8);
            }
        }
    TL_MARK(5, 4);
    float rs = 1.f;
    if (a.norm_w) {
        float ss = 0.f;
#pragma unroll
        for (int n = 0; n < 2; ++n)
            if (n < nch) {
                float f[8];
                unpack8(xv[n], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
            }
        ss = wave_sum(ss);
        if (lane == 0) ss_part[wave] = ss;
        __syncthreads();
        ss = ((ss_part[0] + ss_part[1]) + ss_part[2]) + ss_part[3];
        rs = rsqrtf(ss / (float)a.D + a.rms_eps);
        // rs_pub (riders of the fused expert launch, second hand-off form): instead of the normalised ROW only the row's scale leaves
        // this workgroup -- ONE 8-byte write-through store {rs, epoch} (data and tag in one granule: no drain, no separate flag); the
        // GEMM workgroups loaded the raw row at launch and finish the norm themselves with exactly this rs (the same bits)
        if (rs_pub && tid == 0) {
            typedef uint32_t u32x2_g __attribute__((ext_vector_type(2)));
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(rs_pub, 0, 8 * UMOE_EP_PARTS, 0x00020000);
            const u32x2_g g2 = {(uint32_t)__float_as_int(rs), pub_epoch};
            __builtin_amdgcn_raw_buffer_store_b64(g2, rsrc, s * 8, 0, 16);
        }
    }
    TL_MARK(5, 5);
    float acc[UMOE_MAXE];
#pragma unroll
    for (int e = 0; e < UMOE_MAXE; ++e) acc[e] = 0.f;
#pragma unroll
    for (int n = 0; n < 2; ++n)
        if (n < nch) {
            const int c = (n * 4 + wave) * 64 + lane;
            float f[8];
            unpack8(xv[n], f);
            if (a.norm_w) {
                float w[8];
                unpack8(nw[n], w);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = rbf(w[j] * rbf(f[j] * rs));
                xv[n] = pack8(f);
            }
            if (a.h_out) {
                if (pub_flag) {
                    typedef uint32_t u32x4_pub __attribute__((ext_vector_type(4)));
                    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(a.h_out + (size_t)s * a.D, 0, a.D * 2, 0x00020000);
                    const u32x4_pub v4 = {xv[n].x, xv[n].y, xv[n].z, xv[n].w};
                    __builtin_amdgcn_raw_buffer_store_b128(v4, rsrc, c * 16, 0, 16);      // aux 16 = sc1 (agent scope write-through)
                } else {
                    st16(a.h_out + (size_t)s * a.D + c * 8, xv[n]);
                }
            }
            if (NORM_ONLY) continue;
#pragma unroll
            for (int e = 0; e < NEc; ++e) {
                float w[8];
                unpack8(gwv[e][n], w);
                float d = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) d += f[j] * w[j];
                acc[e] += d;
            }
        }
    if (NORM_ONLY) return;
    const float mine = reduce16_to_lanes(acc, lane);
    if (lane < UMOE_MAXE) lg_part[wave][lane] = mine;
    if (pub_flag) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its sc1 stores
    __syncthreads();
    if (pub_flag && tid < UMOE_FLAG_REPL)       // one store per replica of the flag line (umoe_common.h)
        __hip_atomic_store(reinterpret_cast<umoe_gu32*>(reinterpret_cast<uintptr_t>(pub_flag + tid * 16 + s)), pub_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave != 0) return;
    float full = -INFINITY;
    if (lane < NEc) full = round_t(((lg_part[0][lane] + lg_part[1][lane]) + lg_part[2][lane]) + lg_part[3][lane], T);
    TL_MARK(5, 6);
    route_from_logits<ND, NF, TB>(a, s, lane, full TL_PASS);
}
