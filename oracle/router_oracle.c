/*
 * router_oracle.c -- CPU restatement of the UniMoE-Audio DCMoE Top-P router.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product path: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.
 *
 * What it restates (reference = /root/reference, read as text):
 *   - Top-P expert count          utils/UniMoE_Audio_core.py:157-167
 *   - iterative arg-max mixer     utils/UniMoE_Audio_core.py:94-154 (eval branch,
 *                                 :116 `selected_experts = max_ind`) and its driver
 *                                 loop utils/UniMoE_Audio_core.py:262-282
 *   - renormalise / padding mask / shared-always-on   core.py:284-291
 *   - global routing weight       utils/UniMoE_Audio_core.py:178-193
 *   - weight * mask for the MoE layer                 core.py:447
 *
 * Arithmetic contract (what "bit-exact given identical logits" means here).
 * torch-CPU evaluates every elementwise op on a bf16 tensor in fp32 and rounds
 * the result to bf16 once; these were probed in the build container
 * (torch 2.10 CPU, 1.8M rows) and are followed here:
 *   softmax_T(x): m = max x; e_i = expf(x_i - m) in fp32; s = e_0 + e_1 + ...
 *                 (sequential fp32); r = 1/s; p_i = round_T(e_i * r)
 *   cumsum_T    : running sum in fp32 (T = bf16) or fp64 (T = fp32), each
 *                 output rounded to T
 *   x >= 0.7    : the Python scalar is cast to T first (bf16(0.7) = 0.69921875)
 *   arg-max ties: lowest index
 * The one place this file deliberately differs from torch is expf(): torch uses
 * a vectorised 1-ulp expf whose bits depend on the host ISA.  Here exp is a
 * fixed sequence of IEEE fp64 fma operations (exp_det below), rounded to fp32,
 * so that the GPU kernel can reproduce it bit for bit.  It is within 1 fp32 ulp
 * of torch's value; the measured effect on the integer outputs is recorded in
 * DESIGN.md (parity section) and checked by tests/test_oracle_golden.py.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define UMOE_MAX_E 32

/* ---- bf16 helpers ------------------------------------------------------- */
static inline float bf16_to_f32(uint16_t h) {
    uint32_t u = ((uint32_t)h) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline uint16_t f32_to_bf16(float f) { /* round to nearest even, NaN kept */
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float round_t(float v, int is_bf16) { return is_bf16 ? bf16_to_f32(f32_to_bf16(v)) : v; }

/* ---- deterministic exp: fixed fp64 fma sequence, result rounded to fp32 --- */
static inline float exp_det(float xf) {
    if (!(xf > -110.0f)) return (xf != xf) ? xf : 0.0f; /* -inf, very negative -> 0; NaN -> NaN */
    if (xf > 88.0f) xf = 88.0f + (xf - xf);             /* never reached from softmax (args <= 0) */
    const double x = (double)xf;
    const double LOG2E = 1.4426950408889634074;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    const double k = rint(x * LOG2E);
    double r = fma(k, -LN2_HI, x);
    r = fma(k, -LN2_LO, r);
    /* Horner, degree 13: sum r^i / i! */
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    int64_t ki = (int64_t)k;
    uint64_t bits = (uint64_t)(ki + 1023) << 52; /* 2^k, k in [-159, 127] */
    double scale;
    memcpy(&scale, &bits, 8);
    return (float)(p * scale);
}

/* softmax over n values already widened to fp32; -inf entries allowed */
static void softmax_t(const float* x, int n, int is_bf16, float* out) {
    float m = x[0];
    for (int i = 1; i < n; ++i)
        if (x[i] > m) m = x[i];
    float e[UMOE_MAX_E];
    for (int i = 0; i < n; ++i) e[i] = exp_det(x[i] - m);
    float s = e[0];
    for (int i = 1; i < n; ++i) s = s + e[i];
    const float r = 1.0f / s;
    for (int i = 0; i < n; ++i) out[i] = round_t(e[i] * r, is_bf16);
}

/* Top-P count, core.py:157-167 */
static int top_p_count(const float* logit, int n, float top_p, int is_bf16) {
    float p[UMOE_MAX_E];
    softmax_t(logit, n, is_bf16, p);
    /* descending insertion sort of values */
    for (int i = 1; i < n; ++i) {
        float v = p[i];
        int j = i - 1;
        while (j >= 0 && p[j] < v) {
            p[j + 1] = p[j];
            --j;
        }
        p[j + 1] = v;
    }
    const float thr = round_t(top_p, is_bf16);
    int below = 0;
    if (is_bf16) {
        float acc = 0.0f;
        for (int i = 0; i < n; ++i) {
            acc = acc + p[i];
            float c = round_t(acc, 1);
            if (!(c >= thr)) ++below;
        }
    } else {
        double acc = 0.0;
        for (int i = 0; i < n; ++i) {
            acc = acc + (double)p[i];
            float c = (float)acc;
            if (!(c >= thr)) ++below;
        }
    }
    return below + 1;
}

/*
 * Route S tokens.
 *  logits      [S][n_dyn + n_fix]  fp32 (logits_bf16 = 0) or bf16 bit patterns
 *  n_dyn       dynamic columns incl. null experts (9); n_real real routed (8); n_fix shared (2)
 *  top_p       0 => use fixed_top_k for every token (core.py:254-257)
 *  attn_mask   [S] 0/1 or NULL (core.py:286-288)
 * outputs
 *  top_k       [S] int64
 *  sel         [S][n_dyn] int32: expert picked at round j, -1 beyond k
 *  mask        [S][n_dyn + n_fix] int32 (core.py:259,282,288,291)
 *  route_w     [S][n_dyn] fp32 holding T-rounded values (core.py:284)
 *  global_w    [S][n_dyn + n_fix] fp32 holding T-rounded values (core.py:332)
 *  moe_w       [S][n_real] fp32 = global_w * mask (core.py:447)
 * returns 0, or -1 on bad sizes.
 */
int umoe_oracle_router(const void* logits, int logits_bf16, int S, int n_dyn, int n_real, int n_fix, float top_p,
                       int fixed_top_k, double jitter_eps, const uint8_t* attn_mask, int64_t* top_k, int32_t* sel,
                       int32_t* mask, float* route_w, float* global_w, float* moe_w) {
    const int E = n_dyn + n_fix;
    if (E > UMOE_MAX_E || n_dyn < 1 || n_real > n_dyn || S < 0) return -1;
    /* core.py:107 `> (2 * jitter_eps)`: the Python double is cast to T for the compare */
    const float two_eps_t = round_t((float)(2.0 * jitter_eps), logits_bf16);
    for (int s = 0; s < S; ++s) {
        float full[UMOE_MAX_E];
        for (int e = 0; e < E; ++e)
            full[e] = logits_bf16 ? bf16_to_f32(((const uint16_t*)logits)[(size_t)s * E + e])
                                  : ((const float*)logits)[(size_t)s * E + e];
        int k = (top_p != 0.0f) ? top_p_count(full, n_dyn, top_p, logits_bf16) : fixed_top_k;
        if (k > n_dyn) k = n_dyn;
        top_k[s] = k;

        float w[UMOE_MAX_E];
        int32_t m[UMOE_MAX_E];
        for (int e = 0; e < E; ++e) {
            w[e] = 0.0f;
            m[e] = 0;
        }
        float masked[UMOE_MAX_E];
        for (int e = 0; e < n_dyn; ++e) masked[e] = full[e];
        for (int j = 0; j < n_dyn; ++j) sel[(size_t)s * n_dyn + j] = -1;

        for (int j = 0; j < k; ++j) {
            /* core.py:105 max + arg-max (lowest index on ties) */
            int ind = 0;
            float thr = masked[0];
            for (int e = 1; e < n_dyn; ++e)
                if (masked[e] > thr) {
                    thr = masked[e];
                    ind = e;
                }
            /* core.py:106-109 relative-distance mask, T arithmetic */
            float gates[UMOE_MAX_E];
            const float athr = fabsf(thr);
            for (int e = 0; e < n_dyn; ++e) {
                float a = fabsf(full[e]);
                float factor = a > athr ? a : athr;
                float d = round_t(thr - full[e], logits_bf16);
                float q = round_t(d / factor, logits_bf16);
                gates[e] = (q > two_eps_t) ? -INFINITY : masked[e];
            }
            float g[UMOE_MAX_E];
            softmax_t(gates, n_dyn, logits_bf16, g); /* core.py:118 */
            w[ind] = g[ind];                         /* core.py:119, 279 */
            m[ind] += 1;                             /* core.py:275-276 */
            sel[(size_t)s * n_dyn + j] = ind;
            masked[ind] = -INFINITY; /* core.py:139-144 */
        }
        /* core.py:284 renormalise (sum in fp32, rounded to T) */
        float sum = 0.0f;
        for (int e = 0; e < n_dyn; ++e) sum = sum + w[e];
        sum = round_t(sum, logits_bf16);
        const float den = round_t(sum + 1e-6f, logits_bf16);
        for (int e = 0; e < n_dyn; ++e) w[e] = round_t(w[e] / den, logits_bf16);
        /* core.py:286-291 */
        if (attn_mask)
            for (int e = 0; e < E; ++e) m[e] *= (int32_t)(attn_mask[s] != 0);
        for (int e = n_dyn; e < E; ++e) m[e] = 1;
        /* core.py:178-193 */
        float gw[UMOE_MAX_E];
        if (n_fix > 0) {
            float ml[UMOE_MAX_E];
            for (int e = 0; e < E; ++e) ml[e] = m[e] ? full[e] : -INFINITY;
            softmax_t(ml, E, logits_bf16, gw);
            float ds = 0.0f;
            for (int e = 0; e < n_dyn; ++e) ds = ds + gw[e];
            ds = round_t(ds, logits_bf16);
            for (int e = 0; e < n_dyn; ++e) gw[e] = round_t(w[e] * ds, logits_bf16);
        } else {
            for (int e = 0; e < n_dyn; ++e) gw[e] = w[e];
        }
        for (int e = 0; e < E; ++e) {
            mask[(size_t)s * E + e] = m[e];
            global_w[(size_t)s * E + e] = gw[e];
        }
        for (int e = 0; e < n_dyn; ++e) route_w[(size_t)s * n_dyn + e] = w[e];
        for (int e = 0; e < n_real; ++e) moe_w[(size_t)s * n_real + e] = gw[e] * (float)m[e];
    }
    return 0;
}

/* exposed for unit tests of the arithmetic contract */
float umoe_oracle_exp_det(float x) { return exp_det(x); }
int umoe_oracle_top_p_count(const float* logit, int n, float top_p, int is_bf16) {
    return top_p_count(logit, n, top_p, is_bf16);
}

/*
 * Dispatch lists (restates what compress_matrix/decompress_matrix define,
 * utils/UniMoE_Audio_utils.py:436-523: per expert column, the rows whose mask is 1).
 * Row order inside an expert is ascending token index (the reference's argsort order
 * among equal keys is unspecified and does not affect results).
 *  mask   [S][ld_mask] int32, first n_real columns used
 *  counts [n_real], offsets [n_real+1], slot_token [sum counts], slot_of [S][n_real] (-1 if unrouted)
 */
int umoe_oracle_dispatch(const int32_t* mask, int S, int ld_mask, int n_real, int32_t* counts, int32_t* offsets,
                         int32_t* slot_token, int32_t* slot_of) {
    int off = 0;
    for (int e = 0; e < n_real; ++e) {
        offsets[e] = off;
        int c = 0;
        for (int s = 0; s < S; ++s) {
            if (mask[(size_t)s * ld_mask + e]) {
                slot_token[off + c] = s;
                slot_of[(size_t)s * n_real + e] = off + c;
                ++c;
            } else {
                slot_of[(size_t)s * n_real + e] = -1;
            }
        }
        counts[e] = c;
        off += c;
    }
    offsets[n_real] = off;
    return off;
}
