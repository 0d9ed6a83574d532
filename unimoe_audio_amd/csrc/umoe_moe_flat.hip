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

#include "umoe_flat_dev.h"

#define FLAT_RIDER_LDS 512      // bytes of LDS behind the GEMM area for the riders' partial sums (router4_body: 4 + 4 * 16 floats)

__global__ __launch_bounds__(512, 1) void moe_flat_kernel(const flat_args A, const umoe_router_args ra, const umoe_rider_pub pub, const int lds_gemm, const flat_o O) {
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
    const bool oph = O.half > 0;            // o_proj inside this launch (kernel argument: a scalar branch)
    if (token >= 0) {
        // (the router reads the raw rows: with o_proj inside the launch the rider takes its half tile and the wait first, no prefetch behind it)
        if (oph) {
            const flat_u32x4* const nowp[1] = {nullptr};
            flat_u32x4 now0[1];
            flat_oproj_half<1, false>(O, pub, b, smem, (int)threadIdx.x, nowp, now0, 0);
            flat_oproj_wait(O, pub, b, (int)threadIdx.x);
        }
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
    if (oph && token >= 0) __syncthreads();      // (the half tile's reduction slab is the staging area of the rows)
    const int ophm = oph ? (token >= 0 ? 2 : 1) : 0;
    switch (np) {
        case 4: flat_gateup<4>(A, pub, fp0, b, smem, st, (int)threadIdx.x, ophm, O); break;
        case 5: flat_gateup<5>(A, pub, fp0, b, smem, st, (int)threadIdx.x, ophm, O); break;
        case 6: flat_gateup<6>(A, pub, fp0, b, smem, st, (int)threadIdx.x, ophm, O); break;
        case 7: flat_gateup<7>(A, pub, fp0, b, smem, st, (int)threadIdx.x, ophm, O); break;
        default: break;
    }
    for (int sl = 0; sl < FLAT_SLICES; ++sl) {
        const unsigned e16 = (ed >> (16 * sl)) & 0xffffu;
        const int nd = (int)(e16 >> 12), grp = (int)(e16 & 15u), nb0 = (int)((e16 >> 4) & 255u);
        if (nd == 0) break;
        __syncthreads();     // (the reduction slab of the previous GEMM is the staging area of this one)
        if (A.dn_kb[grp] & 1) {
            switch (nd) {
                case 1: flat_down<1, 1>(A, pub, grp, nb0, smem, st, 7 + 4 * sl, (int)threadIdx.x); break;
                case 2: flat_down<2, 1>(A, pub, grp, nb0, smem, st, 7 + 4 * sl, (int)threadIdx.x); break;
                case 3: flat_down<3, 1>(A, pub, grp, nb0, smem, st, 7 + 4 * sl, (int)threadIdx.x); break;
                case 4: flat_down<4, 1>(A, pub, grp, nb0, smem, st, 7 + 4 * sl, (int)threadIdx.x); break;
                case 5: flat_down<5, 1>(A, pub, grp, nb0, smem, st, 7 + 4 * sl, (int)threadIdx.x); break;
                case 6: flat_down<6, 1>(A, pub, grp, nb0, smem, st, 7 + 4 * sl, (int)threadIdx.x); break;
                case 7: flat_down<7, 1>(A, pub, grp, nb0, smem, st, 7 + 4 * sl, (int)threadIdx.x); break;
                case 8: flat_down<8, 1>(A, pub, grp, nb0, smem, st, 7 + 4 * sl, (int)threadIdx.x); break;
                case 9: flat_down<9, 1>(A, pub, grp, nb0, smem, st, 7 + 4 * sl, (int)threadIdx.x); break;
                default: flat_down<10, 1>(A, pub, grp, nb0, smem, st, 7 + 4 * sl, (int)threadIdx.x); break;
            }
        } else {
            switch (nd) {
                case 1: flat_down<1, 2>(A, pub, grp, nb0, smem, st, 7 + 4 * sl, (int)threadIdx.x); break;
                case 2: flat_down<2, 2>(A, pub, grp, nb0, smem, st, 7 + 4 * sl, (int)threadIdx.x); break;
                case 3: flat_down<3, 2>(A, pub, grp, nb0, smem, st, 7 + 4 * sl, (int)threadIdx.x); break;
                case 4: flat_down<4, 2>(A, pub, grp, nb0, smem, st, 7 + 4 * sl, (int)threadIdx.x); break;
                case 5: flat_down<5, 2>(A, pub, grp, nb0, smem, st, 7 + 4 * sl, (int)threadIdx.x); break;
                default: flat_down<6, 2>(A, pub, grp, nb0, smem, st, 7 + 4 * sl, (int)threadIdx.x); break;
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
int umoe_moe_flat(const umoe_gemm_args* gu, const umoe_gemm_args* dn, uint32_t* flags, int flag_words, int n_wg, hipStream_t s,
                  const umoe_gemm_args* oproj, uint32_t* o_flags) {
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
    flat_o O;
    memset(&O, 0, sizeof(O));
    if (oproj) {
        // o_proj + residual inside the launch: its output rows are this launch's raw rows
        if (!(o_flags && oproj->groups_host && oproj->num_groups == 1 && oproj->epilogue == UMOE_EPI_BF16_RESID && oproj->prologue == UMOE_PRO_PLAIN &&
              oproj->ksplit <= 1 && oproj->resid && oproj->a && oproj->out == (void*)r->x && oproj->ldo == r->D && oproj->max_rows == r->S && r->D == 2048 &&
              (oproj->lda & 7) == 0))
            return 1;
        const umoe_group_t& og = oproj->groups_host[0];
        if (og.rows || og.count || og.row_off || og.a_row_base || og.a_col_off || og.out_row_base || og.bias || og.static_count != r->S || og.k != 2048 ||
            og.n_blocks * 16 != r->D || 2 * og.n_blocks > 256)
            return 1;
        O.rows = oproj->a; O.lda_rows = oproj->lda; O.w = og.w; O.resid = oproj->resid; O.flags = o_flags;
        O.x1 = reinterpret_cast<uint16_t*>(oproj->out); O.lda = r->D; O.S = r->S; O.half = 2 * og.n_blocks; O.n_wg = n_wg;
    }
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
    moe_flat_kernel<<<dim3((unsigned)n_wg), 512, lds + FLAT_RIDER_LDS, s>>>(A, rr, pub, (int)lds, O);
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
