// Decode engine: owns the KV cache + workspace and enqueues prefill / one whole decode step
// (36 x [RMSNorm+QKV GEMM, mRoPE+append, attention, o_proj+residual, RMSNorm+router, dispatch,
// grouped gate/up SwiGLU, grouped down, combine+residual] + head GEMM + CFG/sampler + EOS/delay
// bookkeeping) from ONE host call, with no host<->device synchronisation inside the step.
// A captured hipGraph replays the step; every step-dependent scalar lives in device memory.
//
// Restates the control flow of the reference's generate()/_decoder_step()
// (utils/UniMoE_Audio_model.py:918-1231) and Qwen2_5_VLMoEDecoderLayer.forward (:210-256).
#include <string.h>

#include <string>
#include <vector>

#include "umoe_common.h"
#include "umoe_riders_dev.h"

namespace {

struct LayerDev {
    umoe_layer_weights w;
    std::vector<const uint16_t*> exp_gu, exp_dn, sh_gu, sh_dn;
    std::vector<const uint16_t*> rm_eg, rm_eu, rm_ed, rm_sg, rm_su, rm_sd;   // row-major copies (tiled prefill path)
    bool has_rm = false;
    bool set = false;
};

struct Carver {
    char* base = nullptr;
    size_t off = 0;
    template <typename T>
    T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

}  // namespace

struct umoe_engine {
    umoe_engine_cfg c;
    std::vector<LayerDev> layers;
    const uint16_t *final_norm = nullptr, *codec_emb = nullptr, *codec_head_w = nullptr, *cos_tab = nullptr,
                   *sin_tab = nullptr;
    int max_pos = 0;
    int max_delay = 0;
    // device memory owned by the engine
    char* ws = nullptr;
    size_t ws_bytes = 0;
    int cap_tok = 0;  // tokens the workspace is sized for
    uint16_t *k_cache = nullptr, *v_cache = nullptr;
    int32_t* d_delay = nullptr;
    umoe_group_t* d_groups = nullptr;  // per layer: [qkv 1][o 1][gateup G][down G]; then [head 1]
    std::vector<umoe_group_t> h_groups;  // host copy: descriptors travel by value in the GEMM kernel arguments
    std::vector<umoe_group_t> h_gu_pub;  // per layer: the gate/up groups with the SHARED experts first -- their tile-less workgroups
                                         // (the riders) then come early in dispatch order, in front of almost every workgroup that waits for them
    int groups_for_tok = -1;
    // carved buffers
    uint16_t *x = nullptr, *hin = nullptr, *x1 = nullptr, *h2 = nullptr, *qkv = nullptr, *q_r = nullptr, *attn_out = nullptr,
             *hbuf = nullptr, *ybuf = nullptr;
    float *part_o = nullptr, *part_ml = nullptr, *logits = nullptr, *ypart = nullptr;
    int32_t *pos3 = nullptr, *kv_pos = nullptr, *q_pos0 = nullptr, *kv_start = nullptr, *tok_in = nullptr,
            *valid_count = nullptr, *eng_state = nullptr;
    void* r_logits = nullptr;
    int64_t *r_topk = nullptr, *pred = nullptr;
    int32_t *r_sel = nullptr, *r_mask = nullptr, *counts = nullptr, *offsets = nullptr, *slot_token = nullptr,
            *slot_of = nullptr;
    float *r_routing = nullptr, *r_global = nullptr, *r_moe = nullptr;
    // per-layer router statistics kept for parity tests (last step): [layers][rows][E]
    int32_t* all_mask = nullptr;
    int64_t* all_topk = nullptr;
    int T_prompt = 0;
    // per-layer probe of the parity tests (umoe_engine_set_probe): eager steps only
    const uint16_t* probe_teach = nullptr;
    uint16_t *probe_x1 = nullptr, *probe_x = nullptr, *probe_logits = nullptr;
    bool probe_on() const { return probe_teach || probe_x1 || probe_x || probe_logits; }
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    // decode with >= 6 rows: every routed expert is hit with probability ~1, so each expert computes ALL rows (no gather
    // lists, no dispatch kernel, no device-produced row counts in the GEMM prologues) and the combine selects by mask
    bool dense_experts = true;
    bool fuse_router = true;     // UMOE_FUSE_ROUTER: dense decode runs the router inside the gate/up launch (see run_layer)
    bool fuse_cq = true;         // UMOE_FUSE_CQ: the MoE combine of layer l rides in the QKV launch of layer l + 1 (umoe_gemm_riders kind 2)
    int cb_ep_layer = -1;        // expert parallel: the layer whose return slab the stashed combine reads (>= 0 selects rider kind 4)
    bool cb_pending = false;     // a combine stashed at the end of a layer, issued with the next layer's QKV launch
    umoe_combine_args cb_stash{};
    int dense_min_rows = 2;      // UMOE_DENSE_MIN_ROWS: fewest decode rows that take the dense-expert layout (below: ragged dispatch).  Batch 1
                                 // (2 CFG rows, BASELINE configs[0]) hits 5-6 of the 8 experts: streaming all 8 in the fused launches costs
                                 // fewer microseconds than the ragged path's four extra launches -- 2.91 vs 3.33 ms/step
    int expert_launch = 0;       // what the last dense decode layer enqueued for its experts: 0 launch per GEMM, 1 box-grid fused, 2 flat, 3 the
                                 // one-launch expert-parallel MoE half (umoe_engine_info)
    int n_cu = 0;                // compute units of the device (UMOE_FAKE_CUS overrides: tests of the co-residency guards)
    bool fuse_o = false;         // UMOE_FUSE_O=1: with the flat expert launch, o_proj + residual is computed INSIDE it (half a 16-feature tile per
                                 // workgroup, handed over by flags; the first expert weight stage is requested inside the half tile): four
                                 // launches per layer, bit-identical -- and 0.05 ms/step SLOWER in all three forms measured (3.170 / 3.064 /
                                 // 3.069 vs 3.02 / 3.01 / 3.01 ms): the hand-off (publish, 2048-flag wait, row staging) costs the
                                 // launch more than the o_proj launch it removes.  Off by default; kept selectable and parity-tested
    bool flat_moe = true;        // UMOE_FLAT_MOE: both expert GEMMs as ONE workgroup per CU with a byte-balanced static schedule
                                 // (umoe_moe_flat.hip); 0 / shapes that do not fit: the box-grid launch below
    bool fuse_moe = true;        // UMOE_FUSE_MOE: gate/up and down projections of a dense decode layer in ONE launch (umoe_moe_fused)
    bool rider_pub = true;       // UMOE_RIDER_PUB: the riders also produce the normalised rows and hand them to the GEMM workgroups of the
                                 // same launch (umoe_gemm_args.rider_pub): no RMSNorm launch in front of gate/up
    bool tiled_prefill = true;   // UMOE_TILED_PREFILL=0: weight-streaming kernels for every row count (A/B, tests)
    // optional per-kernel-class timing of one eager step (hipEvents on the launch stream)
    bool prof = false;
    std::vector<hipEvent_t> ev;
    std::vector<int> ev_kind;
    size_t ev_used = 0;
    // expert parallel decode (c.ep_size > 1): see include/umoe.h "Peer exchange"
    int E_loc = 0;                        // local routed experts = n_real / ep_size
    char* ep_region = nullptr;            // flags + dispatch slab + return slab (uncached device memory, IPC-exported)
    size_t ep_region_bytes = 0;
    int ep_mem_kind = 0;                  // 1 uncached, 2 fine-grained, 3 plain hipMalloc (what the runtime granted)
    char* ep_peers[UMOE_MAX_EP] = {};
    void* ep_comm = nullptr;
    int ep_mode = -1;                     // -1 = not connected
    uint32_t* ep_words = nullptr;         // [0] decode steps taken (epoch base), [1] sticky error word; own allocation: survives workspace growth
    uint16_t* xg = nullptr;               // RCCL mode: [ep][rows][D] normalised rows of every rank (tile ep_rank is written locally)
    uint16_t *xgp = nullptr, *hpk = nullptr;   // peer modes: operand-order tiles (see carve)
    // the MoE half of an expert-parallel decode layer as ONE launch with the exchange inside (umoe_moe_ep.hip); prepared at connect time
    bool ep_flat = true;                  // UMOE_EP_FLAT=0: the launch-per-kernel exchange below (also taken when the shape has no plan)
    bool epf_ready = false;
    umoe_epf_desc epf{};
    uint32_t* epf_tasks = nullptr;
    bool ep_decode(int n_tok) const { return c.ep_size > 1 && n_tok == c.rows; }
    int groups_per_layer() const { return 2 + 2 * (c.n_real + c.n_fix); }
};

static size_t carve(umoe_engine* e, int n_tok, char* base) {
    const umoe_engine_cfg& c = e->c;
    const int D = c.hidden, QKV = (c.heads + 2 * c.kv_heads) * c.head_dim, HD = c.heads * c.head_dim;
    const int E = c.n_dyn + c.n_fix, G = c.n_real + c.n_fix;
    const int Imax = c.inter_dyn > c.inter_shared ? c.inter_dyn : c.inter_shared;
    const size_t slots = (size_t)n_tok * G;
    const int splits = c.attn_splits > 1 ? c.attn_splits : 1;
    Carver k;
    k.base = base;
    e->x = k.take<uint16_t>((size_t)n_tok * D);
    e->hin = k.take<uint16_t>((size_t)n_tok * D);
    e->x1 = k.take<uint16_t>((size_t)n_tok * D);
    e->h2 = k.take<uint16_t>((size_t)n_tok * D);
    e->qkv = k.take<uint16_t>((size_t)n_tok * QKV);
    e->q_r = k.take<uint16_t>((size_t)n_tok * HD);
    e->attn_out = k.take<uint16_t>((size_t)n_tok * HD);
    e->hbuf = k.take<uint16_t>(slots * Imax);
    // expert parallel: behind the dense-layout rows [expert][row] and the shared experts' rows, ybuf also holds the outputs of
    // the LOCAL experts for every rank's rows, [dest rank][local expert][row], which the return exchange ships
    e->ybuf = k.take<uint16_t>(slots * D + (c.ep_size > 1 ? (size_t)c.n_real * c.rows * D : 0));
    e->xg = k.take<uint16_t>(c.ep_size > 1 ? (size_t)c.ep_size * c.rows * D : 0);
    // peer modes: 16-row tiles in MFMA operand order -- the gathered rows [ep][16*D] and silu(g)*u of every (local expert, rank) [n_real][16*I]
    e->xgp = k.take<uint16_t>(c.ep_size > 1 ? (size_t)c.ep_size * 16 * D : 0);
    e->hpk = k.take<uint16_t>(c.ep_size > 1 ? (size_t)c.n_real * 16 * c.inter_dyn : 0);
    e->part_o = k.take<float>((size_t)n_tok * c.heads * splits * c.head_dim);
    e->part_ml = k.take<float>((size_t)n_tok * c.heads * splits * 2);
    e->logits = k.take<float>((size_t)c.rows * c.codec_channels * c.codec_vocab);
    e->pos3 = k.take<int32_t>((size_t)3 * n_tok);
    e->kv_pos = k.take<int32_t>(n_tok);
    e->q_pos0 = k.take<int32_t>(c.rows);
    e->kv_start = k.take<int32_t>(c.rows);
    e->valid_count = k.take<int32_t>(c.rows);
    e->eng_state = k.take<int32_t>(8);
    e->tok_in = k.take<int32_t>((size_t)c.rows * c.codec_channels);
    e->r_logits = k.take<float>((size_t)n_tok * E);
    e->r_topk = k.take<int64_t>(n_tok);
    e->pred = k.take<int64_t>((size_t)c.rows * c.codec_channels);
    e->r_sel = k.take<int32_t>((size_t)n_tok * c.n_dyn);
    e->r_mask = k.take<int32_t>((size_t)n_tok * E);
    e->counts = k.take<int32_t>(UMOE_MAXE);
    e->offsets = k.take<int32_t>(UMOE_MAXE + 1);
    e->slot_token = k.take<int32_t>((size_t)n_tok * c.n_real + 1);
    e->slot_of = k.take<int32_t>((size_t)n_tok * c.n_real);
    e->r_routing = k.take<float>((size_t)n_tok * c.n_dyn);
    e->r_global = k.take<float>((size_t)n_tok * E);
    e->r_moe = k.take<float>((size_t)n_tok * c.n_real);
    e->all_mask = k.take<int32_t>((size_t)c.layers * c.rows * E);
    e->all_topk = k.take<int64_t>((size_t)c.layers * c.rows);
    return (k.off + 255) & ~(size_t)255;
}

static int ensure_workspace(umoe_engine* e, int n_tok) {
    if (n_tok <= e->cap_tok) return 0;
    // keep the small persistent scalars (valid_count, eng_state, kv_start) across a re-size: they are
    // rewritten by prefill, which is the only caller that grows the workspace.
    if (e->ws) UMOE_HIP(hipFree(e->ws));
    e->ws = nullptr;
    const size_t bytes = carve(e, n_tok, nullptr);
    UMOE_HIP(hipMalloc(&e->ws, bytes));
    UMOE_HIP(hipMemset(e->ws, 0, bytes));
    carve(e, n_tok, e->ws);
    e->ws_bytes = bytes;
    e->cap_tok = n_tok;
    e->groups_for_tok = -1;
    return 0;
}

// group table for a pass over n_tok tokens
static bool dense_mode(const umoe_engine* e, int n_tok) {
    if (e->ep_decode(n_tok)) return true;   // expert parallel decode IS the dense layout: every rank's rows visit every expert
    return e->dense_experts && n_tok == e->c.rows && n_tok <= 16 && n_tok >= e->dense_min_rows;
}

static int build_groups(umoe_engine* e, int n_tok, hipStream_t s) {
    if (e->groups_for_tok == n_tok) return 0;
    const umoe_engine_cfg& c = e->c;
    const bool dense = dense_mode(e, n_tok);
    const int G = c.n_real + c.n_fix, GPL = e->groups_per_layer();
    std::vector<umoe_group_t>& h = e->h_groups;
    h.assign((size_t)c.layers * GPL + 1, umoe_group_t{});
    const int slots_routed = n_tok * c.n_real;
    for (int l = 0; l < c.layers; ++l) {
        const LayerDev& L = e->layers[l];
        UMOE_REQUIRE(L.set, "umoe_engine: layer %d has no weights", l);
        umoe_group_t* g = &h[(size_t)l * GPL];
        memset(g, 0, sizeof(umoe_group_t) * GPL);
        // qkv
        g[0].w = L.w.qkv_w; g[0].bias = L.w.qkv_b; g[0].static_count = n_tok;
        g[0].n_blocks = (c.heads + 2 * c.kv_heads) * c.head_dim / 16; g[0].k = c.hidden;
        // o_proj
        g[1].w = L.w.o_w; g[1].static_count = n_tok; g[1].n_blocks = c.hidden / 16; g[1].k = c.heads * c.head_dim;
        umoe_group_t* gu = g + 2;
        umoe_group_t* dn = g + 2 + G;
        if (e->ep_decode(n_tok)) {
            // expert parallel decode: one group per (local expert x, source rank t) over the gathered rows xg[t][row]; h rows
            // (x*ep + t)*rows; outputs for rank t's rows at ybuf rows yloc0 + (t*E_loc + x)*rows (shipped by the return push),
            // the own rank's straight into the dense layout row (global expert)*rows the combine reads (peer modes)
            const int ep = c.ep_size, El = e->E_loc, yloc0 = (c.n_real + c.n_fix) * n_tok;
            const bool own_direct = e->ep_mode != UMOE_EP_RCCL;
            for (int x = 0; x < El; ++x)
                for (int t = 0; t < ep; ++t) {
                    const int q = x * ep + t;
                    gu[q].w = L.exp_gu[x]; gu[q].n_blocks = 2 * c.inter_dyn / 16; gu[q].k = c.hidden;
                    gu[q].static_count = n_tok; gu[q].a_row_base = t * n_tok; gu[q].out_row_base = q * n_tok;
                    dn[q].w = L.exp_dn[x]; dn[q].n_blocks = c.hidden / 16; dn[q].k = c.inter_dyn;
                    dn[q].static_count = n_tok; dn[q].a_row_base = q * n_tok;
                    dn[q].out_row_base = (t == c.ep_rank && own_direct) ? (c.ep_rank * El + x) * n_tok : yloc0 + (t * El + x) * n_tok;
                }
            for (int i = 0; i < c.n_fix; ++i) {
                const int x = c.n_real + i;
                gu[x].w = L.sh_gu[i]; gu[x].static_count = n_tok;
                gu[x].a_row_base = own_direct ? 0 : c.ep_rank * n_tok;   // peer modes: the row-major h2; RCCL mode: own tile of xg
                gu[x].out_row_base = slots_routed + i * n_tok; gu[x].n_blocks = 2 * c.inter_shared / 16; gu[x].k = c.hidden;
                dn[x].w = L.sh_dn[i]; dn[x].static_count = n_tok; dn[x].a_row_base = slots_routed + i * n_tok;
                dn[x].out_row_base = slots_routed + i * n_tok; dn[x].n_blocks = c.hidden / 16; dn[x].k = c.inter_shared;
            }
            continue;
        }
        const int n_packed = c.ep_size > 1 ? 0 : c.n_real;   // expert parallel prefill: tiled kernels on the row-major tensors only
        for (int x = 0; x < c.n_real; ++x) {
            gu[x].w = x < n_packed ? L.exp_gu[x] : nullptr; gu[x].n_blocks = 2 * c.inter_dyn / 16; gu[x].k = c.hidden;
            dn[x].w = x < n_packed ? L.exp_dn[x] : nullptr; dn[x].n_blocks = c.hidden / 16; dn[x].k = c.inter_dyn;
            if (dense) {   // expert x owns rows [x*n_tok, (x+1)*n_tok) of the h / y buffers, token order
                gu[x].static_count = n_tok; gu[x].out_row_base = x * n_tok;
                dn[x].static_count = n_tok; dn[x].a_row_base = x * n_tok; dn[x].out_row_base = x * n_tok;
            } else {
                gu[x].rows = e->slot_token; gu[x].row_off = e->offsets + x; gu[x].count = e->counts + x;
                dn[x].row_off = e->offsets + x; dn[x].count = e->counts + x;
            }
        }
        for (int i = 0; i < c.n_fix; ++i) {
            const int x = c.n_real + i;
            gu[x].w = L.sh_gu[i]; gu[x].static_count = n_tok; gu[x].out_row_base = slots_routed + i * n_tok;
            gu[x].n_blocks = 2 * c.inter_shared / 16; gu[x].k = c.hidden;
            dn[x].w = L.sh_dn[i]; dn[x].static_count = n_tok; dn[x].a_row_base = slots_routed + i * n_tok;
            dn[x].out_row_base = slots_routed + i * n_tok; dn[x].n_blocks = c.hidden / 16; dn[x].k = c.inter_shared;
        }
    }
    e->h_gu_pub.assign((size_t)c.layers * G, umoe_group_t{});
    for (int l = 0; l < c.layers; ++l) {
        const umoe_group_t* gu = &h[(size_t)l * GPL + 2];
        umoe_group_t* o = &e->h_gu_pub[(size_t)l * G];
        for (int i = 0; i < c.n_fix; ++i) o[i] = gu[c.n_real + i];
        for (int x = 0; x < c.n_real; ++x) o[c.n_fix + x] = gu[x];
    }
    umoe_group_t* hg = &h[(size_t)c.layers * GPL];
    memset(hg, 0, sizeof(umoe_group_t));
    hg->w = e->codec_head_w; hg->static_count = c.rows;
    hg->n_blocks = ceil_div(c.codec_channels * c.codec_vocab, 16); hg->k = c.hidden;
    UMOE_HIP(hipMemcpyAsync(e->d_groups, h.data(), h.size() * sizeof(umoe_group_t), hipMemcpyHostToDevice, s));
    UMOE_HIP(hipStreamSynchronize(s));  // h is a stack-lifetime vector
    e->groups_for_tok = n_tok;
    return 0;
}

extern "C" int umoe_engine_create(const umoe_engine_cfg* cfg, umoe_engine** out) {
    UMOE_REQUIRE(cfg && out, "umoe_engine_create: null argument");
    UMOE_REQUIRE(cfg->head_dim == 128, "umoe_engine: head_dim must be 128 (got %d)", cfg->head_dim);
    UMOE_REQUIRE(cfg->hidden % 128 == 0 && cfg->inter_dyn % 32 == 0 && cfg->inter_shared % 32 == 0,
                 "umoe_engine: hidden %% 128 and intermediate sizes %% 32 must be 0");
    UMOE_REQUIRE(cfg->rows > 0 && cfg->rows % 2 == 0 && cfg->rows / 2 <= 256, "umoe_engine: rows must be 2*batch, batch <= 256");
    UMOE_REQUIRE(cfg->n_dyn + cfg->n_fix <= UMOE_MAXE && cfg->n_real <= cfg->n_dyn, "umoe_engine: bad expert counts");
    const int ep = cfg->ep_size > 1 ? cfg->ep_size : 1;
    UMOE_REQUIRE(ep == 1 || ((ep == 2 || ep == 4 || ep == 8) && cfg->n_real % ep == 0 && cfg->ep_rank >= 0 && cfg->ep_rank < ep &&
                             cfg->rows <= 16 && cfg->n_real + cfg->n_fix <= UMOE_GROUPS_INLINE),
                 "umoe_engine: expert parallel decode needs ep_size 2/4/8 dividing n_real=%d, 0 <= ep_rank < ep_size, rows <= 16 (ep_size=%d rank=%d rows=%d)",
                 cfg->n_real, cfg->ep_size, cfg->ep_rank, cfg->rows);
    umoe_engine* e = new umoe_engine();
    e->c = *cfg;
    e->c.ep_size = ep;
    if (ep == 1) e->c.ep_rank = 0;
    e->E_loc = cfg->n_real / ep;
    if (e->c.attn_splits < 1) e->c.attn_splits = 1;
    e->layers.resize(cfg->layers);
    const size_t kv = (size_t)cfg->layers * cfg->rows * cfg->kv_heads * cfg->Lmax * cfg->head_dim;
    if (hipMalloc(&e->k_cache, kv * 2) != hipSuccess || hipMalloc(&e->v_cache, kv * 2) != hipSuccess ||
        hipMalloc(&e->d_delay, sizeof(int32_t) * cfg->codec_channels) != hipSuccess ||
        hipMalloc(&e->d_groups, sizeof(umoe_group_t) * ((size_t)cfg->layers * e->groups_per_layer() + 1)) != hipSuccess) {
        umoe_set_error("umoe_engine_create: hipMalloc failed (KV cache %zu bytes)", kv * 4);
        umoe_engine_destroy(e);
        return -2;
    }
    UMOE_HIP(hipMemset(e->k_cache, 0, kv * 2));
    UMOE_HIP(hipMemset(e->v_cache, 0, kv * 2));
    if (ensure_workspace(e, cfg->rows)) {
        umoe_engine_destroy(e);
        return -2;
    }
    // [0] decode steps taken, [1] sticky hand-off error, [16..32) row flags of the rider hand-off (umoe_gemm_args.rider_pub)
    if (hipMalloc(&e->ep_words, 32768) != hipSuccess || hipMemset(e->ep_words, 0, 32768) != hipSuccess) {
        umoe_set_error("umoe_engine_create: hipMalloc failed (state words)");
        umoe_engine_destroy(e);
        return -2;
    }
    if (ep > 1) {
        // exchange region: peers store into it over xGMI and this GPU reads it while a kernel runs -> uncached (MTYPE UC) device
        // memory where the runtime grants it, fine-grained otherwise; every access of it is an sc0 sc1 access anyway (umoe_ep.hip)
        const size_t tile = (size_t)cfg->rows * cfg->hidden * 2;
        e->ep_region_bytes = UMOE_EP_FLAG_BYTES + (size_t)ep * tile + (size_t)cfg->n_real * tile;
        void* r = nullptr;
        if (hipExtMallocWithFlags(&r, e->ep_region_bytes, hipDeviceMallocUncached) == hipSuccess) e->ep_mem_kind = 1;
        else if ((void)hipGetLastError(), hipExtMallocWithFlags(&r, e->ep_region_bytes, hipDeviceMallocFinegrained) == hipSuccess) e->ep_mem_kind = 2;
        else if ((void)hipGetLastError(), hipMalloc(&r, e->ep_region_bytes) == hipSuccess) e->ep_mem_kind = 3;
        if (!r || hipMemset(r, 0, e->ep_region_bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            umoe_set_error("umoe_engine_create: exchange region allocation failed (%zu bytes)", e->ep_region_bytes);
            umoe_engine_destroy(e);
            return -2;
        }
        e->ep_region = (char*)r;
    }
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) e->n_cu = cus;
        if (const char* v = getenv("UMOE_FAKE_CUS")) e->n_cu = atoi(v);
    }
    if (const char* v = getenv("UMOE_FLAT_MOE")) e->flat_moe = atoi(v) != 0;
    if (const char* v = getenv("UMOE_EP_FLAT")) e->ep_flat = atoi(v) != 0;
    if (const char* v = getenv("UMOE_FUSE_O")) e->fuse_o = atoi(v) != 0;
    if (const char* v = getenv("UMOE_DENSE_EXPERTS")) e->dense_experts = atoi(v) != 0;
    if (const char* v = getenv("UMOE_TILED_PREFILL")) e->tiled_prefill = atoi(v) != 0;
    if (const char* v = getenv("UMOE_FUSE_ROUTER")) e->fuse_router = atoi(v) != 0;
    if (const char* v = getenv("UMOE_RIDER_PUB")) e->rider_pub = atoi(v) != 0;
    if (const char* v = getenv("UMOE_FUSE_MOE")) e->fuse_moe = atoi(v) != 0;
    if (const char* v = getenv("UMOE_DENSE_MIN_ROWS")) e->dense_min_rows = atoi(v);
    if (const char* v = getenv("UMOE_FUSE_CQ")) e->fuse_cq = atoi(v) != 0;
    *out = e;
    return 0;
}

extern "C" void umoe_engine_destroy(umoe_engine* e) {
    if (!e) return;
    if (e->exec) (void)hipGraphExecDestroy(e->exec);
    if (e->graph) (void)hipGraphDestroy(e->graph);
    for (hipEvent_t x : e->ev) (void)hipEventDestroy(x);
    if (e->ws) (void)hipFree(e->ws);
    if (e->ep_region) (void)hipFree(e->ep_region);
    if (e->ep_words) (void)hipFree(e->ep_words);
    if (e->epf_tasks) (void)hipFree(e->epf_tasks);
    if (e->k_cache) (void)hipFree(e->k_cache);
    if (e->v_cache) (void)hipFree(e->v_cache);
    if (e->d_delay) (void)hipFree(e->d_delay);
    if (e->d_groups) (void)hipFree(e->d_groups);
    delete e;
}

extern "C" int umoe_engine_set_layer(umoe_engine* e, int layer, const umoe_layer_weights* w) {
    UMOE_REQUIRE(e && w && layer >= 0 && layer < e->c.layers, "umoe_engine_set_layer: bad layer %d", layer);
    LayerDev& L = e->layers[layer];
    L.w = *w;
    L.exp_gu.assign(w->exp_gu, w->exp_gu + e->E_loc);     // expert parallel: the LOCAL experts' packed weights only
    L.exp_dn.assign(w->exp_dn, w->exp_dn + e->E_loc);
    L.sh_gu.assign(w->sh_gu, w->sh_gu + e->c.n_fix);
    L.sh_dn.assign(w->sh_dn, w->sh_dn + e->c.n_fix);
    L.has_rm = w->rm_qkv && w->rm_o && w->rm_exp_gate && w->rm_exp_up && w->rm_exp_down &&
               (e->c.n_fix == 0 || (w->rm_sh_gate && w->rm_sh_up && w->rm_sh_down));
    if (L.has_rm) {
        L.rm_eg.assign(w->rm_exp_gate, w->rm_exp_gate + e->c.n_real);
        L.rm_eu.assign(w->rm_exp_up, w->rm_exp_up + e->c.n_real);
        L.rm_ed.assign(w->rm_exp_down, w->rm_exp_down + e->c.n_real);
        if (e->c.n_fix) {
            L.rm_sg.assign(w->rm_sh_gate, w->rm_sh_gate + e->c.n_fix);
            L.rm_su.assign(w->rm_sh_up, w->rm_sh_up + e->c.n_fix);
            L.rm_sd.assign(w->rm_sh_down, w->rm_sh_down + e->c.n_fix);
        }
    }
    L.set = true;
    e->groups_for_tok = -1;
    return 0;
}

extern "C" int umoe_engine_set_globals(umoe_engine* e, const uint16_t* final_norm, const uint16_t* codec_emb,
                                       const uint16_t* codec_head_w, const uint16_t* cos_tab, const uint16_t* sin_tab,
                                       int max_pos, const int32_t* delay_pattern_host) {
    UMOE_REQUIRE(e && final_norm && codec_emb && codec_head_w && cos_tab && sin_tab && delay_pattern_host,
                 "umoe_engine_set_globals: null argument");
    e->final_norm = final_norm; e->codec_emb = codec_emb; e->codec_head_w = codec_head_w;
    e->cos_tab = cos_tab; e->sin_tab = sin_tab; e->max_pos = max_pos;
    UMOE_HIP(hipMemcpy(e->d_delay, delay_pattern_host, sizeof(int32_t) * e->c.codec_channels, hipMemcpyHostToDevice));
    e->max_delay = 0;
    for (int i = 0; i < e->c.codec_channels; ++i) e->max_delay = delay_pattern_host[i] > e->max_delay ? delay_pattern_host[i] : e->max_delay;
    e->groups_for_tok = -1;
    return 0;
}

extern "C" size_t umoe_engine_workspace_bytes(const umoe_engine* e) { return e ? e->ws_bytes : 0; }

// ------------------------------------------------------------------------------------ expert parallel plumbing
extern "C" int umoe_engine_ep_region(umoe_engine* e, void** base_out, size_t* bytes_out) {
    UMOE_REQUIRE(e && base_out && bytes_out, "umoe_engine_ep_region: null argument");
    UMOE_REQUIRE(e->c.ep_size > 1 && e->ep_region, "umoe_engine_ep_region: engine was created with ep_size 1");
    *base_out = e->ep_region;
    *bytes_out = e->ep_region_bytes;
    return 0;
}

extern "C" int umoe_engine_ep_connect(umoe_engine* e, void* const* peers, void* rccl_comm, int mode) {
    UMOE_REQUIRE(e && e->c.ep_size > 1, "umoe_engine_ep_connect: engine was created with ep_size 1");
    UMOE_REQUIRE(mode == UMOE_EP_PEER || mode == UMOE_EP_LOOPBACK || mode == UMOE_EP_RCCL, "umoe_engine_ep_connect: bad mode %d", mode);
    if (mode == UMOE_EP_RCCL) {
        UMOE_REQUIRE(rccl_comm, "umoe_engine_ep_connect: the RCCL mode needs a communicator");
        e->ep_comm = rccl_comm;
        for (int p = 0; p < e->c.ep_size; ++p) e->ep_peers[p] = e->ep_region;
    } else if (mode == UMOE_EP_LOOPBACK) {
        for (int p = 0; p < e->c.ep_size; ++p) e->ep_peers[p] = e->ep_region;
    } else {
        UMOE_REQUIRE(peers, "umoe_engine_ep_connect: the peer mode needs every rank's region");
        for (int p = 0; p < e->c.ep_size; ++p) {
            UMOE_REQUIRE(peers[p], "umoe_engine_ep_connect: region of rank %d is null", p);
            e->ep_peers[p] = (char*)peers[p];
        }
        UMOE_REQUIRE(e->ep_peers[e->c.ep_rank] == e->ep_region, "umoe_engine_ep_connect: peers[ep_rank] must be this engine's own region");
    }
    e->ep_mode = mode;
    e->groups_for_tok = -1;   // the decode group table depends on the mode (where the own rows' outputs go)
    // the one-launch form of the MoE half (umoe_moe_ep.hip): its task lists are computed and uploaded here, outside any graph capture.
    // One workgroup per CU and in-launch hand-offs: every workgroup must be resident, so the launch is sized to the CU count the guards
    // assume (UMOE_FAKE_CUS: several ranks sharing one card take a share each).
    e->epf_ready = false;
    const umoe_engine_cfg& c = e->c;
    if (mode != UMOE_EP_RCCL && e->ep_flat && e->rider_pub && e->fuse_router && e->n_cu > 0 && c.hidden == 2048 && c.n_dyn == 9 && c.n_fix == 2 && c.rows <= 16 &&
        e->E_loc >= 1 && e->E_loc <= UMOE_MT_MAXG) {
        umoe_epf_desc& d = e->epf;
        memset(&d, 0, sizeof(d));
        d.n_wg = e->n_cu < 256 ? e->n_cu : 256;
        d.R = c.ep_size; d.E_loc = e->E_loc; d.S = c.rows; d.D = c.hidden; d.I_dyn = c.inter_dyn; d.I_sh = c.inter_shared; d.n_fix = c.n_fix;
        if (!e->epf_tasks) UMOE_HIP(hipMalloc(&e->epf_tasks, (size_t)256 * UMOE_EPF_MAXT * sizeof(uint32_t)));
        d.tasks_dev = e->epf_tasks;
        const int rc = umoe_moe_ep_prepare(&d, nullptr);
        if (rc < 0) return rc;
        e->epf_ready = rc == 0;
    }
    return 0;
}

extern "C" int umoe_engine_ep_error(umoe_engine* e, umoe_stream_t stream, int* code_out) {
    UMOE_REQUIRE(e && code_out, "umoe_engine_ep_error: null argument");
    uint32_t w[2] = {0, 0};
    UMOE_HIP(hipStreamSynchronize((hipStream_t)stream));
    UMOE_HIP(hipMemcpy(w, e->ep_words, sizeof(w), hipMemcpyDeviceToHost));
    *code_out = (int)w[1];
    return 0;
}

// kernel classes reported by umoe_engine_profile_step
enum { K_QKV = 0, K_ROPE, K_ATTN, K_OPROJ, K_ROUTER, K_DISPATCH, K_GATEUP, K_DOWN, K_COMBINE, K_EMBED, K_HEAD, K_SAMPLE,
       K_DELAY, K_NUM };

static void prof_mark(umoe_engine* e, int kind, hipStream_t s) {
    if (!e->prof) return;
    if (e->ev_used == e->ev.size()) {
        hipEvent_t x;
        if (hipEventCreate(&x) != hipSuccess) return;
        e->ev.push_back(x);
        e->ev_kind.push_back(0);
    }
    e->ev_kind[e->ev_used] = kind;
    (void)hipEventRecord(e->ev[e->ev_used++], s);
}
#define PROF(kind) prof_mark(e, kind, s)

// ------------------------------------------------------------------------------------ DCMoE of one layer, expert parallel decode
// Replaces AudioMOELayer.forward with ep_size > 1 (core.py:446-493: capacity MAX all-reduce :457, all-to-alls :467 / :480) for
// the decode shape.  Order of launches (one stream, all captured in the step graph):
//   RMSNorm (own rows -> tile ep_rank of xg) | PUSH rows to every peer | shared experts gate/up (hides the push) | PULL the peers'
//   rows | local experts gate/up over ep*rows rows (+ the router riders) | down | PUSH outputs to the rows' owners | shared
//   experts down (hides the push) | PULL | combine (selects by the local routing mask, ascending expert order: bit-identical to
//   ep_size 1 because every (expert, 16-row tile) product is computed by the same kernel instantiation with the same K split).
static int run_moe_ep_rccl(umoe_engine* e, int l, int n_tok, hipStream_t s) {
    const umoe_engine_cfg& c = e->c;
    const int D = c.hidden, G = c.n_real + c.n_fix, GPL = e->groups_per_layer(), E = c.n_dyn + c.n_fix;
    const int Imax = c.inter_dyn > c.inter_shared ? c.inter_dyn : c.inter_shared;
    const int ep = c.ep_size, El = e->E_loc, rank = c.ep_rank;
    const LayerDev& L = e->layers[l];
    const umoe_group_t* g = e->d_groups + (size_t)l * GPL;
    const umoe_group_t* gh = e->h_groups.data() + (size_t)l * GPL;
    UMOE_REQUIRE(e->ep_mode >= 0, "umoe_engine: expert parallel engine is not connected (umoe_engine_ep_connect)");
    int rc;
    uint16_t* h2own = e->xg + (size_t)rank * n_tok * D;
    umoe_router_args ra{};
    ra.x = e->x1; ra.gate_w = L.w.gate_w; ra.norm_w = L.w.post_norm; ra.h_out = h2own; ra.S = n_tok; ra.D = D;
    ra.n_dyn = c.n_dyn; ra.n_real = c.n_real; ra.n_fix = c.n_fix; ra.logits_bf16 = 1; ra.top_p = c.top_p;
    ra.fixed_top_k = c.fixed_top_k; ra.jitter_eps = c.jitter_eps; ra.rms_eps = c.rms_eps;
    ra.logits_out = e->r_logits; ra.sel = e->r_sel; ra.routing_w = e->r_routing; ra.global_w = e->r_global; ra.moe_w = e->r_moe;
    ra.expert_mask = e->all_mask + (size_t)l * c.rows * E;
    ra.top_k = e->all_topk + (size_t)l * c.rows;
    const bool ride = e->fuse_router && c.n_dyn == 9 && c.n_fix == 2 && (D == 2048 || D == 4096) && n_tok <= 16;
    if (ride) {              // RMSNorm only; the router body rides in the local experts' gate/up launch (its results feed the combine)
        umoe_router_args rn = ra;
        rn.norm_only = 1;
        rc = umoe_router_fwd(&rn, s);
        ra.h_out = nullptr;
    } else {
        rc = umoe_router_fwd(&ra, s);
    }
    if (rc) return rc;
    PROF(K_ROUTER);
    const size_t tile = (size_t)n_tok * D * 2, chunk_y = (size_t)El * tile;
    umoe_ep_xfer x{};
    for (int p = 0; p < ep; ++p) x.peer_base[p] = e->ep_peers[p];
    x.rank = rank; x.size = ep; x.loopback = e->ep_mode == UMOE_EP_LOOPBACK; x.step = e->ep_words; x.err = e->ep_words + 1;
    x.layer = l; x.layers = c.layers;
    // ---- first exchange: my normalised rows to every peer
    if (e->ep_mode == UMOE_EP_RCCL) {
        rc = umoe_ep_rccl_allgather(e->ep_comm, h2own, e->xg, tile, s);
    } else {
        x.kind = 0; x.src = (const char*)h2own; x.src_stride = 0; x.chunk = tile; x.data_off = UMOE_EP_FLAG_BYTES;
        rc = umoe_ep_push(x, s);
    }
    if (rc) return rc;
    PROF(K_DISPATCH);
    // ---- shared experts gate/up on the own rows while the rows travel.  K split as in the ep_size 1 launch (8 waves, 1-step chunks)
    if (c.n_fix > 0) {
        umoe_gemm_args sg{};
        sg.groups = g + 2 + c.n_real; sg.groups_host = gh + 2 + c.n_real; sg.num_groups = c.n_fix; sg.max_rows = n_tok;
        sg.max_n_blocks = 2 * c.inter_shared / 16; sg.max_k = D; sg.a = e->xg; sg.lda = D; sg.out = e->hbuf; sg.ldo = Imax; sg.n_valid = Imax;
        sg.prologue = UMOE_PRO_PLAIN; sg.epilogue = UMOE_EPI_SWIGLU; sg.nt = 2; sg.waves = 8;
        if ((rc = umoe_grouped_gemm(&sg, s))) return rc;
        PROF(K_GATEUP);
    }
    if (e->ep_mode != UMOE_EP_RCCL) {
        x.dst = (char*)e->xg;
        if ((rc = umoe_ep_pull(x, s))) return rc;
        PROF(K_DISPATCH);
    }
    // ---- local experts over every rank's rows
    umoe_gemm_args gu{};
    gu.groups = g + 2; gu.groups_host = gh + 2; gu.num_groups = c.n_real; gu.max_rows = n_tok; gu.max_n_blocks = 2 * c.inter_dyn / 16;
    gu.max_k = D; gu.a = e->xg; gu.lda = D; gu.out = e->hbuf; gu.ldo = Imax; gu.n_valid = Imax;
    gu.prologue = UMOE_PRO_PLAIN; gu.epilogue = UMOE_EPI_SWIGLU; gu.nt = 14;
    if (ride) gu.fused_router = &ra;
    if ((rc = umoe_grouped_gemm(&gu, s))) return rc;
    PROF(K_GATEUP);
    umoe_gemm_args dn{};
    dn.groups = g + 2 + G; dn.groups_host = gh + 2 + G; dn.num_groups = c.n_real; dn.max_rows = n_tok; dn.max_n_blocks = D / 16;
    dn.max_k = c.inter_dyn; dn.a = e->hbuf; dn.lda = Imax; dn.out = e->ybuf; dn.ldo = D; dn.n_valid = D;
    dn.prologue = UMOE_PRO_PLAIN; dn.epilogue = UMOE_EPI_BF16; dn.nt = 6; dn.waves = 8;
    if ((rc = umoe_grouped_gemm(&dn, s))) return rc;
    PROF(K_DOWN);
    // ---- second exchange: outputs of my experts to the rows' owners (dense layout row = (global expert)*rows + row there)
    uint16_t* yloc = e->ybuf + (size_t)G * n_tok * D;
    if (e->ep_mode == UMOE_EP_RCCL) {
        rc = umoe_ep_all_to_all(e->ep_comm, yloc, e->ybuf, chunk_y, ep, s);
    } else {
        x.kind = 1; x.src = (const char*)yloc; x.src_stride = (long)chunk_y; x.chunk = chunk_y; x.data_off = UMOE_EP_FLAG_BYTES + (size_t)ep * tile;
        rc = umoe_ep_push(x, s);
    }
    if (rc) return rc;
    PROF(K_DISPATCH);
    if (c.n_fix > 0) {       // shared experts down while the outputs travel (K split as in the ep_size 1 launch: 8 waves, 2-step chunks)
        umoe_gemm_args sd{};
        sd.groups = g + 2 + G + c.n_real; sd.groups_host = gh + 2 + G + c.n_real; sd.num_groups = c.n_fix; sd.max_rows = n_tok;
        sd.max_n_blocks = D / 16; sd.max_k = c.inter_shared; sd.a = e->hbuf; sd.lda = Imax; sd.out = e->ybuf; sd.ldo = D; sd.n_valid = D;
        sd.prologue = UMOE_PRO_PLAIN; sd.epilogue = UMOE_EPI_BF16; sd.nt = 2; sd.waves = 8;
        if ((rc = umoe_grouped_gemm(&sd, s))) return rc;
        PROF(K_DOWN);
    }
    if (e->ep_mode != UMOE_EP_RCCL) {
        x.dst = (char*)e->ybuf;
        if ((rc = umoe_ep_pull(x, s))) return rc;
        PROF(K_DISPATCH);
    }
    // ---- combine + residual -> next layer input (as in run_layer)
    umoe_combine_args cb{};
    cb.y_slots = e->ybuf; cb.shared_row0 = -1; cb.slot_of = nullptr; cb.moe_w = e->r_moe;
    cb.expert_mask = ra.expert_mask; cb.mask_ld = E; cb.dense_rows = n_tok;
    cb.y_shared = c.n_fix ? e->ybuf + (size_t)n_tok * c.n_real * D : nullptr; cb.global_w = e->r_global;
    cb.resid = e->x1; cb.out = e->x; cb.S = n_tok; cb.D = D; cb.n_real = c.n_real; cb.n_dyn = c.n_dyn; cb.n_fix = c.n_fix;
    cb.norm_w = (l + 1 < c.layers) ? e->layers[l + 1].w.in_norm : e->final_norm; cb.norm_out = e->hin; cb.rms_eps = c.rms_eps;
    rc = umoe_unpermute_combine_fwd(&cb, s);
    PROF(K_COMBINE);
    return rc;
}

// Peer modes (xGMI stores / loopback): 6 launches beside the two shared-expert GEMMs --
//   RMSNorm + PUSH of the rows (one launch) | shared gate/up | PULL + re-lay into operand order | local experts gate/up over
//   all ep*16 rows, weights streamed ONCE (umoe_gemm_mt.hip; + the router riders) | down, own rows' outputs straight into the own
//   return slab | PUSH outputs | shared down | combine, which waits for the peers' rows itself and reads the slab in place.
// One launch (umoe_moe_ep.hip): riders push / re-lay the rows and route, phases A..D stream the shared and the local experts' weights,
// the down epilogue stores into the owners' return slabs; the combine rides in the next layer's QKV launch (kind 4), the last layer's
// runs as its own launch on the same counters.
static int run_moe_ep_flat(umoe_engine* e, int l, int n_tok, hipStream_t s) {
    const umoe_engine_cfg& c = e->c;
    const int D = c.hidden, E = c.n_dyn + c.n_fix, QKV = (c.heads + 2 * c.kv_heads) * c.head_dim;
    const int Imax = c.inter_dyn > c.inter_shared ? c.inter_dyn : c.inter_shared;
    const int ep = c.ep_size, El = e->E_loc, rank = c.ep_rank;
    const LayerDev& L = e->layers[l];
    int rc;
    umoe_router_args ra{};
    ra.x = e->x1; ra.gate_w = L.w.gate_w; ra.norm_w = L.w.post_norm; ra.h_out = nullptr; ra.S = n_tok; ra.D = D;
    ra.n_dyn = c.n_dyn; ra.n_real = c.n_real; ra.n_fix = c.n_fix; ra.logits_bf16 = 1; ra.top_p = c.top_p;
    ra.fixed_top_k = c.fixed_top_k; ra.jitter_eps = c.jitter_eps; ra.rms_eps = c.rms_eps;
    ra.logits_out = e->r_logits; ra.sel = e->r_sel; ra.routing_w = e->r_routing; ra.global_w = e->r_global; ra.moe_w = e->r_moe;
    ra.expert_mask = e->all_mask + (size_t)l * c.rows * E;
    ra.top_k = e->all_topk + (size_t)l * c.rows;
    umoe_rider_pub pub{};
    pub.flags = e->ep_words + 1024; pub.step = e->ep_words; pub.layer = l; pub.layers = c.layers; pub.err = e->ep_words + 1;
    const size_t tile = (size_t)n_tok * D * 2;
    umoe_epf_desc d = e->epf;
    d.router = &ra; d.pub = &pub; d.rank = rank; d.loopback = e->ep_mode == UMOE_EP_LOOPBACK;
    d.peer_base = e->ep_peers; d.disp_off = UMOE_EP_FLAG_BYTES; d.ret_off = UMOE_EP_FLAG_BYTES + (size_t)ep * tile;
    d.w_lgu = L.exp_gu.data(); d.w_ldn = L.exp_dn.data(); d.w_sgu = L.sh_gu.data(); d.w_sdn = L.sh_dn.data();
    d.xgp = e->xgp; d.hpk = e->hpk;
    d.h_sh = e->hbuf; d.ldh = Imax; d.h_row0 = c.n_real * n_tok;
    d.y_sh = e->ybuf; d.ldy = D; d.y_row0 = c.n_real * n_tok;
    d.flags = e->ep_words + 2048; d.flag_words = 2688 - 2048;
    d.round = e->ep_words + 2;
    if ((rc = umoe_moe_ep(&d, s))) return rc;
    e->expert_launch = 3;
    PROF(K_GATEUP);
    uint16_t* slab_ret = reinterpret_cast<uint16_t*>(e->ep_region + d.ret_off);
    umoe_combine_args cb{};
    cb.y_slots = slab_ret; cb.shared_row0 = -1; cb.slot_of = nullptr; cb.moe_w = e->r_moe;
    cb.expert_mask = ra.expert_mask; cb.mask_ld = E; cb.dense_rows = n_tok;
    cb.y_shared = e->ybuf + (size_t)n_tok * c.n_real * D; cb.global_w = e->r_global;
    cb.resid = e->x1; cb.out = e->x; cb.S = n_tok; cb.D = D; cb.n_real = c.n_real; cb.n_dyn = c.n_dyn; cb.n_fix = c.n_fix;
    cb.norm_w = (l + 1 < c.layers) ? e->layers[l + 1].w.in_norm : e->final_norm; cb.norm_out = e->hin; cb.rms_eps = c.rms_eps;
    const bool cq_fits = n_tok + QKV / 16 <= 2 * e->n_cu;
    if (e->fuse_cq && cq_fits && l + 1 < c.layers && !e->probe_on()) {
        e->cb_stash = cb;
        e->cb_pending = true;
        e->cb_ep_layer = l;
        return 0;
    }
    // own launch: the same counters (umoe_ep_xfer.n_cwg > 0 selects them in combine_kernel<true>)
    umoe_ep_xfer x{};
    for (int p = 0; p < ep; ++p) x.peer_base[p] = e->ep_peers[p];
    x.rank = rank; x.size = ep; x.loopback = d.loopback; x.step = e->ep_words; x.err = e->ep_words + 1;
    x.layer = l; x.layers = c.layers; x.rows = n_tok; x.row_bytes = D * 2; x.kind = 1;
    x.round = e->ep_words + 2; x.n_cwg = d.n_cwg;
    cb.ep_xfer = &x;
    rc = umoe_unpermute_combine_fwd(&cb, s);
    PROF(K_COMBINE);
    return rc;
}

static int run_moe_ep(umoe_engine* e, int l, int n_tok, hipStream_t s) {
    if (e->ep_mode == UMOE_EP_RCCL) return run_moe_ep_rccl(e, l, n_tok, s);
    if (e->epf_ready && n_tok == e->c.rows) return run_moe_ep_flat(e, l, n_tok, s);
    const umoe_engine_cfg& c = e->c;
    const int D = c.hidden, G = c.n_real + c.n_fix, GPL = e->groups_per_layer(), E = c.n_dyn + c.n_fix;
    const int Imax = c.inter_dyn > c.inter_shared ? c.inter_dyn : c.inter_shared;
    const int ep = c.ep_size, El = e->E_loc, rank = c.ep_rank;
    const LayerDev& L = e->layers[l];
    const umoe_group_t* g = e->d_groups + (size_t)l * GPL;
    const umoe_group_t* gh = e->h_groups.data() + (size_t)l * GPL;
    UMOE_REQUIRE(e->ep_mode >= 0, "umoe_engine: expert parallel engine is not connected (umoe_engine_ep_connect)");
    UMOE_REQUIRE(c.n_dyn == 9 && c.n_fix == 2 && (D == 2048 || D == 4096) && El <= UMOE_MT_MAXG,
                 "umoe_engine: the expert parallel peer path is built for n_dyn 9 / n_fix 2, D 2048 / 4096");
    int rc;
    umoe_router_args ra{};
    ra.x = e->x1; ra.gate_w = L.w.gate_w; ra.norm_w = L.w.post_norm; ra.h_out = e->h2; ra.S = n_tok; ra.D = D;
    ra.n_dyn = c.n_dyn; ra.n_real = c.n_real; ra.n_fix = c.n_fix; ra.logits_bf16 = 1; ra.top_p = c.top_p;
    ra.fixed_top_k = c.fixed_top_k; ra.jitter_eps = c.jitter_eps; ra.rms_eps = c.rms_eps;
    ra.logits_out = e->r_logits; ra.sel = e->r_sel; ra.routing_w = e->r_routing; ra.global_w = e->r_global; ra.moe_w = e->r_moe;
    ra.expert_mask = e->all_mask + (size_t)l * c.rows * E;
    ra.top_k = e->all_topk + (size_t)l * c.rows;
    const size_t tile = (size_t)n_tok * D * 2, chunk_y = (size_t)El * tile;
    umoe_ep_xfer x{};
    for (int p = 0; p < ep; ++p) x.peer_base[p] = e->ep_peers[p];
    x.rank = rank; x.size = ep; x.loopback = e->ep_mode == UMOE_EP_LOOPBACK; x.step = e->ep_words; x.err = e->ep_words + 1;
    x.layer = l; x.layers = c.layers; x.rows = n_tok; x.row_bytes = D * 2;
    // ---- RMSNorm of the own rows (-> h2) + first exchange
    x.kind = 0; x.n_sub = 1; x.chunk = tile; x.data_off = UMOE_EP_FLAG_BYTES;
    umoe_router_args rn = ra;
    rn.norm_only = 1;
    if ((rc = umoe_router_norm_push(&rn, x, s))) return rc;
    ra.h_out = nullptr;
    PROF(K_ROUTER);
    // ---- shared experts gate/up on the own rows while the rows travel (K split of the ep_size 1 launch: 8 waves, 1-step chunks)
    umoe_gemm_args sg{};
    sg.groups = g + 2 + c.n_real; sg.groups_host = gh + 2 + c.n_real; sg.num_groups = c.n_fix; sg.max_rows = n_tok;
    sg.max_n_blocks = 2 * c.inter_shared / 16; sg.max_k = D; sg.a = e->h2; sg.lda = D; sg.out = e->hbuf; sg.ldo = Imax; sg.n_valid = Imax;
    sg.prologue = UMOE_PRO_PLAIN; sg.epilogue = UMOE_EPI_SWIGLU; sg.nt = 2; sg.waves = 8;
    if ((rc = umoe_grouped_gemm(&sg, s))) return rc;
    PROF(K_GATEUP);
    // ---- the peers' rows (and the own ones) into operand-order tiles
    x.dst = (char*)e->xgp; x.chunk = tile;
    {
        umoe_ep_xfer xp = x;
        // (the packed tiles are 16 rows apart whatever n_tok is; the slab tiles n_tok rows)
        if ((rc = umoe_ep_pull_pack(xp, e->h2, s))) return rc;
    }
    PROF(K_DISPATCH);
    // ---- local experts over every rank's rows: one pass over each expert's weights
    umoe_mt_args gu{};
    for (int q = 0; q < El; ++q) gu.w[q] = L.exp_gu[q];
    gu.num_groups = El; gu.n_blocks = 2 * c.inter_dyn / 16; gu.k = D; gu.tiles = ep; gu.n_rows = n_tok;
    gu.b = e->xgp; gu.b_group_tiles = 0; gu.h_out = e->hpk; gu.epilogue = UMOE_EPI_SWIGLU;
    gu.fused_router = e->fuse_router ? &ra : nullptr;
    if ((rc = umoe_gemm_mt(&gu, s))) return rc;
    if (!e->fuse_router) {
        if ((rc = umoe_router_fwd(&ra, s))) return rc;
    }
    PROF(K_GATEUP);
    uint16_t* yloc = e->ybuf + (size_t)G * n_tok * D;                        // [dest rank][local expert][row][D]
    uint16_t* slab_ret = reinterpret_cast<uint16_t*>(e->ep_region + UMOE_EP_FLAG_BYTES + (size_t)ep * tile);   // [global expert][row][D]
    umoe_mt_args dn{};
    for (int q = 0; q < El; ++q) dn.w[q] = L.exp_dn[q];
    dn.num_groups = El; dn.n_blocks = D / 16; dn.k = c.inter_dyn; dn.tiles = ep; dn.n_rows = n_tok;
    dn.b = e->hpk; dn.b_group_tiles = ep; dn.ldo = D; dn.epilogue = UMOE_EPI_BF16;
    for (int q = 0; q < El; ++q)
        for (int t = 0; t < ep; ++t)
            dn.y_out[q][t] = (t == rank) ? slab_ret + (size_t)(rank * El + q) * n_tok * D : yloc + (size_t)(t * El + q) * n_tok * D;
    if ((rc = umoe_gemm_mt(&dn, s))) return rc;
    PROF(K_DOWN);
    // ---- second exchange: my experts' outputs to the rows' owners
    x.kind = 1; x.n_sub = El; x.chunk = chunk_y; x.data_off = UMOE_EP_FLAG_BYTES + (size_t)ep * tile;
    x.src = (const char*)yloc; x.src_stride = (long)chunk_y;
    if ((rc = umoe_ep_push(x, s))) return rc;
    PROF(K_DISPATCH);
    umoe_gemm_args sd{};     // shared experts down while the outputs travel (K split of the ep_size 1 launch: 8 waves, 2-step chunks)
    sd.groups = g + 2 + G + c.n_real; sd.groups_host = gh + 2 + G + c.n_real; sd.num_groups = c.n_fix; sd.max_rows = n_tok;
    sd.max_n_blocks = D / 16; sd.max_k = c.inter_shared; sd.a = e->hbuf; sd.lda = Imax; sd.out = e->ybuf; sd.ldo = D; sd.n_valid = D;
    sd.prologue = UMOE_PRO_PLAIN; sd.epilogue = UMOE_EPI_BF16; sd.nt = 2; sd.waves = 8;
    if ((rc = umoe_grouped_gemm(&sd, s))) return rc;
    PROF(K_DOWN);
    // ---- combine + residual -> next layer input; waits for the peers' rows itself, reads the return slab in place
    umoe_combine_args cb{};
    cb.y_slots = slab_ret; cb.shared_row0 = -1; cb.slot_of = nullptr; cb.moe_w = e->r_moe;
    cb.expert_mask = ra.expert_mask; cb.mask_ld = E; cb.dense_rows = n_tok;
    cb.y_shared = e->ybuf + (size_t)n_tok * c.n_real * D; cb.global_w = e->r_global;
    cb.resid = e->x1; cb.out = e->x; cb.S = n_tok; cb.D = D; cb.n_real = c.n_real; cb.n_dyn = c.n_dyn; cb.n_fix = c.n_fix;
    cb.norm_w = (l + 1 < c.layers) ? e->layers[l + 1].w.in_norm : e->final_norm; cb.norm_out = e->hin; cb.rms_eps = c.rms_eps;
    cb.ep_xfer = &x;
    rc = umoe_unpermute_combine_fwd(&cb, s);
    PROF(K_COMBINE);
    return rc;
}

// ------------------------------------------------------------------------------------ one layer
static int run_layer(umoe_engine* e, int l, int n_tok, int T, int splits, hipStream_t s) {
    const umoe_engine_cfg& c = e->c;
    const int D = c.hidden, HD = c.heads * c.head_dim, QKV = (c.heads + 2 * c.kv_heads) * c.head_dim;
    const int G = c.n_real + c.n_fix, GPL = e->groups_per_layer(), E = c.n_dyn + c.n_fix;
    const int Imax = c.inter_dyn > c.inter_shared ? c.inter_dyn : c.inter_shared;
    const LayerDev& L = e->layers[l];
    const umoe_group_t* g = e->d_groups + (size_t)l * GPL;
    const umoe_group_t* gh = e->h_groups.data() + (size_t)l * GPL;   // same table, host side
    const size_t kv_l = (size_t)l * c.rows * c.kv_heads * c.Lmax * c.head_dim;
    int rc;
    // 1. RMSNorm + QKV (+bias)                                   model.py:227, Qwen2_5_VLAttention q/k/v_proj
    umoe_gemm_args a{};
    a.groups = g; a.groups_host = gh; a.num_groups = 1; a.max_rows = n_tok; a.max_n_blocks = QKV / 16; a.max_k = D;
    a.a = e->hin; a.lda = D; a.out = e->qkv; a.ldo = QKV; a.n_valid = QKV;   // hin = RMSNorm(x) from the previous combine
    a.prologue = UMOE_PRO_PLAIN; a.epilogue = UMOE_EPI_BF16;
    PROF(-1);
    // many rows (prefill): the compute-bound tiled MFMA kernel on the row-major weights; decode: weight streaming
    const bool tiled = L.has_rm && n_tok >= 64 && e->tiled_prefill;
    if (tiled) {
        umoe_tgroup_t tg{};
        tg.w = L.w.rm_qkv; tg.bias = L.w.qkv_b; tg.static_count = n_tok; tg.n = QKV; tg.k = D; tg.ldw = D;
        umoe_tgemm_args ta{};
        ta.groups = &tg; ta.num_groups = 1; ta.max_rows = n_tok; ta.a = e->hin; ta.lda = D; ta.out = e->qkv; ta.ldo = QKV;
        ta.epilogue = UMOE_EPI_BF16;
        rc = umoe_tiled_gemm(&ta, s);
    } else {
        rc = 1;
        if (e->cb_pending) {
            // the previous layer's combine (+ residual + this layer's input RMSNorm) rides in this launch and hands the rows over
            e->cb_pending = false;
            umoe_rider2 r2{};
            r2.n_riders = n_tok; r2.cb = e->cb_stash;
            umoe_rider_pub cpub{};
            cpub.flags = e->ep_words + 1024 + 16 * UMOE_FLAG_REPL; cpub.step = e->ep_words; cpub.layer = l; cpub.layers = c.layers; cpub.err = e->ep_words + 1;
            const bool epc = e->cb_ep_layer >= 0;
            umoe_ep_xfer x{};
            if (epc) {       // the rows of the previous layer's routed experts sit in the return slab: the riders wait for the owners' counters
                r2.ep_region = e->ep_region; r2.ep_round = e->ep_words + 2; r2.ep_layer = e->cb_ep_layer; r2.ep_layers = c.layers;
                r2.ep_size = c.ep_size; r2.ep_n_cwg = e->epf.n_cwg; r2.ep_err = e->ep_words + 1;
                for (int p = 0; p < c.ep_size; ++p) x.peer_base[p] = e->ep_peers[p];
                x.rank = c.ep_rank; x.size = c.ep_size; x.loopback = e->ep_mode == UMOE_EP_LOOPBACK; x.step = e->ep_words; x.err = e->ep_words + 1;
                x.layer = e->cb_ep_layer; x.layers = c.layers; x.rows = n_tok; x.row_bytes = D * 2; x.kind = 1;
                x.round = e->ep_words + 2; x.n_cwg = e->epf.n_cwg;
                e->cb_ep_layer = -1;
            }
            rc = umoe_gemm_riders(&a, epc ? 4 : 2, &r2, &cpub, s);
            if (rc == 1) {       // shapes do not fit: the two launches
                if (epc) e->cb_stash.ep_xfer = &x;
                if ((rc = umoe_unpermute_combine_fwd(&e->cb_stash, s)) == 0) rc = 1;
            }
        }
        if (rc == 1) rc = umoe_grouped_gemm(&a, s);
    }
    if (rc) return rc;
    PROF(K_QKV);
    // 2. mRoPE + KV append
    umoe_rope_args r{};
    r.qkv = e->qkv; r.cos_tab = e->cos_tab; r.sin_tab = e->sin_tab; r.pos3 = e->pos3; r.kv_pos = e->kv_pos;
    r.n_tok = n_tok; r.T = T; r.H = c.heads; r.KVH = c.kv_heads; r.hd = c.head_dim;
    r.sec0 = c.mrope0; r.sec1 = c.mrope1; r.sec2 = c.mrope2; r.Lmax = c.Lmax;
    r.q_out = e->q_r; r.k_cache = e->k_cache + kv_l; r.v_cache = e->v_cache + kv_l;
    const bool fuse_rope = (T == 1) && (c.mrope0 % 8 == 0) && ((c.mrope0 + c.mrope1) % 8 == 0);
    if (!fuse_rope) {   // prefill: rope + append for all T positions first
        if ((rc = umoe_qkv_mrope_kvappend(&r, s))) return rc;
        PROF(K_ROPE);
    }
    // 3. attention (decode: mRoPE of q / new k and the KV append are fused into the kernel)
    umoe_attn_args t{};
    t.q = e->q_r; t.k_cache = r.k_cache; t.v_cache = r.v_cache; t.kv_start = e->kv_start; t.q_pos0 = e->q_pos0;
    t.rows = c.rows; t.nq = T; t.H = c.heads; t.KVH = c.kv_heads; t.hd = c.head_dim; t.Lmax = c.Lmax; t.splits = splits;
    t.scale = 1.0f / sqrtf((float)c.head_dim); t.part_o = e->part_o; t.part_ml = e->part_ml; t.out = e->attn_out;
    if (fuse_rope) {
        t.qkv_raw = e->qkv; t.cos_tab = e->cos_tab; t.sin_tab = e->sin_tab; t.pos3 = e->pos3;
        t.sec0 = c.mrope0; t.sec1 = c.mrope1; t.sec2 = c.mrope2;
    }
    if ((rc = (T > 1 ? umoe_attn_prefill_fwd(&t, s) : umoe_attn_decode(&t, s)))) return rc;   // prefill: MFMA flash kernel
    PROF(K_ATTN);
    // 4. o_proj + residual                                        model.py:238
    umoe_gemm_args o{};
    o.groups = g + 1; o.groups_host = gh + 1; o.num_groups = 1; o.max_rows = n_tok; o.max_n_blocks = D / 16; o.max_k = HD;
    o.a = e->attn_out; o.lda = HD; o.resid = e->x; o.out = e->x1; o.ldo = D; o.n_valid = D;
    o.prologue = UMOE_PRO_PLAIN; o.epilogue = UMOE_EPI_BF16_RESID;
    // decode with the flat expert launch: o_proj is computed INSIDE that launch (umoe_moe_flat with the o_proj arguments); the conditions are
    // those under which the flat launch is taken below (the same flags and shapes; should it refuse after all, o_proj is launched there)
    bool o_in_flat = false;
    {
        const char* fv = getenv("UMOE_FLAT_MOE");
        const bool flat = fv ? atoi(fv) != 0 : e->flat_moe;
        const bool densef = dense_mode(e, n_tok) && !e->ep_decode(n_tok);
        const char* ov = getenv("UMOE_FUSE_O");      // (read per enqueue like UMOE_FLAT_MOE: A/B scripts toggle it between captures)
        const bool fuse_o = ov ? atoi(ov) != 0 : e->fuse_o;
        o_in_flat = fuse_o && flat && densef && T == 1 && !tiled && e->fuse_router && e->rider_pub && e->fuse_moe && e->n_cu > 0 && c.n_dyn == 9 && c.n_fix == 2 &&
                    D == 2048 && HD == 2048 && n_tok <= 16 &&
                    umoe_moe_flat_feasible(e->n_cu < 256 ? e->n_cu : 256, n_tok, D, c.inter_dyn, c.inter_shared, c.n_real, c.n_fix);
    }
    if (tiled) {
        umoe_tgroup_t tg{};
        tg.w = L.w.rm_o; tg.static_count = n_tok; tg.n = D; tg.k = HD; tg.ldw = HD;
        umoe_tgemm_args ta{};
        ta.groups = &tg; ta.num_groups = 1; ta.max_rows = n_tok; ta.a = e->attn_out; ta.lda = HD; ta.resid = e->x; ta.out = e->x1;
        ta.ldo = D; ta.epilogue = UMOE_EPI_BF16_RESID;
        rc = umoe_tiled_gemm(&ta, s);
    } else if (!o_in_flat) {
        rc = umoe_grouped_gemm(&o, s);
    } else {
        rc = 0;
    }
    if (rc) return rc;
    if (!o_in_flat) PROF(K_OPROJ);
    if (e->probe_x1 && n_tok == c.rows && !o_in_flat)
        UMOE_HIP(hipMemcpyAsync(e->probe_x1 + (size_t)l * c.rows * D, e->x1, (size_t)c.rows * D * 2, hipMemcpyDeviceToDevice, s));
    if (e->ep_decode(n_tok)) return run_moe_ep(e, l, n_tok, s);
    // 5. RMSNorm + router                                         model.py:240, core.py:246-291
    umoe_router_args ra{};
    const bool dense = dense_mode(e, n_tok);
    ra.x = e->x1; ra.gate_w = L.w.gate_w; ra.norm_w = L.w.post_norm; ra.h_out = e->h2; ra.S = n_tok; ra.D = D;
    ra.n_dyn = c.n_dyn; ra.n_real = c.n_real; ra.n_fix = c.n_fix; ra.logits_bf16 = 1; ra.top_p = c.top_p;
    ra.fixed_top_k = c.fixed_top_k; ra.jitter_eps = c.jitter_eps; ra.rms_eps = c.rms_eps;
    ra.logits_out = e->r_logits; ra.top_k = e->r_topk; ra.sel = e->r_sel; ra.expert_mask = e->r_mask;
    ra.routing_w = e->r_routing; ra.global_w = e->r_global; ra.moe_w = e->r_moe;
    if (n_tok == c.rows) {  // keep per-layer statistics of decode steps for the parity tests
        ra.expert_mask = e->all_mask + (size_t)l * c.rows * E;
        ra.top_k = e->all_topk + (size_t)l * c.rows;
    }
    // dense decode: the GEMMs do not read the routing results (every expert computes every row), only the combine does -- so the
    // router rides INSIDE the gate/up launch as 16 extra workgroups and only the RMSNorm (h2, which gate/up needs) stays in the
    // chain as its own small launch: the 4.4 us serial routing chain per token leaves the critical path.
    const bool fuse_router = dense && e->fuse_router && !tiled && c.n_dyn == 9 && c.n_fix == 2 &&
                             (D == 2048 || D == 4096) && n_tok <= 16;
    // In-launch hand-offs need every workgroup of the launch RESIDENT at once (a waiting workgroup never yields its CU): the gate/up box
    // of 8-wave, 256-register workgroups admits ONE per CU, the flat launch is sized to the CU count itself.  A device that exposes
    // fewer CUs (partition, CU mask) takes the launch-per-kernel path instead of discovering it by a timeout.
    const int gu_box = G * ceil_div(2 * Imax / 16, 14);
    const bool box_fits = e->n_cu <= 0 || gu_box <= e->n_cu;
    const bool flat_ok = dense && e->flat_moe && e->fuse_moe && e->n_cu > 0 && c.n_dyn == 9 && c.n_fix == 2 &&
                         umoe_moe_flat_feasible(e->n_cu < 256 ? e->n_cu : 256, n_tok, D, c.inter_dyn, c.inter_shared, c.n_real, c.n_fix);
    const bool pub_riders = fuse_router && e->rider_pub && (box_fits || flat_ok);
    if (o_in_flat && !(pub_riders && e->fuse_moe && flat_ok)) {      // (cannot happen with the conditions above; never run a layer without its o_proj)
        if ((rc = umoe_grouped_gemm(&o, s))) return rc;
        o_in_flat = false;
    }
    if (pub_riders) {
        rc = 0;                  // no launch here: the riders write h2 inside the gate/up launch and hand it over (ra.h_out stays h2)
    } else if (fuse_router) {
        umoe_router_args rn = ra;
        rn.norm_only = 1;
        rc = umoe_router_fwd(&rn, s);
        ra.h_out = nullptr;
    } else if (dense) {
        rc = umoe_router_fwd(&ra, s);   // no dispatch tables: the combine reads the mask
    } else {
        rc = umoe_router_dispatch_fwd(&ra, e->counts, e->offsets, e->slot_token, e->slot_of, s);
    }
    if (rc) return rc;
    PROF(K_ROUTER);
    umoe_rider_pub rpub{};
    rpub.flags = e->ep_words + 1024; rpub.step = e->ep_words; rpub.layer = l; rpub.layers = c.layers; rpub.err = e->ep_words + 1;
    // 7./8. experts: routed + shared share one launch (the fused / flat launch) or one launch per GEMM
    umoe_gemm_args gu{};
    gu.groups = g + 2; gu.groups_host = gh + 2; gu.num_groups = G; gu.max_rows = n_tok;
    gu.max_n_blocks = 2 * Imax / 16; gu.max_k = D;
    gu.a = e->h2; gu.lda = D; gu.out = e->hbuf; gu.ldo = Imax; gu.n_valid = Imax;
    gu.prologue = UMOE_PRO_PLAIN; gu.epilogue = UMOE_EPI_SWIGLU;
    if (dense) {             // per-CU byte balance decides this kernel (see umoe_gemm.hip): 7 pairs per workgroup, flat slices
        gu.nt = 14;
        if (fuse_router) gu.fused_router = &ra;
        if (pub_riders) {
            gu.rider_pub = &rpub;
            gu.groups_host = e->h_gu_pub.data() + (size_t)l * G;     // shared experts first (riders early in dispatch order)
        }
    }
    // (dense mode with the post-attention RMSNorm in this launch's staging prologue, so that it would not wait for the
    //  router at all, was measured: 46.9 vs 37.7 us per launch -- 387 workgroups redoing the norm of all 16 rows costs
    //  more than the dependency it removes; the router kernel writes the normalised rows h2 once instead)
    if (tiled && G <= 12) {
        umoe_tgroup_t tg[12];
        memset(tg, 0, sizeof(tg));
        for (int x = 0; x < G; ++x) {
            const umoe_group_t& src = gh[2 + x];     // same row tables as the weight-streaming groups
            const bool sh = x >= c.n_real;
            tg[x].w = sh ? L.rm_sg[x - c.n_real] : L.rm_eg[x];
            tg[x].w2 = sh ? L.rm_su[x - c.n_real] : L.rm_eu[x];
            tg[x].rows = src.rows; tg[x].row_off = src.row_off; tg[x].count = src.count; tg[x].static_count = src.static_count;
            tg[x].a_row_base = src.a_row_base; tg[x].out_row_base = src.out_row_base;
            tg[x].n = sh ? c.inter_shared : c.inter_dyn; tg[x].k = D; tg[x].ldw = D;
        }
        umoe_tgemm_args ta{};
        ta.groups = tg; ta.num_groups = G; ta.max_rows = n_tok; ta.a = e->h2; ta.lda = D; ta.out = e->hbuf; ta.ldo = Imax;
        ta.epilogue = UMOE_EPI_SWIGLU;
        rc = umoe_tiled_gemm(&ta, s);
    } else if (!(pub_riders && e->fuse_moe)) {
        rc = umoe_grouped_gemm(&gu, s);
    }
    if (rc) return rc;
    if (!(pub_riders && e->fuse_moe)) PROF(K_GATEUP);
    umoe_gemm_args dn{};
    dn.groups = g + 2 + G; dn.groups_host = gh + 2 + G; dn.num_groups = G; dn.max_rows = n_tok; dn.max_n_blocks = D / 16;
    dn.max_k = Imax;
    dn.a = e->hbuf; dn.lda = Imax; dn.out = e->ybuf; dn.ldo = D; dn.n_valid = D;
    dn.prologue = UMOE_PRO_PLAIN; dn.epilogue = UMOE_EPI_BF16;
    if (dense) dn.nt = 6;    // 220 workgroups of 8 waves: 22.2 vs 23.7 us with 8 blocks per workgroup (scripts/kbench.py flat)
    if (tiled && G <= 12) {
        umoe_tgroup_t tg[12];
        memset(tg, 0, sizeof(tg));
        for (int x = 0; x < G; ++x) {
            const umoe_group_t& src = gh[2 + G + x];
            const bool sh = x >= c.n_real;
            tg[x].w = sh ? L.rm_sd[x - c.n_real] : L.rm_ed[x];
            tg[x].row_off = src.row_off; tg[x].count = src.count; tg[x].static_count = src.static_count;
            tg[x].a_row_base = src.a_row_base; tg[x].out_row_base = src.out_row_base;
            tg[x].n = D; tg[x].k = sh ? c.inter_shared : c.inter_dyn; tg[x].ldw = tg[x].k;
        }
        umoe_tgemm_args ta{};
        ta.groups = tg; ta.num_groups = G; ta.max_rows = n_tok; ta.a = e->hbuf; ta.lda = Imax; ta.out = e->ybuf; ta.ldo = D;
        ta.epilogue = UMOE_EPI_BF16;
        rc = umoe_tiled_gemm(&ta, s);
        if (rc) return rc;
        PROF(K_DOWN);
    } else if (pub_riders && e->fuse_moe) {
        // both expert GEMMs in one launch (words 64.. of ep_words: one flag per gate/up workgroup); shapes that do not allow it
        // fall back to the two launches
        rc = 1;
        {
            const char* fv = getenv("UMOE_FLAT_MOE");      // (read per enqueue: the step graph captures the choice; A/B scripts toggle it)
            const bool flat = fv ? atoi(fv) != 0 : e->flat_moe;
            const int n_wg = e->n_cu < 256 ? e->n_cu : 256;
            if (flat && n_wg > 0) rc = umoe_moe_flat(&gu, &dn, e->ep_words + 64, 512 - 64, n_wg, s, o_in_flat ? &o : nullptr, e->ep_words + 2688);
            e->expert_launch = rc == 0 ? 2 : 0;
            if (rc == 1 && o_in_flat) {      // the flat launch refused after all: o_proj as its own launch in front of whatever runs instead
                if ((rc = umoe_grouped_gemm(&o, s))) return rc;
                rc = 1;
                o_in_flat = false;
            }
        }
        if (rc == 1 && box_fits) {
            rc = umoe_moe_fused(&gu, &dn, e->ep_words + 64, 512 - 64, s);
            if (rc == 0) e->expert_launch = 1;
        }
        UMOE_REQUIRE(!(rc == 1 && !box_fits), "umoe_engine: no expert launch with in-launch hand-offs fits %d compute units (UMOE_RIDER_PUB=0 selects the launch-per-kernel path)", e->n_cu);
        if (rc == 1) {
            if ((rc = umoe_grouped_gemm(&gu, s))) return rc;
            rc = umoe_grouped_gemm(&dn, s);
        }
        if (rc) return rc;
        if (o_in_flat && e->probe_x1 && n_tok == c.rows)      // (x1 was made inside the launch)
            UMOE_HIP(hipMemcpyAsync(e->probe_x1 + (size_t)l * c.rows * D, e->x1, (size_t)c.rows * D * 2, hipMemcpyDeviceToDevice, s));
        PROF(K_GATEUP);          // (the fused launch is booked as gate/up: zero down launches tell the reader which form ran)
    } else {
        rc = umoe_grouped_gemm(&dn, s);
        if (rc) return rc;
        PROF(K_DOWN);
    }
    // 9. combine + residual -> next layer input                   core.py:488,342-351; model.py:242
    umoe_combine_args cb{};
    cb.y_slots = e->ybuf; cb.shared_row0 = -1; cb.slot_of = dense ? nullptr : e->slot_of; cb.moe_w = e->r_moe;
    cb.expert_mask = ra.expert_mask; cb.mask_ld = E; cb.dense_rows = n_tok;
    cb.y_shared = c.n_fix ? e->ybuf + (size_t)n_tok * c.n_real * D : nullptr; cb.global_w = e->r_global;
    cb.resid = e->x1; cb.out = e->x; cb.S = n_tok; cb.D = D; cb.n_real = c.n_real; cb.n_dyn = c.n_dyn; cb.n_fix = c.n_fix;
    // fused RMSNorm for the consumer of x: the next layer's input_layernorm, or the final norm in front of the head
    cb.norm_w = (l + 1 < c.layers) ? e->layers[l + 1].w.in_norm : e->final_norm; cb.norm_out = e->hin; cb.rms_eps = c.rms_eps;
    const bool cq_fits = e->n_cu <= 0 || n_tok + QKV / 16 <= 2 * e->n_cu;      // riders + QKV tiles resident at once (two 4-wave workgroups per CU)
    if (dense && T == 1 && !tiled && e->fuse_cq && e->rider_pub && cq_fits && l + 1 < c.layers && D == 2048 && n_tok <= 16 && c.n_fix >= 1 && !e->probe_on()) {
        e->cb_stash = cb;        // issued by the next layer's QKV launch (run_layer(l + 1) follows immediately)
        e->cb_pending = true;
        return 0;
    }
    rc = umoe_unpermute_combine_fwd(&cb, s);
    PROF(K_COMBINE);
    return rc;
}

// ------------------------------------------------------------------------------------ prefill
// The in-launch hand-offs derive their epoch from the step word (ep_words[0]), which only step_prep_kernel advances.  A prefill whose
// token count equals the decode rows (T == 1) takes the same rider / fused paths as a decode step: it gets an epoch of its own, or
// its waiters would pass at once on the flags of the last decode step of a re-used engine.  Every expert-parallel rank prefills, so
// the ranks' counters stay equal.
__global__ void epoch_bump_kernel(uint32_t* ep_step) {
    if (threadIdx.x == 0) ep_step[0] += 1u;
}

extern "C" int umoe_engine_prefill(umoe_engine* e, const uint16_t* x, const uint8_t* valid_host, int T,
                                   umoe_stream_t stream) {
    return umoe_engine_prefill_pos(e, x, valid_host, T, nullptr, nullptr, stream);
}

// positions / KV slots / first generated position of a prompt into the engine's device state (shared by the engine's own prefill and by
// a prefill computed outside it)
static int prefill_state(umoe_engine* e, const uint8_t* valid_host, int T, const int32_t* pos3_host, const int32_t* next_pos_host, hipStream_t s) {
    const umoe_engine_cfg& c = e->c;
    const int n_tok = c.rows * T;
    // positions: cumsum(mask)-1, masked -> 1 (model.py:1113-1114); kv slot = t; first valid slot per row
    std::vector<int32_t> pos((size_t)3 * n_tok), kvp(n_tok), start(c.rows), q0(c.rows, 0), vc(c.rows);
    for (int r = 0; r < c.rows; ++r) {
        int cnt = 0, first = T;
        for (int t = 0; t < T; ++t) {
            const int v = valid_host[(size_t)r * T + t] != 0;
            cnt += v;
            if (v && first == T) first = t;
            const int p = v ? cnt - 1 : 1;
            for (int k = 0; k < 3; ++k) {
                const int pk = pos3_host ? pos3_host[((size_t)k * c.rows + r) * T + t] : p;
                UMOE_REQUIRE(pk >= 0 && pk < e->max_pos, "umoe_engine_prefill: position %d outside the rope table (%d)", pk, e->max_pos);
                pos[(size_t)k * n_tok + r * T + t] = pk;
            }
            kvp[r * T + t] = t;
        }
        // left padding assumed (tokenizer padding_side="left", mod.py:104): valid keys are [first, L)
        for (int t = first; t < T; ++t)
            UMOE_REQUIRE(valid_host[(size_t)r * T + t], "umoe_engine_prefill: row %d is not left-padded", r);
        start[r] = first;
        vc[r] = next_pos_host ? next_pos_host[r] : cnt;           // position of the first generated token (decode: + steps taken)
        UMOE_REQUIRE(vc[r] >= 0 && vc[r] + c.Lmax - T < e->max_pos, "umoe_engine_prefill: rope table too short");
    }
    UMOE_HIP(hipMemcpyAsync(e->pos3, pos.data(), pos.size() * 4, hipMemcpyHostToDevice, s));
    UMOE_HIP(hipMemcpyAsync(e->kv_pos, kvp.data(), kvp.size() * 4, hipMemcpyHostToDevice, s));
    UMOE_HIP(hipMemcpyAsync(e->kv_start, start.data(), start.size() * 4, hipMemcpyHostToDevice, s));
    UMOE_HIP(hipMemcpyAsync(e->q_pos0, q0.data(), q0.size() * 4, hipMemcpyHostToDevice, s));
    UMOE_HIP(hipMemcpyAsync(e->valid_count, vc.data(), vc.size() * 4, hipMemcpyHostToDevice, s));
    UMOE_HIP(hipStreamSynchronize(s));  // host vectors go out of scope
    return 0;
}

extern "C" int umoe_engine_prefill_pos(umoe_engine* e, const uint16_t* x, const uint8_t* valid_host, int T, const int32_t* pos3_host,
                                       const int32_t* next_pos_host, umoe_stream_t stream) {
    UMOE_REQUIRE(e && x && valid_host && T > 0, "umoe_engine_prefill: bad argument");
    const umoe_engine_cfg& c = e->c;
    UMOE_REQUIRE(T < c.Lmax, "umoe_engine_prefill: prompt length %d does not fit Lmax %d", T, c.Lmax);
    UMOE_REQUIRE(e->final_norm, "umoe_engine_prefill: globals not set");
    UMOE_REQUIRE(c.ep_size == 1 || e->ep_mode >= 0, "umoe_engine_prefill: expert parallel engine is not connected (umoe_engine_ep_connect)");
    UMOE_REQUIRE(c.ep_size == 1 || (c.rows * T >= 64 && e->tiled_prefill), "umoe_engine_prefill: expert parallel prefill runs replicated on the tiled path: needs rows*T >= 64 (got %d)", c.rows * T);
    for (int l = 0; l < c.layers && c.ep_size > 1; ++l)
        UMOE_REQUIRE(e->layers[l].has_rm, "umoe_engine_prefill: this expert parallel engine holds its LOCAL experts only (no row-major tensors of the others): "
                                          "prefill outside the engine and hand the KV cache over (umoe_engine_prefill_external)");
    hipStream_t s = (hipStream_t)stream;
    const int n_tok = c.rows * T;
    int rc;
    if ((rc = ensure_workspace(e, n_tok))) return rc;
    if ((rc = build_groups(e, n_tok, s))) return rc;
    if ((rc = prefill_state(e, valid_host, T, pos3_host, next_pos_host, s))) return rc;
    UMOE_HIP(hipMemcpyAsync(e->x, x, (size_t)n_tok * c.hidden * 2, hipMemcpyDeviceToDevice, s));
    if ((rc = umoe_rmsnorm_residual_fwd(e->x, nullptr, e->layers[0].w.in_norm, c.rms_eps, n_tok, c.hidden, nullptr, e->hin, s)))
        return rc;
    e->cb_pending = false;
    epoch_bump_kernel<<<1, 64, 0, s>>>(e->ep_words);
    UMOE_LAUNCH_CHECK();
    for (int l = 0; l < c.layers; ++l)
        if ((rc = run_layer(e, l, n_tok, T, 1, s))) return rc;
    e->T_prompt = T;
    // switch the group table to decode shape now, outside any later graph capture
    if ((rc = build_groups(e, c.rows, s))) return rc;
    return 0;
}

// The prompt was run OUTSIDE the engine (an expert-parallel rank that holds only its local experts prefills through the module-level
// forward, whose DCMoE blocks exchange rows over torch.distributed, core.py:455-488) and its roped K / V were copied into the cache buffers
// (umoe_engine_buffer "k_cache" / "v_cache", slots [0, T) of every row): only the decode state is set here.
extern "C" int umoe_engine_prefill_external(umoe_engine* e, const uint8_t* valid_host, int T, const int32_t* pos3_host, const int32_t* next_pos_host,
                                            umoe_stream_t stream) {
    UMOE_REQUIRE(e && valid_host && T > 0, "umoe_engine_prefill_external: bad argument");
    const umoe_engine_cfg& c = e->c;
    UMOE_REQUIRE(T < c.Lmax, "umoe_engine_prefill_external: prompt length %d does not fit Lmax %d", T, c.Lmax);
    UMOE_REQUIRE(e->final_norm, "umoe_engine_prefill_external: globals not set");
    UMOE_REQUIRE(c.ep_size == 1 || e->ep_mode >= 0, "umoe_engine_prefill_external: expert parallel engine is not connected (umoe_engine_ep_connect)");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if ((rc = ensure_workspace(e, c.rows * T))) return rc;      // (pos3 / kv_pos are sized by the token count)
    if ((rc = prefill_state(e, valid_host, T, pos3_host, next_pos_host, s))) return rc;
    e->cb_pending = false;
    e->T_prompt = T;
    if ((rc = build_groups(e, c.rows, s))) return rc;
    return 0;
}

// ------------------------------------------------------------------------------------ decode step
// tokens[b][step] -> tok_in (CFG row doubling, model.py:945), positions / cache slots from device state
__global__ void step_prep_kernel(const int32_t* __restrict__ tokens, const int32_t* __restrict__ state, int B, int C,
                                 int Tmax, int T_prompt, int Lmax, const int32_t* __restrict__ valid_count,
                                 int32_t* tok_in, int32_t* pos3, int32_t* kv_pos, int32_t* q_pos0, uint32_t* ep_step) {
    const int row = blockIdx.x, b = row >> 1;
    if (row == 0 && threadIdx.x == 0) {
        ep_step[0] += 1u;   // epoch base of this step's in-launch hand-offs (read by later launches; a prefill bumps it too)
        ep_step[2] += 1u;   // decode steps only: base of the expert-parallel return counters' rounds (umoe_moe_ep.hip)
    }
    const int step = state[4 * B];
    const int n_dec = step - state[4 * B + 4];  // state[4B+4] = dec_step of the first decode call
    const int ts = min(max(step, 0), Tmax - 1);
    if ((int)threadIdx.x < C) tok_in[row * C + threadIdx.x] = tokens[((size_t)b * Tmax + ts) * C + threadIdx.x];
    if (threadIdx.x == 0) {
        const int rows = 2 * B;
        const int slot = min(T_prompt + n_dec, Lmax - 1);
        const int p = valid_count[row] + n_dec;
        pos3[row] = p;
        pos3[rows + row] = p;
        pos3[2 * rows + row] = p;
        kv_pos[row] = slot;
        q_pos0[row] = slot;
    }
}

static int enqueue_step(umoe_engine* e, const umoe_decode_io* io, hipStream_t s) {
    const umoe_engine_cfg& c = e->c;
    const int B = c.rows / 2, C = c.codec_channels, V = c.codec_vocab;
    int rc;
    PROF(-1);
    step_prep_kernel<<<dim3((unsigned)c.rows), 64, 0, s>>>(io->tokens, io->state, B, C, c.Tmax, e->T_prompt, c.Lmax,
                                                           e->valid_count, e->tok_in, e->pos3, e->kv_pos, e->q_pos0, e->ep_words);
    UMOE_LAUNCH_CHECK();
    if ((rc = umoe_codec_embed_sum(e->tok_in, e->codec_emb, c.rows, C, V, c.hidden, e->x, s))) return rc;
    if ((rc = umoe_rmsnorm_residual_fwd(e->x, nullptr, e->layers[0].w.in_norm, c.rms_eps, c.rows, c.hidden, nullptr, e->hin, s)))
        return rc;
    PROF(K_EMBED);
    e->cb_pending = false;
    for (int l = 0; l < c.layers; ++l) {
        if (e->probe_teach) {        // teacher-forced layer input (parity tests): x <- teach[l], hin <- RMSNorm(x)
            UMOE_HIP(hipMemcpyAsync(e->x, e->probe_teach + (size_t)l * c.rows * c.hidden, (size_t)c.rows * c.hidden * 2, hipMemcpyDeviceToDevice, s));
            if ((rc = umoe_rmsnorm_residual_fwd(e->x, nullptr, e->layers[l].w.in_norm, c.rms_eps, c.rows, c.hidden, nullptr, e->hin, s))) return rc;
        }
        if ((rc = run_layer(e, l, c.rows, 1, c.attn_splits, s))) return rc;
        if (e->probe_x)
            UMOE_HIP(hipMemcpyAsync(e->probe_x + (size_t)l * c.rows * c.hidden, e->x, (size_t)c.rows * c.hidden * 2, hipMemcpyDeviceToDevice, s));
        if (e->probe_logits)
            UMOE_HIP(hipMemcpyAsync(e->probe_logits + (size_t)l * c.rows * (c.n_dyn + c.n_fix), e->r_logits, (size_t)c.rows * (c.n_dyn + c.n_fix) * 2, hipMemcpyDeviceToDevice, s));
    }
    // final norm + codec head -> fp32 logits                       model.py:428, 982-983
    umoe_gemm_args h{};
    h.groups = e->d_groups + (size_t)c.layers * e->groups_per_layer();
    h.groups_host = e->h_groups.data() + (size_t)c.layers * e->groups_per_layer(); h.num_groups = 1; h.max_rows = c.rows;
    h.max_n_blocks = ceil_div(C * V, 16); h.max_k = c.hidden; h.a = e->hin; h.lda = c.hidden;   // hin = final norm(x)
    h.out = e->logits; h.ldo = C * V; h.n_valid = C * V;
    h.prologue = UMOE_PRO_PLAIN; h.epilogue = UMOE_EPI_F32;
    if ((rc = umoe_grouped_gemm(&h, s))) return rc;
    PROF(K_HEAD);
    umoe_sample_args sa{};
    sa.logits = e->logits; sa.B = B; sa.C = C; sa.V = V; sa.cfg_scale = io->cfg_scale; sa.temperature = io->temperature;
    sa.top_p = io->top_p; sa.eos_mul = io->eos_mul; sa.top_k = io->top_k; sa.eos = c.eos; sa.min_tokens = io->min_tokens;
    sa.step = io->state + 4 * B; sa.do_sample = io->do_sample; sa.seed = io->seed; sa.pred = e->pred;
    if ((rc = umoe_codec_head_cfg_sample(&sa, s))) return rc;
    PROF(K_SAMPLE);
    rc = umoe_delay_step(e->pred, io->tokens, io->state, e->d_delay, B, C, c.Tmax, c.eos, c.pad, e->max_delay, s);
    PROF(K_DELAY);
    return rc;
}

extern "C" int umoe_engine_decode_step(umoe_engine* e, const umoe_decode_io* io, umoe_stream_t stream) {
    UMOE_REQUIRE(e && io && io->tokens && io->state, "umoe_engine_decode_step: null argument");
    UMOE_REQUIRE(e->T_prompt > 0, "umoe_engine_decode_step: prefill first");
    return enqueue_step(e, io, (hipStream_t)stream);
}

// One eager decode step with a hipEvent after every kernel class; ms[K_NUM] accumulates the elapsed time between
// consecutive events (kernel + the launch gap in front of it), launches[K_NUM] the number of intervals.
extern "C" int umoe_engine_profile_step(umoe_engine* e, const umoe_decode_io* io, umoe_stream_t stream, float* ms,
                                        int* launches, int n) {
    UMOE_REQUIRE(e && io && ms && launches && n >= K_NUM, "umoe_engine_profile_step: need %d output slots", K_NUM);
    hipStream_t s = (hipStream_t)stream;
    e->prof = true;
    e->ev_used = 0;
    const int rc = enqueue_step(e, io, s);
    e->prof = false;
    if (rc) return rc;
    UMOE_HIP(hipStreamSynchronize(s));
    for (size_t i = 1; i < e->ev_used; ++i) {
        const int k = e->ev_kind[i];
        if (k < 0) continue;
        float t = 0.f;
        UMOE_HIP(hipEventElapsedTime(&t, e->ev[i - 1], e->ev[i]));
        ms[k] += t;
        launches[k] += 1;
    }
    return 0;
}

extern "C" int umoe_engine_capture(umoe_engine* e, const umoe_decode_io* io, umoe_stream_t stream) {
    UMOE_REQUIRE(e && io && io->tokens && io->state, "umoe_engine_capture: null argument");
    UMOE_REQUIRE(e->T_prompt > 0, "umoe_engine_capture: prefill first");
    UMOE_REQUIRE(!e->probe_on(), "umoe_engine_capture: the per-layer probe works on eager steps only (umoe_engine_set_probe)");
    hipStream_t s = (hipStream_t)stream;
    if (e->exec) { (void)hipGraphExecDestroy(e->exec); e->exec = nullptr; }
    if (e->graph) { (void)hipGraphDestroy(e->graph); e->graph = nullptr; }
    UMOE_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue_step(e, io, s);
    hipGraph_t g = nullptr;
    const hipError_t ec = hipStreamEndCapture(s, &g);
    if (rc) return rc;
    UMOE_REQUIRE(ec == hipSuccess && g, "umoe_engine_capture: hipStreamEndCapture failed: %s", hipGetErrorString(ec));
    e->graph = g;
    UMOE_HIP(hipGraphInstantiate(&e->exec, e->graph, nullptr, nullptr, 0));
    return 0;
}

extern "C" int umoe_engine_replay(umoe_engine* e, umoe_stream_t stream) {
    UMOE_REQUIRE(e && e->exec, "umoe_engine_replay: no captured step");
    UMOE_HIP(hipGraphLaunch(e->exec, (hipStream_t)stream));
    return 0;
}

extern "C" int umoe_engine_info(umoe_engine* e, const char* key) {
    if (!e || !key) return -1;
    if (!strcmp(key, "expert_launch")) return e->expert_launch;
    if (!strcmp(key, "n_cu")) return e->n_cu;
    return -1;
}

extern "C" int umoe_engine_set_probe(umoe_engine* e, const uint16_t* teach_x, uint16_t* dump_x1, uint16_t* dump_x, uint16_t* dump_logits) {
    UMOE_REQUIRE(e, "umoe_engine_set_probe: null engine");
    UMOE_REQUIRE(e->c.ep_size == 1 || !(teach_x || dump_x1 || dump_x || dump_logits), "umoe_engine_set_probe: not with expert parallel engines");
    e->probe_teach = teach_x; e->probe_x1 = dump_x1; e->probe_x = dump_x; e->probe_logits = dump_logits;
    return 0;
}

extern "C" const void* umoe_engine_buffer(umoe_engine* e, const char* name, size_t* bytes) {
    if (!e || !name) return nullptr;
    const umoe_engine_cfg& c = e->c;
    const int E = c.n_dyn + c.n_fix;
    struct Item { const char* n; const void* p; size_t b; };
    const Item items[] = {
        {"x", e->x, (size_t)c.rows * c.hidden * 2},
        {"logits", e->logits, (size_t)c.rows * c.codec_channels * c.codec_vocab * 4},
        {"pred", e->pred, (size_t)c.rows / 2 * c.codec_channels * 8},
        {"all_mask", e->all_mask, (size_t)c.layers * c.rows * E * 4},
        {"all_topk", e->all_topk, (size_t)c.layers * c.rows * 8},
        {"router_logits", e->r_logits, (size_t)c.rows * E * 2},
        {"k_cache", e->k_cache, (size_t)c.layers * c.rows * c.kv_heads * c.Lmax * c.head_dim * 2},
        {"v_cache", e->v_cache, (size_t)c.layers * c.rows * c.kv_heads * c.Lmax * c.head_dim * 2},
        {"counts", e->counts, (size_t)c.n_real * 4},
        {"xg", e->xg, (size_t)(c.ep_size > 1 ? c.ep_size : 0) * c.rows * c.hidden * 2},
        {"ep_words", e->ep_words, 8},
    };
    for (const Item& it : items)
        if (!strcmp(it.n, name)) {
            if (bytes) *bytes = it.b;
            return it.p;
        }
    return nullptr;
}
