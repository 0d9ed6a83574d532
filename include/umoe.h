/*
 * umoe.h -- C-ABI of libumoe_hip.so, the MI355X (gfx950) implementation of the
 * UniMoE-Audio hot path: DCMoE transformer forward + DAC-token decode loop.
 *
 * The reference (foggy-frost-forest/UniMoE-Audio) is 100% Python and has no FFI; this ABI
 * is what the repo's Python host modules (the .py files of unimoe_audio_amd, which mirror the reference
 * module API) bind with ctypes.  Each entry point cites the reference code it replaces
 * (paths relative to the reference root).  See INTEGRATION.md for the reference-side stub.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (message: umoe_last_error()).
 *   - pointers are DEVICE pointers unless named host_*; nothing is allocated or freed
 *     behind the caller's back except inside umoe_engine_* (which owns its workspace).
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued, never synchronised.
 *   - bf16 tensors are uint16_t bit patterns; "T" means bf16 when `*_bf16` is 1 else fp32.
 *   - row-major everywhere; [S] tokens, [D] hidden, [E] = n_dyn + n_fix router columns.
 */
#ifndef UMOE_H
#define UMOE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* umoe_stream_t;

const char* umoe_last_error(void);
int umoe_abi_version(void);
/* sizeof of an argument struct of this header AS THE LIBRARY WAS BUILT ("umoe_router_args", "umoe_gemm_args", ...; 0 = unknown name).
 * Every struct here is passed by pointer and read whole, and they grow at the end between versions: a binding checks its mirrors
 * against this at load time (unimoe_audio_amd/_lib.py does) instead of letting a stale build read past a shorter struct. */
size_t umoe_struct_size(const char* name);

/* ------------------------------------------------------------------ weight layout
 * nn.Linear weights W[N][K] (bf16) are re-laid once at load time into the MFMA operand
 * order used by every weight-streaming kernel ("WP16"): 16-row x 32-col fragments of
 * v_mfma_f32_16x16x32_bf16, stored so that one wave-instruction reads 1 KiB contiguous.
 * K is split into 4 equal quarters, lane-group h = lane>>4 owns quarter h:
 *   packed[((nb*KB + i)*64 + lane)*8 + j] = W[nb*16 + (lane&15)][(lane>>4)*(K/4) + i*8 + j]
 * with KB = K/32, N padded up to a multiple of 16 with zero rows.  Requires K % 32 == 0.
 */
size_t umoe_packed_elems(int N, int K);
int umoe_pack_weight(const uint16_t* W, int N, int K, uint16_t* packed, umoe_stream_t stream);
/* gate_proj / up_proj of one SwiGLU expert interleaved per 16-row block (block 2i = gate i,
 * block 2i+1 = up i) so one wave produces silu(g)*u without a round trip.  N_packed = 2*I. */
int umoe_pack_gate_up(const uint16_t* Wg, const uint16_t* Wu, int I, int K, uint16_t* packed, umoe_stream_t stream);

/* ------------------------------------------------------------------ router
 * Replaces utils/UniMoE_Audio_core.py:246-291,331-339 (gate GEMM, Top-P count :157-167,
 * sparse mixer :94-154 + driver loop :262-282, renormalise / padding / shared-on :284-291,
 * global weights :178-193) and the `router_weight * expert_mask` of :447.
 * Integer outputs are bit-exact w.r.t. oracle/router_oracle.c given identical logits.
 */
typedef struct {
    /* inputs: either (x, gate_w) or logits_in */
    const uint16_t* x;        /* [S][D] bf16 hidden states (post-norm unless norm_w given), or NULL */
    const uint16_t* gate_w;   /* [E][D] bf16, reference `gate.weight` */
    const uint16_t* norm_w;   /* optional [D]: fuse RMSNorm (model.py:240) in front of the gate */
    uint16_t* h_out;          /* optional [S][D]: normalised hidden states written for the experts */
    const void* logits_in;    /* optional [S][E] T: skip the gate GEMM (parity tests) */
    const uint8_t* attn_mask; /* optional [S] 0/1 padding mask (core.py:286-288) */
    int S, D;
    int n_dyn, n_real, n_fix; /* 9, 8, 2 */
    int logits_bf16;          /* 1: eval (bf16 gate, core.py:251); 0: fp32 gate (core.py:249) */
    float top_p;              /* 0 => fixed_top_k for every token (core.py:254-257) */
    int fixed_top_k;
    double jitter_eps;        /* router_jitter_noise */
    float rms_eps;
    /* outputs (any may be NULL except expert_mask / moe_w / slot tables) */
    void* logits_out;         /* [S][E] T */
    int64_t* top_k;           /* [S] */
    int32_t* sel;             /* [S][n_dyn] expert picked at round j, -1 beyond k */
    int32_t* expert_mask;     /* [S][E] */
    float* routing_w;         /* [S][n_dyn] T-rounded values held in fp32 */
    float* global_w;          /* [S][E]     T-rounded values held in fp32 */
    float* moe_w;             /* [S][n_real] = global_w * mask */
    int norm_only;            /* 1: only h_out = RMSNorm(x) with the router's own arithmetic (x, norm_w, h_out; S <= 256, D 2048 / 4096):
                               * the launch the decode engine puts in front of the expert GEMM when the router itself rides inside
                               * that GEMM's launch (umoe_gemm_args.fused_router) */
    /* training branch of the mixer (core.py:111-137; `training and not ignore_differentiable_router`), taken when `gumbel` is set:
     * round j selects arg-max(masked_gates + gumbel[s][j][:]) instead of the arg-max, and its weight is the softmax multiplier times
     * mask_for_one = 1 if (selected == arg-max of the softmaxed gates or rand_u[s][j] > 0.75) else 0.3333.  The noise is an INPUT
     * (the reference draws gumbel_rsample / torch.rand_like per call): the host supplies it, tests inject the reference's draws. */
    const float* gumbel;      /* optional [S][n_dyn][n_dyn] fp32 Gumbel(0,1) noise, round-major per token */
    const float* rand_u;      /* [S][n_dyn] fp32 uniform [0,1) draws, one per (token, round); required with gumbel */
    float* round_factor;      /* optional out [S][n_dyn]: mask_for_one of round j as 1 / 0.3333 (T-rounded), 0 beyond k; kept for
                               * umoe_router_bwd_ex (AudioMoERoutingFunction.backward, core.py:64-91) */
    /* training-time input jitter with the fp32 gate (core.py:240-249: `hidden_states.float()` times U(1-eps, 1+eps) noise, then the
     * fp32 gate GEMM): the gate sees float(x[s][d]) * x_noise[s][d] with NO rounding to bf16 in between.  The noise is an INPUT like
     * the mixer's.  Needs x / gate_w, excludes norm_w / h_out / logits_in. */
    const float* x_noise;     /* optional [S][D] fp32 */
} umoe_router_args;
int umoe_router_fwd(const umoe_router_args* a, umoe_stream_t stream);

/* aux load-balancing loss (core.py:361-389): n_dyn * sum_e mean_s(mask) * mean_s(softmax(masked logits)); token_weight
 * (optional [S]) = aux_balance_weight expanded per token (core.py:380-385).  out = one fp32 scalar. */
int umoe_aux_loss_fwd(const void* logits, int logits_bf16, const int32_t* expert_mask, const float* token_weight, int S, int E,
                      int n_dyn, float* out, umoe_stream_t stream);
/* the same result from two launches (64 workgroups over contiguous token ranges, fixed-order partial sums in `ws`, then one small
 * workgroup): for many tokens (training).  ws: umoe_aux_loss_workspace_floats() floats, caller-owned.  Sums are re-associated, so
 * the value can differ from umoe_aux_loss_fwd in the last bits. */
int umoe_aux_loss_fwd_ws(const void* logits, int logits_bf16, const int32_t* expert_mask, const float* token_weight, int S, int E,
                         int n_dyn, float* out, float* ws, umoe_stream_t stream);
size_t umoe_aux_loss_workspace_floats(void);

/* Token drop (core.py:302-329; `capacity` = _audio_expert_capacity, core.py:170-175, computed by the host in the reference's
 * float32 arithmetic).  policy 0 "probs": per dynamic column keep the `capacity` selected tokens with the largest logits (the
 * reference's topk(dim=0) over the column, :305-314; shared columns untouched); policy 1 "position": keep the first `capacity`
 * selected tokens of EVERY column in token order (:321-323; the reference's cumsum runs over the shared columns too).  Dropped
 * entries leave routing_w, which is renormalised (:328-329); global_w / moe_w are recomputed over the kept columns (:178-193) with
 * the router's arithmetic.  Ties at the capacity boundary keep the lowest token indices (torch.topk leaves them unspecified).
 * Integer outputs are bit-exact w.r.t. the reference wherever the boundary is not tied. */
int umoe_token_drop(const void* logits, int logits_bf16, const int32_t* expert_mask_in, const float* routing_w_in, int S, int n_dyn,
                    int n_real, int n_fix, int capacity, int policy, int32_t* expert_mask_out, float* routing_w_out, float* global_w,
                    float* moe_w, umoe_stream_t stream);

/* Ragged dispatch tables from the 0/1 mask: the build's replacement for the dense
 * compress_matrix / decompress_matrix pair (utils/UniMoE_Audio_utils.py:436-523) and the
 * capacity MAX of core.py:455-457.  Wavefront ballot + prefix sums, token order preserved.
 *   counts[n_real], offsets[n_real+1], slot_token[S*n_real], slot_of[S][n_real] (-1 = unrouted) */
int umoe_dispatch_build(const int32_t* expert_mask, int S, int ld_mask, int n_real, int32_t* counts,
                        int32_t* offsets, int32_t* slot_token, int32_t* slot_of, umoe_stream_t stream);
/* same tables with every expert's slot range starting on a multiple of `align` (power of two): offsets[e] % align == 0,
 * offsets[n_real] = padded total, padding slots carry token 0 and are not counted.  slot_token needs
 * S*n_real + n_real*(align-1) entries.  Used by training: the weight-gradient GEMMs contract over slot columns of
 * transposed buffers in 16-byte chunks. */
int umoe_dispatch_build_aligned(const int32_t* expert_mask, int S, int ld_mask, int n_real, int align, int32_t* counts,
                                int32_t* offsets, int32_t* slot_token, int32_t* slot_of, umoe_stream_t stream);

/* router + dispatch tables in one call: a single fused launch when S <= 16 (decode), two launches otherwise */
int umoe_router_dispatch_fwd(const umoe_router_args* a, int32_t* counts, int32_t* offsets, int32_t* slot_token,
                             int32_t* slot_of, umoe_stream_t stream);

/* permute: out[slot] = x[slot_token[slot]]  (the gather half of compress_matrix);
 * bwd of unpermute.  rows = offsets[n_real] read on device. */
int umoe_permute_fwd(const uint16_t* x, int D, const int32_t* slot_token, const int32_t* total_slots, int max_slots,
                     uint16_t* out, umoe_stream_t stream);

/* ------------------------------------------------------------------ grouped weight-streaming GEMM
 * One launch covers any number of groups (routed experts with ragged row lists, shared
 * experts with all rows, plain dense layers with one group).
 */
typedef struct {
    const uint16_t* w;        /* WP16-packed weights of this group */
    const float* bias;        /* optional [N] fp32 */
    const int32_t* rows;      /* optional gather list (token index per row); NULL = identity */
    const int32_t* row_off;   /* optional device scalar added to the row number (offsets[e]) */
    const int32_t* count;     /* optional device scalar: number of rows; NULL => static_count */
    int static_count;
    int a_row_base;           /* identity mode: A row = a_row_base + r */
    int out_row_base;         /* output row = out_row_base + (row_off ? *row_off : 0) + r */
    int n_blocks;             /* N/16 (2*I/16 for gate/up pairs) */
    int k;                    /* K */
    int a_col_off;            /* element offset added to every A row of this group (EP receive buffers) */
    int reserved;
} umoe_group_t;

enum { UMOE_PRO_PLAIN = 0, UMOE_PRO_RMSNORM = 1 };
enum {
    UMOE_EPI_BF16 = 0,      /* y = bf16(acc + bias) */
    UMOE_EPI_BF16_RESID = 1, /* y = bf16(resid + bf16(acc + bias)) */
    UMOE_EPI_SWIGLU = 2,     /* pairs: y = bf16(bf16(silu(bf16 g)) * bf16 u)   core.py:31,49 */
    UMOE_EPI_F32 = 3,        /* y = float(bf16(acc))   (codec_head(...).float(), model.py:982) */
    UMOE_EPI_F32_RAW = 4     /* y = acc (fp32, unrounded) */
};

typedef struct {
    const umoe_group_t* groups; /* device array */
    int num_groups;
    int max_rows;             /* upper bound of rows per group (grid sizing; no host sync) */
    int max_n_blocks;         /* max over groups */
    int max_k;
    const uint16_t* a;        /* [*, lda] bf16 activations */
    int lda;
    const uint16_t* norm_w;   /* UMOE_PRO_RMSNORM: [K] */
    float rms_eps;
    const uint16_t* resid;    /* UMOE_EPI_BF16_RESID: [*, ldo] */
    void* out;                /* bf16 or fp32 [*, ldo] */
    int ldo;
    int n_valid;              /* columns >= n_valid are not stored (N not multiple of 16) */
    int prologue, epilogue;
    int nt;                   /* n-blocks per workgroup: 0 = auto, else 1/2/4/5/6/8, SwiGLU also 14 (tuning knob) */
    int waves;                /* waves per workgroup: 0 = auto, 4, 8 (8 only with nt = 8; tuning knob) */
    int ksplit;               /* >1: K split over workgroups; needs UMOE_EPI_F32_RAW: slab s at out + s*part_stride */
    long part_stride;         /* elements between fp32 partial slabs */
    const umoe_group_t* groups_host; /* optional HOST copy of `groups`: with num_groups <= UMOE_GROUPS_INLINE the
                               * descriptors travel in the kernel arguments (one dependent HBM round trip less per launch) */
    int cache_policy;         /* reserved (0): the weight stream is always non-temporal -- a default-policy variant and an
                               * Infinity Cache warm-up were measured and bought nothing */
    const umoe_router_args* fused_router; /* optional HOST pointer (SwiGLU, nt = 14, <= 16 rows, n_dyn 9 / n_fix 2, D 2048 / 4096, S <= 16):
                               * the router of these tokens runs INSIDE this launch as S extra workgroups (its outputs are complete when
                               * the launch is); the GEMM itself must not depend on them (dense-expert decode: h_out NULL, see norm_only) */
    const void* rider_pub;    /* decode engine only (NULL otherwise), with fused_router: HOST pointer to the hand-off descriptor -- the riders
                               * also produce the normalised rows `a` (fused_router->h_out == a) and HAND them to the GEMM workgroups of
                               * this same launch, which stream their first weight chunk while they wait: no RMSNorm launch in front */
} umoe_gemm_args;
#define UMOE_GROUPS_INLINE 12

/* Warm the Infinity Cache (256 MiB, memory side) with `bytes` at `p`: plain loads, nothing stored.  `wgs` workgroups of 256
 * threads, 8 x 16 B per thread in flight.  Used to pull the next weight-streaming kernel's first bytes out of HBM while
 * a latency-bound kernel leaves the memory idle. */
int umoe_prefetch(const void* p, size_t bytes, int wgs, umoe_stream_t stream);
int umoe_grouped_gemm(const umoe_gemm_args* a, umoe_stream_t stream);

/* ------------------------------------------------------------------ tiled MFMA GEMM (prefill / training shapes)
 * Y[rows(g), N(g)] = epilogue( A[rows(g), K] * W_g^T ) with W_g ROW-MAJOR [N][K] bf16 (the reference's nn.Linear layout,
 * core.py:21-28,39-46; no packing: training updates these tensors every step).  Compute-bound counterpart of
 * umoe_grouped_gemm: 128x128x64 tiles through LDS, v_mfma_f32_16x16x32_bf16, fp32 accumulation, the same ragged-group
 * conventions (gather list / device-side count and offset) and the same rounding points in the epilogues.
 * UMOE_EPI_SWIGLU takes gate rows from `w` and up rows from `w2` (N = intermediate size). */
typedef struct {
    const uint16_t* w;        /* [N][ldw] row-major */
    const uint16_t* w2;       /* SwiGLU only: up_proj [N][ldw] */
    const float* bias;        /* optional [N] fp32 */
    const int32_t* rows;      /* optional gather list; NULL = identity */
    const int32_t* row_off;   /* optional device scalar added to the row number */
    const int32_t* count;     /* optional device scalar: rows of this group; NULL => static_count */
    int static_count;
    int a_row_base, out_row_base;
    int n, k, ldw;            /* output features, contraction length (k % 8 == 0), elements between weight rows */
    int a_col_off;
    const int32_t* k_off;     /* optional device scalars: contract over columns [*k_off, *k_off + roundup8(*k_count)) of BOTH */
    const int32_t* k_count;   /* operands instead of [0, k) -- weight gradients over one expert's (8-aligned) slot range */
    int out_col_off;          /* element offset added to every output row of this group (attention heads side by side) */
    int k_compact_a;          /* with k_off / k_count: > 0 = the activation operand is laid out in per-group COMPACT blocks (what
                               * umoe_transpose_slots_compact writes): this group's block starts at element *k_off * k_compact_a (the
                               * operand's total row count) and its rows are roundup8(*k_count) apart, columns [0, roundup8(*k_count)) */
    int k_compact_w;          /* the same for the weight operand (total rows of `w`) */
    int w_kmajor;             /* != 0: `w` is row-major [K][ldw] with the CONTRACTION index as its row and n columns (Y = A W): the input
                               * gradient dX = dY W on an nn.Linear weight as stored, no transposed copy.  UMOE_EPI_BF16 only, every group of
                               * the launch alike, n % 8 == 0, no bias / K window; the launch runs whatever its size on 256 x 256 tiles */
    int k_w1;                 /* w_kmajor with w2 != NULL: rows [0, k_w1) of the contraction come from w, rows [k_w1, k) from w2 (same ldw, */
                              /* k_w1 % 32 == 0): (dG | dU) against Wg then Wu */
} umoe_tgroup_t;

typedef struct {
    const umoe_tgroup_t* groups;  /* HOST array (descriptors travel in the kernel arguments), num_groups <= 12 */
    int num_groups;
    int max_rows;             /* upper bound of rows in any group (grid sizing) */
    const uint16_t* a;        /* [*, lda] bf16 */
    int lda;
    const uint16_t* resid;    /* UMOE_EPI_BF16_RESID: [*, ldo] */
    void* out;                /* bf16 (or fp32 for UMOE_EPI_F32 / _RAW) [*, ldo] */
    int ldo;
    int epilogue;             /* UMOE_EPI_* */
    uint16_t* aux_out;        /* UMOE_EPI_SWIGLU with aux_out != NULL (training): also store the bf16 pre-activations, */
    int ld_aux;               /* gate at aux_out[row][col], up at aux_out[row][n + col] (what autograd would save) */
} umoe_tgemm_args;
int umoe_tiled_gemm(const umoe_tgemm_args* a, umoe_stream_t stream);

/* Weight-gradient form of the tiled GEMM: both operands are row-major with the CONTRACTION index as the row (tokens / expert slots), as
 * autograd holds activations and their gradients -- dW = dY^T X without transposed copies (torch.nn.functional.linear backward,
 * core.py:21-49 through the expert MLPs and every nn.Linear of model.py:672-871):
 *   out_g[m][n] = sum over k in the group's window of  P[k][p_col_off + m] * Q[k][q_col_off + n]      (bf16 in, fp32 sum, bf16 out)
 * The window is static (k_off, k) or read on the device (k_off_dev / k_count_dev: an expert's slot range from
 * umoe_dispatch_build_aligned).  k_split > 1 (static windows, groups that tile one dense [rows][ldo] output with n == ldo): the window is
 * cut into k_split parts whose fp32 partial outputs (ws, part_stride floats apart, umoe_tiled_gemm_tn_workspace_bytes) are summed in
 * fixed order -- for products with too few 256 x 256 output tiles to fill the chip. */
typedef struct {
    const uint16_t* p; int ldp;   /* optional per-group operands (NULL: the launch's) */
    const uint16_t* q; int ldq;
    void* out;                    /* optional per-group output base (NULL: the launch's) */
    int p_col_off, q_col_off;     /* column of P that is output row 0 / column of Q that is output column 0 (multiples of 8) */
    int m, n;                     /* output rows / columns (multiples of 8) */
    int k_off, k;                 /* static window [k_off, k_off + k) */
    const int32_t* k_off_dev;     /* device scalars, both or neither */
    const int32_t* k_count_dev;
    int out_row_base, out_col_off;
} umoe_tn_group_t;
typedef struct {
    const umoe_tn_group_t* groups;   /* HOST array, num_groups <= 24 */
    int num_groups;
    const uint16_t* p; int ldp;      /* [rows][ldp] bf16: its columns become output ROWS */
    const uint16_t* q; int ldq;      /* [rows][ldq] bf16: its columns become output COLUMNS */
    void* out; int ldo;              /* bf16 [*, ldo] */
    int k_split;                     /* 0 / 1: none; > 1: that many parts; < 0: chosen by the library (1..8, from the tile count and K) */
    void* ws; long part_stride;      /* k_split > 1 or < 0: fp32 workspace (umoe_tiled_gemm_tn_workspace_bytes), part_stride = rows * ldo of the output */
} umoe_tgemm_tn_args;
size_t umoe_tiled_gemm_tn_workspace_bytes(const umoe_tgemm_tn_args* a);
int umoe_tiled_gemm_tn_split(const umoe_tgemm_tn_args* a);      /* the K split the call would use (k_split < 0: the library's choice) */
int umoe_tiled_gemm_tn(const umoe_tgemm_tn_args* a, umoe_stream_t stream);

/* Named wrappers required by the scope table; thin calls of umoe_grouped_gemm.
 * umoe_grouped_swiglu_fwd: routed experts, core.py:406-416 + :34-49 on ragged rows.
 * umoe_shared_swiglu_fwd : shared experts, core.py:344-351 + :16-31. */
int umoe_grouped_swiglu_fwd(const umoe_group_t* gateup_groups, const umoe_group_t* down_groups, int num_groups,
                            int max_rows, const uint16_t* x, int D, int I, uint16_t* h_ws, uint16_t* y_slots,
                            umoe_stream_t stream);

int umoe_shared_swiglu_fwd(const umoe_group_t* gateup_groups, const umoe_group_t* down_groups, int n_fix, int S,
                           const uint16_t* x, int D, int I, uint16_t* h_ws, uint16_t* y_shared, umoe_stream_t stream);

/* combine: out[s] = resid[s] + ( sum_e moe_w[s][e] * y[slot_of[s][e]]  (+ shared_i[s] * global_w[s][n_dyn+i]) )
 * with the reference's rounding points (einsum core.py:488, adds :342,:351, residual model.py:242). */
typedef struct {
    const uint16_t* y_slots;  /* [slots][D] routed expert outputs (bf16), or NULL when y_parts is used */
    const int32_t* slot_of;   /* [S][n_real] */
    const float* moe_w;       /* [S][n_real] */
    const uint16_t* y_shared; /* [n_fix][S][D] or NULL */
    const float* global_w;    /* [S][E] */
    const uint16_t* resid;    /* [S][D] or NULL */
    uint16_t* out;            /* [S][D] */
    int S, D, n_real, n_dyn, n_fix;
    const float* y_parts;     /* optional fp32 partial slabs [n_parts][rows][D] of a K-split down GEMM: y = bf16(sum) */
    int n_parts;
    long part_stride;
    int shared_row0;          /* >= 0: the shared experts' rows live in the slabs too, at shared_row0 + i*S + s */
    const uint16_t* norm_w;   /* optional [D]: also write norm_out = RMSNorm(out) * norm_w (next layer's input norm) */
    uint16_t* norm_out;       /* [S][D] */
    float rms_eps;
    const int32_t* expert_mask; /* dense-expert layout (slot_of == NULL): expert e's output row of token s is e*dense_rows + s, */
    int mask_ld, dense_rows;    /* used iff expert_mask[s*mask_ld + e] != 0 (every expert computed all rows; decode, S <= 16) */
    const void* ep_xfer;        /* decode engine only (NULL otherwise): HOST pointer to the exchange descriptor of an expert-parallel
                                 * step -- y_slots is then this rank's return slab and the kernel waits for the peers' rows itself */
} umoe_combine_args;
int umoe_unpermute_combine_fwd(const umoe_combine_args* a, umoe_stream_t stream);

/* ------------------------------------------------------------------ expert-parallel exchange (RCCL over xGMI)
 * The all-to-all of DeepSpeed's _AllToAll in the reference (core.py:467,480; utils.py:332-335).  `comm` is an opaque
 * ncclComm_t; librccl is resolved at first use.  Every rank exchanges slabs of equal size (fixed per-destination capacity
 * of unimoe_audio_amd/ep.py): send + p*bytes goes to peer p, peer p's slab lands at recv + p*bytes; enqueued on `stream`.
 * umoe_ep_unique_id (rank 0, 128 bytes to broadcast out of band) + umoe_ep_comm_create build a communicator where the
 * caller has none to pass in. */
int umoe_ep_unique_id(void* out128);
int umoe_ep_comm_create(const void* uid128, int rank, int nranks, void** comm_out);
int umoe_ep_comm_destroy(void* comm);
int umoe_ep_all_to_all(void* comm, const void* send, void* recv, size_t bytes_per_peer, int nranks, umoe_stream_t stream);

/* Peer exchange of the expert-parallel DECODE engine (umoe_engine_cfg.ep_size > 1): xGMI peer stores inside the captured step
 * graph instead of a collective call.  Every rank owns one exchange REGION (uncached device memory: flags + the dispatch slab
 * [ep][rows][D] + the return slab [n_real][rows][D]); ranks map each other's regions through HIP IPC handles once, at connect
 * time.  Per layer: rank r pushes its normalised rows into tile r of every peer's dispatch slab (the DENSE form of the
 * reference's first all-to-all, core.py:467: at <= 16 rows per rank every expert is hit anyway, so every rank receives every
 * row and the routing result is only needed by the combine), computes its local experts on all ep*rows rows, and pushes the
 * outputs for rank t's rows into the return slab of rank t (second all-to-all, core.py:480).  Payload and flags are
 * system-scope write-through stores (sc0 sc1), drained per wave before ONE lane publishes the epoch; receivers poll with
 * system-scope loads (bounded spin: a timeout sets the engine's error word, see umoe_engine_ep_error) and copy the slab into
 * ordinary device memory for the GEMMs.  Epoch = decode step * layers + layer + 1, counted on the device: graph replays need
 * no host value.  All ranks must run the same sequence of decode steps. */
#define UMOE_MAX_EP 8
int umoe_ep_ipc_export(const void* dev_ptr, void* handle64_out);      /* hipIpcGetMemHandle: 64 opaque bytes */
int umoe_ep_ipc_open(const void* handle64, void** dev_ptr_out);       /* hipIpcOpenMemHandle (lazy peer access) */
int umoe_ep_ipc_close(void* dev_ptr);

/* ------------------------------------------------------------------ backward pieces (training, BASELINE config 3)
 * The contractions run on umoe_tiled_gemm (operands K-contiguous): dX = dY * W uses a transposed weight copy, dW = dY^T X
 * contracts over the slot columns of transposed, 8-aligned, zero-padded buffers built by umoe_transpose_slots
 * (umoe_dispatch_build_aligned + umoe_tgroup_t.k_off / k_count).  Everything below is per-token / elementwise. */

/* dst[c][off_g + r] = src[row(off_g + r)][c], r < counts[g]; zeros up to the next multiple of 8.  row(s) = rows ? rows[s] : s.
 * counts/offsets NULL: one group of max_rows rows at offset 0 (plain transpose).  Replaces autograd's implicit
 * transposes in Linear.backward (core.py:21-49 via torch.nn.functional.linear). */
int umoe_transpose_slots(const uint16_t* src, int ld_src, int C, const int32_t* rows, const int32_t* counts,
                         const int32_t* offsets, int n_groups, int max_rows, uint16_t* dst, int ld_dst, umoe_stream_t stream);
/* the same transpose into per-group COMPACT blocks: group g's block starts at dst + offsets[g] * C and holds C rows of roundup8(counts[g])
 * elements (zero padded) -- a weight-gradient product then walks rows 5.6 KB apart instead of 50 KB (umoe_tgroup_t.k_compact_*).
 * counts / offsets required; dst needs (sum of roundup8 counts) * C elements. */
int umoe_transpose_slots_compact(const uint16_t* src, int ld_src, int C, const int32_t* rows, const int32_t* counts,
                                 const int32_t* offsets, int n_groups, int max_rows, uint16_t* dst, umoe_stream_t stream);

/* SwiGLU backward, core.py:31,49: gu [rows][2I] = (gate | up) pre-activations saved by the forward
 * (umoe_tgemm_args.aux_out), dh [rows][I] -> dgu [rows][2I] = (dgate | dup).  total_rows: device scalar or NULL. */
/* y[i] = bf16(float(x[i]) * noise[i]): the training-time input jitter on the gate's copy of the rows (core.py:240-244: the reference
 * multiplies a float copy and casts back) in one pass instead of three elementwise launches.  n % 8 == 0. */
int umoe_mul_noise(const uint16_t* x, const float* noise, long n, uint16_t* y, umoe_stream_t stream);
int umoe_swiglu_bwd(const uint16_t* dh, int ld_dh, const uint16_t* gu, int ld_gu, int I, const int32_t* total_rows,
                    int max_rows, uint16_t* dgu, int ld_dgu, umoe_stream_t stream);

/* backward of umoe_unpermute_combine_fwd (einsum core.py:488, shared adds :349-351): per selected (token, expert)
 * dy[slot] = bf16(w * dout[s]) and d_moe_w[s][e] = <dout[s], y[slot]>; same for the shared experts with global_w. */
int umoe_unpermute_combine_bwd(const uint16_t* dout, const umoe_combine_args* a, uint16_t* dy_slots, uint16_t* dy_shared,
                               float* d_moe_w, float* d_gw_shared, umoe_stream_t stream);

/* backward of the dispatch gather (compress_matrix, utils.py:436-485): dx[s] = sum of the slot rows of token s
 * (+ the shared experts' input gradients [n_fix][S][D], + an optional extra [S][D] term), fp32 accumulate, one rounding. */
int umoe_permute_bwd(const uint16_t* dxe, const int32_t* slot_of, int n_real, const uint16_t* dx_shared, int n_fix, int S, int D,
                     const uint16_t* extra, uint16_t* dx, umoe_stream_t stream);

/* router backward, shipped configuration (ignore_differentiable_router, core.py:272): d(moe_w), d(shared global weights)
 * -> d_logits [S][E] fp32 through the per-round softmax multipliers (core.py:115-119), the renormalisation (:284) and the
 * global softmax (:178-193).  sel / top_k / expert_mask are the forward's integer outputs; d_logits_in is added (aux loss). */
int umoe_router_bwd(const void* logits, int logits_bf16, const int32_t* sel, const int64_t* top_k, const int32_t* expert_mask,
                    const float* d_moe_w, const float* d_gw_shared, const float* d_logits_in, int S, int n_dyn, int n_real,
                    int n_fix, double jitter_eps, float* d_logits, umoe_stream_t stream);
/* general form: token_drop as in umoe_router_bwd_drop; round_factor (optional [S][n_dyn], umoe_router_args.round_factor) = the
 * differentiable-router training branch: the forward weights carry mask_for_one, the gradient is AudioMoERoutingFunction.backward
 * (core.py:64-91: grad * multiplier * (onehot(selected) - masked_gates), mask_for_one not differentiated) */
int umoe_router_bwd_ex(const void* logits, int logits_bf16, const int32_t* sel, const int64_t* top_k, const int32_t* expert_mask,
                       const float* d_moe_w, const float* d_gw_shared, const float* d_logits_in, int S, int n_dyn, int n_real,
                       int n_fix, double jitter_eps, int token_drop, const float* round_factor, float* d_logits, umoe_stream_t stream);
/* the same with the token-drop branch in the graph (core.py:328-329): `expert_mask` is then the mask AFTER the drop and the routing
 * weights pass through r2 = (r * mask) / (sum(r * mask) + 1e-6) before the global weights */
int umoe_router_bwd_drop(const void* logits, int logits_bf16, const int32_t* sel, const int64_t* top_k, const int32_t* expert_mask,
                         const float* d_moe_w, const float* d_gw_shared, const float* d_logits_in, int S, int n_dyn, int n_real,
                         int n_fix, double jitter_eps, float* d_logits, umoe_stream_t stream);

/* Backward of the expert MLPs down(silu(gate x) * up x), core.py:16-49,406-416, over up to 12 groups per call:
 *   dH = dY Wd ; (dG | dU) = SwiGLU'(G, U, dH) ; dX_slots = dG Wg + dU Wu ; dWd = dY^T H ; dWg = dG^T X ; dWu = dU^T X.
 * umoe_grouped_swiglu_bwd: routed experts, ragged rows (counts/offsets from umoe_dispatch_build_aligned with align 8,
 * slot_token = gather list of x).  umoe_shared_swiglu_bwd: shared experts, counts/offsets NULL, group g owns slot rows
 * [row_base + g*max_rows, +max_rows) (both multiples of 8) and reads x by identity.  All weights row-major (nn.Linear). */
typedef struct {
    int num_groups;
    const uint16_t* const* w_gate;   /* host arrays [num_groups] of device pointers: [I][D], [I][D], [D][I] */
    const uint16_t* const* w_up;
    const uint16_t* const* w_down;
    int D, I;
    const int32_t* counts;           /* device [num_groups] or NULL */
    const int32_t* offsets;          /* device [num_groups + 1] (8-aligned, last = padded total) or NULL */
    const int32_t* slot_token;       /* device gather list (slot -> token) or NULL */
    int max_rows;                    /* tokens S: upper bound of rows per group */
    int slot_rows;                   /* rows of the slot buffers (h, gu, dy, dx_slots) */
    int row_base;                    /* static groups: first slot row of group 0 */
    const uint16_t* x;  int ldx;     /* [S][D] block input */
    const uint16_t* h;  int ldh;     /* [slot_rows][I] forward silu(g)*u */
    const uint16_t* gu; int ldgu;    /* [slot_rows][2I] forward pre-activations (umoe_tgemm_args.aux_out) */
    const uint16_t* dy; int lddy;    /* [slot_rows][D] gradient of the expert outputs */
    uint16_t* dx_slots; int lddx;    /* out [slot_rows][D]: gradient of the gathered inputs (-> umoe_permute_bwd) */
    uint16_t* const* dw_gate;        /* host arrays [num_groups] of device pointers, outputs [I][D], [I][D], [D][I] */
    uint16_t* const* dw_up;
    uint16_t* const* dw_down;
    void* ws; size_t ws_bytes;       /* umoe_swiglu_bwd_workspace_bytes() */
    const uint16_t* w_down_T;        /* optional, both or neither, read ONLY when I % 32 != 0 (otherwise the input gradients read the weights */
    const uint16_t* w_gateup_T;      /* as stored, umoe_tgroup_t.w_kmajor): transposed weight copies of the caller, group g's Wd^T [I][roundup8(D)]
                                      * at w_down_T + g*I*roundup8(D), (Wg^T | Wu^T) [D][2I] at w_gateup_T + g*D*2I -- what the composite
                                      * builds per call itself otherwise */
} umoe_swiglu_bwd_args;
size_t umoe_swiglu_bwd_workspace_bytes(const umoe_swiglu_bwd_args* a);
int umoe_grouped_swiglu_bwd(const umoe_swiglu_bwd_args* a, umoe_stream_t stream);
int umoe_shared_swiglu_bwd(const umoe_swiglu_bwd_args* a, umoe_stream_t stream);

/* backward of umoe_aux_loss_fwd (core.py:361-389): d_logits [S][E] fp32 = *d_aux * d aux / d logits (zero on the columns the
 * mask removed and on the shared columns); ws: 17 floats of scratch (per-expert token fractions, weight sum). */
int umoe_aux_loss_bwd(const void* logits, int logits_bf16, const int32_t* expert_mask, const float* token_weight, int S, int E,
                      int n_dyn, const float* d_aux, float* d_logits, float* ws, umoe_stream_t stream);

/* Qwen2RMSNorm backward (model.py:206-207): h [S][D] the normalised tensor's input, dy the gradient of w * bf16(h * rs);
 * dh = rs * (dy*w - xh * mean(dy*w*xh)) + dsum (gradient arriving on h through the residual path, optional), dw [D].
 * ws: fp32 scratch of min(S,512)*D floats. */
int umoe_rmsnorm_residual_bwd(const uint16_t* h, const uint16_t* w, const uint16_t* dy, const uint16_t* dsum, float eps, int S,
                              int D, uint16_t* dh, uint16_t* dw, float* ws, size_t ws_floats, umoe_stream_t stream);


/* ------------------------------------------------------------------ norm / rope / attention
 * umoe_rmsnorm_residual_fwd: y = w * bf16((x [+ r]) * rsqrt(mean((x+r)^2) + eps)); also writes x+r.
 * Replaces transformers Qwen2RMSNorm (model.py:206-207,227,240,428) + residual adds (:238,:242). */
int umoe_rmsnorm_residual_fwd(const uint16_t* x, const uint16_t* r, const uint16_t* w, float eps, int S, int D,
                              uint16_t* sum_out, uint16_t* y, umoe_stream_t stream);

/* mRoPE + KV append.  qkv [rows*T][(H+2*KVH)*hd] bf16 (bias already added).  cos/sin tables
 * [max_pos][hd/2] bf16 built by the host exactly as Qwen2_5_VLRotaryEmbedding does; pos3 [3][rows*T]
 * position streams; sections = mrope_section (channel block i uses stream i%3).
 * Writes rotated q to q_out [rows*T][H*hd] and rotated k / v into the cache at kv_pos[row*T+t]:
 * cache layout [rows][KVH][Lmax][hd].  Replaces apply_multimodal_rotary_pos_emb + DynamicCache.update. */
typedef struct {
    const uint16_t* qkv;
    const uint16_t* cos_tab;
    const uint16_t* sin_tab;
    const int32_t* pos3;      /* [3][n_tok] */
    const int32_t* kv_pos;    /* [n_tok] cache slot of each token */
    int n_tok, T;             /* n_tok = rows*T; token i belongs to row i / T */
    int H, KVH, hd;
    int sec0, sec1, sec2;
    int Lmax;
    uint16_t* q_out;
    uint16_t* k_cache;
    uint16_t* v_cache;
} umoe_rope_args;
int umoe_qkv_mrope_kvappend(const umoe_rope_args* a, umoe_stream_t stream);

/* Attention backward, training path (reference: eager_attention_forward of the transformers dependency called at
 * model.py:228-237; fp32 softmax, probabilities cast to bf16).  The contractions run on umoe_tiled_gemm over materialised
 * score tiles; these are the row-wise pieces.  scores [heads*Tp][ld] fp32 raw Q K^T, row h*Tp + t sees keys [kv_start, t]:
 * p_out = bf16(softmax(scale * scores)), zero elsewhere (including the padding columns up to ld_p).
 * backward: ds = bf16(scale * p o (dp - sum_j dp_j p_j)). */
int umoe_attn_softmax_fwd(const float* scores, int ld, int heads, int T, int Tp, int kv_start, float scale, uint16_t* p_out,
                          int ld_p, umoe_stream_t stream);
int umoe_attn_softmax_bwd(const uint16_t* p, const uint16_t* dp, int ld, int heads, int T, int Tp, float scale, uint16_t* ds,
                          umoe_stream_t stream);
/* backward of umoe_attn_prefill_fwd over full sequences (nq == T queries per row, one per key position; keys
 * [kv_start[row], t] visible to query t): dq [rows*T][H*hd], dk / dv in cache layout (positions [0, T) written).
 * "Unfused" first version: scores are materialised per (row, kv head) group in the workspace. */
typedef struct {
    const uint16_t* q;             /* [rows*T][H*hd] rotated queries (umoe_qkv_mrope_kvappend q_out) */
    const uint16_t* k_cache;       /* [rows][KVH][Lmax][hd] */
    const uint16_t* v_cache;
    const int32_t* kv_start_host;  /* HOST [rows] */
    const uint16_t* d_out;         /* [rows*T][H*hd] gradient of the attention output */
    int rows, T, H, KVH, hd, Lmax;
    float scale;
    uint16_t* dq;
    uint16_t* dk_cache;
    uint16_t* dv_cache;
    void* ws; size_t ws_bytes;     /* umoe_attn_prefill_bwd_workspace_bytes() */
    const uint16_t* out;           /* optional: the forward's output [rows*T][H*hd] and its log-sum-exp [rows*T][H] (fp32,        */
    const float* lse;              /* umoe_attn_args.lse_out).  Both given (hd 128, <= 8 heads per kv head): fused flash-style     */
                                   /* backward, no score matrices in memory; otherwise the unfused composite.                    */
} umoe_attn_bwd_args;
size_t umoe_attn_prefill_bwd_workspace_bytes(const umoe_attn_bwd_args* a);
int umoe_attn_prefill_bwd(const umoe_attn_bwd_args* a, umoe_stream_t stream);
/* backward of umoe_qkv_mrope_kvappend: dq [n_tok][H*hd], dk / dv in cache layout -> d(qkv) [n_tok][(H+2KVH)*hd] */
int umoe_qkv_mrope_bwd(const umoe_rope_args* a, const uint16_t* dq, const uint16_t* dk_cache, const uint16_t* dv_cache,
                       uint16_t* dqkv, umoe_stream_t stream);



/* Attention over the cache for nq query tokens per row (nq=1 decode, nq=T causal prefill).
 * q [rows*nq][H*hd]; query t of a row sees cache slots [kv_start[row], q_pos0[row] + t].
 * Split over keys (flash-decoding) + umoe_attn_combine.  out [rows*nq][H*hd] bf16.
 * Replaces Qwen2_5_VLAttention's sdpa/eager core (model.py:228-237). */
typedef struct {
    const uint16_t* q;
    const uint16_t* k_cache;
    const uint16_t* v_cache;
    const int32_t* kv_start;  /* [rows] first valid slot (left padding) */
    const int32_t* q_pos0;    /* [rows] cache slot of this call's first query */
    int rows, nq, H, KVH, hd, Lmax;
    int splits;               /* key splits per (row, kv head, query) */
    float scale;
    float* part_o;            /* [rows*nq][H][splits][hd] */
    float* part_ml;           /* [rows*nq][H][splits][2]  */
    uint16_t* out;
    /* decode fusion (nq == 1): when qkv_raw is set the kernel applies mRoPE to q and to the new k itself, appends the
     * new K/V to the cache slot q_pos0[row] (the caches are then written) and `q` is ignored */
    const uint16_t* qkv_raw;  /* [rows][(H+2*KVH)*hd] bias-added QKV of the new token */
    const uint16_t* cos_tab;
    const uint16_t* sin_tab;
    const int32_t* pos3;      /* [3][rows] */
    int sec0, sec1, sec2;
    float* lse_out;           /* optional (umoe_attn_prefill_fwd, MFMA path): log-sum-exp per (query, head) [rows*nq][H], +inf for
                               * queries that see no key; kept for umoe_attn_prefill_bwd */
} umoe_attn_args;
int umoe_attn_decode(const umoe_attn_args* a, umoe_stream_t stream);
/* causal prefill (nq = T queries per row) over keys already appended by umoe_qkv_mrope_kvappend */
int umoe_attn_prefill_fwd(const umoe_attn_args* a, umoe_stream_t stream);

/* ------------------------------------------------------------------ codec side
 * codec_embedding (model.py:655-661): out[r] = sum_c Emb_c[tok[r][c]], bf16 adds in channel order.
 * emb [C][V][D]. */
int umoe_codec_embed_sum(const int32_t* tok, const uint16_t* emb, int rows, int C, int V, int D, uint16_t* out,
                         umoe_stream_t stream);
/* backward of the same (the reference's autograd of the C nn.Embedding gathers, model.py:655-661): d_emb[c][v] = sum over the rows r
 * with tok[r][c] == v of d_out[r], fp32 accumulation in ASCENDING row order (deterministic), one rounding to bf16; rows of the table
 * nobody selected are written as zeros.  D % 8 == 0, D <= 2048 * 8. */
int umoe_codec_embed_sum_bwd(const int32_t* tok, const uint16_t* d_out, int rows, int C, int V, int D, uint16_t* d_emb,
                             umoe_stream_t stream);

/* CFG + masks + sampling on codec-head logits (model.py:991-1017, 873-916).
 * logits [2B][C*V] fp32 (row 2b = uncond, 2b+1 = cond).  temperature 0 or do_sample 0 => arg-max.
 * probs_out optional [B*C][V] (post-filter probabilities, for parity).  rng: Philox-free counter
 * hash seeded by (seed, step, row). */
typedef struct {
    const float* logits;
    int B, C, V;
    float cfg_scale, temperature, top_p, eos_mul;
    int top_k;                /* <=0: none */
    int eos;
    int enable_eos;           /* host flag; or enable_eos_from_step >= 0: enabled iff *step >= that */
    int min_tokens;           /* -1 = None */
    const int32_t* step;      /* device scalar dec_step (may be NULL) */
    int do_sample;
    uint64_t seed;
    int64_t* pred;            /* [B][C] */
    float* probs_out;
} umoe_sample_args;
int umoe_codec_head_cfg_sample(const umoe_sample_args* a, umoe_stream_t stream);

/* One step of generate()'s token bookkeeping on device (model.py:1173-1203 + DecoderOutput.update_one,
 * utils/UniMoE_Audio_utils.py:290-298): EOS detection on channel 0, countdown, forced EOS/PAD by delay
 * pattern, BOS masking, append into tokens[B][Tmax][C], advance *step.  state = int32[4*B + 8]:
 * eos_detected[B], countdown[B], finished[B], prefill_step[B], then {step, max_tokens, all_done, bos_over,
 * step0 (dec_step of the first decode call), 3 reserved}. */
int umoe_delay_step(int64_t* pred, int32_t* tokens, int32_t* state, const int32_t* delay, int B, int C, int Tmax,
                    int eos, int pad, int max_delay, umoe_stream_t stream);

/* Per-channel codec cross-entropy of the training loss (model.py:830-847) on already shifted logits [N][C][V] fp32 and
 * labels [N][C] int64 (-100 = ignore): ch_loss[c] = mean nll over valid labels, total = ch_loss[0] + sum of channels with
 * at least one valid label.  probs (optional [N][C][V]) keeps the softmax rows for umoe_codec_ce_bwd:
 * dlogits = grad * (softmax - onehot) / count_c.  Fixed-order reductions, no atomics. */
int umoe_codec_ce_fwd(const float* logits, const int64_t* labels, int N, int C, int V, float* nll_ws, float* probs,
                      float* ch_loss, int32_t* ch_count, float* total, umoe_stream_t stream);
int umoe_codec_ce_bwd(const float* probs, const int64_t* labels, const int32_t* ch_count, int N, int C, int V, float grad,
                      float* dlogits, umoe_stream_t stream);

/* RVQ (third-party descript-audio-codec 1.0.0 ResidualVectorQuantize, call sites
 * utils/UniMoE_Audio_utils.py:113,123): from_codes = sum_q out_proj_q(codebook_q[code]); nearest = per level
 * in_proj, L2-normalised nearest neighbour, subtract.  All DAC dims are load-time parameters. */
int umoe_rvq_from_codes(const int32_t* codes, const float* codebooks, const float* out_w, const float* out_b, int NQ,
                        int CB, int cd, int Dl, int T, float* z, umoe_stream_t stream);
int umoe_rvq_nearest(const float* z, const float* codebooks, const float* in_w, const float* in_b, const float* out_w,
                     const float* out_b, int NQ, int CB, int cd, int Dl, int T, int32_t* codes, float* resid_ws,
                     umoe_stream_t stream);

/* DAC conv stacks (third-party descript-audio-codec 1.0.0 DAC.encode / DAC.decode behind utils/UniMoE_Audio_utils.py:112-113,123-124;
 * restated from the published architecture, PARITY UNPINNED, every dimension a load-time parameter).  fp32, [B][C][L] layout.
 * The Snake activation of the PRECEDING layer (x + sin(alpha x)^2 / (alpha + 1e-9), per input channel) is fused into the input
 * load when snake_alpha != NULL; bias, the residual add of a ResidualUnit (resid, same shape as y) and the decoder's final tanh
 * (act = 1) into the store.  Weights are the weight-normalised tensors already folded (g * v / |v|, done once at load time).
 *   umoe_dac_conv1d:            w [Cout][Cin][K], Lout = (L + 2 pad - dilation (K - 1) - 1) / stride + 1
 *   umoe_dac_conv_transpose1d:  w [Cin][Cout][K], Lout = (L - 1) stride - 2 pad + K + out_pad */
int umoe_dac_conv1d(const float* x, const float* w, const float* bias, const float* snake_alpha, const float* resid, int B, int Cin,
                    int L, int Cout, int K, int stride, int dilation, int pad, int act, float* y, int* Lout_out, umoe_stream_t stream);
int umoe_dac_conv_transpose1d(const float* x, const float* w, const float* bias, const float* snake_alpha, int B, int Cin, int L,
                              int Cout, int K, int stride, int pad, int out_pad, float* y, int* Lout_out, umoe_stream_t stream);
/* windowed-sinc resampling to the codec's rate (torchaudio.transforms.Resample as used at utils.py:101-110, restated): with o / n =
 * orig / new frequency over their gcd, kern [n][2 width + o] the filter bank (built by the host from the published formula),
 * y[b][frame * n + phase] = sum_t kern[phase][t] * x[b][frame * o - width + t] (zero outside [0, L)). */
int umoe_dac_resample(const float* x, const float* kern, int B, int L, int o, int n, int width, int Lout, float* y, umoe_stream_t stream);

/* Vision tower pieces (reference utils/UniMoE_Audio_utils.py:756-900 over the third-party transformers vision block /
 * attention / MLP / patch merger).  The projections run on umoe_tiled_gemm; these are the rest.
 *   umoe_vision_rope: apply_rotary_pos_emb_vision on q and k inside the fused qkv buffer [S][3][H][hd] (fp32 arithmetic, one rounding);
 *                     cos / sin [S][hd] fp32.
 *   umoe_vision_attn: non-causal attention of token s over the keys [seg_lo[s], seg_hi[s]) of its cu_seqlens segment; out [S][H*hd].
 *   umoe_swiglu_pair: gu [S][2I] = (gate | up) with biases -> h [S][ldh] = bf16(bf16(silu(g)) * u), columns [I, ldh) zero.
 *   umoe_gelu:        exact GELU in place (patch merger). */
int umoe_vision_rope(uint16_t* qkv, const float* cos_t, const float* sin_t, int S, int H, int hd, umoe_stream_t stream);
int umoe_vision_attn(const uint16_t* qkv, const int32_t* seg_lo, const int32_t* seg_hi, int S, int H, int hd, float scale, uint16_t* out,
                     umoe_stream_t stream);
int umoe_swiglu_pair(const uint16_t* gu, int S, int I, int ldh, uint16_t* h, umoe_stream_t stream);
int umoe_gelu(uint16_t* x, long n, umoe_stream_t stream);

/* ------------------------------------------------------------------ decode engine
 * Owns workspace + KV cache and enqueues a whole decode step (36 layers + head + sampler + delay
 * bookkeeping) from one host call, optionally replayed as a hipGraph.  Restates
 * generate()/_decoder_step() control flow (model.py:918-1231) without per-step host syncs. */
typedef struct {
    int hidden, layers, heads, kv_heads, head_dim;
    int n_dyn, n_real, n_fix, inter_dyn, inter_shared;
    int codec_channels, codec_vocab, eos, pad, bos;
    int mrope0, mrope1, mrope2;
    float rms_eps, top_p;
    int fixed_top_k;
    double jitter_eps;
    int rows;                 /* 2 * batch (CFG pairs) */
    int Lmax;                 /* KV slots per row */
    int Tmax;                 /* token buffer length */
    int attn_splits;
    int ep_rank, ep_size;     /* expert parallel decode: this rank owns experts [rank*n_real/size, ...); ep_size in {1,2,4,8},
                               * rows <= 16 per rank; see umoe_engine_ep_connect */
} umoe_engine_cfg;

typedef struct {
    const uint16_t* in_norm;      /* [D] */
    const uint16_t* qkv_w;        /* WP16 of cat(q,k,v) [H*hd + 2*KVH*hd][D] */
    const float* qkv_b;           /* [H*hd + 2*KVH*hd] fp32 */
    const uint16_t* o_w;          /* WP16 [D][H*hd] */
    const uint16_t* post_norm;    /* [D] */
    const uint16_t* gate_w;       /* [E][D] plain bf16 */
    const uint16_t* const* exp_gu; /* host array [n_real] of device ptrs: WP16 gate/up pairs */
    const uint16_t* const* exp_dn; /* host array [n_real]: WP16 down */
    const uint16_t* const* sh_gu;  /* host array [n_fix] */
    const uint16_t* const* sh_dn;  /* host array [n_fix] */
    /* optional ROW-MAJOR copies (the reference's own nn.Linear tensors) for the tiled MFMA path of the prefill; all or none.
     * rm_qkv = cat(q,k,v) [QKV][D], rm_o [D][H*hd]; per expert gate/up [I][D] and down [D][I]. */
    const uint16_t* rm_qkv;
    const uint16_t* rm_o;
    const uint16_t* const* rm_exp_gate;  /* host arrays [n_real] */
    const uint16_t* const* rm_exp_up;
    const uint16_t* const* rm_exp_down;
    const uint16_t* const* rm_sh_gate;   /* host arrays [n_fix] */
    const uint16_t* const* rm_sh_up;
    const uint16_t* const* rm_sh_down;
} umoe_layer_weights;

typedef struct umoe_engine umoe_engine;
int umoe_engine_create(const umoe_engine_cfg* cfg, umoe_engine** out);
void umoe_engine_destroy(umoe_engine* e);
int umoe_engine_set_layer(umoe_engine* e, int layer, const umoe_layer_weights* w);
/* final norm [D], codec embeddings [C][V][D], WP16 codec head [C*V][D], rope tables [max_pos][hd/2] */
int umoe_engine_set_globals(umoe_engine* e, const uint16_t* final_norm, const uint16_t* codec_emb,
                            const uint16_t* codec_head_w, const uint16_t* cos_tab, const uint16_t* sin_tab, int max_pos,
                            const int32_t* delay_pattern_host);
size_t umoe_engine_workspace_bytes(const umoe_engine* e);
/* prefill: x [rows*T][D] input embeddings (host builds them: text embed + codec scatter, model.py:663-670),
 * valid [rows][T] 0/1 attention mask (left padded). Fills the KV cache. */
int umoe_engine_prefill(umoe_engine* e, const uint16_t* x, const uint8_t* valid_host, int T, umoe_stream_t stream);
/* the same with explicit rotary positions (multimodal prompts, reference get_rope_index model.py:513-652): pos3_host [3][rows][T]
 * int32 (NULL = from the mask as above), next_pos_host [rows] = position of the first generated token of each row = max position + 1
 * (the reference's cache_position + rope_deltas; NULL = number of valid tokens) */
int umoe_engine_prefill_pos(umoe_engine* e, const uint16_t* x, const uint8_t* valid_host, int T, const int32_t* pos3_host,
                            const int32_t* next_pos_host, umoe_stream_t stream);
/* the prompt was run outside the engine (an expert-parallel rank that holds ONLY its local experts -- core.py:505,
 * deepspeed_ep_param_aggregation.py:16-48 -- prefills through the module-level forward, whose DCMoE blocks exchange rows between the ranks)
 * and the caller copied the roped K / V of every layer into slots [0, T) of the cache buffers (umoe_engine_buffer "k_cache" / "v_cache",
 * [layer][row][kv_head][Lmax][head_dim]): sets the decode state only (arguments as umoe_engine_prefill_pos) */
int umoe_engine_prefill_external(umoe_engine* e, const uint8_t* valid_host, int T, const int32_t* pos3_host, const int32_t* next_pos_host,
                                 umoe_stream_t stream);
/* decode bookkeeping state + token buffer (device, owned by caller) */
typedef struct {
    int32_t* tokens;          /* [B][Tmax][C], -1 = to be generated */
    int32_t* state;           /* see umoe_delay_step */
    float cfg_scale, temperature, top_p, eos_mul;
    int top_k, do_sample, min_tokens;
    uint64_t seed;
} umoe_decode_io;
int umoe_engine_decode_step(umoe_engine* e, const umoe_decode_io* io, umoe_stream_t stream);
/* graph capture of one decode step; replays read all step-dependent scalars from device memory */
int umoe_engine_capture(umoe_engine* e, const umoe_decode_io* io, umoe_stream_t stream);
int umoe_engine_replay(umoe_engine* e, umoe_stream_t stream);
/* one eager step with a hipEvent (on `stream`) after every kernel class; accumulates into ms[13] / launches[13]:
 * {qkv, rope, attn, oproj, router, dispatch, gateup, down, combine, embed, head, sample, delay} */
int umoe_engine_profile_step(umoe_engine* e, const umoe_decode_io* io, umoe_stream_t stream, float* ms, int* launches,
                             int n);
/* Expert parallel decode (ep_size > 1).  umoe_engine_set_layer then takes exp_gu / exp_dn of the n_real / ep_size LOCAL experts
 * (global ids [ep_rank * E_loc, (ep_rank + 1) * E_loc), core.py:505) and EITHER the row-major tensors of ALL n_real experts (the
 * engine then prefills replicated on its tiled path) OR no row-major expert tensors at all (rm_exp_* NULL: the rank holds its local
 * experts only; prefill through umoe_engine_prefill_external).
 * umoe_engine_ep_region: this rank's exchange region (export with umoe_ep_ipc_export).
 * umoe_engine_ep_connect: peers[p] = rank p's region as mapped in this process (peers[ep_rank] = own base).
 *   mode UMOE_EP_PEER: xGMI peer stores; UMOE_EP_LOOPBACK: single-GPU emulation of ONE rank of an ep_size job for timing
 *   (every peer is this engine itself: tile p of the local slabs stands for rank p; results are not the model's);
 *   UMOE_EP_RCCL: `rccl_comm` (ncclComm_t) carries the same exchange as ncclAllGather + grouped ncclSend/ncclRecv.
 * umoe_engine_ep_error: synchronises `stream` and returns the sticky error word (0 = none; 1 = a receive timed out). */
enum { UMOE_EP_PEER = 0, UMOE_EP_LOOPBACK = 1, UMOE_EP_RCCL = 2 };
int umoe_engine_ep_region(umoe_engine* e, void** base_out, size_t* bytes_out);
int umoe_engine_ep_connect(umoe_engine* e, void* const* peers, void* rccl_comm, int mode);
int umoe_engine_ep_error(umoe_engine* e, umoe_stream_t stream, int* code_out);
/* per-layer probe for parity tests (EAGER decode steps only; capture refuses it).  teach_x [layers][rows][D] bf16 replaces the residual
 * stream at the START of every layer (the layer's input RMSNorm is recomputed from it), so each layer is checked on the oracle's input
 * of that layer without compounding (model.py:210-256 one layer at a time); dump_x1 / dump_x [layers][rows][D] receive the stream after
 * attention + o_proj and after the MoE block, dump_logits [layers][rows][E] the router logits (bf16).  Any pointer may be NULL; all
 * NULL switches the probe off.  With a probe the layer's combine runs as its own launch (it cannot ride in a QKV launch whose input
 * is replaced); every other launch is the decode step's own. */
int umoe_engine_set_probe(umoe_engine* e, const uint16_t* teach_x, uint16_t* dump_x1, uint16_t* dump_x, uint16_t* dump_logits);
/* host-side facts about the engine: "expert_launch" = what the last dense decode layer enqueued for its experts (0 gate/up and down as
 * two launches, 1 the box-grid fused launch moe_fused_kernel, 2 the flat launch moe_flat_kernel, 3 the one-launch expert-parallel MoE half
 * moe_ep_kernel), "n_cu" = compute units the
 * co-residency guards assume; -1 for an unknown key */
int umoe_engine_info(umoe_engine* e, const char* key);
/* introspection for parity tests: device pointers into the workspace */
const void* umoe_engine_buffer(umoe_engine* e, const char* name, size_t* bytes);

#ifdef __cplusplus
}
#endif
#endif /* UMOE_H */
