// Tiled MFMA GEMM for the compute-bound shapes (prefill: 4 800 tokens, training: 6 240 tokens per step).
//
//   Y[rows(g), N] = epilogue( A[rows(g), K] * W_g^T ),  W_g row-major [N][K] bf16
//
// Design (MI355X / gfx950):
//  * one 256-thread workgroup per 128 x 128 output tile, 4 waves as 2 (rows) x 2 (columns), each wave a 64 x 64 sub-tile
//    = 4 x 4 v_mfma_f32_16x16x32_bf16 accumulators (64 registers);
//  * K in steps of 64: the A and W tiles (16 KiB each) go global -> registers -> LDS, double-buffered, one barrier per step;
//    the loads of step t+1 are in flight while step t is multiplied;
//  * LDS image: 128-byte rows (64 bf16), 16-byte chunk c of row r at r*128 + ((c ^ ((r >> 1) & 7)) * 16): the global loads
//    and the LDS writes are contiguous per row (8 lanes x 16 B), the ds_read_b128 operand reads (16 rows x one chunk)
//    touch every bank once;
//  * weights are the MFMA A operand, activations the B operand (as in umoe_gemm.hip): a lane ends with 4 consecutive
//    output features of one token -> 8-byte bf16 stores;
//  * ragged groups: gather list / row count / row offset are read on device, workgroups beyond the count exit.
// Roofline: MFMA (2.5 PFLOP/s dense bf16).  Intensity of a 128x128 tile: 64 flop per byte moved from L2.
#include "umoe_common.h"
#include <string.h>

#define TG_MAXG 12
struct tg_pack { umoe_tgroup_t g[TG_MAXG]; };

// 16-byte chunk c of tile row r.  Rows are 128 B = 32 banks, so two consecutive rows span the 64 banks; an operand read
// takes ONE chunk column of 16 consecutive rows: rows of equal parity must land on 8 different chunk slots -> swizzle by
// (r >> 1) & 7 (conflict-free); swizzling by r & 7 left rows r and r + 8 on the same banks (2-way).
__device__ __forceinline__ int tg_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int EPI>
__global__ __launch_bounds__(256, 2) void tgemm_kernel(const umoe_tgemm_args p, const tg_pack gp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][A 16 KiB | W 16 KiB]
    const umoe_tgroup_t g = gp.g[blockIdx.z];
    const int count = g.count ? *g.count : g.static_count;
    const int roff = g.row_off ? *g.row_off : 0;
    const int row0 = blockIdx.y * 128;
    if (row0 >= count) return;
    // SwiGLU: the tile's 128 weight rows are 64 gate rows and the 64 up rows of the same features
    constexpr bool SW = EPI == UMOE_EPI_SWIGLU;
    constexpr int NTILE = SW ? 64 : 128;
    const int n0 = blockIdx.x * NTILE;
    if (n0 >= g.n) return;
    const int koff = g.k_off ? *g.k_off : 0;                       // multiple of 8 (aligned dispatch)
    const int K = g.k_count ? ((*g.k_count + 7) & ~7) : g.k;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- global -> register tile loads: thread (lr = tid / 8, ch = tid % 8) owns chunk ch of rows lr + 32 * pass -----
    const int lr = tid >> 3, ch = tid & 7;
    const uint16_t* ap[4];
    const uint16_t* wp[4];
    bool aok[4], wok[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int r = row0 + lr + 32 * ps;
        aok[ps] = r < count;
        long arow = 0;
        if (aok[ps]) arow = g.rows ? (long)g.rows[roff + r] : (long)(g.a_row_base + roff + r);
        ap[ps] = p.a + arow * (long)p.lda + g.a_col_off + koff + ch * 8;
        const int tr = lr + 32 * ps;   // tile row 0..127
        int n;
        const uint16_t* wb = g.w;
        if (SW) {
            n = n0 + (tr & 63);
            if (tr >= 64) wb = g.w2;
        } else {
            n = n0 + tr;
        }
        wok[ps] = n < g.n;
        wp[ps] = wb + (long)(wok[ps] ? n : 0) * g.ldw + koff + ch * 8;
    }
    uint4 ra[4], rw[4];
    auto gload = [&](int k0) {
        const bool kok = k0 + ch * 8 < K;   // K % 8 == 0: a chunk is entirely inside or outside
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            ra[ps] = make_uint4(0, 0, 0, 0);
            rw[ps] = make_uint4(0, 0, 0, 0);
            if (kok && aok[ps]) ra[ps] = ld16(ap[ps] + k0);
            if (kok && wok[ps]) rw[ps] = ld16(wp[ps] + k0);
        }
    };
    auto lstore = [&](int buf) {
        char* A = smem + buf * 32768;
        char* W = A + 16384;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int tr = lr + 32 * ps;
            st16(A + tg_off(tr, ch), ra[ps]);
            st16(W + tg_off(tr, ch), rw[ps]);
        }
    };

    f32x4_t acc[4][4];   // [weight sub-tile j][token sub-tile i]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int h = lane >> 4, c16 = lane & 15;
    // wave (wm, wn): token rows [64 wm, +64); weight rows: plain [64 wn, +64); SwiGLU gate [32 wn, +32) and up [64 + 32 wn, +32)
    int wrow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wrow[j] = SW ? ((j < 2 ? 0 : 64) + 32 * wn + 16 * (j & 1)) : (64 * wn + 16 * j);

    const int KT = (K + 63) >> 6;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KT) gload((kt + 1) << 6);
        const char* A = smem + buf * 32768;
        const char* W = A + 16384;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8_t af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(A + tg_off(64 * wm + 16 * i + c16, s * 4 + h)));
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wf[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(W + tg_off(wrow[j] + c16, s * 4 + h)));
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[j][i], 0, 0, 0);
        }
        if (kt + 1 < KT) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: acc[j][i] lane (h, c16): token row 64 wm + 16 i + c16, features (weight rows) wrow[j] + 4 h .. +3 ----
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + 64 * wm + 16 * i + c16;
        if (r >= count) continue;
        const long orow = (long)g.out_row_base + roff + r;
        const int oc = g.out_col_off;
        if (SW) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + 32 * wn + 16 * j + 4 * h;
                if (col >= g.n) continue;
                uint16_t y[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float gt = rbf(acc[j][i][q]);
                    const float up = rbf(acc[j + 2][i][q]);
                    const float si = rbf(gt / (1.0f + expf(-gt)));
                    y[q] = f2bf(si * up);
                    if (p.aux_out && col + q < g.n) {
                        p.aux_out[orow * p.ld_aux + col + q] = f2bf(gt);
                        p.aux_out[orow * p.ld_aux + g.n + col + q] = f2bf(up);
                    }
                }
                uint16_t* o = reinterpret_cast<uint16_t*>(p.out) + orow * p.ldo + oc + col;
                if (col + 3 < g.n && ((p.ldo | oc) & 3) == 0) {
                    *reinterpret_cast<uint2*>(o) = make_uint2((uint32_t)y[0] | ((uint32_t)y[1] << 16), (uint32_t)y[2] | ((uint32_t)y[3] << 16));
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (col + q < g.n) o[q] = y[q];
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = n0 + wrow[j] + 4 * h;
                if (col >= g.n) continue;
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = acc[j][i][q] + ((g.bias && col + q < g.n) ? g.bias[col + q] : 0.f);
                if (EPI == UMOE_EPI_F32 || EPI == UMOE_EPI_F32_RAW) {
                    float* o = reinterpret_cast<float*>(p.out) + orow * p.ldo + oc + col;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (col + q < g.n) o[q] = (EPI == UMOE_EPI_F32) ? rbf(v[q]) : v[q];
                } else {
                    uint16_t y[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float x = rbf(v[q]);
                        if (EPI == UMOE_EPI_BF16_RESID && col + q < g.n) x = bf2f(p.resid[orow * p.ldo + oc + col + q]) + x;
                        y[q] = f2bf(x);
                    }
                    uint16_t* o = reinterpret_cast<uint16_t*>(p.out) + orow * p.ldo + oc + col;
                    if (col + 3 < g.n && ((p.ldo | oc) & 3) == 0) {
                        *reinterpret_cast<uint2*>(o) = make_uint2((uint32_t)y[0] | ((uint32_t)y[1] << 16), (uint32_t)y[2] | ((uint32_t)y[3] << 16));
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (col + q < g.n) o[q] = y[q];
                    }
                }
            }
        }
    }
}

template <int EPI>
static int launch_tgemm(const umoe_tgemm_args* a, int max_n, hipStream_t s) {
    tg_pack gp;
    memset(&gp, 0, sizeof(gp));
    memcpy(gp.g, a->groups, sizeof(umoe_tgroup_t) * a->num_groups);
    static bool configured = false;
    if (!configured) {
        UMOE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tgemm_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        configured = true;
    }
    const int ntile = EPI == UMOE_EPI_SWIGLU ? 64 : 128;
    dim3 grid((unsigned)ceil_div(max_n, ntile), (unsigned)ceil_div(a->max_rows, 128), (unsigned)a->num_groups);
    tgemm_kernel<EPI><<<grid, 256, 65536, s>>>(*a, gp);
    UMOE_LAUNCH_CHECK();
    return 0;
}

extern "C" int umoe_tiled_gemm(const umoe_tgemm_args* a, umoe_stream_t stream) {
    UMOE_REQUIRE(a && a->groups && a->a && a->out, "umoe_tiled_gemm: null argument");
    UMOE_REQUIRE(a->num_groups > 0 && a->num_groups <= TG_MAXG, "umoe_tiled_gemm: 1..%d groups per launch (got %d)", TG_MAXG, a->num_groups);
    UMOE_REQUIRE(a->max_rows > 0 && ceil_div(a->max_rows, 128) <= 65535, "umoe_tiled_gemm: bad max_rows %d", a->max_rows);
    UMOE_REQUIRE((a->lda & 7) == 0, "umoe_tiled_gemm: lda must be a multiple of 8");
    int max_n = 0;
    for (int i = 0; i < a->num_groups; ++i) {
        const umoe_tgroup_t& g = a->groups[i];
        UMOE_REQUIRE(g.w && g.n > 0 && (g.k_count || (g.k > 0 && g.k % 8 == 0 && g.ldw >= g.k)) && g.ldw % 8 == 0 && (g.a_col_off & 7) == 0,
                     "umoe_tiled_gemm: group %d: need w, n > 0, k %% 8 == 0, ldw %% 8 == 0 (n=%d k=%d ldw=%d)", i, g.n, g.k, g.ldw);
        UMOE_REQUIRE(a->epilogue != UMOE_EPI_SWIGLU || g.w2, "umoe_tiled_gemm: SwiGLU needs w2 (up_proj)");
        if (g.n > max_n) max_n = g.n;
    }
    hipStream_t s = (hipStream_t)stream;
    switch (a->epilogue) {
        case UMOE_EPI_BF16: return launch_tgemm<UMOE_EPI_BF16>(a, max_n, s);
        case UMOE_EPI_BF16_RESID:
            UMOE_REQUIRE(a->resid, "umoe_tiled_gemm: residual epilogue needs resid");
            return launch_tgemm<UMOE_EPI_BF16_RESID>(a, max_n, s);
        case UMOE_EPI_SWIGLU: return launch_tgemm<UMOE_EPI_SWIGLU>(a, max_n, s);
        case UMOE_EPI_F32: return launch_tgemm<UMOE_EPI_F32>(a, max_n, s);
        case UMOE_EPI_F32_RAW: return launch_tgemm<UMOE_EPI_F32_RAW>(a, max_n, s);
    }
    UMOE_REQUIRE(false, "umoe_tiled_gemm: bad epilogue %d", a->epilogue);
}
