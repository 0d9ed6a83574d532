// Tiled MFMA GEMM for the compute-bound shapes (prefill: 4 800 tokens, training: 6 240 tokens per step).
//
//   Y[rows(g), N] = epilogue( A[rows(g), K] * W_g^T ),  W_g row-major [N][K] bf16
//
// Design (MI355X / gfx950):
//  * one 256-thread workgroup per 128 x 128 output tile, 4 waves as 2 (rows) x 2 (columns), each wave a 64 x 64 sub-tile
//    = 4 x 4 v_mfma_f32_16x16x32_bf16 accumulators (64 registers);
//  * tiles are staged by LDS-DMA (global_load_lds_dwordx4: no data registers, no ds_write pass; the swizzle sits on the
//    per-lane source address); >= 1024 rows: 256-token tiles (128 x 64 wave tiles), K steps of 32, THREE LDS stages with the
//    DMA of tile t+2 in flight across the barrier (counted vmcnt + raw s_barrier); fewer rows: 128-token tiles, K steps of 64,
//    two stages; register staging kept as a variant;
//  * LDS image: 128-byte rows (64 bf16), 16-byte chunk c of row r at r*128 + ((c ^ ((r >> 1) & 7)) * 16): the global loads
//    and the LDS writes are contiguous per row (8 lanes x 16 B), the ds_read_b128 operand reads (16 rows x one chunk)
//    touch every bank once;
//  * weights are the MFMA A operand, activations the B operand (as in umoe_gemm.hip): a lane ends with 4 consecutive
//    output features of one token -> 8-byte bf16 stores;
//  * ragged groups: gather list / row count / row offset are read on device, workgroups beyond the count exit.
// Roofline: MFMA (2.5 PFLOP/s dense bf16).  Intensity of a 128x128 tile: 64 flop per byte moved from L2.
#include "umoe_common.h"
#include <stdlib.h>
#include <string.h>

#define TG_MAXG 12
struct tg_pack { umoe_tgroup_t g[TG_MAXG]; };

// 16-byte chunk c of tile row r, rows of BKC chunks.  An operand read takes ONE chunk column of 16 consecutive rows; rows
// are BKC*16 B = BKC*4 banks, so 16/BKC consecutive rows span the 64 banks and the rows that share banks (every 16/BKC-th)
// must land on different chunk slots -> swizzle by (r / (16/BKC)) % BKC: conflict-free (SQ_LDS_BANK_CONFLICT = 0).
template <int BKC>
__device__ __forceinline__ int tg_off(int row, int chunk) {
    constexpr int SH = BKC == 8 ? 1 : 2;
    return row * (BKC * 16) + ((chunk ^ ((row >> SH) & (BKC - 1))) << 4);
}

// MI = 16-token sub-tiles per wave: 4 -> 128-token tiles, K steps of 64 (BKC 8), 64 x 64 wave tiles;
//                                   8 -> 256-token tiles, K steps of 32 (BKC 4), 128 x 64 wave tiles (12 operand reads per 32 MFMAs
//                                        instead of 16: the 64 x 64 wave tile is LDS-bandwidth bound at full MFMA rate)
// one 16-byte block of zeros: the LDS-DMA source of chunks outside a tile (rows beyond the count, K tail)
__device__ __attribute__((aligned(16))) uint4 tg_zero16;

// NST: LDS stages.  0 = register staging (2 buffers); 2 = LDS-DMA, 2 buffers, one __syncthreads per K step (its fence waits for
// the DMA); 3 = LDS-DMA, 3 buffers, the DMA of tile t+2 stays in flight ACROSS the barrier: counted s_waitcnt vmcnt(N) + raw s_barrier
// (a wave waits for its own pieces of tile t, the barrier makes every wave's pieces visible; the buffer restaged after the
// barrier was last read one iteration earlier)
template <int EPI, int MI, int BKC, int NST>
__global__ __launch_bounds__(256, 2) void tgemm_kernel(const umoe_tgemm_args p, const tg_pack gp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][A tile | W tile]
    constexpr int TM = 32 * MI;                 // token rows per workgroup
    constexpr int RPP = 256 / BKC;              // tile rows covered by one load pass of the 256 threads
    constexpr int APS = TM / RPP, WPS = 128 / RPP;
    constexpr int ABYTES = TM * BKC * 16, WBYTES = 128 * BKC * 16, BUF = ABYTES + WBYTES;
    constexpr int KS = BKC / 4;                 // MFMA k-steps (32 wide) per K iteration
    constexpr bool GLDS = NST >= 2;
    const umoe_tgroup_t g = gp.g[blockIdx.z];
    const int count = g.count ? *g.count : g.static_count;
    const int roff = g.row_off ? *g.row_off : 0;
    const int row0 = blockIdx.y * TM;
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

    // ---- global -> register tile loads: thread (lr = tid / BKC, ch = tid % BKC) owns chunk ch of rows lr + RPP * pass -----
    const int lr = tid / BKC, ch = tid % BKC;
    const uint16_t* ap[APS];
    const uint16_t* wp[WPS];
    bool aok[APS], wok[WPS];
#pragma unroll
    for (int ps = 0; ps < APS; ++ps) {
        const int r = row0 + lr + RPP * ps;
        aok[ps] = r < count;
        long arow = 0;
        if (aok[ps]) arow = g.rows ? (long)g.rows[roff + r] : (long)(g.a_row_base + roff + r);
        ap[ps] = p.a + arow * (long)p.lda + g.a_col_off + koff + ch * 8;
    }
#pragma unroll
    for (int ps = 0; ps < WPS; ++ps) {
        const int tr = lr + RPP * ps;   // tile row 0..127
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
    uint4 ra[APS], rw[WPS];
    auto gload = [&](int k0) {
        const bool kok = k0 + ch * 8 < K;   // K % 8 == 0: a chunk is entirely inside or outside
#pragma unroll
        for (int ps = 0; ps < APS; ++ps) {
            ra[ps] = make_uint4(0, 0, 0, 0);
            if (kok && aok[ps]) ra[ps] = ld16(ap[ps] + k0);
        }
#pragma unroll
        for (int ps = 0; ps < WPS; ++ps) {
            rw[ps] = make_uint4(0, 0, 0, 0);
            if (kok && wok[ps]) rw[ps] = ld16(wp[ps] + k0);
        }
    };
    auto lstore = [&](int buf) {
        char* A = smem + buf * BUF;
        char* W = A + ABYTES;
#pragma unroll
        for (int ps = 0; ps < APS; ++ps) st16(A + tg_off<BKC>(lr + RPP * ps, ch), ra[ps]);
#pragma unroll
        for (int ps = 0; ps < WPS; ++ps) st16(W + tg_off<BKC>(lr + RPP * ps, ch), rw[ps]);
    };

    // ---- LDS-DMA staging (global_load_lds_dwordx4): one wave-instruction writes 1 KiB of LDS at wave-uniform base +
    //      lane * 16 = 64 / BKC whole tile rows; the swizzle goes on the per-lane SOURCE address (the LDS image is the same
    //      as with register staging).  No data registers, no ds_write pass; chunks outside the tile read a block of zeros.
    constexpr int RPG = 64 / BKC;                         // tile rows per wave-instruction
    constexpr int AGW = TM / RPG / 4, WGW = 128 / RPG / 4;   // groups per wave
    const uint16_t* gap[GLDS ? AGW : 1];
    const uint16_t* gwp[GLDS ? WGW : 1];
    int gch = 0;                                          // this lane's chunk (same for every group: row & swizzle mask repeats)
    if (GLDS) {
        const int rl = lane / BKC, slot = lane % BKC;
#pragma unroll
        for (int j = 0; j < AGW; ++j) {
            const int tr = (wave + 4 * j) * RPG + rl;     // tile row
            constexpr int SH = BKC == 8 ? 1 : 2;
            const int c = slot ^ ((tr >> SH) & (BKC - 1));
            gch = c;                                      // RPG * 4 is a multiple of the swizzle period: c does not depend on j
            const int r = row0 + tr;
            gap[j] = nullptr;
            if (r < count) {
                const long arow = g.rows ? (long)g.rows[roff + r] : (long)(g.a_row_base + roff + r);
                gap[j] = p.a + arow * (long)p.lda + g.a_col_off + koff + c * 8;
            }
        }
#pragma unroll
        for (int j = 0; j < WGW; ++j) {
            const int tr = (wave + 4 * j) * RPG + rl;
            constexpr int SH = BKC == 8 ? 1 : 2;
            const int c = slot ^ ((tr >> SH) & (BKC - 1));
            int n;
            const uint16_t* wb = g.w;
            if (SW) {
                n = n0 + (tr & 63);
                if (tr >= 64) wb = g.w2;
            } else {
                n = n0 + tr;
            }
            gwp[j] = n < g.n ? wb + (long)n * g.ldw + koff + c * 8 : nullptr;
        }
    }
    auto stage = [&](int buf, int k0) {
        const bool kok = k0 + gch * 8 < K;
        const uint16_t* zero = reinterpret_cast<const uint16_t*>(&tg_zero16);
        char* A = smem + buf * BUF;
        char* W = A + ABYTES;
#pragma unroll
        for (int j = 0; j < AGW; ++j) {
            const uint16_t* src = (kok && gap[j]) ? gap[j] + k0 : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(A + (wave + 4 * j) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < WGW; ++j) {
            const uint16_t* src = (kok && gwp[j]) ? gwp[j] + k0 : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(W + (wave + 4 * j) * 1024), 16, 0, 0);
        }
    };

    f32x4_t acc[4][MI];   // [weight sub-tile j][token sub-tile i]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int h = lane >> 4, c16 = lane & 15;
    // wave (wm, wn): token rows [16 MI wm, +16 MI); weight rows: plain [64 wn, +64); SwiGLU gate [32 wn, +32), up [64 + 32 wn, +32)
    int wrow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wrow[j] = SW ? ((j < 2 ? 0 : 64) + 32 * wn + 16 * (j & 1)) : (64 * wn + 16 * j);

    constexpr int BK = BKC * 8;
    const int KT = (K + BK - 1) / BK;
    if (NST == 3) {
        stage(0, 0);
        if (KT > 1) stage(1, BK);
    } else if (GLDS) {
        stage(0, 0);
    } else {
        gload(0);
        lstore(0);
    }
    if (NST != 3) __syncthreads();     // (with an LDS-DMA in flight the barrier's fence waits vmcnt(0): the tile has landed)
    for (int kt = 0; kt < KT; ++kt) {
        int buf = kt & 1;
        if (NST == 3) {
            buf = kt % 3;
            // this wave's pieces of tile kt have landed once at most the NEWER tile's AGW + WGW DMAs are outstanding
            if (kt + 1 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AGW + WGW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < KT) stage((kt + 2) % 3, (kt + 2) * BK);
        } else if (kt + 1 < KT) {
            if (GLDS) stage(buf ^ 1, (kt + 1) * BK);   // buffer buf^1 was last read in iteration kt-1, behind a barrier
            else gload((kt + 1) * BK);
        }
        const char* A = smem + buf * BUF;
        const char* W = A + ABYTES;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8_t af[MI], wf[4];
#pragma unroll
            for (int i = 0; i < MI; ++i)
                af[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(A + tg_off<BKC>(16 * MI * wm + 16 * i + c16, s * 4 + h)));
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wf[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(W + tg_off<BKC>(wrow[j] + c16, s * 4 + h)));
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[j][i], 0, 0, 0);
        }
        if (!GLDS && kt + 1 < KT) lstore(buf ^ 1);
        if (NST != 3) __syncthreads();
    }

    // ---- epilogue: acc[j][i] lane (h, c16): token row 16 MI wm + 16 i + c16, features (weight rows) wrow[j] + 4 h .. +3 ----
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int r = row0 + 16 * MI * wm + 16 * i + c16;
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

template <int EPI, int MI, int BKC, int NST>
static int launch_tgemm_v(const umoe_tgemm_args* a, int max_n, hipStream_t s) {
    tg_pack gp;
    memset(&gp, 0, sizeof(gp));
    memcpy(gp.g, a->groups, sizeof(umoe_tgroup_t) * a->num_groups);
    constexpr int lds = (NST == 3 ? 3 : 2) * (32 * MI * BKC * 16 + 128 * BKC * 16);
    static bool configured = false;
    if (!configured) {
        UMOE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tgemm_kernel<EPI, MI, BKC, NST>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        configured = true;
    }
    const int ntile = EPI == UMOE_EPI_SWIGLU ? 64 : 128;
    dim3 grid((unsigned)ceil_div(max_n, ntile), (unsigned)ceil_div(a->max_rows, 32 * MI), (unsigned)a->num_groups);
    tgemm_kernel<EPI, MI, BKC, NST><<<grid, 256, lds, s>>>(*a, gp);
    UMOE_LAUNCH_CHECK();
    return 0;
}

// tile choice: 256-token tiles (128 x 64 wave tiles) once there are enough rows to fill the chip with them
static int tgemm_tm(const umoe_tgemm_args* a) {
    static int forced = -1;
    if (forced < 0) {
        const char* v = getenv("UMOE_TGEMM_TM");
        forced = v ? atoi(v) : 0;
    }
    if (forced == 128 || forced == 256) return forced;
    static int minwg = -1;
    if (minwg < 0) {
        const char* v = getenv("UMOE_TGEMM_MINWG");
        minwg = v ? atoi(v) : 512;
    }
    if (a->max_rows < 1024) return 128;
    int max_n = 0;
    for (int i = 0; i < a->num_groups; ++i) max_n = a->groups[i].n > max_n ? a->groups[i].n : max_n;
    const long wgs = (long)ceil_div(a->max_rows, 256) * ceil_div(max_n, a->epilogue == UMOE_EPI_SWIGLU ? 64 : 128) * a->num_groups;
    return wgs >= minwg ? 256 : 128;     // big tiles only when they still fill the chip (2 workgroups per CU)
}

template <int EPI>
static int launch_tgemm(const umoe_tgemm_args* a, int max_n, hipStream_t s) {
    // staging variant (UMOE_TGEMM_GLDS, experiments): default = LDS-DMA; 256-token tiles with three LDS stages (the DMA of
    // tile t+2 in flight across the barrier), 128-token tiles with two.  Measured at 4800-6240 rows (scripts/kbench.py tiled):
    // register staging 580-650 TFLOP/s, LDS-DMA two stages 640-744, 256-token tiles + three stages 650-804; 128-token tiles
    // with three 32 KiB stages (one workgroup per CU) 380-500; 128-token tiles, K steps of 32, three stages 525-670.
    static int glds = -1;
    if (glds < 0) {
        const char* v = getenv("UMOE_TGEMM_GLDS");
        glds = v ? atoi(v) : 1;
    }
    if (glds == 43) return launch_tgemm_v<EPI, 4, 4, 3>(a, max_n, s);
    if (tgemm_tm(a) == 256) {
        if (glds == 0) return launch_tgemm_v<EPI, 8, 4, 0>(a, max_n, s);
        if (glds == 2) return launch_tgemm_v<EPI, 8, 4, 2>(a, max_n, s);
        return launch_tgemm_v<EPI, 8, 4, 3>(a, max_n, s);
    }
    if (glds == 0) return launch_tgemm_v<EPI, 4, 8, 0>(a, max_n, s);
    if (glds == 3) return launch_tgemm_v<EPI, 4, 8, 3>(a, max_n, s);
    return launch_tgemm_v<EPI, 4, 8, 2>(a, max_n, s);
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
