// Tiled MFMA GEMM for the compute-bound shapes (prefill: 4 800 tokens, training: 6 240 tokens per step).
//
//   Y[rows(g), N] = epilogue( A[rows(g), K] * W_g^T ),  W_g row-major [N][K] bf16
//
// Two kernels behind umoe_tiled_gemm (same arguments, same results up to fp32 summation order):
//  * tgemm_pp_kernel (further down): 256 x 256 tiles, 8 waves in two groups running one barrier apart ("ping-pong"), chosen for
//    launches of >= 1024 rows that fill the chip with such tiles -- 0.85-1.0 PFLOP/s at the prefill / training shapes;
//  * tgemm_kernel: the small-tile kernel described next, for everything else (few rows, 64-80-tile weight gradients).
//
// Small-tile design (MI355X / gfx950):
//  * one 256-thread workgroup per 128 x 128 output tile, 4 waves as 2 (rows) x 2 (columns), each wave a 64 x 64 sub-tile
//    = 4 x 4 v_mfma_f32_16x16x32_bf16 accumulators (64 registers);
//  * tiles are staged by LDS-DMA (global_load_lds_dwordx4: no data registers, no ds_write pass; the swizzle sits on the
//    per-lane source address); >= 1024 rows: 256-token tiles (128 x 64 wave tiles), K steps of 32, THREE LDS stages with the
//    DMA of tile t+2 in flight across the barrier (counted vmcnt + raw s_barrier); fewer rows: 128-token tiles, K steps of 64,
//    two stages; register staging kept as a variant;
//  * LDS image: 128-byte rows (64 bf16), 16-byte chunk c of row r at r*128 + ((c ^ ((r >> 1) & 7)) * 16): the global loads
//    and the LDS writes are contiguous per row (8 lanes x 16 B), the ds_read_b128 operand reads touch every bank once
//    (64-byte rows of the K-32 variants: see tg_swz);
//  * weights are the MFMA A operand, activations the B operand (as in umoe_gemm.hip): a lane ends with 4 consecutive
//    output features of one token -> 8-byte bf16 stores;
//  * ragged groups: gather list / row count / row offset are read on device, workgroups beyond the count exit.
// Roofline: MFMA (2.5 PFLOP/s dense bf16).  Intensity of a 128x128 tile: 64 flop per byte moved from L2.
#include "umoe_common.h"
#include <stdlib.h>
#include <type_traits>
#include <string.h>

int umoe_tiled_gemm_nn_launch(const umoe_tgemm_args* a, int max_n, hipStream_t s);

#define TG_MAXG 12
struct tg_pack { umoe_tgroup_t g[TG_MAXG]; };

// 16-byte chunk c of tile row r, rows of BKC chunks.  An operand read takes ONE chunk column of 16 consecutive rows; rows
// are BKC*16 B = BKC*4 banks, so 16/BKC consecutive rows span the 64 banks and the rows that share banks (every 16/BKC-th)
// must land on different chunk slots -> swizzle by (r / (16/BKC)) % BKC: conflict-free (SQ_LDS_BANK_CONFLICT = 0).
// BKC = 4 (64-byte rows, four rows per 256-byte bank line): ds_read_b128 serves a wave in four groups of 16 lanes that MIX two
// chunk columns -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS) -- i.e. rows 0-3 and 12-15 with
// chunk h next to rows 4-11 with chunk h + 1.  Rows r, r+4, r+8, r+12 share their four banks' quad, so the slots
// {g(0), g(12), 1^g(4), 1^g(8)} and {g(4), g(8), 1^g(0), 1^g(12)} must each be distinct: g(r) = (-(r >> 2)) & 3 (0, 3, 2, 1).
// (The obvious g = (r >> 2) & 3 is 2-way on every read: SQ_LDS_BANK_CONFLICT = 4 cycles per ds_read_b128, measured.)
template <int BKC>
__device__ __forceinline__ int tg_swz(int row) {
    return BKC == 8 ? ((row >> 1) & 7) : ((0 - (row >> 2)) & 3);
}
template <int BKC>
__device__ __forceinline__ int tg_off(int row, int chunk) {
    return row * (BKC * 16) + ((chunk ^ tg_swz<BKC>(row)) << 4);
}

// Epilogue shared by the tile variants.  acc[j][i]: lane (h = lane >> 4, c16 = lane & 15) holds features fbase[j] + 4 h .. + 3 of
// token row rbase + 16 i + c16 (SwiGLU: acc[j] = gate, acc[j + 2] = up of the same features, j < 2).
template <int EPI, int MI>
__device__ __forceinline__ void tg_epilogue(const umoe_tgemm_args& p, const umoe_tgroup_t& g, const f32x4_t (&acc)[4][MI], const int count,
                                            const int roff, const int rbase, const int (&fbase)[4], const int lane) {
    constexpr bool SW = EPI == UMOE_EPI_SWIGLU;
    const int h = lane >> 4, c16 = lane & 15;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int r = rbase + 16 * i + c16;
        if (r >= count) continue;
        const long orow = (long)g.out_row_base + roff + r;
        const int oc = g.out_col_off;
        if (SW) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = fbase[j] + 4 * h;
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
                const int col = fbase[j] + 4 * h;
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
        ap[ps] = p.a + (g.k_compact_a > 0 ? (long)koff * g.k_compact_a + arow * (long)K : arow * (long)p.lda + koff) + g.a_col_off + ch * 8;
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
        wp[ps] = wb + (g.k_compact_w > 0 ? (long)koff * g.k_compact_w + (long)(wok[ps] ? n : 0) * K : (long)(wok[ps] ? n : 0) * g.ldw + koff) + ch * 8;
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
    // source = zero block + integer byte delta (0 for chunks outside the tile): the choice is arithmetic, so every stage issues the
    // same number of DMA instructions whatever the data (a pointer select in front of the intrinsic compiles to two exec-masked
    // DMAs; the counted vmcnt waits stay safe with them -- extra instructions only over-wait -- but they cost issue slots)
    long gad[GLDS ? AGW : 1];
    long gwd[GLDS ? WGW : 1];
    int gch = 0;                                          // this lane's chunk (same for every group: row & swizzle mask repeats)
    const char* zero_b = reinterpret_cast<const char*>(&tg_zero16);
    if (GLDS) {
        const int rl = lane / BKC, slot = lane % BKC;
#pragma unroll
        for (int j = 0; j < AGW; ++j) {
            const int tr = (wave + 4 * j) * RPG + rl;     // tile row
            const int c = slot ^ tg_swz<BKC>(tr);
            gch = c;                                      // RPG * 4 is a multiple of the swizzle period: c does not depend on j
            const int r = row0 + tr;
            gad[j] = 0;
            if (r < count) {
                const long arow = g.rows ? (long)g.rows[roff + r] : (long)(g.a_row_base + roff + r);
                gad[j] = reinterpret_cast<const char*>(p.a + (g.k_compact_a > 0 ? (long)koff * g.k_compact_a + arow * (long)K : arow * (long)p.lda + koff) + g.a_col_off + c * 8) - zero_b;
            }
        }
#pragma unroll
        for (int j = 0; j < WGW; ++j) {
            const int tr = (wave + 4 * j) * RPG + rl;
            const int c = slot ^ tg_swz<BKC>(tr);
            int n;
            const uint16_t* wb = g.w;
            if (SW) {
                n = n0 + (tr & 63);
                if (tr >= 64) wb = g.w2;
            } else {
                n = n0 + tr;
            }
            gwd[j] = n < g.n ? reinterpret_cast<const char*>(wb + (g.k_compact_w > 0 ? (long)koff * g.k_compact_w + (long)n * K : (long)n * g.ldw + koff) + c * 8) - zero_b : 0;
        }
    }
    auto stage = [&](int buf, int k0) {
        const long live = (k0 + gch * 8 < K) ? -1L : 0L;
        char* A = smem + buf * BUF;
        char* W = A + ABYTES;
#pragma unroll
        for (int j = 0; j < AGW; ++j) {
            const long d = gad[j] == 0 ? 0 : ((gad[j] + 2L * k0) & live);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(zero_b + d),
                                             (__attribute__((address_space(3))) void*)(A + (wave + 4 * j) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < WGW; ++j) {
            const long d = gwd[j] == 0 ? 0 : ((gwd[j] + 2L * k0) & live);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(zero_b + d),
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
    int fbase[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) fbase[j] = n0 + (SW ? 32 * wn + 16 * (j & 1) : wrow[j]);
    tg_epilogue<EPI, MI>(p, g, acc, count, roff, row0 + 16 * MI * wm, fbase, lane);
}

// Epilogue through LDS for the 256 x 256 variant (bf16 outputs): a lane's accumulators are 4 features of one token, so direct
// stores put 32 contiguous bytes per token row (a quarter of a 128-byte line per instruction).  Each wave parks its 128-token x
// 64-feature tile (bias added, rounded to bf16; SwiGLU applied: 32 features) in a private 16 KiB piece of the now idle
// staging ring, token-major, and reads it back 16 bytes per lane: whole rows of 128 (64) contiguous bytes per store, the
// residual read the same way.  Wave-private, and a wave's LDS operations execute in order: no barrier.
//   tile_store_lds: NJ 16-feature groups per row; val(i, j, q) = bf16 bits of feature 16 j + 4 h + q of token 16 i + c16.
template <int NJ, bool RESID, class F>
__device__ __forceinline__ void tile_store_lds(F val, uint16_t* dst, const uint16_t* resid, const long ld, const int ncols, const long orow0,
                                               const int count, const int rbase, const int cbase, const int lane, char* my) {
    constexpr int RB = NJ * 32;         // bytes per staged row
    constexpr int CPR = RB / 16;        // 16-byte chunks per row
    const int h = lane >> 4, c16 = lane & 15;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 16 * i + c16;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const uint32_t y0 = val(i, j, 0), y1 = val(i, j, 1), y2 = val(i, j, 2), y3 = val(i, j, 3);
            const int chunk = 2 * j + (h >> 1);
            *reinterpret_cast<uint2*>(my + row * RB + ((chunk ^ (row & (CPR - 1))) << 4) + (h & 1) * 8) = make_uint2(y0 | (y1 << 16), y2 | (y3 << 16));
        }
    }
    __builtin_amdgcn_wave_barrier();
    constexpr int RPI = 64 / CPR;       // rows per wave-instruction
    const int rl = lane / CPR, c = lane % CPR;
    const int col = cbase + c * 8;
#pragma unroll 4
    for (int it = 0; it < 128 / RPI; ++it) {
        const int row = it * RPI + rl;
        const uint4 d = *reinterpret_cast<const uint4*>(my + row * RB + ((c ^ (row & (CPR - 1))) << 4));
        const int r = rbase + row;
        if (r >= count || col >= ncols) continue;
        uint16_t* o = dst + (orow0 + r) * ld + col;
        if (col + 8 <= ncols) {
            uint4 v = d;
            if (RESID) {
                const uint4 rv = *reinterpret_cast<const uint4*>(resid + (orow0 + r) * ld + col);
                float a[8], b[8];
                unpack8(d, a);
                unpack8(rv, b);
#pragma unroll
                for (int q = 0; q < 8; ++q) a[q] = b[q] + a[q];
                v = pack8(a);
            }
            *reinterpret_cast<uint4*>(o) = v;
        } else {
            const uint16_t* dv = reinterpret_cast<const uint16_t*>(&d);
            for (int q = 0; q < 8 && col + q < ncols; ++q) {
                float x = bf2f(dv[q]);
                if (RESID) x = bf2f(resid[(orow0 + r) * ld + col + q]) + x;
                o[q] = f2bf(x);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

template <int EPI>
__device__ __forceinline__ void tg_epilogue_lds(const umoe_tgemm_args& p, const umoe_tgroup_t& g, const f32x4_t (&acc)[4][8], const int count,
                                                const int roff, const int rbase, const int cbase, const int lane, char* my) {
    constexpr bool SW = EPI == UMOE_EPI_SWIGLU;
    const int h = lane >> 4;
    const long orow0 = (long)g.out_row_base + roff;
    uint16_t* out = reinterpret_cast<uint16_t*>(p.out) + g.out_col_off;
    if constexpr (SW) {
        auto act = [&](int i, int j, int q) -> uint32_t {
            const float gt = rbf(acc[j][i][q]);
            const float up = rbf(acc[j + 2][i][q]);
            const float si = rbf(gt / (1.0f + expf(-gt)));
            return f2bf(si * up);
        };
        tile_store_lds<2, false>(act, out, nullptr, p.ldo, g.n, orow0, count, rbase, cbase, lane, my);
        if (p.aux_out) {   // pre-activations for the backward pass: gate at [row][col], up at [row][n + col]
            tile_store_lds<2, false>([&](int i, int j, int q) -> uint32_t { return f2bf(acc[j][i][q]); }, p.aux_out, nullptr, p.ld_aux, g.n,
                                     orow0, count, rbase, cbase, lane, my);
            tile_store_lds<2, false>([&](int i, int j, int q) -> uint32_t { return f2bf(acc[j + 2][i][q]); }, p.aux_out + g.n, nullptr, p.ld_aux,
                                     g.n, orow0, count, rbase, cbase, lane, my);
        }
    } else {
        auto lin = [&](int i, int j, int q) -> uint32_t {
            const int col = cbase + 16 * j + 4 * h + q;
            return f2bf(acc[j][i][q] + ((g.bias && col < g.n) ? g.bias[col] : 0.f));
        };
        tile_store_lds<4, EPI == UMOE_EPI_BF16_RESID>(lin, out, p.resid + g.out_col_off, p.ldo, g.n, orow0, count, rbase, cbase, lane, my);
    }
}

// ------------------------------------------------------------------------------------ 256 x 256 tiles, two wave groups in ping-pong
// 512 threads = 8 waves as 2 (token halves, wr) x 4 (feature quarters, wc); a wave owns 128 tokens x 64 features = 4 x 8
// accumulators (128 registers).  Each SIMD carries one wave of each group; the groups run ONE BARRIER APART, so while group 0
// issues its MFMAs group 1 reads its operands / issues the next LDS-DMA, and vice versa -- the matrix pipe of every SIMD
// always has a wave in its MFMA segment (MI355X_MICROARCH.md, LDS section: one wave per SIMD cannot hide its own ds_reads).
//   K runs in tiles of 32 (one MFMA k-step); LDS = ring of NS = 4 tiles x [W unit 256 rows x 64 B | token unit 256 rows x 64 B] =
//   128 KiB; a unit is staged by all 8 waves (2 global_load_lds_dwordx4 each) and the DMA runs NS - 1 = THREE tiles ahead
//   (the comments below use NS = 4; NS = 5 is the measured-slower experiment UMOE_TGEMM_RING=5).
//   tile v: L: read W(v) (4 fragments) + tokens(v) (8); stage W(v+3), tokens(v+3); s_waitcnt vmcnt(8), lgkmcnt(0) | barrier | M: 32 MFMA | barrier
//   (rounds 1-2: two phases per tile, 16 MFMAs each: twice the barriers per MFMA)
//   RAW: a wave's vmcnt(8) in front of the barrier that ends L(v) retires ITS pieces of tile v+1 (the 8 newer DMAs are W/tokens of v+2,
//        v+3); the reads of tile v+1 start in L(v+1), behind a barrier that both groups have passed after their waits.
//   WAR: the partner group stages tile v+3 into the slot of tile v-1 in ITS L(v), one barrier behind this group's L(v-1) -- whose reads
//        are complete by then: lgkmcnt(0) stands in front of the barrier that ends every load segment.
//   Tiles beyond K are staged from the zero block, so the vmcnt arithmetic is the same in every iteration.
template <int EPI, int PRIO, int NS>
__global__ __launch_bounds__(512, 2) void tgemm_pp_kernel(const umoe_tgemm_args p, const tg_pack gp, const int epi_lds_mask, const int nx, const int ny, const int nz, const int ragged_order,
                                                           const unsigned total_wgs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef UMOE_PP_STAMPS
    const unsigned long long t_entry = clock64(), w_entry = wall_clock64();
#endif
    constexpr int UNIT = 256 * 64, SLOT = 2 * UNIT;
    constexpr int AH = NS - 1;                 // the DMA runs AH tiles ahead (NS ring slots: 4 = 128 KiB, 5 = all 160 KiB of the CU)
    constexpr bool SW = EPI == UMOE_EPI_SWIGLU;
    constexpr int NTILE = SW ? 128 : 256;
    // XCD-aware tile order.  Workgroups are dealt to the 8 XCDs round-robin by linear id (1-D launch), so the workgroups with
    // lin % 8 == x share an L2.  Two bijective maps from (x, lin / 8) to tiles:
    //  * static groups (every tile is live; dense layers, shared experts, the codec head): XCD x takes a CONTIGUOUS range of the
    //    row-major tile order (q + 1 tiles for the first nwg % 8 XCDs, q for the rest) -- balanced to one tile, which decides a
    //    launch of ~one workgroup per CU, and an XCD pulls ~1/8 of the token rows through its L2;
    //  * ragged groups (live row tiles known only on the device): XCD x takes the row tiles RR = x, x + 8, ... (RR runs over
    //    groups and token tiles; the grid is padded to a multiple of 8 row tiles), sweeping the nx column tiles of a row tile
    //    before the next -- every XCD gets row tiles of EVERY expert (contiguous ranges would put one expert on one XCD
    //    and let the largest expert set the time; measured in the training step: 228 -> 201 us per launch).
    // PERSISTENT tiles (launches whose tiles are all live: grid = one workgroup per CU, total_wgs > gridDim.x): a workgroup walks the
    // tiles lin = blockIdx.x, + gridDim.x, ... (gridDim.x is a multiple of 8, so lin % 8 stays this workgroup's XCD and the maps below
    // hold).  The epilogue's stores are not waited for: they drain while the next tile's first operand tiles are already on their way.
    for (unsigned lin_it = blockIdx.x; lin_it < total_wgs; lin_it += gridDim.x) {
    if (lin_it != blockIdx.x) __syncthreads();       // (the previous tile's LDS epilogue scratch is the ring the DMA is about to fill)
    int bx, by, bz;
    {
        const unsigned lin = lin_it, xcd = lin & 7, seq = lin >> 3;
        if (ragged_order) {
            const unsigned rr = seq / (unsigned)nx;
            bx = (int)(seq - rr * (unsigned)nx);
            const unsigned RR = rr * 8 + xcd;
            if (RR >= (unsigned)(ny * nz)) continue;       // padding row tiles
            bz = (int)(RR / (unsigned)ny);
            by = (int)(RR - (unsigned)bz * (unsigned)ny);
        } else {
            const unsigned nwg = total_wgs, q = nwg >> 3, r = nwg & 7;
            const unsigned id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + seq;
            bx = (int)(id % (unsigned)nx);
            const unsigned RR = id / (unsigned)nx;
            bz = (int)(RR / (unsigned)ny);
            by = (int)(RR - (unsigned)bz * (unsigned)ny);
        }
    }
    const umoe_tgroup_t g = gp.g[bz];
    const int count = g.count ? *g.count : g.static_count;
    const int roff = g.row_off ? *g.row_off : 0;
    const int row0 = by * 256;
    if (row0 >= count) continue;
    const int n0 = bx * NTILE;
    if (n0 >= g.n) continue;
    const int koff = g.k_off ? *g.k_off : 0;
    const int K = g.k_count ? ((*g.k_count + 7) & ~7) : g.k;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- LDS-DMA sources: this wave stages row groups `wave` and `wave + 8` (16 rows each) of every unit ----
    const int rl = lane >> 2, slot = lane & 3;
    const int gch = slot ^ tg_swz<4>(rl);           // source chunk: the swizzle sits on the global address (row groups of 16: same for both)
    // Source address = zero block + byte delta, delta = 0 for rows outside the tile: the choice between a tile chunk and the
    // zero block is ARITHMETIC on an integer.  (A select between two pointers in front of the intrinsic is turned into two
    // exec-masked DMA instructions by hipcc -- the number of VMEM operations per stage would then depend on the data and the
    // counted vmcnt waits below would retire the wrong tile.)
    const char* zero = reinterpret_cast<const char*>(&tg_zero16);
    long tdel[2], wdel[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int tr = (wave + 8 * q) * 16 + rl;    // tile row 0..255
        const int r = row0 + tr;
        tdel[q] = 0;
        if (r < count) {
            const long arow = g.rows ? (long)g.rows[roff + r] : (long)(g.a_row_base + roff + r);
            tdel[q] = reinterpret_cast<const char*>(p.a + (g.k_compact_a > 0 ? (long)koff * g.k_compact_a + arow * (long)K : arow * (long)p.lda + koff) + g.a_col_off + gch * 8) - zero;
        }
        int n;
        const uint16_t* wb = g.w;
        if (SW) {
            n = n0 + (tr & 127);
            if (tr >= 128) wb = g.w2;
        } else {
            n = n0 + tr;
        }
        wdel[q] = n < g.n ? reinterpret_cast<const char*>(wb + (g.k_compact_w > 0 ? (long)koff * g.k_compact_w + (long)n * K : (long)n * g.ldw + koff) + gch * 8) - zero : 0;
    }
    auto stage = [&](const long (&del)[2], const int unit_off, const int tile) {
        const int k0 = tile * 32;
        const long live = (k0 + gch * 8 < K) ? -1L : 0L;      // K tail and the tiles staged past the end read zeros
        char* base = smem + (tile % NS) * SLOT + unit_off;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const long d = del[q] == 0 ? 0 : ((del[q] + 2L * k0) & live);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(zero + d),
                                             (__attribute__((address_space(3))) void*)(base + (wave + 8 * q) * 1024), 16, 0, 0);
        }
    };

    // Steady state (tiles that lie entirely inside K): running source pointers, one 64-bit add per DMA.  Every VALU
    // instruction of a wave in its load segment takes issue cycles from the MFMAs of the other group's wave on the same SIMD
    // (measured with the address arithmetic above in the loop: 16 MFMAs took 360 cycles instead of 256).
    const char* tptr[2];
    const char* wptr[2];
    long tinc[2], winc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        tinc[q] = tdel[q] ? 64 : 0;
        winc[q] = wdel[q] ? 64 : 0;
        tptr[q] = zero + tdel[q] + AH * tinc[q];     // the prologue stages tiles 0..AH-1
        wptr[q] = zero + wdel[q] + AH * winc[q];
    }
    auto stage_run = [&](const char* (&ptr)[2], const long (&inc)[2], const int unit_off, const int slot_off) {
        char* base = smem + slot_off + unit_off;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ptr[q],
                                             (__attribute__((address_space(3))) void*)(base + (wave + 8 * q) * 1024), 16, 0, 0);
            ptr[q] += inc[q];
        }
    };

    f32x4_t acc[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int h = lane >> 4, c16 = lane & 15;
    int wrow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wrow[j] = SW ? ((j < 2 ? 0 : 128) + 32 * wc + 16 * (j & 1)) : (64 * wc + 16 * j);
    // operand reads: ONE base register per unit, the fragments at compile-time offsets (rows step by 16: the swizzle term
    // only depends on c16) -> ds_read_b128 with immediate offsets
    const int wbase = tg_off<4>(wrow[0] + c16, h);
    const int tbase = UNIT + tg_off<4>(128 * wr + c16, h);
    auto wfo = [](int j) { return SW ? ((j & 1) * 1024 + (j >> 1) * 8192) : j * 1024; };

    const int KT = (K + 31) >> 5;
    // prologue: tiles 0..AH-1 in flight, tile 0 landed and visible
#pragma unroll
    for (int u = 0; u < AH; ++u) {
        stage(wdel, 0, u);
        stage(tdel, UNIT, u);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (AH - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();      // group 1 runs one barrier behind group 0 from here on
    if (PRIO == 2 && wr == 1) __builtin_amdgcn_s_setprio(1);   // static priority for the younger half (it loses every arbitration by age)
#ifdef UMOE_PP_STAMPS
    unsigned long long st[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) st[k] = 0;
    st[10] = clock64();          // loop start
#define PP_ST(k) if (v == 8) st[k] = clock64()
#else
#define PP_ST(k)
#endif
    // ONE load segment and ONE MFMA segment of 32 per K tile (rounds 1-2: two of 16 with a barrier pair each): half as many barriers per
    // MFMA, 5-7 % on every shape (scripts/gemm_ab.py).  The fragment reads are waited for IN FRONT of the barrier that ends the load
    // segment (lgkmcnt(0)): the partner group restages the slot of tile v - 1 in its next load segment, one barrier behind this group's
    // last read of it, so those reads have to be complete when anybody passes that barrier (WAR with one phase of distance; RAW: the
    // counted vmcnt in front of the same barrier retires every wave's pieces of tile v + 1, which is read one phase later).
    auto tile_step = [&](const int v, auto steady_tag) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        const int so = (v % NS) * SLOT;              // ring slot of tile v (read); tile v + AH goes to slot (v + AH) % NS
        const int sn = ((v + AH) % NS) * SLOT;
        const char* Wb = smem + so + wbase;
        const char* Tb = smem + so + tbase;
        bf16x8_t wf[4], af[8];
        PP_ST(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Wb + wfo(j)));
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Tb + i * 1024));
        if (STEADY) { stage_run(wptr, winc, 0, sn); stage_run(tptr, tinc, UNIT, sn); }
        else { stage(wdel, 0, v + AH); stage(tdel, UNIT, v + AH); }
        PP_ST(1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (AH - 1)) : "memory");    // all but the AH - 1 newest tiles have landed: tile v + 1 is in
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        PP_ST(2);
        __builtin_amdgcn_s_barrier();
        PP_ST(3);
        if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[j][i], 0, 0, 0);
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        PP_ST(8);
        __builtin_amdgcn_s_barrier();
        PP_ST(9);
    };
    // tiles v + 3 <= KT - 2 lie entirely inside K (only the last tile can hold the K tail): running pointers; the last four
    // iterations stage the tail tile and the zero tiles with the arithmetic addresses
    const int v_steady = KT - (AH + 1) > 0 ? KT - (AH + 1) : 0;
    int v = 0;
    for (; v < v_steady; ++v) tile_step(v, std::true_type{});
    for (; v < KT; ++v) tile_step(v, std::false_type{});
#ifdef UMOE_PP_STAMPS
    st[11] = clock64();          // loop end
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-tile DMAs of the last iterations must not outlive the workgroup
    if (wr == 0) __builtin_amdgcn_s_barrier();      // every wave has now executed the same number of barriers
    __builtin_amdgcn_s_barrier();                   // ... and no DMA of any wave is still on its way into the ring

    if constexpr (EPI == UMOE_EPI_BF16 || EPI == UMOE_EPI_BF16_RESID || EPI == UMOE_EPI_SWIGLU) {
        const bool al16 = ((p.ldo | g.out_col_off) & 7) == 0 && (reinterpret_cast<size_t>(p.out) & 15) == 0 &&
                          (EPI != UMOE_EPI_BF16_RESID || (reinterpret_cast<size_t>(p.resid) & 15) == 0);
        const bool aux16 = !SW || !p.aux_out || (((p.ld_aux | g.n) & 7) == 0 && (reinterpret_cast<size_t>(p.aux_out) & 15) == 0);
        if (al16 && aux16 && ((epi_lds_mask >> (SW ? 2 : (EPI == UMOE_EPI_BF16_RESID ? 1 : 0))) & 1)) {     // 16-byte aligned rows; the aux stores keep the direct path
            tg_epilogue_lds<EPI>(p, g, acc, count, roff, row0 + 128 * wr, n0 + (SW ? 32 : 64) * wc, lane, smem + wave * 16384);
#ifdef UMOE_PP_STAMPS
            if (p.aux_out && !SW && blockIdx.x == 0 && lane == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                unsigned long long* o = reinterpret_cast<unsigned long long*>(p.aux_out) + wave * 16;
#pragma unroll
                for (int k = 0; k < 12; ++k) o[k] = st[k];
                o[12] = t_entry; o[13] = clock64(); o[14] = w_entry; o[15] = wall_clock64();
            }
#endif
            continue;
        }
    }
    int fbase[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) fbase[j] = n0 + (SW ? 32 * wc + 16 * (j & 1) : wrow[j]);
    tg_epilogue<EPI, 8>(p, g, acc, count, roff, row0 + 128 * wr, fbase, lane);
    }   // persistent tile loop
}

template <int EPI, int PRIO, int NS>
static int launch_tgemm_pp_v(const umoe_tgemm_args* a, int max_n, hipStream_t s) {
    tg_pack gp;
    memset(&gp, 0, sizeof(gp));
    memcpy(gp.g, a->groups, sizeof(umoe_tgroup_t) * a->num_groups);
    constexpr int lds = NS * 2 * 256 * 64;
    static bool configured = false;
    if (!configured) {
        UMOE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tgemm_pp_kernel<EPI, PRIO, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        configured = true;
    }
    const int ntile = EPI == UMOE_EPI_SWIGLU ? 128 : 256;
    const int nx = ceil_div(max_n, ntile), ny = ceil_div(a->max_rows, 256), nz = a->num_groups;
    int ragged = 0;
    // (a per-group contraction window read on the device -- the weight-gradient products over each expert's token slots -- makes the
    //  groups as unequal as a row count does: with the static order XCD e got exactly expert e's 88 tiles and the largest expert set the
    //  time of the launch, 540 us in the training step against 283 us for eight equal experts, scripts/wgrad_bench.py)
    for (int i = 0; i < a->num_groups; ++i) ragged |= a->groups[i].count != nullptr || (a->groups[i].k_count != nullptr && a->num_groups > 1);
    const long nwg = ragged ? (long)nx * (((long)ny * nz + 7) & ~7L) : (long)nx * ny * nz;
    UMOE_REQUIRE(nwg < (1L << 31), "umoe_tiled_gemm: too many tiles (%ld)", nwg);
    // persistent tiles where every tile is live (no device-side row counts: empty tiles would unbalance a static walk, the hardware
    // dispatcher balances those launches): one workgroup per CU
    int rows_on_device = 0;
    for (int i = 0; i < a->num_groups; ++i) rows_on_device |= a->groups[i].count != nullptr;
    // (a persistent walk -- fewer workgroups than tiles -- was measured and is not offered: training step 254-259 vs 248-249 ms, the hardware
    //  dispatcher's dynamic placement of one-tile workgroups beats a static walk; codec head, 1225 tiles: 331 vs 329 us)
    (void)rows_on_device;
    dim3 grid((unsigned)nwg);
    // which bf16 epilogues go through LDS (bit 0 plain, 1 residual, 2 SwiGLU).  Measured at 6240 rows (scripts/kbench.py tiled):
    // residual 621 -> 833 TFLOP/s (the residual is read in whole rows too), SwiGLU +2 %, plain -5 % (stays direct)
    static int mask = -1;
    if (mask < 0) {
        const char* v = getenv("UMOE_TGEMM_EPI_LDS");
        mask = v ? atoi(v) : 6;
    }
    tgemm_pp_kernel<EPI, PRIO, NS><<<grid, 512, lds, s>>>(*a, gp, mask, nx, ny, nz, ragged, (unsigned)nwg);
    UMOE_LAUNCH_CHECK();
    return 0;
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
static int launch_tgemm_pp(const umoe_tgemm_args* a, int max_n, hipStream_t s) {
    // s_setprio around every MFMA segment, LDS ring of 4 tiles (128 KiB, the DMA three tiles ahead).  Measured and not offered: no
    // priority or a static one for the younger group (both -20 %), a 5-slot ring (0-4 % slower on every shape and in the training step)
    return launch_tgemm_pp_v<EPI, 1, 4>(a, max_n, s);
}

// 256 x 256 ping-pong tiles once they fill the chip (one workgroup per CU)
static bool tgemm_pp_pays(const umoe_tgemm_args* a, int max_n) {
    static int minwg = -1;
    if (minwg < 0) {
        const char* v = getenv("UMOE_TGEMM_PP_MINWG");
        minwg = v ? atoi(v) : 128;
    }
    if (a->max_rows < 1024) return false;
    const long wgs = (long)ceil_div(a->max_rows, 256) * ceil_div(max_n, a->epilogue == UMOE_EPI_SWIGLU ? 128 : 256) * a->num_groups;
    return wgs >= minwg;
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
    static int pp = -1;
    if (pp < 0) {
        const char* v = getenv("UMOE_TGEMM_PP");
        pp = v ? atoi(v) : 2;
    }
    if (pp == 1 || (pp == 2 && tgemm_pp_pays(a, max_n))) return launch_tgemm_pp<EPI>(a, max_n, s);
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
        if (g.w_kmajor) { UMOE_REQUIRE(g.w && g.n > 0 && g.k > 0, "umoe_tiled_gemm: group %d: need w, n, k", i); if (g.n > max_n) max_n = g.n; continue; }
        UMOE_REQUIRE(g.w && g.n > 0 && (g.k_count || (g.k > 0 && g.k % 8 == 0 && g.ldw >= g.k)) && g.ldw % 8 == 0 && (g.a_col_off & 7) == 0,
                     "umoe_tiled_gemm: group %d: need w, n > 0, k %% 8 == 0, ldw %% 8 == 0 (n=%d k=%d ldw=%d)", i, g.n, g.k, g.ldw);
        UMOE_REQUIRE(a->epilogue != UMOE_EPI_SWIGLU || g.w2, "umoe_tiled_gemm: SwiGLU needs w2 (up_proj)");
        if (g.n > max_n) max_n = g.n;
    }
    hipStream_t s = (hipStream_t)stream;
#ifdef UMOE_TGTIME
    // diagnostics build (make tg: libumoe_hip_tg.so) with UMOE_TGEMM_TIME=1: every call is timed on its own (events + synchronise) and
    // printed with its shape
    static int timed = -1;
    if (timed < 0) timed = getenv("UMOE_TGEMM_TIME") ? 1 : 0;
    static thread_local bool inside = false;
    if (timed && !inside) {
        inside = true;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, s);
        const int rc = umoe_tiled_gemm(a, stream);
        (void)hipEventRecord(e1, s);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        double flop = 0;
        long rows_total = 0;
        for (int i = 0; i < a->num_groups; ++i) {
            const umoe_tgroup_t& g = a->groups[i];
            int cnt = g.static_count, kk = g.k;
            if (g.count) (void)hipMemcpy(&cnt, g.count, 4, hipMemcpyDeviceToHost);
            if (g.k_count) (void)hipMemcpy(&kk, g.k_count, 4, hipMemcpyDeviceToHost);
            rows_total += cnt;
            flop += 2.0 * cnt * (double)g.n * kk * (a->epilogue == UMOE_EPI_SWIGLU ? 2 : 1);
        }
        const umoe_tgroup_t& g0 = a->groups[0];
        fprintf(stderr, "TG epi=%d groups=%d max_rows=%d rows=%ld n=%d k=%d ragged=%d kwin=%d gather=%d pp=%d kmaj=%d us=%.1f tflops=%.0f\n", a->epilogue, a->num_groups,
                a->max_rows, rows_total, max_n, g0.k, g0.count != nullptr, g0.k_count != nullptr, g0.rows != nullptr, (int)tgemm_pp_pays(a, max_n), g0.w_kmajor, ms * 1e3,
                flop / (ms * 1e-3) * 1e-12);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        inside = false;
        return rc;
    }
#endif
    if (a->groups[0].w_kmajor) return umoe_tiled_gemm_nn_launch(a, max_n, s);      // (umoe_tgemm_tn.hip)
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
