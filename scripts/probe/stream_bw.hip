// Probe (diagnostic, not product): what a whole-chip weight stream reaches on MI355X by the way the bytes enter a CU.
//   reg:  512-thread workgroups, every wave streams 1-KiB pieces (16 B per lane, non-temporal) into registers, DEPTH pieces in flight
//   dma:  LDS-DMA (global_load_lds_dwordx4), W loader waves per workgroup, DEPTH pieces in flight per wave, nobody reads the LDS
// Each launch reads `bytes` once from a rotating window of a buffer much larger than the Infinity Cache (cold, like a decode layer).
// Build: hipcc -O3 --offload-arch=gfx950 scripts/probe/stream_bw.hip -o scripts/probe/stream_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <functional>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int DEPTH, bool NT>
__global__ __launch_bounds__(512, 1) void reg_stream(const u32x4* __restrict__ src, size_t pieces_per_wave, uint32_t* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t wid = (size_t)blockIdx.x * 8 + wave;
    const u32x4* p = src + wid * pieces_per_wave * 64 + lane;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = 0; i < pieces_per_wave; i += DEPTH) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = NT ? __builtin_nontemporal_load(p + (i + d) * 64) : p[(i + d) * 64];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}
// two register stages like the product kernel: stage B requested before stage A is consumed
template <int DEPTH>
__global__ __launch_bounds__(512, 1) void reg_stream2(const u32x4* __restrict__ src, size_t pieces_per_wave, uint32_t* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t wid = (size_t)blockIdx.x * 8 + wave;
    const u32x4* p = src + wid * pieces_per_wave * 64 + lane;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 a[DEPTH], b[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) a[d] = __builtin_nontemporal_load(p + d * 64);
    for (size_t i = 0; i < pieces_per_wave; i += 2 * DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) b[d] = __builtin_nontemporal_load(p + (i + DEPTH + d) * 64);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc ^= a[d];
        if (i + 2 * DEPTH < pieces_per_wave) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) a[d] = __builtin_nontemporal_load(p + (i + 2 * DEPTH + d) * 64);
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc ^= b[d];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

__device__ __forceinline__ void glds16(const void* g, uint32_t lds_off, bool nt) {
    unsigned keep;
    if (nt) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds_off) : "memory");
    else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds_off) : "memory");
}
// W loader waves per workgroup (blockDim = 64 W), DEPTH pieces in flight per wave (ring of DEPTH KiB per wave in LDS)
template <int DEPTH, bool NT>
__global__ void dma_stream(const char* __restrict__ src, size_t pieces_per_wave, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), W = blockDim.x >> 6;
    const size_t wid = (size_t)blockIdx.x * W + wave;
    const char* p = src + wid * pieces_per_wave * 1024 + lane * 16;
    const uint32_t base = (uint32_t)(uintptr_t)lds + wave * DEPTH * 1024;
    for (size_t i = 0; i < pieces_per_wave; ++i) {
        glds16(p + i * 1024, base + (uint32_t)(i % DEPTH) * 1024, NT);
        if (i + 1 >= DEPTH) {
            if (DEPTH > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(DEPTH - 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lds[threadIdx.x] == 0x7f && sink[1] == 77) sink[0] = 1;
}

static float time_launches(int reps, const std::function<void(int)>& f) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f(i);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a));
        f(r + 3);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

#include <functional>
int main() {
    const size_t win = 304ull << 20;             // bytes per launch (one decode layer's expert weights)
    const int NWIN = 6;                          // rotating windows: 1.8 GB >> 256 MiB Infinity Cache
    char* buf; uint32_t* sink;
    CK(hipMalloc(&buf, win * NWIN)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, win * NWIN)); CK(hipMemset(sink, 0, 64));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    printf("device %s, %d CUs\n", pr.name, pr.multiProcessorCount);
    auto report = [&](const char* name, size_t bytes, float ms) { printf("%-44s %8.2f us  %6.2f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12); fflush(stdout); };
    for (int nwg : {256, 226}) {
        {   const size_t ppw = (win / 1024 / (nwg * 8)) / 28 * 28; const size_t bytes = ppw * 1024 * nwg * 8; char nm[96];
            snprintf(nm, 96, "reg nt depth 14, %d wgs x 8 waves", nwg);
            report(nm, bytes, time_launches(15, [&](int r) { reg_stream<14, true><<<nwg, 512>>>((const u32x4*)(buf + (r % NWIN) * win), ppw, sink); }));
            snprintf(nm, 96, "reg nt depth 28, %d wgs x 8 waves", nwg);
            report(nm, bytes, time_launches(15, [&](int r) { reg_stream<28, true><<<nwg, 512>>>((const u32x4*)(buf + (r % NWIN) * win), ppw, sink); }));
            snprintf(nm, 96, "reg nt 2 stages of 14, %d wgs x 8 waves", nwg);
            report(nm, bytes, time_launches(15, [&](int r) { reg_stream2<14><<<nwg, 512>>>((const u32x4*)(buf + (r % NWIN) * win), ppw, sink); }));
            snprintf(nm, 96, "reg default-policy depth 14, %d wgs", nwg);
            report(nm, bytes, time_launches(15, [&](int r) { reg_stream<14, false><<<nwg, 512>>>((const u32x4*)(buf + (r % NWIN) * win), ppw, sink); }));
        }
    }
    for (int W : {1, 2, 4, 8}) {
        const int nwg = 256;
        const size_t ppw = win / 1024 / (nwg * W); const size_t bytes = ppw * 1024 * nwg * W; char nm[96];
        CK(hipFuncSetAttribute((const void*)&dma_stream<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CK(hipFuncSetAttribute((const void*)&dma_stream<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CK(hipFuncSetAttribute((const void*)&dma_stream<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CK(hipFuncSetAttribute((const void*)&dma_stream<32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        snprintf(nm, 96, "lds-dma nt, %d loader waves/CU, 16 KiB in flight each", W);
        report(nm, bytes, time_launches(15, [&](int r) { dma_stream<16, true><<<nwg, 64 * W, 16 * 1024 * W>>>(buf + (r % NWIN) * win, ppw, sink); }));
        snprintf(nm, 96, "lds-dma default, %d loader waves/CU, 16 KiB each", W);
        report(nm, bytes, time_launches(15, [&](int r) { dma_stream<16, false><<<nwg, 64 * W, 16 * 1024 * W>>>(buf + (r % NWIN) * win, ppw, sink); }));
        if (W <= 4) {
            snprintf(nm, 96, "lds-dma nt, %d loader waves/CU, 32 KiB each", W);
            report(nm, bytes, time_launches(15, [&](int r) { dma_stream<32, true><<<nwg, 64 * W, 32 * 1024 * W>>>(buf + (r % NWIN) * win, ppw, sink); }));
        }
        snprintf(nm, 96, "lds-dma nt, %d loader waves/CU, 8 KiB each", W);
        report(nm, bytes, time_launches(15, [&](int r) { dma_stream<8, true><<<nwg, 64 * W, 8 * 1024 * W>>>(buf + (r % NWIN) * win, ppw, sink); }));
    }
    return 0;
}
