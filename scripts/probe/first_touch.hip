// Probe: what does the FIRST dependent load of a kernel cost right behind a kernel boundary, and why?
// A 160-workgroup consumer (256 threads, every workgroup reads the same 64 KB tile with 16 x 16-byte loads per thread -- the
// staging pattern of the decode GEMMs) runs behind a 16-workgroup producer in one stream, replayed from a hipGraph.  Stamps
// (wall_clock64, 10 ns): entry, after the first tile landed (s_waitcnt vmcnt(0)), after a SECOND tile (another buffer, same
// kernel: warm instruction cache / TLB path, data still untouched) landed.
//   hipcc -O3 --offload-arch=gfx950 first_touch.hip -o first_touch && ./first_touch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void producer(uint4* dst, int n16, unsigned v) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) dst[i] = make_uint4(v, v + 1, v + 2, (unsigned)i);
}

template <int MODE>   // 0 plain global loads, 1 buffer loads sc1, 2 buffer loads (no sc bits), 3 global loads sc1 (asm), 4 plain behind a 1-lane warm-up load, 5 global loads nt
__global__ __launch_bounds__(256) void consumer(const uint4* __restrict__ a, const uint4* __restrict__ b, unsigned long long* stamps, unsigned* sink) {
    const unsigned long long t0 = wall_clock64();
    uint4 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (MODE == 1 || MODE == 2) {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a), 0, 65536, 0x00020000);
            const u4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (threadIdx.x + 256 * k) * 16, 0, MODE == 1 ? 16 : 0);
            v[k] = make_uint4(t[0], t[1], t[2], t[3]);
        } else if (MODE == 3) {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            u4 t;
            const uint4* ptr = a + threadIdx.x + 256 * k;
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(t) : "v"(ptr) : "memory");
            v[k] = make_uint4(t[0], t[1], t[2], t[3]);
        } else if (MODE == 5) {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            const u4 t = __builtin_nontemporal_load(reinterpret_cast<const u4*>(a) + threadIdx.x + 256 * k);
            v[k] = make_uint4(t[0], t[1], t[2], t[3]);
        } else {
            if (MODE == 4 && k == 0) {
                unsigned w = sink[1 + (threadIdx.x & 1)];
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (w == 0x9999u) return;
            }
            v[k] = a[threadIdx.x + 256 * k];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = wall_clock64();
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += v[k].x ^ v[k].w ^ (v[k].y + v[k].z);
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = b[threadIdx.x + 256 * k];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = wall_clock64();
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += v[k].y ^ v[k].x ^ (v[k].z + v[k].w);
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 4 + 0] = t0;
        stamps[blockIdx.x * 4 + 1] = t1;
        stamps[blockIdx.x * 4 + 2] = t2;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    const int WGS = 160, REPS = 40;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    // placements: (0) tiles in their own small allocations, (1) tiles inside one 1 GiB allocation (2 MiB-aligned offsets)
    char* big;
    CK(hipMalloc(&big, 1ull << 30));
    uint4 *small_a, *small_b, *other;
    CK(hipMalloc(&small_a, 65536)); CK(hipMalloc(&small_b, 65536)); CK(hipMalloc(&other, 65536));
    unsigned long long* stamps; unsigned* sink;
    CK(hipMalloc(&stamps, WGS * 4 * 8)); CK(hipMalloc(&sink, 64)); CK(hipMemset(sink, 0, 64));
    CK(hipMemset(big, 1, 1ull << 30)); CK(hipMemset(small_a, 1, 65536)); CK(hipMemset(small_b, 1, 65536));
    for (int place = 0; place < 1; ++place)
        for (int dep = 0; dep < 2; ++dep)
            for (int mode = 0; mode < 6; ++mode) {
                uint4* a = place ? reinterpret_cast<uint4*>(big + (64ull << 20)) : small_a;
                uint4* b = place ? reinterpret_cast<uint4*>(big + (64ull << 20) + 65536) : small_b;
                uint4* pdst = dep ? a : other;      // dep 1: the producer writes the tile the consumer reads first
                hipGraph_t g; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
                for (int r = 0; r < 4; ++r) {
                    producer<<<16, 256, 0, s>>>(pdst, 4096, (unsigned)r);
                    switch (mode) {
                        case 0: consumer<0><<<WGS, 256, 0, s>>>(a, b, stamps, sink); break;
                        case 1: consumer<1><<<WGS, 256, 0, s>>>(a, b, stamps, sink); break;
                        case 2: consumer<2><<<WGS, 256, 0, s>>>(a, b, stamps, sink); break;
                        case 3: consumer<3><<<WGS, 256, 0, s>>>(a, b, stamps, sink); break;
                        case 4: consumer<4><<<WGS, 256, 0, s>>>(a, b, stamps, sink); break;
                        case 5: consumer<5><<<WGS, 256, 0, s>>>(a, b, stamps, sink); break;
                    }
                }
                CK(hipStreamEndCapture(s, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                std::vector<double> first, second, skew;
                for (int r = 0; r < REPS; ++r) {
                    CK(hipGraphLaunch(ge, s));
                    CK(hipStreamSynchronize(s));
                    std::vector<unsigned long long> h(WGS * 4);
                    CK(hipMemcpy(h.data(), stamps, WGS * 4 * 8, hipMemcpyDeviceToHost));
                    std::vector<double> f, sec;
                    unsigned long long e0 = ~0ull, e1 = 0;
                    for (int w = 0; w < WGS; ++w) {
                        f.push_back((h[w * 4 + 1] - h[w * 4 + 0]) / 100.0);
                        sec.push_back((h[w * 4 + 2] - h[w * 4 + 1]) / 100.0);
                        e0 = std::min(e0, h[w * 4 + 0]); e1 = std::max(e1, h[w * 4 + 0]);
                    }
                    if (r >= 4) { first.push_back(med(f)); second.push_back(med(sec)); skew.push_back((e1 - e0) / 100.0); }
                }
                printf("placement %s, producer writes %s, %s loads: first tile %.2f us, second tile (same kernel, untouched buffer) %.2f us, entry skew %.2f us\n",
                       place ? "inside 1 GiB" : "own 64 KB allocs", dep ? "the tile read" : "another buffer", (const char*[]){"plain global", "buffer sc1", "buffer no-sc", "global sc1", "plain after 1-lane warm-up", "global nt"}[mode], med(first), med(second), med(skew));
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            }
    return 0;
}
