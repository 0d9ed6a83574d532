// Probe: do two HIP streams of one process run concurrently (a bounded spin on stream A released by a kernel on stream B)?
// Decides how the expert-parallel virtual-rank test drives its engines.  hipcc --offload-arch=gfx950 stream_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void spin(volatile unsigned* flag, unsigned* out) {
    const unsigned long long t0 = wall_clock64();
    unsigned ok = 0;
    while (wall_clock64() - t0 < 200000000ull) {   // 2 s at 100 MHz
        if (__hip_atomic_load((unsigned*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) { ok = 1; break; }
        __builtin_amdgcn_s_sleep(8);
    }
    out[0] = ok;
    out[1] = (unsigned)((wall_clock64() - t0) / 100);   // us
}
__global__ void set(unsigned* flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
int main() {
    const int N = 10;
    hipStream_t s[N];
    for (int i = 0; i < N; ++i) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
    unsigned *flag, *out;
    hipMalloc(&flag, 4); hipMalloc(&out, 8);
    for (int k = 1; k < N; ++k) {
        hipMemset(flag, 0, 4); hipDeviceSynchronize();
        spin<<<1, 64, 0, s[0]>>>(flag, out);
        set<<<1, 64, 0, s[k]>>>(flag);
        hipDeviceSynchronize();
        unsigned h[2]; hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
        printf("spin on stream 0, set on stream %d: released=%u after %u us\n", k, h[0], h[1]);
    }
    return 0;
}
