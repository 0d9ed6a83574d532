// Probe (not part of the library): buffer_load_dwordx4 ... lds with a per-lane offset beyond num_records -- does the LDS-DMA write ZEROS for
// the out-of-range lanes (what a tile loader needs for rows behind the end), and does the scalar offset advance all lanes alike?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void k(const uint32_t* src, uint32_t* dst, int soff) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 1024; i += 64) reinterpret_cast<uint32_t*>(smem)[i] = 0xdeadbeefu;
    __syncthreads();
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(src), 0, 4096, 0x00020000);      // 4096 bytes of records
    const int voff = (threadIdx.x & 1) ? (int)0x7ffffff0 : (int)(threadIdx.x * 16);                                // odd lanes out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)smem, 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) dst[i] = reinterpret_cast<uint32_t*>(smem)[i];
}
int main() {
    uint32_t h[2048], o[256];
    for (int i = 0; i < 2048; ++i) h[i] = i;
    uint32_t *d, *r;
    hipMalloc(&d, sizeof(h));
    hipMalloc(&r, sizeof(o));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64, 4096>>>(d, r, 64);
    hipMemcpy(o, r, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) {
            const uint32_t want = (l & 1) ? 0u : (uint32_t)(l * 4 + 16 + e);      // soffset 64 bytes = 16 words
            if (o[l * 4 + e] != want) { if (bad < 6) printf("lane %d elem %d: got %u want %u\n", l, e, o[l * 4 + e], want); ++bad; }
        }
    printf(bad ? "buffer lds oob probe: %d mismatches\n" : "buffer lds oob probe: OK (out-of-range lanes write zeros, soffset advances every lane)\n", bad);
    return bad != 0;
}
