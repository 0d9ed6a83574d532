// Probe (not part of the library): semantics of ds_read_b64_tr_b16 on gfx950, as used by the prefill attention kernel.
// Expectation (cdna_hip_programming.md T10): per 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3 of a
// 4 x 16 block of 16-bit elements; lane i receives column i of the 4 rows (row q in element q).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const uint16_t* __restrict__ src, uint16_t* dst) {
    __shared__ __attribute__((aligned(16))) uint16_t tile[64 * 128];
    for (int i = threadIdx.x; i < 64 * 128; i += 64) tile[i] = src[i];
    __syncthreads();
    const int l = threadIdx.x, h = l >> 4, q = (l & 15) >> 2, p = l & 3;
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(tile + (4 * h + q) * 128 + 32 + 4 * p));
    for (int e = 0; e < 4; ++e) dst[l * 4 + e] = (uint16_t)r[e];
}
int main() {
    uint16_t h_src[64 * 128], h_dst[256];
    for (int r = 0; r < 64; ++r)
        for (int c = 0; c < 128; ++c) h_src[r * 128 + c] = (uint16_t)(r * 128 + c);
    uint16_t *d_src, *d_dst;
    hipMalloc(&d_src, sizeof(h_src));
    hipMalloc(&d_dst, sizeof(h_dst));
    hipMemcpy(d_src, h_src, sizeof(h_src), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d_src, d_dst);
    hipMemcpy(h_dst, d_dst, sizeof(h_dst), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) {
            const int hgrp = l >> 4, i = l & 15;
            const int expect = (4 * hgrp + e) * 128 + 32 + i;   // row 4h+e, column 32+i
            if (h_dst[l * 4 + e] != expect) {
                if (bad < 8) printf("lane %d elem %d: got %d (row %d col %d) expected %d\n", l, e, h_dst[l * 4 + e], h_dst[l * 4 + e] / 128, h_dst[l * 4 + e] % 128, expect);
                ++bad;
            }
        }
    printf(bad ? "tr_read probe: %d mismatches\n" : "tr_read probe: OK (lane i gets column i of rows 4h..4h+3)\n", bad);
    return bad != 0;
}
