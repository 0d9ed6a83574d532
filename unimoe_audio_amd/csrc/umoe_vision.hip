// Vision tower pieces (reference utils/UniMoE_Audio_utils.py:756-900 built on the third-party transformers classes
// Qwen2_5_VLVisionBlock / Qwen2_5_VLVisionAttention / Qwen2_5_VLMLP / Qwen2_5_VLPatchMerger).  The projections run on the tiled MFMA
// GEMM (umoe_tiled_gemm, bias / residual epilogues); this file holds what is not a GEMM:
//   umoe_vision_rope   2-D rotary embedding of q and k inside the fused qkv buffer, fp32 arithmetic, one rounding (apply_rotary_pos_emb_vision)
//   umoe_vision_attn   NON-causal attention inside each cu_seqlens segment (windows of 64 patches, or a whole frame), fp32 softmax
//   umoe_swiglu_pair   silu(gate) * up of the biased vision MLP, with the reference's bf16 rounding points
//   umoe_gelu          exact (erf) GELU of the patch merger
// Roofline: none of these matters -- the tower runs once per request on ~10^3 patches; the GEMMs carry its FLOPs.
#include "umoe_common.h"

// qkv [S][3][H][hd] bf16; cos / sin [S][hd] fp32 (two equal halves, like the reference's cat((rot, rot))); in place on q and k
__global__ __launch_bounds__(256) void vision_rope_kernel(uint16_t* __restrict__ qkv, const float* __restrict__ cs, const float* __restrict__ sn,
                                                          int S, int H, int hd) {
    const int half = hd >> 1;
    const long n = (long)S * 2 * H * half;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int d = (int)(i % half);
        const int h = (int)((i / half) % H);
        const int which = (int)((i / ((long)half * H)) % 2);      // 0 q, 1 k
        const long s = i / ((long)half * H * 2);
        uint16_t* p = qkv + ((s * 3 + which) * H + h) * hd;
        const float a = bf2f(p[d]), b = bf2f(p[d + half]);
        // q * cos + rotate_half(q) * sin, rotate_half = cat(-x2, x1)
        const float c0 = cs[s * hd + d], c1 = cs[s * hd + d + half], s0 = sn[s * hd + d], s1 = sn[s * hd + d + half];
        p[d] = f2bf(a * c0 + (-b) * s0);
        p[d + half] = f2bf(b * c1 + a * s1);
    }
}

// one wave per (query token, head); keys = the token's segment [lo, hi).  hd <= 128.
__global__ __launch_bounds__(64) void vision_attn_kernel(const uint16_t* __restrict__ qkv, const int32_t* __restrict__ seg_lo,
                                                         const int32_t* __restrict__ seg_hi, int S, int H, int hd, float scale,
                                                         uint16_t* __restrict__ out) {
    __shared__ float qs[128];
    const int s = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const long row = (long)3 * H * hd;
    const uint16_t* q = qkv + (long)s * row + (long)h * hd;
    for (int d = lane; d < hd; d += 64) qs[d] = bf2f(q[d]) * scale;
    __syncthreads();
    const int lo = seg_lo[s], hi = seg_hi[s];
    float m = -INFINITY, l = 0.f, o0 = 0.f, o1 = 0.f;      // lane owns output dims lane and lane + 64
    for (int k0 = lo; k0 < hi; k0 += 64) {
        const int j = k0 + lane;
        float sc = -INFINITY;
        if (j < hi) {
            const uint16_t* kr = qkv + (long)j * row + (long)(H + h) * hd;
            float acc = 0.f;
            for (int d = 0; d < hd; d += 8) {
                float f[8];
                unpack8(ld16(kr + d), f);
#pragma unroll
                for (int t = 0; t < 8; ++t) acc += qs[d + t] * f[t];
            }
            sc = acc;
        }
        const float mn = fmaxf(m, wave_max(sc));
        const float alpha = (m == -INFINITY) ? 0.f : __expf(m - mn);
        const float p = (j < hi) ? __expf(sc - mn) : 0.f;
        l = l * alpha + wave_sum(p);
        o0 *= alpha;
        o1 *= alpha;
        const int nk = min(64, hi - k0);
        for (int t = 0; t < nk; ++t) {
            const float pt = __shfl(p, t, 64);
            const uint16_t* vr = qkv + (long)(k0 + t) * row + (long)(2 * H + h) * hd;
            if (lane < hd) o0 += pt * bf2f(vr[lane]);
            if (lane + 64 < hd) o1 += pt * bf2f(vr[lane + 64]);
        }
        m = mn;
    }
    const float inv = l > 0.f ? 1.f / l : 0.f;
    uint16_t* o = out + ((long)s * H + h) * hd;
    if (lane < hd) o[lane] = f2bf(o0 * inv);
    if (lane + 64 < hd) o[lane + 64] = f2bf(o1 * inv);
}

// gu [S][2 I] (gate | up, biases already added by the GEMM) -> h [S][ldh]: bf16(bf16(silu(g)) * u); columns [I, ldh) zeroed (K padding)
__global__ __launch_bounds__(256) void swiglu_pair_kernel(const uint16_t* __restrict__ gu, int S, int I, int ldh, uint16_t* __restrict__ h) {
    const long n = (long)S * ldh;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long s = i / ldh;
        const int c = (int)(i % ldh);
        uint16_t v = 0;
        if (c < I) {
            const float g = bf2f(gu[s * 2 * I + c]), u = bf2f(gu[s * 2 * I + I + c]);
            v = f2bf(rbf(g / (1.0f + expf(-g))) * u);
        }
        h[i] = v;
    }
}

__global__ __launch_bounds__(256) void gelu_kernel(uint16_t* __restrict__ x, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = bf2f(x[i]);
        x[i] = f2bf(0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)));
    }
}

extern "C" int umoe_vision_rope(uint16_t* qkv, const float* cos_t, const float* sin_t, int S, int H, int hd, umoe_stream_t stream) {
    UMOE_REQUIRE(qkv && cos_t && sin_t && S >= 0 && H > 0 && hd > 0 && hd % 2 == 0, "umoe_vision_rope: bad argument");
    if (S == 0) return 0;
    const long n = (long)S * 2 * H * (hd / 2);
    vision_rope_kernel<<<dim3((unsigned)min((n + 255) / 256, 4096L)), 256, 0, (hipStream_t)stream>>>(qkv, cos_t, sin_t, S, H, hd);
    UMOE_LAUNCH_CHECK();
    return 0;
}

extern "C" int umoe_vision_attn(const uint16_t* qkv, const int32_t* seg_lo, const int32_t* seg_hi, int S, int H, int hd, float scale,
                                uint16_t* out, umoe_stream_t stream) {
    UMOE_REQUIRE(qkv && seg_lo && seg_hi && out && S >= 0 && H > 0 && H <= 65535 && hd > 0 && hd <= 128 && hd % 8 == 0, "umoe_vision_attn: bad argument (hd <= 128, hd %% 8 == 0)");
    if (S == 0) return 0;
    vision_attn_kernel<<<dim3((unsigned)S, (unsigned)H), 64, 0, (hipStream_t)stream>>>(qkv, seg_lo, seg_hi, S, H, hd, scale, out);
    UMOE_LAUNCH_CHECK();
    return 0;
}

extern "C" int umoe_swiglu_pair(const uint16_t* gu, int S, int I, int ldh, uint16_t* h, umoe_stream_t stream) {
    UMOE_REQUIRE(gu && h && S >= 0 && I > 0 && ldh >= I, "umoe_swiglu_pair: bad argument");
    if (S == 0) return 0;
    const long n = (long)S * ldh;
    swiglu_pair_kernel<<<dim3((unsigned)min((n + 255) / 256, 8192L)), 256, 0, (hipStream_t)stream>>>(gu, S, I, ldh, h);
    UMOE_LAUNCH_CHECK();
    return 0;
}

extern "C" int umoe_gelu(uint16_t* x, long n, umoe_stream_t stream) {
    UMOE_REQUIRE(x && n >= 0, "umoe_gelu: bad argument");
    if (n == 0) return 0;
    gelu_kernel<<<dim3((unsigned)min((n + 255) / 256, 8192L)), 256, 0, (hipStream_t)stream>>>(x, n);
    UMOE_LAUNCH_CHECK();
    return 0;
}
