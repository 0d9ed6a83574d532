// DAC waveform <-> latent conv stacks (reference utils/UniMoE_Audio_utils.py:112-113,123-124 call the third-party
// descript-audio-codec 1.0.0 `DAC.encode` / `DAC.decode`; the package is absent offline, so the layers are restated from the
// published architecture -- PARITY UNPINNED -- with every dimension a load-time parameter):
//   Snake1d(x) = x + sin(alpha x)^2 / (alpha + 1e-9)  ->  (weight-normalised) Conv1d / ConvTranspose1d, residual units x + y,
//   tanh on the last decoder layer.  fp32 like the reference model.
// One direct kernel per conv flavour, the Snake of the PRECEDING layer fused into the input load (the DAC graph is always
// Snake -> conv), bias, residual add and tanh fused into the store.  Output tile 64 channels x 64 positions per 256-thread
// workgroup, 4 x 4 outputs per thread, input channels in chunks of 8 staged through LDS together with their weight slices.
// Roofline: fp32 VALU / LDS; ~450 GFLOP per 10 s of audio, once per request (not on the per-step path).
#include "umoe_common.h"

namespace {
constexpr int TO = 64, TP = 64, CC = 8;

__device__ __forceinline__ float snake(float v, float a) {
    const float s = __sinf(a * v);
    return v + s * s / (a + 1e-9f);
}

// y[b][co][t] = act( bias[co] + sum_ci sum_k snake?(x[b][ci][t*stride - pad + k*dil]) * w[co][ci][k] ) (+ resid[b][co][t])
__global__ __launch_bounds__(256) void dac_conv1d_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         const float* __restrict__ alpha, const float* __restrict__ resid, int Cin, int L,
                                                         int Cout, int K, int stride, int dil, int pad, int Lout, int act, int XW,
                                                         float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xl = sm;                       // [CC][XW]
    float* wl = sm + CC * XW;             // [CC][K][TO]
    const int t0 = blockIdx.x * TP, co0 = blockIdx.y * TO, b = blockIdx.z, tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const float* xb = x + (size_t)b * Cin * L;
    const int in0 = t0 * stride - pad;    // input position of LDS column 0
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int c0 = 0; c0 < Cin; c0 += CC) {
        for (int i = tid; i < CC * XW; i += 256) {
            const int c = i / XW, p = i % XW, ci = c0 + c, pos = in0 + p;
            float v = 0.f;
            if (ci < Cin && pos >= 0 && pos < L) {
                v = xb[(size_t)ci * L + pos];
                if (alpha) v = snake(v, alpha[ci]);
            }
            xl[i] = v;
        }
        for (int i = tid; i < CC * K * TO; i += 256) {
            const int o = i % TO, k = (i / TO) % K, c = i / (TO * K), ci = c0 + c, co = co0 + o;
            wl[i] = (ci < Cin && co < Cout) ? w[((size_t)co * Cin + ci) * K + k] : 0.f;
        }
        __syncthreads();
        for (int c = 0; c < CC; ++c)
            for (int k = 0; k < K; ++k) {
                const float4 wv = *reinterpret_cast<const float4*>(wl + (c * K + k) * TO + ty * 4);
                const float* xr = xl + c * XW + k * dil;
                float xv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) xv[j] = xr[(tx + 16 * j) * stride];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[0][j] += wv.x * xv[j];
                    acc[1][j] += wv.y * xv[j];
                    acc[2][j] += wv.z * xv[j];
                    acc[3][j] += wv.w * xv[j];
                }
            }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = co0 + ty * 4 + i;
        if (co >= Cout) continue;
        const float bv = bias ? bias[co] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = t0 + tx + 16 * j;
            if (t >= Lout) continue;
            float v = acc[i][j] + bv;
            if (act == 1) v = tanhf(v);
            const size_t o = ((size_t)b * Cout + co) * Lout + t;
            if (resid) v += resid[o];
            y[o] = v;
        }
    }
}

// ConvTranspose1d, gather form: y[b][co][t] = bias[co] + sum_ci sum_{k == (t + pad) mod stride, k < K} snake?(x[b][ci][(t + pad - k) / stride]) * w[ci][co][k]
__global__ __launch_bounds__(256) void dac_convt1d_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          const float* __restrict__ alpha, int Cin, int L, int Cout, int K, int stride, int pad,
                                                          int Lout, int XW, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xl = sm;                       // [CC][XW]
    float* wl = sm + CC * XW;             // [CC][K][TO]
    const int t0 = blockIdx.x * TP, co0 = blockIdx.y * TO, b = blockIdx.z, tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const float* xb = x + (size_t)b * Cin * L;
    const int M = (K + stride - 1) / stride;                 // taps per output position
    const int ibase = (t0 + pad) / stride - (M - 1);         // input index of LDS column 0
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int c0 = 0; c0 < Cin; c0 += CC) {
        for (int i = tid; i < CC * XW; i += 256) {
            const int c = i / XW, p = i % XW, ci = c0 + c, pos = ibase + p;
            float v = 0.f;
            if (ci < Cin && pos >= 0 && pos < L) {
                v = xb[(size_t)ci * L + pos];
                if (alpha) v = snake(v, alpha[ci]);
            }
            xl[i] = v;
        }
        for (int i = tid; i < CC * K * TO; i += 256) {
            const int o = i % TO, k = (i / TO) % K, c = i / (TO * K), ci = c0 + c, co = co0 + o;
            wl[i] = (ci < Cin && co < Cout) ? w[((size_t)ci * Cout + co) * K + k] : 0.f;
        }
        __syncthreads();
        for (int c = 0; c < CC; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tp = t0 + tx + 16 * j + pad;
                const int k0 = tp % stride, i0 = tp / stride - ibase;
                for (int m = 0; m < M; ++m) {
                    const int k = k0 + m * stride;
                    if (k >= K) break;
                    const float xv = xl[c * XW + i0 - m];
                    const float4 wv = *reinterpret_cast<const float4*>(wl + (c * K + k) * TO + ty * 4);
                    acc[0][j] += wv.x * xv;
                    acc[1][j] += wv.y * xv;
                    acc[2][j] += wv.z * xv;
                    acc[3][j] += wv.w * xv;
                }
            }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = co0 + ty * 4 + i;
        if (co >= Cout) continue;
        const float bv = bias ? bias[co] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = t0 + tx + 16 * j;
            if (t < Lout) y[((size_t)b * Cout + co) * Lout + t] = acc[i][j] + bv;
        }
    }
}
// polyphase windowed-sinc resampler (torchaudio.transforms.Resample's default algorithm, reference utils.py:101-110): output sample
// j = frame * n + phase reads K = 2 width + o taps of filter `phase` from the zero-padded input at frame * o - width
__global__ __launch_bounds__(256) void dac_resample_kernel(const float* __restrict__ x, const float* __restrict__ kern, int L, int o, int n,
                                                           int width, int K, int Lout, float* __restrict__ y) {
    const int j = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (j >= Lout) return;
    const int fr = j / n, ph = j % n;
    const float* k = kern + (size_t)ph * K;
    const float* xb = x + (size_t)b * L;
    const int p0 = fr * o - width;
    float acc = 0.f;
    for (int t = 0; t < K; ++t) {
        const int p = p0 + t;
        if (p >= 0 && p < L) acc += k[t] * xb[p];
    }
    y[(size_t)b * Lout + j] = acc;
}
}  // namespace

extern "C" int umoe_dac_resample(const float* x, const float* kern, int B, int L, int o, int n, int width, int Lout, float* y,
                                 umoe_stream_t stream) {
    UMOE_REQUIRE(x && kern && y && B > 0 && L > 0 && o > 0 && n > 0 && width >= 0 && Lout > 0, "umoe_dac_resample: bad argument");
    dac_resample_kernel<<<dim3((unsigned)ceil_div(Lout, 256), (unsigned)B), 256, 0, (hipStream_t)stream>>>(x, kern, L, o, n, width, 2 * width + o, Lout, y);
    UMOE_LAUNCH_CHECK();
    return 0;
}

extern "C" int umoe_dac_conv1d(const float* x, const float* w, const float* bias, const float* snake_alpha, const float* resid, int B,
                               int Cin, int L, int Cout, int K, int stride, int dilation, int pad, int act, float* y, int* Lout_out,
                               umoe_stream_t stream) {
    UMOE_REQUIRE(x && w && y && B > 0 && Cin > 0 && Cout > 0 && K > 0 && stride > 0 && dilation > 0 && pad >= 0 && L > 0,
                 "umoe_dac_conv1d: bad argument");
    const int Lout = (L + 2 * pad - dilation * (K - 1) - 1) / stride + 1;
    UMOE_REQUIRE(Lout > 0, "umoe_dac_conv1d: empty output (L=%d K=%d dilation=%d pad=%d)", L, K, dilation, pad);
    if (Lout_out) *Lout_out = Lout;
    const int XW = (TP - 1) * stride + (K - 1) * dilation + 1;
    const size_t lds = ((size_t)CC * XW + (size_t)CC * K * TO) * sizeof(float);
    UMOE_REQUIRE(lds <= 64 * 1024, "umoe_dac_conv1d: tile needs %zu bytes of LDS", lds);
    dim3 grid((unsigned)ceil_div(Lout, TP), (unsigned)ceil_div(Cout, TO), (unsigned)B);
    dac_conv1d_kernel<<<grid, 256, lds, (hipStream_t)stream>>>(x, w, bias, snake_alpha, resid, Cin, L, Cout, K, stride, dilation, pad, Lout, act, XW, y);
    UMOE_LAUNCH_CHECK();
    return 0;
}

extern "C" int umoe_dac_conv_transpose1d(const float* x, const float* w, const float* bias, const float* snake_alpha, int B, int Cin, int L,
                                         int Cout, int K, int stride, int pad, int out_pad, float* y, int* Lout_out, umoe_stream_t stream) {
    UMOE_REQUIRE(x && w && y && B > 0 && Cin > 0 && Cout > 0 && K > 0 && stride > 0 && pad >= 0 && out_pad >= 0 && L > 0,
                 "umoe_dac_conv_transpose1d: bad argument");
    const int Lout = (L - 1) * stride - 2 * pad + K + out_pad;
    UMOE_REQUIRE(Lout > 0, "umoe_dac_conv_transpose1d: empty output");
    if (Lout_out) *Lout_out = Lout;
    const int M = (K + stride - 1) / stride;
    const int XW = (TP - 1 + stride - 1) / stride + M + 1;
    const size_t lds = ((size_t)CC * XW + (size_t)CC * K * TO) * sizeof(float);
    UMOE_REQUIRE(lds <= 64 * 1024, "umoe_dac_conv_transpose1d: tile needs %zu bytes of LDS", lds);
    dim3 grid((unsigned)ceil_div(Lout, TP), (unsigned)ceil_div(Cout, TO), (unsigned)B);
    dac_convt1d_kernel<<<grid, 256, lds, (hipStream_t)stream>>>(x, w, bias, snake_alpha, Cin, L, Cout, K, stride, pad, Lout, XW, y);
    UMOE_LAUNCH_CHECK();
    return 0;
}
