// Train-step element-wise layers of the tower, fused (train_RL.py:103-122 runs network.py:62-83 in train mode: BatchNorm with batch statistics, the skip
// connection and ReLU are five to six torch launches per site forward + backward; 39 sites make ~230 of a step's ~510 launches and 1.3 ms of its 7 ms):
//   y = relu(bn(x) [+ residual])      bn: batch mean / biased variance over (boards, 8, 8) per channel, eps inside the square root, gamma, beta;
//                                     running_mean / running_var updated with momentum (unbiased variance), exactly torch.nn.BatchNorm2d's train-mode arithmetic
// One workgroup per channel: the channel's n_boards x 64 values (32 KB at batch 128) are read once into registers (re-read from L2 beyond 256 boards), two-pass
// statistics (mean, then the sum of squared deviations — the same formula torch's native kernel uses, no E[x^2] - E[x]^2 cancellation), one pass out.
// Backward (dy' = gy where y > 0): dbeta = sum dy', dgamma = sum dy' * xhat, dx = gamma * invstd * (dy' - dbeta / N - xhat * dgamma / N); the skip connection's
// gradient is dy' itself.  All sums run in a fixed order (lane-strided partials, wave shuffles, then the sixteen wave sums in order): results are reproducible.
// x, y, residual, gy, dx, dres: device [n_boards, channels, 8, 8] f32 NCHW.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "../../include/sigmazero.h"
#include "sz_nn_common.h"

#define BN_NT 1024                                          // threads per workgroup: 16 waves keep enough loads in flight for a latency-bound pass over 32 KB
#define BN_MAXV 4                                           // float4 per thread kept in registers: 1024 threads x 4 x 4 = 16,384 values = 256 boards

__device__ __forceinline__ float bn_block_sum(float v, float* red) {
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();                                                   // `red` may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < BN_NT / 64; w++) t += red[w];                  // the wave sums in order
    return t;
}

// float4 index i of a channel's values -> element offset in the NCHW tensor: board i / 16, positions 4 * (i % 16) ..
__device__ __forceinline__ size_t bn_off(int i, int c, int channels) { return ((size_t)(i >> 4) * channels + c) * 64 + (size_t)(i & 15) * 4; }

__global__ __launch_bounds__(BN_NT) void k_bn_act_fwd(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, float eps,
                                                    const float* __restrict__ residual, float* __restrict__ y, float* __restrict__ save_mean,
                                                    float* __restrict__ save_invstd, int n_boards, int channels) {
    __shared__ float red[BN_NT / 64];
    const int c = blockIdx.x, nv = n_boards * 16;                      // float4 per channel
    float4 v[BN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < BN_MAXV; k++) {
        const int i = threadIdx.x + BN_NT * k;
        if (i < nv) { v[k] = *(const float4*)(x + bn_off(i, c, channels)); s += (v[k].x + v[k].y) + (v[k].z + v[k].w); }
    }
    for (int i = threadIdx.x + BN_NT * BN_MAXV; i < nv; i += BN_NT) { const float4 t = *(const float4*)(x + bn_off(i, c, channels)); s += (t.x + t.y) + (t.z + t.w); }
    const float n = (float)n_boards * 64.f;
    const float mean = bn_block_sum(s, red) / n;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < BN_MAXV; k++) {
        const int i = threadIdx.x + BN_NT * k;
        if (i < nv) { const float a = v[k].x - mean, b = v[k].y - mean, d = v[k].z - mean, e = v[k].w - mean; q += (a * a + b * b) + (d * d + e * e); }
    }
    for (int i = threadIdx.x + BN_NT * BN_MAXV; i < nv; i += BN_NT) {
        const float4 t = *(const float4*)(x + bn_off(i, c, channels));
        const float a = t.x - mean, b = t.y - mean, d = t.z - mean, e = t.w - mean; q += (a * a + b * b) + (d * d + e * e);
    }
    const float ssd = bn_block_sum(q, red);
    const float var = ssd / n, invstd = 1.0f / sqrtf(var + eps);
    if (threadIdx.x == 0) {
        save_mean[c] = mean; save_invstd[c] = invstd;
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (ssd / (n - 1.f));
    }
    const float g = gamma[c], b0 = beta[c];                             // y = ((x - mean) * invstd) * gamma + beta: torch's order of operations
    auto out = [&](int i, const float4 t) {
        const size_t o = bn_off(i, c, channels);
        float4 r = residual ? *(const float4*)(residual + o) : make_float4(0.f, 0.f, 0.f, 0.f);
        r.x = fmaxf(((t.x - mean) * invstd) * g + b0 + r.x, 0.f); r.y = fmaxf(((t.y - mean) * invstd) * g + b0 + r.y, 0.f);
        r.z = fmaxf(((t.z - mean) * invstd) * g + b0 + r.z, 0.f); r.w = fmaxf(((t.w - mean) * invstd) * g + b0 + r.w, 0.f);
        *(float4*)(y + o) = r;
    };
#pragma unroll
    for (int k = 0; k < BN_MAXV; k++) {
        const int i = threadIdx.x + BN_NT * k;
        if (i < nv) out(i, v[k]);
    }
    for (int i = threadIdx.x + BN_NT * BN_MAXV; i < nv; i += BN_NT) out(i, *(const float4*)(x + bn_off(i, c, channels)));
}

__global__ __launch_bounds__(BN_NT) void k_bn_act_bwd(const float* __restrict__ gy, const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gamma,
                                                    const float* __restrict__ save_mean, const float* __restrict__ save_invstd, float* __restrict__ dx,
                                                    float* __restrict__ dres, float* __restrict__ dgamma, float* __restrict__ dbeta, int n_boards, int channels) {
    __shared__ float red[BN_NT / 64];
    const int c = blockIdx.x, nv = n_boards * 16;
    const float mean = save_mean[c], invstd = save_invstd[c];
    float4 d[BN_MAXV], h[BN_MAXV];                                     // dy' and xhat of the cached part
    float sb = 0.f, sg = 0.f;
    auto load = [&](int i, float4& dd, float4& hh) {
        const size_t o = bn_off(i, c, channels);
        const float4 g4 = *(const float4*)(gy + o), x4 = *(const float4*)(x + o), y4 = *(const float4*)(y + o);
        dd = make_float4(y4.x > 0.f ? g4.x : 0.f, y4.y > 0.f ? g4.y : 0.f, y4.z > 0.f ? g4.z : 0.f, y4.w > 0.f ? g4.w : 0.f);
        hh = make_float4((x4.x - mean) * invstd, (x4.y - mean) * invstd, (x4.z - mean) * invstd, (x4.w - mean) * invstd);
    };
    auto add = [&](const float4 dd, const float4 hh) {
        sb += (dd.x + dd.y) + (dd.z + dd.w);
        sg += (dd.x * hh.x + dd.y * hh.y) + (dd.z * hh.z + dd.w * hh.w);
    };
#pragma unroll
    for (int k = 0; k < BN_MAXV; k++) {
        const int i = threadIdx.x + BN_NT * k;
        if (i < nv) { load(i, d[k], h[k]); add(d[k], h[k]); }
    }
    for (int i = threadIdx.x + BN_NT * BN_MAXV; i < nv; i += BN_NT) { float4 dd, hh; load(i, dd, hh); add(dd, hh); }
    const float n = (float)n_boards * 64.f;
    const float db = bn_block_sum(sb, red), dg = bn_block_sum(sg, red);
    if (threadIdx.x == 0) { dbeta[c] = db; dgamma[c] = dg; }
    const float k1 = gamma[c] * invstd, mb = db / n, mg = dg / n;
    auto out = [&](int i, const float4 dd, const float4 hh) {
        const size_t o = bn_off(i, c, channels);
        *(float4*)(dx + o) = make_float4(k1 * (dd.x - mb - hh.x * mg), k1 * (dd.y - mb - hh.y * mg), k1 * (dd.z - mb - hh.z * mg), k1 * (dd.w - mb - hh.w * mg));
        if (dres) *(float4*)(dres + o) = dd;
    };
#pragma unroll
    for (int k = 0; k < BN_MAXV; k++) {
        const int i = threadIdx.x + BN_NT * k;
        if (i < nv) out(i, d[k], h[k]);
    }
    for (int i = threadIdx.x + BN_NT * BN_MAXV; i < nv; i += BN_NT) {
        float4 dd, hh; load(i, dd, hh);                                  // beyond the register cache: read again
        out(i, dd, hh);
    }
}

extern "C" {

// y = relu(batch_norm_train(x; gamma, beta, eps) [+ residual]); running statistics updated in place (NULL: not tracked); save_mean / save_invstd [channels] for backward.
int sz_bn_act_train_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps, const float* residual,
                        float* y, float* save_mean, float* save_invstd, int32_t n_boards, int32_t channels, void* stream) {
    if (!x || !gamma || !beta || !y || !save_mean || !save_invstd || n_boards <= 0 || channels <= 0) return SZ_ERR_INVALID;
    StreamDeviceGuard _guard(stream);
    hipLaunchKernelGGL(k_bn_act_fwd, dim3(channels), dim3(BN_NT), 0, (hipStream_t)stream, x, gamma, beta, running_mean, running_var, momentum, eps, residual, y, save_mean,
                       save_invstd, (int)n_boards, (int)channels);
    HIPCHK(hipGetLastError());
    return SZ_OK;
}
// gradients of the above: dx [like x], dres (NULL without a residual: the skip connection's gradient = gy where y > 0), dgamma, dbeta [channels]
int sz_bn_act_train_bwd(const float* gy, const float* x, const float* y, const float* gamma, const float* save_mean, const float* save_invstd, float* dx, float* dres,
                        float* dgamma, float* dbeta, int32_t n_boards, int32_t channels, void* stream) {
    if (!gy || !x || !y || !gamma || !save_mean || !save_invstd || !dx || !dgamma || !dbeta || n_boards <= 0 || channels <= 0) return SZ_ERR_INVALID;
    StreamDeviceGuard _guard(stream);
    hipLaunchKernelGGL(k_bn_act_bwd, dim3(channels), dim3(BN_NT), 0, (hipStream_t)stream, gy, x, y, gamma, save_mean, save_invstd, dx, dres, dgamma, dbeta, (int)n_boards,
                       (int)channels);
    HIPCHK(hipGetLastError());
    return SZ_OK;
}

}  // extern "C"
