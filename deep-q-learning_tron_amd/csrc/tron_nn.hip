// tron_nn.hip — the activation of every net in the reference, mish(x) = x * tanh(softplus(x))
// (Net/ACNet.py:56-57), as one memory-bound pass forward and one backward.
//
// torch's composed form is three elementwise launches; its fused F.mish is one, but spends ~86 us on a
// 4096 x 32 x 12 x 12 tensor (75 MB in + out: ~15 us of traffic) evaluating exp, log1p and tanh in turn —
// 14 % of the DDQN trainer's GPU time (rocprofv3, profiles/r01_dqn_kernel_stats.csv).  With e = exp(x):
//     tanh(log1p(e)) = ((1+e)^2 - 1) / ((1+e)^2 + 1) = n / (n + 2),   n = e * (e + 2)
// so one exp and one division give the forward value, with no cancellation anywhere (for x -> -inf,
// n -> 2e and the quotient -> e; for x > 20 the result is x to fp32 precision — the same cut-over as
// F.softplus's threshold — which also keeps e*e from overflowing).  Backward:
//     d/dx = t + x * (1 - t^2) * e / (1 + e),   t = n / (n + 2),   1 - t^2 = (2 / (n + 2)) (1 + t).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tron_hip.h"

namespace {

__device__ __forceinline__ float mish1(float x)
{
    if (x > 20.0f) return x;
    const float e = expf(x);
    const float n = e * (e + 2.0f);
    return x * (n / (n + 2.0f));
}
__device__ __forceinline__ float mish_grad1(float x, float gy)
{
    if (x > 20.0f) return gy;
    const float e = expf(x);
    const float n = e * (e + 2.0f);
    const float t = n / (n + 2.0f), u = 2.0f / (n + 2.0f);       // u = 1 - t without the cancellation
    return gy * (t + x * (u * (1.0f + t)) * (e / (1.0f + e)));   // sech^2 = 1 - t^2 = (1 - t)(1 + t)
}

__global__ void k_mish_fwd(const float *__restrict__ x, float *__restrict__ y, size_t n4, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        reinterpret_cast<float4 *>(y)[i] = make_float4(mish1(v.x), mish1(v.y), mish1(v.z), mish1(v.w));
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = mish1(x[i]);
}
__global__ void k_mish_bwd(const float *__restrict__ x, const float *__restrict__ gy, float *__restrict__ gx, size_t n4,
                           size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i], g = reinterpret_cast<const float4 *>(gy)[i];
        reinterpret_cast<float4 *>(gx)[i] =
            make_float4(mish_grad1(v.x, g.x), mish_grad1(v.y, g.y), mish_grad1(v.z, g.z), mish_grad1(v.w, g.w));
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        gx[i] = mish_grad1(x[i], gy[i]);
}

// pre = y + bias[c] (+ residual), out = mish(pre): what follows every convolution of the DQN net
// (Net/DQNNet.py:33-63) in one pass instead of a broadcast add, a residual add and the activation.
// y is overwritten with pre (backward needs it; nobody else needs y).  [N][C][HW] layout, HW % 4 == 0.
__global__ void k_bias_mish_fwd(float *__restrict__ y, const float *__restrict__ bias, const float *__restrict__ res,
                                float *__restrict__ out, uint32_t n4, uint32_t hw4, uint32_t C)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float b = bias[(i / hw4) % C];
        float4 v = reinterpret_cast<const float4 *>(y)[i];
        v.x += b; v.y += b; v.z += b; v.w += b;
        if (res) {
            const float4 r = reinterpret_cast<const float4 *>(res)[i];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        reinterpret_cast<float4 *>(y)[i] = v;
        reinterpret_cast<float4 *>(out)[i] = make_float4(mish1(v.x), mish1(v.y), mish1(v.z), mish1(v.w));
    }
}

// Backward of k_bias_mish_fwd in one pass: grad_pre = grad_out * mish'(pre), and the bias gradient
// sum_{n,hw} grad_pre[n][c][hw] as per-(channel, segment) partial sums — block (c, seg) owns the samples
// [seg * N / SEGS, (seg + 1) * N / SEGS) of channel c — which k_bias_grad_finish adds up in a fixed order
// (deterministic, unlike atomics).  torch's generic reduction spent 163 us per layer on this sum alone.
constexpr int BIAS_SEGS = 64;

__global__ __launch_bounds__(256) void k_bias_mish_bwd(const float *__restrict__ pre, const float *__restrict__ gy,
                                                       float *__restrict__ gx, float *__restrict__ partial, int N, int C,
                                                       int hw4)
{
    __shared__ float red[4], redm[4];
    const int c = blockIdx.x / BIAS_SEGS, seg = blockIdx.x - c * BIAS_SEGS;
    const int n0 = (int)((long long)N * seg / BIAS_SEGS), n1 = (int)((long long)N * (seg + 1) / BIAS_SEGS);
    const int total = (n1 - n0) * hw4;
    float acc = 0.0f, amax = 0.0f;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int n = n0 + i / hw4, q = i - (i / hw4) * hw4;
        const size_t o = ((size_t)n * C + c) * hw4 + q;
        const float4 v = reinterpret_cast<const float4 *>(pre)[o], g = reinterpret_cast<const float4 *>(gy)[o];
        const float4 r = make_float4(mish_grad1(v.x, g.x), mish_grad1(v.y, g.y), mish_grad1(v.z, g.z), mish_grad1(v.w, g.w));
        reinterpret_cast<float4 *>(gx)[o] = r;
        acc += (r.x + r.y) + (r.z + r.w);
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        acc += __shfl_xor(acc, d);
        amax = fmaxf(amax, __shfl_xor(amax, d));
    }
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = acc;
        redm[threadIdx.x >> 6] = amax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        partial[gridDim.x + blockIdx.x] = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));   // for the weight-gradient kernel's scale
    }
}

__global__ void k_bias_grad_finish(const float *__restrict__ partial, float *__restrict__ bias_grad, int C)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.0f;
    for (int k = 0; k < BIAS_SEGS; ++k) s += partial[c * BIAS_SEGS + k];
    bias_grad[c] = s;
}

inline unsigned grid_for(size_t n4)
{
    size_t b = (n4 + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;      // 16 workgroups per CU, grid-stride beyond
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int tron_mish_fwd(const float *x, float *y, int64_t n, void *stream)
{
    if (n < 0 || (n > 0 && (!x || !y))) return TRON_ERR_BAD_ARG;
    if (n == 0) return TRON_OK;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15u) return TRON_ERR_BAD_ARG;
    const size_t n4 = (size_t)n / 4;
    hipLaunchKernelGGL(k_mish_fwd, dim3(grid_for(n4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y, n4,
                       (size_t)n);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

extern "C" int tron_mish_bwd(const float *x, const float *grad_y, float *grad_x, int64_t n, void *stream)
{
    if (n < 0 || (n > 0 && (!x || !grad_y || !grad_x))) return TRON_ERR_BAD_ARG;
    if (n == 0) return TRON_OK;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(grad_y) | reinterpret_cast<uintptr_t>(grad_x)) & 15u)
        return TRON_ERR_BAD_ARG;
    const size_t n4 = (size_t)n / 4;
    hipLaunchKernelGGL(k_mish_bwd, dim3(grid_for(n4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, grad_y,
                       grad_x, n4, (size_t)n);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

extern "C" int tron_bias_mish_fwd(float *y_pre, const float *bias, const float *residual, float *out, int64_t batch,
                                  int32_t channels, int32_t hw, void *stream)
{
    if (batch < 0 || channels < 1 || hw < 1 || !y_pre || !bias || !out) return TRON_ERR_BAD_ARG;
    const int64_t n = batch * channels * hw;
    if (n == 0) return TRON_OK;
    if ((hw & 3) || n > 0xFFFFFFFFll) return TRON_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(y_pre) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(residual)) & 15u)
        return TRON_ERR_BAD_ARG;
    const uint32_t n4 = (uint32_t)(n / 4);
    hipLaunchKernelGGL(k_bias_mish_fwd, dim3(grid_for(n4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), y_pre, bias,
                       residual, out, n4, (uint32_t)(hw / 4), (uint32_t)channels);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}


extern "C" int tron_bias_mish_bwd(const float *pre, const float *grad_out, float *grad_pre, float *bias_grad,
                                  float *scratch, int64_t batch, int32_t channels, int32_t hw, void *stream)
{
    if (batch < 0 || channels < 1 || hw < 1 || !pre || !grad_out || !grad_pre || !bias_grad || !scratch) return TRON_ERR_BAD_ARG;
    if ((hw & 3) || batch > 0x7FFFFFFFll || batch * channels * hw > 0x7FFFFFFFll * 4) return TRON_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(pre) | reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(grad_pre)) & 15u)
        return TRON_ERR_BAD_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_bias_mish_bwd, dim3((unsigned)(channels * BIAS_SEGS)), dim3(256), 0, st, pre, grad_out, grad_pre,
                       scratch, (int)batch, channels, hw / 4);
    hipLaunchKernelGGL(k_bias_grad_finish, dim3((channels + 63) / 64), dim3(64), 0, st, scratch, bias_grad, channels);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}
