// tron_dqn.hip — the small device-side pieces of the batched DDQN trainer around the network (DDQN.py:90-151,313-315):
// each is a handful of elementwise / reduction steps that PyTorch runs as one ~5 us launch apiece — 14 for the loss, 4 for
// the epsilon-greedy mix, 8 for the epsilon schedule — on tensors of a few thousand elements.  One launch each here:
//   k_td_loss        the Double-DQN loss and its gradient at the local net's Q-values
//   k_eps_greedy     actions = greedy or uniform random, per observation, with epsilon read from device memory
//   k_eps_schedule   finished-game counter, 20-game cycles, epsilon = eps0 * rate ^ decays — without leaving the device
//   k_lin_wgrad      an nn.Linear layer's weight and bias gradient with the batch split over workgroups (the library's GEMM
//                    choices for these small outputs run on 1 - 144 workgroups)
//   k_absmax_pow2    max |x| of a gradient tensor and the power of two the split-f16 kernels scale it by (six launches as
//                    tensor expressions), for the learner's head, the ACKTR nets' convolutions and K-FAC's gradient factors
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tron_hip.h"
#include "tron_device.hpp"

namespace {

constexpr int TD_THREADS = 1024;

// loss = mean_b (q[b][a_b] - y_b)^2,  y_b = r_b + gamma * qt_next[b][argmax_a ql_next[b][a]] * (1 - done_b)   (DDQN.py:129-146:
// gather, max(1)[1], gather, MSELoss).  grad_q[b][a] = 2 (q[b][a_b] - y_b) / B at a = a_b, 0 elsewhere.  One workgroup: the sum
// runs in a fixed order (deterministic).  argmax takes the first maximum, as torch.max does.
__global__ __launch_bounds__(TD_THREADS) void k_td_loss(const float *__restrict__ q, const int64_t *__restrict__ actions,
                                                        const float *__restrict__ rewards, const float *__restrict__ dones,
                                                        const float *__restrict__ ql_next, const float *__restrict__ qt_next, float gamma,
                                                        int B, float *__restrict__ loss, float *__restrict__ grad_q)
{
    __shared__ float red[TD_THREADS / 64];
    float acc = 0.0f;
    const float inv = 1.0f / (float)B;
    for (int b = threadIdx.x; b < B; b += TD_THREADS) {
        const float4 ql = reinterpret_cast<const float4 *>(ql_next)[b], qt = reinterpret_cast<const float4 *>(qt_next)[b];
        int am = 0;
        float best = ql.x;
        if (ql.y > best) { best = ql.y; am = 1; }
        if (ql.z > best) { best = ql.z; am = 2; }
        if (ql.w > best) { best = ql.w; am = 3; }
        const float qn = am == 0 ? qt.x : am == 1 ? qt.y : am == 2 ? qt.z : qt.w;
        const float y = rewards[b] + (gamma * qn) * (1.0f - dones[b]);
        const int a = (int)actions[b] & 3;
        const float4 qv = reinterpret_cast<const float4 *>(q)[b];
        const float p = a == 0 ? qv.x : a == 1 ? qv.y : a == 2 ? qv.z : qv.w;
        const float e = p - y;
        acc += e * e;
        const float g = 2.0f * e * inv;
        reinterpret_cast<float4 *>(grad_q)[b] = make_float4(a == 0 ? g : 0.0f, a == 1 ? g : 0.0f, a == 2 ? g : 0.0f, a == 3 ? g : 0.0f);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.0f;
        for (int k = 0; k < TD_THREADS / 64; ++k) s += red[k];
        *loss = s * inv;
    }
}

// DDQN.py:105-110 per observation: u <= epsilon ? uniform action : greedy.  One Philox block (key = seed, counter = call,
// observation quad) gives four observations their two draws each.
__global__ void k_eps_greedy(const int8_t *__restrict__ greedy, int64_t n, const float *__restrict__ epsilon, uint32_t seed,
                             uint32_t stream, uint32_t call_lo, uint32_t call_hi, int8_t *__restrict__ out)
{
    const float eps = *epsilon;
    const int64_t quads = (n + 3) / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t x[4];
        tron::philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), call_lo, call_hi, seed, stream, x);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t j = 4 * i + k;
            if (j >= n) break;
            // 16 bits decide explore / exploit (u = (h + 0.5) / 65536 in (0, 1)), 2 bits the action
            const float u = ((float)(x[k] >> 16) + 0.5f) * (1.0f / 65536.0f);
            out[j] = u <= eps ? (int8_t)(x[k] & 3u) : greedy[j];
        }
    }
}

// DDQN.py:313-315 once per env step: games += finished; one decay of epsilon per finished 20-game cycle, at most
// decays_max of them; epsilon = eps0 * rate ^ decays (the closed form of the repeated multiply).  state = {games, cycles,
// decays, decays_max} (int64, device); one workgroup.
__global__ __launch_bounds__(1024) void k_eps_schedule(const int8_t *__restrict__ done, int64_t n, int64_t *__restrict__ state,
                                                       int64_t cycle, double eps0, double rate, double *__restrict__ eps_out,
                                                       float *__restrict__ eps_out_f32)
{
    __shared__ int red[16];
    int cnt = 0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) cnt += done[i] != 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int k = 0; k < 16; ++k) total += red[k];
        const int64_t games = state[0] + total;
        const int64_t cycles = games / cycle;
        int64_t decays = state[2] + (cycles - state[1]);
        decays = decays < state[3] ? decays : state[3];
        state[0] = games;
        state[1] = cycles;
        state[2] = decays;
        const double e = eps0 * pow(rate, (double)decays);
        *eps_out = e;
        *eps_out_f32 = (float)e;
    }
}

// max |x| and the power of two that brings it to [2^(t-1), 2^t): what the split-f16 kernels scale a gradient tensor by on its
// way into f16 (aminmax + frexp + ldexp + where on the device: six small launches as tensor expressions).  out = {scale,
// max |x|, (scratch: the maximum's bits while the blocks run), (scratch: blocks done)}, zeroed by the caller; non-negative
// floats order like their bit patterns, so the block maxima meet in an integer atomicMax and the last block to finish
// writes the result.
// (At most 256 workgroups of 1 024 threads: the two atomics per workgroup serialise on their address at ~25 ns each — 2 048
// small workgroups took 51 us for a 9 MB tensor.)
__global__ __launch_bounds__(1024) void k_absmax_pow2(const float *__restrict__ x, int64_t n, int target_exp, float *__restrict__ out)
{
    __shared__ float red[16];
    float m = 0.0f;
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 1024) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (blockIdx.x == 0)
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 1024) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = 0.0f;
        for (int k = 0; k < 16; ++k) m = fmaxf(m, red[k]);
        unsigned int *bits = reinterpret_cast<unsigned int *>(out + 2), *ticket = reinterpret_cast<unsigned int *>(out + 3);
        if (m == m) atomicMax(bits, __float_as_uint(m));                // (a NaN block maximum is left out)
        __threadfence();
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
            const float mx = __uint_as_float(atomicMax(bits, 0u));
            const int e = (int)((__float_as_uint(mx) >> 23) & 255u) - 126;   // mx = f 2^e, f in [0.5, 1)
            int k = target_exp - e;
            k = k < -60 ? -60 : (k > 60 ? 60 : k);                        // (the scale's square must stay finite)
            out[0] = mx > 0.0f ? __uint_as_float((uint32_t)(127 + k) << 23) : 1.0f;
            out[1] = mx;
        }
    }
}

// The weight and bias gradient of an nn.Linear layer (gw[o][i] = sum_b gy[b][o] x[b][i], gb[o] = sum_b gy[b][o]): outputs of a
// few thousand to 150 K elements reduced over the whole batch.  The library picks GEMM kernels with one to 144 workgroups for
// these shapes (26-41 us each at a batch of 4 096) and a separate 7-19 us column reduction for the bias.  Here the batch is
// split over workgroups: a workgroup = a 64 x 64 tile of gw for one slice of the batch (16 rows at a time through LDS, a
// thread = 4 x 4 outputs, plain f32 FMAs: 1.2 GFLOP for the largest layer), partial sums per slice, k_lin_finish adds the
// slices in a fixed order (deterministic) — both gradients from one pass over gy.
constexpr int LW_T = 64, LW_KB = 16;
__global__ __launch_bounds__(256) void k_lin_wgrad(const float *__restrict__ gy, const float *__restrict__ x, int B, int O, int I, int nsplit,
                                                   float *__restrict__ pw, float *__restrict__ pb)
{
    __shared__ float sg[LW_KB][LW_T + 4], sx[LW_KB][LW_T + 4];
    const int tiles_i = (I + LW_T - 1) / LW_T;
    const int o0 = ((int)blockIdx.x / tiles_i) * LW_T, i0 = ((int)blockIdx.x % tiles_i) * LW_T, s = blockIdx.y;
    const int b_lo = (int)((int64_t)B * s / nsplit), b_hi = (int)((int64_t)B * (s + 1) / nsplit);
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    float acc[4][4], accb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = 0.0f;
    for (int b0 = b_lo; b0 < b_hi; b0 += LW_KB) {
        // 16 rows x 64 columns of each operand: 1 024 floats per operand, 4 per thread (row = tid / 16, 4 consecutive columns)
        {
            const int r = tid >> 4, c4 = (tid & 15) * 4, b = b0 + r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sg[r][c4 + e] = (b < b_hi && o0 + c4 + e < O) ? gy[(size_t)b * O + o0 + c4 + e] : 0.0f;
                sx[r][c4 + e] = (b < b_hi && i0 + c4 + e < I) ? x[(size_t)b * I + i0 + c4 + e] : 0.0f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < LW_KB; ++r) {
            const float4 gv = *reinterpret_cast<const float4 *>(&sg[r][ty * 4]);
            const float4 xv = *reinterpret_cast<const float4 *>(&sx[r][tx * 4]);
            const float ga[4] = {gv.x, gv.y, gv.z, gv.w}, xa[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                accb[a] += ga[a];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[a][c] = __fmaf_rn(ga[a], xa[c], acc[a][c]);
            }
        }
        __syncthreads();
    }
    float *w = pw + (size_t)s * O * I;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int o = o0 + ty * 4 + a;
        if (o >= O) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (i0 + tx * 4 + c < I) w[(size_t)o * I + i0 + tx * 4 + c] = acc[a][c];
        if (pb && i0 == 0 && tx == 0) pb[(size_t)s * O + o] = accb[a];
    }
}

__global__ void k_lin_finish(const float *__restrict__ pw, const float *__restrict__ pb, int O, int I, int nsplit, float *__restrict__ gw,
                             float *__restrict__ gb)
{
    const int total = O * I + (gb ? O : 0);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const bool is_w = i < O * I;
        const float *p = is_w ? pw + i : pb + (i - O * I);
        const size_t stride = is_w ? (size_t)O * I : (size_t)O;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};                              // four independent chains, joined in a fixed order
        int s = 0;
        for (; s + 3 < nsplit; s += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += p[(size_t)(s + k) * stride];
        }
        for (int k = 0; s < nsplit; ++s, ++k) acc[k] += p[(size_t)s * stride];
        const float v = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        if (is_w) gw[i] = v;
        else gb[i - O * I] = v;
    }
}

inline int lin_nsplit(int64_t B, int O, int I)
{
    const int tiles = ((O + LW_T - 1) / LW_T) * ((I + LW_T - 1) / LW_T);
    int n = 768 / tiles;                                                  // three workgroups per CU over the launch
    if (n > B / 32) n = (int)(B / 32);                                    // a slice is at least 32 rows
    return n < 1 ? 1 : (n > 256 ? 256 : n);
}

}  // namespace

extern "C" int64_t tron_linear_wgrad_workspace(int64_t batch, int32_t out_features, int32_t in_features)
{
    if (batch < 1 || out_features < 1 || in_features < 1 || batch >= (1ll << 31) || (int64_t)out_features * in_features >= (1ll << 28)) return 0;
    return (int64_t)lin_nsplit(batch, out_features, in_features) * ((int64_t)out_features * in_features + out_features) * (int64_t)sizeof(float);
}

extern "C" int tron_linear_wgrad(const float *grad_out, const float *input, int64_t batch, int32_t out_features, int32_t in_features,
                                 float *grad_weight, float *grad_bias, void *workspace, void *stream)
{
    if (!grad_out || !input || !grad_weight || !workspace || batch < 0 || out_features < 1 || in_features < 1) return TRON_ERR_BAD_ARG;
    if (batch >= (1ll << 31) || (int64_t)out_features * in_features >= (1ll << 28)) return TRON_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int O = out_features, I = in_features;
    if (batch == 0) {
        if (hipMemsetAsync(grad_weight, 0, (size_t)O * I * sizeof(float), st) != hipSuccess ||
            (grad_bias && hipMemsetAsync(grad_bias, 0, (size_t)O * sizeof(float), st) != hipSuccess)) { (void)hipGetLastError(); return TRON_ERR_LAUNCH; }
        return TRON_OK;
    }
    const int nsplit = lin_nsplit(batch, O, I);
    float *pw = reinterpret_cast<float *>(workspace), *pb = pw + (size_t)nsplit * O * I;
    const int tiles = ((O + LW_T - 1) / LW_T) * ((I + LW_T - 1) / LW_T);
    hipLaunchKernelGGL(k_lin_wgrad, dim3((unsigned)tiles, (unsigned)nsplit), dim3(256), 0, st, grad_out, input, (int)batch, O, I, nsplit, pw,
                       grad_bias ? pb : nullptr);
    const int total = O * I + (grad_bias ? O : 0);
    hipLaunchKernelGGL(k_lin_finish, dim3((unsigned)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024)), dim3(256), 0, st, pw, pb, O, I,
                       nsplit, grad_weight, grad_bias);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

// ---- Adam + soft update of every parameter tensor in one launch (DDQN.py:52,149-150,153-165) -----------------------------------
// torch's fused Adam is one multi-tensor launch of 43 us for these 22 tensors (0.5-1.2 M elements: a block walks its tensor's chunks
// in turn), its step counters a second launch, the soft update two more (75 us per learn step together); the arithmetic is 5 reads
// and 4 writes of 2-5 MB.  Here the tensors' pointers ride in the kernel arguments and a block = one 1 024-element chunk of one tensor.
namespace {
struct AdamJobs {
    float *p[TRON_ADAM_MAX_TENSORS], *m[TRON_ADAM_MAX_TENSORS], *v[TRON_ADAM_MAX_TENSORS], *t[TRON_ADAM_MAX_TENSORS];
    const float *g[TRON_ADAM_MAX_TENSORS];
    int chunk0[TRON_ADAM_MAX_TENSORS + 1];                               // first chunk of tensor k (prefix sums)
    int numel[TRON_ADAM_MAX_TENSORS];
    float step_size[TRON_ADAM_MAX_TENSORS], bc2_sqrt[TRON_ADAM_MAX_TENSORS];
    int n;
};
constexpr int ADAM_CHUNK = 1024;
__global__ __launch_bounds__(256) void k_adam_soft(AdamJobs J, float beta1, float beta2, float eps, float tau)
{
    int k = 0;
    while (k + 1 < J.n && (int)blockIdx.x >= J.chunk0[k + 1]) ++k;       // (block-uniform: scalar loads from the kernel arguments)
    const int base = ((int)blockIdx.x - J.chunk0[k]) * ADAM_CHUNK, n = J.numel[k];
    float *__restrict__ p = J.p[k], *__restrict__ m = J.m[k], *__restrict__ v = J.v[k], *__restrict__ t = J.t[k];
    const float *__restrict__ g = J.g[k];
    const float ss = J.step_size[k], bc = J.bc2_sqrt[k];
#pragma unroll
    for (int j = 0; j < ADAM_CHUNK / 256; ++j) {
        const int i = base + j * 256 + (int)threadIdx.x;
        if (i >= n) break;
        float w = p[i];
        if (g) {
            // torch/optim's single-tensor formulas in f32: m <- lerp(m, g, 1 - b1); v <- b2 v + (1 - b2) g g;
            // w <- w - (lr / (1 - b1^step)) m / (sqrt(v) / sqrt(1 - b2^step) + eps)
            const float gi = g[i];
            const float mi = __fmaf_rn(1.0f - beta1, gi - m[i], m[i]);
            const float vi = __fmaf_rn(beta2, v[i], (1.0f - beta2) * gi * gi);
            m[i] = mi;
            v[i] = vi;
            w -= ss * (mi / (__fsqrt_rn(vi) / bc + eps));
            p[i] = w;
        }
        if (t) t[i] = __fmaf_rn(tau, w, (1.0f - tau) * t[i]);           // theta_target <- tau theta + (1 - tau) theta_target
    }
}
}  // namespace

extern "C" int tron_adam_soft_update(int32_t n, float *const *params, const float *const *grads, float *const *exp_avg,
                                     float *const *exp_avg_sq, float *const *targets, const int64_t *numel, const double *steps,
                                     double lr, double beta1, double beta2, double eps, double tau, void *stream)
{
    if (n < 0 || (n && (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !steps))) return TRON_ERR_BAD_ARG;
    if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0 && tau >= 0.0 && tau <= 1.0)) return TRON_ERR_BAD_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    for (int k = 0; k < n;) {
        AdamJobs J{};
        int cnt = 0;
        J.chunk0[0] = 0;
        for (; k < n && cnt < TRON_ADAM_MAX_TENSORS; ++k) {
            if (!params[k] || numel[k] < 0 || numel[k] >= (1ll << 31) - ADAM_CHUNK) return TRON_ERR_BAD_ARG;
            if (grads[k] && (!exp_avg[k] || !exp_avg_sq[k] || !(steps[k] >= 1.0))) return TRON_ERR_BAD_ARG;
            if (numel[k] == 0 || (!grads[k] && !(targets && targets[k]))) continue;
            J.p[cnt] = params[k];
            J.g[cnt] = grads[k];
            J.m[cnt] = exp_avg[k];
            J.v[cnt] = exp_avg_sq[k];
            J.t[cnt] = targets ? targets[k] : nullptr;
            J.numel[cnt] = (int)numel[k];
            if (grads[k]) {
                J.step_size[cnt] = (float)(lr / (1.0 - pow(beta1, steps[k])));
                J.bc2_sqrt[cnt] = (float)sqrt(1.0 - pow(beta2, steps[k]));
            }
            const int64_t chunks = (numel[k] + ADAM_CHUNK - 1) / ADAM_CHUNK;
            if ((int64_t)J.chunk0[cnt] + chunks >= (1ll << 31)) return TRON_ERR_UNSUPPORTED;
            J.chunk0[cnt + 1] = J.chunk0[cnt] + (int)chunks;
            ++cnt;
        }
        J.n = cnt;
        if (!cnt) continue;
        hipLaunchKernelGGL(k_adam_soft, dim3((unsigned)J.chunk0[cnt]), dim3(256), 0, st, J, (float)beta1, (float)beta2, (float)eps, (float)tau);
        if (hipGetLastError() != hipSuccess) return TRON_ERR_LAUNCH;
    }
    return TRON_OK;
}

extern "C" int tron_absmax_pow2(const float *x, int64_t n, int32_t target_exp, float *out4, void *stream)
{
    if (!x || !out4 || n < 0 || target_exp < -60 || target_exp > 60) return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(x) & 15u) || (reinterpret_cast<uintptr_t>(out4) & 15u)) return TRON_ERR_BAD_ARG;
    const int64_t blocks = (n / 4 + 4095) / 4096;                        // (four float4 per thread before another workgroup pays)
    hipLaunchKernelGGL(k_absmax_pow2, dim3((unsigned)(blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks))), dim3(1024), 0,
                       reinterpret_cast<hipStream_t>(stream), x, n, target_exp, out4);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

extern "C" int tron_ddqn_td_loss(const float *q, const int64_t *actions, const float *rewards, const float *dones,
                                 const float *q_local_next, const float *q_target_next, float gamma, int64_t batch, float *loss,
                                 float *grad_q, void *stream)
{
    if (!q || !actions || !rewards || !dones || !q_local_next || !q_target_next || !loss || !grad_q || batch < 1) return TRON_ERR_BAD_ARG;
    if (batch > (1ll << 24)) return TRON_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(q_local_next) | reinterpret_cast<uintptr_t>(q_target_next) |
         reinterpret_cast<uintptr_t>(grad_q)) & 15u)
        return TRON_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_td_loss, dim3(1), dim3(TD_THREADS), 0, reinterpret_cast<hipStream_t>(stream), q, actions, rewards, dones,
                       q_local_next, q_target_next, gamma, (int)batch, loss, grad_q);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

extern "C" int tron_eps_greedy(const int8_t *greedy, int64_t n, const float *epsilon, uint32_t seed, uint32_t stream_id,
                               uint64_t call, int8_t *actions, void *stream)
{
    if (!greedy || !epsilon || !actions || n < 0) return TRON_ERR_BAD_ARG;
    if (n == 0) return TRON_OK;
    const int64_t quads = (n + 3) / 4, blocks = (quads + 255) / 256;
    hipLaunchKernelGGL(k_eps_greedy, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       greedy, n, epsilon, seed, stream_id, (uint32_t)call, (uint32_t)(call >> 32), actions);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

extern "C" int tron_eps_schedule(const int8_t *done, int64_t n, int64_t *state4, int64_t games_per_cycle, double eps0, double rate,
                                 double *epsilon_out, float *epsilon_out_f32, void *stream)
{
    if (!done || !state4 || !epsilon_out || !epsilon_out_f32 || n < 0 || games_per_cycle < 1) return TRON_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_eps_schedule, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), done, n, state4, games_per_cycle,
                       eps0, rate, epsilon_out, epsilon_out_f32);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}
