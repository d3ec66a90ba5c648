// tron_head.hip — what follows the 3x3 trunk in the reference's DQN net (Net/DQNNet.py:52-63):
//     x = pool(x); x = mish(conv7(x)); x = x.view(-1, 64*3*3); x = mish(fc1(x)); x = mish(fc2(x));
//     q = actor2(mish(actor1(x)))                      (dropout is the identity in eval mode)
// as one C-ABI call, tron_dqn_head_fwd, for gradient-free forwards (policy, targets).
//
// All of it is GEMM-shaped once conv7 is written densely: on the 6x6 pooled planes a 7x7 / stride 2 / pad 3 convolution
// has 3x3 outputs and most taps fall on padding, so the dense map [64*6*6 = 2304] -> [64*3*3 = 576] (2.65 MFLOP per
// sample) is cheaper than the nominal convolution (3.6 MFLOP).  Every product runs on the f16 matrix cores with both
// operands split in two halves (v = hi + lo 2^-11, three MFMAs per k-slab, f32 accumulation — csrc/tron_conv_f16.hip
// has the derivation and the error bound), and the layers hand their outputs on already split:
//     k_pool_split12    avg-pool 3/2/1 (count_include_pad) of the trunk's f32 output -> split f16 rows [B][2304]
//     k_head_weights    every weight in the form its GEMM wants, one launch: conv7 -> the dense matrix [576][2304] split
//                       (dense7_split), the nn.Linear weights [N][K] split (split_rows)     (per call: nothing cached)
//     k_gemm_f16x3      C = act(A W^T + bias), A and W split f16 row-major; C as f32 and / or split f16
//     k_q_head          actor2 (64 -> 4) in f32 + argmax
// GEMM tile: 128 x 64 per 8-wave workgroup (wave = 32 x 32 = 2 x 2 MFMA tiles), K in chunks of 64 staged by 16-byte
// copies into single-buffered LDS (60 KB: two workgroups share a CU and cover each other's barriers).
//
// The same layers on the TRAINING path (loss.backward() through them, DDQN.py:148):
//     12x12:           tron_pool12 / tron_conv7_dense around tron_gemm_f16x3 (conv7's dense form: three plain GEMMs)
//     26x26 / 34x34:   tron_pool_conv7_fwd / _bwd — the pooled planes kept channels-last and split; forward = the CONV7 implicit
//                      GEMM; input gradient = four DGRAD7 implicit GEMMs (one per parity class of the pooled pixel: a stride-2
//                      convolution's taps split by parity) + k_pool_bwd_cl; weight gradient = k_conv7_wgrad (both operands are
//                      channels-last, i.e. the product runs over their SLOW index: ds_read_b64_tr_b16 reads them transposed out
//                      of LDS, the reads' row addresses do the stride-2 im2col, LDS-DMA stages an image under the previous one's
//                      MFMAs); tron_conv7_fwd / _bwd — conv7 alone, NCHW in and out, for the module K-FAC hooks.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tron_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr float ACT_SCALE = 1.0f / 64.0f, ACT_UNSCALE = 64.0f, LO_SCALE = 2048.0f, LO_UNSCALE = 1.0f / 2048.0f;

__device__ __forceinline__ float mish1(float x)                         // as in tron_conv.hip
{
    const float e = __builtin_amdgcn_exp2f(x * 1.44269504088896341f);
    const float n = __fmaf_rn(e, e, e + e);
    const float d = n + 2.0f;
    float r = __builtin_amdgcn_rcpf(d);
    r = __fmaf_rn(r, __fmaf_rn(-d, r, 1.0f), r);
    const float y = x * (n * r);
    return x > 20.0f ? x : y;
}
__device__ __forceinline__ void split(float v, f16 &hi, f16 &lo)
{
    hi = (f16)v;
    lo = (f16)((v - (float)hi) * LO_SCALE);
}

// x f32 [B][C][12][12] -> AvgPool2d(3, stride 2, padding 1) (divisor 9 everywhere: count_include_pad) -> rows
// [B][C * 6 * 6] in NCHW-flatten order, pre-scaled by 2^-6 and split.  One thread = one pooled row of one plane: three
// input rows as 16-byte loads (a 12-float row is 48 bytes, so every row is aligned), six outputs as 4-byte stores.
__global__ __launch_bounds__(256) void k_pool_split12(const float *__restrict__ x, int64_t rows, f16 *__restrict__ oh,
                                                      f16 *__restrict__ ol)
{
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += stride) {
        const int py = (int)(i % 6);
        const int64_t plane = i / 6;
        const float *p = x + plane * 144;
        float col[13];                                                   // col[1 + xx] = sum over the 3 rows; col[0] = pad
#pragma unroll
        for (int k = 0; k < 13; ++k) col[k] = 0.0f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = 2 * py + dy;
            if (yy < 0) continue;                                        // yy <= 11 always (py <= 5, dy <= 1)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(p + yy * 12 + q * 4);
                col[1 + q * 4] += v[0];
                col[2 + q * 4] += v[1];
                col[3 + q * 4] += v[2];
                col[4 + q * 4] += v[3];
            }
        }
        f16 h[6], l[6];
#pragma unroll
        for (int px = 0; px < 6; ++px)                                   // window columns 2px-1 .. 2px+1 -> col[2px .. 2px+2]
            split((col[2 * px] + col[2 * px + 1] + col[2 * px + 2]) * (1.0f / 9.0f) * ACT_SCALE, h[px], l[px]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            *reinterpret_cast<f16x2 *>(oh + i * 6 + 2 * k) = (f16x2){h[2 * k], h[2 * k + 1]};
            *reinterpret_cast<f16x2 *>(ol + i * 6 + 2 * k) = (f16x2){l[2 * k], l[2 * k + 1]};
        }
    }
}

// conv7 (Co x Ci x 7 x 7, stride 2, pad 3) on PS x PS planes as a dense [Co*OS*OS][Ci*PS*PS] matrix, split
// (octets: the columns are ordered (channel octet, pixel, channel within the octet) — the order in which the PX16
// pooling writes a row, 16 bytes per (octet, pixel) — instead of NCHW-flatten (channel, pixel))
__device__ __forceinline__ void dense7_split(const float *__restrict__ w, int Co, int Ci, int PS, int OS, int octets, f16 *__restrict__ oh,
                                             f16 *__restrict__ ol, int first, int step)
{
    const int K = Ci * PS * PS, N = Co * OS * OS;
    const int total = N * K;
    for (int i = first; i < total; i += step) {
        const int n = i / K, k = i - n * K;
        const int co = n / (OS * OS), op = n - co * (OS * OS), oy = op / OS, ox = op - oy * OS;
        const int ci = octets ? (k / (PS * PS * 8)) * 8 + (k & 7) : k / (PS * PS);
        const int ip = octets ? (k % (PS * PS * 8)) >> 3 : k - ci * (PS * PS);
        const int iy = ip / PS, ix = ip - iy * PS;
        const int ky = iy - 2 * oy + 3, kx = ix - 2 * ox + 3;
        const float v = (ky >= 0 && ky < 7 && kx >= 0 && kx < 7) ? w[((size_t)(co * Ci + ci) * 7 + ky) * 7 + kx] : 0.0f;
        f16 h, l;
        split(v, h, l);
        oh[i] = h;
        ol[i] = l;
    }
}

__device__ __forceinline__ void split_rows(const float *__restrict__ w, int total, f16 *__restrict__ oh, f16 *__restrict__ ol, int first,
                                           int step)
{
    for (int i = first; i < total; i += step) {
        f16 h, l;
        split(w[i], h, l);
        oh[i] = h;
        ol[i] = l;
    }
}

// ---- the same two layers for the TRAINING path (f32; the products run as library GEMMs on the dense matrix) -----------
// AvgPool2d(3, stride 2, padding 1) of 12x12 planes, f32 -> f32 [B][C][6][6]; one thread = one pooled row
__global__ __launch_bounds__(256) void k_pool12(const float *__restrict__ x, int64_t rows, float *__restrict__ y)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += stride) {
        const int py = (int)(i % 6);
        const float *p = x + (i / 6) * 144;
        float col[13];
#pragma unroll
        for (int k = 0; k < 13; ++k) col[k] = 0.0f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = 2 * py + dy;
            if (yy < 0) continue;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(p + yy * 12 + q * 4);
                col[1 + q * 4] += v[0];
                col[2 + q * 4] += v[1];
                col[3 + q * 4] += v[2];
                col[4 + q * 4] += v[3];
            }
        }
        float o[6];
#pragma unroll
        for (int px = 0; px < 6; ++px) o[px] = (col[2 * px] + col[2 * px + 1] + col[2 * px + 2]) * (1.0f / 9.0f);
        float *d = y + i * 6;                                            // 24-byte rows: 8-byte aligned
#pragma unroll
        for (int k = 0; k < 3; ++k) *reinterpret_cast<float2 *>(d + 2 * k) = make_float2(o[2 * k], o[2 * k + 1]);
    }
}

// The same pooling for the other board sizes (26 x 26 planes: the 24x24 boards' tail is pooling + library kernels): one
// thread = one pooled row of one plane, the three input rows as 8-byte loads (a row of an even side is 8-byte aligned).
template <int S>
__global__ __launch_bounds__(256) void k_pool_rows(const float *__restrict__ x, int64_t rows, float *__restrict__ y)
{
    constexpr int PS = S / 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += stride) {
        const int py = (int)(i % PS);
        const float *p = x + (i / PS) * (S * S);
        float col[S + 1];                                                // col[1 + xx] = sum over the 3 rows; col[0] = pad
#pragma unroll
        for (int k = 0; k <= S; ++k) col[k] = 0.0f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = 2 * py + dy;
            if (yy < 0) continue;                                        // yy <= S - 1 always
#pragma unroll
            for (int q = 0; q < S / 2; ++q) {
                const float2 v = *reinterpret_cast<const float2 *>(p + yy * S + 2 * q);
                col[1 + 2 * q] += v.x;
                col[2 + 2 * q] += v.y;
            }
        }
        float *d = y + i * PS;
#pragma unroll
        for (int px = 0; px < PS; ++px) d[px] = (col[2 * px] + col[2 * px + 1] + col[2 * px + 2]) * (1.0f / 9.0f);
    }
}

// its backward: gx[y][x] = (1/9) * sum of gy over the windows that contain (y, x) — rows (y+1)/2 and, for odd y, also
// (y+2)/2... written per INPUT row: one thread = one input row of 12, 16-byte stores
__global__ __launch_bounds__(256) void k_pool12_bwd(const float *__restrict__ gy, int64_t rows, float *__restrict__ gx)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += stride) {
        const int y = (int)(i % 12);
        const float *g = gy + (i / 12) * 36;
        // window row py covers input rows 2py-1 .. 2py+1: even y belongs to py = y/2 only, odd y to (y-1)/2 and (y+1)/2
        const int p0 = y >> 1, p1 = (y & 1) ? p0 + 1 : -1;
        float r[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) r[k] = g[p0 * 6 + k] + ((p1 >= 0 && p1 < 6) ? g[p1 * 6 + k] : 0.0f);
        float o[12];
#pragma unroll
        for (int x = 0; x < 12; ++x) o[x] = ((x & 1) ? r[x >> 1] + ((x >> 1) + 1 < 6 ? r[(x >> 1) + 1] : 0.0f) : r[x >> 1]) * (1.0f / 9.0f);
        float *d = gx + i * 12;
#pragma unroll
        for (int q = 0; q < 3; ++q) *reinterpret_cast<f32x4 *>(d + 4 * q) = (f32x4){o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
    }
}

// the same backward for the other even sides (26: 24x24 boards, 34: 32x32): one thread = one input row, 8-byte stores
template <int S>
__global__ __launch_bounds__(256) void k_pool_rows_bwd(const float *__restrict__ gy, int64_t rows, float *__restrict__ gx)
{
    constexpr int PS = S / 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += stride) {
        const int y = (int)(i % S);
        const float *g = gy + (i / S) * (PS * PS);
        const int p0 = y >> 1, p1 = (y & 1) ? p0 + 1 : -1;              // even y: window row y/2 only; odd y: (y-1)/2 and (y+1)/2
        float r[PS + 1];
#pragma unroll
        for (int k = 0; k < PS; ++k) r[k] = g[p0 * PS + k] + ((p1 >= 0 && p1 < PS) ? g[p1 * PS + k] : 0.0f);
        r[PS] = 0.0f;
        float *d = gx + i * S;
#pragma unroll
        for (int q = 0; q < S / 2; ++q)                                  // x = 2q: column q; x = 2q + 1: columns q and q + 1
            *reinterpret_cast<float2 *>(d + 2 * q) = make_float2(r[q] * (1.0f / 9.0f), (r[q] + r[q + 1]) * (1.0f / 9.0f));
    }
}

// conv7 weight [Co][Ci][7][7] -> dense [Co*9][Ci*36] f32 (fold = 0), or the dense matrix's gradient -> the weight's:
// dW[co][ci][ky][kx] = sum over the <= 9 outputs (oy, ox) whose tap (ky, kx) lands inside the 6x6 plane (fold = 1)
__global__ void k_dense7(const float *__restrict__ src, float *__restrict__ dst, int Co, int Ci, int fold)
{
    if (!fold) {
        const int K = Ci * 36, total = Co * 9 * K;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
            const int n = i / K, k = i - n * K;
            const int co = n / 9, op = n - co * 9, oy = op / 3, ox = op - oy * 3;
            const int ci = k / 36, ip = k - ci * 36, iy = ip / 6, ix = ip - iy * 6;
            const int ky = iy - 2 * oy + 3, kx = ix - 2 * ox + 3;
            dst[i] = (ky >= 0 && ky < 7 && kx >= 0 && kx < 7) ? src[((size_t)(co * Ci + ci) * 7 + ky) * 7 + kx] : 0.0f;
        }
    } else {
        const int K = Ci * 36, total = Co * Ci * 49;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
            const int kx = i % 7, ky = (i / 7) % 7, ci = (i / 49) % Ci, co = i / (49 * Ci);
            float s = 0.0f;
#pragma unroll
            for (int oy = 0; oy < 3; ++oy)
#pragma unroll
                for (int ox = 0; ox < 3; ++ox) {
                    const int iy = 2 * oy + ky - 3, ix = 2 * ox + kx - 3;
                    if (iy >= 0 && iy < 6 && ix >= 0 && ix < 6) s += src[(size_t)(co * 9 + oy * 3 + ox) * K + ci * 36 + iy * 6 + ix];
                }
            dst[i] = s;
        }
    }
}

// ---- 24x24 boards: 26x26 trunk planes -> 13x13 pooled -> conv7 -> 7x7 ---------------------------------------------------
// avg-pool 3/2/1 of x f32 [B][64][26][26] into the channels-last split image [B][13][13][64], pre-scaled by 2^-6.  One
// workgroup = one pooled row of one image: the 64 x 13 results go through LDS so that the stores are contiguous 128-byte
// pixels.  (No zero halo: the CONV7 GEMM's loader predicates the taps that fall outside — the halo'd 19 x 19 image was
// 2.1x the bytes to write and to fetch.)
template <int S>
__global__ __launch_bounds__(256) void k_pool_split_cl(const float *__restrict__ x, f16 *__restrict__ oh, f16 *__restrict__ ol)
{
    constexpr int PS = S / 2, C = 64;
    __shared__ float raw[C * 3 * S];                                     // [c][dy][x]: the three input rows of every channel
    __shared__ f16 th[PS * C], tl[PS * C];                               // [px][c]
    const int py = blockIdx.x % PS;
    const int64_t b = blockIdx.x / PS;
    const float *img = x + b * C * S * S;
    // rows 2 py - 1 .. 2 py + 1 of all channels as 8-byte loads, consecutive lanes on consecutive pairs of a row
    for (int i = threadIdx.x; i < C * 3 * (S / 2); i += 256) {
        const int q = i % (S / 2), rowi = i / (S / 2), dy = rowi % 3, c = rowi / 3;
        const int yy = 2 * py + dy - 1;
        float2 v = make_float2(0.0f, 0.0f);
        if (yy >= 0) v = *reinterpret_cast<const float2 *>(img + ((size_t)c * S + yy) * S + 2 * q);   // yy <= S - 1 always
        raw[(c * 3 + dy) * S + 2 * q] = v.x;
        raw[(c * 3 + dy) * S + 2 * q + 1] = v.y;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * PS; i += 256) {
        const int c = i % C, px = i / C;                                 // lanes walk the channels: th / tl rows are contiguous
        float sum = 0.0f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = 2 * px + dx;
                if (xx >= 0) sum += raw[(c * 3 + dy) * S + xx];          // xx <= S - 1 always
            }
        f16 h, l;
        split(sum * (1.0f / 9.0f) * ACT_SCALE, h, l);
        th[px * C + c] = h;
        tl[px * C + c] = l;
    }
    __syncthreads();
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const size_t row = ((size_t)b * PS + py) * PS * C;                   // the row's 13 pixels x 64 channels are contiguous
    for (int i = threadIdx.x; i < PS * C / 2; i += 256) {
        *reinterpret_cast<f16x2 *>(oh + row + 2 * i) = (f16x2){th[2 * i], th[2 * i + 1]};
        *reinterpret_cast<f16x2 *>(ol + row + 2 * i) = (f16x2){tl[2 * i], tl[2 * i + 1]};
    }
}

// ---- the same two poolings reading the trunk's output as the PX16 image of csrc/tron_conv_ws.hip ----------------------
// PX16: per image [hi | lo][channel octet][pixel][8 channels] f16, value / 64 = hi + lo 2^-11 — already in this file's
// scale, so the pooled value is the average of (hi + lo 2^-11) over the window, split again.
typedef _Float16 f16x8h __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void px16_window_sum(const unsigned char *img, int S, int half_bytes, int oct, int py, int px, float (&sum)[8])
{
    // all 18 loads are issued before the first use: positions outside the image are clamped and weighted 0
    f16x8h h[9], l[9];
    float wgt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int yy = 2 * py + t / 3 - 1, xx = 2 * px + t % 3 - 1;
        const bool in = yy >= 0 && yy < S && xx >= 0 && xx < S;
        const int yc = yy < 0 ? 0 : (yy >= S ? S - 1 : yy), xc = xx < 0 ? 0 : (xx >= S ? S - 1 : xx);
        const unsigned char *p = img + ((size_t)oct * S * S + yc * S + xc) * 16;
        h[t] = *reinterpret_cast<const f16x8h *>(p);
        l[t] = *reinterpret_cast<const f16x8h *>(p + half_bytes);
        wgt[t] = in ? 1.0f : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        // (tap by tap in f32: the arithmetic tron_conv_ws_kernel.hpp's WS_POOL pass repeats on the same values, bit for bit)
        float acc = 0.0f;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc += wgt[t] * ((float)h[t][j] + (float)l[t][j] * LO_UNSCALE);
        sum[j] = acc;
    }
}

// 12x12: rows [B][64 * 36] with the columns in (octet, pooled pixel, channel) order (dense7_split's `octets` order): a
// thread pools one (octet, pooled pixel) — eight channels — and writes their 16 bytes per half; no LDS, no barrier.
__global__ __launch_bounds__(256) void k_pool_split12_px(const unsigned char *__restrict__ x, int64_t B, f16 *__restrict__ oh,
                                                         f16 *__restrict__ ol)
{
    constexpr int S = 12, C = 64, HALF = (C / 8) * S * S * 16;
    const int64_t total = B * 288;                                       // (image, octet, pooled pixel)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / 288;
        const int r = (int)(i - b * 288), oct = r / 36, pp = r - oct * 36;
        float sum[8];
        px16_window_sum(x + (size_t)b * 2 * HALF, S, HALF, oct, pp / 6, pp % 6, sum);
        f16x8h h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f16 hh, ll;
            split(sum[j] * (1.0f / 9.0f), hh, ll);
            h[j] = hh;
            l[j] = ll;
        }
        *reinterpret_cast<f16x8h *>(oh + (size_t)i * 8) = h;
        *reinterpret_cast<f16x8h *>(ol + (size_t)i * 8) = l;
    }
}

// 12x12, the training path's pooled planes f32 [B][64][6][6] (k_pool12's output) from conv6's PX16 image: the learner's trunk ends
// in a PX16 image, never in f32 planes.  One thread = one pooled pixel x one channel octet.
__global__ __launch_bounds__(256) void k_pool12_from_px(const unsigned char *__restrict__ x, int64_t B, float *__restrict__ y)
{
    constexpr int S = 12, PS = 6, C = 64, HALF = (C / 8) * S * S * 16;
    const int64_t total = B * 8 * PS * PS;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i % (PS * PS)), py = r / PS, px = r % PS;
        const int64_t bo = i / (PS * PS), b = bo >> 3;
        const int oct = (int)(bo & 7);
        float sum[8];
        px16_window_sum(x + (size_t)b * 2 * HALF, S, HALF, oct, py, px, sum);
#pragma unroll
        for (int j = 0; j < 8; ++j) y[((size_t)b * C + oct * 8 + j) * (PS * PS) + r] = sum[j] * (64.0f / 9.0f);   // (the image carries value / 64)
    }
}

// 26x26: the channels-last split image [B][13][13][64] (what k_pool_split_cl<26> writes).  A workgroup = 64 pooled pixels x 8
// octets, a WAVE = one octet: a wave's loads of one tap sit 32 bytes apart in ONE octet plane (whole cache lines over a row's three
// taps) — with a thread = (pooled pixel, octet) and the octet fastest they were 16 bytes in each of eight planes, re-fetched from L2
// tap after tap (252 us at 4 096 images; this: 222).  The 16-byte results go through LDS so that the stores are the pixels'
// contiguous 128 bytes (a thread walking the eight octets itself, storing 16 bytes at a 128-byte stride, was slower than either).
__global__ __launch_bounds__(512) void k_pool_split26_px(const unsigned char *__restrict__ x, int64_t B, f16 *__restrict__ oh,
                                                           f16 *__restrict__ ol)
{
    constexpr int S = 26, PS = 13, C = 64, HALF = (C / 8) * S * S * 16, ROW = 144;
    __shared__ __attribute__((aligned(16))) unsigned char th[64 * ROW], tl[64 * ROW];
    const int lane = threadIdx.x & 63, oct = threadIdx.x >> 6;
    const int64_t total = B * PS * PS, pix0 = (int64_t)blockIdx.x * 64, pix = pix0 + lane;
    if (pix < total) {
        const int64_t b = pix / (PS * PS);
        const int r = (int)(pix - b * (PS * PS)), py = r / PS, px = r % PS;
        float sum[8];
        px16_window_sum(x + (size_t)b * 2 * HALF, S, HALF, oct, py, px, sum);
        f16x8h h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f16 hh, ll;
            split(sum[j] * (1.0f / 9.0f), hh, ll);
            h[j] = hh;
            l[j] = ll;
        }
        *reinterpret_cast<f16x8h *>(th + lane * ROW + oct * 16) = h;
        *reinterpret_cast<f16x8h *>(tl + lane * ROW + oct * 16) = l;
    }
    __syncthreads();
    const int p = threadIdx.x >> 3, part = threadIdx.x & 7;
    if (pix0 + p < total) {
        *reinterpret_cast<f16x8h *>(oh + (size_t)(pix0 + p) * C + part * 8) = *reinterpret_cast<const f16x8h *>(th + p * ROW + part * 16);
        *reinterpret_cast<f16x8h *>(ol + (size_t)(pix0 + p) * C + part * 8) = *reinterpret_cast<const f16x8h *>(tl + p * ROW + part * 16);
    }
}

// conv7's weight [64][64][7][7] -> [co][tap = 7 ky + kx][ci], split: the W matrix of the CONV7 GEMM
__device__ __forceinline__ void conv7w_split(const float *__restrict__ w, f16 *__restrict__ oh, f16 *__restrict__ ol, int first, int step)
{
    const int total = 64 * 49 * 64;
    for (int i = first; i < total; i += step) {
        const int co = i / (49 * 64), r = i - co * (49 * 64), tap = r / 64, ci = r - tap * 64;
        f16 h, l;
        split(w[((size_t)co * 64 + ci) * 49 + tap], h, l);
        oh[i] = h;
        ol[i] = l;
    }
}

// fc1's weight [256][64 * 49] (columns in NCHW flatten order co * 49 + p) -> columns in the CONV7 GEMM's output order
// p * 64 + co, split
__device__ __forceinline__ void fc1_split_nhwc(const float *__restrict__ w, f16 *__restrict__ oh, f16 *__restrict__ ol, int first, int step)
{
    const int total = 256 * 3136;
    for (int i = first; i < total; i += step) {
        const int n = i / 3136, k = i - n * 3136, p = k / 64, co = k - p * 64;
        f16 h, l;
        split(w[(size_t)n * 3136 + co * 49 + p], h, l);
        oh[i] = h;
        ol[i] = l;
    }
}

// Every weight of the head in the form its GEMM wants, in ONE launch (blockIdx.y = which): conv7 (dense matrix at 12x12,
// [co][tap][ci] at 26x26), fc1 (columns permuted at 26x26), fc2, actor1 — four launches of ~5 us each otherwise, per forward.
struct HeadWeights {
    const float *conv7, *fc1, *fc2, *actor1;
    f16 *d7h, *d7l, *w1h, *w1l, *w2h, *w2l, *w3h, *w3l;
    int side26, octets, n_fc1;
};
__global__ void k_head_weights(HeadWeights a)
{
    const int first = blockIdx.x * blockDim.x + threadIdx.x, step = gridDim.x * blockDim.x;
    switch (blockIdx.y) {
    case 0:
        if (a.side26) conv7w_split(a.conv7, a.d7h, a.d7l, first, step);
        else dense7_split(a.conv7, 64, 64, 6, 3, a.octets, a.d7h, a.d7l, first, step);
        break;
    case 1:
        if (a.side26) fc1_split_nhwc(a.fc1, a.w1h, a.w1l, first, step);
        else split_rows(a.fc1, a.n_fc1, a.w1h, a.w1l, first, step);
        break;
    case 2: split_rows(a.fc2, 128 * 256, a.w2h, a.w2l, first, step); break;
    default: split_rows(a.actor1, 64 * 128, a.w3h, a.w3l, first, step); break;
    }
}

constexpr int GM = 128, GN = 64, GK = 64;          // workgroup tile, K chunk
constexpr int GPITCH = GK * 2 + 32;                // bytes per LDS row: 32 mod 64 is conflict-free for ds_read_b128 by (row = lane % 16,
                                                   // column = lane / 16) with the instruction's lane groups; 16 mod 32 (144) is two-way
constexpr int G_THREADS = 512;
constexpr int A_HALF = GM * GPITCH, W_HALF = GN * GPITCH;
constexpr int G_LDS = 2 * A_HALF + 2 * W_HALF;     // 61 440 bytes

// C[M][N] = act((A W^T) * 64 + bias[n / bias_div]); A = Ah + Al 2^-11 (pre-scaled by 2^-6), W = Wh + Wl 2^-11, all f16
// row-major with K contiguous.  N % 64 == 0, K % 64 == 0.  out_f32 and / or (out_h, out_l) (pre-scaled by 2^-6 again).
//
// CONV7 (24x24 boards): the same GEMM as the 7x7 / stride 2 / pad 3 convolution of 13x13 pooled planes — A is then the
// channels-last split image [image][13 x 13 pixels][64 ci] (k_pool_split_cl), row m = (image, oy, ox), and K chunk kc is
// the 64 channels of pixel (2 oy + ky - 3, 2 ox + kx - 3), (ky, kx) = (kc / 7, kc % 7) — or zeros where that pixel is
// conv7's padding (30 % of the pieces: not fetched): an implicit GEMM whose im2col is a few integer operations per staged
// piece.  W is conv7's weight as [co][tap][ci] (conv7w_split); the output rows [image][oy][ox][co] are the next GEMM's A
// rows as they are.
// DGRAD7 (training, tron_pool_conv7_bwd): conv7's input gradient as four such GEMMs, one per parity class (iy & 1, ix & 1) of
// the pooled pixel — tap ky reaches input row iy from output row (iy + 3 - ky) / 2, so a row's taps all have the parity of
// iy + 3: 3 x 3, 3 x 4, 4 x 3 and 4 x 4 taps for the classes (0,0) (0,1) (1,0) (1,1), no MFMA spent on the 3/4 of the 49 taps
// that cannot reach a pixel.  A is the gradient at conv7's output, channels-last split [image][OS x OS][64 co]; row m =
// (image, yy, xx) of the class (iy = 2 yy + py), chunk kc = (jy, jx) is output pixel (yy + (py ? 2 : 1) - jy, ...) or zeros
// outside; W is [ci][class tap][co] (conv7w_dgrad_split); output row m lands at pooled pixel (iy, ix) of a channels-last f32 image.
enum { G_PLAIN = 0, G_CONV7 = 1, G_DGRAD7 = 2 };
template <int PS>
struct G7 {
    static constexpr int P = PS, PIX = PS * PS, O = (PS + 1) / 2, OPIX = O * O;
};
constexpr int P7 = 13, P7_PIX = P7 * P7, O7 = 7, O7_PIX = O7 * O7;
struct RowGeo {                       // what a_piece_bytes needs of an A row: fixed over the K loop
    int b, y, x;
};
template <int MODE, int PS>
__device__ __forceinline__ RowGeo row_geo(int m, int cls)
{
    RowGeo r{m, 0, 0};
    if constexpr (MODE == G_CONV7) {
        using G = G7<PS>;
        r.b = m / G::OPIX;
        const int p = m - r.b * G::OPIX;
        r.y = p / G::O;
        r.x = p - r.y * G::O;
    } else if constexpr (MODE == G_DGRAD7) {
        const int ny = (cls & 2) ? PS / 2 : (PS + 1) / 2, nx = (cls & 1) ? PS / 2 : (PS + 1) / 2;
        r.b = m / (ny * nx);
        const int p = m - r.b * (ny * nx);
        r.y = p / nx;
        r.x = p - r.y * nx;
    }
    return r;
}
// byte offset of the 128-byte K chunk kc of an A row, or -1: the tap falls on zero padding (CONV7) / outside the output (DGRAD7)
template <int MODE, int PS>
__device__ __forceinline__ int64_t a_piece_bytes(const RowGeo &r, int kc, int K, int cls)
{
    if constexpr (MODE == G_PLAIN) return (int64_t)r.b * K * 2 + kc * GK * 2;
    else if constexpr (MODE == G_CONV7) {
        using G = G7<PS>;
        const int ky = kc / 7, kx = kc - ky * 7;
        const int iy = 2 * r.y + ky - 3, ix = 2 * r.x + kx - 3;
        if ((unsigned)iy >= (unsigned)G::P || (unsigned)ix >= (unsigned)G::P) return -1;
        return ((int64_t)r.b * G::PIX + iy * G::P + ix) * 128;
    } else {
        using G = G7<PS>;
        const int tx = (cls & 1) ? 4 : 3;
        const int jy = kc / tx, jx = kc - jy * tx;
        const int oy = r.y + ((cls & 2) ? 2 : 1) - jy, ox = r.x + ((cls & 1) ? 2 : 1) - jx;
        if ((unsigned)oy >= (unsigned)G::O || (unsigned)ox >= (unsigned)G::O) return -1;
        return ((int64_t)r.b * G::OPIX + oy * G::O + ox) * 128;
    }
}

template <int MODE, int PS>
__global__ __launch_bounds__(G_THREADS, 4) void k_gemm_f16x3(const f16 *__restrict__ Ah, const f16 *__restrict__ Al,
                                                             const f16 *__restrict__ Wh, const f16 *__restrict__ Wl,
                                                             const float *__restrict__ bias, int bias_div, int M, int N,
                                                             int K, int act, float *__restrict__ out_f32,
                                                             f16 *__restrict__ out_h, f16 *__restrict__ out_l,
                                                             const float *__restrict__ a_scale, int cls)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *a_h = lds, *a_l = lds + A_HALF, *w_h = lds + 2 * A_HALF, *w_l = w_h + W_HALF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 3, wn = wave >> 2, li = lane & 15, g = lane >> 4;
    const int nblocks = N / GN;
    const int m0 = (blockIdx.x / nblocks) * GM, n0 = (blockIdx.x % nblocks) * GN;   // neighbours share the A rows in L2

    // staging: 16-byte pieces; A: [2 halves][128 rows][8 pieces], W: [2 halves][64 rows][8 pieces]
    f32x4 ra[4], rw[2];
    RowGeo rg[2];                                                         // (pieces j and j + 2 of a thread are the two halves of one row)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int m = m0 + ((tid + j * G_THREADS) & 1023) / 8;
        m = m < M ? m : M - 1;
        rg[j] = row_geo<MODE, PS>(m, cls);
    }
    auto load_chunk = [&](int kc) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = tid + j * G_THREADS, half = q >> 10, pc = q & 7;
            const int64_t ab = a_piece_bytes<MODE, PS>(rg[j & 1], kc, K, cls);
            ra[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (MODE == G_PLAIN || ab >= 0) ra[j] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const unsigned char *>(half ? Al : Ah) + ab + pc * 16);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = tid + j * G_THREADS, half = q >> 9, r = q & 511, row = r >> 3, pc = r & 7;
            rw[j] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const unsigned char *>(half ? Wl : Wh) +
                                                     ((size_t)(n0 + row) * K + (size_t)kc * GK) * 2 + pc * 16);
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = tid + j * G_THREADS, half = q >> 10, r = q & 1023, row = r >> 3, pc = r & 7;
            *reinterpret_cast<f32x4 *>(lds + half * A_HALF + row * GPITCH + pc * 16) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = tid + j * G_THREADS, half = q >> 9, r = q & 511, row = r >> 3, pc = r & 7;
            *reinterpret_cast<f32x4 *>(lds + 2 * A_HALF + half * W_HALF + row * GPITCH + pc * 16) = rw[j];
        }
    };

    f32x4 acc0[2][2], acc1[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            acc0[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc1[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    const int a_off = (wm * 32 + li) * GPITCH + g * 16, b_off = (wn * 32 + li) * GPITCH + g * 16;
    const int nk = K / GK;
    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        if (kc + 1 < nk) load_chunk(kc + 1);                            // in flight under the MFMAs
#pragma unroll
        for (int s = 0; s < GK / 32; ++s) {
            f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[t] = *reinterpret_cast<const f16x8 *>(a_h + a_off + t * 16 * GPITCH + s * 64);
                al[t] = *reinterpret_cast<const f16x8 *>(a_l + a_off + t * 16 * GPITCH + s * 64);
                bh[t] = *reinterpret_cast<const f16x8 *>(w_h + b_off + t * 16 * GPITCH + s * 64);
                bl[t] = *reinterpret_cast<const f16x8 *>(w_l + b_off + t * 16 * GPITCH + s * 64);
            }
            // hi*lo, hi*hi, lo*hi over the four tiles each: the lo*hi MFMA adds to the accumulator the hi*lo one wrote,
            // and a dependent MFMA issues 48 cycles after its producer — eight MFMAs apart nobody waits
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc1[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bl[n], acc1[t][n], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc0[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bh[n], acc0[t][n], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc1[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh[n], acc1[t][n], 0, 0, 0);
        }
        __syncthreads();                                                 // everyone is done reading this chunk
        if (kc + 1 < nk) {
            store_chunk();
            __syncthreads();
        }
    }

    // (A arrives pre-scaled by 2^-6 — and, tron_gemm_f16x3's gradient operands, by the power of two *a_scale on top)
    const float out_scale = a_scale ? ACT_UNSCALE / *a_scale : ACT_UNSCALE;
    // epilogue: D row = 4 * (lane >> 4) + r, column = lane & 15
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int col = n0 + wn * 32 + n * 16 + li;
        const float bv = bias ? bias[col / bias_div] : 0.0f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 v = (acc0[t][n] + acc1[t][n] * LO_UNSCALE) * out_scale + bv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 32 + t * 16 + 4 * g + r;
                if (m >= M) continue;
                const float y = act ? mish1(v[r]) : v[r];
                if constexpr (MODE == G_DGRAD7) {                        // row m of the class -> its pooled pixel, channels-last
                    const RowGeo o = row_geo<MODE, PS>(m, cls);
                    const int iy = 2 * o.y + ((cls & 2) ? 1 : 0), ix = 2 * o.x + (cls & 1);
                    out_f32[((size_t)o.b * G7<PS>::PIX + iy * PS + ix) * N + col] = y;
                    continue;
                }
                if (out_f32) out_f32[(size_t)m * N + col] = y;
                if (out_h) {
                    f16 hh, ll;
                    split(y * ACT_SCALE, hh, ll);
                    out_h[(size_t)m * N + col] = hh;
                    out_l[(size_t)m * N + col] = ll;
                }
            }
        }
    }
}


// ---- the same product on a 128 x 192 tile (N % 192 == 0: conv7's dense form, N = 576) --------------------------------------
// k_gemm_f16x3's 128 x 64 tile pulls (128 + 64) rows of K through L2 -> LDS per 128 x 64 outputs; this one (128 + 192) rows per
// 128 x 192: 1.8x less feed per output, and a wave's A fragments serve six column tiles instead of two (LDS reads per MFMA 0.44
// against 0.67).  8 waves = 4 (rows) x 2 (columns), a wave 32 x 96 outputs = 2 x 6 MFMA tiles, 96 accumulator registers.
// splitk > 1: slice blockIdx.y of the K chunks, raw sums to partial[slice][M][N] (k_gemm_finish adds them in slice order).
constexpr int WN = 192, WW_HALF = WN * GPITCH;
constexpr int GW_LDS = 2 * A_HALF + 2 * WW_HALF;    // 102 400 bytes
__global__ __launch_bounds__(G_THREADS, 2) void k_gemm_wide(const f16 *__restrict__ Ah, const f16 *__restrict__ Al,
                                                            const f16 *__restrict__ Wh, const f16 *__restrict__ Wl,
                                                            const float *__restrict__ bias, int bias_div, int M, int N, int K, int act,
                                                            float *__restrict__ out_f32, f16 *__restrict__ out_h, f16 *__restrict__ out_l,
                                                            const float *__restrict__ a_scale, int splitk, float *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *a_h = lds, *a_l = lds + A_HALF, *w_h = lds + 2 * A_HALF, *w_l = w_h + WW_HALF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 3, wn = wave >> 2, li = lane & 15, g = lane >> 4;
    const int nblocks = N / WN;
    const int m0 = (blockIdx.x / nblocks) * GM, n0 = (blockIdx.x % nblocks) * WN;

    f32x4 ra[4], rw[6];
    auto load_chunk = [&](int kc) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {                                    // A: [2 halves][128 rows][8 pieces]
            const int q = tid + j * G_THREADS, half = q >> 10, r = q & 1023, row = r >> 3, pc = r & 7;
            int m = m0 + row;
            m = m < M ? m : M - 1;
            ra[j] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const unsigned char *>(half ? Al : Ah) + ((size_t)m * K + (size_t)kc * GK) * 2 + pc * 16);
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {                                    // W: [2 halves][192 rows][8 pieces]
            const int q = tid + j * G_THREADS, half = q >= 1536 ? 1 : 0, r = q - half * 1536, row = r >> 3, pc = r & 7;
            rw[j] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const unsigned char *>(half ? Wl : Wh) + ((size_t)(n0 + row) * K + (size_t)kc * GK) * 2 + pc * 16);
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = tid + j * G_THREADS, half = q >> 10, r = q & 1023, row = r >> 3, pc = r & 7;
            *reinterpret_cast<f32x4 *>(lds + half * A_HALF + row * GPITCH + pc * 16) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int q = tid + j * G_THREADS, half = q >= 1536 ? 1 : 0, r = q - half * 1536, row = r >> 3, pc = r & 7;
            *reinterpret_cast<f32x4 *>(lds + 2 * A_HALF + half * WW_HALF + row * GPITCH + pc * 16) = rw[j];
        }
    };

    f32x4 acc0[2][6], acc1[2][6];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            acc0[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc1[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    const int a_off = (wm * 32 + li) * GPITCH + g * 16, b_off = (wn * 96 + li) * GPITCH + g * 16;
    const int nk_all = K / GK, per = (nk_all + splitk - 1) / splitk;
    const int kb = (int)blockIdx.y * per, nk = kb + per < nk_all ? kb + per : nk_all;
    if (kb < nk) {
        load_chunk(kb);
        store_chunk();
    }
    __syncthreads();
    for (int kc = kb; kc < nk; ++kc) {
        if (kc + 1 < nk) load_chunk(kc + 1);                            // in flight under the MFMAs
#pragma unroll
        for (int s = 0; s < GK / 32; ++s) {
            f16x8 ah[2], al[2], bh[6], bl[6];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[t] = *reinterpret_cast<const f16x8 *>(a_h + a_off + t * 16 * GPITCH + s * 64);
                al[t] = *reinterpret_cast<const f16x8 *>(a_l + a_off + t * 16 * GPITCH + s * 64);
            }
#pragma unroll
            for (int n = 0; n < 6; ++n) {
                bh[n] = *reinterpret_cast<const f16x8 *>(w_h + b_off + n * 16 * GPITCH + s * 64);
                bl[n] = *reinterpret_cast<const f16x8 *>(w_l + b_off + n * 16 * GPITCH + s * 64);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int n = 0; n < 6; ++n) acc1[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bl[n], acc1[t][n], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int n = 0; n < 6; ++n) acc0[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bh[n], acc0[t][n], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int n = 0; n < 6; ++n) acc1[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh[n], acc1[t][n], 0, 0, 0);
        }
        __syncthreads();
        if (kc + 1 < nk) {
            store_chunk();
            __syncthreads();
        }
    }
    const float out_scale = a_scale ? ACT_UNSCALE / *a_scale : ACT_UNSCALE;
#pragma unroll
    for (int n = 0; n < 6; ++n) {
        const int col = n0 + wn * 96 + n * 16 + li;
        const float bv = bias ? bias[col / bias_div] : 0.0f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 raw = acc0[t][n] + acc1[t][n] * LO_UNSCALE;
            const f32x4 v = raw * out_scale + bv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 32 + t * 16 + 4 * g + r;
                if (m >= M) continue;
                if (partial) {
                    partial[((size_t)blockIdx.y * M + m) * N + col] = raw[r];
                    continue;
                }
                const float y = act ? mish1(v[r]) : v[r];
                if (out_f32) out_f32[(size_t)m * N + col] = y;
                if (out_h) {
                    f16 hh, ll;
                    split(y * ACT_SCALE, hh, ll);
                    out_h[(size_t)m * N + col] = hh;
                    out_l[(size_t)m * N + col] = ll;
                }
            }
        }
    }
}

// the K slices of k_gemm_wide added in slice order, then its epilogue; four columns per thread
__global__ void k_gemm_finish(const float *__restrict__ partial, int splitk, int64_t M, int N, const float *__restrict__ bias,
                              int bias_div, int act, const float *__restrict__ a_scale, float *__restrict__ out_f32,
                              f16 *__restrict__ out_h, f16 *__restrict__ out_l)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4, total = M * N;
    if (i >= total) return;
    f32x4 v = *reinterpret_cast<const f32x4 *>(partial + i);
    for (int s = 1; s < splitk; ++s) v += *reinterpret_cast<const f32x4 *>(partial + (size_t)s * total + i);
    const float out_scale = a_scale ? ACT_UNSCALE / *a_scale : ACT_UNSCALE;
    const int col = (int)(i % N);
    f32x4 y;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float x = v[r] * out_scale + (bias ? bias[(col + r) / bias_div] : 0.0f);
        y[r] = act ? mish1(x) : x;
    }
    if (out_f32) *reinterpret_cast<f32x4 *>(out_f32 + i) = y;
    if (out_h) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f16 hh, ll;
            split(y[r] * ACT_SCALE, hh, ll);
            out_h[i + r] = hh;
            out_l[i + r] = ll;
        }
    }
}

// actor2: q[b][a] = x[b] . w[a] + bias[a] in f32 (64 x 4 MACs per sample), and the greedy action
__global__ void k_q_head(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias, int B,
                         int K, float *__restrict__ q, int8_t *__restrict__ greedy)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float acc[4] = {bias[0], bias[1], bias[2], bias[3]};
    for (int k = 0; k < K; k += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(x + (size_t)b * K + k);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float4 ww = *reinterpret_cast<const float4 *>(w + a * K + k);
            acc[a] = __fmaf_rn(v.x, ww.x, acc[a]);
            acc[a] = __fmaf_rn(v.y, ww.y, acc[a]);
            acc[a] = __fmaf_rn(v.z, ww.z, acc[a]);
            acc[a] = __fmaf_rn(v.w, ww.w, acc[a]);
        }
    }
    if (q) *reinterpret_cast<float4 *>(q + (size_t)b * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    if (greedy) {                                                        // first maximum, like torch.argmax / np.argmax
        int best = 0;
#pragma unroll
        for (int a = 1; a < 4; ++a)
            if (acc[a] > acc[best]) best = a;
        greedy[b] = (int8_t)best;
    }
}

inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

// Which launches take k_gemm_wide, and in how many K slices (0: k_gemm_f16x3).  Measured (profiles/r04_head_gemm_wide.txt): the wide
// tile wins where K is long — conv7's dense forward (K = 2 304: 89 -> 75 us at 4 096 rows, 150 -> 121 at 8 192, 1 035 -> 804 at
// 65 536) and its weight gradient (K = the batch: 99 -> 80 us) — and loses at K = 576 (conv7's input gradient: 68 -> 74 us).
// Few tiles are cut along K so that most of the 256 CUs have one: the slices' raw sums go through `partial` and k_gemm_finish.
inline int wide_splitk(int64_t M, int N, int K)
{
    if (N % WN != 0 || K % GK != 0 || K < 1024 || M >= (1ll << 31)) return 0;
    const int64_t tiles = ((M + GM - 1) / GM) * (N / WN);
    const int nk = K / GK;
    if (tiles < 32) return 0;
    return (tiles < 80 && nk >= 16) ? 4 : ((tiles < 160 && nk >= 8) ? 2 : 1);
}
inline int64_t wide_partial_bytes(int64_t M, int N, int K)
{
    const int s = wide_splitk(M, N, K);
    return s > 1 ? align256((int64_t)s * M * N * 4) : 0;
}


struct HeadPlan {                      // byte offsets into the workspace
    int64_t a7h, a7l, d7h, d7l, c7h, c7l, w1h, w1l, c1h, c1l, w2h, w2l, c2h, c2l, w3h, w3l, c3, part, total;
};
HeadPlan plan(int64_t B, int K7, int N7, int64_t a7_per_image = 0, int d7_rows = 0)
{
    HeadPlan p{};
    int64_t o = 0;
    auto take = [&](int64_t bytes) { const int64_t at = o; o = align256(o + bytes); return at; };
    p.a7h = take(B * (a7_per_image ? a7_per_image : K7) * 2); p.a7l = take(B * (a7_per_image ? a7_per_image : K7) * 2);
    p.d7h = take((int64_t)(d7_rows ? d7_rows : N7) * K7 * 2); p.d7l = take((int64_t)(d7_rows ? d7_rows : N7) * K7 * 2);
    p.c7h = take(B * N7 * 2); p.c7l = take(B * N7 * 2);
    p.w1h = take(256ll * N7 * 2); p.w1l = take(256ll * N7 * 2);
    p.c1h = take(B * 256 * 2); p.c1l = take(B * 256 * 2);
    p.w2h = take(128 * 256 * 2); p.w2l = take(128 * 256 * 2);
    p.c2h = take(B * 128 * 2); p.c2l = take(B * 128 * 2);
    p.w3h = take(64 * 128 * 2); p.w3l = take(64 * 128 * 2);
    p.c3 = take(B * 64 * 4);
    p.part = take(a7_per_image ? 0 : wide_partial_bytes(B, N7, K7));     // (conv7's dense form: the K slices of k_gemm_wide)
    p.total = o;
    return p;
}

template <int MODE, int PS = 13>
int gemm(const f16 *Ah, const f16 *Al, const f16 *Wh, const f16 *Wl, const float *bias, int bias_div, int64_t M, int N, int K,
         int act, float *out_f32, f16 *out_h, f16 *out_l, hipStream_t st, const float *a_scale = nullptr, int cls = 0,
         float *partial_ws = nullptr)
{
    static uint64_t prepared = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_NO_DEVICE; }
    if (!(prepared & (1ull << (dev & 63)))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_f16x3<MODE, PS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                G_LDS) != hipSuccess)
            (void)hipGetLastError();
        prepared |= 1ull << (dev & 63);
    }
    if (M >= (1ll << 31) || N % GN != 0 || K % GK != 0) return TRON_ERR_UNSUPPORTED;
    if (M == 0) return TRON_OK;
    if (MODE == G_PLAIN) {
        const int splitk = wide_splitk(M, N, K);
        if (splitk == 1 || (splitk > 1 && partial_ws)) {
            static uint64_t prepared_wide = 0;
            if (!(prepared_wide & (1ull << (dev & 63)))) {
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_wide), hipFuncAttributeMaxDynamicSharedMemorySize, GW_LDS) != hipSuccess)
                    (void)hipGetLastError();
                prepared_wide |= 1ull << (dev & 63);
            }
            const int64_t tiles = ((M + GM - 1) / GM) * (N / WN);
            hipLaunchKernelGGL(k_gemm_wide, dim3((unsigned)tiles, splitk), dim3(G_THREADS), GW_LDS, st, Ah, Al, Wh, Wl, bias, bias_div, (int)M, N, K, act,
                               out_f32, out_h, out_l, a_scale, splitk, splitk > 1 ? partial_ws : nullptr);
            if (splitk > 1)
                hipLaunchKernelGGL(k_gemm_finish, dim3((unsigned)((M * N / 4 + 255) / 256)), dim3(256), 0, st, partial_ws, splitk, M, N, bias, bias_div,
                                   act, a_scale, out_f32, out_h, out_l);
            return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
        }
    }
    const int64_t blocks = ((M + GM - 1) / GM) * (N / GN);
    hipLaunchKernelGGL((k_gemm_f16x3<MODE, PS>), dim3((unsigned)blocks), dim3(G_THREADS), G_LDS, st, Ah, Al, Wh, Wl, bias, bias_div,
                       (int)M, N, K, act, out_f32, out_h, out_l, a_scale, cls);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

// ---- operands of tron_gemm_f16x3: f32 matrices -> split f16 [rows][K padded to 64], K contiguous --------------------------
// as given ([rows][K], K % 64 == 0) ...
__global__ void k_split_scaled(const float *__restrict__ a, int64_t total, const float *__restrict__ scale, float fixed, f16 *__restrict__ oh,
                               f16 *__restrict__ ol)
{
    const float sc = fixed * (scale ? *scale : 1.0f);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        f16 h, l;
        split(a[i] * sc, h, l);
        oh[i] = h;
        ol[i] = l;
    }
}
// ... or transposed (given as [K][rows]): a 64 x 64 tile through LDS, K zero-padded to kpad
__global__ __launch_bounds__(256) void k_transpose_split_scaled(const float *__restrict__ a, int64_t K, int rows, int64_t kpad,
                                                                const float *__restrict__ scale, float fixed, f16 *__restrict__ oh,
                                                                f16 *__restrict__ ol)
{
    __shared__ float tile[64][65];
    const float sc = fixed * (scale ? *scale : 1.0f);
    const int64_t k0 = (int64_t)blockIdx.x * 64;
    const int j0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int kk = i >> 6, jj = i & 63;
        tile[kk][jj] = (k0 + kk < K && j0 + jj < rows) ? a[(size_t)(k0 + kk) * rows + j0 + jj] * sc : 0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 8; i += 256) {                   // (row jj, 8 consecutive k)
        const int jj = i >> 3, g = i & 7;
        if (j0 + jj >= rows) continue;
        f16x8 h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            f16 hh, ll;
            split(tile[g * 8 + e][jj], hh, ll);
            h[e] = hh;
            l[e] = ll;
        }
        *reinterpret_cast<f16x8 *>(oh + (size_t)(j0 + jj) * kpad + k0 + g * 8) = h;
        *reinterpret_cast<f16x8 *>(ol + (size_t)(j0 + jj) * kpad + k0 + g * 8) = l;
    }
}

// ---- training side of the same layers at 24x24 / 32x32 boards (tron_pool_conv7_fwd / _bwd) -------------------------------
// conv7's weight [64][64][7][7] -> the four DGRAD7 matrices [class][ci][class tap = jy * tx + jx][co], split; class c = 2 py + px
// starts at 64 * 64 * {0, 9, 21, 33} elements; tap (ky, kx) = ((py ? 0 : 1) + 2 jy, (px ? 0 : 1) + 2 jx)
__global__ void k_conv7w_dgrad_split(const float *__restrict__ w, f16 *__restrict__ oh, f16 *__restrict__ ol)
{
    const int total = 49 * 64 * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int cls = i < 9 * 4096 ? 0 : (i < 21 * 4096 ? 1 : (i < 33 * 4096 ? 2 : 3));
        const int base = (cls == 0 ? 0 : (cls == 1 ? 9 : (cls == 2 ? 21 : 33))) * 4096;
        const int ty = (cls & 2) ? 4 : 3, tx = (cls & 1) ? 4 : 3, taps = ty * tx;
        const int r = i - base, ci = r / (taps * 64), q = r - ci * (taps * 64), tap = q >> 6, co = q & 63;
        const int jy = tap / tx, jx = tap - jy * tx;
        const int ky = ((cls & 2) ? 0 : 1) + 2 * jy, kx = ((cls & 1) ? 0 : 1) + 2 * jx;
        f16 h, l;
        split(w[((size_t)co * 64 + ci) * 49 + ky * 7 + kx], h, l);
        oh[i] = h;
        ol[i] = l;
    }
}
__global__ void k_conv7w_fwd_split(const float *__restrict__ w, f16 *__restrict__ oh, f16 *__restrict__ ol)
{
    conv7w_split(w, oh, ol, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// y[b][co * OPIX + p] = mish(pre[b][p][co]): conv7's pre-activation rows (the CONV7 GEMM's output order) -> the flattened NCHW
// activation fc1 reads (DQNNet.py:55: x.view(-1, 64 * 7 * 7)).  One image per workgroup pass, through LDS.
template <int OPIX>
__global__ __launch_bounds__(256) void k_mish_cl_to_nchw(const float *__restrict__ pre, float *__restrict__ y, int B, int act)
{
    __shared__ float t[OPIX * 65];
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const float *src = pre + (size_t)b * OPIX * 64;
        for (int i = threadIdx.x; i < OPIX * 64; i += 256) t[(i >> 6) * 65 + (i & 63)] = act ? mish1(src[i]) : src[i];
        __syncthreads();
        float *dst = y + (size_t)b * OPIX * 64;
        for (int i = threadIdx.x; i < OPIX * 64; i += 256) {
            const int co = i / OPIX, pp = i - co * OPIX;
            dst[i] = t[pp * 65 + co];
        }
        __syncthreads();
    }
}

__device__ __forceinline__ float mish_grad1(float x)                    // d mish / dx, as csrc/tron_conv_f16.hip's mish_grad4
{
    const uint32_t eb = __float_as_uint(__builtin_amdgcn_exp2f(x * 1.44269504088896341f));
    const float e = __uint_as_float(eb < 0x5D5E0B6Bu ? eb : 0x5D5E0B6Bu);      // e^x capped at 1e18: t is exactly 1 up there
    const float n = __fmaf_rn(e, e, e + e);                              // tanh(softplus x) = n / (n + 2)
    const float r = __builtin_amdgcn_rcpf(n + 2.0f), q = __builtin_amdgcn_rcpf(e + 1.0f);
    const float t = n * r;
    return t + x * ((r + r) * (1.0f + t)) * (e * q);                     // t + x (1 - t^2) sigmoid(x)
}

// gp[b][p][co] = gy[b][co * OPIX + p] * mish'(pre[b][p][co]), written channels-last split (scaled by 2^-6 and the power of two
// *scale: the DGRAD7 GEMMs' and the weight-gradient kernel's operand) — and its sum over (b, p) per channel, one partial row
// per workgroup (k_colsum_finish adds them in order).
template <int OPIX>
__global__ __launch_bounds__(256) void k_mish_bwd_to_cl(const float *__restrict__ gy, const float *__restrict__ pre, int B,
                                                        const float *__restrict__ scale, f16 *__restrict__ oh, f16 *__restrict__ ol,
                                                        float *__restrict__ partial)
{
    __shared__ float t[OPIX * 65];
    __shared__ float red[256];
    const float sc = *scale * ACT_SCALE;
    const int co = threadIdx.x & 63;
    float sum = 0.0f;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const float *src = gy + (size_t)b * OPIX * 64;
        for (int i = threadIdx.x; i < OPIX * 64; i += 256) {
            const int c = i / OPIX, pp = i - c * OPIX;
            t[pp * 65 + c] = src[i];
        }
        __syncthreads();
        const float *pr = pre + (size_t)b * OPIX * 64;
        for (int i = threadIdx.x; i < OPIX * 64; i += 256) {            // i & 63 == co for every i of this thread
            const float g = pre ? t[(i >> 6) * 65 + co] * mish_grad1(pr[i]) : t[(i >> 6) * 65 + co];   // (pre == NULL: no activation behind the convolution)
            sum += g;
            f16 h, l;
            split(g * sc, h, l);
            oh[(size_t)b * OPIX * 64 + i] = h;
            ol[(size_t)b * OPIX * 64 + i] = l;
        }
        __syncthreads();
    }
    red[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x < 64) partial[blockIdx.x * 64 + co] = (red[co] + red[co + 64]) + (red[co + 128] + red[co + 192]);
}
__global__ __launch_bounds__(1024) void k_colsum_finish(const float *__restrict__ partial, int rows, float *__restrict__ out)
{
    __shared__ float red[1024];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;                // sixteen threads per column, each a fixed sixteenth of the rows
    float a0 = 0.f, a1 = 0.f;                                            // (independent chains: a dependent load -> add chain is ~65 ns a link)
    int r = q;
    for (; r + 16 < rows; r += 32) {
        a0 += partial[r * 64 + c];
        a1 += partial[(r + 16) * 64 + c];
    }
    if (r < rows) a0 += partial[r * 64 + c];
    red[threadIdx.x] = a0 + a1;
    __syncthreads();
    if (q == 0) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            s0 += red[c + 128 * k];
            s1 += red[c + 128 * k + 64];
        }
        out[c] = s0 + s1;
    }
}

// gx[b][c][y][x] = (1/9) sum of gpool[b][py][px][c] over the pooling windows that contain (y, x): AvgPool2d(3, 2, 1)'s backward
// from the channels-last pooled gradient (the DGRAD7 GEMMs' output) to the NCHW planes the trunk's backward reads.  One
// workgroup = 16 channels of one image: their planes are written as one contiguous run of whole cache lines.
template <int S>
__global__ __launch_bounds__(256) void k_pool_bwd_cl(const float *__restrict__ gpool, float *__restrict__ gx)
{
    constexpr int PS = S / 2, HW = S * S;
    __shared__ float t[(PS * PS + 1) * 17];                              // [pooled pixel][16 channels (+1: bank spread)], + a pixel of zeros
    const int cg = blockIdx.x & 3;
    const int64_t b = blockIdx.x >> 2;
    for (int i = threadIdx.x; i < PS * PS * 16; i += 256) {
        const int c = i & 15, p = i >> 4;
        t[p * 17 + c] = gpool[((size_t)b * PS * PS + p) * 64 + 16 * cg + c];
    }
    if (threadIdx.x < 17) t[PS * PS * 17 + threadIdx.x] = 0.0f;
    __syncthreads();
    float *dst = gx + ((size_t)b * 64 + 16 * cg) * HW;
    for (int i = threadIdx.x; i < 16 * HW / 2; i += 256) {               // (channel, row, pair of columns): consecutive float2
        const int q = i % (S / 2), r = i / (S / 2), y = r % S, c = r / S;
        // even row y: window row y / 2 only; odd: (y - 1) / 2 and (y + 1) / 2.  Column 2 q: window column q; 2 q + 1: q and q + 1.
        const int p0 = y >> 1, p1 = ((y & 1) && p0 + 1 < PS) ? p0 + 1 : -1;
        const int q1 = q + 1 < PS ? q + 1 : -1;
        const int Z = PS * PS;                                           // the pixel of zeros
        const int i00 = p0 * PS + q, i01 = q1 >= 0 ? p0 * PS + q1 : Z, i10 = p1 >= 0 ? p1 * PS + q : Z, i11 = (p1 >= 0 && q1 >= 0) ? p1 * PS + q1 : Z;
        const float a0 = t[i00 * 17 + c] + t[i10 * 17 + c], a1 = t[i01 * 17 + c] + t[i11 * 17 + c];
        *reinterpret_cast<float2 *>(dst + 2 * i) = make_float2(a0 * (1.0f / 9.0f), (a0 + a1) * (1.0f / 9.0f));
    }
}

// The convolution alone (tron_conv7_fwd / _bwd: the module the K-FAC hooks sit on takes and returns NCHW tensors): pooled planes
// f32 [B][64][PIX] -> the channels-last split image [B][PIX][64] (pre-scaled by 2^-6), and a channels-last f32 gradient back to
// NCHW, through LDS.
template <int PIX>
__global__ __launch_bounds__(256) void k_nchw_to_cl_split(const float *__restrict__ x, f16 *__restrict__ oh, f16 *__restrict__ ol)
{
    // one workgroup = 32 pixels x all 64 channels: 128-byte reads per channel, one contiguous 4 KB run of whole pixels per half
    // (a 16-channel slab per workgroup wrote every 128-byte pixel in four pieces at four different times: 0.9 TB/s)
    constexpr int NBLK = (PIX + 31) / 32;
    __shared__ float t[64 * 33];
    const int p0 = ((int)blockIdx.x % NBLK) * 32;
    const int64_t b = blockIdx.x / NBLK;
    const float *src = x + (size_t)b * 64 * PIX;
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {
        const int c = i >> 5, p = i & 31;
        t[c * 33 + p] = p0 + p < PIX ? src[c * PIX + p0 + p] : 0.0f;
    }
    __syncthreads();
    const int oct = threadIdx.x & 7, p = threadIdx.x >> 3;
    if (p0 + p >= PIX) return;
    f16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        f16 hh, ll;
        split(t[(oct * 8 + j) * 33 + p] * ACT_SCALE, hh, ll);
        h[j] = hh;
        l[j] = ll;
    }
    const size_t o = ((size_t)b * PIX + p0 + p) * 64 + 8 * oct;
    *reinterpret_cast<f16x8 *>(oh + o) = h;
    *reinterpret_cast<f16x8 *>(ol + o) = l;
}
template <int PIX>
__global__ __launch_bounds__(256) void k_cl_to_nchw(const float *__restrict__ g, float *__restrict__ out)
{
    __shared__ float t[PIX * 17];
    const int cg = blockIdx.x & 3;
    const int64_t b = blockIdx.x >> 2;
    for (int i = threadIdx.x; i < PIX * 16; i += 256) t[(i >> 4) * 17 + (i & 15)] = g[((size_t)b * PIX + (i >> 4)) * 64 + 16 * cg + (i & 15)];
    __syncthreads();
    float *dst = out + ((size_t)b * 64 + 16 * cg) * PIX;
    for (int i = threadIdx.x; i < 16 * PIX; i += 256) {
        const int c = i / PIX, p = i - c * PIX;
        dst[i] = t[p * 17 + c];
    }
}

// conv7's WEIGHT gradient, dW[co][ci][ky][kx] = sum over images and output pixels m = (oy, ox) of
//     gp[m][co] * pooled[2 oy + ky - 3][2 ox + kx - 3][ci]
// — a product over the SLOW index of two channels-last operands (both are [pixel][64 channels] rows), so the MFMA operands
// (8 consecutive k per lane for a fixed channel) are read out of LDS transposed: ds_read_b64_tr_b16 hands every lane of a
// 16-lane group four k-rows of its channel column, each row's address supplied by one lane — which also makes conv7's
// stride-2 im2col free: the lane that supplies row m of the pooled operand points at pixel (2 oy + ky - 3, 2 ox + kx - 3), or
// at a row of zeros where that is padding.  No transposed or im2col copy of either operand is ever written.
//   * a workgroup (4 waves, one per SIMD with the whole register file) owns ONE ky and all seven kx, a slice of the images, all
//     64 x 64 channel pairs: wave w = input channels 16 w .. 16 w + 15 x four 16-channel output tiles x seven taps = 28 tiles,
//     224 accumulator registers; the gradient fragments of a k-slab are read once and used for the seven taps;
//   * per image it needs the gradient rows (OPIX x 64, hi and lo) and the pooled rows iy = 2 oy + ky - 3 (the valid ones of O
//     rows x PS pixels), 36 KB at 13x13: copied global -> LDS by LDS-DMA into one of two stages while the other is multiplied,
//     one barrier per image.  Pooled pixels are stored even columns first, then odd, so that consecutive ox are consecutive
//     LDS rows for either parity of kx.  LDS rows are 160 bytes apart (128 of data): the eight consecutive rows a 32-lane
//     half reads then start on eight different bank octets — and every address is base + immediate;
//   * the k of a slab that lane (i, g) holds in element e is 16 (e >> 2) + 4 g + (e & 3) — any assignment works as long as
//     both operands use it, and this one makes each transposed read's two 16-lane groups of a half fetch rows 8 h .. 8 h + 7;
//   * K per image = OPIX pixels padded to a multiple of 32 with rows of zeros (49 -> 64: 1.3x; 81 -> 96: 1.19x).
// Per-(slice, ky) partial sums, added in a fixed order and folded to [co][ci][7][7] by k_conv7_wgrad_finish (deterministic).
constexpr int W7_THREADS = 256, W7_SLICES = 36, W7_PITCH = 160, W7_PCS = W7_PITCH / 16;
template <int PS>
struct W7 {
    static constexpr int O = (PS + 1) / 2, OPIX = O * O, PIX = PS * PS;
    static constexpr int SLABS = (OPIX + 31) / 32;
    static constexpr int GROWS = OPIX + 1, PROWS = O * PS + 1;           // + the row of zeros
    static constexpr int ROWS = 2 * GROWS + 2 * PROWS, STAGE = ROWS * W7_PITCH, LDS = 2 * STAGE;
    static constexpr int NSLOT = (ROWS * W7_PCS + W7_THREADS - 1) / W7_THREADS;
    static constexpr int G_LO = GROWS * W7_PITCH, P_BASE = 2 * GROWS * W7_PITCH, P_LO = PROWS * W7_PITCH;
    static_assert(LDS <= 160 * 1024 && NSLOT <= 32, "layout");
};
// ds_read_b64_tr_b16 by inline asm (EXEC must be all ones).  Behind __builtin_amdgcn_ds_read_tr16_b64 hipcc (ROCm 7.2) puts an
// s_waitcnt vmcnt(0) in front of the first transposed read that follows an LDS-DMA — the next image's copy, issued right before
// the multiply, was waited for before the multiply began: no overlap (found on csrc/tron_conv_ws_train.hip's k_wgrad_px, round 4).
// The asm is invisible to that pass, so the waits are written here: a step's reads go out one step ahead, `lds_tr_wait()`
// (s_waitcnt lgkmcnt(0) + sched_barrier) stands between a read and the first use of its registers, and the two 8-byte halves
// are joined into the MFMA operand only after that wait.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4 lds_tr(uint32_t addr)
{
    s16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=&v"(r) : "v"(addr));
    return r;
}
__device__ __forceinline__ void lds_tr_wait()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ f16x8 join8(s16x4 a, s16x4 b)
{
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(f16x8, v);
}

template <int PS>
__global__ __launch_bounds__(W7_THREADS) void k_conv7_wgrad(const f16 *__restrict__ gph, const f16 *__restrict__ gpl,
                                                            const f16 *__restrict__ ph, const f16 *__restrict__ pl, int B, int nsl,
                                                            float *__restrict__ partial)
{
    using C = W7<PS>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, g = lane >> 4;
    const int ky = (int)blockIdx.x % 7, sl = (int)blockIdx.x / 7;
    const int nimg = sl < B ? (B - sl + nsl - 1) / nsl : 0;              // this workgroup's images: sl, sl + nsl, ...

    for (int i = tid * 16; i < C::LDS; i += W7_THREADS * 16) *reinterpret_cast<f32x4 *>(lds + i) = (f32x4){0.f, 0.f, 0.f, 0.f};

    // what each DMA slot of this thread copies: LDS piece P = slot * 256 + tid (linear: row P / 10, 16-byte position P % 10,
    // positions 8 and 9 are the padding).  slot_src: the source for this workgroup's first image, or nullptr: the piece stays zero
    const unsigned char *slot_src[C::NSLOT];
    uint32_t pooled_slots = 0;                                            // bit j: slot j reads a pooled array (its image stride)
#pragma unroll
    for (int j = 0; j < C::NSLOT; ++j) {
        const int P = j * W7_THREADS + tid, row = P / W7_PCS, pos = P - row * W7_PCS;
        const unsigned char *src = nullptr;
        if (pos < 8 && row < 2 * C::GROWS) {
            const int lo = row >= C::GROWS, r = row - lo * C::GROWS;
            if (r < C::OPIX) src = reinterpret_cast<const unsigned char *>(lo ? gpl : gph) + ((size_t)sl * C::OPIX + r) * 128 + pos * 16;
        } else if (pos < 8 && row < C::ROWS) {
            const int q = row - 2 * C::GROWS, lo = q >= C::PROWS, r = q - lo * C::PROWS;
            const int oy = r / PS, c = r - oy * PS, ix = c < C::O ? 2 * c : 2 * (c - C::O) + 1, iy = 2 * oy + ky - 3;
            if (r < C::O * PS && iy >= 0 && iy < PS) {
                src = reinterpret_cast<const unsigned char *>(lo ? pl : ph) + ((size_t)sl * C::PIX + iy * PS + ix) * 128 + pos * 16;
                pooled_slots |= 1u << j;
            }
        }
        slot_src[j] = src;
    }
    auto dma_image = [&](int stage) {                                     // copies the next image of this workgroup, then steps the sources
#pragma unroll
        for (int j = 0; j < C::NSLOT; ++j)
            if (slot_src[j]) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)slot_src[j],
                                                 (__attribute__((address_space(3))) void *)(lds + stage * C::STAGE + j * (W7_THREADS * 16) + wave * 1024),
                                                 16, 0, 0);
                slot_src[j] += (size_t)nsl * (((pooled_slots >> j) & 1u) ? C::PIX * 128 : C::OPIX * 128);
            }
    };

    // LDS byte offsets (within a stage) of the rows this lane supplies to the transposed reads: k = 32 s + 16 j + 4 g + (li >> 2)
    int a_addr[C::SLABS][2], b_addr[7][C::SLABS][2];
#pragma unroll
    for (int s = 0; s < C::SLABS; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = 32 * s + 16 * j + 4 * g + (li >> 2);
            const int mm = m < C::OPIX ? m : C::OPIX;                    // beyond the image: the row of zeros
            a_addr[s][j] = mm * W7_PITCH + (li & 3) * 8;                 // output tile t: + 32 t
            const int oy = m / C::O, ox = m - oy * C::O;
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const int ix = 2 * ox + kx - 3;
                const bool ok = m < C::OPIX && ix >= 0 && ix < PS;
                const int R = ok ? oy * PS + ((ix & 1) ? C::O + (ix >> 1) : (ix >> 1)) : C::O * PS;
                b_addr[kx][s][j] = C::P_BASE + R * W7_PITCH + (4 * wave + (li & 3)) * 8;
            }
        }

    f32x4 acc0[7][4], acc1[7][4];
#pragma unroll
    for (int kx = 0; kx < 7; ++kx)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc0[kx][t] = acc1[kx][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // One (slab, tap) step = 12 MFMAs; the next step's pooled fragments (and, at a slab's last tap, the next slab's gradient
    // fragments) are requested first and land under them.  The sched_barrier keeps the compiler from hoisting every read of
    // the image to the top (which it does at one wave per SIMD, and then spills).
    auto multiply = [&](const unsigned char *Sp) {
        const uint32_t S = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)Sp;
        s16x4 ra[2][4][4], rb[2][4];                                     // raw halves: gradient [slab parity][tile][h0 h1 l0 l1], pooled [step parity][h0 h1 l0 l1]
        auto read_a = [&](int slot, int s) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                ra[slot][t][0] = lds_tr(S + a_addr[s][0] + 32 * t);
                ra[slot][t][1] = lds_tr(S + a_addr[s][1] + 32 * t);
                ra[slot][t][2] = lds_tr(S + C::G_LO + a_addr[s][0] + 32 * t);
                ra[slot][t][3] = lds_tr(S + C::G_LO + a_addr[s][1] + 32 * t);
            }
        };
        auto read_b = [&](int slot, int kx, int s) {
            rb[slot][0] = lds_tr(S + b_addr[kx][s][0]);
            rb[slot][1] = lds_tr(S + b_addr[kx][s][1]);
            rb[slot][2] = lds_tr(S + C::P_LO + b_addr[kx][s][0]);
            rb[slot][3] = lds_tr(S + C::P_LO + b_addr[kx][s][1]);
        };
        read_a(0, 0);
        read_b(0, 0, 0);
        f16x8 ah[4], al[4];
#pragma unroll
        for (int s = 0; s < C::SLABS; ++s)
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const int step = s * 7 + kx;
                lds_tr_wait();                                           // this step's operands have landed
                if (kx == 0) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        ah[t] = join8(ra[s & 1][t][0], ra[s & 1][t][1]);
                        al[t] = join8(ra[s & 1][t][2], ra[s & 1][t][3]);
                    }
                }
                const f16x8 ch = join8(rb[step & 1][0], rb[step & 1][1]), cl = join8(rb[step & 1][2], rb[step & 1][3]);
                const int nk = kx == 6 ? 0 : kx + 1, ns = kx == 6 ? s + 1 : s;
                if (ns < C::SLABS) {
                    read_b((step + 1) & 1, nk, ns);
                    if (kx == 6) read_a(ns & 1, ns);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) acc1[kx][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], cl, acc1[kx][t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc0[kx][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], ch, acc0[kx][t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc1[kx][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], ch, acc1[kx][t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
    };

    __syncthreads();                                                     // the zeros are in before any copy lands
    if (nimg > 0) dma_image(0);
    for (int n = 0; n < nimg; ++n) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // image n is in; everybody is done with the other stage
        if (n + 1 < nimg) dma_image((n + 1) & 1);
        multiply(lds + (n & 1) * C::STAGE);
    }

    // D row = 4 g + r (output channel within the tile), column = li (input channel): partial[slice][ky][kx][co][ci]
    float *out = partial + ((size_t)(sl * 7 + ky) * 7) * 4096;
#pragma unroll
    for (int kx = 0; kx < 7; ++kx)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4 v = acc0[kx][t] + acc1[kx][t] * LO_UNSCALE;
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)kx * 4096 + (16 * t + 4 * g + r) * 64 + 16 * wave + li] = v[r];
        }
}

// dW[co][ci][tap] = (2^12 / scale) * sum over the slices of partial[slice][tap][co][ci]   (both operands carry 2^-6, the
// gradient the power of two *scale on top)
__global__ void k_conv7_wgrad_finish(const float *__restrict__ partial, int nsl, const float *__restrict__ scale, float *__restrict__ gw)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;                 // (tap, co, ci), ci fastest
    if (i >= 49 * 4096) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int s = 0;
    for (; s + 3 < nsl; s += 4) {
        a0 += partial[(size_t)(s + 0) * 49 * 4096 + i];
        a1 += partial[(size_t)(s + 1) * 49 * 4096 + i];
        a2 += partial[(size_t)(s + 2) * 49 * 4096 + i];
        a3 += partial[(size_t)(s + 3) * 49 * 4096 + i];
    }
    for (; s < nsl; ++s) a0 += partial[(size_t)s * 49 * 4096 + i];
    const int tap = i >> 12, coci = i & 4095;
    gw[(size_t)coci * 49 + tap] = ((a0 + a1) + (a2 + a3)) * (4096.0f / *scale);
}

}  // namespace

// C[M][N] = A B^T + bias[n] on the split-f16 matrix cores (three MFMAs per slab, f32 accumulation: f32-grade), f32 in and out.
// A: [M][K], or (a_transposed) given as [K][M]; B: [N][K], or (b_transposed) given as [K][N]; N % 64 == 0; K % 64 == 0 for an
// operand given K-contiguous (a transposed one is zero-padded).  a_scale (may be NULL): one f32 on the device, a power of two
// A is multiplied by on its way into f16 and C divided by — gradient operands.  workspace: tron_gemm_f16x3_workspace bytes.
extern "C" int64_t tron_gemm_f16x3_workspace(int64_t M, int32_t N, int64_t K)
{
    if (M < 1 || N < 1 || K < 1 || N % 64 != 0 || M >= (1ll << 31) || K >= (1ll << 31)) return 0;
    const int64_t kpad = (K + 63) / 64 * 64;
    return 2 * align256(M * kpad * 2) + 2 * align256((int64_t)N * kpad * 2) + wide_partial_bytes(M, N, kpad) + 256;
}

extern "C" int tron_gemm_f16x3(const float *A, int32_t a_transposed, const float *B, int32_t b_transposed, const float *bias,
                               const float *a_scale, float *C, int64_t M, int32_t N, int64_t K, void *workspace, void *stream)
{
    if (!A || !B || !C || !workspace || M < 0 || N < 1 || K < 1) return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(C)) & 15u) return TRON_ERR_BAD_ARG;
    if (M == 0) return TRON_OK;
    if (N % 64 != 0 || M >= (1ll << 31) || K >= (1ll << 31)) return TRON_ERR_UNSUPPORTED;
    if ((!a_transposed || !b_transposed) && K % 64 != 0) return TRON_ERR_UNSUPPORTED;
    const int64_t kpad = (K + 63) / 64 * 64;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    unsigned char *ws = reinterpret_cast<unsigned char *>(workspace);
    const int64_t abytes = align256(M * kpad * 2), bbytes = align256((int64_t)N * kpad * 2);
    f16 *ah = reinterpret_cast<f16 *>(ws), *al = reinterpret_cast<f16 *>(ws + abytes);
    f16 *bh = reinterpret_cast<f16 *>(ws + 2 * abytes), *bl = reinterpret_cast<f16 *>(ws + 2 * abytes + bbytes);
    auto blocks_for = [](int64_t total) { const int64_t b = (total + 255) / 256; return (unsigned)(b < 8192 ? b : 8192); };
    if (a_transposed)
        hipLaunchKernelGGL(k_transpose_split_scaled, dim3((unsigned)(kpad / 64), (unsigned)((M + 63) / 64)), dim3(256), 0, st, A, K, (int)M, kpad,
                           a_scale, ACT_SCALE, ah, al);
    else
        hipLaunchKernelGGL(k_split_scaled, dim3(blocks_for(M * K)), dim3(256), 0, st, A, M * K, a_scale, ACT_SCALE, ah, al);
    if (b_transposed)
        hipLaunchKernelGGL(k_transpose_split_scaled, dim3((unsigned)(kpad / 64), (unsigned)(N / 64)), dim3(256), 0, st, B, K, N, kpad,
                           (const float *)nullptr, 1.0f, bh, bl);
    else
        hipLaunchKernelGGL(k_split_scaled, dim3(blocks_for((int64_t)N * K)), dim3(256), 0, st, B, (int64_t)N * K, (const float *)nullptr, 1.0f, bh, bl);
    if (hipGetLastError() != hipSuccess) return TRON_ERR_LAUNCH;
    return gemm<G_PLAIN>(ah, al, bh, bl, bias, 1, M, N, (int)kpad, 0, C, nullptr, nullptr, st, a_scale, 0,
                         reinterpret_cast<float *>(ws + 2 * abytes + 2 * bbytes));
}


extern "C" int64_t tron_dqn_head_workspace(int64_t batch, int32_t side)
{
    if (batch < 1 || (side != 12 && side != 26)) return 0;
    if (side == 26) return plan(batch, 64 * 49, 64 * 49, (int64_t)P7_PIX * 64, 64).total;   // (K7 = conv7's 49 taps x 64 ci)
    return plan(batch, 64 * 6 * 6, 64 * 3 * 3).total;
}

// src: 0 = f32 planes, 1 = PX16 image, 2 = the pooled rows tron_conv3x3_ws_fwd_pool12 wrote (12x12 only)
static int head_fwd(const void *trunk, int src, int64_t batch, int32_t side, const float *conv7_w,
                    const float *conv7_b, const float *fc1_w, const float *fc1_b, const float *fc2_w,
                    const float *fc2_b, const float *actor1_w, const float *actor1_b, const float *actor2_w,
                    const float *actor2_b, void *workspace, float *q_out, int8_t *greedy_out, void *stream)
{
    const float *trunk_out = reinterpret_cast<const float *>(trunk);
    const unsigned char *trunk_px = reinterpret_cast<const unsigned char *>(trunk);
    if (!trunk_out || !conv7_w || !conv7_b || !fc1_w || !fc1_b || !fc2_w || !fc2_b || !actor1_w || !actor1_b ||
        !actor2_w || !actor2_b || !workspace || (!q_out && !greedy_out) || batch < 0)
        return TRON_ERR_BAD_ARG;
    if (batch == 0) return TRON_OK;
    if ((side != 12 && side != 26) || batch > (side == 12 ? (1ll << 24) : (1ll << 22)) || (src == 2 && side != 12)) return TRON_ERR_UNSUPPORTED;
    const bool px16 = src != 0;
    if ((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(trunk_out) | reinterpret_cast<uintptr_t>(q_out)) & 15u)
        return TRON_ERR_BAD_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (side == 26) {                                                    // 13x13 pooled planes, 7x7 after conv7: fc1 takes 64 * 49
        constexpr int K7 = 64 * 49, N1 = 64 * 49;
        const HeadPlan p = plan(batch, K7, N1, (int64_t)P7_PIX * 64, 64);
        unsigned char *ws = reinterpret_cast<unsigned char *>(workspace);
        auto H = [&](int64_t off) { return reinterpret_cast<f16 *>(ws + off); };
        if (px16) hipLaunchKernelGGL(k_pool_split26_px, dim3((unsigned)((batch * 169 + 63) / 64)), dim3(512), 0, st, trunk_px, batch, H(p.a7h), H(p.a7l));
        else hipLaunchKernelGGL(k_pool_split_cl<26>, dim3((unsigned)(batch * 13)), dim3(256), 0, st, trunk_out, H(p.a7h), H(p.a7l));
        const HeadWeights hw{conv7_w, fc1_w, fc2_w, actor1_w, H(p.d7h), H(p.d7l), H(p.w1h), H(p.w1l), H(p.w2h), H(p.w2l), H(p.w3h), H(p.w3l),
                             1, 0, 256 * N1};
        hipLaunchKernelGGL(k_head_weights, dim3(1024, 4), dim3(256), 0, st, hw);
        if (hipGetLastError() != hipSuccess) return TRON_ERR_LAUNCH;
        // conv7 as the implicit GEMM [B * 49] x [64] over K = 49 taps x 64 channels; its output rows are fc1's input rows
        int rc = gemm<G_CONV7, 13>(H(p.a7h), H(p.a7l), H(p.d7h), H(p.d7l), conv7_b, 1, batch * O7_PIX, 64, K7, 1, nullptr, H(p.c7h), H(p.c7l), st);
        if (rc == TRON_OK) rc = gemm<G_PLAIN>(H(p.c7h), H(p.c7l), H(p.w1h), H(p.w1l), fc1_b, 1, batch, 256, N1, 1, nullptr, H(p.c1h), H(p.c1l), st);
        if (rc == TRON_OK) rc = gemm<G_PLAIN>(H(p.c1h), H(p.c1l), H(p.w2h), H(p.w2l), fc2_b, 1, batch, 128, 256, 1, nullptr, H(p.c2h), H(p.c2l), st);
        float *c3 = reinterpret_cast<float *>(ws + p.c3);
        if (rc == TRON_OK) rc = gemm<G_PLAIN>(H(p.c2h), H(p.c2l), H(p.w3h), H(p.w3l), actor1_b, 1, batch, 64, 128, 1, c3, nullptr, nullptr, st);
        if (rc != TRON_OK) return rc;
        hipLaunchKernelGGL(k_q_head, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, st, c3, actor2_w, actor2_b, (int)batch, 64,
                           q_out, greedy_out);
        return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
    }
    constexpr int C = 64, PS = 6, OS = 3, K7 = C * PS * PS, N7 = C * OS * OS;
    const HeadPlan p = plan(batch, K7, N7);
    unsigned char *ws = reinterpret_cast<unsigned char *>(workspace);
    auto H = [&](int64_t off) { return reinterpret_cast<f16 *>(ws + off); };
    const int64_t nrows = batch * C * PS;
    const f16 *a7h = H(p.a7h), *a7l = H(p.a7l);
    if (src == 2) {
        a7h = reinterpret_cast<const f16 *>(trunk_px);
        a7l = reinterpret_cast<const f16 *>(trunk_px + align256(batch * K7 * 2));
    } else if (px16) hipLaunchKernelGGL(k_pool_split12_px, dim3((unsigned)((batch * 288 + 255) / 256 < (1 << 20) ? (batch * 288 + 255) / 256 : (1 << 20))), dim3(256), 0, st, trunk_px, batch, H(p.a7h), H(p.a7l));
    else hipLaunchKernelGGL(k_pool_split12, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, st, trunk_out, nrows, H(p.a7h), H(p.a7l));
    const HeadWeights hw{conv7_w, fc1_w, fc2_w, actor1_w, H(p.d7h), H(p.d7l), H(p.w1h), H(p.w1l), H(p.w2h), H(p.w2l), H(p.w3h), H(p.w3l),
                         0, px16 ? 1 : 0, 256 * N7};
    hipLaunchKernelGGL(k_head_weights, dim3(1024, 4), dim3(256), 0, st, hw);
    if (hipGetLastError() != hipSuccess) return TRON_ERR_LAUNCH;
    int rc = gemm<G_PLAIN>(a7h, a7l, H(p.d7h), H(p.d7l), conv7_b, OS * OS, batch, N7, K7, 1, nullptr, H(p.c7h), H(p.c7l), st, nullptr, 0,
                           reinterpret_cast<float *>(ws + p.part));
    if (rc == TRON_OK) rc = gemm<G_PLAIN>(H(p.c7h), H(p.c7l), H(p.w1h), H(p.w1l), fc1_b, 1, batch, 256, N7, 1, nullptr, H(p.c1h), H(p.c1l), st);
    if (rc == TRON_OK) rc = gemm<G_PLAIN>(H(p.c1h), H(p.c1l), H(p.w2h), H(p.w2l), fc2_b, 1, batch, 128, 256, 1, nullptr, H(p.c2h), H(p.c2l), st);
    float *c3 = reinterpret_cast<float *>(ws + p.c3);
    if (rc == TRON_OK) rc = gemm<G_PLAIN>(H(p.c2h), H(p.c2l), H(p.w3h), H(p.w3l), actor1_b, 1, batch, 64, 128, 1, c3, nullptr, nullptr, st);
    if (rc != TRON_OK) return rc;
    hipLaunchKernelGGL(k_q_head, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, st, c3, actor2_w, actor2_b, (int)batch, 64,
                       q_out, greedy_out);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

extern "C" int tron_dqn_head_fwd(const float *trunk_out, int64_t batch, int32_t side, const float *conv7_w,
                                 const float *conv7_b, const float *fc1_w, const float *fc1_b, const float *fc2_w,
                                 const float *fc2_b, const float *actor1_w, const float *actor1_b, const float *actor2_w,
                                 const float *actor2_b, void *workspace, float *q_out, int8_t *greedy_out, void *stream)
{
    return head_fwd(trunk_out, 0, batch, side, conv7_w, conv7_b, fc1_w, fc1_b, fc2_w, fc2_b, actor1_w, actor1_b, actor2_w,
                    actor2_b, workspace, q_out, greedy_out, stream);
}

extern "C" int tron_dqn_head_fwd_px16(const void *trunk_px16, int64_t batch, int32_t side, const float *conv7_w,
                                      const float *conv7_b, const float *fc1_w, const float *fc1_b, const float *fc2_w,
                                      const float *fc2_b, const float *actor1_w, const float *actor1_b,
                                      const float *actor2_w, const float *actor2_b, void *workspace, float *q_out,
                                      int8_t *greedy_out, void *stream)
{
    return head_fwd(trunk_px16, 1, batch, side, conv7_w, conv7_b, fc1_w, fc1_b, fc2_w, fc2_b, actor1_w, actor1_b, actor2_w,
                    actor2_b, workspace, q_out, greedy_out, stream);
}

// the head from the pooled rows of tron_conv3x3_ws_fwd_pool12 (10x10 boards: conv6 + pooling were one launch); workspace as above
extern "C" int tron_dqn_head_fwd_pooled(const void *pooled, int64_t batch, int32_t side, const float *conv7_w,
                                        const float *conv7_b, const float *fc1_w, const float *fc1_b, const float *fc2_w,
                                        const float *fc2_b, const float *actor1_w, const float *actor1_b,
                                        const float *actor2_w, const float *actor2_b, void *workspace, float *q_out,
                                        int8_t *greedy_out, void *stream)
{
    return head_fwd(pooled, 2, batch, side, conv7_w, conv7_b, fc1_w, fc1_b, fc2_w, fc2_b, actor1_w, actor1_b, actor2_w,
                    actor2_b, workspace, q_out, greedy_out, stream);
}

// ---- training at 24x24 / 32x32 boards: pool + conv7 + mish (DQNNet.py:52-54; the actor-critic nets' ACNet.py tail alike) with
// everything on this file's kernels, forward and backward (Net/activations.py::_PoolConv7CL) --------------------------------
namespace {
struct P7Plan {
    int64_t out4, gph, gpl, wh, wl, gpool, bpart, wpart, pre, total;
};
P7Plan p7_plan(int64_t B, int PS)
{
    const int O = (PS + 1) / 2;
    P7Plan p{};
    int64_t o = 0;
    auto take = [&](int64_t bytes) { const int64_t at = o; o = align256(o + bytes); return at; };
    p.out4 = take(256);
    p.wh = take(49ll * 4096 * 2); p.wl = take(49ll * 4096 * 2);
    p.gph = take(B * O * O * 128); p.gpl = take(B * O * O * 128);
    p.gpool = take(B * PS * PS * 256);
    p.bpart = take(1024ll * 64 * 4);
    p.wpart = take((int64_t)W7_SLICES * 49 * 4096 * 4);
    p.pre = take(B * O * O * 256);                                       // (tron_conv7_fwd's channels-last output before the transpose)
    p.total = o;
    return p;
}
inline bool p7_side_ok(int32_t side) { return side == 26 || side == 34; }

// CONVONLY: x is the pooled planes f32 [B][64][PS][PS], y the convolution's output f32 [B][64][O][O] (+ bias, no activation)
template <int S, bool CONVONLY>
int p7_fwd(const float *x, int64_t B, const float *w, const float *bias, void *saved, float *pre, float *y, void *workspace, hipStream_t st,
           const unsigned char *x_px16 = nullptr)
{
    constexpr int PS = S / 2, O = (PS + 1) / 2;
    const P7Plan p = p7_plan(B, PS);
    unsigned char *ws = reinterpret_cast<unsigned char *>(workspace), *sv = reinterpret_cast<unsigned char *>(saved);
    f16 *ph = reinterpret_cast<f16 *>(sv), *pl = reinterpret_cast<f16 *>(sv + align256(B * PS * PS * 128));
    f16 *wh = reinterpret_cast<f16 *>(ws + p.wh), *wl = reinterpret_cast<f16 *>(ws + p.wl);
    if (CONVONLY) {
        pre = reinterpret_cast<float *>(ws + p.pre);
        hipLaunchKernelGGL(k_nchw_to_cl_split<PS * PS>, dim3((unsigned)(B * ((PS * PS + 31) / 32))), dim3(256), 0, st, x, ph, pl);
    } else if (x_px16) {                                                 // the learner's trunk ends in a PX16 image (S = 26 only)
        hipLaunchKernelGGL(k_pool_split26_px, dim3((unsigned)((B * 169 + 63) / 64)), dim3(512), 0, st, x_px16, B, ph, pl);
    } else {
        hipLaunchKernelGGL(k_pool_split_cl<S>, dim3((unsigned)(B * PS)), dim3(256), 0, st, x, ph, pl);   // (a 16-channel-slab form that reads each plane once, contiguously, was 11 % slower: LDS-bound)
    }
    hipLaunchKernelGGL(k_conv7w_fwd_split, dim3(784), dim3(256), 0, st, w, wh, wl);
    if (hipGetLastError() != hipSuccess) return TRON_ERR_LAUNCH;
    const int rc = gemm<G_CONV7, PS>(ph, pl, wh, wl, bias, 1, B * O * O, 64, 64 * 49, 0, pre, nullptr, nullptr, st);
    if (rc != TRON_OK) return rc;
    hipLaunchKernelGGL(k_mish_cl_to_nchw<O * O>, dim3((unsigned)(B < 4096 ? B : 4096)), dim3(256), 0, st, pre, y, (int)B, CONVONLY ? 0 : 1);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

template <int S, bool CONVONLY>
int p7_bwd(const float *gy, const float *pre, const void *saved, const float *w, int64_t B, float *gx, float *gw, float *gb, void *workspace,
           hipStream_t st, float *gpool_out = nullptr)
{
    constexpr int PS = S / 2, O = (PS + 1) / 2, OPIX = O * O;
    const P7Plan p = p7_plan(B, PS);
    unsigned char *ws = reinterpret_cast<unsigned char *>(workspace);
    const unsigned char *sv = reinterpret_cast<const unsigned char *>(saved);
    const f16 *ph = reinterpret_cast<const f16 *>(sv), *pl = reinterpret_cast<const f16 *>(sv + align256(B * PS * PS * 128));
    float *out4 = reinterpret_cast<float *>(ws + p.out4), *bpart = reinterpret_cast<float *>(ws + p.bpart);
    f16 *gph = reinterpret_cast<f16 *>(ws + p.gph), *gpl = reinterpret_cast<f16 *>(ws + p.gpl);
    // the gradient's power-of-two scale from max |gy| (|mish'| <= 1.09: one binade of headroom below f16's range is kept)
    if (hipMemsetAsync(out4, 0, 16, st) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_LAUNCH; }
    int rc = tron_absmax_pow2(gy, B * OPIX * 64, 15, out4, st);
    if (rc != TRON_OK) return rc;
    const int bblocks = (int)(B < 1024 ? B : 1024);
    hipLaunchKernelGGL(k_mish_bwd_to_cl<OPIX>, dim3((unsigned)bblocks), dim3(256), 0, st, gy, pre, (int)B, out4, gph, gpl, bpart);
    if (gb) hipLaunchKernelGGL(k_colsum_finish, dim3(1), dim3(1024), 0, st, bpart, bblocks, gb);
    if (hipGetLastError() != hipSuccess) return TRON_ERR_LAUNCH;
    if (gx || gpool_out) {
        f16 *wh = reinterpret_cast<f16 *>(ws + p.wh), *wl = reinterpret_cast<f16 *>(ws + p.wl);
        float *gpool = gpool_out ? gpool_out : reinterpret_cast<float *>(ws + p.gpool);   // (gpool_out: the caller takes the pooled gradient, channels-last)
        hipLaunchKernelGGL(k_conv7w_dgrad_split, dim3(784), dim3(256), 0, st, w, wh, wl);
        if (hipGetLastError() != hipSuccess) return TRON_ERR_LAUNCH;
        const int first[4] = {0, 9, 21, 33};
        for (int cls = 0; cls < 4 && rc == TRON_OK; ++cls) {
            const int ny = (cls & 2) ? PS / 2 : (PS + 1) / 2, nx = (cls & 1) ? PS / 2 : (PS + 1) / 2;
            const int taps = ((cls & 2) ? 4 : 3) * ((cls & 1) ? 4 : 3);
            rc = gemm<G_DGRAD7, PS>(gph, gpl, wh + first[cls] * 4096, wl + first[cls] * 4096, nullptr, 1, B * ny * nx, 64, taps * 64, 0, gpool,
                                    nullptr, nullptr, st, out4, cls);
        }
        if (rc != TRON_OK) return rc;
        if (gpool_out) { /* the pooling's backward is the caller's (tron_px16_grad_from_pooled) */ }
        else if (CONVONLY) hipLaunchKernelGGL(k_cl_to_nchw<PS * PS>, dim3((unsigned)(B * 4)), dim3(256), 0, st, gpool, gx);
        else hipLaunchKernelGGL(k_pool_bwd_cl<S>, dim3((unsigned)(B * 4)), dim3(256), 0, st, gpool, gx);
        if (hipGetLastError() != hipSuccess) return TRON_ERR_LAUNCH;
    }
    if (gw) {
        static uint64_t prepared = 0;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_NO_DEVICE; }
        if (!(prepared & (1ull << (dev & 63)))) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv7_wgrad<PS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    W7<PS>::LDS) != hipSuccess)
                (void)hipGetLastError();
            prepared |= 1ull << (dev & 63);
        }
        const int nsl = (int)(B < W7_SLICES ? B : W7_SLICES);
        float *wpart = reinterpret_cast<float *>(ws + p.wpart);
        hipLaunchKernelGGL(k_conv7_wgrad<PS>, dim3((unsigned)(7 * nsl)), dim3(W7_THREADS), W7<PS>::LDS, st, gph, gpl, ph, pl, (int)B, nsl, wpart);
        hipLaunchKernelGGL(k_conv7_wgrad_finish, dim3(49 * 4096 / 256), dim3(256), 0, st, wpart, nsl, out4, gw);
        if (hipGetLastError() != hipSuccess) return TRON_ERR_LAUNCH;
    }
    return TRON_OK;
}
}  // namespace

extern "C" int64_t tron_pool_conv7_saved_bytes(int64_t batch, int32_t side)
{
    if (batch < 1 || !p7_side_ok(side) || batch > (1ll << 20)) return 0;
    const int PS = side / 2;
    return 2 * align256(batch * PS * PS * 128);
}

extern "C" int64_t tron_pool_conv7_workspace(int64_t batch, int32_t side)
{
    if (batch < 1 || !p7_side_ok(side) || batch > (1ll << 20)) return 0;
    return p7_plan(batch, side / 2).total;
}

extern "C" int tron_pool_conv7_fwd(const float *x, int64_t batch, int32_t side, const float *weight, const float *bias, void *saved,
                                   float *pre, float *y, void *workspace, void *stream)
{
    if (!x || !weight || !bias || !saved || !pre || !y || !workspace || batch < 0) return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(saved) | reinterpret_cast<uintptr_t>(pre) | reinterpret_cast<uintptr_t>(y) |
         reinterpret_cast<uintptr_t>(workspace)) & 15u)
        return TRON_ERR_BAD_ARG;
    if (batch == 0) return TRON_OK;
    if (!p7_side_ok(side) || batch > (1ll << 20)) return TRON_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return side == 26 ? p7_fwd<26, false>(x, batch, weight, bias, saved, pre, y, workspace, st)
                      : p7_fwd<34, false>(x, batch, weight, bias, saved, pre, y, workspace, st);
}

extern "C" int tron_pool_conv7_bwd(const float *grad_y, const float *pre, const void *saved, const float *weight, int64_t batch, int32_t side,
                                   float *grad_x, float *grad_weight, float *grad_bias, void *workspace, void *stream)
{
    if (!grad_y || !pre || !saved || !weight || !workspace || batch < 0) return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(grad_y) | reinterpret_cast<uintptr_t>(saved) | reinterpret_cast<uintptr_t>(pre) |
         reinterpret_cast<uintptr_t>(grad_x) | reinterpret_cast<uintptr_t>(workspace)) & 15u)
        return TRON_ERR_BAD_ARG;
    if (!p7_side_ok(side) || batch > (1ll << 20)) return TRON_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (batch == 0) {                                                    // nothing to sum: zero parameter gradients
        if (grad_weight && hipMemsetAsync(grad_weight, 0, 49 * 4096 * sizeof(float), st) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_LAUNCH; }
        if (grad_bias && hipMemsetAsync(grad_bias, 0, 64 * sizeof(float), st) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_LAUNCH; }
        return TRON_OK;
    }
    return side == 26 ? p7_bwd<26, false>(grad_y, pre, saved, weight, batch, grad_x, grad_weight, grad_bias, workspace, st)
                      : p7_bwd<34, false>(grad_y, pre, saved, weight, batch, grad_x, grad_weight, grad_bias, workspace, st);
}

// The same two calls for a trunk that ends in a PX16 image (Net/activations.py::_BodyPX, csrc/tron_conv_ws_train.hip): the forward
// pools conv6's PX16 output (26x26 only), the backward stops at the POOLED gradient — grad_pooled f32 [batch][13 x 13][64],
// channels-last — whose pooling backward tron_px16_grad_from_pooled fuses with the activation backward of conv6.
extern "C" int tron_pool_conv7_fwd_px16(const void *x_px16, int64_t batch, int32_t side, const float *weight, const float *bias, void *saved,
                                        float *pre, float *y, void *workspace, void *stream)
{
    if (!x_px16 || !weight || !bias || !saved || !pre || !y || !workspace || batch < 0) return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(x_px16) | reinterpret_cast<uintptr_t>(saved) | reinterpret_cast<uintptr_t>(pre) | reinterpret_cast<uintptr_t>(y) |
         reinterpret_cast<uintptr_t>(workspace)) & 15u)
        return TRON_ERR_BAD_ARG;
    if (batch == 0) return TRON_OK;
    if (side != 26 || batch > (1ll << 20)) return TRON_ERR_UNSUPPORTED;
    return p7_fwd<26, false>(nullptr, batch, weight, bias, saved, pre, y, workspace, reinterpret_cast<hipStream_t>(stream),
                             reinterpret_cast<const unsigned char *>(x_px16));
}

extern "C" int tron_pool_conv7_bwd_pooled(const float *grad_y, const float *pre, const void *saved, const float *weight, int64_t batch,
                                          int32_t side, float *grad_pooled, float *grad_weight, float *grad_bias, void *workspace, void *stream)
{
    if (!grad_y || !pre || !saved || !weight || !grad_pooled || !workspace || batch < 0) return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(grad_y) | reinterpret_cast<uintptr_t>(saved) | reinterpret_cast<uintptr_t>(pre) |
         reinterpret_cast<uintptr_t>(grad_pooled) | reinterpret_cast<uintptr_t>(workspace)) & 15u)
        return TRON_ERR_BAD_ARG;
    if (side != 26 || batch > (1ll << 20)) return TRON_ERR_UNSUPPORTED;
    if (batch == 0) return TRON_OK;
    return p7_bwd<26, false>(grad_y, pre, saved, weight, batch, nullptr, grad_weight, grad_bias, workspace, reinterpret_cast<hipStream_t>(stream),
                             grad_pooled);
}

extern "C" int tron_pool12_px16(const void *x_px16, float *pooled, int64_t batch, void *stream)
{
    if (!x_px16 || !pooled || batch < 0 || (reinterpret_cast<uintptr_t>(x_px16) & 15u)) return TRON_ERR_BAD_ARG;
    if (batch == 0) return TRON_OK;
    const int64_t total = batch * 8 * 36;
    hipLaunchKernelGGL(k_pool12_from_px, dim3((unsigned)((total + 255) / 256 < (1 << 20) ? (total + 255) / 256 : (1 << 20))), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const unsigned char *>(x_px16), batch, pooled);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

// conv7 alone on the same kernels — the nn.Conv2d module itself (Net/activations.py::Conv7), NCHW in and out, for callers that need the
// layer's input and output as tensors: KFACOptimizer's hooks (kfac.py:156-189) on the actor-critic nets' conv7.
extern "C" int tron_conv7_fwd(const float *pooled, int64_t batch, int32_t pooled_side, const float *weight, const float *bias, void *saved,
                              float *y, void *workspace, void *stream)
{
    if (!pooled || !weight || !saved || !y || !workspace || batch < 0) return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(pooled) | reinterpret_cast<uintptr_t>(saved) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(workspace)) & 15u)
        return TRON_ERR_BAD_ARG;
    if (batch == 0) return TRON_OK;
    if ((pooled_side != 13 && pooled_side != 17) || batch > (1ll << 20)) return TRON_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return pooled_side == 13 ? p7_fwd<26, true>(pooled, batch, weight, bias, saved, nullptr, y, workspace, st)
                             : p7_fwd<34, true>(pooled, batch, weight, bias, saved, nullptr, y, workspace, st);
}

extern "C" int tron_conv7_bwd(const float *grad_y, const void *saved, const float *weight, int64_t batch, int32_t pooled_side,
                              float *grad_pooled, float *grad_weight, float *grad_bias, void *workspace, void *stream)
{
    if (!grad_y || !saved || !weight || !workspace || batch < 0) return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(grad_y) | reinterpret_cast<uintptr_t>(saved) | reinterpret_cast<uintptr_t>(grad_pooled) |
         reinterpret_cast<uintptr_t>(workspace)) & 15u)
        return TRON_ERR_BAD_ARG;
    if ((pooled_side != 13 && pooled_side != 17) || batch > (1ll << 20)) return TRON_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (batch == 0) {
        if (grad_weight && hipMemsetAsync(grad_weight, 0, 49 * 4096 * sizeof(float), st) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_LAUNCH; }
        if (grad_bias && hipMemsetAsync(grad_bias, 0, 64 * sizeof(float), st) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_LAUNCH; }
        return TRON_OK;
    }
    return pooled_side == 13 ? p7_bwd<26, true>(grad_y, nullptr, saved, weight, batch, grad_pooled, grad_weight, grad_bias, workspace, st)
                             : p7_bwd<34, true>(grad_y, nullptr, saved, weight, batch, grad_pooled, grad_weight, grad_bias, workspace, st);
}

// ---- training-path pieces of the same layers (Net/activations.py::_PoolConv7): pooling forward / backward on 12x12
// planes and conv7's dense form and its gradient's fold-back; the GEMMs between them are library calls ------------------
extern "C" int tron_pool12(const float *x, float *y, int64_t planes, int32_t backward, void *stream)
{
    if (!x || !y || planes < 0) return TRON_ERR_BAD_ARG;
    if (planes == 0) return TRON_OK;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15u) return TRON_ERR_BAD_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t rows = planes * (backward ? 12 : 6);
    const unsigned blocks = (unsigned)((rows + 255) / 256 < (1 << 20) ? (rows + 255) / 256 : (1 << 20));
    if (backward) hipLaunchKernelGGL(k_pool12_bwd, dim3(blocks), dim3(256), 0, st, x, rows, y);
    else hipLaunchKernelGGL(k_pool12, dim3(blocks), dim3(256), 0, st, x, rows, y);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

extern "C" int tron_pool_s2(const float *x, float *y, int64_t planes, int32_t side, void *stream)
{
    if (!x || !y || planes < 0) return TRON_ERR_BAD_ARG;
    if (planes == 0) return TRON_OK;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15u) return TRON_ERR_BAD_ARG;
    if (side == 12) return tron_pool12(x, y, planes, 0, stream);
    if (side != 26 && side != 34) return TRON_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t rows = planes * (side / 2);
    const unsigned blocks = (unsigned)((rows + 255) / 256 < (1 << 20) ? (rows + 255) / 256 : (1 << 20));
    if (side == 26) hipLaunchKernelGGL(k_pool_rows<26>, dim3(blocks), dim3(256), 0, st, x, rows, y);
    else hipLaunchKernelGGL(k_pool_rows<34>, dim3(blocks), dim3(256), 0, st, x, rows, y);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

extern "C" int tron_pool_s2_bwd(const float *grad_y, float *grad_x, int64_t planes, int32_t side, void *stream)
{
    if (!grad_y || !grad_x || planes < 0) return TRON_ERR_BAD_ARG;
    if (planes == 0) return TRON_OK;
    if ((reinterpret_cast<uintptr_t>(grad_y) | reinterpret_cast<uintptr_t>(grad_x)) & 15u) return TRON_ERR_BAD_ARG;
    if (side == 12) return tron_pool12(grad_y, grad_x, planes, 1, stream);
    if (side != 26 && side != 34) return TRON_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t rows = planes * side;
    const unsigned blocks = (unsigned)((rows + 255) / 256 < (1 << 20) ? (rows + 255) / 256 : (1 << 20));
    if (side == 26) hipLaunchKernelGGL(k_pool_rows_bwd<26>, dim3(blocks), dim3(256), 0, st, grad_y, rows, grad_x);
    else hipLaunchKernelGGL(k_pool_rows_bwd<34>, dim3(blocks), dim3(256), 0, st, grad_y, rows, grad_x);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

extern "C" int tron_conv7_dense(const float *src, float *dst, int32_t cout, int32_t cin, int32_t fold, void *stream)
{
    if (!src || !dst || cout < 1 || cin < 1 || (int64_t)cout * cin * 324 > (1ll << 30)) return TRON_ERR_BAD_ARG;
    const int total = fold ? cout * cin * 49 : cout * 9 * cin * 36;
    hipLaunchKernelGGL(k_dense7, dim3((total + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst, cout, cin,
                       fold);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}
