// tron_kfac.hip — the K-FAC input-patch extraction for the ACKTR path (reference: Net/kfac.py:28-38
// `_extract_patches`, whose own TODO at kfac.py:9-12 asks for a fused kernel).
//
// For a conv layer with input x [B][C][H][W], kernel kh x kw, padding p, stride s, the A-factor is
// E[a a^T] over all (sample, output position) pairs, a = the C*kh*kw input patch under that output.
// torch's F.unfold builds the patch matrix with ONE im2col launch PER SAMPLE (1.1 M launches and 31 %
// of the GPU time of one ACKTR iteration at 16 384 envs x 32x32, rocprofv3 round 1).  This kernel
// writes the GEMM-ready matrix P [B*OH*OW][C*kh*kw] (row = sample-major then output row-major, column
// = c*kh*kw + i*kw + j: F.unfold's channel order, transposed) in one launch per chunk of samples.
//
// One workgroup per (sample, output row): the kh input rows of every channel that this output row
// reads are staged in LDS once, zero-padded, then the OW patch rows go out as coalesced stores —
// consecutive lanes write consecutive columns of a row.  HBM-write bound: reads C*kh*W, writes
// OW*C*kh*kw floats.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tron_hip.h"
#include "tron_kfac_px.hpp"
#include <stdlib.h>

namespace {

__global__ void k_extract_patches(const float *__restrict__ x, int C, int H, int W, int kh, int kw, int pad, int stride,
                                  int OH, int OW, float *__restrict__ out)
{
    extern __shared__ float xs[];                 // [C][kh][Wp], Wp = W + 2*pad
    const int n = blockIdx.x / OH, oy = blockIdx.x - n * OH;
    const int Wp = W + 2 * pad;
    const int d = C * kh * kw;
    const float *xn = x + (size_t)n * C * H * W;
    const int rows = C * kh;
    for (int e = threadIdx.x; e < rows * Wp; e += blockDim.x) {
        const int r = e / Wp, xx = e - r * Wp;
        const int c = r / kh, i = r - c * kh;
        const int iy = oy * stride - pad + i, ix = xx - pad;
        xs[e] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? xn[((size_t)c * H + iy) * W + ix] : 0.0f;
    }
    __syncthreads();
    float *o = out + ((size_t)n * OH + oy) * (size_t)OW * d;
    for (int col = threadIdx.x; col < d; col += blockDim.x) {
        const int c = col / (kh * kw), r = col - c * kh * kw;
        const int i = r / kw, j = r - i * kw;
        const float *src = xs + (c * kh + i) * Wp + j;
        for (int ox = 0; ox < OW; ++ox) o[(size_t)ox * d + col] = src[ox * stride];
    }
}

}  // namespace

extern "C" int tron_extract_patches(const float *x, int64_t batch, int32_t channels, int32_t height, int32_t width,
                                    int32_t kh, int32_t kw, int32_t pad, int32_t stride, float *out, void *stream)
{
    if (!x || !out || batch < 0 || channels < 1 || height < 1 || width < 1 || kh < 1 || kw < 1 || pad < 0 || stride < 1)
        return TRON_ERR_BAD_ARG;
    const int OH = (height + 2 * pad - kh) / stride + 1, OW = (width + 2 * pad - kw) / stride + 1;
    if (OH < 1 || OW < 1) return TRON_ERR_BAD_ARG;
    if (batch == 0) return TRON_OK;
    const size_t smem = (size_t)channels * kh * (width + 2 * pad) * sizeof(float);
    if (smem > 64u * 1024u || batch * (int64_t)OH > 0x7FFFFFFF) return TRON_ERR_UNSUPPORTED;
    const int d = channels * kh * kw;
    int threads = (d + 63) / 64 * 64;
    if (threads > 1024) threads = 1024;
    hipLaunchKernelGGL(k_extract_patches, dim3((unsigned)(batch * OH)), dim3(threads), smem,
                       reinterpret_cast<hipStream_t>(stream), x, channels, height, width, kh, kw, pad, stride, OH, OW, out);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

// ---- the input factor itself: gram = P^T P (kfac.py:41-58 `compute_cov_a`: a.t() @ (a / batch_size)) -----------------------
// The factor of a convolution is the Gram matrix of its patch matrix P [batch * OH * OW][d = C kh kw] — at 16 384 envs x 32x32
// with five rollout steps that is 95 M rows per layer and update, and as `extract_patches` + an f32 library GEMM it was 60 % of
// an ACKTR iteration (profiles/r03_acktr_config5_kernel_rows.txt: 112 TFLOP/s).  Here the product runs on the f16 matrix cores
// with both operands split in two halves (v / 64 = hi + lo 2^-11: three MFMAs per 32-deep slab, f32 accumulation; the error
// analysis is csrc/tron_conv_f16.hip's) and only the tiles on and above the diagonal are computed:
//   k_patches_t      x -> P^T as split f16 [d padded to 64][rows padded to 64] (K = rows contiguous: what both MFMA operands
//                    want), a thread = 8 consecutive output positions of one patch column, 16-byte stores
//   k_transpose_split  the same for a Linear layer's input a [rows][d]: a^T split, through an LDS tile
//   k_gram_f16x3     partial[s] += X X^T over K range s of the chunk: 128 x 64 tiles, 8 waves, K chunks of 64 through LDS
//                    (csrc/tron_head.hip's GEMM loop); blockIdx = (tile, K split) so that small d still fills the chip
//   k_gram_finish    gram[i][j] = scale * sum_s partial[s][min(i,j)][max(i,j)]          (fixed order: deterministic)
#ifndef TRON_GRAM_V2         // 0: the register-staged 128 x 64 kernel (A/B measurements)
#define TRON_GRAM_V2 1
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr float ACT_SCALE = 1.0f / 64.0f, LO_SCALE = 2048.0f, LO_UNSCALE = 1.0f / 2048.0f;
constexpr float GRAM_UNSCALE = 4096.0f;            // both operands carry 2^-6

__device__ __forceinline__ void split(float v, f16 &hi, f16 &lo)
{
    hi = (f16)v;
    lo = (f16)((v - (float)hi) * LO_SCALE);
}

__global__ __launch_bounds__(256) void k_patches_t(const float *__restrict__ x, int n_img, int C, int H, int W, int kh, int kw,
                                                   int pad, int stride, int OH, int OW, int d, int dpad, int64_t rows,
                                                   int64_t rows_pad, const float *__restrict__ in_scale, f16 *__restrict__ xh,
                                                   f16 *__restrict__ xl)
{
    // blockIdx.y = patch column j, blockIdx.x strides over groups of 8 rows: 32-bit index arithmetic only (rows_pad < 2^31)
    const float sc = ACT_SCALE * (in_scale ? *in_scale : 1.0f);
    const int groups = (int)(rows_pad / 8);
    const int j = blockIdx.y;
    const int c = j / (kh * kw), t = j - c * (kh * kw), ky = t / kw, kx = t - ky * kw;
    for (int grp = blockIdx.x * blockDim.x + threadIdx.x; grp < groups; grp += gridDim.x * blockDim.x) {
        const int r0 = grp * 8;
        f16x8 h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = l[e] = (f16)0.0f;
        if (j < d && r0 < rows) {
            const int b = r0 / (OH * OW);
            int rem = r0 - b * (OH * OW), oy = rem / OW, ox = rem - oy * OW;
            // 32-bit offsets into this pass's images (a pass is at most 512 MB of patches: < 2^31 input floats)
            int plane = (b * C + c) * H * W, iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = 0.0f;
                if (r0 + e < rows && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = x[plane + iy * W + ix];
                f16 hh, ll;
                split(v * sc, hh, ll);
                h[e] = hh;
                l[e] = ll;
                ix += stride;
                if (++ox == OW) {
                    ox = 0;
                    ix = kx - pad;
                    iy += stride;
                    if (++oy == OH) { oy = 0; iy = ky - pad; plane += C * H * W; }
                }
            }
        }
        *reinterpret_cast<f16x8 *>(xh + (size_t)j * rows_pad + r0) = h;
        *reinterpret_cast<f16x8 *>(xl + (size_t)j * rows_pad + r0) = l;
    }
}

// The same matrix written plane by plane: a workgroup = one (image, channel) plane, staged once in LDS with its zero
// halo; the kh * kw patch columns of that channel are shifted views of the staged plane, written four output positions
// per thread (8-byte stores: rows of an image start at image * OH * OW, a multiple of 4 when OH * OW is).  The image tensor is
// read once instead of kh * kw times, the index arithmetic is LDS-local, the kernel runs at the rate of its stores.  The
// last image's workgroups also zero the K padding (rows .. rows_pad) of their columns.
__global__ __launch_bounds__(256) void k_patches_t_plane(const float *__restrict__ x, int n_img, int C, int H, int W, int kh, int kw,
                                                         int pad, int stride, int OH, int OW, int64_t rows, int64_t rows_pad,
                                                         const float *__restrict__ in_scale, f16 *__restrict__ xh, f16 *__restrict__ xl)
{
    extern __shared__ float plane[];                                     // [(H + 2 pad)][(W + 2 pad)]
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const float sc = ACT_SCALE * (in_scale ? *in_scale : 1.0f);
    const int b = blockIdx.x / C, c = blockIdx.x - b * C;
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const float *src = x + ((size_t)b * C + c) * H * W;
    for (int i = threadIdx.x; i < Hp * Wp; i += 256) {
        const int y = i / Wp - pad, xx = i % Wp - pad;
        plane[i] = ((unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W) ? src[y * W + xx] * sc : 0.0f;
    }
    __syncthreads();
    const int per = OH * OW, quads = per / 4, taps = kh * kw;
    const size_t r_img = (size_t)b * per;
    for (int i = threadIdx.x; i < taps * quads; i += 256) {
        const int t = i / quads, q = i - t * quads, ky = t / kw, kx = t - ky * kw;
        f16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int pos = 4 * q + e, oy = pos / OW, ox = pos - oy * OW;
            f16 hh, ll;
            split(plane[(oy * stride + ky) * Wp + ox * stride + kx], hh, ll);
            h[e] = hh;
            l[e] = ll;
        }
        const size_t o = (size_t)(c * taps + t) * rows_pad + r_img + 4 * q;
        *reinterpret_cast<f16x4 *>(xh + o) = h;
        *reinterpret_cast<f16x4 *>(xl + o) = l;
    }
    if (b == n_img - 1) {                                                // the K padding of this channel's columns
        const int tail = (int)(rows_pad - rows);
        for (int i = threadIdx.x; i < taps * tail; i += 256) {
            const size_t o = (size_t)(c * taps + i / tail) * rows_pad + rows + i % tail;
            xh[o] = (f16)0.0f;
            xl[o] = (f16)0.0f;
        }
    }
}

// a f32 [rows][d] -> a^T split [dpad][rows_pad]; a workgroup = a 64 x 64 tile through LDS (reads along d, writes along rows)
__global__ __launch_bounds__(256) void k_transpose_split(const float *__restrict__ a, int64_t rows, int d, int dpad, int64_t rows_pad,
                                                         const float *__restrict__ in_scale, f16 *__restrict__ xh, f16 *__restrict__ xl)
{
    __shared__ float tile[64][65];
    const float sc = ACT_SCALE * (in_scale ? *in_scale : 1.0f);
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int j0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int rr = i >> 6, jj = i & 63;
        tile[rr][jj] = (r0 + rr < rows && j0 + jj < d) ? a[(size_t)(r0 + rr) * d + j0 + jj] : 0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 8; i += 256) {                   // (column jj, 8 consecutive rows)
        const int jj = i >> 3, g = i & 7;
        f16x8 h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            f16 hh, ll;
            split(tile[g * 8 + e][jj] * sc, hh, ll);
            h[e] = hh;
            l[e] = ll;
        }
        *reinterpret_cast<f16x8 *>(xh + (size_t)(j0 + jj) * rows_pad + r0 + g * 8) = h;
        *reinterpret_cast<f16x8 *>(xl + (size_t)(j0 + jj) * rows_pad + r0 + g * 8) = l;
    }
}

constexpr int GM = 128, GN = 64, GK = 64;
constexpr int GPITCH = GK * 2 + 32;                // LDS row pitch in bytes: conflict-free ds_read_b128 (tron_head.hip)
constexpr int G_THREADS = 512;
constexpr int A_HALF = GM * GPITCH, W_HALF = GN * GPITCH;
constexpr int G_LDS = 2 * A_HALF + 2 * W_HALF;     // 61 440 bytes: two workgroups per CU

__global__ __launch_bounds__(G_THREADS, 4) void k_gram_f16x3(const f16 *__restrict__ Xh, const f16 *__restrict__ Xl, int d, int dpad,
                                                             int64_t pitch, int nk, int ksplit, float *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *a_h = lds, *a_l = lds + A_HALF, *w_h = lds + 2 * A_HALF, *w_l = w_h + W_HALF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 3, wn = wave >> 2, li = lane & 15, g = lane >> 4;
    const int nblocks = dpad / GN, ntiles = ((d + GM - 1) / GM) * nblocks;
    const int tile = blockIdx.x % ntiles, s = blockIdx.x / ntiles;
    const int m0 = (tile / nblocks) * GM, n0 = (tile % nblocks) * GN;
    if (n0 + GN <= m0) return;                                          // strictly below the diagonal: k_gram_finish mirrors
    const int k_lo = (int)((int64_t)nk * s / ksplit), k_hi = (int)((int64_t)nk * (s + 1) / ksplit);
    if (k_lo >= k_hi) return;

    f32x4 ra[4], rw[2];
    auto load_chunk = [&](int kc) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = tid + j * G_THREADS, half = q >> 10, r = q & 1023, row = r >> 3, pc = r & 7;
            int m = m0 + row;
            m = m < dpad ? m : dpad - 1;
            ra[j] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const unsigned char *>(half ? Xl : Xh) +
                                                     ((size_t)m * pitch + (size_t)kc * GK) * 2 + pc * 16);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = tid + j * G_THREADS, half = q >> 9, r = q & 511, row = r >> 3, pc = r & 7;
            rw[j] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const unsigned char *>(half ? Xl : Xh) +
                                                     ((size_t)(n0 + row) * pitch + (size_t)kc * GK) * 2 + pc * 16);
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
    load_chunk(k_lo);
    store_chunk();
    __syncthreads();
    for (int kc = k_lo; kc < k_hi; ++kc) {
        if (kc + 1 < k_hi) load_chunk(kc + 1);                          // in flight under the MFMAs
#pragma unroll
        for (int sl = 0; sl < GK / 32; ++sl) {
            f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[t] = *reinterpret_cast<const f16x8 *>(a_h + a_off + t * 16 * GPITCH + sl * 64);
                al[t] = *reinterpret_cast<const f16x8 *>(a_l + a_off + t * 16 * GPITCH + sl * 64);
                bh[t] = *reinterpret_cast<const f16x8 *>(w_h + b_off + t * 16 * GPITCH + sl * 64);
                bl[t] = *reinterpret_cast<const f16x8 *>(w_l + b_off + t * 16 * GPITCH + sl * 64);
            }
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
        __syncthreads();
        if (kc + 1 < k_hi) {
            store_chunk();
            __syncthreads();
        }
    }
    // D row = 4 * (lane >> 4) + r, column = lane & 15; this split's partial sums are added to what earlier chunks left
    float *out = partial + (size_t)s * d * dpad;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int col = n0 + wn * 32 + n * 16 + li;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 v = (acc0[t][n] + acc1[t][n] * LO_UNSCALE) * GRAM_UNSCALE;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 32 + t * 16 + 4 * g + r;
                if (m < d) out[(size_t)m * dpad + col] += v[r];
            }
        }
    }
}

// The same product on 128 x 128 tiles fed by LDS-DMA (TRON_GRAM_V2, the default).  k_gram_f16x3 above stages its operands
// through registers and ds_write_b128 — 48 stores per K chunk and workgroup at 13 LDS cycles each keep the CU's one store
// path busy 80 % of the time — and its 128 x 64 tiles pull 49 KB from L2 per 1 M MACs.  Here a 16-wave workgroup (one per CU,
// 4 waves per SIMD, each wave a 32 x 32 sub-tile) owns a 128 x 128 tile; a K chunk of 32 — rows m0 .. m0+127 and
// n0 .. n0+127 of X, hi and lo: 32 KB — is copied global -> LDS by 32 `global_load_lds_dwordx4` (2 per wave) into a ring of
// FOUR stages: three chunks in flight while one is multiplied, one barrier per chunk.  LDS rows are 64 bytes without padding;
// a row's four 16-byte pieces are stored XOR-ed with (-(row >> 2)) & 3 — the per-lane SOURCE address does the swizzle, the
// destination of an LDS-DMA is linear — which puts the 16 lanes of every ds_read_b128 group on 16 different bank quads.
constexpr int G2_T = 128, G2_K = 32, G2_THREADS = 1024, G2_STAGES = 4;
constexpr int G2_STAGE = 2 * 2 * G2_T * G2_K * 2, G2_LDS = G2_STAGES * G2_STAGE;          // 32 KB per stage, 128 KB

__global__ __launch_bounds__(G2_THREADS, 4) void k_gram2_f16x3(const f16 *__restrict__ Xh, const f16 *__restrict__ Xl, int d, int dpad,
                                                               int64_t pitch, int nk, int ksplit, float *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 3, wn = wave >> 2, li = lane & 15, g = lane >> 4;
    const int nb = (dpad + G2_T - 1) / G2_T, ntiles = nb * nb;
    const int tile = blockIdx.x % ntiles, s = blockIdx.x / ntiles;
    const int bm = tile / nb, bn = tile % nb, m0 = bm * G2_T, n0 = bn * G2_T;
    if (bn < bm) return;                                                // below the diagonal: k_gram_finish mirrors
    const int k_lo = (int)((int64_t)nk * s / ksplit), k_hi = (int)((int64_t)nk * (s + 1) / ksplit);
    if (k_lo >= k_hi) return;

    // A stage = a K chunk of 32: [hi | lo][256 rows: A then W][64 B], 32 blocks of 1 KB (16 rows each); this wave copies blocks
    // wave and wave + 16.  A row's four 16-byte pieces are stored at piece ^ sw(row), sw(row) = (-(row >> 2)) & 3: the four
    // rows r, r+4, r+8, r+12 that share a ds_read_b128 lane group's bank span then sit in four different slots.
    const unsigned char *src[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int b = wave + 16 * j, half = b >> 4, row = 16 * (b & 15) + (lane >> 2), piece = (lane & 3) ^ ((0 - (row >> 2)) & 3);
        int grow = row < G2_T ? m0 + row : n0 + row - G2_T;
        grow = grow < dpad ? grow : dpad - 1;
        src[j] = reinterpret_cast<const unsigned char *>(half ? Xl : Xh) + (size_t)grow * pitch * 2 + piece * 16;
    }
    auto dma_chunk = [&](int kc, int stage) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[j] + (size_t)kc * (G2_K * 2)),
                                             (__attribute__((address_space(3))) void *)(lds + stage * G2_STAGE + (wave + 16 * j) * 1024), 16, 0, 0);
    };

    f32x4 acc0[2][2], acc1[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            acc0[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc1[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    int a_off[2], b_off[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ra = wm * 32 + t * 16 + li, rb = G2_T + wn * 32 + t * 16 + li;
        a_off[t] = ra * 64 + ((g ^ ((0 - (ra >> 2)) & 3)) * 16);
        b_off[t] = rb * 64 + ((g ^ ((0 - (rb >> 2)) & 3)) * 16);
    }
    // Three chunks are in flight while one is multiplied (a chunk's 32 KB take longer to arrive than its 768 MFMA cycles per
    // SIMD last): chunk c lives in stage (c - k_lo) % 4; every iteration issues exactly one chunk (past the end: the last one
    // again, unused), so "this wave's copies of the chunk about to be used have landed" is always vmcnt(4).
    const int last = k_hi - 1;
    dma_chunk(k_lo, 0);
    dma_chunk(k_lo + 1 < k_hi ? k_lo + 1 : last, 1);
    dma_chunk(k_lo + 2 < k_hi ? k_lo + 2 : last, 2);
    for (int kc = k_lo; kc < k_hi; ++kc) {
        const int st = (kc - k_lo) & 3;
        // (a bare s_barrier: __syncthreads() carries a fence, and with LDS-DMA writes in flight hipcc turns that into vmcnt(0) —
        //  draining the three chunks this ring exists to keep in flight; the counted wait above is all the ordering needed:
        //  LDS is only written by these copies and only read below)
        asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");     // everybody's copies of chunk kc are in; stage (st + 3) % 4 is free
        dma_chunk(kc + 3 < k_hi ? kc + 3 : last, (st + 3) & 3);
        const unsigned char *base = lds + st * G2_STAGE;
        f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            ah[t] = *reinterpret_cast<const f16x8 *>(base + a_off[t]);
            al[t] = *reinterpret_cast<const f16x8 *>(base + G2_STAGE / 2 + a_off[t]);
            bh[t] = *reinterpret_cast<const f16x8 *>(base + b_off[t]);
            bl[t] = *reinterpret_cast<const f16x8 *>(base + G2_STAGE / 2 + b_off[t]);
        }
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // (the surplus copies: nothing may still be writing LDS at exit)
    float *out = partial + (size_t)s * d * dpad;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int col = n0 + wn * 32 + n * 16 + li;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 v = (acc0[t][n] + acc1[t][n] * LO_UNSCALE) * GRAM_UNSCALE;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 32 + t * 16 + 4 * g + r;
                if (m < d && col < dpad) out[(size_t)m * dpad + col] += v[r];
            }
        }
    }
}

// The gradient factor of a convolution straight from the NCHW tensor g [B][C][per]: G = sum over images and positions of
// g[b, :, pos] g[b, :, pos]^T (C = 32 or 64 output channels).  The tensor is already "channel rows, positions contiguous" image by
// image, so instead of writing its split transpose (k_patches_t_plane as a 1x1 patch matrix: 4 bytes written and read again
// per element) a workgroup walks its share of the images slab by slab (32 positions): 256 threads load the C x 32 f32 block,
// split it in registers into a double-buffered LDS image [C rows][64 B, pieces XOR-swizzled as in k_gram2], and four waves
// multiply it with itself (wave = a 32 x 32 block of G, or 16 x 16 at C = 32).  Memory-bound by construction: g is read once.
template <int C>
__global__ __launch_bounds__(256) void k_gram_nchw(const float *__restrict__ g, int64_t B, int per, const float *__restrict__ in_scale,
                                                   int nsplit, float *__restrict__ partial)
{
    constexpr int TW = C / 32;                                           // MFMA tiles per wave and side: 2 (C = 64) or 1
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][C * 64];   // [buffer][hi | lo][row][64 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, gq = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;
    const float sc = ACT_SCALE * (in_scale ? *in_scale : 1.0f);
    const int s = blockIdx.x;
    const int64_t b_lo = B * s / nsplit, b_hi = B * (s + 1) / nsplit;
    const int slabs = (per + 31) / 32;
    // staging: thread -> (row c, quad q of the slab's 8 float4): C * 8 float4 per slab, C / 32 per thread
    constexpr int NLD = C * 8 / 256;
    f32x4 v[NLD];
    auto load = [&](int64_t b, int sl) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + j * 256, c = i >> 3, q = i & 7, pos = sl * 32 + q * 4;
            v[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (pos < per) v[j] = *reinterpret_cast<const f32x4 *>(g + ((size_t)b * C + c) * per + pos);   // per % 4 == 0
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + j * 256, c = i >> 3, q = i & 7;          // q-th float4 = k 4q .. 4q+3: half of piece q / 2
            typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
            f16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f16 hh, ll;
                split(v[j][e] * sc, hh, ll);
                h[e] = hh;
                l[e] = ll;
            }
            const int off = c * 64 + (((q >> 1) ^ ((0 - (c >> 2)) & 3)) * 16) + (q & 1) * 8;
            *reinterpret_cast<f16x4 *>(&lds[buf][0][off]) = h;
            *reinterpret_cast<f16x4 *>(&lds[buf][1][off]) = l;
        }
    };
    f32x4 acc0[TW][TW], acc1[TW][TW];
#pragma unroll
    for (int a = 0; a < TW; ++a)
#pragma unroll
        for (int b = 0; b < TW; ++b) {
            acc0[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc1[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    int a_off[TW], b_off[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        const int ra = (wm * TW + t) * 16 + li, rb = (wn * TW + t) * 16 + li;
        a_off[t] = ra * 64 + ((gq ^ ((0 - (ra >> 2)) & 3)) * 16);
        b_off[t] = rb * 64 + ((gq ^ ((0 - (rb >> 2)) & 3)) * 16);
    }
    const int64_t total = (b_hi - b_lo) * slabs;
    if (total > 0) {
        load(b_lo, 0);
        stage(0);
        __syncthreads();
        for (int64_t it = 0; it < total; ++it) {
            const int buf = (int)(it & 1);
            const int64_t nx = it + 1;
            if (nx < total) load(b_lo + nx / slabs, (int)(nx % slabs));  // in flight under the MFMAs
            f16x8 ah[TW], al[TW], bh[TW], bl[TW];
#pragma unroll
            for (int t = 0; t < TW; ++t) {
                ah[t] = *reinterpret_cast<const f16x8 *>(&lds[buf][0][a_off[t]]);
                al[t] = *reinterpret_cast<const f16x8 *>(&lds[buf][1][a_off[t]]);
                bh[t] = *reinterpret_cast<const f16x8 *>(&lds[buf][0][b_off[t]]);
                bl[t] = *reinterpret_cast<const f16x8 *>(&lds[buf][1][b_off[t]]);
            }
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int n = 0; n < TW; ++n) {
                    acc1[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bl[n], acc1[t][n], 0, 0, 0);
                    acc0[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bh[n], acc0[t][n], 0, 0, 0);
                    acc1[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh[n], acc1[t][n], 0, 0, 0);
                }
            if (nx < total) stage(buf ^ 1);                               // (the other buffer: nobody reads it during this slab)
            __syncthreads();
        }
    }
    float *out = partial + (size_t)s * C * C;
#pragma unroll
    for (int n = 0; n < TW; ++n) {
        const int col = (wn * TW + n) * 16 + li;
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            const f32x4 r4 = (acc0[t][n] + acc1[t][n] * LO_UNSCALE) * GRAM_UNSCALE;
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)((wm * TW + t) * 16 + 4 * gq + r) * C + col] = r4[r];
        }
    }
}

__global__ void k_gram_finish(const float *__restrict__ partial, int ksplit, int d, int dpad, float scale,
                              const float *__restrict__ in_scale, float *__restrict__ gram)
{
    if (in_scale) scale /= (*in_scale) * (*in_scale);                   // (a power of two: exact)
    const int64_t total = (int64_t)d * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / d), c = (int)(i - (int64_t)r * d);
        const int lo = r < c ? r : c, hi = r < c ? c : r;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};                              // four independent chains, joined in a fixed order
        int s = 0;
        for (; s + 3 < ksplit; s += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += partial[((size_t)(s + k) * d + lo) * dpad + hi];
        }
        for (int k = 0; s < ksplit; ++s, ++k) acc[k] += partial[((size_t)s * d + lo) * dpad + hi];
        gram[i] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) * scale;
    }
}

constexpr int64_t GRAM_CHUNK_BYTES = 512ll << 20;   // the split transposed operand of one pass

struct GramPlan {
    int dpad, ksplit;
    int64_t rows_chunk, x_bytes, partial_bytes, total;   // rows per pass (multiple of 64), bytes of one half of X
};
inline GramPlan gram_plan(int64_t rows_total, int64_t rows_unit, int d)
{
    GramPlan p;
    p.dpad = (d + 63) / 64 * 64;
    int64_t rows = GRAM_CHUNK_BYTES / (4 * (int64_t)p.dpad);            // hi + lo, 2 bytes each
    rows = rows / rows_unit * rows_unit;                                // whole images per pass
    if (rows < rows_unit) rows = rows_unit;
    if (rows > rows_total) rows = (rows_total + rows_unit - 1) / rows_unit * rows_unit;
    p.rows_chunk = (rows + 63) / 64 * 64;
#if TRON_GRAM_V2
    const int nb2 = (p.dpad + G2_T - 1) / G2_T, upper = nb2 * (nb2 + 1) / 2;
    int ks = 768 / upper;                                               // one 16-wave workgroup per CU: three rounds of the chip
#else
    const int nblocks = p.dpad / GN, mt = (d + GM - 1) / GM;
    int upper = 0;
    for (int bm = 0; bm < mt; ++bm)
        for (int bn = 0; bn < nblocks; ++bn) upper += (bn * GN + GN > bm * GM);
    int ks = 2048 / (upper > 0 ? upper : 1);                            // ~8 workgroups per CU in flight over the launch
#endif
    const int64_t nk = p.rows_chunk / GK;
    if (ks > nk / 16) ks = (int)(nk / 16);                              // a split walks at least 16 K chunks
    if (ks < 1) ks = 1;
    if (ks > 256) ks = 256;
    p.ksplit = ks;
    p.x_bytes = (int64_t)p.dpad * p.rows_chunk * 2;
    p.partial_bytes = (int64_t)ks * d * p.dpad * 4;
    p.total = 2 * p.x_bytes + p.partial_bytes + 512;
    return p;
}

int gram_pass(const f16 *xh, const f16 *xl, const GramPlan &p, int d, int64_t rows_pad, float *partial, hipStream_t st)
{
    static uint64_t prepared = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_NO_DEVICE; }
    if (!(prepared & (1ull << (dev & 63)))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_gram_f16x3), hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(k_gram2_f16x3), hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS) != hipSuccess)
            (void)hipGetLastError();
        prepared |= 1ull << (dev & 63);
    }
#if TRON_GRAM_V2
    const int nb2 = (p.dpad + G2_T - 1) / G2_T;
    hipLaunchKernelGGL(k_gram2_f16x3, dim3((unsigned)(nb2 * nb2 * p.ksplit)), dim3(G2_THREADS), G2_LDS, st, xh, xl, d, p.dpad, rows_pad,
                       (int)(rows_pad / G2_K), p.ksplit, partial);
#else
    const int ntiles = ((d + GM - 1) / GM) * (p.dpad / GN);
    hipLaunchKernelGGL(k_gram_f16x3, dim3((unsigned)(ntiles * p.ksplit)), dim3(G_THREADS), G_LDS, st, xh, xl, d, p.dpad, rows_pad,
                       (int)(rows_pad / GK), p.ksplit, partial);
#endif
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

// k_gram_nchw (a 1x1 "patch matrix" read straight from the NCHW tensor) writes one C x C block of partial sums per split.
// The split count and the bytes it needs are decided HERE for both the size query and the call: a Gram plan over batch * per
// rows is smaller than the blocks when an image has few positions (conv7's 4x4 / 6x6 outputs), so the query takes the maximum.
inline bool nchw_gram_geometry(int32_t channels, int32_t kh, int32_t kw, int32_t pad, int32_t stride, int64_t per)
{
    return kh == 1 && kw == 1 && pad == 0 && stride == 1 && (channels == 32 || channels == 64) && per % 4 == 0;
}
inline bool kfac_px_enabled()                                            // TRON_KFAC_PX=0: the patch-matrix path (A/B measurements)
{
    static const bool on = [] { const char *e = getenv("TRON_KFAC_PX"); return !(e && e[0] == '0'); }();
    return on;
}
inline int nchw_gram_splits(int64_t batch) { return (int)(batch < 2048 ? batch : 2048); }
inline int64_t nchw_gram_bytes(int64_t batch, int32_t channels)
{
    return (int64_t)nchw_gram_splits(batch) * channels * channels * (int64_t)sizeof(float) + 512;
}

}  // namespace

extern "C" int64_t tron_kfac_patch_gram_workspace(int64_t batch, int32_t channels, int32_t height, int32_t width, int32_t kh,
                                                  int32_t kw, int32_t pad, int32_t stride)
{
    if (batch < 1 || channels < 1 || height < 1 || width < 1 || kh < 1 || kw < 1 || pad < 0 || stride < 1) return 0;
    const int OH = (height + 2 * pad - kh) / stride + 1, OW = (width + 2 * pad - kw) / stride + 1;
    if (OH < 1 || OW < 1 || (int64_t)channels * kh * kw > 8192) return 0;
    const int64_t per = (int64_t)OH * OW;
    int64_t need = gram_plan(batch * per, per, channels * kh * kw).total;
    if (kfac_px_enabled() && tron_kfac_px_supported(batch, channels, height, width, kh, kw, pad, stride)) {
        const int64_t px = tron_kfac_px_workspace(batch, channels, height);
        if (px > need) need = px;
    }
    if (nchw_gram_geometry(channels, kh, kw, pad, stride, per)) {
        const int64_t direct = nchw_gram_bytes(batch, channels);
        if (direct > need) need = direct;
    }
    return need;
}

extern "C" int tron_kfac_patch_gram(const float *x, int64_t batch, int32_t channels, int32_t height, int32_t width, int32_t kh,
                                    int32_t kw, int32_t pad, int32_t stride, float scale, const float *in_scale, float *gram,
                                    void *workspace, void *stream)
{
    if (!x || !gram || !workspace || batch < 0 || channels < 1 || height < 1 || width < 1 || kh < 1 || kw < 1 || pad < 0 || stride < 1)
        return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(gram)) & 15u) return TRON_ERR_BAD_ARG;
    const int OH = (height + 2 * pad - kh) / stride + 1, OW = (width + 2 * pad - kw) / stride + 1;
    if (OH < 1 || OW < 1) return TRON_ERR_BAD_ARG;
    const int d = channels * kh * kw;
    if (d > 8192) return TRON_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (batch == 0) return hipMemsetAsync(gram, 0, (size_t)d * d * sizeof(float), st) == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
    const int64_t per = (int64_t)OH * OW;
    // a 3x3 / pad 1 / stride 1 layer's input factor at the trunk's shapes: from the image's PX16 window, no patch matrix (csrc/tron_kfac_px.hip)
    if (!in_scale && kfac_px_enabled() && tron_kfac_px_supported(batch, channels, height, width, kh, kw, pad, stride) &&
        tron_kfac_px_workspace(batch, channels, height) > 0)
        return tron_kfac_px_gram(x, batch, channels, height, scale, gram, workspace, st);
    if (nchw_gram_geometry(channels, kh, kw, pad, stride, per) && (reinterpret_cast<uintptr_t>(x) & 15u) == 0) {
        // a gradient factor (or any 1x1 "patch matrix"): straight from the NCHW tensor, k_gram_nchw
        const int nsplit = nchw_gram_splits(batch);
        float *partial = reinterpret_cast<float *>(workspace);           // nsplit * C * C floats = nchw_gram_bytes(): what _workspace() reserved
        if (channels == 64)
            hipLaunchKernelGGL(k_gram_nchw<64>, dim3(nsplit), dim3(256), 0, st, x, batch, (int)per, in_scale, nsplit, partial);
        else
            hipLaunchKernelGGL(k_gram_nchw<32>, dim3(nsplit), dim3(256), 0, st, x, batch, (int)per, in_scale, nsplit, partial);
        hipLaunchKernelGGL(k_gram_finish, dim3(16), dim3(256), 0, st, partial, nsplit, d, d, scale, in_scale, gram);
        return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
    }
    const GramPlan p = gram_plan(batch * per, per, d);
    unsigned char *wsb = reinterpret_cast<unsigned char *>(workspace);
    f16 *xh = reinterpret_cast<f16 *>(wsb), *xl = reinterpret_cast<f16 *>(wsb + p.x_bytes);
    float *partial = reinterpret_cast<float *>(wsb + 2 * p.x_bytes);
    if (hipMemsetAsync(partial, 0, (size_t)p.partial_bytes, st) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_LAUNCH; }
    const int64_t imgs_chunk = p.rows_chunk / per > 0 ? p.rows_chunk / per : 1;
    if (imgs_chunk * channels * height * width >= (1ll << 31)) return TRON_ERR_UNSUPPORTED;   // (k_patches_t indexes a pass's input with 32 bits)
    for (int64_t i = 0; i < batch; i += imgs_chunk) {
        const int64_t n = batch - i < imgs_chunk ? batch - i : imgs_chunk;
        const int64_t rows = n * per, rows_pad = (rows + 63) / 64 * 64;
        const size_t plane_bytes = (size_t)(height + 2 * pad) * (width + 2 * pad) * sizeof(float);
        if (per % 4 == 0 && plane_bytes <= 48 * 1024 && n * channels < (1ll << 31)) {
            // (columns d .. dpad of the operand stay unwritten: k_gram2 clamps its row index to dpad - 1 and never stores what those
            //  rows produce where k_gram_finish reads)
            hipLaunchKernelGGL(k_patches_t_plane, dim3((unsigned)(n * channels)), dim3(256), plane_bytes, st,
                               x + (size_t)i * channels * height * width, (int)n, channels, height, width, kh, kw, pad, stride, OH, OW, rows,
                               rows_pad, in_scale, xh, xl);
            const int rc = gram_pass(xh, xl, p, d, rows_pad, partial, st);
            if (rc != TRON_OK) return rc;
            continue;
        }
        int64_t gblocks = (rows_pad / 8 + 255) / 256;                    // (x: groups of 8 rows, grid-stride; y: patch column)
        const int64_t want = 8192 / p.dpad > 0 ? 8192 / p.dpad : 1;       // ~32 workgroups per CU over the launch, tens of items per thread
        if (gblocks > want) gblocks = want;
        hipLaunchKernelGGL(k_patches_t, dim3((unsigned)gblocks, (unsigned)p.dpad), dim3(256), 0, st, x + (size_t)i * channels * height * width, (int)n, channels,
                           height, width, kh, kw, pad, stride, OH, OW, d, p.dpad, rows, rows_pad, in_scale, xh, xl);
        const int rc = gram_pass(xh, xl, p, d, rows_pad, partial, st);
        if (rc != TRON_OK) return rc;
    }
    hipLaunchKernelGGL(k_gram_finish, dim3((unsigned)(((int64_t)d * d + 255) / 256 < 4096 ? ((int64_t)d * d + 255) / 256 : 4096)), dim3(256),
                       0, st, partial, p.ksplit, d, p.dpad, scale, in_scale, gram);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

extern "C" int64_t tron_kfac_gram_workspace(int64_t rows, int32_t d)
{
    if (rows < 1 || d < 1 || d > 8192) return 0;
    return gram_plan(rows, 64, d).total;
}

extern "C" int tron_kfac_gram(const float *a, int64_t rows, int32_t d, float scale, const float *in_scale, float *gram,
                              void *workspace, void *stream)
{
    if (!a || !gram || !workspace || rows < 0 || d < 1) return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(gram)) & 15u) return TRON_ERR_BAD_ARG;
    if (d > 8192) return TRON_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (rows == 0) return hipMemsetAsync(gram, 0, (size_t)d * d * sizeof(float), st) == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
    const GramPlan p = gram_plan(rows, 64, d);
    unsigned char *wsb = reinterpret_cast<unsigned char *>(workspace);
    f16 *xh = reinterpret_cast<f16 *>(wsb), *xl = reinterpret_cast<f16 *>(wsb + p.x_bytes);
    float *partial = reinterpret_cast<float *>(wsb + 2 * p.x_bytes);
    if (hipMemsetAsync(partial, 0, (size_t)p.partial_bytes, st) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_LAUNCH; }
    for (int64_t i = 0; i < rows; i += p.rows_chunk) {
        const int64_t n = rows - i < p.rows_chunk ? rows - i : p.rows_chunk, rows_pad = (n + 63) / 64 * 64;
        hipLaunchKernelGGL(k_transpose_split, dim3((unsigned)(rows_pad / 64), (unsigned)(p.dpad / 64)), dim3(256), 0, st,
                           a + (size_t)i * d, n, d, p.dpad, rows_pad, in_scale, xh, xl);
        const int rc = gram_pass(xh, xl, p, d, rows_pad, partial, st);
        if (rc != TRON_OK) return rc;
    }
    hipLaunchKernelGGL(k_gram_finish, dim3((unsigned)(((int64_t)d * d + 255) / 256 < 4096 ? ((int64_t)d * d + 255) / 256 : 4096)), dim3(256),
                       0, st, partial, p.ksplit, d, p.dpad, scale, in_scale, gram);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}
