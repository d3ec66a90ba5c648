// tron_conv_wgrad.hip — the weight gradient of the CNN's 3x3 convolutions (what loss.backward() computes for
// conv1..conv6 of Net/DQNNet.py:10-17 in DDQN.py:148):
//     dW[co][ci][ky][kx] = sum over b, y, x of  g[b][co][y][x] * in[b][ci][y + ky - 1][x + kx - 1]
// on the f16 matrix cores with both operands split in two halves (v = hi + lo 2^-11, three MFMAs per k-slab, f32
// accumulation — csrc/tron_conv_f16.hip has the error bound), NCHW f32 in, f32 out, no layout transposes.
//
// The GEMM: M = cout, N = cin per tap, K = pixels of all images.  Pixels are the contiguous axis of NCHW, so both
// operands stage into LDS as [channel][k] rows and every MFMA operand is one aligned 16-byte read.  The 3x3 taps
// become SHIFTS of the input rows against the gradient rows: each image is laid out on a 14-wide, 192-long haloed
// strip (position q = 14 y + x; columns 12-13, row 12 and the tail stay zero), so the input pixel under tap (ky, kx)
// of gradient position q is input position q + 14 (ky - 1) + (kx - 1), and wherever that leaves the image the gradient
// strip holds a zero.  A wave reads a 40-element window of its input rows once per k-slab and cuts the nine shifted
// operands out of it in registers (v_alignbit for the odd shifts) — 27 MFMAs per 16x16 output tile per slab against
// 7 LDS reads.  192 / 144 = 1.33x padded work buys alignment and nine taps per read.
//
// One workgroup (4 waves, one per SIMD with the whole register file: the accumulators of a 32 x 16 x 9 output block
// are 144 registers per lane; 64 output channels are two workgroups on the two halves, because 288 accumulators do
// not fit the 256 AGPRs) walks its share of the images, the next image's rows in flight in registers while the
// current one is multiplied; the per-workgroup sums go to a workspace and a second kernel adds them up in a fixed
// order (deterministic, unlike atomics).  Gradients are tiny numbers: they are scaled by a power of two taken from their
// largest magnitude (handed in by tron_bias_mish_bwd, or found by a pre-pass) so that the f16 halves stay normal.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tron_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int SIDE = 12, HW = SIDE * SIDE, ROWP = 14, IMG_K = 192, SLABS_IMG = IMG_K / 32, MARGIN = 16;
constexpr int THREADS = 256, GRID_MAX = 256, STAGE2 = 8, ABSMAX_BLOCKS = 1024;
constexpr float IN_SCALE = 1.0f / 64.0f, LO_SCALE = 2048.0f;

constexpr int odd16(int bytes) { return (((bytes + 15) / 16) | 1) * 16; }    // row pitch: an odd number of 16-byte units

constexpr int COT = 2;                                         // 16-channel output tiles per wave: 32 channels per workgroup

template <int CIT, int IMGS>
struct Cfg {
    static constexpr int KSPLIT = 4 / CIT;                     // waves sharing an input tile split the slabs
    static constexpr int SL = SLABS_IMG * IMGS / KSPLIT;       // slabs per wave per round
    static constexpr int GP_PITCH = odd16(IMGS * IMG_K * 2), IN_PITCH = odd16((IMGS * IMG_K + 2 * MARGIN) * 2);
    static constexpr int GP_HALF = COT * 16 * GP_PITCH, IN_HALF = CIT * 16 * IN_PITCH;
    static constexpr int LDS = 2 * GP_HALF + 2 * IN_HALF;
    static constexpr int GP_ITEMS = IMGS * COT * 16 * SIDE, IN_ITEMS = IMGS * CIT * 16 * SIDE;
    static constexpr int NIT = (GP_ITEMS + IN_ITEMS + THREADS - 1) / THREADS;
    static_assert(SLABS_IMG * IMGS % KSPLIT == 0, "slabs must split evenly");
};

__device__ __forceinline__ uint32_t pack2(f16 a, f16 b)
{
    const f16x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, v);
}

// the operand that starts IDX elements into the 40-element window D (20 dwords)
template <int IDX>
__device__ __forceinline__ f16x8 window_frag(const uint32_t (&D)[20])
{
    u32x4 v;
    if constexpr (IDX % 2 == 0) {
        v = (u32x4){D[IDX / 2], D[IDX / 2 + 1], D[IDX / 2 + 2], D[IDX / 2 + 3]};
    } else {
        constexpr int k = IDX / 2;
        v = (u32x4){__builtin_amdgcn_alignbit(D[k + 1], D[k], 16), __builtin_amdgcn_alignbit(D[k + 2], D[k + 1], 16),
                    __builtin_amdgcn_alignbit(D[k + 3], D[k + 2], 16), __builtin_amdgcn_alignbit(D[k + 4], D[k + 3], 16)};
    }
    return __builtin_bit_cast(f16x8, v);
}

// per-block maxima of |g| -> one power-of-two scale that puts the largest magnitude in [2^13, 2^14)
__device__ __forceinline__ float grad_scale(const float *__restrict__ absmax, int n_absmax, float *red)
{
    float m = 0.0f;
    for (int i = threadIdx.x; i < n_absmax; i += THREADS) m = fmaxf(m, absmax[i]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const uint32_t bits = __float_as_uint(m);
    const int e = (int)((bits >> 23) & 255u) - 126;                     // m = f 2^e, f in [0.5, 1)
    if (!(m > 0.0f) || e < -100 || e > 100) return 1.0f;                // zero, denormal, huge or NaN gradients: unscaled
    return __uint_as_float((uint32_t)(127 + 14 - e) << 23);            // 2^(14 - e)
}

template <int CIT, int IMGS>
__global__ __launch_bounds__(THREADS, 1) void k_wgrad(const float *__restrict__ in, const float *__restrict__ gp,
                                                      const float *__restrict__ absmax, int n_absmax,
                                                      float *__restrict__ partial, int batch, int cin, int cout,
                                                      int nrounds)
{
    using C = Cfg<CIT, IMGS>;
    constexpr int COUT = COT * 16;                                       // this workgroup's output channels
    const int nhalves = cout / COUT, co0 = ((int)blockIdx.x % nhalves) * COUT, wg = (int)blockIdx.x / nhalves;
    const int wgs = (int)gridDim.x / nhalves;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ float red[4];
    unsigned char *gp_h = lds, *gp_l = lds + C::GP_HALF, *in_h = lds + 2 * C::GP_HALF, *in_l = in_h + C::IN_HALF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    const int cit = wave % CIT, ks = wave / CIT;

    for (int i = tid * 16; i < C::LDS; i += THREADS * 16) *reinterpret_cast<f32x4 *>(lds + i) = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float gscale = grad_scale(absmax, n_absmax, red);            // (contains the barrier after the clear)
    const int in_items = IMGS * cin * SIDE;

    f32x4 acc0[COT][9], acc1[COT][9];
#pragma unroll
    for (int t = 0; t < COT; ++t)
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            acc0[t][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc1[t][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

    // staging items: one 12-float row of one channel of one image (gradient rows first, then input rows)
    f32x4 pf[C::NIT][3];
#define TRON_WG_LOAD(round_)                                                                                             \
    _Pragma("unroll") for (int k_ = 0; k_ < C::NIT; ++k_) {                                                              \
        const int it_ = tid + k_ * THREADS;                                                                              \
        const float *src_ = nullptr;                                                                                     \
        if (it_ < C::GP_ITEMS) {                                                                                         \
            const int j_ = it_ / (COUT * SIDE), rem_ = it_ - j_ * (COUT * SIDE);                                         \
            const int64_t img_ = (int64_t)(round_) * IMGS + j_;                                                          \
            if (img_ < batch) src_ = gp + ((img_ * cout + co0) * SIDE + rem_) * SIDE;                                            \
        } else if (it_ - C::GP_ITEMS < in_items) {                                                                       \
            const int i2_ = it_ - C::GP_ITEMS, j_ = i2_ / (cin * SIDE), rem_ = i2_ - j_ * (cin * SIDE);                  \
            const int64_t img_ = (int64_t)(round_) * IMGS + j_;                                                          \
            if (img_ < batch) src_ = in + (img_ * cin * SIDE + rem_) * SIDE;                                             \
        }                                                                                                                \
        if (src_) {                                                                                                      \
            pf[k_][0] = reinterpret_cast<const f32x4 *>(src_)[0];                                                        \
            pf[k_][1] = reinterpret_cast<const f32x4 *>(src_)[1];                                                        \
            pf[k_][2] = reinterpret_cast<const f32x4 *>(src_)[2];                                                        \
        } else {                                                                                                         \
            pf[k_][0] = pf[k_][1] = pf[k_][2] = (f32x4){0.f, 0.f, 0.f, 0.f};                                             \
        }                                                                                                                \
    }
#define TRON_WG_STORE()                                                                                                  \
    _Pragma("unroll") for (int k_ = 0; k_ < C::NIT; ++k_) {                                                              \
        const int it_ = tid + k_ * THREADS;                                                                              \
        unsigned char *dh_ = nullptr;                                                                                    \
        int half_ = 0;                                                                                                   \
        float sc_ = gscale;                                                                                              \
        if (it_ < C::GP_ITEMS) {                                                                                         \
            const int j_ = it_ / (COUT * SIDE), rem_ = it_ - j_ * (COUT * SIDE), c_ = rem_ / SIDE, y_ = rem_ - c_ * SIDE; \
            dh_ = gp_h + c_ * C::GP_PITCH + (j_ * IMG_K + y_ * ROWP) * 2;                                                \
            half_ = C::GP_HALF;                                                                                          \
        } else if (it_ - C::GP_ITEMS < in_items) {                                                                       \
            const int i2_ = it_ - C::GP_ITEMS, j_ = i2_ / (cin * SIDE), rem_ = i2_ - j_ * (cin * SIDE);                  \
            const int c_ = rem_ / SIDE, y_ = rem_ - c_ * SIDE;                                                           \
            dh_ = in_h + c_ * C::IN_PITCH + (MARGIN + j_ * IMG_K + y_ * ROWP) * 2;                                       \
            half_ = C::IN_HALF;                                                                                          \
            sc_ = IN_SCALE;                                                                                              \
        }                                                                                                                \
        if (dh_) {                                                                                                       \
            _Pragma("unroll") for (int q_ = 0; q_ < 3; ++q_) {                                                           \
                const f32x4 v_ = pf[k_][q_] * sc_;                                                                       \
                const f16 h0_ = (f16)v_[0], h1_ = (f16)v_[1], h2_ = (f16)v_[2], h3_ = (f16)v_[3];                        \
                const f16 l0_ = (f16)((v_[0] - (float)h0_) * LO_SCALE), l1_ = (f16)((v_[1] - (float)h1_) * LO_SCALE);    \
                const f16 l2_ = (f16)((v_[2] - (float)h2_) * LO_SCALE), l3_ = (f16)((v_[3] - (float)h3_) * LO_SCALE);    \
                reinterpret_cast<uint32_t *>(dh_)[2 * q_] = pack2(h0_, h1_);                                             \
                reinterpret_cast<uint32_t *>(dh_)[2 * q_ + 1] = pack2(h2_, h3_);                                         \
                reinterpret_cast<uint32_t *>(dh_ + half_)[2 * q_] = pack2(l0_, l1_);                                     \
                reinterpret_cast<uint32_t *>(dh_ + half_)[2 * q_ + 1] = pack2(l2_, l3_);                                 \
            }                                                                                                            \
        }                                                                                                                \
    }
#define TRON_WG_TAP(tap_, idx_)                                                                                          \
    {                                                                                                                    \
        const f16x8 bh_ = window_frag<idx_>(Dh), bl_ = window_frag<idx_>(Dl);                                            \
        _Pragma("unroll") for (int t = 0; t < COT; ++t)                                                                  \
            acc0[t][tap_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bh_, acc0[t][tap_], 0, 0, 0);                  \
        _Pragma("unroll") for (int t = 0; t < COT; ++t)                                                                  \
            acc1[t][tap_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bl_, acc1[t][tap_], 0, 0, 0);                  \
        _Pragma("unroll") for (int t = 0; t < COT; ++t)                                                                  \
            acc1[t][tap_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh_, acc1[t][tap_], 0, 0, 0);                  \
    }

    int round = wg;
    TRON_WG_LOAD(round)
    for (; round < nrounds; round += wgs) {
        __syncthreads();                                                 // the previous round's operands are consumed
        TRON_WG_STORE()
        __syncthreads();
        TRON_WG_LOAD(round + wgs)                             // in flight under the MFMAs (zeros past the end)
        const unsigned char *a_h = gp_h + li * C::GP_PITCH + g * 16, *a_l = a_h + C::GP_HALF;
        const unsigned char *b_h = in_h + (cit * 16 + li) * C::IN_PITCH + g * 16, *b_l = b_h + C::IN_HALF;
#pragma unroll
        for (int sl = 0; sl < C::SL; ++sl) {
            const int s = ks * C::SL + sl;
            f16x8 ah[COT], al[COT];
#pragma unroll
            for (int t = 0; t < COT; ++t) {
                ah[t] = *reinterpret_cast<const f16x8 *>(a_h + t * 16 * C::GP_PITCH + s * 64);
                al[t] = *reinterpret_cast<const f16x8 *>(a_l + t * 16 * C::GP_PITCH + s * 64);
            }
            uint32_t Dh[20], Dl[20];                                     // elements [32 s + 8 g - 16, + 40) of the input rows
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const u32x4 vh = *reinterpret_cast<const u32x4 *>(b_h + s * 64 + k * 16);
                const u32x4 vl = *reinterpret_cast<const u32x4 *>(b_l + s * 64 + k * 16);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    Dh[4 * k + c] = vh[c];
                    Dl[4 * k + c] = vl[c];
                }
            }
            // tap (ky, kx): shift 14 (ky - 1) + (kx - 1), window index 16 + shift
            TRON_WG_TAP(0, 1) TRON_WG_TAP(1, 2) TRON_WG_TAP(2, 3)
            TRON_WG_TAP(3, 15) TRON_WG_TAP(4, 16) TRON_WG_TAP(5, 17)
            TRON_WG_TAP(6, 29) TRON_WG_TAP(7, 30) TRON_WG_TAP(8, 31)
        }
    }
#undef TRON_WG_LOAD
#undef TRON_WG_STORE
#undef TRON_WG_TAP

    // D row = 4 g + r = co within the tile, column = li = ci within the tile
    const float unscale = 64.0f / gscale;
    float *dst = partial + ((size_t)(wg * C::KSPLIT + ks) * cout + co0) * cin * 9;
    const int ci = cit * 16 + li;
    if (ci < cin) {
#pragma unroll
        for (int t = 0; t < COT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float *row = dst + ((size_t)(t * 16 + 4 * g + r) * cin + ci) * 9;
#pragma unroll
                for (int k = 0; k < 9; ++k) row[k] = (acc0[t][k][r] + acc1[t][k][r] * (1.0f / LO_SCALE)) * unscale;
            }
    }
}

// sums of the per-workgroup partial gradients in a fixed order: stage 1 adds STAGE2 interleaved subsets, stage 2 those
__global__ void k_wgrad_reduce(const float *__restrict__ partial, int nparts, int stride, int W, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W) return;
    float s = 0.0f;
    for (int p = blockIdx.y; p < nparts; p += stride) s += partial[(size_t)p * W + i];
    out[(size_t)blockIdx.y * W + i] = s;
}

__global__ __launch_bounds__(256) void k_absmax(const float *__restrict__ x, size_t n4, float *__restrict__ out)
{
    __shared__ float red[4];
    float m = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4 *>(x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

struct Plan { int64_t absmax, partial, stage2, total; };
Plan plan(int cin, int cout)
{
    const int64_t W = (int64_t)cout * cin * 9;
    Plan p{};
    p.absmax = 0;
    p.partial = ABSMAX_BLOCKS * 4;
    const int ksplit = cin <= 16 ? 4 : cin <= 32 ? 2 : 1;               // Cfg::KSPLIT partial sums per workgroup
    p.stage2 = p.partial + (int64_t)GRID_MAX * ksplit * W * 4;
    p.total = p.stage2 + (int64_t)STAGE2 * W * 4;
    return p;
}

template <int CIT, int IMGS>
int launch(const float *in, const float *gp, const float *absmax, int n_absmax, float *grad_w, int64_t batch, int cin,
           int cout, unsigned char *ws, hipStream_t st)
{
    using C = Cfg<CIT, IMGS>;
    static uint64_t prepared = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_NO_DEVICE; }
    auto kern = k_wgrad<CIT, IMGS>;
    if (!(prepared & (1ull << (dev & 63)))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess)
            (void)hipGetLastError();
        prepared |= 1ull << (dev & 63);
    }
    const int W = cout * cin * 9, nhalves = cout / (COT * 16);
    const Plan p = plan(cin, cout);
    const int nrounds = (int)((batch + IMGS - 1) / IMGS);
    const int wgs = nrounds < GRID_MAX / nhalves ? nrounds : GRID_MAX / nhalves;   // workgroups per output half
    float *partial = reinterpret_cast<float *>(ws + p.partial), *stage2 = reinterpret_cast<float *>(ws + p.stage2);
    hipLaunchKernelGGL(kern, dim3(wgs * nhalves), dim3(THREADS), C::LDS, st, in, gp, absmax, n_absmax, partial, (int)batch, cin,
                       cout, nrounds);
    const int nparts = wgs * C::KSPLIT;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((W + 255) / 256, STAGE2), dim3(256), 0, st, partial, nparts, STAGE2, W, stage2);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((W + 255) / 256, 1), dim3(256), 0, st, stage2, STAGE2, 1, W, grad_w);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

}  // namespace

extern "C" int64_t tron_conv3x3_wgrad_workspace(int32_t cin, int32_t cout)
{
    if (cin < 1 || cin > 64 || cout < 1 || cout > 64) return 0;
    return plan(cin, cout).total;
}

extern "C" int tron_conv3x3_wgrad(const float *in, const float *grad_pre, const float *grad_absmax, int32_t n_absmax,
                                  float *grad_weight, int64_t batch, int32_t cin, int32_t cout, int32_t side,
                                  void *workspace, void *stream)
{
    if (!in || !grad_pre || !grad_weight || !workspace || batch < 0 || (grad_absmax && n_absmax < 1)) return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(grad_pre) | reinterpret_cast<uintptr_t>(workspace) |
         reinterpret_cast<uintptr_t>(grad_weight)) & 15u)
        return TRON_ERR_BAD_ARG;
    const bool small = cin == 3 || cin == 4;
    if (side != SIDE || !(small || cin == 32 || cin == 64) || !(cout == 32 || cout == 64) || batch > (1ll << 24))
        return TRON_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (batch == 0) return hipMemsetAsync(grad_weight, 0, (size_t)cout * cin * 9 * 4, st) == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
    unsigned char *ws = reinterpret_cast<unsigned char *>(workspace);
    if (!grad_absmax) {
        const size_t n4 = (size_t)batch * cout * HW / 4;
        const unsigned blocks = (unsigned)((n4 + 255) / 256 < ABSMAX_BLOCKS ? (n4 + 255) / 256 : ABSMAX_BLOCKS);
        hipLaunchKernelGGL(k_absmax, dim3(blocks), dim3(256), 0, st, grad_pre, n4, reinterpret_cast<float *>(ws));
        grad_absmax = reinterpret_cast<const float *>(ws);
        n_absmax = (int32_t)blocks;
    }
    if (small) return launch<1, 2>(in, grad_pre, grad_absmax, n_absmax, grad_weight, batch, cin, cout, ws, st);
    if (cin == 32) return launch<2, 1>(in, grad_pre, grad_absmax, n_absmax, grad_weight, batch, cin, cout, ws, st);
    return launch<4, 1>(in, grad_pre, grad_absmax, n_absmax, grad_weight, batch, cin, cout, ws, st);
}
