// tron_conv_f16.hip — the same fused 3x3 convolution as tron_conv.hip (Net/DQNNet.py:10-17,33-50: conv + bias +
// residual + mish, NCHW f32 in and out, conv1 straight from the int8 observation codes), computed on the f16
// matrix cores at fp32-grade accuracy by splitting every operand in two halves.
//
// Why.  gfx950 has no reduced-precision fast path for f32 inputs: v_mfma_f32_16x16x4_f32 runs at the f32 vector
// rate, 1/16 of the f16 rate.  An f32 value v is hi + lo * 2^-11 with hi = f16(v) and lo = f16((v - hi) * 2^11)
// to within 2^-22 |v|, and a product of two f16 values is exact in f32, so
//     a * b  =  ah*bh  +  (ah*bl + al*bh) * 2^-11                      (+ al*bl * 2^-22, dropped: 2^-22 relative)
// is three v_mfma_f32_16x16x32_f16 (32 k-values in 16 cycles each) where the f32 path needs eight 16x16x4 MFMAs of
// 32 cycles: 5.3x less matrix-pipe time, f32 accumulation throughout, error per product 2^-22 against f32's 2^-24.
// Activations are scaled by 2^-6 before the split (|x| up to 4e6 stays inside f16; tiny values lose nothing
// because lo picks up what hi's subnormal rounding drops); the scale is undone, exactly, in the epilogue.
//
// Mapping (GEMM view: M = pixels, N = output channels, K = taps x input channels).
//   * A workgroup owns P whole images (4 at 12x12) and 32 or 64 output channels: 4 waves along M (one image each
//     at 12x12), 1 or 2 along N; a wave holds MT x 2 tiles of 16 px x 16 channels, two f32 accumulators each.
//   * K is walked in chunks of 16 input channels x all 9 taps (padded to 10: the tenth tap has zero weights), five
//     32-deep slabs per chunk: lane group g of an MFMA covers (tap 2s + g/2, channel octet g%2).  LDS holds the
//     chunk's input planes channel-innermost with their zero halo, [pixel][16 ci] f16 twice (hi, lo), and the
//     weights as [tap][cout][16 ci] f16 twice, so every operand fragment — 8 consecutive k of one row — is ONE
//     ds_read_b128.  Rows are 48 bytes (32 + 16 pad) to spread the banks.
//   * The next chunk's global loads are issued before the slabs and converted / written to LDS after them
//     (registers double-buffer, LDS is single-buffered: 137 KB per workgroup with 64 channels).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tron_hip.h"
#include "tron_conv.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int CIC = 16;           // input channels per K chunk
constexpr int TAPS_PAD = 10;      // 9 taps + one with zero weights: 5 slabs of 2 taps x 16 channels
constexpr int SLABS = 5;
constexpr int PITCH = 48;         // bytes per LDS row: 16 f16 + 16 bytes of padding
constexpr int NWM = 4;            // waves along M
constexpr int NT = 2;             // 16-channel N tiles per wave
constexpr float ACT_SCALE = 1.0f / 64.0f, ACT_UNSCALE = 64.0f, LO_SCALE = 2048.0f, LO_UNSCALE = 1.0f / 2048.0f;

template <int S_>
struct Cfg {
    static constexpr int S = S_;
    static constexpr int SP = S + 2;
    static constexpr int PLANE = SP * SP;
    static constexpr int SS = S * S;
    static constexpr int P = (S * S <= 144) ? (576 / (S * S)) : 1;
    static constexpr int PX = P * SS;
    static constexpr int PW = ((PX + NWM - 1) / NWM + 3) & ~3;         // pixels per M wave, multiple of 4
    static constexpr int MT = (PW + 15) / 16;
    static constexpr int IN_BYTES = P * PLANE * PITCH;                  // one half (hi or lo) of the input tile
    static_assert(S % 2 == 0, "even sides only");
    static_assert(MT <= 9, "accumulators: MT x NT x 2 x 4 registers must leave room for two waves per SIMD");
};

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

// v -> (hi, lo): v = hi + lo * 2^-11 up to 2^-22 |v|
__device__ __forceinline__ void split(float v, f16 &hi, f16 &lo)
{
    hi = (f16)v;
    lo = (f16)((v - (float)hi) * LO_SCALE);
}

// SMALL = the conv1 instantiation (cin 3 or 4, a single chunk whose absent channels stay zero); with in_codes its
// input is the int8 observation codes, otherwise the f32 planes.  COUT_WG = 32 * NWN output channels per workgroup.
template <int S, int NWN, bool SMALL>
__global__ __launch_bounds__(64 * NWM * NWN, NWN == 2 ? 2 : 1) void k_conv3x3_f16(
    const void *__restrict__ in, const float *__restrict__ wgt, const float *__restrict__ bias,
    const float *__restrict__ res, float *__restrict__ out, float *__restrict__ pre_out, int B, int cin, int cout,
    int groups, float plane4, int apply_mish, int in_codes)
{
    using C = Cfg<S>;
    constexpr int THREADS = 64 * NWM * NWN;
    constexpr int COUT_WG = 32 * NWN;
    constexpr int W_BYTES = TAPS_PAD * COUT_WG * PITCH;                 // one half of the weight tile
    constexpr int IN_Q = C::P * CIC * C::SS / 4;                        // float4s of an input chunk [P][16][SS/4]
    constexpr int W_Q = COUT_WG * CIC * 9 / 4;                          // float4s of a weight chunk [co][16 x 9]
    constexpr int IN_LD = (IN_Q + THREADS - 1) / THREADS, W_LD = (W_Q + THREADS - 1) / THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *in_h = lds, *in_l = lds + C::IN_BYTES, *w_h = lds + 2 * C::IN_BYTES, *w_l = w_h + W_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & (NWM - 1), wn = wave / NWM;
    const int li = lane & 15, g = lane >> 4, tsel = g >> 1, oct = g & 1;
    const int group = blockIdx.x % groups, chalf = blockIdx.x / groups;            // cout > COUT_WG: channel slices
    const int co0 = chalf * COUT_WG;
    const int img0 = group * C::P;
    const int nchunks = SMALL ? 1 : cin / CIC;
    const int last_img = B - 1 - img0;

    // zero all of LDS once: halo pixels, padding, absent channels and the tenth tap stay zero for good
    for (int i = tid; i < (2 * C::IN_BYTES + 2 * W_BYTES) / 16; i += THREADS)
        reinterpret_cast<uint4 *>(lds)[i] = make_uint4(0u, 0u, 0u, 0u);

    // per-lane operand bases (bytes)
    int a_base[C::MT];
#pragma unroll
    for (int t = 0; t < C::MT; ++t) {
        int px = wm * C::PW + 16 * t + li;
        px = px < C::PX ? px : C::PX - 1;
        const int img = px / C::SS, p = px - img * C::SS;
        const int y = p / S, x = p - y * S;
        a_base[t] = (img * C::PLANE + y * C::SP + x) * PITCH + oct * 16;
    }
    // byte offset of slab s's tap for this lane: taps 2s (tsel 0) / 2s+1 (tsel 1); the tenth tap (weights zero) reads
    // the ninth's pixels.  Two literals and a select per use instead of five registers held across the MFMA loop.
    auto tap_offset = [&](int s) {
        const int t0 = 2 * s, t1 = (2 * s + 1 < 9) ? 2 * s + 1 : 8;
        const int o0 = ((t0 / 3) * C::SP + (t0 % 3)) * PITCH, o1 = ((t1 / 3) * C::SP + (t1 % 3)) * PITCH;
        return tsel ? o1 : o0;
    };
    const int b_base = (tsel * COUT_WG + wn * 32 + li) * PITCH + oct * 16;

    const size_t wg_base = ((size_t)img0 * cout + co0) * C::SS;
    const float *wgt_wg = wgt + (size_t)co0 * cin * 9;
    const float *in_wg = (SMALL && in_codes) ? nullptr : reinterpret_cast<const float *>(in) + (size_t)img0 * cin * C::SS;

    f32x4 acc0[C::MT][NT], acc1[C::MT][NT];
#pragma unroll
    for (int t = 0; t < C::MT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            acc0[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc1[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

    f32x4 rin[IN_LD], rw[W_LD];
    uint32_t rcodes = 0x01010101u;
    float rw1[5];

#define TRON_LOAD_CHUNK(c_)                                                                                          \
    do {                                                                                                              \
        if (SMALL) {                                                                                                  \
            if (in_codes) {                                                                                           \
                const int w_ = tid < C::PX / 4 ? tid : C::PX / 4 - 1;                                                 \
                const int im_ = (w_ * 4) / C::SS;                                                                     \
                const int ims_ = im_ < last_img ? im_ : last_img;                                                     \
                rcodes = reinterpret_cast<const uint32_t *>(in)[(size_t)(img0 + ims_) * (C::SS / 4) + (w_ - im_ * (C::SS / 4))]; \
            } else {                                                                                                  \
                _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                       \
                    int q_ = tid + j * THREADS;                                                                       \
                    q_ = q_ < C::P * cin * (C::SS / 4) ? q_ : C::P * cin * (C::SS / 4) - 1;                           \
                    const int im_ = q_ / (cin * (C::SS / 4)), r_ = q_ - im_ * (cin * (C::SS / 4));                    \
                    const int ims_ = im_ < last_img ? im_ : last_img;                                                 \
                    rin[j < IN_LD ? j : 0] = *reinterpret_cast<const f32x4 *>(in_wg + (ims_ * cin * C::SS + r_ * 4)); \
                }                                                                                                     \
            }                                                                                                         \
            _Pragma("unroll") for (int j = 0; j < 5; ++j) {                                                           \
                int i_ = tid + j * THREADS;                                                                           \
                i_ = i_ < COUT_WG * cin * 9 ? i_ : COUT_WG * cin * 9 - 1;                                             \
                rw1[j] = wgt_wg[i_];                                                                                  \
            }                                                                                                         \
        } else {                                                                                                      \
            int tidl_ = tid;                                                                                          \
            asm volatile("" : "+v"(tidl_));                                                                           \
            _Pragma("unroll") for (int j = 0; j < IN_LD; ++j) {                                                       \
                int q_ = tidl_ + j * THREADS;                                                                         \
                q_ = q_ < IN_Q ? q_ : IN_Q - 1;                                                                       \
                const int im_ = q_ / (CIC * C::SS / 4), r_ = q_ - im_ * (CIC * C::SS / 4);                            \
                const int ims_ = im_ < last_img ? im_ : last_img;                                                     \
                rin[j] = *reinterpret_cast<const f32x4 *>(in_wg + ((ims_ * cin + (c_) * CIC) * C::SS + r_ * 4));      \
            }                                                                                                         \
            _Pragma("unroll") for (int j = 0; j < W_LD; ++j) {                                                        \
                int q_ = tidl_ + j * THREADS;                                                                         \
                q_ = q_ < W_Q ? q_ : W_Q - 1;                                                                         \
                const int co_ = q_ / 36, j4_ = q_ - co_ * 36;                                                         \
                rw[j] = *reinterpret_cast<const f32x4 *>(wgt_wg + ((co_ * cin + (c_) * CIC) * 9 + j4_ * 4));          \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)

    // one activation -> its (hi, lo) halves at pixel `pix_` (padded index), channel `ci_` of the chunk
#define TRON_PUT_IN(pix_, ci_, v_)                                                                                   \
    do {                                                                                                              \
        f16 h_, l_;                                                                                                   \
        split((v_) * ACT_SCALE, h_, l_);                                                                              \
        *reinterpret_cast<f16 *>(in_h + (pix_) * PITCH + (ci_) * 2) = h_;                                             \
        *reinterpret_cast<f16 *>(in_l + (pix_) * PITCH + (ci_) * 2) = l_;                                             \
    } while (0)
#define TRON_PUT_W(tap_, co_, ci_, v_)                                                                               \
    do {                                                                                                              \
        f16 h_, l_;                                                                                                   \
        split((v_), h_, l_);                                                                                          \
        *reinterpret_cast<f16 *>(w_h + ((tap_) * COUT_WG + (co_)) * PITCH + (ci_) * 2) = h_;                          \
        *reinterpret_cast<f16 *>(w_l + ((tap_) * COUT_WG + (co_)) * PITCH + (ci_) * 2) = l_;                          \
    } while (0)

#define TRON_STORE_CHUNK()                                                                                           \
    do {                                                                                                              \
        if (SMALL) {                                                                                                  \
            if (in_codes) {                                                                                           \
                if (tid < C::PX / 4) {                                                                                \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                   \
                        const int px_ = tid * 4 + e;                                                                  \
                        const int im_ = px_ / C::SS, p_ = px_ - im_ * C::SS;                                          \
                        const int y_ = p_ / S, xx_ = p_ - y_ * S;                                                     \
                        const int v_ = im_ <= last_img ? (int)(int8_t)(rcodes >> (8 * e)) : 1;                        \
                        const int pix_ = im_ * C::PLANE + (y_ + 1) * C::SP + (xx_ + 1);                               \
                        TRON_PUT_IN(pix_, 0, (v_ == -1) ? 1.0f : 0.0f);                         /* util.py:18-19 */  \
                        TRON_PUT_IN(pix_, 1, (v_ == -2) ? 1.0f : (v_ == 10) ? 10.0f : 0.0f);    /* util.py:20-21,26-27 */ \
                        TRON_PUT_IN(pix_, 2, (v_ == -3) ? 1.0f : (v_ == -10) ? 10.0f : 0.0f);                         \
                        if (cin == 4) TRON_PUT_IN(pix_, 3, plane4);                                                   \
                    }                                                                                                 \
                }                                                                                                     \
            } else {                                                                                                  \
                _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                       \
                    const int q_ = tid + j * THREADS;                                                                 \
                    if (q_ < C::P * cin * (C::SS / 4)) {                                                              \
                        const int im_ = q_ / (cin * (C::SS / 4)), r_ = q_ - im_ * (cin * (C::SS / 4));                \
                        const int ci_ = r_ / (C::SS / 4), p0_ = (r_ - ci_ * (C::SS / 4)) * 4;                         \
                        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                               \
                            const int p_ = p0_ + e;                                                                   \
                            const int y_ = p_ / S, xx_ = p_ - y_ * S;                                                 \
                            TRON_PUT_IN(im_ * C::PLANE + (y_ + 1) * C::SP + (xx_ + 1), ci_,                           \
                                        im_ <= last_img ? rin[j < IN_LD ? j : 0][e] : 0.0f);                          \
                        }                                                                                             \
                    }                                                                                                 \
                }                                                                                                     \
            }                                                                                                         \
            _Pragma("unroll") for (int j = 0; j < 5; ++j) {                                                           \
                const int i_ = tid + j * THREADS;                                                                     \
                if (i_ < COUT_WG * cin * 9) {                                                                         \
                    const int co_ = i_ / (cin * 9), k_ = i_ - co_ * (cin * 9);                                        \
                    const int ci_ = k_ / 9, tap_ = k_ - ci_ * 9;                                                      \
                    TRON_PUT_W(tap_, co_, ci_, rw1[j]);                                                               \
                }                                                                                                     \
            }                                                                                                         \
        } else {                                                                                                      \
            /* an opaque copy of tid: the ~40 LDS destinations are recomputed here each chunk; hoisted out of the  */ \
            /* chunk loop they would sit in registers across the MFMAs, which have none to spare                  */ \
            int tidv_ = tid;                                                                                          \
            asm volatile("" : "+v"(tidv_));                                                                           \
            _Pragma("unroll") for (int j = 0; j < IN_LD; ++j) {                                                       \
                const int q_ = tidv_ + j * THREADS;                                                                   \
                if (q_ < IN_Q) {                                                                                      \
                    const int im_ = q_ / (CIC * C::SS / 4), r_ = q_ - im_ * (CIC * C::SS / 4);                        \
                    const int ci_ = r_ / (C::SS / 4), p0_ = (r_ - ci_ * (C::SS / 4)) * 4;                             \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                   \
                        const int p_ = p0_ + e;                                                                       \
                        const int y_ = p_ / S, xx_ = p_ - y_ * S;                                                     \
                        TRON_PUT_IN(im_ * C::PLANE + (y_ + 1) * C::SP + (xx_ + 1), ci_, im_ <= last_img ? rin[j][e] : 0.0f); \
                    }                                                                                                 \
                }                                                                                                     \
            }                                                                                                         \
            _Pragma("unroll") for (int j = 0; j < W_LD; ++j) {                                                        \
                const int q_ = tidv_ + j * THREADS;                                                                   \
                if (q_ < W_Q) {                                                                                       \
                    const int co_ = q_ / 36, j4_ = q_ - co_ * 36;                                                     \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                   \
                        const int k_ = j4_ * 4 + e;                                                                   \
                        const int ci_ = k_ / 9, tap_ = k_ - ci_ * 9;                                                  \
                        TRON_PUT_W(tap_, co_, ci_, rw[j][e]);                                                         \
                    }                                                                                                 \
                }                                                                                                     \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)

    __syncthreads();
    TRON_LOAD_CHUNK(0);
    TRON_STORE_CHUNK();
    __syncthreads();

    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        if (more) TRON_LOAD_CHUNK(c + 1);                                // in flight under the MFMAs
#pragma unroll
        for (int s = 0; s < SLABS; ++s) {
            f16x8 bh[NT], bl[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int bo = b_base + (2 * s * COUT_WG + n * 16) * PITCH;
                bh[n] = *reinterpret_cast<const f16x8 *>(w_h + bo);
                bl[n] = *reinterpret_cast<const f16x8 *>(w_l + bo);
            }
            // two-deep register pipeline over the M tiles: tile t+1's fragments are read while tile t's six MFMAs
            // issue; sched_barrier keeps the compiler from hoisting a whole slab of reads (72 registers) at once
            f16x8 ah[2], al[2];
            const int toff = tap_offset(s);
            ah[0] = *reinterpret_cast<const f16x8 *>(in_h + a_base[0] + toff);
            al[0] = *reinterpret_cast<const f16x8 *>(in_l + a_base[0] + toff);
#pragma unroll
            for (int t = 0; t < C::MT; ++t) {
                if (t + 1 < C::MT) {
                    ah[(t + 1) & 1] = *reinterpret_cast<const f16x8 *>(in_h + a_base[t + 1] + toff);
                    al[(t + 1) & 1] = *reinterpret_cast<const f16x8 *>(in_l + a_base[t + 1] + toff);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    acc0[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t & 1], bh[n], acc0[t][n], 0, 0, 0);
                    acc1[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t & 1], bl[n], acc1[t][n], 0, 0, 0);
                    acc1[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t & 1], bh[n], acc1[t][n], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (more) {
            __syncthreads();                                             // everyone is done reading this chunk
            TRON_STORE_CHUNK();
            __syncthreads();
        }
    }
#undef TRON_LOAD_CHUNK
#undef TRON_STORE_CHUNK
#undef TRON_PUT_IN
#undef TRON_PUT_W

    // epilogue (as tron_conv.hip): D row = 4 * (lane >> 4) + r (pixel), column = lane & 15 (channel)
    const int pxw_end = (wm + 1) * C::PW < C::PX ? (wm + 1) * C::PW : C::PX;
    int o[C::MT];
    bool live[C::MT];
#pragma unroll
    for (int t = 0; t < C::MT; ++t) {
        const int px = wm * C::PW + 16 * t + 4 * g;
        const int img = px / C::SS, p = px - img * C::SS;
        live[t] = px < pxw_end && img <= last_img;
        o[t] = (img * cout + wn * 32 + li) * C::SS + p;
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const float bv = bias ? bias[co0 + wn * 32 + n * 16 + li] : 0.0f;
#pragma unroll
        for (int t = 0; t < C::MT; ++t) acc0[t][n] = (acc0[t][n] + acc1[t][n] * LO_UNSCALE) * ACT_UNSCALE + bv;
    }
    if (res) {
        const float *res_wg = res + wg_base;
        f32x4 r[C::MT][NT];
#pragma unroll
        for (int t = 0; t < C::MT; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n)
                r[t][n] = live[t] ? *reinterpret_cast<const f32x4 *>(res_wg + o[t] + n * 16 * C::SS) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < C::MT; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc0[t][n] += r[t][n];
    }
    float *out_wg = out + wg_base;
    float *pre_wg = pre_out ? pre_out + wg_base : nullptr;
#pragma unroll
    for (int t = 0; t < C::MT; ++t) {
        if (!live[t]) continue;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            f32x4 v = acc0[t][n];
            if (pre_wg) *reinterpret_cast<f32x4 *>(pre_wg + o[t] + n * 16 * C::SS) = v;
            if (apply_mish) v = (f32x4){mish1(v[0]), mish1(v[1]), mish1(v[2]), mish1(v[3])};
            *reinterpret_cast<f32x4 *>(out_wg + o[t] + n * 16 * C::SS) = v;
        }
    }
}

template <int S, int NWN, bool SMALL>
int launch(const void *in, const float *wgt, const float *bias, const float *res, float *out, float *pre_out, int64_t B,
           int cin, int cout, float plane4, int apply_mish, int in_codes, hipStream_t st)
{
    using C = Cfg<S>;
    constexpr int COUT_WG = 32 * NWN;
    constexpr size_t LDS_BYTES = 2 * (size_t)C::IN_BYTES + 2 * (size_t)TAPS_PAD * COUT_WG * PITCH;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    auto kern = k_conv3x3_f16<S, NWN, SMALL>;
    static uint64_t prepared = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_NO_DEVICE; }
    if (!(prepared & (1ull << (dev & 63)))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)LDS_BYTES) != hipSuccess)
            (void)hipGetLastError();
        prepared |= 1ull << (dev & 63);
    }
    const int64_t groups = (B + C::P - 1) / C::P;
    const int64_t blocks = groups * (cout / COUT_WG);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * NWM * NWN), LDS_BYTES, st, in, wgt, bias, res, out, pre_out,
                       (int)B, cin, cout, (int)groups, plane4, apply_mish, in_codes);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

}  // namespace

// called by tron_conv3x3_fwd (tron_conv.hip) after it validated the arguments; TRON_ERR_UNSUPPORTED = not this shape
int tron_conv3x3_f16x3(const void *in, int in_is_codes, const float *weight, const float *bias, const float *residual,
                       float *out, float *pre_out, int64_t batch, int cin, int cout, int side, float plane4,
                       int apply_mish, hipStream_t st)
{
    const bool small = cin == 3 || cin == 4;
    if (side != 12 || (!small && cin % CIC != 0)) return TRON_ERR_UNSUPPORTED;
    if (small)
        return launch<12, 1, true>(in, weight, bias, residual, out, pre_out, batch, cin, cout, plane4, apply_mish,
                                   in_is_codes ? 1 : 0, st);
    if (cout == 64)
        return launch<12, 2, false>(in, weight, bias, residual, out, pre_out, batch, cin, cout, plane4, apply_mish, 0, st);
    return launch<12, 1, false>(in, weight, bias, residual, out, pre_out, batch, cin, cout, plane4, apply_mish, 0, st);
}
