// tron_conv.hip — the 3x3 convolutions of the reference's CNN (Net/DQNNet.py:10-17,33-50: conv1..conv6, each
// followed by bias, an optional residual add and mish) as ONE fused gfx950 kernel per layer:
// implicit GEMM on the fp32 matrix cores (v_mfma_f32_16x16x4_f32 — exact fp32, a k-ordered fmaf chain, so
// Q-values stay within the 1e-5 the north star asks for), bias + residual + mish in the epilogue, NCHW in
// and out (no layout transposes), and for conv1 the int8 observation codes (map.py:67-84) are expanded
// to the pop_up planes (util.py:11-37) while they are staged — the f32 planes never exist in HBM.
//
// Mapping (GEMM view: M = pixels, N = output channels, K = input channels x 9 taps).
//   * A workgroup of 4 waves owns P whole images (P = 4 at 12x12: one image per wave; P = 1 at 26x26: a
//     quarter image per wave) and ALL output channels: every wave holds MT x NT accumulator tiles of
//     16 px x 16 channels (144-176 VGPRs) — the input is read from HBM once, the weights stream from L2.
//   * K is walked in chunks of 8 input channels.  A chunk's input planes sit in LDS with their zero halo
//     ([ci][image][S+2][S+2], fp32), its weights as [k-step][k-sub][cout]; both double-buffered: while the
//     matrix cores chew on chunk c (18 k-steps x MT x NT MFMAs, ~20 k cycles) the next chunk's global
//     loads are in flight and are written to the other buffer afterwards.
//   * The A operand of a k-step is read straight from the padded planes: lane (pixel i, k-sub s) reads
//     plane[ci0 + s][y_i + ky][x_i + kx] — per-lane base in a VGPR, (ci, tap) in the instruction's
//     immediate offset — so there is no im2col and no address arithmetic in the loop.
// The matrix pipe is the bound: one 16x16x4 fp32 MFMA holds it for 32 cycles against ~0.4 ds_read_b32 per
// MFMA; HBM traffic is one pass over the activations (~1 byte per 60 flops).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tron_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CIC = 8;           // input channels per K chunk
constexpr int KSTEPS = 18;       // 9 taps x 2 channel quads per chunk
constexpr int NWAVES = 4;
constexpr int THREADS = 64 * NWAVES;

template <int S_, int NT_>
struct Cfg {
    static constexpr int S = S_, NT = NT_;
    static constexpr int SP = S + 2;                       // padded side
    static constexpr int PLANE = SP * SP;                  // one padded plane, floats
    static constexpr int SS = S * S;
    static constexpr int P = (S * S <= 144) ? (576 / (S * S)) : 1;   // images per workgroup
    static constexpr int PX = P * SS;                      // pixels per workgroup
    static constexpr int PW = ((PX + NWAVES - 1) / NWAVES + 3) & ~3;   // pixels per wave, multiple of 4
    static constexpr int MT = (PW + 15) / 16;              // 16-pixel M tiles per wave
    static constexpr int CI_STRIDE = ((P * PLANE + 15) & ~31) + 16;    // floats; = 16 mod 32: the two k-subs
                                                                       // of a ds_read_b32 lane group land 16 banks apart
    static constexpr int COUT = 16 * NT;
    static constexpr int WROW = COUT + 16;                 // floats per (k-step, k-sub) weight row; = 16 mod 32 likewise
    static constexpr int IN_BUF = CIC * CI_STRIDE;         // floats per input buffer
    static constexpr int W_BUF = KSTEPS * 4 * WROW;        // floats per weight buffer
    static constexpr int IN_LD = (P * CIC * SS / 4 + THREADS - 1) / THREADS;   // float4 global loads per thread per chunk
    static constexpr int W_LD = (COUT * CIC * 9 / 4 + THREADS - 1) / THREADS;   // float4 loads of the weight slice W[co][8c..8c+8)[3][3]
    static constexpr size_t LDS_BYTES = (size_t)(2 * IN_BUF + 2 * W_BUF) * 4;
    static_assert(S % 2 == 0, "even sides only: 4-pixel groups must not straddle images");
    static_assert(P * PLANE <= CI_STRIDE, "plane stride");
    static_assert((CIC - 1) * CI_STRIDE * 4 + (2 * SP + 2) * 4 < 65536, "ds_read immediate offset");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

__device__ __forceinline__ float mish1(float x)              // same closed form as csrc/tron_nn.hip
{
    if (x > 20.0f) return x;
    const float e = expf(x);
    const float n = e * (e + 2.0f);
    return x * (n / (n + 2.0f));
}

// CODES: `in` is int8 observation codes [B][S*S]; channels 0..2 are the pop_up planes (wall, my, enemy),
// channel 3 the constant `plane4` when cin == 4 (Game.prob_map, game.py:124-132), the rest of the chunk zero.
template <int S, int NT, bool CODES>
__global__ __launch_bounds__(THREADS) void k_conv3x3(const void *__restrict__ in, const float *__restrict__ wgt,
                                                      const float *__restrict__ bias, const float *__restrict__ res,
                                                      float *__restrict__ out, float *__restrict__ pre_out, int B,
                                                      int cin, float plane4, int apply_mish)
{
    using C = Cfg<S, NT>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // LDS map (floats): input buffer 0 | input buffer 1 | weight buffer 0 | weight buffer 1

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, ks = lane >> 4;
    const int img0 = blockIdx.x * C::P;
    const int nchunks = CODES ? 1 : cin / CIC;

    // zero both input buffers once: the halo (and, for CODES, the unused channels) stay zero for good
    // (CODES: and the weight buffer, whose rows of the absent channels must read as zero)
    for (int i = tid; i < (2 * C::IN_BUF + (CODES ? C::W_BUF : 0)) / 4; i += THREADS)
        reinterpret_cast<float4 *>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    // per-lane operand bases
    int a_off[C::MT];
#pragma unroll
    for (int t = 0; t < C::MT; ++t) {
        int px = wave * C::PW + 16 * t + li;
        px = px < C::PX ? px : C::PX - 1;                       // surplus lanes read a valid pixel, never store
        const int img = px / C::SS, p = px - img * C::SS;
        const int y = p / S, x = p - y * S;
        a_off[t] = ks * C::CI_STRIDE + img * C::PLANE + y * C::SP + x;
    }
    const int b_off = ks * C::WROW + li;

    // D row = 4 * (lane >> 4) + r (pixel), column = lane & 15 (channel): 4 consecutive pixels per lane.
    // Accumulators start at the bias (the C input of the MFMA chain).
    const int pxw_end = (wave + 1) * C::PW < C::PX ? (wave + 1) * C::PW : C::PX;
    // workgroup-uniform 64-bit bases + 32-bit per-lane offsets (a workgroup's slice is < 1 MB)
    const float *res_wg = res ? res + (size_t)img0 * C::COUT * C::SS : nullptr;
    float *out_wg = out + (size_t)img0 * C::COUT * C::SS;
    float *pre_wg = pre_out ? pre_out + (size_t)img0 * C::COUT * C::SS : nullptr;
    f32x4 acc[C::MT][NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const float bv = bias ? bias[n * 16 + li] : 0.0f;
#pragma unroll
        for (int t = 0; t < C::MT; ++t) acc[t][n] = (f32x4){bv, bv, bv, bv};
    }

    // Staging registers of the next chunk.  Every global load is unconditional (indices are clamped into the
    // buffers; what a surplus thread or a past-the-batch image loads is dropped at the LDS write), so the loads
    // of a chunk go out back to back and complete under the MFMAs — no per-load waits, no private-memory arrays.
    f32x4 rin[C::IN_LD], rw[C::W_LD];               // native vectors: HIP's float4 struct arrays end up in scratch
    uint32_t rcodes = 0x01010101u;
    constexpr int IN_Q = C::P * CIC * C::SS / 4;          // float4s of one input chunk: [P][CIC][SS/4]
    constexpr int W_Q = C::COUT * CIC * 9 / 4;             // float4s of one weight chunk: [COUT][8 ci x 9 taps]
    float rw1[5];                                          // conv1's whole weight (<= 32 x 36 floats), CODES only
    const int last_img = B - 1 - img0;                     // >= 0: the grid covers ceil(B / P) image groups

#define TRON_LOAD_CHUNK(c_)                                                                                          \
    do {                                                                                                              \
        if (CODES) {                                                                                                  \
            const int w_ = tid < C::PX / 4 ? tid : C::PX / 4 - 1;                                                     \
            const int im_ = (w_ * 4) / C::SS;                                                                         \
            const int ims_ = im_ < last_img ? im_ : last_img;                                                         \
            rcodes = reinterpret_cast<const uint32_t *>(in)[(size_t)(img0 + ims_) * (C::SS / 4) + (w_ - im_ * (C::SS / 4))]; \
        } else {                                                                                                      \
            const float *x_ = reinterpret_cast<const float *>(in);                                                    \
            _Pragma("unroll") for (int j = 0; j < C::IN_LD; ++j) {                                                    \
                int q_ = tid + j * THREADS;                                                                           \
                q_ = q_ < IN_Q ? q_ : IN_Q - 1;                                                                       \
                const int im_ = q_ / (CIC * C::SS / 4), r_ = q_ - im_ * (CIC * C::SS / 4);                            \
                const int ims_ = im_ < last_img ? im_ : last_img;                                                     \
                rin[j] = *reinterpret_cast<const f32x4 *>(x_ + (size_t)img0 * cin * C::SS + ((ims_ * cin + (c_) * CIC) * C::SS + r_ * 4)); \
            }                                                                                                         \
        }                                                                                                             \
        if (CODES) {                        /* conv1: W[32][cin][3][3], cin 3 or 4: rows of 27 / 36 floats, scalar loads */ \
            _Pragma("unroll") for (int j = 0; j < 5; ++j) {                                                           \
                int i_ = tid + j * THREADS;                                                                           \
                i_ = i_ < C::COUT * cin * 9 ? i_ : C::COUT * cin * 9 - 1;                                             \
                rw1[j] = wgt[i_];                                                                                     \
            }                                                                                                         \
        } else {                            /* W[co][cin][3][3]: the 72 floats of channels 8c..8c+7 are contiguous per co */ \
            _Pragma("unroll") for (int j = 0; j < C::W_LD; ++j) {                                                     \
                int q_ = tid + j * THREADS;                                                                           \
                q_ = q_ < W_Q ? q_ : W_Q - 1;                                                                         \
                const int co_ = q_ / 18, j4_ = q_ - co_ * 18;                                                         \
                rw[j] = *reinterpret_cast<const f32x4 *>(wgt + ((co_ * cin + (c_) * CIC) * 9 + j4_ * 4));             \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)

#define TRON_STORE_CHUNK(b_)                                                                                         \
    do {                                                                                                              \
        float *ib_ = lds + (b_) * C::IN_BUF, *wb_ = lds + 2 * C::IN_BUF + (b_) * C::W_BUF;                            \
        if (CODES) {                                                                                                  \
            if (tid < C::PX / 4) {                                                                                    \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                       \
                    const int px_ = tid * 4 + e;                                                                      \
                    const int im_ = px_ / C::SS, p_ = px_ - im_ * C::SS;                                              \
                    const int y_ = p_ / S, xx_ = p_ - y_ * S;                                                         \
                    const int v_ = im_ <= last_img ? (int)(int8_t)(rcodes >> (8 * e)) : 1;                            \
                    float *d_ = ib_ + im_ * C::PLANE + (y_ + 1) * C::SP + (xx_ + 1);                                  \
                    d_[0] = (v_ == -1) ? 1.0f : 0.0f;                                   /* util.py:18-19 */          \
                    d_[C::CI_STRIDE] = (v_ == -2) ? 1.0f : (v_ == 10) ? 10.0f : 0.0f;    /* util.py:20-21,26-27 */    \
                    d_[2 * C::CI_STRIDE] = (v_ == -3) ? 1.0f : (v_ == -10) ? 10.0f : 0.0f;                            \
                    if (cin == 4) d_[3 * C::CI_STRIDE] = plane4;                                                      \
                }                                                                                                     \
            }                                                                                                         \
        } else {                                                                                                      \
            _Pragma("unroll") for (int j = 0; j < C::IN_LD; ++j) {                                                    \
                const int q_ = tid + j * THREADS;                                                                     \
                if (q_ < IN_Q) {                                                                                      \
                    const int im_ = q_ / (CIC * C::SS / 4), r_ = q_ - im_ * (CIC * C::SS / 4);                        \
                    const int ci_ = r_ / (C::SS / 4), p0_ = (r_ - ci_ * (C::SS / 4)) * 4;                             \
                    const bool live_ = im_ <= last_img;                                                               \
                    const float v_[4] = {rin[j][0], rin[j][1], rin[j][2], rin[j][3]};                                     \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                   \
                        const int p_ = p0_ + e;                                                                       \
                        const int y_ = p_ / S, xx_ = p_ - y_ * S;                                                     \
                        ib_[ci_ * C::CI_STRIDE + im_ * C::PLANE + (y_ + 1) * C::SP + (xx_ + 1)] = live_ ? v_[e] : 0.0f; \
                    }                                                                                                 \
                }                                                                                                     \
            }                                                                                                         \
        }                                                                                                             \
        /* weights -> [k-step = tap * 2 + channel quad][k-sub = channel & 3][cout] */                                  \
        if (CODES) {                                                                                                  \
            _Pragma("unroll") for (int j = 0; j < 5; ++j) {                                                           \
                const int i_ = tid + j * THREADS;                                                                     \
                if (i_ < C::COUT * cin * 9) {                                                                         \
                    const int co_ = i_ / (cin * 9), k_ = i_ - co_ * (cin * 9);                                        \
                    const int ci_ = k_ / 9, tap_ = k_ - ci_ * 9;                                                      \
                    wb_[(tap_ * 8 + ci_) * C::WROW + co_] = rw1[j];                                                   \
                }                                                                                                     \
            }                                                                                                         \
        } else {                                                                                                      \
            _Pragma("unroll") for (int j = 0; j < C::W_LD; ++j) {                                                     \
                const int q_ = tid + j * THREADS;                                                                     \
                if (q_ < W_Q) {                                                                                       \
                    const int co_ = q_ / 18, j4_ = q_ - co_ * 18;                                                     \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                   \
                        const int k_ = j4_ * 4 + e;                                                                   \
                        const int ci_ = k_ / 9, tap_ = k_ - ci_ * 9;                                                  \
                        wb_[((tap_ * 2 + (ci_ >> 2)) * 4 + (ci_ & 3)) * C::WROW + co_] = rw[j][e];                    \
                    }                                                                                                 \
                }                                                                                                     \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)

    __syncthreads();                  // zero fill done
    TRON_LOAD_CHUNK(0);
    TRON_STORE_CHUNK(0);
    __syncthreads();

    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        if (more) TRON_LOAD_CHUNK(c + 1);                        // in flight under the MFMAs below
        const float *ib = lds + (c & 1) * C::IN_BUF, *wb = lds + 2 * C::IN_BUF + (c & 1) * C::W_BUF;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int cq = 0; cq < 2; ++cq) {
                const int koff = cq * 4 * C::CI_STRIDE + (tap / 3) * C::SP + (tap % 3);
                const int kstep = tap * 2 + cq;
                float a[C::MT], b[NT];
#pragma unroll
                for (int t = 0; t < C::MT; ++t) a[t] = ib[a_off[t] + koff];
#pragma unroll
                for (int n = 0; n < NT; ++n) b[n] = wb[b_off + kstep * 4 * C::WROW + n * 16];
#pragma unroll
                for (int t = 0; t < C::MT; ++t)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[n], acc[t][n], 0, 0, 0);
            }
        }
        if (more) TRON_STORE_CHUNK((c + 1) & 1);
        __syncthreads();
    }

#undef TRON_LOAD_CHUNK
#undef TRON_STORE_CHUNK

    // epilogue.  The residual is added to the finished sum (adding it first would round every one of the K
    // partial sums at the residual's magnitude).  All of a lane's residual loads are issued together — the
    // staging and operand registers are dead by now — so the wave pays ONE memory round trip, not one per tile.
    int o[C::MT];
    bool live[C::MT];
#pragma unroll
    for (int t = 0; t < C::MT; ++t) {
        const int px = wave * C::PW + 16 * t + 4 * ks;
        const int img = px / C::SS, p = px - img * C::SS;
        live[t] = px < pxw_end && img0 + img < B;
        o[t] = (img * C::COUT + li) * C::SS + p;
    }
    if (res_wg) {
        f32x4 r[C::MT][NT];
#pragma unroll
        for (int t = 0; t < C::MT; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n)
                r[t][n] = live[t] ? *reinterpret_cast<const f32x4 *>(res_wg + o[t] + n * 16 * C::SS) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < C::MT; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[t][n] += r[t][n];
    }
#pragma unroll
    for (int t = 0; t < C::MT; ++t) {
        if (!live[t]) continue;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            f32x4 v = acc[t][n];
            if (pre_wg) *reinterpret_cast<f32x4 *>(pre_wg + o[t] + n * 16 * C::SS) = v;
            if (apply_mish) v = (f32x4){mish1(v[0]), mish1(v[1]), mish1(v[2]), mish1(v[3])};
            *reinterpret_cast<f32x4 *>(out_wg + o[t] + n * 16 * C::SS) = v;
        }
    }
}

template <int S, int NT, bool CODES>
int launch_conv(const void *in, const float *wgt, const float *bias, const float *res, float *out, float *pre_out,
                int64_t B, int cin, float plane4, int apply_mish, hipStream_t st)
{
    using C = Cfg<S, NT>;
    auto kern = k_conv3x3<S, NT, CODES>;
    static uint64_t prepared = 0;     // hipFuncSetAttribute is per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_NO_DEVICE; }
    if (!(prepared & (1ull << (dev & 63)))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)C::LDS_BYTES) != hipSuccess)
            (void)hipGetLastError();
        prepared |= 1ull << (dev & 63);
    }
    const int64_t blocks = (B + C::P - 1) / C::P;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(THREADS), C::LDS_BYTES, st, in, wgt, bias, res, out, pre_out,
                       (int)B, cin, plane4, apply_mish);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

}  // namespace

extern "C" int tron_conv3x3_fwd(const void *in, int32_t in_is_codes, const float *weight, const float *bias,
                                const float *residual, float *out, float *pre_out, int64_t batch, int32_t cin,
                                int32_t cout, int32_t side, float plane4, int32_t apply_mish, void *stream)
{
    if (!in || !weight || !out || batch < 0 || cin < 1) return TRON_ERR_BAD_ARG;
    if (batch == 0) return TRON_OK;
    if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(weight) | reinterpret_cast<uintptr_t>(out) |
         reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(pre_out)) & 15u)
        return TRON_ERR_BAD_ARG;
    if (in_is_codes ? (cin != 3 && cin != 4) : (cin % CIC != 0)) return TRON_ERR_UNSUPPORTED;
    if (batch * cout * side * side > 0x7FFFFFFFll * 4) return TRON_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define TRON_CONV_CASE(S_, NT_)                                                                                            \
    if (side == S_ && cout == 16 * NT_)                                                                                    \
        return in_is_codes ? launch_conv<S_, NT_, true>(in, weight, bias, residual, out, pre_out, batch, cin, plane4,     \
                                                        apply_mish, st)                                                    \
                           : launch_conv<S_, NT_, false>(in, weight, bias, residual, out, pre_out, batch, cin, plane4,    \
                                                         apply_mish, st);
    TRON_CONV_CASE(12, 2)
    TRON_CONV_CASE(12, 4)
    TRON_CONV_CASE(26, 2)
    TRON_CONV_CASE(26, 4)
#undef TRON_CONV_CASE
    return TRON_ERR_UNSUPPORTED;
}
