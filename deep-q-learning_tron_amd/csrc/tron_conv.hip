// tron_conv.hip — the 3x3 convolutions of the reference's CNN (Net/DQNNet.py:10-17,33-50: conv1..conv6, each
// followed by bias, an optional residual add and mish) as ONE fused gfx950 kernel per layer:
// implicit GEMM on the fp32 matrix cores (v_mfma_f32_16x16x4_f32 — exact fp32, a k-ordered fmaf chain, so
// Q-values stay within the 1e-5 the north star asks for), bias + residual + mish in the epilogue, NCHW in
// and out (no layout transposes), the nn.Conv2d weight read as it is (reordered while it is staged), and
// for conv1 the int8 observation codes (map.py:67-84) expanded to the pop_up planes (util.py:11-37) on the
// way into LDS — the f32 planes never exist in HBM.
//
// Mapping (GEMM view: M = pixels, N = output channels, K = input channels x 9 taps).
//   * A workgroup of 4 waves owns P whole images (P = 4 at 12x12: one image per wave; P = 1 at 26x26: a
//     quarter image per wave) and 32 output channels: every wave holds MT x 2 accumulator tiles of
//     16 px x 16 channels (72-88 registers).  A 64-channel layer is two such workgroups per image group,
//     dispatched 8 apart so that they run at the same time on the same XCD and share the input in its L2.
//   * 77 KB of LDS and < 256 registers: TWO workgroups per CU, so one's staging, barriers, prologue and
//     epilogue hide under the other's MFMAs.
//   * K is walked in chunks of 8 input channels.  A chunk's input planes sit in LDS with their zero halo
//     ([ci][image][S+2][S+2], fp32), its weights as [k-step][k-sub][cout]; both double-buffered: the next
//     chunk's global loads are issued before the chunk's 18 k-steps (x MT x 2 MFMAs) and written to the other
//     buffer half way through them, in the matrix cores' shadow.
//   * The A operand of a k-step is read straight from the padded planes: lane (pixel i, k-sub s) reads
//     plane[ci0 + s][y_i + ky][x_i + kx] — per-lane base in a VGPR, (ci, tap) in the instruction's
//     immediate offset — so there is no im2col and no address arithmetic in the loop.  Operands of k-step
//     s+1 are fetched before the MFMAs of k-step s are issued.
// The matrix pipe is the bound: one 16x16x4 fp32 MFMA holds it for 32 cycles against ~0.6 ds_read_b32 per
// MFMA; HBM traffic is one pass over the activations (~1 byte per 60 flops).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tron_hip.h"
#include "tron_conv.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CIC = 8;           // input channels per K chunk
constexpr int KSTEPS = 18;       // 9 taps x 2 channel quads per chunk
constexpr int NWAVES = 4;
constexpr int THREADS = 64 * NWAVES;
constexpr int NT = 2;            // 16-channel N tiles per workgroup
constexpr int COUT_WG = 16 * NT; // output channels per workgroup

template <int S_>
struct Cfg {
    static constexpr int S = S_;
    static constexpr int SP = S + 2;                       // padded side
    static constexpr int PLANE = SP * SP;                  // one padded plane, floats
    static constexpr int SS = S * S;
    static constexpr int P = (S * S <= 144) ? (576 / (S * S)) : 1;   // images per workgroup
    static constexpr int PX = P * SS;                      // pixels per workgroup
    static constexpr int PW = ((PX + NWAVES - 1) / NWAVES + 3) & ~3;   // pixels per wave, multiple of 4
    static constexpr int MT = (PW + 15) / 16;              // 16-pixel M tiles per wave
    static constexpr int CI_STRIDE = ((P * PLANE + 15) & ~31) + 16;    // floats; = 16 mod 32: the two k-subs
                                                                       // of a ds_read_b32 lane group land 16 banks apart
    static constexpr int WROW = COUT_WG + 16;              // floats per (k-step, k-sub) weight row; = 16 mod 32 likewise
    static constexpr int IN_BUF = CIC * CI_STRIDE;         // floats per input buffer
    static constexpr int W_BUF = KSTEPS * 4 * WROW;        // floats per weight buffer
    static constexpr int IN_Q = P * CIC * SS / 4;          // float4s of one input chunk: [P][CIC][SS/4]
    static constexpr int W_Q = COUT_WG * CIC * 9 / 4;      // float4s of one weight chunk: [32 co][8 ci x 9 taps]
    static constexpr int IN_LD = (IN_Q + THREADS - 1) / THREADS;   // float4 global loads per thread per chunk
    static constexpr int W_LD = (W_Q + THREADS - 1) / THREADS;
    static constexpr int DUMP = 2 * IN_BUF + 2 * W_BUF;    // 64 floats that absorb the LDS writes of surplus threads
    static constexpr size_t LDS_BYTES = (size_t)(DUMP + 64) * 4;
    static_assert(S % 2 == 0, "even sides only: 4-pixel groups must not straddle images");
    static_assert(P * PLANE <= CI_STRIDE, "plane stride");
    static_assert((CIC - 1) * CI_STRIDE * 4 + (2 * SP + 2) * 4 < 65536, "ds_read immediate offset");
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};

// mish(x) = x * tanh(softplus(x)) = x * n / (n + 2), n = e (e + 2), e = exp(x) (csrc/tron_nn.hip has the derivation).
// The epilogue runs 72-88 of these per lane with the matrix pipe idle, so it is built from the two hardware
// transcendentals: exp2 of x * log2(e) — its argument rounding costs |x| 2^-24 relative in e, which mish turns into
// < 4e-8 ABSOLUTE (x^2 e^x <= 0.55 where e matters, and n / (n + 2) -> 1 where e is large) — and a reciprocal
// refined by one Newton step (v_rcp_f32 alone is 1 ulp) instead of the ~10-instruction IEEE division.
__device__ __forceinline__ float mish1(float x)
{
    const float e = __builtin_amdgcn_exp2f(x * 1.44269504088896341f);
    const float n = __fmaf_rn(e, e, e + e);
    const float d = n + 2.0f;
    float r = __builtin_amdgcn_rcpf(d);
    r = __fmaf_rn(r, __fmaf_rn(-d, r, 1.0f), r);
    const float y = x * (n * r);
    return x > 20.0f ? x : y;                        // e * e overflows beyond x = 44; the quotient is 1 from 20 on
}

// operands of k-step KS (compile-time) of the chunk in (ib, wb) -> a[MT], b[NT]
template <class C, int KS>
__device__ __forceinline__ void load_frags(const float *ib, const float *wb, const int (&a_off)[C::MT], int b_off,
                                           float (&a)[C::MT], float (&b)[NT])
{
    constexpr int tap = KS >> 1, cq = KS & 1;
    constexpr int koff = cq * 4 * C::CI_STRIDE + (tap / 3) * C::SP + (tap % 3);
#pragma unroll
    for (int t = 0; t < C::MT; ++t) a[t] = ib[a_off[t] + koff];
#pragma unroll
    for (int n = 0; n < NT; ++n) b[n] = wb[b_off + KS * 4 * C::WROW + n * 16];
}

template <class C>
__device__ __forceinline__ void mfma_step(f32x4 (&acc)[C::MT][NT], const float (&a)[C::MT], const float (&b)[NT])
{
#pragma unroll
    for (int t = 0; t < C::MT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[n], acc[t][n], 0, 0, 0);
}

// k-steps [K0, K1) of one chunk as a two-deep register pipeline: the operands of step s+1 are read while the
// MFMAs of step s issue; sched_barrier keeps the compiler from hoisting more reads (and registers) than that.
// On entry a0/b0 hold step K0's operands; on exit they hold step K1's when K1 < KSTEPS.
template <class C, int K0, int K1>
__device__ __forceinline__ void k_range(f32x4 (&acc)[C::MT][NT], const float *ib, const float *wb,
                                        const int (&a_off)[C::MT], int b_off, float (&a0)[C::MT], float (&b0)[NT],
                                        float (&a1)[C::MT], float (&b1)[NT])
{
    if constexpr (K0 < K1) {
        if constexpr (K0 + 1 < KSTEPS) load_frags<C, K0 + 1>(ib, wb, a_off, b_off, a1, b1);
        __builtin_amdgcn_sched_barrier(0);          // reads first: left alone the scheduler sinks them behind the MFMAs
        mfma_step<C>(acc, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        k_range<C, K0 + 1, K1>(acc, ib, wb, a_off, b_off, a1, b1, a0, b0);     // buffers swap roles
    }
}

// CODES = the conv1 instantiation (cin 3 or 4, one K chunk whose absent channels are zero).  With in_codes `in` is
// int8 observation codes [B][S*S]: channels 0..2 are the pop_up planes (wall, my, enemy), channel 3 the constant
// `plane4` when cin == 4 (Game.prob_map, game.py:124-132).  Without, `in` is those planes as f32 [B][cin][S*S].
template <int S, bool CODES>
__global__ __launch_bounds__(THREADS, 2) void k_conv3x3(const void *__restrict__ in, const float *__restrict__ wgt,
                                                        const float *__restrict__ bias, const float *__restrict__ res,
                                                        float *__restrict__ out, float *__restrict__ pre_out, int B,
                                                        int cin, int cout, int groups, float plane4, int apply_mish,
                                                        int in_codes)
{
    using C = Cfg<S>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // LDS map (floats): input buffer 0 | input buffer 1 | weight buffer 0 | weight buffer 1 | dump

    // block -> (image group, channel half).  With two halves, blocks id and id + 8 — same XCD under the observed
    // round-robin placement, dispatched together — take the two halves of one image group (speed only).
    const int id = blockIdx.x;
    const int group = cout == 64 ? ((id >> 4) * 8 + (id & 7)) : id;
    const int co0 = cout == 64 ? ((id >> 3) & 1) * COUT_WG : 0;
    if (group >= groups) return;                              // whole workgroup leaves before any barrier

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, ks = lane >> 4;
    const int img0 = group * C::P;
#ifdef TRON_CONV_STAMPS     // diagnostic build only: pre_out is a stamp buffer u64[blocks][6], never an output
    unsigned long long *stamps = reinterpret_cast<unsigned long long *>(pre_out) + (size_t)id * 6;
    pre_out = nullptr;
    if (tid == 0) { stamps[0] = __builtin_amdgcn_s_memtime(); stamps[1] = __builtin_amdgcn_s_memrealtime(); }
#endif
    const int nchunks = CODES ? 1 : cin / CIC;
    const int last_img = B - 1 - img0;                         // >= 0

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

    // workgroup-uniform 64-bit bases + 32-bit per-lane offsets (a workgroup's slice is < 1 MB)
    const size_t wg_base = ((size_t)img0 * cout + co0) * C::SS;
    const float *wgt_wg = wgt + (size_t)co0 * cin * 9;
    const float *in_wg = (CODES && in_codes) ? nullptr : reinterpret_cast<const float *>(in) + (size_t)img0 * cin * C::SS;

    // D row = 4 * (lane >> 4) + r (pixel), column = lane & 15 (channel): 4 consecutive pixels per lane.
    // Accumulators start at the bias (the C input of the MFMA chain).
    f32x4 acc[C::MT][NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const float bv = bias ? bias[co0 + n * 16 + li] : 0.0f;
#pragma unroll
        for (int t = 0; t < C::MT; ++t) acc[t][n] = (f32x4){bv, bv, bv, bv};
    }

    // Staging registers of the next chunk.  Every global load is unconditional (indices are clamped into the
    // buffers; what a surplus thread or a past-the-batch image loads is dropped at the LDS write), so the loads
    // of a chunk go out back to back and complete under the MFMAs — no per-load waits, no private-memory arrays.
    f32x4 rin[C::IN_LD], rw[C::W_LD];               // native vectors: HIP's float4 struct arrays end up in scratch
    uint32_t rcodes = 0x01010101u;
    float rw1[5];                                   // conv1's whole weight slice (<= 32 x 36 floats), CODES only

#define TRON_LOAD_CHUNK(c_)                                                                                          \
    do {                                                                                                              \
        if (CODES) {                                                                                                  \
            if (in_codes) {                                                                                           \
                const int w_ = tid < C::PX / 4 ? tid : C::PX / 4 - 1;                                                 \
                const int im_ = (w_ * 4) / C::SS;                                                                     \
                const int ims_ = im_ < last_img ? im_ : last_img;                                                     \
                rcodes = reinterpret_cast<const uint32_t *>(in)[(size_t)(img0 + ims_) * (C::SS / 4) + (w_ - im_ * (C::SS / 4))]; \
            } else {                        /* f32 planes [P][cin][SS]: <= 3 float4 per thread */                      \
                _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                       \
                    int q_ = tid + j * THREADS;                                                                       \
                    q_ = q_ < C::P * cin * (C::SS / 4) ? q_ : C::P * cin * (C::SS / 4) - 1;                           \
                    const int im_ = q_ / (cin * (C::SS / 4)), r_ = q_ - im_ * (cin * (C::SS / 4));                    \
                    const int ims_ = im_ < last_img ? im_ : last_img;                                                 \
                    rin[j] = *reinterpret_cast<const f32x4 *>(in_wg + (ims_ * cin * C::SS + r_ * 4));                 \
                }                                                                                                     \
            }                                                                                                         \
            /* conv1: W[32][cin][3][3], cin 3 or 4: rows of 27 / 36 floats, scalar loads */                            \
            _Pragma("unroll") for (int j = 0; j < 5; ++j) {                                                           \
                int i_ = tid + j * THREADS;                                                                           \
                i_ = i_ < COUT_WG * cin * 9 ? i_ : COUT_WG * cin * 9 - 1;                                             \
                rw1[j] = wgt_wg[i_];                                                                                  \
            }                                                                                                         \
        } else {                                                                                                      \
            _Pragma("unroll") for (int j = 0; j < C::IN_LD; ++j) {                                                    \
                int q_ = tid + j * THREADS;                                                                           \
                q_ = q_ < C::IN_Q ? q_ : C::IN_Q - 1;                                                                 \
                const int im_ = q_ / (CIC * C::SS / 4), r_ = q_ - im_ * (CIC * C::SS / 4);                            \
                const int ims_ = im_ < last_img ? im_ : last_img;                                                     \
                rin[j] = *reinterpret_cast<const f32x4 *>(in_wg + ((ims_ * cin + (c_) * CIC) * C::SS + r_ * 4));      \
            }                                                                                                         \
            /* W[co][cin][3][3]: the 72 floats of channels 8c..8c+7 are contiguous per co */                           \
            _Pragma("unroll") for (int j = 0; j < C::W_LD; ++j) {                                                     \
                int q_ = tid + j * THREADS;                                                                           \
                q_ = q_ < C::W_Q ? q_ : C::W_Q - 1;                                                                   \
                const int co_ = q_ / 18, j4_ = q_ - co_ * 18;                                                         \
                rw[j] = *reinterpret_cast<const f32x4 *>(wgt_wg + ((co_ * cin + (c_) * CIC) * 9 + j4_ * 4));          \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)

// One float4 of the staged chunk -> LDS: pieces 0..IN_LD-1 are input, the rest weights.  Branch-free (a surplus
// thread's writes go to the dump slot), so a piece can be scheduled between the MFMAs of the k-step it precedes.
#define TRON_STORE_PIECE(j_, b_)                                                                                     \
    do {                                                                                                              \
        if ((j_) < C::IN_LD) {                                                                                        \
            const int q_ = tid + (j_) * THREADS;                                                                      \
            const bool ok_ = ((j_) + 1) * THREADS <= C::IN_Q || q_ < C::IN_Q;                                         \
            const int im_ = q_ / (CIC * C::SS / 4), r_ = q_ - im_ * (CIC * C::SS / 4);                                \
            const int ci_ = r_ / (C::SS / 4), p0_ = (r_ - ci_ * (C::SS / 4)) * 4;                                     \
            const bool live_ = im_ <= last_img;                                                                       \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                           \
                const int p_ = p0_ + e;                                                                               \
                const int y_ = p_ / S, xx_ = p_ - y_ * S;                                                             \
                const int d_ = (b_) * C::IN_BUF + ci_ * C::CI_STRIDE + im_ * C::PLANE + (y_ + 1) * C::SP + (xx_ + 1); \
                lds[ok_ ? d_ : C::DUMP + lane] = live_ ? rin[(j_) < C::IN_LD ? (j_) : 0][e] : 0.0f;                   \
            }                                                                                                         \
        } else if ((j_) < C::IN_LD + C::W_LD) {                                                                       \
            const int jw_ = (j_) < C::IN_LD ? 0 : ((j_) - C::IN_LD < C::W_LD ? (j_) - C::IN_LD : 0);              \
            const int q_ = tid + jw_ * THREADS;                                                                       \
            const bool ok_ = (jw_ + 1) * THREADS <= C::W_Q || q_ < C::W_Q;                                            \
            const int co_ = q_ / 18, j4_ = q_ - co_ * 18;                                                             \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                           \
                const int k_ = j4_ * 4 + e;                                                                           \
                const int ci_ = k_ / 9, tap_ = k_ - ci_ * 9;                                                          \
                const int d_ = 2 * C::IN_BUF + (b_) * C::W_BUF + ((tap_ * 2 + (ci_ >> 2)) * 4 + (ci_ & 3)) * C::WROW + co_; \
                lds[ok_ ? d_ : C::DUMP + lane] = rw[jw_][e];                                                          \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)

#define TRON_STORE_CHUNK(b_)                                                                                         \
    do {                                                                                                              \
        float *ib_ = lds + (b_) * C::IN_BUF, *wb_ = lds + 2 * C::IN_BUF + (b_) * C::W_BUF;                            \
        if (CODES) {                                                                                                  \
            if (in_codes) {                                                                                           \
                if (tid < C::PX / 4) {                                                                                \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                   \
                        const int px_ = tid * 4 + e;                                                                  \
                        const int im_ = px_ / C::SS, p_ = px_ - im_ * C::SS;                                          \
                        const int y_ = p_ / S, xx_ = p_ - y_ * S;                                                     \
                        const int v_ = im_ <= last_img ? (int)(int8_t)(rcodes >> (8 * e)) : 1;                        \
                        float *d_ = ib_ + im_ * C::PLANE + (y_ + 1) * C::SP + (xx_ + 1);                              \
                        d_[0] = (v_ == -1) ? 1.0f : 0.0f;                                   /* util.py:18-19 */      \
                        d_[C::CI_STRIDE] = (v_ == -2) ? 1.0f : (v_ == 10) ? 10.0f : 0.0f;    /* util.py:20-21,26-27 */ \
                        d_[2 * C::CI_STRIDE] = (v_ == -3) ? 1.0f : (v_ == -10) ? 10.0f : 0.0f;                        \
                        if (cin == 4) d_[3 * C::CI_STRIDE] = plane4;                                                  \
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
                            ib_[ci_ * C::CI_STRIDE + im_ * C::PLANE + (y_ + 1) * C::SP + (xx_ + 1)] = im_ <= last_img ? rin[j][e] : 0.0f; \
                        }                                                                                             \
                    }                                                                                                 \
                }                                                                                                     \
            }                                                                                                         \
            /* weights -> [k-step = tap * 2 + channel quad][k-sub = channel & 3][cout] */                              \
            _Pragma("unroll") for (int j = 0; j < 5; ++j) {                                                           \
                const int i_ = tid + j * THREADS;                                                                     \
                if (i_ < COUT_WG * cin * 9) {                                                                         \
                    const int co_ = i_ / (cin * 9), k_ = i_ - co_ * (cin * 9);                                        \
                    const int ci_ = k_ / 9, tap_ = k_ - ci_ * 9;                                                      \
                    wb_[(tap_ * 8 + ci_) * C::WROW + co_] = rw1[j];                                                   \
                }                                                                                                     \
            }                                                                                                         \
        } else {                                                                                                      \
            _Pragma("unroll") for (int j = 0; j < C::IN_LD + C::W_LD; ++j) TRON_STORE_PIECE(j, b_);                   \
        }                                                                                                             \
    } while (0)

    __syncthreads();                  // zero fill done
    TRON_LOAD_CHUNK(0);
    TRON_STORE_CHUNK(0);
    __syncthreads();

    float a0[C::MT], b0[NT], a1[C::MT], b1[NT];
    for (int c = 0; c + 1 < nchunks; ++c) {
        TRON_LOAD_CHUNK(c + 1);                                  // in flight under the MFMAs below
        const float *ib = lds + (c & 1) * C::IN_BUF, *wb = lds + 2 * C::IN_BUF + (c & 1) * C::W_BUF;
        load_frags<C, 0>(ib, wb, a_off, b_off, a0, b0);
        k_range<C, 0, KSTEPS / 2>(acc, ib, wb, a_off, b_off, a0, b0, a1, b1);
        // half way (an odd number of steps done: the live operands sit in a1/b1) the next chunk's loads have landed:
        // one float4 of them goes to LDS in front of each of the remaining k-steps, between that step's MFMAs
        static_assert(C::IN_LD + C::W_LD <= 9, "one staging piece per k-step of the second half");
        const int nb = (c + 1) & 1;
        if constexpr (!CODES) {
#define TRON_STEP(K_, A_, B_, NA_, NB_, PIECE_)                                                      \
    do {                                                                                              \
        if constexpr ((K_) + 1 < KSTEPS) load_frags<C, (K_) + 1>(ib, wb, a_off, b_off, NA_, NB_);     \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        TRON_STORE_PIECE(PIECE_, nb);          /* scheduled between this step's MFMAs */             \
        mfma_step<C>(acc, A_, B_);                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    } while (0)
            TRON_STEP(9, a1, b1, a0, b0, 0);
            TRON_STEP(10, a0, b0, a1, b1, 1);
            TRON_STEP(11, a1, b1, a0, b0, 2);
            TRON_STEP(12, a0, b0, a1, b1, 3);
            TRON_STEP(13, a1, b1, a0, b0, 4);
            TRON_STEP(14, a0, b0, a1, b1, 5);
            TRON_STEP(15, a1, b1, a0, b0, 6);
            TRON_STEP(16, a0, b0, a1, b1, 7);
            TRON_STEP(17, a1, b1, a0, b0, 8);
#undef TRON_STEP
        }
        __syncthreads();
    }
    {                                                            // the last chunk: nothing left to stage
        const int c = nchunks - 1;
        const float *ib = lds + (c & 1) * C::IN_BUF, *wb = lds + 2 * C::IN_BUF + (c & 1) * C::W_BUF;
        load_frags<C, 0>(ib, wb, a_off, b_off, a0, b0);
        k_range<C, 0, KSTEPS>(acc, ib, wb, a_off, b_off, a0, b0, a1, b1);
    }

#undef TRON_LOAD_CHUNK
#undef TRON_STORE_CHUNK
#undef TRON_STORE_PIECE

#ifdef TRON_CONV_STAMPS
    if (tid == 0) { stamps[2] = __builtin_amdgcn_s_memtime(); stamps[3] = __builtin_amdgcn_s_memrealtime(); }
#endif
    // epilogue.  The residual is added to the finished sum (adding it first would round every one of the K
    // partial sums at the residual's magnitude).  All of a lane's residual loads are issued together — the
    // staging and operand registers are dead by now — so the wave pays ONE memory round trip, not one per tile.
    const int pxw_end = (wave + 1) * C::PW < C::PX ? (wave + 1) * C::PW : C::PX;
    int o[C::MT];
    bool live[C::MT];
#pragma unroll
    for (int t = 0; t < C::MT; ++t) {
        const int px = wave * C::PW + 16 * t + 4 * ks;
        const int img = px / C::SS, p = px - img * C::SS;
        live[t] = px < pxw_end && img <= last_img;
        o[t] = (img * cout + li) * C::SS + p;
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
            for (int n = 0; n < NT; ++n) acc[t][n] += r[t][n];
    }
    float *out_wg = out + wg_base;
    float *pre_wg = pre_out ? pre_out + wg_base : nullptr;
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
#ifdef TRON_CONV_STAMPS
    if (tid == 0) { stamps[4] = __builtin_amdgcn_s_memtime(); stamps[5] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

template <int S, bool CODES>
int launch_conv(const void *in, const float *wgt, const float *bias, const float *res, float *out, float *pre_out,
                int64_t B, int cin, int cout, float plane4, int apply_mish, int in_codes, hipStream_t st)
{
    using C = Cfg<S>;
    auto kern = k_conv3x3<S, CODES>;
    static uint64_t prepared = 0;     // hipFuncSetAttribute is per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_NO_DEVICE; }
    if (!(prepared & (1ull << (dev & 63)))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)C::LDS_BYTES) != hipSuccess)
            (void)hipGetLastError();
        prepared |= 1ull << (dev & 63);
    }
    const int64_t groups = (B + C::P - 1) / C::P;
    const int64_t blocks = cout == 64 ? ((groups + 7) / 8) * 16 : groups;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(THREADS), C::LDS_BYTES, st, in, wgt, bias, res, out, pre_out,
                       (int)B, cin, cout, (int)groups, plane4, apply_mish, in_codes);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

}  // namespace

extern "C" int tron_conv3x3_fwd(const void *in, int32_t in_fmt, const float *weight, const float *bias,
                                const float *residual, float *out, float *pre_out, int64_t batch, int32_t cin,
                                int32_t cout, int32_t side, float plane4, int32_t apply_mish, int32_t math, void *workspace,
                                void *out_split, void *stream)
{
    if (!in || !weight || (!out && !out_split) || batch < 0 || cin < 1) return TRON_ERR_BAD_ARG;
    if (in_fmt < TRON_CONV_IN_F32 || in_fmt > TRON_CONV_IN_SPLIT16) return TRON_ERR_BAD_ARG;
    const int in_is_codes = in_fmt == TRON_CONV_IN_CODES;
    // the split-f16 activation image exists only between layers of the split kernel
    const bool needs_f16 = in_fmt == TRON_CONV_IN_SPLIT16 || out_split != nullptr;
    const bool f16x3 = math == TRON_CONV_F16X3 || math == TRON_CONV_F16X3_PRESPLIT;
    if (math == TRON_CONV_F16X3_PRESPLIT && !workspace) return TRON_ERR_BAD_ARG;
    if (needs_f16 && (!f16x3 || !workspace || (side != 12 && side != 26 && side != 34) || cout % 16 != 0 ||
                      (in_fmt == TRON_CONV_IN_SPLIT16 && cin % 16 != 0)))
        return TRON_ERR_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(out_split) & 15u) return TRON_ERR_BAD_ARG;
    if (batch == 0) return TRON_OK;
    if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(weight) | reinterpret_cast<uintptr_t>(out) |
         reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(pre_out)) & 15u)
        return TRON_ERR_BAD_ARG;
    if (cout != 32 && cout != 64) return TRON_ERR_UNSUPPORTED;
    const bool small = cin == 3 || cin == 4;                 // conv1: from codes or from its f32 planes
    if (in_is_codes && !small) return TRON_ERR_BAD_ARG;
    if (small ? cout != 32 : (cin % CIC != 0)) return TRON_ERR_UNSUPPORTED;
    if (batch > (1ll << 24)) return TRON_ERR_UNSUPPORTED;
    if (math != TRON_CONV_F32 && !f16x3) return TRON_ERR_BAD_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (f16x3) {                        // shapes the split kernel has no instantiation for take the f32 kernel
        if (workspace && (reinterpret_cast<uintptr_t>(workspace) & 15u)) return TRON_ERR_BAD_ARG;
        const int rc = tron_conv3x3_f16x3(in, in_fmt, weight, bias, residual, out, pre_out, batch, cin, cout, side,
                                          plane4, apply_mish, workspace, out_split, 0, nullptr, 0, math == TRON_CONV_F16X3_PRESPLIT, st);
        if (rc != TRON_ERR_UNSUPPORTED || needs_f16) return rc;
    }
#define TRON_CONV_CASE(S_)                                                                                                \
    if (side == S_)                                                                                                        \
        return small ? launch_conv<S_, true>(in, weight, bias, residual, out, pre_out, batch, cin, cout, plane4,          \
                                             apply_mish, in_is_codes ? 1 : 0, st)                                          \
                     : launch_conv<S_, false>(in, weight, bias, residual, out, pre_out, batch, cin, cout, plane4,         \
                                              apply_mish, 0, st);
    TRON_CONV_CASE(12)
    TRON_CONV_CASE(26)
#undef TRON_CONV_CASE
    return TRON_ERR_UNSUPPORTED;
}

// the data gradient of the same layer: the same convolution on the gradient with the weight's channel axes swapped and
// its taps reversed; only the split-f16 kernel builds that weight image (k_split_weights)
extern "C" int tron_conv3x3_dgrad(const float *grad_pre, const float *weight, const float *grad_absmax, int32_t n_absmax,
                                  float *grad_in, int64_t batch, int32_t cin, int32_t cout, int32_t side, void *workspace,
                                  void *stream)
{
    if (!grad_pre || !weight || !grad_in || !workspace || batch < 0 || (grad_absmax && n_absmax < 1)) return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(grad_pre) | reinterpret_cast<uintptr_t>(weight) | reinterpret_cast<uintptr_t>(grad_in) |
         reinterpret_cast<uintptr_t>(workspace)) & 15u)
        return TRON_ERR_BAD_ARG;
    if (batch == 0) return TRON_OK;
    if ((cin != 32 && cin != 64) || cout % CIC != 0 || cout < CIC || cout > 64 || batch > (1ll << 24)) return TRON_ERR_UNSUPPORTED;
    return tron_conv3x3_f16x3(grad_pre, TRON_CONV_IN_F32, weight, nullptr, nullptr, grad_in, nullptr, batch, cout, cin, side, 0.0f,
                              0, workspace, nullptr, 1, grad_absmax, n_absmax, 0, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int64_t tron_conv3x3_workspace(int32_t cin, int32_t cout) { return tron_conv3x3_f16x3_workspace(cin, cout); }
