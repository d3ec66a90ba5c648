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
typedef __attribute__((address_space(1))) const f32x4 gvec4;      // global memory, explicitly
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int SIDE = 12, HW = SIDE * SIDE, ROWP = 14, IMG_K = 192, SLABS_IMG = IMG_K / 32, MARGIN = 16;
constexpr int THREADS = 256, GRID_MAX = 256, STAGE2 = 8, ABSMAX_BLOCKS = 1024;
constexpr float IN_SCALE = 1.0f / 64.0f, LO_SCALE = 2048.0f;
#ifndef TRON_WG_ABLATE       // diagnostic builds: 1 no MFMA slabs, 2 no staging inside the loop, 3 no global loads
#define TRON_WG_ABLATE 0
#endif

// row pitch for ds_read_b128 by (row = lane % 16, 16-byte column = lane / 16): 32 mod 64 bytes is conflict-free with the
// instruction's lane groups (MI355X guide, LDS table); any other multiple of 16 is two-way
constexpr int pitch_for(int bytes) { return (bytes + 31) / 64 * 64 + 32; }

constexpr int COT = 2;                                         // 16-channel output tiles per wave: 32 channels per workgroup

template <int CIT, int IMGS>
struct Cfg {
    static constexpr int KSPLIT = 4 / CIT;                     // waves sharing an input tile split the slabs
    static constexpr int SL = SLABS_IMG * IMGS / KSPLIT;       // slabs per wave per round
    static constexpr int GP_PITCH = pitch_for(IMGS * IMG_K * 2), IN_PITCH = pitch_for((IMGS * IMG_K + 2 * MARGIN) * 2);
    static constexpr int GP_HALF = COT * 16 * GP_PITCH, IN_HALF = CIT * 16 * IN_PITCH;
    static constexpr int LDS = 2 * GP_HALF + 2 * IN_HALF;      // operands; a 64-byte dump slot per thread follows
    static constexpr int LDS_ALL = LDS + THREADS * 64;
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

#ifdef TRON_WG_STAMPS        // diagnostic build only (scripts/wgrad_stamps.py): per-workgroup clocks, read back by tron_wgrad_stamps
__device__ unsigned long long g_stamps[GRID_MAX * 8];
#define TRON_WG_STAMP(i_) if (tid == 0) { g_stamps[blockIdx.x * 8 + (i_)] = __builtin_amdgcn_s_memtime(); }
#define TRON_WG_STAMP_RT(i_) if (tid == 0) { g_stamps[blockIdx.x * 8 + (i_)] = __builtin_amdgcn_s_memrealtime(); }
#else
#define TRON_WG_STAMP(i_)
#define TRON_WG_STAMP_RT(i_)
#endif

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
    TRON_WG_STAMP(0)
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

    // Staging items: one 12-float row of one channel of one image (gradient rows first, then input rows); thread i owns
    // items i, i + THREADS, ...  An item's source (round 0) and LDS destination are the same in every round and are
    // computed once; its half offset, image stride, scale and image index follow from "gradient or input?", a compare of
    // tid with a constant per k, recomputed at each use instead of held in registers across the MFMA loop.  Surplus
    // items read the tensor's first row and write a per-thread dump slot; images past the batch are multiplied by a
    // zero scale.  No branches: the pieces can be scheduled between MFMAs.
    const float *it_src[C::NIT];
    int it_dst[C::NIT];
#define TRON_WG_IS_GP(k_) (tid + (k_) * THREADS < C::GP_ITEMS)
#define TRON_WG_REAL(k_) (tid + (k_) * THREADS < C::GP_ITEMS + in_items)
#define TRON_WG_J(k_) (IMGS == 1 ? 0 : TRON_WG_IS_GP(k_) ? (tid + (k_) * THREADS) / (COUT * SIDE) : (tid + (k_) * THREADS - C::GP_ITEMS) / (cin * SIDE))
#pragma unroll
    for (int k = 0; k < C::NIT; ++k) {
        const int it = tid + k * THREADS;
        it_src[k] = gp;
        it_dst[k] = C::LDS + tid * 64 - (TRON_WG_IS_GP(k) ? 0 : 2 * C::GP_HALF);   // dump: hi at +0, lo at +32 (TRON_WG_STORE)
        if (it < C::GP_ITEMS) {
            const int j = it / (COUT * SIDE), rem = it - j * (COUT * SIDE), c = rem / SIDE, y = rem - c * SIDE;
            it_src[k] = gp + ((size_t)(j * cout + co0) * SIDE + rem) * SIDE;
            it_dst[k] = c * C::GP_PITCH + (j * IMG_K + y * ROWP) * 2;              // relative to gp_h
        } else if (it - C::GP_ITEMS < in_items) {
            const int i2 = it - C::GP_ITEMS, j = i2 / (cin * SIDE), rem = i2 - j * (cin * SIDE);
            const int c = rem / SIDE, y = rem - c * SIDE;
            it_src[k] = in + ((size_t)j * cin * SIDE + rem) * SIDE;
            it_dst[k] = c * C::IN_PITCH + (MARGIN + j * IMG_K + y * ROWP) * 2;     // relative to in_h
        }
    }
    f32x4 pf[C::NIT][3];                                                 // raw rows in flight
    uint32_t cvh[C::NIT][6], cvl[C::NIT][6];                             // the same rows as packed f16 halves
    // (an explicit global-memory pointer: as a generic one these are flat_loads, whose lgkmcnt entangles the LDS waits of
    // the MFMA loop with the HBM latency)
#define TRON_WG_LOAD(round_)                                                                                             \
    _Pragma("unroll") for (int k_ = 0; k_ < C::NIT; ++k_) {                                                              \
        const bool ok_ = TRON_WG_REAL(k_) && (int64_t)(round_) * IMGS + TRON_WG_J(k_) < batch;                           \
        const int stride_ = IMGS * HW * (TRON_WG_IS_GP(k_) ? cout : cin);                                                \
        const gvec4 *src_ = (const gvec4 *)(uintptr_t)(ok_ ? it_src[k_] + (int64_t)(round_) * stride_ : gp);             \
        pf[k_][0] = src_[0];                                                                                             \
        pf[k_][1] = src_[1];                                                                                             \
        pf[k_][2] = src_[2];                                                                                             \
    }
    // one third of an item: 4 floats -> 2 + 2 packed dwords; piece p_ = 3 * item + third
#define TRON_WG_CONVERT_PIECE(round_, p_)                                                                                \
    {                                                                                                                    \
        constexpr int k_ = (p_) / 3, q_ = (p_) % 3;                                                                      \
        const float sc_ = ((int64_t)(round_) * IMGS + TRON_WG_J(k_) < batch) ? (TRON_WG_IS_GP(k_) ? gscale : IN_SCALE) : 0.0f; \
        const f32x4 v_ = pf[k_][q_] * sc_;                                                                               \
        const f16 h0_ = (f16)v_[0], h1_ = (f16)v_[1], h2_ = (f16)v_[2], h3_ = (f16)v_[3];                                \
        const f16 l0_ = (f16)((v_[0] - (float)h0_) * LO_SCALE), l1_ = (f16)((v_[1] - (float)h1_) * LO_SCALE);            \
        const f16 l2_ = (f16)((v_[2] - (float)h2_) * LO_SCALE), l3_ = (f16)((v_[3] - (float)h3_) * LO_SCALE);            \
        cvh[k_][2 * q_] = pack2(h0_, h1_);                                                                               \
        cvh[k_][2 * q_ + 1] = pack2(h2_, h3_);                                                                           \
        cvl[k_][2 * q_] = pack2(l0_, l1_);                                                                               \
        cvl[k_][2 * q_ + 1] = pack2(l2_, l3_);                                                                           \
    }
#define TRON_WG_STORE()                                                                                                  \
    _Pragma("unroll") for (int k_ = 0; k_ < C::NIT; ++k_) {                                                              \
        const bool gp_ = TRON_WG_IS_GP(k_);                                                                              \
        uint32_t *dh_ = reinterpret_cast<uint32_t *>((gp_ ? gp_h : in_h) + it_dst[k_]);                                  \
        uint32_t *dl_ = reinterpret_cast<uint32_t *>((gp_ ? gp_h : in_h) + it_dst[k_] +                                  \
                                                     (!TRON_WG_REAL(k_) ? 32 : gp_ ? C::GP_HALF : C::IN_HALF));          \
        _Pragma("unroll") for (int q_ = 0; q_ < 6; ++q_) {                                                               \
            dh_[q_] = cvh[k_][q_];                                                                                       \
            dl_[q_] = cvl[k_][q_];                                                                                       \
        }                                                                                                                \
    }

    // ---- the MFMA loop.  One wave per SIMD: nobody else covers an LDS wait, so every operand is read one tap GROUP
    // (three taps = 18 MFMAs) before its first use, into registers whose previous contents are dead by then, and
    // sched_barriers keep the compiler from sinking the reads down to their uses.  A wave's input window slides along
    // one contiguous strip: slab s covers elements [32 s - 16, 32 s + 24) (+ 8 g) as blocks 0..4 of 8 elements, block 0
    // of slab s + 1 is block 4 of slab s.  Group 0 (ky = 0: window indices 1, 2, 3) uses blocks 0-1, group 1 (15, 16, 17)
    // blocks 1-3, group 2 (29, 30, 31) blocks 3-4.
#define TRON_WG_BLOCK(k_, s_)                                                                                            \
    {                                                                                                                    \
        const u32x4 vh_ = *reinterpret_cast<const u32x4 *>(b_h + (s_) * 64 + (k_) * 16);                                 \
        const u32x4 vl_ = *reinterpret_cast<const u32x4 *>(b_l + (s_) * 64 + (k_) * 16);                                 \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                                  \
            Dh[4 * (k_) + c] = vh_[c];                                                                                   \
            Dl[4 * (k_) + c] = vl_[c];                                                                                   \
        }                                                                                                                \
    }
#define TRON_WG_A(b_, s_)                                                                                                \
    _Pragma("unroll") for (int t = 0; t < COT; ++t) {                                                                    \
        ah[b_][t] = *reinterpret_cast<const f16x8 *>(a_h + t * 16 * C::GP_PITCH + (s_) * 64);                            \
        al[b_][t] = *reinterpret_cast<const f16x8 *>(a_l + t * 16 * C::GP_PITCH + (s_) * 64);                            \
    }
    // A tap group: taps tap0_ .. tap0_ + 2 (tap (ky, kx) is shift 14 (ky - 1) + (kx - 1) = window index 16 + shift), operands
    // fh / fl cut from the window beforehand.  The three products of a tap are issued group-wide in three phases with
    // scheduling barriers between them, hi*lo first and lo*hi last: both add to the same accumulator, a dependent MFMA
    // issues 48 cycles after its producer (an independent one 16), and left alone the compiler puts them back to back.
    // Phase 3 also cuts the NEXT group's operands (nidx0_: its first window index) out of the blocks read during phase 1.
    // tp_: the group's first tap-step of the round; the staging pieces that ride in the taps' shadow are counted from it.
#define TRON_WG_PIECE_AT(tp_)                                                                                            \
    {                                                                                                                    \
        constexpr int piece_ = (tp_) - (C::SL * 9 - C::NIT * 3);                                                         \
        if constexpr (piece_ >= 0 && TRON_WG_ABLATE != 2) TRON_WG_CONVERT_PIECE(round + wgs, piece_ >= 0 ? piece_ : 0)   \
    }
#define TRON_WG_PHASE_SCHED(reads_)                                                                                      \
    __builtin_amdgcn_sched_group_barrier(0x100, reads_, 0);                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 3 * COT; ++i_) {                                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                                               \
    }                                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);
#define TRON_WG_CUT(idx0_)                                                                                               \
    fh[0] = window_frag<idx0_>(Dh); fh[1] = window_frag<(idx0_) + 1>(Dh); fh[2] = window_frag<(idx0_) + 2>(Dh);          \
    fl[0] = window_frag<idx0_>(Dl); fl[1] = window_frag<(idx0_) + 1>(Dl); fl[2] = window_frag<(idx0_) + 2>(Dl);
#define TRON_WG_GROUP(b_, tap0_, tp_, reads_, nidx0_, slide_)                                                            \
    {                                                                                                                    \
        TRON_WG_PIECE_AT(tp_)                                                                                            \
        _Pragma("unroll") for (int t = 0; t < COT; ++t)                                                                  \
            _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                                \
                acc1[t][(tap0_) + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b_][t], fl[j], acc1[t][(tap0_) + j], 0, 0, 0); \
        TRON_WG_PHASE_SCHED(reads_)                                                                                      \
        TRON_WG_PIECE_AT((tp_) + 1)                                                                                      \
        _Pragma("unroll") for (int t = 0; t < COT; ++t)                                                                  \
            _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                                \
                acc0[t][(tap0_) + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b_][t], fh[j], acc0[t][(tap0_) + j], 0, 0, 0); \
        TRON_WG_PHASE_SCHED(0)                                                                                           \
        TRON_WG_PIECE_AT((tp_) + 2)                                                                                      \
        _Pragma("unroll") for (int t = 0; t < COT; ++t)                                                                  \
            _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                                \
                acc1[t][(tap0_) + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[b_][t], fh[j], acc1[t][(tap0_) + j], 0, 0, 0); \
        if constexpr ((nidx0_) > 0) {                                                                                    \
            if constexpr (slide_) {                           /* the window slides: block 4 is the next slab's block 0 */ \
                _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                          \
                    Dh[c] = Dh[16 + c];                                                                                  \
                    Dl[c] = Dl[16 + c];                                                                                  \
                }                                                                                                        \
            }                                                                                                            \
            TRON_WG_CUT(nidx0_)                                                                                          \
        }                                                                                                                \
        TRON_WG_PHASE_SCHED(0)                                                                                           \
    }
#define TRON_WG_SLAB(sl_)                                                                                                \
    if constexpr ((sl_) < C::SL && TRON_WG_ABLATE != 1) {                                                                \
        constexpr int b_ = (sl_) & 1;                                                                                    \
        constexpr bool more_ = (sl_) + 1 < C::SL;                                                                        \
        TRON_WG_BLOCK(2, s0 + (sl_)) TRON_WG_BLOCK(3, s0 + (sl_))                                                        \
        TRON_WG_GROUP(b_, 0, (sl_) * 9, 4, 15, false)                                                                    \
        TRON_WG_BLOCK(4, s0 + (sl_))                                                                                     \
        TRON_WG_GROUP(b_, 3, (sl_) * 9 + 3, 2, 29, false)                                                                \
        if constexpr (more_) {                                                                                           \
            TRON_WG_A(b_ ^ 1, s0 + (sl_) + 1)                                                                            \
            TRON_WG_BLOCK(1, s0 + (sl_) + 1)                  /* block 1 is dead after group 1 */                        \
        }                                                                                                                \
        TRON_WG_GROUP(b_, 6, (sl_) * 9 + 6, more_ ? 2 * COT + 2 : 0, more_ ? 1 : 0, true)                                \
    }
    static_assert(C::SL * 9 >= C::NIT * 3 && C::SL <= 6 && C::NIT <= 5, "one staging piece per tap");

    const unsigned char *a_h = gp_h + li * C::GP_PITCH + g * 16, *a_l = a_h + C::GP_HALF;
    const unsigned char *b_h = in_h + (cit * 16 + li) * C::IN_PITCH + g * 16, *b_l = b_h + C::IN_HALF;
    const int s0 = ks * C::SL;
    f16x8 ah[2][COT], al[2][COT];
    uint32_t Dh[20], Dl[20];                                             // the window: blocks 0..4, four dwords each
    f16x8 fh[3], fl[3];                                                  // the current tap group's operands

    int round = wg;
    TRON_WG_LOAD(round)
    {
        TRON_WG_CONVERT_PIECE(round, 0) TRON_WG_CONVERT_PIECE(round, 1) TRON_WG_CONVERT_PIECE(round, 2)
        if constexpr (C::NIT > 1) { TRON_WG_CONVERT_PIECE(round, 3) TRON_WG_CONVERT_PIECE(round, 4) TRON_WG_CONVERT_PIECE(round, 5) }
        if constexpr (C::NIT > 2) { TRON_WG_CONVERT_PIECE(round, 6) TRON_WG_CONVERT_PIECE(round, 7) TRON_WG_CONVERT_PIECE(round, 8) }
        if constexpr (C::NIT > 3) { TRON_WG_CONVERT_PIECE(round, 9) TRON_WG_CONVERT_PIECE(round, 10) TRON_WG_CONVERT_PIECE(round, 11) }
        if constexpr (C::NIT > 4) { TRON_WG_CONVERT_PIECE(round, 12) TRON_WG_CONVERT_PIECE(round, 13) TRON_WG_CONVERT_PIECE(round, 14) }
    }
    TRON_WG_STAMP(1) TRON_WG_STAMP_RT(2)
#ifdef TRON_WG_STAMPS
    unsigned long long t_stage = 0, t_mark = 0;
#endif
    for (; round < nrounds; round += wgs) {
#ifdef TRON_WG_STAMPS
        t_mark = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();                                                 // the previous round's operands are consumed
        if (TRON_WG_ABLATE != 2) { TRON_WG_STORE() }
        __syncthreads();
#ifdef TRON_WG_STAMPS
        t_stage += __builtin_amdgcn_s_memtime() - t_mark;
#endif
        if (TRON_WG_ABLATE != 2 && TRON_WG_ABLATE != 3) { TRON_WG_LOAD(round + wgs) }   // lands under the first taps' MFMAs
        if (TRON_WG_ABLATE != 1) {
            TRON_WG_A(0, s0) TRON_WG_BLOCK(0, s0) TRON_WG_BLOCK(1, s0)
            TRON_WG_CUT(1)
            __builtin_amdgcn_sched_barrier(0);
        }
        TRON_WG_SLAB(0) TRON_WG_SLAB(1) TRON_WG_SLAB(2) TRON_WG_SLAB(3) TRON_WG_SLAB(4) TRON_WG_SLAB(5)
    }
    TRON_WG_STAMP(3) TRON_WG_STAMP_RT(4)
#ifdef TRON_WG_STAMPS
    if (tid == 0) g_stamps[blockIdx.x * 8 + 5] = t_stage;
#endif
#undef TRON_WG_IS_GP
#undef TRON_WG_REAL
#undef TRON_WG_J
#undef TRON_WG_LOAD
#undef TRON_WG_CONVERT_PIECE
#undef TRON_WG_STORE
#undef TRON_WG_BLOCK
#undef TRON_WG_A
#undef TRON_WG_PIECE_AT
#undef TRON_WG_GROUP
#undef TRON_WG_PHASE_SCHED
#undef TRON_WG_CUT
#undef TRON_WG_SLAB

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

// sums of the per-workgroup partial gradients in a fixed order, one launch: a workgroup = 32 outputs x the STAGE2 = 8 interleaved
// subsets of the partials (subset y = partials y, y + 8, ...); thread (output, subset) adds its subset on eight independent chains
// (a single chain is nparts / 8 dependent load -> add steps: 33 us for conv1's 1 024 partials), joined in a fixed order; the eight
// subset sums meet in LDS and are joined in the same fixed order.  (Two launches before: subsets, then their sum.)
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float *__restrict__ partial, int nparts, int W, float *__restrict__ out)
{
    __shared__ float sub[STAGE2][32];
    const int il = threadIdx.x & 31, y = threadIdx.x >> 5, i = blockIdx.x * 32 + il;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < W) {
        int p = y;
        for (; p + 7 * STAGE2 < nparts; p += 8 * STAGE2) {
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += partial[(size_t)(p + k * STAGE2) * W + i];
        }
        for (int k = 0; p < nparts; p += STAGE2, ++k) s[k] += partial[(size_t)p * W + i];
    }
    sub[y][il] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (y == 0 && i < W)
        out[i] = ((sub[0][il] + sub[1][il]) + (sub[2][il] + sub[3][il])) + ((sub[4][il] + sub[5][il]) + (sub[6][il] + sub[7][il]));
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
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_ALL) != hipSuccess)
            (void)hipGetLastError();
        prepared |= 1ull << (dev & 63);
    }
    const int W = cout * cin * 9, nhalves = cout / (COT * 16);
    const Plan p = plan(cin, cout);
    const int nrounds = (int)((batch + IMGS - 1) / IMGS);
    const int wgs = nrounds < GRID_MAX / nhalves ? nrounds : GRID_MAX / nhalves;   // workgroups per output half
    float *partial = reinterpret_cast<float *>(ws + p.partial);
    hipLaunchKernelGGL(kern, dim3(wgs * nhalves), dim3(THREADS), C::LDS_ALL, st, in, gp, absmax, n_absmax, partial, (int)batch, cin,
                       cout, nrounds);
    const int nparts = wgs * C::KSPLIT;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((W + 31) / 32), dim3(256), 0, st, partial, nparts, W, grad_w);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

// conv1's weight gradient on boards whose image does not fit the strip layout (26x26, 34x34: cin = 3 or 4 planes, 32 output
// channels — 4.8 GFLOP at 4 096 x 26x26, against 354 MB of gradient to read): plain f32 FMAs, one pass over the gradient.
// A workgroup walks its images; the image's input planes sit in LDS with a zero halo (9 - 21 KB).  Thread (co, row group):
// for each of its rows, the gradient row of its channel in registers (S floats) x the CIN x 3 haloed input rows around it,
// read from LDS as broadcasts (the 32 lanes of a channel group share every address), 9 CIN accumulators per thread.  The
// eight row groups of a channel meet in LDS; per-workgroup partial sums, k_wgrad_reduce adds them in a fixed order.
template <int S, int CIN>
__global__ __launch_bounds__(256) void k_wgrad_small(const float *__restrict__ in, const float *__restrict__ gp, float *__restrict__ partial,
                                                     int batch)
{
    constexpr int HP = S + 2, TAPS = CIN * 9, NQ = HP / 4;               // HP % 4 == 0: a haloed row is NQ aligned 16-byte reads
    static_assert(HP % 4 == 0 && S % 2 == 0, "row reads");
    __shared__ __attribute__((aligned(16))) float pl[CIN * HP * HP];
    __shared__ float red[8 * 32 * TAPS];
    const int tid = threadIdx.x, co = tid & 31, rg = tid >> 5;
    for (int i = tid; i < CIN * HP * HP; i += 256) pl[i] = 0.0f;
    float acc[TAPS];
#pragma unroll
    for (int k = 0; k < TAPS; ++k) acc[k] = 0.0f;
    for (int b = blockIdx.x; b < batch; b += gridDim.x) {
        __syncthreads();                                                 // (the halo's zeros / the previous image's readers)
        const float *img = in + (size_t)b * CIN * S * S;
        for (int i = tid; i < CIN * S * S; i += 256) {
            const int c = i / (S * S), r = i - c * (S * S), y = r / S, x = r - y * S;
            pl[(c * HP + y + 1) * HP + x + 1] = img[i];
        }
        __syncthreads();
        const float *g = gp + ((size_t)b * 32 + co) * S * S;
        for (int y = rg; y < S; y += 8) {
            float gr[S];
#pragma unroll
            for (int q = 0; q < S / 2; ++q) {
                const float2 v = *reinterpret_cast<const float2 *>(g + y * S + 2 * q);
                gr[2 * q] = v.x;
                gr[2 * q + 1] = v.y;
            }
#pragma unroll
            for (int c = 0; c < CIN; ++c)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    float row[HP];                                       // haloed input row y + ky - 1: row[x + kx] is the pixel under tap kx
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(&pl[(c * HP + y + ky) * HP + 4 * q]);
                        row[4 * q] = v[0]; row[4 * q + 1] = v[1]; row[4 * q + 2] = v[2]; row[4 * q + 3] = v[3];
                    }
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        float a = acc[(c * 3 + ky) * 3 + kx];
#pragma unroll
                        for (int x = 0; x < S; ++x) a = __fmaf_rn(gr[x], row[x + kx], a);
                        acc[(c * 3 + ky) * 3 + kx] = a;
                    }
                }
        }
    }
#pragma unroll
    for (int k = 0; k < TAPS; ++k) red[(rg * 32 + co) * TAPS + k] = acc[k];
    __syncthreads();
    float *out = partial + (size_t)blockIdx.x * 32 * TAPS;               // [co][ci][tap]: the weight's own order
    for (int i = tid; i < 32 * TAPS; i += 256) {
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s0 += red[(2 * r) * 32 * TAPS + i];
            s1 += red[(2 * r + 1) * 32 * TAPS + i];
        }
        out[i] = s0 + s1;
    }
}

template <int S, int CIN>
int launch_small(const float *in, const float *gp, float *grad_w, int64_t batch, unsigned char *ws, hipStream_t st)
{
    const Plan p = plan(CIN, 32);
    constexpr int W = 32 * CIN * 9;
    float *partial = reinterpret_cast<float *>(ws + p.partial);
    const int grid = (int)(batch < 3 * GRID_MAX ? batch : 3 * GRID_MAX);   // three resident workgroups per CU (138 - 170 VGPRs): one round.  (plan(): room for 4 GRID_MAX partial rows at cin <= 16)
    hipLaunchKernelGGL((k_wgrad_small<S, CIN>), dim3((unsigned)grid), dim3(256), 0, st, in, gp, partial, (int)batch);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((W + 31) / 32), dim3(256), 0, st, partial, grid, W, grad_w);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

}  // namespace

#ifdef TRON_WG_STAMPS
extern "C" int tron_wgrad_stamps(unsigned long long *host_dst)
{
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * GRID_MAX * 8) == hipSuccess ? 0 : -1;
}
#endif

// csrc/tron_conv_wgrad_rows.hip: the row-streaming kernel for images that do not fit one strip (26x26)
int tron_wgrad_rows(const float *in, const float *gp, const float *absmax, int n_absmax, float *partial, int64_t batch, int cin,
                    int cout, int side, int grid_max, int *nparts, hipStream_t st);

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
    const bool rows = (side == 26 || side == 34) && (cin == 32 || cin == 64) && (cout == 32 || cout == 64) && !(cin == 64 && cout == 32);
    const bool small_big = small && cout == 32 && (side == 26 || side == 34);   // conv1 at 24x24 / 32x32 boards: f32 FMAs
    if (!(side == SIDE || rows || small_big) || !(small || cin == 32 || cin == 64) || !(cout == 32 || cout == 64) || batch > (1ll << 24))
        return TRON_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (batch == 0) return hipMemsetAsync(grad_weight, 0, (size_t)cout * cin * 9 * 4, st) == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
    unsigned char *ws = reinterpret_cast<unsigned char *>(workspace);
    if (small_big) {                                                     // (no scale: the products stay in f32)
        if (side == 26) return cin == 3 ? launch_small<26, 3>(in, grad_pre, grad_weight, batch, ws, st) : launch_small<26, 4>(in, grad_pre, grad_weight, batch, ws, st);
        return cin == 3 ? launch_small<34, 3>(in, grad_pre, grad_weight, batch, ws, st) : launch_small<34, 4>(in, grad_pre, grad_weight, batch, ws, st);
    }
    if (!grad_absmax) {
        const size_t n4 = (size_t)batch * cout * side * side / 4;
        const unsigned blocks = (unsigned)((n4 + 255) / 256 < ABSMAX_BLOCKS ? (n4 + 255) / 256 : ABSMAX_BLOCKS);
        hipLaunchKernelGGL(k_absmax, dim3(blocks), dim3(256), 0, st, grad_pre, n4, reinterpret_cast<float *>(ws));
        grad_absmax = reinterpret_cast<const float *>(ws);
        n_absmax = (int32_t)blocks;
    }
    if (rows) {                                                          // 24x24 / 32x32 boards: rows streamed through LDS
        const Plan p = plan(cin, cout);
        const int W = cout * cin * 9;
        float *partial = reinterpret_cast<float *>(ws + p.partial);
        int nparts = 0;
        const int rc = tron_wgrad_rows(in, grad_pre, grad_absmax, n_absmax, partial, batch, cin, cout, side, GRID_MAX, &nparts, st);
        if (rc != TRON_OK) return rc;
        hipLaunchKernelGGL(k_wgrad_reduce, dim3((W + 31) / 32), dim3(256), 0, st, partial, nparts, W, grad_weight);
        return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
    }
    if (small) return launch<1, 2>(in, grad_pre, grad_absmax, n_absmax, grad_weight, batch, cin, cout, ws, st);
    if (cin == 32) return launch<2, 1>(in, grad_pre, grad_absmax, n_absmax, grad_weight, batch, cin, cout, ws, st);
    return launch<4, 1>(in, grad_pre, grad_absmax, n_absmax, grad_weight, batch, cin, cout, ws, st);
}
