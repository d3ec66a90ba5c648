// tron_conv_f16.hip — the same fused 3x3 convolution as tron_conv.hip (Net/DQNNet.py:10-17,33-50: conv + bias +
// residual + mish, NCHW f32 in and out, conv1 straight from the int8 observation codes), computed on the f16
// matrix cores at fp32-grade accuracy by splitting every operand in two halves.
//
// Why.  gfx950 has no reduced-precision fast path for f32 inputs: v_mfma_f32_16x16x4_f32 runs at the f32 vector
// rate, 1/16 of the f16 rate.  An f32 value v is hi + lo * 2^-11 with hi = f16(v) and lo = f16((v - hi) * 2^11)
// to within 2^-22 |v|, and a product of two f16 values is exact in f32, so
//     a * b  =  ah*bh  +  (ah*bl + al*bh) * 2^-11                      (+ al*bl * 2^-22, dropped: 2^-22 relative)
// is three v_mfma_f32_16x16x32_f16 (32 k-values in 16 cycles each) where the f32 path needs eight 16x16x4 MFMAs of
// 32 cycles: 5.3x less matrix-pipe time, f32 accumulation throughout; measured error against float64 is BELOW the
// f32 MFMA kernel's (the hi*hi products are exact and a 32-deep slab is summed in one instruction).
// Activations are scaled by 2^-6 before the split (|x| up to 4e6 stays inside f16; tiny values lose nothing
// because lo picks up what hi's subnormal rounding drops); the scale is undone, exactly, in the epilogue.
//
// Mapping (GEMM view: M = pixels, N = output channels, K = taps x input channels).
//   * A workgroup of 8 waves owns P whole images (2 at 12x12: 288 pixels = 18 M tiles) and ALL output channels (32
//     or 64): 4 waves along M x 2 along N; a wave holds up to MT x NT tiles of 16 px x 16 channels, two f32
//     accumulators each (hi*hi and the cross terms).  The 18 tiles are dealt 5/5/4/4 and the two N halves take the
//     M slots in opposite order, so the two waves of a SIMD (waves w and w + 4) always carry 9 tiles together.
//   * K is walked in chunks of 16 input channels x 9 taps (padded to 10: the tenth tap has zero weights), five
//     32-deep slabs per chunk: lane group g of an MFMA covers (tap 2s + g/2, channel octet g%2).  LDS holds the
//     chunk's input planes channel-innermost with their zero halo, [pixel][16 ci] f16 twice (hi, lo), and the
//     weights as [tap][cout][16 ci] f16 twice, so every operand fragment — 8 consecutive k of one row — is ONE
//     ds_read_b128; 32-byte rows keep a tile's 16 rows on 16 different bank quads.
//   * Weights are split ONCE per call by a small kernel into exactly that LDS image (`workspace`), so staging them
//     is a linear 16-byte copy.  Activations are split while they are staged.  Both are double-buffered in LDS
//     (129 KB with 64 channels): the next chunk is loaded from memory at the start of a chunk and written to the
//     other buffers one piece per tile-step, in the shadow of the chunk's MFMAs; one barrier per chunk.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tron_hip.h"
#include "tron_conv.hpp"

#ifndef TRON_F16_ABLATE      // diagnostic builds only (wrong results): 1 = no staging of the next chunk, 2 = A fragments not re-read, 5 = one split-image store per tile instead of eight, 6 = no weight copies, 7 = no input loads
#define TRON_F16_ABLATE 0
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int CIC = 16;           // input channels per K chunk
constexpr int TAPS_PAD = 10;      // 9 taps + one with zero weights: 5 slabs of 2 taps x 16 channels
constexpr int SLABS = 5;
constexpr int PITCH = 32;         // bytes per LDS row: 16 f16
constexpr int NWM = 4, NWN = 2;   // waves along M and N
constexpr int THREADS = 64 * NWM * NWN;
constexpr float ACT_SCALE = 1.0f / 64.0f, ACT_UNSCALE = 64.0f, LO_SCALE = 2048.0f, LO_UNSCALE = 1.0f / 2048.0f;

// A workgroup's region: P whole images (small boards) or one row band of one image (NB bands; larger boards).  A band
// starts on an even row, so its first pixel's offset (row * S, S even) is a multiple of 4: epilogue float4s stay aligned.
#ifndef TRON_F16_PRIO       // 0: leave the issue arbitration to wave age (A/B switch for measurements)
#define TRON_F16_PRIO 1
#endif

template <int S_>
struct Cfg {
    static constexpr int S = S_;
    // Row pitch of the padded LDS planes, in pixels (32 bytes each): column 0 and columns S+1 .. SP-1 are zero halo.  A
    // 16-pixel M tile spans two or three image rows, and a ds_read_b128 lane group holds eight lanes of one 16-byte
    // column: their pixels must differ mod 8 to sit on different banks, so every jump a tile makes (row to row: SP - S,
    // image to image: 3 SP - S at 12x12) has to be a multiple of 8 pixels.  S + 2 (the minimal halo) made most fragment
    // reads two-way bank conflicts.
    static constexpr int SP = S + 8;
    static constexpr int SS = S * S;
    static constexpr int P = (S * S <= 144) ? (288 / (S * S)) : 1;     // images per workgroup
    static constexpr int NB = (S * S <= 400) ? 1 : (S <= 26 ? 2 : 4);   // row bands per image (34x34: 8 + 8 + 8 + 10 rows)
    static constexpr int SPLIT = NB == 1 ? S : ((S / NB) & ~1);         // band b = rows [b SPLIT, (b + 1) SPLIT), the last one up to S
    static constexpr int LAST = S - (NB - 1) * SPLIT;                   // rows of the last band
    static constexpr int ROWS_MAX = LAST > SPLIT ? LAST : SPLIT;
    static constexpr int PLANE = (ROWS_MAX + 2) * SP;                   // padded plane of one image's band in LDS, pixels
    static constexpr int TILES_MAX = (P * ROWS_MAX * S + 15) / 16;      // 16-pixel M tiles of the larger band
    static constexpr int TILES_MIN = (P * (LAST < SPLIT ? LAST : SPLIT) * S + 15) / 16;
    static constexpr int MT = (TILES_MAX + NWM - 1) / NWM;              // most tiles a wave gets
    static constexpr int MT_MIN = TILES_MIN / NWM;                      // fewest: staging rides on these tile-steps
    static constexpr int IN_HALF = P * PLANE * PITCH;                   // hi (or lo) image of one input chunk
    static constexpr int IN_ITEMS = P * (ROWS_MAX + 2) * S * (CIC / 4); // (pixel incl. halo rows, channel quad) items, at most
    static constexpr int IN_LD = (IN_ITEMS + THREADS - 1) / THREADS;
    static_assert(S % 2 == 0, "even sides only");
    static_assert(P == 1 || NB == 1, "several images or several bands, not both");
    static_assert(MT <= 6, "accumulators: MT x NT x 2 x 4 registers");
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

// Four values at once, written on vectors so that the compiler packs the arithmetic two per instruction (v_pk_*): the
// epilogue is VALU-issue-bound (docs/DESIGN_history_r01_r03.md 4b).  No select for large x: e^x is capped at 1e18 (an unsigned integer min
// on the bits — e is never negative — so no NaN-canonicalising v_max comes with it); n = e (e + 2) then stays finite
// and n / (n + 2) is exactly 1 up there, so the product is x itself.  One rcp (1 ulp), no Newton step.
__device__ __forceinline__ f32x4 mish4(f32x4 x)
{
    const f32x4 t = x * 1.44269504088896341f;
    f32x4 e;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t b = __float_as_uint(__builtin_amdgcn_exp2f(t[i]));
        e[i] = __uint_as_float(b < 0x5D5E0B6Bu ? b : 0x5D5E0B6Bu);      // min(e, 1e18)
    }
    const f32x4 n = __builtin_elementwise_fma(e, e, e + e);
    const f32x4 d = n + 2.0f;
    const f32x4 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]), __builtin_amdgcn_rcpf(d[2]), __builtin_amdgcn_rcpf(d[3])};
    return x * (n * r);
}

// gy * mish'(x) the same way (tron_nn.hip's mish_grad1 with exp2 / rcp): t = n / (n + 2), 1 - t^2 = (2 / (n + 2)) (1 + t),
// mish' = t + x (1 - t^2) e / (1 + e).  With e capped at 1e18, t is exactly 1 and the second term 0 up there.
__device__ __forceinline__ f32x4 mish_grad4(f32x4 x, f32x4 gy)
{
    const f32x4 t2 = x * 1.44269504088896341f;
    f32x4 e;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t b = __float_as_uint(__builtin_amdgcn_exp2f(t2[i]));
        e[i] = __uint_as_float(b < 0x5D5E0B6Bu ? b : 0x5D5E0B6Bu);
    }
    const f32x4 n = __builtin_elementwise_fma(e, e, e + e);
    const f32x4 d = n + 2.0f, e1 = e + 1.0f;
    const f32x4 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]), __builtin_amdgcn_rcpf(d[2]), __builtin_amdgcn_rcpf(d[3])};
    const f32x4 q = {__builtin_amdgcn_rcpf(e1[0]), __builtin_amdgcn_rcpf(e1[1]), __builtin_amdgcn_rcpf(e1[2]), __builtin_amdgcn_rcpf(e1[3])};
    const f32x4 t = n * r, u = r + r;
    return gy * (t + x * (u * (1.0f + t)) * (e * q));
}

// v -> (hi, lo): v = hi + lo * 2^-11 up to 2^-22 |v|
__device__ __forceinline__ void split(float v, f16 &hi, f16 &lo)
{
    hi = (f16)v;
    lo = (f16)((v - (float)hi) * LO_SCALE);
}

// W[cout][cin][3][3] f32 -> workspace f16 [chunk][half][tap 10][cout][16 ci]: per chunk exactly the LDS weight image
// (hi image, then lo image; tenth tap and channels >= cin zero)
// dgrad: `w` is the FORWARD layer's weight [cin][cout][3][3] and the image built is that of the data-gradient
// convolution: channel axes swapped, taps reversed (W'[co][ci][tap] = W[ci][co][8 - tap]).
__global__ void k_split_weights(const float *__restrict__ w, int cout, int cin, int nchunks, int dgrad, f16 *__restrict__ ws)
{
    const int per_half = TAPS_PAD * cout * CIC;
    const int total = nchunks * per_half;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int c = i / per_half, r = i - c * per_half;
        const int tap = r / (cout * CIC), r2 = r - tap * (cout * CIC);
        const int co = r2 / CIC, cl = r2 - co * CIC;
        const int ci = c * CIC + cl;
        const float v = !(tap < 9 && ci < cin) ? 0.0f
                        : dgrad            ? w[((size_t)ci * cout + co) * 9 + (8 - tap)]
                                           : w[((size_t)co * cin + ci) * 9 + tap];
        f16 h, l;
        split(v, h, l);
        ws[(size_t)c * 2 * per_half + r] = h;
        ws[(size_t)c * 2 * per_half + per_half + r] = l;
    }
}

// the same for up to SPLIT_MAX layers in one launch (blockIdx.y = layer): a forward pass splits all its weights at once
constexpr int SPLIT_MAX = 8;
struct SplitJobs {
    const float *w[SPLIT_MAX];
    f16 *ws[SPLIT_MAX];
    int cout[SPLIT_MAX], cin[SPLIT_MAX];
};
__global__ void k_split_weights_multi(SplitJobs jobs)
{
    const int k = blockIdx.y, cout = jobs.cout[k], cin = jobs.cin[k], nchunks = (cin + CIC - 1) / CIC;
    const float *__restrict__ w = jobs.w[k];
    f16 *__restrict__ ws = jobs.ws[k];
    const int per_half = TAPS_PAD * cout * CIC;
    const int total = nchunks * per_half;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int c = i / per_half, r = i - c * per_half;
        const int tap = r / (cout * CIC), r2 = r - tap * (cout * CIC);
        const int co = r2 / CIC, cl = r2 - co * CIC;
        const int ci = c * CIC + cl;
        const float v = (tap < 9 && ci < cin) ? w[((size_t)co * cin + ci) * 9 + tap] : 0.0f;
        f16 h, l;
        split(v, h, l);
        ws[(size_t)c * 2 * per_half + r] = h;
        ws[(size_t)c * 2 * per_half + per_half + r] = l;
    }
}

// SMALL = the conv1 instantiation (cin 3 or 4: one chunk whose absent channels are zero; input = int8 observation
// codes when in_codes, else the f32 planes).  NT 16-channel tiles per wave: cout = 32 * NT.
// PERSIST (an even number of chunks): a workgroup walks image groups blockIdx.x, + gridDim.x, ... and stages the
// NEXT group's first chunk during the current group's last one — a group then starts on operands that are already in
// LDS instead of waiting for its first loads (12 % of a workgroup's life, stamped).
#ifdef TRON_CONV_WAVE_STAMPS  // diagnostic build only (scripts/conv_wave_stamps.py): per-wave cycles, read back by tron_conv_wave_stamps
__device__ unsigned long long g_wave_stamps[256 * 8 * 4];
#endif

//
// MBWD (the learner's backward, tron_conv3x3_dgrad_mish): the call is a data gradient whose result is carried through the
// activation of the layer BELOW before it is written: out = (conv + res) * mish'(zprev) — res is then the gradient that
// reaches the same tensor along a residual connection, zprev the layer below's pre-activation (laid out like out) — and
// every wave keeps the sums and the largest magnitudes of what it wrote, per channel, in its own LDS slots and leaves them
// in stats[2][workgroup][wave][16 NT] when it is done (k_mbwd_finish adds them up in a fixed order: the layer below's bias
// gradient and the scale its consumers need).  The epilogue's loads and its mish' arithmetic are exposed (one workgroup per
// CU, all waves in the same phase): +57 us on 168 at 4 096 x 12x12 x 64 channels, +250 on 824 at 26x26 — against 85 + 66 and
// 473 + 366 us for the passes it replaces.  (Tried: touching the group's zprev / res lines by 4-byte LDS-DMA while the last
// chunk computes, so that the epilogue's loads hit L2 — no change: the phase is bound by its arithmetic and stores.)
template <int S, int NT, bool SMALL, bool PERSIST, bool MBWD = false>
__global__ __launch_bounds__(THREADS, 2) void k_conv3x3_f16(
    const void *__restrict__ in, const f16 *__restrict__ ws, const float *__restrict__ bias,
    const float *__restrict__ res, float *__restrict__ out, float *__restrict__ pre_out, int B, int cin, float plane4,
    int apply_mish, int in_fmt, unsigned char *__restrict__ out_s16, const float *__restrict__ absmax, int n_absmax,
    int ngroups, const float *__restrict__ zprev, float *__restrict__ stats)
{
    static_assert(!PERSIST || !SMALL, "persistent groups: chunked input");   // (NB > 1: the host keeps the grid a multiple of NB,
                                                                             //  so a workgroup's band never changes)
    const int in_codes = in_fmt == 1;
    const bool in_s16 = !SMALL && in_fmt == 2;
    using C = Cfg<S>;
    constexpr int COUT = 32 * NT;
    constexpr int W_HALF = TAPS_PAD * COUT * PITCH;                     // hi (or lo) image of one weight chunk
    constexpr int IN_BUF = 2 * C::IN_HALF;
    constexpr int W_Q = 2 * W_HALF / 16;                                // 16-byte pieces of a weight chunk
    constexpr int W_LD = (W_Q + THREADS - 1) / THREADS;
    constexpr int W_BUF = 2 * W_HALF;
    constexpr int STEPS = SLABS * C::MT_MIN;                            // tile-steps every wave runs per chunk
    // steps that store weight pieces / input float4s: late enough for the loads issued at the chunk's start to have landed
    constexpr int STAGE_IN0 = STEPS - C::IN_LD;                         // (the weights go global -> LDS directly: TRON_DMA_W)
    static_assert(STAGE_IN0 >= 6, "give the global loads time");
    static_assert(STAGE_IN0 + C::IN_LD <= STEPS, "one staged piece per tile-step");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // LDS: in[0] (hi|lo) | in[1] (hi|lo) | w[0] (hi|lo) | w[1] (hi|lo) | dump (1 KB: where surplus threads' staging
    // writes go, so that the staging pieces are branch-free and can be scheduled between MFMAs)
    unsigned char *dump = lds + 2 * IN_BUF + 2 * W_BUF;
    float *sacc = reinterpret_cast<float *>(dump + 1024);               // MBWD: [wave][sum | max][16 NT <= 32], touched by its wave only

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // waves w and w + 4 share a SIMD: the second N half walks the M slots backwards, so a SIMD gets 5 + 4 tiles
    const int wn = wave / NWM, wm = wn ? NWM - 1 - (wave & (NWM - 1)) : (wave & (NWM - 1));
    const int li = lane & 15, g = lane >> 4, tsel = g >> 1, oct = g & 1;
    int grp = blockIdx.x;                                               // image group (x band): PERSIST walks grp += gridDim.x
    int img0 = (grp / C::NB) * C::P;
    const int band = grp % C::NB;                                       // neighbours in the grid share an image: halo rows hit L2
    const int r0 = band * C::SPLIT, rows = band == C::NB - 1 ? C::LAST : C::SPLIT;
    const int rpx = rows * S;                                           // pixels of one image in this region
    const int npx = C::P * rpx;
    const int tiles = (npx + 15) >> 4;
    const int t_base = tiles / NWM, t_rem = tiles % NWM;
    const int tile0 = wm * t_base + (wm < t_rem ? wm : t_rem);           // first M tile of this wave
    const int my_mt = t_base + (wm < t_rem ? 1 : 0);
    const int nchunks = SMALL ? 1 : cin / CIC;
    int last_img = B - 1 - img0;
    const int cout = COUT;

#ifdef TRON_CONV_STAMPS     // diagnostic build only: pre_out is a stamp buffer u64[blocks][6], never an output
    unsigned long long *stamps = reinterpret_cast<unsigned long long *>(pre_out) + (size_t)blockIdx.x * 6;
    pre_out = nullptr;
    if (tid == 0) { stamps[0] = __builtin_amdgcn_s_memtime(); stamps[1] = __builtin_amdgcn_s_memrealtime(); }
#endif
    // Scale of the f32 activations on their way into f16: 2^-6 for the network's activations; a gradient tensor (the
    // data-gradient call) is orders of magnitude smaller and brings its per-block maxima along: the power of two that
    // puts its largest magnitude in [2^13, 2^14) keeps the halves in f16's normal range.
    float act_scale = ACT_SCALE, act_unscale = ACT_UNSCALE;
    if (absmax) {
        __shared__ float red[THREADS / 64];
        float m = 0.0f;
        for (int i = tid; i < n_absmax; i += THREADS) m = fmaxf(m, absmax[i]);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
        if (lane == 0) red[wave] = m;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < THREADS / 64; ++k) m = fmaxf(m, red[k]);
        const int e = (int)((__float_as_uint(m) >> 23) & 255u) - 126;   // m = f 2^e, f in [0.5, 1)
        if (m > 0.0f && e >= -100 && e <= 100) {
            act_scale = __uint_as_float((uint32_t)(127 + 14 - e) << 23);
            act_unscale = __uint_as_float((uint32_t)(127 - 14 + e) << 23);
        }
    }
    if (MBWD) sacc[tid] = 0.0f;                                         // (8 waves x 64 slots)
    // zero both input buffers once: halo pixels (and, for SMALL, the absent channels) stay zero for good
    for (int i = tid; i < 2 * IN_BUF / 16; i += THREADS) reinterpret_cast<uint4 *>(lds)[i] = make_uint4(0u, 0u, 0u, 0u);

    // per-lane operand bases (bytes)
    int a_base[C::MT];
#pragma unroll
    for (int t = 0; t < C::MT; ++t) {
        int px = 16 * (tile0 + t) + li;
        px = px < npx ? px : npx - 1;                                   // surplus tile slots / pixels: valid address, unused
        const int img = px / rpx, p = px - img * rpx;
        const int y = p / S, x = p - y * S;                             // y: row within the band
        a_base[t] = (img * C::PLANE + y * C::SP + x) * PITCH + oct * 16;
    }
    // byte offset of slab s's tap for this lane: taps 2s (tsel 0) / 2s+1 (tsel 1); the tenth tap (weights zero) reads
    // the ninth's pixels.  Two literals and a select per use instead of five registers held across the MFMA loop.
    auto tap_offset = [&](int s) {
        const int t0 = 2 * s, t1 = (2 * s + 1 < 9) ? 2 * s + 1 : 8;
        const int o0 = ((t0 / 3) * C::SP + (t0 % 3)) * PITCH, o1 = ((t1 / 3) * C::SP + (t1 % 3)) * PITCH;
        return tsel ? o1 : o0;
    };
    const int b_base = (tsel * COUT + wn * 16 * NT + li) * PITCH + oct * 16;

    size_t wg_base = (size_t)img0 * cout * C::SS;
    const float *in_wg = (SMALL && in_codes) ? nullptr : reinterpret_cast<const float *>(in) + (size_t)img0 * cin * C::SS;
    // the split-f16 activation image ("S16"): per image [16-channel chunk][hi | lo][pixel][16 ci] f16 — as many bytes as
    // the f32 tensor, laid out so that a chunk's rows are copied into the padded LDS planes 16 bytes at a time
    const unsigned char *in16_wg = reinterpret_cast<const unsigned char *>(in) + (size_t)img0 * cin * C::SS * 4;
    // what the staging macros read from: the current group's images, or (PERSIST, last chunk) the next group's
    const float *ld_in_wg = in_wg;
    const unsigned char *ld_in16 = in16_wg;
    int ld_last = last_img;

    f32x4 acc0[C::MT][NT], acc1[C::MT][NT];
#pragma unroll
    for (int t = 0; t < C::MT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            acc0[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc1[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

    f32x4 rin[C::IN_LD];
    const int epx = (rows + 2) * S;                                     // a band's pixels plus its two halo rows
    const int n_items = C::P * epx * (CIC / 4);                         // f32 input: (pixel, quad) items; S16 input: as many 16-byte pieces
    const int erows = rows + 2, npr = 2 * S, nph = C::P * erows * npr;  // S16: pieces per row, per half
    // the staging index math divides by band-dependent (wave-uniform) values: multiply-high by a reciprocal computed
    // once (exact while x * d < 2^32; x, d < 2^12 here) instead of a ~25-instruction integer division per use
    const uint32_t m_epx = 0xFFFFFFFFu / (uint32_t)epx + 1u, m_nph = 0xFFFFFFFFu / (uint32_t)nph + 1u,
                   m_ernpr = 0xFFFFFFFFu / (uint32_t)(erows * npr) + 1u;
#define TRON_DIV(x_, m_) ((int)__umulhi((uint32_t)(x_), (m_)))

    // ---- staging pieces -------------------------------------------------------------------------------------
    // one activation -> its (hi, lo) halves at pixel `pix_` (padded index), channel `ci_`, input buffer at `ib_`
#define TRON_PUT_IN(ib_, pix_, ci_, v_)                                                                              \
    do {                                                                                                              \
        f16 h_, l_;                                                                                                   \
        split((v_) * ACT_SCALE, h_, l_);                                                                              \
        *reinterpret_cast<f16 *>((ib_) + (pix_) * PITCH + (ci_) * 2) = h_;                                            \
        *reinterpret_cast<f16 *>((ib_) + C::IN_HALF + (pix_) * PITCH + (ci_) * 2) = l_;                               \
    } while (0)
    // Input staging.  An item = (pixel, channel quad): thread i takes items i, i + THREADS, ...; consecutive lanes hold
    // consecutive quads of a pixel and then the next pixel, so a wave's 8-byte LDS writes are one contiguous 512-byte
    // run (lane = pixel with the channel fixed would put every lane of a group on the same bank: rows are 32 bytes).
    // The four values of an item are four dword loads from four channel planes (16 consecutive pixels per plane and
    // instruction).  Unconditional: indices clamped, surplus dropped at the LDS write.
#define TRON_LOAD_IN(c_)                                                                                             \
    do {                                                                                                              \
        int tidl_ = tid;                                                                                              \
        asm volatile("" : "+v"(tidl_));                      /* keep the address math out of registers across chunks */ \
        if (in_s16) {                                         /* already split by the producing layer: 16-byte pieces */ \
            _Pragma("unroll") for (int j = 0; j < C::IN_LD; ++j) {                                                    \
                int q_ = tidl_ + j * THREADS;                                                                         \
                q_ = q_ < n_items ? q_ : n_items - 1;                                                                 \
                const int h_ = TRON_DIV(q_, m_nph), r_ = q_ - h_ * nph;                                                          \
                const int im_ = TRON_DIV(r_, m_ernpr), r2_ = r_ - im_ * (erows * npr);                                   \
                const int re_ = r2_ / npr, pc_ = r2_ - re_ * npr;                                                     \
                int gr_ = r0 - 1 + re_;                                                                               \
                gr_ = gr_ < 0 ? 0 : (gr_ >= S ? S - 1 : gr_);                                                         \
                const int ims_ = im_ < ld_last ? im_ : ld_last;                                                     \
                rin[j] = *reinterpret_cast<const f32x4 *>(ld_in16 + ((size_t)ims_ * cin * C::SS * 4 +                 \
                                                                      ((c_) * 2 + h_) * C::SS * 32 + gr_ * S * 32 + pc_ * 16)); \
            }                                                                                                         \
        } else {                                                                                                      \
        _Pragma("unroll") for (int j = 0; j < C::IN_LD; ++j) {                                                        \
            int q_ = tidl_ + j * THREADS;                                                                             \
            q_ = q_ < n_items ? q_ : n_items - 1;                                                                     \
            const int px_ = q_ >> 2, quad_ = q_ & 3;                                                                  \
            const int im_ = TRON_DIV(px_, m_epx), p_ = px_ - im_ * epx;             /* p_: pixel within the band + halo rows */  \
            const int re_ = p_ / S, xx_ = p_ - re_ * S;                                                               \
            int gr_ = r0 - 1 + re_;                                       /* image row; clamped: dropped at the write */ \
            gr_ = gr_ < 0 ? 0 : (gr_ >= S ? S - 1 : gr_);                                                             \
            const int ims_ = im_ < ld_last ? im_ : ld_last;                                                         \
            const float *src_ = ld_in_wg + ((ims_ * cin + (c_) * CIC + quad_ * 4) * C::SS + gr_ * S + xx_);              \
            rin[j] = (f32x4){src_[0], src_[C::SS], src_[2 * C::SS], src_[3 * C::SS]};                                 \
        }                                                                                                             \
        }                                                                                                             \
    } while (0)
    // staged item j_ -> input buffer ib_ (4 hi halves in one 8-byte write, 4 lo halves in another)
#define TRON_STORE_IN(ib_, j_)                                                                                       \
    do {                                                                                                              \
        int tidv_ = tid;                                                                                              \
        asm volatile("" : "+v"(tidv_));                                                                               \
        const int q_ = tidv_ + (j_) * THREADS;                                                                        \
        if (in_s16) {                                                                                                 \
            const int h_ = TRON_DIV(q_, m_nph), r_ = q_ - h_ * nph;                                                              \
            const int im_ = TRON_DIV(r_, m_ernpr), r2_ = r_ - im_ * (erows * npr);                                       \
            const int re_ = r2_ / npr, pc_ = r2_ - re_ * npr;                                                         \
            const int gr_ = r0 - 1 + re_;                                                                             \
            const bool ok_ = q_ < n_items && gr_ >= 0 && gr_ < S && im_ <= ld_last;                                  \
            const int off_ = h_ * C::IN_HALF + (im_ * C::PLANE + re_ * C::SP + 1) * PITCH + pc_ * 16;                 \
            *reinterpret_cast<f32x4 *>(ok_ ? (ib_) + off_ : dump + lane * 16) = rin[j_];   /* branch-free: see dump */  \
        } else {                                                                                                      \
        const int px_ = q_ >> 2, quad_ = q_ & 3;                                                                      \
        const int im_ = TRON_DIV(px_, m_epx), p_ = px_ - im_ * epx;                                                              \
        const int re_ = p_ / S, xx_ = p_ - re_ * S;                                                                   \
        const int gr_ = r0 - 1 + re_;                                                                                 \
        const bool ok_ = q_ < n_items && gr_ >= 0 && gr_ < S;             /* rows outside the image stay zero */        \
        const int off_ = (im_ * C::PLANE + re_ * C::SP + (xx_ + 1)) * PITCH + quad_ * 8;                              \
        f16x4 h_, l_;                                                                                                 \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                               \
            f16 hh_, ll_;                                                                                             \
            split((im_ <= ld_last ? rin[j_][e] : 0.0f) * act_scale, hh_, ll_);                                       \
            h_[e] = hh_;                                                                                              \
            l_[e] = ll_;                                                                                              \
        }                                                                                                             \
        *reinterpret_cast<f16x4 *>(ok_ ? (ib_) + off_ : dump + lane * 16) = h_;                                       \
        *reinterpret_cast<f16x4 *>(ok_ ? (ib_) + C::IN_HALF + off_ : dump + lane * 16 + 8) = l_;                      \
        }                                                                                                             \
    } while (0)
    // weight chunk c_: a linear copy of its pre-split image, piece j_
    // Weight chunk c_ -> LDS buffer wb_: its pre-split image is copied as it is, so it goes global -> LDS directly
    // (global_load_lds_dwordx4: per-lane source, destination = a wave-uniform base + 16 * lane) — no staging registers,
    // no ds_write_b128 (the slow store path: 13 cycles per wave-instruction), nothing to schedule into the MFMA loop.
    // A copy is retired by the issuing wave's vmcnt(0) followed by the chunk barrier.  (Tried on top of this and dropped:
    // input pieces two chunks ahead in a second register set — with a DMA in flight, and across the chunk loop's back
    // edge, hipcc waits vmcnt(0) at the first use of any staged register, which drains the younger loads as well, so the
    // deeper prefetch buys nothing short of hand-placed waits around inline-asm loads: 1.26 -> 1.33 ms for the trunk.)
#define TRON_DMA_W(wb_, c_)                                                                                          \
    do {                                                                                                              \
        _Pragma("unroll") for (int j = 0; j < W_LD; ++j) {                                                            \
            if ((j + 1) * THREADS <= W_Q || wave * 64 + j * THREADS < W_Q) {   /* whole waves in or out: W_Q % 64 == 0 */ \
                __builtin_amdgcn_global_load_lds(                                                                     \
                    (const __attribute__((address_space(1))) void *)(reinterpret_cast<const unsigned char *>(ws) +    \
                                                                     (size_t)(c_) * W_BUF + (size_t)(tid + j * THREADS) * 16), \
                    (__attribute__((address_space(3))) void *)((wb_) + (wave * 64 + j * THREADS) * 16), 16, 0, 0);    \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)

    // ---- one chunk: 5 slabs x up to MT tile-steps; STAGE_: also bring chunk c+1 into the other buffers ---------------
#define TRON_CHUNK(STAGE_, c_, cn_)                                                                                    \
    do {                                                                                                              \
        const unsigned char *in_h = lds + ((c_) & 1) * IN_BUF, *in_l = in_h + C::IN_HALF;                             \
        const unsigned char *w_h = lds + 2 * IN_BUF + ((c_) & 1) * W_BUF, *w_l = w_h + W_HALF;                        \
        unsigned char *nxt_in_ = lds + (((c_) + 1) & 1) * IN_BUF;                                                     \
        unsigned char *nxt_w_ = lds + 2 * IN_BUF + (((c_) + 1) & 1) * W_BUF;                                          \
        if (STAGE_ && TRON_F16_ABLATE != 1 && TRON_F16_ABLATE != 4) {                                                 \
            if (TRON_F16_ABLATE != 6) TRON_DMA_W(nxt_w_, cn_);                                                        \
            if (TRON_F16_ABLATE != 7) TRON_LOAD_IN(cn_);                                                              \
        }                                                                                                             \
        _Pragma("unroll") for (int s = 0; s < SLABS; ++s) {                                                           \
            /* Vector issue goes to the older wave of a SIMD first: left alone, waves 0-3 run each chunk ahead of waves  */ \
            /* 4-7 and then wait ~1.5 K cycles at its barrier while those finish alone.  The two waves of a SIMD swap    */ \
            /* priority every slab, so neither gets more than a slab ahead.                                              */ \
            if (TRON_F16_PRIO) {                                                                                      \
                if (wn) { if (s & 1) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1); }             \
                else    { if (s & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }             \
            }                                                                                                         \
            f16x8 bh[NT], bl[NT];                                                                                     \
            _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                                          \
                const int bo = b_base + (2 * s * COUT + n * 16) * PITCH;                                              \
                bh[n] = *reinterpret_cast<const f16x8 *>(w_h + bo);                                                   \
                bl[n] = *reinterpret_cast<const f16x8 *>(w_l + bo);                                                   \
            }                                                                                                         \
            /* three-deep register pipeline over the M tiles: tile t+2's fragments are requested while tile t's MFMAs */ \
            /* issue — one tile-step (96 cycles) does not cover the LDS latency under load                           */ \
            /* The lo*hi product adds to the accumulator the hi*lo product wrote, and a dependent MFMA issues 48 cycles   */ \
            /* after its producer (an independent one 16): tile t's lo*hi MFMAs are issued two tile-steps later, so that  */ \
            /* a whole step lies between the pair wherever the compiler puts them inside a step (ring of 5 fragments).    */ \
            f16x8 ah[5], al[5];                                                                                       \
            const int toff = tap_offset(s);                                                                           \
            ah[0] = *reinterpret_cast<const f16x8 *>(in_h + a_base[0] + toff);                                        \
            al[0] = *reinterpret_cast<const f16x8 *>(in_l + a_base[0] + toff);                                        \
            ah[1] = *reinterpret_cast<const f16x8 *>(in_h + a_base[1] + toff);                                        \
            al[1] = *reinterpret_cast<const f16x8 *>(in_l + a_base[1] + toff);                                        \
            _Pragma("unroll") for (int t = 0; t < C::MT; ++t) {                                                       \
                if (t + 2 < C::MT && TRON_F16_ABLATE != 2) {                                                          \
                    ah[(t + 2) % 5] = *reinterpret_cast<const f16x8 *>(in_h + a_base[t + 2] + toff);                  \
                    al[(t + 2) % 5] = *reinterpret_cast<const f16x8 *>(in_l + a_base[t + 2] + toff);                  \
                }                                                                                                     \
                __builtin_amdgcn_sched_barrier(0);                                                                    \
                if (STAGE_ && t < C::MT_MIN && TRON_F16_ABLATE != 1 && TRON_F16_ABLATE != 3) { /* staging rides in the shadow of this tile's MFMAs */ \
                    const int step = s * C::MT_MIN + t;                                                               \
                    if (step >= STAGE_IN0 && step < STAGE_IN0 + C::IN_LD) TRON_STORE_IN(nxt_in_, step - STAGE_IN0);   \
                }                                                                                                     \
                if (STAGE_ && t < C::MT_MIN && TRON_F16_ABLATE != 1 && TRON_F16_ABLATE != 3 &&                        \
                    s * C::MT_MIN + t >= STAGE_IN0 && s * C::MT_MIN + t < STAGE_IN0 + C::IN_LD) {                     \
                    /* spread the piece's VALU / LDS-write instructions between this tile's MFMAs */                  \
                    _Pragma("unroll") for (int i = 0; i < 3 * NT; ++i) {                                              \
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                            \
                        __builtin_amdgcn_sched_group_barrier(0x006, 8, 0);                                            \
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                            \
                    }                                                                                                 \
                }                                                                                                     \
                if (t >= 2 && (t - 2 < C::MT_MIN || my_mt > t - 2)) {                                                 \
                    _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                    \
                        acc1[t >= 2 ? t - 2 : 0][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[(t + 3) % 5], bh[n], acc1[t >= 2 ? t - 2 : 0][n], 0, 0, 0); \
                }                                                                                                     \
                if (t < C::MT_MIN || my_mt > t) {                                                                     \
                    _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                                  \
                        acc1[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t % 5], bl[n], acc1[t][n], 0, 0, 0);   \
                        acc0[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t % 5], bh[n], acc0[t][n], 0, 0, 0);   \
                    }                                                                                                 \
                }                                                                                                     \
                __builtin_amdgcn_sched_barrier(0);                                                                    \
            }                                                                                                         \
            _Pragma("unroll") for (int t = (C::MT >= 2 ? C::MT - 2 : 0); t < C::MT; ++t) {   /* the last two tiles' lo*hi */ \
                if (t < C::MT_MIN || my_mt > t) {                                                                     \
                    _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                    \
                        acc1[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t % 5], bh[n], acc1[t][n], 0, 0, 0);   \
                }                                                                                                     \
            }                                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
        }                                                                                                             \
    } while (0)

    // ---- prologue: chunk 0 -------------------------------------------------------------------------------------
    __syncthreads();                                                     // zero fill done
    if (SMALL) {
        // conv1: split the input here (27 / 36 k-values per output: nothing to amortise); one pass over the band's
        // pixels and halo rows
        for (int e = tid; e < C::P * epx; e += THREADS) {
            const int im = e / epx, p = e - im * epx;
            const int re = p / S, x = p - re * S;
            const int gr = r0 - 1 + re;
            if (gr < 0 || gr >= S) continue;
            const int pix = im * C::PLANE + re * C::SP + (x + 1);
            const bool have = im <= last_img;
            if (in_codes) {
                const int v = have ? (int)reinterpret_cast<const int8_t *>(in)[(size_t)(img0 + im) * C::SS + gr * S + x] : 1;
                TRON_PUT_IN(lds, pix, 0, (v == -1) ? 1.0f : 0.0f);                              // util.py:18-19
                TRON_PUT_IN(lds, pix, 1, (v == -2) ? 1.0f : (v == 10) ? 10.0f : 0.0f);         // util.py:20-21,26-27
                TRON_PUT_IN(lds, pix, 2, (v == -3) ? 1.0f : (v == -10) ? 10.0f : 0.0f);
                if (cin == 4) TRON_PUT_IN(lds, pix, 3, have ? plane4 : 0.0f);
            } else {
                for (int ci = 0; ci < cin; ++ci)
                    TRON_PUT_IN(lds, pix, ci, have ? in_wg[(im * cin + ci) * C::SS + gr * S + x] : 0.0f);
            }
        }
    } else {
        TRON_LOAD_IN(0);
#pragma unroll
        for (int j = 0; j < C::IN_LD; ++j) TRON_STORE_IN(lds, j);
    }
    TRON_DMA_W(lds + 2 * IN_BUF, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#ifdef TRON_CONV_STAMPS
    if (tid == 0) { stamps[2] = __builtin_amdgcn_s_memtime(); stamps[3] = __builtin_amdgcn_s_memrealtime(); }
#endif
    // ---- the chunks: one barrier each (next chunk's buffers written, this chunk's buffers free again) ---------------
#ifdef TRON_CONV_WAVE_STAMPS
    unsigned long long ws_loop = 0, ws_bar = 0, ws_epi = 0, ws_groups = 0;
#endif
    for (;;) {                                                          // image groups (one pass unless PERSIST)
#ifdef TRON_CONV_WAVE_STAMPS
    const unsigned long long ws_t0 = __builtin_amdgcn_s_memtime();
#endif
    for (int c = 0; c + 1 < nchunks; ++c) {
        TRON_CHUNK(true, c, c + 1);
#ifdef TRON_CONV_WAVE_STAMPS
        const unsigned long long ws_b0 = __builtin_amdgcn_s_memtime();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this wave's weight copies (LDS-DMA) have landed ...
        __syncthreads();                                                // ... and after the barrier everybody's have
#ifdef TRON_CONV_WAVE_STAMPS
        ws_bar += __builtin_amdgcn_s_memtime() - ws_b0;
#endif
    }
    const int grp_next = grp + (int)gridDim.x;
    if (PERSIST) {                                                      // the last chunk brings in the next group's chunk 0
        const int img0n = ((grp_next < ngroups ? grp_next : grp) / C::NB) * C::P;   // (past the end: this group's again, unused)
        ld_in_wg = reinterpret_cast<const float *>(in) + (size_t)img0n * cin * C::SS;
        ld_in16 = reinterpret_cast<const unsigned char *>(in) + (size_t)img0n * cin * C::SS * 4;
        ld_last = B - 1 - img0n;
    }
    TRON_CHUNK(PERSIST, nchunks - 1, 0);

#undef TRON_CHUNK
#undef TRON_DMA_W
#undef TRON_STORE_IN
#undef TRON_DIV
#undef TRON_LOAD_IN
#undef TRON_PUT_IN

#ifdef TRON_CONV_STAMPS
    if (tid == 0) { stamps[4] = __builtin_amdgcn_s_memtime(); stamps[5] = __builtin_amdgcn_s_memrealtime(); }
#endif
#ifdef TRON_CONV_WAVE_STAMPS
    const unsigned long long ws_t1 = __builtin_amdgcn_s_memtime();
    ws_loop += ws_t1 - ws_t0;
#endif
    // ---- epilogue (as tron_conv.hip): D row = 4 * (lane >> 4) + r (pixel), column = lane & 15 (channel) ------------
    int o[C::MT];
    bool live[C::MT];
    int g_e = g;                                                        // (an opaque copy: in the persistent variant the output
    asm volatile("" : "+v"(g_e));                                       //  offsets below are loop-invariant, and hoisted they
#pragma unroll                                                          //  would sit in registers across the whole MFMA loop)
    for (int t = 0; t < C::MT; ++t) {
        const int px = 16 * (tile0 + t) + 4 * g_e;                       // npx is a multiple of 4: all four pixels or none
        const int img = px / rpx, p = px - img * rpx;
        live[t] = t < my_mt && px < npx && img <= last_img;
        o[t] = (img * cout + wn * 16 * NT + li) * C::SS + r0 * S + p;
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const float bv = bias ? bias[wn * 16 * NT + n * 16 + li] : 0.0f;
#pragma unroll
        for (int t = 0; t < C::MT; ++t) acc0[t][n] = (acc0[t][n] + acc1[t][n] * LO_UNSCALE) * act_unscale + bv;
    }
    if (res) {
        const float *res_wg = res + wg_base;
        f32x4 r[C::MT][NT];
#pragma unroll
        for (int t = 0; t < C::MT; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n)
                r[t][n] = live[t] ? *reinterpret_cast<const f32x4 *>(res_wg + (uint32_t)(o[t] + n * 16 * C::SS)) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < C::MT; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc0[t][n] += r[t][n];
    }
    if (MBWD) {
        const float *z_wg = zprev + wg_base;
        float bsum[NT], bmax[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) bsum[n] = bmax[n] = 0.0f;
#pragma unroll
        for (int t = 0; t < C::MT; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const f32x4 z = live[t] ? *reinterpret_cast<const f32x4 *>(z_wg + (uint32_t)(o[t] + n * 16 * C::SS)) : (f32x4){0.f, 0.f, 0.f, 0.f};
                const f32x4 v = live[t] ? mish_grad4(z, acc0[t][n]) : (f32x4){0.f, 0.f, 0.f, 0.f};
                acc0[t][n] = v;
                bsum[n] += (v[0] + v[1]) + (v[2] + v[3]);
                bmax[n] = fmaxf(fmaxf(bmax[n], fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
            }
#pragma unroll
        for (int n = 0; n < NT; ++n) {                                  // lanes li, li + 16, li + 32, li + 48 hold one channel
            bsum[n] += __shfl_xor(bsum[n], 16);
            bsum[n] += __shfl_xor(bsum[n], 32);
            bmax[n] = fmaxf(bmax[n], __shfl_xor(bmax[n], 16));
            bmax[n] = fmaxf(bmax[n], __shfl_xor(bmax[n], 32));
            if (g_e == 0) {
                float *slot = sacc + wave * 64 + n * 16 + li;
                slot[0] += bsum[n];
                slot[32] = fmaxf(slot[32], bmax[n]);
            }
        }
    }
    float *out_wg = out ? out + wg_base : nullptr;
    float *pre_wg = pre_out ? pre_out + wg_base : nullptr;
    unsigned char *o16_wg = out_s16 ? out_s16 + wg_base * 4 : nullptr;
#pragma unroll
    for (int t = 0; t < C::MT; ++t) {
        if (!live[t]) continue;
        const int px = 16 * (tile0 + t) + 4 * g_e;
        const int img = px / rpx, pg = r0 * S + (px - img * rpx);        // image, pixel within the image
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            f32x4 v = acc0[t][n];
            const uint32_t ob = (uint32_t)(o[t] + n * 16 * C::SS);       // unsigned 32-bit offsets from a uniform base: no 64-bit address math per store
            if (pre_wg) *reinterpret_cast<f32x4 *>(pre_wg + ob) = v;
            if (apply_mish) v = mish4(v);
            if (out_wg) *reinterpret_cast<f32x4 *>(out_wg + ob) = v;
            if (o16_wg) {                 // the same values as the next layer's operand halves: [chunk][hi | lo][pixel][16 ci]
                unsigned char *d = o16_wg + (uint32_t)(img * cout * C::SS * 4 + ((wn * NT + n) * 2) * C::SS * 32 + pg * 32 + li * 2);
                const f32x4 sv = v * ACT_SCALE;
                const f16x4 hh = __builtin_convertvector(sv, f16x4);
                const f16x4 ll = __builtin_convertvector((sv - __builtin_convertvector(hh, f32x4)) * LO_SCALE, f16x4);
                if (TRON_F16_ABLATE == 5) {   /* diagnostic: one store instead of eight, every value still computed */
                    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
                    const u16x4 a = __builtin_bit_cast(u16x4, hh), b = __builtin_bit_cast(u16x4, ll);
                    *reinterpret_cast<unsigned short *>(d) = (unsigned short)(a[0] ^ a[1] ^ a[2] ^ a[3] ^ b[0] ^ b[1] ^ b[2] ^ b[3]);
                } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    *reinterpret_cast<f16 *>(d + r * 32) = hh[r];
                    *reinterpret_cast<f16 *>(d + C::SS * 32 + r * 32) = ll[r];
                }
                }
            }
        }
    }
#ifdef TRON_CONV_WAVE_STAMPS
    ws_epi += __builtin_amdgcn_s_memtime() - ws_t1;
    ws_groups += 1;
    if (lane == 0 && blockIdx.x < 256) {
        unsigned long long *d = g_wave_stamps + (blockIdx.x * 8 + wave) * 4;
        d[0] = ws_loop; d[1] = ws_bar; d[2] = ws_epi; d[3] = ws_groups;
    }
#endif
    if (!PERSIST || grp_next >= ngroups) break;                         // (uniform over the workgroup)
    // ---- next group: its chunk 0 is in LDS already (staged during the last chunk above) ----------------------------
    grp = grp_next;
    img0 = (grp / C::NB) * C::P;
    last_img = B - 1 - img0;
    wg_base = (size_t)img0 * cout * C::SS;
    in_wg = ld_in_wg;
    in16_wg = ld_in16;
#pragma unroll
    for (int t = 0; t < C::MT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            acc0[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc1[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // (the next group's weight chunk 0: LDS-DMA)
    __syncthreads();                                                    // the staged chunk is complete, the old buffers are free
    }
    if (MBWD && lane < 16 * NT) {
        const size_t slot = ((size_t)blockIdx.x * (THREADS / 64) + wave) * (16 * NT) + lane;
        stats[slot] = sacc[wave * 64 + lane];
        stats[(size_t)gridDim.x * (THREADS / 64) * (16 * NT) + slot] = sacc[wave * 64 + 32 + lane];
    }
}

template <int S, int NT, bool SMALL, bool PERSIST, bool MBWD = false>
int launch(const void *in, const f16 *ws, const float *bias, const float *res, float *out, float *pre_out, int64_t B,
           int cin, float plane4, int apply_mish, int in_fmt, void *out_s16, const float *absmax, int n_absmax, hipStream_t st,
           const float *zprev = nullptr, float *stats = nullptr, int64_t *grid_out = nullptr, int64_t grid_max = 0)
{
    using C = Cfg<S>;
    constexpr size_t LDS_BYTES = 4 * (size_t)C::IN_HALF + 4 * (size_t)TAPS_PAD * 32 * NT * PITCH + 1024 + (MBWD ? THREADS * 4 : 0);
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    auto kern = k_conv3x3_f16<S, NT, SMALL, PERSIST, MBWD>;
    static uint64_t prepared = 0;
    static int cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_NO_DEVICE; }
    if (!(prepared & (1ull << (dev & 63)))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)LDS_BYTES) != hipSuccess)
            (void)hipGetLastError();
        hipDeviceProp_t prop;
        cus[dev & 63] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
        (void)hipGetLastError();
        prepared |= 1ull << (dev & 63);
    }
    const int64_t groups = (B + C::P - 1) / C::P * C::NB;
    // PERSIST: one workgroup per CU (the kernel's LDS allows no more) walking groups blockIdx.x, + grid, ...
    const int64_t grid = PERSIST && groups > cus[dev & 63] ? cus[dev & 63] / C::NB * C::NB : groups;
    if (grid_out) {
        if (grid > grid_max) return TRON_ERR_UNSUPPORTED;               // (the caller sized its partial-sum scratch for grid_max workgroups)
        *grid_out = grid;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(THREADS), LDS_BYTES, st, in, ws, bias, res, out, pre_out, (int)B,
                       cin, plane4, apply_mish, in_fmt, reinterpret_cast<unsigned char *>(out_s16), absmax, n_absmax, (int)groups,
                       zprev, stats);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

// stats[2][workgroups][8 waves][16 NT] of a MBWD launch -> bias_grad[c] (sums, fixed order) and absmax[c]; one workgroup per channel
__global__ __launch_bounds__(256) void k_mbwd_finish(const float *__restrict__ stats, int groups, int nt, float *__restrict__ bias_grad,
                                                     float *__restrict__ absmax)
{
    __shared__ float rs[256], rm[256];
    const int c = blockIdx.x, per = 16 * nt, wn = c / per, col = c - wn * per;
    const size_t half = (size_t)groups * (THREADS / 64) * per;
    float s = 0.0f, m = 0.0f;
    for (int i = threadIdx.x; i < groups * NWM; i += 256) {             // (group, wave row) pairs; wave = wn * NWM + row
        const size_t slot = ((size_t)(i / NWM) * (THREADS / 64) + wn * NWM + (i % NWM)) * per + col;
        s += stats[slot];
        m = fmaxf(m, stats[half + slot]);
    }
    rs[threadIdx.x] = s;
    rm[threadIdx.x] = m;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) {
            rs[threadIdx.x] += rs[threadIdx.x + d];
            rm[threadIdx.x] = fmaxf(rm[threadIdx.x], rm[threadIdx.x + d]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        bias_grad[c] = rs[0];
        absmax[c] = rm[0];
    }
}

constexpr int64_t MBWD_GRID_MAX = 1024;         // workgroups a MBWD launch may have (persistent: one per CU)

}  // namespace

#ifdef TRON_CONV_WAVE_STAMPS
extern "C" int tron_conv_wave_stamps(unsigned long long *host_dst)
{
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_wave_stamps), sizeof(g_wave_stamps)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int tron_conv3x3_split_weights(const float *const *weights, const int32_t *cins, const int32_t *couts,
                                          void *const *workspaces, int32_t n, void *stream)
{
    if (!weights || !cins || !couts || !workspaces || n < 1 || n > SPLIT_MAX) return TRON_ERR_BAD_ARG;
    SplitJobs jobs{};
    int most = 0;
    for (int k = 0; k < n; ++k) {
        if (!weights[k] || !workspaces[k] || cins[k] < 1 || couts[k] < 1 || cins[k] > 1024 || couts[k] > 1024 ||
            (reinterpret_cast<uintptr_t>(workspaces[k]) & 15u))
            return TRON_ERR_BAD_ARG;
        jobs.w[k] = weights[k];
        jobs.ws[k] = reinterpret_cast<f16 *>(workspaces[k]);
        jobs.cin[k] = cins[k];
        jobs.cout[k] = couts[k];
        const int total = (cins[k] + CIC - 1) / CIC * TAPS_PAD * couts[k] * CIC;
        most = total > most ? total : most;
    }
    hipLaunchKernelGGL(k_split_weights_multi, dim3((most + 255) / 256, n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), jobs);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

int64_t tron_conv3x3_f16x3_workspace(int cin, int cout)
{
    const int64_t nchunks = (cin + CIC - 1) / CIC;
    return nchunks * 2 * TAPS_PAD * cout * CIC * (int64_t)sizeof(f16);
}

// called by tron_conv3x3_fwd (tron_conv.hip) after it validated the arguments; TRON_ERR_UNSUPPORTED = not this shape
int tron_conv3x3_f16x3(const void *in, int in_fmt, const float *weight, const float *bias, const float *residual,
                       float *out, float *pre_out, int64_t batch, int cin, int cout, int side, float plane4,
                       int apply_mish, void *workspace, void *out_split, int dgrad, const float *grad_absmax,
                       int n_absmax, int presplit, hipStream_t st)
{
    const bool small = cin == 3 || cin == 4;
    if ((side != 12 && side != 26 && side != 34) || (!small && cin % CIC != 0) || !workspace) return TRON_ERR_UNSUPPORTED;
    f16 *ws = reinterpret_cast<f16 *>(workspace);
    const int nchunks = (cin + CIC - 1) / CIC;
    const int total = nchunks * TAPS_PAD * cout * CIC;
    if (!presplit)
        hipLaunchKernelGGL(k_split_weights, dim3((total + 255) / 256), dim3(256), 0, st, weight, cout, cin, nchunks, dgrad, ws);
    const bool persist = !small && nchunks % 2 == 0;
#define TRON_F16_ARGS in, ws, bias, residual, out, pre_out, batch, cin, plane4, apply_mish, in_fmt, out_split, grad_absmax, n_absmax, st
#define TRON_F16_CASE(S_, PERSIST_OK_)                                                                                    \
    if (side == S_) {                                                                                                     \
        if (small) return launch<S_, 1, true, false>(TRON_F16_ARGS);                                                      \
        if (PERSIST_OK_ && persist)                                                                                       \
            return cout == 64 ? launch<S_, 2, false, PERSIST_OK_>(TRON_F16_ARGS) : launch<S_, 1, false, PERSIST_OK_>(TRON_F16_ARGS); \
        return cout == 64 ? launch<S_, 2, false, false>(TRON_F16_ARGS) : launch<S_, 1, false, false>(TRON_F16_ARGS);      \
    }
    TRON_F16_CASE(12, true)
    TRON_F16_CASE(26, true)
    TRON_F16_CASE(34, true)              // 32x32 boards (BASELINE config 5: the ACKTR nets' trunk, Net/ACNet.py)
#undef TRON_F16_CASE
#undef TRON_F16_ARGS
    return TRON_ERR_UNSUPPORTED;
}

// ---- the learner's backward: input gradient + the activation gradient of the layer below in one launch ---------------
extern "C" int64_t tron_conv3x3_dgrad_mish_workspace(int64_t batch, int32_t cin, int32_t cout, int32_t side)
{
    if (batch < 0 || (cin != 32 && cin != 64) || (cout != 32 && cout != 64) || (side != 12 && side != 26)) return 0;
    const int64_t wbytes = (tron_conv3x3_f16x3_workspace(cout, cin) + 255) / 256 * 256;
    return wbytes + 2 * MBWD_GRID_MAX * (THREADS / 64) * (cin / 2) * (int64_t)sizeof(float);
}

extern "C" int tron_conv3x3_dgrad_mish(const float *grad_pre, const float *weight, const float *grad_absmax, int32_t n_absmax,
                                       const float *extra_grad, const float *pre_below, float *grad_pre_below,
                                       float *bias_grad_below, float *absmax_below, int64_t batch, int32_t cin, int32_t cout,
                                       int32_t side, void *workspace, void *stream)
{
    if (!grad_pre || !weight || !pre_below || !grad_pre_below || !bias_grad_below || !absmax_below || !workspace || batch < 0 ||
        (grad_absmax && n_absmax < 1))
        return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(grad_pre) | reinterpret_cast<uintptr_t>(weight) | reinterpret_cast<uintptr_t>(extra_grad) |
         reinterpret_cast<uintptr_t>(pre_below) | reinterpret_cast<uintptr_t>(grad_pre_below) | reinterpret_cast<uintptr_t>(workspace)) & 15u)
        return TRON_ERR_BAD_ARG;
    if ((cin != 32 && cin != 64) || (cout != 32 && cout != 64) || (side != 12 && side != 26) || batch > (1ll << 24))
        return TRON_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (batch == 0) {
        if (hipMemsetAsync(bias_grad_below, 0, cin * sizeof(float), st) != hipSuccess ||
            hipMemsetAsync(absmax_below, 0, cin * sizeof(float), st) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_LAUNCH; }
        return TRON_OK;
    }
    // the kernel's view: a convolution from `cout` gradient channels to `cin` channels
    f16 *ws = reinterpret_cast<f16 *>(workspace);
    const int nchunks = cout / CIC;
    const int total = nchunks * TAPS_PAD * cin * CIC;
    float *stats = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(workspace) + (tron_conv3x3_f16x3_workspace(cout, cin) + 255) / 256 * 256);
    hipLaunchKernelGGL(k_split_weights, dim3((total + 255) / 256), dim3(256), 0, st, weight, cin, cout, nchunks, 1, ws);
    int rc = TRON_ERR_UNSUPPORTED;
    int64_t groups = 0;
#define TRON_MBWD_ARGS grad_pre, ws, nullptr, extra_grad, grad_pre_below, nullptr, batch, cout, 0.0f, 0, TRON_CONV_IN_F32, nullptr, grad_absmax, n_absmax, st, pre_below, stats, &groups, MBWD_GRID_MAX
    if (side == 12)
        rc = cin == 64 ? launch<12, 2, false, true, true>(TRON_MBWD_ARGS) : launch<12, 1, false, true, true>(TRON_MBWD_ARGS);
    else
        rc = cin == 64 ? launch<26, 2, false, true, true>(TRON_MBWD_ARGS) : launch<26, 1, false, true, true>(TRON_MBWD_ARGS);
#undef TRON_MBWD_ARGS
    if (rc != TRON_OK) return rc;
    hipLaunchKernelGGL(k_mbwd_finish, dim3(cin), dim3(256), 0, st, stats, (int)groups, cin / 32, bias_grad_below, absmax_below);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}
