// tron_conv_ws_kernel.hpp — the weight-stationary 3x3 convolution kernel (see tron_conv_ws.hip for the design) as a template
// that two translation units instantiate: tron_conv_ws.hip (MODE = WS_INFER: the gradient-free chain) and
// tron_conv_ws_train.hip (WS_TRAIN: the learner's forward, which also keeps every layer's pre-activation as a PX16 image;
// WS_BWD: the input gradient — the same loop on the 180-degree-rotated, transposed weights reading a gradient image — with
// the activation backward of the layer below, its bias sums and the scale of what it writes in the epilogue).  Everything
// here sits in an anonymous namespace: each translation unit gets its own copy (co-compiled instantiations perturb each
// other's register allocation, so the tuned inference variants keep a translation unit of their own).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/tron_hip.h"

// 34x34 images (32x32 boards: the actor-critic nets): the gradient-free forward's instantiations live beside the training ones in
// tron_conv_ws_train.hip (tron_conv_ws.hip keeps the tuned 12x12 / 26x26 inference variants to itself); tron_conv3x3_ws_fwd calls this
int ws_fwd_side34(const void *in_px16, const void *wfrag, const float *bias, const void *res_px16, void *out_px16, float *out_f32,
                  float *pre_f32, int64_t batch, int32_t cin, int32_t cout, int32_t apply_mish, hipStream_t st);

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#ifndef TRON_WS_POOL_ABLATE // diagnostic builds only (wrong results; scripts/ws_ablate.sh poolflags): 1 = WS_POOL without its pooling pass,
#define TRON_WS_POOL_ABLATE 0 // 2 = and without the barrier in front of it, 3 = and the epilogue's LDS write goes to the dump
#endif
#ifndef TRON_WS_ABLATE      // diagnostic builds only (wrong results; scripts/ws_ablate.sh): 1 = no epilogue arithmetic, 2 = B fragments
#define TRON_WS_ABLATE 0    // not re-read from LDS, 3 = no image DMA, 4 = no stores, 5 = no MFMAs, 6 = no item barrier
#endif
constexpr float ACT_SCALE = 1.0f / 64.0f, ACT_UNSCALE = 64.0f, LO_SCALE = 2048.0f, LO_UNSCALE = 1.0f / 2048.0f;

constexpr int align256(int x) { return (x + 255) & ~255; }

// Geometry of one instantiation.  S: image side.  R: image rows per work item (R == S: whole images; larger boards are
// cut into bands of at most R rows that bring one halo row above and below).  IPI: whole images per item.  WAVES: 4 (one
// wave per SIMD) or 8 (two: waves w and w + 4 share a SIMD and an M tile and split the pixels).  TPS: pixel tiles a wave
// carries through K together.
template <int S_, int R_, int CIN_, int COUT_, int IPI_, int WAVES_, int TPS_>
struct Geo {
    static constexpr int S = S_, R = R_, CIN = CIN_, COUT = COUT_, IPI = IPI_, WAVES = WAVES_, TPS = TPS_;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int NB = (S + R - 1) / R;                           // bands per image
    static constexpr int SS = S * S;
    static constexpr int ROWB = S * 16;                                  // bytes of one image row in an octet plane
    static constexpr int PLANE = align256((R + 2) * ROWB);               // LDS octet plane: halo row, R rows, halo row
    static constexpr int ZONE = align256(256 + 2 * ROWB);                // zeros in front of each 32-channel block
    static constexpr int CBLK = ZONE + 4 * PLANE;                        // one 32-channel block (4 octets)
    static constexpr int NCB = CIN / 32;
    static constexpr int HALF = NCB * CBLK;                              // hi (or lo) image
    static constexpr int IMG = 2 * HALF;
    static constexpr int BUF = (IPI * IMG + 1023) / 1024 * 1024;         // whole 1 KB DMA pieces
    static constexpr int NS = 9 * NCB;                                   // 32-deep k slabs: (tap, channel block)
    static constexpr int NCT = COUT / 16;                                // 16-channel M tiles
    static constexpr int PG = WAVES / NCT;                               // pixel groups: waves = NCT x PG
    static constexpr int NDMA = (BUF + 1023) / 1024;                     // 1 KB LDS-DMA pieces per item: the whole buffer, linearly
    static constexpr int DMA_PER_WAVE = (NDMA + WAVES - 1) / WAVES;
    static constexpr int PLANEG = SS * 16;                               // an octet plane of the PX16 image in memory
    static constexpr int HALFG_IN = (CIN / 8) * PLANEG, HALFG_OUT = (COUT / 8) * PLANEG;
    static constexpr int NST = 2 * TPS;                                  // stores per step: TPS tiles x (hi, lo)
    // LDS: two item buffers | per wave: residual pieces of two steps | per wave: the DMA source-offset table
    static constexpr int RES_OFF = 2 * BUF, RES_WAVE = 2 * TPS * 1024;
    static constexpr int TAB_OFF = RES_OFF + WAVES * RES_WAVE, TAB_WAVE = DMA_PER_WAVE * 128;       // (16-byte units in 16 bits)
    static constexpr int SINK_OFF = TAB_OFF + WAVES * TAB_WAVE;          // 1 KB where a piece past the buffer's end lands
    static constexpr size_t LDS_BYTES = (size_t)SINK_OFF + 1024;
    static_assert(NCT * PG == WAVES && NCT >= 1 && PG >= 1, "waves: M tiles x pixel groups");
    static_assert(CIN % 32 == 0 && COUT % 16 == 0, "channel blocks");
    static_assert(IPI == 1 || NB == 1, "several images or several bands, not both");
    static_assert(2 * ROWB + CBLK * (NCB - 1) + HALF + 16 <= 65536, "ds_read immediate offsets");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(NS % TPS == 0 || TPS == 1, "epilogue stages over the slabs");
};

// as tron_conv_f16.hip's: four values at once, e^x capped by an unsigned min (no select), one rcp
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

__device__ __forceinline__ void split(float v, f16 &hi, f16 &lo)
{
    hi = (f16)v;
    lo = (f16)((v - (float)hi) * LO_SCALE);
}

// ---- weights: W[cout][cin][3][3] f32 -> fragment image f16 [M tile][slab][hi | lo][lane][8] -------------------------
// Lane l of an A fragment holds row l & 15 (output channel 16 ct + row), k = 8 (l >> 4) + j: slab s = (tap s / NCB,
// channel block s % NCB), input channel 32 cb + 8 (l >> 4) + j.  A wave reads its fragments with 16-byte loads.
constexpr int WS_SPLIT_MAX = 8;
struct WsJobs {
    const float *w[WS_SPLIT_MAX];
    f16 *ws[WS_SPLIT_MAX];
    int cout[WS_SPLIT_MAX], cin[WS_SPLIT_MAX];
    int rot[WS_SPLIT_MAX];               // 1: the input gradient's weights — cout / cin are those of the BACKWARD convolution (the forward
};                                       //    layer's cin / cout) and W'[co][ci][tap] = W[ci][co][8 - tap] (transposed, rotated by 180 degrees)
__global__ void k_ws_split_weights(WsJobs jobs)
{
    const int k = blockIdx.y, cout = jobs.cout[k], cin = jobs.cin[k], ncb = cin / 32, ns = 9 * ncb;
    const float *__restrict__ w = jobs.w[k];
    f16 *__restrict__ ws = jobs.ws[k];
    const int total = (cout / 16) * ns * 64 * 8;                        // (tile, slab, lane, j)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i & 7, lane = (i >> 3) & 63, r = i >> 9;
        const int s = r % ns, ct = r / ns;
        const int tap = s / ncb, cb = s - tap * ncb;
        const int co = 16 * ct + (lane & 15), ci = 32 * cb + 8 * (lane >> 4) + j;
        f16 h, l;
        split(jobs.rot[k] ? w[((size_t)ci * cout + co) * 9 + (8 - tap)] : w[((size_t)co * cin + ci) * 9 + tap], h, l);
        const size_t o = ((size_t)(ct * ns + s) * 2) * 512 + lane * 8 + j;
        ws[o] = h;
        ws[o + 512] = l;
    }
}

// where predicated-off stores go: every step issues the same number of vector-memory operations, so that the counted
// s_waitcnt vmcnt(N) below are exact (N must not exceed the number of younger operations actually issued)
__device__ unsigned char g_ws_dump[152 * 1024];                          // (as large as the largest hi-to-lo image distance: the lo half lands inside it too)
__device__ uint4 g_ws_zero[64];                                          // 1 KB of zeros: what a DMA lane copies into halo rows, zones and plane padding

#ifdef TRON_WS_STAMPS        // diagnostic build only (scripts/ws_stamps.py): per-wave cycle counts, read back by tron_conv_ws_stamps
__device__ unsigned long long g_ws_stamps[256 * 12 * 8];         // [workgroup][wave <= 12][8]
#endif

// One step = TPS pixel tiles of 16 pixels through all of K, with the PREVIOUS step's epilogue (bias, residual, mish,
// split, stores) spread over this step's slabs, one LDS-DMA piece of the NEXT item per slab, and the B fragments of
// the next slab requested tile by tile: the whole step is one basic block.
//
// MODE (the epilogue; the loop is the same):
//   WS_INFER  out = act(conv(in) + bias + res)                                      (the gradient-free chain)
//   WS_TRAIN  the same, and the pre-activation z = conv(in) + bias + res is kept as a second PX16 image `pre_px`: what the
//             backward pass multiplies by mish'(z) (DDQN.py:148 -> DQNNet.py:37-48)
//   WS_BWD    `in` is a GRADIENT image (PX16 of g * s_in, s_in a power of two kept beside the image), the weights are the
//             forward layer's rotated and transposed (k_ws_split_weights, rot), and
//                 out = (conv(in) / s_in + res / s_res) * mish'(zb)   written as PX16 of out * s_out
//             — res the gradient that reaches the same tensor along a residual connection, zb the pre-activation of the
//             layer below — with the per-channel sums of out (that layer's bias gradient) and the largest |out| left in
//             `stats` per workgroup (k_wsb_finish adds them up in a fixed order).  F32OUT also writes out as f32 NCHW.
//   WS_POOL   WS_INFER whose output never leaves the CU as an image: the epilogue writes the item's PX16 values into an LDS
//             image O ([channel quad][pixel][4 values: hi + lo 2^-11 as f32]); once an item's last epilogue is in (flushed at the item's end),
//             the workgroup average-pools it (3x3 / stride 2 / pad 1, DQNNet.py:20,52 — the arithmetic of tron_head.hip's
//             px16_window_sum on the same values, so the result has the bits of the two-kernel path) and stores the
//             pooled rows the head's conv7 GEMM reads: `out` = hi rows, `pre_px` = lo rows, [B][(octet, pooled pixel, channel)].
//             Whole 12x12 images per item only (tron_conv_ws_pool.hip).
//   WS_POOL_TRAIN  WS_TRAIN the same way (the learner's forward, DDQN.py:127: conv6's output feeds nothing but the pooling): the
//             pre-activation image is written as in WS_TRAIN, the output is pooled in LDS and leaves as the f32 planes
//             [B][64][6 x 6] of tron_pool12_px16 (`out`), with its bits.
enum { WS_INFER = 0, WS_TRAIN = 1, WS_BWD = 2, WS_POOL = 3, WS_POOL_TRAIN = 4 };
// A gradient image's device record: info = {s, 1 / s, -, -, max |g| per channel [<= 64]} (include/tron_hip.h).
struct WsBwd {
    const unsigned char *zb;     // PX16 pre-activation of the layer below [B][COUT][S][S] (COUT = this launch's output channels)
    const float *in_info;        // record of `in`
    const float *res_info;       // record of `res` (RES)
    const float *wnorm;          // largest absolute row sum of the rotated weights: max |conv(in)| <= max |in| * wnorm
    float *out_info;             // record of `out`: {s_out, 1 / s_out} are written here (the maxima: k_wsb_finish)
    float *stats;                // [2][gridDim.x][COUT]: column sums | column maxima of |out|
};
__device__ __forceinline__ float pow2_at_most(float x)                   // the largest power of two <= x (x > 0, finite); exponent clamped to +-60
{
    int e = (int)((__float_as_uint(x) >> 23) & 0xFFu) - 127;
    e = e < -60 ? -60 : (e > 60 ? 60 : e);
    return __uint_as_float((uint32_t)(e + 127) << 23);
}

template <class G, bool RES, bool F32OUT, int MODE = WS_INFER>
__global__ __launch_bounds__(G::THREADS, G::WAVES / 4) void k_conv_ws(
    const unsigned char *__restrict__ in, const f16x8 *__restrict__ wfrag, const float *__restrict__ bias,
    const unsigned char *__restrict__ res, unsigned char *__restrict__ out, float *__restrict__ out_f32,
    float *__restrict__ pre_f32, int B, int apply_mish, int nitems, unsigned char *__restrict__ pre_px, WsBwd bw)
{
    constexpr bool TRAIN = MODE == WS_TRAIN || MODE == WS_POOL_TRAIN, BWD = MODE == WS_BWD, POOL = MODE == WS_POOL || MODE == WS_POOL_TRAIN;
    static_assert(!BWD || G::TPS == 1, "the gradient epilogue is written for one tile per step");
    static_assert(!POOL || (G::S == 12 && G::R == 12 && G::IPI == 1 && G::TPS == 1 && G::COUT == 64 && !F32OUT), "pooled output: whole 12x12 images");
    constexpr int S = G::S, NS = G::NS, NCB = G::NCB, TPS = G::TPS, THREADS = G::THREADS;
    constexpr int NST = G::NST * ((POOL ? 0 : 1) + (TRAIN ? 1 : 0));     // stores per step: (hi, lo) of the output (not POOL), and of the pre-activation
    // (POOL) the LDS output image behind the sink: quad q of pixel p at O_OFF + q O_QS + 16 p; lanes without a pixel write O_DUMP
    constexpr int O_OFF = G::SINK_OFF + 1024, O_QS = G::SS * 16 + 16, O_DUMP = O_OFF + (G::COUT / 4) * O_QS;
    constexpr int O_FLAG = O_DUMP + 1024;                                // a word per wave: pooling passes it has finished
    extern __shared__ __attribute__((aligned(256))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int ct = wave % G::NCT, pg = wave / G::NCT;                    // (8 waves: w and w + 4 share a SIMD: same M tile, other pixels)

    // this wave's weights: all of K for its 16 output channels, hi and lo fragments
    f16x8 wh[NS], wl[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        wh[s] = wfrag[((ct * NS + s) * 2 + 0) * 64 + lane];
        wl[s] = wfrag[((ct * NS + s) * 2 + 1) * 64 + lane];
    }
    // D row = 4 (lane >> 4) + r: this lane's four output channels 16 ct + 4 g + r, at pixel lane & 15 of a tile
    const int co0 = 16 * ct + 4 * g;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (!BWD && bias) bv = *reinterpret_cast<const f32x4 *>(bias + co0);
    float inv_in = 1.0f, inv_res = 1.0f, s_out = 1.0f;                  // (BWD) the images' scales
    if (BWD) {
        // the scale of what this launch writes, from a bound on it (every wave computes the same number): max |conv(in) + res| <=
        // max |in| x wnorm + max |res|, times |mish'| <= 1.1, mapped to at most 2^15 by a power of two
        float m_in = lane < G::CIN ? bw.in_info[4 + lane] : 0.0f, m_res = (RES && lane < G::COUT) ? bw.res_info[4 + lane] : 0.0f;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            m_in = __builtin_fmaxf(m_in, __shfl_xor(m_in, d, 64));
            m_res = __builtin_fmaxf(m_res, __shfl_xor(m_res, d, 64));
        }
        const float bound = (m_in * bw.wnorm[0] + m_res) * 1.1f;
        const float sc = (bound > 0.0f && bound < 3.0e38f) ? pow2_at_most(32768.0f / bound) : 1.0f;
        inv_in = bw.in_info[1] * ACT_UNSCALE;
        inv_res = RES ? bw.res_info[1] * ACT_UNSCALE : 0.0f;
        s_out = sc * ACT_SCALE;
        if (blockIdx.x == 0 && tid == 0) {
            bw.out_info[0] = sc;
            bw.out_info[1] = 1.0f / sc;
        }
    }
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};                                   // (BWD) this lane's channels: sums of what it wrote,
    float cmax = 0.0f;                                                   //       and the largest magnitude among them

    for (int i = tid; i < (BWD ? G::TAB_OFF : 2 * G::BUF) / 16; i += THREADS) reinterpret_cast<uint4 *>(lds)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (POOL && tid < G::WAVES) reinterpret_cast<int *>(lds + O_FLAG)[tid] = 0;

    // the band this workgroup serves (the grid is a multiple of NB, so it is always the same one)
    const int band = blockIdx.x % G::NB;
    const int r0 = band * G::R, rows_b = (S - r0 < G::R) ? S - r0 : G::R;
    const int npx = G::IPI * rows_b * S;                                 // pixels of an item
    const int ntiles = (npx + 15) >> 4;
    const int my_nt = pg < ntiles ? (ntiles - pg + G::PG - 1) / G::PG : 0;
    const int nsteps = (my_nt + TPS - 1) / TPS;
    unsigned char *res_stage = lds + G::RES_OFF + wave * G::RES_WAVE;    // this wave's residual pieces, two steps deep
    uint16_t *dma_tab = reinterpret_cast<uint16_t *>(lds + G::TAB_OFF + wave * G::TAB_WAVE);
    static_assert(2 * G::HALFG_IN * G::IPI / 16 < 0xFFFF, "source offsets in 16 bits");

    // An item goes global -> LDS as NDMA linear 1 KB pieces of the LDS buffer (LDS-DMA writes base + 16 * lane; the
    // SOURCE is per lane): a lane whose 16 destination bytes are image data copies them from the PX16 image, a lane
    // whose destination is a halo row outside the image, a zero zone or plane padding copies zeros — no partial
    // pieces, no branches, and the buffer's zeros are rewritten with every item.  The source offset of (piece, lane)
    // does not depend on the item: computed once, kept in LDS (piece j of this wave = piece wave + WAVES * j).
    for (int j = 0; j < G::DMA_PER_WAVE; ++j) {
        const int d = (wave + G::WAVES * j) * 1024 + lane * 16;          // byte of the item buffer
        int off = -1;
        if (d < G::IPI * G::IMG) {
            const int il = d / G::IMG, d1 = d - il * G::IMG;
            const int h = d1 / G::HALF, d2 = d1 - h * G::HALF;
            const int cb = d2 / G::CBLK, d3 = d2 - cb * G::CBLK - G::ZONE;
            if (d3 >= 0) {
                const int o4 = d3 / G::PLANE, d4 = d3 - o4 * G::PLANE;
                const int row = d4 / G::ROWB, irow = r0 - 1 + row;       // LDS row (0 = halo above) -> image row
                if (row < rows_b + 2 && irow >= 0 && irow < S)
                    off = (il * 2 + h) * G::HALFG_IN + (4 * cb + o4) * G::PLANEG + irow * G::ROWB + (d4 - row * G::ROWB);
            }
        }
        dma_tab[j * 64 + lane] = (uint16_t)(off < 0 ? 0xFFFF : off >> 4);
    }

    auto dma_piece = [&](int item, int buf, int j) {                     // (j < DMA_PER_WAVE; a piece past the buffer's end: zeros into the dump)
        const int q = wave + G::WAVES * j;
        const uint32_t off16 = dma_tab[j * 64 + lane];
        const int off = off16 == 0xFFFFu ? -1 : (int)(off16 << 4);
        const int ip = item / G::NB;
        const int il = (G::IPI > 1 && off >= 2 * G::HALFG_IN) ? 1 : 0;   // (the table's offset counts from the item's first image)
        int img = ip * G::IPI + il;
        img = img < B ? img : B - 1;                                     // the second image of the last pair may not exist: any valid one
        const unsigned char *src = off >= 0 ? in + (size_t)img * 2 * G::HALFG_IN + (off - il * 2 * G::HALFG_IN)
                                            : reinterpret_cast<const unsigned char *>(g_ws_zero) + lane * 16;
        unsigned char *dst = q < G::NDMA ? lds + buf * G::BUF + q * 1024 : lds + G::SINK_OFF;
        if (TRON_WS_ABLATE != 3)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    };
    // the next item's pieces ride on the slabs of this item's steps 0 .. nsteps - 2 (one per slab), so that the last
    // step's NST stores are younger than every piece: the counted wait at the item boundary then covers the pieces
    const bool dma_in_loop = nsteps >= 2 && (nsteps - 1) * NS >= G::DMA_PER_WAVE;

    // per-lane LDS byte offsets of tile t's B fragments for the three horizontal taps (vertical tap, channel block and
    // half are immediates on top).  A lane whose tap falls outside the row reads the zero zone, at the 16-byte slot its
    // real address would have had (the read stays conflict-free).
    auto bases = [&](int t, int bufoff, int (&b)[3]) {
        int o = 16 * t + li;
        o = o < npx ? o : npx - 1;
        const int il = G::IPI > 1 ? (o >= G::SS ? 1 : 0) : 0, rem = o - il * G::SS;
        const int y = rem / S, x = rem - y * S;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int xx = x + kx - 1, pp = rem + kx - 1;
            const bool ok = xx >= 0 && xx < S;
            b[kx] = bufoff + il * G::IMG + (ok ? G::ZONE + g * G::PLANE + pp * 16 : (pp & 15) * 16);
        }
    };

    // ---- the state of the step whose epilogue is pending --------------------------------------------------------------
    f32x4 pv[TPS], pe[TPS];                                              // value; e^x, then n / (n + 2)
    f32x4 pz[BWD ? TPS : 1];                                             // (BWD) the layer below's pre-activation at the pending tile
    f16x4 phh[TPS];
    unsigned char *pp[TPS];                                              // where this lane's 8 bytes of the hi image go
    unsigned char *ppz[TPS];                                             // (TRAIN) ... and of the pre-activation's hi image
    bool pok[TPS];                                                       // (BWD) the pending tile's pixel exists: it counts in the sums
    int64_t pfo[TPS];                                                    // (F32OUT) element offset of its four f32 values, or -1
    int pt[TPS], p_ip = 0, p_par = 0;                                    // tile indices, image (pair) index, residual-stage parity
    int plo[TPS];                                                        // (POOL) where this lane's 16 bytes go in the LDS output image
    bool pend_b = false;                                                 // (POOL) this wave has yet to see the last pooling pass finished (below)
    int npass = 0;                                                       // (POOL) pooling passes started so far
#pragma unroll
    for (int k = 0; k < TPS; ++k) {
        plo[k] = O_DUMP + lane * 16;
        pv[k] = pe[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (BWD) pz[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        phh[k] = (f16x4){0, 0, 0, 0};
        pp[k] = ppz[k] = g_ws_dump + lane * 16;
        pok[k] = false;
        pfo[k] = -1;
        pt[k] = -1;
    }

    // image and in-image pixel of (tile t, pixel pl of the tile); false if the tile / pixel / image does not exist
    auto locate = [&](int t, int pl, int ip, int &img, int &pixg) {
        const int o = 16 * t + pl;
        const int oc = (t >= 0 && o < npx) ? o : 0;
        const int il = G::IPI > 1 ? (oc >= G::SS ? 1 : 0) : 0;
        img = ip * G::IPI + il;
        pixg = r0 * S + oc - il * G::SS;
        const bool ok = t >= 0 && t < ntiles && o < npx && img < B;
        if (!ok) img = 0, pixg = 0;
        return ok;
    };

    // the epilogue of pending tile k, cut into six stages that ride on consecutive slabs of the next step
    auto epi = [&](int k, int stage) {
        if (TRON_WS_ABLATE == 1 && stage >= 1 && stage <= 4) return;
        if (stage == 0) {
            int img, pixg;
            const bool ok = locate(pt[k], li, p_ip, img, pixg);
            const uint32_t poff = (uint32_t)((2 * ct + (g >> 1)) * G::PLANEG + pixg * 16 + (g & 1) * 8);
            unsigned char *po = out + (size_t)img * 2 * G::HALFG_OUT + poff;      // (img, pixg are 0 when !ok: a valid address either way)
            pp[k] = (ok && out && TRON_WS_ABLATE != 7) ? po : g_ws_dump + lane * 16;
            asm volatile("" : "+v"(pp[k]));
            if (POOL) {
                plo[k] = (ok && TRON_WS_POOL_ABLATE < 3) ? O_OFF + (4 * ct + g) * O_QS + pixg * 16 : O_DUMP + lane * 16;
                asm volatile("" : "+v"(plo[k]));
            }
            if (TRAIN) {
                unsigned char *pz = pre_px + (size_t)img * 2 * G::HALFG_OUT + poff;
                ppz[k] = (ok && pre_px) ? pz : g_ws_dump + lane * 16;
                asm volatile("" : "+v"(ppz[k]));
            }
            if (F32OUT) pfo[k] = ok ? ((int64_t)img * G::COUT + co0) * G::SS + pixg : -1;
            if (BWD) {
                // the staged pieces [z of the layer below | res]: this lane's 8 + 8 bytes of each, [hi | lo][octet of the pair][pixel][16 B]
                pok[k] = ok;
                const unsigned char *zs = res_stage + k * 1024 + (g >> 1) * 256 + li * 16 + (g & 1) * 8;
                const f16x4 zh = *reinterpret_cast<const f16x4 *>(zs), zl = *reinterpret_cast<const f16x4 *>(zs + 512);
                pz[k] = (__builtin_convertvector(zh, f32x4) + __builtin_convertvector(zl, f32x4) * LO_UNSCALE) * ACT_UNSCALE;
                if (RES) {
                    const unsigned char *rs = res_stage + (TPS + k) * 1024 + (g >> 1) * 256 + li * 16 + (g & 1) * 8;
                    const f16x4 rh = *reinterpret_cast<const f16x4 *>(rs), rl = *reinterpret_cast<const f16x4 *>(rs + 512);
                    pv[k] += (__builtin_convertvector(rh, f32x4) + __builtin_convertvector(rl, f32x4) * LO_UNSCALE) * inv_res;
                }
            } else if (RES) {
                // this lane's 8 + 8 residual bytes from the wave's staged piece: [hi | lo][octet of the pair][pixel][16 B]
                const unsigned char *rs = res_stage + (p_par * TPS + k) * 1024 + (g >> 1) * 256 + li * 16 + (g & 1) * 8;
                const f16x4 rh = *reinterpret_cast<const f16x4 *>(rs), rl = *reinterpret_cast<const f16x4 *>(rs + 512);
                pv[k] += (__builtin_convertvector(rh, f32x4) + __builtin_convertvector(rl, f32x4) * LO_UNSCALE) * ACT_UNSCALE;
            }
        } else if (stage == 1) {
            if (F32OUT && pre_f32 && pfo[k] >= 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pre_f32[pfo[k] + r * G::SS] = pv[k][r];
            }
            if (TRAIN) {                                                  // the pre-activation as a PX16 image of its own
                const f32x4 zs = pv[k] * ACT_SCALE;
                const f16x4 zh = __builtin_convertvector(zs, f16x4);
                const f16x4 zl = __builtin_convertvector((zs - __builtin_convertvector(zh, f32x4)) * LO_SCALE, f16x4);
                *reinterpret_cast<f16x4 *>(ppz[k]) = zh;
                *reinterpret_cast<f16x4 *>(ppz[k] + G::HALFG_OUT) = zl;
            }
            const f32x4 t = (BWD ? pz[k] : pv[k]) * 1.44269504088896341f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t b = __float_as_uint(__builtin_amdgcn_exp2f(t[i]));
                pe[k][i] = __uint_as_float(b < 0x5D5E0B6Bu ? b : 0x5D5E0B6Bu);      // min(e^x, 1e18): see tron_conv_f16.hip
            }
            asm volatile("" : "+v"(pe[k]));                              // (opaque: the stage stays on its slab instead of sinking to its use)
        } else if (stage == 2) {
            const f32x4 e = pe[k];
            const f32x4 n = __builtin_elementwise_fma(e, e, e + e);
            const f32x4 d = n + 2.0f;
            const f32x4 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]), __builtin_amdgcn_rcpf(d[2]), __builtin_amdgcn_rcpf(d[3])};
            if (BWD) {
                // mish'(z) = t + z (1 - t^2) e / (1 + e), t = n / (n + 2), 1 - t^2 = (2 / (n + 2)) (1 + t)   (tron_nn.hip; e capped at
                // 1e18: t = 1 and the second term 0 for large z, t = 0 and e / (1 + e) = 0 for very negative z)
                const f32x4 t = n * r, e1 = e + 1.0f;
                const f32x4 q = {__builtin_amdgcn_rcpf(e1[0]), __builtin_amdgcn_rcpf(e1[1]), __builtin_amdgcn_rcpf(e1[2]), __builtin_amdgcn_rcpf(e1[3])};
                pe[k] = t + pz[k] * ((r + r) * (t + 1.0f)) * (e * q);
            } else {
                pe[k] = n * r;
            }
            asm volatile("" : "+v"(pe[k]));
        } else if (stage == 3 && BWD) {
            f32x4 gv = pv[k] * pe[k];
#pragma unroll
            for (int r = 0; r < 4; ++r) gv[r] = pok[k] ? gv[r] : 0.0f;     // (a select, not a product: whatever sits in a missing tile's registers stays out of the sums)
            csum += gv;
            cmax = __builtin_fmaxf(cmax, __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(gv[0]), __builtin_fabsf(gv[1])),
                                                         __builtin_fmaxf(__builtin_fabsf(gv[2]), __builtin_fabsf(gv[3]))));
            if (F32OUT && out_f32 && pfo[k] >= 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) out_f32[pfo[k] + r * G::SS] = gv[r];
            }
            pv[k] = gv * s_out;
            asm volatile("" : "+v"(pv[k]));
        } else if (stage == 3) {
            if (!F32OUT || apply_mish) pv[k] = pv[k] * pe[k];           // (the fast variant is only launched with the activation on)
            if (F32OUT && out_f32 && pfo[k] >= 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) out_f32[pfo[k] + r * G::SS] = pv[k][r];
            }
            pv[k] = pv[k] * ACT_SCALE;
            asm volatile("" : "+v"(pv[k]));
        } else if (stage == 4) {
            phh[k] = __builtin_convertvector(pv[k], f16x4);
            pv[k] = (pv[k] - __builtin_convertvector(phh[k], f32x4)) * LO_SCALE;
            asm volatile("" : "+v"(pv[k]), "+v"(phh[k]));
        } else {
            const f16x4 ll = __builtin_convertvector(pv[k], f16x4);
            if (TRON_WS_ABLATE == 4) { asm volatile("" ::"v"(ll), "v"(phh[k]), "v"(pp[k])); return; }
            if (POOL) {                                                  // the value the PX16 image would have carried: hi + lo 2^-11
                if (pend_b) {                                            // (the previous image's pooling pass must be over: its waves say so)
                    for (;;) {
                        const volatile __attribute__((address_space(3))) int *f = (const volatile __attribute__((address_space(3))) int *)(lds + O_FLAG);
                        int done = f[0];
#pragma unroll
                        for (int w = 1; w < G::WAVES; ++w) { const int d = f[w]; done = d < done ? d : done; }
                        if (__builtin_amdgcn_readfirstlane(done) >= npass) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    pend_b = false;
                }
                *reinterpret_cast<f32x4 *>(lds + plo[k]) = __builtin_convertvector(phh[k], f32x4) + __builtin_convertvector(ll, f32x4) * LO_UNSCALE;
                return;
            }
            *reinterpret_cast<f16x4 *>(pp[k]) = phh[k];
            *reinterpret_cast<f16x4 *>(pp[k] + G::HALFG_OUT) = ll;
        }
    };
    // slab -> (pending tile, stage): six stages per tile over NS / TPS slabs
    auto epi_at_slab = [&](int s) {
        constexpr int PER = NS / TPS;                                    // slabs per pending tile: 18, 9, 6 or 3
        const int k = s / PER, j = s % PER;
        if (k >= TPS) return;
        if (PER >= 6) {
            constexpr int SP = PER / 6;                                  // a stage every SP slabs
            if (j % SP == 0 && j / SP < 6) epi(k, j / SP);
        } else {
            epi(k, 2 * j);
            epi(k, 2 * j + 1);
        }
    };

    // (POOL) image `img`'s pooled rows from the LDS output image.  A unit = (channel quad, pooled pixel): four channels, the window's
    // nine values added in px16_window_sum's order, / 9, split.  12 is even, so only the row above and the column left of the image
    // are padding: such a tap reads 16 bytes of zeros (the zero zone at the start of LDS) instead, every other tap is the unit's
    // base + a constant.  576 units: one per thread, and the 64 left over go to the first wave of the last pixel group (which has a
    // tile fewer per item than the first group's).  What a thread needs of its units is computed once per launch — the pass costs
    // vector issue slots and LDS round trips, the scarce things in this kernel (with the indices worked out in place a unit was 130
    // instructions; three units one after the other in the waves of one pixel group left the other group waiting at the item's end).
    constexpr int PUNITS = 2;
    int pbase[POOL ? PUNITS : 1], pout[POOL ? PUNITS : 1];               // LDS base of the window; output byte offset in the row | top | left << 1, or -1
    if (POOL) {
#pragma unroll
        for (int k = 0; k < PUNITS; ++k) {
            const int u = k == 0 ? tid : THREADS + tid - (G::PG - 1) * 64 * G::NCT;
            const bool mine = k == 0 || (wave == (G::PG - 1) * G::NCT);
            const int q = u / 36, pq = u - q * 36, py = pq / 6, px = pq - py * 6;
            pbase[k] = O_OFF + q * O_QS + ((2 * py - 1) * S + (2 * px - 1)) * 16;
            const int obytes = TRAIN ? (4 * q * 36 + pq) * 4 : ((q >> 1) * 36 + pq) * 16 + (q & 1) * 8;     // f32 planes | split rows
            pout[k] = (mine && u >= 0 && u < 16 * 36) ? (obytes | (py == 0 ? 1 : 0) | (px == 0 ? 2 : 0)) : -1;
        }
        static_assert(!POOL || (THREADS == 512 && G::NCT == 4 && G::PG == 2), "the units' deal");
    }
    auto pool_pass = [&](int img) {
        unsigned char *oh = out + (size_t)img * (TRAIN ? 9216 : 4608), *ol = pre_px + (size_t)img * 4608;   // (TRAIN: `out` = f32 [B][64][36])
#pragma unroll
        for (int k = 0; k < PUNITS; ++k) {
            if (pout[k] < 0) continue;
            const int b = pbase[k];
            const bool top = pout[k] & 1, left = pout[k] & 2;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3, dx = t % 3, c = (dy * S + dx) * 16;
                const bool pad = (dy == 0 && top) || (dx == 0 && left);
                acc += *reinterpret_cast<const f32x4 *>(lds + ((dy > 0 && dx > 0) ? b + c : (pad ? 0 : b + c)));
            }
            if (TRAIN) {                                                 // k_pool12_from_px's planes: the image carries value / 64
                float *po = reinterpret_cast<float *>(oh + (pout[k] & ~3));
#pragma unroll
                for (int r = 0; r < 4; ++r) po[r * 36] = acc[r] * (64.0f / 9.0f);
                continue;
            }
            const f32x4 sum = acc * (1.0f / 9.0f);
            const f16x4 hh = __builtin_convertvector(sum, f16x4);
            const f16x4 ll = __builtin_convertvector((sum - __builtin_convertvector(hh, f32x4)) * LO_SCALE, f16x4);
            *reinterpret_cast<f16x4 *>(oh + (pout[k] & ~3)) = hh;
            *reinterpret_cast<f16x4 *>(ol + (pout[k] & ~3)) = ll;
        }
    };
    int pooled_img = -1;                                                 // (POOL) the image whose output sits in O, waiting for its pass

    __builtin_amdgcn_s_waitcnt(0);                                       // (the builtin, not asm: the compiler's own wait insertion then
    __syncthreads();                                                     //  knows the weight loads are done and adds no vmcnt(0) in the loop)
    int item = blockIdx.x, cur = 0, par = 0;
#ifdef TRON_WS_STAMPS
    unsigned long long st_wait = 0, st_bar = 0, st_dma = 0, st_steps = 0, st_items = 0, st_res = 0;
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (item < nitems)
        for (int j = 0; j < G::DMA_PER_WAVE; ++j) dma_piece(item, 0, j);
    bool first = true;
    for (; item < nitems; item += (int)gridDim.x, cur ^= 1) {
        // this wave's pieces of `item` (and the residual pieces of the pending step) have landed: everything but the
        // pending epilogue's last NST stores, which are the youngest operations in flight
#ifdef TRON_WS_STAMPS
        const unsigned long long st_a = __builtin_amdgcn_s_memtime();
#endif
        if (POOL) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (and this wave's writes of the previous item's output image)
        else if (first || F32OUT || !dma_in_loop) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
        first = false;
#ifdef TRON_WS_STAMPS
        const unsigned long long st_b = __builtin_amdgcn_s_memtime();
#endif
        if (TRON_WS_ABLATE != 6) asm volatile("s_barrier" ::: "memory");  // ... everybody's have; the other buffer is free
#ifdef TRON_WS_STAMPS
        const unsigned long long st_c = __builtin_amdgcn_s_memtime();
#endif
        const int nxt = item + (int)gridDim.x;
        if (nxt < nitems && !dma_in_loop)
            for (int j = 0; j < G::DMA_PER_WAVE; ++j) dma_piece(nxt, cur ^ 1, j);
#ifdef TRON_WS_STAMPS
        const unsigned long long st_d = __builtin_amdgcn_s_memtime();
        st_wait += st_b - st_a; st_bar += st_c - st_b; st_dma += st_d - st_c; st_items += 1;
#endif
        const int ip = item / G::NB;
        const int bufoff = cur * G::BUF;
        for (int st = 0; st < nsteps; ++st, par ^= 1) {
            int tl[TPS], b[TPS][3];
#pragma unroll
            for (int k = 0; k < TPS; ++k) {
                const int t = pg + (st * TPS + k) * G::PG;
                tl[k] = t < ntiles ? t : -1;
                bases(t < ntiles ? t : ntiles - 1, bufoff, b[k]);
            }
            // BWD: every step stages the layer below's pre-activation (and the residual gradient) at its tiles, into ONE set of
            // slots per wave — [z | res] — instead of two by step parity (LDS: the 64-channel 26x26 variant has no room
            // for four pieces per wave): the pieces of this step go out right after the pending epilogue's stage 0 has
            // read the previous ones (slab 0 below; a read issued before an LDS-DMA returns the old bytes)
            auto bwd_pieces = [&]() {
#pragma unroll
                for (int k = 0; k < TPS; ++k) {
                    int img, pixg;
                    locate(tl[k], lane & 15, ip, img, pixg);
                    const size_t off = ((size_t)img * 2 + (lane >> 5)) * G::HALFG_OUT + (2 * ct + ((lane >> 4) & 1)) * G::PLANEG + pixg * 16;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(bw.zb + off),
                                                     (__attribute__((address_space(3))) void *)(res_stage + k * 1024), 16, 0, 0);
                    if (RES)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(res + off),
                                                         (__attribute__((address_space(3))) void *)(res_stage + (TPS + k) * 1024), 16, 0, 0);
                }
            };
            if (BWD) {
                if (st > 0) {                                            // the pending step's pieces: all but its NST stores
                    if (F32OUT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
                }
            } else if (RES) {
                if (st > 0) {                                            // the pending step's residual pieces: all but its NST stores
                    if (F32OUT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
                }
                // this step's residual: one 1 KB piece per tile, lane -> (half, octet of this wave's pair, pixel)
#pragma unroll
                for (int k = 0; k < TPS; ++k) {
                    int img, pixg;
                    locate(tl[k], lane & 15, ip, img, pixg);
                    const unsigned char *src = res + ((size_t)img * 2 + (lane >> 5)) * G::HALFG_OUT +
                                               (2 * ct + ((lane >> 4) & 1)) * G::PLANEG + pixg * 16;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(res_stage + (par * TPS + k) * 1024), 16, 0, 0);
                }
            }
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            f32x4 ah[TPS], ax[TPS], ay[TPS];
#pragma unroll
            for (int k = 0; k < TPS; ++k) ah[k] = ax[k] = ay[k] = z;
            // B fragments AHEAD slabs ahead, tile by tile: tile k's fragments of slab s + AHEAD are requested right after
            // its MFMAs of slab s were issued.  lgkmcnt counts to 15 (with more in flight the compiler waits for ALL), so
            // 2 TPS AHEAD reads stay below that: three slabs ahead with one tile per step (a slab is then only three
            // MFMAs, 48 cycles — less than an LDS round trip), one slab ahead with three tiles.
            constexpr int AHEAD = 1, RING = AHEAD + 1;                  // (three slabs ahead measured no faster: the LDS latency is not what the loop waits for)
            static_assert(2 * TPS * AHEAD + 2 * TPS <= 15, "LDS reads in flight");
            f16x8 rbh[RING][TPS], rbl[RING][TPS];
            auto fetch = [&](int s, int k) {
                const int tap = s / NCB, cb = s % NCB, ky = tap / 3, kx = tap % 3;
                const int imm = ky * G::ROWB + cb * G::CBLK;
                rbh[s % RING][k] = *reinterpret_cast<const f16x8 *>(lds + b[k][kx] + imm);
                rbl[s % RING][k] = *reinterpret_cast<const f16x8 *>(lds + b[k][kx] + imm + G::HALF);
            };
#pragma unroll
            for (int s = 0; s < AHEAD; ++s)
#pragma unroll
                for (int k = 0; k < TPS; ++k) fetch(s, k);
            // which of the next item's pieces this step carries: piece j0 + s / DSTRIDE at slab s (wave-uniform: one
            // branch per step, none per slab — a branch around a DMA makes the compiler drain lgkmcnt at the join)
            const int j0 = (dma_in_loop && nxt < nitems && st < nsteps - 1) ? st * NS : G::DMA_PER_WAVE;
            auto kloop = [&](auto with_dma) {
#pragma unroll
                for (int s = 0; s < NS; ++s) {
#pragma unroll
                    for (int k = 0; k < TPS; ++k) {
                        if (s + AHEAD < NS && TRON_WS_ABLATE != 2) fetch(s + AHEAD, k);
                        if (TRON_WS_ABLATE != 5) {
                            ax[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], rbl[s % RING][k], ax[k], 0, 0, 0);
                            ah[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], rbh[s % RING][k], ah[k], 0, 0, 0);
                            ay[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s], rbh[s % RING][k], ay[k], 0, 0, 0);
                        } else {
                            asm volatile("" ::"v"(rbl[s % RING][k]), "v"(rbh[s % RING][k]));
                        }
                    }
                    epi_at_slab(s);                                      // a stage of one pending tile's epilogue
                    if (BWD && s == 0) bwd_pieces();                     // (after stage 0's reads of the previous pieces)
                    if (decltype(with_dma)::value && s < G::DMA_PER_WAVE) dma_piece(nxt, cur ^ 1, j0 + s < G::DMA_PER_WAVE ? j0 + s : G::DMA_PER_WAVE - 1);
                    // the order inside the slab: per tile two LDS reads, then its three MFMAs each followed by a few of
                    // the other vector instructions — an MFMA leaves 8 of its 16 cycles of vector issue free
#pragma unroll
                    for (int k = 0; k < TPS; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                        for (int m = 0; m < 3; ++m) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x006, TPS == 1 ? 6 : 3, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (j0 < G::DMA_PER_WAVE) kloop(std::true_type{}); else kloop(std::false_type{});
            // this step becomes the pending one
#pragma unroll
            for (int k = 0; k < TPS; ++k) {
                if (BWD) pv[k] = (ah[k] + (ax[k] + ay[k]) * LO_UNSCALE) * inv_in;
                else pv[k] = (ah[k] + (ax[k] + ay[k]) * LO_UNSCALE) * ACT_UNSCALE + bv;
                pt[k] = tl[k];
            }
            p_ip = ip;
            p_par = par;
            if (POOL && st == 0 && pooled_img >= 0) {
                // The previous item's last epilogue rode on this step: once everybody's has (the barrier) its image is whole and
                // every wave pools its share.  Nobody may write the image again before all of it has been read: a wave looks at the
                // others' pass counters in LDS before its first write of this item's output (stage 5 of the epilogue under step 1,
                // some 15 slabs from here: the counters have long moved by then).  A second s_barrier instead would line the waves
                // up once more per item.
#ifdef TRON_WS_STAMPS
                const unsigned long long st_p0 = __builtin_amdgcn_s_memtime();
#endif
                if (TRON_WS_POOL_ABLATE < 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                ++npass;
                if (TRON_WS_POOL_ABLATE < 1) pool_pass(pooled_img);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the pass's reads have returned)
                if (lane == 0) *(volatile __attribute__((address_space(3))) int *)(lds + O_FLAG + 4 * wave) = npass;
                pend_b = true;
#ifdef TRON_WS_STAMPS
                st_res += __builtin_amdgcn_s_memtime() - st_p0;           // (the barrier and the pass)
#endif
            }
        }
        if (POOL) pooled_img = ip;
#ifdef TRON_WS_STAMPS
        st_steps += __builtin_amdgcn_s_memtime() - st_d;
#endif
    }
#ifdef TRON_WS_STAMPS
    if (lane == 0 && blockIdx.x < 256) {
        unsigned long long *d = g_ws_stamps + (blockIdx.x * 12 + wave) * 8;
        d[0] = st_wait; d[1] = st_bar; d[2] = st_dma; d[3] = st_steps; d[4] = st_items;
        d[5] = __builtin_amdgcn_s_memtime() - st_t0; d[6] = __builtin_amdgcn_s_memrealtime() - st_r0; d[7] = st_res;
    }
#endif
    // the last step's epilogue
    if (RES || BWD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < NS; ++s) epi_at_slab(s);
    if (POOL) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (pooled_img >= 0) pool_pass(pooled_img);
    }
    if (BWD) {
        // this workgroup's column sums / maxima: over the 16 pixels of a tile row (lanes with equal g), then over the waves
        // that share an M tile, in a fixed order; k_wsb_finish adds the workgroups up
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) csum[r] += __shfl_xor(csum[r], d, 64);
            cmax = __builtin_fmaxf(cmax, __shfl_xor(cmax, d, 64));
        }
        __syncthreads();                                                 // (everybody is done with the item buffers)
        float *sacc = reinterpret_cast<float *>(lds), *smax = sacc + G::WAVES * 16;
        if (li == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sacc[wave * 16 + 4 * g + r] = csum[r];
            smax[wave * 4 + g] = cmax;
        }
        __syncthreads();
        if (tid < G::COUT) {
            const int ctile = tid >> 4, c16 = tid & 15;
            float sm = 0.0f, mx = 0.0f;
            for (int q = 0; q < G::PG; ++q) {
                const int w = ctile + G::NCT * q;
                sm += sacc[w * 16 + c16];
                mx = __builtin_fmaxf(mx, smax[w * 4 + (c16 >> 2)]);
            }
            bw.stats[(size_t)blockIdx.x * G::COUT + tid] = sm;
            bw.stats[((size_t)gridDim.x + blockIdx.x) * G::COUT + tid] = mx;
        }
    }
}

// ---- conv1 (DQNNet.py:10,34): int8 observation codes -> PX16, as a table sum ---------------------------------------
// The input of conv1 is util.pop_up's planes of the env's codes (util.py:11-37): every cell holds one of six codes, so a
// cell's contribution to output channel co through tap k is one of six numbers: T[code][k][co] = sum over planes of
// plane value x W[co][plane][k] (wall 1; own body 1 / head 10; enemy body 1 / head 10; + plane4 x W[co][3][k] for every
// cell when there is a fourth plane, game.py:124-132).  out = mish(bias + sum over the 9 taps of T[code at the tap]):
// nine LDS reads and adds per output instead of 27 / 36 multiply-adds, no matrix work worth a matrix core.
// One thread = one pixel x one channel octet: 16-byte table reads, one 16-byte PX16 store per half.
template <int S, bool TRAIN = false>
__global__ __launch_bounds__(256) void k_conv1_px(const int8_t *__restrict__ codes, const float *__restrict__ w,
                                                  const float *__restrict__ bias, int cin, float plane4, int64_t B,
                                                  unsigned char *__restrict__ out, unsigned char *__restrict__ pre = nullptr)
{
    // One thread = one pixel, all 32 channels: the nine neighbour codes are read and decoded once.  The table row of a
    // code is indexed by the code's low nibble (1 -> 1, -1 -> 15, -2 -> 14, -3 -> 13, 10 -> 10, -10 -> 6: all
    // different; row 0 = outside the image = zeros), so decoding a code is one AND.
    constexpr int SS = S * S, COUT = 32, ROW = 9 * COUT + 4;            // table row (floats), padded: rows land on different banks
    __shared__ __attribute__((aligned(16))) float T[16 * ROW];
    for (int i = threadIdx.x; i < 16 * 9 * COUT; i += blockDim.x) {
        const int nib = i / (9 * COUT), r = i - nib * 9 * COUT, k = r / COUT, co = r - k * COUT;
        const float *wc = w + (size_t)co * cin * 9 + k;
        float v = 0.0f;                                                  // EMPTY (1) and the unused nibbles: no plane is set
        if (nib == 15) v = wc[0];                                        // WALL -1      (map.py:67-81, util.py:18-27)
        else if (nib == 14) v = wc[9];                                   // own body -2
        else if (nib == 10) v = 10.0f * wc[9];                           // own head 10
        else if (nib == 13) v = wc[18];                                  // enemy body -3
        else if (nib == 6) v = 10.0f * wc[18];                           // enemy head -10
        const bool real = nib == 1 || nib == 15 || nib == 14 || nib == 10 || nib == 13 || nib == 6;
        if (real && cin == 4) v += plane4 * wc[27];
        T[nib * ROW + k * COUT + co] = real ? v : 0.0f;
    }
    __syncthreads();
    const int64_t total = B * SS;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t img = i / SS;
        const int p = (int)(i - img * SS);
        const int y = p / S, x = p - y * S;
        const int8_t *c = codes + img * SS;
        int row[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
            const bool inside = yy >= 0 && yy < S && xx >= 0 && xx < S;
            const int v = inside ? (int)c[yy * S + xx] : 0;
            row[k] = (v & 15) * ROW + k * COUT;
        }
        unsigned char *op = out + (size_t)img * 2 * 4 * SS * 16 + (size_t)p * 16;
#pragma unroll
        for (int oct = 0; oct < 4; ++oct) {
            f32x4 a0 = *reinterpret_cast<const f32x4 *>(bias + oct * 8), a1 = *reinterpret_cast<const f32x4 *>(bias + oct * 8 + 4);
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float *t = T + row[k] + oct * 8;
                a0 += *reinterpret_cast<const f32x4 *>(t);
                a1 += *reinterpret_cast<const f32x4 *>(t + 4);
            }
            if (TRAIN) {                                                 // the pre-activation as a PX16 image of its own (the learner's backward)
                const f32x4 z0 = a0 * ACT_SCALE, z1 = a1 * ACT_SCALE;
                const f16x4 zh0 = __builtin_convertvector(z0, f16x4), zh1 = __builtin_convertvector(z1, f16x4);
                const f16x4 zl0 = __builtin_convertvector((z0 - __builtin_convertvector(zh0, f32x4)) * LO_SCALE, f16x4);
                const f16x4 zl1 = __builtin_convertvector((z1 - __builtin_convertvector(zh1, f32x4)) * LO_SCALE, f16x4);
                unsigned char *zp = pre + (size_t)img * 2 * 4 * SS * 16 + (size_t)p * 16;
                *reinterpret_cast<f16x8 *>(zp + (size_t)oct * SS * 16) = __builtin_shufflevector(zh0, zh1, 0, 1, 2, 3, 4, 5, 6, 7);
                *reinterpret_cast<f16x8 *>(zp + (size_t)(4 + oct) * SS * 16) = __builtin_shufflevector(zl0, zl1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            a0 = mish4(a0) * ACT_SCALE;
            a1 = mish4(a1) * ACT_SCALE;
            const f16x4 h0 = __builtin_convertvector(a0, f16x4), h1 = __builtin_convertvector(a1, f16x4);
            const f16x4 l0 = __builtin_convertvector((a0 - __builtin_convertvector(h0, f32x4)) * LO_SCALE, f16x4);
            const f16x4 l1 = __builtin_convertvector((a1 - __builtin_convertvector(h1, f32x4)) * LO_SCALE, f16x4);
            const f16x8 hh = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            const f16x8 ll = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
            *reinterpret_cast<f16x8 *>(op + (size_t)oct * SS * 16) = hh;
            *reinterpret_cast<f16x8 *>(op + (size_t)(4 + oct) * SS * 16) = ll;
        }
    }
}

// ---- PX16 -> f32 NCHW (what the head of the net, or a caller that wants plain tensors, reads) -----------------------
__global__ __launch_bounds__(256) void k_px16_to_f32(const unsigned char *__restrict__ in, float *__restrict__ out, int64_t B, int C, int SS)
{
    const int64_t total = B * (C / 8) * SS;                              // (image, octet, pixel)
    const size_t half = (size_t)(C / 8) * SS * 16;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t img = i / ((C / 8) * SS);
        const int r = (int)(i - img * (C / 8) * SS), oct = r / SS, p = r - oct * SS;
        const unsigned char *ip = in + (size_t)img * 2 * half + (size_t)(oct * SS + p) * 16;
        const f16x8 hh = *reinterpret_cast<const f16x8 *>(ip), ll = *reinterpret_cast<const f16x8 *>(ip + half);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            out[((size_t)img * C + oct * 8 + j) * SS + p] = ((float)hh[j] + (float)ll[j] * LO_UNSCALE) * ACT_UNSCALE;
    }
}

int device_cus()
{
    static int cus[64];
    static uint64_t known = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
    if (!(known & (1ull << (dev & 63)))) {
        hipDeviceProp_t prop;
        cus[dev & 63] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
        (void)hipGetLastError();
        known |= 1ull << (dev & 63);
    }
    return cus[dev & 63];
}

template <class G, int MODE = WS_INFER>
int launch_ws(const void *in, const void *wfrag, const float *bias, const void *res, void *out, float *out_f32,
              float *pre_f32, int64_t B, int apply_mish, hipStream_t st, void *pre_px = nullptr, WsBwd bw = WsBwd{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr},
              int *grid_out = nullptr)
{
    const int64_t nitems = (B + G::IPI - 1) / G::IPI * G::NB;
    if (nitems > 0x7fffffff) return TRON_ERR_UNSUPPORTED;
    int grid = device_cus() / G::NB * G::NB;
    if (nitems < grid) grid = (int)nitems;
    if (grid_out) *grid_out = grid;
    constexpr size_t LDS_ALL = G::LDS_BYTES;
    static uint64_t prepared[4] = {0, 0, 0, 0};                         // per variant, one bit per device (the attribute is per device)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_NO_DEVICE; }
    const uint64_t dev_bit = 1ull << (dev & 63);
#define TRON_WS_LAUNCH(RES_, F32_)                                                                                    \
    do {                                                                                                              \
        auto kern = k_conv_ws<G, RES_, F32_, MODE>;                                                                   \
        if (!(prepared[RES_ * 2 + F32_] & dev_bit)) {                                                                 \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)LDS_ALL) != hipSuccess)                                                      \
                (void)hipGetLastError();                                                                              \
            prepared[RES_ * 2 + F32_] |= dev_bit;                                                                     \
        }                                                                                                             \
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(G::THREADS), LDS_ALL, st,                                    \
                           reinterpret_cast<const unsigned char *>(in), reinterpret_cast<const f16x8 *>(wfrag), bias, \
                           reinterpret_cast<const unsigned char *>(res), reinterpret_cast<unsigned char *>(out),      \
                           out_f32, pre_f32, (int)B, apply_mish, (int)nitems, reinterpret_cast<unsigned char *>(pre_px), bw); \
    } while (0)
    // anything but the chain's inner layers takes the general variant (BWD: the one that also writes f32 planes)
    const bool f32o = MODE == WS_BWD ? out_f32 != nullptr : (out_f32 || pre_f32 || !apply_mish || !out);
    if (res) { if (f32o) TRON_WS_LAUNCH(true, true); else TRON_WS_LAUNCH(true, false); }
    else     { if (f32o) TRON_WS_LAUNCH(false, true); else TRON_WS_LAUNCH(false, false); }
#undef TRON_WS_LAUNCH
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

// ds_read_b64_tr_b16 by inline asm, not by __builtin_amdgcn_ds_read_tr16_b64: behind the builtin hipcc (ROCm 7.2) puts an
// s_waitcnt vmcnt(0) in front of every transposed read that follows an LDS-DMA (it cannot tell the read from the copy's
// destination), which serialised each of an item's nine copies with the multiply — 11 us per item instead of 3 (the .s showed
// ten vmcnt(0) per item).  The asm is invisible to that pass; what it costs is that the waits are ours: the reads of a tap go
// out one tap ahead, `lds_wait()` (s_waitcnt lgkmcnt(0) + sched_barrier, so that no consumer moves above it) stands between a
// read and the first use of its registers, and the two 8-byte halves are joined into the MFMA operand only AFTER that wait (a
// copy the compiler might make of them then reads landed data).  EXEC must be all ones (the read crosses lanes).
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4 lds_tr(uint32_t addr)
{
    s16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=&v"(r) : "v"(addr));
    return r;
}
__device__ __forceinline__ void lds_wait()
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


}  // namespace
