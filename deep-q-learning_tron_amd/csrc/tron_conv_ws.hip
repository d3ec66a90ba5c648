// tron_conv_ws.hip — the CNN's 3x3 convolutions (Net/DQNNet.py:10-17,33-50: conv + bias + residual + mish) for
// gradient-free forwards, WEIGHT-STATIONARY on the split-f16 matrix cores (the arithmetic of tron_conv_f16.hip:
// v = hi + lo 2^-11, three v_mfma_f32_16x16x32_f16 per k-slab, f32 accumulation).
//
// What bounded tron_conv_f16.hip (wave stamps): every 16-channel chunk re-pulled 41 KB of split weights
// from L2 into LDS per workgroup, input pieces were converted and stored to LDS by VALU work inside the MFMA loop, and
// the epilogue wrote the next layer's operand image with 2-byte stores.  Here
//   * the WEIGHTS LIVE IN REGISTERS for the whole launch: GEMM view M = output channels, N = pixels, K = taps x input
//     channels; a wave owns one 16-channel M tile and ALL of K — 9 (cin 32) or 18 (cin 64) slabs of 32 k-values, hi and
//     lo fragments, 72 / 144 VGPRs — loaded once per launch; a persistent workgroup of four waves (one per SIMD, the
//     whole register file) then streams images past them;
//   * activations travel between layers as the PX16 image: per image [hi | lo][channel octet][pixel][8 channels] f16,
//     already scaled by 2^-6 and split.  An octet plane is contiguous, so an image goes global -> LDS by LDS-DMA
//     (global_load_lds_dwordx4) with no register, no VALU and no ds_write, double-buffered one image ahead;
//   * a B fragment (8 consecutive k of one pixel) is ONE ds_read_b128 at "per-tile lane base + immediate": the three
//     horizontal taps have their own base (a lane whose tap leaves the row reads a zero zone instead — the planes carry
//     no padding columns), vertical taps, channel blocks and the lo half are immediates.  Planes start on multiples of
//     256 bytes and a tile is 16 consecutive pixels, so the 16 lanes of every LDS read group hit 16 different 16-byte
//     slots: conflict-free at any tap shift (scripts/lds_bank_search.py);
//   * with channels on the MFMA rows a lane ends up holding 4 consecutive channels of one pixel: the epilogue writes
//     the next layer's PX16 image with 8-byte stores that pair up to full 256-byte runs per wave instruction;
//   * K = 9 taps x 32 channels is exactly nine slabs: no tenth zero tap (tron_conv_f16.hip pads 9 to 10).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tron_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int THREADS = 256;
constexpr float ACT_SCALE = 1.0f / 64.0f, ACT_UNSCALE = 64.0f, LO_SCALE = 2048.0f, LO_UNSCALE = 1.0f / 2048.0f;

constexpr int align256(int x) { return (x + 255) & ~255; }

// Geometry of one instantiation.  S: image side.  R: image rows per work item (R == S: whole images; larger boards are
// cut into bands of at most R rows that bring one halo row above and below).  IPI: whole images per item (small
// channel counts: two images give every pixel group the same number of tiles).
template <int S_, int R_, int CIN_, int COUT_, int IPI_>
struct Geo {
    static constexpr int S = S_, R = R_, CIN = CIN_, COUT = COUT_, IPI = IPI_;
    static constexpr int NB = (S + R - 1) / R;                           // bands per image
    static constexpr int SS = S * S;
    static constexpr int ROWB = S * 16;                                  // bytes of one image row in an octet plane
    static constexpr int PLANE = align256((R + 2) * ROWB);               // LDS octet plane: halo row, R rows, halo row
    static constexpr int ZONE = align256(256 + 2 * ROWB);                // zeros in front of each 32-channel block
    static constexpr int CBLK = ZONE + 4 * PLANE;                        // one 32-channel block (4 octets)
    static constexpr int NCB = CIN / 32;
    static constexpr int HALF = NCB * CBLK;                              // hi (or lo) image
    static constexpr int IMG = 2 * HALF;
    static constexpr int BUF = IPI * IMG;
    static constexpr int NS = 9 * NCB;                                   // 32-deep k slabs: (tap, channel block)
    static constexpr int NCT = COUT / 16;                                // 16-channel M tiles
    static constexpr int PG = 4 / NCT;                                   // pixel groups: waves = NCT x PG
    static constexpr int NPIECE = ((R + 2) * ROWB + 1023) / 1024;        // 1 KB LDS-DMA pieces per plane, at most
    static constexpr int NDMA = IPI * 2 * (CIN / 8) * NPIECE;
    static constexpr int DMA_PER_WAVE = (NDMA + 3) / 4;
    static constexpr int PLANEG = SS * 16;                               // an octet plane of the PX16 image in memory
    static constexpr int HALFG_IN = (CIN / 8) * PLANEG, HALFG_OUT = (COUT / 8) * PLANEG;
    static constexpr size_t LDS_BYTES = 2 * (size_t)BUF;
    static_assert(NCT * PG == 4 && NCT >= 1, "four waves: M tiles x pixel groups");
    static_assert(CIN % 32 == 0 && COUT % 16 == 0, "channel blocks");
    static_assert(IPI == 1 || NB == 1, "several images or several bands, not both");
    static_assert(2 * ROWB + CBLK * (NCB - 1) + HALF + 16 <= 65536, "ds_read immediate offsets");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
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
};
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
        split(w[((size_t)co * cin + ci) * 9 + tap], h, l);
        const size_t o = ((size_t)(ct * ns + s) * 2) * 512 + lane * 8 + j;
        ws[o] = h;
        ws[o + 512] = l;
    }
}

template <class G, bool RES>
__global__ __launch_bounds__(THREADS, 1) void k_conv_ws(
    const unsigned char *__restrict__ in, const f16x8 *__restrict__ wfrag, const float *__restrict__ bias,
    const unsigned char *__restrict__ res, unsigned char *__restrict__ out, float *__restrict__ out_f32,
    float *__restrict__ pre_f32, int B, int apply_mish, int nitems)
{
    constexpr int S = G::S, NS = G::NS, NCB = G::NCB;
    extern __shared__ __attribute__((aligned(256))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int ct = wave % G::NCT, pg = wave / G::NCT;

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
    if (bias) bv = *reinterpret_cast<const f32x4 *>(bias + co0);

    for (int i = tid; i < (int)(G::LDS_BYTES / 16); i += THREADS) reinterpret_cast<uint4 *>(lds)[i] = make_uint4(0u, 0u, 0u, 0u);

    // the band this workgroup serves (the grid is a multiple of NB: halo rows that are zero for it stay zero)
    const int band = blockIdx.x % G::NB;
    const int r0 = band * G::R, rows_b = (S - r0 < G::R) ? S - r0 : G::R;
    const int ipx = rows_b * S;                                          // pixels of one image in an item
    const int npx = G::IPI * ipx;
    const int ntiles = (npx + 15) >> 4;
    const int my_nt = pg < ntiles ? (ntiles - pg + G::PG - 1) / G::PG : 0;
    const int row_lo = r0 > 0 ? r0 - 1 : 0, row_hi = (r0 + rows_b + 1 < S) ? r0 + rows_b + 1 : S;
    const int dbytes = (row_hi - row_lo) * G::ROWB, drow = (row_lo - (r0 - 1)) * G::ROWB;

    // item -> LDS buffer `buf`: every octet plane's rows [row_lo, row_hi) as 1 KB LDS-DMA pieces, dealt over the waves
    auto dma = [&](int item, int buf) {
        const int ip = item / G::NB;
#pragma unroll
        for (int j = 0; j < G::DMA_PER_WAVE; ++j) {
            const int q = wave + 4 * j;
            if (q >= G::NDMA) break;
            const int pc = q % G::NPIECE, pl = q / G::NPIECE;
            const int o = pl % (G::CIN / 8), h = (pl / (G::CIN / 8)) & 1, il = pl / (2 * (G::CIN / 8));
            const int img = ip * G::IPI + il;
            const int off = pc * 1024 + lane * 16;
            if (img < B && off < dbytes) {
                const unsigned char *src = in + ((size_t)img * 2 + h) * G::HALFG_IN + o * G::PLANEG + row_lo * G::ROWB + off;
                unsigned char *dst = lds + buf * G::BUF + il * G::IMG + h * G::HALF + (o >> 2) * G::CBLK + G::ZONE +
                                     (o & 3) * G::PLANE + drow + pc * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
            }
        }
    };

    // per-lane LDS byte offsets of tile t's B fragments for the three horizontal taps (vertical tap, channel block and
    // half are immediates on top).  A lane whose tap falls outside the row reads the zero zone, at the 16-byte slot its
    // real address would have had (the read stays conflict-free).
    auto bases = [&](int t, int bufoff, int (&b)[3]) {
        int o = 16 * t + li;
        o = o < npx ? o : npx - 1;
        const int il = G::IPI > 1 ? o / G::SS : 0, rem = G::IPI > 1 ? o - il * G::SS : o;
        const int y = rem / S, x = rem - y * S;
        const int p = y * S + x;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int xx = x + kx - 1, pp = p + kx - 1;
            const bool ok = xx >= 0 && xx < S;
            b[kx] = bufoff + il * G::IMG + (ok ? G::ZONE + g * G::PLANE + pp * 16 : (pp & 15) * 16);
        }
    };

    auto epilogue = [&](int t, int ip, f32x4 a0, f32x4 a1) {
        const int o = 16 * t + li;
        const int oc = o < npx ? o : npx - 1;
        const int il = G::IPI > 1 ? oc / G::SS : 0, rem = G::IPI > 1 ? oc - il * G::SS : oc;
        const int img = ip * G::IPI + il;
        const bool valid = o < npx && img < B;
        const int pixg = r0 * S + rem;                                   // pixel within the image (rem = y * S + x)
        f32x4 v = (a0 + a1 * LO_UNSCALE) * ACT_UNSCALE + bv;
        const uint32_t poff = (uint32_t)((2 * ct + (g >> 1)) * G::PLANEG + pixg * 16 + (g & 1) * 8);
        if (RES) {
            const unsigned char *rp = res + (size_t)img * 2 * G::HALFG_OUT + poff;
            f16x4 rh = {0, 0, 0, 0}, rl = {0, 0, 0, 0};
            if (valid) {
                rh = *reinterpret_cast<const f16x4 *>(rp);
                rl = *reinterpret_cast<const f16x4 *>(rp + G::HALFG_OUT);
            }
            v += (__builtin_convertvector(rh, f32x4) + __builtin_convertvector(rl, f32x4) * LO_UNSCALE) * ACT_UNSCALE;
        }
        if (!valid) return;
        const size_t f32o = ((size_t)img * G::COUT + co0) * G::SS + pixg;
        if (pre_f32) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pre_f32[f32o + r * G::SS] = v[r];
        }
        if (apply_mish) v = mish4(v);
        if (out_f32) {
#pragma unroll
            for (int r = 0; r < 4; ++r) out_f32[f32o + r * G::SS] = v[r];
        }
        if (out) {
            unsigned char *op = out + (size_t)img * 2 * G::HALFG_OUT + poff;
            const f32x4 sv = v * ACT_SCALE;
            const f16x4 hh = __builtin_convertvector(sv, f16x4);
            const f16x4 ll = __builtin_convertvector((sv - __builtin_convertvector(hh, f32x4)) * LO_SCALE, f16x4);
            *reinterpret_cast<f16x4 *>(op) = hh;
            *reinterpret_cast<f16x4 *>(op + G::HALFG_OUT) = ll;
        }
    };

    __syncthreads();                                                     // zero fill done
    int item = blockIdx.x, cur = 0;
    if (item < nitems) dma(item, 0);
    for (; item < nitems; item += (int)gridDim.x, cur ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this wave's pieces of `item` have landed ...
        __syncthreads();                                                 // ... everybody's have; the other buffer is free
        const int nxt = item + (int)gridDim.x;
        if (nxt < nitems) dma(nxt, cur ^ 1);
        const int ip = item / G::NB;
        const int bufoff = cur * G::BUF;
        for (int tp = 0; tp < my_nt; tp += 2) {
            const int t0 = pg + tp * G::PG, t1 = t0 + G::PG;
            const bool two = tp + 1 < my_nt;                             // (wave-uniform)
            int b0[3], b1[3];
            bases(t0, bufoff, b0);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if (two) {
                // two tiles at once: six independent accumulator chains (a dependent MFMA issues 48 cycles after its
                // producer, an independent one 16)
                bases(t1, bufoff, b1);
                f32x4 h0 = z, x0 = z, y0 = z, h1 = z, x1 = z, y1 = z;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int tap = s / NCB, cb = s % NCB, ky = tap / 3, kx = tap % 3;
                    const int imm = ky * G::ROWB + cb * G::CBLK;
                    const f16x8 bh0 = *reinterpret_cast<const f16x8 *>(lds + b0[kx] + imm);
                    const f16x8 bl0 = *reinterpret_cast<const f16x8 *>(lds + b0[kx] + imm + G::HALF);
                    const f16x8 bh1 = *reinterpret_cast<const f16x8 *>(lds + b1[kx] + imm);
                    const f16x8 bl1 = *reinterpret_cast<const f16x8 *>(lds + b1[kx] + imm + G::HALF);
                    x0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], bl0, x0, 0, 0, 0);
                    h0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], bh0, h0, 0, 0, 0);
                    x1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], bl1, x1, 0, 0, 0);
                    h1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], bh1, h1, 0, 0, 0);
                    y0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s], bh0, y0, 0, 0, 0);
                    y1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s], bh1, y1, 0, 0, 0);
                }
                epilogue(t0, ip, h0, x0 + y0);
                epilogue(t1, ip, h1, x1 + y1);
            } else {
                f32x4 h0 = z, x0 = z, y0 = z, h1 = z, x1 = z, y1 = z;   // odd / even slabs on their own chains
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int tap = s / NCB, cb = s % NCB, ky = tap / 3, kx = tap % 3;
                    const int imm = ky * G::ROWB + cb * G::CBLK;
                    const f16x8 bh0 = *reinterpret_cast<const f16x8 *>(lds + b0[kx] + imm);
                    const f16x8 bl0 = *reinterpret_cast<const f16x8 *>(lds + b0[kx] + imm + G::HALF);
                    if (s & 1) {
                        x1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], bl0, x1, 0, 0, 0);
                        h1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], bh0, h1, 0, 0, 0);
                        y1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s], bh0, y1, 0, 0, 0);
                    } else {
                        x0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], bl0, x0, 0, 0, 0);
                        h0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], bh0, h0, 0, 0, 0);
                        y0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s], bh0, y0, 0, 0, 0);
                    }
                }
                epilogue(t0, ip, h0 + h1, (x0 + x1) + (y0 + y1));
            }
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
template <int S>
__global__ __launch_bounds__(256) void k_conv1_px(const int8_t *__restrict__ codes, const float *__restrict__ w,
                                                  const float *__restrict__ bias, int cin, float plane4, int64_t B,
                                                  unsigned char *__restrict__ out)
{
    constexpr int SS = S * S, COUT = 32, ROW = 9 * COUT + 4;            // table row of one code (floats), padded against bank conflicts
    __shared__ __attribute__((aligned(16))) float T[7 * ROW];
    for (int i = threadIdx.x; i < 7 * 9 * COUT; i += blockDim.x) {
        const int code = i / (9 * COUT), r = i - code * 9 * COUT, k = r / COUT, co = r - k * COUT;
        const float *wc = w + (size_t)co * cin * 9 + k;
        // code index: 0 = outside the image, 1 = EMPTY (1), 2 = WALL (-1), 3 = own body (-2), 4 = own head (10),
        //             5 = enemy body (-3), 6 = enemy head (-10)            (map.py:67-81)
        float v = 0.0f;
        if (code == 2) v = wc[0];
        else if (code == 3) v = wc[9];
        else if (code == 4) v = 10.0f * wc[9];
        else if (code == 5) v = wc[18];
        else if (code == 6) v = 10.0f * wc[18];
        if (code != 0 && cin == 4) v += plane4 * wc[27];
        T[code * ROW + k * COUT + co] = v;
    }
    __syncthreads();
    const int64_t total = B * SS * 4;                                    // (image, pixel, octet)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t img = i / (SS * 4);
        const int r = (int)(i - img * SS * 4), oct = r / SS, p = r - oct * SS;    // octet-major: a wave's stores are contiguous
        const int y = p / S, x = p - y * S;
        const int8_t *c = codes + img * SS;
        f32x4 a0 = *reinterpret_cast<const f32x4 *>(bias + oct * 8), a1 = *reinterpret_cast<const f32x4 *>(bias + oct * 8 + 4);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
            int code = 0;
            if (yy >= 0 && yy < S && xx >= 0 && xx < S) {
                const int v = c[yy * S + xx];
                code = v == 1 ? 1 : v == -1 ? 2 : v == -2 ? 3 : v == 10 ? 4 : v == -3 ? 5 : v == -10 ? 6 : 1;
            }
            const float *t = T + code * ROW + k * COUT + oct * 8;
            a0 += *reinterpret_cast<const f32x4 *>(t);
            a1 += *reinterpret_cast<const f32x4 *>(t + 4);
        }
        a0 = mish4(a0) * ACT_SCALE;
        a1 = mish4(a1) * ACT_SCALE;
        f16x8 hh, ll;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f16 h, l;
            split(a0[j], h, l);
            hh[j] = h; ll[j] = l;
            split(a1[j], h, l);
            hh[4 + j] = h; ll[4 + j] = l;
        }
        unsigned char *op = out + (size_t)img * 2 * 4 * SS * 16 + (size_t)(oct * SS + p) * 16;
        *reinterpret_cast<f16x8 *>(op) = hh;
        *reinterpret_cast<f16x8 *>(op + 4 * SS * 16) = ll;
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

template <class G>
int launch_ws(const void *in, const void *wfrag, const float *bias, const void *res, void *out, float *out_f32,
              float *pre_f32, int64_t B, int apply_mish, hipStream_t st)
{
    const int64_t nitems = (B + G::IPI - 1) / G::IPI * G::NB;
    if (nitems > 0x7fffffff) return TRON_ERR_UNSUPPORTED;
    int grid = device_cus() / G::NB * G::NB;
    if (nitems < grid) grid = (int)nitems;
    static bool prepared[2] = {false, false};
#define TRON_WS_LAUNCH(RES_)                                                                                          \
    do {                                                                                                              \
        auto kern = k_conv_ws<G, RES_>;                                                                               \
        if (!prepared[RES_]) {                                                                                        \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)G::LDS_BYTES) != hipSuccess)                                                 \
                (void)hipGetLastError();                                                                              \
            prepared[RES_] = true;                                                                                    \
        }                                                                                                             \
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(THREADS), G::LDS_BYTES, st,                               \
                           reinterpret_cast<const unsigned char *>(in), reinterpret_cast<const f16x8 *>(wfrag), bias, \
                           reinterpret_cast<const unsigned char *>(res), reinterpret_cast<unsigned char *>(out),      \
                           out_f32, pre_f32, (int)B, apply_mish, (int)nitems);                                        \
    } while (0)
    if (res) TRON_WS_LAUNCH(true); else TRON_WS_LAUNCH(false);
#undef TRON_WS_LAUNCH
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

}  // namespace

extern "C" int64_t tron_px16_bytes(int64_t batch, int32_t channels, int32_t side)
{
    if (batch < 0 || channels < 8 || channels % 8 || side < 1) return 0;
    return batch * channels * side * side * 4;
}

extern "C" int64_t tron_conv3x3_ws_workspace(int32_t cin, int32_t cout)
{
    if (cin < 32 || cin % 32 || cout < 16 || cout % 16) return 0;
    return (int64_t)(cout / 16) * 9 * (cin / 32) * 2 * 64 * 16;
}

extern "C" int tron_conv3x3_ws_split_weights(const float *const *weights, const int32_t *cins, const int32_t *couts,
                                             void *const *workspaces, int32_t n, void *stream)
{
    if (!weights || !cins || !couts || !workspaces || n < 1 || n > WS_SPLIT_MAX) return TRON_ERR_BAD_ARG;
    WsJobs jobs{};
    int most = 0;
    for (int k = 0; k < n; ++k) {
        if (!weights[k] || !workspaces[k] || (reinterpret_cast<uintptr_t>(workspaces[k]) & 15u)) return TRON_ERR_BAD_ARG;
        if (tron_conv3x3_ws_workspace(cins[k], couts[k]) == 0 || cins[k] > 1024 || couts[k] > 1024) return TRON_ERR_UNSUPPORTED;
        jobs.w[k] = weights[k];
        jobs.ws[k] = reinterpret_cast<f16 *>(workspaces[k]);
        jobs.cin[k] = cins[k];
        jobs.cout[k] = couts[k];
        const int total = (couts[k] / 16) * 9 * (cins[k] / 32) * 512;
        most = total > most ? total : most;
    }
    hipLaunchKernelGGL(k_ws_split_weights, dim3((most + 255) / 256, n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), jobs);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

extern "C" int tron_conv3x3_ws_fwd(const void *in_px16, const void *wfrag, const float *bias, const void *res_px16,
                                   void *out_px16, float *out_f32, float *pre_f32, int64_t batch, int32_t cin,
                                   int32_t cout, int32_t side, int32_t apply_mish, void *stream)
{
    if (!in_px16 || !wfrag || batch < 0 || (!out_px16 && !out_f32 && !pre_f32)) return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(in_px16) | reinterpret_cast<uintptr_t>(wfrag) | reinterpret_cast<uintptr_t>(res_px16) |
         reinterpret_cast<uintptr_t>(out_px16) | reinterpret_cast<uintptr_t>(bias)) & 15u)
        return TRON_ERR_BAD_ARG;
    if (batch == 0) return TRON_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define TRON_WS_CASE(S_, R_, CI_, CO_, IPI_)                                                                          \
    if (side == S_ && cin == CI_ && cout == CO_)                                                                      \
        return launch_ws<Geo<S_, R_, CI_, CO_, IPI_>>(in_px16, wfrag, bias, res_px16, out_px16, out_f32, pre_f32, batch, apply_mish, st);
    TRON_WS_CASE(12, 12, 32, 32, 2)
    TRON_WS_CASE(12, 12, 32, 64, 1)
    TRON_WS_CASE(12, 12, 64, 64, 1)
    TRON_WS_CASE(26, 13, 32, 32, 1)
    TRON_WS_CASE(26, 13, 32, 64, 1)
    TRON_WS_CASE(26, 9, 64, 64, 1)
#undef TRON_WS_CASE
    return TRON_ERR_UNSUPPORTED;
}

extern "C" int tron_conv1_px16(const int8_t *codes, const float *weight, const float *bias, int32_t cin, float plane4,
                               int64_t batch, int32_t side, void *out_px16, void *stream)
{
    if (!codes || !weight || !bias || !out_px16 || batch < 0 || (cin != 3 && cin != 4)) return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(out_px16) | reinterpret_cast<uintptr_t>(bias)) & 15u) return TRON_ERR_BAD_ARG;
    if (batch == 0) return TRON_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = batch * side * side * 4;
    const unsigned grid = (unsigned)((total + 255) / 256 < 256 * 16 ? (total + 255) / 256 : 256 * 16);
    if (side == 12) hipLaunchKernelGGL(k_conv1_px<12>, dim3(grid), dim3(256), 0, st, codes, weight, bias, cin, plane4, batch, reinterpret_cast<unsigned char *>(out_px16));
    else if (side == 26) hipLaunchKernelGGL(k_conv1_px<26>, dim3(grid), dim3(256), 0, st, codes, weight, bias, cin, plane4, batch, reinterpret_cast<unsigned char *>(out_px16));
    else if (side == 34) hipLaunchKernelGGL(k_conv1_px<34>, dim3(grid), dim3(256), 0, st, codes, weight, bias, cin, plane4, batch, reinterpret_cast<unsigned char *>(out_px16));
    else return TRON_ERR_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

extern "C" int tron_px16_to_f32(const void *in_px16, float *out, int64_t batch, int32_t channels, int32_t side, void *stream)
{
    if (!in_px16 || !out || batch < 0 || channels < 8 || channels % 8 || side < 1) return TRON_ERR_BAD_ARG;
    if (batch == 0) return TRON_OK;
    const int64_t total = batch * (channels / 8) * side * side;
    const unsigned grid = (unsigned)((total + 255) / 256 < 256 * 32 ? (total + 255) / 256 : 256 * 32);
    hipLaunchKernelGGL(k_px16_to_f32, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const unsigned char *>(in_px16), out, batch, channels, side * side);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}
