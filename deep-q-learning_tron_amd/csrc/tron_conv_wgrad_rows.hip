// tron_conv_wgrad_rows.hip — the weight gradient of the 3x3 convolutions (tron_conv3x3_wgrad, include/tron_hip.h;
// DDQN.py:148 through conv2..conv6 of Net/DQNNet.py:10-17) for boards whose image does not fit
// csrc/tron_conv_wgrad.hip's one-strip-per-image layout: 26x26 observations (24x24 boards, BASELINE config 3).
//     dW[co][ci][ky][kx] = sum over b, y, x of  g[b][co][y][x] * in[b][ci][y + ky - 1][x + kx - 1]
// Same arithmetic as there: split-f16 matrix cores (v = hi + lo 2^-11, three MFMAs per k-slab, f32 accumulation), the
// gradient scaled by the power of two its block maxima give, per-workgroup partial sums added in a fixed order.
//
// The GEMM: M = cout, N = cin per tap, K = pixels.  One image ROW is one 32-deep k-slab (26 pixels + 6 zeros), and the
// rows of a workgroup's images stream through LDS: step u stages one input row and one gradient row (f32 -> split f16,
// both halves) while the waves multiply the slab staged three steps earlier — a ring of four input rows (a slab needs
// rows y-1, y, y+1; the fourth is being written) and two gradient rows, one barrier per step, images back to back.
//   * vertical taps are different ring slots of the input; horizontal taps are three copies of the GRADIENT row, stored
//     shifted by -1, 0, +1 pixel (the smaller operand: 32 channels per workgroup against up to 64), so that every MFMA
//     operand is one aligned ds_read_b128: tap (ky, kx) = copy kx of g row y  x  input row y + ky - 1;
//   * a wave owns one (16 output channels x 16 input channels) tile and all nine taps: 18 accumulators = 72 registers;
//     8 waves = 2 output tiles x 4 input tiles (cin 64; the other 32 output channels are a second workgroup) or
//     4 x 2 (cin 32, cout 64), 4 waves = 2 x 2 (32 -> 32).  Two waves per SIMD: one wave's conversion work runs under the
//     other's MFMAs;
//   * rows are 96 bytes apart per channel (8 + 32 + 8 f16): 32 mod 64, the conflict-free pitch for ds_read_b128 by
//     (row = lane % 16, column = lane / 16) (MI355X guide, LDS table).
// Work done on padding: 32/26 in K and (S + 2)/S steps per image: 1.33x, the same factor as the 12x12 kernel's strip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tron_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifndef TRON_WR_ABLATE      // diagnostic builds only (wrong results): 1 = no LDS stores of the staged rows, 2 = no MFMAs, 3 = no global loads,
#define TRON_WR_ABLATE 0    // 4 = no barrier
#endif
constexpr int ROWEL = 48, ROWB = ROWEL * 2, PADL = 8;                   // an LDS row: 8 zeros, 32 pixels (>= S real), 8 zeros
constexpr float IN_SCALE = 1.0f / 64.0f, LO_SCALE = 2048.0f;

template <int S_, int CIN_, int COW_, int HALVES_ = 1>
struct RCfg {
    static constexpr int S = S_, CIN = CIN_, COW = COW_;                 // COW: output channels per workgroup
    // HALVES = 2 (34x34: a row does not fit one 32-deep slab): every image is walked twice, once per column half.  Half h takes
    // the gradient's columns h HW .. h HW + HW - 1 against the input's columns from one to the left of them to one to the right —
    // both read from column h (HW - 1) on, so that slab position = column - h (HW - 1): the gradient sits at positions h .. h + HW - 1,
    // the input at 0 .. HW, everything else of the slab is zero.  Same kernel otherwise; 2 (S + 2) steps per image.
    static constexpr int HALVES = HALVES_, HW = S / HALVES;              // gradient columns per half
    static constexpr int NPOS = HALVES == 1 ? S : HW + 1;                // slab positions in use (input columns per half)
    static constexpr int NCI = CIN / 16, NCO = COW / 16, WAVES = NCI * NCO, THREADS = 64 * WAVES;
    static constexpr int P = S + 2;                                      // steps per image: input rows -1 .. S
    static constexpr int HS = S / 2;                                     // float pairs per row
    static constexpr int IN_SLOT = CIN * ROWB, IN_HALF = 4 * IN_SLOT;    // ring of four rows
    static constexpr int G_SLOT = COW * ROWB, G_HALF = 3 * 2 * G_SLOT;   // [copy kx][ring of two rows]
    static constexpr int LDS = 2 * IN_HALF + 2 * G_HALF;
    static_assert(S % 2 == 0 && NPOS <= 32 && (HALVES == 1 || HW % 2 == 1) && (WAVES == 4 || WAVES == 8), "one 32-deep slab per (half) row; 4 or 8 waves");
    static_assert(LDS <= 160 * 1024, "LDS");
};

__device__ __forceinline__ uint32_t pack2(f16 a, f16 b)
{
    const f16x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, v);
}

// per-block maxima of |g| -> one power-of-two scale that puts the largest magnitude in [2^13, 2^14) (as tron_conv_wgrad.hip)
template <int THREADS>
__device__ __forceinline__ float grad_scale(const float *__restrict__ absmax, int n_absmax, float *red)
{
    float m = 0.0f;
    for (int i = threadIdx.x; i < n_absmax; i += THREADS) m = fmaxf(m, absmax[i]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = 0.0f;
#pragma unroll
    for (int k = 0; k < THREADS / 64; ++k) m = fmaxf(m, red[k]);
    const uint32_t bits = __float_as_uint(m);
    const int e = (int)((bits >> 23) & 255u) - 126;                     // m = f 2^e, f in [0.5, 1)
    if (!(m > 0.0f) || e < -100 || e > 100) return 1.0f;
    return __uint_as_float((uint32_t)(127 + 14 - e) << 23);            // 2^(14 - e)
}

template <class C>
__global__ __launch_bounds__(C::THREADS, 2) void k_wgrad_rows(const float *__restrict__ in, const float *__restrict__ gp,
                                                              const float *__restrict__ absmax, int n_absmax,
                                                              float *__restrict__ partial, int batch, int cout)
{
    constexpr int S = C::S, CIN = C::CIN, COW = C::COW, P = C::P, THREADS = C::THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ float red[8];
    unsigned char *in_h = lds, *g_h = lds + 2 * C::IN_HALF;              // lo halves: + IN_HALF / + G_HALF
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, g = lane >> 4;
    const int nhalves = cout / COW, co0 = ((int)blockIdx.x % nhalves) * COW, wg = (int)blockIdx.x / nhalves;
    const int wgs = (int)gridDim.x / nhalves;
    const int nb = wg < batch ? (batch - wg + wgs - 1) / wgs : 0;       // this workgroup's images: wg, wg + wgs, ...
    const int nbv = nb * C::HALVES;                                      // ... as (image, column half) pairs
    const int it = wave % C::NCI, ct = wave / C::NCI;                    // this wave's input / output channel tile

    for (int i = tid * 16; i < C::LDS; i += THREADS * 16) *reinterpret_cast<f32x4 *>(lds + i) = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float gscale = grad_scale<THREADS>(absmax, n_absmax, red);     // (contains the barrier after the clear)

    f32x4 acc0[9], acc1[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc0[k] = acc1[k] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Staging.  An LDS store instruction costs the CU's one store path 4-13 cycles whatever its lanes carry (a first
    // version with 2-byte stores for the shifted gradient copies spent 75 % of its time there), so: an item is FOUR
    // pixels of one channel's row (two 8-byte loads; 8-byte LDS stores), a wave-pass is 9 rows x 7 items and is either all
    // input rows or all gradient rows (wave-uniform: the other kind's instructions are skipped, not masked), and the
    // shifted copies are cut from the neighbouring lanes' registers (DPP wave shifts + v_alignbit), not stored piecewise.
    constexpr int Q = (C::NPOS + 3) / 4, RPW = 64 / Q;                   // items per row, rows per wave-pass
    constexpr int HALVES = C::HALVES, HW = C::HW;
    constexpr int NPI = (CIN + RPW - 1) / RPW, NPG = (COW + RPW - 1) / RPW, NROUND = (NPI + NPG + C::WAVES - 1) / C::WAVES;
    const int rip = lane / Q, q = lane - rip * Q;                        // row within the pass, item within the row
    int st_src[NROUND], st_dst[NROUND];                                  // source offset within the image / LDS offset; -1: nothing
    // which of an item's four positions 4 q + j carry data: the input's [0, NPOS), the gradient's [h, h + HW) for half h of two,
    // [0, S) for a whole row (the rest of the slab stays zero); the second 8-byte load is skipped where both its pixels are out
#pragma unroll
    for (int r = 0; r < NROUND; ++r) {
        const int pidx = wave + r * C::WAVES;
        const bool is_in = pidx < NPI;
        const int row = (is_in ? pidx : pidx - NPI) * RPW + rip;
        const bool ok = rip < RPW && row < (is_in ? CIN : COW) && pidx < NPI + NPG;
        st_src[r] = ok ? row * S * S + 4 * q : -1;
        st_dst[r] = row * ROWB + (PADL + 4 * q) * 2;
    }
    auto pos_lo = [&](bool is_in, int half) { return (!is_in && HALVES == 2) ? half : 0; };
    auto pos_hi = [&](bool is_in, int half) { return HALVES == 1 ? S : (is_in ? HW + 1 : half + HW); };
    float2 pf[NROUND][2];
    // what step u stages: input entry u (image u / P, row u % P - 1) and the gradient row of slab u - 2
    // (eb, gb count (image, half) pairs: image = eb / HALVES, half = eb % HALVES)
    auto load = [&](int eb, int er, int gb, int gy) {
        const bool in_ok = eb < nbv && er >= 0 && er < S, g_ok = gb >= 0 && gb < nbv && gy >= 0 && gy < S;
        const int ie = in_ok ? eb / HALVES : 0, ih = in_ok ? eb % HALVES : 0, ge = g_ok ? gb / HALVES : 0, gh = g_ok ? gb % HALVES : 0;
        const float *in_row = in + ((size_t)(wg + ie * wgs) * CIN * S + (in_ok ? er : 0)) * S + ih * (HW - 1);
        const float *g_row = gp + (((size_t)(wg + ge * wgs) * cout + co0) * S + (g_ok ? gy : 0)) * S + gh * (HW - 1);
#pragma unroll
        for (int r = 0; r < NROUND; ++r) {
            const int pidx = wave + r * C::WAVES;
            if (pidx >= NPI + NPG) continue;                             // (wave-uniform)
            const bool is_in = pidx < NPI;
            const float *src = (is_in ? in_row : g_row) + (st_src[r] >= 0 ? st_src[r] : 0);
            pf[r][0] = *reinterpret_cast<const float2 *>(src);
            pf[r][1] = *reinterpret_cast<const float2 *>(4 * q + 2 < pos_hi(is_in, is_in ? ih : gh) ? src + 2 : src);   // (never past the row / the tensor)
        }
    };
    auto store = [&](int slot_in, int slot_g, bool in_ok, bool g_ok, int in_half, int g_half) {
#pragma unroll
        for (int r = 0; r < NROUND; ++r) {
            const int pidx = wave + r * C::WAVES;
            if (pidx >= NPI + NPG) continue;
            const bool is_in = pidx < NPI;                               // (wave-uniform)
            const float sc = st_src[r] < 0 ? 0.0f : is_in ? (in_ok ? IN_SCALE : 0.0f) : (g_ok ? gscale : 0.0f);
            const int lo = pos_lo(is_in, is_in ? in_half : g_half), span = pos_hi(is_in, is_in ? in_half : g_half) - lo;
            auto at = [&](int j, float v) { return (unsigned)(4 * q + j - lo) < (unsigned)span ? v * sc : 0.0f; };
            const float v0 = at(0, pf[r][0].x), v1 = at(1, pf[r][0].y), v2 = at(2, pf[r][1].x), v3 = at(3, pf[r][1].y);
            const f16 h0 = (f16)v0, h1 = (f16)v1, h2 = (f16)v2, h3 = (f16)v3;
            const f16 l0 = (f16)((v0 - (float)h0) * LO_SCALE), l1 = (f16)((v1 - (float)h1) * LO_SCALE);
            const f16 l2 = (f16)((v2 - (float)h2) * LO_SCALE), l3 = (f16)((v3 - (float)h3) * LO_SCALE);
            const uint32_t H0 = pack2(h0, h1), H1 = pack2(h2, h3), L0 = pack2(l0, l1), L1 = pack2(l2, l3);
            const bool live = st_src[r] >= 0;
            if (is_in) {
                unsigned char *d = in_h + slot_in * C::IN_SLOT + st_dst[r];
                if (live) {
                    *reinterpret_cast<uint2 *>(d) = make_uint2(H0, H1);
                    *reinterpret_cast<uint2 *>(d + C::IN_HALF) = make_uint2(L0, L1);
                }
            } else {
                // copy kx holds g[x] at pixel x + kx - 1.  copy 1 = the row itself; copy 0 = one pixel to the left: this
                // item's pixels 1..3 and the right neighbour's pixel 0; copy 2 = one to the right: the left neighbour's
                // pixel 3 and this item's 0..2.  Neighbours are the adjacent lanes (row ends: zeros: a row's last item has
                // its tail masked to zero, and lane 0 / the first item of a row take an explicit zero).
                uint32_t RH = __builtin_amdgcn_update_dpp(0u, H0, 0x130, 0xf, 0xf, true);      // wave_shl:1: lane i <- lane i + 1
                uint32_t RL = __builtin_amdgcn_update_dpp(0u, L0, 0x130, 0xf, 0xf, true);
                uint32_t LH = __builtin_amdgcn_update_dpp(0u, H1, 0x138, 0xf, 0xf, true);      // wave_shr:1: lane i <- lane i - 1
                uint32_t LL = __builtin_amdgcn_update_dpp(0u, L1, 0x138, 0xf, 0xf, true);
                if (q == Q - 1) RH = 0u, RL = 0u;                        // (the row's last item: nothing to its right)
                if (q == 0) LH = 0u, LL = 0u;
                unsigned char *d = g_h + slot_g * C::G_SLOT + st_dst[r];
                if (live) {
                    const uint32_t mh = __builtin_amdgcn_alignbit(H1, H0, 16), ml = __builtin_amdgcn_alignbit(L1, L0, 16);   // {1, 2}
                    *reinterpret_cast<uint2 *>(d) = make_uint2(mh, __builtin_amdgcn_alignbit(RH, H1, 16));                  // copy 0
                    *reinterpret_cast<uint2 *>(d + C::G_HALF) = make_uint2(ml, __builtin_amdgcn_alignbit(RL, L1, 16));
                    *reinterpret_cast<uint2 *>(d + 2 * C::G_SLOT) = make_uint2(H0, H1);                                      // copy 1
                    *reinterpret_cast<uint2 *>(d + 2 * C::G_SLOT + C::G_HALF) = make_uint2(L0, L1);
                    *reinterpret_cast<uint2 *>(d + 4 * C::G_SLOT) = make_uint2(__builtin_amdgcn_alignbit(H0, LH, 16), mh);  // copy 2
                    *reinterpret_cast<uint2 *>(d + 4 * C::G_SLOT + C::G_HALF) = make_uint2(__builtin_amdgcn_alignbit(L0, LL, 16), ml);
                }
            }
        }
    };

    const unsigned char *a_base = g_h + (16 * ct + li) * ROWB + (PADL + 8 * g) * 2;     // + (2 kx + slot) * G_SLOT
    const unsigned char *b_base = in_h + (16 * it + li) * ROWB + (PADL + 8 * g) * 2;    // + slot * IN_SLOT

    // (image, index) of the entry staged at step u, and of steps u - 2 (gradient row) and u - 3 (the slab multiplied)
    int eb = 0, ei = 0;
    const int total = nbv * P + 3;
    load(0, -1, -1, 0);
    for (int u = 0; u < total; ++u) {
        const int u2 = u - 2, u3 = u - 3;
        const int gb = u2 >= 0 ? u2 / P : -1, gy = u2 >= 0 ? u2 - gb * P : 0;
        const int vb = u3 >= 0 ? u3 / P : -1, vy = u3 >= 0 ? u3 - vb * P : 0;
        // 1. this step's rows (loaded during the previous step) into their ring slots
        if (TRON_WR_ABLATE != 1)
            store(u & 3, u & 1, eb < nbv && ei >= 1 && ei <= S, gb >= 0 && gb < nbv && gy < S, eb % HALVES, gb >= 0 ? gb % HALVES : 0);
        // 2. the next step's rows: in flight under this step's MFMAs
        {
            int nb_e = eb, ni = ei + 1;
            if (ni == P) { ni = 0; ++nb_e; }
            const int n2 = u - 1;                                        // (u + 1) - 2
            const int ngb = n2 >= 0 ? n2 / P : -1, ngy = n2 >= 0 ? n2 - ngb * P : 0;
            if (TRON_WR_ABLATE != 3) load(nb_e, ni - 1, ngb, ngy);
            eb = nb_e;
            ei = ni;
        }
        // 3. the slab staged three steps ago: g row vy of image vb (ring slot of step u - 1) x input entries u-3, u-2, u-1
        if (vb >= 0 && vb < nbv && vy < S && TRON_WR_ABLATE != 2) {
            const int gs = (u - 1) & 1;
            f16x8 ah[3], al[3], bh[3], bl[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                ah[k] = *reinterpret_cast<const f16x8 *>(a_base + (2 * k + gs) * C::G_SLOT);
                al[k] = *reinterpret_cast<const f16x8 *>(a_base + (2 * k + gs) * C::G_SLOT + C::G_HALF);
                const int slot = (u3 + k) & 3;
                bh[k] = *reinterpret_cast<const f16x8 *>(b_base + slot * C::IN_SLOT);
                bl[k] = *reinterpret_cast<const f16x8 *>(b_base + slot * C::IN_SLOT + C::IN_HALF);
            }
            // three phases, hi*lo first and lo*hi last: both add to the same accumulator, and a dependent MFMA issues 48
            // cycles after its producer (an independent one 16)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    acc1[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kx], bl[ky], acc1[ky * 3 + kx], 0, 0, 0);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    acc0[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kx], bh[ky], acc0[ky * 3 + kx], 0, 0, 0);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    acc1[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[kx], bh[ky], acc1[ky * 3 + kx], 0, 0, 0);
        }
        if (TRON_WR_ABLATE != 4) __syncthreads();                        // this step's writes before the next step's reads; its reads before their slot is rewritten
    }

    // D row = 4 g + r = co within the tile, column = li = ci within the tile
    const float unscale = 64.0f / gscale;
    float *dst = partial + ((size_t)wg * cout + co0) * CIN * 9;
    const int ci = 16 * it + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float *row = dst + ((size_t)(16 * ct + 4 * g + r) * CIN + ci) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) row[k] = (acc0[k][r] + acc1[k][r] * (1.0f / LO_SCALE)) * unscale;
    }
}

template <class C>
int launch_rows(const float *in, const float *gp, const float *absmax, int n_absmax, float *partial, int64_t batch, int cout,
                int grid_max, int *nparts, hipStream_t st)
{
    static uint64_t prepared = 0;                                       // one bit per device: the attribute is per device
    auto kern = k_wgrad_rows<C>;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_NO_DEVICE; }
    if (!(prepared & (1ull << (dev & 63)))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess)
            (void)hipGetLastError();
        prepared |= 1ull << (dev & 63);
    }
    const int nhalves = cout / C::COW;
    const int wgs = (int)(batch < grid_max / nhalves ? batch : grid_max / nhalves);
    hipLaunchKernelGGL(kern, dim3(wgs * nhalves), dim3(C::THREADS), C::LDS, st, in, gp, absmax, n_absmax, partial, (int)batch, cout);
    *nparts = wgs;
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

}  // namespace

// called by tron_conv3x3_wgrad (tron_conv_wgrad.hip) for sides it has no strip layout for; writes `*nparts` partial sums
// [part][cout][cin][9] to `partial`.  TRON_ERR_UNSUPPORTED = not this shape.
int tron_wgrad_rows(const float *in, const float *gp, const float *absmax, int n_absmax, float *partial, int64_t batch, int cin,
                    int cout, int side, int grid_max, int *nparts, hipStream_t st)
{
    if (side == 34) {                                                    // 32x32 boards: two column halves per row
        if (cin == 64 && cout == 64) return launch_rows<RCfg<34, 64, 32, 2>>(in, gp, absmax, n_absmax, partial, batch, cout, grid_max, nparts, st);
        if (cin == 32 && cout == 64) return launch_rows<RCfg<34, 32, 64, 2>>(in, gp, absmax, n_absmax, partial, batch, cout, grid_max, nparts, st);
        if (cin == 32 && cout == 32) return launch_rows<RCfg<34, 32, 32, 2>>(in, gp, absmax, n_absmax, partial, batch, cout, grid_max, nparts, st);
        return TRON_ERR_UNSUPPORTED;
    }
    if (side != 26) return TRON_ERR_UNSUPPORTED;
    if (cin == 64 && cout == 64) return launch_rows<RCfg<26, 64, 32>>(in, gp, absmax, n_absmax, partial, batch, cout, grid_max, nparts, st);
    if (cin == 32 && cout == 64) return launch_rows<RCfg<26, 32, 64>>(in, gp, absmax, n_absmax, partial, batch, cout, grid_max, nparts, st);
    if (cin == 32 && cout == 32) return launch_rows<RCfg<26, 32, 32>>(in, gp, absmax, n_absmax, partial, batch, cout, grid_max, nparts, st);
    return TRON_ERR_UNSUPPORTED;
}
