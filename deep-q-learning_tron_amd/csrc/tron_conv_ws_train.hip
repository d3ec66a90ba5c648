// tron_conv_ws_train.hip — the LEARNER's side of the weight-stationary design (DDQN.py:115-151 on DQNNet.py:33-50): the trunk's
// forward, input gradients and weight gradients with every tensor between two layers a PX16 image — no f32 NCHW activation,
// pre-activation or gradient tensor exists between conv1 and conv6.
//
//   forward    k_conv_ws<WS_TRAIN> (tron_conv_ws_kernel.hpp): the gradient-free chain's kernel, which also keeps each layer's
//              pre-activation z as a PX16 image (what mish'(z) is computed from on the way back);
//   gradient images: a gradient tensor g is stored as PX16 of g * s, s a power of two kept beside the image in a 4-float
//              record {s, 1 / s, max |g|, -} on the device.  Gradients are 1e-6-sized after a mean-reduced loss: s brings them
//              into f16's range.  s must be known BEFORE the kernel that writes the image runs, so it is chosen from a bound:
//              max |conv^T(g) + r| <= max |g| * (largest absolute row sum of the rotated weights) + max |r|, times mish' <= 1.1;
//              s = the power of two that maps that bound to at most 2^15 — no overflow, and the true maximum (typically 10-30x
//              below the bound) sits ~2^10 above f16's normal minimum: hi + lo keep 22 bits wherever it matters;
//   k_gout_px  the chain's entry: g (f32 NCHW, from the head's backward) * mish'(z6) -> gradient image + bias sums + max;
//   input gradient   k_conv_ws<WS_BWD>: the same loop on the rotated, transposed weights, epilogue = (+ residual gradient) *
//              mish'(z below), column sums (the layer below's bias gradient), max, PX16 store at the next scale;
//   weight gradient  k_wgrad_px: dW[co][ci][tap] = sum over pixels of g[co][p] * a[ci][p + tap] straight from the two
//              pixel-major images: K = pixels, both MFMA operands read TRANSPOSED out of LDS (ds_read_b64_tr_b16: a lane
//              supplies the address of one pixel row and receives one channel column, so the nine taps are address offsets
//              into a zero-haloed window and no operand is ever converted, shifted in registers or im2col-ed).
#include "tron_conv_ws_kernel.hpp"

namespace {

// ---- scales ------------------------------------------------------------------------------------------------------------
// largest absolute row sum of the rotated weights of a layer: max over the backward convolution's output channel (= the
// forward layer's input channel ci) of sum_{co, tap} |W[co][ci][tap]|
__global__ __launch_bounds__(64) void k_ws_wnorm(WsJobs jobs, float *__restrict__ wnorm)
{
    const int k = blockIdx.x, cinF = jobs.cout[k], coutF = jobs.cin[k];  // (rot jobs carry the backward convolution's cout / cin)
    const float *__restrict__ w = jobs.w[k];
    float m = 0.0f;
    for (int ci = threadIdx.x; ci < cinF; ci += 64) {
        float s = 0.0f;
        for (int co = 0; co < coutF; ++co)
            for (int t = 0; t < 9; ++t) s += __builtin_fabsf(w[((size_t)co * cinF + ci) * 9 + t]);
        m = __builtin_fmaxf(m, s);
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) m = __builtin_fmaxf(m, __shfl_xor(m, d, 64));
    if (threadIdx.x == 0) wnorm[k] = m;
}

// stats[2][groups][C] of a k_conv_ws<WS_BWD> (or k_gout_px) launch -> bias_grad[c] = the column sums and info[2] = the largest
// magnitude written.  One workgroup per channel, a thread per group, fixed-order tree: deterministic.  (The first version — one
// workgroup, one thread per channel walking the groups — took 78 us per call: 256 dependent loads per thread.)
__global__ __launch_bounds__(256) void k_wsb_finish(const float *__restrict__ stats, int groups, int C, float *__restrict__ bias_grad,
                                                    float *__restrict__ chmax)
{
    __shared__ float rs[256], rm[256];
    const int c = blockIdx.x;
    float s = 0.0f, m = 0.0f;
    for (int q = threadIdx.x; q < groups; q += 256) {
        s += stats[(size_t)q * C + c];
        m = __builtin_fmaxf(m, stats[((size_t)groups + q) * C + c]);
    }
    rs[threadIdx.x] = s;
    rm[threadIdx.x] = m;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) {
            rs[threadIdx.x] += rs[threadIdx.x + d];
            rm[threadIdx.x] = __builtin_fmaxf(rm[threadIdx.x], rm[threadIdx.x + d]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (bias_grad) bias_grad[c] = rs[0];
        chmax[c] = rm[0];
    }
}

// ---- the chain's entry: gp = g * mish'(z) as a gradient image ----------------------------------------------------------
// g: f32 [B][C][SS] (the gradient at the trunk's output, from the head's backward), z: PX16 pre-activation of the last layer.
// blockIdx = (image group, channel octet); a thread walks pixels of its group's images: 8 channels x one pixel at a time
// (coalesced along the pixels of each channel row), keeps the 8 column sums and the maximum, and the block leaves them in
// stats[2][groups][C] (k_wsb_finish).  info_in = tron_absmax_pow2's {s, max |g|, ...}: s brings max |g| below 2^14, and
// |mish'| <= 1.1 keeps the product below 2^15.
//
// SRC: where g comes from.  0: f32 [B][C][SS] planes.  1 / 2: the gradient of the POOLED planes (AvgPool2d(3, 2, 1), DQNNet.py:52)
// — the pooling's backward is taken on the fly, g[y][x] = (1/9) sum of the windows that contain (y, x) (at most four), so the
// f32 gradient planes of the trunk's output are never written: 1 = pooled planes f32 [B][C][PS*PS] (12x12: the dense conv7 path),
// 2 = channels-last f32 [B][PS*PS][C] (26x26: the DGRAD7 GEMMs' output).  scale4[0] must then bring (4/9) max |pooled g| * 1.1
// below 2^15: tron_absmax_pow2(pooled g, 15).
constexpr int GOUT_TILE = 16 * 36 * 8;                                   // floats: 16 images of 6x6 (or one of up to 24x24: 13x13 here)
__host__ __device__ inline int gout_images(int S) { return S <= 12 ? 16 : 1; }   // (26x26: 676 pixels fill the threads; four at a time measured 6 % slower)
template <int SRC>
__global__ __launch_bounds__(256) void k_gout_px(const float *__restrict__ gout, const unsigned char *__restrict__ z, int64_t B, int C,
                                                 int S, const float *__restrict__ scale4, unsigned char *__restrict__ out,
                                                 float *__restrict__ stats, float *__restrict__ info)
{
    const int SS = S * S, PS = S / 2, PP = PS * PS;
    __shared__ float red[256 * 9];
    // (SRC 1 / 2) the pooled gradients of NI images' octet: [image][pooled pixel][8 channels].  NI images are walked as one index
    // space (image, pixel): a 12x12 image alone keeps 144 of the 256 threads busy and costs two barriers — 124 us at 4 096 images,
    // 86 us with 16 at a time (profiles/r04_gout_px.txt)
    __shared__ __attribute__((aligned(16))) float tile[SRC ? GOUT_TILE : 8];
    const int NI = SRC ? gout_images(S) : 1;
    const int oct = blockIdx.y, groups = gridDim.x, grp = blockIdx.x;
    const float s = scale4[0];
    if (grp == 0 && oct == 0 && threadIdx.x == 0) { info[0] = s; info[1] = 1.0f / s; }
    const size_t half = (size_t)(C / 8) * SS * 16;
    float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, mx = 0.0f;
    for (int64_t img0 = (int64_t)grp * NI; img0 < B; img0 += (int64_t)groups * NI) {   // this block's images, NI at a time
        const int ni = B - img0 < NI ? (int)(B - img0) : NI;
        if (SRC) {
            __syncthreads();                                             // (the previous images' tiles are no longer read)
            if (SRC == 1) {
                for (int i = threadIdx.x; i < ni * 8 * PP; i += 256) {   // planes [B][C][PP]: the octet's 8 planes are contiguous
                    const int il = i / (8 * PP), r = i - il * 8 * PP, j = r / PP, pp = r - j * PP;
                    tile[(il * PP + pp) * 8 + j] = gout[((size_t)(img0 + il) * C + oct * 8) * PP + r];
                }
            } else {
                for (int i = threadIdx.x; i < ni * 2 * PP; i += 256) {   // channels-last [B][PP][C]: 8 consecutive floats per pooled pixel
                    const int il = i / (2 * PP), r = i - il * 2 * PP, pp = r >> 1, q = r & 1;
                    *reinterpret_cast<f32x4 *>(tile + (il * PP + pp) * 8 + 4 * q) =
                        *reinterpret_cast<const f32x4 *>(gout + ((size_t)(img0 + il) * PP + pp) * C + oct * 8 + 4 * q);
                }
            }
            __syncthreads();
        }
        for (int ip = threadIdx.x; ip < ni * SS; ip += 256) {
            const int il = ip / SS, p = ip - il * SS;
            const int64_t img = img0 + il;
            const unsigned char *zp = z + (size_t)img * 2 * half + ((size_t)oct * SS + p) * 16;
            const f16x8 zh = *reinterpret_cast<const f16x8 *>(zp), zl = *reinterpret_cast<const f16x8 *>(zp + half);
            float gg[8];
            if (SRC == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) gg[j] = gout[((size_t)img * C + oct * 8 + j) * SS + p];
            } else {
                // window row py covers input rows 2 py - 1 .. 2 py + 1: an even y belongs to py = y / 2 only, an odd one to
                // (y - 1) / 2 and (y + 1) / 2 (the latter if it exists); columns alike
                const int y = p / S, x = p - y * S;
                const int py0 = y >> 1, px0 = x >> 1;
                const bool two_y = (y & 1) && py0 + 1 < PS, two_x = (x & 1) && px0 + 1 < PS;
                const float *t00 = tile + (il * PP + py0 * PS + px0) * 8;
                const float wx = two_x ? 1.0f : 0.0f, wy = two_y ? 1.0f : 0.0f;
                const float *t01 = two_x ? t00 + 8 : t00, *t10 = two_y ? t00 + PS * 8 : t00, *t11 = (two_x && two_y) ? t00 + PS * 8 + 8 : t00;
#pragma unroll
                for (int j = 0; j < 8; ++j) gg[j] = (t00[j] + wx * t01[j] + wy * t10[j] + wx * wy * t11[j]) * (1.0f / 9.0f);
            }
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float zz = ((float)zh[j] + (float)zl[j] * LO_UNSCALE) * ACT_UNSCALE;
                const uint32_t eb = __float_as_uint(__builtin_amdgcn_exp2f(zz * 1.44269504088896341f));
                const float e = __uint_as_float(eb < 0x5D5E0B6Bu ? eb : 0x5D5E0B6Bu);
                const float n = e * (e + 2.0f), r = __builtin_amdgcn_rcpf(n + 2.0f), t = n * r;
                const float m = t + zz * ((r + r) * (t + 1.0f)) * (e * __builtin_amdgcn_rcpf(e + 1.0f));
                v[j] = gg[j] * m;
                sum[j] += v[j];
                mx = __builtin_fmaxf(mx, __builtin_fabsf(v[j]));
                v[j] *= s * ACT_SCALE;
            }
            f16x8 hh, ll;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                hh[j] = (f16)v[j];
                ll[j] = (f16)((v[j] - (float)hh[j]) * LO_SCALE);
            }
            unsigned char *op = out + (size_t)img * 2 * half + ((size_t)oct * SS + p) * 16;
            *reinterpret_cast<f16x8 *>(op) = hh;
            *reinterpret_cast<f16x8 *>(op + half) = ll;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[j * 256 + threadIdx.x] = sum[j];
    red[8 * 256 + threadIdx.x] = mx;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) {
#pragma unroll
            for (int j = 0; j < 8; ++j) red[j * 256 + threadIdx.x] += red[j * 256 + threadIdx.x + d];
            red[8 * 256 + threadIdx.x] = __builtin_fmaxf(red[8 * 256 + threadIdx.x], red[8 * 256 + threadIdx.x + d]);
        }
        __syncthreads();
    }
    if (threadIdx.x < 8) {
        stats[(size_t)grp * C + oct * 8 + threadIdx.x] = red[threadIdx.x * 256];
        stats[((size_t)groups + grp) * C + oct * 8 + threadIdx.x] = red[8 * 256];
    }
}

// ---- the weight gradient from two PX16 images ----------------------------------------------------------------------------
// GEMM view: M = output channels (A = the gradient image), N = input channels (B = the activation image, shifted per tap),
// K = pixels.  Geometry of one instantiation:
//   S     image side;  NI images stacked per "stack" (12x12: 2, so that 8-row bands are whole 32-pixel slabs: 96 = 3 x 32);
//   R     rows of the stack per band; a band's gradient pixels are contiguous in memory, its activation window brings one
//         halo row above and below (and the all-zero separator row between two stacked images);
//   COT / CIT  the channels of each operand one workgroup holds (a 64-channel layer at 26x26 is split over two workgroup
//         kinds by input channels: 64 + 64 channels of both operands, double-buffered, do not fit 160 KB);
//   a wave owns a (32 co) x (16 ci) block and all nine taps — 2 x 9 x 2 accumulator tiles = 144 registers — and the slabs
//   s = kgroup, kgroup + KG, ... of every item; waves = blocks x KG = 8.
// A workgroup serves ONE band index (its DMA source table, built once, is the band's), bands get workgroups in proportion to
// their slabs.  An item = (stack, band): both operands go global -> LDS by LDS-DMA through a per-unit source table (a unit
// whose destination is halo, separator, padding pixel or plane padding copies from a zero page), double-buffered one item
// ahead, one barrier per item, the next item's pieces riding between the taps' MFMAs.
template <int S_, int NI_, int R_, int CO_, int CI_, int COT_, int CIT_, int KG_>
struct WCfg {
    static constexpr int S = S_, NI = NI_, R = R_, CO = CO_, CI = CI_, COT = COT_, CIT = CIT_, KG = KG_;   // (CO / CI: the layer's; COT / CIT: a workgroup's)
    static constexpr int SS = S * S, ROWS = NI * S, NB = (ROWS + R - 1) / R;
    static constexpr int GPX = R * S, NSLAB = (GPX + 31) / 32, GP = NSLAB * 32;
    static constexpr int WR = R + 2 + (NI - 1), WC = S + 2;
    static constexpr int odd128(int x) { return ((x + 127) / 256) * 256 + 128; }     // smallest y >= x with y = 128 (mod 256)
    static constexpr int G_PLANE = odd128(GP * 16), A_PLANE = odd128(WR * WC * 16);
    static constexpr int G_HALF = (COT / 8) * G_PLANE, A_HALF = (CIT / 8) * A_PLANE;
    static constexpr int A_OFF = 2 * G_HALF;
    static constexpr int ITEM = (2 * G_HALF + 2 * A_HALF + 1023) / 1024 * 1024;
    static constexpr int NPIECE = ITEM / 1024, PPW = (NPIECE + 7) / 8;
    static constexpr int TAB_OFF = 2 * ITEM, TAB_BYTES = 8 * PPW * 128;
    static constexpr int LDS = TAB_OFF + TAB_BYTES + 1024;               // + the sink of a piece past the item's end
    static constexpr int NBLK = (COT / 32) * (CIT / 16);
    static_assert(NBLK * KG == 8, "eight waves");
    static_assert(LDS <= 160 * 1024, "LDS");
    static_assert(2 * NI * 8 * SS < 0xFFFF, "source offsets in 16 bits");
};

// row index (with halo and separator rows counted) of stack row t: image i = t / S occupies e = i (S + 1) + 1 ... i (S + 1) + S
template <class C> __device__ __forceinline__ int stack_e(int t) { return t + t / C::S + 1; }

struct WBands { int n[8]; };                                             // workgroups per band

template <class C>
__global__ __launch_bounds__(512, 2) void k_wgrad_px(const unsigned char *__restrict__ gimg, const unsigned char *__restrict__ aimg,
                                                     int B, int kinds, WBands bands, float *__restrict__ partial)
{
    constexpr int S = C::S, SS = C::SS, CO = C::CO, CI = C::CI;
    extern __shared__ __attribute__((aligned(256))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, g = lane >> 4;
    // workgroup kinds = the (co, ci) channel tiles of a layer one workgroup does not hold whole.  The kinds of one item sit 8
    // blocks apart — the same XCD under round-robin placement — so the operand they share is read from memory once (speed only).
    const int kind = ((int)blockIdx.x >> 3) % kinds, wg = (((int)blockIdx.x >> 3) / kinds) * 8 + ((int)blockIdx.x & 7);
    const int nci = CI / C::CIT, co0 = (kind / nci) * C::COT, ci0 = (kind % nci) * C::CIT;
    // this workgroup's band, its index among the band's workgroups and their number
    int band = 0, wb = wg, nwb = bands.n[0];
    while (wb >= nwb && band + 1 < C::NB) { wb -= nwb; ++band; nwb = bands.n[band]; }
    if (wb >= nwb) return;                                               // (the grid is rounded up to whole groups of 8 x kinds)
    const int r0 = band * C::R, nrows = (C::ROWS - r0 < C::R) ? C::ROWS - r0 : C::R, npx = nrows * S;
    const int nslab = (npx + 31) >> 5;
    const int e0 = stack_e<C>(r0) - 1, e_last = stack_e<C>(r0 + nrows - 1) + 1;   // the window's first and last row (halo rows included)
    const int blk = wave % C::NBLK, kgroup = wave / C::NBLK;
    const int cot2 = blk / (C::CIT / 16), cit = blk % (C::CIT / 16);

    for (int i = tid * 16; i < C::TAB_OFF; i += 512 * 16) *reinterpret_cast<uint4 *>(lds + i) = make_uint4(0u, 0u, 0u, 0u);

    // the DMA source table: unit u (16 bytes) of the item buffer <- unit tab[u] of the stack's gradient / activation images,
    // or 0xFFFF: zeros.  Piece j of this wave = piece wave + 8 j of the item.
    uint16_t *tab = reinterpret_cast<uint16_t *>(lds + C::TAB_OFF) + wave * C::PPW * 64;
    const size_t g_img_units = (size_t)2 * (CO / 8) * SS, a_img_units = (size_t)2 * (CI / 8) * SS;   // 16-byte units of one image
    for (int j = 0; j < C::PPW; ++j) {
        const int d = (wave + 8 * j) * 1024 + lane * 16;
        int off = -1;
        if (d < C::A_OFF) {                                              // gradient: [hi | lo][octet][pixel of the band]
            const int h = d / C::G_HALF, d1 = d - h * C::G_HALF, o = d1 / C::G_PLANE, px = (d1 - o * C::G_PLANE) >> 4;
            if (px < npx) {
                const int t = r0 + px / S, img = t / S, y = t - img * S, x = px - (px / S) * S;
                off = (int)(img * g_img_units) + (h * (CO / 8) + co0 / 8 + o) * SS + y * S + x;
            }
        } else if (d < C::A_OFF + 2 * C::A_HALF) {                       // activation window: [hi | lo][octet][row][column with halo]
            const int da = d - C::A_OFF, h = da / C::A_HALF, d1 = da - h * C::A_HALF, o = d1 / C::A_PLANE, u = (d1 - o * C::A_PLANE) >> 4;
            const int wr = u / C::WC, wc = u - wr * C::WC, e = e0 + wr;
            const int img = e / (S + 1), y = e - img * (S + 1) - 1;      // y = -1: a halo / separator row
            if (e <= e_last && wc >= 1 && wc <= S && y >= 0 && img < C::NI)
                off = (int)(img * a_img_units) + (h * (CI / 8) + ci0 / 8 + o) * SS + y * S + (wc - 1);
        }
        tab[j * 64 + lane] = (uint16_t)(off < 0 ? 0xFFFF : off);
    }

    const int nstack = (B + C::NI - 1) / C::NI;
    auto dma_piece = [&](int stack, int buf, int j) {                    // (a stack past the last one: zeros — selects, no branch around the copy)
        const int q = wave + 8 * j;
        const uint32_t o16 = tab[j * 64 + lane];
        const bool is_g = q * 1024 + lane * 16 < C::A_OFF;
        // the second image of the last stack may not exist: its units read zeros
        const size_t per = is_g ? g_img_units : a_img_units;
        const bool ok = stack < nstack && o16 != 0xFFFFu && (C::NI == 1 || (int64_t)stack * C::NI + (o16 >= per ? 1 : 0) < B);
        const unsigned char *base = is_g ? gimg + (size_t)stack * C::NI * g_img_units * 16 : aimg + (size_t)stack * C::NI * a_img_units * 16;
        const unsigned char *src = ok ? base + (size_t)o16 * 16 : reinterpret_cast<const unsigned char *>(g_ws_zero) + lane * 16;
        unsigned char *dst = q < C::NPIECE ? lds + buf * C::ITEM + q * 1024 : lds + C::TAB_OFF + C::TAB_BYTES;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    };

    // LDS byte offsets (within an item buffer) of the pixel rows this lane supplies to the transposed reads.  The k a lane
    // holds in element e of a slab's fragment is 16 (e >> 2) + 4 g + (e & 3) (any assignment works if both operands share it):
    // read j of a slab covers pixels 32 s + 16 j + 4 g + (li >> 2), this lane supplies the 8 bytes (li & 3) of that pixel's row.
    // Gradient planes are dense: slab s = + 512 s.  Activation rows have halo columns: one base per (slab, read).
    const int q4 = li >> 2, p4 = li & 3;
    int g_addr[2], a_addr[C::NSLAB][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        g_addr[j] = (4 * cot2 + (p4 >> 1)) * C::G_PLANE + (16 * j + 4 * g + q4) * 16 + (p4 & 1) * 8;       // co tile t: + 2 t G_PLANE
#pragma unroll
        for (int s = 0; s < C::NSLAB; ++s) {
            int px = 32 * s + 16 * j + 4 * g + q4;
            px = px < npx ? px : 0;                                      // (padding pixels: the gradient is zero there, any finite operand will do)
            const int t = r0 + px / S, x = px - (px / S) * S;
            a_addr[s][j] = C::A_OFF + (2 * cit + (p4 >> 1)) * C::A_PLANE + ((stack_e<C>(t) - e0) * C::WC + x + 1) * 16 + (p4 & 1) * 8;
        }
    }

    f32x4 acc0[9][2], acc1[9][2];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc0[k][t] = acc1[k][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    __syncthreads();                                                     // zeros and tables are in before any copy lands
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds;
    int stack = wb;
    if (stack < nstack)
        for (int j = 0; j < C::PPW; ++j) dma_piece(stack, 0, j);
    for (int cur = 0; stack < nstack; stack += nwb, cur ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // this item is in; everybody is done with the other buffer
        const int nxt = stack + nwb;
        const uint32_t I = lds_base + cur * C::ITEM;                     // LDS byte address of this item's buffer
        int pj = 0;                                                      // the next item's pieces ride between the taps: piece 9 si + tap at (slab si, tap)
#pragma unroll
        for (int si = 0; si < (C::NSLAB + C::KG - 1) / C::KG; ++si) {
            const int s = kgroup + si * C::KG;
            if (s < nslab) {
                s16x4 rg[8], ra[2][4];                                   // raw halves: gradient (2 tiles x hi / lo x 2 reads), activation (ring of two taps)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    rg[4 * t + 0] = lds_tr(I + g_addr[0] + s * 512 + 2 * t * C::G_PLANE);
                    rg[4 * t + 1] = lds_tr(I + g_addr[1] + s * 512 + 2 * t * C::G_PLANE);
                    rg[4 * t + 2] = lds_tr(I + C::G_HALF + g_addr[0] + s * 512 + 2 * t * C::G_PLANE);
                    rg[4 * t + 3] = lds_tr(I + C::G_HALF + g_addr[1] + s * 512 + 2 * t * C::G_PLANE);
                }
                int a0 = a_addr[0][0], a1 = a_addr[0][1];
#pragma unroll
                for (int q = 1; q < C::NSLAB; ++q)
                    if (s == q) { a0 = a_addr[q][0]; a1 = a_addr[q][1]; }
                auto read_tap = [&](int slot, int tap) {
                    const int sh = (tap / 3 - 1) * C::WC * 16 + (tap % 3 - 1) * 16;
                    ra[slot][0] = lds_tr(I + a0 + sh);
                    ra[slot][1] = lds_tr(I + a1 + sh);
                    ra[slot][2] = lds_tr(I + C::A_HALF + a0 + sh);
                    ra[slot][3] = lds_tr(I + C::A_HALF + a1 + sh);
                };
                read_tap(0, 0);
                f16x8 gh[2], gl[2];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    lds_wait();                                          // this tap's operands (and, at tap 0, the gradient's) have landed
                    if (tap == 0) {
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            gh[t] = join8(rg[4 * t + 0], rg[4 * t + 1]);
                            gl[t] = join8(rg[4 * t + 2], rg[4 * t + 3]);
                        }
                    }
                    const f16x8 bh = join8(ra[tap & 1][0], ra[tap & 1][1]), bl = join8(ra[tap & 1][2], ra[tap & 1][3]);
                    if (tap < 8) read_tap((tap + 1) & 1, tap + 1);
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc1[tap][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gh[t], bl, acc1[tap][t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc0[tap][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gh[t], bh, acc0[tap][t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc1[tap][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gl[t], bh, acc1[tap][t], 0, 0, 0);
                    if (si * 9 + tap < C::PPW) { dma_piece(nxt, cur ^ 1, si * 9 + tap); pj = si * 9 + tap + 1; }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        for (; pj < C::PPW; ++pj) dma_piece(nxt, cur ^ 1, pj);           // (waves with few or no slabs of this band)
    }

    // D row = 4 g + r (output channel within the tile), column = li (input channel): partial[part][tap][co][ci], part = (workgroup, kgroup)
    float *out = partial + ((size_t)wg * C::KG + kgroup) * 9 * CO * CI;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 v = acc0[tap][t] + acc1[tap][t] * LO_UNSCALE;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[((size_t)tap * CO + co0 + 32 * cot2 + 16 * t + 4 * g + r) * CI + ci0 + 16 * cit + li] = v[r];
        }
}

// dW[co][ci][tap] = (2^12 / s) * sum over the parts of partial[part][tap][co][ci] (both operands carry 2^-6, the gradient its
// scale s on top).  A workgroup = 64 consecutive outputs (16 threads x float4) x 16 part groups (parts q, q + 16, ...: two chains
// each), joined in a fixed order: deterministic, wide loads, thousands of independent loads in flight.
__global__ __launch_bounds__(256) void k_wgrad_px_finish(const float *__restrict__ partial, int nparts, int CO, int CI,
                                                         const float *__restrict__ ginfo, float *__restrict__ gw)
{
    __shared__ f32x4 red[256];
    const int total = 9 * CO * CI, col = threadIdx.x & 15, pg = threadIdx.x >> 4, i = blockIdx.x * 64 + 4 * col;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    if (i < total) {                                                     // (total is a multiple of 4: whole float4s)
        int p = pg;
        for (; p + 16 < nparts; p += 32) {
            a0 += *reinterpret_cast<const f32x4 *>(partial + (size_t)p * total + i);
            a1 += *reinterpret_cast<const f32x4 *>(partial + (size_t)(p + 16) * total + i);
        }
        if (p < nparts) a0 += *reinterpret_cast<const f32x4 *>(partial + (size_t)p * total + i);
    }
    red[threadIdx.x] = a0 + a1;
    __syncthreads();
    if (threadIdx.x < 16 && i < total) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 16; q += 4) v += (red[col + 16 * q] + red[col + 16 * (q + 1)]) + (red[col + 16 * (q + 2)] + red[col + 16 * (q + 3)]);
        const float unscale = 4096.0f * ginfo[1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = i + r, tap = o / (CO * CI), rem = o - tap * CO * CI, co = rem / CI, ci = rem - co * CI;
            gw[((size_t)co * CI + ci) * 9 + tap] = v[r] * unscale;
        }
    }
}

template <class C>
int launch_wgrad_px(const void *gimg, const void *aimg, const float *ginfo, int64_t B, int CO, int CI, float *partial, float *gw,
                    hipStream_t st, int64_t *ws_need)
{
    static_assert(C::NB <= 8, "bands");
    const int kinds = (CO / C::COT) * (CI / C::CIT);                     // workgroup kinds: channel tiles of both operands
    const int cus = device_cus();
    // workgroups per band in proportion to the band's slabs; every kind gets the same split
    int per_kind = cus / kinds;
    if (per_kind < C::NB) per_kind = C::NB;
    int slabs[8], tot = 0, nwg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < C::NB; ++b) {
        const int r0 = b * C::R, nr = C::ROWS - r0 < C::R ? C::ROWS - r0 : C::R;
        slabs[b] = (nr * C::S + 31) / 32;
        tot += slabs[b];
    }
    const int64_t nstack = (B + C::NI - 1) / C::NI;
    int used = 0;
    for (int b = 0; b < C::NB; ++b) {
        int n = per_kind * slabs[b] / tot;
        n = n < 1 ? 1 : n;
        if (n > nstack) n = (int)nstack;
        nwg[b] = n;
        used += n;
    }
    const int64_t parts = (int64_t)used * C::KG;                         // per kind: disjoint (co, ci) tiles, so the kinds share the parts
    if (ws_need) { *ws_need = parts * 9 * CO * CI * (int64_t)sizeof(float); return TRON_OK; }
    static uint64_t prepared = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_NO_DEVICE; }
    if (!(prepared & (1ull << (dev & 63)))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_wgrad_px<C>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess)
            (void)hipGetLastError();
        prepared |= 1ull << (dev & 63);
    }
    WBands bands;
    for (int b = 0; b < 8; ++b) bands.n[b] = nwg[b];
    const int grid = (used + 7) / 8 * 8 * kinds;                        // (wg = 8 (block / (8 kinds)) + block % 8: whole groups of 8 per kind)
    hipLaunchKernelGGL(k_wgrad_px<C>, dim3((unsigned)grid), dim3(512), C::LDS, st, reinterpret_cast<const unsigned char *>(gimg),
                       reinterpret_cast<const unsigned char *>(aimg), (int)B, kinds, bands, partial);
    hipLaunchKernelGGL(k_wgrad_px_finish, dim3((unsigned)((9 * CO * CI + 63) / 64)), dim3(256), 0, st, partial, (int)parts, CO, CI, ginfo, gw);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

template <class F>
int dispatch_wgrad_px(int side, int cin, int cout, F &&f)
{
    if (side == 12) {
        if (cin == 64 && cout == 64) return f(WCfg<12, 2, 8, 64, 64, 64, 64, 1>{});
        if (cin == 32 && cout == 64) return f(WCfg<12, 2, 8, 64, 32, 64, 32, 2>{});
        if (cin == 32 && cout == 32) return f(WCfg<12, 2, 8, 32, 32, 32, 32, 4>{});
    } else if (side == 26) {
        if (cin == 64 && cout == 64) return f(WCfg<26, 1, 6, 64, 64, 64, 32, 2>{});
        if (cin == 32 && cout == 64) return f(WCfg<26, 1, 6, 64, 32, 64, 32, 2>{});
        if (cin == 32 && cout == 32) return f(WCfg<26, 1, 6, 32, 32, 32, 32, 4>{});
    } else if (side == 34) {                                             // (32 + 32 channels of both operands per workgroup kind: 2 x 59 KB of LDS)
        if (cin == 64 && cout == 64) return f(WCfg<34, 1, 5, 64, 64, 32, 32, 4>{});
        if (cin == 32 && cout == 64) return f(WCfg<34, 1, 5, 64, 32, 32, 32, 4>{});
        if (cin == 32 && cout == 32) return f(WCfg<34, 1, 5, 32, 32, 32, 32, 4>{});
    }
    return TRON_ERR_UNSUPPORTED;
}

// ---- conv1's weight gradient from the int8 codes and the gradient image ---------------------------------------------------
// dW[co][plane][tap] = sum over images and pixels of g[co][p] * plane value at p + tap (util.pop_up's planes of the codes: wall 1;
// own body 1 / head 10; enemy body 1 / head 10; a fourth plane = plane4 on every cell, game.py:124-132).  GEMM view: M = the 32
// output channels (A = the gradient image by transposed LDS reads, as k_wgrad_px), N = the (plane, tap) pairs — 27 or 36 of 48
// columns — K = pixels; the B operand does not exist in memory: a lane assembles its eight f16 plane values per slab from a
// small f16 plane window in LDS that is rebuilt from the codes with every item (0, 1 and 10 are exact in f16: no lo half, two
// MFMAs per product) — kept in three copies shifted by 0, 1 and 2 columns, so that the four consecutive cells a lane needs under
// any tap are ONE aligned 8-byte read (eight 2-byte reads per fragment cost 45 % of the kernel).  A slab = 32 k positions = one image row (26x26) or two (12x12) padded to 32 / 16: positions past the row
// read the gradient plane's zero unit.  An item = an image (12x12) or half of one (13 rows of 26x26); the next item's loads are in
// flight (in registers) under this item's slabs and go to the one LDS buffer behind a barrier — 22 / 48 KB of LDS per workgroup, so
// several workgroups of 4 waves share a CU and cover each other's load latency; persistent over the items.
template <int S_>
struct C1W {
    static constexpr int S = S_, SS = S * S;
    static constexpr int R = S <= 16 ? S : (S + 1) / 2, NBAND = (S + R - 1) / R;
    static constexpr int XW = S <= 16 ? 16 : 32, RPS = 32 / XW, NSLAB = (R + RPS - 1) / RPS;
    static constexpr int GP = R * S;
    static constexpr int odd128(int x) { return ((x + 127) / 256) * 256 + 128; }
    static constexpr int G_PLANE = odd128((GP + 1) * 16), G_HALF = 4 * G_PLANE, G_BYTES = 2 * G_HALF;    // (+ 1: the zero unit)
    static constexpr int PW = (S + 2 + 3) / 4 * 4, PR = NSLAB * RPS + 2, CELLS = PR * PW;               // (rows of whole 8-byte groups)
    static constexpr int P_PLANE = ((CELLS + XW + 8) * 2 + 63) / 64 * 64, NPL = 13;                      // (wall, own, enemy, inside) x 3 column shifts, zeros
    static constexpr int P_OFF = G_BYTES, ITEM = (G_BYTES + NPL * P_PLANE + 255) / 256 * 256;
    static constexpr int NCOL = 48, RED = 4 * 32 * NCOL * 4, LDS = ITEM > RED ? ITEM : RED;             // ONE item buffer (the next item waits in registers); the final sums of the four waves
    static constexpr int GU = 8 * GP, GU_PT = (GU + 255) / 256, CELL_PT = (CELLS + 255) / 256;
    static_assert(S % R == 0 && S <= XW && LDS <= 160 * 1024, "geometry");
};

#ifndef TRON_C1W_ABLATE      // diagnostic builds only (wrong results): 1 no B build, 2 no transposed reads, 3 no LDS stores of the item, 4 no global loads, 5 no MFMAs
#define TRON_C1W_ABLATE 0
#endif
template <class C>
__global__ __launch_bounds__(256) void k_conv1_wgrad_px(const int8_t *__restrict__ codes, const unsigned char *__restrict__ gimg, int B,
                                                        float *__restrict__ partial)
{
    constexpr int S = C::S, SS = C::SS;
    extern __shared__ __attribute__((aligned(256))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, g = lane >> 4;
    for (int i = tid * 16; i < C::LDS; i += 256 * 16) *reinterpret_cast<uint4 *>(lds + i) = make_uint4(0u, 0u, 0u, 0u);
    const int nitems = B * C::NBAND;

    f32x4 gu[C::GU_PT];                                                  // (ext vectors, not HIP's uint4 struct: the latter stayed in scratch)
    int cc[C::CELL_PT];
    auto load_item = [&](int item) __attribute__((always_inline)) {
        const int b = item / C::NBAND, r0 = (item - b * C::NBAND) * C::R;
#pragma unroll
        for (int k = 0; k < C::GU_PT; ++k) {
            int u = tid + 256 * k;
            u = u < C::GU ? u : C::GU - 1;                               // (past the item's units: any valid unit, never stored)
            const int pl8 = u / C::GP, px = u - pl8 * C::GP;
            if (TRON_C1W_ABLATE == 4) { gu[k] = (f32x4){0.f, 0.f, 0.f, 0.f}; continue; }
            gu[k] = *reinterpret_cast<const f32x4 *>(gimg + (((size_t)b * 8 + pl8) * SS + r0 * S + px) * 16);
        }
#pragma unroll
        for (int k = 0; k < C::CELL_PT; ++k) {
            const int c = tid + 256 * k;
            const int wr = c / C::PW, y = r0 + wr - 1, x = c - wr * C::PW - 1;
            const bool in = c < C::CELLS && y >= 0 && y < S && x >= 0 && x < S;
            const int8_t v = codes[in ? (size_t)b * SS + y * S + x : (size_t)b * SS];
            cc[k] = in ? (int)v : 0;
        }
    };
    auto store_item = [&]() __attribute__((always_inline)) {
        unsigned char *I = lds;
#pragma unroll
        for (int k = 0; k < C::GU_PT; ++k) {
            const int u = tid + 256 * k;
            if (u < C::GU && TRON_C1W_ABLATE != 3) {
                const int pl8 = u / C::GP, px = u - pl8 * C::GP;
                *reinterpret_cast<f32x4 *>(I + (pl8 >> 2) * C::G_HALF + (pl8 & 3) * C::G_PLANE + px * 16) = gu[k];
            }
            if (TRON_C1W_ABLATE == 3) asm volatile("" ::"v"(gu[k]));
        }
#pragma unroll
        for (int k = 0; k < C::CELL_PT; ++k) {
            const int c = tid + 256 * k;
            if (c < C::CELLS) {
                const int nib = cc[k] & 15;                              // k_conv1_px's decode: 15 wall, 14 / 10 own body / head, 13 / 6 enemy body / head, 1 empty
                const f16 one = (f16)1.0f, ten = (f16)10.0f, zero = (f16)0.0f;
                const f16 v[4] = {nib == 15 ? one : zero, nib == 14 ? one : (nib == 10 ? ten : zero), nib == 13 ? one : (nib == 6 ? ten : zero),
                                  (nib == 1 || nib == 15 || nib == 14 || nib == 10 || nib == 13 || nib == 6) ? one : zero};
                const int wc = c % C::PW;
                f16 *P = reinterpret_cast<f16 *>(I + C::P_OFF) + c;
#pragma unroll
                for (int pl = 0; pl < 4; ++pl)
#pragma unroll
                    for (int sh = 0; sh < 3; ++sh)                       // copy sh holds cell (row, column + sh) at (row, column)
                        if (wc >= sh) P[(pl * 3 + sh) * (C::P_PLANE / 2) - sh] = v[pl];
            }
        }
    };

    // A operand (gradient), read j of a slab: this lane supplies the 8 bytes (li & 3) of k position 16 j + 4 g + (li >> 2)
    const int q4 = li >> 2, p4 = li & 3;
    int a_row[2], a_x[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = 16 * j + 4 * g + q4;
        a_row[j] = i / C::XW;
        a_x[j] = i - a_row[j] * C::XW;
    }
    const int a_plane = (p4 >> 1) * C::G_PLANE + (p4 & 1) * 8;            // M tile t: + 2 t G_PLANE; lo half: + G_HALF
    // B operand (planes), N tile nt: column 16 nt + li = (plane, tap); element e sits at k position 16 (e >> 2) + 4 g + (e & 3)
    int b_base[3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
        const int c = 16 * nt + li, pl = c < 36 ? c / 9 : -1, tap = c < 36 ? c - (c / 9) * 9 : 4;
        b_base[nt] = C::P_OFF + (pl < 0 ? C::NPL - 1 : pl * 3 + tap % 3) * C::P_PLANE + 2 * ((tap / 3) * C::PW + 4 * g);
    }

    f32x4 acc0[2][3], acc1[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc0[t][n] = acc1[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds;

    int item = blockIdx.x;
    __syncthreads();                                                     // the zeros are in
    if (item < nitems) {
        load_item(item);
        store_item();
    }
    __syncthreads();
    for (; item < nitems; item += (int)gridDim.x) {
        const int nxt = item + (int)gridDim.x;
        if (nxt < nitems) load_item(nxt);                                // in flight under this item's slabs
        const uint32_t I = lds_base;
        const unsigned char *Ip = lds;
        for (int s = wave; s < C::NSLAB; s += 4) {
            s16x4 rg[8];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int y = s * C::RPS + a_row[j];
                const bool ok = a_x[j] < S && y < C::R;
                const uint32_t a = I + a_plane + (ok ? (y * S + a_x[j]) * 16 : C::GP * 16);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (TRON_C1W_ABLATE == 2) { asm volatile("" : "=v"(rg[4 * t + j]), "=v"(rg[4 * t + 2 + j]) : "v"(a)); continue; }
                    rg[4 * t + j] = lds_tr(a + 2 * t * C::G_PLANE);
                    rg[4 * t + 2 + j] = lds_tr(a + C::G_HALF + 2 * t * C::G_PLANE);
                }
            }
            f16x8 bf[3];
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                const f16 *P = reinterpret_cast<const f16 *>(Ip + b_base[nt] + 2 * s * C::RPS * C::PW);
                typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
                const f16x4v r0 = *reinterpret_cast<const f16x4v *>(P), r1 = *reinterpret_cast<const f16x4v *>(P + (C::XW == 16 ? C::PW : 16));
                bf[nt] = __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, 6, 7);
                if (TRON_C1W_ABLATE == 1) asm volatile("" : "=v"(bf[nt]) : "v"(P));
            }
            lds_wait();
            f16x8 gh[2], gl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                gh[t] = join8(rg[4 * t + 0], rg[4 * t + 1]);
                gl[t] = join8(rg[4 * t + 2], rg[4 * t + 3]);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) {
                    if (TRON_C1W_ABLATE == 5) { asm volatile("" ::"v"(gh[t]), "v"(gl[t]), "v"(bf[nt])); continue; }
                    acc0[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gh[t], bf[nt], acc0[t][nt], 0, 0, 0);
                    acc1[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gl[t], bf[nt], acc1[t][nt], 0, 0, 0);
                }
        }
        __syncthreads();                                                 // everybody is done reading this item
        if (nxt < nitems) store_item();
        __syncthreads();
    }
    // the four waves' sums, joined in a fixed order: partial[workgroup][co][column]; D row = 4 g + r (co within the tile), column = li
    float *red = reinterpret_cast<float *>(lds);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const f32x4 v = acc0[t][nt] + acc1[t][nt] * LO_UNSCALE;
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wave * 32 + 16 * t + 4 * g + r) * C::NCOL + 16 * nt + li] = v[r];
        }
    __syncthreads();
    for (int i = tid; i < 32 * C::NCOL; i += 256)
        partial[(size_t)blockIdx.x * 32 * C::NCOL + i] = (red[i] + red[32 * C::NCOL + i]) + (red[2 * 32 * C::NCOL + i] + red[3 * 32 * C::NCOL + i]);
}

// dW[co][plane][tap] = (64 / s) * sum over the workgroups of partial[workgroup][co][9 plane + tap]  (x plane4 for the fourth plane):
// one workgroup per output channel, a thread = one column x one of five part groups, joined in a fixed order
__global__ __launch_bounds__(768) void k_conv1_wgrad_finish(const float *__restrict__ partial, int nparts, int cin, float plane4,
                                                            const float *__restrict__ ginfo, float *__restrict__ gw)
{
    __shared__ float red[16][48];
    const int co = blockIdx.x, col = threadIdx.x % 48, pg = threadIdx.x / 48;
    float a0 = 0.0f, a1 = 0.0f;
    int p = pg;
    for (; p + 16 < nparts; p += 32) {
        a0 += partial[((size_t)p * 32 + co) * 48 + col];
        a1 += partial[((size_t)(p + 16) * 32 + co) * 48 + col];
    }
    if (p < nparts) a0 += partial[((size_t)p * 32 + co) * 48 + col];
    red[pg][col] = a0 + a1;
    __syncthreads();
    if (threadIdx.x < 9 * cin) {
        const int c = threadIdx.x;
        float v = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; q += 4) v += (red[q][c] + red[q + 1][c]) + (red[q + 2][c] + red[q + 3][c]);
        gw[(size_t)co * cin * 9 + c] = v * (64.0f * ginfo[1]) * (c >= 27 ? plane4 : 1.0f);
    }
}

template <class C>
int launch_conv1_wgrad(const int8_t *codes, const void *gimg, const float *ginfo, int64_t B, int cin, float plane4, float *partial, float *gw,
                       hipStream_t st, int64_t *ws_need)
{
    const int cus = device_cus();
    int per_cu = (160 * 1024) / C::LDS;
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    int64_t grid = (int64_t)cus * per_cu;
    if (grid > B * C::NBAND) grid = B * C::NBAND;
    if (ws_need) { *ws_need = (int64_t)cus * 4 * 32 * C::NCOL * (int64_t)sizeof(float); return TRON_OK; }
    static uint64_t prepared = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_NO_DEVICE; }
    if (!(prepared & (1ull << (dev & 63)))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv1_wgrad_px<C>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess)
            (void)hipGetLastError();
        prepared |= 1ull << (dev & 63);
    }
    hipLaunchKernelGGL(k_conv1_wgrad_px<C>, dim3((unsigned)grid), dim3(256), C::LDS, st, codes, reinterpret_cast<const unsigned char *>(gimg), (int)B, partial);
    hipLaunchKernelGGL(k_conv1_wgrad_finish, dim3(32), dim3(768), 0, st, partial, (int)grid, cin, plane4, ginfo, gw);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

}  // namespace

extern "C" int64_t tron_conv1_wgrad_px16_workspace(int64_t batch, int32_t side)
{
    int64_t need = 0;
    if (batch < 1 || batch >= (1ll << 30)) return 0;
    if (side == 12) launch_conv1_wgrad<C1W<12>>(nullptr, nullptr, nullptr, batch, 3, 0.0f, nullptr, nullptr, nullptr, &need);
    else if (side == 26) launch_conv1_wgrad<C1W<26>>(nullptr, nullptr, nullptr, batch, 3, 0.0f, nullptr, nullptr, nullptr, &need);
    return need;
}

extern "C" int tron_conv1_wgrad_px16(const int8_t *codes, const void *grad_px16, const float *grad_info, int64_t batch, int32_t side,
                                     int32_t cin, float plane4, float *grad_weight, void *workspace, void *stream)
{
    if (!codes || !grad_px16 || !grad_info || !grad_weight || !workspace || batch < 0 || (cin != 3 && cin != 4)) return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(grad_px16) | reinterpret_cast<uintptr_t>(workspace)) & 15u) return TRON_ERR_BAD_ARG;
    if (batch >= (1ll << 30)) return TRON_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (batch == 0)
        return hipMemsetAsync(grad_weight, 0, (size_t)32 * cin * 9 * sizeof(float), st) == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
    float *partial = reinterpret_cast<float *>(workspace);
    if (side == 12) return launch_conv1_wgrad<C1W<12>>(codes, grad_px16, grad_info, batch, cin, plane4, partial, grad_weight, st, nullptr);
    if (side == 26) return launch_conv1_wgrad<C1W<26>>(codes, grad_px16, grad_info, batch, cin, plane4, partial, grad_weight, st, nullptr);
    return TRON_ERR_UNSUPPORTED;
}

extern "C" int tron_conv1_px16_train(const int8_t *codes, const float *weight, const float *bias, int32_t cin, float plane4,
                                     int64_t batch, int32_t side, void *out_px16, void *pre_px16, void *stream)
{
    if (!codes || !weight || !bias || !out_px16 || !pre_px16 || batch < 0 || (cin != 3 && cin != 4)) return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(out_px16) | reinterpret_cast<uintptr_t>(pre_px16) | reinterpret_cast<uintptr_t>(bias)) & 15u) return TRON_ERR_BAD_ARG;
    if (batch == 0) return TRON_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = batch * side * side;
    const unsigned grid = (unsigned)((total + 255) / 256 < 256 * 8 ? (total + 255) / 256 : 256 * 8);
    unsigned char *o = reinterpret_cast<unsigned char *>(out_px16), *z = reinterpret_cast<unsigned char *>(pre_px16);
    if (side == 12) hipLaunchKernelGGL((k_conv1_px<12, true>), dim3(grid), dim3(256), 0, st, codes, weight, bias, cin, plane4, batch, o, z);
    else if (side == 26) hipLaunchKernelGGL((k_conv1_px<26, true>), dim3(grid), dim3(256), 0, st, codes, weight, bias, cin, plane4, batch, o, z);
    else return TRON_ERR_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

#define TRON_WST_CASES(MODE_, ARGS_)                                                                                    \
    TRON_WST_CASE(12, 12, 32, 32, 2, 12, MODE_, ARGS_)                                                                  \
    TRON_WST_CASE(12, 12, 32, 64, 1, 12, MODE_, ARGS_)                                                                  \
    TRON_WST_CASE(12, 12, 64, 64, 1, 8, MODE_, ARGS_)                                                                   \
    TRON_WST_CASE(26, 13, 32, 32, 1, 12, MODE_, ARGS_)                                                                  \
    TRON_WST_CASE(26, 13, 32, 64, 1, 12, MODE_, ARGS_)                                                                  \
    TRON_WST_CASE(26, 7, 64, 64, 1, 8, MODE_, ARGS_)                                                                    \
    TRON_WST_CASE(34, 12, 32, 32, 1, 8, MODE_, ARGS_)     /* 32x32 boards (the ACKTR nets): bands of 12 + 12 + 10 rows, eight waves (LDS) */ \
    TRON_WST_CASE(34, 12, 32, 64, 1, 8, MODE_, ARGS_)                                                                   \
    TRON_WST_CASE(34, 5, 64, 64, 1, 8, MODE_, ARGS_)      /* seven bands of 5 (4) rows: 2 x 66 KB of LDS */

int ws_fwd_side34(const void *in_px16, const void *wfrag, const float *bias, const void *res_px16, void *out_px16, float *out_f32,
                  float *pre_f32, int64_t batch, int32_t cin, int32_t cout, int32_t apply_mish, hipStream_t st)
{
    const int side = 34;
#define TRON_WST_CASE(S_, R_, CI_, CO_, IPI_, WAVES_, MODE_, ARGS_)                                                     \
    if (S_ == 34 && side == S_ && cin == CI_ && cout == CO_) return launch_ws<Geo<S_, R_, CI_, CO_, IPI_, WAVES_, 1>, MODE_> ARGS_;
    TRON_WST_CASES(WS_INFER, (in_px16, wfrag, bias, res_px16, out_px16, out_f32, pre_f32, batch, apply_mish, st))
#undef TRON_WST_CASE
    return TRON_ERR_UNSUPPORTED;
}

extern "C" int tron_conv3x3_ws_train_fwd(const void *in_px16, const void *wfrag, const float *bias, const void *res_px16,
                                         void *out_px16, float *out_f32, void *pre_px16, int64_t batch, int32_t cin,
                                         int32_t cout, int32_t side, void *stream)
{
    if (!in_px16 || !wfrag || !bias || !pre_px16 || batch < 0 || (!out_px16 && !out_f32)) return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(in_px16) | reinterpret_cast<uintptr_t>(wfrag) | reinterpret_cast<uintptr_t>(res_px16) |
         reinterpret_cast<uintptr_t>(out_px16) | reinterpret_cast<uintptr_t>(pre_px16) | reinterpret_cast<uintptr_t>(bias)) & 15u)
        return TRON_ERR_BAD_ARG;
    if (batch == 0) return TRON_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define TRON_WST_CASE(S_, R_, CI_, CO_, IPI_, WAVES_, MODE_, ARGS_)                                                     \
    if (side == S_ && cin == CI_ && cout == CO_) return launch_ws<Geo<S_, R_, CI_, CO_, IPI_, WAVES_, 1>, MODE_> ARGS_;
    TRON_WST_CASES(WS_TRAIN, (in_px16, wfrag, bias, res_px16, out_px16, out_f32, nullptr, batch, 1, st, pre_px16))
#undef TRON_WST_CASE
    return TRON_ERR_UNSUPPORTED;
}

extern "C" int tron_conv3x3_ws_split_weights_bwd(const float *const *weights, const int32_t *cins, const int32_t *couts,
                                                 void *const *workspaces, float *wnorms, int32_t n, void *stream)
{
    if (!weights || !cins || !couts || !workspaces || !wnorms || n < 1 || n > WS_SPLIT_MAX) return TRON_ERR_BAD_ARG;
    WsJobs jobs{};
    int most = 0;
    for (int k = 0; k < n; ++k) {
        if (!weights[k] || !workspaces[k] || (reinterpret_cast<uintptr_t>(workspaces[k]) & 15u)) return TRON_ERR_BAD_ARG;
        if (tron_conv3x3_ws_workspace(couts[k], cins[k]) == 0 || cins[k] > 1024 || couts[k] > 1024) return TRON_ERR_UNSUPPORTED;
        jobs.w[k] = weights[k];
        jobs.ws[k] = reinterpret_cast<f16 *>(workspaces[k]);
        jobs.cin[k] = couts[k];                                          // the backward convolution: cout channels in, cin channels out
        jobs.cout[k] = cins[k];
        jobs.rot[k] = 1;
        const int total = (cins[k] / 16) * 9 * (couts[k] / 32) * 512;
        most = total > most ? total : most;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_ws_split_weights, dim3((most + 255) / 256, n), dim3(256), 0, st, jobs);
    hipLaunchKernelGGL(k_ws_wnorm, dim3(n), dim3(64), 0, st, jobs, wnorms);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

extern "C" int64_t tron_px16_grad_workspace(int64_t batch, int32_t channels)
{
    if (batch < 1 || channels < 8 || channels > 64 || channels % 8) return 0;
    return (int64_t)2 * 1024 * channels * (int64_t)sizeof(float) + 256;  // stats[2][<= 1024 groups][C]
}

extern "C" int tron_px16_grad_from_f32(const float *grad_out, const void *pre_px16, const float *scale4, int64_t batch,
                                       int32_t channels, int32_t side, void *grad_px16, float *grad_info, float *bias_grad,
                                       void *workspace, void *stream)
{
    if (!grad_out || !pre_px16 || !scale4 || !grad_px16 || !grad_info || !workspace || batch < 0) return TRON_ERR_BAD_ARG;
    if (channels < 8 || channels > 64 || channels % 8 || side < 1) return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(pre_px16) | reinterpret_cast<uintptr_t>(grad_px16) | reinterpret_cast<uintptr_t>(workspace)) & 15u)
        return TRON_ERR_BAD_ARG;
    if (batch == 0) return TRON_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int groups = (int)(batch < 1024 ? batch : 1024);
    float *stats = reinterpret_cast<float *>(workspace);
    hipLaunchKernelGGL(k_gout_px<0>, dim3((unsigned)groups, (unsigned)(channels / 8)), dim3(256), 0, st, grad_out,
                       reinterpret_cast<const unsigned char *>(pre_px16), batch, channels, side, scale4,
                       reinterpret_cast<unsigned char *>(grad_px16), stats, grad_info);
    hipLaunchKernelGGL(k_wsb_finish, dim3((unsigned)channels), dim3(256), 0, st, stats, groups, channels, bias_grad, grad_info + 4);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

extern "C" int tron_px16_grad_from_pooled(const float *grad_pooled, int32_t channels_last, const void *pre_px16, const float *scale4,
                                          int64_t batch, int32_t channels, int32_t side, void *grad_px16, float *grad_info,
                                          float *bias_grad, void *workspace, void *stream)
{
    if (!grad_pooled || !pre_px16 || !scale4 || !grad_px16 || !grad_info || !workspace || batch < 0) return TRON_ERR_BAD_ARG;
    if (channels < 8 || channels > 64 || channels % 8 || side < 2 || side % 2) return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(pre_px16) | reinterpret_cast<uintptr_t>(grad_px16) | reinterpret_cast<uintptr_t>(workspace) |
         reinterpret_cast<uintptr_t>(grad_pooled)) & 15u)
        return TRON_ERR_BAD_ARG;
    if (gout_images(side) * (side / 2) * (side / 2) * 8 > GOUT_TILE) return TRON_ERR_UNSUPPORTED;      // (the kernel's LDS tile: pooled planes up to 13 x 13)
    if (batch == 0) return TRON_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t batches = (batch + gout_images(side) - 1) / gout_images(side);
    int groups = (int)(batches < 1024 ? batches : 1024);
    float *stats = reinterpret_cast<float *>(workspace);
    const dim3 grid((unsigned)groups, (unsigned)(channels / 8));
    if (channels_last)
        hipLaunchKernelGGL(k_gout_px<2>, grid, dim3(256), 0, st, grad_pooled, reinterpret_cast<const unsigned char *>(pre_px16), batch, channels,
                           side, scale4, reinterpret_cast<unsigned char *>(grad_px16), stats, grad_info);
    else
        hipLaunchKernelGGL(k_gout_px<1>, grid, dim3(256), 0, st, grad_pooled, reinterpret_cast<const unsigned char *>(pre_px16), batch, channels,
                           side, scale4, reinterpret_cast<unsigned char *>(grad_px16), stats, grad_info);
    hipLaunchKernelGGL(k_wsb_finish, dim3((unsigned)channels), dim3(256), 0, st, stats, groups, channels, bias_grad, grad_info + 4);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

extern "C" int64_t tron_conv3x3_ws_dgrad_workspace(int32_t cin, int32_t cout)
{
    if (tron_conv3x3_ws_workspace(cout, cin) == 0 || cin > 64) return 0;
    return (int64_t)2 * 1024 * cin * (int64_t)sizeof(float) + 256;       // stats[2][workgroups <= 1024][cin] | scal[4]
}

// (conv^T(grad, W) + extra) * mish'(pre_below): cin / cout are the FORWARD layer's; the output has cin channels
extern "C" int tron_conv3x3_ws_dgrad(const void *grad_px16, const float *grad_info, const void *wfrag_rot, const float *wnorm,
                                     const void *extra_px16, const float *extra_info, const void *pre_below_px16, void *out_px16,
                                     float *out_f32, float *out_info, float *bias_grad_below, int64_t batch, int32_t cin,
                                     int32_t cout, int32_t side, void *workspace, void *stream)
{
    if (!grad_px16 || !grad_info || !wfrag_rot || !wnorm || !pre_below_px16 || !out_info || !workspace || batch < 0) return TRON_ERR_BAD_ARG;
    if ((!out_px16 && !out_f32) || (extra_px16 && !extra_info)) return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(grad_px16) | reinterpret_cast<uintptr_t>(wfrag_rot) | reinterpret_cast<uintptr_t>(extra_px16) |
         reinterpret_cast<uintptr_t>(pre_below_px16) | reinterpret_cast<uintptr_t>(out_px16) | reinterpret_cast<uintptr_t>(workspace)) & 15u)
        return TRON_ERR_BAD_ARG;
    if (batch == 0) return TRON_OK;
    if (tron_conv3x3_ws_dgrad_workspace(cin, cout) == 0) return TRON_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float *stats = reinterpret_cast<float *>(workspace);
    WsBwd bw{reinterpret_cast<const unsigned char *>(pre_below_px16), grad_info, extra_px16 ? extra_info : nullptr, wnorm, out_info, stats};
    int grid = 0, rc = TRON_ERR_UNSUPPORTED;
    // the backward convolution has the forward layer's cout channels in and cin channels out
#define TRON_WSB_CASE(S_, R_, CI_, CO_, IPI_, WAVES_)                                                                   \
    if (side == S_ && cout == CI_ && cin == CO_)                                                                        \
        rc = launch_ws<Geo<S_, R_, CI_, CO_, IPI_, WAVES_, 1>, WS_BWD>(grad_px16, wfrag_rot, nullptr, extra_px16, out_px16, out_f32, nullptr, \
                                                                       batch, 1, st, nullptr, bw, &grid);
    TRON_WSB_CASE(12, 12, 32, 32, 2, 8)          // (eight waves: the gradient epilogue's state does not fit three waves per SIMD)
    TRON_WSB_CASE(12, 12, 64, 32, 1, 8)
    TRON_WSB_CASE(12, 12, 64, 64, 1, 8)
    TRON_WSB_CASE(26, 13, 32, 32, 1, 8)
    TRON_WSB_CASE(26, 7, 64, 32, 1, 8)
    TRON_WSB_CASE(26, 7, 64, 64, 1, 8)
    TRON_WSB_CASE(34, 12, 32, 32, 1, 8)
    TRON_WSB_CASE(34, 5, 64, 32, 1, 8)
    TRON_WSB_CASE(34, 5, 64, 64, 1, 8)
#undef TRON_WSB_CASE
    if (rc != TRON_OK) return rc;
    hipLaunchKernelGGL(k_wsb_finish, dim3((unsigned)cin), dim3(256), 0, st, stats, grid, cin, bias_grad_below, out_info + 4);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

extern "C" int64_t tron_conv3x3_wgrad_px16_workspace(int64_t batch, int32_t cin, int32_t cout, int32_t side)
{
    if (batch < 1) return 0;
    int64_t need = 0;
    const int rc = dispatch_wgrad_px(side, cin, cout, [&](auto cfg) {
        return launch_wgrad_px<decltype(cfg)>(nullptr, nullptr, nullptr, batch, cout, cin, nullptr, nullptr, nullptr, &need);
    });
    return rc == TRON_OK ? need + 256 : 0;
}

extern "C" int tron_conv3x3_wgrad_px16(const void *in_px16, const void *grad_px16, const float *grad_info, float *grad_weight,
                                       int64_t batch, int32_t cin, int32_t cout, int32_t side, void *workspace, void *stream)
{
    if (!in_px16 || !grad_px16 || !grad_info || !grad_weight || !workspace || batch < 0) return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(in_px16) | reinterpret_cast<uintptr_t>(grad_px16) | reinterpret_cast<uintptr_t>(workspace)) & 15u)
        return TRON_ERR_BAD_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (batch == 0)
        return hipMemsetAsync(grad_weight, 0, (size_t)cout * cin * 9 * sizeof(float), st) == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
    return dispatch_wgrad_px(side, cin, cout, [&](auto cfg) {
        return launch_wgrad_px<decltype(cfg)>(grad_px16, in_px16, grad_info, batch, cout, cin, reinterpret_cast<float *>(workspace),
                                              grad_weight, st, nullptr);
    });
}
