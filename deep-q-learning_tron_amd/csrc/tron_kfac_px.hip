// tron_kfac_px.hip — K-FAC's input factor of a 3x3 / pad 1 / stride 1 convolution (kfac.py:41-58: A = E[a a^T], a = the layer's
// 9 C patch vector) WITHOUT the patch matrix.  A is a 9 x 9 grid of C x C blocks,
//     block(t, t')[c][c'] = sum over images and output positions p of x[c][p + t] x[c'][p + t']        (x zero outside the image),
// and only the 45 blocks with t' >= t are needed (A is symmetric).  csrc/tron_kfac.hip materialises P^T (9x the image bytes, as
// split f16) and runs a Gram kernel on it; here both MFMA operands of a block are the SAME zero-haloed window of the image in
// LDS read at two tap offsets — the scheme of csrc/tron_conv_ws_train.hip's k_wgrad_px (K = pixels, operands read transposed
// with ds_read_b64_tr_b16 from a pixel-major PX16 image, items = row bands double-buffered by table-driven LDS-DMA).
//   * the image goes f32 NCHW -> PX16 once (k_nchw_to_px16: 4 bytes read, 4 written per element);
//   * the 45 (t, t') pairs are dealt to FIVE workgroup kinds of nine pairs each: kind k takes row k (t' = k .. 8) and row 9 - k
//     (t' = 9 - k .. 8) — (9 - k) + k = 9 pairs and at most two M shifts per kind; a wave owns a (32 c) x (16 c') block of all
//     nine pairs = 144 accumulator registers, as k_wgrad_px's nine taps;
//   * a kind reads the window once per item: five kinds = 5 x (1 + halo) image reads instead of nine patch copies written and
//     re-read; the kinds of an item sit 8 blocks apart (one XCD: the window comes out of its L2 for four of them);
//   * padding pixels of a band's last slab are zeroed in the M fragments (a select after the wait: the window holds real data there);
//   * per-(workgroup, K group) partial sums, joined in a fixed order by k_kfac_px_finish, which also places block (t, t') at
//     A[c 9 + t][c' 9 + t'] and its mirror (the reference's patch order: channel major, kfac.py:28-38) — bitwise symmetric.
#include "tron_conv_ws_kernel.hpp"
#include "tron_kfac_px.hpp"

namespace {

// ---- f32 NCHW -> PX16 ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_nchw_to_px16(const float *__restrict__ x, int64_t B, int C, int SS, unsigned char *__restrict__ out)
{
    const int64_t total = B * (C / 8) * SS;                              // (image, octet, pixel): coalesced along the pixels of each channel row
    const size_t half = (size_t)(C / 8) * SS * 16;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t img = i / ((C / 8) * SS);
        const int r = (int)(i - img * (C / 8) * SS), oct = r / SS, p = r - oct * SS;
        f16x8 hh, ll;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = x[((size_t)img * C + oct * 8 + j) * SS + p] * ACT_SCALE;
            hh[j] = (f16)v;
            ll[j] = (f16)((v - (float)hh[j]) * LO_SCALE);
        }
        unsigned char *op = out + (size_t)img * 2 * half + ((size_t)oct * SS + p) * 16;
        *reinterpret_cast<f16x8 *>(op) = hh;
        *reinterpret_cast<f16x8 *>(op + half) = ll;
    }
}

template <int S_, int NI_, int R_, int C_, int KG_>
struct KCfg {
    static constexpr int S = S_, NI = NI_, R = R_, C = C_, KG = KG_;
    static constexpr int SS = S * S, ROWS = NI * S, NB = (ROWS + R - 1) / R;
    static constexpr int GPX = R * S, NSLAB = (GPX + 31) / 32;
    static constexpr int WR = R + 2 + (NI - 1), WC = S + 2;
    static constexpr int odd128(int x) { return ((x + 127) / 256) * 256 + 128; }
    static constexpr int PLANE = odd128(WR * WC * 16), HALF = (C / 8) * PLANE;
    static constexpr int ITEM = (2 * HALF + 1023) / 1024 * 1024;
    static constexpr int NPIECE = ITEM / 1024, PPW = (NPIECE + 7) / 8;
    static constexpr int TAB_OFF = 2 * ITEM, TAB_BYTES = 8 * PPW * 128;
    static constexpr int LDS = TAB_OFF + TAB_BYTES + 1024;
    static constexpr int NBLK = (C / 32) * (C / 16);
    static_assert(NBLK * KG == 8, "eight waves");
    static_assert(LDS <= 160 * 1024, "LDS");
    static_assert(2 * NI * 8 * SS < 0xFFFF, "source offsets in 16 bits");
    static_assert(GPX % 4 == 0, "padding pixels come in whole 4-pixel read groups");
};
template <class C> __device__ __forceinline__ int kstack_e(int t) { return t + t / C::S + 1; }

struct KBands { int n[8]; };

// pair i of kind k: (t, t')
__host__ __device__ constexpr int pair_t(int kind, int i) { return i < 9 - kind ? kind : 9 - kind; }
__host__ __device__ constexpr int pair_u(int kind, int i) { return i < 9 - kind ? kind + i : 9 - kind + (i - (9 - kind)); }
__device__ __forceinline__ int tap_shift(int t, int WC) { return (t / 3 - 1) * WC * 16 + (t % 3 - 1) * 16; }

template <class C, int KIND>
__device__ __forceinline__ void kfac_px_body(const unsigned char *__restrict__ ximg, int B, int wg, const KBands &bands,
                                             float *__restrict__ partial, unsigned char *lds)
{
    constexpr int S = C::S, SS = C::SS, CH = C::C;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, g = lane >> 4;
    int band = 0, wb = wg, nwb = bands.n[0];
    while (wb >= nwb && band + 1 < C::NB) { wb -= nwb; ++band; nwb = bands.n[band]; }
    if (wb >= nwb) return;
    const int r0 = band * C::R, nrows = (C::ROWS - r0 < C::R) ? C::ROWS - r0 : C::R, npx = nrows * S;
    const int nslab = (npx + 31) >> 5;
    const int e0 = kstack_e<C>(r0) - 1, e_last = kstack_e<C>(r0 + nrows - 1) + 1;
    const int blk = wave % C::NBLK, kgroup = wave / C::NBLK;
    const int cot2 = blk / (CH / 16), cit = blk % (CH / 16);

    for (int i = tid * 16; i < C::TAB_OFF; i += 512 * 16) *reinterpret_cast<uint4 *>(lds + i) = make_uint4(0u, 0u, 0u, 0u);
    uint16_t *tab = reinterpret_cast<uint16_t *>(lds + C::TAB_OFF) + wave * C::PPW * 64;
    const size_t img_units = (size_t)2 * (CH / 8) * SS;
    for (int j = 0; j < C::PPW; ++j) {
        const int d = (wave + 8 * j) * 1024 + lane * 16;
        int off = -1;
        if (d < 2 * C::HALF) {
            const int h = d / C::HALF, d1 = d - h * C::HALF, o = d1 / C::PLANE, u = (d1 - o * C::PLANE) >> 4;
            const int wr = u / C::WC, wc = u - wr * C::WC, e = e0 + wr;
            const int img = e / (S + 1), y = e - img * (S + 1) - 1;
            if (e <= e_last && wc >= 1 && wc <= S && y >= 0 && img < C::NI)
                off = (int)(img * img_units) + (h * (CH / 8) + o) * SS + y * S + (wc - 1);
        }
        tab[j * 64 + lane] = (uint16_t)(off < 0 ? 0xFFFF : off);
    }
    const int nstack = (B + C::NI - 1) / C::NI;
    auto dma_piece = [&](int stack, int buf, int j) {
        const int q = wave + 8 * j;
        const uint32_t o16 = tab[j * 64 + lane];
        const bool ok = stack < nstack && o16 != 0xFFFFu && (C::NI == 1 || (int64_t)stack * C::NI + (o16 >= img_units ? 1 : 0) < B);
        const unsigned char *src = ok ? ximg + ((size_t)stack * C::NI * img_units + o16) * 16
                                      : reinterpret_cast<const unsigned char *>(g_ws_zero) + lane * 16;
        unsigned char *dst = q < C::NPIECE ? lds + buf * C::ITEM + q * 1024 : lds + C::TAB_OFF + C::TAB_BYTES;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    };

    // the pixel rows this lane supplies to the transposed reads (k_wgrad_px's assignment): read j of slab s covers pixels
    // 32 s + 16 j + 4 g + (li >> 2); `ok[s][j]`: that 4-pixel group exists (else the M fragment's half is zeroed)
    const int q4 = li >> 2, p4 = li & 3;
    int addr[C::NSLAB][2];
    bool ok[C::NSLAB][2];
#pragma unroll
    for (int s = 0; s < C::NSLAB; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int px = 32 * s + 16 * j + 4 * g + q4;
            ok[s][j] = 32 * s + 16 * j + 4 * g < npx;
            px = px < npx ? px : 0;
            const int t = r0 + px / S, x = px - (px / S) * S;
            addr[s][j] = (p4 >> 1) * C::PLANE + ((kstack_e<C>(t) - e0) * C::WC + x + 1) * 16 + (p4 & 1) * 8;
        }
    const int m_off = 4 * cot2 * C::PLANE, n_off = 2 * cit * C::PLANE;   // this wave's channel tiles: M 32 channels (two tiles), N 16

    f32x4 acc0[9][2], acc1[9][2];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc0[k][t] = acc1[k][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // partial[kind][part][pair][c][c'], part = (workgroup, kgroup); D row = 4 g + r -> c within the tile, column li -> c'.
    // The matrix unit's f32 accumulate is not round-to-nearest over a long chain: an all-positive sum (the factor's diagonal) drifts
    // low by ~3.6e-10 of the sum per MFMA (measured: -2.4e-6 after 6 700 steps at 8 192 x 34 x 34), so a wave adds its accumulators
    // into its own slots of `partial` every FLUSH_STEPS slab steps (VALU adds, IEEE) and starts them again from zero.
    float *out = partial + ((size_t)wg * C::KG + kgroup) * 9 * CH * CH;
    auto spill = [&](bool add) {
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x4 v = acc0[i][t] + acc1[i][t] * LO_UNSCALE;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float *o = out + ((size_t)i * CH + 32 * cot2 + 16 * t + 4 * g + r) * CH + 16 * cit + li;
                    *o = add ? *o + v[r] : v[r];
                }
                acc0[i][t] = acc1[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
    };
    constexpr int FLUSH_STEPS = 256;
    const int steps_per_stack = kgroup < nslab ? (nslab - kgroup + C::KG - 1) / C::KG : 0;
    int since = 0;
    bool wrote = false;

    __syncthreads();
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds;
    int stack = wb;
    if (stack < nstack)
        for (int j = 0; j < C::PPW; ++j) dma_piece(stack, 0, j);
    constexpr int TA = KIND, TB = 9 - KIND, NA = 9 - KIND;              // row TA: pairs 0 .. NA - 1; row TB: the rest (none for kind 0)
    for (int cur = 0; stack < nstack; stack += nwb, cur ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        const int nxt = stack + nwb;
        const uint32_t I = lds_base + cur * C::ITEM;
        int pj = 0;
#pragma unroll
        for (int si = 0; si < (C::NSLAB + C::KG - 1) / C::KG; ++si) {
            const int s = kgroup + si * C::KG;
            if (s < nslab) {
                int a0 = addr[0][0], a1 = addr[0][1];
                bool v0 = ok[0][0], v1 = ok[0][1];
#pragma unroll
                for (int q = 1; q < C::NSLAB; ++q)
                    if (s == q) { a0 = addr[q][0]; a1 = addr[q][1]; v0 = ok[q][0]; v1 = ok[q][1]; }
                // M fragments: this wave's 32 channels at the row's tap shift, hi and lo (raw halves: [row a | row b][tile][h0 h1 l0 l1])
                s16x4 rm[2][2][4], rn[2][4];
                auto read_m = [&](int which, int t) {
                    const int sh = tap_shift(t, C::WC);
#pragma unroll
                    for (int tile = 0; tile < 2; ++tile) {
                        const uint32_t b = I + m_off + 2 * tile * C::PLANE + sh;
                        rm[which][tile][0] = lds_tr(b + a0);
                        rm[which][tile][1] = lds_tr(b + a1);
                        rm[which][tile][2] = lds_tr(b + C::HALF + a0);
                        rm[which][tile][3] = lds_tr(b + C::HALF + a1);
                    }
                };
                auto read_n = [&](int slot, int t) {
                    const uint32_t b = I + n_off + tap_shift(t, C::WC);
                    rn[slot][0] = lds_tr(b + a0);
                    rn[slot][1] = lds_tr(b + a1);
                    rn[slot][2] = lds_tr(b + C::HALF + a0);
                    rn[slot][3] = lds_tr(b + C::HALF + a1);
                };
                read_m(0, TA);
                if (KIND > 0) read_m(1, TB);
                read_n(0, pair_u(KIND, 0));
                f16x8 mh[2][2], ml[2][2];
                const s16x4 zero4 = {0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    lds_wait();
                    if (i == 0) {
#pragma unroll
                        for (int w = 0; w < (KIND > 0 ? 2 : 1); ++w)
#pragma unroll
                            for (int tile = 0; tile < 2; ++tile) {
                                mh[w][tile] = join8(v0 ? rm[w][tile][0] : zero4, v1 ? rm[w][tile][1] : zero4);
                                ml[w][tile] = join8(v0 ? rm[w][tile][2] : zero4, v1 ? rm[w][tile][3] : zero4);
                            }
                    }
                    const f16x8 nh = join8(rn[i & 1][0], rn[i & 1][1]), nl = join8(rn[i & 1][2], rn[i & 1][3]);
                    if (i < 8) read_n((i + 1) & 1, pair_u(KIND, i + 1));
                    const int w = i < NA ? 0 : 1;
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc1[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(mh[w][t], nl, acc1[i][t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc0[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(mh[w][t], nh, acc0[i][t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc1[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ml[w][t], nh, acc1[i][t], 0, 0, 0);
                    if (si * 9 + i < C::PPW) { dma_piece(nxt, cur ^ 1, si * 9 + i); pj = si * 9 + i + 1; }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        for (; pj < C::PPW; ++pj) dma_piece(nxt, cur ^ 1, pj);
        since += steps_per_stack;
        if (since >= FLUSH_STEPS && nxt < nstack) {                      // (wave-uniform; the last stack's sums go out below)
            spill(wrote);
            wrote = true;
            since = 0;
        }
    }

    spill(wrote);
}

template <class C>
__global__ __launch_bounds__(512, 2) void k_kfac_px(const unsigned char *__restrict__ ximg, int B, KBands bands, int parts_per_kind,
                                                    float *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(256))) unsigned char lds[];
    // the five kinds of one item sit 8 blocks apart: the same XCD under round-robin placement (speed only)
    const int kind = ((int)blockIdx.x >> 3) % 5, wg = (((int)blockIdx.x >> 3) / 5) * 8 + ((int)blockIdx.x & 7);
    float *p = partial + (size_t)kind * parts_per_kind * 9 * C::C * C::C;
    switch (kind) {                                                      // (workgroup-uniform)
    case 0: kfac_px_body<C, 0>(ximg, B, wg, bands, p, lds); break;
    case 1: kfac_px_body<C, 1>(ximg, B, wg, bands, p, lds); break;
    case 2: kfac_px_body<C, 2>(ximg, B, wg, bands, p, lds); break;
    case 3: kfac_px_body<C, 3>(ximg, B, wg, bands, p, lds); break;
    default: kfac_px_body<C, 4>(ximg, B, wg, bands, p, lds); break;
    }
}

// A[c 9 + t][c' 9 + t'] = A[c' 9 + t'][c 9 + t] = scale * 4096 * sum over the parts of partial[kind][part][pair][c][c'] for the 45
// pairs t' >= t (for t' = t only c' >= c is taken and mirrored: the factor is symmetric bit for bit).  One thread per entry of the
// 45 C x C blocks, four chains over the parts joined in a fixed order.
__global__ __launch_bounds__(256) void k_kfac_px_finish(const float *__restrict__ partial, int parts_per_kind, int C, float scale,
                                                        float *__restrict__ gram)
{
    const int CC = C * C, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 45 * CC) return;
    const int pr = i / CC, r = i - pr * CC, c = r / C, cp = r - c * C;
    const int kind = pr / 9, idx = pr - kind * 9, t = pair_t(kind, idx), u = pair_u(kind, idx);
    if (t == u && cp < c) return;
    const float *src = partial + ((size_t)kind * parts_per_kind * 9 + idx) * CC + r;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    int p = 0;
    for (; p + 3 < parts_per_kind; p += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] += src[(size_t)(p + k) * 9 * CC];
    }
    for (int k = 0; p < parts_per_kind; ++p, ++k) a[k] += src[(size_t)p * 9 * CC];
    const float v = ((a[0] + a[1]) + (a[2] + a[3])) * (scale * 4096.0f);
    const int d = 9 * C, row = c * 9 + t, col = cp * 9 + u;
    gram[(size_t)row * d + col] = v;
    gram[(size_t)col * d + row] = v;
}

template <class C>
int64_t kfac_px_run(const float *x, const unsigned char *x_px16, int64_t B, float scale, float *gram, void *workspace, hipStream_t st, bool size_only)
{
    static_assert(C::NB <= 8, "bands");
    const int cus = device_cus();
    int per_kind = cus / 5;
    if (per_kind < C::NB) per_kind = C::NB;
    int slabs[8], tot = 0;
    KBands bands{};
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
        bands.n[b] = n;
        used += n;
    }
    const int wgs = (used + 7) / 8 * 8, parts = wgs * C::KG;             // (inactive workgroups of the last group of 8 write nothing: their parts are zeroed)
    const int64_t px_bytes = (B * C::C * C::SS * 4 + 255) / 256 * 256;
    const int64_t part_bytes = (int64_t)5 * parts * 9 * C::C * C::C * (int64_t)sizeof(float);
    if (size_only) return px_bytes + part_bytes + 256;
    unsigned char *px = reinterpret_cast<unsigned char *>(workspace);      // (x_px16 given: the image is read where it is; the slot stays unused)
    float *partial = reinterpret_cast<float *>(px + px_bytes);
    static uint64_t prepared = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_NO_DEVICE; }
    if (!(prepared & (1ull << (dev & 63)))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_kfac_px<C>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess)
            (void)hipGetLastError();
        prepared |= 1ull << (dev & 63);
    }
    if (wgs != used && hipMemsetAsync(partial, 0, (size_t)part_bytes, st) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_LAUNCH; }
    const int64_t total = B * (C::C / 8) * C::SS;
    if (!x_px16)
        hipLaunchKernelGGL(k_nchw_to_px16, dim3((unsigned)((total + 255) / 256 < (1 << 20) ? (total + 255) / 256 : (1 << 20))), dim3(256), 0, st, x, B,
                           C::C, C::SS, px);
    hipLaunchKernelGGL(k_kfac_px<C>, dim3((unsigned)(wgs * 5)), dim3(512), C::LDS, st, x_px16 ? x_px16 : px, (int)B, bands, parts, partial);
    hipLaunchKernelGGL(k_kfac_px_finish, dim3((unsigned)((45 * C::C * C::C + 255) / 256)), dim3(256), 0, st, partial, parts, C::C, scale, gram);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

template <class F>
int64_t kfac_px_dispatch(int C, int S, F &&f)
{
    if (S == 12) {
        if (C == 64) return f(KCfg<12, 2, 8, 64, 1>{});
        if (C == 32) return f(KCfg<12, 2, 8, 32, 4>{});
    } else if (S == 26) {
        if (C == 64) return f(KCfg<26, 1, 6, 64, 1>{});
        if (C == 32) return f(KCfg<26, 1, 6, 32, 4>{});
    } else if (S == 34) {
        if (C == 64) return f(KCfg<34, 1, 6, 64, 1>{});
        if (C == 32) return f(KCfg<34, 1, 6, 32, 4>{});
    }
    return INT64_MIN;                                                    // no instantiation for this shape
}

}  // namespace

bool tron_kfac_px_supported(int64_t batch, int C, int H, int W, int kh, int kw, int pad, int stride)
{
    return batch >= 1 && batch < (1ll << 30) && kh == 3 && kw == 3 && pad == 1 && stride == 1 && H == W && (C == 32 || C == 64) &&
           (H == 12 || H == 26 || H == 34);
}

int64_t tron_kfac_px_workspace(int64_t batch, int C, int S)
{
    const int64_t n = kfac_px_dispatch(C, S, [&](auto cfg) { return kfac_px_run<decltype(cfg)>(nullptr, nullptr, batch, 0.0f, nullptr, nullptr, nullptr, true); });
    return n == INT64_MIN || n < 0 ? 0 : n;
}

int tron_kfac_px_gram(const float *x, int64_t batch, int C, int S, float scale, float *gram, void *workspace, hipStream_t st)
{
    const int64_t rc = kfac_px_dispatch(C, S, [&](auto cfg) { return kfac_px_run<decltype(cfg)>(x, nullptr, batch, scale, gram, workspace, st, false); });
    return rc == INT64_MIN ? TRON_ERR_UNSUPPORTED : (int)rc;
}

// ---- public entries on PX16 images (include/tron_hip.h) ---------------------------------------------------------------------
extern "C" int tron_px16_from_f32(const float *x, void *out_px16, int64_t batch, int32_t channels, int32_t side, void *stream)
{
    if (!x || !out_px16 || batch < 0 || channels < 8 || channels % 8 || side < 1) return TRON_ERR_BAD_ARG;
    if (reinterpret_cast<uintptr_t>(out_px16) & 15u) return TRON_ERR_BAD_ARG;
    if (batch == 0) return TRON_OK;
    const int64_t total = batch * (channels / 8) * side * side;
    hipLaunchKernelGGL(k_nchw_to_px16, dim3((unsigned)((total + 255) / 256 < (1 << 20) ? (total + 255) / 256 : (1 << 20))), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x, batch, channels, side * side, reinterpret_cast<unsigned char *>(out_px16));
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

extern "C" int64_t tron_kfac_gram_px16_workspace(int64_t batch, int32_t channels, int32_t side)
{
    return tron_kfac_px_supported(batch, channels, side, side, 3, 3, 1, 1) ? tron_kfac_px_workspace(batch, channels, side) : 0;
}

extern "C" int tron_kfac_gram_px16(const void *x_px16, int64_t batch, int32_t channels, int32_t side, float scale, float *gram,
                                   void *workspace, void *stream)
{
    if (!x_px16 || !gram || !workspace || batch < 1) return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(x_px16) | reinterpret_cast<uintptr_t>(workspace)) & 15u) return TRON_ERR_BAD_ARG;
    if (!tron_kfac_px_supported(batch, channels, side, side, 3, 3, 1, 1)) return TRON_ERR_UNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t rc = kfac_px_dispatch(channels, side, [&](auto cfg) {
        return kfac_px_run<decltype(cfg)>(nullptr, reinterpret_cast<const unsigned char *>(x_px16), batch, scale, gram, workspace, st, false);
    });
    return rc == INT64_MIN ? TRON_ERR_UNSUPPORTED : (int)rc;
}
