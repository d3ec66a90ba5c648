// tron_conv_ws_pool.hip — conv6 of the gradient-free chain at 12x12 (DQNNet.py:48-52: mish(conv6(x) + res), then AvgPool2d(3, 2, 1))
// as ONE launch: the weight-stationary kernel of tron_conv_ws_kernel.hpp in its WS_POOL mode.  conv6's 64 x 12 x 12 output
// never exists in memory: the epilogue leaves it in an LDS image, the workgroup pools it there and stores the 64 x 6 x 6 pooled
// rows the head's conv7 GEMM reads (tron_dqn_head_fwd_pooled).  Per image 36.9 KB less written, 36.9 KB less read and one
// launch less than tron_conv3x3_ws_fwd + the head's k_pool_split12_px, with the same bits (the pooling adds the same hi / lo
// f16 values in the same order).  A translation unit of its own: co-compiled instantiations perturb each other's register
// allocation (tron_conv_ws_kernel.hpp).
#include "tron_conv_ws_kernel.hpp"

#ifdef TRON_WS_STAMPS
extern "C" int tron_conv_ws_pool_stamps(unsigned long long *host_dst)
{
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_ws_stamps), sizeof(g_ws_stamps)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int64_t tron_pooled12_bytes(int64_t batch)
{
    if (batch < 0 || batch > (1ll << 24)) return 0;
    return 2 * ((batch * 64 * 36 * 2 + 255) / 256 * 256);
}

namespace {
// one launch of k_conv_ws<Geo<12, 12, 64, 64, 1, 8, 1>, residual, no f32 planes, MODE>: `out` / `pre_px` as the mode reads them
template <int MODE>
int launch_pool(const void *in_px16, const void *wfrag, const float *bias, const void *res_px16, void *out, void *pre_px,
                int64_t batch, void *stream)
{
    using G = Geo<12, 12, 64, 64, 1, 8, 1>;
    constexpr size_t LDS_ALL = G::LDS_BYTES + (G::COUT / 4) * (G::SS * 16 + 16) + 1024 + 32;     // + the output image, its dump, the pass counters
    static_assert(LDS_ALL <= 160 * 1024, "LDS");
    auto kern = k_conv_ws<G, true, false, MODE>;
    static uint64_t prepared = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_NO_DEVICE; }
    if (!(prepared & (1ull << (dev & 63)))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_ALL) != hipSuccess)
            (void)hipGetLastError();
        prepared |= 1ull << (dev & 63);
    }
    int grid = device_cus();
    if (batch < grid) grid = (int)batch;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(G::THREADS), LDS_ALL, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const unsigned char *>(in_px16), reinterpret_cast<const f16x8 *>(wfrag), bias,
                       reinterpret_cast<const unsigned char *>(res_px16), reinterpret_cast<unsigned char *>(out), (float *)nullptr,
                       (float *)nullptr, (int)batch, 1, (int)batch, reinterpret_cast<unsigned char *>(pre_px),
                       WsBwd{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr});
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}
}  // namespace

// in_px16 / res_px16: PX16 images [batch][64][12][12]; wfrag: conv6's fragment image (tron_conv3x3_ws_split_weights);
// pooled: tron_pooled12_bytes(batch) bytes — [hi rows | lo rows], a row = 64 x 6 x 6 f16 in (octet, pooled pixel, channel) order.
extern "C" int tron_conv3x3_ws_fwd_pool12(const void *in_px16, const void *wfrag, const float *bias, const void *res_px16,
                                          void *pooled, int64_t batch, void *stream)
{
    if (!in_px16 || !wfrag || !bias || !res_px16 || !pooled || batch < 0) return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(in_px16) | reinterpret_cast<uintptr_t>(wfrag) | reinterpret_cast<uintptr_t>(res_px16) |
         reinterpret_cast<uintptr_t>(pooled) | reinterpret_cast<uintptr_t>(bias)) & 15u)
        return TRON_ERR_BAD_ARG;
    if (batch == 0) return TRON_OK;
    if (batch > (1ll << 24)) return TRON_ERR_UNSUPPORTED;
    unsigned char *hi = reinterpret_cast<unsigned char *>(pooled);
    return launch_pool<WS_POOL>(in_px16, wfrag, bias, res_px16, hi, hi + (batch * 64 * 36 * 2 + 255) / 256 * 256, batch, stream);
}

// The learner's forward (DDQN.py:127 on DQNNet.py:48-52): the same launch also keeps conv6's pre-activation as the PX16 image
// pre_px16 [batch][64][12][12] (tron_conv3x3_ws_train_fwd's), and the pooled output leaves as f32 planes pooled_f32 [batch][64][6][6]
// — what tron_pool12_px16 makes of conv6's PX16 output, bit for bit; that output itself (which only the pooling reads) is not stored.
extern "C" int tron_conv3x3_ws_train_fwd_pool12(const void *in_px16, const void *wfrag, const float *bias, const void *res_px16,
                                                void *pre_px16, float *pooled_f32, int64_t batch, void *stream)
{
    if (!in_px16 || !wfrag || !bias || !res_px16 || !pre_px16 || !pooled_f32 || batch < 0) return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(in_px16) | reinterpret_cast<uintptr_t>(wfrag) | reinterpret_cast<uintptr_t>(res_px16) |
         reinterpret_cast<uintptr_t>(pre_px16) | reinterpret_cast<uintptr_t>(pooled_f32) | reinterpret_cast<uintptr_t>(bias)) & 15u)
        return TRON_ERR_BAD_ARG;
    if (batch == 0) return TRON_OK;
    if (batch > (1ll << 24)) return TRON_ERR_UNSUPPORTED;
    return launch_pool<WS_POOL_TRAIN>(in_px16, wfrag, bias, res_px16, pooled_f32, pre_px16, batch, stream);
}
