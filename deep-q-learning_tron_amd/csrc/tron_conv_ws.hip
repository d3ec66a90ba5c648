// tron_conv_ws.hip — the CNN's 3x3 convolutions (Net/DQNNet.py:10-17,33-50: conv + bias + residual + mish) for
// gradient-free forwards, WEIGHT-STATIONARY on the split-f16 matrix cores (the arithmetic of tron_conv_f16.hip:
// v = hi + lo 2^-11, three v_mfma_f32_16x16x32_f16 per k-slab, f32 accumulation).
//
// What bounded tron_conv_f16.hip (docs/DESIGN_history_r01_r03.md 4b, wave stamps): every 16-channel chunk re-pulled 41 KB of split weights
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
#include "tron_conv_ws_kernel.hpp"

#ifdef TRON_WS_STAMPS
extern "C" int tron_conv_ws_stamps(unsigned long long *host_dst)
{
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_ws_stamps), sizeof(g_ws_stamps)) == hipSuccess ? 0 : -1;
}
#endif

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
    WsJobs jobs{};                                                      // (rot = 0: forward fragments)
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
#ifndef TRON_WS_WAVES        // 4: one wave per SIMD, three tiles per step; 8: two waves per SIMD, one tile per step
#define TRON_WS_WAVES 8
#endif
#ifndef TRON_WS_TPS
#define TRON_WS_TPS (TRON_WS_WAVES == 4 ? 3 : 1)
#endif
#ifndef TRON_WS_WAVES32      // waves of the 32-input-channel layers: 72 weight registers, so three waves per SIMD fit, and 12 waves
#define TRON_WS_WAVES32 12   // deal a 12x12 item's 9 (18) tiles evenly; 1-4 % faster than 8 (profiles/r03_ws_layer_bench_*.txt)
#endif
    if (side == 34) return ws_fwd_side34(in_px16, wfrag, bias, res_px16, out_px16, out_f32, pre_f32, batch, cin, cout, apply_mish, st);
#define TRON_WS_CASE(S_, R_, CI_, CO_, IPI_, WAVES_)                                                                  \
    if (side == S_ && cin == CI_ && cout == CO_)                                                                      \
        return launch_ws<Geo<S_, R_, CI_, CO_, IPI_, WAVES_, TRON_WS_TPS>>(in_px16, wfrag, bias, res_px16, out_px16, out_f32, pre_f32, batch, apply_mish, st);
    TRON_WS_CASE(12, 12, 32, 32, 2, TRON_WS_WAVES32)
    TRON_WS_CASE(12, 12, 32, 64, 1, TRON_WS_WAVES32)
    TRON_WS_CASE(12, 12, 64, 64, 1, TRON_WS_WAVES)
    TRON_WS_CASE(26, 13, 32, 32, 1, TRON_WS_WAVES32)
    TRON_WS_CASE(26, 13, 32, 64, 1, TRON_WS_WAVES32)
    TRON_WS_CASE(26, 7, 64, 64, 1, TRON_WS_WAVES)
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
    const int64_t total = batch * side * side;
    const unsigned grid = (unsigned)((total + 255) / 256 < 256 * 8 ? (total + 255) / 256 : 256 * 8);
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
