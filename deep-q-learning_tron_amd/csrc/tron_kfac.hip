// tron_kfac.hip — the K-FAC input-patch extraction for the ACKTR path (reference: Net/kfac.py:28-38
// `_extract_patches`, whose own TODO at kfac.py:9-12 asks for a fused kernel).
//
// For a conv layer with input x [B][C][H][W], kernel kh x kw, padding p, stride s, the A-factor is
// E[a a^T] over all (sample, output position) pairs, a = the C*kh*kw input patch under that output.
// torch's F.unfold builds the patch matrix with ONE im2col launch PER SAMPLE (1.1 M launches and 31 %
// of the GPU time of one ACKTR iteration at 16 384 envs x 32x32, rocprofv3 round 1).  This kernel
// writes the GEMM-ready matrix P [B*OH*OW][C*kh*kw] (row = sample-major then output row-major, column
// = c*kh*kw + i*kw + j: F.unfold's channel order, transposed) in one launch per chunk of samples.
//
// One workgroup per (sample, output row): the kh input rows of every channel that this output row
// reads are staged in LDS once, zero-padded, then the OW patch rows go out as coalesced stores —
// consecutive lanes write consecutive columns of a row.  HBM-write bound: reads C*kh*W, writes
// OW*C*kh*kw floats.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tron_hip.h"

namespace {

__global__ void k_extract_patches(const float *__restrict__ x, int C, int H, int W, int kh, int kw, int pad, int stride,
                                  int OH, int OW, float *__restrict__ out)
{
    extern __shared__ float xs[];                 // [C][kh][Wp], Wp = W + 2*pad
    const int n = blockIdx.x / OH, oy = blockIdx.x - n * OH;
    const int Wp = W + 2 * pad;
    const int d = C * kh * kw;
    const float *xn = x + (size_t)n * C * H * W;
    const int rows = C * kh;
    for (int e = threadIdx.x; e < rows * Wp; e += blockDim.x) {
        const int r = e / Wp, xx = e - r * Wp;
        const int c = r / kh, i = r - c * kh;
        const int iy = oy * stride - pad + i, ix = xx - pad;
        xs[e] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? xn[((size_t)c * H + iy) * W + ix] : 0.0f;
    }
    __syncthreads();
    float *o = out + ((size_t)n * OH + oy) * (size_t)OW * d;
    for (int col = threadIdx.x; col < d; col += blockDim.x) {
        const int c = col / (kh * kw), r = col - c * kh * kw;
        const int i = r / kw, j = r - i * kw;
        const float *src = xs + (c * kh + i) * Wp + j;
        for (int ox = 0; ox < OW; ++ox) o[(size_t)ox * d + col] = src[ox * stride];
    }
}

}  // namespace

extern "C" int tron_extract_patches(const float *x, int64_t batch, int32_t channels, int32_t height, int32_t width,
                                    int32_t kh, int32_t kw, int32_t pad, int32_t stride, float *out, void *stream)
{
    if (!x || !out || batch < 0 || channels < 1 || height < 1 || width < 1 || kh < 1 || kw < 1 || pad < 0 || stride < 1)
        return TRON_ERR_BAD_ARG;
    const int OH = (height + 2 * pad - kh) / stride + 1, OW = (width + 2 * pad - kw) / stride + 1;
    if (OH < 1 || OW < 1) return TRON_ERR_BAD_ARG;
    if (batch == 0) return TRON_OK;
    const size_t smem = (size_t)channels * kh * (width + 2 * pad) * sizeof(float);
    if (smem > 64u * 1024u || batch * (int64_t)OH > 0x7FFFFFFF) return TRON_ERR_UNSUPPORTED;
    const int d = channels * kh * kw;
    int threads = (d + 63) / 64 * 64;
    if (threads > 1024) threads = 1024;
    hipLaunchKernelGGL(k_extract_patches, dim3((unsigned)(batch * OH)), dim3(threads), smem,
                       reinterpret_cast<hipStream_t>(stream), x, channels, height, width, kh, kw, pad, stride, OH, OW, out);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}
