// Microbenchmark: cycles per v_mfma_f32_16x16x32_f16 for one wave per SIMD, with N independent accumulators, with
// and without VALU fillers (v_perm_b32) in each gap.   hipcc --offload-arch=gfx950 -O3 mfma_rate.cpp -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int FILL>
__global__ __launch_bounds__(256, 1) void k(const f16x8 *in, f32x4 *out, unsigned long long *cyc, int iters)
{
    f16x8 a = in[threadIdx.x], b = in[threadIdx.x + 256];
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){(float)i, 0, 0, 0};
    unsigned x = threadIdx.x, y = threadIdx.x * 3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));   // (the intrinsic on equal inputs gets CSE'd)
#pragma unroll
            for (int f = 0; f < FILL; ++f) { x = __builtin_amdgcn_perm(x, y, 0x05040302u + f); y += x; }
        }
    }
    asm volatile("s_nop 7\n s_nop 7\n s_nop 7" ::: "memory");            // let the last MFMAs retire before anything reads them
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = acc[0];
    for (int i = 1; i < NACC; ++i) s += acc[i];
    s[0] += (float)(x + y);
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int FILL>
void run(const f16x8 *in, f32x4 *out, unsigned long long *cyc, const char *name)
{
    const int iters = 2000;
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<NACC, FILL>), dim3(256), dim3(256), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-40s %6.2f cycles per MFMA (median over 256 workgroups)\n", name, (double)h[128] / ((double)iters * NACC));
}

int main()
{
    f16x8 *in; f32x4 *out; unsigned long long *cyc;
    hipMalloc(&in, 512 * 16); hipMemset(in, 0, 512 * 16);
    hipMalloc(&out, 256 * 256 * 16); hipMalloc(&cyc, 256 * 8);
    run<1, 0>(in, out, cyc, "1 accumulator, no filler");
    run<4, 0>(in, out, cyc, "4 accumulators, no filler");
    run<16, 0>(in, out, cyc, "16 accumulators, no filler");
    run<36, 0>(in, out, cyc, "36 accumulators (AGPR pressure), no filler");
    run<16, 1>(in, out, cyc, "16 accumulators, 2 VALU per gap");
    run<16, 2>(in, out, cyc, "16 accumulators, 4 VALU per gap");
    run<16, 3>(in, out, cyc, "16 accumulators, 6 VALU per gap");
    return 0;
}
